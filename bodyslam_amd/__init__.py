"""bodyslam_amd: MI355X-native (gfx950) implementation of BodySLAM's per-frame depth + pose +
back-projection hot path behind the reference's own Python call surface.  See DESIGN.md."""
__version__ = "0.1.0"
