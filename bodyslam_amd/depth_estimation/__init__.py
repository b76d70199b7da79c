from .interface import DepthEstimator  # noqa: F401
