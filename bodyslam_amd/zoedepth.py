"""MDEM on MI355X: ZoeDepth (ZoeD_NK: BEiT-L/16 backbone, DPT neck, relative head, metric-bins head)
as a prebuilt plan of HIP kernel launches through the C ABI (bodyslam_amd/_lib.py).

What the reference does for this path: ``torch.hub.load("isl-org/ZoeDepth", "ZoeD_NK")`` then
``model.infer_pil(image, output_type="pil")`` per frame (BodySLAM_Refactored/src/depth_estimation/
interface.py:46,61; BodySLAM_not_refactored/MDEM/mdem_interface.py:37-44,68).  The network itself is
un-vendored; layer-by-layer citations below are to the installed weight-compatible restatement
("HF" = transformers 5.15.0 models/zoedepth/modeling_zoedepth.py, models/beit/modeling_beit.py,
models/zoedepth/image_processing_pil_zoedepth.py).  Parameter names are HF state-dict names.

Data layout in HBM (B frames, flip-aug doubles the image batch: NB = 2B):
  residual stream      fp32 [NB*S, hidden]            S = 1 + hp*wp tokens (cls first)
  GEMM operands        fp16/bf16, K-major weights [N, K]; conv weights [O][I/64][kh][kw][64]
  Q, K / V^T           [NB, heads, Sp, 64] / [NB, heads, 64, Sp], Sp = S rounded up to 64, zero padded
  rel-pos bias         fp32 [layers][heads, (2hp-1)(2wp-1)+3] table per window (wp = 32: gathered in LDS by bs_attention_table;
                       other windows: materialised [heads, Sp, Sp] with -1e30 in padded key columns)
  conv activations     NHWC 16-bit
  bins / attractors    fp32 NHWC, both heads side by side ([nyu 64 | kitti 64], [16 | 16])
  precision="accurate" every GEMM / conv operand is a pair: activations [rows, 2C] = (hi16 | hi8 | lo8) (or (hi | lo) 16-bit
                       pairs where a K is not a multiple of 128), weights [W_hi16 | W_lo8 | W_hi8]; one launch evaluates
                       A_hi W_hi + A_hi W_lo + A_lo W_hi, the corrections on the FP8 MFMA (Numerics)
PyTorch is used for allocation, views and one-off weight re-layout only.
"""
from __future__ import annotations

import math
import os
import warnings
from dataclasses import dataclass
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import _lib as L

LOG2E = 1.4426950408889634


@dataclass
class ZoeConfig:
    """The part of HF ZoeDepthConfig/BeitConfig the forward depends on (defaults: ZoeD_NK)."""
    hidden: int = 1024
    layers: int = 24
    heads: int = 16
    intermediate: int = 4096
    taps: Tuple[int, ...] = (6, 12, 18, 24)
    image_size: int = 384
    patch: int = 16
    ln_eps: float = 1e-12
    neck_hidden: Tuple[int, ...] = (256, 512, 1024, 1024)
    fusion: int = 256
    rel_features: int = 32
    bottleneck: int = 256
    bin_dim: int = 128
    n_attractors: int = 16
    n_bins: int = 64
    min_temp: float = 0.0212
    max_temp: float = 50.0
    pt_layers: int = 4
    pt_hidden: int = 128
    pt_inter: int = 1024
    pt_heads: int = 4
    head_names: Tuple[str, ...] = ("nyu", "kitti")     # one name = a single-head model (ZoeD_N: ("nyu",), ZoeD_K: ("kitti",))
    add_projection: bool = True    # HF ZoeDepthConfig.add_projection; the engine follows the WEIGHTS (relative_head.projection.* present or not)
    level_attractors: Tuple[int, ...] = (16, 8, 4, 1)  # single-head models only (HF num_attractors); the NK head uses n_attractors

    @property
    def single_head(self) -> bool:
        return len(self.head_names) == 1

    @property
    def seed_mlp(self) -> int:      # HF defaults in the single head (modeling_zoedepth.py:1132-1142) vs bin_dim // 2 in the NK head
        return 256 if self.single_head else self.bin_dim // 2

    @property
    def proj_mlp(self) -> int:
        return 128 if self.single_head else self.bin_dim // 2

    @property
    def clb_hidden(self) -> int:    # (in + condition) // bottleneck_factor: factor 2 (single, with the relative depth) or 4 (NK)
        return (self.rel_features + 1 + self.bin_dim) // 2 if self.single_head else (self.rel_features + self.bin_dim) // 4

    def attractors_at(self, level: int) -> int:
        return self.level_attractors[level] if self.single_head else self.n_attractors


# Which correction products accurate mode evaluates per backbone GEMM class (tools/probes/precision_classes.py,
# tools/probes/weight_mean_correction.py, DESIGN.md Numerics).
#   "full"  A_hi W_hi + A_hi8 W_lo8 + A_lo8 W_hi8 on every row (2 pass-equivalents): depth L1 vs the fp32 oracle 1.3e-5 m
#   "wcls"  the activation-rounding correction on the cls-token tile only (its error is the only coherent one): 1.5 passes,
#           4.3e-5 / 2.0e-5 m on two weight seeds
#   "wmean" patch tiles run ONE 16-bit pass; the weight-rounding error A dW^T, a coherent offset, is replaced by its
#           token-independent part 1 (mean_tokens(A) dW^T) -- a per-image bias from a column mean and a tiny GEMM; the cls tile
#           keeps both corrections: ~1.0 pass, 4.8e-5 / 2.6e-5 m on those two seeds, 1.7e-5 ... 6.3e-5 m over eight
#           (profiles/r02_accurate_seeds.txt: "wcls" 1.6e-5 ... 5.6e-5, "full" <= 2.1e-5 on the same seeds)
# The neck keeps both products (weight correction only: 0.9-2.0e-4).
# "auto" (the default): the engine measures, on the device and with the weights it was given, which of these each class tolerates
# (ZoeDepthEngine.calibrate, run before the first plan is built): the random-weight studies above say nothing about a trained
# checkpoint's outlier channels and layer-scale, so no fixed choice is trusted.
# "pairs" (not a calibration candidate): the operands as (hi | lo) 16-bit pairs and the product as three 16-bit passes, ~22 significant
# bits per operand -- the REFERENCE precision (precision="reference"), against which calibrate() takes its absolute error.
BACKBONE_CLASSES = ("qkv", "o", "fc1", "fc2")
ACCURATE_CLASS_MODES = "auto"
AUTO_CANDIDATES = ("wmean", "wcls", "full")       # cheapest first
#   "wstat" (round 6; a candidate in front of "wmean" only with BS_AUTO_WSTAT=1) "wmean" with the token-independent part of the weight-rounding error
#           taken from the CALIBRATION frames' channel means instead of each image's own: a static fp32 bias row per product (backbone_bias_corr), no
#           bs_col_mean / bs_rank1_bias launches at run time (192 per forward batch, 2.2 ms per 128 network inputs).  Measured: alone a class costs what
#           "wmean" costs it (1.2-1.3e-5 m), but the four together read 3.46e-5 m where "wmean" reads 2.81e-5 -- the image-dependent remainder
#           Sum_c (mean_t a_tc - E[a_c]) dw_c is coherent over an image -- and the neck loses products to it: +0.2 ... +1.1 % frames/s on three weight seeds,
#           -2 % on the fourth (profiles/r06_calibration_experiments.txt (8)).  Not a default: the gain is inside the box-to-box spread and the static means
#           stand on the calibration frames' statistics.
AUTO_TOL_CLASS_M = 4.0e-5                        # depth L1 against the best mode's result that ONE class may cost
AUTO_TOL_TOTAL_M = 6.0e-5                        # ... and the chosen combination as a whole
AUTO_TOL_ABS_M = 8.0e-5                          # ... and the chosen combination against the 3-pass REFERENCE engine on the device (the
                                                 # north star's tolerance is 1e-4 m; the margin covers frames other than the calibration frame)
TOLERANCE_M = 1.0e-4                             # BASELINE.json north_star: depth L1 vs the reference; calibrate() warns above it
# Attention operands: "single" = Q, K, V^T and the probabilities as single 16-bit values (bs_attention_table); "corr" = each with its
# rounding residual as a second 16-bit value, three MFMA passes per product (bs_attention_table_corr).  Seeded Gaussian weights lose
# 2.6e-6 m to "single"; weights with outlier channels behind the LayerNorms (trained BEiT checkpoints) 2.9e-4 m
# (tools/probes/outlier_rounding_study.py) -- so the choice is calibrated per weight set like the GEMM classes ("auto").
ACCURATE_ATTN_MODE = "auto"
# neck: "full", or the list of weight-key prefixes that keep both products (the rest: weight-rounding correction only).
NECK_RELHEAD_WONLY = "ro,ra,nc,fu,pj,mh"         # everything but the relative head keeps both (mh: the bins head's bottleneck conv)
# What a class's cheap mode saves, in executed GFLOP per network input (16-bit-pass equivalents; derived from the shapes in
# ZoeDepthEngine._saving_gflop, round 6 -- rounds 2-5 carried millisecond figures measured once at B = 64).  When the chosen combination
# misses the total tolerance, the class that pays the most depth error per unit saved goes back up first.
AUTO_ATTN_RATE_FACTOR = 1.5                      # the attention kernels run at ~2/3 of the GEMMs' rate: a pass saved there is worth more time
# The relative head (rh.projection, rh.conv1: 40 % of the conv stack's time) with the weight-rounding correction only is +1.1 % frames/s for
# +0.7-1.8e-5 m (profiles/r03_neck_relhead_wonly.txt): a candidate since round 4, when calibrate() got an ABSOLUTE reference.  Without that
# reference (no source weights) the neck stays "full".
AUTO_NECK_CANDIDATES = (NECK_RELHEAD_WONLY, "full")
# Round 5: the neck is calibrated PER SITE (VERDICT r4 #1c).  With the absolute reference available, calibrate() measures what each of the
# neck's / heads' FP8-format products costs when it alone drops the activation-rounding correction, orders the sites by depth error per
# FLOP saved and keeps the longest prefix of that order whose combination stays under AUTO_TOL_NECK_ABS_M against the reference
# (tools/probes/neck_site_study.py: on the bench weights 11 ... 19 sites, not only the relative head, fit that budget).  The choice is
# the neck mode "wonly:<site>,<site>,..." (ZoeDepthEngine.neck_site_wonly).  Sites below AUTO_NECK_SITE_MIN_SHARE of the neck's FLOPs are
# not worth a calibration forward EACH; since round 6 they are candidates as GROUPS (AUTO_NECK_SMALL_GROUPS: one forward per group and
# stage) -- in round 5 none of them was ever calibrated and all ran both corrections, 2-3x their algorithmic work (VERDICT r5 #1 ii).
AUTO_TOL_NECK_ABS_M = 5.0e-5
# (round 6: a calibration forward over four frames costs ~0.1 s, so every product above 0.1 % of the neck's FLOPs is a candidate of its own --
# the first grouping tried, "all reassemble / projection GEMMs", hid one sensitive member behind 6.4e-5 m for the whole group)
AUTO_NECK_SITE_MIN_SHARE = 0.001
AUTO_NECK_SMALL_GROUPS = {"tiny": ("ro", "ra", "nc", "fu", "pj", "rh")}      # by weight-key prefix, first match; the bins head's bottleneck conv (mh.) stays out
# Round 5, second stage: the large weight-only sites may drop the weight-rounding correction TOO -- one 16-bit pass, `f8_skip_from = -1`, the
# form "wonly:...;plain:<site>,..." (ZoeDepthEngine.neck_site_plain).  tools/probes/neck_plain_study.py: a site alone moves the map by 2-3e-5 m,
# but twelve of them together leave the distance to the reference where it was (4.5 -> 4.9e-5 m) -- weight rounding in the neck is incoherent
# noise, like its activation rounding -- while two small sites (nc2, ra3.down) alone cost 3-8e-5.  So the stage walks the sites by FLOPs, largest
# first, and keeps a site when the combination stays under AUTO_TOL_NECK_PLAIN_ABS_M against the reference.  BS_NECK_PLAIN=0 switches the stage off.
# Round 6 (advisor): every stage is judged on the WORST of AUTO_CAL_FRAMES calibration frames (round 5: one frame, the
# plain stage on the mean of two), a site is kept only when it ALSO stays within `tol_total` of the best mode, and the final combination is
# validated on AUTO_HOLDOUT_FRAMES frames that no decision has seen: above AUTO_TOL_HOLDOUT_M there, the latest relaxations are withdrawn.
# And the one-pass sites carry a STATIC bias correction: the token-independent part dW E[a] of their weight-rounding error, from the
# calibration frames' channel means (site_bias_corr; the data-free-quantisation bias correction) -- no run-time cost.
AUTO_NECK_PLAIN_MIN_SHARE = 0.001
AUTO_TOL_NECK_PLAIN_ABS_M = 6.0e-5
# The two neck stages are gated on what they ADD, not only on where they end (round-5 advisor): a weight set whose backbone choice already sits
# at or above AUTO_TOL_NECK_ABS_M (the outlier-channel weights: 5.1e-5 m with every class "full" -- the floor of e4m3 correction planes on
# those weights) got no neck relaxation at all in round 5, although one-pass products with the static bias correction cost it a few 1e-6.
# Stage 1 may spend up to max(AUTO_TOL_NECK_ABS_M, backbone choice + AUTO_NECK_WONLY_INCREMENT_M), stage 2 up to max(AUTO_TOL_NECK_PLAIN_ABS_M,
# stage 1's result + AUTO_NECK_PLAIN_INCREMENT_M), both capped at AUTO_TOL_NECK_CAP_M; the hold-out check (AUTO_TOL_HOLDOUT_M) stands behind them.
AUTO_NECK_WONLY_INCREMENT_M = 0.5e-5
AUTO_NECK_PLAIN_INCREMENT_M = 0.5e-5
AUTO_TOL_NECK_CAP_M = 6.5e-5
AUTO_CAL_FRAMES = 4
AUTO_HOLDOUT_FRAMES = 4
AUTO_TOL_HOLDOUT_M = 7.0e-5
AUTO_CAL_SEEDS = (11, 12, 13, 14, 15, 16, 17, 18)        # synthetic.make_sequence(1, H, W, seed): the calibration frames ...
AUTO_HOLDOUT_SEEDS = (21, 22, 23, 24, 25, 26, 27, 28)    # ... and the held-out ones
ACCURATE_NECK_MODE = "full"
# (Round 3 also had neck_corr="f4": e2m1 correction planes with E8M0 block scales on the FP4 MFMA -- +1.4 % frames/s for 1.5x the depth error,
# profiles/r03_fp4_corrections.txt.  It never paid and was removed in round 4; the correction products run on the block-scaled FP8 MFMA.)

_CALIBRATION_CACHE: Dict[Tuple, dict] = {}        # process-wide: (weights fingerprint, geometry, tolerances, ...) -> calibration report
# environment switches that change the arithmetic a calibration measures (A / B runs): part of the cache key (round-5 advisor)
_ARITHMETIC_SWITCHES = ("BS_PJ_LOWRES", "BS_UPCONV_FUSED", "BS_CLB_COMPOSED", "BS_RELU_OUT", "BS_NECK_PLAIN", "BS_MLP2", "BS_NECK_BIAS_CORR",
                        "BS_PROJECTOR_LEVEL", "BS_AUTO_WSTAT")

ZOED_NK = ZoeConfig()
ZOED_N = ZoeConfig(head_names=("nyu",))
ZOED_K = ZoeConfig(head_names=("kitti",))


def pad_sizes(h: int, w: int) -> Tuple[int, int]:
    """HF image_processing_pil_zoedepth.py:181-201."""
    return int(np.sqrt(h / 2) * 3), int(np.sqrt(w / 2) * 3)


def net_size(h: int, w: int, out_hw=(384, 512), multiple: int = 32) -> Tuple[int, int]:
    """Network input size for an (h, w) frame: pad, keep_aspect_ratio resize, multiple of 32
    (HF image_processing_pil_zoedepth.py:72-108; upstream PrepForMidas ensure_multiple_of=32)."""
    ph, pw = pad_sizes(h, w)
    hp, wp = h + 2 * ph, w + 2 * pw
    sh, sw = out_hw[0] / hp, out_hw[1] / wp
    if abs(1 - sw) < abs(1 - sh):
        sh = sw
    else:
        sw = sh
    return int(np.round(sh * hp / multiple) * multiple), int(np.round(sw * wp / multiple) * multiple)


def _relative_position_index(wh: int, ww: int) -> torch.Tensor:
    """HF modeling_beit.py:194-218."""
    nrd = (2 * wh - 1) * (2 * ww - 1) + 3
    coords = torch.stack(torch.meshgrid(torch.arange(wh), torch.arange(ww), indexing="ij")).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += wh - 1
    rel[:, :, 1] += ww - 1
    rel[:, :, 0] *= 2 * ww - 1
    idx = torch.zeros((wh * ww + 1,) * 2, dtype=rel.dtype)
    idx[1:, 1:] = rel.sum(-1)
    idx[0, 0:] = nrd - 3
    idx[0:, 0] = nrd - 2
    idx[0, 0] = nrd - 1
    return idx


class ZoeDepthEngine:
    """Weights resident on the GPU + per-shape launch plans.

    ``weights``: HF-named fp32 CPU/GPU tensors (an ``Intel/zoedepth-nyu-kitti`` state dict, or the
    synthetic set used by the tests / bench).  ``dtype``: GEMM operand type (fp16 or bf16); all
    accumulation, LayerNorm, softmax, residual stream and the bins head are fp32.
    """

    def __init__(self, weights: Dict[str, torch.Tensor], cfg: Optional[ZoeConfig] = None, dtype=torch.float16,
                 device: int = 0, target_hw: Tuple[int, int] = (384, 512), precision: str = "fast",
                 class_modes: Optional[Dict[str, str]] = None, neck_mode: Optional[str] = None, attn_mode: Optional[str] = None):
        """precision: "fast" = one MFMA pass per product (16-bit operands, fp32 accumulate);
        "accurate" = split-precision products (Numerics): every GEMM / conv operand of the backbone, the DPT neck,
        the relative head and the projector path of the bins head is a (hi, lo) pair of 16-bit values; one launch evaluates
        A_hi W_hi + A_lo W_hi + A_hi W_lo (3 MFMA passes).  Attention (Q, K, V, P) and the small bins-head MLPs stay single.
        class_modes: per backbone GEMM class ("qkv", "o", "fc1", "fc2") which correction products accurate mode evaluates --
        "full" A_hi W_hi + A_hi W_lo + A_lo W_hi (2 pass-equivalents); "wcls" the same on the cls-token rows and A_hi W_hi +
        A_hi W_lo (the weight-rounding correction only, 1.5 pass-equivalents) on the patch rows -- activation rounding is per-row
        noise that stays incoherent in the depth map except on the cls rows, whose error shifts the whole map; probes: "w" (no
        activation correction anywhere), "a" A_hi W_hi + A_lo W_hi, "single" one 16-bit pass.  Default ACCURATE_CLASS_MODES
        (DESIGN.md Numerics: which rounding errors the depth map sees)."""
        L.init(device)
        assert precision in ("fast", "accurate", "reference")
        if precision == "reference":
            # every GEMM / conv as three 16-bit passes on (hi | lo) pairs, attention on split operands: ~22 significant bits per operand
            # everywhere (fp32 accumulation throughout) -- what calibrate() measures the production modes against, on the device
            class_modes, neck_mode, attn_mode = "pairs", "pairs", "corr"
        self.precision = precision
        self.acc = precision != "fast"
        self._sd = weights if self.acc else None     # (kept for calibrate(): the reference engine is built from the same weights)
        self.target_hw = target_hw       # the processor's resize target (384x512 for every released checkpoint)
        self.cfg = cfg or ZoeConfig()
        assert dtype in (torch.float16, torch.bfloat16)
        self.dtype = dtype
        self.dev = torch.device("cuda", device)
        self.w: Dict[str, torch.Tensor] = {}
        self._bias_cache: Dict[Tuple[int, int], list] = {}
        self._plans: Dict[Tuple, "_ZoePlan"] = {}
        self._raw_tables = []
        self.f8s: Dict[str, Tuple[int, int]] = {}
        cm = class_modes if class_modes is not None else (os.environ.get("BS_ACCURATE_CLASS_MODE") or ACCURATE_CLASS_MODES)
        if isinstance(cm, str) and cm != "auto":          # one mode for all four classes (diagnostics, A/B runs of bench.py)
            cm = {k: cm for k in BACKBONE_CLASSES}
        # "auto": every class starts at "full" and calibrate() -- run before the first plan is built -- lowers what the weights allow
        self.auto_classes = self.acc and isinstance(cm, str)
        self.auto_modes = self.auto_classes
        self.calibration: Optional[dict] = None
        self.class_modes = {k: "full" for k in BACKBONE_CLASSES}
        if isinstance(cm, dict):
            self.class_modes.update(cm)
        assert all(v in ("full", "w", "wcls", "wmean", "wstat", "a", "single", "pairs") for v in self.class_modes.values()), self.class_modes
        am_ = attn_mode or os.environ.get("BS_ATTN_MODE") or (ACCURATE_ATTN_MODE if self.acc else "single")
        assert am_ in ("auto", "single", "corr"), am_
        assert self.acc or am_ != "corr", "attn_mode='corr' belongs to precision='accurate'"
        # "auto": starts at "corr" (the best mode); calibrate() lowers it when the weights allow.  (A geometry the split-precision
        # kernel is not built for -- network widths other than 512, odd hp -- runs "single" whatever this says: _ZoePlan.)
        self.auto_attn = am_ == "auto"
        self.attn_mode = "corr" if am_ == "auto" else am_
        self.auto_modes = self.auto_modes or (self.acc and self.auto_attn)
        self.single_keys = set()
        self.wmode: Dict[str, str] = {}
        # static bias correction of the neck's one-pass products (calibrate(), second stage): dw_sum[key] = (W - round16(W)) summed over
        # the filter taps, fp32 [N, Cin], kept from ingestion; site_bias_corr[key] = dw_sum[key] @ E[a] with the calibration frames' channel
        # means -- added to the product's bias while it runs one pass (the token-independent part of its weight-rounding error)
        self.dw_sum: Dict[str, torch.Tensor] = {}
        self.backbone_bias_corr: Dict[str, torch.Tensor] = {}     # "wstat": weight key -> fp32 [1, N] = dW E[a] over the calibration frames' patch rows
        self.site_bias_corr: Dict[str, torch.Tensor] = {}
        self._bias_corr_cache: Dict[str, torch.Tensor] = {}
        # DPT neck / heads (no cls rows there): "full" = both correction products, "w" = the weight-rounding correction only
        self.neck_mode = neck_mode or os.environ.get("BS_NECK_MODE") or ACCURATE_NECK_MODE
        # (probes: a comma-separated list of weight-key prefixes that KEEP both products, e.g. "ro,ra,nc": everything else "w")
        # the attractor levels: 2 = add + resize + MLP in one launch (bs_mlp2_add), 1 = bs_add_resized + bs_mlp2, 0 = three launches (A / B)
        self.fuse_mlp = int(os.environ.get("BS_MLP2", "2"))
        c_ = self.cfg
        # the neck switches to the (hi16 | hi8 | lo8) operand format as a whole: every K / Cin on it must be whole 128-byte FP8 stages
        self.neck_f8 = self.acc and self.neck_mode != "pairs" and all(v % 128 == 0 for v in (c_.hidden, c_.fusion, c_.fusion // 2, *c_.neck_hidden))
        with torch.no_grad():
            self._ingest(weights)

    # ------------------------------------------------------------------------------------------
    # weight ingestion (one-off re-layout; names on the left are this engine's, on the right HF's)
    # ------------------------------------------------------------------------------------------
    def _h(self, t: torch.Tensor) -> torch.Tensor:
        return t.to(self.dev, dtype=self.dtype).contiguous()

    def _f(self, t: torch.Tensor) -> torch.Tensor:
        return t.to(self.dev, dtype=torch.float32).contiguous()

    def _conv_w(self, t: torch.Tensor) -> torch.Tensor:
        """[O, I, kh, kw] -> [O][I/64][kh][kw][64] (the K order of bs_gemm's conv mode: chunk, tap, channel)."""
        return self._h(L.conv_weight(t.permute(0, 2, 3, 1)))

    def _split(self, t: torch.Tensor):
        hi = t.to(self.dtype)
        lo = (t - hi.float()).to(self.dtype)
        return hi, lo

    def _w8(self, key: str, t: torch.Tensor) -> torch.Tensor:
        """backbone GEMM weight [N, K]; accurate: rows of [W_hi16 | W_lo8 | W_hi8] bytes (bs_gemm's FP8 correction segment);
        the two plane scales go to self.f8s[key]."""
        if not self.acc:
            return self._h(t)
        mode = self.class_modes.get(key.split(".")[-2], "full") if key[0] == "l" else "full"      # "l7.qkv.w" -> class "qkv"
        if mode == "single":
            self.single_keys.add(key)
            return self._h(t)
        if mode == "pairs":              # reference precision: three 16-bit passes on (hi | lo) pairs
            return self._wn(t)
        if t.shape[1] % 128 != 0:        # the FP8 segment walks whole 128-byte stages per plane: fall back to three 16-bit passes
            return self._wn(t)
        if mode == "w" and t.shape[1] % 256 != 0:
            mode = "full"
        w8, sb = L.f8_weight(t, self.dtype, planes={"full": "both", "wcls": "both", "wmean": "both", "wstat": "both", "w": "lo", "a": "hi_only"}[mode], device=self.dev)
        self.f8s[key] = sb
        self.wmode[key] = mode
        if key[0] == "l" and mode in ("full", "wcls", "wmean", "wstat"):
            # "full" / "wcls" / "wmean" read the same packed rows; "wmean" also needs dW = W - round16(W) as bf16 (fp32's exponent
            # range), the W operand of the rank-1 correction GEMM: kept for all three, so the mode is a PLAN-time choice
            # (set_class_modes / calibrate) and not a re-ingestion
            td = t.to(self.dev)
            self.w[key + ".lo"] = (td - td.to(self.dtype).float()).to(torch.bfloat16).contiguous()
        return w8.to(self.dev)

    def mode_of(self, wkey: str) -> Optional[str]:
        """correction mode of a weight at plan time: a backbone class packed for the switchable modes follows class_modes"""
        m = self.wmode.get(wkey)
        if m in ("full", "wcls", "wmean", "wstat") and wkey[0] == "l":
            return self.class_modes.get(wkey.split(".")[-2], m)
        return m

    def neck_site_wonly(self, wkey: str) -> bool:
        """does this neck / head product run the weight-rounding correction only (1.5 pass-equivalents) under the current neck mode?
        neck_mode: "full" (both products everywhere), "w" (none), "wonly:a,b,..." (exactly these sites weight-only: the per-site
        calibration's form), or a comma list of weight-key prefixes that KEEP both products (everything else weight-only)."""
        nm = self.neck_mode
        if wkey.endswith("w_cls") or nm in ("full", "pairs"):
            # the per-image readout bias (cls row x W_cls, added to every token of the image) always gets both products: an error there
            # is a coherent offset of the whole map
            return False
        if nm == "w":
            return True
        if nm.startswith("wonly:"):
            return any(wkey in part.split(":", 1)[1].split(",") for part in nm.split(";") if ":" in part)
        return not any(wkey.startswith(p_) for p_ in nm.split(","))

    def neck_site_plain(self, wkey: str) -> bool:
        """does this neck / head product run NO correction at all (one 16-bit pass)?  Only the per-site form names such sites:
        "wonly:a,b;plain:c,d" (a plain site also counts as weight-only for its producers: nobody reads the lo8 plane).  The fused
        up-convolution (rh.conv2.w) has no one-pass form on (hi16 | hi8 | lo8) rows and stays weight-only."""
        nm = self.neck_mode
        if not nm.startswith("wonly:") or wkey.endswith("w_cls") or wkey == "rh.conv2.w":
            return False
        return any(part.startswith("plain:") and wkey in part[6:].split(",") for part in nm.split(";"))

    def set_class_modes(self, modes: Dict[str, str], neck_mode: Optional[str] = None, attn_mode: Optional[str] = None) -> None:
        """switch the backbone classes between "full" / "wcls" / "wmean" (and the neck mode, the attention mode) without re-ingesting
        the weights; plans built so far are dropped"""
        for k, v in modes.items():
            assert k in BACKBONE_CLASSES and v in ("full", "wcls", "wmean", "wstat"), (k, v)
            assert self.class_modes[k] in ("full", "wcls", "wmean", "wstat"), f"class {k} was ingested as {self.class_modes[k]!r}: not switchable"
        self.class_modes.update(modes)
        if neck_mode is not None:
            assert (neck_mode == "pairs") == (self.neck_mode == "pairs"), "the neck's operand format is fixed at ingestion"
            self.neck_mode = neck_mode
        if attn_mode is not None:
            assert attn_mode in ("single", "corr") and (self.acc or attn_mode == "single")
            self.attn_mode = attn_mode
        self._plans.clear()

    def apply_calibration(self, report: dict) -> None:
        """take over a calibration made elsewhere (rank 0 of a sharded run broadcasts its report: every rank must run the same
        arithmetic, and ranks calibrating on their own could fall on different sides of a threshold)"""
        self.set_class_modes({k: v for k, v in report["class_modes"].items() if self.class_modes[k] in ("full", "wcls", "wmean", "wstat")},
                             report["neck_mode"], report.get("attn_mode"))
        self.backbone_bias_corr = {k: torch.tensor(v, dtype=torch.float32, device=self.dev).view(1, -1)
                                   for k, v in (report.get("backbone_bias_corr") or {}).items()}
        self.site_bias_corr = {k: torch.tensor(v, dtype=torch.float32, device=self.dev) for k, v in (report.get("site_bias_corr") or {}).items()}
        self._bias_corr_cache.clear()
        self.calibration = dict(report)

    def reference_depth(self, frames_u8: torch.Tensor) -> torch.Tensor:
        """depth map of `frames_u8` [B,H,W,3] from a REFERENCE-precision engine built from the same weights (three 16-bit passes on
        (hi | lo) pairs for every product, split-precision attention; ~22 significant bits per operand): the on-device stand-in for
        the fp32 oracle.  The engine is built, used and dropped here (2 GB, a few seconds)."""
        assert self._sd is not None, "the engine no longer holds its source weights (release_weights())"
        ref = ZoeDepthEngine(self._sd, self.cfg, dtype=self.dtype, device=self.dev.index or 0, target_hw=self.target_hw, precision="reference")
        B, H, W = int(frames_u8.shape[0]), int(frames_u8.shape[1]), int(frames_u8.shape[2])
        plan = _ZoePlan(ref, B, H, W, True)
        plan.frames.copy_(frames_u8)
        plan.run(None)
        d = plan.depth_m.clone()
        torch.cuda.synchronize(self.dev)
        del plan, ref
        torch.cuda.empty_cache()
        return d

    def release_weights(self) -> None:
        """drop the reference to the caller's weight dict (after this, calibrate() has no absolute reference)"""
        self._sd = None

    def _saving_gflop(self, S: int) -> Dict[str, float]:
        """what a cheap mode saves per network input, in executed GFLOP (16-bit-pass equivalents): "wmean" against "full" is one pass of the
        class's products; the single-operand attention against the split-precision one is two passes of 4 S^2 hidden (weighted by
        AUTO_ATTN_RATE_FACTOR).  step_up()'s yardstick (the error a choice costs per unit it saves)."""
        c = self.cfg
        per = lambda n, k: 2.0 * S * n * k * c.layers / 1e9
        return {"qkv": per(3 * c.hidden, c.hidden), "o": per(c.hidden, c.hidden), "fc1": per(c.intermediate, c.hidden),
                "fc2": per(c.hidden, c.intermediate), "attn": 2.0 * AUTO_ATTN_RATE_FACTOR * 4.0 * S * S * c.hidden * c.layers / 1e9}

    def calibrate(self, H: int = 480, W: int = 640, frames_u8: Optional[torch.Tensor] = None, tol_class: float = AUTO_TOL_CLASS_M,
                  tol_total: float = AUTO_TOL_TOTAL_M, neck_candidates: Optional[Sequence[str]] = None, tol_abs: float = AUTO_TOL_ABS_M,
                  reference: bool = True, holdout_u8: Optional[torch.Tensor] = None) -> dict:
        """Choose, with THESE weights on THIS device, the cheapest correction mode per backbone GEMM class, for the attention operands
        and for the neck that keeps the depth maps of the calibration frames within `tol_class` metres (L1, the WORST frame) of the BEST
        mode's result (all classes "full", attention "corr"), check the combination against `tol_total` and step the most expensive
        offender back up until it holds.  Then the ABSOLUTE check: the chosen combination against a reference-precision engine built from
        the same weights (reference_depth: three 16-bit passes everywhere) -- if that exceeds `tol_abs` the stepping continues, and if even
        the best mode misses the north star's 1e-4 m the report carries a "warning" (bench.py prints it, DepthEstimator warns).  With the
        reference at hand the neck is then calibrated per product in two stages (AUTO_TOL_NECK_ABS_M, AUTO_TOL_NECK_PLAIN_ABS_M) and the
        result validated on held-out frames (AUTO_TOL_HOLDOUT_M).
        `frames_u8` [N,H,W,3]: the caller's own calibration frames -- ALL of them are used, every decision is judged on the worst one
        (default: AUTO_CAL_FRAMES synthetic frames); `holdout_u8`: frames no decision may see (default: AUTO_HOLDOUT_FRAMES synthetic ones).
        One forward over the calibration frames per candidate (about fifty plans, each dropped after use).  The report is kept in
        ``self.calibration``.  Only what was left on "auto" is calibrated; classes / attention given a fixed mode keep it."""
        assert self.acc, "calibrate() is for precision='accurate'"
        import time as _time
        t_start = _time.perf_counter()
        # the same weights, geometry, tolerances and arithmetic switches give the same choice (every kernel is deterministic): a process that
        # builds several engines from one weight set (the bench's legs, a test module) calibrates once
        ckey = None
        if frames_u8 is None and holdout_u8 is None and self._sd is not None and reference and neck_candidates is None:
            ckey = (self._weights_fingerprint(), repr(self.cfg), str(self.dtype), H, W, tuple(self.target_hw), tol_class, tol_total, tol_abs,
                    self.auto_classes, self.auto_attn, tuple(sorted(self.class_modes.items())), self.attn_mode, self.neck_mode,
                    self.fuse_mlp, self.add_projection, self.neck_f8,
                    AUTO_TOL_NECK_ABS_M, AUTO_TOL_NECK_PLAIN_ABS_M, AUTO_TOL_HOLDOUT_M, AUTO_CAL_FRAMES, AUTO_HOLDOUT_FRAMES,
                    AUTO_NECK_WONLY_INCREMENT_M, AUTO_NECK_PLAIN_INCREMENT_M, AUTO_TOL_NECK_CAP_M,
                    tuple(os.environ.get(k_, "") for k_ in _ARITHMETIC_SWITCHES))
            hit = _CALIBRATION_CACHE.get(ckey)
            if hit is not None:
                self.apply_calibration(hit)
                if "warning" in hit:
                    warnings.warn("ZoeDepthEngine.calibrate: " + hit["warning"])
                return dict(hit)
        from .synthetic import make_sequence
        synth = lambda seeds: torch.from_numpy(np.concatenate([make_sequence(1, H, W, seed=s_) for s_ in seeds], 0)).to(self.dev)
        if frames_u8 is None:
            frames_u8 = synth(AUTO_CAL_SEEDS[:AUTO_CAL_FRAMES])
        frames_u8 = frames_u8.to(self.dev).contiguous()
        ncal = int(frames_u8.shape[0])
        H, W = int(frames_u8.shape[1]), int(frames_u8.shape[2])
        switchable = [k for k in BACKBONE_CLASSES if self.class_modes[k] in ("full", "wcls", "wmean", "wstat")] if self.auto_classes else []
        neck_cands = ["full"] if (not self.neck_f8 or not self.auto_classes or (neck_candidates is None and (not reference or self._sd is None))) \
            else list(neck_candidates or AUTO_NECK_CANDIDATES)
        # default (no explicit candidates, absolute reference available): the neck is calibrated per site AFTER the backbone classes and
        # the attention have been settled with the neck on both products (below); the group candidates are for explicit requests
        per_site = neck_candidates is None and len(neck_cands) > 1
        if per_site:
            neck_cands = ["full"]
        nh_, nw_ = net_size(H, W, self.target_hw)
        corr_ok = nw_ // self.cfg.patch == 32 and (nh_ // self.cfg.patch) % 2 == 0 and nh_ // self.cfg.patch <= 40
        attn_best = ("corr" if corr_ok else "single") if self.auto_attn else self.attn_mode
        saved_auto, self.auto_modes = self.auto_modes, False
        neck0 = self.neck_mode
        self.site_bias_corr = {}
        self._bias_corr_cache.clear()
        site_flops: Dict[str, float] = {}

        def depth(modes, neck, attn, frames=frames_u8, means=None):
            """depth maps [n,H,W] of `frames` under a mode combination (one plan, dropped after use); means: a dict that receives the channel
            means of every neck product's input rows (the run then goes launch by launch through the plan's tap path)"""
            self.set_class_modes(modes, neck, attn)
            plan = _ZoePlan(self, int(frames.shape[0]), H, W, True)
            plan.frames.copy_(frames)
            plan.run(None if means is None else {"__site_means__": means})
            d = plan.depth_m.clone()
            torch.cuda.synchronize(self.dev)
            site_flops.update(plan.site_flops)
            del plan
            return d

        l1f = lambda a, b: (a - b).abs().flatten(1).mean(1)              # per-frame L1
        worst = lambda a, b: float(l1f(a, b).max())

        full = {k: "full" for k in switchable}
        neck_full = "full" if (len(neck_cands) > 1 or per_site) else neck0
        bmeans: Dict[str, torch.Tensor] = {}
        ref = depth(full, neck_full, attn_best, means=bmeans if (switchable and os.environ.get("BS_AUTO_WSTAT") == "1") else None)
        # The yardstick of every decision below is run twice: the same launches must give the same bits.  (Earlier in round 6 one forward in a few
        # hundred differed beside another process allocating on the same GPU; it was located in the log-binomial kernel, whose shipped form has not
        # shown it since -- DESIGN section 7 -- and the check costs one forward and stays.)  A yardstick that does not reproduce is measured a
        # third time and the report says so.
        ref2 = depth(full, neck_full, attn_best)
        rerun_equal = bool(torch.equal(ref, ref2))
        if not rerun_equal:
            ref3 = depth(full, neck_full, attn_best)
            ref = ref2 if torch.equal(ref2, ref3) else ref
            warnings.warn("ZoeDepthEngine.calibrate: two runs of the same plan differed (is another process using this GPU?); the calibration's choices may not "
                          "be reproducible")
        del ref2
        # "wstat": the static correction rows dW E[a] of every backbone product, from the channel means of its patch rows on the calibration frames
        # (taken with every correction on: the means of the 16-bit values do not depend on the mode to any digit that matters here)
        self.backbone_bias_corr = {k_[3:]: (self.w[k_[3:] + ".lo"].double() * m_.double()).sum(1).float().view(1, -1).contiguous()
                                   for k_, m_ in bmeans.items() if k_[3:] + ".lo" in self.w}       # (deterministic form: see site_bias_corr below)
        del bmeans
        report = {"frame": f"{H}x{W}", "frames": ncal, "statistic": "worst frame (max over the calibration frames of the per-frame mean |d - d_ref|)",
                  "tol_class_m": tol_class, "tol_total_m": tol_total, "tol_abs_m": tol_abs, "l1_vs_full_m": {}, "yardstick_rerun_equal": rerun_equal}
        truth = hold = truth_h = None
        if reference and self._sd is not None:
            if holdout_u8 is None and per_site and AUTO_HOLDOUT_FRAMES > 0:
                holdout_u8 = synth(AUTO_HOLDOUT_SEEDS[:AUTO_HOLDOUT_FRAMES])
            if holdout_u8 is not None:
                hold = holdout_u8.to(self.dev).contiguous()
                assert tuple(hold.shape[1:]) == tuple(frames_u8.shape[1:])
            t_all = self.reference_depth(frames_u8 if hold is None else torch.cat([frames_u8, hold], 0))     # ONE reference engine for both sets
            truth, truth_h = t_all[:ncal], (t_all[ncal:] if hold is not None else None)
            report["l1_best_vs_reference_m"] = worst(ref, truth)          # the floor: nothing the calibration chooses can be closer than this
        chosen, cost = dict(full), {}
        cands_cls = (("wstat",) if os.environ.get("BS_AUTO_WSTAT") == "1" else ()) + tuple(AUTO_CANDIDATES)
        for k in switchable:
            for cand in cands_cls:
                if cand == "full":
                    chosen[k], cost[k] = "full", 0.0
                    break
                l1 = worst(depth({**full, k: cand}, neck_full, attn_best), ref)
                report["l1_vs_full_m"][f"{k}:{cand}"] = l1
                if l1 <= tol_class:
                    chosen[k], cost[k] = cand, l1
                    break
        attn = attn_best
        if self.auto_attn and attn_best == "corr":
            l1 = worst(depth(full, neck_full, "single"), ref)
            report["l1_vs_full_m"]["attn:single"] = l1
            if l1 <= tol_class:
                attn, cost["attn"] = "single", l1
        neck = neck_full
        for cand in neck_cands:
            if cand == "full":
                break
            l1 = worst(depth(full, cand, attn_best), ref)
            report["l1_vs_full_m"][f"neck:{cand}"] = l1
            if l1 <= tol_class:
                neck, cost["neck"] = cand, l1
                break
        saving = self._saving_gflop(1 + (nh_ // self.cfg.patch) * (nw_ // self.cfg.patch))
        passes_saved = {"wstat": 1.02, "wmean": 1.0, "wcls": 0.5, "full": 0.0}

        def step_up():
            """the live choice that pays the most depth error per unit of work saved goes one step back up; False when none is left"""
            nonlocal neck, attn
            live = [k_ for k_ in cost if cost[k_] > 0.0]
            if not live:
                return False
            def worth(k_):
                if k_ == "neck":
                    return 0.5 * sum(site_flops.values()) / max(2 * ncal, 1) / 1e9
                return saving[k_] * (1.0 if k_ == "attn" else passes_saved[chosen[k_]])
            worst_k = max(live, key=lambda k_: cost[k_] / max(worth(k_), 1e-9))
            if worst_k == "neck":
                neck = neck_full
            elif worst_k == "attn":
                attn = attn_best
            else:
                chosen[worst_k] = cands_cls[min(cands_cls.index(chosen[worst_k]) + 1, len(cands_cls) - 1)]
            cost[worst_k] = 0.0 if (worst_k in ("neck", "attn") or chosen[worst_k] == "full") else report["l1_vs_full_m"].get(f"{worst_k}:{chosen[worst_k]}", 0.0)
            return True

        # the combination, against the best mode and then against the reference
        while True:
            cheap = any(v != "full" for v in chosen.values()) or neck != neck_full or attn != attn_best
            d_c = depth(chosen, neck, attn) if cheap else ref
            total = worst(d_c, ref) if cheap else 0.0
            l1_abs = worst(d_c, truth) if truth is not None else None
            if neck != neck_full and neck_candidates is None and l1_abs is not None and l1_abs > AUTO_TOL_NECK_ABS_M:
                neck, cost["neck"] = neck_full, 0.0           # the cheaper neck is only worth half the tolerance (see AUTO_TOL_NECK_ABS_M)
                continue
            if total <= tol_total and (l1_abs is None or l1_abs <= tol_abs):
                break
            if not step_up():
                break
        report["l1_backbone_choice_vs_reference_m"] = l1_abs
        wsites, plain, cand_sites = [], [], {}

        def site_mode(names, pl=()):
            """the neck mode that runs the named candidates weight-only and those of `pl` on one pass"""
            return "wonly:" + ",".join(sorted(k_ for n_ in names for k_ in cand_sites[n_])) + \
                (";plain:" + ",".join(sorted(k_ for n_ in pl for k_ in cand_sites[n_])) if pl else "")

        def executed_gflop(chosen_, attn_, wsites_, plain_):
            """executed work per network input in 16-bit-pass GFLOP (backbone classes + attention + neck products): what the outer loop below
            minimises -- a proxy for time that needs no timing run"""
            g = sum(saving[k_] * {"wstat": 0.98, "wmean": 1.0, "wcls": 1.5, "full": 2.0}.get(chosen_.get(k_, self.class_modes[k_]), 2.0) for k_ in BACKBONE_CLASSES)
            g += saving["attn"] * (0.5 if attn_ == "single" else 1.5)
            wk = {k_ for n_ in wsites_ for k_ in cand_sites[n_]}
            pk = {k_ for n_ in plain_ for k_ in cand_sites[n_]}
            for k_, f_ in site_flops.items():
                g += f_ / max(2 * ncal, 1) / 1e9 * (1.0 if k_ in pk else (1.5 if k_ in wk else 2.0))
            return g

        def neck_stages(chosen_, attn_, d_c_, l1_abs_, total_):
            """the two per-product stages of the neck under one backbone choice -> dict(neck, wsites, plain, l1_abs, total, rep, corr)"""
            out = dict(neck=neck_full, wsites=[], plain=[], l1_abs=l1_abs_, total=total_, rep=None, corr={})
            tol_neck1 = min(max(AUTO_TOL_NECK_ABS_M, l1_abs_ + AUTO_NECK_WONLY_INCREMENT_M), AUTO_TOL_NECK_CAP_M)
            if l1_abs_ > AUTO_TOL_NECK_CAP_M - AUTO_NECK_WONLY_INCREMENT_M:
                return out
            # ---- candidate by candidate: what each product (or group of small products) costs when it alone drops the activation-rounding
            # correction (against the combination chosen so far), then the longest prefix of the error-per-FLOP order that stays within the
            # stage's budget against the reference AND within tol_total of the best mode.  The error grows along that order
            # (tools/probes/neck_site_study.py), so the prefix is bisected.
            tot_f = sum(site_flops.values()) or 1.0
            if not cand_sites:
                for k_, f_ in site_flops.items():
                    if k_.endswith("w_cls") or k_.startswith("mh."):
                        continue
                    if f_ >= AUTO_NECK_SITE_MIN_SHARE * tot_f:
                        cand_sites[k_] = [k_]
                    else:
                        g_ = next((n_ for n_, pre in AUTO_NECK_SMALL_GROUPS.items() if any(k_.startswith(p_) for p_ in pre)), None)
                        if g_ is not None:
                            cand_sites.setdefault("group:" + g_, []).append(k_)
            cflops = {n_: sum(site_flops[k_] for k_ in ks) for n_, ks in cand_sites.items()}
            err = {n_: worst(depth(chosen_, site_mode([n_]), attn_), d_c_) for n_ in cand_sites}
            order = sorted(cand_sites, key=lambda n_: err[n_] / cflops[n_])
            lo, hi, kept = 0, len(order), None
            while lo < hi:
                mid = (lo + hi + 1) // 2
                d_s = depth(chosen_, site_mode(order[:mid]), attn_)
                a_, t_ = worst(d_s, truth), worst(d_s, ref)
                if a_ <= tol_neck1 and t_ <= tol_total:
                    lo, kept = mid, (a_, t_)
                else:
                    hi = mid - 1
            ws = list(order[:lo])
            rep = {"tol_abs_m": tol_neck1, "groups": {n_: ks for n_, ks in cand_sites.items() if n_.startswith("group:")},
                   "l1_alone_vs_chosen_m": {n_: round(err[n_], 8) for n_ in order}, "weight_only": ws,
                   "flops_share_weight_only": round(sum(cflops[n_] for n_ in ws) / tot_f, 4)}
            out.update(wsites=ws, rep=rep)
            if lo > 0:
                out.update(neck=site_mode(ws), l1_abs=kept[0], total=kept[1])
            if lo > 0 and os.environ.get("BS_NECK_PLAIN", "1") != "0":
                # ---- second stage: one 16-bit pass for the weight-only candidates, largest first, with the static bias correction
                means: Dict[str, torch.Tensor] = {}
                self.site_bias_corr = {}
                self._bias_corr_cache.clear()
                depth(chosen_, out["neck"], attn_, means=means)      # channel means of every product's input rows on the calibration frames
                for k_, m_ in means.items():
                    k_ = k_[3:]                                       # "in:<weight key>"
                    if k_ in self.dw_sum:
                        # (multiply + tree sum in fp64, not a library GEMV: split-K atomics would make the low bits -- and with them a borderline
                        # decision below -- differ between two processes calibrating the same weights)
                        self.site_bias_corr[k_] = (self.dw_sum[k_].double() * m_.double()).sum(1).float().contiguous()
                cands = sorted((n_ for n_ in ws if cand_sites[n_] != ["rh.conv2.w"] and (n_.startswith("group:") or cflops[n_] >= AUTO_NECK_PLAIN_MIN_SHARE * tot_f)),
                               key=lambda n_: -cflops[n_])
                a0 = worst(depth(chosen_, out["neck"], attn_), truth)
                tol_neck2 = min(max(AUTO_TOL_NECK_PLAIN_ABS_M, a0 + AUTO_NECK_PLAIN_INCREMENT_M), AUTO_TOL_NECK_CAP_M)
                pl, trail, a_now = [], {}, a0
                for n_ in cands:
                    d_p = depth(chosen_, site_mode(ws, pl + [n_]), attn_)
                    a_, t_ = worst(d_p, truth), worst(d_p, ref)
                    trail[n_] = round(a_, 8)
                    if a_ <= tol_neck2 and t_ <= tol_total:
                        pl.append(n_)
                        a_now = a_
                        out.update(l1_abs=a_, total=t_)
                rep.update(plain_tol_abs_m=tol_neck2, l1_weight_only_m=round(a0, 8), l1_with_candidate_plain_m=trail, plain=list(pl),
                           l1_plain_m=round(a_now, 8), flops_share_plain=round(sum(cflops[n_] for n_ in pl) / tot_f, 4),
                           static_bias_correction=sorted(self.site_bias_corr))
                out.update(plain=pl, corr=dict(self.site_bias_corr))
                if pl:
                    out["neck"] = site_mode(ws, pl)
            return out

        if per_site and truth is not None and l1_abs is not None:
            best = neck_stages(chosen, attn, d_c, l1_abs, total)
            best_cost = executed_gflop(chosen, attn, best["wsites"], best["plain"])
            # (The stages run in sequence: the backbone classes take the cheapest modes their own tolerances allow, the neck gets what is left of
            # its budget -- on some weight sets little: eight seeds, round 6: a backbone choice at 5.4e-5 m leaves the neck two products and the
            # rate 8 % under the median.  Trading the other way was tried and does not pay -- a tighter backbone budget (3.5e-5 / 4.2e-5 m) made
            # those seeds 21 % / 4 % SLOWER, and a search over single class steps with the neck stages redone under each found no combination
            # with less executed work on four seeds: a backbone class one step up costs more than the neck products it frees.
            # profiles/r06_calibration_experiments.txt)
            neck, wsites, plain, l1_abs, total = best["neck"], best["wsites"], best["plain"], best["l1_abs"], best["total"]
            self.site_bias_corr = dict(best["corr"])
            self._bias_corr_cache.clear()
            if best["rep"] is not None:
                report["neck_sites"] = best["rep"]
            report["executed_gflop_per_input"] = round(best_cost, 1)
        # ---- held-out validation: frames no decision above has seen.  Above the line the latest relaxations are withdrawn, newest first
        if hold is not None:
            hv = {"tol_m": AUTO_TOL_HOLDOUT_M, "frames": int(hold.shape[0]), "withdrawn": []}
            neck_base = neck if not wsites else neck_full
            while True:
                per = l1f(depth(chosen, neck, attn, frames=hold), truth_h)
                if float(per.max()) <= AUTO_TOL_HOLDOUT_M:
                    break
                if plain:
                    hv["withdrawn"].append("plain:" + plain.pop())
                elif wsites:
                    hv["withdrawn"].append("wonly:" + wsites.pop())
                elif step_up():
                    hv["withdrawn"].append("backbone step")
                else:
                    break
                neck = site_mode(wsites, plain) if wsites else neck_base
            if hv["withdrawn"]:
                d_c = depth(chosen, neck, attn)
                l1_abs, total = worst(d_c, truth), worst(d_c, ref)
                if "neck_sites" in report:
                    report["neck_sites"].update(weight_only=list(wsites), plain=list(plain))
            hv.update(l1_frames_m=[round(float(v), 8) for v in per], l1_max_m=round(float(per.max()), 8), l1_mean_m=round(float(per.mean()), 8))
            report["holdout"] = hv
        if self.auto_attn and not corr_ok:
            # the split-precision kernel could not be tried on this geometry (every candidate above ran "single"): the engine keeps "corr",
            # which a plan of a capable geometry then uses and this one falls back from (_ZoePlan) -- "single" must be earned by a measurement
            attn = "corr"
            report["attn_note"] = f"attention not calibrated on a {nh_}x{nw_} network input (no split-precision kernel for it): kept at 'corr'"
        # the static corrections travel with the report (a sharded run's ranks all apply rank 0's): only those of the sites that run one pass
        keep_corr = {k_ for part in neck.split(";") if part.startswith("plain:") for k_ in part[6:].split(",")}
        self.site_bias_corr = {k_: v for k_, v in self.site_bias_corr.items() if k_ in keep_corr}
        self._bias_corr_cache.clear()
        final_modes = {**{k: self.class_modes[k] for k in BACKBONE_CLASSES}, **chosen}
        self.backbone_bias_corr = {k_: v for k_, v in self.backbone_bias_corr.items() if final_modes.get(k_.split(".")[-2]) == "wstat"}
        report.update(class_modes=final_modes, neck_mode=neck, attn_mode=attn,
                      l1_total_vs_full_m=total, l1_abs_vs_reference_m=l1_abs,
                      site_bias_corr={k_: v.cpu().tolist() for k_, v in self.site_bias_corr.items()},
                      backbone_bias_corr={k_: v.view(-1).cpu().tolist() for k_, v in self.backbone_bias_corr.items()})
        if l1_abs is not None and l1_abs > TOLERANCE_M:
            fixed = [k for k in BACKBONE_CLASSES if k not in switchable]
            report["warning"] = (f"depth L1 of the calibration frames against the reference-precision engine is {l1_abs:.2e} m with every calibrated "
                                 f"choice at its most accurate: above the {TOLERANCE_M:.0e} m tolerance for these weights"
                                 + (f" (modes fixed by the caller, not calibrated: {({k: self.class_modes[k] for k in fixed})})" if fixed else ""))
            warnings.warn("ZoeDepthEngine.calibrate: " + report["warning"])
        elif truth is None:
            report["note"] = "no absolute reference (the engine holds no source weights): l1_total_vs_full_m is relative to the best mode only"
        if truth is not None and report.get("l1_best_vs_reference_m", 0.0) > 0.5 * TOLERANCE_M and "warning" not in report:
            # (round 6, the outlier-channel weights: the best mode reads 5.1e-5 m on the calibration frames and 1.2e-5 ... 2.2e-4 m on eight frames of
            # the bench's sequence -- on such weights the error of the e4m3 correction planes varies several-fold from frame to frame)
            report["margin_note"] = (f"even with every correction on, the calibration frames read {report['l1_best_vs_reference_m']:.2e} m against the reference-precision "
                                     f"engine -- more than half the {TOLERANCE_M:.0e} m tolerance: frames with other statistics may exceed it.  precision='reference' "
                                     "(three 16-bit passes per product) holds ~1.5e-5 m on such weights at about half the rate")
            warnings.warn("ZoeDepthEngine.calibrate: " + report["margin_note"])
        self.set_class_modes(chosen, neck, attn)
        self.auto_modes = saved_auto
        torch.cuda.synchronize(self.dev)
        report["calibrate_s"] = round(_time.perf_counter() - t_start, 2)
        self.calibration = report
        if ckey is not None:
            _CALIBRATION_CACHE[ckey] = dict(report)
        torch.cuda.empty_cache()
        return report

    def _weights_fingerprint(self) -> str:
        """blake2b over every source tensor's name, shape and bytes: what identifies a weight set for the calibration cache (round-5
        advisor: the (count, sum, sum of squares) of rounds 4-5 could not tell a permutation from the original)"""
        import hashlib
        h = hashlib.blake2b(digest_size=16)
        for k in sorted(self._sd):
            t = self._sd[k].detach().cpu().contiguous()
            h.update(k.encode())
            h.update(str((tuple(t.shape), t.dtype)).encode())
            h.update(memoryview(t.reshape(-1).view(torch.uint8).numpy()))
        return h.hexdigest()

    def _w8conv(self, key: str, t: torch.Tensor) -> torch.Tensor:
        """conv weight [O, I, kh, kw] for the FP8-correction conv path (L.f8_conv_weight); scales to self.f8s[key]."""
        w8, sb = L.f8_conv_weight(t.permute(0, 2, 3, 1), self.dtype, device=self.dev)
        self.f8s[key] = sb
        return w8.to(self.dev)

    def _wp(self, key: str, t: torch.Tensor) -> torch.Tensor:
        """plain neck / head weight: FP8-correction packing when the whole neck runs that format, else three 16-bit passes"""
        if self.neck_f8 and t.shape[1] % 128 == 0:
            self.dw_sum[key] = (t - t.to(self.dtype).float()).to(self.dev).contiguous()
        return self._w8(key, t) if self.neck_f8 else self._wn(t)

    def _wc(self, key: str, t: torch.Tensor) -> torch.Tensor:
        if self.neck_f8:
            self.dw_sum[key] = (t - t.to(self.dtype).float()).sum((2, 3)).to(self.dev).contiguous()          # [O, I]: summed over the taps
        return self._w8conv(key, t) if self.neck_f8 else self._wn_conv(t)

    def site_bias(self, wkey: str, bias: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
        """the bias a neck / head product runs with: its own, plus -- while the product runs ONE 16-bit pass and the calibration left a
        correction for it -- the static part dW E[a] of the weight-rounding error that pass makes (site_bias_corr)"""
        corr = self.site_bias_corr.get(wkey)
        if corr is None or not self.neck_site_plain(wkey) or os.environ.get("BS_NECK_BIAS_CORR", "1") == "0":
            return bias
        hit = self._bias_corr_cache.get(wkey)
        if hit is None:
            hit = self._bias_corr_cache[wkey] = (corr if bias is None else (bias + corr)).contiguous()
        return hit

    def _wn(self, t: torch.Tensor) -> torch.Tensor:
        """plain GEMM weight [N, K]; accurate: [N, 3K] = [W_hi | W_hi | W_lo] against A = [hi | lo], then hi again."""
        if not self.acc:
            return self._h(t)
        hi, lo = self._split(t.to(self.dev))                 # (on the device: the reference precision splits all 300 M backbone weights)
        return torch.cat([hi, hi, lo], 1).contiguous()

    def _wn_conv(self, t: torch.Tensor) -> torch.Tensor:
        """neck conv weight [O, I, kh, kw]; accurate: segment 0 = [W_hi | W_hi] against the 2I (hi | lo) channels,
        segment 1 = W_lo against the hi channels again; each segment in conv K order (chunk, tap, channel)."""
        if not self.acc:
            return self._conv_w(t)
        k = t.to(self.dev).permute(0, 2, 3, 1)                      # [O, kh, kw, I]
        hi, lo = self._split(k)
        seg0 = L.conv_weight(torch.cat([hi, hi], -1))
        return torch.cat([seg0, L.conv_weight(lo)], 1).to(self.dev).contiguous()

    def _ingest(self, sd: Dict[str, torch.Tensor]):
        c, w = self.cfg, self.w
        g = lambda k: sd[k].detach().float()
        pe = "backbone.beit.embeddings."
        w["pe.w"] = self._w8("pe.w", g(pe + "patch_embeddings.projection.weight").reshape(c.hidden, -1))
        w["pe.b"] = self._f(g(pe + "patch_embeddings.projection.bias"))
        w["cls"] = self._f(g(pe + "cls_token").reshape(-1))
        for l in range(c.layers):
            p = f"backbone.beit.layers.{l}."
            w[f"l{l}.ln1.g"], w[f"l{l}.ln1.b"] = self._f(g(p + "layernorm_before.weight")), self._f(g(p + "layernorm_before.bias"))
            w[f"l{l}.ln2.g"], w[f"l{l}.ln2.b"] = self._f(g(p + "layernorm_after.weight")), self._f(g(p + "layernorm_after.bias"))
            w[f"l{l}.qkv.w"] = self._w8(f"l{l}.qkv.w", torch.cat([g(p + "attention.q_proj.weight"), g(p + "attention.k_proj.weight"),
                                                                  g(p + "attention.v_proj.weight")], 0))
            # k_proj has no bias (HF modeling_beit.py:305-307)
            w[f"l{l}.qkv.b"] = self._f(torch.cat([g(p + "attention.q_proj.bias"), torch.zeros(c.hidden),
                                                  g(p + "attention.v_proj.bias")], 0))
            w[f"l{l}.o.w"], w[f"l{l}.o.b"] = self._w8(f"l{l}.o.w", g(p + "attention.o_proj.weight")), self._f(g(p + "attention.o_proj.bias"))
            w[f"l{l}.fc1.w"], w[f"l{l}.fc1.b"] = self._w8(f"l{l}.fc1.w", g(p + "mlp.fc1.weight")), self._f(g(p + "mlp.fc1.bias"))
            w[f"l{l}.fc2.w"], w[f"l{l}.fc2.b"] = self._w8(f"l{l}.fc2.w", g(p + "mlp.fc2.weight")), self._f(g(p + "mlp.fc2.bias"))
            w[f"l{l}.lam1"], w[f"l{l}.lam2"] = self._f(g(p + "lambda_1")), self._f(g(p + "lambda_2"))
            self._raw_tables.append(g(p + "relative_position_bias.relative_position_bias_table"))
        factors = (4, 2, 1, 0.5)
        for i, (ch, f) in enumerate(zip(c.neck_hidden, factors)):
            ro = g(f"neck.reassemble_stage.readout_projects.{i}.0.weight")
            w[f"ro{i}.w_tok"] = self._wp(f"ro{i}.w_tok", ro[:, :c.hidden])          # concat(token, cls) @ W^T = token @ W1^T + cls @ W2^T
            w[f"ro{i}.w_cls"] = self._wp(f"ro{i}.w_cls", ro[:, c.hidden:])
            w[f"ro{i}.b"] = self._f(g(f"neck.reassemble_stage.readout_projects.{i}.0.bias"))
            p = f"neck.reassemble_stage.layers.{i}."
            w[f"ra{i}.proj.w"] = self._wp(f"ra{i}.proj.w", g(p + "projection.weight").reshape(ch, c.hidden))
            w[f"ra{i}.proj.b"] = self._f(g(p + "projection.bias"))
            if f > 1:
                s = int(f)
                wt = g(p + "resize.weight")                         # ConvTranspose2d [Cin, Cout, s, s]
                w[f"ra{i}.up.w"] = self._wp(f"ra{i}.up.w", wt.permute(2, 3, 1, 0).reshape(s * s * ch, ch))  # n = (ky*s+kx)*Cout + co
                w[f"ra{i}.up.b"] = self._f(g(p + "resize.bias").repeat(s * s))
            elif f < 1:
                w[f"ra{i}.down.w"] = self._wc(f"ra{i}.down.w", g(p + "resize.weight"))
                w[f"ra{i}.down.b"] = self._f(g(p + "resize.bias"))
            w[f"nc{i}.w"] = self._wc(f"nc{i}.w", g(f"neck.convs.{i}.weight"))
        for i in range(4):
            p = f"neck.fusion_stage.layers.{i}."
            w[f"fu{i}.proj.w"] = self._wp(f"fu{i}.proj.w", g(p + "projection.weight").reshape(c.fusion, c.fusion))
            w[f"fu{i}.proj.b"] = self._f(g(p + "projection.bias"))
            for r in (1, 2):
                for cv in (1, 2):
                    w[f"fu{i}.r{r}.c{cv}.w"] = self._wc(f"fu{i}.r{r}.c{cv}.w", g(p + f"residual_layer{r}.convolution{cv}.weight"))
                    w[f"fu{i}.r{r}.c{cv}.b"] = self._f(g(p + f"residual_layer{r}.convolution{cv}.bias"))
        # relative_head.projection exists only in checkpoints converted with config.add_projection (HF modeling_zoedepth.py:344-346)
        self.add_projection = "relative_head.projection.weight" in sd
        for n in (("projection",) if self.add_projection else ()) + ("conv1",):
            w[f"rh.{n}.w"] = self._wc(f"rh.{n}.w", g(f"relative_head.{n}.weight"))
            w[f"rh.{n}.b"] = self._f(g(f"relative_head.{n}.bias"))
        # conv2 follows the x2 upsampling; it is evaluated as nine 1x1 tap products at the LOW resolution (one GEMM, n = tap * Cout + o)
        # that bs_upconv_tapsum interpolates and sums (see the plan): weight [O, I, 3, 3] -> [(ky, kx, o), I]
        w2 = g("relative_head.conv2.weight")
        w["rh.conv2.w"] = self._wp("rh.conv2.w", w2.permute(2, 3, 0, 1).reshape(9 * w2.shape[0], w2.shape[1]))
        w["rh.conv2.b"] = self._f(g("relative_head.conv2.bias"))
        # ---- metric head
        mh = "metric_head."
        B_, E = c.bottleneck, c.bin_dim
        # conv2 and the seed regressors work at the bottleneck resolution (12 x 16 pixels per image: no time at all) and decide where every
        # bin STARTS: as single 16-bit products they cost 4.3e-5 + 7.1e-5 m of depth on outlier-channel weights
        # (tools/probes/outlier_rounding_study.py mh16) -- in accurate mode they are split-precision products like the neck.
        w["mh.conv2.w"] = self._wp("mh.conv2.w", g(mh + "conv2.weight").reshape(B_, B_))
        w["mh.conv2.b"] = self._f(g(mh + "conv2.bias"))
        # Two head slots are carried side by side (buffers, block-diagonal weights) and an image is routed to one of them.  A
        # single-head model (ZoeD_N / ZoeD_K) fills slot 0 and leaves slot 1 as zeros with every image routed to slot 0.
        single = c.single_head
        sq = lambda k: g(k).reshape(g(k).shape[0], -1)
        SM, PM, HID = c.seed_mlp, c.proj_mlp, c.clb_hidden

        def pref(kind, slot, i=None):
            if single:
                if slot:
                    return None
                return {"seed": mh + "seed_bin_regressor.", "att": mh + f"attractors.{i}.", "clb": mh + "conditional_log_binomial.mlp."}[kind]
            n = c.head_names[slot]
            return {"seed": mh + f"seed_bin_regressors.{n}.", "att": mh + f"attractors.{n}.{i}.", "clb": mh + f"conditional_log_binomial.{n}.mlp."}[kind]

        def two(kind, suffix, i=None, flat=True, rep=1):
            """the tensor of both slots (slot 1 = zeros for a single head); rep replicates rows (attractor counts below 4)"""
            t0 = g(pref(kind, 0, i) + suffix)
            t0 = t0.reshape(t0.shape[0], -1) if (flat and t0.dim() > 1) else t0
            if rep > 1:
                t0 = t0.repeat_interleave(rep, 0)
            p1 = pref(kind, 1, i)
            if p1 is None:
                return t0, torch.zeros_like(t0)
            t1 = g(p1 + suffix)
            t1 = t1.reshape(t1.shape[0], -1) if (flat and t1.dim() > 1) else t1
            return t0, (t1.repeat_interleave(rep, 0) if rep > 1 else t1)

        s0, s1 = two("seed", "conv1.weight")
        w["seed.c1.w"] = self._wn(torch.cat([s0, s1], 0))                                                     # [2*SM, 256]
        w["seed.c1.b"] = self._f(torch.cat(two("seed", "conv1.bias")))
        w["seedproj.c1.w"] = self._h(sq(mh + "seed_projector.conv1.weight"))                                 # [PM, 256]
        w["seedproj.c1.b"] = self._f(g(mh + "seed_projector.conv1.bias"))
        s0, s1 = two("seed", "conv2.weight")
        w["seed.c2.w"] = self._wn(torch.block_diag(s0, s1))                                                   # [2*nb, 2*SM]
        w["seed.c2.b"] = self._f(torch.cat(two("seed", "conv2.bias")))
        w["seedproj.c2.w"] = self._h(sq(mh + "seed_projector.conv2.weight"))                                 # [E, PM]
        w["seedproj.c2.b"] = self._f(g(mh + "seed_projector.conv2.bias"))
        self.na_eff = []
        for i in range(4):
            p = mh + f"projectors.{i}."
            w[f"pj{i}.c1.w"], w[f"pj{i}.c1.b"] = self._wp(f"pj{i}.c1.w", sq(p + "conv1.weight")), self._f(g(p + "conv1.bias"))
            w[f"pj{i}.c2.w"], w[f"pj{i}.c2.b"] = self._wn(sq(p + "conv2.weight")), self._f(g(p + "conv2.bias"))
            w[f"at{i}.c1.w"] = self._h(torch.cat(two("att", "conv1.weight", i), 0))                           # [2E, E]
            w[f"at{i}.c1.b"] = self._f(torch.cat(two("att", "conv1.bias", i)))
            # kind "mean": an attractor replicated r times leaves the mean unchanged -> counts below 4 are padded by replication
            na = c.attractors_at(i)
            rep = 1 if na % 4 == 0 else 4 // math.gcd(na, 4)
            self.na_eff.append(na * rep)
            w[f"at{i}.c2.w"] = self._h(torch.block_diag(*two("att", "conv2.weight", i, rep=rep)))              # [2*na, 2E]
            w[f"at{i}.c2.b"] = self._f(torch.cat(two("att", "conv2.bias", i, rep=rep)))
        R = c.rel_features
        X = 1 if single else 0                                     # the single head's extra input: the relative depth
        w00, w01 = two("clb", "0.weight")                          # [HID, R + X + E] = [last R | (rel depth) | emb E]
        w["clb.emb.w"] = self._wn(torch.cat([w00[:, R + X:], w01[:, R + X:]], 0))                               # [2*HID, E]
        w["clb.emb.b"] = self._f(torch.cat(two("clb", "0.bias")))
        # Round 5: the embedding half of the log-binomial MLP's first layer reads the LAST projector's output emb = W_c2 e1 + b_c2 (a 1x1 convolution
        # without activation, HF modeling_zoedepth.py:749-772): two linear maps in a row.  Composed here in fp32 -- (W_clb W_c2) e1 + (W_clb b_c2 +
        # b_clb) -- the product reads the projector's 64-channel hidden map instead of the 128-channel embedding (half the bytes, half the K).
        wce = torch.cat([w00[:, R + X:], w01[:, R + X:]], 0)                                                   # [2*HID, E]
        wc2_3, bc2_3 = sq(mh + "projectors.3.conv2.weight"), g(mh + "projectors.3.conv2.bias")                 # [E, PM], [E]
        w["clb.e1.w"] = self._wn(wce @ wc2_3)                                                                  # [2*HID, PM]
        w["clb.e1.b"] = self._f(wce @ bc2_3 + torch.cat(two("clb", "0.bias")))
        w["clb.w0_last"] = self._f(torch.stack([w00[:, :R], w01[:, :R]]))                                      # [2, HID, R]
        w["clb.w2"] = self._f(torch.stack(two("clb", "2.weight")))                                             # [2, 4, HID]
        w["clb.b2"] = self._f(torch.stack(two("clb", "2.bias")))                                               # [2, 4]
        if single:
            r0 = torch.cat([w00[:, R], sq("relative_head.conv3.weight").flatten(), g("relative_head.conv3.bias").flatten()])
            w["clb.rel"] = self._f(torch.stack([r0, torch.zeros_like(r0)]))                                    # [2, HID + R + 1]
            return
        # ---- router (HF modeling_zoedepth.py:775-962)
        pt = mh + "patch_transformer."
        w["rt.emb.w"] = self._h(sq(pt + "embedding_convPxP.weight"))
        w["rt.emb.b"] = self._f(g(pt + "embedding_convPxP.bias"))
        for l in range(c.pt_layers):
            p = pt + f"transformer_encoder.{l}."
            w[f"rt{l}.qkv.w"] = self._h(torch.cat([g(p + "self_attn.query.weight"), g(p + "self_attn.key.weight"), g(p + "self_attn.value.weight")], 0))
            w[f"rt{l}.qkv.b"] = self._f(torch.cat([g(p + "self_attn.query.bias"), g(p + "self_attn.key.bias"), g(p + "self_attn.value.bias")]))
            w[f"rt{l}.o.w"], w[f"rt{l}.o.b"] = self._h(g(p + "self_attn.out_proj.weight")), self._f(g(p + "self_attn.out_proj.bias"))
            w[f"rt{l}.l1.w"], w[f"rt{l}.l1.b"] = self._h(g(p + "linear1.weight")), self._f(g(p + "linear1.bias"))
            w[f"rt{l}.l2.w"], w[f"rt{l}.l2.b"] = self._h(g(p + "linear2.weight")), self._f(g(p + "linear2.bias"))
            for n in (1, 2):
                w[f"rt{l}.n{n}.g"], w[f"rt{l}.n{n}.b"] = self._f(g(p + f"norm{n}.weight")), self._f(g(p + f"norm{n}.bias"))
        w["cl.l1.w"], w["cl.l1.b"] = self._h(g(mh + "mlp_classifier.linear1.weight")), self._f(g(mh + "mlp_classifier.linear1.bias"))
        l2w, l2b = g(mh + "mlp_classifier.linear2.weight"), g(mh + "mlp_classifier.linear2.bias")
        w["cl.l2.w"] = self._h(torch.cat([l2w, torch.zeros(2, l2w.shape[1])], 0))                              # N padded 2 -> 4
        w["cl.l2.b"] = self._f(torch.cat([l2b, torch.zeros(2)]))

    # ------------------------------------------------------------------------------------------
    def _rel_bias(self, hp: int, wp: int, Sp: int):
        """[layers] x fp32 [heads, Sp, Sp]: the table re-interpolated for an (hp, wp) window and
        gathered (HF modeling_beit.py:220-265); -1e30 in the padded key columns."""
        key = (hp, wp)
        if key in self._bias_cache:
            return self._bias_cache[key]
        c = self.cfg
        old = 2 * (c.image_size // c.patch) - 1
        nh_, nw_ = 2 * hp - 1, 2 * wp - 1
        idx = _relative_position_index(hp, wp).view(-1)
        S = hp * wp + 1
        out = []
        for tab in self._raw_tables:
            sub = tab[: old * old].reshape(1, old, old, -1).permute(0, 3, 1, 2)
            new = F.interpolate(sub, size=(nh_, nw_), mode="bilinear").permute(0, 2, 3, 1).reshape(nh_ * nw_, -1)
            full = torch.cat([new, tab[old * old:]])
            b = full[idx].view(S, S, -1).permute(2, 0, 1)
            pad = torch.full((c.heads, Sp, Sp), -1.0e30)
            pad[:, :S, :S] = b * LOG2E          # bs_attention works in the log2 domain
            pad[:, S:, :S] = 0.0
            out.append(pad.to(self.dev))
        self._bias_cache[key] = out
        return out

    def _rel_table(self, hp: int, wp: int):
        """[layers] x fp32 [heads, (2hp-1)(2wp-1)+3]: the bias table re-interpolated for an (hp, wp) window (HF modeling_beit.py:
        220-245), times log2(e), patch-pair entries in reversed order -- the operand of bs_attention_table (no [heads, Sp, Sp]
        tensor is materialised)."""
        key = ("tab", hp, wp)
        if key in self._bias_cache:
            return self._bias_cache[key]
        c = self.cfg
        old = 2 * (c.image_size // c.patch) - 1
        nh_, nw_ = 2 * hp - 1, 2 * wp - 1
        out = []
        for tab in self._raw_tables:
            sub = tab[: old * old].reshape(1, old, old, -1).permute(0, 3, 1, 2)
            new = F.interpolate(sub, size=(nh_, nw_), mode="bilinear").permute(0, 2, 3, 1).reshape(nh_ * nw_, -1)
            full = torch.cat([torch.flip(new, dims=[0]), tab[old * old:]])       # [ntab, heads]: patch-pair entries reversed, cls entries last
            out.append((full.t() * LOG2E).contiguous().to(self.dev))
        self._bias_cache[key] = out
        return out

    def plan_for(self, B: int, H: int, W: int, flip: bool = True) -> "_ZoePlan":
        if self.auto_modes and self.calibration is None:
            self.calibrate(H, W)
        key = (B, H, W, flip)
        if key not in self._plans:
            self._plans[key] = _ZoePlan(self, B, H, W, flip)
        return self._plans[key]

    def infer(self, frames_u8: torch.Tensor, flip_aug: bool = True, taps: Optional[dict] = None, want_u16: bool = True,
              graph: bool = False):
        """uint8 [B,H,W,3] on the GPU -> (depth metres fp32 [B,H,W], uint16 metres*256 [B,H,W] as int16 storage).

        Outputs are the plan's static buffers (valid until the next call with the same shape).  graph=True: the plan of this
        shape is captured into a HIP graph on first use and replayed afterwards (one launch call per forward)."""
        assert frames_u8.dtype == torch.uint8 and frames_u8.is_cuda and frames_u8.dim() == 4 and frames_u8.shape[-1] == 3
        B, H, W, _ = frames_u8.shape
        if B == 0:      # an empty batch: nothing to launch
            return (torch.empty(0, H, W, device=self.dev), torch.empty(0, H, W, device=self.dev, dtype=torch.int16))
        plan = self.plan_for(B, H, W, flip_aug)
        plan.frames.copy_(frames_u8)
        if graph and taps is None:
            plan.plan.capture()
        plan.run(taps)
        return plan.depth_m, plan.depth_u16


class _Pool:
    """Plan-time buffer reuse.  The launch sequence of a plan is fixed, so the lifetime of every intermediate is known while the
    plan is being built: ``free(t)`` -- placed right after the call that reads ``t`` last -- returns its block, and a later
    ``alloc`` of at most that size takes it over (best fit).  Kernels of the main lane run in program order, so a block handed
    on at build position p is only ever overwritten by work issued after p.  A side lane (Plan.lane = 1; unused since round 5) must not use pooled blocks
    (``hold``): it runs beside main-lane kernels that were added later."""

    def __init__(self, dev):
        self.dev = dev
        self.free_blocks = []        # uint8 tensors
        self.blocks = []             # every block ever made (engine export: pooled intermediates carry no data)
        self.owner = {}              # data_ptr -> block
        self.hold = False
        self.bytes_new = 0

    def alloc(self, shape, dtype):
        n = 1
        for d in shape:
            n *= int(d)
        nbytes = max(n * torch.empty((), dtype=dtype).element_size(), 16)
        blk = None
        if not self.hold:
            fits = [b for b in self.free_blocks if b.numel() >= nbytes]
            if fits:
                blk = min(fits, key=lambda b: b.numel())
                self.free_blocks = [b for b in self.free_blocks if b is not blk]
        if blk is None:
            blk = torch.empty((nbytes + 255) // 256 * 256, dtype=torch.uint8, device=self.dev)
            self.bytes_new += blk.numel()
            self.blocks.append(blk)
        t = blk[:nbytes].view(dtype).view(*shape)
        self.owner[t.data_ptr()] = blk
        return t

    def free(self, *tensors):
        for t in tensors:
            blk = self.owner.pop(t.data_ptr(), None) if t is not None else None
            if blk is not None:
                self.free_blocks.append(blk)


class _ZoePlan:
    """All buffers + the launch sequence for one (frames, H, W, flip) configuration."""

    def __init__(self, eng: ZoeDepthEngine, B: int, H: int, W: int, flip: bool):
        self.eng = eng
        c, w, dt_, dev = eng.cfg, eng.w, eng.dtype, eng.dev
        NB = 2 * B if flip else B
        nh_, nw_ = net_size(H, W, eng.target_hw)
        hp, wp = nh_ // c.patch, nw_ // c.patch
        T0 = hp * wp
        S = T0 + 1
        Sp = (S + 63) // 64 * 64
        Hd = c.hidden
        self.geom = dict(B=B, NB=NB, H=H, W=W, nh=nh_, nw=nw_, hp=hp, wp=wp, S=S, Sp=Sp)
        self.site_flops: Dict[str, float] = {}      # FP8-format neck / head products of this plan: weight key -> algorithmic FLOPs
        P = L.Plan(dev)
        self.plan = P
        # intermediates of the neck / heads come from a pool and are handed on after their last reader (free); the backbone's
        # buffers (whose padding rows must stay zero) are plain allocations
        pool = _Pool(dev)
        self.pool = pool
        e16 = lambda *s: pool.alloc(s, dt_)
        z16 = lambda *s: torch.zeros(*s, device=dev, dtype=dt_)
        e32 = lambda *s: pool.alloc(s, torch.float32)
        free = pool.free
        use_tab = wp == 32 and hp <= 40         # every 512-wide network input: bias from the per-head table held in LDS
        bias = eng._rel_table(hp, wp) if use_tab else eng._rel_bias(hp, wp, Sp)
        # Row order of the token tensors (residual stream, LN / attention outputs, MLP hidden).  Grouped (with the table
        # attention): rows [0, NB) are the cls tokens of the NB images, row NB + b*T0 + t is patch t of image b -- the cls rows
        # sit in the first GEMM tile, which alone evaluates the activation-rounding correction ("wcls", bs_gemm f8_wonly_from).
        # Otherwise image-major, cls first ([NB, S, hidden]).
        # The cls group is padded to whole 256-row GEMM tiles (CP rows, NB of them used): a tile then holds either cls rows or patch
        # rows, never both, so which rows get the activation-rounding correction does not depend on the batch size.
        grouped = use_tab
        self.grouped = grouped
        CP = (NB + 255) // 256 * 256 if grouped else 0
        MT = CP + NB * T0 if grouped else NB * S          # rows of the token tensors

        acc = eng.acc
        m2 = 2 if acc else 1          # channel multiplier of (hi | lo) activations
        np3 = 3 if acc else 1         # K passes of a GEMM whose weights AND activations are split
        SP = 16 if acc else 0         # "split" flag (bit 4 of the dtype argument) of the pointwise producers
        self.frames = torch.empty(B, H, W, 3, device=dev, dtype=torch.uint8)
        patches = e16(NB * T0, 3 * c.patch * c.patch * m2)
        x = torch.zeros(MT, Hd, device=dev, dtype=torch.float32)           # (padding rows of the grouped layout stay finite)
        xn = z16(MT, Hd * m2)
        # split-precision attention (eng.attn_mode == "corr"): Q, K, V^T are allocated twice over, the rounding residuals behind the
        # values (bs_gemm_desc.qkv_lo_off -> bs_attention_table_corr).  Built for the table kernel's pipelined form (512-wide inputs,
        # even hp: 384x512 and 416x512); other geometries run the single-operand kernel.
        corr = bool(eng.acc and eng.attn_mode == "corr" and use_tab and hp % 2 == 0)
        self.attn_corr = corr
        QN = NB * c.heads * Sp * 64
        q, k, vt = z16((2 if corr else 1) * NB, c.heads, Sp, 64), z16((2 if corr else 1) * NB, c.heads, Sp, 64), z16((2 if corr else 1) * NB, c.heads, 64, Sp)
        ao = z16(MT, Hd * m2)
        hid = z16(MT, c.intermediate * m2)
        def PE(C):
            """elements between consecutive pixels / rows of a neck activation with C channels"""
            return C * m2

        taps16 = [z16(MT, PE(Hd)) for _ in c.taps]

        f8s = eng.f8s

        single = eng.single_keys

        def fmt(*wkeys):
            """producer format flag of an activation: 32 = (hi16 | hi8 | lo8) when every consumer GEMM runs its corrections on
            the FP8 MFMA, 16 = (hi | lo) 16-bit pairs otherwise (accurate mode), 0 = single (fast mode, or every consumer is a
            single-pass GEMM)."""
            if not acc or all(k_ in single for k_ in wkeys):
                return 0
            return 32 if all(k_ in f8s for k_ in wkeys) else 16

        def am(wkey):
            """row multiplier of the activation a backbone GEMM reads: 1 = single rows, 2 = pair rows"""
            return 2 if fmt(wkey) else 1

        def prow(wkey):
            """dtype-argument bits of a producer whose consumer GEMM runs no FP8 stage on the patch rows ("wmean"): only the rows
            below CP (the cls tile) need their FP8 planes.  bs_layernorm: rows << 8"""
            return (CP << 8) if (acc and grouped and fmt(wkey) == 32 and eng.mode_of(wkey) in ("wmean", "wstat")) else 0

        MEAN_STEP = 8      # the rank-1 correction's token mean uses every 8th patch row (probe: same depth result as the full mean)

        def bgemm(name, A, wkey, out, M, N, K, **kw):
            """backbone GEMM.  Accurate mode: A_hi W_hi + A_hi W_lo + A_lo W_hi in one launch -- the two corrections on the
            block-scaled FP8 MFMA where the weight was packed for it (A = [hi16 | hi8 | lo8], 2 pass-equivalents), else as
            K segments of 16-bit (hi | lo) pairs (3 passes)."""
            if acc and grouped and wkey in f8s and wkey[0] == "l":
                # (calibrate(): the channel means of the product's patch rows, what "wstat"'s static correction is formed from)
                P.mark("in:" + wkey, A, ("chanmean", CP * 2 * K, NB * T0, 2 * K, K))
            if acc and wkey in single:
                P.gemm(name, A, w[wkey], out, M=M, N=N, K=K, lda=K, precision_passes=1, **kw)
            elif acc and wkey in f8s and eng.mode_of(wkey) == "w":
                # weight-rounding correction only: the FP8 segment is A_hi8 x W_lo8 (K bytes per row, both halves on the same scales)
                sb0, _ = f8s[wkey]
                P.gemm(name, A, w[wkey], out, M=M, N=N, K=K, lda=2 * K, f8_seg=K,
                       f8_scales=(127 - L.F8_ACT_HI_EXP, sb0, 127 - L.F8_ACT_HI_EXP, sb0), precision_passes=1, **kw)
            elif acc and wkey in f8s and eng.mode_of(wkey) == "wstat" and grouped and wkey in eng.backbone_bias_corr:
                # "wmean" with the calibration frames' channel means in place of the image's own: the correction is a constant row, added to the
                # patch rows through the bias2 path as ONE group (the cls tile runs both FP8 corrections and must not get it)
                sb0, sb1 = f8s[wkey]
                P.gemm(name, A, w[wkey], out, M=M, N=N, K=K, lda=2 * K, f8_seg=2 * K, f8_skip_from=CP, bias2=(eng.backbone_bias_corr[wkey], CP, NB * T0),
                       f8_scales=(127 - L.F8_ACT_HI_EXP, sb0, 127 - L.F8_ACT_LO_EXP, sb1), precision_passes=1, **kw)
            elif acc and wkey in f8s and eng.mode_of(wkey) in ("wmean", "wstat") and grouped:
                # cls tile: both FP8 corrections.  Patch tiles: ONE 16-bit pass; the weight-rounding error A dW^T is replaced by its
                # token-independent part 1 (mean_tokens(A) dW^T), a per-image bias formed by a column-mean kernel over a sample of
                # the image's patch rows and bs_rank1_bias, a [NB, K] x [K, N] product (Numerics)
                sb0, sb1 = f8s[wkey]
                abar = pool.alloc((NB, K), torch.bfloat16)
                b2 = e32(NB, N)
                P.add(name + ".cm", "bs_col_mean", A, 2 * K, CP, T0, NB, MEAN_STEP, K, abar, b2, NB * N, L.dt(A))
                P.add(name + ".r1", "bs_rank1_bias", abar, w[wkey + ".lo"], b2, NB, N, K)
                P.gemm(name, A, w[wkey], out, M=M, N=N, K=K, lda=2 * K, f8_seg=2 * K, f8_skip_from=CP, bias2=(b2, CP, T0),
                       f8_scales=(127 - L.F8_ACT_HI_EXP, sb0, 127 - L.F8_ACT_LO_EXP, sb1), precision_passes=1, **kw)
                free(abar, b2)
            elif acc and wkey in f8s:
                sb0, sb1 = f8s[wkey]
                wonly = CP if (eng.mode_of(wkey) == "wcls" and grouped) else 0      # tiles past the cls group: weight correction only
                P.gemm(name, A, w[wkey], out, M=M, N=N, K=K, lda=2 * K, f8_seg=2 * K, f8_wonly_from=wonly,
                       f8_scales=(127 - L.F8_ACT_HI_EXP, sb0, 127 - L.F8_ACT_LO_EXP, sb1), precision_passes=1, **kw)
            else:
                P.gemm(name, A, w[wkey], out, M=M, N=N, K=K * np3, lda=K * m2, seg1=K if acc else 0, precision_passes=np3, **kw)

        # ---- Z1 + Z2: pre-processing fused with the patch gather, patch embedding, cls token
        PK = 3 * c.patch * c.patch
        P.add("preprocess", "bs_preprocess_patches", self.frames, patches, B, H, W, nh_, nw_, int(flip), L.dt(patches) | fmt("pe.w"))
        TOK = ("tokens_grouped" if grouped else "tokens", NB, S, Hd, CP)
        if grouped:
            P.add("cls", "bs_fill_rows", x, w["cls"], NB, 1, Hd)                 # rows 0 .. NB-1
            bgemm("patch_embed", patches, "pe.w", x, NB * T0, Hd, PK, bias=w["pe.b"], out_group=(NB * T0, 0, CP))   # rows CP ..
        else:
            P.add("cls", "bs_fill_rows", x, w["cls"], NB, S, Hd)
            bgemm("patch_embed", patches, "pe.w", x, NB * T0, Hd, PK, bias=w["pe.b"], out_group=(T0, S, 1))
        free(patches)
        P.mark("embed", x, TOK)
        # ---- Z3: BEiT layers
        ti = 0
        for l in range(c.layers):
            P.add(f"l{l}.ln1", "bs_layernorm", x, w[f"l{l}.ln1.g"], w[f"l{l}.ln1.b"], xn, None, MT, Hd, c.ln_eps,
                  L.dt(xn) | fmt(f"l{l}.qkv.w") | prow(f"l{l}.qkv.w"))
            bgemm(f"l{l}.qkv", xn, f"l{l}.qkv.w", q, MT, 3 * Hd, Hd, bias=w[f"l{l}.qkv.b"],
                  qkv=(Hd, S, Sp, LOG2E / math.sqrt(64.0), k, vt, use_tab, NB if grouped else 0, CP, QN if corr else 0))
            if corr:
                P.add(f"l{l}.attn", "bs_attention_table_corr", q, k, vt, q.view(-1)[QN:], k.view(-1)[QN:], vt.view(-1)[QN:], bias[l], ao, NB,
                      c.heads, hp, wp, Sp, CP, L.dt(q) | fmt(f"l{l}.o.w") | (64 if prow(f"l{l}.o.w") else 0))
            elif use_tab:
                P.add(f"l{l}.attn", "bs_attention_table", q, k, vt, bias[l], ao, NB, c.heads, hp, wp, Sp, CP,
                      L.dt(q) | fmt(f"l{l}.o.w") | (64 if prow(f"l{l}.o.w") else 0))
            else:
                P.add(f"l{l}.attn", "bs_attention", q, k, vt, bias[l], ao, NB, c.heads, S, Sp, L.dt(q) | fmt(f"l{l}.o.w"))
            bgemm(f"l{l}.o", ao, f"l{l}.o.w", x, MT, Hd, Hd, bias=w[f"l{l}.o.b"], scale=w[f"l{l}.lam1"], res=x, ldr=Hd)
            P.add(f"l{l}.ln2", "bs_layernorm", x, w[f"l{l}.ln2.g"], w[f"l{l}.ln2.b"], xn, None, MT, Hd, c.ln_eps,
                  L.dt(xn) | fmt(f"l{l}.fc1.w") | prow(f"l{l}.fc1.w"))
            hfmt = fmt(f"l{l}.fc2.w")
            bgemm(f"l{l}.fc1", xn, f"l{l}.fc1.w", hid, MT, c.intermediate, Hd, bias=w[f"l{l}.fc1.b"], act=L.ACT_GELU,
                  ldo=c.intermediate * am(f"l{l}.fc2.w"), out_split_off=c.intermediate if hfmt else 0,
                  out_f8=(L.F8_ACT_HI_EXP, L.F8_ACT_LO_EXP) if hfmt == 32 else None,
                  # fc2 in "wcls" mode reads the lo8 plane of its cls tile only
                  out_lo8_rows=CP if (hfmt == 32 and grouped and eng.mode_of(f"l{l}.fc2.w") in ("wcls", "wmean", "wstat")) else 0,
                  out_planes_rows=CP if prow(f"l{l}.fc2.w") else 0)
            bgemm(f"l{l}.fc2", hid, f"l{l}.fc2.w", x, MT, Hd, c.intermediate, bias=w[f"l{l}.fc2.b"], scale=w[f"l{l}.lam2"], res=x, ldr=Hd)
            P.mark(f"layer{l + 1}", x, TOK)
            if (l + 1) in c.taps:
                if acc:
                    P.add(f"tap{ti}", "bs_cast_split", x, taps16[ti], MT, Hd, L.dt(xn) | (32 if eng.neck_f8 else 0))
                else:
                    P.add(f"tap{ti}", "bs_cast", x, taps16[ti], x.numel(), L.dt(xn))
                ti += 1

        # ---- neck helpers.  Activations are NHWC 16-bit; in accurate mode every tensor is a pair per pixel:
        # (hi16 | hi8 | lo8) with them on the FP8 MFMA (nf8: the whole neck, when every K is whole FP8 stages), or (hi | lo) 16-bit
        # pairs with the product as three K segments.
        nf8 = eng.neck_f8
        NSP = (32 if nf8 else 16) if acc else 0          # format flag of the pointwise producers whose output stays (hi16 | hi8 | lo8)
        RZ = 1 | ((4 if nf8 else 2) if acc else 0)       # bs_resize_bilinear_nhwc flag: align_corners | pair format
        F8O = (L.F8_ACT_HI_EXP, L.F8_ACT_LO_EXP)

        def mfmt(C):
            """format code of a marked neck tensor (tests/test_zoedepth_gpu.py to_nchw): 2 = (hi16 | hi8 | lo8), 1 = (hi | lo)"""
            return (2 if nf8 else 1) if acc else 0

        def f8kw(wkey):
            sb0, sb1 = f8s[wkey]
            wonly = eng.neck_site_wonly(wkey)
            return dict(f8_scales=(127 - L.F8_ACT_HI_EXP, sb0, 127 - L.F8_ACT_LO_EXP, sb1), f8_wonly_from=-1 if wonly else 0,
                        f8_skip_from=-1 if eng.neck_site_plain(wkey) else 0)

        def okw(Cout, out_pairs, out8):
            """output-format arguments of a neck GEMM writing Cout channels per row / pixel"""
            if not (acc and out_pairs):
                return dict(ldo=Cout, out_split_off=0)
            return dict(ldo=Cout * m2, out_split_off=Cout, out_f8=F8O if (nf8 and out8) else None)

        def nplain(name, A, wkey, out, M, N, K, shuffle=None, out_pairs=True, **kw):
            """plain GEMM on pair operands; out_pairs=False leaves the output alone (fp32 or caller-specified)"""
            out8 = kw.pop("out8", True)
            ok = okw(shuffle[1] if shuffle else N, out_pairs, out8)
            if "ldo" in kw:
                ok["ldo"] = kw.pop("ldo")
            if "split_off" in kw:
                ok["out_split_off"] = kw.pop("split_off")
            if not out_pairs:
                ok = dict(ldo=ok["ldo"], out_split_off=0)
            if acc and wkey in f8s:
                self.site_flops[wkey] = self.site_flops.get(wkey, 0.0) + 2.0 * M * N * K      # (calibrate(): what a site's second product is worth)
                lda_ = kw.pop("lda", 2 * K)
                # (calibrate(): the channel means of this product's input, and the static bias correction they give while it runs one pass;
                # a product with a per-group bias -- the readout's token half -- gets its correction through the product that forms that bias)
                P.mark("in:" + wkey, A, ("chanmean", kw.get("a_offset", 0), M, lda_, K))
                if not kw.get("bias_group_rows"):
                    kw["bias"] = eng.site_bias(wkey, kw.get("bias"))
                P.gemm(name, A, w[wkey], out, M=M, N=N, K=K, lda=lda_, f8_seg=2 * K, shuffle=shuffle,
                       precision_passes=1, **ok, **f8kw(wkey), **kw)
            else:
                ok.pop("out_f8", None)
                P.gemm(name, A, w[wkey], out, M=M, N=N, K=K * np3, lda=kw.pop("lda", K * m2), seg1=K if acc else 0,
                       shuffle=shuffle, precision_passes=np3, **ok, **kw)

        def lo8_rows(*consumer_wkeys):
            """out_lo8_rows of a producer whose output is read only by the named products: 256 (the first tile alone writes the lo8 plane) when
            every one of them runs the weight-rounding correction only -- nobody reads that plane then, the epilogue need not form it"""
            return 256 if (acc and nf8 and all(k_ in f8s and eng.neck_site_wonly(k_) for k_ in consumer_wkeys)) else 0

        def hi8_rows(*consumer_wkeys):
            """out_planes_rows of a producer whose output is read only by the named products: 256 (the first tile alone writes its planes) when every
            one of them runs ONE 16-bit pass (the calibration's "plain" sites) -- nobody reads the hi8 plane either"""
            return 256 if (acc and nf8 and all(k_ in f8s and eng.neck_site_plain(k_) for k_ in consumer_wkeys)) else 0

        def nconv(name, A, wkey, out, hh, ww, Ci, Co, stride=1, **kw):
            use8 = acc and wkey in f8s
            g_ = L.conv_geom(hh, ww, Ci if use8 else Ci * m2, 3, 3, stride, 1)
            ho, wo = g_[3], g_[4]
            has_res = "res" in kw
            out_relu = kw.pop("out_relu", None)
            assert out_relu is None or use8
            if use8:
                if out_relu is not None:
                    kw["out_relu"] = out_relu
                self.site_flops[wkey] = self.site_flops.get(wkey, 0.0) + 2.0 * NB * ho * wo * Co * 9 * Ci
                P.mark("in:" + wkey, A, ("chanmean", 0, NB * hh * ww, 2 * Ci, Ci))
                kw["bias"] = eng.site_bias(wkey, kw.get("bias"))
                P.gemm(name, A, w[wkey], out, M=NB * ho * wo, N=Co, K=9 * Ci, lda=2 * Ci, conv=g_, f8_seg=2 * Ci, ldo=2 * Co,
                       ldr=2 * Co if has_res else 0, res_f8=has_res, out_split_off=Co, out_f8=F8O, precision_passes=1, **f8kw(wkey), **kw)
            else:
                if has_res and acc:
                    kw["res_split_off"] = Co
                P.gemm(name, A, w[wkey], out, M=NB * ho * wo, N=Co, K=9 * Ci * np3, lda=Ci * m2, conv=g_, seg1=Ci if acc else 0,
                       ldo=Co * m2, ldr=Co * m2 if has_res else 0, out_split_off=Co if acc else 0, precision_passes=np3, **kw)
            return ho, wo

        # ---- Z4: reassemble (readout project, 1x1 projection, resize) + neck 3x3 convs
        feats, fshape, feats_relu = [], [], []
        # (bs_gemm_desc.out2_relu, see the residual units below: measured a wash -- the seven bs_relu_split launches cost 1.55 ms per step, the second
        # output adds 1.1-1.8 ms to the producing convolutions' epilogues, profiles/r05_plan_call_times.txt -- and left off; BS_RELU_OUT=1 turns it on)
        relu_out = bool(acc and nf8 and os.environ.get("BS_RELU_OUT", "0") == "1")
        cb = e32(NB, Hd)
        r16 = e16(NB * T0, PE(Hd))
        for i, ch in enumerate(c.neck_hidden):
            t16 = taps16[i]
            # cls half of the readout: per-image bias vector  c_b = cls_b @ W_cls^T + b   (A rows = the cls row of every image)
            # (its bias also carries the token half's static correction while that product runs one pass: c_b is added to every token row)
            nplain(f"ro{i}.cls", t16, f"ro{i}.w_cls", cb, NB, Hd, Hd, out_pairs=False, lda=(1 if grouped else S) * PE(Hd),
                   bias=eng.site_bias(f"ro{i}.w_tok", w[f"ro{i}.b"]))
            # token half: the patch rows of every image, + c_b, GELU.  Grouped rows: a plain GEMM over rows NB..; image-major rows:
            # rows 1..S-1 of every image (a 1-row "conv" with a -1 column crop)
            if grouped:
                nplain(f"ro{i}.tok", t16, f"ro{i}.w_tok", r16, NB * T0, Hd, Hd, a_offset=CP * PE(Hd), bias=cb, bias_group_rows=T0,
                       act=L.ACT_GELU)
            elif acc and f"ro{i}.w_tok" in f8s:
                P.gemm(f"ro{i}.tok", t16, w[f"ro{i}.w_tok"], r16, M=NB * T0, N=Hd, K=Hd, lda=2 * Hd, conv=(1, S, Hd, 1, T0, 1, 1, 1, 0, -1),
                       f8_seg=2 * Hd, bias=cb, bias_group_rows=T0, act=L.ACT_GELU, ldo=2 * Hd, out_split_off=Hd, out_f8=F8O if nf8 else None,
                       precision_passes=1, **f8kw(f"ro{i}.w_tok"))
            else:
                P.gemm(f"ro{i}.tok", t16, w[f"ro{i}.w_tok"], r16, M=NB * T0, N=Hd, K=Hd * np3, lda=Hd * m2, conv=(1, S, Hd * m2, 1, T0, 1, 1, 1, 0, -1),
                       seg1=Hd if acc else 0, bias=cb, bias_group_rows=T0, act=L.ACT_GELU, ldo=Hd * m2, out_split_off=Hd if acc else 0,
                       precision_passes=np3)
            pr = e16(NB * T0, PE(ch))
            nplain(f"ra{i}.proj", r16, f"ra{i}.proj.w", pr, NB * T0, ch, Hd, bias=w[f"ra{i}.proj.b"])
            if i == 0 or i == 1:
                s_ = 4 if i == 0 else 2
                up = e16(NB, hp * s_, wp * s_, PE(ch))
                nplain(f"ra{i}.up", pr, f"ra{i}.up.w", up, NB * T0, s_ * s_ * ch, ch, shuffle=(s_, ch, hp, wp), bias=w[f"ra{i}.up.b"])
                free(pr)
                fh, fw, src = hp * s_, wp * s_, up
            elif i == 2:
                fh, fw, src = hp, wp, pr
            else:
                fh, fw = (hp + 2 - 3) // 2 + 1, (wp + 2 - 3) // 2 + 1
                src = e16(NB, fh, fw, PE(ch))
                nconv(f"ra{i}.down", pr, f"ra{i}.down.w", src, hp, wp, ch, ch, stride=2, bias=w[f"ra{i}.down.b"])
                free(pr)
            P.mark(f"reassemble{i}", src, ("nhwc", NB, fh, fw, ch, mfmt(ch)))
            f16_ = e16(NB, fh, fw, PE(c.fusion))
            # Round 5: the fusion stage's residual units read x (the skip) AND relu(x) (their first convolution's input): the producing
            # convolution's epilogue can write both (bs_gemm_desc.out2_relu) instead of a bs_relu_split launch re-reading x (off by default, see relu_out).
            fr_ = e16(NB, fh, fw, PE(c.fusion)) if relu_out else None
            nconv(f"nc{i}", src, f"nc{i}.w", f16_, fh, fw, ch, c.fusion, out_relu=fr_)
            feats_relu.append(fr_)
            free(src)                                                  # (level 2: src is pr)
            P.mark(f"neckconv{i}", f16_, ("nhwc", NB, fh, fw, c.fusion, mfmt(c.fusion)))
            feats.append(f16_)
            fshape.append((fh, fw))
        free(r16, cb)
        bott, (bh_, bw_) = feats[3], fshape[3]
        # The router and the seed regressors depend only on the bottleneck map.  Rounds 2-4 issued them on a side stream beside the fusion stage;
        # measured in round 5 (bench, alternating runs: 413.0 / 411.9 frames/s with the side lane, 413.1 / 406.0 without) that lane hides nothing
        # that can be seen -- the launches fill the chip -- and it is gone: one lane, and these intermediates are pooled like all others.
        # ---- Z7: metric-bins head
        Mb = NB * bh_ * bw_
        xb = e16(Mb, c.bottleneck * m2)                                   # (hi | lo) pairs in accurate mode
        nplain("mh.conv2", bott, "mh.conv2.w", xb, Mb, c.bottleneck, c.bottleneck, bias=w["mh.conv2.b"], out8=False)
        self.logits = torch.empty(NB, 4, device=dev, dtype=torch.float32)      # (a plan output: not pooled)
        self.route = torch.zeros(NB, dtype=torch.int32, device=dev)       # single-head models: every image stays on slot 0
        if not c.single_head:
            # router: 4-layer post-norm transformer over (1 + bh*bw) tokens, classifier on token 0
            D, St = c.pt_hidden, bh_ * bw_ + 1
            pos = torch.arange(0, St, dtype=torch.float32).unsqueeze(1)
            div = torch.exp(torch.arange(0, D, 2, dtype=torch.float32).unsqueeze(0) * (-torch.log(torch.full((), 10000.0)) / D))
            pe_tab = torch.cat([torch.sin(pos * div), torch.cos(pos * div)], dim=1).to(dev)          # [St, D]
            self._pe_src = pe_tab.unsqueeze(0).expand(NB, St, D).contiguous().view(NB * St, D)
            e32b = e32(NB * St, D)
            e16b = e16(NB * St, D)
            self._router_init = (e32b, self._pe_src)
            # e = pos_enc (token 0 is the zero "cls" pad) ; tokens 1.. += embedding conv
            P.add("rt.init", "bs_copy_f32", self._pe_src, e32b, e32b.numel())
            P.gemm("rt.emb", xb, w["rt.emb.w"], e32b, M=Mb, N=D, K=c.bottleneck, lda=c.bottleneck * m2, bias=w["rt.emb.b"], res=e32b, ldr=D,
                   out_group=(bh_ * bw_, St, 1))
            P.add("rt.cast", "bs_cast", e32b, e16b, e32b.numel(), L.dt(e16b))
            qkv32 = e32(NB * St, 3 * D)
            at16 = e16(NB * St, D)
            tmp32 = e32(NB * St, D)
            h16 = e16(NB * St, c.pt_inter)
            for l in range(c.pt_layers):
                P.gemm(f"rt{l}.qkv", e16b, w[f"rt{l}.qkv.w"], qkv32, M=NB * St, N=3 * D, K=D, lda=D, bias=w[f"rt{l}.qkv.b"])
                P.add(f"rt{l}.attn", "bs_small_attention", qkv32, at16, NB, St, c.pt_heads, L.dt(at16))
                P.gemm(f"rt{l}.o", at16, w[f"rt{l}.o.w"], tmp32, M=NB * St, N=D, K=D, lda=D, bias=w[f"rt{l}.o.b"], res=e32b, ldr=D)
                P.add(f"rt{l}.n1", "bs_layernorm", tmp32, w[f"rt{l}.n1.g"], w[f"rt{l}.n1.b"], e16b, e32b, NB * St, D, 1e-5, L.dt(e16b))
                P.gemm(f"rt{l}.l1", e16b, w[f"rt{l}.l1.w"], h16, M=NB * St, N=c.pt_inter, K=D, lda=D, bias=w[f"rt{l}.l1.b"], act=L.ACT_RELU)
                P.gemm(f"rt{l}.l2", h16, w[f"rt{l}.l2.w"], tmp32, M=NB * St, N=D, K=c.pt_inter, lda=c.pt_inter, bias=w[f"rt{l}.l2.b"], res=e32b, ldr=D)
                P.add(f"rt{l}.n2", "bs_layernorm", tmp32, w[f"rt{l}.n2.g"], w[f"rt{l}.n2.b"], e16b, e32b, NB * St, D, 1e-5, L.dt(e16b))
            c1 = e16(NB, D)
            P.gemm("cl.l1", e16b, w["cl.l1.w"], c1, M=NB, N=D, K=D, lda=St * D, bias=w["cl.l1.b"], act=L.ACT_RELU)
            P.gemm("cl.l2", c1, w["cl.l2.w"], self.logits, M=NB, N=4, K=D, lda=D, bias=w["cl.l2.b"])
            P.add("route", "bs_route_argmax", self.logits, 4, self.route, NB)
            P.mark("logits", self.logits, ("raw",))
        # seeds + seed projector
        E, nb = c.bin_dim, c.n_bins
        SM, PM, HID = c.seed_mlp, c.proj_mlp, c.clb_hidden          # hidden widths: 64 / 64 / 40 (NK head), 256 / 128 / 80 (single head)
        KB = c.bottleneck
        sh = e16(Mb, 2 * SM * m2)                                  # [seed regressor slot 0 | slot 1] hidden units, pairs in accurate mode
        P.gemm("seed.c1", xb, w["seed.c1.w"], sh, M=Mb, N=2 * SM, K=KB * np3, lda=KB * m2, seg1=KB if acc else 0, bias=w["seed.c1.b"], act=L.ACT_RELU,
               ldo=2 * SM * m2, out_split_off=2 * SM if acc else 0, precision_passes=np3)
        bins_prev = e32(NB, bh_, bw_, 2 * nb)
        P.gemm("seed.c2", sh, w["seed.c2.w"], bins_prev, M=Mb, N=2 * nb, K=2 * SM * np3, lda=2 * SM * m2, seg1=2 * SM if acc else 0, bias=w["seed.c2.b"],
               act=L.ACT_SOFTPLUS, precision_passes=np3)
        # seed projector: its own small GEMM pair (1.3e-6 m as single products: stays single); the projector embeddings feed the last
        # 1x1 convs of the head almost directly, so accurate mode keeps them as (hi | lo) pairs
        shp = e16(Mb, PM)
        P.gemm("seedproj.c1", xb, w["seedproj.c1.w"], shp, M=Mb, N=PM, K=KB, lda=KB * m2, bias=w["seedproj.c1.b"], act=L.ACT_RELU)
        emb_prev = e16(Mb, E * m2)
        P.gemm("seedproj.c2", shp, w["seedproj.c2.w"], emb_prev, M=Mb, N=E, K=PM, lda=PM, bias=w["seedproj.c2.b"],
               ldo=E * m2, out_split_off=E if acc else 0)
        # ---- Z5: fusion stage (pre-activation residual units, x2 bilinear, 1x1 projection)
        Fc = c.fusion

        def res_unit(name, xin, hh, ww, other=None, xin_relu=None, want_relu=False):
            """y = conv2(relu(conv1(relu(x)))) + x (+ other).  xin_relu: relu(x) as its producer wrote it (else a bs_relu_split launch forms it);
            want_relu: also return relu(y), written by conv2's epilogue, for the next unit."""
            t = e16(NB, hh, ww, PE(Fc))
            y = e16(NB, hh, ww, PE(Fc))
            yr = e16(NB, hh, ww, PE(Fc)) if (want_relu and relu_out) else None
            if acc:
                xr = xin_relu
                if xr is None:
                    xr = e16(NB, hh, ww, PE(Fc))
                    # (bit 6: no lo8 plane when the first convolution, the only reader, is weight-only)
                    P.add(name + ".relu", "bs_relu_split", xin, xr, NB * hh * ww, Fc, L.dt(xr) | (32 if nf8 else 0) | (64 if lo8_rows(name + ".c1.w") else 0) | (128 if hi8_rows(name + ".c1.w") else 0))
                nconv(name + ".c1", xr, name + ".c1.w", t, hh, ww, Fc, Fc, bias=w[name + ".c1.b"], act=L.ACT_RELU, out_lo8_rows=lo8_rows(name + ".c2.w"),
                      out_planes_rows=hi8_rows(name + ".c2.w"))
                free(xr)
            else:
                nconv(name + ".c1", xin, name + ".c1.w", t, hh, ww, Fc, Fc, relu_a=True, bias=w[name + ".c1.b"], act=L.ACT_RELU)
            nconv(name + ".c2", t, name + ".c2.w", y, hh, ww, Fc, Fc, bias=w[name + ".c2.b"], res=xin, res2=other, out_relu=yr)
            free(t)
            return y, yr

        # Round 5: the bins head's projectors (pj{i}.c1: 1x1 conv 256 -> 64 + ReLU on the fused maps, HF modeling_zoedepth.py:749-772) read an
        # UPSAMPLED map too: their convolution runs on the low-resolution projection output (a quarter of the pixels; the product is kept as
        # (hi | lo) pairs) and bs_resize_bias_relu_nhwc upsamples it, adds the bias and applies the ReLU where the level needs it -- instead of
        # reading the 256-channel fused map at full resolution (pj3.c1: 6.4 GB in for 1.6 GB out, 1.36 ms).  BS_PJ_LOWRES=0: the direct form.
        pj_lowres = os.environ.get("BS_PJ_LOWRES", "1") != "0"
        pj_low = []
        fused_list = []
        fused = None
        for li in range(4):
            feat = feats[3 - li]
            fh, fw = fshape[3 - li]
            if fused is None:
                cur, cur_relu = feat, feats_relu[3 - li]                    # (the bottleneck map)
                own = False
            else:
                cur, cur_relu = res_unit(f"fu{li}.r1", feat, fh, fw, other=fused, xin_relu=feats_relu[3 - li], want_relu=True)     # fused + residual_layer1(feat)
                free(feat)
                if pj_lowres:
                    free(fused)                                             # (its projector input was taken at the low resolution)
                own = True
            cur_in = cur
            cur, _ = res_unit(f"fu{li}.r2", cur, fh, fw, xin_relu=cur_relu)
            if own:
                free(cur_in)
            # HF upsamples, then applies the 1x1 projection (modeling_zoedepth.py:316-322).  Both are linear and the bilinear weights
            # sum to 1, so projection(interpolate(x)) = interpolate(projection(x)) exactly in real arithmetic: the projection runs at
            # the LOW resolution (a quarter of the FLOPs and of the bytes), the resize writes the fused map directly.
            lowp = e16(NB, fh, fw, PE(Fc))
            nplain(f"fu{li}.proj", cur, f"fu{li}.proj.w", lowp, NB * fh * fw, Fc, Fc, bias=w[f"fu{li}.proj.b"])
            free(cur)
            if pj_lowres:
                z = e16(NB * fh * fw, PM * m2)
                nplain(f"pj{li}.c1", lowp, f"pj{li}.c1.w", z, NB * fh * fw, PM, Fc, out8=False)
                pj_low.append((z, fh, fw))
            fused = e16(NB, 2 * fh, 2 * fw, PE(Fc))
            # (the last fused map is read by the relative head's first convolution only: when that product is weight-only the resize does not
            # form the lo8 plane -- flag bit 3 -- and the tap's format code says so: 3 = (hi16 | hi8 | -))
            nolo = bool(li == 3 and pj_lowres and lo8_rows("rh.projection.w" if eng.add_projection else "rh.conv1.w"))
            nopl = bool(nolo and hi8_rows("rh.projection.w" if eng.add_projection else "rh.conv1.w"))
            P.add(f"fu{li}.up", "bs_resize_bilinear_nhwc", lowp, fused, NB, fh, fw, Fc, 2 * fh, 2 * fw, RZ | (8 if nolo else 0) | (16 if nopl else 0), L.dt(fused))
            free(lowp)
            P.mark(f"fused{li}", fused, ("nhwc", NB, 2 * fh, 2 * fw, Fc, 3 if nolo else mfmt(Fc)))
            fused_list.append((fused, 2 * fh, 2 * fw))
        # ---- Z6: relative head (conv3 + ReLU -> relative depth is dead code for the NK output and not launched)
        f3, h3, w3 = fused_list[3]
        if eng.add_projection:
            rp = e16(NB, h3, w3, PE(Fc))
            nconv("rh.projection", f3, "rh.projection.w", rp, h3, w3, Fc, Fc, bias=w["rh.projection.b"], act=L.ACT_RELU, out_lo8_rows=lo8_rows("rh.conv1.w"))
        else:
            rp = f3
        if eng.add_projection and pj_lowres:
            free(f3)
        r1 = e16(NB, h3, w3, (Fc // 2) * m2)
        nconv("rh.conv1", rp, "rh.conv1.w", r1, h3, w3, Fc, Fc // 2, bias=w["rh.conv1.b"], out_lo8_rows=lo8_rows("rh.conv2.w"))
        if eng.add_projection or pj_lowres:
            free(rp)
        # HF: interpolate x2 (align_corners), conv2 3x3 128 -> 32, ReLU (modeling_zoedepth.py:358-362).  Both are linear and the resize acts
        # per channel, so conv2(up(x))(p) = sum_tap up(W_tap x)(p + d_tap): the nine 1x1 tap products run as ONE plain GEMM at the low
        # resolution (N = 9 * 32, a quarter of the conv's FLOPs; the N = 32 conv ran at 20 % of the MFMA peak, bound by the LDS fill
        # rate) and bs_upconv_tapsum gathers / interpolates / sums them -- the upsampled map is never materialised.
        # Round 5: one launch (bs_upconv_fused, csrc/upconv_fused.hip) -- the tap products of a 16 x 16 output tile's low-resolution window are
        # formed by MFMA into LDS and interpolated from there, the 7.2 GB fp32 tap-product tensor of the two-launch path (bs_gemm +
        # bs_upconv_tapsum, kept for the (hi | lo) pair formats and for A / B runs: BS_UPCONV_FUSED=0) never exists.
        fused_up = (os.environ.get("BS_UPCONV_FUSED", "1") != "0" and Fc // 2 == 128 and c.rel_features == 32
                    and ((not acc) or (nf8 and "rh.conv2.w" in f8s)))
        last = None
        if fused_up:
            last = e16(NB, 2 * h3, 2 * w3, c.rel_features * m2)
            if acc:
                sb0, sb1 = f8s["rh.conv2.w"]
                umode, usc = (1 if eng.neck_site_wonly("rh.conv2.w") else 2), (127 - L.F8_ACT_HI_EXP, sb0, 127 - L.F8_ACT_LO_EXP, sb1)
                self.site_flops["rh.conv2.w"] = 2.0 * NB * h3 * w3 * 9 * c.rel_features * (Fc // 2)
            else:
                umode, usc = 0, (127, 127, 127, 127)
            P.add("rh.conv2", "bs_upconv_fused", r1, w["rh.conv2.w"], w["rh.conv2.b"], last, NB, h3, w3, Fc // 2, c.rel_features, 2 * h3, 2 * w3,
                  RZ, 1, umode, *usc, L.dt(last))
            # the reference's product: conv 3x3 (Fc/2 -> rel_features) at the UPSAMPLED resolution; what runs: nine tap products at the low one,
            # plus their FP8 correction stage(s)
            f_low = 2.0 * NB * h3 * w3 * 9 * c.rel_features * (Fc // 2)
            P.tag_stack(4.0 * f_low, f_low * (1.0, 1.5, 2.0)[umode])       # (executed: bs_gemm's convention, an FP8 stage = half a pass)
            free(r1)
        else:
            y9 = e32(NB, h3, w3, 9 * c.rel_features)
            # (K = 128: a block's main loop is four K steps, the launch is bound by block turnover -- the 128x64 tile, 3 blocks per CU,
            # takes 3.8 ms where the 128x128 one takes 5.9, tools/probes/rh_conv2_tiles.py)
            nplain("rh.conv2", r1, "rh.conv2.w", y9, NB * h3 * w3, 9 * c.rel_features, Fc // 2, out_pairs=False, tile=2)
            free(r1)
            last = e16(NB, 2 * h3, 2 * w3, c.rel_features * m2)
            P.add("rh.tapsum", "bs_upconv_tapsum", y9, w["rh.conv2.b"], last, NB, h3, w3, c.rel_features, 2 * h3, 2 * w3, RZ, 1, L.dt(last))
            free(y9)
        P.mark("rel_features", last, ("nhwc", NB, 2 * h3, 2 * w3, c.rel_features, (2 if nf8 else 1) if acc else 0))
        # ---- Z7 (continued): projector / attractor levels on the fusion outputs
        ph_, pw_ = bh_, bw_
        clb_composed = os.environ.get("BS_CLB_COMPOSED", "1") != "0"       # (A / B switch: 0 = the embedding product on the 128-channel embedding)
        # Round 6: the level's projector path in ONE launch (csrc/projector.hip, bs_projector_level): e1 = relu(resize(z) + b), emb = W_c2 e1 + b,
        # x = round16(emb + resize(emb_prev)) and -- at the last level -- the composed log-binomial embedding product, with e1 and emb never
        # in memory (rounds 2-5: bs_resize_bias_relu_nhwc + a 3-pass bs_gemm + the sum inside bs_mlp2_add + another 3-pass bs_gemm: 2 368 B
        # through HBM per finest-level pixel where this moves 640).  BS_PROJECTOR_LEVEL=0: the four launches (A / B, and the path of every
        # geometry / format the kernel is not built for: rows that are no multiple of 32 pixels, single 16-bit operands).
        proj_level = (os.environ.get("BS_PROJECTOR_LEVEL", "1") != "0" and acc and pj_lowres and clb_composed and PM == 64 and E == 128
                      and 2 * HID <= 80 and (2 * HID) % 16 == 0 and eng.fuse_mlp
                      and all(fw_ % 32 == 0 and 2 * zw_ == fw_ and 2 * zh_ == fh_ for (_, fh_, fw_), (_, zh_, zw_) in zip(fused_list, pj_low))
                      and all(tuple(w[f"at{i_}.c1.w"].shape) == (256, 128) and 2 * eng.na_eff[i_] <= 32 for i_ in range(4)))
        Eh = None
        for i in range(4):
            feat, fh, fw = fused_list[i]
            Mi = NB * fh * fw
            if proj_level:
                z, zh, zw = pj_low[i]
                assert (zh, zw) == (ph_, pw_), "the projector's low-resolution map and the previous level's embedding share a grid"
                last_lv = i == 3
                x16 = e16(Mi, E)
                emb = None if last_lv else e16(Mi, E * m2)
                if last_lv:
                    Eh = e32(Mi, 2 * HID)
                P.add(f"pj{i}.level", "bs_projector_level", z, w[f"pj{i}.c1.b"], emb_prev, w[f"pj{i}.c2.w"], w[f"pj{i}.c2.b"],
                      w["clb.e1.w"] if last_lv else None, w["clb.e1.b"] if last_lv else None, x16, emb, Eh, NB, zh, zw, fh, fw, PM, E, 2 * HID,
                      L.dt(x16))
                f_alg = 2.0 * Mi * PM * (E + (2 * HID if last_lv else 0))
                P.tag_stack(f_alg, 3.0 * f_alg)                          # (three 16-bit passes on (hi | lo) pairs, as the bs_gemm launches it replaces)
                free(z, emb_prev)
                na = eng.na_eff[i]
                A = e32(Mi, 2 * na)
                P.add(f"at{i}.mlp", "bs_mlp2", x16, E, w[f"at{i}.c1.w"], w[f"at{i}.c1.b"], w[f"at{i}.c2.w"], w[f"at{i}.c2.b"], A, Mi, E, 2 * E,
                      2 * na, L.ACT_SOFTPLUS_FAST, L.dt(x16))
                P.tag_stack(2.0 * Mi * (E * 2 * E + 2 * E * 2 * na), 2.0 * Mi * (E * 2 * E + 2 * E * 2 * na))      # the attractor's two 1x1 convolutions
                free(x16)
                bins = e32(NB, fh, fw, 2 * nb)
                P.add(f"at{i}.step", "bs_attractor_step", A, bins_prev, bins, self.route, NB, ph_, pw_, fh, fw, 2, nb, na)
                free(A, bins_prev)
                P.mark(f"bins{i}", bins, ("nhwc_route", NB, fh, fw, 2 * nb))
                bins_prev, emb_prev, ph_, pw_ = bins, emb, fh, fw
                continue
            e1 = e16(Mi, PM * m2)
            if pj_lowres:
                z, zh, zw = pj_low[i]
                P.add(f"pj{i}.up", "bs_resize_bias_relu_nhwc", z, w[f"pj{i}.c1.b"], e1, NB, zh, zw, PM, fh, fw, 1 | (2 if acc else 0), L.dt(e1))
                free(z)
            else:
                nplain(f"pj{i}.c1", feat, f"pj{i}.c1.w", e1, Mi, PM, Fc, bias=w[f"pj{i}.c1.b"], act=L.ACT_RELU, out8=False)
                free(feat)                                         # the fused map's last reader (fused 3 fed the relative head earlier)
            emb = e16(Mi, E * m2)
            P.gemm(f"pj{i}.c2", e1, w[f"pj{i}.c2.w"], emb, M=Mi, N=E, K=PM * np3, lda=PM * m2, seg1=PM if acc else 0,
                   ldo=E * m2, out_split_off=E if acc else 0, bias=w[f"pj{i}.c2.b"], precision_passes=np3)
            if i == 3 and clb_composed:
                e1_last = e1                                       # (read once more by the composed log-binomial embedding product below)
            else:
                free(e1)
            na = eng.na_eff[i]                                     # attractors of this level (replicated up to a multiple of 4)
            A = e32(Mi, 2 * na)
            fuse = E == 128 and 2 * na <= 32 and eng.fuse_mlp and tuple(w[f"at{i}.c1.w"].shape) == (256, 128)
            if fuse and eng.fuse_mlp > 1:
                # the level in ONE launch: emb + resize(emb_prev) is formed inside the MLP kernel (its hi half is all the MLP reads) and
                # never stored -- bit-identical to bs_add_resized + bs_mlp2 (csrc/mlp2.hip, FUSE)
                P.add(f"at{i}.mlp", "bs_mlp2_add", emb, emb_prev, w[f"at{i}.c1.w"], w[f"at{i}.c1.b"], w[f"at{i}.c2.w"], w[f"at{i}.c2.b"], A, NB, ph_,
                      pw_, fh, fw, E, 2 * E, 2 * na, L.ACT_SOFTPLUS_FAST, L.dt(emb) | (16 if acc else 0))
                P.tag_stack(2.0 * Mi * (E * 2 * E + 2 * E * 2 * na), 2.0 * Mi * (E * 2 * E + 2 * E * 2 * na))      # the attractor's two 1x1 convolutions
                free(emb_prev)
                y = None
            else:
                y = e16(Mi, E * m2)
                P.add(f"at{i}.add", "bs_add_resized", emb, emb_prev, y, NB, ph_, pw_, fh, fw, E, L.dt(y) | (16 if acc else 0))
                free(emb_prev)
            if y is None:
                pass
            elif fuse:
                # both 1x1 convolutions in one launch: the 256-channel hidden map (3.2 GB at the finest level) never reaches memory;
                # bit-identical to the two launches below (bs_mlp2, csrc/mlp2.hip)
                P.add(f"at{i}.mlp", "bs_mlp2", y, E * m2, w[f"at{i}.c1.w"], w[f"at{i}.c1.b"], w[f"at{i}.c2.w"], w[f"at{i}.c2.b"], A, Mi, E, 2 * E,
                      2 * na, L.ACT_SOFTPLUS_FAST, L.dt(y))
                P.tag_stack(2.0 * Mi * (E * 2 * E + 2 * E * 2 * na), 2.0 * Mi * (E * 2 * E + 2 * E * 2 * na))
                free(y)
            else:
                a1 = e16(Mi, 2 * E)
                P.gemm(f"at{i}.c1", y, w[f"at{i}.c1.w"], a1, M=Mi, N=2 * E, K=E, lda=E * m2, bias=w[f"at{i}.c1.b"], act=L.ACT_RELU)
                free(y)
                P.gemm(f"at{i}.c2", a1, w[f"at{i}.c2.w"], A, M=Mi, N=2 * na, K=2 * E, lda=2 * E, bias=w[f"at{i}.c2.b"], act=L.ACT_SOFTPLUS_FAST)
                free(a1)
            bins = e32(NB, fh, fw, 2 * nb)
            P.add(f"at{i}.step", "bs_attractor_step", A, bins_prev, bins, self.route, NB, ph_, pw_, fh, fw, 2, nb, na)
            free(A, bins_prev)
            P.mark(f"bins{i}", bins, ("nhwc_route", NB, fh, fw, 2 * nb))
            bins_prev, emb_prev, ph_, pw_ = bins, emb, fh, fw
        if proj_level:
            pass                                                   # (Eh came out of the last level's launch)
        elif clb_composed:
            Eh = e32(NB * ph_ * pw_, 2 * HID)
            P.gemm("clb.emb", e1_last, w["clb.e1.w"], Eh, M=NB * ph_ * pw_, N=2 * HID, K=PM * np3, lda=PM * m2, seg1=PM if acc else 0, bias=w["clb.e1.b"],
                   precision_passes=np3)
            free(e1_last)
        else:
            Eh = e32(NB * ph_ * pw_, 2 * HID)
            P.gemm("clb.emb", emb_prev, w["clb.emb.w"], Eh, M=NB * ph_ * pw_, N=2 * HID, K=E * np3, lda=E * m2, seg1=E if acc else 0, bias=w["clb.emb.b"],
                   precision_passes=np3)
        free(emb_prev)
        self.depth_net = torch.empty(NB, nh_, nw_, device=dev, dtype=torch.float32)        # plan outputs are not pooled
        assert (nh_, nw_) == (2 * h3, 2 * w3)
        P.mark("clb_eh", Eh, ("raw",))
        P.add("logbinom", "bs_logbinom_depth_ex", last, Eh, bins_prev, w["clb.w0_last"], w["clb.w2"], w["clb.b2"], w.get("clb.rel"), HID,
              self.route, self.depth_net, NB, nh_, nw_, ph_, pw_, c.min_temp, c.max_temp, L.dt(last) | NSP)
        P.mark("depth_net", self.depth_net, ("raw",))
        # ---- Z8: flip average + bicubic + crop + x256 -> uint16
        free(last, Eh, bins_prev)
        self.depth_m = torch.empty(B, H, W, device=dev, dtype=torch.float32)
        self.depth_u16 = torch.empty(B, H, W, device=dev, dtype=torch.int16)   # uint16 payload
        P.add("postprocess", "bs_postprocess_depth", self.depth_net, self.depth_m, self.depth_u16, B, H, W, nh_, nw_, int(flip))

    def run(self, taps: Optional[dict] = None):
        self.plan.run(taps)
