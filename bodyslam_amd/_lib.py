"""ctypes binding of libbodyslam_hip.so (include/bodyslam_hip.h).

PyTorch is plumbing here: it owns device memory and the stream; every compute call below hands raw
device pointers to the C ABI.  There is NO fallback: if the shared library is missing or a call
fails, an exception is raised (BodySlamHipError) -- the product path never silently computes on
the CPU or through torch ops.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BODYSLAM_HIP_LIB") or os.path.join(_HERE, "libbodyslam_hip.so")      # (the override: A/B runs of two builds in one call)

F32, F16, BF16 = 0, 1, 2
ACT_NONE, ACT_RELU, ACT_GELU, ACT_SOFTPLUS, ACT_SOFTPLUS_FAST = 0, 1, 2, 3, 4
OUT_PLAIN, OUT_SHUFFLE, OUT_QKV = 0, 1, 2

_TORCH2BS = {torch.float32: F32, torch.float16: F16, torch.bfloat16: BF16}
BS2TORCH = {F32: torch.float32, F16: torch.float16, BF16: torch.bfloat16}


class BodySlamHipError(RuntimeError):
    pass


class GemmDesc(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("W", C.c_void_p), ("dtype", C.c_int32),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("lda", C.c_int32), ("conv", C.c_int32),
        ("Hin", C.c_int32), ("Win", C.c_int32), ("Cin", C.c_int32), ("Hout", C.c_int32), ("Wout", C.c_int32),
        ("KH", C.c_int32), ("KW", C.c_int32), ("stride", C.c_int32), ("pad_h", C.c_int32), ("pad_w", C.c_int32),
        ("relu_a", C.c_int32),
        ("bias", C.c_void_p), ("bias_group_rows", C.c_int32), ("act", C.c_int32),
        ("scale", C.c_void_p), ("res", C.c_void_p), ("res_dtype", C.c_int32), ("ldr", C.c_int32), ("res2", C.c_void_p),
        ("out", C.c_void_p), ("out2", C.c_void_p), ("out3", C.c_void_p),
        ("out_dtype", C.c_int32), ("ldo", C.c_int32), ("out_mode", C.c_int32),
        ("out_group_rows", C.c_int32), ("out_group_stride", C.c_int32), ("out_row_offset", C.c_int32),
        ("shuffle_s", C.c_int32), ("shuffle_cout", C.c_int32),
        ("qkv_hidden", C.c_int32), ("qkv_tokens", C.c_int32), ("qkv_sp", C.c_int32), ("q_scale", C.c_float),
        ("tile", C.c_int32), ("seg1", C.c_int32), ("out_split_off", C.c_int32), ("res_split_off", C.c_int32),
        ("f8_seg", C.c_int32), ("f8_scales", C.c_uint32), ("out_f8", C.c_int32), ("res_f8", C.c_int32),
        ("qkv_cls_last", C.c_int32), ("qkv_cls_rows", C.c_int32), ("qkv_patch_row0", C.c_int32), ("f8_wonly_from", C.c_int32), ("out_lo8_rows", C.c_int32),
        ("f8_skip_from", C.c_int32), ("bias2_row0", C.c_int32), ("bias2_group_rows", C.c_int32), ("out_planes_rows", C.c_int32),
        ("bias2", C.c_void_p),
        ("qkv_lo_off", C.c_int32), ("out2_relu", C.c_int32),
    ]


_lib: Optional[C.CDLL] = None

_SIGS = {
    "bs_init": [C.c_int],
    "bs_version": [],
    "bs_gemm": [C.POINTER(GemmDesc), C.c_void_p],
    "bs_gemm_tile": [C.POINTER(GemmDesc)],
    "bs_attention": [C.c_void_p] * 5 + [C.c_int32] * 5 + [C.c_void_p],
    "bs_attention_table": [C.c_void_p] * 5 + [C.c_int32] * 7 + [C.c_void_p],
    "bs_attention_table_corr": [C.c_void_p] * 8 + [C.c_int32] * 7 + [C.c_void_p],
    "bs_layernorm": [C.c_void_p] * 5 + [C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_void_p],
    "bs_cast": [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p],
    "bs_copy_f32": [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p],
    "bs_cast_split": [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p],
    "bs_relu_split": [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p],
    "bs_preprocess_patches": [C.c_void_p, C.c_void_p] + [C.c_int32] * 7 + [C.c_void_p],
    "bs_preprocess_image": [C.c_void_p, C.c_void_p] + [C.c_int32] * 6 + [C.c_void_p],
    "bs_fill_rows": [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p],
    "bs_resize_bilinear_nhwc": [C.c_void_p, C.c_void_p] + [C.c_int32] * 8 + [C.c_void_p],
    "bs_upconv_tapsum": [C.c_void_p] * 3 + [C.c_int32] * 9 + [C.c_void_p],
    "bs_upconv_fused": [C.c_void_p] * 4 + [C.c_int32] * 15 + [C.c_void_p],
    "bs_resize_bias_relu_nhwc": [C.c_void_p] * 3 + [C.c_int32] * 8 + [C.c_void_p],
    "bs_col_mean": [C.c_void_p, C.c_int64] + [C.c_int32] * 5 + [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p],
    "bs_rank1_bias": [C.c_void_p] * 3 + [C.c_int32] * 3 + [C.c_void_p],
    "bs_depth_u16_to_m": [C.c_void_p, C.c_int64, C.c_double, C.c_double, C.c_void_p, C.c_void_p],
    "bs_engine_load": [C.c_char_p, C.c_void_p],
    "bs_engine_destroy": [C.c_void_p],
    "bs_engine_io": [C.c_void_p, C.c_char_p, C.c_void_p, C.c_void_p],
    "bs_engine_device_bytes": [C.c_void_p],
    "bs_engine_run": [C.c_void_p, C.c_void_p],
    "bs_engine_upload": [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64],
    "bs_engine_download": [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64],
    "bs_zoedepth_forward": [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p],
    "bs_cyclepose_forward": [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p],
    "bs_tsdf_frames_upload": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p],
    "bs_tsdf_touch_batch": [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                            C.c_void_p],
    "bs_tsdf_integrate_batch": [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p,
                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_void_p],
    "bs_odo_prepare": [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p],
    "bs_odo_pyrdown": [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_double, C.c_void_p],
    "bs_odo_sobel": [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p],
    "bs_odo_accumulate": [C.c_void_p] * 8 + [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_void_p,
                                              C.c_int32, C.c_void_p],
    "bs_odo_step": [C.c_void_p] * 8 + [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_void_p,
                                        C.c_void_p, C.c_int32, C.c_void_p],
    "bs_tsdf_touch": [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p,
                      C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p],
    "bs_tsdf_integrate": [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                          C.c_int32, C.c_double, C.c_double, C.c_void_p, C.c_void_p],
    "bs_tsdf_extract": [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_void_p, C.c_void_p,
                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    "bs_tsdf_mesh": [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_void_p, C.c_int32] + [C.c_void_p] * 7,
    "bs_attractor_step": [C.c_void_p] * 4 + [C.c_int32] * 8 + [C.c_void_p],
    "bs_add_resized": [C.c_void_p] * 3 + [C.c_int32] * 7 + [C.c_void_p],
    "bs_mlp2": [C.c_void_p, C.c_int32] + [C.c_void_p] * 5 + [C.c_int32] * 6 + [C.c_void_p],
    "bs_mlp2_add": [C.c_void_p] * 7 + [C.c_int32] * 10 + [C.c_void_p],
    "bs_projector_level": [C.c_void_p] * 10 + [C.c_int32] * 9 + [C.c_void_p],
    "bs_logbinom_depth": [C.c_void_p] * 8 + [C.c_int32] * 5 + [C.c_float, C.c_float, C.c_int32, C.c_void_p],
    "bs_logbinom_depth_ex": [C.c_void_p] * 7 + [C.c_int32] + [C.c_void_p] * 2 + [C.c_int32] * 5 + [C.c_float, C.c_float, C.c_int32, C.c_void_p],
    "bs_small_attention": [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p],
    "bs_route_argmax": [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p],
    "bs_postprocess_depth": [C.c_void_p] * 3 + [C.c_int32] * 6 + [C.c_void_p],
    "bs_cyclepose_im2col": [C.c_void_p] * 3 + [C.c_int32] * 4 + [C.c_void_p],
    "bs_cyclepose_im2col_window": [C.c_void_p] * 3 + [C.c_int32] * 8 + [C.c_void_p],
    "bs_instnorm_relu_nhwc": [C.c_void_p] * 4 + [C.c_int32] * 3 + [C.c_float, C.c_int32, C.c_void_p],
    "bs_avgpool_nhwc": [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p],
    "bs_cyclepose_head": [C.c_void_p] * 12 + [C.c_int32] * 3 + [C.c_void_p],
    "bs_backproject": [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_double, C.c_double] + [C.c_void_p] * 6,
    "bs_pose_chain": [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p],
    "bs_pose_chain_from": [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p],
    "bs_pixel_to_3d": [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p],
}
EXPORTS = sorted(list(_SIGS) + ["bs_last_error"])


def load_library() -> C.CDLL:
    """dlopen the in-tree library and declare every prototype.  No GPU is needed for this."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BodySlamHipError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"(or `make -C bodyslam_amd/csrc`).  There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, args in _SIGS.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int
    lib.bs_last_error.restype = C.c_char_p
    lib.bs_last_error.argtypes = []
    lib.bs_engine_device_bytes.restype = C.c_int64
    _lib = lib
    return lib


_inited_device: Optional[int] = None


def init(device: int = 0) -> None:
    global _inited_device
    lib = load_library()
    if _inited_device == device:
        return
    if not torch.cuda.is_available():
        raise BodySlamHipError("bodyslam_amd needs an MI355X (gfx950) GPU: torch.cuda.is_available() is False "
                               "and there is no CPU fallback")
    check(lib.bs_init(device), "bs_init")
    _inited_device = device


def check(status: int, what: str) -> None:
    if status != 0:
        msg = load_library().bs_last_error().decode(errors="replace")
        raise BodySlamHipError(f"{what} failed with status {status}: {msg}")


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def dt(t: torch.Tensor) -> int:
    return _TORCH2BS[t.dtype]


def p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


# ---------------------------------------------------------------------------------------------
# thin typed wrappers (argument checking that needs tensor metadata lives here; numeric argument
# validation lives in the C library)
# ---------------------------------------------------------------------------------------------
def make_gemm_desc(A: torch.Tensor, W: torch.Tensor, out: torch.Tensor, *, M: int, N: int, K: int, lda: int,
                   conv=None, relu_a: bool = False, bias: Optional[torch.Tensor] = None, bias_group_rows: int = 0,
                   act: int = ACT_NONE, scale: Optional[torch.Tensor] = None, res: Optional[torch.Tensor] = None,
                   res2: Optional[torch.Tensor] = None, ldr: int = 0, ldo: Optional[int] = None, out_group=None,
                   shuffle=None, qkv=None, a_offset: int = 0, tile: int = 0, seg1: int = 0, out_split_off: int = 0,
                   res_split_off: int = 0, f8_seg: int = 0, f8_scales=(127, 127, 127, 127), out_f8=None, res_f8: bool = False,
                   f8_wonly_from: int = 0, out_lo8_rows: int = 0, f8_skip_from: int = 0, bias2=None, out_planes_rows: int = 0,
                   out_relu: Optional[torch.Tensor] = None) -> GemmDesc:
    """Fill a bs_gemm_desc.  conv = (Hin, Win, Cin, Hout, Wout, KH, KW, stride, pad_h, pad_w) or None;
    out_group = (rows, stride, offset); shuffle = (s, Cout, Hgrid, Wgrid);
    qkv = (hidden, tokens, Sp, q_scale, out_k, out_vt[, cls_last[, cls_rows[, patch_row0[, lo_off]]]]); a_offset in elements;
    (lo_off > 0: out / out_k / out_vt are allocated twice over and the rounding residuals go lo_off elements behind the values)
    bias2 = (fp32 [groups, N], row0, group_rows)."""
    d = GemmDesc()
    d.A = A.data_ptr() + a_offset * A.element_size()
    d.W = W.data_ptr()
    assert A.dtype == W.dtype and A.dtype in (torch.float16, torch.bfloat16), (A.dtype, W.dtype)
    d.dtype = dt(A)
    d.M, d.N, d.K, d.lda = M, N, K, lda
    if conv is not None:
        d.conv = 1
        (d.Hin, d.Win, d.Cin, d.Hout, d.Wout, d.KH, d.KW, d.stride, d.pad_h, d.pad_w) = conv
    d.relu_a = int(relu_a)
    if bias is not None:
        assert bias.dtype == torch.float32
        d.bias = bias.data_ptr()
    d.bias_group_rows = bias_group_rows
    d.act = act
    if scale is not None:
        assert scale.dtype == torch.float32
        d.scale = scale.data_ptr()
    if res is not None:
        d.res = res.data_ptr()
        d.res_dtype = dt(res)
        d.ldr = ldr if ldr else N
    if res2 is not None:
        assert res is not None and res2.dtype == A.dtype
        d.res2 = res2.data_ptr()
    d.out = out.data_ptr()
    d.out_dtype = dt(out)
    d.ldo = N if ldo is None else ldo
    if out_group is not None:
        d.out_group_rows, d.out_group_stride, d.out_row_offset = out_group
    if shuffle is not None:
        d.out_mode = OUT_SHUFFLE
        d.shuffle_s, d.shuffle_cout, d.Hout, d.Wout = shuffle
    if qkv is not None:
        d.out_mode = OUT_QKV
        hidden, tokens, sp, q_scale, out_k, out_vt = qkv[:6]
        d.qkv_cls_last = int(bool(qkv[6])) if len(qkv) > 6 else 0
        d.qkv_cls_rows = int(qkv[7]) if len(qkv) > 7 else 0
        d.qkv_patch_row0 = int(qkv[8]) if len(qkv) > 8 else d.qkv_cls_rows
        d.qkv_lo_off = int(qkv[9]) if len(qkv) > 9 else 0
        d.qkv_hidden, d.qkv_tokens, d.qkv_sp, d.q_scale = hidden, tokens, sp, q_scale
        d.out2 = out_k.data_ptr()
        d.out3 = out_vt.data_ptr()
    d.tile = tile
    d.seg1, d.out_split_off, d.res_split_off = seg1, out_split_off, res_split_off
    d.f8_seg = f8_seg
    d.f8_scales = f8_scales[0] | (f8_scales[1] << 8) | (f8_scales[2] << 16) | (f8_scales[3] << 24)
    d.out_f8 = 0 if out_f8 is None else ((out_f8[0] & 0xff) | ((out_f8[1] & 0xff) << 8))
    d.res_f8 = int(res_f8)
    d.f8_wonly_from = f8_wonly_from
    d.out_lo8_rows = out_lo8_rows
    d.f8_skip_from = f8_skip_from
    d.out_planes_rows = out_planes_rows
    if bias2 is not None:
        b2, row0, grows = bias2
        assert b2.dtype == torch.float32 and b2.shape[-1] == N
        d.bias2, d.bias2_row0, d.bias2_group_rows = b2.data_ptr(), row0, grows
    if out_relu is not None:         # second output: relu(y) in out's (hi16 | hi8 | lo8) format (bs_gemm_desc.out2_relu)
        assert qkv is None and out_f8 is not None and out_relu.dtype == out.dtype and out_relu.numel() == out.numel()
        d.out2 = out_relu.data_ptr()
        d.out2_relu = 1
    return d


def gemm(A: torch.Tensor, W: torch.Tensor, out: torch.Tensor, **kw) -> None:
    d = make_gemm_desc(A, W, out, **kw)
    check(load_library().bs_gemm(C.byref(d), stream_ptr()), "bs_gemm")


def _gemm_bytes(d) -> float:
    rows_in = float(d.M if not d.conv else d.M / max(d.stride * d.stride, 1))
    rows_f8 = 0.0 if d.f8_skip_from < 0 else (rows_in if (d.conv or not d.f8_skip_from) else float(min(d.f8_skip_from, d.M)))
    taps = d.KH * d.KW if d.conv else 1
    a = rows_in * (d.Cin if d.conv else d.K) * 2 + rows_f8 * d.f8_seg
    w = float(d.N) * (d.K * 2 + taps * d.f8_seg)
    if d.out_dtype == F32:
        per_row = [4.0, 4.0, 4.0]
    elif d.out_f8:              # (hi16 | hi8 | lo8): 4 B; without the lo8 plane 3 B; hi16 alone 2 B
        per_row = [4.0, 3.0, 2.0]
    else:
        per_row = [2.0 * (2 if d.out_split_off else 1)] * 3
    full = float(d.M)
    r_planes = float(min(d.out_planes_rows, d.M)) if d.out_planes_rows else full            # rows that keep any plane
    r_lo = float(min(d.out_lo8_rows, d.M)) if d.out_lo8_rows else r_planes                   # rows that keep the lo8 plane
    r_lo = min(r_lo, r_planes)
    out = d.N * (r_lo * per_row[0] + (r_planes - r_lo) * per_row[1] + (full - r_planes) * per_row[2])
    res = float(d.M) * d.N * (4 if (d.res and d.res_dtype == F32) else 0)
    return a + w + out + res


class Plan:
    """A prebuilt sequence of C-ABI calls over static buffers: descriptors and pointer arguments are
    marshalled once, so replaying a forward costs one ctypes call per kernel (and the whole sequence
    can be captured into a HIP graph).  `mark(name, tensor)` records a named intermediate for tests."""

    def __init__(self, device=None):
        # the device whose streams the plan is issued on (an engine passes its own; None = torch's current device at run time)
        self.device = None if device is None else torch.device(device)
        self.calls = []      # (cfunc, args) ; args exclude the trailing stream
        self.names = []
        self.keep = []       # keeps descriptors / tensors alive
        self.keep_descs = [] # the bs_gemm descriptors in call order (engine export)
        self.marks = {}      # call index -> [(name, tensor)]
        self.gemm_info = {}  # call index -> dict(tile, conv, flops, bytes)
        self.stack_info = {} # call index -> dict(name, alg_flops, flops): launches other than bs_gemm that carry matrix-core work of a GEMM / conv
                             # the reference has (the fused up-convolution, the attractor MLPs): timed with the GEMMs for the roofline
        self.lane = 0        # lane of the calls being added: 0 = the caller's stream, 1 = the plan's side stream
        self.lanes = []      # per call
        self.events = None   # a list: run() records a HIP event pair around every bs_gemm launch into it (see run_timed)
        self._graph = None   # torch.cuda.CUDAGraph of the captured sequence (capture())
        self._side = None    # torch side stream + fork / join events, created at the first run
        self._events = {}

    # ---- two-lane plans: calls added while `lane` is 1 are issued on the plan's side stream.  `signal(k)` records event k on
    # the current lane's stream, `wait(k)` makes the current lane's stream wait for it: a run of small, latency-bound kernels
    # that only depends on data available at some point of the main lane overlaps the main lane's big kernels from there on
    # (the bins head's router / seeds / attractor chain beside the fusion stage and the relative head).
    def signal(self, k: int):
        self.calls.append(("signal", (k, self.lane)))
        self.names.append(f"signal{k}")
        self.lanes.append(self.lane)

    def wait(self, k: int):
        self.calls.append(("wait", (k, self.lane)))
        self.names.append(f"wait{k}")
        self.lanes.append(self.lane)

    def _streams(self):
        main = torch.cuda.current_stream(self.device)
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
            self._events = {}
        return main, self._side

    def _sync_op(self, op, args, main, side):
        k, lane = args
        ev = self._events.get(k)
        if ev is None:
            ev = self._events[k] = torch.cuda.Event()
        st = side if lane else main
        if op == "signal":
            ev.record(st)
        else:
            st.wait_event(ev)

    def gemm(self, name, A, W, out, **kw):
        passes = kw.pop("precision_passes", 1)
        d = make_gemm_desc(A, W, out, **kw)
        kw["precision_passes"] = passes
        self.keep.append((d, A, W, out, kw))
        self.keep_descs.append(d)
        self.calls.append((load_library().bs_gemm, (C.byref(d),)))
        self.names.append(name)
        self.lanes.append(self.lane)
        # bookkeeping for the roofline: which kernel instantiation and how many algorithmic FLOPs
        self.gemm_info[len(self.calls) - 1] = dict(
            name=name, tile=load_library().bs_gemm_tile(C.byref(d)), conv=bool(d.conv),
            # executed work in 16-bit-MFMA-equivalents: an FP8 correction stage covers 128 k in the time of 64
            # (tiles past f8_wonly_from run only the first FP8 half)
            # (tiles past f8_skip_from run none)
            flops=2.0 * d.N * (d.M * d.K + (0 if d.f8_skip_from < 0 else (min(d.f8_skip_from, d.M) if d.f8_skip_from else d.M)) * (d.KH * d.KW if d.conv else 1) * d.f8_seg
                               / (4 if d.f8_wonly_from else 2)),
            # algorithmic FLOPs exclude the extra passes of a split-precision product
            alg_flops=2.0 * d.M * d.N * (d.K / kw.get("precision_passes", 1)),
            # algorithmic HBM bytes: A once (16-bit values + the FP8 planes of the rows that run FP8 stages), W once, the output
            # (+ the fp32 residual read)
            bytes=_gemm_bytes(d))

    def add(self, name, fn_name, *args):
        cargs = []
        for a in args:
            if isinstance(a, torch.Tensor):
                self.keep.append(a)
                cargs.append(a.data_ptr())
            else:
                cargs.append(a)
        self.calls.append((getattr(load_library(), fn_name), tuple(cargs)))
        self.names.append(name)
        self.lanes.append(self.lane)

    def tag_stack(self, alg_flops: float, flops: float):
        """the call added last carries this much matrix-core work (algorithmic / executed in 16-bit-pass equivalents)"""
        self.stack_info[len(self.calls) - 1] = dict(name=self.names[-1], alg_flops=float(alg_flops), flops=float(flops))

    def mark(self, name, tensor, meta=None):
        self.marks.setdefault(len(self.calls), []).append((name, tensor, meta))

    def run_timed(self, events: list):
        """Like run(), with a HIP event pair (recorded on the launch stream) around every bs_gemm launch;
        appends (call_index, start_event, end_event) to `events`."""
        main, side = self._streams()
        sts = (main, side)
        ptrs = (main.cuda_stream, side.cuda_stream)
        for i, (fn, args) in enumerate(self.calls):
            if isinstance(fn, str):
                self._sync_op(fn, args, main, side)
                continue
            ln = self.lanes[i]
            if i in self.gemm_info or i in self.stack_info:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(sts[ln])
                rc = fn(*args, ptrs[ln])
                e1.record(sts[ln])
                events.append((i, e0, e1))
            else:
                rc = fn(*args, ptrs[ln])
            if rc:
                check(rc, self.names[i])

    def capture(self):
        """Capture the whole launch sequence (both lanes, their fork / join events, the GEMM tail launches) into one HIP graph;
        run() then replays it with a single hipGraphLaunch instead of ~600 ctypes calls.  The buffers are static, so the
        captured pointers stay valid; inputs are copied into them before run() as usual.  Worth it for the reference's
        one-frame-per-call pattern (host time 3.6 ms -> ~0.1 ms per forward; the GPU time does not change)."""
        if self._graph is not None:
            return
        self.run()                              # warm-up: first-use attribute calls, lazy streams and events
        torch.cuda.synchronize(self.device)
        g = torch.cuda.CUDAGraph()
        cap = torch.cuda.Stream(device=self.device)
        cap.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(cap):
            with torch.cuda.graph(g, stream=cap, capture_error_mode="thread_local"):
                self._events = {}               # events recorded during capture belong to the capture
                self.run()
        self._events = {}
        torch.cuda.current_stream(self.device).wait_stream(cap)
        self._graph = g

    def run(self, taps: Optional[dict] = None):
        if taps is None and self.events is not None:      # a caller (bench.py) asked for per-launch HIP events
            return self.run_timed(self.events)
        if taps is None and self._graph is not None and not torch.cuda.is_current_stream_capturing():
            self._graph.replay()
            return
        if taps is None:
            main, side = self._streams()
            ptrs = (main.cuda_stream, side.cuda_stream)
            for i, (fn, args) in enumerate(self.calls):
                if isinstance(fn, str):
                    self._sync_op(fn, args, main, side)
                    continue
                rc = fn(*args, ptrs[self.lanes[i]])
                if rc:
                    check(rc, self.names[i])
            return
        st = torch.cuda.current_stream(self.device).cuda_stream       # tap mode (tests): one stream, program order
        means = taps.get("__site_means__")
        for i in range(len(self.calls) + 1):
            for (name, t, meta) in self.marks.get(i, []):
                if meta and meta[0] == "chanmean":
                    # the channel means of a product's 16-bit input rows as the launch that follows will read them (the calibration's static
                    # bias correction): only when the caller asks for them, never cloned
                    if means is not None:
                        _, off, M, lda, K = meta
                        means[name] = t.view(-1)[off: off + M * lda].view(M, lda)[:, :K].float().mean(0)
                    continue
                if means is None:               # (a run made for the channel means clones nothing else)
                    taps[name] = (t.clone(), meta)
            if i < len(self.calls):
                fn, args = self.calls[i]
                if isinstance(fn, str):
                    continue
                check(fn(*args, st), self.names[i])


_F8_ON_DEVICE = {}      # device index -> does torch's fp32 -> e4m3 cast on that device give the CPU cast's bytes (checked once per process)


def _f8_pack_device(device):
    """where the weight planes are formed: on `device` when torch's e4m3 cast there reproduces the CPU cast bit for bit (checked once on a
    sample that covers ties, the subnormal range and the saturation bound), else on the host.  One-off weight re-layout, not the compute
    path -- but a ZoeD_NK engine packs 300 M weights, 40 s of an 8-core host against < 1 s on the GPU."""
    if device is None or os.environ.get("BS_PACK_ON_HOST") == "1":       # (the switch: A / B of the two packings)
        return torch.device("cpu")
    device = torch.device(device)
    if device.type != "cuda":
        return torch.device("cpu")
    ok = _F8_ON_DEVICE.get(device.index)
    if ok is None:
        g = torch.Generator().manual_seed(7)
        t = torch.cat([torch.randn(4096, generator=g) * 100.0, torch.randn(4096, generator=g) * 1e-2, torch.linspace(-460.0, 460.0, 4097),
                       torch.arange(0, 4096, dtype=torch.float32) * 2.0 ** -11, torch.tensor([0.0, -0.0, 448.0, -448.0, 2.0 ** -9, 2.0 ** -10, 3 * 2.0 ** -11])])
        t = t.clamp(-448.0, 448.0)
        try:
            ok = bool(torch.equal(t.to(torch.float8_e4m3fn).view(torch.uint8), t.to(device).to(torch.float8_e4m3fn).view(torch.uint8).cpu()))
        except Exception:
            ok = False
        _F8_ON_DEVICE[device.index] = ok
    return device if ok else torch.device("cpu")


def f8_weight(w: torch.Tensor, dtype, planes: str = "both", device=None) -> tuple:
    """fp32 [N, K] -> ([N, 2K] `dtype`-typed rows of [W_hi16 | W_lo8 | W_hi8] bytes, (sb0, sb1)): the weight side of bs_gemm's
    FP8 correction segment.  W_hi8 = e4m3(W_hi * 2^e_hi), W_lo8 = e4m3((W - W_hi) * 2^e_lo) with per-matrix power-of-two
    scales that put the largest magnitude just under e4m3's 448; sb0 / sb1 are the E8M0 exponents bs_gemm applies to the
    lo / hi plane (127 - e).  planes: "both" (default), "lo" = [W_hi16 | W_lo8] only, "hi_only" = the W_lo8 plane zeroed.
    device: where the planes are formed and returned (_f8_pack_device; default: the host)."""
    import math
    w = w.detach().to(_f8_pack_device(device)).float()
    hi = w.to(dtype)
    lo = w - hi.float()

    def plane(t):
        mx = float(t.abs().max())
        e = 0 if mx == 0.0 else min(int(math.floor(math.log2(448.0 / mx))), 100)
        return (t * (2.0 ** e)).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8), e

    hi8, e_hi = plane(hi.float())
    lo8, e_lo = plane(lo)
    hi16 = hi.contiguous().view(torch.uint8).view(w.shape[0], -1)
    if planes == "lo":            # [W_hi16 | W_lo8]: the weight-rounding correction only (bs_gemm f8_seg = K)
        row = torch.cat([hi16, lo8], 1).contiguous()
    elif planes == "hi_only":     # probe: no weight-rounding correction (the W_lo8 plane is zero)
        row = torch.cat([hi16, torch.zeros_like(lo8), hi8], 1).contiguous()
    else:
        row = torch.cat([hi16, lo8, hi8], 1).contiguous()
    return row.view(dtype), (127 - e_lo, 127 - e_hi)


def f8_conv_weight(w_ohwi: torch.Tensor, dtype, device=None) -> tuple:
    """fp32 [O, kh, kw, I] -> the conv-mode counterpart of f8_weight: K order [W_hi16: chunk64, tap, 64][W_lo8: chunk128, tap, 128]
    [W_hi8: chunk128, tap, 128] (the kernel's chunk walk continues from the 16-bit channels through the two FP8 planes)."""
    import math
    w = w_ohwi.detach().to(_f8_pack_device(device)).float()
    O, kh, kw, I = w.shape
    assert I % 128 == 0, I
    hi = w.to(dtype)
    lo = w - hi.float()

    def plane(t):
        mx = float(t.abs().max())
        e = 0 if mx == 0.0 else min(int(math.floor(math.log2(448.0 / mx))), 100)
        q = (t * (2.0 ** e)).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8)
        return q.reshape(O, kh, kw, I // 128, 128).permute(0, 3, 1, 2, 4).reshape(O, -1), e

    hi8, e_hi = plane(hi.float())
    lo8, e_lo = plane(lo)
    hi16 = conv_weight(hi).view(torch.uint8).view(O, -1)
    row = torch.cat([hi16, lo8, hi8], 1).contiguous()
    return row.view(dtype), (127 - e_lo, 127 - e_hi)


F8_ACT_HI_EXP, F8_ACT_LO_EXP = 0, 11          # include/bodyslam_hip.h BS_F8_ACT_*_EXP


# ---------------------------------------------------------------------------------------------
def conv_weight(w_ohwi: torch.Tensor) -> torch.Tensor:
    """[O, kh, kw, I] -> the K order bs_gemm's conv mode walks: [O][I/64 chunks][kh][kw][64] flattened to [O, kh*kw*I]."""
    O, kh, kw, I = w_ohwi.shape
    assert I % 64 == 0, I
    return w_ohwi.reshape(O, kh, kw, I // 64, 64).permute(0, 3, 1, 2, 4).reshape(O, -1).contiguous()


def conv_geom(Hin, Win, Cin, KH, KW, stride, pad):
    Hout = (Hin + 2 * pad - KH) // stride + 1
    Wout = (Win + 2 * pad - KW) // stride + 1
    return (Hin, Win, Cin, Hout, Wout, KH, KW, stride, pad, pad)


def attention(q, k, vt, bias, out, B, nh, S, Sp):
    check(load_library().bs_attention(p(q), p(k), p(vt), p(bias), p(out), B, nh, S, Sp, dt(q), stream_ptr()), "bs_attention")


def attention_table(q, k, vt, table, out, B, nh, hp, wp, Sp, split=0, grouped=0):
    check(load_library().bs_attention_table(p(q), p(k), p(vt), p(table), p(out), B, nh, hp, wp, Sp, int(grouped), dt(q) | split, stream_ptr()),
          "bs_attention_table")


def attention_table_corr(q, k, vt, q_lo, k_lo, vt_lo, table, out, B, nh, hp, wp, Sp, split=0, grouped=0):
    """split-precision operands: x = x16 + x_lo (bs_attention_table_corr)"""
    check(load_library().bs_attention_table_corr(p(q), p(k), p(vt), p(q_lo), p(k_lo), p(vt_lo), p(table), p(out), B, nh, hp, wp, Sp, int(grouped),
                                                 dt(q) | split, stream_ptr()), "bs_attention_table_corr")


def layernorm(x, gamma, beta, out16, out32, rows, cols, eps, dtype=F16):
    check(load_library().bs_layernorm(p(x), p(gamma), p(beta), p(out16), p(out32), rows, cols, eps,
                                      dt(out16) if out16 is not None else dtype, stream_ptr()), "bs_layernorm")


def cast(x, out):
    check(load_library().bs_cast(p(x), p(out), x.numel(), dt(out), stream_ptr()), "bs_cast")


def cast_split(x, out, rows, cols, f8=False):
    """fp32 [rows, cols] -> 16-bit [rows, 2*cols] = (hi | lo),  hi = round16(x), lo = round16(x - hi);
    f8: (hi16 | hi8 | lo8), the operand format of the FP8 correction passes."""
    check(load_library().bs_cast_split(p(x), p(out), rows, cols, dt(out) | (32 if f8 else 0), stream_ptr()), "bs_cast_split")


def relu_split(x, out, rows, cols, f8=False):
    """(hi | lo) [rows, 2*cols] -> re-split relu(hi + lo); f8: the same on (hi16 | hi8 | lo8) rows."""
    check(load_library().bs_relu_split(p(x), p(out), rows, cols, dt(out) | (32 if f8 else 0), stream_ptr()), "bs_relu_split")


def preprocess_patches(frames, out, B, H, W, nh, nw, flip):
    check(load_library().bs_preprocess_patches(p(frames), p(out), B, H, W, nh, nw, int(flip), dt(out), stream_ptr()),
          "bs_preprocess_patches")


def preprocess_image(frames, out, B, H, W, nh, nw, flip):
    check(load_library().bs_preprocess_image(p(frames), p(out), B, H, W, nh, nw, int(flip), stream_ptr()), "bs_preprocess_image")


def fill_rows(x, v, B, rows_per_image, cols):
    check(load_library().bs_fill_rows(p(x), p(v), B, rows_per_image, cols, stream_ptr()), "bs_fill_rows")


def resize_bilinear_nhwc(x, out, B, Hin, Win, Cch, Hout, Wout, align_corners=True, split=False):
    check(load_library().bs_resize_bilinear_nhwc(p(x), p(out), B, Hin, Win, Cch, Hout, Wout, int(align_corners) | (4 if split == 2 else (2 if split else 0)), dt(x),
                                                 stream_ptr()), "bs_resize_bilinear_nhwc")


def depth_u16_to_m(depth_u16: torch.Tensor, depth_scale: float, depth_trunc: float) -> torch.Tensor:
    """int16-stored uint16 depth [..., H, W] on the device -> fp32 metres, values >= depth_trunc -> 0 (include/bodyslam_hip.h)"""
    assert depth_u16.dtype == torch.int16 and depth_u16.is_cuda and depth_u16.is_contiguous()
    out = torch.empty(depth_u16.shape, dtype=torch.float32, device=depth_u16.device)
    check(load_library().bs_depth_u16_to_m(p(depth_u16), depth_u16.numel(), float(depth_scale), float(depth_trunc), p(out), stream_ptr()),
          "bs_depth_u16_to_m")
    return out


def col_mean(A, lda, row0, rows_per_group, groups, row_step, K, out, zero=None):
    """bf16 [groups, K] means over every row_step-th row of each group of the 16-bit matrix A; `zero` (fp32) is cleared as well
    (include/bodyslam_hip.h)"""
    assert out.dtype == torch.bfloat16 and (zero is None or zero.dtype == torch.float32)
    check(load_library().bs_col_mean(p(A), lda, row0, rows_per_group, groups, row_step, K, p(out), p(zero), 0 if zero is None else zero.numel(),
                                     dt(A), stream_ptr()), "bs_col_mean")


def rank1_bias(abar, dw, out):
    """out [G, N] fp32 (zeroed) += abar [G, K] bf16 @ dw [N, K]^T bf16 (include/bodyslam_hip.h)"""
    assert abar.dtype == torch.bfloat16 and dw.dtype == torch.bfloat16 and out.dtype == torch.float32
    G, K = abar.shape
    N = dw.shape[0]
    assert dw.shape[1] == K and tuple(out.shape) == (G, N)
    check(load_library().bs_rank1_bias(p(abar), p(dw), p(out), G, N, K, stream_ptr()), "bs_rank1_bias")


def upconv_tapsum(y, bias, out, B, Hin, Win, Cout, Hout, Wout, align_corners=True, split=False, relu=True):
    """conv3x3(interpolate x2(x)) from the low-resolution tap products y [B, Hin, Win, 9*Cout] fp32 (include/bodyslam_hip.h)"""
    check(load_library().bs_upconv_tapsum(p(y), p(bias), p(out), B, Hin, Win, Cout, Hout, Wout,
                                          int(align_corners) | (4 if split == 2 else (2 if split else 0)), int(relu), dt(out), stream_ptr()),
          "bs_upconv_tapsum")


def upconv_fused(x, w, bias, out, B, Hin, Win, Cin, Cout, mode=0, split=0, relu=True, f8_scales=(127, 127, 127, 127)):
    """relu(conv3x3(interpolate x2, align_corners(x))) in one launch from the low-resolution input (include/bodyslam_hip.h)"""
    check(load_library().bs_upconv_fused(p(x), p(w), p(bias), p(out), B, Hin, Win, Cin, Cout, 2 * Hin, 2 * Win,
                                         1 | (4 if split == 2 else (2 if split else 0)), int(relu), mode, *[int(v) for v in f8_scales], dt(out),
                                         stream_ptr()), "bs_upconv_fused")


def resize_bias_relu_nhwc(x, bias, out, B, Hin, Win, Cch, Hout, Wout, split=False):
    """relu(bilinear(x, align_corners) + bias) on NHWC 16-bit rows / (hi | lo) pairs (include/bodyslam_hip.h)"""
    check(load_library().bs_resize_bias_relu_nhwc(p(x), p(bias), p(out), B, Hin, Win, Cch, Hout, Wout, 1 | (2 if split else 0), dt(out), stream_ptr()),
          "bs_resize_bias_relu_nhwc")


def attractor_step(A, bins_prev, bins_out, route, B, Hp, Wp, H, W, groups, n_bins, n_attr):
    check(load_library().bs_attractor_step(p(A), p(bins_prev), p(bins_out), p(route), B, Hp, Wp, H, W, groups, n_bins, n_attr,
                                           stream_ptr()), "bs_attractor_step")


def mlp2(x, ldx, W1, b1, W2, b2, out, M, K1, N1, N2, act2=ACT_SOFTPLUS_FAST):
    """out = act2(round16(relu(x W1^T + b1)) W2^T + b2) in one launch (include/bodyslam_hip.h: bs_mlp2)"""
    check(load_library().bs_mlp2(p(x), ldx, p(W1), p(b1), p(W2), p(b2), p(out), M, K1, N1, N2, act2, dt(x), stream_ptr()), "bs_mlp2")


def mlp2_add(emb, prev, W1, b1, W2, b2, out, B, Hp, Wp, H, W, K1, N1, N2, act2=ACT_SOFTPLUS_FAST, split=False):
    """bs_add_resized + bs_mlp2 in one launch (include/bodyslam_hip.h: bs_mlp2_add)"""
    check(load_library().bs_mlp2_add(p(emb), p(prev), p(W1), p(b1), p(W2), p(b2), p(out), B, Hp, Wp, H, W, K1, N1, N2, act2,
                                     dt(emb) | (16 if split else 0), stream_ptr()), "bs_mlp2_add")


def projector_level(z, b_c1, emb_prev, Wc2, b_c2, We, b_e, x_out, emb_out, eh_out, B, Hl, Wl, H, W, PM, E, NE):
    """one level of the bins head's projector path in one launch (include/bodyslam_hip.h: bs_projector_level)"""
    check(load_library().bs_projector_level(p(z), p(b_c1), p(emb_prev), p(Wc2), p(b_c2), p(We), p(b_e), p(x_out), p(emb_out), p(eh_out), B, Hl, Wl,
                                            H, W, PM, E, NE, dt(z), stream_ptr()), "bs_projector_level")


def add_resized(x, prev, out, B, Hp, Wp, H, W, Cch, split=False):
    check(load_library().bs_add_resized(p(x), p(prev), p(out), B, Hp, Wp, H, W, Cch, dt(x) | (16 if split else 0), stream_ptr()),
          "bs_add_resized")


def logbinom_depth(last, Eh, bins, w0_last, w2, b2, route, depth, B, H, W, He, We, min_temp, max_temp):
    check(load_library().bs_logbinom_depth(p(last), p(Eh), p(bins), p(w0_last), p(w2), p(b2), p(route), p(depth), B, H, W, He, We,
                                           min_temp, max_temp, dt(last), stream_ptr()), "bs_logbinom_depth")


def small_attention(qkv, out, B, S, nheads):
    check(load_library().bs_small_attention(p(qkv), p(out), B, S, nheads, dt(out), stream_ptr()), "bs_small_attention")


def route_argmax(logits, ld, route, B):
    check(load_library().bs_route_argmax(p(logits), ld, p(route), B, stream_ptr()), "bs_route_argmax")


def postprocess_depth(depth_net, depth_m, depth_u16, B, H, W, nh, nw, flip):
    check(load_library().bs_postprocess_depth(p(depth_net), p(depth_m), p(depth_u16), B, H, W, nh, nw, int(flip), stream_ptr()),
          "bs_postprocess_depth")


def cyclepose_im2col(frames, pairs, out, P, H, W):
    check(load_library().bs_cyclepose_im2col(p(frames), p(pairs), p(out), P, H, W, dt(out), stream_ptr()), "bs_cyclepose_im2col")


def instnorm_relu_nhwc(x, out, out_f32, scratch, P, HW, Cch, eps=1e-5):
    check(load_library().bs_instnorm_relu_nhwc(p(x), p(out), p(out_f32), p(scratch), P, HW, Cch, eps, dt(out), stream_ptr()),
          "bs_instnorm_relu_nhwc")


def avgpool_nhwc(x, out, P, HW, Cch):
    check(load_library().bs_avgpool_nhwc(p(x), p(out), P, HW, Cch, stream_ptr()), "bs_avgpool_nhwc")


def cyclepose_head(pooled, x2, w_skip_pool, w_skip_x2, b_skip, w1, b1, w2, b2, pose7, T, scratch, P, HW, Cch):
    check(load_library().bs_cyclepose_head(p(pooled), p(x2), p(w_skip_pool), p(w_skip_x2), p(b_skip), p(w1), p(b1), p(w2), p(b2),
                                           p(pose7), p(T), p(scratch), P, HW, Cch, stream_ptr()), "bs_cyclepose_head")


def backproject(depth_u16, K4, depth_scale, depth_trunc, poses, xyz, idx, count, scratch, B, H, W):
    Karr = (C.c_double * 4)(*[float(v) for v in K4])
    check(load_library().bs_backproject(p(depth_u16), B, H, W, C.cast(Karr, C.c_void_p), float(depth_scale), float(depth_trunc),
                                        p(poses), p(xyz), p(idx), p(count), p(scratch), stream_ptr()), "bs_backproject")


def pose_chain(t_rel, N, g0, g_abs):
    g0arr = None
    if g0 is not None:
        g0arr = C.cast((C.c_double * 16)(*[float(v) for v in g0]), C.c_void_p)
    check(load_library().bs_pose_chain(p(t_rel), N, g0arr, p(g_abs), stream_ptr()), "bs_pose_chain")


def pose_chain_from(t_rel, N, g0_dev, g_abs):
    """g0_dev: fp64 [16] device tensor (e.g. the last pose of the previous call's output)"""
    check(load_library().bs_pose_chain_from(p(t_rel), N, p(g0_dev), p(g_abs), stream_ptr()), "bs_pose_chain_from")


def pixel_to_3d(uvd, K4, out, n):
    Karr = (C.c_double * 4)(*[float(v) for v in K4])
    check(load_library().bs_pixel_to_3d(p(uvd), n, C.cast(Karr, C.c_void_p), p(out), stream_ptr()), "bs_pixel_to_3d")
