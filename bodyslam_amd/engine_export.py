"""Engine files: a plan compiled once by the Python builder, run afterwards by ANY host through the C ABI alone (SURVEY.md section 8(b):
``bs_zoedepth_forward`` / ``bs_cyclepose_forward``; the reference's host is Python, depth_estimation/interface.py:39-45 and
mpem_interface.py:61-99 -- a C, C++ or Go host binds the same two calls).

A plan (``_lib.Plan``) is a fixed sequence of C-ABI launches over static device buffers.  ``export_plan`` writes it down:

  * every device buffer the launches touch: workspace (size only), zero-initialised (size only) or constant (weights in their packed
    device form, tables: size + bytes),
  * every launch: entry-point name, stream lane, arguments -- integers, floats, and device pointers as (buffer, offset); a
    ``bs_gemm`` descriptor goes in as its raw bytes plus the (field offset -> buffer, offset) relocations of its pointer fields,
  * the fork / join events between the two lanes,
  * named input / output regions ("frames", "depth_m", "depth_u16"; "frames", "pairs", "T").

``bs_engine_load`` (csrc/engine.hip) allocates the buffers, uploads the constants and resolves the entry points inside the library;
``bs_engine_run`` issues the launches.  Nothing of the product's Python runs at inference time.  The file is specific to the (model,
batch, frame size, precision, dtype) the plan was built for -- like the plan itself.

    python -m bodyslam_amd.engine_export zoedepth weights.pt out.bseng --batch 1 --height 480 --width 640
"""
from __future__ import annotations

import bisect
import ctypes as C
import struct
from typing import Dict, Iterable, List, Tuple

import torch

from . import _lib as L

MAGIC = b"BSENG02\0"      # csrc/engine.hip kMagic: bumped with every change of an enum's meaning or of bs_gemm_desc's layout
KIND_WORKSPACE, KIND_ZERO, KIND_DATA = 0, 1, 2
ARG_I64, ARG_F64, ARG_PTR, ARG_NULL, ARG_DESC = 0, 1, 2, 3, 4
OP_CALL, OP_SIGNAL, OP_WAIT = 0, 1, 2


def _tensors(obj, out: list):
    """every CUDA tensor reachable from obj (tuples, lists, dicts)"""
    if isinstance(obj, torch.Tensor):
        if obj.is_cuda:
            out.append(obj)
    elif isinstance(obj, (tuple, list)):
        for o in obj:
            _tensors(o, out)
    elif isinstance(obj, dict):
        for o in obj.values():
            _tensors(o, out)


class _Buffers:
    """the distinct device storages behind a set of tensors; pointer -> (buffer index, offset)"""

    def __init__(self, tensors: Iterable[torch.Tensor], workspace_ptrs: set):
        seen: Dict[int, torch.Tensor] = {}
        for t in tensors:
            st = t.untyped_storage()
            if st.data_ptr() not in seen:
                seen[st.data_ptr()] = t
        self.base = sorted(seen)
        self.size = [seen[b].untyped_storage().nbytes() for b in self.base]
        self.rep = [seen[b] for b in self.base]
        self.workspace = [b in workspace_ptrs for b in self.base]

    def locate(self, ptr: int, what: str) -> Tuple[int, int]:
        i = bisect.bisect_right(self.base, ptr) - 1
        if i < 0 or ptr >= self.base[i] + max(self.size[i], 1):
            raise ValueError(f"engine export: {what} points at device memory the plan does not keep alive ({ptr:#x})")
        return i, ptr - self.base[i]

    def bytes_of(self, i: int) -> torch.Tensor:
        st = self.rep[i].untyped_storage()
        return torch.empty(0, dtype=torch.uint8, device=self.rep[i].device).set_(st, 0, (st.nbytes(),)).cpu()


def _pack_name(s: str, n: int) -> bytes:
    b = s.encode()
    assert len(b) < n, s
    return b + b"\0" * (n - len(b))


def export_plan(plan: L.Plan, path: str, io: Dict[str, torch.Tensor], workspace: Iterable[torch.Tensor] = ()) -> dict:
    """write `plan` as an engine file.  io: name -> tensor (a view of one of the plan's static buffers).  workspace: tensors whose
    storages hold nothing the launches rely on (the plan's pooled intermediates): only their size is written."""
    ts: List[torch.Tensor] = []
    _tensors(plan.keep, ts)
    _tensors(list(io.values()), ts)
    ws = list(workspace)
    _tensors(ws, ts)
    bufs = _Buffers(ts, {t.untyped_storage().data_ptr() for t in ws})
    io_ptrs = {t.untyped_storage().data_ptr() for t in io.values()}
    desc_of = {}
    gi = 0
    for i, (fn, args) in enumerate(plan.calls):
        if not isinstance(fn, str) and fn.__name__ == "bs_gemm":
            desc_of[i] = plan.keep_descs[gi]
            gi += 1
    ptr_fields = [(name, getattr(L.GemmDesc, name).offset) for name, ty in L.GemmDesc._fields_ if ty is C.c_void_p]

    calls = bytearray()
    n_ops = 0
    for i, (fn, args) in enumerate(plan.calls):
        if isinstance(fn, str):
            k, lane = args
            calls += struct.pack("<II", OP_SIGNAL if fn == "signal" else OP_WAIT, lane) + _pack_name("", 48) + struct.pack("<I", 1)
            calls += struct.pack("<Iq", ARG_I64, int(k))
            n_ops += 1
            continue
        name = fn.__name__
        sig = L._SIGS[name][:-1]                       # the trailing stream is supplied at run time
        assert len(sig) == len(args), (name, len(sig), len(args))
        calls += struct.pack("<II", OP_CALL, plan.lanes[i]) + _pack_name(name, 48) + struct.pack("<I", len(args))
        for j, (ty, a) in enumerate(zip(sig, args)):
            if name == "bs_gemm":
                d = desc_of[i]
                raw = bytes(C.string_at(C.addressof(d), C.sizeof(d)))
                rel = []
                for fname, off in ptr_fields:
                    v = getattr(d, fname)
                    if v:
                        b, o = bufs.locate(int(v), f"{plan.names[i]}: bs_gemm_desc.{fname}")
                        rel.append((off, b, o))
                calls += struct.pack("<II", ARG_DESC, len(raw)) + raw + struct.pack("<I", len(rel))
                for off, b, o in rel:
                    calls += struct.pack("<IIq", off, b, o)
            elif ty is C.c_void_p:
                if a is None or a == 0:
                    calls += struct.pack("<Iq", ARG_NULL, 0)
                elif isinstance(a, int):
                    b, o = bufs.locate(a, f"{plan.names[i]}: argument {j} of {name}")
                    calls += struct.pack("<IIq", ARG_PTR, b, o)
                else:
                    raise ValueError(f"engine export: {name} argument {j} is a host pointer; only device pointers and scalars can be exported")
            elif ty in (C.c_float, C.c_double):
                calls += struct.pack("<Id", ARG_F64, float(a))
            else:
                calls += struct.pack("<Iq", ARG_I64, int(a))
        n_ops += 1

    # buffers: the static inputs / outputs and the pooled intermediates carry no data; anything else is written unless it is all zero
    table, blobs, off = bytearray(), [], 0
    n_const = 0
    for i in range(len(bufs.base)):
        kind = KIND_WORKSPACE
        if not bufs.workspace[i] and bufs.base[i] not in io_ptrs:
            data = bufs.bytes_of(i)
            if bool(data.any()):
                kind = KIND_DATA
                blobs.append(data.numpy().tobytes())
                n_const += bufs.size[i]
            else:
                kind = KIND_ZERO
        elif bufs.base[i] in io_ptrs:
            kind = KIND_ZERO
        table += struct.pack("<qIq", bufs.size[i], kind, off if kind == KIND_DATA else -1)
        if kind == KIND_DATA:
            off += (bufs.size[i] + 255) // 256 * 256
    ios = bytearray()
    for name, t in io.items():
        b, o = bufs.locate(t.data_ptr(), f"io {name}")
        ios += _pack_name(name, 32) + struct.pack("<Iqq", b, o, t.numel() * t.element_size())
    header = MAGIC + struct.pack("<IIIIq", len(bufs.base), n_ops, len(io), C.sizeof(L.GemmDesc), len(calls))
    with open(path, "wb") as f:
        f.write(header)
        f.write(table)
        f.write(ios)
        f.write(calls)
        pad = (-f.tell()) % 256
        f.write(b"\0" * pad)
        for blob in blobs:
            f.write(blob)
            f.write(b"\0" * ((-len(blob)) % 256))
    return dict(buffers=len(bufs.base), ops=n_ops, constant_bytes=n_const, workspace_bytes=sum(bufs.size) - n_const)


def export_zoedepth(engine, B: int, H: int, W: int, path: str, flip_aug: bool = True) -> dict:
    """the depth network of ``ZoeDepthEngine`` for B frames of H x W as an engine file; I/O: frames u8 [B,H,W,3] -> depth_m fp32
    [B,H,W], depth_u16 [B,H,W]"""
    zp = engine.plan_for(B, H, W, flip_aug)
    return export_plan(zp.plan, path, {"frames": zp.frames, "depth_m": zp.depth_m, "depth_u16": zp.depth_u16}, workspace=zp.pool.blocks)


def export_cyclepose(engine, n_frames: int, n_pairs: int, H: int, W: int, path: str) -> dict:
    """the pose branch of ``CyclePoseEngine`` for n_pairs pairs over n_frames frames; I/O: frames u8 [N,H,W,3], pairs int32 [P,2]
    -> T fp32 [P,16]"""
    pp = engine.plan_for(n_frames, n_pairs, H, W)
    return export_plan(pp.plan, path, {"frames": pp.frames, "pairs": pp.pairs, "T": pp.T})


def main(argv=None):
    import argparse
    from .weights import load_cyclepose_checkpoint, load_zoedepth_weights
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("model", choices=["zoedepth", "cyclepose"])
    ap.add_argument("weights")
    ap.add_argument("out")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--precision", default="accurate")
    a = ap.parse_args(argv)
    if a.model == "zoedepth":
        from .zoedepth import ZoeDepthEngine
        info = export_zoedepth(ZoeDepthEngine(load_zoedepth_weights(a.weights), precision=a.precision), a.batch, a.height, a.width, a.out)
    else:
        from .cyclepose import CyclePoseEngine
        info = export_cyclepose(CyclePoseEngine(load_cyclepose_checkpoint(a.weights), precision=a.precision), a.batch + 1, a.batch, a.height, a.width,
                                a.out)
    print(info)


if __name__ == "__main__":
    main()
