"""A compiled engine file run through the C ABI alone (``bs_engine_load`` / ``bs_zoedepth_forward`` / ``bs_cyclepose_forward``,
include/bodyslam_hip.h).  This wrapper is the ctypes binding a Python host would write -- it uses nothing of the plan
builder (bodyslam_amd/zoedepth.py, cyclepose.py); a C host makes the same five calls (examples/zoedepth_host.c)."""
from __future__ import annotations

import ctypes as C
from typing import Tuple

import torch

from . import _lib as L


class Engine:
    def __init__(self, path: str, device: int = 0):
        L.init(device)
        self.dev = torch.device("cuda", device)
        self._h = C.c_void_p()
        L.check(L.load_library().bs_engine_load(path.encode(), C.byref(self._h)), "bs_engine_load")

    def close(self):
        if self._h:
            L.load_library().bs_engine_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    @property
    def device_bytes(self) -> int:
        return int(L.load_library().bs_engine_device_bytes(self._h))

    def io(self, name: str) -> Tuple[int, int]:
        """(device address, bytes) of a named static input / output"""
        p, n = C.c_void_p(), C.c_int64()
        L.check(L.load_library().bs_engine_io(self._h, name.encode(), C.byref(p), C.byref(n)), "bs_engine_io")
        return int(p.value), int(n.value)

    def zoedepth_forward(self, frames_u8: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """uint8 [B,H,W,3] on the GPU -> (depth metres fp32 [B,H,W], uint16 metres * 256 [B,H,W] as int16 storage)"""
        assert frames_u8.is_cuda and frames_u8.dtype == torch.uint8 and frames_u8.dim() == 4 and frames_u8.shape[-1] == 3
        fr = frames_u8.contiguous()
        B, H, W, _ = fr.shape
        dm = torch.empty(B, H, W, device=self.dev)
        du = torch.empty(B, H, W, device=self.dev, dtype=torch.int16)
        L.check(L.load_library().bs_zoedepth_forward(self._h, L.p(fr), B, H, W, L.p(dm), L.p(du), L.stream_ptr()), "bs_zoedepth_forward")
        return dm, du

    def cyclepose_forward(self, frames_u8: torch.Tensor, pairs: torch.Tensor) -> torch.Tensor:
        """frames uint8 [N,H,W,3], pairs int32 [P,2] -> T fp32 [P,4,4]"""
        fr, pr = frames_u8.contiguous(), pairs.to(torch.int32).contiguous()
        N, H, W, _ = fr.shape
        T = torch.empty(pr.shape[0], 16, device=self.dev)
        L.check(L.load_library().bs_cyclepose_forward(self._h, L.p(fr), N, H, W, L.p(pr), pr.shape[0], L.p(T), L.stream_ptr()), "bs_cyclepose_forward")
        return T.view(-1, 4, 4)
