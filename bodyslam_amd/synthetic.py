"""Synthetic endoscopy-like sequences for the bench and the parity tests:
per channel a sum of 8 random 2-D sinusoids (mean ~106/255, like the reference's fixture images),
i.i.d. noise sigma = 8, and a cumulative 1-px shift per frame so that consecutive frames differ."""
from __future__ import annotations

import numpy as np


def make_sequence(n: int, h: int = 480, w: int = 640, seed: int = 0) -> np.ndarray:
    rng = np.random.default_rng(seed)
    yy, xx = np.meshgrid(np.arange(h + n, dtype=np.float32), np.arange(w + n, dtype=np.float32), indexing="ij")
    base = np.zeros((h + n, w + n, 3), np.float32)
    for c in range(3):
        f = np.full((h + n, w + n), 106.0, np.float32)
        for _ in range(8):
            fy, fx = rng.uniform(0.002, 0.02, size=2)
            ph = rng.uniform(0, 2 * np.pi)
            amp = rng.uniform(5, 20)
            f += amp * np.sin(2 * np.pi * (fy * yy + fx * xx) + ph).astype(np.float32)
        base[..., c] = f
    out = np.empty((n, h, w, 3), np.uint8)
    for i in range(n):
        fr = base[i:i + h, i:i + w] + rng.normal(0.0, 8.0, size=(h, w, 3)).astype(np.float32)
        out[i] = np.clip(np.rint(fr), 0, 255).astype(np.uint8)
    return out
