"""Seeded synthetic 8-bit frames (SURVEY.md section 8d "Synthetic generator").

pix(f,r,c) = clamp(base + noise, 0, 255)
base  = 200 if ((r//32 + c//32) & 1) else 56          # 32-px checkerboard
noise = (splitmix64(seed ^ (f << 40) ^ (r*W + c)) & 31) - 16
seed  = 0x5EED0000 + stream_id

Integer-only, so the numpy (host) and torch (device) generators agree bit for bit.
"""
from __future__ import annotations

import numpy as np

SEED_BASE = 0x5EED0000
_M64 = (1 << 64) - 1


def splitmix64_np(x: np.ndarray) -> np.ndarray:
    """splitmix64 output function on uint64 state x (wrapping arithmetic)."""
    with np.errstate(over="ignore"):
        z = x.astype(np.uint64) + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _checker(rows: int, cols: int) -> np.ndarray:
    r = np.arange(rows)[:, None] // 32
    c = np.arange(cols)[None, :] // 32
    return np.where(((r + c) & 1) == 1, 200, 56).astype(np.int64)


def frame_np(rows: int, cols: int, frame: int = 0, stream_id: int = 0, kind: str = "checker") -> np.ndarray:
    """One synthetic frame on the host.  kind: checker | noise | constant | impulse."""
    if kind == "constant":
        return np.full((rows, cols), 128, np.uint8)
    if kind == "impulse":
        a = np.zeros((rows, cols), np.uint8)
        a[rows // 2, cols // 2] = 255
        return a
    seed = np.uint64(SEED_BASE + stream_id)
    idx = (np.arange(rows, dtype=np.uint64)[:, None] * np.uint64(cols) + np.arange(cols, dtype=np.uint64)[None, :])
    h = splitmix64_np(seed ^ (np.uint64(frame) << np.uint64(40)) ^ idx)
    if kind == "noise":
        return (h & np.uint64(255)).astype(np.uint8)
    if kind != "checker":
        raise ValueError(f"unknown synthetic frame kind {kind!r}")
    noise = (h & np.uint64(31)).astype(np.int64) - 16
    return np.clip(_checker(rows, cols) + noise, 0, 255).astype(np.uint8)


def frames_np(n: int, rows: int, cols: int, stream_id: int = 0, first_frame: int = 0) -> np.ndarray:
    return np.stack([frame_np(rows, cols, first_frame + f, stream_id) for f in range(n)])


def _i64(v: int) -> int:
    """Two's-complement int64 view of a uint64 constant."""
    v &= _M64
    return v - (1 << 64) if v >= (1 << 63) else v


def frames_torch(n: int, rows: int, cols: int, stream_id: int = 0, first_frame: int = 0, device="cpu", noise_every: int = 0):
    """Same frames as frames_np, generated with torch int64 ops on ``device``.

    noise_every = k > 0: every k-th frame (f % k == k - 1) is the uniform-noise KAT frame of SURVEY 8d
    (frame_np(kind="noise")) instead of the checkerboard - content that populates the orientation stage.

    int64 add/mul wrap like uint64; the logical right shifts are emulated by masking
    the sign-extended bits of torch's arithmetic shift.
    """
    import torch

    def lsr(z, s):
        return (z >> s) & ((1 << (64 - s)) - 1)

    dev = torch.device(device)
    idx = (torch.arange(rows, dtype=torch.int64, device=dev)[:, None] * cols
           + torch.arange(cols, dtype=torch.int64, device=dev)[None, :])
    r = torch.arange(rows, device=dev)[:, None] // 32
    c = torch.arange(cols, device=dev)[None, :] // 32
    base = torch.where(((r + c) & 1) == 1, 200, 56).to(torch.int64)
    out = torch.empty((n, rows, cols), dtype=torch.uint8, device=dev)
    seed = SEED_BASE + stream_id
    for f in range(n):
        x = idx ^ _i64(seed ^ ((first_frame + f) << 40))
        z = x + _i64(0x9E3779B97F4A7C15)
        z = (z ^ lsr(z, 30)) * _i64(0xBF58476D1CE4E5B9)
        z = (z ^ lsr(z, 27)) * _i64(0x94D049BB133111EB)
        z = z ^ lsr(z, 31)
        if noise_every > 0 and f % noise_every == noise_every - 1:
            out[f] = (z & 255).to(torch.uint8)
            continue
        noise = (z & 31) - 16
        out[f] = torch.clamp(base + noise, 0, 255).to(torch.uint8)
    return out
