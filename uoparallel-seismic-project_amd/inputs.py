"""Host-side input handling: forward-star / start files, VBOX files, synthetic models.

Mirrors what the reference main() does before its sweep loop
(serial_new/sweep-tt-multistart.c:77-147) so that Python drivers (tests,
bench.py) feed the C ABI exactly what the C host program feeds it.
"""
from __future__ import annotations

import os
import struct

import numpy as np

from ._lib import FS

DATA_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data")
DELTA = 10.0            # serial_new/sweep-tt-multistart.c:108

FS_DTYPE = np.dtype([("i", "<i4"), ("j", "<i4"), ("k", "<i4"), ("d", "<f4")])
assert FS_DTYPE.itemsize == 16


def star_path(name: str) -> str:
    """data/stars/<name>-FS.txt for the shipped stars '3', '5', '818' (copies of the
    reference's docs/*-FS.txt input files), or `name` itself if it is a path."""
    if os.path.exists(name):
        return name
    return os.path.join(DATA_DIR, "stars", f"{name}-FS.txt")


def starts_path(name: str) -> str:
    if os.path.exists(name):
        return name
    return os.path.join(DATA_DIR, "starts", f"start-{name}-241-241-51.txt")


def read_triples(path: str) -> np.ndarray:
    """`N` followed by N integer triples: the format of both the forward-star
    and the start-point files (serial_new/sweep-tt-multistart.c:16-24,:113,:121,:135,:143)."""
    with open(path) as f:
        tok = f.read().split()
    n = int(tok[0])
    if len(tok) < 1 + 3 * n:
        raise ValueError(f"{path}: expected {n} triples")
    return np.array(tok[1:1 + 3 * n], dtype=np.int32).reshape(n, 3)


def make_fs(offsets: np.ndarray, delta: float = DELTA) -> np.ndarray:
    """struct FS array as the reference main() prepares it (:120-128):
    d = (float)sqrt((double)(i*i+j*j+k*k)), then d = delta * d in float."""
    offsets = np.asarray(offsets, dtype=np.int32).reshape(-1, 3)
    fs = np.zeros(len(offsets), dtype=FS_DTYPE)
    fs["i"], fs["j"], fs["k"] = offsets[:, 0], offsets[:, 1], offsets[:, 2]
    d2 = (offsets.astype(np.int64) ** 2).sum(axis=1)
    d = np.sqrt(d2.astype(np.float64)).astype(np.float32)
    fs["d"] = np.float32(delta) * d
    return fs


def fs_pointer(fs: np.ndarray):
    assert fs.dtype == FS_DTYPE and fs.flags["C_CONTIGUOUS"]
    return fs.ctypes.data_as(__import__("ctypes").POINTER(FS))


# ---------------------------------------------------------------------------
# VBOX files (formats/VBOXFORMAT.txt:29-46 as include/velocityboxfiler.h implements it)
# ---------------------------------------------------------------------------

def vbox_checksum(words: np.ndarray) -> int:
    """Signed-byte word sum of include/velocityboxfiler.h:240-252 (c4 is int8_t, :79)."""
    w = np.ascontiguousarray(words, dtype="<u4").astype(np.uint64)
    corr = ((w & 0x80) << 1) + ((w & 0x8000) << 1) + ((w & 0x800000) << 1)
    return int((w.sum() - corr.sum()) % (1 << 32))


def write_vbox(path: str, v: np.ndarray, origin=(1, 1, 1)) -> None:
    v = np.ascontiguousarray(v, dtype="<f4")
    header = struct.pack("<4s6i", b"vbox", *origin, *v.shape)
    body = header + v.tobytes()
    cs = vbox_checksum(np.frombuffer(body, dtype="<u4"))
    with open(path, "wb") as f:
        f.write(body)
        f.write(struct.pack("<I", cs))


def read_vbox(path: str):
    """Returns (origin, v[nx,ny,nz]); raises on a bad magic or checksum."""
    with open(path, "rb") as f:
        blob = f.read()
    if blob[:4] != b"vbox":
        raise ValueError(f"{path}: not a vbox file")
    ox, oy, oz, nx, ny, nz = struct.unpack("<6i", blob[4:28])
    n = nx * ny * nz
    if len(blob) != 32 + 4 * n:
        raise ValueError(f"{path}: size does not match header")
    words = np.frombuffer(blob[:28 + 4 * n], dtype="<u4")
    (stored,) = struct.unpack("<I", blob[28 + 4 * n:])
    if vbox_checksum(words) != stored:
        raise ValueError(f"{path}: checksum mismatch")
    v = np.frombuffer(blob, dtype="<f4", count=n, offset=28).reshape(nx, ny, nz).copy()
    return (ox, oy, oz), v


# ---------------------------------------------------------------------------
# synthetic velocity models (the real docs/velocity-241-241-51*.txt are not in
# the reference tree: .MISSING_LARGE_BLOBS:1-2; recipe: SURVEY.md Appendix C)
# ---------------------------------------------------------------------------

def velocity_model(nx: int, ny: int, nz: int, seed: int = 20160507) -> np.ndarray:
    """Depth gradient + lateral sinusoid + +-0.005 uniform noise, values ~0.155-0.305."""
    i = np.arange(nx)[:, None, None]
    j = np.arange(ny)[None, :, None]
    k = np.arange(nz)[None, None, :]
    base = (0.18 + 0.10 * k / max(nz - 1, 1) + 0.02 * np.sin(i / 9) * np.cos(j / 11))
    noise = np.random.default_rng(seed).uniform(-0.005, 0.005, size=(nx, ny, nz))
    return base.astype(np.float32) + noise.astype(np.float32)


def velocity_model_device(nx: int, ny: int, nz: int, seed: int, device):
    """Same smooth part as velocity_model, noise from a counter-based hash of
    (i,j,k,seed) so large grids can be generated on the GPU slab by slab and
    reproduced without storing them.  Returns a float32 torch tensor on `device`."""
    import torch

    out = torch.empty((nx, ny, nz), dtype=torch.float32, device=device)
    j = torch.arange(ny, device=device, dtype=torch.float64)[None, :, None]
    k = torch.arange(nz, device=device, dtype=torch.float64)[None, None, :]
    jk = (torch.arange(ny, device=device, dtype=torch.int64)[None, :, None] * nz
          + torch.arange(nz, device=device, dtype=torch.int64)[None, None, :])
    slab = max(1, (1 << 24) // (ny * nz))
    for x0 in range(0, nx, slab):
        x1 = min(nx, x0 + slab)
        ii = torch.arange(x0, x1, device=device, dtype=torch.int64)[:, None, None]
        i = ii.to(torch.float64)
        base = 0.18 + 0.10 * k / max(nz - 1, 1) + 0.02 * torch.sin(i / 9) * torch.cos(j / 11)
        # 64-bit mix (wrapping int64 arithmetic); top 24 bits -> uniform [0,1)
        h = (ii * (ny * nz) + jk) * 0x1E3779B97F4A7C15 + ((seed * 0x632BE59BD9B4E019) & 0x7FFFFFFFFFFFFFFF)
        h = (h ^ (h >> 31)) * 0x2545F4914F6CDD1D
        h = h ^ (h >> 29)
        u = ((h >> 20) & 0xFFFFFF).to(torch.float64) / float(1 << 24)
        out[x0:x1] = base.to(torch.float32) + ((u - 0.5) * 0.01).to(torch.float32)
    return out


def scaled_starts(starts241: np.ndarray, nx: int, ny: int, nz: int) -> np.ndarray:
    """Start points of a 241x241x51 file mapped onto another grid: (i,j) scaled by
    nx/241, ny/241, k = nz-1."""
    s = np.asarray(starts241, dtype=np.int64).reshape(-1, 3)
    out = np.empty_like(s)
    out[:, 0] = np.minimum(s[:, 0] * nx // 241, nx - 1)
    out[:, 1] = np.minimum(s[:, 1] * ny // 241, ny - 1)
    out[:, 2] = nz - 1
    return out.astype(np.int32)
