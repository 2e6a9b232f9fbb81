"""Python mirror of the C ABI in include/ttsweep.h (thin ctypes layer, no compute).

`TravelTimeSolver` wraps one `ttsweep_ctx`.  Method names follow the C entry
points; array arguments use the reference's FLOATBOX layout ([x][y][z], z fastest,
include/floatbox.h:127-129).  All arithmetic happens in libttsweep.so on the GPU;
if the library or a HIP device is missing every call raises.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import FS, PullEntry, Start, Stats
from .inputs import FS_DTYPE


class TTSweepError(RuntimeError):
    pass


def _check(rc: int, what: str) -> int:
    if rc < 0:
        raise TTSweepError(f"{what}: {_lib.last_error()}")
    return rc


def _require(cond: bool, what: str):
    """Argument checks that must survive `python -O` (a wrong shape, dtype or device handed
    to the C ABI is an out-of-bounds or cross-device access on the GPU)."""
    if not cond:
        raise TTSweepError(what)


def device_count() -> int:
    return _check(_lib.lib().ttsweep_device_count(), "ttsweep_device_count")


def build_pull_star(fs: np.ndarray, starstart: int = 0, starstop: int | None = None):
    """Host-only: the pull form of the star as a list of (di,dj,dk,flags,h)."""
    fs = np.ascontiguousarray(fs, dtype=FS_DTYPE)
    if starstop is None:
        starstop = len(fs) - 1
    cap = 2 * max(len(fs), 1)
    buf = (PullEntry * cap)()
    n = _check(_lib.lib().ttsweep_build_pull_star(fs.ctypes.data, starstart, starstop, buf, cap),
               "ttsweep_build_pull_star")
    return [(e.di, e.dj, e.dk, e.flags, e.h) for e in buf[:n]]


def relaxations_per_sweep(shape, fs: np.ndarray, starstart: int = 0, starstop: int | None = None) -> int:
    fs = np.ascontiguousarray(fs, dtype=FS_DTYPE)
    if starstop is None:
        starstop = len(fs) - 1
    return _lib.lib().ttsweep_relaxations_per_sweep(*map(int, shape), fs.ctypes.data,
                                                    starstart, starstop)


class TravelTimeSolver:
    """One solver context: a grid size, a star range and a device.

    starstop defaults to len(fs)-1, the (exclusive) bound the reference call site
    passes (serial_new/sweep-tt-multistart.c:160)."""

    def __init__(self, shape, fs: np.ndarray, starstart: int = 0, starstop: int | None = None,
                 device: int = 0):
        self._L = _lib.lib()
        self.shape = tuple(int(n) for n in shape)
        self.fs = np.ascontiguousarray(fs, dtype=FS_DTYPE)
        self.starstart = starstart
        self.starstop = len(self.fs) - 1 if starstop is None else starstop
        self.device = device
        self._ctx = self._L.ttsweep_create(device, *self.shape, self.fs.ctypes.data,
                                           self.starstart, self.starstop)
        if not self._ctx:
            raise TTSweepError(f"ttsweep_create: {_lib.last_error()}")

    # -- lifecycle ----------------------------------------------------------
    def close(self):
        if getattr(self, "_ctx", None):
            self._L.ttsweep_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_option(self, key: int, value: int):
        _check(self._L.ttsweep_set_option(self._ctx, key, value), "ttsweep_set_option")

    # -- inputs -------------------------------------------------------------
    def set_velocity(self, v):
        """v: numpy float32 [nx,ny,nz] (host) or a CUDA/HIP torch tensor (device)."""
        if isinstance(v, np.ndarray):
            v = np.ascontiguousarray(v, dtype=np.float32)
            _require(v.shape == self.shape, f"velocity shape {v.shape} != {self.shape}")
            _check(self._L.ttsweep_set_velocity(self._ctx, v.ctypes.data), "ttsweep_set_velocity")
        else:
            self._require_device_tensor(v, self.shape, "velocity")
            import torch
            torch.cuda.current_stream(v.device).synchronize()
            _check(self._L.ttsweep_set_velocity_device(self._ctx, v.data_ptr()),
                   "ttsweep_set_velocity_device")

    def _require_device_tensor(self, t, shape, what):
        import torch
        _require(isinstance(t, torch.Tensor) and t.is_cuda, f"{what}: not a device tensor")
        _require(t.dtype == torch.float32 and t.is_contiguous(), f"{what}: must be contiguous float32")
        _require(tuple(t.shape) == tuple(shape), f"{what}: shape {tuple(t.shape)} != {tuple(shape)}")
        _require(t.device.index == self.device,
                 f"{what}: tensor on device {t.device.index}, solver on device {self.device}")

    # -- the hot path -------------------------------------------------------
    @staticmethod
    def _starts_array(starts):
        starts = np.asarray(starts, dtype=np.int32).reshape(-1, 3)
        arr = (Start * len(starts))()
        for s, (i, j, k) in enumerate(starts):
            arr[s] = Start(int(i), int(j), int(k))
        return arr

    def solve(self, starts, tt_boxes) -> int:
        """In-place solve of host boxes (numpy float32 arrays, one per start).
        Returns 1 if anything improved, 0 if all boxes were already converged."""
        arr = self._starts_array(starts)
        _require(len(tt_boxes) == len(arr), "one box per start")
        ptrs = (C.c_void_p * len(arr))()
        for s, box in enumerate(tt_boxes):
            _require(isinstance(box, np.ndarray) and box.dtype == np.float32, f"box {s}: float32 ndarray")
            _require(box.flags["C_CONTIGUOUS"] and box.shape == self.shape, f"box {s}: shape / layout")
            ptrs[s] = box.ctypes.data
        return _check(self._L.ttsweep_solve(self._ctx, len(arr), arr, ptrs), "ttsweep_solve")

    def solve_device(self, starts, tt, init: bool = True) -> int:
        """Solve with the boxes resident in HBM.  tt: torch float32 tensor
        [nstart,nx,ny,nz] on this solver's device, written in place."""
        import torch
        arr = self._starts_array(starts)
        self._require_device_tensor(tt, (len(arr),) + self.shape, "travel-time boxes")
        ptrs = (C.c_void_p * len(arr))()
        stride = tt.stride(0) * 4
        for s in range(len(arr)):
            ptrs[s] = tt.data_ptr() + s * stride
        torch.cuda.current_stream(tt.device).synchronize()
        return _check(self._L.ttsweep_solve_device(self._ctx, len(arr), arr, ptrs, int(init)),
                      "ttsweep_solve_device")

    def changed(self, nstart: int):
        """Per-start outcome of the last solve: 1 where a travel time of that start improved (the
        reference's changed[s], serial_new/sweep-tt-multistart.c:158-164)."""
        out = (C.c_int * nstart)()
        m = _check(self._L.ttsweep_get_changed(self._ctx, out, nstart), "ttsweep_get_changed")
        return [int(out[s]) for s in range(m)]

    def validate_device(self, start, tt):
        """(open_edges, cells_infinite, cells_unsupported) of one box in HBM (torch tensor
        [nx,ny,nz]): the reference's store conditions evaluated on the device, and the cells
        no store can have produced; (0, 0, 0) exactly for the converged box."""
        import torch
        self._require_device_tensor(tt, self.shape, "travel-time box")
        torch.cuda.current_stream(tt.device).synchronize()
        st = Start(int(start[0]), int(start[1]), int(start[2]))
        a, b, c = C.c_longlong(0), C.c_longlong(0), C.c_longlong(0)
        _check(self._L.ttsweep_validate_device(self._ctx, C.byref(st), tt.data_ptr(), C.byref(a),
                                               C.byref(b), C.byref(c)), "ttsweep_validate_device")
        return a.value, b.value, c.value

    def stats(self) -> dict:
        st = Stats()
        _check(self._L.ttsweep_get_stats(self._ctx, C.byref(st)), "ttsweep_get_stats")
        return {name: getattr(st, name) for name, _ in Stats._fields_}


def solve_multi(devices, v: np.ndarray, fs: np.ndarray, starts, tt_boxes, starstart: int = 0,
                starstop: int | None = None, changed: list | None = None) -> int:
    """ttsweep_solve_multi[_changed]: shard the starts, balanced by cost, over `devices` (host boxes).
    changed: a list that receives the per-start outcome (serial_new/...:158-164: changed[s])."""
    v = np.ascontiguousarray(v, dtype=np.float32)
    fs = np.ascontiguousarray(fs, dtype=FS_DTYPE)
    if starstop is None:
        starstop = len(fs) - 1
    arr = TravelTimeSolver._starts_array(starts)
    _require(len(tt_boxes) == len(arr), "one box per start")
    ptrs = (C.c_void_p * len(arr))()
    for s, box in enumerate(tt_boxes):
        _require(box.dtype == np.float32 and box.flags["C_CONTIGUOUS"] and box.shape == v.shape,
                 f"box {s}: float32, C order, shape of the velocity volume")
        ptrs[s] = box.ctypes.data
    dev = (C.c_int * len(devices))(*devices)
    nx, ny, nz = v.shape
    if changed is not None:
        out = (C.c_int * len(arr))()
        rc = _check(_lib.lib().ttsweep_solve_multi_changed(len(devices), dev, nx, ny, nz, fs.ctypes.data,
                                                           starstart, starstop, v.ctypes.data, len(arr), arr,
                                                           ptrs, out), "ttsweep_solve_multi_changed")
        changed[:] = list(out)
        return rc
    return _check(_lib.lib().ttsweep_solve_multi(len(devices), dev, nx, ny, nz, fs.ctypes.data,
                                                 starstart, starstop, v.ctypes.data, len(arr), arr,
                                                 ptrs), "ttsweep_solve_multi")


def sweepXYZ(v: np.ndarray, tt: np.ndarray, fs: np.ndarray, start, starstart: int = 0,
             starstop: int | None = None) -> int:
    """The one-call drop-in (ttsweep_sweepXYZ): converge `tt` in place on device 0."""
    v = np.ascontiguousarray(v, dtype=np.float32)
    fs = np.ascontiguousarray(fs, dtype=FS_DTYPE)
    _require(tt.dtype == np.float32 and tt.flags["C_CONTIGUOUS"] and tt.shape == v.shape,
             "tt: float32, C order, shape of the velocity volume")
    if starstop is None:
        starstop = len(fs) - 1
    nx, ny, nz = v.shape
    return _check(_lib.lib().ttsweep_sweepXYZ(v.ctypes.data, tt.ctypes.data, nx, ny, nz,
                                              fs.ctypes.data, starstart, starstop,
                                              int(start[0]), int(start[1]), int(start[2])),
                  "ttsweep_sweepXYZ")


MULTI_LOOPBACK, MULTI_NO_RCCL = 1, 2
GATHER_NONE, GATHER_RCCL, GATHER_PEER = 0, 1, 2


def solve_multi_device(devices, v: np.ndarray, fs: np.ndarray, starts, tt_root, starstart: int = 0,
                       starstop: int | None = None, flags: int = 0):
    """ttsweep_solve_multi_device: shard the starts over `devices`, solve in HBM, gather the boxes on
    devices[0] into tt_root (torch float32 tensor [nstart, nx, ny, nz] on that device).  Returns
    (rc, changed per start, gather path)."""
    import torch
    v = np.ascontiguousarray(v, dtype=np.float32)
    fs = np.ascontiguousarray(fs, dtype=FS_DTYPE)
    if starstop is None:
        starstop = len(fs) - 1
    arr = TravelTimeSolver._starts_array(starts)
    n = len(arr)
    _require(tt_root.is_cuda and tt_root.dtype == torch.float32 and tt_root.is_contiguous()
             and tuple(tt_root.shape) == (n,) + tuple(v.shape) and tt_root.device.index == devices[0],
             "tt_root: contiguous float32 [nstart, nx, ny, nz] on devices[0]")
    ptrs = (C.c_void_p * n)()
    for s in range(n):
        ptrs[s] = tt_root.data_ptr() + s * tt_root.stride(0) * 4
    dev = (C.c_int * len(devices))(*devices)
    changed = (C.c_int * max(n, 1))()
    path = C.c_int(0)
    nx, ny, nz = v.shape
    torch.cuda.synchronize(tt_root.device)
    rc = _check(_lib.lib().ttsweep_solve_multi_device(len(devices), dev, nx, ny, nz, fs.ctypes.data, starstart, starstop,
                                                      v.ctypes.data, n, arr, ptrs, flags, changed, C.byref(path)),
                "ttsweep_solve_multi_device")
    return rc, [int(changed[s]) for s in range(n)], path.value
