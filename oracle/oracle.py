"""ctypes front-end of the CPU checker.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module, and only as the checker / CPU baseline (see ttsweep_oracle.h).

Two libraries:
  * _build/libttoracle.so - our plain-C restatement (ttsweep_oracle.c); travels
    to the GPU box as source and is (re)built there with gcc by `build()`.
  * _ref/libttref.so      - the unmodified reference translation unit behind
    ref_wrapper.c; exists only where /root/reference does (never on the GPU
    box); used to pin the restatement and to generate tests/golden.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "_build", "libttoracle.so")
REF_SO = os.path.join(HERE, "_ref", "libttref.so")
REFERENCE_ROOT = os.environ.get("TTSWEEP_REFERENCE", "/root/reference")

FS_DTYPE = np.dtype([("i", "<i4"), ("j", "<i4"), ("k", "<i4"), ("d", "<f4")])
_fp = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


def build(verbose: bool = False) -> None:
    """Compile the restatement (always) and the reference wrapper (if the
    reference checkout is present)."""
    out = subprocess.run(["make", "-C", HERE, f"REF={REFERENCE_ROOT}"],
                         capture_output=True, text=True)
    if verbose or out.returncode:
        print(out.stdout + out.stderr)
    if out.returncode:
        raise RuntimeError("oracle build failed")


_oracle = None
_ref = None


def lib():
    global _oracle
    if _oracle is None:
        if not os.path.exists(ORACLE_SO):
            build()
        L = C.CDLL(ORACLE_SO)
        L.oracle_star_prepare.argtypes = [C.c_void_p, C.c_int, C.c_float]
        L.oracle_star_prepare.restype = None
        L.oracle_tt_init.argtypes = [_fp] + [C.c_int] * 6
        L.oracle_tt_init.restype = None
        L.oracle_sweepXYZ.argtypes = [_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                      C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.oracle_sweepXYZ.restype = C.c_long
        L.oracle_sweep_dir.argtypes = L.oracle_sweepXYZ.argtypes + [C.c_int] * 3
        L.oracle_sweep_dir.restype = C.c_long
        L.oracle_converge.argtypes = L.oracle_sweepXYZ.argtypes + [C.c_int, C.c_int,
                                                                   C.POINTER(C.c_long)]
        L.oracle_converge.restype = C.c_int
        L.oracle_validate.argtypes = L.oracle_sweepXYZ.argtypes + [C.POINTER(C.c_long)]
        L.oracle_validate.restype = C.c_long
        L.oracle_vbox_checksum.argtypes = [C.c_uint, C.c_void_p, C.c_long]
        L.oracle_vbox_checksum.restype = C.c_uint
        _oracle = L
    return _oracle


def have_ref() -> bool:
    return os.path.exists(REF_SO) or os.path.exists(
        os.path.join(REFERENCE_ROOT, "serial_new", "sweep-tt-multistart.c"))


def ref():
    """The reference build, or None when the reference is not available."""
    global _ref
    if _ref is None:
        if not os.path.exists(REF_SO):
            if not have_ref():
                return None
            build()
        if not os.path.exists(REF_SO):
            return None
        L = C.CDLL(REF_SO)
        L.ttref_setup.argtypes = [C.c_int, C.c_int, C.c_int, _fp, C.c_int, _ip, C.c_float,
                                  C.c_int, _ip]
        L.ttref_setup.restype = C.c_int
        L.ttref_sweep.argtypes = [C.c_int, C.c_int, C.c_int]
        L.ttref_sweep.restype = C.c_int
        L.ttref_sweep_default.argtypes = [C.c_int]
        L.ttref_sweep_default.restype = C.c_int
        L.ttref_tt.argtypes = [C.c_int]
        L.ttref_tt.restype = C.POINTER(C.c_float)
        L.ttref_fs_d.argtypes = [C.c_int]
        L.ttref_fs_d.restype = C.c_float
        L.ttref_store_vbox.argtypes = [C.c_char_p] + [C.c_int] * 6 + [_fp]
        L.ttref_store_vbox.restype = C.c_int
        L.ttref_load_vbox.argtypes = [C.c_char_p, _ip, _fp, C.c_long]
        L.ttref_load_vbox.restype = C.c_int
        L.ttref_text_to_vbox.argtypes = [C.c_char_p, C.c_char_p]
        L.ttref_text_to_vbox.restype = C.c_int
        L.ttref_teardown.restype = None
        _ref = L
    return _ref


# --------------------------------------------------------------------------
# inputs
# --------------------------------------------------------------------------

def read_triples(path: str) -> np.ndarray:
    """Star / start file: a count N followed by N integer triples
    (serial_new/sweep-tt-multistart.c:16-24)."""
    with open(path) as f:
        tok = f.read().split()
    n = int(tok[0])
    return np.array(tok[1:1 + 3 * n], dtype=np.int32).reshape(n, 3)


def make_star(offsets: np.ndarray, delta: float = 10.0) -> np.ndarray:
    """struct FS array with d filled by the restated star preparation."""
    offsets = np.ascontiguousarray(offsets, dtype=np.int32).reshape(-1, 3)
    fs = np.zeros(len(offsets), dtype=FS_DTYPE)
    fs["i"], fs["j"], fs["k"] = offsets[:, 0], offsets[:, 1], offsets[:, 2]
    lib().oracle_star_prepare(fs.ctypes.data, len(fs), delta)
    return fs


def tt_init(shape, start) -> np.ndarray:
    tt = np.empty(shape, dtype=np.float32)
    lib().oracle_tt_init(tt.reshape(-1), *shape, *map(int, start))
    return tt


# --------------------------------------------------------------------------
# the restatement
# --------------------------------------------------------------------------

def sweep(v, tt, fs, start, starstart=0, starstop=None, direction=None) -> int:
    """One pass (reference order, or the given (+-1,+-1,+-1) ordering) in place."""
    nx, ny, nz = v.shape
    if starstop is None:
        starstop = len(fs) - 1          # the reference call site, :160
    args = (v.reshape(-1), tt.reshape(-1), nx, ny, nz, fs.ctypes.data, starstart, starstop,
            int(start[0]), int(start[1]), int(start[2]))
    if direction is None:
        return lib().oracle_sweepXYZ(*args)
    return lib().oracle_sweep_dir(*args, *map(int, direction))


def converge(v, fs, start, starstart=0, starstop=None, order=0, max_sweeps=0, tt=None):
    """Sweep to the fixed point.  Returns (tt, sweeps, stores)."""
    nx, ny, nz = v.shape
    if starstop is None:
        starstop = len(fs) - 1
    if tt is None:
        tt = tt_init(v.shape, start)
    stores = C.c_long(0)
    n = lib().oracle_converge(v.reshape(-1), tt.reshape(-1), nx, ny, nz, fs.ctypes.data,
                              starstart, starstop, int(start[0]), int(start[1]), int(start[2]),
                              order, max_sweeps, C.byref(stores))
    return tt, n, stores.value


def validate(v, tt, fs, start, starstart=0, starstop=None):
    """(open_edges, cells_still_infinite); (0, 0) at the fixed point."""
    nx, ny, nz = v.shape
    if starstop is None:
        starstop = len(fs) - 1
    ninf = C.c_long(0)
    n = lib().oracle_validate(v.reshape(-1), np.ascontiguousarray(tt).reshape(-1), nx, ny, nz,
                              fs.ctypes.data, starstart, starstop,
                              int(start[0]), int(start[1]), int(start[2]), C.byref(ninf))
    return n, ninf.value


def vbox_checksum(words: np.ndarray, seed: int = 0) -> int:
    words = np.ascontiguousarray(words, dtype="<u4")
    return lib().oracle_vbox_checksum(seed, words.ctypes.data, words.size)


# --------------------------------------------------------------------------
# the reference itself (only where /root/reference exists)
# --------------------------------------------------------------------------

def ref_converge(v, offsets, starts, delta=10.0, starstart=0, starstop=None, max_sweeps=0):
    """Loop the reference's own sweepXYZ until it returns 0 for every start
    (old/sweep-serial/sweep-tt-multistart.c:189-211).  Returns (list of tt, sweeps)."""
    R = ref()
    if R is None:
        raise RuntimeError("reference build not available")
    nx, ny, nz = v.shape
    offsets = np.ascontiguousarray(offsets, dtype=np.int32).reshape(-1, 3)
    starts = np.ascontiguousarray(starts, dtype=np.int32).reshape(-1, 3)
    if starstop is None:
        starstop = len(offsets) - 1
    if not R.ttref_setup(nx, ny, nz, np.ascontiguousarray(v, dtype=np.float32).reshape(-1),
                         len(offsets), offsets.reshape(-1), delta, len(starts),
                         starts.reshape(-1)):
        raise RuntimeError("ttref_setup failed (limits: FSMAX 818, STARTMAX 12)")
    sweeps = 0
    while True:
        any_change = 0
        for s in range(len(starts)):
            any_change += R.ttref_sweep(s, starstart, starstop)
        sweeps += 1
        if not any_change or (max_sweeps and sweeps >= max_sweeps):
            break
    out = [np.ctypeslib.as_array(R.ttref_tt(s), shape=(nx, ny, nz)).copy()
           for s in range(len(starts))]
    R.ttref_teardown()
    return out, sweeps


def ref_sweeps(v, offsets, start, nsweeps, delta=10.0, starstart=0, starstop=None):
    """State after exactly nsweeps reference-order passes (order-dependent)."""
    out, _ = ref_converge(v, offsets, [start], delta, starstart, starstop, max_sweeps=nsweeps)
    return out[0]
