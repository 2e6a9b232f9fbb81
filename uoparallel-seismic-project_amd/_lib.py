"""Build and load libttsweep.so (the HIP kernels behind the C ABI of include/ttsweep.h).

The shared object is built in-tree (csrc/libttsweep.so) so that it travels with
the repository snapshot.  There is no Python or CPU fallback: if the library
cannot be loaded, importing the solver fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libttsweep.so")


def use_library(path: str) -> None:
    """Load another build of the same sources instead of csrc/libttsweep.so (A/B builds of
    tools/exp; bench.py --lib).  Must be called before the library is first used; the path in
    use is reported by bench.py in its output line."""
    global LIB_PATH
    if _lib is not None:
        raise RuntimeError("libttsweep.so is already loaded")
    LIB_PATH = os.path.abspath(path)


class FS(C.Structure):
    """struct FS (serial_new/sweep-tt-multistart.c:46-49) == ttsweep_fs."""
    _fields_ = [("i", C.c_int), ("j", C.c_int), ("k", C.c_int), ("d", C.c_float)]


class Start(C.Structure):
    """struct START (serial_new/sweep-tt-multistart.c:56-58) == ttsweep_start."""
    _fields_ = [("i", C.c_int), ("j", C.c_int), ("k", C.c_int)]


class PullEntry(C.Structure):
    _fields_ = [("di", C.c_int), ("dj", C.c_int), ("dk", C.c_int), ("flags", C.c_int),
                ("h", C.c_float)]


class Stats(C.Structure):
    _fields_ = [("nstart", C.c_int), ("sweeps_max", C.c_int),
                ("sweeps_total", C.c_longlong), ("cells_relaxed", C.c_longlong),
                ("cells", C.c_longlong),
                ("relaxations_per_sweep", C.c_longlong), ("launches", C.c_longlong),
                ("sweep_kernel_ms", C.c_double), ("solve_ms", C.c_double),
                ("kernel_variant", C.c_int), ("fallbacks", C.c_int)]


# every symbol include/ttsweep.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("ttsweep_abi_version", C.c_int, []),
    ("ttsweep_device_count", C.c_int, []),
    ("ttsweep_last_error", C.c_char_p, []),
    ("ttsweep_warmup", C.c_int, [C.c_int]),
    ("ttsweep_create", C.c_void_p, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]),
    ("ttsweep_destroy", None, [C.c_void_p]),
    ("ttsweep_set_option", C.c_int, [C.c_void_p, C.c_int, C.c_longlong]),
    ("ttsweep_set_velocity", C.c_int, [C.c_void_p, C.c_void_p]),
    ("ttsweep_set_velocity_device", C.c_int, [C.c_void_p, C.c_void_p]),
    ("ttsweep_solve", C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    ("ttsweep_get_changed", C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    ("ttsweep_solve_multi_device", C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                             C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    ("ttsweep_solve_device", C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]),
    ("ttsweep_get_stats", C.c_int, [C.c_void_p, C.c_void_p]),
    ("ttsweep_validate_device", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p]),
    ("ttsweep_solve_multi", C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                      C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    ("ttsweep_solve_multi_changed", C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                              C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("ttsweep_sweepXYZ", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                   C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    ("ttsweep_build_pull_star", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]),
    ("ttsweep_relaxations_per_sweep", C.c_longlong, [C.c_int, C.c_int, C.c_int, C.c_void_p,
                                                     C.c_int, C.c_int]),
]

OPT_TIMING, OPT_KERNEL, OPT_MAX_SWEEPS, OPT_MAX_BATCH = 1, 2, 3, 4
OPT_GATE_SPEED_MILLI, OPT_GATE_R0_MILLI, OPT_PAIR_MIN_STARTS, OPT_PREPASS_ENTRIES = 5, 6, 7, 8
OPT_ASYNC, OPT_ASYNC_LOW, OPT_ASYNC_HIGH, OPT_ASYNC_SPECIAL, OPT_ASYNC_POLICY, OPT_DEFER_MARGIN_MILLI, OPT_ASYNC_WINDOW_MILLI, OPT_ASYNC_GATE_MILLI, OPT_ASYNC_GATE_FAST_MILLI, OPT_ASYNC_TIMEOUT_MILLI = 9, 10, 11, 12, 13, 14, 15, 16, 17, 18
OPT_TILE_IN_PLACE = 19
OPT_QUEUES = 20
OPT_ASYNC_INUNIT = 21
OPT_ASYNC_HANDOFF = 22
OPT_ASYNC_WAVES = 23
OPT_TILE_ORDER = 24
KERNEL_AUTO, KERNEL_CELL, KERNEL_STRIP, KERNEL_TILE = 0, 1, 2, 3


def build(verbose: bool = False) -> str:
    """Compile csrc/ for gfx950 with hipcc (cross-compiles without a GPU)."""
    out = subprocess.run(["make", "-C", CSRC], capture_output=True, text=True)
    if verbose or out.returncode:
        print(out.stdout + out.stderr)
    if out.returncode:
        raise RuntimeError("building libttsweep.so failed")
    return LIB_PATH


_lib = None


def _load_hip_runtime():
    """libttsweep.so is linked without a HIP runtime of its own (-no-hip-rt): make
    one visible (RTLD_GLOBAL) before loading it.  When PyTorch is installed its
    bundled libamdhip64 is used, so that torch (device memory, RCCL) and the sweep
    library share ONE runtime in the process; otherwise the system ROCm one."""
    candidates = []
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec and spec.origin:
            candidates.append(os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so"))
    except Exception:
        pass
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    candidates += [os.path.join(rocm, "lib", "libamdhip64.so"), "libamdhip64.so"]
    errors = []
    for path in candidates:
        if os.path.sep in path and not os.path.exists(path):
            continue
        try:
            return C.CDLL(path, mode=C.RTLD_GLOBAL)
        except OSError as e:
            errors.append(f"{path}: {e}")
    raise RuntimeError("no HIP runtime (libamdhip64) could be loaded: " + "; ".join(errors))


_hip = None


def lib():
    """The loaded library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: run __graft_entry__.build() (hipcc) first; "
                "there is no CPU fallback for the sweep")
        global _hip
        _hip = _load_hip_runtime()
        L = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(L, name)       # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def last_error() -> str:
    return lib().ttsweep_last_error().decode()
