"""MI355X-native multi-start travel-time sweep (host-side Python mirror of include/ttsweep.h).

The directory name contains a hyphen, so import it through `ttsweep_pkg.load()` at
the repository root (it registers this package as `uoparallel_seismic_project_amd`).
"""
from . import _lib, inputs, multistart, solver
from ._lib import (KERNEL_AUTO, KERNEL_CELL, KERNEL_STRIP, KERNEL_TILE, OPT_GATE_R0_MILLI, OPT_GATE_SPEED_MILLI,
                   OPT_PAIR_MIN_STARTS, OPT_PREPASS_ENTRIES, OPT_ASYNC, OPT_ASYNC_LOW, OPT_ASYNC_HIGH, OPT_ASYNC_SPECIAL, OPT_ASYNC_POLICY, OPT_DEFER_MARGIN_MILLI, OPT_ASYNC_WINDOW_MILLI, OPT_ASYNC_GATE_MILLI, OPT_ASYNC_GATE_FAST_MILLI, OPT_ASYNC_TIMEOUT_MILLI, OPT_TILE_IN_PLACE, OPT_QUEUES, OPT_ASYNC_INUNIT, OPT_ASYNC_HANDOFF, OPT_ASYNC_WAVES, OPT_TILE_ORDER,
                   OPT_KERNEL, OPT_MAX_BATCH, OPT_MAX_SWEEPS, OPT_TIMING)
from .solver import (TravelTimeSolver, TTSweepError, build_pull_star, device_count,
                     relaxations_per_sweep, solve_multi, sweepXYZ)

__all__ = ["_lib", "inputs", "multistart", "solver", "TravelTimeSolver", "TTSweepError",
           "build_pull_star", "device_count", "relaxations_per_sweep", "solve_multi", "sweepXYZ",
           "KERNEL_AUTO", "KERNEL_CELL", "KERNEL_STRIP", "KERNEL_TILE", "OPT_TIMING", "OPT_KERNEL",
           "OPT_MAX_SWEEPS", "OPT_MAX_BATCH", "OPT_GATE_SPEED_MILLI", "OPT_GATE_R0_MILLI", "OPT_PAIR_MIN_STARTS",
           "OPT_PREPASS_ENTRIES", "OPT_ASYNC", "OPT_ASYNC_LOW", "OPT_ASYNC_HIGH", "OPT_ASYNC_SPECIAL", "OPT_ASYNC_POLICY", "OPT_DEFER_MARGIN_MILLI", "OPT_ASYNC_WINDOW_MILLI", "OPT_ASYNC_GATE_MILLI", "OPT_ASYNC_GATE_FAST_MILLI", "OPT_ASYNC_TIMEOUT_MILLI", "OPT_TILE_IN_PLACE", "OPT_QUEUES", "OPT_ASYNC_INUNIT", "OPT_ASYNC_HANDOFF", "OPT_ASYNC_WAVES", "OPT_TILE_ORDER"]
