import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: CPU test that takes tens of seconds")


@pytest.fixture(scope="session")
def oracle():
    """The CPU checker (test infrastructure only)."""
    import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def pkg():
    """The product package with libttsweep.so built (hipcc cross-compiles on CPU)."""
    import ttsweep_pkg
    P = ttsweep_pkg.load()
    P._lib.build()          # `make` is incremental: a stale .so is never tested against new sources
    return P


class Golden:
    def __init__(self, name):
        self.z = np.load(os.path.join(GOLDEN, f"{name}.npz"))
        self.meta = json.loads(bytes(self.z["meta"]).decode())
        self.v = self.z["v"]

    def star(self, sname):
        return self.z[f"star_{sname}"]

    def cases(self):
        """(key, star name, offsets, start, converged tt, reference sweeps)"""
        for key, m in self.meta.items():
            if f"tt_{key}" in self.z and f"start_{key}" in self.z:
                sname = key.split("_")[0]
                yield key, sname, self.star(sname), self.z[f"start_{key}"], self.z[f"tt_{key}"], m["sweeps"]


@pytest.fixture(scope="session", params=["g24", "g9"])
def golden(request):
    return Golden(request.param)


@pytest.fixture(scope="session")
def golden24():
    return Golden("g24")


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_bit_equal(got, want, what=""):
    g, w = bits(got), bits(want)
    if not np.array_equal(g, w):
        bad = np.argwhere(g != w)
        first = tuple(bad[0])
        rel = np.abs(got.astype(np.float64) - want) / np.maximum(np.abs(want), 1e-30)
        raise AssertionError(
            f"{what}: {len(bad)} of {g.size} cells differ; first at {first}: got {got[first]!r} "
            f"want {want[first]!r}; max rel {np.nanmax(np.where(np.isfinite(rel), rel, 0)):.3e}")
