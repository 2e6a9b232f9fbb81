"""CPU tier: the C-ABI library loads, exports every symbol include/ttsweep.h declares,
its host-only helpers are right, and the product path fails loudly without a GPU."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "ttsweep.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ttsweep_[a-zA-Z_]\w*)\s*\(", text)))


def test_library_exports_every_declared_symbol(pkg):
    L = pkg._lib.lib()
    names = declared_symbols()
    assert len(names) >= 14
    bound = {n for n, _, _ in pkg._lib.SYMBOLS}
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/ttsweep.h but not exported"
        assert n in bound, f"{n} has no ctypes binding"
    assert L.ttsweep_abi_version() == 6


def test_struct_layouts_match_reference_structs(pkg):
    """struct FS 16 B, struct START 12 B."""
    import ctypes as C
    assert C.sizeof(pkg._lib.FS) == 16 and pkg._lib.FS.d.offset == 12
    assert C.sizeof(pkg._lib.Start) == 12
    assert pkg.inputs.FS_DTYPE.itemsize == 16


def test_star_preparation_matches_oracle(pkg, oracle):
    """Product-side make_fs (host code) against the restated reference statements."""
    for name in ("3", "5", "818"):
        offs = pkg.inputs.read_triples(pkg.inputs.star_path(name))
        a = pkg.inputs.make_fs(offs)
        b = oracle.make_star(offs)
        assert np.array_equal(a["d"].view(np.uint32), b["d"].view(np.uint32))
        assert np.array_equal(a["i"], b["i"]) and np.array_equal(a["k"], b["k"])


def pull_star_python(fs, starstart, starstop):
    """Independent derivation of the liveness rule."""
    out = {}
    for l in range(starstart, starstop):
        e = (int(fs["i"][l]), int(fs["j"][l]), int(fs["k"][l]))
        if e == (0, 0, 0):
            continue
        h = np.float32(fs["d"][l]) * np.float32(0.5)
        out[(e, h.tobytes())] = out.get((e, h.tobytes()), 0) | 1
        m = (-e[0], -e[1], -e[2])
        out[(m, h.tobytes())] = out.get((m, h.tobytes()), 0) | 2
    return out


def test_pull_star_symmetric(pkg):
    """Shipped stars are point-symmetric: every offset is live both ways except the
    pair +-off[S-1] made one-sided by the exclusive bound (the dead edge)."""
    for name, size in (("3", 98), ("5", 422), ("818", 818)):
        offs = pkg.inputs.read_triples(pkg.inputs.star_path(name))
        fs = pkg.inputs.make_fs(offs)
        pull = pkg.build_pull_star(fs)
        assert len(pull) == size
        last = tuple(offs[-1])
        flags = {(di, dj, dk): f for di, dj, dk, f, h in pull}
        assert flags[last] == 2 and flags[tuple(-x for x in last)] == 1
        assert sum(1 for f in flags.values() if f == 3) == size - 2
        want = pull_star_python(fs, 0, len(fs) - 1)
        got = {((di, dj, dk), np.float32(h).tobytes()): f for di, dj, dk, f, h in pull}
        assert got == want


def test_pull_star_general(pkg):
    rng = np.random.default_rng(3)
    offs = rng.integers(-4, 5, size=(30, 3)).astype(np.int32)
    offs[4] = 0                     # an (ignored) zero offset
    offs[7] = offs[2]               # a duplicate
    fs = pkg.inputs.make_fs(offs)
    for a, b in ((0, 29), (3, 17), (10, 10)):
        pull = pkg.build_pull_star(fs, a, b)
        got = {((di, dj, dk), np.float32(h).tobytes()): f for di, dj, dk, f, h in pull}
        assert got == pull_star_python(fs, a, b)


def test_relaxation_counts_match_survey_table(pkg):
    """exact in-bounds relaxations per start*sweep."""
    want = {("3", (241, 241, 51)): 2.7808e8, ("5", (241, 241, 51)): 1.1835e9,
            ("818", (241, 241, 51)): 2.2462e9, ("818", (512, 512, 256)): 5.3715e10,
            ("818", (1024, 1024, 512)): 4.3416e11}
    for (name, shape), approx in want.items():
        fs = pkg.inputs.make_fs(pkg.inputs.read_triples(pkg.inputs.star_path(name)))
        n = pkg.relaxations_per_sweep(shape, fs)
        assert abs(n - approx) / approx < 1e-4, (name, shape, n)
        # brute force on the offsets
        o = np.abs(np.stack([fs["i"], fs["j"], fs["k"]], 1)[:-1].astype(np.int64))
        assert n == int(np.prod(np.maximum(np.array(shape) - o, 0), axis=1).sum())


def test_no_cpu_fallback(pkg):
    """Without a HIP device the product path must fail loudly, not fall back."""
    try:
        n = pkg.device_count()
    except pkg.TTSweepError:
        n = 0
    if n > 0:
        pytest.skip("a GPU is present")
    fs = pkg.inputs.make_fs(pkg.inputs.read_triples(pkg.inputs.star_path("3")))
    with pytest.raises(pkg.TTSweepError):
        pkg.TravelTimeSolver((8, 8, 8), fs)
    v = np.ones((4, 4, 4), np.float32)
    tt = np.full((4, 4, 4), np.inf, np.float32)
    with pytest.raises(pkg.TTSweepError):
        pkg.sweepXYZ(v, tt, fs, (0, 0, 0))
    assert np.isinf(tt).all()


def test_vbox_python_io_matches_reference_writer(pkg, tmp_path):
    from conftest import GOLDEN
    ref_file = os.path.join(GOLDEN, "ref_written_6x5x4.vbox")
    want = np.load(os.path.join(GOLDEN, "ref_written_6x5x4_values.npy"))
    origin, v = pkg.inputs.read_vbox(ref_file)
    assert origin == (1, 1, 1) and np.array_equal(v, want)
    out = tmp_path / "mine.vbox"
    pkg.inputs.write_vbox(str(out), want, origin=(1, 1, 1))
    assert out.read_bytes() == open(ref_file, "rb").read()
    bad = bytearray(out.read_bytes())
    bad[40] ^= 1
    out.write_bytes(bytes(bad))
    with pytest.raises(ValueError):
        pkg.inputs.read_vbox(str(out))


def test_synthetic_model_digest(pkg):
    """The 241x241x51 synthetic model is the one SURVEY.md Appendix C hashed."""
    import hashlib
    import struct
    v = pkg.inputs.velocity_model(241, 241, 51, 20160507)
    body = struct.pack("<4s6i", b"vbox", 1, 1, 1, 241, 241, 51) + v.tobytes()
    cs = pkg.inputs.vbox_checksum(np.frombuffer(body, dtype="<u4"))
    blob = body + struct.pack("<I", cs)
    assert hashlib.sha256(blob).hexdigest() == \
        "c050f8c40f4bcfdb443f51577429637ae234c874702152ad9a9abd3b45419a26"
