"""CPU tier: pin the oracle restatement against the reference's own outputs.

The reference has no golden vectors of its own, so the pins
are (a) fixtures recorded from the unmodified reference translation unit
(tests/golden/make_golden.py) and (b), where /root/reference is present, the
live reference build on fresh random inputs.
"""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, assert_bit_equal


def test_star_lengths_match_reference(oracle, golden24):
    """fs[].d as the reference main() computes it (serial_new/...:122,:127)."""
    fs = oracle.make_star(golden24.star("818"))
    assert np.array_equal(fs["d"].view(np.uint32), golden24.z["fs_d_818"].view(np.uint32))


def test_converged_boxes_match_reference(oracle, golden):
    n = 0
    for key, sname, offs, start, want, ref_sweeps in golden.cases():
        fs = oracle.make_star(offs)
        tt, sweeps, _ = oracle.converge(golden.v, fs, start)
        assert sweeps == ref_sweeps, key
        assert_bit_equal(tt, want, key)
        assert oracle.validate(golden.v, tt, fs, start)[0] == 0, key
        n += 1
    assert n == 20


def test_eight_orderings_reach_the_same_bits(oracle, golden):
    """The fixed point does not depend on the relaxation order."""
    for key, sname, offs, start, want, _ in golden.cases():
        fs = oracle.make_star(offs)
        tt, _, _ = oracle.converge(golden.v, fs, start, order=1)
        assert_bit_equal(tt, want, key)


def test_single_pass_states_and_counts(oracle, golden):
    """Order-dependent one-pass states pin the sweep body and its change count
    (serial_new/...:198-256) line by line."""
    for sname in ("3", "818", "nonsym"):
        m = golden.meta[f"pass_{sname}"]
        fs = oracle.make_star(golden.star(sname))
        start = m["start"]
        tt = oracle.tt_init(golden.v.shape, start)
        for n, want_count in enumerate(m["counts"]):
            got = oracle.sweep(golden.v, tt, fs, start)
            assert got == want_count, (sname, n)
            assert_bit_equal(tt, golden.z[f"pass{n + 1}_{sname}"], f"{sname} pass {n + 1}")


def test_star_subrange(oracle, golden):
    m = golden.meta["3_range_5_60"]
    fs = oracle.make_star(golden.star("3"))
    tt, sweeps, _ = oracle.converge(golden.v, fs, m["start"], starstart=5, starstop=60)
    assert sweeps == m["sweeps"]
    assert_bit_equal(tt, golden.z["tt_3_range_5_60"], "range")


def test_quirks_change_the_answer(oracle, golden24):
    """Ignoring the exclusive star bound changes cells: the pins above
    would not pass with starstop = starsize."""
    offs = golden24.star("3")
    fs = oracle.make_star(offs)
    start = golden24.z["start_3_deadin"]
    full, _, _ = oracle.converge(golden24.v, fs, start, starstop=len(fs))
    assert not np.array_equal(full.view(np.uint32), golden24.z["tt_3_deadin"].view(np.uint32))


def test_against_live_reference(oracle):
    """Fresh random inputs through the reference build (only beside /root/reference)."""
    if oracle.ref() is None:
        pytest.skip("reference checkout not present (GPU box)")
    rng = np.random.default_rng(7)
    for shape in ((13, 9, 8), (6, 17, 10)):
        v = rng.uniform(0.1, 0.4, size=shape).astype(np.float32)
        for n_off in (5, 40):
            offs = rng.integers(-3, 4, size=(n_off, 3)).astype(np.int32)
            offs = offs[np.any(offs != 0, axis=1)]
            start = [int(rng.integers(0, n)) for n in shape]
            (want,), ref_sweeps = oracle.ref_converge(v, offs, [start])
            fs = oracle.make_star(offs)
            tt, sweeps, _ = oracle.converge(v, fs, start)
            assert sweeps == ref_sweeps
            assert_bit_equal(tt, want, f"{shape} {n_off}")
            tt8, _, _ = oracle.converge(v, fs, start, order=1)
            assert_bit_equal(tt8, want, f"{shape} {n_off} 8-ord")


def test_vbox_checksum_matches_reference_writer(oracle):
    """Signed-byte checksum (include/velocityboxfiler.h:240-252) on a file the
    reference writer produced."""
    blob = open(os.path.join(GOLDEN, "ref_written_6x5x4.vbox"), "rb").read()
    words = np.frombuffer(blob[:-4], dtype="<u4")
    stored = int(np.frombuffer(blob[-4:], dtype="<u4")[0])
    assert oracle.vbox_checksum(words) == stored
    naive = int(words.astype(np.uint64).sum() % (1 << 32))
    assert naive != stored      # the documented "sum of uint32" is NOT what the code does


@pytest.mark.slow
def test_full_size_3fs_digest(oracle):
    """241x241x51, 3-FS, start (120,120,50): the oracle (8 orderings) reproduces the
    SHA-256 of the reference's converged box (102 reference-order sweeps)."""
    path = os.path.join(GOLDEN, "big_digests.json")
    if not os.path.exists(path):
        pytest.skip("big digests not generated")
    want = json.load(open(path)).get("syn241_3_120_120_50")
    if want is None:
        pytest.skip("3-FS digest not generated")
    import ttsweep_pkg
    P = ttsweep_pkg.load()
    v = P.inputs.velocity_model(241, 241, 51, 20160507)
    fs = oracle.make_star(oracle.read_triples(P.inputs.star_path("3")))
    tt, sweeps, _ = oracle.converge(v, fs, (120, 120, 50), order=1)
    assert hashlib.sha256(tt.tobytes()).hexdigest() == want["sha256"]


def test_torch_checker_counts_what_the_oracle_validator_counts(oracle, pkg):
    """tests/torch_checker.py (plain PyTorch, used on the GPU tier for the grids no CPU run
    reaches) against the oracle's validator on the reference's own states: the converged box
    and the states after one and two reference passes, symmetric and non-symmetric stars."""
    import json
    import os

    import torch
    from conftest import GOLDEN
    from torch_checker import fixed_point_counts
    z = np.load(os.path.join(GOLDEN, "g24.npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    v = z["v"]
    for sname in ("818", "nonsym", "3"):
        if f"pass_{sname}" not in meta:
            continue
        offs, start = z[f"star_{sname}"], meta[f"pass_{sname}"]["start"]
        ofs, fs = oracle.make_star(offs), pkg.inputs.make_fs(offs)
        conv, _, _ = oracle.converge(v, ofs, start)
        for tt in (conv, z[f"pass1_{sname}"], z[f"pass2_{sname}"]):
            want = oracle.validate(v, tt, ofs, start)
            got = fixed_point_counts(torch.from_numpy(v), torch.from_numpy(np.ascontiguousarray(tt)), fs, start)
            assert got[:2] == want and got[2] == 0, (sname, got, want)
        bad = conv.copy()
        far = tuple(0 if 2 * s >= n else n - 1 for s, n in zip(start, conv.shape))
        bad[far] = np.float32(0.5) * bad[far]
        opened, _, unsupported = fixed_point_counts(torch.from_numpy(v), torch.from_numpy(bad), fs, start)
        assert opened > 0 and unsupported >= 1
