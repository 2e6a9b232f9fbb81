"""GPU tier (-m gpu): the HIP path, called through the C ABI, against
  * the committed golden fixtures recorded from the reference itself,
  * the CPU oracle on the same seeded inputs,
  * size-independent properties at BASELINE.json's full grid size.
Bar: bit-exact (BASELINE.json asks <= 1e-5 relative; the converged state is
order-independent, so equality is achievable and is what is asserted)."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, assert_bit_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P(pkg):
    n = pkg.device_count()
    assert n > 0, "no HIP device: the GPU tier must run on an MI355X (there is no CPU fallback)"
    return pkg


def gpu_converge(P, v, fs, starts, starstart=0, starstop=None, tts=None, kernel=None):
    """kernel: 1 CELL, 2 STRIP (its own choice of unit size and driver), 3 TILE; 21 / 22: STRIP with
    units of one / two planes whatever the number of starts (latency / throughput mode), the solve as
    ONE launch (ring planners + workers); 23 / 24: the same units, a launch pair per pass."""
    starts = np.asarray(starts, dtype=np.int32).reshape(-1, 3)
    with P.TravelTimeSolver(v.shape, fs, starstart, starstop) as sol:
        if kernel in (21, 22, 23, 24):
            sol.set_option(P.OPT_KERNEL, 2)
            sol.set_option(P.OPT_PAIR_MIN_STARTS, 1 << 20 if kernel in (21, 23) else 0)
            sol.set_option(P.OPT_ASYNC, 1 if kernel in (21, 22) else 0)
        elif kernel is not None:
            sol.set_option(P.OPT_KERNEL, kernel)
        sol.set_velocity(v)
        if tts is None:
            tts = []
            for st in starts:
                tt = np.full(v.shape, np.inf, dtype=np.float32)
                tt[tuple(st)] = 0
                tts.append(tt)
        rc = sol.solve(starts, tts)
        return tts, rc, sol.stats()


KERNELS = [pytest.param(1, id="cell"), pytest.param(21, id="strip1"), pytest.param(22, id="strip2"),
           pytest.param(23, id="strip1-passes"), pytest.param(24, id="strip2-passes")]


def tile_supports(offs):
    """Mirror of the library's rule for the TILE kernel (small stars only)."""
    pull = {tuple(o) for o in offs[:-1]} | {tuple(-o) for o in offs[:-1]}
    pull.discard((0, 0, 0))
    return 0 < len(pull) <= 26 and all(abs(a) <= 2 and abs(b) <= 2 and abs(c) <= 4 for a, b, c in pull)


@pytest.mark.parametrize("kernel", KERNELS + [pytest.param(3, id="tile")])
def test_golden_cases_bit_exact(P, golden, kernel):
    """Every fixture: 5 stars (3 shipped, one non-symmetric, 6-neighbour) x 4 start
    kinds (interior, corner, dead edge inside / outside the grid), every kernel that
    accepts the star (TILE: the small ones)."""
    n = 0
    for key, sname, offs, start, want, _ in golden.cases():
        if kernel == 3 and not tile_supports(offs):
            with pytest.raises(P.TTSweepError):
                gpu_converge(P, golden.v, P.inputs.make_fs(offs), [start], kernel=kernel)
            continue
        fs = P.inputs.make_fs(offs)
        (tt,), rc, st = gpu_converge(P, golden.v, fs, [start], kernel=kernel)
        assert rc == 1, key
        assert st["kernel_variant"] == (2 if kernel > 20 else kernel)
        assert_bit_equal(tt, want, key)
        assert st["sweeps_total"] >= 2
        n += 1
    assert n == (20 if kernel != 3 else 8)


def test_golden_star_subrange(P, golden):
    m = golden.meta["3_range_5_60"]
    fs = P.inputs.make_fs(golden.star("3"))
    (tt,), _, _ = gpu_converge(P, golden.v, fs, [m["start"]], starstart=5, starstop=60)
    assert_bit_equal(tt, golden.z["tt_3_range_5_60"], "range")


def test_batch_of_starts_equals_single_solves(P, golden24):
    """All four start kinds of one star in one batched solve."""
    offs = golden24.star("818")
    fs = P.inputs.make_fs(offs)
    keys = ["818_mid", "818_corner", "818_deadin", "818_deadout"]
    starts = [golden24.z[f"start_{k}"] for k in keys]
    tts, rc, st = gpu_converge(P, golden24.v, fs, starts)
    assert rc == 1 and st["nstart"] == 4
    for k, tt in zip(keys, tts):
        assert_bit_equal(tt, golden24.z[f"tt_{k}"], k)


def test_host_solve_in_memory_sized_batches(P, golden24):
    """ttsweep_solve splits the starts into batches that fit the device (forced here to
    3 + 1); results and accumulated counters are those of the single-batch solve."""
    offs = golden24.star("818")
    fs = P.inputs.make_fs(offs)
    keys = ["818_mid", "818_corner", "818_deadin", "818_deadout"]
    starts = np.array([golden24.z[f"start_{k}"] for k in keys], dtype=np.int32)
    with P.TravelTimeSolver(golden24.v.shape, fs) as sol:
        sol.set_option(P.OPT_MAX_BATCH, 3)
        sol.set_velocity(golden24.v)
        tts = []
        for st in starts:
            tt = np.full(golden24.v.shape, np.inf, dtype=np.float32)
            tt[tuple(st)] = 0
            tts.append(tt)
        assert sol.solve(starts, tts) == 1
        st = sol.stats()
    assert st["nstart"] == 4 and st["sweeps_total"] >= 8
    for k, tt in zip(keys, tts):
        assert_bit_equal(tt, golden24.z[f"tt_{k}"], k)


def test_resume_and_idempotence(P, golden24, oracle):
    """The solve takes the box as its initial state: starting from the reference's
    state after ONE pass reaches the same fixed point, and solving a converged box
    returns 0 and leaves it untouched (the loop exit of serial_new/...:152,166)."""
    m = golden24.meta["pass_818"]
    fs = P.inputs.make_fs(golden24.star("818"))
    tt = golden24.z["pass1_818"].copy()
    (tt,), rc, _ = gpu_converge(P, golden24.v, fs, [m["start"]], tts=[tt])
    assert rc == 1
    want, _, _ = oracle.converge(golden24.v, oracle.make_star(golden24.star("818")), m["start"])
    assert_bit_equal(tt, want, "resume")
    before = tt.copy()
    (tt,), rc, st = gpu_converge(P, golden24.v, fs, [m["start"]], tts=[tt])
    assert rc == 0 and st["sweeps_total"] <= 2     # confirming pass (+ one speculative pass)
    assert_bit_equal(tt, before, "idempotent")


@pytest.mark.parametrize("kernel", [1, 2, 3])
def test_convergence_cap_error_leaves_the_context_usable(P, oracle, kernel):
    """An error inside the driver loop (here: the sweep cap, TTSWEEP_OPT_MAX_SWEEPS) is reported,
    the passes already queued are drained, and the same context then solves the same problem
    correctly (the round-1 review's robustness item)."""
    rng = np.random.default_rng(90 + kernel)
    shape = (37, 34, 45)
    v = rng.uniform(0.1, 0.5, size=shape).astype(np.float32)
    offs = P.inputs.read_triples(P.inputs.star_path("six" if kernel == 3 else "3"))
    fs = P.inputs.make_fs(offs)
    starts = np.array([[3, 30, 7], [35, 2, 40]], np.int32)

    def fresh():
        out = []
        for st in starts:
            t = np.full(shape, np.inf, np.float32)
            t[tuple(st)] = 0
            out.append(t)
        return out

    with P.TravelTimeSolver(shape, fs) as sol:
        sol.set_option(P.OPT_KERNEL, kernel)
        sol.set_velocity(v)
        sol.set_option(P.OPT_MAX_SWEEPS, 2)
        with pytest.raises(Exception, match="did not converge"):
            sol.solve(starts, fresh())
        sol.set_option(P.OPT_MAX_SWEEPS, 100000)
        tts = fresh()
        assert sol.solve(starts, tts) == 1
        assert sol.stats()["kernel_variant"] == kernel
    for st, tt in zip(starts, tts):
        want, _, _ = oracle.converge(v, oracle.make_star(offs), st, order=1)
        assert_bit_equal(tt, want, f"kernel {kernel} after the error, start {st}")


def test_sweepXYZ_dropin(P, golden24):
    """ttsweep_sweepXYZ: non-zero on the first call, 0 on the next (drop-in contract)."""
    key = "3_mid"
    fs = P.inputs.make_fs(golden24.star("3"))
    start = golden24.z[f"start_{key}"]
    tt = np.full(golden24.v.shape, np.inf, dtype=np.float32)
    tt[tuple(start)] = 0
    assert P.sweepXYZ(golden24.v, tt, fs, start) > 0
    assert_bit_equal(tt, golden24.z[f"tt_{key}"], key)
    assert P.sweepXYZ(golden24.v, tt, fs, start) == 0


@pytest.mark.parametrize("shape,seed", [((33, 70, 19), 11), ((70, 33, 40), 12), ((5, 4, 3), 13),
                                        ((1, 1, 1), 14), ((2, 300, 2), 15), ((130, 3, 66), 16),
                                        ((70, 66, 130), 17), ((65, 129, 67), 18)])
@pytest.mark.parametrize("kernel", KERNELS)
def test_seeded_random_vs_oracle(P, oracle, shape, seed, kernel):
    """Ragged / tiny / thin grids (smaller than the star radius on some axes), and grids
    with every axis above 64 (several lane tiles per plane: the border zones of the
    activity flags), with random velocities and a random asymmetric star, against the CPU
    oracle."""
    rng = np.random.default_rng(seed)
    v = rng.uniform(0.1, 0.5, size=shape).astype(np.float32)
    offs = rng.integers(-7, 8, size=(60, 3)).astype(np.int32)
    offs = offs[np.any(offs != 0, axis=1)]
    start = [int(rng.integers(0, n)) for n in shape]
    want, _, _ = oracle.converge(v, oracle.make_star(offs), start, order=1)
    (tt,), _, st = gpu_converge(P, v, P.inputs.make_fs(offs), [start], kernel=kernel)
    assert st["kernel_variant"] == (2 if kernel > 20 else kernel)
    assert_bit_equal(tt, want, str(shape))


def test_fuzz_shapes_stars_starts_vs_oracle(P, oracle):
    """Ten seeded random cases - grid shape (1..140 per axis), star (818-FS, 5-FS, random
    asymmetric stars of radius <= 7 or <= 3), 1-3 starts per solve - against the CPU oracle."""
    rng = np.random.default_rng(2026)
    for case in range(10):
        shape = tuple(int(x) for x in rng.integers(1, 90, size=3))
        if rng.random() < 0.3:
            shape = tuple(int(x) for x in rng.integers(60, 140, size=3))
        if np.prod(shape) > 250000:
            shape = (shape[0] // 2 + 1, shape[1], shape[2] // 2 + 1)
        v = rng.uniform(0.05, 0.6, size=shape).astype(np.float32)
        kind = int(rng.integers(0, 4))
        if kind == 0:
            offs = P.inputs.read_triples(P.inputs.star_path("818"))
        elif kind == 1:
            offs = P.inputs.read_triples(P.inputs.star_path("5"))
        else:
            r = 7 if kind == 2 else 3
            offs = rng.integers(-r, r + 1, size=(int(rng.integers(3, 80)), 3)).astype(np.int32)
            offs = offs[np.any(offs != 0, axis=1)]
        nstart = int(rng.integers(1, 4))
        starts = np.stack([rng.integers(0, n, size=nstart) for n in shape], axis=1).astype(np.int32)
        # the library's own choice of kernel, or the unit kernel forced into either mode
        tts, _, st = gpu_converge(P, v, P.inputs.make_fs(offs), starts, kernel=[None, 21, 22][case % 3])
        ofs = oracle.make_star(offs)
        for start, tt in zip(starts, tts):
            want, _, _ = oracle.converge(v, ofs, start, order=1)
            assert_bit_equal(tt, want, f"case {case} {shape} star kind {kind} start {tuple(start)}")


def test_large_radius_star_falls_back_to_cell_kernel(P, oracle):
    """Offsets beyond +-7 (FSRADIUSMAX, serial_new/...:41) are outside the STRIP
    kernel's window; the library must still solve them (CELL kernel)."""
    rng = np.random.default_rng(99)
    shape = (20, 30, 25)
    v = rng.uniform(0.1, 0.5, size=shape).astype(np.float32)
    offs = np.array([[9, 0, 0], [0, -10, 1], [1, 1, 8], [-1, 0, 0], [0, 0, 1], [2, 2, 2]], np.int32)
    start = (3, 4, 5)
    want, _, _ = oracle.converge(v, oracle.make_star(offs), start, order=1)
    (tt,), _, st = gpu_converge(P, v, P.inputs.make_fs(offs), [start])
    assert st["kernel_variant"] == 1
    assert_bit_equal(tt, want, "radius 10")


def test_device_resident_solve(P, golden24):
    """ttsweep_solve_device with init on the device (the bench path)."""
    import torch
    offs = golden24.star("5")
    fs = P.inputs.make_fs(offs)
    keys = ["5_mid", "5_corner"]
    starts = np.array([golden24.z[f"start_{k}"] for k in keys], dtype=np.int32)
    dev = torch.device("cuda:0")
    with P.TravelTimeSolver(golden24.v.shape, fs) as sol:
        sol.set_velocity(torch.from_numpy(golden24.v).to(dev))
        tt = torch.empty((2,) + golden24.v.shape, dtype=torch.float32, device=dev)
        assert sol.solve_device(starts, tt, init=True) == 1
        out = tt.cpu().numpy()
    for n, k in enumerate(keys):
        assert_bit_equal(out[n], golden24.z[f"tt_{k}"], k)


# ---------------------------------------------------------------------------
# BASELINE.json full size: 241 x 241 x 51
# ---------------------------------------------------------------------------

@pytest.fixture(scope="module")
def full(P):
    return P.inputs.velocity_model(241, 241, 51, 20160507)


def test_full_size_digests(P, full):
    """Converged 241x241x51 boxes equal the reference's (SHA-256 recorded from runs of
    the unmodified reference: tests/golden/big_digests.json): 3-FS and 818-FS from start
    (120,120,50) - including the dead-edge cell (113,119,49) - and 818-FS from several
    start points of the BASELINE start-24 file, solved as one batch."""
    digests = json.load(open(os.path.join(GOLDEN, "big_digests.json")))
    by_star = {}
    for key, want in digests.items():
        _, sname, i, j, k = key.split("_")
        by_star.setdefault(sname, []).append(((int(i), int(j), int(k)), want))
    assert "3" in by_star and "818" in by_star
    for sname, cases in by_star.items():
        fs = P.inputs.make_fs(P.inputs.read_triples(P.inputs.star_path(sname)))
        tts, rc, st = gpu_converge(P, full, fs, [c[0] for c in cases])
        assert rc == 1
        for tt, (start, want) in zip(tts, cases):
            for pos, b in want["spots"].items():
                p = tuple(map(int, pos.split(",")))
                assert int(tt[p].view(np.uint32)) == b, (sname, start, p)
            assert hashlib.sha256(tt.tobytes()).hexdigest() == want["sha256"], (sname, start)


@pytest.mark.parametrize("nstart,options", [
    (1, {}),                                                    # the latency instance by the default rule
    (3, {}),                                                    # one GPU's shard of BASELINE config 3
    (3, {"OPT_ASYNC_WAVES": 4}),
    (3, {"OPT_ASYNC_WAVES": 8, "OPT_ASYNC_HANDOFF": 3}),
    (3, {"OPT_ASYNC_WAVES": 4, "OPT_ASYNC_HANDOFF": 1, "OPT_ASYNC_SPECIAL": 1}),
    (8, {"OPT_ASYNC_WAVES": 8, "OPT_ASYNC_HANDOFF": 1}),       # (eight rings, one start each: the case that found the
                                                                #  dead-edge entry without a reserved position)
    (24, {"OPT_ASYNC_HANDOFF": 3}),
], ids=["1-default", "3-default", "3-waves4", "3-waves8-handoff3", "3-waves4-handoff1-special1", "8-waves8-handoff1", "24-handoff3"])
def test_full_size_small_shards_and_handoff(P, full, nstart, options):
    """BASELINE config 3's shards (3 of the 24 starts per GPU) and the other sizes the round-5 schedule treats
    differently - the eight-wave latency instance, direct hand-off - on the full-size grid, device-resident and
    initialised on the device: every box is the reference's, bit for bit (SHA-256 from the unmodified reference's runs,
    tests/golden/big_digests.json), the launch never gives up, and a second solve of the converged boxes stores
    nothing."""
    import torch
    digests = json.load(open(os.path.join(GOLDEN, "big_digests.json")))
    starts = np.asarray(P.inputs.read_triples(P.inputs.starts_path("24")), dtype=np.int32)[:nstart]
    fs = P.inputs.make_fs(P.inputs.read_triples(P.inputs.star_path("818")))
    dev = torch.device("cuda:0")
    tt = torch.empty((nstart,) + full.shape, dtype=torch.float32, device=dev)
    with P.TravelTimeSolver(full.shape, fs) as sol:
        for key, value in options.items():
            sol.set_option(getattr(P, key), value)
        sol.set_velocity(full)
        assert sol.solve_device(starts, tt, init=True) == 1
        st = sol.stats()
        assert st["launches"] == 1 and st["fallbacks"] == 0 and st["kernel_variant"] == 2
        host = tt.cpu().numpy()
        assert sol.solve_device(starts, tt, init=False) == 0
        assert sol.stats()["fallbacks"] == 0
        assert torch.equal(tt.cpu(), torch.from_numpy(host))
    for s, box in zip(starts, host):
        want = digests["syn241_818_%d_%d_%d" % tuple(int(x) for x in s)]["sha256"]
        assert hashlib.sha256(box.tobytes()).hexdigest() == want, tuple(s)


def test_full_size_fixed_point_properties(P, oracle, full):
    """start-4 (the BASELINE config), 818-FS: no INFINITY left, start at 0, a second
    solve changes nothing, and the oracle's validator finds no relaxable edge
    (one CPU pass, ~15 s) for one of the starts."""
    fs = P.inputs.make_fs(P.inputs.read_triples(P.inputs.star_path("818")))
    starts = P.inputs.read_triples(P.inputs.starts_path("4"))
    tts, rc, st = gpu_converge(P, full, fs, starts)
    assert rc == 1
    for s, tt in zip(starts, tts):
        assert np.isfinite(tt).all()
        assert tt[tuple(s)] == 0 and (tt >= 0).all()
    again = [t.copy() for t in tts]
    _, rc2, st2 = gpu_converge(P, full, fs, starts, tts=again)
    assert rc2 == 0 and st2["sweeps_total"] <= 2 * len(starts)
    for a, b in zip(again, tts):
        assert_bit_equal(a, b, "idempotent")
    open_edges, ninf = oracle.validate(full, tts[0], oracle.make_star(
        oracle.read_triples(P.inputs.star_path("818"))), starts[0])
    assert (open_edges, ninf) == (0, 0)


# ---------------------------------------------------------------------------
# BASELINE.json config 4: synthetic 512 x 512 x 256 (no CPU oracle run is feasible:
# one reference sweep is ~5 min).  Size-independent properties instead.
# ---------------------------------------------------------------------------

def sampled_open_edges(v, tt, offs, start, nsample, seed):
    """The reference's own store conditions (serial_new/...:219-249) evaluated with
    numpy float32 arithmetic at `nsample` random centre cells x all star entries
    l < starsize-1.  Returns the number of (cell, offset) pairs a reference sweep
    would still store through (0 at the fixed point)."""
    rng = np.random.default_rng(seed)
    shape = np.array(v.shape)
    L = len(offs) - 1
    d = (np.float32(10.0) * np.sqrt((offs[:L].astype(np.int64) ** 2).sum(1).astype(np.float64))
         .astype(np.float32)).astype(np.float32)
    c = np.stack([rng.integers(0, n, size=nsample) for n in shape], axis=1)
    c = c[~np.all(c == np.asarray(start), axis=1)]                  # centre == start is skipped
    # make sure the neighbourhood of the start and the grid corners are covered
    open_edges = 0
    for lo in range(0, len(c), 2048):
        cc = c[lo:lo + 2048]
        o = cc[:, None, :] + offs[None, :L, :]
        ok = np.all((o >= 0) & (o < shape), axis=2)
        oc = np.clip(o, 0, shape - 1)
        vc = v[cc[:, 0], cc[:, 1], cc[:, 2]][:, None]
        vo = v[oc[..., 0], oc[..., 1], oc[..., 2]]
        tc = tt[cc[:, 0], cc[:, 1], cc[:, 2]][:, None]
        to = tt[oc[..., 0], oc[..., 1], oc[..., 2]]
        delay = ((d[None, :] * (vc + vo).astype(np.float32)).astype(np.float32)
                 * np.float32(0.5)).astype(np.float32)
        bad = ((delay + to).astype(np.float32) < tc) | ((delay + tc).astype(np.float32) < to)
        open_edges += int((bad & ok).sum())
    return open_edges


def test_device_validator_matches_oracle_validator(P, golden24, oracle):
    """ttsweep_validate_device counts exactly what the oracle's validator counts: 0 on
    converged boxes, the same positive numbers on the reference's one- and two-pass
    states (818-FS and the non-symmetric star).  Its third count (cells no store can have
    produced) is 0 on every state the reference passes through and > 0 on a box with a
    value that is too small."""
    import torch
    dev = torch.device("cuda:0")
    for sname in ("818", "nonsym"):
        offs = golden24.star(sname)
        m = golden24.meta[f"pass_{sname}"]
        start = m["start"]
        ofs = oracle.make_star(offs)
        conv, _, _ = oracle.converge(golden24.v, ofs, start)
        with P.TravelTimeSolver(golden24.v.shape, P.inputs.make_fs(offs)) as sol:
            sol.set_velocity(golden24.v)
            for tt in (conv, golden24.z[f"pass1_{sname}"], golden24.z[f"pass2_{sname}"]):
                want = oracle.validate(golden24.v, tt, ofs, start)
                got = sol.validate_device(start, torch.from_numpy(np.ascontiguousarray(tt)).to(dev))
                assert got[:2] == want and got[2] == 0, (sname, got, want)
            assert sol.validate_device(start, torch.from_numpy(conv).to(dev)) == (0, 0, 0)
            # a value below the fixed point: its neighbours can now improve (open edges) and
            # nothing supports the value itself
            bad = conv.copy()
            far = tuple(0 if 2 * s >= n else n - 1 for s, n in zip(start, conv.shape))
            bad[far] = np.float32(0.5) * bad[far]
            opened, _, unsupported = sol.validate_device(start, torch.from_numpy(bad).to(dev))
            assert opened > 0 and unsupported >= 1, (sname, opened, unsupported)


def test_sampled_checker_detects_unconverged_state(P, golden24, oracle):
    """The sampled validator itself: 0 on a converged fixture, > 0 one pass earlier."""
    offs = golden24.star("818")
    m = golden24.meta["pass_818"]
    start = np.array(m["start"])
    conv, _, _ = oracle.converge(golden24.v, oracle.make_star(offs), start)
    assert sampled_open_edges(golden24.v, conv, offs, start, 4000, 1) == 0
    assert sampled_open_edges(golden24.v, golden24.z["pass1_818"], offs, start, 4000, 1) > 0


def test_512_grid_properties(P):
    """818-FS on 512x512x256: the STRIP and CELL kernels (independent implementations,
    different layouts and schedules) agree bit for bit, no INFINITY is left, the start
    stays 0, a second solve changes nothing, and the reference's store conditions hold
    at every cell (device validator) and at 20000 sampled cells (numpy, independent)."""
    import torch
    shape = (512, 512, 256)
    dev = torch.device("cuda:0")
    v_dev = P.inputs.velocity_model_device(*shape, 20160507, dev)
    offs = P.inputs.read_triples(P.inputs.star_path("818"))
    fs = P.inputs.make_fs(offs)
    starts = P.inputs.scaled_starts(P.inputs.read_triples(P.inputs.starts_path("111")), *shape)[:1]
    out = {}
    for kernel in (2, 1):
        with P.TravelTimeSolver(shape, fs) as sol:
            sol.set_option(P.OPT_KERNEL, kernel)
            sol.set_velocity(v_dev)
            tt = torch.empty((1,) + shape, dtype=torch.float32, device=dev)
            assert sol.solve_device(starts, tt, init=True) == 1
            if kernel == 2:
                assert sol.solve_device(starts, tt, init=False) == 0
                assert sol.validate_device(starts[0], tt[0]) == (0, 0, 0)   # every cell, on the device
            out[kernel] = tt[0].cpu().numpy()
    assert np.array_equal(out[1].view(np.uint32), out[2].view(np.uint32))
    tt = out[2]
    assert np.isfinite(tt).all() and tt[tuple(starts[0])] == 0 and (tt >= 0).all()
    v = v_dev.cpu().numpy()
    assert sampled_open_edges(v, tt, offs, starts[0], 20000, 2) == 0
    # every cell once more, by code that shares nothing with the library (plain PyTorch)
    from torch_checker import fixed_point_counts
    assert fixed_point_counts(v_dev, torch.from_numpy(tt).to(dev), fs, starts[0]) == (0, 0, 0)


SHELL26 = np.array([[a, b, c] for a in (-1, 0, 1) for b in (-1, 0, 1) for c in (-1, 0, 1) if (a, b, c) != (0, 0, 0)]
                   + [[1, 1, 1]], np.int32)       # 26-neighbour shell + the entry the exclusive bound drops


@pytest.mark.parametrize("shape,seed", [((33, 70, 19), 31), ((70, 33, 40), 32), ((5, 4, 3), 33), ((1, 1, 1), 34),
                                        ((2, 100, 2), 35), ((9, 8, 65), 36), ((17, 16, 97), 37)])
def test_tile_kernel_small_stars_vs_oracle(P, oracle, shape, seed):
    """TILE kernel (ordered tile sweeps) on ragged / tiny / thin grids with random velocities:
    the 6-neighbour star, the 26-neighbour shell (diagonal offsets: neighbours on the same
    hyperplane), and random asymmetric stars within its reach (entries that are live in one
    direction only: exact liveness path), 1-3 starts, against the CPU oracle."""
    rng = np.random.default_rng(seed)
    v = rng.uniform(0.1, 0.5, size=shape).astype(np.float32)
    six = P.inputs.read_triples(P.inputs.star_path("six"))
    rnd = np.stack([rng.integers(-2, 3, size=9), rng.integers(-2, 3, size=9), rng.integers(-4, 5, size=9)], axis=1)
    rnd = rnd[np.any(rnd != 0, axis=1)].astype(np.int32)
    for name, offs in (("six", six), ("shell26", SHELL26), ("random", rnd)):
        assert tile_supports(offs), name
        nstart = int(rng.integers(1, 4))
        starts = np.stack([rng.integers(0, n, size=nstart) for n in shape], axis=1).astype(np.int32)
        tts, rc, st = gpu_converge(P, v, P.inputs.make_fs(offs), starts, kernel=3)
        assert st["kernel_variant"] == 3
        for start, tt in zip(starts, tts):
            want, _, _ = oracle.converge(v, oracle.make_star(offs), start, order=1)
            assert_bit_equal(tt, want, f"{name} {shape} start {start}")
        # small stars take the TILE kernel by default, and a converged box is left alone
        tts2, rc2, st2 = gpu_converge(P, v, P.inputs.make_fs(offs), starts, tts=[t.copy() for t in tts])
        assert rc2 == 0 and st2["kernel_variant"] == 3
        for a, b in zip(tts, tts2):
            assert_bit_equal(b, a, f"{name} re-solved")


@pytest.mark.parametrize("shape,seed", [((20, 37, 97), 41), ((40, 9, 70), 42)])
def test_tile_kernel_resumes_from_a_damaged_box(P, oracle, shape, seed):
    """TILE kernel started from a box that is not a fresh one (init = 0): a converged box with a
    block reset to INFINITY and another raised is still a state above the fixed point, so the
    solve has to return 1 and restore the converged box bit for bit (tile activity and the z
    faces are then derived from the box, not from the start point)."""
    rng = np.random.default_rng(seed)
    v = rng.uniform(0.1, 0.5, size=shape).astype(np.float32)
    six = P.inputs.read_triples(P.inputs.star_path("six"))
    rnd = np.stack([rng.integers(-2, 3, size=9), rng.integers(-2, 3, size=9), rng.integers(-4, 5, size=9)], axis=1)
    rnd = rnd[np.any(rnd != 0, axis=1)].astype(np.int32)
    for name, offs in (("six", six), ("random", rnd)):
        starts = np.stack([rng.integers(0, n, size=2) for n in shape], axis=1).astype(np.int32)
        fs = P.inputs.make_fs(offs)
        tts, rc, st = gpu_converge(P, v, fs, starts, kernel=3)
        assert st["kernel_variant"] == 3
        damaged = [t.copy() for t in tts]
        for t, start in zip(damaged, starts):
            lo = [int(rng.integers(0, n)) for n in shape]
            hi = [min(n, l + int(rng.integers(1, 40))) for n, l in zip(shape, lo)]
            t[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]] = np.inf
            t[: shape[0] // 2, :, 30:36] *= np.float32(1.5)
            t[tuple(start)] = 0
        fixed, rc2, st2 = gpu_converge(P, v, fs, starts, tts=damaged, kernel=3)
        assert rc2 == 1 and st2["kernel_variant"] == 3
        for a, b, start in zip(fixed, tts, starts):
            assert_bit_equal(a, b, f"{name} {shape} start {start} after damage")


TILE_DRIVERS = [pytest.param({}, id="columns-in-place"), pytest.param({"OPT_TILE_IN_PLACE": 0}, id="columns-padded"),
                pytest.param({"OPT_ASYNC": 0}, id="hyperplane-launches")]


@pytest.mark.parametrize("options", TILE_DRIVERS)
@pytest.mark.parametrize("shape,seed", [((20, 37, 96), 51), ((9, 8, 64), 52), ((8, 8, 32), 53), ((33, 70, 19), 54),
                                        ((17, 16, 97), 55), ((40, 24, 160), 56), ((3, 2, 32), 57)])
def test_tile_six_star_drivers_vs_oracle(P, oracle, shape, seed, options):
    """The plain 6-neighbour star under the TILE kernel's three drivers - column pipelines in ONE launch, relaxing
    in the caller's own arrays (rows of whole tiles: nz % 32 == 0) or in the library's padded volumes, and one launch
    per tile hyperplane - on ragged grids (partial tiles along x and y, rows of one to five tiles): fresh boxes
    initialised on the device, boxes that arrive with values (host boxes), a damaged converged box, and the
    converged boxes solved again; all bit for bit the CPU oracle's fixed point."""
    import torch
    rng = np.random.default_rng(seed)
    v = rng.uniform(0.1, 0.5, size=shape).astype(np.float32)
    six = P.inputs.read_triples(P.inputs.star_path("six"))
    fs = P.inputs.make_fs(six)
    nstart = int(rng.integers(1, 4))
    starts = np.stack([rng.integers(0, n, size=nstart) for n in shape], axis=1).astype(np.int32)
    want = [oracle.converge(v, oracle.make_star(six), st, order=1)[0] for st in starts]
    dev = torch.device("cuda:0")
    with P.TravelTimeSolver(shape, fs) as sol:
        sol.set_option(P.OPT_KERNEL, 3)
        for key, val in options.items():
            sol.set_option(getattr(P, key), val)
        sol.set_velocity(v)
        # fresh boxes, initialised on the device
        tt = torch.empty((nstart,) + shape, dtype=torch.float32, device=dev)
        assert sol.solve_device(starts, tt, init=True) == 1
        st = sol.stats()
        assert st["kernel_variant"] == 3 and st["fallbacks"] == 0
        assert (st["launches"] == 1) == ("OPT_ASYNC" not in options)
        for s in range(nstart):
            assert_bit_equal(tt[s].cpu().numpy(), want[s], f"{shape} start {starts[s]} (device boxes)")
        # the converged boxes again: nothing to do
        assert sol.solve_device(starts, tt, init=False) == 0
        for s in range(nstart):
            assert_bit_equal(tt[s].cpu().numpy(), want[s], f"{shape} start {starts[s]} (second solve)")
        # a damaged box: a block back at INFINITY, a slab raised
        dmg = tt.clone()
        lo = [int(rng.integers(0, n)) for n in shape]
        dmg[:, lo[0]:lo[0] + 9, lo[1]:lo[1] + 5, lo[2]:lo[2] + 40] = float("inf")
        dmg[:, : max(shape[0] // 2, 1), :, : shape[2] // 2] *= 1.5
        for s in range(nstart):
            dmg[s][tuple(starts[s])] = 0
        assert sol.solve_device(starts, dmg, init=False) == 1
        for s in range(nstart):
            assert_bit_equal(dmg[s].cpu().numpy(), want[s], f"{shape} start {starts[s]} (after damage)")
        # host boxes (staged by the library, solved as boxes that arrive with values)
        boxes = []
        for st_ in starts:
            b = np.full(shape, np.inf, dtype=np.float32)
            b[tuple(st_)] = 0
            boxes.append(b)
        assert sol.solve(starts, boxes) == 1
        for s in range(nstart):
            assert_bit_equal(boxes[s], want[s], f"{shape} start {starts[s]} (host boxes)")


@pytest.mark.parametrize("kernel,star", [(1, "3"), (21, "3"), (23, "3"), (3, "six"), (30, "six")])
def test_per_start_outcome(P, kernel, star):
    """ttsweep_get_changed: what the reference's driver prints and sums per start (serial_new/...:158-164) -
    1 exactly for the boxes a call improved: all of a fresh batch, none of a converged one, and exactly the
    damaged ones of a batch in which some boxes arrive converged (the call's return value is their OR)."""
    import torch
    shape = (40, 33, 64)
    rng = np.random.default_rng(77)
    v = rng.uniform(0.1, 0.5, size=shape).astype(np.float32)
    fs = P.inputs.make_fs(P.inputs.read_triples(P.inputs.star_path(star)))
    starts = np.array([[3, 4, 5], [20, 20, 40], [39, 0, 63], [10, 30, 1]], np.int32)
    dev = torch.device("cuda:0")
    with P.TravelTimeSolver(shape, fs) as sol:
        if kernel in (21, 23):
            sol.set_option(P.OPT_KERNEL, 2)
            sol.set_option(P.OPT_ASYNC, 1 if kernel == 21 else 0)
        elif kernel == 30:
            sol.set_option(P.OPT_KERNEL, 3)
            sol.set_option(P.OPT_ASYNC, 0)
        else:
            sol.set_option(P.OPT_KERNEL, kernel)
        sol.set_velocity(v)
        tt = torch.empty((4,) + shape, dtype=torch.float32, device=dev)
        assert sol.solve_device(starts, tt, init=True) == 1
        assert sol.changed(4) == [1, 1, 1, 1]
        good = tt.clone()
        assert sol.solve_device(starts, tt, init=False) == 0
        assert sol.changed(4) == [0, 0, 0, 0]
        tt[1, 5:20, 3:9, 10:50] = float("inf")
        tt[3, :, :, 30:] *= 1.25
        tt[3][tuple(starts[3])] = 0
        assert sol.solve_device(starts, tt, init=False) == 1
        assert sol.changed(4) == [0, 1, 0, 1]
        assert torch.equal(tt.view(torch.int32), good.view(torch.int32))
        # host boxes: the same through ttsweep_solve (staged batches), and the confirming call
        boxes = [good[s].cpu().numpy().copy() for s in range(4)]
        boxes[2][0:7, :, :] = np.inf
        boxes[2][tuple(starts[2])] = 0
        assert sol.solve(starts, boxes) == 1
        assert sol.changed(4) == [0, 0, 1, 0]
        assert sol.solve(starts, boxes) == 0
        assert sol.changed(4) == [0, 0, 0, 0]
        assert sol.changed(2) == [0, 0]


@pytest.mark.parametrize("flags,devices,want_path", [(0, [0], "none"), (1, [0], "rccl-or-peer"), (3, [0], "peer"),
                                                     (0, [0, 0], "peer"), (1, [0, 0, 0], "peer")])
def test_solve_multi_device_gathers_on_the_root(P, golden24, flags, devices, want_path):
    """ttsweep_solve_multi_device (the C host's multi-GPU form with the result set resident on the root device):
    on the one GPU of this tier the device list is {0} - every box is solved in its slot -, {0} with the loop-back
    flag - the boxes travel through ONE RCCL group of ncclSend / ncclRecv pairs to the root itself (or through peer
    copies when RCCL is not to be had) -, and lists that name the device more than once (RCCL refuses them: peer
    copies).  Always: the golden boxes bit for bit, per-start outcome 1, in the caller's device array."""
    import torch
    v = golden24.v
    fs = P.inputs.make_fs(golden24.star("818"))
    keys = ["818_mid", "818_corner", "818_deadin", "818_deadout"]
    starts = np.asarray([golden24.z[f"start_{k}"] for k in keys], np.int32).reshape(-1, 3)
    want = [golden24.z[f"tt_{k}"] for k in keys]
    n = len(starts)
    tt = torch.full((n,) + v.shape, -1.0, dtype=torch.float32, device="cuda:0")
    rc, changed, path = P.solver.solve_multi_device(devices, v, fs, starts, tt, flags=flags)
    assert rc == 1 and changed == [1] * n
    names = {P.solver.GATHER_NONE: "none", P.solver.GATHER_RCCL: "rccl", P.solver.GATHER_PEER: "peer"}
    assert names[path] in want_path.split("-or-"), (names[path], want_path)
    for s in range(n):
        assert_bit_equal(tt[s].cpu().numpy(), want[s], f"start {starts[s]} devices {devices} flags {flags}")


def test_rccl_one_rank_group(P, golden24):
    """First contact with RCCL on the GPU tier: a process group of ONE rank under backend "nccl" (= RCCL on ROCm; all a
    one-GPU box offers).  multistart.solve_star_split runs its two all-reduces (MAX on the int flag, MIN on the float
    box) through it, multistart.gather_boxes its grouped send / receive pairs (batch_isend_irecv: the root sends its
    boxes to itself) - results bit for bit the golden boxes."""
    import socket
    import torch
    import torch.distributed as dist
    v = golden24.v
    offs = golden24.star("818")
    fs = P.inputs.make_fs(offs)
    keys = ["818_mid", "818_corner"]
    starts = [golden24.z[f"start_{k}"] for k in keys]
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    assert not dist.is_initialized()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        dev = torch.device("cuda:0")
        with P.TravelTimeSolver(v.shape, fs) as sol:
            sol.set_velocity(v)
            # one start, the star "split" over the one rank: every round ends with the real collectives
            box = torch.full(v.shape, float("inf"), dtype=torch.float32, device=dev)
            box[tuple(starts[0])] = 0
            rounds = P.multistart.solve_star_split(
                box, lambda b: sol.solve_device([starts[0]], b.unsqueeze(0), init=False) == 1, dist, collectives=True)
            assert rounds == 2
            assert_bit_equal(box.cpu().numpy(), golden24.z[f"tt_{keys[0]}"], "star split, one-rank nccl group")
            # two starts solved here, gathered "to the root" through RCCL send / receive pairs
            local = torch.empty((2,) + v.shape, dtype=torch.float32, device=dev)
            assert sol.solve_device(starts, local, init=True) == 1
            out = P.multistart.gather_boxes(local, 2, dist, path="device", loopback=True)
            assert out is not None and out.data_ptr() != local.data_ptr()
            for s, k in enumerate(keys):
                assert_bit_equal(out[s].cpu().numpy(), golden24.z[f"tt_{k}"], f"gathered {k}")
    finally:
        dist.destroy_process_group()


def test_tile_kernel_512_grid_matches_cell_kernel(P):
    """The HBM-bound regime at size: 6-neighbour star on 512x512x256, two starts.  TILE
    (ordered sweeps) and CELL (one hop per pass, an independent implementation) agree bit for
    bit; the device validator finds nothing to improve and nothing too small; a second solve
    reports no change."""
    import torch
    shape = (512, 512, 256)
    dev = torch.device("cuda:0")
    v_dev = P.inputs.velocity_model_device(*shape, 20160507, dev)
    fs = P.inputs.make_fs(P.inputs.read_triples(P.inputs.star_path("six")))
    starts = P.inputs.scaled_starts(P.inputs.read_triples(P.inputs.starts_path("111")), *shape)[:2]
    out = {}
    for kernel in (3, 1):
        with P.TravelTimeSolver(shape, fs) as sol:
            sol.set_option(P.OPT_KERNEL, kernel)
            sol.set_velocity(v_dev)
            tt = torch.empty((2,) + shape, dtype=torch.float32, device=dev)
            assert sol.solve_device(starts, tt, init=True) == 1
            assert sol.stats()["kernel_variant"] == kernel
            if kernel == 3:
                for s in range(2):
                    assert sol.validate_device(starts[s], tt[s]) == (0, 0, 0)
                assert sol.solve_device(starts, tt, init=False) == 0
            out[kernel] = tt.clone()
    assert torch.equal(out[1].view(torch.int32), out[3].view(torch.int32))
    assert bool(torch.isfinite(out[3]).all()) and float(out[3].min()) == 0.0


def test_tile_kernel_1024_grid_six_star(P):
    """BASELINE.json config 5's "HBM-roofline run": the 6-neighbour star on 1024x1024x512
    (volumes of 2.2 GB), two starts in one batch, pinned by the device validator and by the
    plain-PyTorch checker (every cell, both) and by idempotence."""
    st = _batched_config_check(P, (1024, 1024, 512), 2, star="six", torch_checked=2)
    assert st["kernel_variant"] == 3


def _batched_config_check(P, shape, nstart, star="818", torch_checked=0):
    """One batched device-resident solve of `nstart` scaled start-111 points on a synthetic
    grid; every box is then pinned by the device validator (nothing can improve, nothing is
    too small, nothing left at INFINITY) - the first `torch_checked` boxes also by the
    plain-PyTorch checker, which shares no code with the library -, and a second solve of the
    converged boxes must report "no change" and leave every bit alone."""
    import torch
    dev = torch.device("cuda:0")
    v_dev = P.inputs.velocity_model_device(*shape, 20160507, dev)
    fs = P.inputs.make_fs(P.inputs.read_triples(P.inputs.star_path(star)))
    starts = P.inputs.scaled_starts(P.inputs.read_triples(P.inputs.starts_path("111")), *shape)[:nstart]
    assert len(starts) == nstart
    with P.TravelTimeSolver(shape, fs) as sol:
        sol.set_velocity(v_dev)
        tt = torch.empty((nstart,) + shape, dtype=torch.float32, device=dev)
        assert sol.solve_device(starts, tt, init=True) == 1
        st = sol.stats()
        assert st["nstart"] == nstart
        for s in range(nstart):
            assert sol.validate_device(starts[s], tt[s]) == (0, 0, 0), f"start {s} {starts[s]}"
            assert float(tt[s][tuple(starts[s])]) == 0.0
        from torch_checker import fixed_point_counts
        for s in range(min(torch_checked, nstart)):
            assert fixed_point_counts(v_dev, tt[s], fs, starts[s]) == (0, 0, 0), f"start {s} {starts[s]} (torch)"
        assert bool(torch.isfinite(tt).all()) and float(tt.min()) == 0.0
        digest = [int(tt[s].view(torch.int32).to(torch.int64).sum().item()) for s in range(nstart)]
        assert sol.solve_device(starts, tt, init=False) == 0
        assert digest == [int(tt[s].view(torch.int32).to(torch.int64).sum().item()) for s in range(nstart)]
        return st


def test_config_512_grid_eight_starts(P):
    """BASELINE.json config 4 at its stated batch: 512x512x256, 8 starts, 818-FS, one
    batched solve (8 travel-time volumes resident, work list of 8 x the unit grid)."""
    _batched_config_check(P, (512, 512, 256), 8, torch_checked=2)


def test_config_1024_grid_one_gpu_share(P):
    """BASELINE.json config 5, one GPU's share of the 111 starts (ceil(111 / 8) = 14) on
    1024x1024x512 with the 818-FS star: 14 padded volumes of 2.3 GB in one batch."""
    _batched_config_check(P, (1024, 1024, 512), 14)


def test_volume_above_2_gib(P):
    """818-FS on 1024x1024x512 (padded volumes of 2.3 GB: byte offsets beyond 2^31 inside one
    volume).  No second relaxation is fast enough here, so the box is pinned by two independent
    fixed-point checks over every cell and offset - the library's device validator and a
    plain-PyTorch restatement of the reference's store conditions -: nothing can improve and
    nothing is too small, i.e. it is the fixed point; plus the size-independent properties
    (start 0, finite, non-negative, a second solve changes nothing)."""
    import torch
    shape = (1024, 1024, 512)
    dev = torch.device("cuda:0")
    v_dev = P.inputs.velocity_model_device(*shape, 20160507, dev)
    fs = P.inputs.make_fs(P.inputs.read_triples(P.inputs.star_path("818")))
    starts = P.inputs.scaled_starts(P.inputs.read_triples(P.inputs.starts_path("24")), *shape)[:1]
    with P.TravelTimeSolver(shape, fs) as sol:
        sol.set_velocity(v_dev)
        tt = torch.empty((1,) + shape, dtype=torch.float32, device=dev)
        assert sol.solve_device(starts, tt, init=True) == 1
        assert sol.stats()["kernel_variant"] == 2
        assert sol.validate_device(starts[0], tt[0]) == (0, 0, 0)
        # ... and by the plain-PyTorch restatement of the reference's store conditions (every cell,
        # every offset; tests/torch_checker.py, pinned against the oracle on the CPU tier)
        from torch_checker import fixed_point_counts
        assert fixed_point_counts(v_dev, tt[0], fs, starts[0]) == (0, 0, 0)
        assert float(tt[0][tuple(starts[0])]) == 0.0
        assert bool(torch.isfinite(tt).all()) and float(tt.min()) == 0.0
        before = tt.clone()
        assert sol.solve_device(starts, tt, init=False) == 0
        assert torch.equal(before.view(torch.int32), tt.view(torch.int32))


@pytest.mark.parametrize("sname,key,nslice", [("818", "818_mid", 3), ("5", "5_corner", 2),
                                             ("818", "818_deadin", 8)])
def test_star_split_single_start(P, golden24, sname, key, nslice):
    """One start, the star split over `nslice` contexts (the single-start multi-GPU
    scheme; the "ranks" are played in one process here): each context
    relaxes only offsets [lo, hi), rounds are min-combined, and the result is the fixture
    of the whole star, bit for bit."""
    import torch
    dev = torch.device("cuda:0")
    offs = golden24.star(sname)
    fs = P.inputs.make_fs(offs)
    start = golden24.z[f"start_{key}"]
    shape = golden24.v.shape
    sols, fns = [], []
    try:
        for lo, hi in P.multistart.star_slices(len(fs) - 1, nslice):
            sol = P.TravelTimeSolver(shape, fs, starstart=lo, starstop=hi)
            sol.set_velocity(golden24.v)
            sols.append(sol)
            fns.append(lambda box, sol=sol: sol.solve_device([start], box[None], init=False) == 1)
        tt = np.full(shape, np.inf, dtype=np.float32)
        tt[tuple(start)] = 0
        box = torch.from_numpy(tt).to(dev)
        rounds = P.multistart.solve_star_split_local(box, fns)
    finally:
        for sol in sols:
            sol.close()
    assert rounds >= 2
    assert_bit_equal(box.cpu().numpy(), golden24.z[f"tt_{key}"], key)


@pytest.mark.parametrize("speed,r0", [(0, 0), (500, 1000), (2000, 3000), (9000, 30000)])
def test_result_does_not_depend_on_the_gate(P, golden24, speed, r0):
    """The distance gate only decides WHEN a unit is relaxed: switched off, crawling or far
    ahead of the front, the converged boxes are the same bits (818-FS, interior, corner and
    dead-edge starts in one batch)."""
    fs = P.inputs.make_fs(golden24.star("818"))
    keys = ["818_mid", "818_corner", "818_deadin", "818_deadout"]
    starts = np.array([golden24.z[f"start_{k}"] for k in keys], dtype=np.int32)
    tts = []
    for st in starts:
        tt = np.full(golden24.v.shape, np.inf, dtype=np.float32)
        tt[tuple(st)] = 0
        tts.append(tt)
    with P.TravelTimeSolver(golden24.v.shape, fs) as sol:
        sol.set_option(P.OPT_GATE_SPEED_MILLI, speed)
        sol.set_option(P.OPT_GATE_R0_MILLI, r0)
        sol.set_velocity(golden24.v)
        assert sol.solve(starts, tts) == 1 and sol.stats()["kernel_variant"] == 2
    for k, tt in zip(keys, tts):
        assert_bit_equal(tt, golden24.z[f"tt_{k}"], f"{k} gate {speed}/{r0}")


SCHEDULES = [
    # (one launch per solve?, {option: value}) - every knob that decides WHEN a unit is relaxed or told
    (1, {}),
    (1, {"OPT_ASYNC_POLICY": 0}),
    (1, {"OPT_ASYNC_POLICY": 0, "OPT_ASYNC_LOW": 1, "OPT_ASYNC_HIGH": 2}),
    (1, {"OPT_ASYNC_POLICY": 1, "OPT_ASYNC_GATE_MILLI": 100, "OPT_ASYNC_LOW": 200, "OPT_ASYNC_HIGH": 4000}),
    (1, {"OPT_ASYNC_POLICY": 1, "OPT_ASYNC_GATE_MILLI": 20000, "OPT_ASYNC_SPECIAL": 1}),
    (1, {"OPT_ASYNC_POLICY": 2, "OPT_ASYNC_WINDOW_MILLI": 3000, "OPT_ASYNC_SPECIAL": 1 << 20}),
    (1, {"OPT_ASYNC_POLICY": 2, "OPT_ASYNC_WINDOW_MILLI": 0}),
    (1, {"OPT_DEFER_MARGIN_MILLI": -1000000000}),
    (1, {"OPT_DEFER_MARGIN_MILLI": -4000}),
    (1, {"OPT_DEFER_MARGIN_MILLI": 0, "OPT_GATE_SPEED_MILLI": 0}),
    (1, {"OPT_DEFER_MARGIN_MILLI": 6000, "OPT_PAIR_MIN_STARTS": 0}),
    (1, {"OPT_ASYNC_INUNIT": 0}),
    (1, {"OPT_ASYNC_INUNIT": 1, "OPT_PAIR_MIN_STARTS": 0}),
    (1, {"OPT_ASYNC_INUNIT": 8, "OPT_DEFER_MARGIN_MILLI": 1000}),
    (1, {"OPT_ASYNC_INUNIT": 3, "OPT_PAIR_MIN_STARTS": 0, "OPT_ASYNC_POLICY": 0}),
    # direct hand-off (round 5): workers publish successor units (1), their own unit (2), both; with every ring policy
    # that allows it, a tiny ring fill, no gate, eager deferral, in-unit passes, one- and two-plane units
    (1, {"OPT_ASYNC_HANDOFF": 1}),
    (1, {"OPT_ASYNC_HANDOFF": 2}),
    (1, {"OPT_ASYNC_HANDOFF": 3}),
    (1, {"OPT_ASYNC_HANDOFF": 3, "OPT_PAIR_MIN_STARTS": 0}),
    (1, {"OPT_ASYNC_HANDOFF": 3, "OPT_ASYNC_POLICY": 0, "OPT_ASYNC_LOW": 1, "OPT_ASYNC_HIGH": 2}),
    (1, {"OPT_ASYNC_HANDOFF": 3, "OPT_GATE_SPEED_MILLI": 0, "OPT_DEFER_MARGIN_MILLI": -4000}),
    (1, {"OPT_ASYNC_HANDOFF": 1, "OPT_ASYNC_GATE_MILLI": 100, "OPT_ASYNC_GATE_FAST_MILLI": 100, "OPT_ASYNC_INUNIT": 2}),
    (1, {"OPT_ASYNC_HANDOFF": 3, "OPT_ASYNC_SPECIAL": 1, "OPT_DEFER_MARGIN_MILLI": -1000000000, "OPT_QUEUES": 1}),
    (1, {"OPT_ASYNC_HANDOFF": 3, "OPT_ASYNC_POLICY": 2, "OPT_ASYNC_WINDOW_MILLI": 3000}),     # (policy 2: hand-off stays off)
    # the latency instance (round 5): a unit relaxed by eight waves, four slabs in flight (one-plane units only: with
    # units of two planes the option is ignored)
    (1, {"OPT_ASYNC_WAVES": 8}),
    (1, {"OPT_ASYNC_WAVES": 8, "OPT_ASYNC_INUNIT": 2}),
    (1, {"OPT_ASYNC_WAVES": 8, "OPT_ASYNC_HANDOFF": 3, "OPT_ASYNC_INUNIT": 3, "OPT_DEFER_MARGIN_MILLI": 0}),
    (1, {"OPT_ASYNC_WAVES": 8, "OPT_ASYNC_POLICY": 0, "OPT_ASYNC_LOW": 1, "OPT_ASYNC_HIGH": 2, "OPT_ASYNC_SPECIAL": 1}),
    (1, {"OPT_ASYNC_WAVES": 8, "OPT_PAIR_MIN_STARTS": 0}),
    (1, {"OPT_ASYNC_WAVES": 4, "OPT_PAIR_MIN_STARTS": 1000000}),
    (0, {"OPT_DEFER_MARGIN_MILLI": -1000000000}),
    (0, {"OPT_DEFER_MARGIN_MILLI": -4000}),
    (0, {"OPT_DEFER_MARGIN_MILLI": 0, "OPT_PAIR_MIN_STARTS": 0}),
    (0, {"OPT_DEFER_MARGIN_MILLI": 3000, "OPT_GATE_SPEED_MILLI": 700}),
]


@pytest.mark.parametrize("one_launch,options", SCHEDULES, ids=[f"{a}-{'-'.join(f'{k[4:]}={v}' for k, v in o.items()) or 'defaults'}"
                                                           for a, o in SCHEDULES])
def test_result_does_not_depend_on_the_schedule(P, golden24, one_launch, options):
    """The STRIP kernel's schedule - a solve as one launch (ring policies, fill marks, gate per round, window,
    how often the dead-edge cells are relaxed, how often a unit that improved is relaxed again against its own
    planes before it is handed back) or as a launch pair per pass, and the deferral of the
    bits for units behind the front (off, eager, far too eager: the late relaxations then improve cells and
    the solve goes on) - never changes a bit of the converged boxes: 818-FS, interior, corner and dead-edge
    starts in one batch, against the reference's boxes."""
    fs = P.inputs.make_fs(golden24.star("818"))
    keys = ["818_mid", "818_corner", "818_deadin", "818_deadout"]
    starts = np.array([golden24.z[f"start_{k}"] for k in keys], dtype=np.int32)
    tts = []
    for st in starts:
        tt = np.full(golden24.v.shape, np.inf, dtype=np.float32)
        tt[tuple(st)] = 0
        tts.append(tt)
    with P.TravelTimeSolver(golden24.v.shape, fs) as sol:
        sol.set_option(P.OPT_ASYNC, one_launch)
        for key, value in options.items():
            sol.set_option(getattr(P, key), value)
        sol.set_velocity(golden24.v)
        assert sol.solve(starts, tts) == 1
        st = sol.stats()
        assert st["kernel_variant"] == 2 and (st["launches"] == 1) == bool(one_launch)
        # ... and the converged boxes are recognised as such by the same schedule
        again = [t.copy() for t in tts]
    for k, tt in zip(keys, tts):
        assert_bit_equal(tt, golden24.z[f"tt_{k}"], f"{k} {one_launch} {options}")
    with P.TravelTimeSolver(golden24.v.shape, fs) as sol:
        sol.set_option(P.OPT_ASYNC, one_launch)
        for key, value in options.items():
            sol.set_option(getattr(P, key), value)
        sol.set_velocity(golden24.v)
        assert sol.solve(starts, again) == 0        # (a fresh context: nothing is answered from digests)
    for a, b in zip(again, tts):
        assert_bit_equal(a, b, "idempotent")


def test_one_launch_solve_that_gives_up_is_finished_by_the_pass_driver(P, full):
    """Every wait inside the one-launch solve has a wall-clock limit, so that a protocol error could not
    hang the device.  Here the limit is set absurdly low (1 ms; the 4-start full-size solve takes ~10): the
    launch drains with the boxes somewhere on their way, the pass driver takes over from them, and the
    converged boxes are the reference's, bit for bit (SHA-256 of the reference's own runs)."""
    digests = json.load(open(os.path.join(GOLDEN, "big_digests.json")))
    cases = []
    for key, want in digests.items():
        _, sname, i, j, k = key.split("_")
        if sname == "818" and len(cases) < 4:
            cases.append(((int(i), int(j), int(k)), want["sha256"]))
    fs = P.inputs.make_fs(P.inputs.read_triples(P.inputs.star_path("818")))
    starts = np.array([c[0] for c in cases], dtype=np.int32)
    tts = []
    for st in starts:
        tt = np.full(full.shape, np.inf, dtype=np.float32)
        tt[tuple(st)] = 0
        tts.append(tt)
    with P.TravelTimeSolver(full.shape, fs) as sol:
        sol.set_option(P.OPT_ASYNC, 1)
        sol.set_option(P.OPT_ASYNC_TIMEOUT_MILLI, 1)
        sol.set_velocity(full)
        assert sol.solve(starts, tts) == 1
        assert sol.stats()["launches"] > 2          # (the one launch, then passes)
    for tt, (start, want) in zip(tts, cases):
        assert hashlib.sha256(tt.tobytes()).hexdigest() == want, start


@pytest.mark.parametrize("nstart", [255, 256])
def test_many_starts_on_a_small_grid(P, nstart):
    """255 starts are the most a one-launch solve takes (eight rings of up to 32 starts; a ring entry holds the
    start in 8 bits), 256 go to the pass driver: both against the CELL kernel, bit for bit, duplicates and
    corner starts included."""
    rng = np.random.default_rng(77)
    shape = (21, 18, 12)
    v = rng.uniform(0.2, 1.0, size=shape).astype(np.float32)
    fs = P.inputs.make_fs(P.inputs.read_triples(P.inputs.star_path("5")))
    starts = np.stack([rng.integers(0, n, size=nstart) for n in shape], axis=1).astype(np.int32)
    starts[0] = (0, 0, 0)
    starts[1] = (20, 17, 11)
    starts[2] = starts[3]

    def fresh():
        out = []
        for st in starts:
            t = np.full(shape, np.inf, np.float32)
            t[tuple(st)] = 0
            out.append(t)
        return out

    ref, _, _ = gpu_converge(P, v, fs, starts, tts=fresh(), kernel=1)
    got, rc, st = gpu_converge(P, v, fs, starts, tts=fresh(), kernel=2)
    assert rc == 1 and st["kernel_variant"] == 2 and (st["launches"] == 1) == (nstart <= 255)
    for a, b in zip(got, ref):
        assert_bit_equal(a, b, f"{nstart} starts")


@pytest.mark.parametrize("queues,nstart,one_launch", [(1, 32, True), (1, 40, False), (2, 64, True), (2, 70, False)])
def test_starts_per_ring_limit_with_few_queues(P, queues, nstart, one_launch):
    """A device (or partition) with fewer than 8 XCDs has fewer planner rings, and a ring serves at most 32 starts:
    beyond that the launch-per-pass driver has to run - not an error (TTSWEEP_OPT_QUEUES forces the queue count
    the census would find there).  Both drivers against the CELL kernel, bit for bit."""
    rng = np.random.default_rng(79)
    shape = (21, 18, 12)
    v = rng.uniform(0.2, 1.0, size=shape).astype(np.float32)
    fs = P.inputs.make_fs(P.inputs.read_triples(P.inputs.star_path("3")))
    starts = np.stack([rng.integers(0, n, size=nstart) for n in shape], axis=1).astype(np.int32)
    ref, _, _ = gpu_converge(P, v, fs, starts, kernel=1)
    with P.TravelTimeSolver(shape, fs) as sol:
        sol.set_option(P.OPT_KERNEL, 2)
        sol.set_option(P.OPT_QUEUES, queues)
        sol.set_velocity(v)
        got = _boxes(shape, starts)
        assert sol.solve(starts, got) == 1
        st = sol.stats()
        assert st["kernel_variant"] == 2 and st["fallbacks"] == 0 and (st["launches"] == 1) == one_launch
    for a, b in zip(got, ref):
        assert_bit_equal(a, b, f"{nstart} starts, {queues} queue(s)")
    with pytest.raises(Exception):
        with P.TravelTimeSolver(shape, fs) as sol:
            sol.set_option(P.OPT_QUEUES, 9)


@pytest.mark.parametrize("queues", [1, 3, 8])
def test_column_driver_with_fewer_claim_sequences(P, oracle, queues):
    """The column driver deals the tile positions over one claim sequence per XCD; a partition shows fewer XCDs
    (TTSWEEP_OPT_QUEUES forces the count): one, three and eight sequences, the same fixed point, bit for bit."""
    import torch
    shape = (70, 33, 96)
    rng = np.random.default_rng(61)
    v = rng.uniform(0.1, 0.5, size=shape).astype(np.float32)
    six = P.inputs.read_triples(P.inputs.star_path("six"))
    fs = P.inputs.make_fs(six)
    starts = np.array([[5, 30, 90], [69, 0, 0]], dtype=np.int32)
    want = [oracle.converge(v, oracle.make_star(six), st, order=1)[0] for st in starts]
    with P.TravelTimeSolver(shape, fs) as sol:
        sol.set_option(P.OPT_KERNEL, 3)
        sol.set_option(P.OPT_QUEUES, queues)
        sol.set_velocity(v)
        tt = torch.empty((2,) + shape, dtype=torch.float32, device=torch.device("cuda:0"))
        assert sol.solve_device(starts, tt, init=True) == 1
        st = sol.stats()
        assert st["kernel_variant"] == 3 and st["launches"] == 1 and st["fallbacks"] == 0
        for s in range(2):
            assert_bit_equal(tt[s].cpu().numpy(), want[s], f"{queues} sequences, start {starts[s]}")


@pytest.mark.parametrize("order", [-1, 0, 1, 2, 3, 5, 11, 24, 111, 114, 115, 204, 311, 412, 429, 117, 18, 119])
def test_column_driver_sequences_of_orderings(P, oracle, order):
    """TTSWEEP_OPT_TILE_ORDER: the sequence of the eight orderings each start's sweeps follow (which table, from which
    corner, which axis in which role - column_order_sequence) changes how many sweeps and how much work a solve is,
    never the fixed point: ragged grids, starts in corners, on faces and in the middle, fresh and damaged boxes."""
    import torch
    six = P.inputs.read_triples(P.inputs.star_path("six"))
    fs = P.inputs.make_fs(six)
    for shape, seed in (((70, 33, 96), 71), ((24, 41, 160), 72)):
        rng = np.random.default_rng(seed)
        v = rng.uniform(0.1, 0.5, size=shape).astype(np.float32)
        if order < 0:       # the default looks at the velocity profile of a start's vertical line: give it one (faster with depth)
            v = (v * 0.05 + np.linspace(0.15, 0.4, shape[2], dtype=np.float32)[None, None, :]).astype(np.float32)
        starts = np.array([[0, 0, 0], [shape[0] - 1, shape[1] - 1, shape[2] - 1], [shape[0] // 2, 3, shape[2] - 2],
                           [shape[0] // 3, shape[1] // 2, shape[2] // 2], [shape[0] - 2, 1, 40]], dtype=np.int32)
        want = [oracle.converge(v, oracle.make_star(six), st, order=1)[0] for st in starts]
        with P.TravelTimeSolver(shape, fs) as sol:
            sol.set_option(P.OPT_KERNEL, 3)
            sol.set_option(P.OPT_TILE_ORDER, order)
            sol.set_velocity(v)
            tt = torch.empty((len(starts),) + shape, dtype=torch.float32, device=torch.device("cuda:0"))
            assert sol.solve_device(starts, tt, init=True) == 1
            st = sol.stats()
            assert st["kernel_variant"] == 3 and st["launches"] == 1 and st["fallbacks"] == 0
            for s in range(len(starts)):
                assert_bit_equal(tt[s].cpu().numpy(), want[s], f"order {order}, {shape}, start {starts[s]}")
            assert sol.solve_device(starts, tt, init=False) == 0
            tt[:, : shape[0] // 2, :, 10:50] *= 1.25
            tt[:, 3:9, 2:30, :] = float("inf")
            for s in range(len(starts)):
                tt[s][tuple(starts[s])] = 0
            assert sol.solve_device(starts, tt, init=False) == 1
            for s in range(len(starts)):
                assert_bit_equal(tt[s].cpu().numpy(), want[s], f"order {order}, {shape}, start {starts[s]} (after damage)")
    with pytest.raises(Exception):
        with P.TravelTimeSolver((8, 8, 32), fs) as sol:
            sol.set_option(P.OPT_TILE_ORDER, 131)


def test_column_rest_is_declared_once_per_start(P):
    """Three starts that come to rest in different sweeps, many times over: successive sweeps of the column driver
    overlap, and two of them can both end without an improvement before either hears of the other - the start
    must still count as ONE finished start (counted twice, the launch ended while the third start was still being
    relaxed: 9 cells short of the fixed point in two of five runs)."""
    import torch
    shape, seed = (40, 24, 160), 56
    rng = np.random.default_rng(seed)
    v = rng.uniform(0.1, 0.5, size=shape).astype(np.float32)
    fs = P.inputs.make_fs(P.inputs.read_triples(P.inputs.star_path("six")))
    nstart = int(rng.integers(1, 4))
    starts = np.stack([rng.integers(0, n, size=nstart) for n in shape], axis=1).astype(np.int32)
    assert nstart == 3
    dev = torch.device("cuda:0")
    with P.TravelTimeSolver(shape, fs) as ref:
        ref.set_option(P.OPT_KERNEL, 3)
        ref.set_option(P.OPT_ASYNC, 0)
        ref.set_velocity(v)
        want = torch.empty((nstart,) + shape, dtype=torch.float32, device=dev)
        assert ref.solve_device(starts, want, init=True) == 1
    for in_place in (1, 0):
        with P.TravelTimeSolver(shape, fs) as sol:
            sol.set_option(P.OPT_KERNEL, 3)
            sol.set_option(P.OPT_TILE_IN_PLACE, in_place)
            sol.set_velocity(v)
            for rep in range(40):
                tt = torch.empty((nstart,) + shape, dtype=torch.float32, device=dev)
                assert sol.solve_device(starts, tt, init=True) == 1
                st = sol.stats()
                assert st["launches"] == 1 and st["fallbacks"] == 0
                assert torch.equal(tt, want), f"repetition {rep}, in place {in_place}"


def test_converged_box_with_one_finite_unit_reports_no_change(P, oracle):
    """A star that only moves along z reaches one z-column: every finite cell sits in ONE
    activity unit, which is exactly the case the distance gate treats as "grown from one
    source".  Units the gate holds back keep the start active but are not improvements: a
    second solve of the converged box must return 0 (a reference-style `while (anychange)`
    driver would otherwise never terminate), and velocities that are negative or not finite
    are refused (zero is accepted, as the reference accepts it: test_zero_velocities_vs_oracle)."""
    rng = np.random.default_rng(5)
    shape = (70, 66, 30)        # z is the only axis that fits a wave: it becomes the lane axis
    v = rng.uniform(0.1, 0.5, size=shape).astype(np.float32)
    offs = np.array([[0, 0, 1], [0, 0, -1], [0, 0, 2], [0, 0, -2], [1, 1, 1]], np.int32)
    start = (20, 33, 7)
    want, _, _ = oracle.converge(v, oracle.make_star(offs), start, order=1)
    assert np.isinf(want).sum() == want.size - shape[2]
    for kernel in (2, 1):
        with P.TravelTimeSolver(shape, P.inputs.make_fs(offs)) as sol:
            sol.set_option(P.OPT_KERNEL, kernel)
            sol.set_velocity(v)
            tt = np.full(shape, np.inf, dtype=np.float32)
            tt[start] = 0
            assert sol.solve([start], [tt]) == 1
            assert_bit_equal(tt, want, f"kernel {kernel}")
            # (copies: a call with the very arrays of the previous one is answered from their
            # digests, test_confirming_call_is_answered_without_device_work - here the kernels'
            # own report is wanted)
            for _ in range(2):
                again = tt.copy()
                assert sol.solve([start], [again]) == 0 and sol.stats()["sweeps_total"] > 0
                assert_bit_equal(again, want, f"kernel {kernel}, re-solved")
            for bad in (-1.0, -1e-40, np.nan, np.inf, -np.inf):
                w = v.copy()
                w[3, 4, 5] = bad
                with pytest.raises(P.TTSweepError):
                    sol.set_velocity(w)
            with pytest.raises(P.TTSweepError):     # a refused volume is not kept
                sol.solve([start], [tt])


def test_solve_multi_shards_starts_over_devices(P, golden24):
    """ttsweep_solve_multi with two device slots (both GPU 0 on a one-GPU box: two
    contexts, two host threads): every start ends bit-equal to its fixture."""
    offs = golden24.star("818")
    fs = P.inputs.make_fs(offs)
    keys = ["818_mid", "818_corner", "818_deadin", "818_deadout"]
    starts = [golden24.z[f"start_{k}"] for k in keys]
    for devices in ([0, 0], [0, 0, 0, 0, 0]):       # more device slots than starts, too
        tts = []
        for st in starts:
            tt = np.full(golden24.v.shape, np.inf, dtype=np.float32)
            tt[tuple(st)] = 0
            tts.append(tt)
        changed = []
        assert P.solve_multi(devices, golden24.v, fs, starts, tts, changed=changed) == 1
        assert changed == [1] * len(starts)
        for k, tt in zip(keys, tts):
            assert_bit_equal(tt, golden24.z[f"tt_{k}"], k)
        assert P.solve_multi(devices, golden24.v, fs, starts, tts) == 0
        # the per-start outcome (serial_new/...:158-164: changed[s]): one box reset, the others at their fixed point
        tts[2][...] = np.inf
        tts[2][tuple(starts[2])] = 0
        assert P.solve_multi(devices, golden24.v, fs, starts, tts, changed=changed) == 1
        assert changed == [0, 0, 1, 0]
        assert_bit_equal(tts[2], golden24.z[f"tt_{keys[2]}"], keys[2])


def test_context_reuse_and_edge_arguments(P, golden24, oracle):
    """One context, several solves with different start counts; zero starts; a second
    velocity volume; a start on every face of the grid; bad arguments raise."""
    offs = golden24.star("5")
    fs = P.inputs.make_fs(offs)
    ofs = oracle.make_star(offs)
    v1 = golden24.v
    v2 = (golden24.v * np.float32(1.7)).astype(np.float32)
    nx, ny, nz = v1.shape
    faces = np.array([[0, 5, 5], [nx - 1, 5, 5], [5, 0, 5], [5, ny - 1, 5], [5, 5, 0], [5, 5, nz - 1],
                      [nx - 1, ny - 1, nz - 1]], dtype=np.int32)

    def boxes(starts):
        out = []
        for st in starts:
            tt = np.full(v1.shape, np.inf, dtype=np.float32)
            tt[tuple(st)] = 0
            out.append(tt)
        return out

    with P.TravelTimeSolver(v1.shape, fs) as sol:
        sol.set_velocity(v1)
        assert sol.solve(np.zeros((0, 3), np.int32), []) == 0
        for v, group in ((v1, faces[:2]), (v1, faces), (v2, faces[3:4]), (v1, faces[4:])):
            sol.set_velocity(v)
            tts = boxes(group)
            assert sol.solve(group, tts) == 1
            for st, tt in zip(group, tts):
                want, _, _ = oracle.converge(v, ofs, st, order=1)
                assert_bit_equal(tt, want, f"start {st}")
        with pytest.raises(P.TTSweepError):
            sol.solve(np.array([[nx, 0, 0]], np.int32), boxes([[0, 0, 0]]))
    with pytest.raises(P.TTSweepError):
        P.TravelTimeSolver((0, 4, 4), fs)
    with pytest.raises(P.TTSweepError):
        P.TravelTimeSolver(v1.shape, fs, device=99)


def test_duplicate_and_zero_offsets_in_the_star(P, oracle):
    """A star file may repeat an offset or contain (0,0,0); both are harmless in the
    reference and must be here (pull star: duplicates merged, zero dropped)."""
    rng = np.random.default_rng(21)
    shape = (18, 40, 22)
    v = rng.uniform(0.1, 0.5, size=shape).astype(np.float32)
    offs = np.array([[1, 0, 0], [0, 0, 0], [1, 0, 0], [0, -2, 1], [-1, 0, 0], [0, 2, -1], [0, 0, 3],
                     [0, 0, -3], [3, 3, 3]], np.int32)
    for start in ((9, 20, 11), (0, 0, 0)):
        want, _, _ = oracle.converge(v, oracle.make_star(offs), start, order=1)
        for kernel in (1, 21, 22):
            (tt,), _, _ = gpu_converge(P, v, P.inputs.make_fs(offs), [start], kernel=kernel)
            assert_bit_equal(tt, want, f"{start} kernel {kernel}")


def _boxes(shape, starts):
    tts = []
    for st in starts:
        tt = np.full(shape, np.inf, dtype=np.float32)
        tt[tuple(st)] = 0
        tts.append(tt)
    return tts


@pytest.mark.parametrize("kernel", KERNELS + [pytest.param(3, id="tile")])
def test_tiny_velocities(P, oracle, kernel):
    """The small end of the velocity range.  The reference halves the ROUNDED product
    d * (v[c] + v[o]) (serial_new/sweep-tt-multistart.c:216), the kernels multiply by d / 2; the
    two agree bit for bit while that product is a normal number - a first version of this test
    with velocities of 1e-42 found 1-ulp differences in the denormal range, on every kernel.
    The fast kernels are therefore used for volumes whose delays cannot be denormal (no 0 < v < 2^-124 /
    d_min), and everything they are given - here 2e-39 .. 1e-30, delays and travel times down to
    the smallest normal numbers - matches the oracle bit for bit (the kernels are built with
    -fno-honor-nans -mno-amdgpu-ieee and use v_pk_mul_f32 / v_pk_add_f32 / v_min3_f32).  [Round 4: volumes
    below the limit are no longer refused - second half of this test.]"""
    rng = np.random.default_rng(77)
    shape = (26, 40, 21)
    offs = P.inputs.read_triples(P.inputs.star_path("six" if kernel == 3 else "5"))
    fs = P.inputs.make_fs(offs)
    tiny = float(2.0 ** -124 / 10.0)                # d_min = delta * 1 = 10 for both stars
    v = (10.0 ** rng.uniform(np.log10(tiny) + 0.01, -30.0, size=shape)).astype(np.float32)
    v[3:9, 5:30, 2:15] = np.float32(tiny * 1.0001)  # a block at the smallest accepted velocity (one start inside it)
    starts = np.array([[13, 20, 20], [0, 0, 0], [5, 10, 7]], dtype=np.int32)
    ofs = oracle.make_star(offs)
    tts, rc, st = gpu_converge(P, v, fs, starts, kernel=kernel)
    assert rc == 1 and st["kernel_variant"] == (2 if kernel > 20 else kernel)
    smallest = np.inf
    for start, tt in zip(starts, tts):
        want, _, _ = oracle.converge(v, ofs, start, order=1)
        assert np.isfinite(want).all()
        smallest = min(smallest, float(want[want > 0].min()))
        assert_bit_equal(tt, want, f"kernel {kernel}, start {start}")
    assert smallest < 1e-37                         # (travel times a few binades above the denormal range)
    # below the limit: accepted as the reference accepts it, solved by the per-cell kernel's instance that rounds
    # a delay as the reference does (the product, then the half) - delays and travel times down to single
    # denormal steps, bit for bit; the next volume without such values gets the chosen kernel back
    with P.TravelTimeSolver(shape, fs) as sol:
        sol.set_option(P.OPT_KERNEL, 2 if kernel > 20 else kernel)
        for k, bad in enumerate((tiny * 0.99, 1e-42, 1.5e-45)):
            w = v.copy()
            w[1, 2, 3] = np.float32(bad)
            if k:       # (the block around the third start: denormal delays, denormal travel times)
                w[3:9, 5:30, 2:15] = (10.0 ** rng.uniform(-44.5, -38.0, size=(6, 25, 13))).astype(np.float32)
            sol.set_velocity(w)
            got = _boxes(shape, starts)
            assert sol.solve(starts, got) == 1
            assert sol.stats()["kernel_variant"] == 1
            for start, tt in zip(starts, got):
                want, _, _ = oracle.converge(w, ofs, start, order=1)
                assert_bit_equal(tt, want, f"kernel {kernel}, start {start}, a velocity of {bad}")
            if k:
                assert 0 < want[want > 0].min() < 1.17e-38
        sol.set_velocity(v)
        got = _boxes(shape, starts[:1])
        assert sol.solve(starts[:1], got) == 1 and sol.stats()["kernel_variant"] == (2 if kernel > 20 else kernel)
        assert_bit_equal(got[0], tts[0], f"kernel {kernel}: back on the chosen kernel")


@pytest.mark.parametrize("kernel", KERNELS + [pytest.param(3, id="tile")])
def test_zero_velocities_vs_oracle(P, oracle, kernel):
    """A region of zero velocity (zero delays between its cells, both signs of zero): accepted
    as the reference accepts it (serial_new/sweep-tt-multistart.c:216 has no test), same fixed
    point bit for bit."""
    shape = (22, 35, 18)
    v = P.inputs.velocity_model(*shape, seed=21).copy()
    v[4:12, 10:25, 3:14] = 0.0
    v[15:20, 2:8, 0:18] = -0.0
    offs = P.inputs.read_triples(P.inputs.star_path("six" if kernel == 3 else "818"))
    starts = np.array([[11, 17, 17], [6, 15, 8], [0, 34, 0]], dtype=np.int32)     # (the second one inside the zero block)
    ofs = oracle.make_star(offs)
    tts, rc, _ = gpu_converge(P, v, P.inputs.make_fs(offs), starts, kernel=kernel)
    assert rc == 1
    for start, tt in zip(starts, tts):
        want, _, _ = oracle.converge(v, ofs, start, order=1)
        assert_bit_equal(tt, want, f"kernel {kernel}, start {start}")


@pytest.mark.parametrize("entries", [98, 146, 26])
def test_prepass_leaves_the_fixed_point_unchanged(P, golden24, entries):
    """TTSWEEP_OPT_PREPASS_ENTRIES (old/wavefront-openmp/wave-multistart.c:210-215: a short
    sub-range of the star is swept before the whole one; 146 = that program's fsindex[3] for the
    818 star, 98 = the entries no longer than 3 cells): the converged boxes are the golden 818
    boxes bit for bit, for all four start kinds, host and device-resident solves."""
    import torch
    offs = golden24.star("818")
    fs = P.inputs.make_fs(offs)
    keys = ["818_mid", "818_corner", "818_deadin", "818_deadout"]
    starts = np.array([golden24.z[f"start_{k}"] for k in keys], dtype=np.int32)
    with P.TravelTimeSolver(golden24.v.shape, fs) as sol:
        sol.set_velocity(golden24.v)
        sol.set_option(P.OPT_PREPASS_ENTRIES, entries)     # (after the velocity: it is handed on)
        tts = _boxes(golden24.v.shape, starts)
        assert sol.solve(starts, tts) == 1
        st = sol.stats()
        assert st["nstart"] == 4 and st["sweeps_total"] > 8
        for k, tt in zip(keys, tts):
            assert_bit_equal(tt, golden24.z[f"tt_{k}"], f"{k} pre-pass {entries}")
        dev = torch.empty((4,) + golden24.v.shape, dtype=torch.float32, device="cuda:0")
        assert sol.solve_device(starts, dev, init=True) == 1
        for n, k in enumerate(keys):
            assert_bit_equal(dev[n].cpu().numpy(), golden24.z[f"tt_{k}"], f"{k} pre-pass {entries}, device")
        sol.set_option(P.OPT_PREPASS_ENTRIES, 0)           # off again
        tts = _boxes(golden24.v.shape, starts)
        assert sol.solve(starts, tts) == 1
        for k, tt in zip(keys, tts):
            assert_bit_equal(tt, golden24.z[f"tt_{k}"], f"{k} after the pre-pass was switched off")
    with P.TravelTimeSolver(golden24.v.shape, fs) as sol:  # option before the velocity
        sol.set_option(P.OPT_PREPASS_ENTRIES, entries)
        sol.set_velocity(golden24.v)
        tts = _boxes(golden24.v.shape, starts[:2])
        assert sol.solve(starts[:2], tts) == 1
        for k, tt in zip(keys[:2], tts):
            assert_bit_equal(tt, golden24.z[f"tt_{k}"], f"{k} pre-pass {entries}, option first")


def test_confirming_call_is_answered_without_device_work(P, golden24):
    """A reference-style driver calls once more with the boxes it was given back to hear 0
    (serial_new/sweep-tt-multistart.c:151-170).  ttsweep_solve answers that call from a digest of
    the boxes - per box, so batched and one-start-at-a-time hosts both profit - and any change
    to a box, its start or the velocity makes it solve again."""
    offs = golden24.star("818")
    fs = P.inputs.make_fs(offs)
    keys = ["818_mid", "818_corner", "818_deadin"]
    starts = np.array([golden24.z[f"start_{k}"] for k in keys], dtype=np.int32)
    with P.TravelTimeSolver(golden24.v.shape, fs) as sol:
        sol.set_velocity(golden24.v)
        tts = _boxes(golden24.v.shape, starts)
        assert sol.solve(starts, tts) == 1 and sol.stats()["sweeps_total"] > 0
        assert sol.solve(starts, tts) == 0 and sol.stats()["sweeps_total"] == 0     # answered from the digests
        assert sol.stats()["nstart"] == 3
        for s in range(3):                                                          # one start at a time, too
            assert sol.solve(starts[s:s + 1], tts[s:s + 1]) == 0 and sol.stats()["sweeps_total"] == 0
        for k, tt in zip(keys, tts):
            assert_bit_equal(tt, golden24.z[f"tt_{k}"], k)
        # a damaged box is solved again (and repaired) ...
        tts[1][7, 8, 9] = np.inf
        assert sol.solve(starts, tts) == 1 and sol.stats()["sweeps_total"] > 0
        assert_bit_equal(tts[1], golden24.z["tt_818_corner"], "repaired")
        assert sol.solve(starts, tts) == 0 and sol.stats()["sweeps_total"] == 0
        # ... a box whose values are too SMALL is not a fixed point of anything this context
        # wrote: it is solved (nothing can raise a value: the reference cannot either)
        tts[0][3, 3, 3] = np.float32(0.5) * tts[0][3, 3, 3]
        rc = sol.solve(starts, tts)
        assert rc in (0, 1) and sol.stats()["sweeps_total"] > 0
        # ... as is the same array with another start, and everything after a new velocity
        tts = _boxes(golden24.v.shape, starts)
        assert sol.solve(starts, tts) == 1
        other = starts.copy()
        other[2] = [1, 2, 3]
        assert sol.solve(other, tts) in (0, 1) and sol.stats()["sweeps_total"] > 0
        tts = _boxes(golden24.v.shape, starts)
        assert sol.solve(starts, tts) == 1
        sol.set_velocity(golden24.v)
        assert sol.solve(starts, tts) == 0 and sol.stats()["sweeps_total"] > 0      # really solved: nothing to improve
