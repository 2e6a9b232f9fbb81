"""GPU tier: the plain-C host program (uoparallel-seismic-project_amd/host/
sweep-tt-multistart.c), i.e. the reference's main() with sweepXYZ forwarding to the
C ABI, run end to end on real files: .vbox in, star + start text files in,
output.tt out (format of serial_new/sweep-tt-multistart.c:176-194)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

HOST_DIR = os.path.join(ROOT, "uoparallel-seismic-project_amd", "host")
EXE = os.path.join(HOST_DIR, "sweep-tt-multistart")


@pytest.fixture(scope="module")
def exe(pkg):
    r = subprocess.run(["make", "-C", HOST_DIR], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    return EXE


@pytest.mark.gpu
def test_host_program_end_to_end(exe, pkg, oracle, tmp_path):
    shape = (30, 26, 14)
    v = pkg.inputs.velocity_model(*shape, seed=8)
    pkg.inputs.write_vbox(str(tmp_path / "model.vbox"), v)
    starts = np.array([[15, 13, 13], [0, 0, 0], [29, 3, 7]], dtype=np.int32)
    (tmp_path / "starts.txt").write_text(
        f"{len(starts)}\n" + "".join(f"{i} {j} {k}\n" for i, j, k in starts))
    star = pkg.inputs.star_path("818")
    env = dict(os.environ, TTSWEEP_BINARY_OUTPUT="tt-")
    r = subprocess.run([exe, "model.vbox", star, "starts.txt"], cwd=tmp_path, env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    out = r.stdout
    # the reference's progress lines (serial_new/...:80-166)
    assert "Velocity model dimensions: 30 x 26 x 14" in out
    assert "Forward star size: 818" in out
    assert "Delta: 10.000000" in out
    assert "starting point 2: 29 3 7" in out
    assert "sweep 1 begin" in out and "sweep 2 finished: anychange = 0" in out
    assert "sweep 3 begin" not in out
    # per start, as the reference's loop prints it (:158-164; from ttsweep_get_changed): every box moved in the
    # first pass, none in the confirming one
    changed = [l for l in out.splitlines() if l.startswith(">>> start")]
    assert changed == [f">>> start {s}: changed == 1" for s in range(3)] + [f">>> start {s}: changed == 0" for s in range(3)]

    lines = (tmp_path / "output.tt").read_text().splitlines()
    assert lines[0] == "30 26 14"
    n = v.size
    assert len(lines) == 1 + len(starts) * (1 + n)
    fs = oracle.make_star(oracle.read_triples(star))
    for s, st in enumerate(starts):
        base = 1 + s * (1 + n)
        assert lines[base] == f"starting point: {s}"
        want, _, _ = oracle.converge(v, fs, st, order=1)
        # spot-check the text of a few cells and parse all values
        first = lines[base + 1]
        assert first == "travel time for (0,0,0): %f 0 0 0" % want[0, 0, 0]
        got = np.array([float(l.split(": ")[1].split()[0]) for l in lines[base + 1: base + 1 + n]])
        exp = np.array([float("%f" % x) for x in want.reshape(-1)])
        assert np.array_equal(got, exp), s
        # the compact binary volume holds the exact floats
        origin, box = pkg.inputs.read_vbox(str(tmp_path / f"tt-{s}.vbox"))
        assert origin == (1, 1, 1) and np.array_equal(box.view(np.uint32), want.view(np.uint32))


def _write_inputs(pkg, tmp_path, shape, nstart, seed=3):
    v = pkg.inputs.velocity_model(*shape, seed=seed)
    pkg.inputs.write_vbox(str(tmp_path / "model.vbox"), v)
    rng = np.random.default_rng(seed)
    starts = np.stack([rng.integers(0, n, size=nstart) for n in shape], axis=1).astype(np.int32)
    (tmp_path / "starts.txt").write_text(
        f"{nstart}\n" + "".join(f"{i} {j} {k}\n" for i, j, k in starts))
    return v, starts


def test_host_program_refuses_more_starts_than_STARTMAX(exe, pkg, tmp_path):
    """STARTMAX is 128 here (12 in the reference, serial_new/sweep-tt-multistart.c:44, which
    start-24 and start-111 overflow): 129 starts are refused before anything is allocated or any
    GPU call is made (runs on the CPU tier)."""
    _write_inputs(pkg, tmp_path, (12, 10, 8), 129)
    r = subprocess.run([exe, "model.vbox", pkg.inputs.star_path("3"), "starts.txt"], cwd=tmp_path,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "Bad number of starting points" in r.stdout and "maximum 128" in r.stdout


@pytest.mark.gpu
def test_host_program_with_STARTMAX_starts(exe, pkg, oracle, tmp_path):
    """Exactly STARTMAX = 128 start points go through the host program (static arrays full)."""
    shape = (12, 10, 8)
    v, starts = _write_inputs(pkg, tmp_path, shape, 128)
    star = pkg.inputs.star_path("3")
    env = dict(os.environ, TTSWEEP_NO_OUTPUT="1", TTSWEEP_BINARY_OUTPUT="tt-")
    r = subprocess.run([exe, "model.vbox", star, "starts.txt"], cwd=tmp_path, env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "starting point 127:" in r.stdout and "sweep 2 finished: anychange = 0" in r.stdout
    fs = oracle.make_star(oracle.read_triples(star))
    for s in (0, 63, 127):
        want, _, _ = oracle.converge(v, fs, starts[s], order=1)
        _, box = pkg.inputs.read_vbox(str(tmp_path / f"tt-{s}.vbox"))
        assert np.array_equal(box.view(np.uint32), want.view(np.uint32)), s


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["1", "24"])
def test_host_program_full_size_digests(exe, pkg, tmp_path, which):
    """BASELINE configs 1 and 3 through the drop-in main() itself (serial_new/sweep-tt-multistart.c:70-195 as the
    caller, :151-170 as the loop being replaced): the synthetic 241x241x51 model as a VBOX file, docs/818-FS.txt,
    docs/start-1 / start-24 verbatim; every travel-time volume the program writes is, bit for bit, the box the
    UNMODIFIED reference converged to (SHA-256 recorded from oracle/_ref by tests/golden/make_golden.py)."""
    import hashlib
    import json
    digests = json.load(open(os.path.join(ROOT, "tests", "golden", "big_digests.json")))
    v = pkg.inputs.velocity_model(241, 241, 51, 20160507)
    pkg.inputs.write_vbox(str(tmp_path / "model.vbox"), v)
    starts = pkg.inputs.read_triples(pkg.inputs.starts_path(which))
    env = dict(os.environ, TTSWEEP_NO_OUTPUT="1", TTSWEEP_BINARY_OUTPUT="tt-")
    r = subprocess.run([exe, "model.vbox", pkg.inputs.star_path("818"), pkg.inputs.starts_path(which)], cwd=tmp_path,
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    out = r.stdout
    assert "Velocity model dimensions: 241 x 241 x 51" in out and "Forward star size: 818" in out
    assert f"starting point {len(starts) - 1}:" in out
    assert "sweep 2 finished: anychange = 0" in out and "sweep 3 begin" not in out
    changed = [l for l in out.splitlines() if l.startswith(">>> start")]
    n = len(starts)
    assert changed == [f">>> start {s}: changed == 1" for s in range(n)] + [f">>> start {s}: changed == 0" for s in range(n)]
    for s, st in enumerate(starts):
        want = digests["syn241_818_%d_%d_%d" % tuple(int(x) for x in st)]["sha256"]
        origin, box = pkg.inputs.read_vbox(str(tmp_path / f"tt-{s}.vbox"))
        assert origin == (1, 1, 1) and box.shape == v.shape and box.dtype == np.float32
        assert hashlib.sha256(np.ascontiguousarray(box).tobytes()).hexdigest() == want, (s, tuple(st))


def test_host_program_rejects_bad_input(exe, tmp_path):
    r = subprocess.run([exe], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 1 and "usage" in r.stdout
    (tmp_path / "junk.vbox").write_bytes(b"not a vbox file at all")
    r = subprocess.run([exe, "junk.vbox", "x", "y"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 1 and "Cannot open velocity model file" in r.stdout


@pytest.mark.gpu
def test_bench_prints_one_contract_line(pkg, exe):
    """bench.py (the driver's entry point): exactly one JSON line on stdout with the keys of
    the contract, the roofline and (with --no-cpu) no CPU leg; value = cells relaxed / time."""
    import json
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0",
                        "--nstarts", "2", "--no-cpu", "--no-traffic"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 1 and d["warmup"] == 0 and d["vs_baseline"] is None
    assert d["unit"] == "Mcells*sweeps/s" and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"] and d["config"]["starts"] == 2
    assert d["metric"].startswith("Mcells*sweeps/s (241x241x51, 818-offset star, 2 starts")
    for rf, bound, peak in ((d["roofline"], "valu", 78.6), (d["roofline_hbm"], "hbm", 8000.0)):
        for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "avg_launch_ms"):
            assert key in rf, key
        assert rf["bound"] == bound and rf["peak"] == peak and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert d["roofline"]["traffic"] is None         # --no-traffic: nothing is quoted from a file
    assert d["value"] > 0 and d["ms_per_step"] > 0 and "cpu_baseline" not in d
    assert abs(d["time_to_solution"]["gpu_all_starts_s"] - d["ms_per_step"] / 1e3) < 1e-12
    e2e = d["end_to_end_host_program"]              # the plain-C host program on the same two starts
    assert e2e["process_wall_seconds"] > 0 and e2e["sweep_loop_wall_seconds"] > 0
    cells = 241 * 241 * 51
    eq = d["config"]["full_sweep_equivalents_per_start_mean"]
    assert abs(d["value"] - eq * 2 * cells / (d["ms_per_step"] / 1e3) / 1e6) < 1e-6 * d["value"]
