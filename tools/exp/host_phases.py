"""Phase times of ttsweep_solve inside the plain-C host program (debug build of the library in
gpurun_exp/dbglib, TTSWEEP_TRACE=1): pinning, upload, solve, download, unpinning."""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ttsweep_pkg
P = ttsweep_pkg.load()
shape = (241, 241, 51)
v = P.inputs.velocity_model(*shape, 20160507)
starts = P.inputs.read_triples(P.inputs.starts_path("24"))
d = tempfile.mkdtemp(prefix="ttsweep_host_", dir="/tmp")
vfile = os.path.join(d, "model.vbox")
P.inputs.write_vbox(vfile, v, (1, 1, 1))
sfile = os.path.join(d, "starts.txt")
with open(sfile, "w") as f:
    f.write(f"{len(starts)}\n" + "".join(f"{i} {j} {k}\n" for i, j, k in starts))
exe = os.path.join(ROOT, "uoparallel-seismic-project_amd", "host", "sweep-tt-multistart")
for trace in (False, True):
    env = dict(os.environ, TTSWEEP_NO_OUTPUT="1")
    if trace:
        env.update(TTSWEEP_TRACE="1", LD_LIBRARY_PATH=os.path.join(ROOT, "gpurun_exp", "dbglib") + ":" + env.get("LD_LIBRARY_PATH", ""))
    t0 = time.perf_counter()
    r = subprocess.run([exe, vfile, P.inputs.star_path("818"), sfile], capture_output=True, text=True, env=env, cwd=d)
    print(f"trace={trace}: rc {r.returncode}, wall {time.perf_counter() - t0:.3f} s")
    for ln in (r.stdout + r.stderr).splitlines():
        if ln.startswith("ttsweep_solve:") or ln.startswith("ttsweep: sweep loop") or "ms on device" in ln:
            print("   ", ln)
