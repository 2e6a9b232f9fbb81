import sys, os, json
sys.path.insert(0, os.getcwd())
import bench, ttsweep_pkg
P = ttsweep_pkg.load()
starts = P.inputs.read_triples(P.inputs.starts_path("24"))
v = P.inputs.velocity_model(241, 241, 51, 20160507)
for k in range(3):
    print(json.dumps(bench.host_program_end_to_end(P, (241, 241, 51), "818", starts, v))[:260])
