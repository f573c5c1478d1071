"""Ad-hoc: N steps of a workload with every periodic exchange routed through the transport (self-exchange), for rocprofv3.
python tools/gpu_debug/gpu_selfx_prof.py [workload] [steps] [transport]"""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import bench
from roms_amd import tiling
wl = sys.argv[1] if len(sys.argv) > 1 else "benchmark1"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
tr = sys.argv[3] if len(sys.argv) > 3 else "peer"
cs = bench.params_for(wl, ntimes=n + 10)
cs["ninfo"] = 1
run = tiling.TiledRun(cs, self_exchange=True, transport=tr)
run.step(3); run.sync()
t0 = time.perf_counter(); run.step(n); run.sync(); t1 = time.perf_counter()
print("selfx %s %.3f ms/step" % (tr, 1e3 * (t1 - t0) / n), flush=True)
run.close()
