"""Ad-hoc: step time with every periodic ghost exchange routed through pack -> RCCL send/recv -> unpack
(self-exchange test aid, one GPU): python tools/gpu_debug/gpu_selfx_time.py [workload] [steps] [rccl|peer]"""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import bench
from roms_amd import tiling
wl = sys.argv[1] if len(sys.argv) > 1 else "benchmark1"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
tr = sys.argv[3] if len(sys.argv) > 3 else "rccl"
cs = bench.params_for(wl, ntimes=n + 10)
cs["ninfo"] = 1
for selfx in (False, True):
    run = tiling.TiledRun(cs, self_exchange=selfx, transport=tr if selfx else None)
    run.step(3); run.sync()
    x0 = run.ctx.L.roms_hip_exchange_count(run.ctx.h)
    t0 = time.perf_counter(); run.step(n); run.sync(); t1 = time.perf_counter()
    x1 = run.ctx.L.roms_hip_exchange_count(run.ctx.h)
    print(tr, "self_exchange=%s: %.3f ms/step, %d exchanges/step" % (selfx, 1e3 * (t1 - t0) / n, (x1 - x0) // n), flush=True)
    run.close()
