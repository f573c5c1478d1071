"""Self-exchange step time of BENCHMARK at given tile dimensions: python tools/gpu_debug/selfx_dims_time.py Lm Mm N [steps]"""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import bench
from roms_amd import tiling
Lm, Mm, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
n = int(sys.argv[4]) if len(sys.argv) > 4 else 30
cs = bench.params_for("benchmark1", Lm, Mm, N, ntimes=n + 10)
cs["ninfo"] = 1
run = tiling.TiledRun(cs, self_exchange=True, transport="peer")
run.step(4); run.sync()
x0 = run.ctx.L.roms_hip_exchange_count(run.ctx.h)
t0 = time.perf_counter(); run.step(n); run.sync(); t1 = time.perf_counter()
x1 = run.ctx.L.roms_hip_exchange_count(run.ctx.h)
print("selfx %dx%dx%d %.3f ms/step, %d exchanges/step" % (Lm, Mm, N, 1e3 * (t1 - t0) / n, (x1 - x0) // n), flush=True)
run.close()
