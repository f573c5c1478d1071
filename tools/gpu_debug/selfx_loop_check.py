"""The persistent barotropic loop across tile edges, self-exchange form (one GPU: the tile is its own W/E neighbour through
the mailbox): fields against the single-tile run, then the step time with the loop and with the pair launches.
python tools/gpu_debug/selfx_loop_check.py [workload] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import numpy as np
import bench
from roms_amd import tiling
wl = sys.argv[1] if len(sys.argv) > 1 else "benchmark1"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
names = ["zeta", "ubar", "vbar", "u", "v", "t", "W", "Hz", "rho", "Akv", "Huon", "DU_avg1", "DU_avg2", "Zt_avg1", "rzeta", "rubar", "rufrc", "ru"]
cs = bench.params_for(wl, ntimes=n + 20)
cs["ninfo"] = 1
ref = tiling.TiledRun(cs)
ref.step(5); ref.sync()
want = {k: ref.gather(k).copy() for k in names}
ref.close()
for loop in ("1", "0"):
    os.environ["ROMS_HIP_LOOP"] = loop
    run = tiling.TiledRun(cs, self_exchange=True, transport="peer")
    run.step(5); run.sync()
    bad = [k for k in names if not np.array_equal(run.gather(k), want[k])]
    x0 = run.ctx.L.roms_hip_exchange_count(run.ctx.h)
    t0 = time.perf_counter(); run.step(n); run.sync(); t1 = time.perf_counter()
    x1 = run.ctx.L.roms_hip_exchange_count(run.ctx.h)
    print("SELFXLOOP", wl, "loop=%s: %.3f ms/step, %d exchanges/step, mismatching %s" % (loop, 1e3 * (t1 - t0) / n, (x1 - x0) // n, bad), flush=True)
    run.close()
