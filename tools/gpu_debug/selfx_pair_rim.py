"""The pair launches handing their rim across the tile edge themselves (tiles too large for the persistent loop): BENCHMARK at the
given tile size as its own W/E neighbour through the mailbox, fields against the single-tile run, step time with and without.
python tools/gpu_debug/selfx_pair_rim.py Lm Mm N [steps]"""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import numpy as np
import bench
from roms_amd import tiling
Lm, Mm, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
n = int(sys.argv[4]) if len(sys.argv) > 4 else 30
names = ["zeta", "ubar", "vbar", "u", "v", "t", "W", "Hz", "rho", "Akv", "Huon", "DU_avg1", "DU_avg2", "Zt_avg1", "rzeta", "rubar"]
cs = bench.params_for("benchmark1", Lm, Mm, N, ntimes=n + 20)
cs["ninfo"] = 1
ref = tiling.TiledRun(cs); ref.step(6); ref.sync()
want = {k: ref.gather(k).copy() for k in names}; ref.close()
for rim in ("1", "0"):
    os.environ["ROMS_HIP_PAIR_RIM"] = rim
    run = tiling.TiledRun(cs, self_exchange=True, transport="peer")
    run.step(6); run.sync()
    bad = [k for k in names if not np.array_equal(run.gather(k), want[k])]
    x0 = run.ctx.L.roms_hip_exchange_count(run.ctx.h)
    t0 = time.perf_counter(); run.step(n); run.sync(); t1 = time.perf_counter()
    x1 = run.ctx.L.roms_hip_exchange_count(run.ctx.h)
    print("PAIRRIM %dx%dx%d rim=%s: %.3f ms/step, %d exchanges/step, mismatching %s" % (Lm, Mm, N, rim, 1e3 * (t1 - t0) / n, (x1 - x0) // n, bad), flush=True)
    run.close()
