"""The tiled form over many steps: BENCHMARK1 as its own W/E neighbour (mailbox, the loop across the tile edge) for N steps against
the single-tile run, every field bit for bit -- the arrival counters, parities and rim planes over N launches.
python tools/gpu_debug/selfx_long.py [steps] [workload]"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import numpy as np
import bench
from roms_amd import tiling
n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
wl = sys.argv[2] if len(sys.argv) > 2 else "benchmark1"
names = ["zeta", "ubar", "vbar", "u", "v", "t", "W", "Hz", "rho", "Akv", "Huon", "DU_avg1", "Zt_avg1", "wvel"]
cs = bench.params_for(wl, ntimes=n + 2)
cs["ninfo"] = 1
ref = tiling.TiledRun(cs); ref.step(n); ref.sync()
want = {k: ref.gather(k).copy() for k in names}; ref.close()
run = tiling.TiledRun(cs, self_exchange=True, transport="peer"); run.step(n); run.sync()
bad = [k for k in names if not np.array_equal(run.gather(k), want[k])]
print("SELFXLONG", wl, n, "steps, exchanges", run.ctx.L.roms_hip_exchange_count(run.ctx.h), "mismatching", bad, "max|u|", float(np.abs(want["u"]).max()))
run.close()
