"""Ad-hoc timing of the barotropic PAIR kernel (not a test): python tools/gpu_debug/gpu_pair_probe.py [workload]
ROMS_HIP_DBG_STOP=99: per-stage wall_clock64 stamps"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import numpy as np
import bench
from roms_amd import hiplib, tiling
wl = sys.argv[1] if len(sys.argv) > 1 else "benchmark1"
cs = bench.params_for(wl)
cs["ninfo"] = 0
run = tiling.TiledRun(cs)
run.step(2)
run.sync()
ctx = run.ctx
def pairs(n):
    # the index juggling of main3d.F:810-918 for fast steps 2, 3, ...: indx1 toggles per pair
    indx1 = 1
    for k in range(n):
        ctx.set_stepping(iif=2 + (k % 20), predictor=1, kstp=3 - indx1, krhs=indx1, knew=3, indx1=indx1)
        ctx.L.roms_hip_step2d_pair(ctx.h)
        indx1 = 3 - indx1
hiplib.kprof(1)
pairs(200)
run.sync()
t = hiplib.kprof_table()
hiplib.kprof(0)
print("pair", {k: round(v[0] / v[1] * 1e6, 2) for k, v in t.items()})
if os.environ.get("ROMS_HIP_DBG_STOP") == "99":
    pairs(3)
    run.sync()
    x = ctx.download("xr").ravel()
    d = run.host.dims
    nb = int(os.environ.get("NB", "256"))
    T = x[:nb * 16].reshape(nb, 16)[:, :10]
    t0 = T[:, 0].min()
    print("ticks (10 ns): start, loads issued, barrier, P fluxes, P zeta, P momentum, bc, C fluxes, C zeta, end")
    for b in (0, 1, 17, nb // 2 + 8, nb - 1):
        print("  block", b, (T[b] - t0).astype(int).tolist())
    print("  mean per stage:", np.round((T - T[:, :1]).mean(axis=0), 1).tolist(), "max end:", int((T[:, 9] - t0).max()),
          "max start:", int((T[:, 0] - t0).max()))
