"""Ad-hoc timing of the barotropic kernel (not a test): python tools/gpu_debug/gpu_step2d_probe.py [workload]"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import bench
from roms_amd import hiplib, tiling
wl = sys.argv[1] if len(sys.argv) > 1 else "benchmark1"
cs = bench.params_for(wl)
cs["ninfo"] = 0
run = tiling.TiledRun(cs)
run.step(2)
run.sync()
ctx = run.ctx
for pred in (1, 0):
    if pred:
        ctx.set_stepping(iif=2, predictor=1, kstp=2, krhs=1, knew=3)
    else:
        ctx.set_stepping(iif=2, predictor=0, kstp=1, krhs=3, knew=2)
    hiplib.kprof(1)
    for _ in range(200):
        ctx.L.roms_hip_step2d(ctx.h)
    run.sync()
    t = hiplib.kprof_table()
    hiplib.kprof(0)
    print("pred" if pred else "corr", os.environ.get("ROMS_HIP_DBG_STOP", "0"), os.environ.get("ROMS_HIP_TILE2D", "-"),
          {k: round(v[0] / v[1] * 1e6, 2) for k, v in t.items()})

if os.environ.get("ROMS_HIP_DBG_STOP") == "99":
    import numpy as np
    for pred in (1, 0):
        if pred:
            ctx.set_stepping(iif=2, predictor=1, kstp=2, krhs=1, knew=3)
        else:
            ctx.set_stepping(iif=2, predictor=0, kstp=1, krhs=3, knew=2)
        for _ in range(3):
            ctx.L.roms_hip_step2d(ctx.h)
        run.sync()
        x = ctx.download("xr").ravel()
        nb = 256 if wl == "benchmark1" else 512
        T = x[:nb * 8].reshape(nb, 8)[:, :6]
        t0 = T[:, 0].min()
        print("pred" if pred else "corr", "ticks (10 ns) rel. to first block start: start, loads issued, after barrier, after st2, after st3, end")
        for b in (0, 1, 17, nb // 2 + 8, nb - 1):
            print("  block", b, (T[b] - t0).astype(int).tolist())
        print("  mean per stage:", np.round((T - T[:, :1]).mean(axis=0), 1).tolist(), "max end:", int((T[:, 5] - t0).max()), "max start:", int((T[:, 0] - t0).max()))
