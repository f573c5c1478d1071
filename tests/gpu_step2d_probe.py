"""Ad-hoc timing of the barotropic kernel (not a test): python tests/gpu_step2d_probe.py [workload]"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
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
