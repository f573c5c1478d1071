import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import bench
from roms_amd import tiling
cs = bench.params_for("upwelling", 512, 64, 30, ntimes=60)
cs["NSperiodic"] = 1; cs["ninfo"] = 1
for selfx in (False, True):
    run = tiling.TiledRun(cs, self_exchange=selfx, transport=(sys.argv[1] if len(sys.argv) > 1 else "rccl") if selfx else None)
    run.step(3); run.sync()
    x0 = run.ctx.L.roms_hip_exchange_count(run.ctx.h)
    t0 = time.perf_counter(); run.step(20); run.sync(); t1 = time.perf_counter()
    x1 = run.ctx.L.roms_hip_exchange_count(run.ctx.h)
    print("SX8T self_exchange=%s: %.3f ms/step, %d exchanges/step" % (selfx, 1e3 * (t1 - t0) / 20, (x1 - x0) // 20), flush=True)
    run.close()
