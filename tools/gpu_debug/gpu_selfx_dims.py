import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..')))
import bench
from roms_amd import tiling
Lm, Mm, N, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
cs = bench.params_for("benchmark1", Lm, Mm, N, ntimes=n + 10); cs["ninfo"] = 1
for selfx in (False, True):
    run = tiling.TiledRun(cs, self_exchange=selfx, transport="peer" if selfx else None)
    run.step(5); run.sync()
    t0 = time.perf_counter(); run.step(n); run.sync(); t1 = time.perf_counter()
    print(f"{Lm}x{Mm}x{N} self_exchange={selfx}: {1e3*(t1-t0)/n:.3f} ms/step", flush=True)
    run.close()
