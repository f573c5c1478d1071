"""time per step of the KELVIN application (open boundaries) on the GPU: roms_kelvin.in size and a BENCHMARK1-size channel"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from roms_amd import tiling
from tests import util
for kw in (dict(), dict(Lm=512, Mm=64, N=30)):
    cs = util.cases.kelvin(**kw)
    cs["ninfo"] = 0
    run = tiling.TiledRun(cs, weak=False)
    run.step(10)
    run.ctx.sync()
    t0 = time.perf_counter()
    run.step(40)
    run.ctx.sync()
    dt = (time.perf_counter() - t0) / 40
    print("kelvin", cs["Lm"], cs["Mm"], cs["N"], "ndtfast", cs["ndtfast"], "ms/step %.3f" % (dt * 1e3), flush=True)
    run.close()
