"""one GLS configuration for a kernel profile: argv = Lm Mm N"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from roms_amd import tiling
from tests import util
Lm, Mm, N = [int(x) for x in sys.argv[1:4]]
cs = util.cases.upwelling_gls(Lm=Lm, Mm=Mm, N=N, hadv=("U3", "U3"), vadv=("C4", "C4"))
cs["ninfo"] = 0
run = tiling.TiledRun(cs, weak=False)
run.step(12)
run.ctx.sync()
run.close()
