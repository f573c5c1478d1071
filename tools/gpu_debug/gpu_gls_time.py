"""time per step of UPWELLING with the generic length-scale closure against the same grid with the analytic mixing
(ANA_VMIX), and the closure's kernels by themselves (roms_hip_profile regions are per entry: region 18 = mixing)"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from roms_amd import tiling
from tests import util
for Lm, Mm, N in ((512, 64, 30), (512, 512, 50)):
    for name, cs in (("ana_vmix", util.cases.upwelling(Lm=Lm, Mm=Mm, N=N, hadv=("U3", "U3"), vadv=("C4", "C4"))),
                     ("gls k-eps KC", util.cases.upwelling_gls(Lm=Lm, Mm=Mm, N=N, hadv=("U3", "U3"), vadv=("C4", "C4"))),
                     ("gls gen CA", util.cases.upwelling_gls(form="upwelling_gls_cb", closure="gen", Lm=Lm, Mm=Mm, N=N,
                                                             hadv=("U3", "U3"), vadv=("C4", "C4")))):
        cs["ninfo"] = 0
        run = tiling.TiledRun(cs, weak=False)
        run.step(10)
        run.ctx.sync()
        t0 = time.perf_counter()
        run.step(30)
        run.ctx.sync()
        dt = (time.perf_counter() - t0) / 30
        print(name, Lm, Mm, N, "ms/step %.3f" % (dt * 1e3), flush=True)
        run.close()
