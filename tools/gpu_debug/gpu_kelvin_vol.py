import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from tests import util
f, meta = util.load_fixture("kelvin_small_steps.npz")
side = util.HipSide(util.case_from_meta(meta))
side.main3d(meta["nsteps"])
print(util.fmt_diag(side.diag()), meta["diag"][-1], side.diag())
