import numpy as np, os, sys
from tests import util, refdrive as rd, cases
from roms_amd import hiplib
hadv = tuple(sys.argv[1].split(",")); vadv = tuple(sys.argv[2].split(","))
app, cs = rd.make_case("overflow_small")
cs["hadv"], cs["vadv"] = hadv, vadv
saved = rd.quiet(); R = rd.reference(app, cs); rd.unquiet(saved)
O = rd.oracle_from(R, cs); O.start()
b = R.bounds(0); nd = cs["ndtfast"]
w = np.stack([R.table(5, 2 * nd), R.table(6, 2 * nd)])
cfg = cases.hip_cfg(cs, R.table(7, 8)[0], b[58], w, R.table(1, cs["N"]), R.table(2, cs["N"]), R.table(3, cs["N"]+1), R.table(4, cs["N"]+1))
H = hiplib.Context(cfg, util.EMU_LIB)
util.push_state(O, H); H.start()
out = []
for s in range(10):
    O.main3d_step(); H.main3d(1)
    a, bb = H.download("t"), O.field("t")
    if not np.array_equal(a, bb): out.append((s, int(np.count_nonzero(a != bb)), float(np.abs(a - bb).max())))
print("RESULT", hadv, vadv, out[:3])
