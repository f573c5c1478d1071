"""biharmonic case on the device: after how many main3d steps does it leave the oracle, and in which fields?"""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from tests import util
cs = util.case_for("upwelling_bih_small")
g = util.load_init("upwelling_small", util.nghost_for(cs))
O = util.make_oracle(cs, g); O.start()
H = util.make_hip(cs, g); H.start()
for step in range(1, 13):
    O.main3d_step(); H.main3d(1); H.sync()
    bad = []
    for n in util.PROGNOSTIC:
        a, b = H.download(n), O.field(n)
        if not np.array_equal(a, b): bad.append((n, int((a != b).sum()), float("%.1e" % util.relrms(a, b))))
    print("STEP", step, bad[:10], flush=True)
