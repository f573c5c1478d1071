"""Ad-hoc GPU probe (not a test): per-field differences HIP vs oracle and first timings."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from tests import util

tag = sys.argv[1] if len(sys.argv) > 1 else "upwelling"
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
cs = util.case_for(tag, hadv=("U3", "HSIMT"), vadv=("C4", "HSIMT"))
if len(sys.argv) > 3 and sys.argv[3] == "novmix":
    cs["options"] = tuple(o for o in cs["options"] if o != "ANA_VMIX")
g = util.load_init(tag, util.nghost_for(cs))
O = util.make_oracle(cs, g)
H = util.make_hip(cs, g)
O.start(); H.start()
t0 = time.time(); O.main3d_step(nsteps); t_or = time.time() - t0
H.sync(); t0 = time.time(); H.main3d(nsteps); H.sync(); t_hip = time.time() - t0
cells = cs["Lm"] * cs["Mm"] * cs["N"]
print(f"oracle {t_or/nsteps*1e3:.3f} ms/step  hip {t_hip/nsteps*1e3:.3f} ms/step  cells={cells}")
for n in util.PROGNOSTIC:
    a, b = H.download(n), O.field(n)
    print(f"{n:8s} relrms={util.relrms(a,b):.3e} maxabs={np.abs(a-b).max():.3e} nbad={(a!=b).sum()}/{a.size}")
H.profile(True)
H.main3d(20); H.sync()
names = {4:"set_data",6:"set_vbc",9:"step2d",12:"depth/massflux/zeta/wvel",13:"omega",14:"rho_eos",18:"vmix",21:"rhs3d_tile",22:"pre_step3d",23:"prsgrd",24:"t3dmix",30:"uv3dmix",34:"step3d_uv",35:"step3d_t"}
tot = 0
for rid, nm in names.items():
    s, n = H.region(rid)
    tot += s
    print(f"region {rid:2d} {nm:28s} {s/20*1e6:10.1f} us/step  calls/step={n/20:.1f}")
print(f"sum of regions {tot/20*1e6:.1f} us/step (profiled, serialised)")
