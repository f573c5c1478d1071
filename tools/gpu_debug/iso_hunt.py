"""OVERFLOW / MIX_ISO_TS on the device: where does the run-to-run varying deviation of `t` come from?

  python tools/gpu_debug/iso_hunt.py step [n]   one main3d step in n fresh contexts: deviation from the oracle, equality between runs
  python tools/gpu_debug/iso_hunt.py kern       kernel by kernel over the first two steps, the oracle's state pushed before EVERY
                                                entry, every entry run twice from the same pushed state (array_equal of the two)
Environment knobs (ROMS_HIP_OVERLAP=0, ROMS_HIP_KPROF=1, ...) are the caller's.  TEST INFRASTRUCTURE (uses the oracle)."""
import os
import sys

import numpy as np

os.environ.setdefault("ROMS_HIP_ALLOW_ISO", "1")
from tests import util  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "step"
tag = os.environ.get("HUNT_TAG", "overflow_small")
cs = util.case_for(tag)
if os.environ.get("HUNT_MIX"):
    cs["options"] = tuple(os.environ["HUNT_MIX"] if o == "MIX_ISO_TS" else o for o in cs["options"])
g = util.load_init(os.environ.get("HUNT_INIT", tag), util.nghost_for(cs))
if os.environ.get("HUNT_KPROF"):
    from roms_amd import hiplib
    hiplib.kprof(int(os.environ["HUNT_KPROF"]))

FIELDS = [n for n in util.STATE_FIELDS]
LIBP = util.EMU_LIB if os.environ.get("HUNT_EMU") else None


def snapshot(H):
    out = {}
    for n in FIELDS:
        try:
            out[n] = H.download(n).copy()
        except KeyError:
            pass
    return out


if mode == "step":
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    nsteps = int(os.environ.get("HUNT_STEPS", "1"))
    O = util.make_oracle(cs, g)
    O.start()
    O.main3d_step(nsteps)
    first = None
    for r in range(n):
        H = util.make_hip(cs, g, LIBP)
        H.start()
        H.main3d(nsteps)
        H.sync()
        S = snapshot(H)
        H.close()
        dev = {k: float("%.2e" % util.relrms(S[k], O.field(k))) for k in ("zeta", "t", "v", "rho", "Hz", "W") if k in S}
        same = None if first is None else [k for k in S if not np.array_equal(S[k], first[k])]
        if first is None:
            first = S
        print("RESULT step run", r, "dev", dev, "differs from run 0 in", same, flush=True)
        if r == 0:
            a = S["t"]; b = O.field("t")
            bad = np.argwhere(a != b).ravel()
            ni, nj, N = O.ni, O.nj, cs["N"]
            print("RESULT   t mismatches", bad.size, "of", a.size)
            for q in bad[:12]:
                it, rem = divmod(int(q), 3 * N * nj * ni); lev, rem = divmod(rem, N * nj * ni); k, rem = divmod(rem, nj * ni); j, i = divmod(rem, ni)
                print("RESULT     itrc %d lev %d k %d j %d i %d  hip %.17e orc %.17e" % (it + 1, lev + 1, k + 1, j, i, a[q], b[q]))

if mode == "kern":
    from tests import refdrive as rd
    O = util.make_oracle(cs, g)
    H = util.make_hip(cs, g, LIBP)
    O.start(); H.start()
    st = dict(iic=1, iif=1, nstp=1, nnew=1, nrhs=1, kstp=1, knew=1, krhs=1, predictor=0, indx1=1, time=0.0, nfast=int(g["bounds"][58]))
    KEYS = ("iic", "iif", "nstp", "nnew", "nrhs", "kstp", "knew", "krhs", "predictor", "time", "indx1")
    sub = dict(rhs3d=["pre_step3d", "prsgrd", "t3dmix2", "rhs3d_tile", "uv3dmix2"])
    for step in (1, 2):
        for kern, s_ in rd.main3d_sequence(cs, st, first=(step == 1)):
            for kk in sub.get(kern, [kern]):
                if kk == "diag":
                    continue
                for k in KEYS:
                    setattr(O.step, k, s_[k])
                O.step.tdays = s_["time"] / 86400.0
                util.push_state(O, H)       # the oracle's state BEFORE the entry
                outs = []
                for rep in range(2):
                    if rep:
                        util.push_state(O, H)
                    if kk == "wvelocity":
                        H.call(kk, s_["nstp"])
                    else:
                        H.call(kk)
                    H.sync()
                    outs.append(snapshot(H))
                if kk == "wvelocity":
                    O.call(kk, None, s_["nstp"])
                else:
                    O.call(kk)
                unstable = [n for n in outs[0] if not np.array_equal(outs[0][n], outs[1][n])]
                dev = {n: float("%.1e" % util.relrms(outs[0][n], O.field(n))) for n in outs[0] if util.relrms(outs[0][n], O.field(n)) > 1e-13}
                print("RESULT kern step", step, kk, "iif", s_.get("iif"), "pred", s_.get("predictor"), "unstable", unstable, "dev", dev, flush=True)
    H.close()
