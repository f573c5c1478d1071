"""Child-process bodies of tests/test_oracle_vs_ref.py: the C oracle against the reference's own object
code (oracle/_ref).  One reference configuration per process.  TEST INFRASTRUCTURE.

usage: python -m tests.refchild <mode> <case tag> [key=value ...]
  main3d    nsteps=N [hadv=a,b vadv=a,b NtileI=n NtileJ=n tol=x]   whole steps, every state array, every step
  kernels   [hadv= vadv=]      the six core routines one by one on a randomly perturbed mid-run state
  physics                      BENCHMARK physics routines one by one on a perturbed state
  obc       preset=A..F        zetabc, u2dbc, v2dbc, u3dbc, v3dbc, t3dbc with open-boundary kinds on all four edges
"""
import sys

import numpy as np

from tests import refdrive as rd
from tests import util


def parse(argv):
    kw = {}
    for a in argv:
        k, v = a.split("=")
        if k in ("hadv", "vadv", "lbc_tke"):
            kw[k] = tuple(v.split(","))
        elif k == "preset":
            kw[k] = v
        elif k == "tol":
            kw[k] = float(v)
        else:
            kw[k] = int(v)
    return kw


def _diag_of(O):
    """the numbers orc_diag left in the oracle during the step just made (main3d.F:355)"""
    import ctypes as C
    out = (C.c_double * 16)()
    O.L.orc_get_diag(C.c_void_p(O.h), out)
    return out


def mode_main3d(tag, kw):
    """ref_main3d (reference kernels in main3d.F order) against orc_main3d_step."""
    nsteps = kw.pop("nsteps", 10)
    tol = kw.pop("tol", 0.0)
    if "preset" in kw:                       # the kinds of an obc-mode preset on the edges of a whole run
        kw["lbc"] = OBC_PRESETS[kw.pop("preset")]
    kick = kw.pop("kick", 0)                 # per cent of 1 m/s: random velocities added in front of step 3 (a bottom boundary layer to speak of)
    app, cs = rd.make_case(tag, **kw)
    saved = rd.quiet()
    R = rd.reference(app, cs)
    O = rd.oracle_from(R, cs)
    O.start()
    names = rd.shared_fields(R, O)
    log = []
    dlines = []
    worst = 0.0
    for s in range(1, nsteps + 1):
        if kick and s == 3:
            perturb(R, O, np.random.default_rng(9), [("u", 0.01 * kick), ("v", 0.01 * kick)])
        dg = R.main3d(1)
        O.main3d_step(1)
        bad = rd.mismatches(R, O, names)
        if bad:
            worst = max(worst, max(b[2] for b in bad))
            log.append((s, bad[:6]))
        dlines.append(rd.oracle_diag_line(list(_diag_of(O))))
    rs = R.get_stepping()
    ok_step = (rs["iic"] == O.step.iic and rs["indx1"] == O.step.indx1 and rs["time"] == O.step.time)
    rd.unquiet(saved)
    printed = rd.diag_lines()
    ndiag = sum(1 for a, b in zip(printed, dlines) if a == b)
    moved = float(max(np.abs(O.field("u")).max(), np.abs(O.field("v")).max()))      # (OVERFLOW flows along eta only)
    if cs.get("bkpp"):
        print("bottom boundary layer: max(hbbl + h)", float((O.field("hbbl") + O.field("h")).max()))
    print("fields", len(names), "steps", nsteps, "max|u|", moved, "worst relrms", worst, "stepping", ok_step,
          "diag lines equal", ndiag, "of", len(printed))
    if tol == 0.0 and (ndiag != nsteps or len(printed) != nsteps):
        print("MISMATCH diag lines", [(a, b) for a, b in zip(printed, dlines) if a != b][:3])
        ok_step = False
    for entry in log[:5]:
        print("MISMATCH step", entry)
    if ok_step and moved > 0 and worst <= tol:
        print("MAIN3D-OK bitwise" if not log else "MAIN3D-OK within %g" % tol)


def perturb(R, O, rng, amps):
    for n, amp in amps:
        a = O.field(n)
        a[:] += amp * rng.standard_normal(a.size)
        R.put(n, a)


def mode_kernels(tag, kw):
    """step2d (first predictor, a corrector, the last predictor), omega, pre_step3d, rhs3d, step3d_uv,
    step3d_t on a perturbed state: non-smooth inputs reach the limiter / upstream-switch branches a smooth
    run never takes."""
    app, cs = rd.make_case(tag, **kw)
    saved = rd.quiet()
    R = rd.reference(app, cs)
    O = rd.oracle_from(R, cs)
    O.start()
    R.main3d(3)
    O.main3d_step(3)
    names = rd.shared_fields(R, O)
    rng = np.random.default_rng(11)
    res = []
    nfast = R.bounds(0)[58]

    def both(kernel, **st):
        for k, v in st.items():
            setattr(O.step, k, v)
        rd.sync_stepping(R, O)
        R.call(kernel)
        O.call(kernel)
        res.append((kernel, dict(st), rd.mismatches(R, O, names)))

    amps = [("u", 0.02), ("v", 0.02), ("zeta", 0.05), ("ubar", 0.01), ("vbar", 0.01), ("Huon", 20.0),
            ("Hvom", 20.0), ("W", 1.0), ("ru", 1.0), ("rv", 1.0), ("rufrc", 10.0), ("rvfrc", 10.0),
            ("rzeta", 1e-3), ("rubar", 1.0), ("rvbar", 1.0), ("DU_avg1", 5.0), ("DV_avg1", 5.0),
            ("DU_avg2", 5.0), ("DV_avg2", 5.0), ("Zt_avg1", 0.01), ("Akv", 1e-4), ("Akt", 1e-5)]
    perturb(R, O, rng, amps)
    t = O.field("t")
    t[:] += 0.05 * rng.standard_normal(t.size) * (t != 0)
    R.put("t", t)
    base = dict(iic=4, nstp=2, nnew=1, nrhs=2)
    if cs.get("wet_dry"):                 # the initial masks from the perturbed free surface and barotropic flow (initial.F:467)
        both("wetdry", kstp=1, **base)
    both("omega", **base)
    both("pre_step3d", **base)
    perturb(R, O, rng, [("W", 0.5)])
    both("rhs3d", **base)
    both("step2d", iif=1, predictor=1, indx1=1, kstp=1, knew=3, krhs=1, **base)
    both("step2d", iif=1, predictor=0, indx1=2, kstp=1, knew=2, krhs=3, **base)
    both("step2d", iif=2, predictor=1, indx1=2, kstp=1, knew=3, krhs=2, **base)
    both("step2d", iif=2, predictor=0, indx1=1, kstp=2, knew=1, krhs=3, **base)
    both("step2d", iif=nfast + 1, predictor=1, indx1=1, kstp=2, knew=3, krhs=1, **base)
    both("set_depth", **base)
    both("step3d_uv", iif=nfast + 1, **base)
    both("omega", **base)
    both("step3d_t", **base)
    rd.unquiet(saved)
    nbad = 0
    for kernel, st, bad in res:
        if bad:
            nbad += 1
            print("MISMATCH", kernel, st, bad[:6])
    print("calls", len(res), "fields", len(names))
    if nbad == 0:
        print("KERNELS-OK bitwise")


# kinds per variable and edge (west, south, east, north) of the routine-level open-boundary tests: every kind of every
# routine appears on every edge in one preset or another
OBC_PRESETS = {
    "A": dict(zeta=("Cha",) * 4, ubar=("Fla",) * 4, vbar=("Fla",) * 4, u=("Rad",) * 4, v=("Rad",) * 4, temp=("Rad",) * 4, salt=("Rad",) * 4),
    "B": dict(zeta=("Che",) * 4, ubar=("Shc",) * 4, vbar=("Shc",) * 4, u=("Gra",) * 4, v=("Gra",) * 4, temp=("Cla",) * 4, salt=("Gra",) * 4),
    "C": dict(zeta=("Rad",) * 4, ubar=("Rad",) * 4, vbar=("Rad",) * 4, u=("RadNud",) * 4, v=("RadNud",) * 4, temp=("RadNud",) * 4, salt=("RadNud",) * 4),
    "D": dict(zeta=("RadNud",) * 4, ubar=("RadNud",) * 4, vbar=("RadNud",) * 4, u=("Cla",) * 4, v=("Cla",) * 4, temp=("Gra",) * 4, salt=("Cla",) * 4),
    "E": dict(zeta=("Cla",) * 4, ubar=("Cla",) * 4, vbar=("Cla",) * 4, u=("Clo",) * 4, v=("Clo",) * 4, temp=("Clo",) * 4, salt=("Clo",) * 4),
    # a different kind on every edge
    "F": dict(zeta=("Cha", "Rad", "Gra", "Clo"), ubar=("Fla", "Rad", "Shc", "Clo"), vbar=("Shc", "Clo", "Rad", "Fla"),
              u=("Rad", "Clo", "Gra", "Cla"), v=("Clo", "Rad", "Cla", "Gra"), temp=("Rad", "Cla", "Clo", "Gra"), salt=("Gra", "RadNud", "Rad", "Clo")),
    "G": dict(zeta=("Clo", "Cha", "Rad", "Che"), ubar=("Gra", "Fla", "Clo", "Shc"), vbar=("Rad", "Gra", "Fla", "Clo"),
              u=("Gra", "Rad", "Clo", "RadNud"), v=("Rad", "Gra", "RadNud", "Clo"), temp=("Clo", "Rad", "Gra", "Cla"), salt=("Rad", "Clo", "Cla", "Gra")),
}
BRY = [n + "_" + e for n in ("zeta", "ubar", "vbar", "u", "v", "t") for e in ("west", "east", "south", "north")]


def mode_obc(tag, kw):
    """The six boundary-condition routines of the reference (zetabc_tile, u2dbc_tile, v2dbc_tile through ref_bc2d;
    t3dbc_tile, u3dbc_tile, v3dbc_tile through ref_bc3d) against the oracle's on a random state with random boundary
    data, for the stepping variants that select `know` and `dt2d` (zetabc.F:100-112)."""
    import ctypes as C
    preset = kw.pop("preset")
    kw["lbc"] = OBC_PRESETS[preset]
    kw.update(bry_all=1, Znudg=0.5, M2nudg=0.25, M3nudg=2.0, Tnudg=(1.0, 3.0), obcfac=4.0)
    app, cs = rd.make_case(tag, **kw)
    saved = rd.quiet()
    R = rd.reference(app, cs)
    O = rd.oracle_from(R, cs)
    O.start()
    names = [n for n in rd.shared_fields(R, O) if n in ("zeta", "ubar", "vbar", "u", "v", "t")]
    rng = np.random.default_rng(5)
    res = []
    nb = 0
    for n in BRY:
        a = O.field(n)
        a[:] = rng.standard_normal(a.size) * (0.05 if n[0] in "zuv" else 1.0) + (10.0 if n[0] == "t" else 0.0)
        if R.has(n):
            R.put(n, a)
            nb += 1
    for var, amp, base in (("zeta", 0.2, 0.0), ("ubar", 0.1, 0.0), ("vbar", 0.1, 0.0), ("u", 0.1, 0.0), ("v", 0.1, 0.0), ("t", 0.5, 10.0)):
        a = O.field(var)
        a[:] = base + amp * rng.standard_normal(a.size)
        R.put(var, a)
    for n in ("sustr", "svstr", "bustr", "bvstr"):
        a = O.field(n)
        a[:] = 1e-4 * rng.standard_normal(a.size)
        R.put(n, a)
    a = O.field("h")                      # (an application's analytic depth need not cover the rim of a grid it was not made for)
    a[:] = 40.0 + 10.0 * rng.random(a.size)
    if cs.get("wet_dry"):                 # shallow: the free surface matters in the phase speeds (Shchepetkin, WET_DRY form); and
        a[:] = 1.5 + 0.5 * rng.random(a.size)   # every value the wet masks take at velocity points (wetdry_mask_tile: 0, +-1, 2)
        for n in ("umask_wet", "vmask_wet"):
            m = O.field(n)
            m[:] = rng.choice(np.array([0.0, 1.0, -1.0, 2.0]), m.size)
            R.put(n, m)
    R.put("h", a)

    def both(what, lev, **st):
        for k, v in st.items():
            setattr(O.step, k, v)
        rd.sync_stepping(R, O)
        if what == "bc2d":
            R.L.ref_bc2d(C.c_int(lev))
        else:
            R.L.ref_bc3d(C.c_int(lev))
        for tile in range(cs["NtileI"] * cs["NtileJ"]):
            getattr(O.L, "orc_" + what)(C.c_void_p(O.h), C.c_int(tile), C.c_int(lev))
        res.append((what, lev, dict(st), rd.mismatches(R, O, names)))

    base = dict(iic=4, nstp=2, nnew=1, nrhs=2)
    both("bc2d", 3, iif=1, predictor=1, indx1=1, kstp=1, knew=3, krhs=1, **base)
    both("bc2d", 2, iif=1, predictor=0, indx1=2, kstp=1, knew=2, krhs=3, **base)
    both("bc2d", 3, iif=2, predictor=1, indx1=2, kstp=1, knew=3, krhs=2, **base)
    both("bc2d", 1, iif=2, predictor=0, indx1=1, kstp=2, knew=1, krhs=3, **base)
    both("bc3d", 1, **base)
    both("bc3d", 3, **base)
    both("bc3d", 2, iic=5, nstp=1, nnew=2, nrhs=1)
    rd.unquiet(saved)
    nbad = 0
    # (to stderr: the reference's own set-up report is still in the Fortran unit's buffer and reaches stdout at exit)
    for what, lev, st, bad in res:
        if bad:
            nbad += 1
            print("MISMATCH", what, lev, st, bad[:6], file=sys.stderr)
    print("calls", len(res), "fields", len(names), "boundary arrays", nb, file=sys.stderr)
    if nbad == 0 and nb == 24 and len(names) == 6:
        print("OBC-OK bitwise", file=sys.stderr)


def mode_physics(tag, kw):
    """The physics of the headline bench configuration, routine by routine (Next round #2)."""
    app, cs = rd.make_case(tag, **kw)
    saved = rd.quiet()
    R = rd.reference(app, cs)
    O = rd.oracle_from(R, cs)
    O.start()
    names = rd.shared_fields(R, O)
    rng = np.random.default_rng(5)
    st = dict(iic=4, iif=1, nstp=2, nnew=1, nrhs=2, kstp=1, knew=1, krhs=1, predictor=0, indx1=1,
              time=4 * cs["dt"] + 0.3 * 86400.0)     # daytime at the BENCHMARK longitudes: srflx > 0
    for k, v in st.items():
        setattr(O.step, k, v)
    O.step.tdays = st["time"] / 86400.0
    rd.sync_stepping(R, O)
    perturb(R, O, rng, [("u", 0.05), ("v", 0.05), ("zeta", 0.1), ("ubar", 0.02), ("vbar", 0.02)])
    t = O.field("t")
    t[:] += 0.05 * rng.standard_normal(t.size) * (t != 0)
    R.put("t", t)
    O.field("Zt_avg1")[:] = O.field("zeta")[:O.ni * O.nj]
    R.put("Zt_avg1", O.field("Zt_avg1"))
    perturb(R, O, rng, [("DU_avg1", 5.0), ("DV_avg1", 5.0)])
    seq = ["set_depth", "set_massflux", "rho_eos", "set_data", "bulk_flux", "set_vbc", "lmd_vmix", "omega",
           "wvelocity", "set_zeta", "prsgrd", "t3dmix2", "uv3dmix2", "diag"]
    res = []
    for k in seq:
        if k == "bulk_flux" and "BULK_FLUXES" not in cs["options"]:
            continue
        if k == "lmd_vmix" and "LMD_MIXING" not in cs["options"]:
            k = "ana_vmix"
        R.call(k)
        if k == "wvelocity":
            O.call(k, None, st["nstp"])
        elif k == "diag":
            od = O.diag()
        else:
            O.call(k)
        res.append((k, rd.mismatches(R, O, names)))
    # diag.F keeps only avgkp in mod_scalars (the other sums are reset after printing, diag.F:536-552);
    # the rest is compared through the line it prints (7 significant digits)
    rt = R.table(7, 14)
    rd.unquiet(saved)
    printed = rd.diag_lines()
    dg_ok = rt[8] == od[2] and len(printed) >= 1 and printed[-1] == rd.oracle_diag_line(od)
    nbad = 0
    for k, bad in res:
        if bad:
            nbad += 1
            print("MISMATCH", k, bad[:8])
    print("diag", "equal" if dg_ok else ("DIFFERENT", rt[8], printed[-1:], rd.oracle_diag_line(od)))
    print("srflx max", float(O.field("srflx").max()) if "SOLAR_SOURCE" in cs["options"] else None,
          "Akv max", float(O.field("Akv").max()))
    if nbad == 0 and dg_ok:
        print("PHYSICS-OK bitwise")


AVG_FIELDS = ["avg_zeta", "avg_ubar", "avg_vbar", "avg_u", "avg_v", "avg_omega", "avg_w", "avg_rho", "avg_t", "avg_ZZ",
              "avg_U2", "avg_V2", "avg_UU", "avg_VV", "avg_UV", "avg_Huon", "avg_Hvom", "avg_TT", "avg_UT", "avg_VT",
              "avg_HuonT", "avg_HvomT"]


def mode_avg(tag, kw):
    """set_avg.F (the reference built with AVERAGES, oracle/ref/upwelling_avg.h) against orc_set_avg: both sides
    step kernel by kernel in main3d's order with set_avg behind set_zeta (main3d.F:562); all 22 time-averaged
    arrays after every call -- the set, accumulate and convert phases of several windows."""
    nsteps, nAVG, ntsAVG = kw.pop("nsteps", 9), kw.pop("nAVG", 3), kw.pop("ntsAVG", 1)
    app, cs = rd.make_case(tag, **kw)
    saved = rd.quiet()
    R = rd.reference(app, cs)
    O = rd.oracle_from(R, cs)
    O.start()
    R.L.ref_set_avg_window(nAVG, ntsAVG, 0, 1)
    O.set_avg_window(nAVG, ntsAVG, 0, 1)
    nfast = R.bounds(0)[58]
    st = dict(iic=1, iif=1, nstp=1, nnew=1, nrhs=1, kstp=1, knew=1, krhs=1, predictor=0, indx1=1, time=0.0, nfast=nfast)
    names = rd.shared_fields(R, O)
    log, ncmp, nonzero = [], 0, 0
    for step in range(1, nsteps + 1):
        for kern, s_ in rd.main3d_sequence(cs, st, first=(step == 1)):
            for k, v in s_.items():
                if k != "nfast":
                    setattr(O.step, k, v)
            O.step.tdays = s_["time"] / 86400.0
            rd.sync_stepping(R, O)
            R.call(kern)
            if kern == "wvelocity":
                O.call(kern, None, s_["nstp"])
            elif kern == "diag":
                O.diag()
            else:
                O.call(kern)
            if kern == "set_zeta":
                R.call("set_avg")
                O.call("set_avg")
                bad = rd.mismatches(R, O, AVG_FIELDS)
                ncmp += 1
                nonzero += int(np.abs(O.field("avg_UV")).max() > 0.0)
                if bad:
                    log.append((step, bad[:6]))
    bad_state = rd.mismatches(R, O, names)
    rd.unquiet(saved)
    print("steps", nsteps, "nAVG", nAVG, "ntsAVG", ntsAVG, "set_avg calls compared", ncmp, "with data", nonzero)
    for e in log[:5]:
        print("MISMATCH step", e)
    if bad_state:
        print("MISMATCH state", bad_state[:6])
    if not log and not bad_state and nonzero >= nsteps - ntsAVG - 1:
        print("AVG-OK bitwise")


DIA_FIELDS = ["DiaTwrk", "DiaTrc", "dia_zeta"]
DIAUV_FIELDS = ["DiaRU", "DiaRV", "DiaRUfrc", "DiaRVfrc", "DiaU3wrk", "DiaV3wrk", "DiaU2wrk", "DiaV2wrk", "DiaU2int", "DiaV2int",
                "DiaRUbar", "DiaRVbar", "DiaU2d", "DiaV2d", "DiaU3d", "DiaV3d"]


def mode_dia(tag, kw):
    """DIAGNOSTICS_TS (the reference built from upwelling.h as shipped) against the oracle: both sides step kernel by kernel
    in main3d's order with set_diags behind set_zeta (main3d.F:559); DiaTwrk after every kernel, DiaTrc / avgzeta after
    every set_diags -- the set, accumulate and convert phases of several windows."""
    nsteps, nDIA, ntsDIA, uv = kw.pop("nsteps", 9), kw.pop("nDIA", 3), kw.pop("ntsDIA", 1), kw.pop("uv", 0)
    app, cs = rd.make_case(tag, **kw)
    saved = rd.quiet()
    R = rd.reference(app, cs)
    O = rd.oracle_from(R, cs)
    O.start()
    R.L.ref_set_dia_window(nDIA, ntsDIA, 0, 1)
    O.set_dia_window(nDIA, ntsDIA, 0, 1, uv=bool(uv))
    fields = DIA_FIELDS + (DIAUV_FIELDS if uv else [])
    nfast = R.bounds(0)[58]
    st = dict(iic=1, iif=1, nstp=1, nnew=1, nrhs=1, kstp=1, knew=1, krhs=1, predictor=0, indx1=1, time=0.0, nfast=nfast)
    names = rd.shared_fields(R, O)
    log, ncmp, nonzero = [], 0, 0
    for step in range(1, nsteps + 1):
        for kern, s_ in rd.main3d_sequence(cs, st, first=(step == 1)):
            for k, v in s_.items():
                if k != "nfast":
                    setattr(O.step, k, v)
            O.step.tdays = s_["time"] / 86400.0
            rd.sync_stepping(R, O)
            R.call(kern)
            if kern == "wvelocity":
                O.call(kern, None, s_["nstp"])
            elif kern == "diag":
                O.diag()
            else:
                O.call(kern)
            if kern == "set_zeta":
                R.call("set_diags")
                O.call("set_diags")
                ncmp += 1
                nonzero += int(np.abs(O.field("DiaTrc")).max() > 0.0)
            if kern in ("set_zeta", "rhs3d", "step3d_t") or (uv and kern in ("step2d", "step3d_uv")):
                bad = rd.mismatches(R, O, fields)
                if bad:
                    log.append((step, kern, bad[:6]))
    bad_state = rd.mismatches(R, O, names)
    rd.unquiet(saved)
    print("steps", nsteps, "nDIA", nDIA, "ntsDIA", ntsDIA, "set_diags calls compared", ncmp, "with data", nonzero)
    for e in log[:5]:
        print("MISMATCH", e)
    if bad_state:
        print("MISMATCH state", bad_state[:6])
    if not log and not bad_state and nonzero >= nsteps - ntsDIA - 1:
        print("DIA-OK bitwise")


if __name__ == "__main__":
    mode, tag = sys.argv[1], sys.argv[2]
    {"main3d": mode_main3d, "kernels": mode_kernels, "physics": mode_physics, "avg": mode_avg, "dia": mode_dia, "obc": mode_obc}[mode](tag, parse(sys.argv[3:]))
