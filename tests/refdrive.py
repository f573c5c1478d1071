"""Drive the reference's own object code (oracle/_ref, see oracle/ref/build_ref.sh) next to the C oracle.
TEST INFRASTRUCTURE; runs only where /root/reference and oracle/_ref exist (the build container).

One reference configuration can live in a process (Fortran module state), so every function here is
meant to be called from a fresh `python -c` / `python -m` child (tests/test_oracle_vs_ref.py,
tests/golden/make_golden.py)."""
import os
import sys

import numpy as np

from tests import cases, util

# every state array both sides expose (absent ones -- cpp options -- are skipped at run time)
FIELDS = util.STATE_FIELDS
CASES = {
    # tag: (reference library, case constructor kwargs)
    "upwelling_small": ("upwelling", dict(Lm=14, Mm=18, N=8)),
    "upwelling": ("upwelling", dict()),
    "benchmark_small": ("benchmark", dict(Lm=24, Mm=16, N=10)),
    "benchmark1": ("benchmark", dict()),
    "upwelling_kpp_small": ("upwelling_kpp", dict(Lm=14, Mm=18, N=8)),
    # LMD_DDMIX (round 6): oracle/ref/upwelling_kpp_ddmix.h (linear EOS), benchmark.h -DLMD_DDMIX (nonlinear EOS); the state of cases.ddmix_state
    "upwelling_kpp_ddmix_small": ("upwelling_kpp_ddmix", dict(Lm=14, Mm=18, N=8)),
    "benchmark_ddmix_small": ("benchmark_ddmix", dict(Lm=24, Mm=16, N=10)),
    # LMD_BKPP (round 6): benchmark.h -DLMD_BKPP, oracle/ref/upwelling_kpp.h -DLMD_BKPP
    "benchmark_bkpp_small": ("benchmark_bkpp", dict(Lm=24, Mm=16, N=10)),
    "upwelling_kpp_bkpp_small": ("upwelling_kpp_bkpp", dict(Lm=14, Mm=18, N=8)),
    "benchmark_wetdry_ddmix_small": ("benchmark_wetdry_ddmix", dict(Lm=24, Mm=16, N=10)),      # ... under WET_DRY
    # the UPWELLING case built WITH its time-averaged output (oracle/ref/upwelling_avg.h): pins set_avg.F
    "upwelling_avg_small": ("upwelling_avg", dict(Lm=14, Mm=18, N=8)),
    # ROMS/Include/upwelling.h AS SHIPPED (AVERAGES, DIAGNOSTICS_TS, DIAGNOSTICS_UV): pins the per-term tracer tendencies
    "upwelling_diag_small": ("upwelling_diag", dict(Lm=14, Mm=18, N=8)),
    # UPWELLING with the logarithmic bottom drag (oracle/ref/upwelling_logdrag.h)
    "upwelling_logdrag_small": ("upwelling_logdrag", dict(Lm=14, Mm=18, N=8)),
    # UPWELLING with WINDBASIN's option set: no UV_ADV, no horizontal mixing (oracle/ref/upwelling_noadv.h)
    "upwelling_noadv_small": ("upwelling_noadv", dict(Lm=14, Mm=18, N=8)),
    # UPWELLING with land/sea masking (oracle/ref/upwelling_mask.h; the masks are cases.land_mask)
    "upwelling_mask_small": ("upwelling_mask", dict(Lm=14, Mm=18, N=8)),
    "benchmark_mask_small": ("benchmark_mask", dict(Lm=24, Mm=16, N=10)),
    # ... with MASKING + WET_DRY (oracle/ref/benchmark_wetdry.h): bulk fluxes, solar source, KPP, geopotential mixing on the beach
    "benchmark_wetdry_small": ("benchmark_wetdry", dict(Lm=24, Mm=16, N=10)),
    # UPWELLING with MASKING + WET_DRY (oracle/ref/upwelling_wetdry.h; bathymetry and initial ridge: cases.wetdry_depth)
    "upwelling_wetdry_small": ("upwelling_wetdry", dict(Lm=14, Mm=18, N=8)),
    "upwelling_wetdry_obc_small": ("upwelling_wetdry", dict(Lm=14, Mm=18, N=8)),      # closed basin: all four walls
    "upwelling_avg_mask_small": ("upwelling_avg_mask", dict(Lm=14, Mm=18, N=8)),      # AVERAGES + MASKING
    "upwelling_wetdry_avg_small": ("upwelling_wetdry_avg", dict(Lm=14, Mm=18, N=8)),  # AVERAGES + WET_DRY (round 6)
    # WET_DRY with the closures, the viscosity along geopotentials, the other pressure Jacobians (round 6: upwelling_wetdry_*.h)
    "upwelling_wetdry_gls_small": ("upwelling_wetdry_gls", dict(Lm=14, Mm=18, N=8, variant="gls")),
    "upwelling_wetdry_my25_small": ("upwelling_wetdry_my25", dict(Lm=14, Mm=18, N=8, variant="my25")),
    "upwelling_wetdry_geouv_small": ("upwelling_wetdry_geouv", dict(Lm=14, Mm=18, N=8, variant="geouv")),
    "upwelling_wetdry_prs31_small": ("upwelling_wetdry_prs31", dict(Lm=14, Mm=18, N=8, variant="prs31")),
    "upwelling_wetdry_prs44_small": ("upwelling_wetdry_prs44", dict(Lm=14, Mm=18, N=8, variant="prs44")),
    "upwelling_wetdry_iso_small": ("upwelling_wetdry_iso", dict(Lm=14, Mm=18, N=8, variant="iso")),
    # open boundaries: the reference's own KELVIN application (ROMS/Include/kelvin.h, RADIATION_2D) ...
    # more of the reference's own test applications (ROMS/Include/seamount.h, grav_adj.h as shipped)
    # the standard density Jacobian (prsgrd31.h), plain and weighted (WJ_GRADP)
    "upwelling_prs31_small": ("upwelling_prs31", dict(Lm=14, Mm=18, N=8)),
    "upwelling_wjgradp_small": ("upwelling_wjgradp", dict(Lm=14, Mm=18, N=8, wj=True)),
    "upwelling_prs40_small": ("upwelling_prs40", dict(Lm=14, Mm=18, N=8)),          # PJ_GRADP, prsgrd40.h
    "upwelling_prs42_small": ("upwelling_prs42", dict(Lm=14, Mm=18, N=8, scheme=42)),   # PJ_GRADPQ2, prsgrd42.h
    "upwelling_prs44_small": ("upwelling_prs44", dict(Lm=14, Mm=18, N=8, scheme=44)),   # PJ_GRADPQ4, prsgrd44.h
    # biharmonic mixing along s-surfaces (oracle/ref/upwelling_bih.h: UV_VIS4, TS_DIF4)
    # harmonic viscosity along geopotential surfaces under MASKING (oracle/ref/upwelling_geouv.h: MIX_GEO_UV, uv3dmix2_geo.h)
    "upwelling_geouv_small": ("upwelling_geouv", dict(Lm=14, Mm=18, N=8)),
    "upwelling_bihgeouv_small": ("upwelling_bihgeouv", dict(Lm=14, Mm=18, N=8)),     # UV_VIS4 + MIX_GEO_UV under MASKING (uv3dmix4_geo.h; round 6)
    "upwelling_bihgeouv_closed_small": ("upwelling_bihgeouv", dict(Lm=14, Mm=18, N=8)),
    "upwelling_bih_small": ("upwelling_bih", dict(Lm=14, Mm=18, N=8)),
    "upwelling_bihgeo_small": ("upwelling_bihgeo", dict(Lm=14, Mm=18, N=8)),         # ... tracers along geopotentials (t3dmix4_geo.h)
    "upwelling_bihiso_small": ("upwelling_bihiso", dict(Lm=14, Mm=18, N=8)),         # ... along isopycnals (t3dmix4_iso.h)
    # the generic length-scale closure: upwelling.h with -DGLS_MIXING, and its other compile-time forms
    "upwelling_gls_small": ("upwelling_gls", dict(Lm=14, Mm=18, N=8)),
    "upwelling_gls_kw_small": ("upwelling_gls", dict(Lm=14, Mm=18, N=8, closure="k-omega")),
    "upwelling_gls_ca_small": ("upwelling_gls_ca", dict(Lm=14, Mm=18, N=8, form="upwelling_gls_ca", closure="gen")),
    "upwelling_gls_cb_small": ("upwelling_gls_cb", dict(Lm=14, Mm=18, N=8, form="upwelling_gls_cb", closure="k-kl")),
    "upwelling_gls_gal_small": ("upwelling_gls_gal", dict(Lm=14, Mm=18, N=8, form="upwelling_gls_gal", closure="k-omega")),
    # the Mellor-Yamada 2.5 closure: upwelling.h with -DMY25_MIXING, and with Galperin's functions / K_C4ADVECTION
    "upwelling_my25_small": ("upwelling_my25", dict(Lm=14, Mm=18, N=8)),
    "upwelling_my25_gal_small": ("upwelling_my25_gal", dict(Lm=14, Mm=18, N=8, form="upwelling_my25_gal")),
    "seamount": ("seamount", dict()),
    "seamount_small": ("seamount", dict(Lm=20, Mm=18, N=8)),
    "overflow": ("overflow", dict()),                              # roms_overflow.in: MIX_ISO_TS, Vtransform 1 / Vstretching 1
    "overflow_small": ("overflow", dict(Lm=4, Mm=40, N=10)),
    "grav_adj": ("grav_adj", dict()),
    "grav_adj_small": ("grav_adj", dict(Lm=32, Mm=4, N=10)),
    "kelvin": ("kelvin_splines", dict()),
    "kelvin_small": ("kelvin_splines", dict(Lm=16, Mm=12, N=6)),
    "kelvin_geouv_small": ("kelvin_geouv", dict(Lm=16, Mm=12, N=6)),   # open boundaries + MIX_GEO_UV (oracle/ref/kelvin_geouv.h; round 6)
    "benchmark_iso_small": ("benchmark_iso", dict(Lm=24, Mm=16, N=10)),   # MIX_ISO_TS with the nonlinear EOS (oracle/ref/benchmark_iso.h; round 6)
    "kelvin_gls_small": ("kelvin_gls", dict(Lm=16, Mm=12, N=6)),       # open boundaries + GLS_MIXING (oracle/ref/kelvin_gls.h)
    "kelvin_plain_small": ("kelvin", dict(Lm=16, Mm=12, N=6, plain=True)),   # kelvin.h as shipped: the plain vertical solvers
    "kelvin_plain": ("kelvin", dict(plain=True)),
    # ... and closed-basin variants of the other libraries for the routine-level tests (no RADIATION_2D; MASKING)
    "upwelling_obc_small": ("upwelling", dict(Lm=14, Mm=18, N=8)),
    # closed basins (round 6): whole steps between four walls -- the corner values of every boundary routine -- and the biharmonic
    # operators' conditions on the first operator at western / eastern walls (t3dmix4_geo.h:475-600, t3dmix4_iso.h:504-618, t3dmix4_s.h)
    "upwelling_closed_small": ("upwelling", dict(Lm=14, Mm=18, N=8)),
    "upwelling_mask_closed_small": ("upwelling_mask", dict(Lm=14, Mm=18, N=8)),      # (with open kinds on the edges: the masked open-boundary whole runs, VolCons)
    "upwelling_bih_closed_small": ("upwelling_bih", dict(Lm=14, Mm=18, N=8)),
    "upwelling_bihgeo_closed_small": ("upwelling_bihgeo", dict(Lm=14, Mm=18, N=8)),
    "upwelling_bihiso_closed_small": ("upwelling_bihiso", dict(Lm=14, Mm=18, N=8)),
    "upwelling_mask_obc_small": ("upwelling_mask", dict(Lm=14, Mm=18, N=8)),
}


_LOG = None
_TEXT = []


def _cleanup():
    if _LOG and os.path.exists(_LOG):
        os.unlink(_LOG)


import atexit
atexit.register(_cleanup)


def quiet():
    """send the reference's Fortran stdout (set-up report, diag lines) to a scratch file (fd 1)"""
    global _LOG
    import tempfile
    sys.stdout.flush()
    f = tempfile.NamedTemporaryFile(prefix="romsref_stdout_", suffix=".log", delete=False)
    _LOG = f.name
    saved = os.dup(1)
    os.dup2(f.fileno(), 1)
    f.close()
    return saved


def unquiet(saved):
    os.dup2(saved, 1)


def diag_lines():
    """The `diag` report the reference printed since quiet() (diag.F:473-500, FORMAT 30/40): a list of
    ((avgke, avgpe, avgkp, volume) as printed strings, (Ci, Cj, Ck), (Cu, Cv, Cw, maxspeed) strings)."""
    import re
    out = []
    if _LOG is None:
        return out
    flt = r"[-+]?\d\.\d{6}(?:E[-+]\d{2}|[-+]\d{3})"       # 1pe14.6: a three-digit exponent drops the E (3.489962-286)
    with open(_LOG, "rb") as f:
        lines = [b.decode("latin-1") for b in f.read().split(b"\n")]   # DateTime is unset: arbitrary bytes
    os.unlink(_LOG)
    k = 0
    while k < len(lines) - 1:
        a = re.findall(flt, lines[k])
        m = re.search(r"\((\d+),(\d+),(\d+)\)", lines[k + 1])
        if len(a) >= 4 and m and re.match(r"^\s*\d+\s", lines[k]):
            b = re.findall(flt, lines[k + 1])
            out.append((tuple(a[-4:]), tuple(int(x) for x in m.groups()), tuple(b[-4:])))
            _TEXT.append((lines[k], lines[k + 1]))
            k += 2
        else:
            k += 1
    return out


def diag_text():
    """the two printed lines of every diag report diag_lines() found, verbatim"""
    return list(_TEXT)


def fmt_e(x):
    """Fortran 1pe14.6 / 1pe13.6 digits of x"""
    s = "%.6E" % x
    m, e = s.split("E")
    return s if len(e) == 3 else m + e          # as Fortran's 1pe14.6 prints a three-digit exponent: 3.489962-286


def oracle_diag_line(od):
    """orc_diag numbers in the shape of diag_lines() entries"""
    return (tuple(fmt_e(v) for v in od[:4]), (int(od[8]), int(od[9]), int(od[10])),
            tuple(fmt_e(v) for v in (od[5], od[6], od[7], od[4])))


def make_case(tag, **kw):
    app, base = CASES[tag]
    k = dict(base)
    k.update({a: b for a, b in kw.items() if a not in ("bry_all", "Znudg", "M2nudg", "M3nudg", "Tnudg", "obcfac", "lbc_tke", "clima", "volcons")})
    ctor = dict(upwelling=cases.upwelling, benchmark=cases.benchmark, upwelling_kpp=cases.upwelling_kpp,
                upwelling_avg=cases.upwelling, upwelling_diag=cases.upwelling, upwelling_logdrag=cases.upwelling_logdrag, upwelling_noadv=cases.upwelling_noadv,
                upwelling_mask=cases.upwelling_mask, upwelling_wetdry=cases.upwelling_wetdry, benchmark_mask=cases.benchmark_mask, benchmark_wetdry=cases.benchmark_wetdry,
                upwelling_avg_mask=cases.upwelling_mask, upwelling_wetdry_avg=cases.upwelling_wetdry, upwelling_wetdry_gls=cases.upwelling_wetdry_x, upwelling_wetdry_my25=cases.upwelling_wetdry_x,
                upwelling_wetdry_geouv=cases.upwelling_wetdry_x, upwelling_wetdry_prs31=cases.upwelling_wetdry_x, upwelling_wetdry_prs44=cases.upwelling_wetdry_x, upwelling_wetdry_iso=cases.upwelling_wetdry_x, kelvin=cases.kelvin, kelvin_splines=cases.kelvin, kelvin_gls=cases.kelvin_gls, kelvin_geouv=cases.kelvin_geouv, benchmark_iso=cases.benchmark_iso, seamount=cases.seamount, grav_adj=cases.grav_adj, overflow=cases.overflow, upwelling_prs31=cases.upwelling_prs31, upwelling_bih=cases.upwelling_bih, upwelling_geouv=cases.upwelling_geouv, upwelling_bihgeouv=cases.upwelling_bihgeouv, upwelling_bihgeo=cases.upwelling_bihgeo, upwelling_bihiso=cases.upwelling_bihiso,
                upwelling_wjgradp=cases.upwelling_prs31, upwelling_kpp_ddmix=cases.upwelling_kpp_ddmix, benchmark_ddmix=cases.benchmark_ddmix, benchmark_bkpp=cases.benchmark_bkpp, upwelling_kpp_bkpp=cases.upwelling_kpp_bkpp, benchmark_wetdry_ddmix=cases.benchmark_wetdry_ddmix, upwelling_prs40=cases.upwelling_prs40, upwelling_prs42=cases.upwelling_prs4x, upwelling_prs44=cases.upwelling_prs4x, upwelling_gls=cases.upwelling_gls, upwelling_gls_ca=cases.upwelling_gls,
                upwelling_gls_cb=cases.upwelling_gls, upwelling_gls_gal=cases.upwelling_gls,
                upwelling_my25=cases.upwelling_my25, upwelling_my25_gal=cases.upwelling_my25)[app]
    lbc = k.pop("lbc", None)
    cs = ctor(**k)
    if tag.endswith("_obc_small") or tag.endswith("_closed_small"):
        cs["EWperiodic"] = 0                 # all four edges are boundaries
    if tag.endswith("_closed_small"):
        cs["closed_state"] = 1               # ... with the set-up arrays of the periodic channel as input data (reference())
        if "mix4" in cs:                     # (VISC4 = 4e8, TNU4 = 2e7 of the channel cases blow up between four walls within 13 steps --
            cs["visc4"], cs["tnu4"] = 4.0e7, (2.0e6, 1.0e6)      # in the reference and, bit for bit, in the oracle; a tenth is stable)
    if lbc is not None:
        cs["lbc"] = lbc
    for n in ("bry_all", "Znudg", "M2nudg", "M3nudg", "Tnudg", "obcfac", "lbc_tke", "clima", "volcons"):
        if n in kw:
            cs[n] = kw[n]
    if cs.get("clima") and "mix4" in cs and cs.get("visc4", 0.0) > 4.0e7:
        # (nudging towards the climatology of cases.clima_arrays sets the channel in motion: with VISC4 = 4e8, TNU4 = 2e7 the run blows up
        # within ten steps -- in the reference and, bit for bit, in the oracle; a tenth is stable)
        cs["visc4"], cs["tnu4"] = 4.0e7, (2.0e6, 1.0e6)
    return app, cs


def reference(app, cs):
    """Reference state after its `initial` sequence."""
    from oracle import ref
    ip, rp = cases.ref_params(cs)
    if cs.get("clima"):                 # climatology nudging: the glue reads the switches from the environment (ref_glue.F90)
        os.environ["ROMS_REF_CLIMA"] = str(int(cs["clima"]))
    else:
        os.environ.pop("ROMS_REF_CLIMA", None)
    R = ref.Ref(app, ip, rp)
    if "MASKING" in cs["options"]:      # the masks are input data (a grid file's mask_rho ...): set before `initial`
        for n, a in cases.land_mask(cs, R.LBi, R.UBi, R.LBj, R.UBj).items():
            if n != "pmask":            # metrics.F derives the slipperiness mask itself
                R.put(n, a)
    R.initial()
    if cs.get("wet_dry"):               # bathymetry and initial free surface of the wetting/drying case, then initial.F:467
        for n, a in cases.wetdry_depth(cs, R.LBi, R.UBi, R.LBj, R.UBj).items():
            if n == "zeta":
                z = R.get("zeta").reshape(3, -1)
                z[:] = a.reshape(1, -1)
                R.put("zeta", z)
            else:
                R.put(n, a)
        R.call("wetdry")
    if cs.get("closed_state"):          # UPWELLING as a closed basin: ana_grid.h gives its bathymetry for a periodic channel only (h = 0
        # otherwise: NaN from the first rho_eos on) -- the set-up arrays of the periodic fixture, cut to the closed basin's arrays, are
        # the input data of the case on both sides (util.closed_basin_state)
        g = util.closed_basin_state(cs, util.load_init(util.init_tag(cs), util.nghost_for(dict(cs, EWperiodic=1))))
        for n in util.INIT_FIELDS:
            if n in g and R.has(n) and R.get(n).size == np.asarray(g[n]).size:
                R.put(n, g[n])
    if cs.get("ddmix"):                 # a state with double diffusion in it (the analytic salinity is uniform): data, cases.ddmix_state
        R.put("t", cases.ddmix_state(cs, R.get("t"), R.LBi, R.UBi, R.LBj, R.UBj))
    if cs.get("clima"):                 # the climatology and coefficient arrays: data (cases.clima_arrays), the reference's compact
        nij = (R.UBi - R.LBi + 1) * (R.UBj - R.LBj + 1)                              # tracer index = the nudged tracers in order
        ca = cases.clima_arrays(cs, nij)
        nudged = [it for it in (1, 2) if cs["clima"] & (1 << it)]
        per = nij * cs["N"]
        if nudged:
            for n in ("tclm", "Tnudgcof"):
                R.put(n, np.concatenate([ca[n][(it - 1) * per:it * per] for it in nudged]))
        if cs["clima"] & 1:
            for n in ("uclm", "vclm", "M3nudgcof"):
                R.put(n, ca[n])
        if cs["clima"] & 32:
            for n in ("ubarclm", "vbarclm", "M2nudgcof"):
                R.put(n, ca[n])
    return R


def oracle_from(R, cs):
    """Oracle state holding a copy of the reference's set-up tables and initial arrays."""
    from oracle import orc
    b = R.bounds(0)
    nd = cs["ndtfast"]
    w = np.stack([R.table(5, 2 * nd), R.table(6, 2 * nd)])
    O = orc.Oracle(cases.oracle_cfg(cs, R.table(7, 8)[0], b[58], w))
    if "mix4" in cs:
        O.set_mix4(*cs["mix4"])
    if cs.get("wet_dry"):
        O.set_wetdry(cs["Dcrit"])
    if cs.get("mix_geo_uv"):
        O.set_geouv()
    if cs.get("prsgrd"):
        O.set_prsgrd(cs["prsgrd"])
    if cs.get("ddmix"):
        O.set_ddmix()
    if cs.get("bkpp"):
        O.set_bkpp()
    if cs.get("clima"):
        O.set_clima(cs["clima"])
        for n, a in cases.clima_arrays(cs, O.ni * O.nj).items():
            O.field(n)[:] = a
    for n in util.INIT_FIELDS + (util.WET_FIELDS if cs.get("wet_dry") else []):
        if R.has(n):
            O.field(n)[:] = R.get(n)
    for k, n in enumerate(["sc_r", "Cs_r", "sc_w", "Cs_w"]):
        O.field(n)[:] = R.table(k + 1, O.field(n).size)
    return O


def shared_fields(R, O):
    out = []
    for n in FIELDS + util.WET_FIELDS + util.BKPP_FIELDS:
        if not R.has(n):
            continue
        try:
            O.field(n)
        except KeyError:
            continue
        out.append(n)
    return out


def mismatches(R, O, names):
    """[(field, number of differing doubles, relative RMS)] over `names`."""
    bad = []
    for n in names:
        a, c = R.get(n), O.field(n)
        if not np.array_equal(a, c):
            bad.append((n, int(np.count_nonzero(a != c)), float(util.relrms(c, a))))
    return bad


def sync_stepping(R, O):
    """copy the oracle's stepping indices into the reference's mod_stepping"""
    s = O.step
    R.set_stepping(s.iic, s.iif, s.nstp, s.nnew, s.nrhs, s.kstp, s.knew, s.krhs, s.predictor, s.time, s.indx1)


def stepping_dict(O):
    s = O.step
    return dict(iic=s.iic, iif=s.iif, nstp=s.nstp, nnew=s.nnew, nrhs=s.nrhs, kstp=s.kstp, knew=s.knew,
                krhs=s.krhs, indx1=s.indx1, predictor=s.predictor, time=s.time)


def main3d_sequence(cs, st, first):
    """The kernel calls of one main3d pass (main3d.F:216-1148) as (kernel name, stepping updates) pairs,
    for drivers that call the C ABI / the reference wrappers one by one.  `st` = dict with iic, indx1,
    time; mutated as main3d mutates mod_stepping."""
    seq = []
    nstp = 1 + (st["iic"] - 1) % 2
    st.update(nstp=nstp, nnew=3 - nstp, nrhs=nstp)
    seq.append(("set_data", dict(st)))
    if first:
        seq += [("ini_zeta", dict(st)), ("set_depth", dict(st)), ("ini_fields", dict(st))]
    seq += [("set_massflux", dict(st)), ("rho_eos", dict(st)), ("diag", dict(st))]
    if "BULK_FLUXES" in cs["options"]:
        seq.append(("bulk_flux", dict(st)))
    seq.append(("set_vbc", dict(st)))
    seq.append(("ana_vmix" if "ANA_VMIX" in cs["options"] else "lmd_vmix", dict(st)))
    seq += [("omega", dict(st)), ("wvelocity", dict(st)), ("set_zeta", dict(st)), ("rhs3d", dict(st))]
    nfast = st["nfast"]
    for my_iif in range(1, nfast + 2):
        nxt = 3 - st["indx1"]
        st.update(predictor=1, iif=my_iif, kstp=st["indx1"] if my_iif == 1 else 3 - st["indx1"], knew=3,
                  krhs=st["indx1"])
        seq.append(("step2d", dict(st)))
        st.update(predictor=0, knew=nxt, kstp=3 - nxt, krhs=3)
        if my_iif < nfast + 1:
            st["indx1"] = nxt
            seq.append(("step2d", dict(st)))
    seq += [("set_depth", dict(st)), ("step3d_uv", dict(st)), ("omega", dict(st)), ("step3d_t", dict(st))]
    st["iic"] += 1
    st["time"] += cs["dt"]
    return seq
