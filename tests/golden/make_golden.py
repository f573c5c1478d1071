#!/usr/bin/env python3
"""Generate the golden fixtures from the reference's own Fortran (oracle/_ref).

Runs only in the build container (needs /root/reference and oracle/_ref/*.so built by
oracle/ref/build_ref.sh).  One configuration per process (the reference keeps its state in
Fortran modules), so this script re-invokes itself per case.

Fixtures (npz, float64 exact):
  <case>_init.npz     set-up tables and the state after the reference's initial sequence
                      (ana_grid, set_scoord, set_weights, metrics, ini_hmixcoef, set_depth,
                      ana_initial, set_depth0/set_zeta_timeavg/set_depth, set_massflux, rho_eos)
  <case>_steps.npz    the reference's state (reference kernels in main3d.F order, ref_glue.F90:
                      ref_main3d) after steps 1, 2, 3 and 100, and every diag line it printed
  <case>_kernels.npz  steps 1 and 2 kernel by kernel: the state before the step, then for every
                      reference kernel call of main3d the arrays it changed (of the 2*nfast+1
                      step2d calls the first four and the last are kept)
  <workload>_sample.npz  BASELINE-size runs (BENCHMARK1/2/3, UPWELLING, 512x512x50, config 5): a
                      regular (xi,eta) sub-sample, all levels, of the reference's end state
  bounds_*.npz        BOUNDS/DOMAIN tables of get_bounds.F for several tilings
The GPU-box tests (tests/test_gpu_vs_reference.py) feed these inputs to the C ABI and compare
with the reference's OWN outputs; only data travels.
"""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)

GRID2D = ["h", "f", "fomn", "pm", "pn", "om_r", "on_r", "om_u", "on_u", "om_v", "on_v", "om_p", "on_p", "omn",
          "pmon_r", "pnom_r", "pmon_p", "pnom_p", "pmon_u", "pnom_u", "pmon_v", "pnom_v", "angler", "xr", "yr",
          "rdrag", "visc2_r", "visc2_p", "diff2"]
STATE = ["Hz", "z_r", "z_w", "Huon", "Hvom", "zeta", "ubar", "vbar", "u", "v", "t", "rho", "pden", "rhoA", "rhoS",
         "Zt_avg1", "Akv", "Akt"]
EXTRA = {"benchmark": ["dmde", "dndx", "lonr", "latr", "rdrag2", "bvf", "alpha", "beta", "hsbl"],
         "kelvin": ["xp", "yp", "rdrag2"], "seamount": ["rdrag2"]}
ALLSTATE = STATE + ["rzeta", "rubar", "rvbar", "W", "wvel", "ru", "rv", "rufrc", "rvfrc", "DU_avg1", "DU_avg2",
                    "DV_avg1", "DV_avg2", "sustr", "svstr", "bustr", "bvstr", "stflx", "btflx", "stflux", "btflux"]


def quiet():
    """silence the Fortran stdout of the reference (it prints its set-up report)"""
    sys.stdout.flush()
    devnull = os.open(os.devnull, os.O_WRONLY)
    saved = os.dup(1)
    os.dup2(devnull, 1)
    return saved


def make_case(case):
    from oracle import ref
    from tests import cases
    tag, kw = case.split(":") if ":" in case else (case, "")
    kwargs = eval("dict(%s)" % kw) if kw else {}
    app = ("benchmark" if tag.startswith("benchmark") else "kelvin" if tag.startswith("kelvin") else
           "seamount" if tag.startswith("seamount") else "grav_adj" if tag.startswith("grav_adj") else
           "overflow" if tag.startswith("overflow") else "upwelling")
    cs = getattr(cases, app)(**kwargs)
    ip, rp = cases.ref_params(cs)
    saved = quiet()
    R = ref.Ref("kelvin_splines" if app == "kelvin" else app, ip, rp)   # (kelvin: oracle/ref/kelvin_splines.h)
    R.initial()
    b = R.bounds(0)
    N, nd = cs["N"], cs["ndtfast"]
    out = dict(
        bounds=np.array(b[:60], dtype=np.int32),
        sc_r=R.table(1, N), Cs_r=R.table(2, N), sc_w=R.table(3, N + 1), Cs_w=R.table(4, N + 1),
        weight=np.stack([R.table(5, 2 * nd), R.table(6, 2 * nd)]), scalars=R.table(7, 8),
    )
    names = GRID2D + STATE + EXTRA.get(app, [])
    for n in names:
        out[n] = R.get(n)
    np.savez_compressed(os.path.join(HERE, f"{tag}_init.npz"), **out)
    os.dup2(saved, 1)
    print(f"wrote {tag}_init.npz  ({len(names)} fields, nfast={b[58]})")


VSTRETCH_SETS = [(2, 2, 5.0, 0.4, 8), (2, 2, 7.0, 2.0, 12), (2, 2, 6.0, 0.0, 8), (3, 2, 1.3, 2.1, 8), (3, 2, 0.65, 0.58, 12), (5, 2, 5.0, 1.5, 8),
                 (5, 2, 0.0, 0.0, 12), (5, 1, 4.0, 0.0, 8), (2, 1, 5.0, 0.4, 8)]


def make_vstretch(one=None):
    """vstretch_tables.npz: sc_r, Cs_r, sc_w, Cs_w of the reference's set_scoord.F for the stretching functions the BASELINE
    applications do not use -- Vstretching 2 (Shchepetkin 2005), 3 (Geyer), 5 (Souza) -- with several parameter sets and both
    Vtransform; z_r, Hz of the state at rest beside them (UPWELLING 14x18, N = 8 / 12).  One process per set: the
    reference library holds one configuration per process."""
    if one is None:
        out = {}
        for q in range(len(VSTRETCH_SETS)):
            subprocess.check_call([sys.executable, __file__, "--vstretch-one", str(q)])
            d = np.load(f"/tmp/vstretch_{q}.npz")
            out.update({k: d[k] for k in d.files})
        np.savez_compressed(os.path.join(HERE, "vstretch_tables.npz"), nsets=np.array(len(VSTRETCH_SETS)), **out)
        print(f"wrote vstretch_tables.npz ({len(VSTRETCH_SETS)} parameter sets)")
        return
    from oracle import ref
    from tests import cases
    q = int(one)
    vs, vt, ths, thb, N = VSTRETCH_SETS[q]
    cs = cases.upwelling(Lm=14, Mm=18, N=N)
    cs.update(Vstretching=vs, Vtransform=vt, theta_s=ths, theta_b=thb, Tcline=(25.0 if vt == 2 else 10.0))
    ip, rp = cases.ref_params(cs)
    saved = quiet()
    R = ref.Ref("upwelling", ip, rp)
    R.initial()
    out = {f"par{q}": np.array([vs, vt, ths, thb, N, cs["Tcline"]])}
    for k, (tab, n) in enumerate((("sc_r", N), ("Cs_r", N), ("sc_w", N + 1), ("Cs_w", N + 1))):
        out[f"{tab}{q}"] = R.table(k + 1, n)
    out[f"z_r{q}"] = R.get("z_r")
    out[f"Hz{q}"] = R.get("Hz")
    os.dup2(saved, 1)
    np.savez_compressed(f"/tmp/vstretch_{q}.npz", **out)


def make_bounds():
    from oracle import ref
    from tests import cases
    spec = sys.argv[3]
    app, Lm, Mm, nti, ntj, hs = spec.split(",")
    kw = dict(Lm=int(Lm), Mm=int(Mm), NtileI=int(nti), NtileJ=int(ntj))
    if app == "upwelling":
        kw.update(hadv=("U3", hs), vadv=("C4", hs))
    cs = getattr(cases, app)(**kw)
    ip, rp = cases.ref_params(cs)
    saved = quiet()
    R = ref.Ref(app, ip, rp)
    tabs = np.array([R.bounds(t)[:58] for t in range(int(nti) * int(ntj))], dtype=np.int32)
    os.dup2(saved, 1)
    np.savez_compressed(os.path.join(HERE, f"bounds_{app}_{Lm}x{Mm}_{nti}x{ntj}_{hs}.npz"), table=tabs,
                        ewp=cs["EWperiodic"], nsp=cs["NSperiodic"])
    print("wrote bounds", spec)


KEEP = ["zeta", "ubar", "vbar", "u", "v", "t", "W", "wvel", "Hz", "z_r", "z_w", "Huon", "Hvom", "rho", "ru", "rv",
        "Zt_avg1", "DU_avg1", "DV_avg1", "DU_avg2", "DV_avg2", "rufrc", "rvfrc", "rzeta", "rubar", "rvbar",
        "Akv", "Akt", "hsbl", "hbbl", "ghats", "stflx", "sustr", "svstr", "bustr", "bvstr", "srflx", "bvf",
        "tke", "gls", "Lscale", "Akk", "Akp",
        "rmask_wet", "umask_wet", "vmask_wet", "pmask_wet", "rmask_full", "umask_full", "vmask_full", "pmask_full", "rmask_wet_avg"]


def _kw(args):
    from tests import refchild
    return refchild.parse(args)


def make_steps(name, tag, args):
    """<name>_steps.npz: the reference after steps 1, 2, 3 and nsteps."""
    import json
    from tests import refdrive as rd
    kw = _kw(args)
    nsteps = kw.pop("nsteps", 100)
    kick_ = kw.pop("kick", 0)
    app, cs = rd.make_case(tag, **kw)
    cs["kick"] = kick_
    saved = rd.quiet()
    R = rd.reference(app, cs)
    names = [n for n in KEEP if R.has(n)]
    out = {}
    kick = cs.pop("kick", 0)
    for s in range(1, nsteps + 1):
        if kick and s == 3:                  # random velocities (per cent of 1 m/s) in front of step 3; the arrays travel with the fixture
            rng = np.random.default_rng(9)
            for n in ("u", "v"):
                a = np.array(R.get(n), dtype=float)
                a = a + 0.01 * kick * rng.standard_normal(a.shape)
                R.put(n, a)
                out[f"kick_{n}"] = np.array(R.get(n))
        R.main3d(1)
        if s in (1, 2, 3, nsteps):
            for n in names:
                out[f"s{s}_{n}"] = R.get(n)
    st = R.get_stepping()
    rd.unquiet(saved)
    lines = rd.diag_lines()
    assert len(lines) == nsteps
    out["meta"] = np.array(json.dumps(dict(tag=tag, case={k: v for k, v in cs.items()}, nsteps=nsteps, fields=names,
                                           stepping=st, diag=lines, diag_text=rd.diag_text())))
    np.savez_compressed(os.path.join(HERE, f"{name}_steps.npz"), **out)
    print(f"wrote {name}_steps.npz ({len(names)} fields x 4 snapshots, {nsteps} diag lines)")


def make_kernels(name, tag, args):
    """<name>_kernels.npz: steps 1 and 2 through the reference's kernel wrappers one call at a time."""
    import json
    from tests import refdrive as rd
    from tests import util
    kw = _kw(args)
    app, cs = rd.make_case(tag, **kw)
    saved = rd.quiet()
    R = rd.reference(app, cs)
    names = [n for n in util.STATE_FIELDS if R.has(n)]
    nfast = R.bounds(0)[58]
    st = dict(iic=1, iif=1, nstp=1, nnew=1, nrhs=1, kstp=1, knew=1, krhs=1, predictor=0, indx1=1, time=0.0,
              nfast=nfast)
    out, seqs = {}, []
    snap = {n: R.get(n) for n in names}
    for step in (1, 2):
        for n in names:
            out[f"s{step}_in_{n}"] = snap[n]
        seq = rd.main3d_sequence(cs, st, first=(step == 1))
        n2d = [k for k, (kern, _) in enumerate(seq) if kern == "step2d"]
        keep2d = set(n2d[:4] + n2d[-1:])
        entries = []
        for k, (kern, s_) in enumerate(seq):
            R.set_stepping(s_["iic"], s_.get("iif", 1), s_["nstp"], s_["nnew"], s_["nrhs"], s_.get("kstp", 1),
                           s_.get("knew", 1), s_.get("krhs", 1), s_.get("predictor", 0), s_["time"], s_["indx1"])
            R.call(kern)
            e = dict(k=kern, st={a: b for a, b in s_.items() if a != "nfast"}, saved=False, out=[])
            if kern != "step2d" or k in keep2d:
                now = {n: R.get(n) for n in names}
                e["saved"] = True
                for n in names:
                    if not np.array_equal(now[n], snap[n]):
                        e["out"].append(n)
                        out[f"s{step}_{k:03d}_{n}"] = now[n]
                snap = now
            entries.append(e)
        seqs.append(entries)
    rd.unquiet(saved)
    lines = rd.diag_lines()
    # the same two steps through ref_main3d (another process wrote <name>_steps.npz): must agree bit for bit
    chk = np.load(os.path.join(HERE, f"{name}_steps.npz"))
    for n in KEEP:
        if f"s2_{n}" in chk.files:
            assert np.array_equal(chk[f"s2_{n}"], snap[n]), n
    out["meta"] = np.array(json.dumps(dict(tag=tag, case=cs, fields=names, nfast=nfast, seq=seqs, diag=lines)))
    np.savez_compressed(os.path.join(HERE, f"{name}_kernels.npz"), **out)
    print(f"wrote {name}_kernels.npz ({sum(len(e['out']) for q in seqs for e in q)} output arrays)")


SAMPLE_FIELDS = ["zeta", "ubar", "vbar", "u", "v", "t", "W", "wvel", "Hz", "rho", "Akv", "Akt", "Huon", "DU_avg1",
                 "Zt_avg1", "hsbl"]


def make_sample(workload, nsteps):
    """<workload>_sample.npz: the reference's end state at a BASELINE size, sub-sampled."""
    import json
    import resource
    import bench
    from tests import refdrive as rd
    resource.setrlimit(resource.RLIMIT_STACK, (resource.RLIM_INFINITY, resource.RLIM_INFINITY))
    cs = bench.params_for(workload, ntimes=nsteps)
    app = {"benchmark": "benchmark", "upwelling_kpp": "upwelling_kpp"}.get(cs["app"], "upwelling")
    saved = rd.quiet()
    R = rd.reference(app, cs)
    import time
    t0 = time.perf_counter()
    R.main3d(1)                      # the first step (start-up branches, post_initial) apart
    t1 = time.perf_counter()
    R.main3d(nsteps - 1)
    t2 = time.perf_counter()
    ni, nj = R.ni, R.nj
    si, sj = max(1, ni // 48), max(1, nj // 24)
    ii, jj = np.arange(0, ni, si), np.arange(0, nj, sj)
    out = dict(ii=ii, jj=jj)
    names = [n for n in SAMPLE_FIELDS if R.has(n)]
    for n in names:
        a = R.get(n).reshape(-1, nj, ni)
        out[n] = np.ascontiguousarray(a[:, jj][:, :, ii])
        out[n + "_rms"] = np.sqrt(np.mean(a ** 2))
    rd.unquiet(saved)
    lines = rd.diag_lines()
    # the reference's own object code timed on one core of the build container: steps 2..nsteps
    cells = cs["Lm"] * cs["Mm"] * cs["N"]
    ref_rate = cells * (nsteps - 1) / (t2 - t1) if nsteps > 1 else 0.0
    out["meta"] = np.array(json.dumps(dict(workload=workload, case=cs, nsteps=nsteps, fields=names, ni=ni, nj=nj,
                                           diag=lines[-1:], ref_first_step_s=t1 - t0, ref_steps_s=t2 - t1,
                                           ref_cell_updates_per_s=ref_rate)))
    print(f"reference Fortran (amdflang -O2, 1 core): {ref_rate:.4g} cell-updates/s over steps 2..{nsteps} ({t2 - t1:.1f} s)")
    np.savez_compressed(os.path.join(HERE, f"{workload}_sample.npz"), **out)
    print(f"wrote {workload}_sample.npz ({len(names)} fields, {len(ii)}x{len(jj)} columns, {nsteps} steps)")


STEP_CASES = [
    # fixture name, refdrive case tag, arguments
    ("upwelling_small_hsimt", "upwelling_small", ["nsteps=100", "hadv=U3,HSIMT", "vadv=C4,HSIMT"]),
    ("upwelling_small_mpdata", "upwelling_small", ["nsteps=100", "hadv=MPDATA,MPDATA", "vadv=MPDATA,MPDATA"]),
    ("upwelling_small_a4spl", "upwelling_small", ["nsteps=30", "hadv=A4,C2", "vadv=SPLINES,C2"]),
    ("upwelling_small_c4su3", "upwelling_small", ["nsteps=30", "hadv=C4,SU3", "vadv=A4,C4"]),
    ("benchmark_small", "benchmark_small", ["nsteps=100"]),
    ("upwelling_kpp_small", "upwelling_kpp_small", ["nsteps=100"]),
    # WINDBASIN's option set on UPWELLING's functions: no UV_ADV, no UV_VIS2, no TS_DIF2 (oracle/ref/upwelling_noadv.h)
    ("upwelling_noadv_small", "upwelling_noadv_small", ["nsteps=60", "hadv=U3,HSIMT", "vadv=C4,HSIMT"]),
    # MASKING: the reference built with oracle/ref/upwelling_mask.h, land of cases.land_mask
    ("upwelling_mask_small", "upwelling_mask_small", ["nsteps=60", "hadv=U3,HSIMT", "vadv=C4,HSIMT"]),
    ("benchmark_mask_small", "benchmark_mask_small", ["nsteps=60"]),      # oracle/ref/benchmark_mask.h
    # MASKING with MPDATA: mpdata_adiff.F's masked blocks
    ("upwelling_mask_small_mpdata", "upwelling_mask_small", ["nsteps=60", "hadv=MPDATA,MPDATA", "vadv=MPDATA,MPDATA"]),
    # open boundaries: the reference's KELVIN application (oracle/ref/kelvin_splines.h)
    ("kelvin_small", "kelvin_small", ["nsteps=96"]),
    ("kelvin_plain_small", "kelvin_plain_small", ["nsteps=96"]),     # ROMS/Include/kelvin.h as shipped: plain vertical solvers
    ("kelvin_plain_small_volcons", "kelvin_plain_small", ["nsteps=60", "volcons=5"]),    # ... with VolCons(west) = VolCons(east) = T (obc_volcons.F), round 6
    # two more of the reference's test applications: SEAMOUNT (pressure-gradient test) and GRAV_ADJ (lock exchange, MPDATA)
    ("seamount_small", "seamount_small", ["nsteps=100"]),
    ("grav_adj_small", "grav_adj_small", ["nsteps=100"]),
    # the generic length-scale closure: upwelling.h -DGLS_MIXING (Kantha-Clayson, k-epsilon), Canuto A masked ("gen"),
    # Canuto B with CHARNOK / CRAIG_BANNER / K_C2ADVECTION (k-kl)
    ("upwelling_small_prs40", "upwelling_prs40_small", ["nsteps=60"]),       # PJ_GRADP, prsgrd40.h
    ("benchmark_small_ddmix", "benchmark_ddmix_small", ["nsteps=60"]),       # LMD_DDMIX, nonlinear EOS (benchmark.h -DLMD_DDMIX; the state of cases.ddmix_state)
    ("upwelling_kpp_small_ddmix", "upwelling_kpp_ddmix_small", ["nsteps=40"]),   # ... linear EOS (upwelling_kpp_ddmix.h)
    # LMD_BKPP (benchmark.h / upwelling_kpp.h -DLMD_BKPP), round 6; kick=30: random velocities of 0.3 m/s in front of step 3 (from rest
    # the layer stays empty), stored with the fixture
    ("benchmark_small_bkpp", "benchmark_bkpp_small", ["nsteps=20", "kick=30"]),
    ("upwelling_kpp_small_bkpp", "upwelling_kpp_bkpp_small", ["nsteps=20", "kick=30"]),
    ("upwelling_small_prs42", "upwelling_prs42_small", ["nsteps=60"]),       # PJ_GRADPQ2, prsgrd42.h (one tile: its second pass, oracle/orc_prs4x.c)
    ("upwelling_small_prs44", "upwelling_prs44_small", ["nsteps=60"]),       # PJ_GRADPQ4, prsgrd44.h
    ("upwelling_small_bih", "upwelling_bih_small", ["nsteps=60"]),           # UV_VIS4 + TS_DIF4 along s-surfaces (upwelling_bih.h)
    ("upwelling_small_geouv", "upwelling_geouv_small", ["nsteps=60"]),       # UV_VIS2 along geopotentials under MASKING (uv3dmix2_geo.h; upwelling_geouv.h)
    ("upwelling_small_bihgeouv", "upwelling_bihgeouv_small", ["nsteps=60"]), # UV_VIS4 along geopotentials under MASKING (uv3dmix4_geo.h; upwelling_bihgeouv.h), round 6
    ("upwelling_small_bihiso", "upwelling_bihiso_small", ["nsteps=60"]),     # ... along isopycnals (t3dmix4_iso.h; upwelling_bihiso.h)
    ("upwelling_small_bihgeo", "upwelling_bihgeo_small", ["nsteps=60"]),     # ... the tracers along geopotentials (t3dmix4_geo.h; upwelling_bihgeo.h)
    ("upwelling_small_wetdry", "upwelling_wetdry_small", ["nsteps=60"]),     # MASKING + WET_DRY (upwelling_wetdry.h; cases.wetdry_depth)
    ("upwelling_small_wetdry_mpdata", "upwelling_wetdry_small", ["nsteps=40", "hadv=MPDATA,MPDATA", "vadv=MPDATA,MPDATA"]),   # mpdata_adiff.F's wet masks
    ("benchmark_small_wetdry", "benchmark_wetdry_small", ["nsteps=60"]),     # WET_DRY with bulk fluxes, solar source, KPP, t3dmix2_geo (benchmark_wetdry.h)
    ("benchmark_small_wetdry_mpdata", "benchmark_wetdry_small", ["nsteps=40", "hadv=MPDATA,MPDATA", "vadv=MPDATA,MPDATA"]),
    ("upwelling_gls_small", "upwelling_gls_small", ["nsteps=60"]),
    ("upwelling_gls_ca_small", "upwelling_gls_ca_small", ["nsteps=60"]),
    ("upwelling_gls_cb_small", "upwelling_gls_cb_small", ["nsteps=60"]),
    ("upwelling_my25_small", "upwelling_my25_small", ["nsteps=60"]),            # MY25_MIXING: upwelling.h -DMY25_MIXING
    ("overflow_small", "overflow_small", ["nsteps=60"]),                        # OVERFLOW: overflow.h (MIX_ISO_TS, Vtransform 1)
]
KERNEL_CASES = ["upwelling_small_hsimt", "upwelling_small_mpdata", "benchmark_small", "upwelling_kpp_small"]
def make_avg():
    """upwelling_small_avg.npz: the 22 time-averaged arrays of the reference's set_avg.F (reference built with AVERAGES,
    oracle/ref/upwelling_avg.h) after the window-closing steps 4 and 7 of a run with nAVG = 3, ntsAVG = 1 -- stepped
    through the reference's kernel wrappers in main3d's order with set_avg behind set_zeta (main3d.F:562)."""
    from tests import refdrive as rd
    from tests import refchild
    app, cs = rd.make_case("upwelling_avg_small")
    saved = rd.quiet()
    R = rd.reference(app, cs)
    R.L.ref_set_avg_window(3, 1, 0, 1)
    nfast = R.bounds(0)[58]
    st = dict(iic=1, iif=1, nstp=1, nnew=1, nrhs=1, kstp=1, knew=1, krhs=1, predictor=0, indx1=1, time=0.0, nfast=nfast)
    out = {}
    for step in range(1, 8):
        for kern, s_ in rd.main3d_sequence(cs, st, first=(step == 1)):
            R.set_stepping(s_["iic"], s_.get("iif", 1), s_["nstp"], s_["nnew"], s_["nrhs"], s_.get("kstp", 1),
                           s_.get("knew", 1), s_.get("krhs", 1), s_.get("predictor", 0), s_["time"], s_["indx1"])
            R.call(kern)
            if kern == "set_zeta":
                R.call("set_avg")
                if step in (4, 7):
                    for n in refchild.AVG_FIELDS:
                        out[f"s{step}_{n}"] = R.get(n)
    os.dup2(saved, 1)
    np.savez_compressed(os.path.join(HERE, "upwelling_small_avg.npz"), nAVG=3, ntsAVG=1, **out)
    print("wrote upwelling_small_avg.npz", len(out), "arrays")


def make_dia():
    """upwelling_small_dia.npz (DIAGNOSTICS_TS and DIAGNOSTICS_UV): DiaTwrk, DiaTrc, avgzeta and the accumulated momentum
    terms DiaU2d, DiaV2d, DiaU3d, DiaV3d (+ DiaU3wrk, DiaV2wrk of one step) of the reference's set_diags.F (reference built from
    ROMS/Include/upwelling.h AS SHIPPED -- DIAGNOSTICS_TS defined -- oracle/ref/build_ref.sh upwelling_diag) after steps 4
    and 7 of a run with nDIA = 3, ntsDIA = 1 (the window-closing calls: DiaTrc converted, ghost points filled), and DiaTwrk
    at the end of step 5 (the raw terms of one step) -- stepped through the reference's kernel wrappers in main3d's order
    with set_diags behind set_zeta (main3d.F:559)."""
    from tests import refdrive as rd
    from tests import refchild
    app, cs = rd.make_case("upwelling_diag_small")
    saved = rd.quiet()
    R = rd.reference(app, cs)
    R.L.ref_set_dia_window(3, 1, 0, 1)
    nfast = R.bounds(0)[58]
    st = dict(iic=1, iif=1, nstp=1, nnew=1, nrhs=1, kstp=1, knew=1, krhs=1, predictor=0, indx1=1, time=0.0, nfast=nfast)
    out = {}
    for step in range(1, 8):
        for kern, s_ in rd.main3d_sequence(cs, st, first=(step == 1)):
            R.set_stepping(s_["iic"], s_.get("iif", 1), s_["nstp"], s_["nnew"], s_["nrhs"], s_.get("kstp", 1),
                           s_.get("knew", 1), s_.get("krhs", 1), s_.get("predictor", 0), s_["time"], s_["indx1"])
            R.call(kern)
            if kern == "set_zeta":
                R.call("set_diags")
                if step in (4, 7):
                    for n in ("DiaTrc", "dia_zeta", "DiaU2d", "DiaV2d", "DiaU3d", "DiaV3d"):
                        out[f"s{step}_{n}"] = R.get(n)
        if step == 5:
            out["e5_DiaTwrk"] = R.get("DiaTwrk")
            out["e5_DiaU3wrk"] = R.get("DiaU3wrk")
            out["e5_DiaV2wrk"] = R.get("DiaV2wrk")
            out["e5_t"] = R.get("t")
    os.dup2(saved, 1)
    np.savez_compressed(os.path.join(HERE, "upwelling_small_dia.npz"), nDIA=3, ntsDIA=1, **out)
    print("wrote upwelling_small_dia.npz", len(out), "arrays", float(np.abs(out["s7_DiaTrc"]).max()))


# (round 3: the full-size cases run 12 steps -- past the start-up branches iic <= ntfirst + 1, into the AB3 steady state)
SAMPLES = [("upwelling", 100), ("benchmark1", 100), ("benchmark2", 100), ("benchmark3", 100), ("ns512", 100), ("ns512u3", 100),
           ("config5", 100)]


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--case":
        make_case(sys.argv[2])
    elif len(sys.argv) > 3 and sys.argv[1] == "--steps":
        make_steps(sys.argv[2], sys.argv[3], sys.argv[4:])
    elif len(sys.argv) > 3 and sys.argv[1] == "--kernels":
        make_kernels(sys.argv[2], sys.argv[3], sys.argv[4:])
    elif len(sys.argv) > 3 and sys.argv[1] == "--sample":
        make_sample(sys.argv[2], int(sys.argv[3]))
    elif len(sys.argv) > 2 and sys.argv[1] == "--vstretch-one":
        make_vstretch(sys.argv[2])
    elif len(sys.argv) > 1 and sys.argv[1] == "--vstretch":
        make_vstretch()
    elif len(sys.argv) > 1 and sys.argv[1] == "--avg":
        make_avg()
    elif len(sys.argv) > 1 and sys.argv[1] == "--dia":
        make_dia()
    elif len(sys.argv) > 1 and sys.argv[1] == "--reference-runs":
        py = sys.executable
        subprocess.check_call([py, __file__, "--avg"])
        subprocess.check_call([py, __file__, "--dia"])
        for name, tag, args in STEP_CASES:
            subprocess.check_call([py, __file__, "--steps", name, tag] + args)
            if name in KERNEL_CASES:
                subprocess.check_call([py, __file__, "--kernels", name, tag] + [a for a in args if "nsteps" not in a])
        for wl, n in SAMPLES:
            subprocess.check_call([py, __file__, "--sample", wl, str(n)])
    elif len(sys.argv) > 2 and sys.argv[1] == "--bounds":
        make_bounds()
    else:
        py = sys.executable
        for case in ["upwelling", "upwelling_small:Lm=14,Mm=18,N=8", "benchmark_small:Lm=24,Mm=16,N=10",
                     "benchmark_small_g3:Lm=24,Mm=16,N=10,hadv=('MPDATA','MPDATA'),vadv=('MPDATA','MPDATA')",   # (three ghost points: MPDATA tracers)
                     "kelvin_small:Lm=16,Mm=12,N=6", "kelvin", "seamount_small:Lm=20,Mm=18,N=8", "seamount",
                     "grav_adj_small:Lm=32,Mm=4,N=10", "grav_adj"]:
            subprocess.check_call([py, __file__, "--case", case])
        for spec in ["upwelling,41,80,1,1,HSIMT", "upwelling,41,80,2,2,HSIMT", "upwelling,41,80,2,4,U3",
                     "upwelling,41,80,3,3,U3", "benchmark,512,64,1,1,U3", "benchmark,512,64,2,2,U3",
                     "benchmark,2048,256,2,4,U3"]:
            subprocess.check_call([py, __file__, "--bounds", "-", spec])
