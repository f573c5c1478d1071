"""Fortran host driver (roms_amd/host): its set-up (roms.in -> grid, s-coordinate, filter weights,
metrics, initial fields) must reproduce the reference's arrays held in tests/golden/*_init.npz bit
for bit -- those fixtures were written by the reference's own set_scoord / set_weights / ana_grid /
metrics / ana_initial / set_depth (tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest

from tests import util

from roms_amd.hostlib import HOST_FIELDS  # noqa: E402  (the arrays the host set-up owns)


def _host(cs):
    from roms_amd import hostlib
    if not os.path.exists(hostlib.LIB):
        from roms_amd import build
        build.build_hip()
        build.build_host()
    return hostlib.Host(params=cs)


@pytest.mark.parametrize("tag", ["upwelling", "upwelling_small", "benchmark_small", "kelvin_small", "kelvin", "seamount_small",
                                 "seamount", "grav_adj_small", "grav_adj", "overflow_small", "overflow"])
def test_host_setup_matches_reference(tag):
    cs = util.case_for(tag)
    g = util.load_init(tag, util.nghost_for(cs))
    H = _host(cs)
    try:
        b = g["bounds"]
        assert [H.dims[k] for k in ("LBi", "UBi", "LBj", "UBj")] == [int(x) for x in b[:4]]
        assert H.dims["Nghost"] == int(b[54]) and H.dims["nfast"] == int(b[58])
        assert H.reals["hc"] == g["scalars"][0] and H.reals["hmin"] == g["scalars"][1]
        assert H.reals["hmax"] == g["scalars"][2]
        w = np.stack([H.get("weight1"), H.get("weight2")])
        assert np.array_equal(w, np.asarray(g["weight"]).reshape(2, -1))
        checked = 0
        for n in HOST_FIELDS:
            if n in g:
                assert np.array_equal(H.get(n), g[n]), n
                checked += 1
        assert checked >= 40
    finally:
        H.finalize()


def test_roms_in_reader_handles_reference_syntax(tmp_path):
    """d-exponents, n*value repeats, continuation lines, comments (Utility/inp_decode.F syntax)."""
    from roms_amd import hostlib
    text = """
! a comment line with MyAppCPP == WRONG
    MyAppCPP = UPWELLING
          Lm == 14            ! Number of I-direction INTERIOR RHO-points
          Mm == 18
           N == 8
   Hadvection == U3       \\                     ! temperature
                 HSIMT                          ! salinity
   Vadvection == C4       \\
                 HSIMT
   LBC(isFsur) ==   Per     Clo     Per     Clo         ! free-surface
ad_LBC(isFsur) ==   Clo     Clo     Clo     Clo
      NTIMES == 7
          DT == 300.0d0
     NDTFAST == 30
        TNU2 == 2*0.0d0
     AKT_BAK == 1.0d-6 1.0d-6
  Vtransform == 2
 Vstretching == 4
     THETA_S == 3.0d0
     THETA_B == 0.0d0
      TCLINE == 25.0d0
"""
    f = tmp_path / "roms_test.in"
    f.write_text(text)
    H = hostlib.Host(infile=str(f))
    try:
        assert (H.dims["Lm"], H.dims["Mm"], H.dims["N"], H.dims["ntimes"]) == (14, 18, 8, 7)
        assert H.dims["hadv"][:2] == [8, 4] and H.dims["vadv"][:2] == [3, 4]
        assert H.dims["EWper"] == 1 and H.dims["NSper"] == 0 and H.dims["Nghost"] == 3
        g = util.load_init("upwelling_small", 3)
        assert np.array_equal(H.get("z_r"), g["z_r"]) and np.array_equal(H.get("t"), g["t"])
    finally:
        H.finalize()


@pytest.mark.parametrize("q", range(9))
def test_vertical_stretching_functions_match_reference(q):
    """Round 6: Vstretching 2 (Shchepetkin 2005), 3 (Geyer), 5 (Souza's quadratic levels) of set_scoord.F:240-337,478-526, several
    parameter sets and both Vtransform: sc_r, Cs_r, sc_w, Cs_w and the depths z_r, Hz of the state at rest equal the arrays the
    reference's own object code wrote (tests/golden/vstretch_tables.npz, make_golden.py --vstretch) bit for bit."""
    from tests import cases
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "vstretch_tables.npz"))
    vs, vt, ths, thb, N, tcl = d[f"par{q}"]
    cs = cases.upwelling(Lm=14, Mm=18, N=int(N))
    cs.update(Vstretching=int(vs), Vtransform=int(vt), theta_s=float(ths), theta_b=float(thb), Tcline=float(tcl))
    H = _host(cs)
    try:
        for n in ("sc_r", "Cs_r", "sc_w", "Cs_w"):
            assert np.array_equal(H.get(n), d[f"{n}{q}"]), (n, H.get(n), d[f"{n}{q}"])
        for n in ("z_r", "Hz"):
            a, b = H.get(n), d[f"{n}{q}"]
            assert a.size == b.size and np.array_equal(a.ravel(), b.ravel()), n
        assert np.all(np.diff(d[f"Cs_w{q}"]) > 0)
    finally:
        H.finalize()


def test_host_needs_device_for_run():
    cs = util.case_for("upwelling_small")
    H = _host(cs)
    try:
        assert H.lib.roms_host_run(1, 0) == 8          # no device context yet: usage error
    finally:
        H.finalize()


# ---------------------------------------------------------------------------------------------------
# roms.in / application-header surface: what the reference would apply or stop on
# must not run silently with different physics here.
# ---------------------------------------------------------------------------------------------------
BASE_IN = """
    MyAppCPP = %(app)s
          Lm == 14
          Mm == 18
           N == 8
   Hadvection == %(h1)s  \\
                 %(h2)s
   Vadvection == %(v1)s \\
                 C4
   LBC(isFsur) ==   %(fs)s
   LBC(isUbar) ==   %(ub)s
   LBC(isVbar) ==   Per     Clo     Per     Clo
   LBC(isUvel) ==   Per     Clo     Per     Clo
   LBC(isVvel) ==   Per     Clo     Per     Clo
   LBC(isMtke) ==   Per     Clo     Per     Clo
   LBC(isTvar) ==   Per     Clo     Per     Clo \\
                    %(tv)s
ad_LBC(isFsur) ==   Rad     Rad     Rad     Rad
      %(extra)s
      NTIMES == 5
          DT == 300.0d0
     NDTFAST == 30
"""
GOOD = dict(app="UPWELLING", v1="C4", h1="U3", h2="U3", fs="Per Clo Per Clo", ub="Per Clo Per Clo", tv="Per Clo Per Clo", extra="")


def _setup(tmp_path, header=None, **kw):
    from roms_amd import hostlib
    f = tmp_path / "roms_case.in"
    f.write_text(BASE_IN % dict(GOOD, **kw))
    return hostlib.Host(infile=str(f), header=header)


def test_reader_accepts_the_reference_spellings(tmp_path):
    """inp_decode.F upper-cases its values and takes the long names of the advection schemes."""
    H = _setup(tmp_path, h1="upstream3", h2="Hsimt", fs="PER clo per CLO")
    try:
        assert H.dims["hadv"][:2] == [8, 4] and H.dims["EWper"] == 1 and H.dims["NSper"] == 0
    finally:
        H.finalize()


@pytest.mark.parametrize("kw,needle", [
    (dict(fs="Red Clo Red Clo"), "LBC(isFsur) = Red"),                       # reduced physics: not built
    (dict(ub="Per Clo Per Nes"), "LBC(isUbar) = Nes"),                        # nesting
    (dict(ub="Clo Clo Clo Clo"), "periodicity differs"),                      # per-variable periodicity
    (dict(fs="Cha Clo Cha Clo"), "periodicity differs"),                      # an open edge is not a periodic one
    (dict(tv="Per Mix Per Clo"), "LBC(isTvar) = Mix"),                        # the salinity line
    (dict(fs="Per Clo Clo Clo"), "opposite edge"),
    (dict(h1="WENO5"), "unknown scheme"),
    (dict(h1="MPDATA"), "MPDATA must be chosen for both"),
    (dict(extra="LuvSrc == T"), "LuvSrc == T"),
    (dict(extra="Vstretching == 6"), "Vstretching"),
    (dict(extra="Ngrids = 2"), "Ngrids"),
])
def test_reader_stops_on_settings_it_cannot_honour(tmp_path, kw, needle):
    """exit_flag 5 with the reason, as checkdefs.F / inp_par.F stop on illegal configurations."""
    from roms_amd import hostlib
    with pytest.raises(hostlib.HostError) as e:
        _setup(tmp_path, **kw).finalize()
    assert e.value.exit_flag == 5 and needle in str(e.value), str(e.value)


def test_header_without_advection_and_mixing_sets_up(tmp_path):
    """An application header with the option set of the reference's WINDBASIN -- no UV_ADV, no UV_VIS2, no TS_DIF2 -- on
    UPWELLING's functions (oracle/ref/upwelling_noadv.h, the header the reference is built from for this pin): the option mask
    carries none of the three bits (refused until round 4: the combination was not pinned)."""
    from roms_amd import hiplib
    hdr = os.path.join(os.path.dirname(__file__), "..", "oracle", "ref", "upwelling_noadv.h")
    H = _setup(tmp_path, header=hdr)
    opt = H.dims["options"]
    assert not (opt & (hiplib.OPTIONS["UV_ADV"] | hiplib.OPTIONS["UV_VIS2"] | hiplib.OPTIONS["TS_DIF2"]))
    assert opt & hiplib.OPTIONS["UV_COR"] and opt & hiplib.OPTIONS["ANA_VMIX"]
    H.finalize()


def test_header_with_viscosity_along_geopotentials_sets_up(tmp_path):
    """UV_VIS2 + MIX_GEO_UV (oracle/ref/upwelling_geouv.h, the header the reference is built from for this pin): the host passes
    the option bit of ABI version 4's upper word and the library takes it (uv3dmix2_geo.h, k_uvmix_geo.h); together with
    MIX_S_UV it is a configuration error."""
    from roms_amd import hiplib, hostlib
    hdr = os.path.join(os.path.dirname(__file__), "..", "oracle", "ref", "upwelling_geouv.h")
    H = _setup(tmp_path, header=hdr)
    assert H.dims["options"] & hiplib.OPTIONS["UV_VIS2"] and H.dims["options"] & hiplib.OPTIONS["MASKING"]
    H.finalize()
    bad = tmp_path / "both.h"
    bad.write_text(open(hdr).read() + "\n#define MIX_S_UV\n")
    with pytest.raises(hostlib.HostError) as e:
        _setup(tmp_path, header=str(bad)).finalize()
    assert e.value.exit_flag == 5 and "MIX_S_UV, MIX_GEO_UV" in str(e.value), str(e.value)


def test_header_with_biharmonic_mixing_along_isopycnals_sets_up(tmp_path):
    """TS_DIF4 + MIX_ISO_TS (oracle/ref/upwelling_bihiso.h: t3dmix4_iso.h, k_t3dmix2_iso in its modes 2 and 3), as a header
    and as the built-in application UPWELLING_BIHISO."""
    from roms_amd import hiplib
    hdr = os.path.join(os.path.dirname(__file__), "..", "oracle", "ref", "upwelling_bihiso.h")
    for kw in (dict(header=hdr), dict(app="UPWELLING_BIHISO")):
        H = _setup(tmp_path, **kw)
        assert H.dims["options"] & hiplib.OPTIONS["MIX_ISO_TS"] and H.dims["options"] & hiplib.OPTIONS["TS_DIF2"]
        H.finalize()


def test_masking_option_and_analytic_masks(tmp_path):
    """MASKING: built-in application UPWELLING_MASK and oracle/ref/upwelling_mask.h as header both set the bit; the
    host's analytic land (roms_host.f90:analytic_masks, psi mask by the rule of metrics.F) equals tests' cases.land_mask,
    which the reference's own metrics.F output was compared with (tests/refdrive.py); MPDATA goes with it."""
    from roms_amd import hiplib, hostlib
    from tests import cases
    hdr = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "oracle", "ref", "upwelling_mask.h"))
    for kw in (dict(header=hdr), dict(app="UPWELLING_MASK")):
        H = _setup(tmp_path, **kw)
        try:
            assert H.dims["options"] & hiplib.OPTIONS["MASKING"]
            cs = dict(Lm=H.dims["Lm"], Mm=H.dims["Mm"], EWperiodic=H.dims["EWper"], NSperiodic=H.dims["NSper"])
            m = cases.land_mask(cs, H.dims["LBi"], H.dims["UBi"], H.dims["LBj"], H.dims["UBj"])
            for n, a in m.items():
                assert np.array_equal(H.get(n), a.ravel()), n
            assert (m["rmask"] == 0).sum() > 10 and (m["pmask"] == 2).sum() > 4
        finally:
            H.finalize()
    # MPDATA under MASKING is built too (mpdata_adiff.F's masked blocks): the set-up goes through
    H = _setup(tmp_path, app="UPWELLING_MASK", h1="MPDATA", v1="MPDATA")
    try:
        assert H.dims["options"] & hiplib.OPTIONS["MASKING"]
    finally:
        H.finalize()


def test_pressure_gradient_scheme_follows_the_header(tmp_path):
    """prsgrd.F:16-26: DJ_GRADPS -> prsgrd32.h; none of the options -> prsgrd31.h (WJ_GRADP: weighted); PJ_GRADP -> prsgrd40.h;
    PJ_GRADPQ2 / PJ_GRADPQ4 -> prsgrd42.h / prsgrd44.h (round 6: bits of the upper option word; they win over every other
    scheme, as prsgrd.F tests them first)"""
    from roms_amd import hiplib, hostlib
    ref = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "oracle", "ref"))
    for hdr, p31, wj in (("upwelling_prs31.h", True, False), ("upwelling_wjgradp.h", True, True), ("upwelling_logdrag.h", False, False)):
        H = _setup(tmp_path, header=os.path.join(ref, hdr))
        try:
            assert bool(H.dims["options"] & hiplib.OPTIONS["PRSGRD31"]) == p31 and bool(H.dims["options"] & hiplib.OPTIONS["WJ_GRADP"]) == wj
        finally:
            H.finalize()
    H = _setup(tmp_path, header=os.path.join(ref, "upwelling_prs40.h"))          # PJ_GRADP -> prsgrd40.h
    try:
        assert H.dims["options"] & hiplib.OPTIONS["PRSGRD40"] and not H.dims["options"] & hiplib.OPTIONS["PRSGRD31"]
    finally:
        H.finalize()
    for hdr in ("upwelling_prs42.h", "upwelling_prs44.h"):
        H = _setup(tmp_path, header=os.path.join(ref, hdr))
        try:
            assert not H.dims["options"] & (hiplib.OPTIONS["PRSGRD40"] | hiplib.OPTIONS["PRSGRD31"])
        finally:
            H.finalize()
    both = tmp_path / "pj.h"                                      # (PJ_GRADPQ4 in front of PJ_GRADP: prsgrd.F:16)
    both.write_text(open(os.path.join(ref, "upwelling_prs40.h")).read() + "\n#define PJ_GRADPQ4\n")
    H = _setup(tmp_path, header=str(both))
    try:
        assert not H.dims["options"] & hiplib.OPTIONS["PRSGRD40"]
    finally:
        H.finalize()


def test_logdrag_header_selects_the_option(tmp_path):
    """a custom application header with UV_LOGDRAG (oracle/ref/upwelling_logdrag.h is one) sets its bit; two drag
    laws at once stop the set-up"""
    from roms_amd import hiplib, hostlib
    hdr = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "oracle", "ref", "upwelling_logdrag.h"))
    H = _setup(tmp_path, header=hdr)
    try:
        assert H.dims["options"] & hiplib.OPTIONS["UV_LOGDRAG"] and not H.dims["options"] & hiplib.OPTIONS["UV_QDRAG"]
    finally:
        H.finalize()
    bad = tmp_path / "two_drags.h"
    bad.write_text(open(hdr).read() + "\n#define UV_LDRAG\n")
    with pytest.raises(hostlib.HostError) as e:
        _setup(tmp_path, header=str(bad)).finalize()
    assert e.value.exit_flag == 5 and "UV_LOGDRAG" in str(e.value)


def test_reader_takes_the_output_keywords(tmp_path):
    """NRREC, NRST, NHIS, LcycleRST, the file names and the Hout switches (read_phypar.F) reach the output module."""
    H = _setup(tmp_path, extra="NRREC == -1\n NRST == 288\n NHIS == 72\n LcycleRST == F\n"
                               " RSTNAME == out/my_rst.nc\n HISNAME == my_his.nc\n ININAME == in/roms_ini.nc")
    try:
        c = H.output_config()
        assert c == dict(nrrec=-1, nRST=288, nHIS=72, LcycleRST=False, nAVG=0, ntsAVG=1, ininame="in/roms_ini.nc",
                         rstname="out/my_rst.nc", hisname="my_his.nc", avgname="roms_avg.nc")
    finally:
        H.finalize()


def test_application_header_is_read_like_cpp_would(tmp_path):
    """#define / #undef / #ifdef / #if defined || && ! / #elif / #else / comments / continuation lines:
    the directive set of ROMS/Include/*.h.  The option mask must follow the live branches only."""
    from roms_amd import hiplib
    hdr = tmp_path / "my_upwelling.h"
    hdr.write_text("""/*
** custom application: a comment block with #define NOT_AN_OPTION inside
*/
#define UV_ADV
#define UV_COR      /* trailing comment */
#define UV_LDRAG
#define UV_VIS2
# define MIX_S_UV
#define TS_DIF2
#define MIX_S_TS
#define SPLINES_VDIFF
#define SPLINES_VVISC
#define DJ_GRADPS
#define SOLVE3D
#if defined BIO_FENNEL || \\
    defined SOLVE3D
# define SALINITY   /* only the clause on the continuation line is true: the branch must be live */
#endif
#define ANA_GRID
#define ANA_INITIAL
#define ANA_SMFLUX
#define ANA_STFLUX
#define ANA_SSFLUX
#define ANA_BTFLUX
#define ANA_BSFLUX
#undef  MIX_GEO_TS
#if defined GLS_MIXING || \\
    defined MY25_MIXING
# define KANTHA_CLAYSON
#elif !defined SOLVE3D && (defined UV_ADV)
# define TS_DIF4
#else
# define ANA_VMIX
#endif
#ifdef BIO_FENNEL
# define CARBON
#endif
#ifndef PERFECT_RESTART
# define AVERAGES
#endif
#if 0
# define UV_VIS4
#endif
""")
    H = _setup(tmp_path, header=str(hdr))
    try:
        want = 0
        for o in ("UV_ADV", "UV_COR", "UV_VIS2", "TS_DIF2", "ANA_VMIX", "SALINITY", "APP_UPWELLING"):
            want |= hiplib.OPTIONS[o]
        assert H.dims["options"] == want, hex(H.dims["options"])
    finally:
        H.finalize()


@pytest.mark.parametrize("line,needle", [("#define TS_DIF4", "TS_DIF4"), ("#define GLS_MIXING", "GLS_MIXING"),
                                         ("#define WET_DRY\n#undef DJ_GRADPS\n#define PJ_GRADP", "WET_DRY"), ("#define LMD_BKPP\n#define LMD_DDMIX", "LMD_BKPP together with"),
                                         ("#define UV_QDRAG", "exactly one of UV_LDRAG, UV_QDRAG, UV_LOGDRAG")])
def test_application_header_with_unbuilt_options_stops(tmp_path, line, needle):
    """An option whose code is not in the library (biharmonic mixing, GLS, wetting and drying with the standard density
    Jacobian, another pressure-gradient scheme ...) is a configuration error (exit_flag 5), never a silent no-op."""
    from roms_amd import hostlib
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    base = open(os.path.join(root, "oracle", "ref", "upwelling_kpp.h")).read()
    hdr = tmp_path / "bad.h"
    hdr.write_text(base + "\n" + line + "\n")
    with pytest.raises(hostlib.HostError) as e:
        _setup(tmp_path, header=str(hdr)).finalize()
    assert e.value.exit_flag == 5 and needle in str(e.value), str(e.value)


def test_biharmonic_header_and_keywords_drive_the_run_the_oracle_makes():
    """oracle/ref/upwelling_bih.h (UV_VIS4 + TS_DIF4 along s-surfaces, the header the reference build the oracle is pinned to
    was made from) read in place, VISC4 / TNU4 from roms.in: three ghost points (inp_par.F:214), the harmonic coefficients
    zero, and six steps of the emulated kernels through the Fortran host equal the oracle's bit for bit -- the built-in
    UPWELLING_BIH option list likewise."""
    from roms_amd import hostlib
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    emu = os.path.join(root, "tests", "emu")
    if not os.path.exists(os.path.join(emu, "libroms_host_emu.so")):
        pytest.skip("emulation not built")
    cs = util.case_for("upwelling_bih_small")
    g = util.load_init("upwelling_small", util.nghost_for(cs))
    O = util.make_oracle(cs, g)
    O.start()
    O.main3d_step(6)
    for params, header in ((dict(cs, ninfo=0), None), (dict(cs, ninfo=0, app="upwelling"), os.path.join(root, "oracle", "ref", "upwelling_bih.h"))):
        H = hostlib.Host(params=params, lib_path=os.path.join(emu, "libroms_host_emu.so"), hip_lib_path=util.EMU_LIB, header=header)
        try:
            assert H.dims["Nghost"] == 3
            assert np.all(H.get("visc2_r") == 0.0) and np.all(H.get("diff2") == 0.0)
            ctx = H.device_init(0)
            assert np.array_equal(ctx.download("visc4_r"), np.full(ctx.download("visc4_r").shape, np.sqrt(cs["visc4"])))
            H.run(6)
            for n in ("u", "v", "t", "zeta", "ubar"):
                assert np.array_equal(ctx.download(n), O.field(n)), (header, n)
        finally:
            H.finalize()


def test_custom_header_of_config5_gives_the_kpp_options():
    """oracle/ref/upwelling_kpp.h -- the custom application header the reference build of BASELINE config 5
    is made with -- read by the product's own reader equals the built-in UPWELLING_KPP list."""
    from roms_amd import hostlib
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    cs = util.case_for("upwelling_kpp_small")
    H = hostlib.Host(params=cs)
    builtin = H.dims["options"]
    H.finalize()
    H = hostlib.Host(params=dict(cs, app="upwelling"), header=os.path.join(root, "oracle", "ref", "upwelling_kpp.h"))
    try:
        assert H.dims["options"] == builtin
    finally:
        H.finalize()


@pytest.mark.ref
@pytest.mark.parametrize("name,app,dims", [("roms_upwelling.in", "upwelling", (41, 80, 16, 1440, 30, 42, 3)),
                                           ("roms_benchmark1.in", "benchmark", (512, 64, 30, 200, 20, 29, 2)),
                                           ("roms_benchmark2.in", "benchmark", (1024, 128, 30, 200, 20, 29, 2))])
def test_reference_input_files_and_headers_read_in_place(name, app, dims):
    """The reference's own ROMS/External/*.in and ROMS/Include/<app>.h (read where they lie; build container
    only): dimensions, stepping, advection schemes, periodicity, ghost points, and an option mask equal to the
    built-in list of the application."""
    from roms_amd import hostlib
    ext = "/root/reference/ROMS/External/" + name
    if not os.path.exists(ext):
        pytest.skip("no reference tree here")
    H = hostlib.Host(infile=ext)
    try:
        d = H.dims
        builtin = d["options"]
        assert (d["Lm"], d["Mm"], d["N"], d["ntimes"], d["ndtfast"], d["nfast"], d["Nghost"]) == dims
        assert d["EWper"] == 1 and d["NSper"] == 0
        assert d["hadv"][:2] == ([8, 4] if app == "upwelling" else [8, 8])
    finally:
        H.finalize()
    H = hostlib.Host(infile=ext, header=f"/root/reference/ROMS/Include/{app}.h")
    try:
        assert H.dims["options"] == builtin
        # output: both .in files ask for averages every 72 steps; only upwelling.h defines AVERAGES
        c = H.output_config()
        assert (c["nHIS"], c["nRST"], c["LcycleRST"], c["nrrec"]) == ((72, 288, True, 0) if app == "upwelling" else
                                                                      (1000, 1000, True, 0))
        assert c["nAVG"] == (72 if app == "upwelling" else 0) and c["avgname"] == "roms_avg.nc"   # NAVG == 1000 in the
        # BENCHMARK inputs, but benchmark.h has no AVERAGES
    finally:
        H.finalize()


def _unknown_keys(H):
    import ctypes as C
    buf = C.create_string_buffer(80)
    n = H.lib.roms_host_unknown_keys(buf, 80)
    return n, buf.value.decode()


@pytest.mark.ref
def test_every_reference_input_file_is_read_without_skipping_a_physics_keyword():
    """Every ROMS/External/roms_*.in of the reference (read where it lies): the set-up either completes or stops with
    exit_flag 5 and a reason (an application / option / boundary condition this build does not have) -- and in both cases
    every keyword of the file is one the reader honours, checks or has classified as unable to change the forward time
    step (docs/ROMS_IN_KEYWORDS.md: all 612 keywords of read_phypar.F).  A keyword outside that table would be skipped
    silently: none may occur."""
    import glob
    from roms_amd import hostlib
    files = sorted(glob.glob("/root/reference/ROMS/External/roms_*.in"))
    if not files:
        pytest.skip("no reference tree here")
    assert len(files) >= 30
    done, stopped = [], []
    for f in files:
        try:
            H = hostlib.Host(infile=f)
        except hostlib.HostError as e:
            assert e.exit_flag == 5 and len(str(e)) > 40, (f, str(e))
            stopped.append(os.path.basename(f))
            lib = hostlib.load(None)
            H = None
        else:
            done.append(os.path.basename(f))
        import ctypes as C
        buf = C.create_string_buffer(80)
        n = hostlib.load(None).roms_host_unknown_keys(buf, 80)
        assert n == 0, (f, n, buf.value.decode())
        if H is not None:
            H.finalize()
    assert {"roms_upwelling.in", "roms_benchmark1.in", "roms_benchmark2.in", "roms_benchmark3.in", "roms_kelvin.in",
            "roms_seamount.in", "roms_grav_adj.in", "roms_overflow.in"} <= set(done)
    assert len(stopped) > 20          # the other applications of the reference: analytic set-ups this host does not have


@pytest.mark.ref
def test_keyword_table_covers_the_reference_reader():
    """docs/ROMS_IN_KEYWORDS.md lists every CASE of ROMS/Utility/read_phypar.F, and the generated include the reader
    compiles holds exactly the keywords classified inert."""
    import re
    src = "/root/reference/ROMS/Utility/read_phypar.F"
    if not os.path.exists(src):
        pytest.skip("no reference tree here")
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    keys = set(re.findall(r"CASE \('([A-Za-z0-9_()%]*)'", open(src).read()))
    rows = re.findall(r"^\| `([^`]+)` \| (\w+) \|", open(os.path.join(root, "docs", "ROMS_IN_KEYWORDS.md")).read(), re.M)
    assert {k for k, _ in rows} == keys and len(rows) == len(keys) == 612
    inc = set(re.findall(r"'([^']+)'", "".join(l for l in open(os.path.join(root, "roms_amd", "host", "roms_in_inert.inc"))
                                                 if l.lstrip().startswith("CASE"))))
    assert inc == {k for k, c in rows if c == "inert"}
    reader = open(os.path.join(root, "roms_amd", "host", "roms_host.f90")).read()
    for k, c in rows:
        if c in ("honoured", "checked"):
            assert f"'{k}'" in reader, k


@pytest.mark.ref
@pytest.mark.parametrize("app", ["upwelling", "benchmark"])
def test_option_echo_is_the_reference_report(tmp_path, app):
    """romsM's " Activated C-preprocessing Options:" block (roms_host.f90:echo_cppdefs) for the reference's own
    application header: the options are those cpp leaves defined after cppdefs.h + globaldefs.h (cpp -dM, run here)
    that checkdefs.F has a line for, in checkdefs.F's order, each with checkdefs.F's text."""
    import re
    import subprocess
    from roms_amd import hostlib
    inc = "/root/reference/ROMS/Include"
    if not os.path.isdir(inc):
        pytest.skip("no reference tree here")
    src = open("/root/reference/ROMS/Utility/checkdefs.F").read()
    pairs = re.findall(r"WRITE \(stdout,20\) '([A-Z0-9_]+)',\s*&\s*\n\s*&\s*'((?:[^']|'')*)'", src)
    order = [p[0] for p in pairs]
    A = app.upper()
    out = subprocess.run(["/usr/bin/cpp", "-P", "-traditional", "-w", "-dM", f"-D{A}", f'-DROMS_HEADER="{app}.h"',
                          f'-DHEADER="{app}.h"', "-DNONLINEAR", f"-I{inc}", f"{inc}/cppdefs.h"], capture_output=True, text=True,
                         cwd=str(tmp_path)).stdout
    macros = {l.split()[1] for l in out.splitlines() if l.startswith("#define")}
    want = [(n, t) for n, t in pairs if n in macros]
    assert len(want) > 25
    name = "roms_upwelling.in" if app == "upwelling" else "roms_benchmark1.in"
    H = hostlib.Host(infile="/root/reference/ROMS/External/" + name, header=f"{inc}/{app}.h")
    try:
        f = str(tmp_path / "echo.txt")
        assert H.lib.roms_host_echo_cppdefs(f.encode()) == 0
        lines = [l.rstrip("\n") for l in open(f)]
    finally:
        H.finalize()
    assert lines[1] == " Activated C-preprocessing Options:" and lines[0] == "" and lines[2] == ""
    assert lines[3].split()[0] == A
    got = [(l[1:26].strip(), l[26:].strip()) for l in lines[4:] if l.strip()]
    assert got == want, (got, want)


def test_open_boundaries_through_the_host(tmp_path):
    """LBC lines with open kinds reach the library's configuration (load_lbc, inp_decode.F:1616-1660); the nudging time
    scales follow inp_par.F:696-752; the reference's KELVIN application sets up as shipped (kelvin.h: no SPLINES_VDIFF /
    SPLINES_VVISC -> the plain vertical solvers) and in its variant with the spline solvers (oracle/ref/kelvin_splines.h,
    built in as KELVIN_SPLINES), also from the reference's own roms_kelvin.in and kelvin.h (NAT = 1: the second tracer
    rides along)."""
    from roms_amd import hostlib
    cs = util.case_for("kelvin_small")
    H = hostlib.Host(params=cs)
    try:
        assert H.dims["EWper"] == 0 and H.dims["NSper"] == 0
        assert H.dims["options"] & hiplib_opt("RADIATION_2D") and H.dims["options"] & hiplib_opt("APP_KELVIN")
    finally:
        H.finalize()
    # as shipped (ROMS/Include/kelvin.h: no SPLINES_VDIFF / SPLINES_VVISC): the plain vertical solvers are selected
    H = hostlib.Host(params=util.case_for("kelvin_plain_small"))
    try:
        assert H.dims["options"] & hiplib_opt("PLAIN_VDIFF") and H.dims["options"] & hiplib_opt("PLAIN_VVISC")
    finally:
        H.finalize()
    ref_in = "/root/reference/ROMS/External/roms_kelvin.in"
    if os.path.exists(ref_in):
        H = hostlib.Host(infile=ref_in, header="/root/reference/ROMS/Include/kelvin.h")
        try:
            assert (H.dims["Lm"], H.dims["Mm"], H.dims["N"]) == (50, 30, 10) and H.dims["ntimes"] == 96
            assert H.dims["options"] & hiplib_opt("PLAIN_VDIFF") and H.dims["options"] & hiplib_opt("RADIATION_2D")
            g = util.load_init("kelvin", 2)
            for n in ("h", "f", "pm", "pn", "z_r", "t", "xp", "yp"):
                assert np.array_equal(H.get(n), g[n]), n
        finally:
            H.finalize()


def hiplib_opt(name):
    from roms_amd import hiplib
    return hiplib.OPTIONS[name]


@pytest.mark.parametrize("app,dims", [("seamount", (49, 48, 13)), ("grav_adj", (128, 4, 40)), ("overflow", (4, 128, 20))])
def test_reference_test_applications_from_their_own_files(app, dims):
    """ROMS/External/roms_<app>.in with ROMS/Include/<app>.h, both read in place: the set-up equals the reference's (the
    fixture its `initial` wrote); the headers' output options (AVERAGES, DIAGNOSTICS_*, ANA_DIAG) select no time-stepping code."""
    from roms_amd import hostlib
    ref_in, ref_h = f"/root/reference/ROMS/External/roms_{app}.in", f"/root/reference/ROMS/Include/{app}.h"
    if not os.path.exists(ref_in):
        pytest.skip("needs the reference tree")
    H = hostlib.Host(infile=ref_in, header=ref_h)
    try:
        assert (H.dims["Lm"], H.dims["Mm"], H.dims["N"]) == dims
        g = util.load_init(app, H.dims["Nghost"])
        for n in ("h", "f", "pm", "pn", "z_r", "z_w", "t", "Hz", "sc_r", "Cs_r"):
            assert np.array_equal(H.get(n), g[n]), n
    finally:
        H.finalize()
