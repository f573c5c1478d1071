"""Kernel-logic tests without a GPU: the HIP kernel sources compiled with -DROMS_CPU_EMU
(tests/emu/build_emu.sh; blocks/threads run serially on the host) against the oracle, bit for bit.
This is a development aid for machines without a GPU; the product library has no CPU path and the
-m gpu tests are the parity tests proper."""
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import util


@pytest.fixture(scope="module")
def emu():
    subprocess.check_call([os.path.join(os.path.dirname(util.EMU_LIB), "build_emu.sh")],
                          stdout=subprocess.DEVNULL)
    return util.EMU_LIB


@pytest.mark.parametrize("hadv,vadv", [(("U3", "HSIMT"), ("C4", "HSIMT")), (("U3", "U3"), ("C4", "C4")),
                                       (("C4", "A4"), ("SPLINES", "A4")), (("C2", "SU3"), ("C2", "SU3")),
                                       (("MPDATA", "MPDATA"), ("MPDATA", "MPDATA")),
                                       (("MPDATA", "HSIMT"), ("MPDATA", "HSIMT")), (("U3", "MPDATA"), ("C4", "MPDATA"))])
def test_main3d_sequence_bitwise(emu, hadv, vadv):
    cs = util.case_for("upwelling_small", hadv=hadv, vadv=vadv)
    g = util.load_init("upwelling_small", util.nghost_for(cs))
    O = util.make_oracle(cs, g)
    H = util.make_hip(cs, g, emu)
    O.start()
    H.start()
    for _ in range(6):           # covers the Euler, AB2 and AB3 start-up branches
        O.main3d_step()
        H.main3d(1)
        for n in util.PROGNOSTIC:
            assert np.array_equal(H.download(n), O.field(n)), n
    assert O.diag() == H.diag()
    H.close()


@pytest.mark.parametrize("hadv,vadv,ng", [(("U3", "U3"), ("C4", "C4"), 2),
                                          (("MPDATA", "MPDATA"), ("MPDATA", "MPDATA"), 3)])
def test_closed_basin_all_edges(emu, hadv, vadv, ng):
    """Closed western/eastern edges too (all four walls): exercises the non-periodic branches of
    every kernel (edge replication of curvature terms, corner fills, wall-normal velocities; with
    MPDATA the wall values of Ta, Ua, Va)."""
    cs = util.case_for("upwelling_small", hadv=hadv, vadv=vadv)
    g = util.load_init("upwelling_small", ng)
    cs["EWperiodic"] = 0
    # re-embed the periodic fixture (LBi=-2..UBi) into closed-basin arrays (0..Im+1)
    LBi, UBi, LBj, UBj = [int(x) for x in g["bounds"][:4]]
    ni, nj = UBi - LBi + 1, UBj - LBj + 1
    Lm = cs["Lm"]
    Im = Lm + ((Lm + 2) // 2 - (Lm + 1) // 2)
    for k, a in list(g.items()):
        if a.ndim == 1 and a.size >= ni * nj and a.size % (ni * nj) == 0:
            a = a.reshape(-1, nj, ni)[:, :, (0 - LBi):(Im + 1 - LBi) + 1]
            g[k] = np.ascontiguousarray(a).ravel()
    O = util.make_oracle(cs, g)
    H = util.make_hip(cs, g, emu)
    O.start()
    H.start()
    for _ in range(4):
        O.main3d_step()
        H.main3d(1)
        for n in util.PROGNOSTIC:
            assert np.array_equal(H.download(n), O.field(n)), n
    assert np.isfinite(O.diag()[0])
    H.close()


def test_upwelling_kpp_mpdata_bitwise(emu):
    """BASELINE config 5 physics on a small grid: UPWELLING + KPP (linear EOS with bvf/alpha/beta) + MPDATA."""
    cs = util.case_for("upwelling_kpp_small")
    g = util.load_init("upwelling_small", 3)
    O = util.make_oracle(cs, g)
    H = util.make_hip(cs, g, emu)
    O.start()
    H.start()
    for _ in range(5):
        O.main3d_step()
        H.main3d(1)
        for n in util.PROGNOSTIC:
            assert np.array_equal(H.download(n), O.field(n)), n
    assert float(O.field("Akv").max()) > 1.0e-5          # the closure is active
    H.close()


def test_logarithmic_bottom_drag_bitwise(emu):
    """UV_LOGDRAG (set_vbc.F:591-635; oracle pinned to the reference built from oracle/ref/upwelling_logdrag.h):
    the kernel's branch against the oracle's over 6 steps, and it does differ from the linear drag."""
    cs = util.case_for("upwelling_logdrag_small")
    g = util.load_init("upwelling_small", util.nghost_for(cs))
    O = util.make_oracle(cs, g)
    H = util.make_hip(cs, g, emu)
    O.start()
    H.start()
    for _ in range(6):
        O.main3d_step()
        H.main3d(1)
        for n in util.PROGNOSTIC:
            assert np.array_equal(H.download(n), O.field(n)), n
    lin = util.make_oracle(util.case_for("upwelling_small", hadv=cs["hadv"], vadv=cs["vadv"]), g)
    lin.start()
    lin.main3d_step(6)
    assert np.abs(lin.field("bustr") - O.field("bustr")).max() > 1e-9
    H.close()


@pytest.mark.parametrize("hadv,vadv", [(("U3", "HSIMT"), ("C4", "HSIMT")), (("A4", "C4"), ("SPLINES", "C4")), (("C2", "SU3"), ("C2", "A4")),
                                       (("MPDATA", "MPDATA"), ("MPDATA", "MPDATA")), (("U3", "MPDATA"), ("C4", "MPDATA"))])
def test_land_sea_masking_bitwise(emu, hadv, vadv):
    """MASKING (island + headland of cases.land_mask; oracle pinned to the reference built from oracle/ref/upwelling_mask.h):
    every masked kernel branch -- barotropic step, closed-boundary fills, EOS, pressure gradient, advection incl. HSIMT,
    mixing, step3d_uv/t, MPDATA's anti-diffusive velocities and limiter, the first-step loads of ini_fields -- against the oracle's over 8 steps, bit for bit; land
    stays land (u, v, zeta, rho zero there) and the run differs from the unmasked one."""
    cs = util.case_for("upwelling_mask_small", hadv=hadv, vadv=vadv)
    g = util.with_masks(cs, util.load_init("upwelling_small", util.nghost_for(cs)))
    O = util.make_oracle(cs, g)
    H = util.make_hip(cs, g, emu)
    assert np.array_equal(H.download("rmask"), g["rmask"]) and (g["rmask"] == 0).sum() > 10
    O.start()
    H.start()
    for _ in range(8):
        O.main3d_step()
        H.main3d(1)
        for n in util.PROGNOSTIC:
            a, b = H.download(n), O.field(n)
            assert np.isfinite(b).all(), n
            assert np.array_equal(a, b), (n, float(np.abs(a - b).max()))
    land = g["rmask"] == 0
    nn = land.size
    for n in ("zeta", "rho", "t"):
        a = H.download(n)
        assert not a.reshape(-1, nn)[:, land].any(), n
    assert np.abs(H.download("u")).max() > 0
    plain = util.make_oracle(util.case_for("upwelling_small", hadv=hadv, vadv=vadv), g)
    plain.start()
    plain.main3d_step(8)
    assert np.abs(plain.field("zeta") - O.field("zeta")).max() > 1e-9
    H.close()


@pytest.mark.parametrize("hadv,vadv,ewp", [(("U3", "HSIMT"), ("C4", "HSIMT"), 1), (("A4", "C4"), ("SPLINES", "C4"), 1), (("U3", "U3"), ("C4", "C4"), 0)])
def test_wetting_and_drying_bitwise(emu, hadv, vadv, ewp):
    """WET_DRY (a beach that dries above the still water level and a ridge of water that runs up it, cases.wetdry_depth; oracle
    pinned to the reference built from oracle/ref/upwelling_wetdry.h): the masks of every fast step and the time-averaged ones
    (k_wetdry), the wet/dry branches of the barotropic kernel, of prsgrd / rhs3d / t3dmix2 / uv3dmix2 / step3d_uv / set_vbc with
    the bottom-stress limiter, the first-step loads and the wetting/drying conditions of the boundary routines -- periodic
    channel and closed basin (all four walls: v2dbc's western edge as written) -- against the oracle over 30 steps, bit for bit;
    cells flip between wet and dry on the way."""
    cs = util.case_for("upwelling_wetdry_small", hadv=hadv, vadv=vadv)
    if ewp:
        g = util.with_wetdry(cs, util.load_init("upwelling_small", util.nghost_for(cs)))
    else:
        cs["EWperiodic"] = 0
        g = util.with_wetdry(cs, util.closed_basin_state(cs, util.load_init("upwelling_small", util.nghost_for(cs))))
    O = util.make_oracle(cs, g)
    H = util.make_hip(cs, g, emu)
    O.call("wetdry_ini")
    H.wetdry_ini()
    for n in util.WET_FIELDS[:8]:
        assert np.array_equal(H.download(n), O.field(n)), n
    O.start()
    H.start()
    flips = 0
    prev = O.field("rmask_wet").copy()
    for _ in range(30):
        O.main3d_step()
        H.main3d(1)
        for n in util.PROGNOSTIC + util.WET_FIELDS:
            a, b = H.download(n), O.field(n)
            assert np.isfinite(b).all(), n
            assert np.array_equal(a, b), (n, float(np.abs(a - b).max()))
        flips += int((O.field("rmask_wet") != prev).sum())
        prev = O.field("rmask_wet").copy()
    assert flips > 0 and np.abs(H.download("u")).max() > 0
    dry = O.field("rmask_wet") == 0
    assert dry.sum() > (g["rmask"] == 0).sum()                      # more than the land is dry
    H.close()


WD_OBC = {
    # open western and eastern edges, walls south and north (the beach ends at the northern wall)
    "we": dict(zeta=("Che", "Clo", "Rad", "Clo"), ubar=("Shc", "Clo", "Rad", "Clo"), vbar=("Shc", "Clo", "Gra", "Clo"),
               u=("RadNud", "Clo", "Gra", "Clo"), v=("Gra", "Clo", "RadNud", "Clo"), temp=("RadNud", "Clo", "Cla", "Clo"),
               salt=("Cla", "Clo", "Gra", "Clo")),
    # all four edges open: the northern one lies on the dry beach (zetabc's "water level above bed elevation"), u's gradient
    # condition at the southern edge is the one the reference leaves without the wet mask
    "four": dict(zeta=("Cha", "Gra", "Che", "Cla"), ubar=("Fla", "Gra", "Shc", "Rad"), vbar=("Fla", "Shc", "Gra", "Rad"),
                 u=("Gra", "Gra", "Rad", "Cla"), v=("Rad", "Gra", "Gra", "Gra"), temp=("Rad", "Gra", "Gra", "Rad"),
                 salt=("Gra", "Rad", "Rad", "Gra")),
}


@pytest.mark.parametrize("variant", sorted(WD_OBC))
def test_wetting_and_drying_with_open_boundaries_bitwise(emu, variant):
    """WET_DRY together with open boundaries (k_obc.h: Chapman with MAX(depth, Dcrit), Shchepetkin with the free surface in its
    depth, the wet masks on every open kind of u3dbc / v3dbc but the one the reference guards with an undefined name, the
    wetting/drying conditions behind zetabc / u2dbc / v2dbc; the oracle's routines are pinned to the reference's under every
    kind, tests/test_oracle_vs_ref.py): 20 steps against the oracle, bit for bit, masks included."""
    cs = util.case_for("upwelling_wetdry_small", hadv=("U3", "U3"), vadv=("C4", "C4"))
    cs["EWperiodic"] = 0
    cs["lbc"] = WD_OBC[variant]
    cs.update(Znudg=0.5, M2nudg=0.25, M3nudg=2.0, Tnudg=(1.0, 3.0), obcfac=4.0)
    g = util.with_wetdry(cs, util.closed_basin_state(cs, util.load_init("upwelling_small", util.nghost_for(cs))))
    O = util.make_oracle(cs, g)
    H = util.make_hip(cs, g, emu)
    rng = np.random.default_rng(3)
    from roms_amd import hiplib
    for n in hiplib.BRY_FIELDS:
        a = O.field(n)
        a[:] = (15.0 if n[0] == "t" else 0.0) + 0.01 * rng.standard_normal(a.size)
        H.upload(n, a)
    O.start()
    H.start()
    for _ in range(20):
        O.main3d_step()
        H.main3d(1)
        for n in util.PROGNOSTIC + util.WET_FIELDS:
            a, b = H.download(n), O.field(n)
            assert np.isfinite(b).all(), n
            assert np.array_equal(a, b), (n, float(np.abs(a - b).max()))
    assert np.abs(O.field("u")).max() > 0.01 and (O.field("rmask_wet") == 0).sum() > (g["rmask"] == 0).sum()
    H.close()


def test_wetting_and_drying_through_the_fortran_host_matches_the_oracle(emu):
    """roms.in (MyAppCPP = UPWELLING_WETDRY, DCRIT) -> Fortran host (WET_DRY switches MASKING on; its analytic beach and ridge of
    water; roms_hip_wetdry_config / _ini) -> C ABI -> kernels: the host's bathymetry and initial free surface are the numbers of
    cases.wetdry_depth, and 12 steps give the same bits as the oracle started from the state the host uploaded."""
    from roms_amd import hostlib
    cs = util.case_for("upwelling_wetdry_small", hadv=("U3", "HSIMT"), vadv=("C4", "HSIMT"))
    g = util.with_wetdry(cs, util.load_init("upwelling_small", util.nghost_for(cs)))
    H = hostlib.Host(params=dict(cs, ninfo=0), lib_path=os.path.join(os.path.dirname(emu), "libroms_host_emu.so"), hip_lib_path=emu)
    ctx = H.device_init()
    assert np.array_equal(ctx.download("h"), g["h"]) and np.array_equal(ctx.download("zeta"), g["zeta"])
    assert np.array_equal(ctx.download("rmask"), g["rmask"])
    g2 = dict(g)
    for n in util.INIT_FIELDS + util.WET_FIELDS:
        if n in g or n in util.WET_FIELDS:
            g2[n] = ctx.download(n)
    O = util.make_oracle(cs, g2)
    O.start()
    O.main3d_step(12)
    H.run(12)
    for n in ("zeta", "u", "v", "t", "ubar", "vbar", "rho", "W", "rmask_wet", "umask_wet", "vmask_wet", "pmask_wet", "rmask_full"):
        assert np.array_equal(ctx.download(n), O.field(n)), n
    assert (O.field("rmask_wet") == 0).sum() > (g["rmask"] == 0).sum()
    H.finalize()


@pytest.mark.parametrize("drop", [("UV_ADV",), ("UV_VIS2",), ("TS_DIF2",), ("UV_ADV", "UV_VIS2", "TS_DIF2")])
@pytest.mark.parametrize("hadv,vadv", [(("U3", "HSIMT"), ("C4", "HSIMT")), (("MPDATA", "MPDATA"), ("MPDATA", "MPDATA"))])
def test_applications_without_advection_or_mixing_bitwise(emu, drop, hadv, vadv):
    """An application header without UV_ADV, UV_VIS2 or TS_DIF2 (the option set of the reference's WINDBASIN; refused until
    round 4): the kernels skip what the reference's cpp drops -- rhs3d.F:765-1330, step2d_LF_AM3.h:1246-1660, uv3dmix2 /
    t3dmix2 -- and give the oracle's bits over 12 steps.  The oracle is pinned to the reference built without the three
    (oracle/ref/upwelling_noadv.h; tests/test_oracle_vs_ref.py, tests/golden/upwelling_noadv_small_steps.npz)."""
    cs = util.case_for("upwelling_small", hadv=hadv, vadv=vadv)
    cs["options"] = tuple(o for o in cs["options"] if o not in drop)
    g = util.load_init("upwelling_small", util.nghost_for(cs))
    O = util.make_oracle(cs, g)
    H = util.make_hip(cs, g, emu)
    O.start()
    H.start()
    O.main3d_step(12)
    H.main3d(12)
    for n in util.PROGNOSTIC:
        assert np.array_equal(H.download(n), O.field(n)), n
    H.close()


def test_wet_dry_refusals_hold_in_either_call_order(emu):
    """The combinations WET_DRY is not built with are refused with exit_flag 5 wherever they are asked for: the biharmonic
    operators (option bits of the same roms_hip_config since ABI version 4) by roms_hip_create, DIAGNOSTICS_TS
    by its configuration call (the refusal used to sit in the WET_DRY call only, and the host made that
    one first), and the host's own reader stops a WET_DRY run that asks for diagnostics.  (AVERAGES with WET_DRY is built
    since round 6: test_time_averages_bitwise.)"""
    from roms_amd import hiplib, hostlib
    cs = util.case_for("upwelling_wetdry_small", hadv=("U3", "HSIMT"), vadv=("C4", "HSIMT"))
    g = util.with_wetdry(cs, util.load_init("upwelling_small", util.nghost_for(cs)))
    for call in (lambda H: H.dia_config(4),):
        H = util.make_hip(cs, g, emu)
        with pytest.raises(hiplib.RomsHipError, match="WET_DRY"):
            call(H)
        H.close()
    with pytest.raises(hiplib.RomsHipError, match="WET_DRY"):
        util.make_hip(dict(cs, mix4=(0, 1), visc4=0.0, tnu4=(1.0, 1.0)), g, emu)
    with pytest.raises(hiplib.RomsHipError, match="WET_DRY"):
        util.make_hip(dict(cs, dia_uv=True), g, emu).dia_config(4, uv=True)
    with pytest.raises(hostlib.HostError, match="WET_DRY together with DIAGNOSTICS"):
        hostlib.Host(params=dict(cs, ninfo=0, NDIA=4, Dout={"idDtrc(iThadv)": (True, True)}), lib_path=os.path.join(os.path.dirname(emu), "libroms_host_emu.so"),
                     hip_lib_path=emu)


@pytest.mark.parametrize("tag,adv", [("benchmark_wetdry_small", None), ("benchmark_wetdry_small", "MPDATA"), ("upwelling_wetdry_small", "MPDATA")])
def test_wet_dry_with_bulk_fluxes_kpp_geopotential_mixing_and_mpdata_bitwise(emu, tag, adv):
    """WET_DRY beyond round 4's set (oracle pinned to the reference built from oracle/ref/benchmark_wetdry.h and
    upwelling_wetdry.h): the wet masks of bulk_flux.F, of the solar source in pre_step3d.F:903, of t3dmix2_geo.h and of
    mpdata_adiff.F, with KPP on the beach -- the kernels against the oracle's over 10 steps, bit for bit."""
    kw = dict(hadv=(adv, adv), vadv=(adv, adv)) if adv else {}
    if tag.startswith("upwelling") and not adv:
        kw = dict(hadv=("U3", "HSIMT"), vadv=("C4", "HSIMT"))
    cs = util.case_for(tag, **kw)
    g = util.with_wetdry(cs, util.load_init(util.init_tag(cs), util.nghost_for(cs)))
    O = util.make_oracle(cs, g)
    H = util.make_hip(cs, g, emu)
    O.start()
    H.start()
    for _ in range(10):
        O.main3d_step()
        H.main3d(1)
        for n in util.PROGNOSTIC + ["rmask_wet", "umask_wet", "vmask_wet"]:
            a, b = H.download(n), O.field(n)
            assert np.isfinite(b).all(), n
            assert np.array_equal(a, b), (n, float(np.abs(a - b).max()))
    assert (H.download("rmask_wet") == 0).any() and (H.download("rmask_wet") == 1).any()
    H.close()


def test_land_sea_masking_benchmark_physics_bitwise(emu):
    """MASKING with the BENCHMARK physics (oracle pinned to the reference built from oracle/ref/benchmark_mask.h): the
    masked branches of the nonlinear EOS, the COARE bulk fluxes, KPP (surface boundary layer) and the geopotential
    tracer mixing kernels against the oracle's over 8 steps, bit for bit; directly and through the Fortran host
    (MyAppCPP = BENCHMARK_MASK: its own analytic land)."""
    from roms_amd import hostlib
    cs = util.case_for("benchmark_mask_small")
    g = util.with_masks(cs, util.load_init("benchmark_small", util.nghost_for(cs)))
    O = util.make_oracle(cs, g)
    H = util.make_hip(cs, g, emu)
    O.start()
    H.start()
    for _ in range(8):
        O.main3d_step()
        H.main3d(1)
        for n in util.PROGNOSTIC:
            a, b = H.download(n), O.field(n)
            assert np.isfinite(b).all(), n
            assert np.array_equal(a, b), (n, float(np.abs(a - b).max()))
    land = g["rmask"] == 0
    for n in ("rho", "hsbl", "stflx", "t"):
        assert not H.download(n).reshape(-1, land.size)[:, land].any(), n
    H.close()
    Hh = hostlib.Host(params=dict(cs, ninfo=0), lib_path=os.path.join(os.path.dirname(emu), "libroms_host_emu.so"), hip_lib_path=emu)
    ctx = Hh.device_init()
    Hh.run(8)
    for n in ("zeta", "u", "v", "t", "rho", "Akv", "hsbl"):
        assert np.array_equal(ctx.download(n), O.field(n)), n
    Hh.finalize()


def test_masked_run_through_the_fortran_host_matches_the_oracle(emu):
    """roms.in (MyAppCPP = UPWELLING_MASK) -> Fortran host (its own analytic land, option surface, uploads) -> C ABI ->
    kernels: the same bits as the oracle driven with tests' cases.land_mask, 6 steps."""
    from roms_amd import hostlib
    cs = util.case_for("upwelling_mask_small", hadv=("U3", "HSIMT"), vadv=("C4", "HSIMT"))
    g = util.with_masks(cs, util.load_init("upwelling_small", util.nghost_for(cs)))
    O = util.make_oracle(cs, g)
    O.start()
    O.main3d_step(6)
    H = hostlib.Host(params=dict(cs, ninfo=0), lib_path=os.path.join(os.path.dirname(emu), "libroms_host_emu.so"), hip_lib_path=emu)
    ctx = H.device_init()
    H.run(6)
    for n in ("zeta", "u", "v", "t", "ubar", "vbar", "rho", "W"):
        assert np.array_equal(ctx.download(n), O.field(n)), n
    H.finalize()


@pytest.mark.parametrize("hadv,vadv,ng,ewp", [(("U3", "U3"), ("C4", "C4"), 2, 1), (("U3", "HSIMT"), ("C4", "HSIMT"), 3, 1),
                                              (("U3", "U3"), ("C4", "C4"), 2, 0)])
def test_ns_periodic(emu, hadv, vadv, ng, ewp):
    """A periodic eta direction (doubly periodic, and eta-periodic with closed xi walls): the north-south
    periodic branches of every kernel -- ghost rows by local copy, no wall values, full-range index
    bounds.  The EW-periodic fixture is re-embedded: ghost rows are periodic images of the interior rows
    (the fields need not be smooth across the seam for a bit-for-bit comparison over a few steps)."""
    cs, g = util.ns_periodic_case(hadv, vadv, ng, ewp)
    O = util.make_oracle(cs, g)
    H = util.make_hip(cs, g, emu)
    O.start()
    H.start()
    for _ in range(4):
        O.main3d_step()
        H.main3d(1)
        for n in util.PROGNOSTIC:
            a, b = H.download(n), O.field(n)
            assert np.isfinite(b).all(), n
            assert np.array_equal(a, b), (n, float(np.abs(a - b).max()))
    H.close()


def test_romsM_run_report_is_the_reference_text(emu, tmp_path):
    """romsM (linked against the emulated kernels here; the GPU-box twin is in test_gpu_vs_reference.py):
    roms.in -> Fortran host -> C ABI -> kernels, and on standard output the reference's run report, character
    for character (the emulated exp() is glibc's, as the reference's)."""
    exe = os.path.join(os.path.dirname(util.EMU_LIB), "romsM_emu")
    util.check_romsM_report(exe, tmp_path, exact=True)
    # the reference's KELVIN application (open boundaries), as shipped and with the spline vertical solvers
    util.check_romsM_report(exe, tmp_path, exact=True, fixture="kelvin_plain_small_steps.npz")
    util.check_romsM_report(exe, tmp_path, exact=True, fixture="kelvin_small_steps.npz")
    # ... with VolCons(west) == VolCons(east) == T in roms.in (obc_volcons.F; round 6)
    util.check_romsM_report(exe, tmp_path, exact=True, fixture="kelvin_plain_small_volcons_steps.npz")
    # SEAMOUNT and GRAV_ADJ
    util.check_romsM_report(exe, tmp_path, exact=True, fixture="seamount_small_steps.npz")
    util.check_romsM_report(exe, tmp_path, exact=True, fixture="grav_adj_small_steps.npz")
    util.check_romsM_report(exe, tmp_path, exact=True, fixture="overflow_small_steps.npz")            # OVERFLOW, MIX_ISO_TS
    util.check_romsM_report(exe, tmp_path, exact=True, fixture="upwelling_gls_small_steps.npz")       # GLS_MIXING
    util.check_romsM_report(exe, tmp_path, exact=True, fixture="upwelling_gls_cb_small_steps.npz")


def test_product_partition_matches_reference_get_bounds(emu):
    """The host's tile rectangles and the library's derived BOUNDS/DOMAIN entries == get_bounds.F's tables for
    UPWELLING 1x1/2x2/2x4/3x3 and BENCHMARK1 1x1, BENCHMARK1 2x2, BENCHMARK3 2x4 (tests/golden/bounds_*.npz)."""
    host = os.path.join(os.path.dirname(emu), "libroms_host_emu.so")
    assert util.check_tile_bounds(host_lib=host, hip_lib=emu) >= 30


@pytest.mark.parametrize("tag,nAVG,ntsAVG", [("upwelling_small", 3, 1), ("benchmark_small", 2, 2), ("upwelling_small", 1, 1),
                                             ("upwelling_wetdry_small", 3, 1), ("upwelling_wetdry_small", 1, 1)])
def test_time_averages_bitwise(emu, tag, nAVG, ntsAVG):
    """set_avg (k_avg.h) inside roms_hip_main3d against the oracle's (pinned to set_avg.F): all 22 averaged arrays
    after every step -- set, add and convert phases of several windows; averaging does not change the run.
    upwelling_wetdry (round 6): WET_DRY -- every field times the full mask (land x wet) of its grid type, the sums divided by the
    number of steps the point was wet (set_avg.F:257-288, :302 ..., :2980-2988); the oracle equals the reference built from
    oracle/ref/upwelling_wetdry_avg.h bit for bit (tests/test_oracle_vs_ref.py)."""
    from roms_amd import hiplib
    cs = util.case_for(tag)
    g = util.load_init(util.init_tag(cs), util.nghost_for(cs))
    if cs.get("wet_dry"):
        g = util.with_wetdry(cs, g)
    O = util.make_oracle(cs, g)
    H = util.make_hip(cs, g, emu)
    O.set_avg_window(nAVG, ntsAVG)
    H.avg_config(nAVG, ntsAVG)
    O.start()
    H.start()
    for step in range(1, 8):
        O.main3d_step()
        H.main3d(1)
        for n in hiplib.Context.AVG_FIELDS:
            assert np.array_equal(H.download(n), O.field(n)), (step, n)
        for n in ("u", "t", "zeta"):
            assert np.array_equal(H.download(n), O.field(n)), (step, n)
    assert np.abs(H.download("avg_UV")).max() > 0.0
    H.close()


OBC_VARIANTS = {
    # the reference's KELVIN application as shipped (roms_kelvin.in): Chapman / Flather west, radiation east, walls south / north
    "kelvin": None,
    # ... with the plain tridiagonal vertical solvers kelvin.h itself selects (no SPLINES_VDIFF / SPLINES_VVISC)
    "plain": None,
    # the other kinds on the open edges, nudging towards uploaded boundary data
    "mixed": dict(zeta=("Che", "Clo", "RadNud", "Clo"), ubar=("Shc", "Clo", "RadNud", "Clo"), vbar=("Shc", "Clo", "Gra", "Clo"),
                  u=("RadNud", "Clo", "Gra", "Clo"), v=("Gra", "Clo", "RadNud", "Clo"), temp=("RadNud", "Clo", "Cla", "Clo"),
                  salt=("Cla", "Clo", "Gra", "Clo")),
    # ... and with nudging towards climatology on: the radiation + nudging edges take their time scales from the coefficient arrays
    # (u3dbc_im.F:160-171, t3dbc_im.F:157-167, u2dbc_im.F:171-183: obc_out the coefficient at the point, obc_in = obcfac * obc_out)
    "mixed_clima": "mixed",
    # all four edges open
    "four": dict(zeta=("Cha", "Rad", "Rad", "Che"), ubar=("Fla", "Rad", "Rad", "Shc"), vbar=("Fla", "Rad", "Rad", "Shc"),
                 u=("Rad", "Rad", "Rad", "Gra"), v=("Rad", "Rad", "Rad", "Gra"), temp=("Rad", "Gra", "Rad", "Rad"),
                 salt=("Gra", "Rad", "Rad", "Rad")),
}


@pytest.mark.parametrize("variant", sorted(OBC_VARIANTS))
def test_open_boundaries_bitwise(emu, variant):
    """Open boundaries (k_obc.h: zetabc, u2dbc, v2dbc, u3dbc, v3dbc, t3dbc with radiation, Chapman, Flather, Shchepetkin,
    clamped, gradient conditions; the oracle is pinned to the reference routines and to whole runs of the reference's
    KELVIN application): the Kelvin wave entering through the western boundary, 12 steps against the oracle, bit for bit;
    the boundary really is open (the wave arrives: |u| grows from rest)."""
    lbc = OBC_VARIANTS[variant]
    lbc = OBC_VARIANTS[lbc] if isinstance(lbc, str) else lbc
    kw = {} if lbc is None else dict(lbc=lbc)
    cs = util.case_for("kelvin_plain_small" if variant == "plain" else "kelvin_small", **kw)
    if variant == "mixed_clima":
        cs["clima"] = 39
    if variant.startswith("mixed"):
        cs.update(Znudg=0.5, M2nudg=0.25, M3nudg=2.0, Tnudg=(1.0, 3.0), obcfac=4.0)
    g = util.load_init("kelvin_small", util.nghost_for(cs))
    O = util.make_oracle(cs, g)
    H = util.make_hip(cs, g, emu)
    if variant not in ("kelvin", "plain"):      # boundary data the analytic KELVIN functions do not provide: uploaded, as a caller's set_data would
        rng = np.random.default_rng(3)
        from roms_amd import hiplib
        for n in hiplib.BRY_FIELDS:
            if n.startswith(("u_", "v_", "t_")) or n.endswith(("south", "north")):
                a = O.field(n)
                a[:] = (10.0 if n[0] == "t" else 0.0) + 0.01 * rng.standard_normal(a.size)
                H.upload(n, a)
    O.start()
    H.start()
    for _ in range(12):
        O.main3d_step()
        H.main3d(1)
        for n in util.PROGNOSTIC:
            a, b = H.download(n), O.field(n)
            assert np.isfinite(b).all(), n
            assert np.array_equal(a, b), (n, float(np.abs(a - b).max()))
    assert np.abs(O.field("u")).max() > 0.05
    H.close()


@pytest.mark.parametrize("variant,vc", [("kelvin", 5), ("plain", 5), ("four", 15), ("four", 10), ("masked", 15)])
def test_volume_conservation_across_open_edges_bitwise(emu, variant, vc):
    """Round 6, VolCons (obc_volcons.F): behind the boundary conditions of every barotropic call the cross-section and the mass flux
    of the conserving edges are summed in the reference's order and give the correction velocity ubar_xs (k_obc.h: k_obc_flux, one
    block, terms in LDS, one thread adds them up in order); the next call takes it off the inflow in its mass fluxes along those
    edges (k_step2d.h: set_DUV_bc_tile, obc_volcons.F:236-370).  The oracle equals the reference with VolCons on -- KELVIN west +
    east, all four edges, 2x2 tiles, under MASKING, kernel by kernel (tests/test_oracle_vs_ref.py).  12 steps against the oracle,
    every bit: the KELVIN application, all four edges open, and a masked basin with open kinds on its four edges; the run differs
    from the one without."""
    from roms_amd import hiplib
    if variant == "masked":
        from tests.refchild import OBC_PRESETS
        cs = util.case_for("upwelling_mask_small")
        cs["lbc"] = OBC_PRESETS["F"]
        cs["EWperiodic"] = 0
        g = util.closed_basin_state(cs, util.with_masks(cs, util.load_init("upwelling_small", util.nghost_for(cs))))
    else:
        kw = {} if OBC_VARIANTS[variant] is None else dict(lbc=OBC_VARIANTS[variant])
        cs = util.case_for("kelvin_plain_small" if variant == "plain" else "kelvin_small", **kw)
        g = util.load_init("kelvin_small", util.nghost_for(cs))
    cs0 = dict(cs)
    cs["volcons"] = vc
    O = util.make_oracle(cs, g)
    O0 = util.make_oracle(cs0, g)
    H = util.make_hip(cs, g, emu)
    if variant == "four":
        rng = np.random.default_rng(3)
        for n in hiplib.BRY_FIELDS:
            if n.startswith(("u_", "v_", "t_")) or n.endswith(("south", "north")):
                a = O.field(n)
                a[:] = (10.0 if n[0] == "t" else 0.0) + 0.01 * rng.standard_normal(a.size)
                O0.field(n)[:] = a
                H.upload(n, a)
    O.start(); O0.start(); H.start()
    rng = np.random.default_rng(11)
    for step in range(12):
        if variant == "masked" and step == 2:     # (the basin at rest: set it in motion)
            for n, amp in (("t", 0.05), ("u", 1e-3), ("v", 1e-3)):
                a = O.field(n).copy()
                a += amp * rng.standard_normal(a.size) * (a != 0.0 if n != "t" else 1.0)
                O.field(n)[:] = a
                O0.field(n)[:] = a
                H.upload(n, a)
        O.main3d_step(); O0.main3d_step(); H.main3d(1)
        for n in util.PROGNOSTIC:
            a, b = H.download(n), O.field(n)
            assert np.isfinite(b).all(), n
            assert np.array_equal(a, b), (step, n, float(np.abs(a - b).max()))
    assert not np.array_equal(O.field("ubar"), O0.field("ubar"))
    H.close()


def test_volume_conservation_on_more_than_one_tile_stops(emu):
    """VolCons on a partition: host and library stop with exit_flag 5 and the reason (the sum over the tiles is a reduction across ranks)"""
    from roms_amd import hiplib, hostlib
    cs = util.case_for("kelvin_small")
    cs["volcons"] = 5
    cs["NtileI"] = 2
    with pytest.raises(hostlib.HostError) as e:
        hostlib.Host(params=cs, lib_path=os.path.join(os.path.dirname(emu), "libroms_host_emu.so"), hip_lib_path=emu)
    assert e.value.exit_flag == 5 and "VolCons on more than one tile" in str(e.value)
    g = util.load_init("kelvin_small", util.nghost_for(cs))
    from tests import cases
    cfg = cases.hip_cfg(cs, float(g["scalars"][0]), int(g["bounds"][58]), g["weight"], g["sc_r"], g["Cs_r"], g["sc_w"], g["Cs_w"])
    cfg.NtileI = 2
    with pytest.raises(hiplib.RomsHipError) as e2:
        hiplib.Context(cfg, emu)
    assert "exit_flag=5" in str(e2.value) and "VolCons on more than one tile" in str(e2.value)


def test_open_boundary_kinds_the_library_does_not_have_stop():
    """roms_hip_create: exit_flag 5 with the reason for a kind that is not built, never a silent closed wall"""
    from roms_amd import hiplib
    cs = util.case_for("kelvin_small")
    g = util.load_init("kelvin_small", util.nghost_for(cs))
    for lbc, needle in ((dict(zeta=("Fla", "Clo", "Rad", "Clo")), "not built"), (dict(u=("Cha", "Clo", "Rad", "Clo")), "not built")):
        cs["lbc"] = lbc
        with pytest.raises(hiplib.RomsHipError) as e:
            util.make_hip(cs, g, util.EMU_LIB)
        assert "exit_flag=5" in str(e.value) and needle in str(e.value), str(e.value)
    cs = util.case_for("kelvin_small", lbc=dict(u=("Rad", "Clo", "Rad", "Clo")))
    cs["hadv"] = cs["vadv"] = ("MPDATA", "MPDATA")
    with pytest.raises(hiplib.RomsHipError) as e:
        util.make_hip(cs, g, util.EMU_LIB)      # (a closed basin: the arrays do not depend on the number of ghost points)
    assert "exit_flag=5" in str(e.value) and "MPDATA" in str(e.value)


def test_isopycnic_mixing_only_in_its_pinned_combination():
    """MIX_ISO_TS (t3dmix2_iso.h) is pinned to the reference through OVERFLOW and, since round 6, with MASKING + WET_DRY
    (oracle/ref/upwelling_wetdry_iso.h) and with the nonlinear equation of state (oracle/ref/benchmark_iso.h): only MIX_GEO_TS
    beside it stops roms_hip_create, exit_flag 5 and the reason."""
    from roms_amd import hiplib
    g = util.load_init("overflow_small", 2)
    for extra, needle in (("MIX_GEO_TS", "exclude"),):
        cs = util.case_for("overflow_small")
        cs["options"] = tuple(cs["options"]) + (extra,)
        gg = util.with_masks(cs, g) if extra == "MASKING" else g
        with pytest.raises(hiplib.RomsHipError) as e:
            util.make_hip(cs, gg, util.EMU_LIB)
        assert "exit_flag=5" in str(e.value) and needle in str(e.value), str(e.value)


@pytest.mark.parametrize("tag", ["seamount_small", "grav_adj_small", "overflow_small"])
def test_more_reference_applications_bitwise(emu, tag):
    """SEAMOUNT (ROMS/Include/seamount.h: no-slip walls GAMMA2 = -1, Akima advection, harmonic mixing along geopotentials
    without KPP, quadratic drag, no vertical mixing closure) and GRAV_ADJ (grav_adj.h: the lock exchange -- MPDATA tracers in
    a closed channel four points wide and periodic across, no rotation, no drag), both pinned bit for bit to the reference:
    12 steps against the oracle, bit for bit; the fronts move."""
    cs = util.case_for(tag)
    g = util.load_init(tag, util.nghost_for(cs))
    O = util.make_oracle(cs, g)
    H = util.make_hip(cs, g, emu)
    O.start()
    H.start()
    for _ in range(12):
        O.main3d_step()
        H.main3d(1)
        for n in util.PROGNOSTIC:
            a, b = H.download(n), O.field(n)
            assert np.isfinite(b).all(), n
            assert np.array_equal(a, b), (n, float(np.abs(a - b).max()))
    assert max(np.abs(O.field("u")).max(), np.abs(O.field("v")).max()) > 1e-4      # (OVERFLOW flows along eta only)
    H.close()


@pytest.mark.parametrize("tag", ["upwelling_prs31_small", "upwelling_wjgradp_small", "upwelling_prs40_small", "upwelling_prs42_small", "upwelling_prs44_small"])
def test_standard_density_jacobian_bitwise(emu, tag):
    """(upwelling_prs42 / _prs44, round 6: PJ_GRADPQ2 / PJ_GRADPQ4, the finite-volume Jacobians with a reconstructed density
    profile, prsgrd42.h / prsgrd44.h -- k_prs4x.h; pinned the same way, oracle/orc_prs4x.c.)
    prsgrd31.h (an application without DJ_GRADPS; WJ_GRADP: the weighted form), k_prs31: 8 steps against the oracle
    (pinned to the reference built from oracle/ref/upwelling_prs31.h / upwelling_wjgradp.h), bit for bit; the result differs
    from the prsgrd32.h run.  upwelling_prs40: PJ_GRADP, the finite-volume scheme prsgrd40.h (k_prs40), pinned the same way."""
    cs = util.case_for(tag)
    g = util.load_init("upwelling_small", util.nghost_for(cs))
    O = util.make_oracle(cs, g)
    H = util.make_hip(cs, g, emu)
    O32 = util.make_oracle(util.case_for("upwelling_small"), g)
    O.start(); H.start(); O32.start()
    for _ in range(8):
        O.main3d_step(); H.main3d(1); O32.main3d_step()
        for n in util.PROGNOSTIC:
            assert np.array_equal(H.download(n), O.field(n)), n
    assert not np.array_equal(O.field("u"), O32.field("u"))
    H.close()


@pytest.mark.parametrize("tag,clima,adv", [("upwelling_small", 7, ("U3", "HSIMT")), ("upwelling_small", 4, ("MPDATA", "MPDATA")),
                                           ("upwelling_small", 32, ("U3", "HSIMT")), ("upwelling_mask_small", 39, ("U3", "U3")),       # bit 5: LnudgeM2CLM, step2d_LF_AM3.h:2179
                                           ("upwelling_mask_small", 3, ("U3", "U3")), ("benchmark_small", 7, None)])
def test_climatology_nudging_bitwise(emu, tag, clima, adv):
    """Nudging towards climatology (LnudgeM3CLM: rhs3d.F:654-680 in k_rhs3d_pt; LtracerCLM + LnudgeTCLM: step3d_t.F:1866-1878,
    k_tnudge between t3dbc and the mask + exchange) -- 10 steps against the oracle (pinned bit for bit to the reference with
    the switches on, tests/test_oracle_vs_ref.py), bit for bit; the result differs from the run without nudging."""
    kw = {} if adv is None else dict(hadv=adv, vadv=tuple("C4" if x == "U3" else x for x in adv))
    cs = util.case_for(tag, **kw)
    cs["clima"] = clima
    g = util.load_init(util.init_tag(cs), util.nghost_for(cs))
    if "MASKING" in cs["options"]:
        g = util.with_masks(cs, g)
    O = util.make_oracle(cs, g)
    H = util.make_hip(cs, g, emu)
    cs0 = dict(cs); cs0.pop("clima")
    O2 = util.make_oracle(cs0, g)
    O.start(); H.start(); O2.start()
    for _ in range(10):
        O.main3d_step(); H.main3d(1); O2.main3d_step()
        for n in util.PROGNOSTIC:
            a, b = H.download(n), O.field(n)
            assert np.array_equal(a, b), (n, int((a != b).sum()), float(np.abs(a - b).max()))
    assert not np.array_equal(O.field("t"), O2.field("t"))
    H.close()


def test_viscosity_along_geopotentials_bitwise(emu):
    """UV_VIS2 + MIX_GEO_UV under MASKING (uv3dmix2_geo.h: the five kernels of k_uvmix_geo.h) -- 10 steps against the oracle
    (pinned bit for bit to the reference built from oracle/ref/upwelling_geouv.h), bit for bit; the result differs from
    the run with the viscosity along s-surfaces."""
    cs = util.case_for("upwelling_geouv_small")
    g = util.with_masks(cs, util.load_init("upwelling_small", util.nghost_for(cs)))
    O = util.make_oracle(cs, g)
    H = util.make_hip(cs, g, emu)
    O2 = util.make_oracle(util.case_for("upwelling_mask_small"), g)
    O.start(); H.start(); O2.start()
    for _ in range(10):
        O.main3d_step(); H.main3d(1); O2.main3d_step()
        for n in util.PROGNOSTIC + ["rufrc", "rvfrc"]:
            a, b = H.download(n), O.field(n)
            assert np.array_equal(a, b), (n, int((a != b).sum()), float(np.abs(a - b).max()))
    assert not np.array_equal(O.field("u"), O2.field("u"))
    H.close()


@pytest.mark.parametrize("closed", [False, True])
def test_biharmonic_viscosity_along_geopotentials_bitwise(emu, closed):
    """Round 6: UV_VIS4 + MIX_GEO_UV under MASKING (uv3dmix4_geo.h:296-1478: the kernels of k_uvmix_geo.h in their modes 1 and 2 and
    k_uvg_lapbc between them) -- 10 steps against the oracle (pinned bit for bit to the reference built from
    oracle/ref/upwelling_bihgeouv.h: channel, four walls, 2x2 tiles, kernel by kernel on perturbed states), bit for bit, in
    the periodic channel and between four walls (the corner averages of LapU, LapV); the result differs from the harmonic run."""
    cs = util.case_for("upwelling_bihgeouv_small")
    g = util.with_masks(cs, util.load_init("upwelling_small", util.nghost_for(cs)))
    csh = util.case_for("upwelling_geouv_small", hadv=cs["hadv"], vadv=cs["vadv"])
    gh = util.with_masks(csh, util.load_init("upwelling_small", util.nghost_for(cs)))
    if closed:
        cs["EWperiodic"] = 0
        g = util.closed_basin_state(cs, g)
    O = util.make_oracle(cs, g)
    H = util.make_hip(cs, g, emu)
    O.start(); H.start()
    rng = np.random.default_rng(7)
    for step in range(10):
        if step == 2:
            for n, amp in (("u", 2e-3), ("v", 2e-3)):
                a = O.field(n).copy()
                a += amp * rng.standard_normal(a.size) * (a != 0.0)
                O.field(n)[:] = a
                H.upload(n, a)
        O.main3d_step(); H.main3d(1)
        for n in util.PROGNOSTIC + ["rufrc", "rvfrc"]:
            a, b = H.download(n), O.field(n)
            assert np.array_equal(a, b), (step, n, int((a != b).sum()), float(np.abs(a - b).max()))
    assert np.isfinite(O.field("u")).all() and np.abs(O.field("u")).max() > 1e-4
    H.close()


@pytest.mark.parametrize("tag", ["upwelling_bih_small", "upwelling_bihgeo_small", "upwelling_bihiso_small"])
def test_biharmonic_mixing_bitwise(emu, tag):
    """UV_VIS4 + TS_DIF4 along s-surfaces (uv3dmix4_s.h, t3dmix4_s.h, the UV_VIS4 block of step2d_LF_AM3.h): k_uv4_lap +
    k_uv3dmix4_s, k_t3dmix4, k_step2d_vis4 -- 10 steps against the oracle (pinned bit for bit to the reference built from
    oracle/ref/upwelling_bih.h), bit for bit; the result differs from the harmonic run.  _bihgeo: the tracers along
    geopotential surfaces (t3dmix4_geo.h: the marching kernel k_t3dmix2_geo in its modes 2 and 3; upwelling_bihgeo.h);
    _bihiso: along isopycnic surfaces (t3dmix4_iso.h: k_t3dmix2_iso in the same modes; upwelling_bihiso.h)."""
    cs = util.case_for(tag)
    itag = "upwelling_small"
    g = util.load_init(itag, util.nghost_for(cs))
    O = util.make_oracle(cs, g)
    H = util.make_hip(cs, g, emu)
    O2 = util.make_oracle(util.case_for(itag), g)
    O.start(); H.start(); O2.start()
    for _ in range(10):
        O.main3d_step(); H.main3d(1); O2.main3d_step()
        for n in util.PROGNOSTIC:
            a, b = H.download(n), O.field(n)
            assert np.array_equal(a, b), (n, int((a != b).sum()), float(np.abs(a - b).max()))
    assert not np.array_equal(O.field("u"), O2.field("u")) and not np.array_equal(O.field("t"), O2.field("t"))
    H.close()


GLS_TAGS = ["upwelling_gls_small", "upwelling_gls_small:k-omega", "upwelling_gls_ca_small:gen", "upwelling_gls_cb_small:k-kl",
            "upwelling_gls_gal_small:k-omega",
            # the Mellor-Yamada 2.5 closure (my25_corstep.F on the same kernels): upwelling.h -DMY25_MIXING, and Galperin / K_C4ADVECTION
            "upwelling_my25_small", "upwelling_my25_gal_small"]


@pytest.mark.parametrize("tag", GLS_TAGS)
def test_generic_length_scale_closure_bitwise(emu, tag):
    """GLS_MIXING (k_gls.h: gls_prestep.F, gls_corstep.F, tkebc_im.F) in its five pinned forms -- Kantha-Clayson (upwelling.h
    with -DGLS_MIXING; k-epsilon and k-omega parameters), Canuto A under MASKING ("gen"), Canuto B with K_C2ADVECTION,
    CHARNOK and CRAIG_BANNER (k-kl: the wall function), Galperin with K_C4ADVECTION: 12 steps against the oracle, every
    array bit for bit (the emulated build calls the same libm `pow` as the oracle); the closure is active (Akv leaves
    its background value)."""
    cs = util.case_for(tag)
    g = util.load_init("upwelling_small", util.nghost_for(cs))
    if "MASKING" in cs["options"]:
        g = util.with_masks(cs, g)
    g = util.with_gls(cs, g)
    O = util.make_oracle(cs, g)
    H = util.make_hip(cs, g, emu)
    O.start(); H.start()
    for _ in range(12):
        O.main3d_step(); H.main3d(1)
        for n in util.PROGNOSTIC:
            a, b = H.download(n), O.field(n)
            assert np.isfinite(b).all(), n
            assert np.array_equal(a, b), (n, int(np.count_nonzero(a != b)), float(np.abs(a - b).max()))
    assert O.field("Akv").max() > 1.05 * cs["Akv_bak"]
    H.close()


@pytest.mark.parametrize("tag", GLS_TAGS)
def test_generic_length_scale_closure_through_the_fortran_host(emu, tag, monkeypatch):
    """roms.in (the GLS_* block, AKK_BAK, AKP_BAK, CHARNOK_ALPHA, CRGBAN_CW, LBC(isMtke)) and the application header ->
    Fortran host (option surface, initialize_mixing's values for tke, gls, Akk, Akp, Lscale) -> C ABI -> kernels: the
    oracle's bits after 8 steps.  The shipped upwelling.h form is selected as a user of the reference does, by -DGLS_MIXING
    on the cpp command line (ROMS_CPP_FLAGS, with the built-in UPWELLING list); the others by their headers under
    oracle/ref/; the kernel-by-kernel sequence (main3d_kernels: gls_prestep behind rhs3d, gls_corstep behind omega) gives
    the same bits as the fused entry."""
    from roms_amd import hostlib
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    cs = util.case_for(tag)
    g = util.load_init("upwelling_small", util.nghost_for(cs))
    if "MASKING" in cs["options"]:
        g = util.with_masks(cs, g)
    O = util.make_oracle(cs, util.with_gls(cs, g))
    O.start()
    O.main3d_step(8)
    # three ways to say which form: the built-in list of the application name, the header, -DGLS_MIXING beside the shipped one
    ways = [dict(params=dict(cs, ninfo=0)), dict(params=dict(cs, ninfo=0), kernels=True)]
    if cs["app"] in ("upwelling_gls", "upwelling_my25"):
        ways.append(dict(params=dict(cs, ninfo=0, app="upwelling"), flags="-DGLS_MIXING" if cs["app"] == "upwelling_gls" else "-DMY25_MIXING"))
    else:
        ways.append(dict(params=dict(cs, ninfo=0, app="upwelling"), header=os.path.join(root, "oracle", "ref", cs["app"] + ".h")))
    for w in ways:
        if "flags" in w:
            monkeypatch.setenv("ROMS_CPP_FLAGS", w["flags"])
        H = hostlib.Host(params=w["params"], lib_path=os.path.join(os.path.dirname(emu), "libroms_host_emu.so"),
                         hip_lib_path=emu, header=w.get("header"))
        ctx = H.device_init()
        H.run(8, kernels=w.get("kernels", False))
        for n in ("zeta", "u", "v", "t", "Akv", "Akt", "tke", "gls", "Lscale", "Akk", "Akp"):
            assert np.array_equal(ctx.download(n), O.field(n)), (n, w.keys())
        H.finalize()


@pytest.mark.parametrize("lbc_tke", [None, ("Gra", "Clo", "Rad", "Clo"), ("Rad", "Clo", "Gra", "Clo")])
def test_generic_length_scale_closure_with_open_boundaries(emu, lbc_tke):
    """KELVIN (Chapman / Flather / radiation edges, k_obc.h) with GLS_MIXING: 20 steps, every array the oracle's bits; the
    bottom stress of the Kelvin wave drives the closure hard (Akv four orders above its background).  LBC(isMtke): the default
    (closed where not periodic), and tkebc_im.F's radiation condition on the eastern | western edge with the gradient
    condition opposite (the oracle pinned to the reference with the same kinds: tests/test_oracle_vs_ref.py)."""
    cs, g = util.kelvin_gls_case()
    if lbc_tke:
        cs["lbc_tke"] = lbc_tke
    O = util.make_oracle(cs, g)
    H = util.make_hip(cs, g, emu)
    O.start(); H.start()
    for _ in range(20):
        O.main3d_step(); H.main3d(1)
        for n in util.PROGNOSTIC:
            assert np.array_equal(H.download(n), O.field(n)), n
    assert O.field("Akv").max() > 1e4 * cs["Akv_bak"]
    H.close()


def _determinism_cases():
    from tests.test_gpu_parity import DETERMINISM_SMALL
    return DETERMINISM_SMALL


@pytest.mark.parametrize("tag,kw", _determinism_cases())
def test_launch_order_and_poisoned_scratch_change_nothing(emu, tag, kw, monkeypatch):
    """Two aids of the emulated build against the defect class a serial emulation hides (round 4): ROMS_EMU_ORDER=reverse
    runs the blocks / threads of every launch in the opposite order -- a launch whose threads exchange values through
    global memory depends on the order; ROMS_HIP_POISON=1 fills the work arrays with NaN at create and at every step --
    a read of scratch nobody wrote shows.  Both must leave every bit of every application case where it was."""
    from tests.test_gpu_parity import _case_state, _end_state
    cs, g = _case_state(tag, kw)
    a = _end_state(cs, g, 4, emu)
    for env in (dict(ROMS_EMU_ORDER="reverse"), dict(ROMS_HIP_POISON="1"), dict(ROMS_EMU_ORDER="reverse", ROMS_HIP_POISON="1")):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        b = _end_state(cs, g, 4, emu)
        for k in env:
            monkeypatch.delenv(k)
        for n in a:
            assert np.array_equal(a[n], b[n], equal_nan=True), (tag, env, n)


@pytest.mark.parametrize("tag,kw,nDIA,ntsDIA", [
    ("upwelling_small", dict(hadv=("U3", "HSIMT"), vadv=("C4", "HSIMT")), 3, 1),           # roms_upwelling.in's schemes
    ("upwelling_small", dict(hadv=("A4", "C2"), vadv=("SPLINES", "C2")), 2, 2),
    ("benchmark_small", {}, 2, 1),                                                           # geopotential mixing: the S-diffusion term
    ("overflow_small", {}, 1, 1),                                                            # isopycnic mixing, both tracers on splines
    ("upwelling_mask_small", {}, 3, 1)])
def test_tracer_diagnostics_bitwise(emu, tag, kw, nDIA, ntsDIA):
    """DIAGNOSTICS_TS on the emulated kernels: DiaTwrk, DiaTrc and avgzeta after every step of several windows against the
    oracle (pinned to the reference built from upwelling.h as shipped: tests/test_oracle_vs_ref.py::test_set_diags_bitwise),
    bit for bit; the prognostic fields unchanged by the switch."""
    from tests.test_gpu_parity import _case_state
    cs, g = _case_state(tag, kw)
    O = util.make_oracle(cs, g)
    H = util.make_hip(cs, g, emu)
    O.set_dia_window(nDIA, ntsDIA)
    H.dia_config(nDIA, ntsDIA)
    O.start()
    H.start()
    seen = 0
    for step in range(1, 8):
        O.main3d_step()
        H.main3d(1)
        for n in ("DiaTwrk", "DiaTrc", "dia_zeta", "t", "u", "zeta"):
            a, b = H.download(n), O.field(n)
            assert np.array_equal(a, b), (tag, step, n, int((a != b).sum()))
        seen += int(np.abs(O.field("DiaTrc")).max() > 0.0)
    assert seen >= 4
    H.close()


DIAUV_FIELDS = ["DiaRU", "DiaRV", "DiaRUfrc", "DiaRVfrc", "DiaU3wrk", "DiaV3wrk", "DiaU2wrk", "DiaV2wrk", "DiaU2int", "DiaV2int",
                "DiaRUbar", "DiaRVbar", "DiaU2d", "DiaV2d", "DiaU3d", "DiaV3d"]


@pytest.mark.parametrize("tag,kw,nDIA,ntsDIA", [
    ("upwelling_small", dict(hadv=("U3", "HSIMT"), vadv=("C4", "HSIMT")), 3, 1),           # the option set pinned to the reference
    ("upwelling_small", dict(hadv=("U3", "U3"), vadv=("C4", "C4")), 2, 2),
    ("benchmark_small", {}, 2, 1),                                                           # curvilinear terms, quadratic drag, KPP
    ("upwelling_mask_small", {}, 3, 1)])                                                     # land/sea masks
def test_momentum_diagnostics_bitwise(emu, tag, kw, nDIA, ntsDIA):
    """DIAGNOSTICS_UV on the emulated kernels: the sixteen arrays of mod_diags.F (the two levels of every 3-D right-hand-side
    term, their vertical sums, the fast-time integrals of the 2-D terms, DiaU3wrk / DiaU2wrk and the accumulated output)
    after every step of several windows against the oracle (pinned to the reference built from upwelling.h as shipped:
    tests/test_oracle_vs_ref.py::test_set_diags_uv_bitwise), bit for bit; the prognostic fields unchanged by the switch."""
    from tests.test_gpu_parity import _case_state
    cs, g = _case_state(tag, kw)
    O = util.make_oracle(cs, g)
    H = util.make_hip(dict(cs, dia_uv=True), g, emu)
    O.set_dia_window(nDIA, ntsDIA, uv=True)
    H.dia_config(nDIA, ntsDIA, uv=True)
    O.start()
    H.start()
    seen = 0
    for step in range(1, 8):
        O.main3d_step()
        H.main3d(1)
        for n in DIAUV_FIELDS + ["DiaTwrk", "DiaTrc", "t", "u", "v", "ubar", "zeta"]:
            a, b = H.download(n), O.field(n)
            assert np.array_equal(a, b), (tag, step, n, int((a != b).sum()), float(np.abs(a - b).max()))
        seen += int(np.abs(O.field("DiaU3d")).max() > 0.0)
    assert seen >= 4
    H.close()


@pytest.mark.parametrize("tag", ["benchmark_small", "benchmark_mask_small", "upwelling_kpp_small"])
def test_kpp_block_form_bitwise(emu, tag):
    """k_lmd_blk (ROMS_HIP_LMDCOL=3: 64 columns per block, the sweeps without a recurrence on (column, level) pairs, the
    three splines side by side; the default of small grids on the device) against the oracle over 6 steps, every bit; in a
    child process -- the library reads the switch once."""
    import textwrap
    code = textwrap.dedent("""
        import sys
        import numpy as np
        sys.path.insert(0, %r)
        from tests import util
        tag = %r
        cs = util.case_for(tag)
        g = util.load_init(util.init_tag(cs), util.nghost_for(cs))
        if "MASKING" in cs["options"]:
            g = util.with_masks(cs, g)
        O = util.make_oracle(cs, g)
        H = util.make_hip(cs, g, %r)
        O.start(); H.start()
        for _ in range(6):
            O.main3d_step(); H.main3d(1)
            for n in util.PROGNOSTIC + ["Akv", "Akt", "hsbl", "ghats"]:
                a, b = H.download(n), O.field(n)
                assert np.array_equal(a, b), (n, int((a != b).sum()), float(np.abs(a - b).max()))
        H.close()
        print("BLK-OK")
    """) % (os.path.join(os.path.dirname(__file__), ".."), tag, emu)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, ROMS_HIP_LMDCOL="3"), timeout=600)
    assert "BLK-OK" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


@pytest.mark.parametrize("tag", ["benchmark_bkpp_small", "upwelling_kpp_bkpp_small"])
def test_bottom_boundary_layer_of_the_k_profile_scheme_bitwise(emu, tag):
    """Round 6, LMD_BKPP (lmd_bkpp.F:95-806 with RI_SPLINES and the file's own SASHA; k_lmd.h: k_lmd_bkpp behind k_lmd_interior +
    k_lmd_skpp, lmd_finish as its last operation): the oracle equals the reference built with -DLMD_BKPP (benchmark.h: nonlinear EOS,
    bulk fluxes, shortwave; upwelling_kpp.h: linear EOS) from rest and with random velocities added -- a layer 40 to 190 m thick --
    on one tile and 2x2.  12 steps against the oracle, every bit, hbbl included; from step 3 on (random velocities of 0.3 m/s) the
    layer reaches several levels and the mixing coefficients differ from the run without the option."""
    cs = util.case_for(tag)
    g = util.load_init(util.init_tag(cs), util.nghost_for(cs))
    cs0 = dict(cs); cs0.pop("bkpp")
    O = util.make_oracle(cs, g); O0 = util.make_oracle(cs0, g)
    H = util.make_hip(cs, g, emu)
    O.start(); O0.start(); H.start()
    rng = np.random.default_rng(9)
    for step in range(12):
        if step == 2:
            for n in ("u", "v"):
                a = O.field(n).copy()
                a += 0.3 * rng.standard_normal(a.size)
                O.field(n)[:] = a; O0.field(n)[:] = a
                H.upload(n, a)
        O.main3d_step(); O0.main3d_step(); H.main3d(1)
        for n in util.PROGNOSTIC + ["hbbl"]:
            a, b = H.download(n), O.field(n)
            assert np.isfinite(b).all(), (step, n)
            assert np.array_equal(a, b), (step, n, float(np.abs(a - b).max()))
    assert float((O.field("hbbl") + np.asarray(g["h"]).ravel()).max()) > 30.0
    assert not np.array_equal(O.field("Akv"), O0.field("Akv"))
    H.close()


def test_bottom_boundary_layer_only_in_its_pinned_combinations():
    """roms_hip_create: LMD_BKPP with MASKING (or WET_DRY, LMD_DDMIX) is refused with the reason -- no reference build pins it"""
    from roms_amd import hiplib
    cs = util.case_for("upwelling_mask_small")
    cs["options"] = tuple(cs["options"]) + ("LMD_MIXING",) if "LMD_MIXING" not in cs["options"] else cs["options"]
    cs["bkpp"] = 1
    g = util.with_masks(cs, util.load_init("upwelling_small", util.nghost_for(cs)))
    with pytest.raises(hiplib.RomsHipError) as e:
        util.make_hip(cs, g, util.EMU_LIB)
    assert "exit_flag=5" in str(e.value) and "LMD_BKPP" in str(e.value), str(e.value)


@pytest.mark.parametrize("tag", ["benchmark_ddmix_small", "upwelling_kpp_ddmix_small"])
@pytest.mark.parametrize("form", ["0", "1", "2", "3"])
def test_double_diffusive_mixing_bitwise(emu, tag, form):
    """LMD_DDMIX (round 6): lmd_vmix.F:360-428 in every form of the KPP kernels (ROMS_HIP_LMDCOL = 0: two kernels, 1: the column
    kernel, 2: one thread kernel, 3: the block kernel) with alfaobeta of both equations of state (k_eos_alfaobeta) on the state of
    cases.ddmix_state -- salt fingering capped and not, diffusive convection with Rrho on both sides of 0.5 -- against the oracle
    (pinned bit for bit to the reference built with the option: tests/test_oracle_vs_ref.py) over 6 steps, every bit; the
    mixing coefficients differ from the run without the option.  In a child process: the library reads the switch once."""
    import textwrap
    code = textwrap.dedent("""
        import sys
        import numpy as np
        sys.path.insert(0, %r)
        from tests import util
        tag = %r
        cs = util.case_for(tag)
        g = util.with_ddmix_state(cs, util.load_init(util.init_tag(cs), util.nghost_for(cs)))
        O = util.make_oracle(cs, g)
        cs0 = dict(cs); cs0.pop("ddmix")
        O0 = util.make_oracle(cs0, g)
        H = util.make_hip(cs, g, %r)
        O.start(); H.start(); O0.start()
        for _ in range(6):
            O.main3d_step(); H.main3d(1); O0.main3d_step()
            for n in util.PROGNOSTIC + ["Akv", "Akt", "hsbl", "ghats"]:
                a, b = H.download(n), O.field(n)
                assert np.array_equal(a, b), (n, int((a != b).sum()), float(np.abs(a - b).max()))
        assert (O.field("Akt") != O0.field("Akt")).sum() > 100
        H.close()
        print("DDMIX-OK")
    """) % (os.path.join(os.path.dirname(__file__), ".."), tag, emu)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, ROMS_HIP_LMDCOL=form), timeout=600)
    assert "DDMIX-OK" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


@pytest.mark.parametrize("tag", ["upwelling_small", "upwelling_bih_small", "upwelling_bihgeo_small", "upwelling_bihiso_small"])
def test_closed_basin_with_biharmonic_mixing_bitwise(emu, tag):
    """Round 6: four walls.  The set-up arrays of the periodic channel cut to the closed basin's arrays (util.closed_basin_state:
    the reference's ana_grid.h gives UPWELLING no bathymetry without a periodic direction) -- whole steps against the oracle, which
    is pinned bit for bit to the reference on exactly this state (tests/test_oracle_vs_ref.py: *_closed_small, whole steps and
    kernel by kernel on perturbed states): the corner values of every boundary routine, the barotropic engines with the corner
    averages fused into their stores (k_haloblock.h: HB_CORNERS), and the conditions on the first biharmonic operator at the
    western / eastern walls and the corners -- t3dmix4_s.h, t3dmix4_geo.h:475-600, t3dmix4_iso.h:504-618 (k_bench.h:
    T3D4_WE_WALLS; refused until round 6) -- with a tenth of the channel cases' VISC4 / TNU4 (those blow up between four
    walls, in the reference too)."""
    cs = util.case_for(tag)
    cs["EWperiodic"] = 0
    if "mix4" in cs:
        cs["visc4"], cs["tnu4"] = 4.0e7, (2.0e6, 1.0e6)
    g = util.closed_basin_state(cs, util.load_init("upwelling_small", util.nghost_for(cs)))
    O = util.make_oracle(cs, g)
    H = util.make_hip(cs, g, emu)
    O.start(); H.start()
    rng = np.random.default_rng(5)
    for step in range(8):
        if step == 2:        # a perturbation that reaches the walls and the corners (the state at rest is uniform along xi)
            for n, amp in (("t", 0.05), ("u", 1e-3), ("v", 1e-3)):
                a = O.field(n).copy()
                a += amp * rng.standard_normal(a.size) * (a != 0.0 if n != "t" else 1.0)
                O.field(n)[:] = a
                H.upload(n, a)
        O.main3d_step(); H.main3d(1)
        for n in util.PROGNOSTIC:
            assert np.array_equal(H.download(n), O.field(n)), (step, n)
    assert np.isfinite(O.diag()[0])
    H.close()


@pytest.mark.parametrize("variant", ["gls", "my25", "geouv", "prs31", "prs44", "iso"])
def test_wet_dry_with_closures_geopotential_viscosity_and_other_jacobians_bitwise(emu, variant):
    """Round 6: WET_DRY together with GLS_MIXING / MY25_MIXING (their routines carry no WET_DRY statement; the closure sees the
    masked, limited state), MIX_GEO_UV (uv3dmix2_geo.h's wet masks) and the Jacobians prsgrd31.h / prsgrd44.h (ru, rv times the wet
    masks) -- refused until now as unpinned; the oracle equals the reference built from oracle/ref/upwelling_wetdry_<variant>.h over
    40 steps.  20 steps against the oracle, every bit; the shore line moves."""
    cs = util.case_for("upwelling_wetdry_%s_small" % variant)
    g = util.with_wetdry(cs, util.load_init("upwelling_small", util.nghost_for(cs)))
    if "gls_flags" in cs:
        g = util.with_gls(cs, g)
    O = util.make_oracle(cs, g)
    H = util.make_hip(cs, g, emu)
    O.start(); H.start()
    wet0 = O.field("rmask_wet").copy()
    moved = False
    for step in range(20):
        O.main3d_step(); H.main3d(1)
        for n in util.PROGNOSTIC + ["rmask_wet", "umask_wet", "vmask_wet"] + (["tke", "gls", "Akv"] if "gls_flags" in cs else []):
            a, b = H.download(n), O.field(n)
            assert np.array_equal(a, b), (step, n, int((a != b).sum()), float(np.abs(a - b).max()))
        moved = moved or not np.array_equal(wet0, O.field("rmask_wet"))
    assert moved
    H.close()


@pytest.mark.parametrize("tag,clima", [("upwelling_wetdry_small", 7), ("upwelling_bih_small", 7), ("upwelling_bihgeo_small", 7),
                                       ("upwelling_geouv_small", 39), ("upwelling_bihgeouv_small", 7)])
def test_climatology_nudging_with_wetting_biharmonic_and_geopotential_options_bitwise(emu, tag, clima):
    """Round 6: nudging of the 3-D momentum and the tracers towards climatology (rhs3d.F:654-680, step3d_t.F:1866-1878) together with
    WET_DRY, UV_VIS4 / TS_DIF4 and MIX_GEO_UV (with the harmonic form also LnudgeM2CLM: clima 39) -- refused until now as unpinned;
    the oracle equals the reference with the switches on over 30 steps (tests/test_oracle_vs_ref.py).  15 steps, every bit."""
    cs = util.case_for(tag)
    cs["clima"] = clima
    if "mix4" in cs:
        cs["visc4"], cs["tnu4"] = min(cs["visc4"], 4.0e7), tuple(min(x, y) for x, y in zip(cs["tnu4"], (2.0e6, 1.0e6)))
    g = util.load_init("upwelling_small", util.nghost_for(cs))
    if cs.get("wet_dry"):
        g = util.with_wetdry(cs, g)
    elif "MASKING" in cs["options"]:
        g = util.with_masks(cs, g)
    O = util.make_oracle(cs, g)
    H = util.make_hip(cs, g, emu)
    O.start(); H.start()
    for step in range(15):
        O.main3d_step(); H.main3d(1)
        for n in util.PROGNOSTIC:
            a, b = H.download(n), O.field(n)
            assert np.array_equal(a, b), (step, n, int((a != b).sum()), float(np.abs(a - b).max()))
    assert np.abs(O.field("u")).max() > 1e-3
    H.close()


def test_double_diffusive_mixing_with_wet_dry_bitwise(emu):
    """LMD_DDMIX under WET_DRY (BENCHMARK with MASKING + WET_DRY, bulk fluxes, KPP: oracle/ref/benchmark_wetdry.h -DLMD_DDMIX pins the
    oracle over 40 steps): 10 steps against the oracle on the state of cases.ddmix_state, every bit."""
    cs = util.case_for("benchmark_wetdry_small")
    cs["ddmix"] = 1
    g = util.with_ddmix_state(cs, util.with_wetdry(cs, util.load_init(util.init_tag(cs), util.nghost_for(cs))))
    O = util.make_oracle(cs, g)
    H = util.make_hip(cs, g, emu)
    O.start(); H.start()
    for step in range(10):
        O.main3d_step(); H.main3d(1)
        for n in util.PROGNOSTIC + ["Akt", "Akv", "rmask_wet"]:
            a, b = H.download(n), O.field(n)
            assert np.array_equal(a, b), (step, n, int((a != b).sum()), float(np.abs(a - b).max()))
    H.close()


@pytest.mark.parametrize("tag", ["kelvin_geouv_small", "benchmark_iso_small"])
def test_more_pinned_combinations_bitwise(emu, tag):
    """Round 6: MIX_GEO_UV beside open boundaries (KELVIN with the viscosity along geopotentials, VISC2 = 50: oracle/ref/kelvin_geouv.h)
    and MIX_ISO_TS with the nonlinear equation of state (BENCHMARK with its tracer mixing along isopycnals: oracle/ref/benchmark_iso.h)
    -- refused until now as unpinned; the oracle equals the reference over 40 steps.  20 steps against the oracle, every bit."""
    cs = util.case_for(tag)
    g = util.load_init(util.init_tag(cs), util.nghost_for(cs))
    O = util.make_oracle(cs, g)
    H = util.make_hip(cs, g, emu)
    O.start(); H.start()
    for step in range(20):
        O.main3d_step(); H.main3d(1)
        for n in util.PROGNOSTIC:
            a, b = H.download(n), O.field(n)
            assert np.array_equal(a, b), (step, n, int((a != b).sum()), float(np.abs(a - b).max()))
    H.close()
