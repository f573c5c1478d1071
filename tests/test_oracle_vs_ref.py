"""Oracle against the reference's OWN Fortran (oracle/_ref, built from /root/reference by
oracle/ref/build_ref.sh).  Only runs where that build exists (the build container); skipped
elsewhere.  One configuration per process, so each case runs in a subprocess."""
import os
import subprocess
import sys

import pytest

from oracle import ref

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
pytestmark = pytest.mark.ref

SCRIPT = r'''
import sys, os, numpy as np
sys.path.insert(0, %(root)r)
from oracle import orc, ref
from tests import cases, util
cs = cases.upwelling(Lm=14, Mm=18, N=8, hadv=%(hadv)r, vadv=%(vadv)r)
ip, rp = cases.ref_params(cs)
R = ref.Ref("upwelling", ip, rp); R.initial()
b = R.bounds(0)
w = np.stack([R.table(5, 60), R.table(6, 60)])
O = orc.Oracle(cases.oracle_cfg(cs, R.table(7, 8)[0], b[58], w))
assert b[:54] == O.bounds(0)
for n in util.INIT_FIELDS:
    if R.has(n): O.field(n)[:] = R.get(n)
for k, n in enumerate(["sc_r", "Cs_r", "sc_w", "Cs_w"]): O.field(n)[:] = R.table(k + 1, O.field(n).size)
st = O.step
st.iic = 4; st.iif = 1; st.nstp = 2; st.nnew = 1; st.nrhs = 2; st.kstp = 1; st.knew = 1; st.krhs = 1
st.predictor = 0; st.time = 900.0; st.tdays = 900.0 / 86400.0
R.set_stepping(st.iic, st.iif, st.nstp, st.nnew, st.nrhs, st.kstp, st.knew, st.krhs, st.predictor, st.time)
rng = np.random.default_rng(7)
for n, amp in [("u", 0.05), ("v", 0.05), ("zeta", 0.1), ("t", 0.01), ("ubar", 0.02), ("vbar", 0.02)]:
    a = O.field(n); a[:] += amp * rng.standard_normal(a.size)
O.field("Zt_avg1")[:] = O.field("zeta")[:O.ni * O.nj]
for n in ["u", "v", "zeta", "t", "Zt_avg1", "ubar", "vbar"]: R.put(n, O.field(n))
seq = [("set_depth", ["Hz", "z_r", "z_w"]), ("set_massflux", ["Huon", "Hvom"]),
       ("rho_eos", ["rho", "pden", "rhoA", "rhoS"]), ("set_vbc", ["bustr", "bvstr", "stflx", "btflx"]),
       ("ana_vmix", ["Akv", "Akt"]), ("ana_smflux", ["sustr", "svstr"]), ("prsgrd", ["ru", "rv"]),
       ("t3dmix2", ["t"]), ("uv3dmix2", ["u", "v", "rufrc", "rvfrc"]), ("set_zeta", ["zeta"]),
       ("wvelocity", ["wvel"]), ("ini_zeta", ["zeta", "Zt_avg1"]), ("ini_fields", ["u", "v", "ubar", "vbar", "t"])]
for k, fl in seq:
    R.call(k)
    if k == "wvelocity": O.call(k, None, st.nstp)
    elif k == "ana_smflux": O.call("set_data")
    else: O.call(k)
    for n in fl:
        assert np.array_equal(R.get(n), O.field(n)), (k, n)
print("PINNED-OK")
'''


@pytest.mark.parametrize("hadv,vadv", [(("U3", "HSIMT"), ("C4", "HSIMT")), (("U3", "U3"), ("C4", "C4"))])
def test_pinned_kernels_bitwise(hadv, vadv):
    if not ref.available("upwelling"):
        pytest.skip("reference build oracle/_ref not available here")
    code = SCRIPT % dict(root=ROOT, hadv=hadv, vadv=vadv)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert "PINNED-OK" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]


MPDATA_SCRIPT = r'''
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, %(root)r)
from oracle import orc, ref
from tests import cases, util
cs = cases.upwelling(Lm=14, Mm=18, N=8, hadv=("MPDATA", "MPDATA"), vadv=("MPDATA", "MPDATA"))
ip, rp = cases.ref_params(cs)
R = ref.Ref("upwelling", ip, rp); R.initial()
b = R.bounds(0)
w = np.stack([R.table(5, 60), R.table(6, 60)])
O = orc.Oracle(cases.oracle_cfg(cs, R.table(7, 8)[0], b[58], w))
for n in util.INIT_FIELDS:
    if R.has(n): O.field(n)[:] = R.get(n)
for k, n in enumerate(["sc_r", "Cs_r", "sc_w", "Cs_w"]): O.field(n)[:] = R.table(k + 1, O.field(n).size)
rng = np.random.default_rng(3)
for n, amp in [("u", 0.05), ("v", 0.05), ("t", 0.3), ("W", 5.0), ("Huon", 50.0), ("Hvom", 50.0)]:
    a = O.field(n); a[:] += amp * rng.standard_normal(a.size); R.put(n, a)
LBi, UBi, LBj, UBj = b[0], b[1], b[2], b[3]
ni, nj, N = UBi - LBi + 1, UBj - LBj + 1, cs["N"]
Istr, Iend, Jstr, Jend = 1, cs["Lm"], 1, cs["Mm"]
IminS, ImaxS, JminS, JmaxS = Istr - 3, Iend + 3, Jstr - 3, Jend + 3
nis, njs = ImaxS - IminS + 1, JmaxS - JminS + 1
for itrc in (1, 2):
    # Ta: positive field on the global arrays
    Tg = (10.0 + 3.0 * rng.random((N, nj, ni)) if itrc == 1 else 35.0 + 0.5 * rng.random((N, nj, ni)))
    Tg[:, :, 5:9] = Tg[:, :, 5:6]      # some equal neighbours -> the eps2 branch
    Ts = np.zeros((N, njs, nis))
    j0, j1 = max(JminS, LBj), min(JmaxS, UBj); i0, i1 = max(IminS, LBi), min(ImaxS, UBi)
    Ts[:, j0 - JminS:j1 - JminS + 1, i0 - IminS:i1 - IminS + 1] = Tg[:, j0 - LBj:j1 - LBj + 1, i0 - LBi:i1 - LBi + 1]
    Ua = np.zeros(N * njs * nis); Va = np.zeros_like(Ua); Wa = np.zeros((N + 1) * njs * nis)
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    R.L.ref_mpdata_adiff(C.c_int(itrc), P(np.ascontiguousarray(Ts)), P(Ua), P(Va), P(Wa))
    # oracle
    oHz = np.zeros((N, nj, ni)); Hz = O.arr("Hz")
    oHz[:] = 1.0 / np.where(Hz != 0, Hz, 1.0)
    Uo = np.zeros(N * nj * ni); Vo = np.zeros_like(Uo); Wo = np.zeros((N + 1) * nj * ni)
    Tgc = np.ascontiguousarray(Tg)
    O.L.orc_mpdata_adiff(C.c_void_p(O.h), C.c_int(0), C.c_int(itrc), P(Tgc), P(Uo), P(Vo), P(Wo), P(np.ascontiguousarray(oHz)))
    def crop(a, nk):
        return a.reshape(nk, njs, nis)[:, j0 - JminS:j1 - JminS + 1, i0 - IminS:i1 - IminS + 1]
    def cropg(a, nk):
        return a.reshape(nk, nj, ni)[:, j0 - LBj:j1 - LBj + 1, i0 - LBi:i1 - LBi + 1]
    for nm, r_, o_, nk in [("Ua", Ua, Uo, N), ("Va", Va, Vo, N), ("Wa", Wa, Wo, N + 1)]:
        a, bb = crop(r_, nk), cropg(o_, nk)
        d = np.argwhere(a != bb)
        assert np.count_nonzero(a) > 500 and len(d) == 0, (itrc, nm, len(d))
print("MPDATA-PINNED-OK")
'''


def test_mpdata_adiff_bitwise():
    """mpdata_adiff_tile (mpdata_adiff.F:38): anti-diffusive velocities Ua, Va, Wa with the FCT limiter,
    both tracers, random positive fields incl. equal neighbours -- oracle vs the reference's object code."""
    p = subprocess.run([sys.executable, "-c", MPDATA_SCRIPT % dict(root=ROOT)], capture_output=True, text=True,
                       timeout=600)
    assert "MPDATA-PINNED-OK" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]


KPP_SCRIPT = r'''
import sys, os, numpy as np
sys.path.insert(0, %(root)r)
from oracle import orc, ref
from tests import cases, util
cs = cases.upwelling_kpp(Lm=14, Mm=18, N=8)
ip, rp = cases.ref_params(cs)
R = ref.Ref("upwelling_kpp", ip, rp); R.initial()
b = R.bounds(0)
w = np.stack([R.table(5, 60), R.table(6, 60)])
O = orc.Oracle(cases.oracle_cfg(cs, R.table(7, 8)[0], b[58], w))
for n in util.INIT_FIELDS:
    if R.has(n): O.field(n)[:] = R.get(n)
for k, n in enumerate(["sc_r", "Cs_r", "sc_w", "Cs_w"]): O.field(n)[:] = R.table(k + 1, O.field(n).size)
st = O.step
st.iic = 4; st.iif = 1; st.nstp = 2; st.nnew = 1; st.nrhs = 2; st.kstp = 1; st.knew = 1; st.krhs = 1
st.predictor = 0; st.time = 900.0; st.tdays = 900.0 / 86400.0
R.set_stepping(st.iic, st.iif, st.nstp, st.nnew, st.nrhs, st.kstp, st.knew, st.krhs, st.predictor, st.time)
rng = np.random.default_rng(11)
for n, amp in [("u", 0.05), ("v", 0.05), ("zeta", 0.1), ("t", 0.05)]:
    a = O.field(n); a[:] += amp * rng.standard_normal(a.size); R.put(n, a)
for n, amp in [("sustr", 1e-4), ("svstr", 1e-4), ("stflx", 1e-5), ("bustr", 1e-5), ("bvstr", 1e-5)]:
    a = O.field(n); a[:] = amp * rng.standard_normal(a.size); R.put(n, a)
bad = []
def cmp(tag, names):
    for n in names:
        if not R.has(n): continue
        a, bb = R.get(n), O.field(n)
        if not np.array_equal(a, bb): bad.append((tag, n, float(np.abs(a - bb).max()), int(np.count_nonzero(a != bb))))
R.call("set_depth"); O.call("set_depth"); cmp("set_depth", ["Hz", "z_r", "z_w"])
R.call("rho_eos"); O.call("rho_eos"); cmp("rho_eos", ["rho", "pden", "rhoA", "rhoS", "bvf", "alpha", "beta"])
R.call("lmd_vmix"); O.call("lmd_vmix"); cmp("lmd_vmix", ["Akv", "Akt", "hsbl", "ghats"])
print("nonzero bvf", np.count_nonzero(O.field("bvf")), "Akv max", O.field("Akv").max(), "hsbl min", O.field("hsbl").min())
print("KPP-LINEAR-EOS-PINNED-OK" if not bad else bad)
'''


def test_upwelling_kpp_linear_eos_and_lmd_bitwise():
    """BASELINE config 5's physics (upwelling.h + the KPP options): linear EOS with BV_FREQUENCY and the
    expansion coefficients (rho_eos.F:751-780), then lmd_vmix, oracle vs the reference's object code."""
    from oracle import ref
    if not ref.available("upwelling_kpp"):
        pytest.skip("oracle/_ref/libromsref_upwelling_kpp.so not built")
    p = subprocess.run([sys.executable, "-c", KPP_SCRIPT % dict(root=ROOT)], capture_output=True, text=True,
                       timeout=600)
    assert "KPP-LINEAR-EOS-PINNED-OK" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]


# ----------------------------------------------------------------------------------------------------
# The six core routines (step2d, omega, pre_step3d, rhs3d, step3d_uv, step3d_t) and whole main3d passes.
# Bodies: tests/refchild.py (one reference configuration per process).
# ----------------------------------------------------------------------------------------------------
def _child(mode, tag, *args, timeout=900):
    import resource

    def big_stack():
        # the reference keeps its private (IminS:ImaxS,JminS:JmaxS,N) work arrays on the stack
        resource.setrlimit(resource.RLIMIT_STACK, (resource.RLIM_INFINITY, resource.RLIM_INFINITY))

    lib = {"upwelling_kpp_small": "upwelling_kpp", "upwelling_avg_small": "upwelling_avg", "upwelling_diag_small": "upwelling_diag",
           "upwelling_logdrag_small": "upwelling_logdrag", "upwelling_noadv_small": "upwelling_noadv", "upwelling_mask_small": "upwelling_mask",
           "benchmark_mask_small": "benchmark_mask", "benchmark_wetdry_small": "benchmark_wetdry", "upwelling_avg_mask_small": "upwelling_avg_mask",
           "kelvin": "kelvin_splines", "kelvin_small": "kelvin_splines", "kelvin_plain_small": "kelvin", "kelvin_plain": "kelvin",
           "upwelling_obc_small": "upwelling", "upwelling_mask_obc_small": "upwelling_mask", "seamount": "seamount",
           "seamount_small": "seamount", "grav_adj": "grav_adj", "grav_adj_small": "grav_adj", "overflow": "overflow", "overflow_small": "overflow",
           "upwelling_bihgeo_small": "upwelling_bihgeo", "upwelling_bihiso_small": "upwelling_bihiso", "upwelling_geouv_small": "upwelling_geouv", "upwelling_wetdry_small": "upwelling_wetdry", "upwelling_wetdry_obc_small": "upwelling_wetdry", "upwelling_prs31_small": "upwelling_prs31", "upwelling_wjgradp_small": "upwelling_wjgradp"}.get(tag, "benchmark" if tag.startswith("benchmark") else "upwelling")
    if not ref.available(lib):
        pytest.skip(f"oracle/_ref/libromsref_{lib}.so not built here")
    p = subprocess.run([sys.executable, "-m", "tests.refchild", mode, tag] + list(args), capture_output=True,
                       text=True, timeout=timeout, cwd=ROOT, preexec_fn=big_stack, errors="replace")
    return p.stdout[-3000:] + p.stderr[-2000:]


MAIN3D_CASES = [
    # tag, arguments                                                             what it covers
    ("upwelling_small", ["nsteps=100", "hadv=U3,HSIMT", "vadv=C4,HSIMT"]),       # roms_upwelling.in schemes
    ("upwelling_small", ["nsteps=100", "hadv=MPDATA,MPDATA", "vadv=MPDATA,MPDATA"]),
    ("upwelling_small", ["nsteps=30", "hadv=A4,C2", "vadv=SPLINES,C2"]),
    ("upwelling_small", ["nsteps=30", "hadv=C4,SU3", "vadv=A4,C4"]),
    ("upwelling_small", ["nsteps=30", "hadv=U3,U3", "vadv=C4,C4"]),
    ("upwelling_small", ["nsteps=20", "NtileI=2", "NtileJ=2", "hadv=U3,HSIMT", "vadv=C4,HSIMT"]),
    ("upwelling_small", ["nsteps=20", "NtileI=3", "NtileJ=1", "hadv=MPDATA,MPDATA", "vadv=MPDATA,MPDATA"]),
    # climatology nudging (LnudgeM3CLM: rhs3d.F:654-680; LtracerCLM + LnudgeTCLM: step3d_t.F:1866-1878): clima = bit 0 the 3-D
    # momentum, bit itrc tracer itrc; the climatology and coefficient arrays are data (cases.clima_arrays)
    ("upwelling_small", ["nsteps=40", "hadv=U3,HSIMT", "vadv=C4,HSIMT", "clima=7"]),
    ("upwelling_small", ["nsteps=20", "hadv=MPDATA,MPDATA", "vadv=MPDATA,MPDATA", "clima=4", "NtileI=2", "NtileJ=2"]),
    ("upwelling_small", ["nsteps=20", "hadv=U3,U3", "vadv=C4,C4", "clima=1"]),
    # ... bit 5: the 2-D momentum (LnudgeM2CLM, step2d_LF_AM3.h:2179-2203; round 6), alone and with the others, tiled
    ("upwelling_small", ["nsteps=40", "hadv=U3,HSIMT", "vadv=C4,HSIMT", "clima=32"]),
    ("upwelling_small", ["nsteps=20", "hadv=U3,U3", "vadv=C4,C4", "clima=39", "NtileI=2", "NtileJ=2"]),
    ("benchmark_small", ["nsteps=100"]),                                         # KPP, bulk fluxes, nonlinear EOS
    ("benchmark_small", ["nsteps=20", "NtileI=2", "NtileJ=2"]),
    ("upwelling_kpp_small", ["nsteps=100"]),                                     # BASELINE config 5 physics
    ("upwelling_logdrag_small", ["nsteps=40"]),                                  # UV_LOGDRAG (set_vbc.F:591-635)
    # WINDBASIN's option set on UPWELLING's functions (oracle/ref/upwelling_noadv.h): no UV_ADV, no UV_VIS2, no TS_DIF2
    ("upwelling_noadv_small", ["nsteps=40", "hadv=U3,HSIMT", "vadv=C4,HSIMT"]),
    ("upwelling_noadv_small", ["nsteps=20", "NtileI=2", "NtileJ=2", "hadv=MPDATA,MPDATA", "vadv=MPDATA,MPDATA"]),
    ("upwelling_mask_small", ["nsteps=60", "hadv=U3,HSIMT", "vadv=C4,HSIMT"]),   # MASKING: island + headland
    ("upwelling_mask_small", ["nsteps=30", "hadv=U3,U3", "vadv=C4,C4", "clima=39"]),   # ... with nudging of the 2-D and 3-D momentum and both tracers
    ("upwelling_mask_small", ["nsteps=30", "hadv=A4,C4", "vadv=SPLINES,C4"]),
    ("upwelling_mask_small", ["nsteps=20", "NtileI=2", "NtileJ=2", "hadv=U3,HSIMT", "vadv=C4,HSIMT"]),
    ("upwelling_mask_small", ["nsteps=40", "hadv=MPDATA,MPDATA", "vadv=MPDATA,MPDATA"]),   # mpdata_adiff.F's 13 masked blocks
    ("upwelling_mask_small", ["nsteps=20", "NtileI=2", "NtileJ=2", "hadv=MPDATA,MPDATA", "vadv=MPDATA,MPDATA"]),
    ("benchmark_mask_small", ["nsteps=60"]),                                     # MASKING with KPP, bulk fluxes, nonlinear EOS, geopotential mixing
    ("benchmark_mask_small", ["nsteps=20", "NtileI=2", "NtileJ=2"]),
    # open boundaries: the reference's KELVIN application (Cha / Fla west, Rad east, RADIATION_2D; analytic boundary data)
    ("kelvin_small", ["nsteps=60"]),
    ("kelvin_small", ["nsteps=20", "NtileI=2", "NtileJ=2"]),
    ("kelvin", ["nsteps=96"]),                                                   # roms_kelvin.in, full size and length
    # ... and as shipped (ROMS/Include/kelvin.h: no SPLINES_VDIFF / SPLINES_VVISC -- the plain tridiagonal vertical solvers)
    ("kelvin_plain_small", ["nsteps=60"]),
    ("kelvin_plain_small", ["nsteps=20", "NtileI=2", "NtileJ=2"]),
    ("kelvin_plain", ["nsteps=96"]),
    # climatology nudging beside open boundaries (round 6): radiation + nudging edges (tests/refchild.py OBC_PRESETS) with
    # LnudgeM2CLM + LnudgeM3CLM + LnudgeTCLM on -- the conditions' time scales come from the coefficient arrays and obcfac
    # VolCons (round 6; obc_volcons.F: obc_flux_tile behind every barotropic call, set_DUV_bc_tile in front of the next): KELVIN's
    # western + eastern edge, all four edges of an all-open basin, 2x2 tiles (the running sums in calling order), under MASKING
    ("kelvin_plain_small", ["nsteps=20", "volcons=5"]),
    ("kelvin_small", ["nsteps=20", "volcons=5"]),
    ("kelvin_plain_small", ["nsteps=20", "volcons=5", "NtileI=2", "NtileJ=2"]),
    ("kelvin_plain_small", ["nsteps=20", "preset=F", "volcons=15"]),
    ("kelvin_plain_small", ["nsteps=20", "preset=G", "volcons=10", "NtileI=2", "NtileJ=2"]),
    ("upwelling_mask_closed_small", ["nsteps=12", "preset=F", "volcons=15"]),
    ("upwelling_mask_closed_small", ["nsteps=12", "preset=B", "volcons=10", "NtileI=2", "NtileJ=2"]),
    ("kelvin_plain_small", ["nsteps=20", "preset=F", "clima=39"]),
    ("kelvin_plain_small", ["nsteps=20", "preset=C", "clima=39", "NtileI=2", "NtileJ=2"]),
    # two more of the reference's test applications: SEAMOUNT (no-slip walls, Akima advection, geopotential mixing without
    # KPP, quadratic drag, no closure) and GRAV_ADJ (lock exchange: MPDATA in a closed channel periodic across, no rotation)
    ("seamount_small", ["nsteps=60"]),
    ("seamount_small", ["nsteps=20", "NtileI=2", "NtileJ=2"]),
    ("seamount", ["nsteps=20"]),                                                 # roms_seamount.in, full size
    ("grav_adj_small", ["nsteps=60"]),
    ("grav_adj_small", ["nsteps=20", "NtileI=2", "NtileJ=1"]),
    # OVERFLOW (overflow.h as shipped but AVERAGES): MIX_ISO_TS through t3dmix2_iso.h, Vtransform 1 / Vstretching 1
    ("overflow_small", ["nsteps=60"]),
    ("overflow_small", ["nsteps=30", "NtileI=1", "NtileJ=2"]),
    ("overflow", ["nsteps=30"]),                                                 # roms_overflow.in, full size
    ("grav_adj", ["nsteps=40"]),                                                 # roms_grav_adj.in, full size
    # the standard density Jacobian prsgrd31.h (no DJ_GRADPS), plain and weighted (WJ_GRADP)
    ("upwelling_prs31_small", ["nsteps=60"]),
    ("upwelling_wjgradp_small", ["nsteps=60"]),
    ("upwelling_prs31_small", ["nsteps=20", "NtileI=2", "NtileJ=2"]),
    # the finite-volume pressure Jacobian of Lin (1997), prsgrd40.h (PJ_GRADP)
    ("upwelling_prs40_small", ["nsteps=60"]),
    ("upwelling_prs40_small", ["nsteps=20", "NtileI=2", "NtileJ=2"]),
    # climatology nudging (clima 39 = LnudgeM3CLM + both tracers + LnudgeM2CLM) together with WET_DRY, biharmonic mixing and the viscosity
    # along geopotentials (round 6; a tenth of VISC4 / TNU4 for the biharmonic cases: refdrive.make_case)
    ("upwelling_wetdry_small", ["nsteps=30", "clima=39"]),
    ("upwelling_bih_small", ["nsteps=30", "clima=39"]),
    ("upwelling_bihgeo_small", ["nsteps=30", "clima=39"]),
    ("upwelling_geouv_small", ["nsteps=30", "clima=39"]),
    ("upwelling_bihgeouv_small", ["nsteps=30", "clima=39"]),
    ("kelvin_geouv_small", ["nsteps=40"]),                                       # MIX_GEO_UV beside open boundaries (kelvin_geouv.h)
    ("kelvin_geouv_small", ["nsteps=20", "NtileI=2", "NtileJ=2"]),
    ("benchmark_iso_small", ["nsteps=40"]),                                      # MIX_ISO_TS with the nonlinear EOS (benchmark_iso.h)
    ("benchmark_iso_small", ["nsteps=20", "NtileI=2", "NtileJ=2"]),
    # WET_DRY with the closures (no WET_DRY statement of their own), the viscosity along geopotentials and the Jacobians prsgrd31 / 44
    # (round 6: oracle/ref/upwelling_wetdry_*.h; PJ_GRADP does not compile with WET_DRY in the reference: prsgrd40.h:98)
    ("upwelling_wetdry_gls_small", ["nsteps=40"]),
    ("upwelling_wetdry_gls_small", ["nsteps=20", "NtileI=2", "NtileJ=2", "hadv=MPDATA,MPDATA", "vadv=MPDATA,MPDATA"]),
    ("upwelling_wetdry_my25_small", ["nsteps=40"]),
    ("upwelling_wetdry_geouv_small", ["nsteps=40"]),
    ("upwelling_wetdry_geouv_small", ["nsteps=20", "NtileI=2", "NtileJ=2"]),
    ("upwelling_wetdry_prs31_small", ["nsteps=40"]),
    ("upwelling_wetdry_prs44_small", ["nsteps=40"]),
    ("upwelling_wetdry_prs44_small", ["nsteps=20", "NtileI=2", "NtileJ=2"]),
    ("upwelling_wetdry_iso_small", ["nsteps=40"]),                              # MIX_ISO_TS: t3dmix2_iso.h's masked and wet gradients
    ("upwelling_wetdry_iso_small", ["nsteps=20", "NtileI=2", "NtileJ=2"]),
    ("benchmark_wetdry_ddmix_small", ["nsteps=40"]),                            # LMD_DDMIX under WET_DRY (benchmark_wetdry.h -DLMD_DDMIX)
    # UV_VIS4 + MIX_GEO_UV (round 6): uv3dmix4_geo.h, the rotated stress tensor twice, under MASKING; in the channel and between four walls
    ("upwelling_bihgeouv_small", ["nsteps=60"]),
    ("upwelling_bihgeouv_small", ["nsteps=20", "NtileI=2", "NtileJ=2", "hadv=U3,U3", "vadv=C4,C4"]),
    ("upwelling_bihgeouv_closed_small", ["nsteps=30"]),
    ("upwelling_bihgeouv_closed_small", ["nsteps=20", "NtileI=2", "NtileJ=2"]),
    # closed basins (round 6): the periodic channel's set-up arrays between four walls (ana_grid.h has no bathymetry for that) --
    # every corner value, and the first biharmonic operator's conditions at western / eastern walls (a tenth of VISC4 / TNU4)
    ("upwelling_closed_small", ["nsteps=30"]),
    ("upwelling_closed_small", ["nsteps=20", "NtileI=2", "NtileJ=2", "hadv=MPDATA,MPDATA", "vadv=MPDATA,MPDATA"]),
    ("upwelling_bih_closed_small", ["nsteps=30"]),
    ("upwelling_bihgeo_closed_small", ["nsteps=30"]),
    ("upwelling_bihgeo_closed_small", ["nsteps=20", "NtileI=2", "NtileJ=2"]),
    ("upwelling_bihiso_closed_small", ["nsteps=30"]),
    # LMD_DDMIX (round 6): double-diffusive mixing in lmd_vmix's interior scheme, alfaobeta from both equations of state; the
    # state of cases.ddmix_state has salt fingering in one half and diffusive convection (Rrho on both sides of 0.5) in the other
    ("upwelling_kpp_ddmix_small", ["nsteps=40"]),
    ("upwelling_kpp_ddmix_small", ["nsteps=20", "NtileI=2", "NtileJ=2", "hadv=U3,HSIMT", "vadv=C4,HSIMT"]),
    # LMD_BKPP (round 6; lmd_bkpp.F with RI_SPLINES and the file's own SASHA): benchmark.h and oracle/ref/upwelling_kpp.h with
    # -DLMD_BKPP.  The runs from rest keep the layer empty (hbbl = -h); kick=30 adds random velocities of 0.3 m/s in front of step 3:
    # a layer 40 - 190 m thick that reaches several levels (the shape functions, the overlap rule with the surface layer)
    ("benchmark_bkpp_small", ["nsteps=40"]),
    ("upwelling_kpp_bkpp_small", ["nsteps=60"]),
    ("benchmark_bkpp_small", ["nsteps=20", "kick=30"]),
    ("upwelling_kpp_bkpp_small", ["nsteps=20", "kick=30"]),
    ("upwelling_kpp_bkpp_small", ["nsteps=20", "kick=30", "NtileI=2", "NtileJ=2"]),
    ("benchmark_bkpp_small", ["nsteps=20", "kick=30", "NtileI=2", "NtileJ=2"]),
    ("benchmark_ddmix_small", ["nsteps=40"]),
    ("benchmark_ddmix_small", ["nsteps=20", "NtileI=2", "NtileJ=2"]),
    # the finite-volume Jacobians of Shchepetkin & McWilliams (2003) with a reconstructed density profile (round 6): prsgrd44.h
    # (PJ_GRADPQ4: quartic, power-law reconciliation; any partition) and prsgrd42.h (PJ_GRADPQ2: parabolic WENO and a second pass
    # that reads rv(Iend+1,j) -- a column no tile computes: one tile only, oracle/orc_prs4x.c)
    ("upwelling_prs44_small", ["nsteps=60"]),
    ("upwelling_prs44_small", ["nsteps=20", "NtileI=2", "NtileJ=2"]),
    ("upwelling_prs44_small", ["nsteps=20", "hadv=U3,HSIMT", "vadv=C4,HSIMT"]),
    ("upwelling_prs42_small", ["nsteps=60"]),
    ("upwelling_prs42_small", ["nsteps=20", "hadv=U3,HSIMT", "vadv=C4,HSIMT"]),
    # biharmonic mixing along s-surfaces (oracle/ref/upwelling_bih.h: UV_VIS4, TS_DIF4): uv3dmix4_s.h, t3dmix4_s.h and the
    # UV_VIS4 block of step2d_LF_AM3.h; three ghost points (inp_par.F:214)
    ("upwelling_bih_small", ["nsteps=60"]),
    ("upwelling_bih_small", ["nsteps=20", "NtileI=2", "NtileJ=2"]),
    ("upwelling_bih_small", ["nsteps=20", "hadv=U3,U3", "vadv=C4,C4"]),
    # harmonic viscosity along geopotential surfaces under MASKING (oracle/ref/upwelling_geouv.h: uv3dmix2_geo.h)
    ("upwelling_geouv_small", ["nsteps=60", "hadv=U3,HSIMT", "vadv=C4,HSIMT"]),
    ("upwelling_geouv_small", ["nsteps=20", "NtileI=2", "NtileJ=2", "hadv=U3,U3", "vadv=C4,C4"]),
    # ... with the tracers along geopotential surfaces (oracle/ref/upwelling_bihgeo.h: t3dmix4_geo.h)
    ("upwelling_bihgeo_small", ["nsteps=60"]),
    ("upwelling_bihgeo_small", ["nsteps=20", "NtileI=2", "NtileJ=2"]),
    # ... along isopycnic surfaces (oracle/ref/upwelling_bihiso.h: t3dmix4_iso.h)
    ("upwelling_bihiso_small", ["nsteps=60"]),
    ("upwelling_bihiso_small", ["nsteps=20", "NtileI=2", "NtileJ=2"]),
    # wetting and drying (oracle/ref/upwelling_wetdry.h: MASKING + WET_DRY): wetdry.F, the WET_DRY branches of step2d, rhs3d,
    # prsgrd32, t3dmix2_s, uv3dmix2_s, step3d_uv, set_vbc (LIMIT_BSTRESS), ini_fields, zetabc / u2dbc / v2dbc / u3dbc / v3dbc;
    # a beach that dries above the still water level, a ridge of water running up it; _obc: a closed basin (all four walls)
    ("upwelling_wetdry_small", ["nsteps=60"]),
    ("upwelling_wetdry_small", ["nsteps=20", "NtileI=2", "NtileJ=2"]),
    ("upwelling_wetdry_small", ["nsteps=30", "hadv=U3,U3", "vadv=C4,C4"]),
    # WET_DRY with the COARE bulk fluxes, the solar source, KPP and geopotential tracer mixing (oracle/ref/benchmark_wetdry.h),
    # and with MPDATA (mpdata_adiff.F's wet masks)
    ("benchmark_wetdry_small", ["nsteps=60"]),
    ("benchmark_wetdry_small", ["nsteps=20", "NtileI=2", "NtileJ=2"]),
    ("benchmark_wetdry_small", ["nsteps=40", "hadv=MPDATA,MPDATA", "vadv=MPDATA,MPDATA"]),
    ("upwelling_wetdry_small", ["nsteps=40", "hadv=MPDATA,MPDATA", "vadv=MPDATA,MPDATA"]),
    ("upwelling_wetdry_obc_small", ["nsteps=40"]),
    ("upwelling_wetdry_obc_small", ["nsteps=20", "NtileI=2", "NtileJ=2"]),
    # the generic length-scale closure (gls_prestep.F, gls_corstep.F, tkebc_im.F): upwelling.h built with -DGLS_MIXING
    # (Kantha-Clayson, N2S2_HORAVG, RI_SPLINES; k-epsilon and k-omega parameters of roms_upwelling.in), and the other
    # compile-time forms: Canuto A under MASKING ("gen" parameters), Canuto B with K_C2ADVECTION, CHARNOK, CRAIG_BANNER
    # (k-kl parameters: the wall function), Galperin with K_C4ADVECTION (k-omega)
    ("upwelling_gls_small", ["nsteps=100"]),
    ("upwelling_gls_kw_small", ["nsteps=60"]),
    ("upwelling_gls_ca_small", ["nsteps=100"]),
    ("upwelling_gls_cb_small", ["nsteps=100"]),
    ("upwelling_gls_gal_small", ["nsteps=60"]),
    ("upwelling_gls_small", ["nsteps=20", "NtileI=2", "NtileJ=2"]),
    ("upwelling_gls_ca_small", ["nsteps=20", "NtileI=2", "NtileJ=2"]),
    ("upwelling_gls_cb_small", ["nsteps=20", "NtileI=3", "NtileJ=1", "hadv=MPDATA,MPDATA", "vadv=MPDATA,MPDATA"]),
    # the Mellor-Yamada level 2.5 closure (my25_prestep.F = gls_prestep.F; my25_corstep.F with its own production, boundary
    # values, stability functions -- and its eastern edge copy onto Iend-1, restated as it stands): upwelling.h built with
    # -DMY25_MIXING (Kantha-Clayson, smoothing, spline shear) and oracle/ref/upwelling_my25_gal.h (Galperin, K_C4ADVECTION)
    ("upwelling_my25_small", ["nsteps=60"]),
    ("upwelling_my25_gal_small", ["nsteps=60"]),
    ("upwelling_my25_small", ["nsteps=20", "NtileI=2", "NtileJ=2"]),
    ("upwelling_my25_gal_small", ["nsteps=20", "NtileI=3", "NtileJ=1", "hadv=MPDATA,MPDATA", "vadv=MPDATA,MPDATA"]),
    # open boundaries together with GLS_MIXING (oracle/ref/kelvin_gls.h): tkebc's zero-gradient edges beside radiating ones,
    # the closure driven by the Kelvin wave's bottom stress
    ("kelvin_gls_small", ["nsteps=60"]),
    ("kelvin_gls_small", ["nsteps=20", "NtileI=2", "NtileJ=2"]),
    # ... and tkebc_im.F's radiation condition (LBC(isMtke) = Rad) on the eastern | western edge, gradient opposite
    ("kelvin_gls_small", ["nsteps=40", "lbc_tke=Gra,Clo,Rad,Clo"]),
    ("kelvin_gls_small", ["nsteps=40", "lbc_tke=Rad,Clo,Gra,Clo", "NtileI=2", "NtileJ=2"]),
    ("upwelling", ["nsteps=100"]),                                               # BASELINE configs[0], full size
    ("benchmark1", ["nsteps=4"]),                                                # BASELINE configs[1], full size
]


@pytest.mark.parametrize("tag,args", MAIN3D_CASES, ids=[t + ":" + "+".join(a) for t, a in MAIN3D_CASES])
def test_main3d_steps_bitwise(tag, args):
    """Whole main3d passes: the reference's own kernels called in main3d.F order (ref_glue.F90:ref_main3d)
    against orc_main3d_step -- every state array compared after every step with array_equal, the stepping
    indices at the end, and every `diag` line the reference prints (diag.F FORMAT 30/40)."""
    out = _child("main3d", tag, *args)
    assert "MAIN3D-OK bitwise" in out, out


@pytest.mark.parametrize("tag,args", [
    ("upwelling_small", ["hadv=U3,HSIMT", "vadv=C4,HSIMT"]),
    ("upwelling_small", ["hadv=MPDATA,MPDATA", "vadv=MPDATA,MPDATA"]),
    ("upwelling_small", ["hadv=A4,SU3", "vadv=SPLINES,A4"]),
    ("upwelling_small", ["hadv=C2,C4", "vadv=C2,C4"]),
    ("benchmark_small", []),
    ("upwelling_kpp_small", []),
    ("upwelling_mask_small", ["hadv=U3,HSIMT", "vadv=C4,HSIMT"]),
    ("upwelling_mask_small", ["hadv=C4,A4", "vadv=C4,A4"]),
    ("upwelling_mask_small", ["hadv=MPDATA,MPDATA", "vadv=MPDATA,MPDATA"]),
    ("benchmark_mask_small", []),
    ("upwelling_wetdry_small", ["hadv=U3,HSIMT", "vadv=C4,HSIMT"]),
    ("upwelling_wetdry_obc_small", ["hadv=A4,C4", "vadv=SPLINES,C4"]),
    # t3dmix4_geo.h (called by rhs3d) in the periodic channel: the conditions on the first operator at the southern and
    # northern walls; *_closed_small (round 6): between four walls -- the western / eastern conditions and the corner averages
    # of t3dmix4_geo.h:475-600, t3dmix4_iso.h:504-618, t3dmix4_s.h on a perturbed state.  (Until round 6 "the reference is NaN in
    # a closed basin": its ana_grid.h gives UPWELLING no bathymetry without a periodic direction -- the set-up arrays of the
    # channel are the case's input now, refdrive.reference -- and the channel cases' VISC4 / TNU4 blow up between four walls.)
    ("upwelling_bihgeouv_small", ["hadv=U3,HSIMT", "vadv=C4,HSIMT"]),                  # uv3dmix4_geo.h (round 6)
    ("upwelling_bihgeouv_small", ["hadv=U3,U3", "vadv=C4,C4", "NtileI=2", "NtileJ=2"]),
    ("upwelling_bihgeouv_closed_small", ["hadv=U3,HSIMT", "vadv=C4,HSIMT"]),
    ("upwelling_closed_small", ["hadv=U3,HSIMT", "vadv=C4,HSIMT"]),
    ("upwelling_closed_small", ["hadv=MPDATA,MPDATA", "vadv=MPDATA,MPDATA", "NtileI=2", "NtileJ=2"]),
    ("upwelling_bih_closed_small", ["hadv=U3,HSIMT", "vadv=C4,HSIMT"]),
    ("upwelling_bihgeo_closed_small", ["hadv=U3,HSIMT", "vadv=C4,HSIMT"]),
    ("upwelling_bihgeo_closed_small", ["hadv=U3,U3", "vadv=C4,C4", "NtileI=2", "NtileJ=2"]),
    ("upwelling_bihiso_closed_small", ["hadv=U3,HSIMT", "vadv=C4,HSIMT"]),
    ("upwelling_bihiso_closed_small", ["hadv=U3,U3", "vadv=C4,C4", "NtileI=2", "NtileJ=2"]),
    ("upwelling_bihgeo_small", ["hadv=U3,HSIMT", "vadv=C4,HSIMT"]),
    ("upwelling_bihgeo_small", ["hadv=U3,U3", "vadv=C4,C4", "NtileI=2", "NtileJ=2"]),
    ("upwelling_bihiso_small", ["hadv=U3,HSIMT", "vadv=C4,HSIMT"]),
    ("upwelling_bihiso_small", ["hadv=U3,U3", "vadv=C4,C4", "NtileI=2", "NtileJ=2"]),
    ("upwelling_geouv_small", ["hadv=U3,HSIMT", "vadv=C4,HSIMT"]),                     # uv3dmix2_geo.h (called by rhs3d)
    ("kelvin_plain_small", ["volcons=15"]),                                              # obc_volcons.F inside step2d_tile
])
def test_core_kernels_bitwise(tag, args):
    """step2d_tile (step2d_LF_AM3.h:163; first predictor, correctors, last predictor), omega_tile (omega.F:96),
    pre_step3d_tile (pre_step3d.F:126), rhs3d (rhs3d.F:25 incl. rhs3d_tile :196), step3d_uv_tile
    (step3d_uv.F:134), step3d_t_tile (step3d_t.F:120): one call each on a randomly perturbed mid-run state."""
    out = _child("kernels", tag, *args)
    assert "KERNELS-OK bitwise" in out, out


OBC_CASES = [(t, p, a) for t in ("kelvin_plain_small", "upwelling_obc_small", "upwelling_mask_obc_small") for p in "ABCDEFG" for a in ([],)] + \
            [("kelvin_plain_small", p, ["NtileI=2", "NtileJ=2"]) for p in "FG"] + \
            [("upwelling_wetdry_obc_small", p, a) for p in "ABEFG" for a in ([],)] + [("upwelling_wetdry_obc_small", "F", ["NtileI=2", "NtileJ=2"])] + \
            [(t, p, ["clima=39"]) for t in ("kelvin_plain_small", "upwelling_obc_small", "upwelling_mask_obc_small") for p in "CDFG"] + \
            [("kelvin_plain_small", "F", ["clima=39", "NtileI=2", "NtileJ=2"])]     # the radiation + nudging presets with the nudging switches on: per-point time scales


@pytest.mark.parametrize("tag,preset,args", OBC_CASES, ids=[f"{t}:{p}" + ("+clima" if "clima=39" in a else "") + ("+tiles" if "NtileI=2" in a else "") for t, p, a in OBC_CASES])
def test_open_boundary_routines_bitwise(tag, preset, args):
    """zetabc_tile, u2dbc_tile, v2dbc_tile, u3dbc_tile, v3dbc_tile, t3dbc_tile of the reference against the oracle's
    (oracle/orc_obc.c) on a random state with random boundary data: radiation with and without nudging, Chapman explicit
    and implicit, Flather, Shchepetkin, clamped, gradient and closed, every kind on every edge (tests/refchild.py:
    OBC_PRESETS), for the four stepping variants that select `know` and `dt2d` and the three 3-D time-level pairs; with
    RADIATION_2D (the reference's kelvin.h), without it (UPWELLING's library on a closed-basin grid), under MASKING, and
    on 2x2 tiles.  Includes the reference's southern free-surface radiation branch, which differences towards the boundary
    row (zetabc.F:455,486-487).  With LnudgeM2CLM / LnudgeM3CLM / LnudgeTCLM on (clima=39) the radiation + nudging edges take
    their time scales from the nudging coefficient arrays and obcfac (u2dbc_im.F:158-183, u3dbc_im.F:113-171, t3dbc_im.F:120-167)."""
    out = _child("obc", tag, "preset=" + preset, *args)
    assert "OBC-OK bitwise" in out, out[-1500:]


@pytest.mark.parametrize("tag", ["benchmark_small", "upwelling_kpp_small", "upwelling_small", "upwelling_logdrag_small",
                                 "upwelling_mask_small", "benchmark_mask_small"])
def test_physics_routines_bitwise(tag):
    """set_depth, set_massflux, rho_eos (rho, pden, rhoA, rhoS, bvf, alpha, beta: rho_eos.F:247-560), the analytic
    atmosphere + ana_srflux (set_data.F), bulk_flux (bulk_flux.F:208), set_vbc (QDRAG), lmd_vmix, omega,
    wvelocity, set_zeta, prsgrd32, t3dmix2 (_geo for BENCHMARK: t3dmix2_geo.h:90), uv3dmix2, diag -- each
    routine's outputs array_equal on a perturbed state; diag through avgkp and its printed line."""
    out = _child("physics", tag)
    assert "PHYSICS-OK bitwise" in out, out


@pytest.mark.parametrize("args", [["nsteps=9", "nAVG=3", "ntsAVG=1"], ["nsteps=8", "nAVG=2", "ntsAVG=3"],
                                  ["nsteps=5", "nAVG=1", "ntsAVG=1"]])
def test_set_avg_bitwise(args):
    """set_avg_tile (set_avg.F:96) for the Aout switches of roms_upwelling.in -- the reference built from
    oracle/ref/upwelling_avg.h (UPWELLING with AVERAGES) against orc_set_avg, all 22 time-averaged arrays after
    every call: the set, accumulate and convert (scale + periodic refill) phases of several windows."""
    out = _child("avg", "upwelling_avg_small", *args)
    assert "AVG-OK bitwise" in out, out


@pytest.mark.parametrize("args", [["nsteps=9", "nDIA=3", "ntsDIA=1"], ["nsteps=8", "nDIA=2", "ntsDIA=3", "hadv=U3,U3", "vadv=C4,C4"],
                                  ["nsteps=6", "nDIA=1", "ntsDIA=1", "hadv=A4,C2", "vadv=SPLINES,C2"],
                                  ["nsteps=6", "nDIA=2", "ntsDIA=1", "NtileI=2", "NtileJ=2"]])
def test_set_diags_bitwise(args):
    """DIAGNOSTICS_TS: the per-term tracer tendencies -- DiaTwrk as pre_step3d.F:925, t3dmix2_s.h:293, step3d_t.F:908-912,
    1357-1362, 1716-1719, 1892-1904 leave it after EVERY kernel of the step, and DiaTrc / avgzeta of set_diags.F (set,
    accumulate, convert phases of several windows) -- the reference built from ROMS/Include/upwelling.h AS SHIPPED
    (oracle/ref/build_ref.sh upwelling_diag: AVERAGES, DIAGNOSTICS_TS, DIAGNOSTICS_UV) against the oracle, array_equal."""
    out = _child("dia", "upwelling_diag_small", *args)
    assert "DIA-OK bitwise" in out, out


@pytest.mark.parametrize("args", [["nsteps=9", "nDIA=3", "ntsDIA=1", "uv=1"], ["nsteps=6", "nDIA=2", "ntsDIA=2", "hadv=U3,U3", "vadv=C4,C4", "uv=1"],
                                  ["nsteps=6", "nDIA=2", "ntsDIA=1", "NtileI=2", "NtileJ=2", "uv=1"]])
def test_set_diags_uv_bitwise(args):
    """DIAGNOSTICS_UV: the per-term momentum tendencies of the reference built from upwelling.h AS SHIPPED -- DiaRU / DiaRV
    (the two levels of every 3-D right-hand-side term), their vertical sums DiaRUfrc / DiaRVfrc, DiaU3wrk / DiaV3wrk,
    the fast-time arrays DiaRUbar, DiaU2int, DiaU2wrk (and V) and the accumulated output DiaU2d, DiaU3d (and V) -- against
    the oracle (orc_diags_uv.c and its hooks in prsgrd, rhs3d, uv3dmix2, pre_step3d, step2d, step3d_uv) after rhs3d, after
    EVERY step2d call, after step3d_uv, step3d_t and set_diags of several windows: array_equal, 16 arrays."""
    out = _child("dia", "upwelling_diag_small", *args)
    assert "DIA-OK bitwise" in out, out


def test_set_avg_on_a_masked_run_bitwise():
    """AVERAGES together with MASKING (reference built from oracle/ref/upwelling_avg_mask.h): the 22 averaged arrays of
    roms_upwelling.in carry no mask arithmetic of their own (set_avg.F uses masks for rotated and vorticity fields only),
    the masked state they accumulate does -- all of them after every call of two windows."""
    out = _child("avg", "upwelling_avg_mask_small", "nsteps=7", "nAVG=3", "ntsAVG=1")
    assert "AVG-OK bitwise" in out, out


@pytest.mark.parametrize("args", [["nsteps=9", "nAVG=3", "ntsAVG=1"], ["nsteps=6", "nAVG=1", "ntsAVG=1"], ["nsteps=9", "nAVG=4", "ntsAVG=2", "NtileI=2", "NtileJ=2"]])
def test_set_avg_with_wetting_and_drying_bitwise(args):
    """AVERAGES together with WET_DRY (round 6; reference built from oracle/ref/upwelling_wetdry_avg.h): every averaged field
    times the full mask (land x wet) of its grid type where it is set and where it is added (set_avg.F:302 ..., :1652 ...), the
    wet-point counters rmask_avg / umask_avg / vmask_avg (:257-288, :1608-1645), the sums divided by them (:2980-2988) -- all
    22 arrays after every call, on the beach-and-ridge state whose shore line moves."""
    out = _child("avg", "upwelling_wetdry_avg_small", *args)
    assert "AVG-OK bitwise" in out, out


# ----------------------------------------------------------------------------------------------------
# main3d.F itself cannot be compiled here (it USEs the NetCDF readers and writers); the fixtures and the main3d
# tests above are driven by oracle/ref/ref_glue.F90:ref_main3d, which calls the reference's kernels in main3d's
# order.  That order is checked against the TEXT of main3d.F / post_initial.F, pre-processed for each application the
# way build_ref.sh pre-processes the sources: a change of the glue (or of the reference) that reorders, drops or
# adds a kernel call, or flips the direction of a tile loop, fails here.
# ----------------------------------------------------------------------------------------------------
def _cpp(path, up, hdrpath, extra=()):
    REF = "/root/reference"
    cmd = ["/usr/bin/cpp", "-P", "-traditional", "-w", f"-D{up}", f'-DROMS_HEADER="{hdrpath}"', '-DHEADER="x.h"',
           "-DLINUX", "-DX86_64", "-DGFORTRAN", "-DNestedGrids=1", f'-DROOT_DIR="{REF}"',
           f'-DANALYTICAL_DIR="{REF}/ROMS/Functionals"', f'-DHEADER_DIR="{REF}/ROMS/Include"', '-DGIT_URL="x"', '-DGIT_REV="x"',
           '-DMY_OS="Linux"', '-DMY_CPU="x86_64"', '-DMY_FORT="gfortran"', '-DMY_FC="flang"', '-DMY_FFLAGS="-O2"', *extra,
           f"-I{REF}/ROMS/Include", f"-I{REF}/ROMS/Nonlinear", f"-I{REF}/ROMS/Functionals", f"-I{REF}/ROMS/Utility",
           f"-I{REF}/ROMS/Drivers", f"-I{REF}/Master", path]
    return subprocess.run(cmd, capture_output=True, text=True, check=True).stdout


def _tile_calls(text, start, end):
    """[(direction of the enclosing tile loop, routine)] of the `CALL name (ng, tile` statements between two markers"""
    import re
    body = text[text.index(start):]
    body = body[:body.index(end)]
    seq, direction = [], None
    for line in body.splitlines():
        s = line.strip()
        m = re.match(r"DO tile=(first|last)_tile\(ng\)", s)
        if m:
            direction = "+" if m.group(1) == "first" else "-"
        m = re.match(r"CALL (\w+) \((?:ng, )?tile", s)
        if m:
            seq.append((direction, m.group(1)))
    return seq


@pytest.mark.parametrize("up,hdr,extra", [
    ("UPWELLING", "upwelling.h", ("-DPERFECT_RESTART",)), ("BENCHMARK", "benchmark.h", ()), ("KELVIN", "kelvin.h", ()),
    ("UPWELLING", os.path.join(ROOT, "oracle", "ref", "upwelling_kpp.h"), ()),
    ("BENCHMARK", os.path.join(ROOT, "oracle", "ref", "benchmark_mask.h"), ()),
    ("KELVIN", os.path.join(ROOT, "oracle", "ref", "kelvin_splines.h"), ()),
])
def test_glue_calls_the_kernels_in_the_order_of_main3d_F(up, hdr, extra):
    if not os.path.isdir("/root/reference/ROMS"):
        pytest.skip("needs the reference tree")
    ref = _tile_calls(_cpp("/root/reference/ROMS/Nonlinear/main3d.F", up, hdr, extra), "STEP_LOOP : DO istep", "END DO STEP_LOOP")
    post = _tile_calls(_cpp("/root/reference/ROMS/Nonlinear/post_initial.F", up, hdr, extra), "SUBROUTINE post_initial", "END SUBROUTINE post_initial")
    glue = _tile_calls(_cpp(os.path.join(ROOT, "oracle", "ref", "ref_glue.F90"), up, hdr, extra), "SUBROUTINE ref_main3d", "END SUBROUTINE ref_main3d")
    glue = [(d, "set_data" if n == "ref_set_data" else n) for d, n in glue]
    assert len(ref) >= 15 and ("-", "step2d") in ref and ("+", "step2d") in ref
    # the glue inlines post_initial (main3d.F:334) where main3d.F calls it: behind set_data of the first step
    k = glue.index(("+", "ini_zeta"))
    inlined = glue[k:k + len(post)]
    assert inlined == post, (inlined, post)
    assert glue[:k] + glue[k + len(post):] == ref, (glue, ref)


def _index_statements(text, start, end):
    """the statements of the barotropic loop's index state machine (main3d.F:810-918), continuation lines joined, blanks dropped"""
    import re
    body = text[text.index(start):]
    body = body[:body.index(end)]
    joined, cur = [], ""
    for line in body.splitlines():
        s = line.strip()
        if not s or s.startswith("!"):
            continue
        s = s.split("!")[0].strip()              # (trailing comments: the glue cites main3d.F lines)
        if s.startswith("&"):
            cur += s[1:].strip()
            continue
        if cur:
            joined.append(cur)
        cur = s.rstrip("&").strip()
    joined.append(cur)
    keep = re.compile(r"^(IF \(|ELSE|END IF|next_indx1=|PREDICTOR_2D_STEP\(ng\)=|iif\(ng\)=|kstp\(ng\)=|knew\(ng\)=|krhs\(ng\)=|indx1\(ng\)=)")
    return [re.sub(r"\s+", "", s) for s in joined if keep.match(s)]


def test_glue_steps_the_barotropic_indices_as_main3d_F_does():
    """kstp / knew / krhs / indx1 / iif / PREDICTOR_2D_STEP through the 2*nfast+1 step2d calls: statement for statement
    the text of main3d.F's LOOP_2D (the oracle's and the library's state machines are compared with this glue at run time)."""
    if not os.path.isdir("/root/reference/ROMS"):
        pytest.skip("needs the reference tree")
    ref = _index_statements(_cpp("/root/reference/ROMS/Nonlinear/main3d.F", "UPWELLING", "upwelling.h", ("-DPERFECT_RESTART",)),
                            "LOOP_2D : DO my_iif", "END DO LOOP_2D")
    glue = _index_statements(_cpp(os.path.join(ROOT, "oracle", "ref", "ref_glue.F90"), "UPWELLING", "upwelling.h", ("-DPERFECT_RESTART",)),
                             "LOOP_2D : DO my_iif", "END DO LOOP_2D")
    assert len(ref) >= 12 and "knew(ng)=3" in ref
    assert glue == ref, (glue, ref)
