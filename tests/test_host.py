"""Fortran host driver (roms_amd/host): its set-up (roms.in -> grid, s-coordinate, filter weights,
metrics, initial fields) must reproduce the reference's arrays held in tests/golden/*_init.npz bit
for bit -- those fixtures were written by the reference's own set_scoord / set_weights / ana_grid /
metrics / ana_initial / set_depth (tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest

from tests import util

HOST_FIELDS = ["h", "f", "fomn", "pm", "pn", "om_r", "on_r", "om_u", "on_u", "om_v", "on_v", "om_p", "on_p", "omn",
               "pmon_r", "pnom_r", "pmon_p", "pnom_p", "pmon_u", "pnom_u", "pmon_v", "pnom_v", "angler", "xr", "yr",
               "rdrag", "rdrag2", "visc2_r", "visc2_p", "diff2", "Hz", "z_r", "z_w", "zeta", "ubar", "vbar", "u", "v",
               "t", "Zt_avg1", "Akv", "Akt", "dmde", "dndx", "lonr", "latr", "sc_r", "Cs_r", "sc_w", "Cs_w"]


def _host(cs):
    from roms_amd import hostlib
    if not os.path.exists(hostlib.LIB):
        from roms_amd import build
        build.build_hip()
        build.build_host()
    return hostlib.Host(params=cs)


@pytest.mark.parametrize("tag", ["upwelling", "upwelling_small", "benchmark_small"])
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


def test_host_needs_device_for_run():
    cs = util.case_for("upwelling_small")
    H = _host(cs)
    try:
        assert H.lib.roms_host_run(1, 0) == 8          # no device context yet: usage error
    finally:
        H.finalize()
