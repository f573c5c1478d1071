"""Helpers shared by the tests: load golden fixtures, build oracle / HIP states from them."""
import os

import numpy as np

from tests import cases

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
EMU_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu", "libroms_hip_emu.so")

INIT_FIELDS = ["h", "f", "fomn", "pm", "pn", "om_r", "on_r", "om_u", "on_u", "om_v", "on_v", "om_p", "on_p", "omn",
               "pmon_r", "pnom_r", "pmon_p", "pnom_p", "pmon_u", "pnom_u", "pmon_v", "pnom_v", "angler", "xr", "yr",
               "rdrag", "visc2_r", "visc2_p", "diff2", "Hz", "z_r", "z_w", "Huon", "Hvom", "zeta", "ubar", "vbar",
               "u", "v", "t", "rho", "pden", "rhoA", "rhoS", "Zt_avg1", "Akv", "Akt",
               "dmde", "dndx", "lonr", "latr", "rdrag2", "bvf", "alpha", "beta", "hsbl"]
STATE_FIELDS = INIT_FIELDS + ["rzeta", "rubar", "rvbar", "W", "wvel", "ru", "rv", "rufrc", "rvfrc", "DU_avg1",
                              "DU_avg2", "DV_avg1", "DV_avg2", "sustr", "svstr", "bustr", "bvstr", "stflx", "btflx",
                              "stflux", "btflux", "srflx", "ghats", "Uwind", "Vwind", "Tair", "Pair", "Hair", "rain",
                              "cloud", "lhflx", "shflx", "lrflx"]
PROGNOSTIC = ["zeta", "ubar", "vbar", "u", "v", "t", "W", "wvel", "Hz", "z_r", "z_w", "Huon", "Hvom", "rho", "ru", "rv",
              "Zt_avg1", "DU_avg1", "DV_avg1", "DU_avg2", "DV_avg2", "rufrc", "rvfrc", "rzeta", "rubar", "rvbar",
              "Akv", "Akt", "hsbl", "ghats", "stflx", "sustr", "svstr", "bustr", "bvstr", "srflx", "bvf"]


def load_init(tag, nghost=None):
    """Golden initial state.  The fixtures were generated with 3 ghost points (HSIMT salinity);
    for a 2-ghost-point configuration the extra periodic ghost column/row is cropped."""
    g = dict(np.load(os.path.join(GOLDEN, f"{tag}_init.npz")))
    b = g["bounds"]
    if nghost is not None and nghost != int(b[54]):
        assert int(b[54]) == 3 and nghost == 2
        LBi, UBi, LBj, UBj = [int(x) for x in b[:4]]
        ni, nj = UBi - LBi + 1, UBj - LBj + 1
        ci = 1 if LBi < 0 else 0      # periodic in xi
        cj = 1 if LBj < 0 else 0
        for k, a in list(g.items()):
            if a.ndim == 1 and a.size >= ni * nj and a.size % (ni * nj) == 0:
                a = a.reshape(-1, nj, ni)[:, cj:nj - cj, ci:ni - ci]
                g[k] = np.ascontiguousarray(a).ravel()
        b = b.copy()
        b[0] += ci; b[1] -= ci; b[2] += cj; b[3] -= cj; b[54] = 2
        g["bounds"] = b
    return g


def nghost_for(cs):
    return 3 if any(x in ("HSIMT", "MPDATA") for x in cs["hadv"]) else 2


def case_for(tag, **kw):
    if tag == "upwelling":
        return cases.upwelling(**kw)
    if tag == "upwelling_small":
        return cases.upwelling(Lm=14, Mm=18, N=8, **kw)
    if tag == "upwelling_kpp_small":
        return cases.upwelling_kpp(Lm=14, Mm=18, N=8, **kw)
    if tag == "benchmark_small":
        return cases.benchmark(Lm=24, Mm=16, N=10, **kw)
    raise KeyError(tag)


def make_oracle(cs, g):
    from oracle import orc
    O = orc.Oracle(cases.oracle_cfg(cs, float(g["scalars"][0]), int(g["bounds"][58]), g["weight"]))
    for n in INIT_FIELDS:
        if n in g:
            O.field(n)[:] = g[n]
    for n in ["sc_r", "Cs_r", "sc_w", "Cs_w"]:
        O.field(n)[:] = g[n]
    return O


def make_hip(cs, g, lib_path=None, device=0, ninfo=0):
    from roms_amd import hiplib
    cfg = cases.hip_cfg(cs, float(g["scalars"][0]), int(g["bounds"][58]), g["weight"], g["sc_r"], g["Cs_r"],
                        g["sc_w"], g["Cs_w"], device=device)
    cfg.ninfo = ninfo
    H = hiplib.Context(cfg, lib_path)
    for n in INIT_FIELDS:
        if n in g:
            H.upload(n, g[n])
    return H


def push_state(O, H, fields=STATE_FIELDS):
    for n in fields:
        H.upload(n, O.field(n))
    s = O.step
    H.set_stepping(iic=s.iic, iif=s.iif, nstp=s.nstp, nnew=s.nnew, nrhs=s.nrhs, kstp=s.kstp, knew=s.knew,
                   krhs=s.krhs, indx1=s.indx1, predictor=s.predictor, time=s.time)


def relrms(a, b):
    d = np.sqrt(np.mean((a - b) ** 2))
    s = np.sqrt(np.mean(b ** 2))
    return d / s if s > 0 else d


def ns_periodic_case(hadv, vadv, ng, ewp):
    """upwelling_small with a periodic eta direction (and closed xi walls if not ewp): the EW-periodic
    fixture re-embedded, ghost rows = periodic images of the interior rows."""
    cs = case_for("upwelling_small", hadv=hadv, vadv=vadv)
    g = load_init("upwelling_small", ng)
    cs["NSperiodic"] = 1
    cs["EWperiodic"] = ewp
    LBi, UBi, LBj, UBj = [int(x) for x in g["bounds"][:4]]
    ni, nj = UBi - LBi + 1, UBj - LBj + 1
    Lm, Mm = cs["Lm"], cs["Mm"]
    Im = Lm + ((Lm + 2) // 2 - (Lm + 1) // 2)
    Jm = Mm + ((Mm + 2) // 2 - (Mm + 1) // 2)
    rows = [((j - 1) % Mm) + 1 - LBj for j in range(-ng, Jm + ng + 1)]        # source row (old local index)
    for k, a in list(g.items()):
        if a.ndim == 1 and a.size >= ni * nj and a.size % (ni * nj) == 0:
            a = a.reshape(-1, nj, ni)[:, rows, :]
            if not ewp:
                a = a[:, :, (0 - LBi):(Im + 1 - LBi) + 1]
            g[k] = np.ascontiguousarray(a).ravel()
    return cs, g
