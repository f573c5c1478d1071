#!/usr/bin/env python3
"""Generate the golden fixtures from the reference's own Fortran (oracle/_ref).

Runs only in the build container (needs /root/reference and oracle/_ref/*.so built by
oracle/ref/build_ref.sh).  One configuration per process (the reference keeps its state in
Fortran modules), so this script re-invokes itself per case.

Fixtures (npz, float64 exact):
  <case>_init.npz     set-up tables and the state after the reference's initial sequence
                      (ana_grid, set_scoord, set_weights, metrics, ini_hmixcoef, set_depth,
                      ana_initial, set_depth0/set_zeta_timeavg/set_depth, set_massflux, rho_eos)
  <case>_kernels.npz  inputs and outputs of every reference kernel that builds here, run on a
                      perturbed state (the "pinned" kernels)
  bounds_*.npz        BOUNDS/DOMAIN tables of get_bounds.F for several tilings
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
EXTRA = {"benchmark": ["dmde", "dndx", "lonr", "latr", "rdrag2", "bvf", "alpha", "beta", "hsbl"]}
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
    app = "benchmark" if tag.startswith("benchmark") else "upwelling"
    cs = getattr(cases, app)(**kwargs)
    ip, rp = cases.ref_params(cs)
    saved = quiet()
    R = ref.Ref(app, ip, rp)
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


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--case":
        make_case(sys.argv[2])
    elif len(sys.argv) > 2 and sys.argv[1] == "--bounds":
        make_bounds()
    else:
        py = sys.executable
        for case in ["upwelling", "upwelling_small:Lm=14,Mm=18,N=8", "benchmark_small:Lm=24,Mm=16,N=10"]:
            subprocess.check_call([py, __file__, "--case", case])
        for spec in ["upwelling,41,80,1,1,HSIMT", "upwelling,41,80,2,2,HSIMT", "upwelling,41,80,2,4,U3",
                     "upwelling,41,80,3,3,U3", "benchmark,512,64,1,1,U3", "benchmark,512,64,2,2,U3",
                     "benchmark,2048,256,2,4,U3"]:
            subprocess.check_call([py, __file__, "--bounds", "-", spec])
