"""Ad-hoc: run the first step kernel by kernel and report the first kernel after which a field is not finite."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import numpy as np
import bench
from roms_amd import tiling
wl = sys.argv[1] if len(sys.argv) > 1 else "benchmark1"
cs = bench.params_for(wl)
cs["ninfo"] = 0
run = tiling.TiledRun(cs)
ctx = run.ctx
FIELDS = ["zeta", "ubar", "vbar", "u", "v", "t", "W", "Hz", "z_r", "z_w", "Huon", "Hvom", "rho", "Akv", "Akt", "ru", "rv", "rufrc",
          "rvfrc", "sustr", "svstr", "stflx", "bustr", "bvstr", "hsbl", "ghats", "rhoA", "rhoS", "Zt_avg1", "DU_avg1", "DU_avg2", "srflx", "bvf", "alpha", "beta", "lhflx", "shflx", "lrflx", "Uwind", "Tair", "Pair", "Hair", "cloud", "rain"]
def check(tag):
    bad = [n for n in FIELDS if not np.isfinite(ctx.download(n)).all()]
    print(f"{tag:14s} {'OK' if not bad else 'NONFINITE: ' + ','.join(bad)}", flush=True)
    return bad
check("start")
L, h = ctx.L, ctx.h
s = ctx.get_stepping()
ctx.set_stepping(nstp=1, nnew=2, nrhs=1)
seq = ["set_data", "ini_zeta", "set_depth", "ini_fields", "set_massflux", "rho_eos", "bulk_flux", "set_vbc", "lmd_vmix", "omega"]
for k in seq:
    ctx.call(k)
    if k == "set_depth":
        a, b = ctx.download("z_r"), run.host.get("z_r")
        d = np.argwhere(a != b)
        print("  z_r vs host: mismatches", len(d), "first", d[:3].ravel(), "n", a.size, flush=True)
        if len(d):
            ni = run.host.dims["UBi"] - run.host.dims["LBi"] + 1; nj = run.host.dims["UBj"] - run.host.dims["LBj"] + 1
            q = d[:, 0]; kk = q // (ni * nj); jj = (q % (ni * nj)) // ni; ii = q % ni
            print("  k range", kk.min(), kk.max(), "j range", jj.min(), jj.max(), "i range", ii.min(), ii.max(), flush=True)
    if check(k):
        sys.exit(0)
L.roms_hip_wvelocity(h, 1); check("wvelocity")
for k in ["set_zeta", "pre_step3d", "prsgrd", "t3dmix2", "rhs3d_tile", "uv3dmix2"]:
    ctx.call(k)
    if check(k):
        sys.exit(0)
ctx.set_stepping(iif=1, predictor=1, kstp=1, knew=3, krhs=1)
ctx.call("step2d")
check("step2d pred")
