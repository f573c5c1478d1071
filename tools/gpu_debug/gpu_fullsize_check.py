"""Ad-hoc (not collected by pytest): BENCHMARK1 at full size on the GPU against the oracle.
python tools/gpu_debug/gpu_fullsize_check.py [nsteps]"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import numpy as np
import bench
from tests import cases, util
from tests.test_host import HOST_FIELDS
from roms_amd import tiling
from oracle import orc

nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
wl = sys.argv[2] if len(sys.argv) > 2 else "benchmark1"
cs = bench.params_for(wl)
cs["ninfo"] = 1
run = tiling.TiledRun(cs)
H = run.host
w = np.stack([H.get("weight1"), H.get("weight2")])
O = orc.Oracle(cases.oracle_cfg(cs, H.reals["hc"], H.dims["nfast"], w))
for n in HOST_FIELDS:
    try:
        O.field(n)[:] = H.get(n)
    except KeyError:
        pass
O.start()
for s in range(nsteps):
    run.step(1)
    O.main3d_step(1)
    errs = {n: util.relrms(run.ctx.download(n), O.field(n)) for n in ["zeta", "ubar", "u", "v", "t", "W", "Akv", "Hz", "rho"]}
    worst = max(errs, key=errs.get)
    print(f"step {s+1}: worst {worst} {errs[worst]:.3e}  zeta {errs['zeta']:.2e} u {errs['u']:.2e} t {errs['t']:.2e}", flush=True)
