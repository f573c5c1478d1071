"""Ad-hoc: the mailbox transport on one GPU -- self-exchange parity (2 and 8 neighbours), then N ranks sharing the
device: python tools/gpu_debug/peer_probe.py"""
import json, os, subprocess, sys, tempfile
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from roms_amd import tiling
from tests import util

names = ["zeta", "ubar", "vbar", "u", "v", "t", "W", "Hz", "rho", "Akv", "Huon", "DU_avg1"]
cs = bench.params_for("benchmark1", 96, 32, 10)
cs["ninfo"] = 1
run = tiling.TiledRun(cs, self_exchange=True, transport="peer")
run.step(3); run.sync()
nx = run.ctx.L.roms_hip_exchange_count(run.ctx.h)
got = {n: run.ctx.download(n).copy() for n in names}
run.close()
ref = tiling.TiledRun(cs)
ref.step(3); ref.sync()
bad = [n for n in names if not np.array_equal(got[n], ref.ctx.download(n))]
ref.close()
print("PEER-SELF", nx, "mismatching", bad, flush=True)

tag, kw = "upwelling_small", {}
fields = ["zeta", "ubar", "vbar", "u", "v", "t", "W", "Hz", "Huon", "rho", "Akv", "DU_avg1"]
c2 = util.case_for(tag, **kw); c2["ninfo"] = 0
r1 = tiling.TiledRun(c2, weak=False); r1.step(4)
refg = {n: r1.gather(n) for n in fields}
r1.close()
for tiles, port in (((2, 1), 29611), ((2, 2), 29612)):
    out = os.path.join(tempfile.mkdtemp(), "t.npz")
    spec = dict(tag=tag, kw=kw, steps=4, tiles=list(tiles), fields=fields, gpu=True, transport="peer")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={tiles[0] * tiles[1]}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "mp", "run_tiles.py"), out, json.dumps(spec)]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=200, env=dict(os.environ, OMP_NUM_THREADS="1", ROMS_HIP_PEER_TIMEOUT="5"))
    if p.returncode:
        print("PEER-TILES", tiles, "FAILED", p.stdout[-1500:], p.stderr[-2500:], flush=True)
        continue
    g = dict(np.load(out))
    print("PEER-TILES", tiles, int(g["nexchanges"]), "mismatching", [n for n in fields if not np.array_equal(g[n], refg[n])], flush=True)
