"""The persistent barotropic loop across REAL tile edges on one GPU: N processes share cuda:0 (ROMS_HIP_LOOP=1 forces the loop
although the ranks share a device: their kernels must all be resident at once -- BENCHMARK1's 256 sub-tiles split over the
ranks fill the 256 CUs exactly), the neighbour is another process's context (its own array origin, the periodic seam on one
side, an interior tile boundary on the other), gathered fields against the single-tile run.
python tools/gpu_debug/loop_2rank_shared.py [tiles e.g. 2x1] [workload] [steps]"""
import json, os, subprocess, sys
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
tiles = tuple(int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "2x1").split("x"))
wl = sys.argv[2] if len(sys.argv) > 2 else "benchmark1"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
import bench
from roms_amd import tiling
fields = ["zeta", "ubar", "vbar", "u", "v", "t", "W", "Hz", "Huon", "rho", "Akv", "DU_avg1", "Zt_avg1", "rubar"]
cs = bench.params_for(wl, ntimes=steps)
cs["ninfo"] = 0
run = tiling.TiledRun(cs, weak=False)
run.step(steps)
ref = {n: run.gather(n) for n in fields}
run.close()
out = "/tmp/loop_2rank.npz"
spec = dict(workload=wl, steps=steps, tiles=list(tiles), fields=fields, gpu=True, probe=True, transport="peer")
cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={tiles[0] * tiles[1]}", "--master-addr", "127.0.0.1",
       "--master-port", "29791", os.path.join(ROOT, "tests", "mp", "run_tiles.py"), out, json.dumps(spec)]
p = subprocess.run(cmd, capture_output=True, text=True, timeout=600,
                   env=dict(os.environ, OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0", ROMS_HIP_PEER_TIMEOUT="10", ROMS_HIP_LOOP="1",
                            ROMS_HIP_LOOP_TIMEOUT=os.environ.get("ROMS_HIP_LOOP_TIMEOUT", "1.0")))
if p.returncode != 0:
    print("LOOP2RANK", tiles, "FAILED", p.stdout[-800:], p.stderr[-2500:])
    sys.exit(1)
got = dict(np.load(out))
bad = [n for n in fields if not np.array_equal(got[n], ref[n])]
print("LOOP2RANK", tiles, wl, "exchanges/step", int(got["nexchanges_steps"]) / steps, "mismatching", bad)
