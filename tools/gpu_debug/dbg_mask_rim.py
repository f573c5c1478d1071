import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
import bench
from roms_amd import tiling
names = ["zeta", "ubar", "vbar", "u", "v", "t", "Hz", "DU_avg1", "Zt_avg1", "rzeta", "rubar", "rvbar"]
cs = bench.params_for("benchmark1_mask", ntimes=30); cs["ninfo"] = 1
ref = tiling.TiledRun(cs); ref.step(1); ref.sync()
want = {n: ref.gather(n).copy() for n in names}; ref.close()
os.environ["ROMS_HIP_LOOP"] = "0"; os.environ["ROMS_HIP_PAIR_RIM"] = "1"
run = tiling.TiledRun(cs, self_exchange=True, transport="peer"); run.step(1); run.sync()
for n in names:
    a, b = run.gather(n), want[n]
    d = np.argwhere(a != b)
    if len(d): print(n, a.shape, "ndiff", len(d), "planes", sorted(set(d[:,0]))[:4], "j", sorted(set(d[:,1]))[:12], "i", sorted(set(d[:,2]))[:16], "maxdiff", float(np.nanmax(np.abs(a-b))))
print("done")
run.close()
