import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
import bench
from roms_amd import tiling
cs = bench.params_for("benchmark1", ntimes=30); cs["ninfo"] = 1
ref = tiling.TiledRun(cs); ref.step(5); ref.sync()
d = ref.host.dims; t = ref.host.tile
names = ["rufrc", "rvfrc", "ru", "rv"]
want = {k: ref.gather(k).copy() for k in names}; ref.close()
run = tiling.TiledRun(cs, self_exchange=True, transport="peer"); run.step(5); run.sync()
for k in names:
    a, b = run.gather(k), want[k]
    diff = np.argwhere(a != b)
    print(k, a.shape, "ndiff", len(diff), "planes", sorted(set(diff[:, 0]))[:8], "j", sorted(set(diff[:, 1]))[:10], "i", sorted(set(diff[:, 2]))[:12], "LBi", d["LBi"], "LBj", d["LBj"])
run.close()
