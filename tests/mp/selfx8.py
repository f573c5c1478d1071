"""Doubly periodic single tile, every ghost exchange through the transport given on the command line (rccl: 8 sends
+ 8 receives to the same peer in one group; peer: the mailbox slots) with the tile as its own eight neighbours,
against the local periodic copies."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import numpy as np
import bench
from roms_amd import tiling
cs = bench.params_for("upwelling", 48, 40, 10)
cs["NSperiodic"] = 1
cs["ninfo"] = 1
names = ["zeta", "ubar", "vbar", "u", "v", "t", "W", "Hz", "Huon", "Hvom"]
run = tiling.TiledRun(cs, self_exchange=True, transport=sys.argv[1] if len(sys.argv) > 1 else "rccl")
run.step(3); run.sync()
nx = run.ctx.L.roms_hip_exchange_count(run.ctx.h)
got = {n: run.gather(n).copy() for n in names}     # (every point the reference defines, not the padding line)
run.close()
ref = tiling.TiledRun(cs)
ref.step(3); ref.sync()
bad = [n for n in names if not np.array_equal(got[n], ref.gather(n))]
fin = all(np.isfinite(got[n]).all() for n in names)
ref.close()
print("SELFX8 exchanges", nx, "finite", fin, "mismatching", bad)
