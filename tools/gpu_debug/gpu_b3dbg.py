import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from roms_amd import hiplib, tiling
cs = bench.params_for(sys.argv[1]); cs["ninfo"] = 1
hiplib.kprof(1)
run = tiling.TiledRun(cs)
run.step(1); run.sync()
print("one step ok")
