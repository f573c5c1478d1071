"""Ad-hoc: per-kernel time per launch (synchronous HIP events) for a workload, optional advection schemes:
   python tools/gpu_debug/gpu_kbreak.py ns512 [steps] [U3,U3 C4,C4] [env K=V ...]"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import bench
from roms_amd import hiplib, tiling
wl = sys.argv[1] if len(sys.argv) > 1 else "ns512"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
kw = {}
if len(sys.argv) > 4:
    kw = dict(hadv=tuple(sys.argv[3].split(",")), vadv=tuple(sys.argv[4].split(",")))
app, Lm, Mm, N = bench.WORKLOADS[wl]
from tests import cases
if kw:
    fn = {"benchmark": cases.benchmark, "upwelling_kpp": cases.upwelling_kpp, "upwelling": cases.upwelling}[app]
    cs = fn(Lm=Lm, Mm=Mm, N=N, ntimes=n + 10, **kw)
else:
    cs = bench.params_for(wl, ntimes=n + 10)
cs["ninfo"] = 1
run = tiling.TiledRun(cs)
run.step(3); run.sync()
hiplib.kprof(1)
run.step(n); run.sync()
tab = hiplib.kprof_table()
hiplib.kprof(0)
rows = sorted(tab.items(), key=lambda kv: -kv[1][0])
tot = sum(v[0] for v in tab.values())
print("workload %s %s: sum of kernels %.0f us/step" % (wl, kw, 1e6 * tot / n))
for k, (s, c) in rows[:int(os.environ.get("KB_ROWS", "40"))]:
    print("  %-18s %8.1f us/launch x %5.1f /step = %8.0f us/step" % (k, 1e6 * s / max(c, 1), c / n, 1e6 * s / n))
import time
t0 = time.perf_counter(); run.step(n); run.sync(); t1 = time.perf_counter()
print("async step: %.3f ms" % (1e3 * (t1 - t0) / n))
run.close()
