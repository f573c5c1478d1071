"""The barotropic engines in a CLOSED basin (neither direction periodic; round 6: the corner averages by the thread of the
point next to the corner, k_haloblock.h:HB_CORNERS, so that the fused boundary stores -- and with them the pair launches without
a halo launch behind them and the persistent loop -- run there too): fields against the separate halo launches
(ROMS_HIP_FUSE_CLOSED=0), the pair launches and the per-call kernel, bit for bit, on a small ragged grid and at BENCHMARK1's
size, with and without land/sea masks; then the step time of each.
python tools/gpu_debug/closed_loop_check.py"""
import os, subprocess, sys, textwrap
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
code = textwrap.dedent("""
    import sys, time
    sys.path.insert(0, %r)
    import numpy as np
    import bench
    from roms_amd import tiling
    dims = tuple(int(x) for x in sys.argv[2].split(",")) if sys.argv[2] else ()
    cs = bench.params_for(sys.argv[3], *dims, ntimes=60)
    cs["EWperiodic"] = 0
    cs["ninfo"] = 1
    run = tiling.TiledRun(cs)
    run.step(int(sys.argv[4])); run.sync()
    names = ["zeta", "ubar", "vbar", "rzeta", "rubar", "rvbar", "Zt_avg1", "DU_avg1", "DU_avg2", "DV_avg1", "DV_avg2", "u", "v", "t", "W", "Hz", "rufrc"]
    np.savez(sys.argv[1], **{n: run.gather(n) for n in names})
    t0 = time.perf_counter(); run.step(40); run.sync(); t1 = time.perf_counter()
    print("MS", 1e3 * (t1 - t0) / 40)
    run.close()
""") % ROOT
import numpy as np
for wl in ("benchmark1", "benchmark1_mask"):
    for dims in ("200,44,10", ""):
        got = {}
        for tag, env in (("loop", {}), ("separate", {"ROMS_HIP_FUSE_CLOSED": "0"}), ("pair", {"ROMS_HIP_LOOP": "0"}), ("percall", {"ROMS_HIP_PAIR": "0"}),
                         ("percall_separate", {"ROMS_HIP_PAIR": "0", "ROMS_HIP_FUSE_CLOSED": "0"})):
            f = "/tmp/cl_%s.npz" % tag
            r = subprocess.run([sys.executable, "-c", code, f, dims, wl, os.environ.get("CL_STEPS", "4")], capture_output=True, text=True, env=dict(os.environ, ROMS_HIP_LOOP_TIMEOUT="0.2", **env), timeout=600)
            ms = [l for l in r.stdout.splitlines() if l.startswith("MS")]
            if not ms:
                print(tag, "FAILED", r.stdout[-500:], r.stderr[-1500:]); continue
            got[tag] = dict(np.load(f))
            print("CLOSEDLOOP", wl, dims or "full", tag, ms[-1], flush=True)
        ref = got.get("percall_separate")
        for tag in ("loop", "separate", "pair", "percall"):
            if tag in got and ref is not None:
                bad = [n for n in ref if not np.array_equal(ref[n], got[tag][n])]
                where = ""
                if bad:
                    d = np.abs(got[tag]["zeta"] - ref["zeta"])
                    d = d.reshape((-1,) + d.shape[-2:])
                    jj, ii = np.nonzero(d.max(axis=0))
                    where = "zeta differs at i(array) %s..%s j %s..%s, max %.3e" % (ii.min() if ii.size else None, ii.max() if ii.size else None, jj.min() if jj.size else None, jj.max() if jj.size else None, d.max())
                print("CLOSEDLOOP", wl, dims or "full", tag, "vs percall_separate: mismatching", bad, where, "finite", all(np.isfinite(got[tag][n]).all() for n in ref), "moving", float(np.abs(ref["u"]).max()))
