"""GPU debugging aid: where does the pair form of the barotropic kernel differ from the per-call form?
usage: python tools/gpu_debug/gpu_pair_diff.py <workload> Lm Mm N nsteps"""
import os, subprocess, sys, textwrap
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
wl, Lm, Mm, N, nsteps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
names = ["zeta", "ubar", "vbar", "rzeta", "rubar", "rvbar", "Zt_avg1", "DU_avg1", "DU_avg2", "DV_avg1", "DV_avg2"]
code = textwrap.dedent("""
    import sys
    sys.path.insert(0, %r)
    import numpy as np
    import bench
    from roms_amd import tiling
    cs = bench.params_for(%r, %d, %d, %d)
    cs["ninfo"] = 1
    run = tiling.TiledRun(cs)
    run.step(%d)
    run.sync()
    t = run.host.tile
    np.savez(sys.argv[1], bounds=np.array([t["LBi"], t["UBi"], t["LBj"], t["UBj"]]), **{n: run.ctx.download(n) for n in %r})
    run.close()
""") % (ROOT, wl, Lm, Mm, N, nsteps, names)
forms = [("percall", {"ROMS_HIP_PAIR": "0"}), ("pair_a", {})]
for spec in sys.argv[6:]:          # extra forms: tag:ENV=val,ENV=val
    tag, _, envs = spec.partition(":")
    forms.append((tag, dict(e.split("=") for e in envs.split(",") if e)))
got = {}
for tag, env in forms:
    f = f"/tmp/pd_{tag}.npz"
    r = subprocess.run([sys.executable, "-c", code, f], capture_output=True, text=True, env=dict(os.environ, **env))
    if r.returncode:
        print(tag, "FAILED", r.stderr[-2000:])
        continue
    got[tag] = dict(np.load(f))
b = got["percall"]["bounds"]
ni, nj = b[1] - b[0] + 1, b[3] - b[2] + 1
for tag in got:
    if tag == "percall":
        continue
    for n in names:
        a, o = got[tag][n].reshape(-1, nj, ni), got["percall"][n].reshape(-1, nj, ni)
        if not np.array_equal(a, o):
            w = np.argwhere(a != o)
            print(tag, n, len(w), "levels", sorted(set(w[:, 0].tolist())), "i", w[:, 2].min() + b[0], w[:, 2].max() + b[0],
                  "j", w[:, 1].min() + b[2], w[:, 1].max() + b[2], "max", float(np.abs(a - o).max()),
                  "first", [(int(l), int(i + b[0]), int(j + b[2])) for l, j, i in w[:8]])
    print(tag, "compared")
