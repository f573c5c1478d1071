#!/usr/bin/env python3
"""bench.py -- grid-cell-updates/sec of the nonlinear 3-D time step (main3d) on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload benchmark1|benchmark2|benchmark3|ns512|ns512u3|upwelling|config5]

A "step" is one pass of main3d's STEP_LOOP (ROMS/Nonlinear/main3d.F:216-1148) with the full physics
of the application (BENCHMARK: nonlinear EOS, KPP, COARE bulk fluxes, geopotential tracer mixing,
nfast+1 LF-AM3 barotropic predictor/corrector pairs ...) and the state resident in HBM.  The default
workload is BASELINE.json configs[1]: BENCHMARK1 512x64x30.  For N > 1 the driver starts one rank per
GPU (torch.distributed, nccl = RCCL); each rank owns one tile of NtileI x NtileJ = N tiles.  The default
workload then runs BASELINE.json's own multi-GPU configurations: N = 4 -> BENCHMARK2 1024x128x30 in 2x2
(tile 512x64, the BENCHMARK1 grid: weak scaling), N = 8 -> BENCHMARK3 2048x256x30 in 2x4 (tile 1024x64),
N = 2 -> two BENCHMARK1 tiles side by side.

One JSON line is printed by rank 0 (contract in the task statement) with two extra objects:
  roofline      dominant kernel: algorithmic bytes / average launch duration measured with HIP events
                on the library's stream inside the timed region
  cpu_baseline  the C oracle (oracle/, a port of the reference's algorithm pinned against its object code)
                timed on ALL host cores (OpenMP threads over shared-memory tiles, the reference's
                shared-memory mode) on a bounded sample of the same workload
and two more: `north_star_pair` = the kernels of "step3d_t + rhs3d" (BASELINE.json north_star) timed in the
breakdown pass of this workload against their 632 algorithmic bytes per cell, and -- default run only --
`north_star_pair_512x512x50` = the same on the grid the north star names (a 6-step pass after the timed region), and
`north_star_pair_512x512x50_stock` = that grid with the tracer schemes of the shipped roms_upwelling.in (HSIMT salinity).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E ~8 TB/s

WORKLOADS = {
    # name: (app, Lm, Mm, N)   -- ROMS/External/roms_benchmark{1,2,3}.in, roms_upwelling.in
    "benchmark1": ("benchmark", 512, 64, 30),
    "benchmark2": ("benchmark", 1024, 128, 30),
    "benchmark3": ("benchmark", 2048, 256, 30),
    "benchmark1_mask": ("benchmark_mask", 512, 64, 30),   # BENCHMARK1 with the host's analytic land: cost of a MASKING run
    "benchmark1_closed": ("benchmark_closed", 512, 64, 30),            # ... as a closed basin (walls west and east instead of the periodic channel)
    "benchmark1_mask_closed": ("benchmark_mask_closed", 512, 64, 30),
    "ns512": ("upwelling", 512, 512, 50),       # north_star roofline size (512x512x50); UPWELLING keeps
                                                # 1 km cells at any size (BENCHMARK's shelf steepens with Mm)
    "ns512u3": ("upwelling_u3c4", 512, 512, 50),  # the same with U3/C4 advection for both tracers: the schemes
                                                # SURVEY.md 8(d) prices the north-star kernel pair on (79 words/cell)
    "upwelling": ("upwelling", 41, 80, 16),
    "config5": ("upwelling_kpp", 256, 512, 50),  # BASELINE configs[4]: UPWELLING + KPP + MPDATA
}

# Algorithmic HBM traffic per launch of each kernel, in "words per cell": the number of distinct full
# 3-D f64 arrays the launch must read or write once (read-modify-write = 2; NT=2 tracers; perfect reuse
# inside the kernel, none across kernels -- SURVEY.md 8(d)'s unit), plus the 2-D arrays per column.
# The per-kernel derivation is the table "Kernels and rooflines" in DESIGN.md.
ALGO_ARRAYS = {
    #                  3-D  2-D
    "k_step2d":       (0, 44),
    "k_step2d_pair":  (0, 88),     # predictor + corrector of one fast step in one launch (k_step2d_pair.h): two step2d
                                   # calls' worth of SURVEY 8(d)'s unit -- the 44 2-D words each call moves in the reference
                                   # -- although the fused kernel itself reads the state once
    "k_step2d_loop":  (0, 88),     # PER PAIR: the fast steps 2 .. nfast in one persistent launch (k_step2d_loop.h); algo_bytes
                                   # multiplies by the nfast - 1 pairs of a launch
    "k_pre_t3":       (10, 2),     # t(nstp), t(nnew) read and t(3) written per tracer; Hz, Huon, Hvom, W read ONCE for both
                                   # tracers (LDS-tiled form from 64 K columns; the point-wise form re-reads them: 15)
    "k_pre_t3h":      (7, 2),
    "k_pre_t3v":      (10, 2),
    "k_pre_new":      (19, 9),
    "k_prs_P":        (3, 0),
    "k_prs_grad":     (6, 2),
    "k_t3dmix2_s":    (7, 5),
    "k_t3dmix2_geo":  (8, 5),
    "k_uv3dmix2_s":   (11, 12),
    "k_uv3dmix2_sum": (4, 4),
    "k_uv3dmix2_col": (9, 12),     # large grids: uv3dmix2 + the coupling sums in one kernel: Hz, u, v, ru, rv read,
                                   # u, v(nnew) read-modify-write; no work arrays
    "k_rhs3d_pt":     (10, 3),     # u, v, Huon, Hvom, W, Hz read; ru, rv read-modify-write
    "k_rhs3d_sum":    (6, 10),     # ru, rv and the four viscous terms of uv3dmix2 (fused main3d sequence)
    "k_s3uv_col":     (8, 6),
    "k_s3uv_couple":  (9, 8),
    "k_s3t_hv":       (10, 2),     # t(3) read, t(nnew) read-modify-write per tracer; Huon, Hvom, W, Hz once (LDS-tiled form;
                                   # the point-wise form re-reads them per tracer: 12)
    "k_s3t_h":        (8, 2),
    "k_s3t_col":      (11, 2),
    "k_omega":        (4, 0),
    "k_eos_nl":       (8, 4),
    "k_rho_eos_lin":  (5, 2),
    "k_lmd_interior": (9, 0),
    "k_lmd_skpp":     (18, 10),    # incl. the convective adjustment of lmd_finish (Akv, Akt read-modify-write)
    "k_lmd_col":      (21, 12),    # N <= 41: lmd_vmix as one column kernel, spline columns in LDS
    "k_lmd_fused":    (21, 12),    # the same column function with its three work columns in 3-D work arrays
    "k_set_depth":    (3, 2),
    "k_set_massflux": (5, 2),
    "k_diag_col":     (7, 3),
    "k_wvel":         (6, 4),      # fused wvelocity: u, v, z_r, W, z_w read, wvel written
    # MPDATA (BASELINE config 5), PER TRACER -- the launch sequence runs once per MPDATA tracer (g_step3d.cpp); SURVEY 8(a)
    # a10 prices the first-order step + corrected fluxes at +16 words, a11 (mpdata_adiff) at about 10 per tracer:
    "k_mp_ta":        (7, 2),      # t(3), t(nnew), Huon, Hvom, W, Hz read; Ta written
    "k_mp_uva":       (7, 4),      # Ta, Huon, Hvom, W, Hz read; Ua, Va written (both directions in one launch)
    "k_mp_wa":        (6, 2),      # Ta, Huon, Hvom, W, Hz read; Wa written
    "k_mp_beta":      (8, 2),      # Ta, t(3), Ua, Va, Wa, Hz read; beta_up, beta_dn written
    "k_mp_apply":     (9, 3),      # (k_mp_limapply) Ta, Ua, Va, Wa, beta_up, beta_dn, Hz, z_r read; t(nnew) written
    "k_mp_vdiff":     (4, 1),      # t(nnew) read and written, Hz, Akt read (plain implicit vertical diffusion :1724-1790)
}
# what the FUSED barotropic pair kernel itself must move once per launch: 2-D words.  Read: zeta, ubar,
# vbar at two levels (6), h, pm, pn, on_u, om_v (5), rhoA, rhoS (2), the five fast-time averages (5), rzeta/rubar/rvbar of
# the older level (3), rufrc, rvfrc (2), the 13 metric arrays of the momentum stage; written: the five averages, the
# r.h.s. level krhs (3), the staged corrector result (3), the committed level (3) -- 50 words against the 88 of two
# step2d calls in SURVEY 8(d)'s per-call unit.  roofline.achieved keeps SURVEY's unit; roofline.achieved_fused this one.
FUSED_2D_WORDS = {"k_step2d_pair": 50}
# ... and the persistent loop per LAUNCH: what the pair kernel reads, once (36 words: the state at two levels, the static
# fields, the averages, the older r.h.s. level, the forcing, the 13 metric arrays) + what it leaves behind (23: the averages,
# the r.h.s. of both levels, the two logical levels, level 3, the staged result) + per pair but the last the rim exchange: 3
# words of own points written, (26 x 18 - 128) / 128 x 3 = 8 words' worth of rim points read
def loop_fused_words(pairs):
    return 36.0 + 23.0 + 11.0 * max(pairs - 1, 0)


def loop_pairs(table, nfast):
    """predictor + corrector pairs one launch of the persistent loop covers: nfast - 1 (fast steps 2 .. nfast), or -- when
    the per-call kernel k_step2d does not run at all -- nfast + 1/2: the first fast step and the auxiliary call
    iif = nfast+1 (half a pair: a predictor call without a corrector) are inside the launch too"""
    percall = table.get("k_step2d", (0.0, 0))[1] if table else 0
    return nfast - 1 if percall > 0 else nfast + 0.5

# North-star kernel pair "step3d_t + rhs3d" (BASELINE.json north_star; SURVEY.md 8(d): rows a4-a8 + a10 =
# pre_step3d, prsgrd, t3dmix2, rhs3d_tile, uv3dmix2, step3d_t): 79 words = 632 bytes per cell for U3/C4
# advection and NT = 2.  The kernels of those rows (launch sequences g_rhs3d.cpp, g_step3d.cpp:run_step3d_t):
PAIR_KERNELS = ("k_swdk", "k_pre_t3", "k_pre_t3h", "k_pre_t3v", "k_pre_new", "k_prs_P", "k_prs_grad", "k_t3dmix2_s",
                "k_t3dmix2_geo", "k_rhs3d_pt", "k_rhs3d_sum", "k_uv3dmix2_s", "k_uv3dmix2_sum", "k_uv3dmix2_col", "k_s3t_hv", "k_s3t_h",
                "k_s3t_col", "k_mp_ta", "k_mp_uva", "k_mp_wa", "k_mp_beta", "k_mp_limit", "k_mp_apply", "k_mp_vdiff")
PAIR_BYTES_PER_CELL = 632.0


def pair_report(table, steps, cells):
    """Time of the north-star kernel pair per step from the per-kernel breakdown pass (synchronous HIP
    events on the library's stream, `steps` steps) against its algorithmic bytes."""
    us = sum(1e6 * table[k][0] for k in PAIR_KERNELS if k in table) / max(steps, 1)
    if us <= 0.0:
        return None
    gbs = PAIR_BYTES_PER_CELL * cells / (us * 1e-6) / 1e9
    return {"rows": "SURVEY 8(a) a4-a8 + a10 (pre_step3d, prsgrd, t3dmix2, rhs3d_tile, uv3dmix2, step3d_t)",
            "kernels": [k for k in PAIR_KERNELS if k in table], "us_per_step": us,
            "algorithmic_bytes_per_cell": PAIR_BYTES_PER_CELL, "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": gbs / HBM_PEAK_GBS,
            "note": "632 B/cell is priced for U3/C4 advection and NT=2; HSIMT/MPDATA tracers and the halo launches of "
                    "those rows (not counted here) add work"}


def whole_step_bytes_per_cell(cs, nfast):
    """SURVEY.md 8(d)'s unfused-kernel model of one baroclinic step, bytes per cell: 166 3-D words with BENCHMARK
    physics (KPP, bulk fluxes, nonlinear EOS, geopotential mixing) or 127 with UPWELLING's, plus the barotropic
    engine's 48 2-D passes per step2d call x (2 nfast + 1) calls spread over N levels."""
    words3d = 166.0 if cs["app"].startswith("benchmark") else 127.0
    if cs["app"] == "upwelling_kpp":         # BASELINE config 5: + lmd_vmix/skpp/finish (34), MPDATA (about 32 per tracer)
        words3d += 34.0 + 64.0
    return 8.0 * (words3d + 48.0 * (2 * nfast + 1) / cs["N"])


def algo_bytes(kernel, Lm, Mm, N, launches_per_step_hint=None, pairs=1):
    if kernel not in ALGO_ARRAYS:
        return None
    a3, a2 = ALGO_ARRAYS[kernel]
    P = Lm * Mm
    return 8.0 * P * (a3 * N + a2) * (pairs if kernel == "k_step2d_loop" else 1)


def pmc_traffic(workload, kernel, world):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/*_traffic.json,
    FETCH_SIZE + WRITE_SIZE passes corrected with the calibration factors measured on k_copy_probe, as
    the MI355X guide prescribes).  None if no such profile of this workload is committed (round 1:
    see DESIGN.md 6.4)."""
    import glob
    if world != 1:
        return None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")), reverse=True):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        if d.get("workload") != workload:
            continue
        ks = d.get("kernels", {})
        # (the profiler lists the instantiation that ran -- k_step2d_loop_b -- the library's timers the family name)
        for name in [kernel] + sorted(k for k in ks if k.startswith(kernel + "_")):
            if name in ks:
                pmc_traffic.source = os.path.relpath(f, ROOT)
                return ks[name].get("hbm_bytes_per_launch")
        return None
    return None


pmc_traffic.source = None


def params_for(workload, Lm=None, Mm=None, N=None, ntimes=10):
    from roms_amd import cases
    app, lm, mm, n = WORKLOADS[workload]
    Lm, Mm, N = Lm or lm, Mm or mm, N or n
    if app.endswith("_closed"):
        cs = params_for({"benchmark_closed": "benchmark1", "benchmark_mask_closed": "benchmark1_mask"}[app], Lm, Mm, N, ntimes)
        cs["EWperiodic"] = 0
    elif app == "benchmark":
        cs = cases.benchmark(Lm=Lm, Mm=Mm, N=N, ntimes=ntimes)
    elif app == "benchmark_mask":       # BENCHMARK with the host's analytic land (MASKING)
        cs = cases.benchmark_mask(Lm=Lm, Mm=Mm, N=N, ntimes=ntimes)
    elif app == "upwelling_kpp":
        cs = cases.upwelling_kpp(Lm=Lm, Mm=Mm, N=N, ntimes=ntimes)
    elif app == "upwelling_u3c4":
        cs = cases.upwelling(Lm=Lm, Mm=Mm, N=N, ntimes=ntimes, hadv=("U3", "U3"), vadv=("C4", "C4"))
    else:
        cs = cases.upwelling(Lm=Lm, Mm=Mm, N=N, ntimes=ntimes)
    return cs


def usable_cores():
    """CPUs this process may really use: the affinity mask, capped by the cgroup CPU quota (a container can
    show 256 CPUs and be allowed 16 of them; OpenMP teams sized by the mask then crawl)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def _cpu_replica(args):
    """One replica of the cpu_baseline workload: the oracle with `threads` OpenMP threads over eta strips."""
    cs, setup, threads, strips, nsteps, go, done = args
    import numpy as np
    from oracle import orc
    from tests import cases as tcases
    c2 = dict(cs, NtileI=1, NtileJ=strips)
    O = orc.Oracle(tcases.oracle_cfg(c2, setup["hc"], setup["nfast"], setup["weight"]))
    for n, a in setup["fields"].items():
        try:
            O.field(n)[:] = a
        except KeyError:
            pass
    O.set_threads(threads)
    O.start()
    O.main3d_step(1)                       # first step: start-up branches
    done.put("ready")
    go.wait()
    t0 = time.perf_counter()
    O.main3d_step(nsteps)
    t1 = time.perf_counter()
    O.close()
    done.put((t0, t1))


def cpu_baseline(cs, H, budget_s=15.0):
    """The reference's algorithm on ALL host cores of this box.  The C oracle (oracle/, a port pinned bit for
    bit against the reference's object code) runs its tile loops as OpenMP threads -- the reference's
    shared-memory mode (Drivers/nl_roms.h:304-310) -- over NtileJ strips along eta (NtileI = 1: with tiles
    along a PERIODIC axis the reference's shared-memory exchange reads neighbour tiles inside the same parallel
    loop and the result depends on thread timing; strips keep every periodic copy inside one tile, and the
    threaded run is bit-identical to the serial one, tests/test_oracle.py::test_threaded_tiles_bitwise).  A grid
    with Mm rows feeds at most Mm/4 threads, so the remaining cores run further replicas of the same workload
    at the same time: value = replicas x cells x steps / wall time of the slowest replica.  This is the ONLY
    place bench.py touches oracle/."""
    import multiprocessing as mp
    import queue as _queue
    import numpy as np
    os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")   # idle OpenMP threads sleep: replicas share the cores
    from oracle import orc
    from roms_amd.hostlib import HOST_FIELDS
    orc.build()
    cores = usable_cores()
    strips = max(1, min(cs["Mm"] // 4, cores))             # >= 4 rows per strip
    threads = strips
    replicas = max(1, cores // threads)
    setup = {"hc": H.reals["hc"], "nfast": H.dims["nfast"],
             "weight": np.stack([H.get("weight1"), H.get("weight2")]), "fields": {}}
    for n in HOST_FIELDS:
        try:
            setup["fields"][n] = H.get(n)
        except KeyError:
            pass
    # size the sample on one replica alone, then run all of them together
    ctx = mp.get_context("fork")
    cells = cs["Lm"] * cs["Mm"] * cs["N"]
    go, done = ctx.Event(), ctx.Queue()
    probe = ctx.Process(target=_cpu_replica, args=((cs, setup, threads, strips, 2, go, done),))
    probe.start()
    try:
        done.get(timeout=120)
        go.set()
        a, b = done.get(timeout=120)
    except _queue.Empty:               # never block the benchmark on its reported baseline
        probe.terminate()
        return {"value": None, "unit": "grid-cell-updates/sec", "cores": threads, "kind": "port",
                "sample": "the oracle did not finish 3 steps in 120 s on this host: no CPU baseline"}
    probe.join()
    per_step = (b - a) / 2
    solo = {"value": cells / per_step, "unit": "grid-cell-updates/sec", "cores": threads, "kind": "port",
            "sample": f"{cs['app'].upper()} {cs['Lm']}x{cs['Mm']}x{cs['N']}, 2 steps after the first, oracle/liborc.so "
                      f"(gcc -O2 -fopenmp), {threads} OpenMP threads over 1x{strips} shared-memory tiles "
                      f"({cores} cores usable), {b - a:.1f} s"}
    if replicas == 1 and per_step * 2 >= 5.0:
        return solo
    nsteps = max(2, min(400, int(budget_s / max(per_step * 1.5, 1e-3))))
    go, done = ctx.Event(), ctx.Queue()
    procs = [ctx.Process(target=_cpu_replica, args=((cs, setup, threads, strips, nsteps, go, done),))
             for _ in range(replicas)]
    for p in procs:
        p.start()
    try:
        for _ in procs:
            done.get(timeout=180)
        go.set()
        spans = [done.get(timeout=max(60.0, 6.0 * budget_s)) for _ in procs]
    except _queue.Empty:               # oversubscribed or throttled host: report the one replica that was timed
        for p in procs:
            p.terminate()
        solo["sample"] += f"; {replicas} concurrent replicas did not finish in time and were stopped"
        return solo
    for p in procs:
        p.join()
    wall = max(t[1] for t in spans) - min(t[0] for t in spans)
    return {"value": replicas * cells * nsteps / wall, "unit": "grid-cell-updates/sec", "cores": replicas * threads,
            "kind": "port",
            "sample": f"{cs['app'].upper()} {cs['Lm']}x{cs['Mm']}x{cs['N']}, {nsteps} steps after the first, "
                      f"oracle/liborc.so (gcc -O2 -fopenmp): {replicas} concurrent replicas x {threads} OpenMP threads "
                      f"over 1x{strips} shared-memory tiles each ({cores} cores available), {wall:.1f} s; one replica "
                      f"alone: {cells / per_step:.3g} cell-updates/s"}


def reference_leg(workload, cs, budget_s=12.0):
    """The reference's OWN object code (oracle/_ref/libromsref_<app>.so: the reference's Fortran where it lies, built by
    oracle/ref/build_ref.sh in the build container; it travels to the GPU box as a built library, its sources do not) on ONE
    host core: main3d steps 2..n of the same workload, n sized to `budget_s` from the time of step 1.  In a process of its
    own (the reference keeps its state in Fortran modules).  None when the library is absent."""
    import subprocess
    app = {"benchmark": "benchmark", "upwelling_kpp": "upwelling_kpp"}.get(cs["app"], "upwelling")
    root = os.path.dirname(os.path.abspath(__file__))
    if not os.path.exists(os.path.join(root, "oracle", "_ref", f"libromsref_{app}.so")):
        return None
    if not os.path.exists("/root/reference/ROMS/External/varinfo.yaml"):
        # the reference reads its variable-metadata table at start-up (mod_ncparam.F:initialize_ncparam, the file named by
        # VARNAME); the GPU box has no reference tree, so its library cannot be started there.  What the same library measured
        # in the build container is recorded with the reference-written sample of this workload (tests/golden/make_golden.py
        # --sample): reported as such, not as a timing of this machine
        try:
            import numpy as _np
            meta = json.loads(str(_np.load(os.path.join(root, "tests", "golden", f"{workload}_sample.npz"))["meta"]))
            c = meta["case"]
            if (c["Lm"], c["Mm"], c["N"]) != (cs["Lm"], cs["Mm"], cs["N"]):      # (an --Lm/--Mm/--N override: the record is of another grid)
                return {"value": None, "kind": "reference", "measured_here": False, "source": "none",
                        "sample": "no reference tree on this machine and the recorded rate is of another grid size"}
            v = c["Lm"] * c["Mm"] * c["N"] * (meta["nsteps"] - 1) / meta["ref_steps_s"]
            return {"value": v, "unit": "grid-cell-updates/sec", "cores": 1, "kind": "reference", "measured_here": False, "source": "recorded",
                    "sample": f"{cs['app'].upper()} {c['Lm']}x{c['Mm']}x{c['N']}, main3d steps 2..{meta['nsteps']} of the reference's own object code, "
                              "RECORDED in the build container (tests/golden/%s_sample.npz), not timed on this machine: the reference "
                              "library reads ROMS/External/varinfo.yaml of its source tree when it starts, and this machine has no reference tree" % workload}
        except Exception as e:      # noqa: BLE001
            return {"value": None, "kind": "reference", "sample": f"no reference tree on this machine (the library reads its varinfo.yaml at start-up) and no recorded rate: {e}"}
    code = (
        "import sys, json, time, threading\n"
        f"sys.path.insert(0, {root!r})\n"
        "def run():\n"
        "    import bench\n"
        "    from tests import refdrive as rd\n"
        f"    cs = bench.params_for({workload!r}, {cs['Lm']}, {cs['Mm']}, {cs['N']}, ntimes=400)\n"
        "    saved = rd.quiet()\n"
        f"    R = rd.reference({app!r}, cs)\n"
        "    t0 = time.perf_counter(); R.main3d(1); t1 = time.perf_counter()\n"
        f"    n = max(2, min(200, int({budget_s} / max(t1 - t0, 1e-4))))\n"
        "    R.main3d(n); t2 = time.perf_counter()\n"
        "    rd.unquiet(saved)\n"
        "    print(json.dumps(dict(n=n, first=t1 - t0, span=t2 - t1)), flush=True)\n"
        "# the reference keeps its private (IminS:ImaxS,JminS:JmaxS,N) work arrays on the stack: a thread with a 1 GiB stack of its\n"
        "# own (an ordinary user cannot raise RLIMIT_STACK beyond the hard limit, and the main thread's stack is fixed at start)\n"
        "threading.stack_size(1 << 30)\n"
        "t = threading.Thread(target=run); t.start(); t.join()\n")
    env = dict(os.environ, OMP_NUM_THREADS="1")

    def big_stack():
        # the reference keeps its private (IminS:ImaxS,JminS:JmaxS,N) work arrays on the stack, and the main thread's stack is
        # sized when the program starts: raise the limit between fork and the start of the child (to the hard limit where an
        # ordinary user may not go further)
        import resource
        soft, hard = resource.getrlimit(resource.RLIMIT_STACK)
        try:
            resource.setrlimit(resource.RLIMIT_STACK, (resource.RLIM_INFINITY, resource.RLIM_INFINITY))
        except (ValueError, OSError):
            resource.setrlimit(resource.RLIMIT_STACK, (hard, hard))

    try:
        p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env, preexec_fn=big_stack)
        lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
        if p.returncode != 0 or not lines:
            return {"value": None, "kind": "reference", "sample": "the reference library did not run here: " +
                    (p.stderr.strip().splitlines() or [f"exit code {p.returncode}"])[-1][:200]}
        r = json.loads(lines[-1])
    except (subprocess.TimeoutExpired, ValueError) as e:
        return {"value": None, "kind": "reference", "sample": f"the reference library did not run here: {e}"}
    cells = cs["Lm"] * cs["Mm"] * cs["N"]
    return {"value": cells * r["n"] / r["span"], "unit": "grid-cell-updates/sec", "cores": 1, "kind": "reference", "measured_here": True, "source": "timed",
            "sample": f"{cs['app'].upper()} {cs['Lm']}x{cs['Mm']}x{cs['N']}, main3d steps 2..{r['n'] + 1} of the reference's own "
                      f"object code (oracle/_ref/libromsref_{app}.so, amdflang -O2, serial), {r['span']:.1f} s"}


def multi_gpu_plan(world, workload, explicit_dims):
    """(workload, (NtileI, NtileJ) or None, weak) of a run on `world` GPUs.  The default workload on 2/4/8 GPUs is
    BASELINE.json's own multi-GPU configuration: BENCHMARK2 1024x128x30 in NtileI x NtileJ = 2x2 on 4 GPUs (tile
    512x64, the BENCHMARK1 grid: weak scaling) and BENCHMARK3 2048x256x30 in 2x4 on 8 (tile 1024x64); on 2 GPUs two
    BENCHMARK1 tiles side by side (1024x64 in 2x1).  Any other workload / GPU count / explicit --Lm --Mm --N: weak
    scaling of the named grid over roms_amd.tiling.partition(world) (weak = True: the named grid is the tile)."""
    baseline_multi = {2: ("benchmark1", (2, 1), True), 4: ("benchmark2", (2, 2), False),
                      8: ("benchmark3", (2, 4), False)}
    if world in baseline_multi and workload == "benchmark1" and not explicit_dims:
        return baseline_multi[world]
    return workload, None, True


def north_star_pass(hiplib, tiling, device, steps=6, warmup=24, workload="ns512u3"):
    """UPWELLING 512x512x50 with U3/C4 advection (workload ns512u3), or with the schemes of the shipped roms_upwelling.in
    (ns512: HSIMT salinity): `steps` steps with synchronous per-kernel HIP events; the kernels of "step3d_t + rhs3d"
    against their 632 algorithmic bytes per cell, and each of them alone.  (A single-tile run stores the boundary values
    and periodic images inside the producing kernels: those rows have no separate halo launches to add.)"""
    cs = params_for(workload, ntimes=steps + warmup)
    cs["ninfo"] = 1
    run = tiling.TiledRun(cs, device=device)
    run.step(warmup)                       # past the start-up branches (iic <= 2) and the clock ramp after the CPU leg
    run.sync()
    # kprof mode 4: every launch carries the start / stop events hipExtLaunchKernel fills from the dispatch's own
    # timestamps, one stream -- the per-kernel durations a serial `rocprofv3 --kernel-trace --stats` run reports
    # (profiles/r04_ns512*_serial_kernel_stats.csv); rounds 1-3 bracketed every launch with marker events and a host
    # synchronisation, which read 4 % higher
    hiplib.kprof(4)
    run.step(steps)
    run.sync()
    table = hiplib.kprof_table()
    hiplib.kprof(0)
    cells = cs["Lm"] * cs["Mm"] * cs["N"]
    rep = pair_report(table, steps, cells)
    per = {}
    for k in rep["kernels"]:
        nb = algo_bytes(k, cs["Lm"], cs["Mm"], cs["N"])
        us = 1e6 * table[k][0] / max(table[k][1], 1)
        per[k] = {"us_per_launch": us, "launches_per_step": table[k][1] / steps,
                  "algorithmic_GBs": (nb / us / 1e3) if nb else None}
    rep["per_kernel"] = per
    rep["workload"] = ("UPWELLING 512x512x50, U3/C4 advection of both tracers (bench.py --workload ns512u3)" if workload == "ns512u3" else
                       "UPWELLING 512x512x50, the schemes of roms_upwelling.in: U3/C4 temperature, HSIMT salinity (bench.py --workload ns512); "
                       "priced on the same 632 B/cell, which HSIMT's extra passes exceed")
    run.check()
    run.close()
    return rep


def whole_step_pass(hiplib, tiling, device, workload, steps=8, warmup=6, vary_salt=False):
    """One of the other BASELINE configurations that fit one GPU, on one tile: `steps` steps timed from the host
    (synchronised on both sides) -> ms per step, cell-updates/s and the whole-step fraction of the HBM peak on SURVEY
    8(d)'s bytes per cell-update; then two steps with every launch timed (the dispatch's own begin / end) -> the
    dominant kernel of that configuration, its time per launch and its fraction on its algorithmic bytes."""
    cs = params_for(workload, ntimes=steps + warmup + 2)
    cs["ninfo"] = 1
    run = tiling.TiledRun(cs, device=device)
    if vary_salt:
        # UPWELLING's analytic salinity is the constant 35: mpdata_adiff's anti-diffusive velocities of a constant tracer are
        # zero and its kernels take their early exits.  The honest MPDATA cost needs a second tracer that varies:
        # S = 35 + 0.05 (T - 14) before the first step (tools/gpu_debug/mp_salt_cost.py)
        t = run.ctx.download("t")
        tt = t.reshape(2, 3, -1)
        tt[1, :, :] = 35.0 + 0.05 * (tt[0, :, :] - 14.0)
        run.ctx.upload("t", tt.reshape(t.shape))
    run.step(warmup)
    run.sync()
    t0 = time.perf_counter()
    run.step(steps)
    run.sync()
    elapsed = time.perf_counter() - t0
    cells = cs["Lm"] * cs["Mm"] * cs["N"]
    wsb = whole_step_bytes_per_cell(cs, run.nfast)
    rep = {"workload": f"{cs['app'].upper()} {cs['Lm']}x{cs['Mm']}x{cs['N']} on one tile (bench.py --workload {workload})" +
                       (", salinity = 35 + 0.05 (T - 14) at the start instead of the constant 35" if vary_salt else ""),
           "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * elapsed / steps, "value": cells * steps / elapsed,
           "unit": "grid-cell-updates/sec", "whole_step_bytes_per_cell": wsb,
           "whole_step_frac": wsb * cells * steps / elapsed / 1e9 / HBM_PEAK_GBS}
    hiplib.kprof(4)
    run.step(2)
    run.sync()
    table = hiplib.kprof_table()
    hiplib.kprof(0)
    ranked = sorted(((k, v) for k, v in table.items() if k in ALGO_ARRAYS), key=lambda kv: -kv[1][0])
    if ranked:
        k, (sec, n) = ranked[0]
        pairs = loop_pairs(table, run.nfast)
        nb = algo_bytes(k, cs["Lm"], cs["Mm"], cs["N"], pairs=pairs)
        fused = loop_fused_words(pairs) if k == "k_step2d_loop" else FUSED_2D_WORDS.get(k)
        fb = 8.0 * cs["Lm"] * cs["Mm"] * fused if fused else nb
        us = 1e6 * sec / max(n, 1)
        rep["dominant"] = {"kernel": k, "us_per_launch": us, "launches_per_step": n / 2.0, "us_per_step": 1e6 * sec / 2.0,
                           "share_of_step": (sec / 2.0) / (elapsed / steps),
                           "frac": fb / us / 1e3 / HBM_PEAK_GBS, "frac_survey_unit": nb / us / 1e3 / HBM_PEAK_GBS}
    run.check()
    run.close()
    return rep


def tiled_form_pass(tiling, device, steps=20, warmup=6):
    """The MULTI-TILE form of the headline step on one GPU: BENCHMARK1 512x64x30 with the tile as its own western and eastern
    neighbour (ROMS_HIP_SELF_EXCHANGE: every periodic ghost line travels through the halo transport instead of a local copy),
    once through the mailbox -- the persistent barotropic loop handing its rim across the tile edge inside the launch, the
    schedule around it -- and once through RCCL send/recv groups (the per-pair launches: a collective library cannot be
    called from inside a kernel).  Every exchange is device-local here: the figures are the cost of the tiled code path, not
    of xGMI.  ms per step (host-timed, synchronised on both sides) and exchange points per step."""
    out = {}
    for tr in ("peer", "rccl"):
        cs = params_for("benchmark1", ntimes=steps + warmup + 2)
        cs["ninfo"] = 1
        try:
            run = tiling.TiledRun(cs, device=device, self_exchange=True, transport=tr)
            run.step(warmup)
            run.sync()
            x0 = run.ctx.L.roms_hip_exchange_count(run.ctx.h)
            t0 = time.perf_counter()
            run.step(steps)
            run.sync()
            dt = time.perf_counter() - t0
            x1 = run.ctx.L.roms_hip_exchange_count(run.ctx.h)
            out["mailbox" if tr == "peer" else "rccl"] = {"ms_per_step": 1e3 * dt / steps, "exchanges_per_step": (x1 - x0) / steps,
                                                          "rccl_ranks": run.rccl_ranks()}
            run.check()
            run.close()
        except Exception as e:       # (a reported aid, never a reason to lose the headline line)
            out["mailbox" if tr == "peer" else "rccl"] = {"error": str(e)[:200]}
    out["workload"] = "BENCHMARK1 512x64x30, one tile that is its own W/E neighbour through the halo transport (one GPU)"
    # ... and the tile of BASELINE's 8-GPU configuration (BENCHMARK3 2048x256x30 in 2x4: 1024x64x30 per GPU): too many sub-tiles
    # for the persistent loop, so the pair launches hand their rim across the tile edge themselves (k_step2d_pair.h)
    try:
        cs = params_for("benchmark1", 1024, 64, 30, ntimes=steps + warmup + 2)
        cs["ninfo"] = 1
        one = tiling.TiledRun(cs, device=device)
        one.step(warmup); one.sync()
        t0 = time.perf_counter(); one.step(steps); one.sync(); t_one = time.perf_counter() - t0
        one.close()
        run = tiling.TiledRun(cs, device=device, self_exchange=True, transport="peer")
        run.step(warmup); run.sync()
        x0 = run.ctx.L.roms_hip_exchange_count(run.ctx.h)
        t0 = time.perf_counter(); run.step(steps); run.sync(); dt = time.perf_counter() - t0
        x1 = run.ctx.L.roms_hip_exchange_count(run.ctx.h)
        run.check(); run.close()
        out["mailbox_1024x64x30"] = {"ms_per_step": 1e3 * dt / steps, "exchanges_per_step": (x1 - x0) / steps, "single_tile_ms_per_step": 1e3 * t_one / steps}
    except Exception as e:
        out["mailbox_1024x64x30"] = {"error": str(e)[:200]}
    return out


def self_launch(ngpus):
    """`python bench.py --gpus N` without torch.distributed.run: start the N ranks as CHILD processes (one per GPU,
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same arguments>`), relay their
    output and return the exit code.  Nothing in this (parent) process has initialised the GPU."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ngpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, ROMS_BENCH_SELF_LAUNCHED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    print("bench.py: starting " + " ".join(cmd), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="benchmark1", choices=sorted(WORKLOADS))
    ap.add_argument("--Lm", type=int)
    ap.add_argument("--Mm", type=int)
    ap.add_argument("--N", type=int)
    ap.add_argument("--averages", type=int, default=0, metavar="NAVG",
                    help="measurement aid: time-average the 22 default Aout fields over windows of NAVG steps "
                         "(AVERAGES of the stock upwelling.h; off in the headline run, as in roms_benchmark*.in)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-leg", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-breakdown", action="store_true")
    ap.add_argument("--no-north-star", action="store_true", help="skip the 512x512x50 pass of the default run")
    ap.add_argument("--breakdown-file", default=None, help="write the per-kernel table (JSON) here")
    ap.add_argument("--share-gpu", action="store_true",
                    help="test aid: all ranks of a --gpus N run on device 0, halo strips staged through the host (gloo)")
    ap.add_argument("--dry-launch", action="store_true",
                    help="launcher check: start the ranks, join a gloo group, report, stop (needs no GPU)")
    ap.add_argument("--transport", default=None, choices=["auto", "peer", "rccl", "dist_staged"],
                    help="halo transport of a multi-GPU run (default: ROMS_HIP_TRANSPORT or the library's default)")
    ap.add_argument("--copy-probe", action="store_true",
                    help="also time the library's streaming-copy kernel on this workload's 3-D arrays")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1 and "WORLD_SIZE" not in os.environ:
            # started plainly (`python bench.py --gpus N`): this process only LAUNCHES -- one rank per GPU as child
            # processes of torch.distributed.run, before anything here has touched the GPU (never an exec) -- and
            # relays rank 0's JSON line and the exit code
            raise SystemExit(self_launch(args.gpus))
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    if args.dry_launch:
        # launcher / rendezvous check (runs without a GPU): every rank joins a gloo group, rank 0 reports
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import torch
        t = torch.tensor([rank + 1], dtype=torch.int64)
        dist.all_reduce(t)
        dist.barrier()
        if rank == 0:
            print(json.dumps({"launch": "ok", "n_gpus": world, "rank_sum": int(t.item()),
                              "self_launched": os.environ.get("ROMS_BENCH_SELF_LAUNCHED") == "1"}), flush=True)
        dist.destroy_process_group()
        return

    if args.cpu_leg:
        # (child of the run below) the CPU leg alone: the host set-up, the oracle on the host cores, one JSON line
        from roms_amd import hostlib as _hl
        cs0 = params_for(args.workload, args.Lm, args.Mm, args.N, ntimes=10)
        H0 = _hl.Host(params=cs0)
        try:
            cb = cpu_baseline(cs0, H0, budget_s=float(os.environ.get("ROMS_BENCH_CPU_BUDGET", "15")))
        finally:
            H0.finalize()
        # ... and the reference's own Fortran on one core beside it (after the port's replicas: the cores are idle again)
        ref1 = reference_leg(args.workload, cs0, budget_s=float(os.environ.get("ROMS_BENCH_REF_BUDGET", "12")))
        if ref1 is not None:
            cb["reference_1core"] = ref1
        print(json.dumps(cb), flush=True)
        return
    if not os.path.exists("/dev/kfd"):
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    # The CPU leg runs first, before anything initialises the GPU in this process (it forks worker processes;
    # even counting devices may open the driver on some ROCm builds).
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # In a process of its own: it forks worker processes, and it loads the host library -- with it the system's HIP
        # runtime -- which must not happen in THIS process before torch has loaded the runtime it ships (two HIP runtimes
        # in one process: the second one finds no device)
        import subprocess
        cmd = [sys.executable, os.path.abspath(__file__), "--cpu-leg", "--workload", args.workload]
        for k in ("Lm", "Mm", "N"):
            if getattr(args, k):
                cmd += [f"--{k}", str(getattr(args, k))]
        try:
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
            lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
            cpu = json.loads(lines[-1]) if p.returncode == 0 and lines else None
            if cpu is None:
                print("bench.py: the CPU baseline leg failed: " + (p.stderr.strip().splitlines() or ["?"])[-1], file=sys.stderr)
        except (subprocess.TimeoutExpired, ValueError) as e:     # never block the benchmark on its reported baseline
            print(f"bench.py: the CPU baseline leg failed: {e}", file=sys.stderr)
    import torch
    if torch.cuda.device_count() < 1 or not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    # --share-gpu (test aid): every rank on device 0, strips staged through the host (RCCL refuses two ranks on one
    # device): the multi-rank path of this script on a 1-GPU box
    device = 0 if args.share_gpu else local_rank
    torch.cuda.set_device(device)
    dist = None
    if world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ):   # started by torch.distributed.run
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
    local_rank = device

    from roms_amd import hiplib, hostlib, tiling
    if os.environ.get("ROMS_HIP_TRACE"):     # debugging aid: every launch synchronous and named on stderr
        hiplib.kprof(1)

    explicit_dims = bool(args.Lm or args.Mm or args.N)
    wl, tiles, weak = multi_gpu_plan(world, args.workload, explicit_dims)
    cs = params_for(wl, args.Lm, args.Mm, args.N, ntimes=args.steps + args.warmup)
    cs["ninfo"] = 1                          # NINFO of roms_benchmark1.in: diagnostics every step
    transport = args.transport or ("dist_staged" if (args.share_gpu and world > 1) else None)
    def make_run():
        try:
            return tiling.TiledRun(cs, rank=rank, world=world, device=local_rank, dist=dist, tiles=tiles, weak=weak,
                                   transport=transport)
        except hiplib.RomsHipError as e:     # no usable halo transport (both probes failed): say why, stop all ranks
            print(f"bench.py rank {rank}: {e}", file=sys.stderr, flush=True)
            raise SystemExit(3)
    run = make_run()
    loop_fallback = None
    if world > 1 and os.environ.get("ROMS_HIP_LOOP") is None and os.environ.get("ROMS_HIP_PAIR_RIM") is None:
        # The persistent barotropic loop crosses the tile edges inside one launch (round 6): its blocks wait -- bounded -- for
        # the neighbouring GPUs' blocks.  No multi-GPU node was available to develop it on, so the first steps are a trial:
        # if any rank reports a wait that gave up (exit_flag 2), EVERY rank goes back to the pair launches (ROMS_HIP_LOOP=0:
        # one exchange per predictor+corrector pair, the form of rounds 3-5) with fresh contexts, and the line says so.
        ok = 1
        try:
            run.step(2)
            run.sync()
        except hiplib.RomsHipError as e:
            ok = 0
            print(f"bench.py rank {rank}: trial steps with the persistent loop across tiles failed: {e}", file=sys.stderr, flush=True)
        tok = torch.tensor([ok], dtype=torch.int32, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tok, op=dist.ReduceOp.MIN)
        if int(tok.item()) == 0:
            loop_fallback = "the trial steps with the rim handed across tiles inside the barotropic launches failed on some rank: pair launches with an exchange behind each (ROMS_HIP_LOOP=0 ROMS_HIP_PAIR_RIM=0)"
            try:
                run.close()
            except Exception:
                pass
            os.environ["ROMS_HIP_LOOP"] = "0"
            os.environ["ROMS_HIP_PAIR_RIM"] = "0"       # (the rim hand-off of the pair launches waits for the neighbours too)
            cs = params_for(wl, args.Lm, args.Mm, args.N, ntimes=args.steps + args.warmup)
            cs["ninfo"] = 1
            run = make_run()
    if args.averages > 0:
        run.ctx.avg_config(args.averages)
    if not weak:                             # cs names the global grid: the tile is its NtileI x NtileJ-th part
        cs = dict(cs, Lm=cs["Lm"] // run.NtileI, Mm=cs["Mm"] // run.NtileJ)
    cells_per_rank = cs["Lm"] * cs["Mm"] * cs["N"]

    def barrier_sync():
        run.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
    red_dev = "cuda" if (dist is not None and dist.get_backend() == "nccl") else "cpu"

    run.step(args.warmup)
    barrier_sync()

    # dominant kernel: per-kernel breakdown over a few extra steps (synchronous events), then the
    # timed region with asynchronous event pairs on that kernel only
    dominant, table, pair = None, {}, None
    if not args.no_breakdown:
        hiplib.kprof(1)
        run.step(2)
        run.sync()
        table = hiplib.kprof_table()
        hiplib.kprof(0)
        pair = pair_report(table, 2, cells_per_rank)
        ranked = sorted(((k, v) for k, v in table.items() if k in ALGO_ARRAYS), key=lambda kv: -kv[1][0])
        if ranked:
            dominant = ranked[0][0]
        if args.breakdown_file and rank == 0:
            with open(args.breakdown_file, "w") as f:
                json.dump({k: {"seconds": v[0], "launches": v[1]} for k, v in table.items()}, f, indent=1)
    if dominant:
        # every launch of the dominant kernel inside the timed region carries a start / stop event pair filled by the
        # launch itself (kprof mode 3: the dispatch's own begin / end timestamps, i.e. the duration rocprofv3 reports;
        # rounds 1-3 bracketed runs of 16 launches with marker events, which counted the gaps between launches too
        # and read 24 % above rocprofv3); all launches of one step in ten are sampled: a pair on EVERY launch of every
        # step slows the step by 6 %
        lps = max(1, table[dominant][1] // 2)                     # launches per step (the breakdown pass ran 2 steps)
        every = int(os.environ.get("ROMS_BENCH_KSTEPS", "10" if args.steps >= 20 else "1"))
        hiplib.kprof(3, dominant, window=(lps, lps * every))
    barrier_sync()

    if world > 1:
        run.exchanges_per_step(0)            # (sets the count the timed region starts from)
    t0 = time.perf_counter()
    run.step(args.steps)
    barrier_sync()
    t1 = time.perf_counter()
    xps = run.exchanges_per_step(args.steps) if world > 1 else 0
    elapsed = t1 - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    roofline = None
    if dominant:
        sec, launches = hiplib.kprof_table().get(dominant, (0.0, 0))
        hiplib.kprof(0)
        if launches > 0:
            avg = sec / launches
            pairs = loop_pairs(table, run.nfast)
            nb = algo_bytes(dominant, cs["Lm"], cs["Mm"], cs["N"], pairs=pairs)
            survey = nb / avg / 1e9
            fused = loop_fused_words(pairs) if dominant == "k_step2d_loop" else FUSED_2D_WORDS.get(dominant)
            # the headline figure prices the kernel on the bytes IT must move (a fused kernel reads the state once for the
            # calls it replaces); SURVEY 8(d)'s per-call unit beside it
            fb = 8.0 * cs["Lm"] * cs["Mm"] * fused if fused else nb
            roofline = {"bound": "hbm", "kernel": dominant, "achieved": fb / avg / 1e9, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": fb / avg / 1e9 / HBM_PEAK_GBS, "traffic": None,
                        "avg_launch_us": avg * 1e6, "launches": launches,
                        "avg_launch_method": "hipExtLaunchKernel start/stop events on every launch of the kernel in one step out of ten inside "
                                             "the timed region (dispatch begin -> end, as rocprofv3 --kernel-trace)",
                        "algorithmic_bytes_per_launch": fb,
                        "survey_unit_bytes_per_launch": nb, "achieved_survey_unit": survey,
                        "frac_survey_unit": survey / HBM_PEAK_GBS}
            if dominant == "k_step2d_loop":
                roofline["pairs_per_launch"] = pairs
                roofline["us_per_pair"] = avg * 1e6 / pairs
                roofline["step2d_calls_per_launch"] = int(round(2 * pairs))
                roofline["bound_note"] = ("the kernel keeps its 13 MB working set in LDS / registers for all the fast steps of a launch and "
                                          "moves only the rim between blocks: it is bound by the dependent LDS / f64 chains of its stages and "
                                          "the rim hand-off (DESIGN.md 3), not by HBM -- `frac` says how little of the memory system it needs")
            # the whole step against the same peak: SURVEY 8(d)'s bytes per cell-update x cells / step time
            wsb = whole_step_bytes_per_cell(cs, run.nfast)
            roofline["whole_step_bytes_per_cell"] = wsb
            roofline["whole_step_frac"] = wsb * cells_per_rank * args.steps / elapsed / 1e9 / HBM_PEAK_GBS
            if dominant in table and table[dominant][1] > 0:
                # the same kernel with nothing else on the chip (synchronous per-kernel pass before the timed
                # region; each launch bracketed by two event markers, which add ~1 us to a 10 us kernel)
                iso = table[dominant][0] / table[dominant][1]
                roofline["isolated_launch_us"] = iso * 1e6
                roofline["frac_isolated"] = fb / iso / 1e9 / HBM_PEAK_GBS

    # the spread of the step time: a pass of its own behind the timed region, a HIP event at
    # every step boundary on the main stream (the timed region above carries no such markers)
    spread = None
    try:
        ts = sorted(run.step_times(min(40, max(10, args.steps))))
        if ts:
            spread = {"n": len(ts), "min": ts[0], "median": ts[len(ts) // 2], "max": ts[-1],
                      "method": "HIP events at the step boundaries of a separate pass (roms_hip_step_timing)"}
    except Exception as e:
        spread = {"error": str(e)[:200]}
    barrier_sync()
    copy_gbs = run.ctx.copy_probe() if (args.copy_probe or rank == 0) else None
    if roofline is not None:
        roofline["measured_copy_GBs"] = copy_gbs
        roofline["traffic"] = pmc_traffic(args.workload, dominant, world)
        roofline["traffic_source"] = (f"{pmc_traffic.source}: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload, committed "
                                      "with the round's profiles; read from that file, not collected in this run") if roofline["traffic"] is not None else None
    run.check()                              # blow-up test of the last diagnostics (exit_flag)
    out = None
    if rank == 0:
        value = cells_per_rank * world * args.steps / elapsed
        # the tile of this run against the tile of the N = 1 line of the same workload: BASELINE's own 8-GPU configuration
        # (BENCHMARK3 in 2x4) has a 1024x64 tile, twice the 512x64 of BENCHMARK1 -- a scaling curve must be normalised
        # by cells per GPU, which the line therefore states
        n1 = WORKLOADS[args.workload]
        n1_tile = (args.Lm or n1[1], args.Mm or n1[2], args.N or n1[3])
        tile = (cs["Lm"], cs["Mm"], cs["N"])
        ratio = (tile[0] * tile[1] * tile[2]) / float(n1_tile[0] * n1_tile[1] * n1_tile[2])
        scaling = "weak" if abs(ratio - 1.0) < 1e-9 else f"weak (tile x{ratio:g} vs n=1)"
        out = {
            "metric": "grid-cell-updates/sec", "value": value, "unit": "grid-cell-updates/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{cs['app'].upper()} {run.global_Lm}x{run.global_Mm}x{cs['N']} "
                                   f"(tile {cs['Lm']}x{cs['Mm']}x{cs['N']} per GPU), dt={cs['dt']:g}s ndtfast={cs['ndtfast']}, "
                                   "analytic grid/initial/forcing, full application physics",
                       "tiles": f"{run.NtileI}x{run.NtileJ}", "nfast": run.nfast,
                       "tile": "%dx%dx%d" % tile, "n1_tile": "%dx%dx%d" % n1_tile, "cells_per_gpu": cells_per_rank,
                       "halo_transport": getattr(run, "transport", None) if world > 1 else "none (single tile)",
                       "rccl_ranks": run.rccl_ranks() if world > 1 else None,
                       "transport_probes": getattr(run, "probe_log", None) if world > 1 else None,
                       "exchanges_per_step": xps, "barotropic_loop_fallback": loop_fallback},
            "roofline": roofline,
            "north_star_pair": pair,
            "step_time_ms": spread,
        }
        out["cpu_baseline"] = cpu
    run.close()
    if world > 1 and (not args.share_gpu or os.environ.get("ROMS_BENCH_FORCE_AB")) and not os.environ.get("ROMS_BENCH_NO_AB"):
        # The same steps through the OTHER device-to-device transport of the library: the headline
        # above used `run.transport` (auto: the mailbox where its probe passes on every rank, else RCCL); here a fresh set of
        # contexts with the other one, a short timed region with the same barriers.  rccl_ranks is read from the live RCCL
        # communicator of whichever of the two runs used it (ncclCommCount).  Guarded: an exception on any rank, or a
        # rendezvous that does not complete within two minutes, ends the comparison -- never the benchmark line.
        import signal
        head_tr = getattr(run, "transport", None)
        other = "rccl" if head_tr == "peer" else "peer"
        ab = {"headline": "mailbox" if head_tr == "peer" else head_tr, ("mailbox" if head_tr == "peer" else str(head_tr)): {"ms_per_step": 1e3 * elapsed / args.steps, "exchanges_per_step": xps}}

        def give_up(signum, frame):
            if rank == 0 and out is not None:
                ab["error"] = "the comparison run did not finish within 120 s"
                out["config"]["transport_ab"] = ab
                print(json.dumps(out), flush=True)
            os._exit(0)
        signal.signal(signal.SIGALRM, give_up)
        signal.alarm(120)
        try:
            n_ab = max(5, min(args.steps, 20))
            cs2 = params_for(wl, args.Lm, args.Mm, args.N, ntimes=n_ab + 12)
            cs2["ninfo"] = 1
            run2 = tiling.TiledRun(cs2, rank=rank, world=world, device=local_rank, dist=dist, tiles=tiles, weak=weak, transport=other)
            run2.step(6)
            run2.sync(); torch.cuda.synchronize(); dist.barrier()
            run2.exchanges_per_step(0)
            ta = time.perf_counter()
            run2.step(n_ab)
            run2.sync(); torch.cuda.synchronize(); dist.barrier()
            tb = torch.tensor([time.perf_counter() - ta], dtype=torch.float64, device=red_dev)
            dist.all_reduce(tb, op=dist.ReduceOp.MAX)
            ab["mailbox" if other == "peer" else other] = {"ms_per_step": 1e3 * float(tb.item()) / n_ab, "exchanges_per_step": run2.exchanges_per_step(n_ab),
                                                           "steps": n_ab}
            if other == "rccl":
                ab["rccl_ranks_live"] = run2.rccl_ranks()
            run2.check()
            run2.close()
        except Exception as e:
            ab["error"] = str(e)[:300]
        signal.alarm(0)
        if out is not None:
            out["config"]["transport_ab"] = ab
            if ab.get("rccl_ranks_live") is not None:
                out["config"]["rccl_ranks"] = ab["rccl_ranks_live"]
    if rank == 0 and world == 1 and args.workload == "benchmark1" and not explicit_dims and not args.no_breakdown \
            and not args.no_north_star:
        # BASELINE.json north_star: "step3d_t + rhs3d at 512x512x50" -- a short pass of that grid (UPWELLING
        # physics, U3/C4 advection: the schemes SURVEY 8(d) prices the pair on) with every kernel timed
        out["north_star_pair_512x512x50"] = north_star_pass(hiplib, tiling, local_rank)
        out["north_star_pair_512x512x50_stock"] = north_star_pass(hiplib, tiling, local_rank, steps=4, warmup=8, workload="ns512")
        # the other BASELINE configurations that fit one GPU, driver-timed: whole steps
        out["whole_step"] = {wl: whole_step_pass(hiplib, tiling, local_rank, wl) for wl in ("benchmark2", "benchmark3", "config5", "ns512")}
        out["whole_step"]["config5_varS"] = whole_step_pass(hiplib, tiling, local_rank, "config5", vary_salt=True)
        # the multi-tile form of the headline step, on this one GPU
        out["tiled_form_selfx"] = tiled_form_pass(tiling, local_rank)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
