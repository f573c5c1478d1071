"""Multi-tile path (one tile per process) on CPU: world_size-2/4 gloo runs of the
emulated kernels driven by the Fortran host must reproduce the single-tile run BIT FOR BIT --
the reference's own tiling invariance (tests/test_oracle.py proves it for the oracle)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import util

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
EMU = os.path.join(ROOT, "tests", "emu")
FIELDS = ["zeta", "ubar", "vbar", "u", "v", "t", "W", "Hz", "Huon", "Hvom", "rho", "Akv", "DU_avg1", "Zt_avg1"]
GLS_FIELDS = ["tke", "gls", "Akt", "Akk", "Akp", "Lscale"]


def _emu_libs():
    if not (os.path.exists(os.path.join(EMU, "libroms_host_emu.so")) and os.path.exists(util.EMU_LIB)):
        subprocess.check_call(["bash", os.path.join(EMU, "build_emu.sh")])


def _fields(tag):
    return FIELDS + (GLS_FIELDS if tag.startswith(("upwelling_gls", "upwelling_my25", "upwelling_wetdry_gls", "upwelling_wetdry_my25")) else [])


def _single(tag, kw, steps):
    from roms_amd import tiling
    cs = util.case_for(tag, **kw)
    cs["ninfo"] = 0
    run = tiling.TiledRun(cs, weak=False, host_lib=os.path.join(EMU, "libroms_host_emu.so"), hip_lib=util.EMU_LIB)
    run.step(steps)
    res = {n: run.gather(n) for n in _fields(tag)}
    d = run.diag()
    run.close()
    return res, d


def _tiled(tmp_path, tag, kw, steps, tiles, port, kernels=False):
    out = str(tmp_path / f"tiles_{tiles[0]}x{tiles[1]}.npz")
    spec = dict(tag=tag, kw=kw, steps=steps, tiles=list(tiles), fields=_fields(tag), kernels=kernels, probe=True)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={tiles[0] * tiles[1]}",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "mp", "run_tiles.py"), out, json.dumps(spec)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    return dict(np.load(out))


def _interior(cs_dims, a):
    return a


@pytest.mark.parametrize("tag,kw,tiles,port", [
    ("upwelling_small", dict(hadv=("U3", "HSIMT"), vadv=("C4", "HSIMT")), (2, 1), 29611),
    # the 8-GPU partition of bench.py: every neighbour of a tile (sides and diagonals) is a different rank
    ("benchmark_small", dict(), (4, 2), 29615),
    # three ghost lines on the high side (MPDATA), corner blocks from the diagonal tiles
    ("upwelling_small", dict(hadv=("MPDATA", "MPDATA"), vadv=("MPDATA", "MPDATA")), (2, 2), 29614),
    # MASKING (the island straddles the tile boundaries, the headland sits on the southern wall of one tile) with MPDATA: the
    # masked cross-gradient terms read umask/vmask in the ghost lines (plain MASKING on 2x2 tiles: upwelling_mask_mid below)
    ("upwelling_mask_small", dict(hadv=("MPDATA", "MPDATA"), vadv=("MPDATA", "MPDATA")), (2, 2), 29622),
    # tiles of 8 points and more: the barotropic steps run as predictor+corrector pairs (k_step2d_pair.h) with one exchange of
    # 5 | 4 lines per pair -- 2x2 (corner blocks of the wide strips), the 8-rank layout, three ghost lines + MPDATA, MASKING
    ("benchmark_mid", dict(), (2, 2), 29617),
    ("benchmark_mid", dict(), (4, 2), 29618),
    ("upwelling_mid", dict(hadv=("MPDATA", "MPDATA"), vadv=("MPDATA", "MPDATA")), (2, 2), 29619),
    ("upwelling_mask_mid", dict(hadv=("U3", "HSIMT"), vadv=("C4", "HSIMT")), (2, 2), 29620),
    ("upwelling_mid", dict(hadv=("U3", "U3"), vadv=("C4", "C4")), (1, 2), 29621),
    # open boundaries (the reference's KELVIN application): the western and eastern conditions read along the edge across
    # the tile boundary (the tangential differences of the radiation condition), corner tiles hold both kinds of edge
    ("kelvin_small", dict(), (2, 2), 29623),
    # SEAMOUNT (no-slip walls) and GRAV_ADJ (MPDATA, closed in xi, periodic and four points wide in eta: tiles along xi)
    ("seamount_small", dict(), (2, 2), 29625),
    ("grav_adj_small", dict(), (2, 1), 29626),
    # GLS_MIXING: tke, gls (index 3 behind gls_prestep, nnew behind gls_corstep), Akv, Akt travel; the smoothing of N2 and
    # shear reads them and the work array across the tile boundary; masked (Canuto A) and with MPDATA's three ghost lines
    ("upwelling_gls_small", dict(), (2, 2), 29627),
    ("upwelling_gls_ca_small:gen", dict(), (2, 2), 29628),
    ("upwelling_gls_cb_small:k-kl", dict(hadv=("MPDATA", "MPDATA"), vadv=("MPDATA", "MPDATA")), (1, 2), 29629),
    # MY25_MIXING: the eastern tiles carry my25_corstep.F's copy onto the interior column Iend-1
    ("upwelling_my25_small", dict(), (2, 2), 29630),
    # OVERFLOW (closed channel four points wide, tiles along eta): MIX_ISO_TS reads the density and the tracer slopes across the
    # tile boundary
    ("overflow_small", dict(), (1, 2), 29631),
    # biharmonic mixing (UV_VIS4, TS_DIF4): the first harmonic operator is formed from the neighbour's points across the tile
    # boundary (three ghost lines), its closed-edge conditions and corner values on the edge tiles only
    ("upwelling_bih_mid", dict(), (2, 2), 29633),
    # WET_DRY: the wet/dry masks of a tile's ghost points are computed from the exchanged free surface (no exchange of masks);
    # the shore line crosses the tile boundaries, the averaged masks read DU_avg1 / DV_avg1 in the ghost lines
    ("upwelling_wetdry_mid", dict(hadv=("U3", "HSIMT"), vadv=("C4", "HSIMT")), (2, 2), 29634),
    # round 5: the viscosity along geopotentials (five point-wise kernels reading u, v, z_r, Hz two points across the tile
    # boundary; MASKING) and the biharmonic tracer mixing along geopotentials (the first operator on the tile widened by one point)
    ("upwelling_geouv_mid", dict(hadv=("U3", "HSIMT"), vadv=("C4", "HSIMT")), (2, 2), 29635),
    ("upwelling_bihgeo_mid", dict(), (2, 2), 29636),
    ("upwelling_bihiso_mid", dict(), (2, 2), 29646),
    # round 6: the BIHARMONIC viscosity along geopotentials (uv3dmix4_geo.h): the first operator on the tile widened by one point
    # -- three ghost lines of u, v, z_r, Hz --, its conditions on the edge tiles only; MASKING
    ("upwelling_bihgeouv_mid", dict(hadv=("U3", "U3"), vadv=("C4", "C4")), (2, 2), 29648),
    # round 6: PJ_GRADPQ4 (prsgrd44.h: the reconstruction runs on the tile widened by one column; no partition dependence)
    ("upwelling_prs44_small", dict(), (2, 2), 29647),
])
def test_tiled_run_bit_identical_to_single_tile(tmp_path, tag, kw, tiles, port):
    _emu_libs()
    steps = 4
    ref, dref = _single(tag, kw, steps)
    got = _tiled(tmp_path, tag, kw, steps, tiles, port)
    assert int(got["nexchanges"]) > 30 * steps          # the strips really travelled
    for n in _fields(tag):
        a, b = got[n], ref[n]
        assert a.shape == b.shape, n
        # compare everything the single-tile run defines; ghost entries that no kernel reads are
        # excluded by comparing only where the single-tile field was ever written or gathered
        assert np.array_equal(a, b), (n, float(np.abs(a - b).max()), np.argwhere(a != b)[:5])
    assert got["diag"][2] == pytest.approx(dref["volume"], rel=1e-14)
    assert got["diag"][0] == pytest.approx(dref["avgke"], rel=1e-12)


@pytest.mark.parametrize("variant,port", [("gls", 29651), ("geouv", 29652), ("prs44", 29653)])
def test_wet_dry_variants_tiled_bit_identical_to_single_tile(tmp_path, monkeypatch, variant, port):
    """Round 6: WET_DRY with the generic length-scale closure / the viscosity along geopotentials / PJ_GRADPQ4 on 2x2 tiles -- the host
    takes the options from the header the reference was built with for the pin (ROMS_APP_HEADER = oracle/ref/upwelling_wetdry_<v>.h);
    the shore line crosses the tile boundaries; every field of the single-tile run bit for bit."""
    _emu_libs()
    monkeypatch.setenv("ROMS_APP_HEADER", os.path.join(ROOT, "oracle", "ref", "upwelling_wetdry_%s.h" % variant))
    tag, kw, steps = "upwelling_wetdry_%s_mid" % variant, dict(hadv=("U3", "HSIMT"), vadv=("C4", "HSIMT")), 4
    ref, dref = _single(tag, kw, steps)
    got = _tiled(tmp_path, tag, kw, steps, (2, 2), port)
    for n in _fields(tag):
        assert np.array_equal(got[n], ref[n]), (n, float(np.abs(got[n] - ref[n]).max()))


def test_quadratic_pressure_jacobian_with_second_pass_stays_on_one_tile():
    """PJ_GRADPQ2 (prsgrd42.h) on two tiles: its second pass reads rv(Iend+1,j,k), which no tile computes (prsgrd42.h:449) -- the
    reference's result depends on the partition there, so host and library stop with exit_flag 5 and say why."""
    from roms_amd import hiplib, hostlib
    _emu_libs()
    cs = util.case_for("upwelling_prs42_small")
    cs["NtileI"] = 2
    with pytest.raises(hostlib.HostError) as e:
        hostlib.Host(params=cs, lib_path=os.path.join(EMU, "libroms_host_emu.so"), hip_lib_path=util.EMU_LIB)
    assert e.value.exit_flag == 5 and "prsgrd42.h:449" in str(e.value)
    g = util.load_init("upwelling_small", util.nghost_for(cs))
    from tests import cases
    cfg = cases.hip_cfg(cs, float(g["scalars"][0]), int(g["bounds"][58]), g["weight"], g["sc_r"], g["Cs_r"], g["sc_w"], g["Cs_w"])
    cfg.NtileI = 2                                       # (what a caller with two tiles along xi would pass)
    with pytest.raises(hiplib.RomsHipError) as e2:
        hiplib.Context(cfg, util.EMU_LIB)
    assert "exit_flag=5" in str(e2.value) and "PJ_GRADPQ2" in str(e2.value)


def test_partition_rule():
    from roms_amd import tiling
    assert [tiling.partition(n) for n in (1, 2, 4, 8, 6)] == [(1, 1), (2, 1), (2, 2), (4, 2), (3, 2)]


def test_bench_multi_gpu_plan_is_the_baseline_configs():
    """bench.py --gpus N of the default workload runs BASELINE.json's own multi-GPU configurations; the tile
    each rank gets follows from the reference's partition rule (get_bounds.F:972-1042)."""
    import bench
    from roms_amd import tiling
    assert bench.multi_gpu_plan(1, "benchmark1", False) == ("benchmark1", None, True)
    assert bench.multi_gpu_plan(2, "benchmark1", False) == ("benchmark1", (2, 1), True)
    assert bench.multi_gpu_plan(4, "benchmark1", False) == ("benchmark2", (2, 2), False)
    assert bench.multi_gpu_plan(8, "benchmark1", False) == ("benchmark3", (2, 4), False)
    assert bench.multi_gpu_plan(8, "ns512", False) == ("ns512", None, True)
    assert bench.multi_gpu_plan(4, "benchmark1", True) == ("benchmark1", None, True)
    for world, (glm, gmm), tile in [(4, (1024, 128), (512, 64)), (8, (2048, 256), (1024, 64))]:
        wl, tiles, weak = bench.multi_gpu_plan(world, "benchmark1", False)
        cs = bench.params_for(wl)
        assert (cs["Lm"], cs["Mm"]) == (glm, gmm) and tiles[0] * tiles[1] == world
        assert (cs["Lm"] // tiles[0], cs["Mm"] // tiles[1]) == tile
    assert tiling.partition(8) == (4, 2) and tiling.partition(2) == (2, 1)


def test_automatic_transport_fails_cleanly_without_a_device_transport(tmp_path):
    """transport="auto" (tiling.py: mailbox, else RCCL, each behind collectives of all ranks) where neither exists -- the
    emulated build: every rank gets the library's message, none hangs in a collective the other has left."""
    import json
    import sys
    _emu_libs()
    spec = dict(tag="upwelling_small", kw={}, steps=1, tiles=[2, 1], fields=["zeta"], transport_cpu="auto")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29641", os.path.join(ROOT, "tests", "mp", "run_tiles.py"), str(tmp_path / "x.npz"), json.dumps(spec)]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=180, env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert p.returncode != 0
    out = p.stdout + p.stderr
    assert "mailbox transport not usable" in out and "RCCL is not part of the CPU-emulated test build" in out, out[-3000:]


def test_exchange_soak_passes_and_detects_a_missing_exchange(tmp_path):
    """roms_hip_exchange_soak (round 4): 200 exchange points back to back without a host synchronisation, every repetition
    coded and verified by a kernel -- it passes on two ranks through the callback transport, and with ONE exchange left
    out (ROMS_HIP_SOAK_FAULT=1: the ghost zone keeps what the fill left there) it stops with the first wrong point."""
    import json
    import sys
    _emu_libs()
    spec = dict(tag="upwelling_mid", kw={}, steps=1, tiles=[2, 1], fields=["zeta"], probe=True)
    for fault, port in (("0", 29643), ("1", 29644)):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(ROOT, "tests", "mp", "run_tiles.py"), str(tmp_path / "x.npz"), json.dumps(spec)]
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=dict(os.environ, OMP_NUM_THREADS="1", ROMS_HIP_SOAK_FAULT=fault))
        if fault == "0":
            assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
        else:
            assert p.returncode != 0 and "exchange soak" in p.stdout + p.stderr, p.stdout[-2000:] + p.stderr[-2000:]


def test_bench_starts_its_own_ranks_when_run_plainly():
    """`python bench.py --gpus 2` without torch.distributed.run (WORLD_SIZE unset): the script starts its ranks as child
    processes itself and relays rank 0's line.  --dry-launch stops after the rendezvous (no GPU needed here)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["OMP_NUM_THREADS"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d == {"launch": "ok", "n_gpus": 2, "rank_sum": 3, "self_launched": True}


def test_bench_refuses_a_mismatched_world():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", OMP_NUM_THREADS="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch"],
                       capture_output=True, text=True, timeout=120, env=env)
    assert p.returncode != 0 and "WORLD_SIZE=3" in (p.stdout + p.stderr)


def test_bench_cpu_leg_runs_in_its_own_process():
    """bench.py times the CPU baseline in a child process (`--cpu-leg`): the leg loads the host library -- and with it the
    system's HIP runtime -- which must not be in the benchmark's own process before torch has loaded the runtime it
    ships (a default `python bench.py` found no device afterwards).  The child prints one JSON object."""
    env = dict(os.environ, ROMS_BENCH_CPU_BUDGET="1", OMP_NUM_THREADS="2")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-leg", "--Lm", "24", "--Mm", "16", "--N", "6"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert d["kind"] == "port" and d["value"] > 0 and d["cores"] >= 1 and "oracle/liborc.so" in d["sample"]
