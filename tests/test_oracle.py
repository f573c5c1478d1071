"""CPU tests of the oracle (no GPU, no reference tree needed): golden set-up tables, tiling
invariance, and reference-independent properties of the core kernels (volume conservation, constancy
of a uniform tracer) that complement the bit-for-bit pins of tests/test_oracle_vs_ref.py and
tests/test_golden_reference.py."""
import os

import numpy as np
import pytest

from tests import util, cases


def test_bounds_tables_match_reference_get_bounds():
    """orc_tile_bounds == BOUNDS/DOMAIN of the reference (get_bounds.F) for several tilings."""
    from oracle import orc
    n = 0
    for f in sorted(os.listdir(util.GOLDEN)):
        if not f.startswith("bounds_"):
            continue
        z = np.load(os.path.join(util.GOLDEN, f))
        _, app, dims, tiling, hs = f[:-4].split("_")
        Lm, Mm = [int(x) for x in dims.split("x")]
        nti, ntj = [int(x) for x in tiling.split("x")]
        kw = dict(Lm=Lm, Mm=Mm, NtileI=nti, NtileJ=ntj)
        if app == "upwelling":
            kw.update(hadv=("U3", hs), vadv=("C4", hs))
        cs = getattr(cases, app)(**kw)
        O = orc.Oracle(cases.oracle_cfg(cs, 25.0, 42, np.zeros((2, 60))))
        for t in range(nti * ntj):
            assert O.bounds(t) == [int(x) for x in z["table"][t][:54]], (f, t)
            n += 1
        O.close()
    assert n >= 30


def _fresh(tag="upwelling_small", **kw):
    cs = util.case_for(tag, **kw)
    g = util.load_init(tag, util.nghost_for(cs))
    return cs, g, util.make_oracle(cs, g)


def test_initial_depths_match_reference():
    """set_depth (pinned): recomputing Hz, z_r, z_w from the golden h reproduces the reference."""
    cs, g, O = _fresh()
    O.step.nstp = 1
    O.step.nrhs = 1          # (before set_massflux, which reads u, v(nrhs))
    O.call("set_depth")
    for n in ["Hz", "z_r", "z_w"]:
        assert np.array_equal(O.field(n), g[n]), n
    O.call("set_massflux")
    O.call("rho_eos")
    for n in ["rho", "rhoA", "rhoS", "Huon", "Hvom"]:
        assert np.array_equal(O.field(n), g[n]), n


@pytest.mark.parametrize("hadv,vadv", [(("U3", "HSIMT"), ("C4", "HSIMT")), (("U3", "U3"), ("C4", "C4")),
                                       (("C4", "A4"), ("SPLINES", "A4")), (("C2", "SU3"), ("C2", "SU3"))])
def test_constant_tracer_and_volume(hadv, vadv):
    """Reference-independent properties: a uniform tracer (S = 35, no fluxes) stays uniform to
    round-off (consistency of omega, pre_step3d, step3d_t with the corrected mass fluxes of
    step3d_uv/step2d) and the total volume is conserved."""
    cs, g, O = _fresh(hadv=hadv, vadv=vadv)
    O.start()
    v0 = None
    for s in range(30):
        O.main3d_step()
        d = O.diag()
        v0 = v0 or d[3]
        assert abs(d[3] - v0) <= 1e-12 * v0
        assert np.isfinite(d[0]) and d[4] < 1.0
    N, nij = cs["N"], O.ni * O.nj
    salt = O.field("t")[3 * N * nij:6 * N * nij].reshape(3, N, O.nj, O.ni)
    b = O.bounds(0)
    inner = salt[:, :, b[6] - b[2]:b[7] - b[2] + 1, b[4] - b[0]:b[5] - b[0] + 1][:2]
    assert np.abs(inner - 35.0).max() < 5e-12


@pytest.mark.parametrize("tiling", [(2, 2), (3, 1), (1, 4)])
def test_tiling_invariance(tiling):
    """The reference's own acceptance criterion (ROMS/Bin/verify.sh): identical results for any
    tile partition.  Shared-memory tiles, 10 steps, all prognostic fields bit-identical."""
    res = []
    for nti, ntj in [(1, 1), tiling]:
        cs = util.case_for("upwelling_small", hadv=("U3", "HSIMT"), vadv=("C4", "HSIMT"))
        cs["NtileI"], cs["NtileJ"] = nti, ntj
        g = util.load_init("upwelling_small", 3)
        O = util.make_oracle(cs, g)
        O.start()
        O.main3d_step(10)
        res.append({n: O.field(n).copy() for n in ["zeta", "ubar", "vbar", "u", "v", "t", "W", "Huon", "Hvom"]})
        O.close()
    for n in res[0]:
        assert np.array_equal(res[0][n], res[1][n]), n


def test_filter_weights_moments():
    """set_weights.F:233-234: the barotropic filter's centres of gravity "must be 1, 1, ~1/2, 1, 1"."""
    g = util.load_init("upwelling")
    w = g["weight"]
    nfast, ndtfast = int(g["bounds"][58]), 30
    k = np.arange(1, 2 * ndtfast + 1)
    assert abs(w[0].sum() - 1.0) < 1e-13 and abs(w[1].sum() - 1.0) < 1e-13
    assert abs((w[0] * k).sum() / ndtfast - 1.0) < 1e-12
    assert np.all(w[:, nfast:] == 0.0)


def test_threaded_tiles_bitwise():
    """The cpu_baseline leg of bench.py runs the oracle's tile loops as OpenMP threads over eta strips
    (NtileI = 1).  That run must be the serial one bit for bit -- same strips, one thread -- and, strips
    keeping every periodic copy inside a tile, also the untiled one."""
    res = []
    for ntj, thr in [(1, 1), (4, 1), (4, 4)]:
        cs = util.case_for("benchmark_small")
        cs["NtileI"], cs["NtileJ"] = 1, ntj
        g = util.load_init("benchmark_small", 2)
        O = util.make_oracle(cs, g)
        O.set_threads(thr)
        O.start()
        O.main3d_step(12)
        res.append({n: O.field(n).copy() for n in ["zeta", "ubar", "vbar", "u", "v", "t", "W", "Huon", "Akv", "Akt",
                                                   "hsbl", "stflx", "sustr"]})
        O.close()
    for n in res[0]:
        assert np.array_equal(res[0][n], res[1][n]), ("strips vs untiled", n)
        assert np.array_equal(res[1][n], res[2][n]), ("threads vs serial", n)
