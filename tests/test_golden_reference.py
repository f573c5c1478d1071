"""The oracle against the fixtures the reference's own object code wrote (tests/golden/*_steps.npz,
*_kernels.npz; generator: tests/golden/make_golden.py --reference-runs).  Runs anywhere (no /root/reference,
no GPU): this is what keeps the oracle pinned on the GPU box, where oracle/_ref does not exist.
Bit-exact: the oracle restates the reference's arithmetic operation by operation."""
import glob
import os

import numpy as np
import pytest

from tests import util

STEPS = sorted(os.path.basename(p) for p in glob.glob(os.path.join(util.GOLDEN, "*_steps.npz")))
KERNELS = sorted(os.path.basename(p) for p in glob.glob(os.path.join(util.GOLDEN, "*_kernels.npz")))


@pytest.mark.parametrize("name", STEPS)
def test_oracle_reproduces_reference_steps(name):
    """main3d passes (main3d.F:216-1148): every kept array after steps 1, 2, 3 and the last, array_equal."""
    f, meta = util.load_fixture(name)
    side = util.OracleSide(util.case_from_meta(meta))
    worst = util.check_steps_fixture(side, f, meta, tol=0.0)
    assert len(worst) >= 25
    side.close()


@pytest.mark.parametrize("name", KERNELS)
def test_oracle_reproduces_reference_kernel_by_kernel(name):
    """every kernel call of steps 1 and 2 on the reference's own input: outputs array_equal, diag line equal
    as printed."""
    f, meta = util.load_fixture(name)
    side = util.OracleSide(util.case_from_meta(meta))
    worst = util.check_kernels_fixture(side, f, meta, tol=0.0)
    assert len(worst) >= 40
    side.close()


@pytest.mark.parametrize("name,nsteps", [("upwelling_sample.npz", None), ("benchmark1_sample.npz", 20)])
def test_host_setup_plus_oracle_reproduce_reference_sample(name, nsteps):
    """BASELINE configs[0] (UPWELLING 41x80x16, 100 steps) and configs[1] (BENCHMARK1 512x64x30): the product's
    Fortran host set-up from roms.in values (roms_amd/host) feeds the oracle, which must land on the
    reference's own end state.  BENCHMARK1 is cut to 20 steps here for time -- its fixture holds step 100 --
    so for it only the set-up and the first diag line are checked against the reference, the rest against the
    100-step GPU test."""
    import numpy as np
    import bench
    from oracle import orc
    from roms_amd import hostlib
    from tests import cases
    from tests.test_host import HOST_FIELDS
    f, meta = util.load_fixture(name)
    cs = bench.params_for(meta["workload"], ntimes=meta["nsteps"])
    H = hostlib.Host(params=cs)
    try:
        w = np.stack([H.get("weight1"), H.get("weight2")])
        O = orc.Oracle(cases.oracle_cfg(cs, H.reals["hc"], H.dims["nfast"], w))
        for n in HOST_FIELDS:
            try:
                O.field(n)[:] = H.get(n)
            except KeyError:
                pass
    finally:
        H.finalize()
    O.start()
    O.main3d_step(nsteps or meta["nsteps"])
    import ctypes as C
    dg = (C.c_double * 16)()
    O.L.orc_get_diag(C.c_void_p(O.h), dg)        # the numbers of the diag call inside the last step (main3d.F:355)
    if nsteps is None:
        ii, jj = f["ii"], f["jj"]
        for n in meta["fields"]:
            a = O.field(n).reshape(-1, meta["nj"], meta["ni"])[:, jj][:, :, ii]
            mask = np.ones(a.shape[2], bool)
            mask[ii > cs["Lm"] + 2 * util.nghost_for(cs)] = False        # the padding column (see util.unpadded)
            assert np.array_equal(a[:, :, mask], f[n][:, :, mask]), n
        assert util.fmt_diag(list(dg)) == [list(x) for x in meta["diag"][-1]]
    else:
        d = list(dg)
        assert np.isfinite(d[:4]).all() and ("%.6E" % d[3])[:6] == meta["diag"][-1][0][3][:6]   # volume, 5 digits
    O.close()


def test_oracle_set_avg_matches_reference_fixture():
    """tests/golden/upwelling_small_avg.npz: the 22 time-averaged arrays the reference's own set_avg.F held after the
    window-closing steps 4 and 7 (nAVG = 3; written by make_golden.py --avg from the reference built with AVERAGES):
    the oracle's, bit for bit, anywhere."""
    z = np.load(os.path.join(util.GOLDEN, "upwelling_small_avg.npz"))
    cs = util.case_for("upwelling_small")
    g = util.load_init("upwelling_small", util.nghost_for(cs))
    O = util.make_oracle(cs, g)
    O.set_avg_window(int(z["nAVG"]), int(z["ntsAVG"]))
    O.start()
    n = 0
    for step in range(1, 8):
        O.main3d_step()
        for key in z.files:
            if key.startswith(f"s{step}_"):
                assert np.array_equal(O.field(key[3:]), z[key]), key
                n += 1
    assert n == 44


def test_oracle_set_diags_matches_reference_fixture():
    """tests/golden/upwelling_small_dia.npz: the per-term tracer tendencies the reference's own set_diags.F held after the
    window-closing steps 4 and 7 (nDIA = 3), and the raw terms DiaTwrk at the end of step 5 (written by make_golden.py --dia
    from the reference built from upwelling.h as shipped, DIAGNOSTICS_TS on): the oracle's, bit for bit, anywhere."""
    z = np.load(os.path.join(util.GOLDEN, "upwelling_small_dia.npz"))
    cs = util.case_for("upwelling_small")
    g = util.load_init("upwelling_small", util.nghost_for(cs))
    O = util.make_oracle(cs, g)
    O.set_dia_window(int(z["nDIA"]), int(z["ntsDIA"]), uv=True)       # ... and the momentum terms (DIAGNOSTICS_UV)
    O.start()
    n = 0
    for step in range(1, 8):
        O.main3d_step()
        for key in z.files:
            if key.startswith(f"s{step}_") or key.startswith(f"e{step}_"):
                assert np.array_equal(O.field(key[3:]), z[key]), key
                n += 1
    assert n == 16 and np.abs(z["s7_DiaTrc"]).max() > 0.0 and np.abs(z["s7_DiaU3d"]).max() > 0.0
