"""Reference-independent physics of the hot path: what the kernels must do whatever the reference's bits are.

The bit-for-bit pins (tests/test_oracle_vs_ref.py, test_golden_reference.py, test_gpu_vs_reference.py) say the
oracle and the HIP kernels compute what the reference computes; these tests say that what all three compute is the
physics the routines stand for.  Each test runs on the C oracle (CPU, always) and on libroms_hip.so through its C ABI
(-m gpu), driven kernel by kernel in main3d's order (tests/refdrive.py:main3d_sequence) so that the analytic wind
stress of the application can be switched off:

  * a resting ocean of uniform density over the UPWELLING shelf stays at rest (prsgrd32 + rhs3d_tile + step3d_uv +
    step2d: no spurious pressure-gradient force in sigma coordinates when there is no stratification);
  * a free-surface wave in a flat channel oscillates with the period of the C-grid dispersion relation of
    sqrt(g*H) (step2d alone, LF-AM3 index machine);
  * omega closes continuity: W(N) = 0 to round-off for arbitrary mass fluxes, and W balances their divergence;
  * a tracer blob in a uniform current moves with the current and its content is conserved to round-off
    (pre_step3d + step3d_t + omega with the fluxes step3d_uv/step2d hand them)."""
import numpy as np
import pytest

from tests import refdrive, util

SIDES = [pytest.param(util.OracleSide, id="oracle"), pytest.param(util.HipSide, id="hip", marks=pytest.mark.gpu)]


class Run:
    """a side + the stepping state of main3d + array views with (level, j, i) indexing"""

    def __init__(self, side_cls, tag="upwelling_small", mods=None, **kw):
        self.cs = util.case_for(tag, **kw)
        self.cs.update(mods or {})
        self.side = side_cls(self.cs)
        g = self.side.g
        self.LBi, self.UBi, self.LBj, self.UBj = [int(x) for x in g["bounds"][:4]]
        self.ni, self.nj = self.side.dims()
        self.N, self.Lm, self.Mm = self.cs["N"], self.cs["Lm"], self.cs["Mm"]
        self.st = dict(iic=1, iif=1, nstp=1, nnew=1, nrhs=1, kstp=1, knew=1, krhs=1, predictor=0, indx1=1, time=0.0,
                       nfast=int(g["bounds"][58]))
        self.first = True

    def get(self, name):
        return np.array(self.side.get(name)).reshape(-1, self.nj, self.ni)

    def put(self, name, a):
        self.side.put(name, np.ascontiguousarray(a, dtype=np.float64).ravel())

    def I(self, i):
        return i - self.LBi

    def J(self, j):
        return j - self.LBj

    def interior(self, a):
        """(.., Mm, Lm) view of the rho points 1..Lm x 1..Mm"""
        return a[..., self.J(1):self.J(self.Mm) + 1, self.I(1):self.I(self.Lm) + 1]

    def step(self, n=1, after=None):
        """n passes of main3d, kernel by kernel; after = {kernel name: callable(run)} hooks"""
        for _ in range(n):
            for kern, st in refdrive.main3d_sequence(self.cs, self.st, self.first):
                self.side.call(kern, st)
                if after and kern in after:
                    after[kern](self)
            self.first = False

    def close(self):
        self.side.close()


def _no_wind(r):
    for n in ("sustr", "svstr"):
        r.put(n, 0.0 * r.get(n))


def _uniform_ts(r):
    t = r.get("t")                      # (NT*3*N, nj, ni)
    N = r.N
    t[:3 * N] = r.cs["T0"]
    t[3 * N:] = r.cs["S0"]
    r.put("t", t)


@pytest.mark.parametrize("side_cls", SIDES)
def test_unstratified_ocean_at_rest_stays_at_rest(side_cls):
    """Uniform T and S over the shelf, no wind: nothing may move.  (With the application's stratification the
    same run -- wind off -- develops the sigma-coordinate pressure-gradient currents; that contrast is asserted
    too, so the test cannot pass on a run that does nothing.)"""
    speeds = {}
    for uniform in (True, False):
        r = Run(side_cls)
        if uniform:
            _uniform_ts(r)
        r.step(5, after={"set_data": _no_wind})
        u, v, z = r.get("u"), r.get("v"), r.get("zeta")
        speeds[uniform] = max(np.abs(u).max(), np.abs(v).max())
        if uniform:
            assert np.isfinite(u).all() and np.isfinite(z).all()
            assert speeds[True] < 1e-12, speeds          # m/s after 5 steps of 300 s: round-off only
            assert np.abs(r.interior(z)).max() < 1e-12
            t = r.interior(r.get("t"))
            assert np.abs(t[:3 * r.N] - r.cs["T0"]).max() < 1e-11 and np.abs(t[3 * r.N:] - r.cs["S0"]).max() < 1e-11
        r.close()
    assert speeds[False] > 1e6 * max(speeds[True], 1e-30)


@pytest.mark.parametrize("side_cls", SIDES)
def test_omega_closes_continuity(side_cls):
    """omega_tile: W(i,j,0) = 0 and W(i,j,N) = 0 (to round-off of the flux sums) for arbitrary Huon, Hvom, and
    between them W(k) - W(k-1) = -(div of the mass fluxes of level k) + the share of the column's net divergence
    that moves the free surface, distributed in proportion to the layer thickness (omega.F:160-210)."""
    r = Run(side_cls)
    rng = np.random.default_rng(7)
    Hu, Hv = r.get("Huon"), r.get("Hvom")
    Hu = Hu + rng.normal(0.0, 50.0, Hu.shape)
    Hv = Hv + rng.normal(0.0, 50.0, Hv.shape)
    # closed southern/northern walls carry no flux; keep the periodic images consistent along xi
    Hv[:, r.J(1), :] = 0.0
    Hv[:, r.J(r.Mm + 1), :] = 0.0
    for a in (Hu, Hv):
        a[:, :, :r.I(1)] = a[:, :, r.I(r.Lm - (r.I(1) - 1)):r.I(r.Lm) + 1]
        a[:, :, r.I(r.Lm + 1):] = a[:, :, r.I(1):r.I(1) + (r.ni - r.I(r.Lm + 1))]
    r.put("Huon", Hu)
    r.put("Hvom", Hv)
    r.side.call("omega", r.st)
    W = r.get("W")                                              # (N+1, nj, ni)
    zw = r.get("z_w")
    js, je, is_, ie = r.J(1), r.J(r.Mm), r.I(1), r.I(r.Lm)
    div = (Hu[:, js:je + 1, is_ + 1:ie + 2] - Hu[:, js:je + 1, is_:ie + 1] +
           Hv[:, js + 1:je + 2, is_:ie + 1] - Hv[:, js:je + 1, is_:ie + 1])          # (N, Mm, Lm)
    Wi, zi = r.interior(W), r.interior(zw)
    scale = np.abs(div).sum(axis=0).max()
    assert np.abs(Wi[0]).max() == 0.0
    assert np.abs(Wi[-1]).max() <= 1e-13 * scale
    net = div.sum(axis=0) / (zi[-1] - zi[0])
    for k in range(1, r.N + 1):
        want = -div[k - 1] + net * (zi[k] - zi[k - 1])
        assert np.abs((Wi[k] - Wi[k - 1]) - want).max() <= 1e-12 * scale, k
    r.close()


@pytest.mark.parametrize("side_cls", SIDES)
def test_free_surface_wave_period_is_sqrt_gH(side_cls):
    """step2d alone (iif >= 2: no coupling to the 3-D forcing): a standing wave zeta = a cos(2 pi x / L) in a flat,
    non-rotating, inviscid periodic channel of depth H oscillates with the angular frequency of the C-grid
    dispersion relation, omega = (2 c / dx) sin(k dx / 2), c = sqrt(g H) -- 0.8 % below c k on this 14-point
    grid; LF-AM3 at omega*dtfast = 0.13 adds < 0.1 %."""
    r = Run(side_cls)
    g, H, amp = 9.81, 100.0, 0.01
    pm = r.get("pm")
    dx = 1.0 / pm[0, r.J(1), r.I(1)]
    L = r.Lm * dx
    dtfast = r.cs["dt"] / r.cs["ndtfast"]
    zero2 = 0.0 * r.get("h")
    r.put("h", zero2 + H)
    for n in ("f", "fomn", "rdrag", "visc2_r", "visc2_p", "rufrc", "rvfrc"):
        r.put(n, 0.0 * r.get(n))
    # VAR_RHO_2D: the pressure gradient carries (1000/rho0 + rhoS) with rhoS, rhoA the vertical means of the
    # density ANOMALY over rho0 (rho_eos.F:420-450); a uniform ocean of density rho0 has both = (rho0-1000)/rho0
    for n in ("rhoA", "rhoS"):
        r.put(n, 0.0 * r.get(n) + (r.cs["rho0"] - 1000.0) / r.cs["rho0"])
    x = (np.arange(r.ni) + r.LBi - 0.5) * dx                     # rho points: x_i = (i - 1/2) dx
    z0 = amp * np.cos(2.0 * np.pi * x / L)[None, :] * np.ones((r.nj, 1))
    r.put("zeta", np.stack([z0, z0, z0]))
    for n in ("ubar", "vbar", "rzeta", "rubar", "rvbar"):
        r.put(n, 0.0 * r.get(n))
    mode = np.cos(2.0 * np.pi * x / L)[r.I(1):r.I(r.Lm) + 1]
    st = dict(r.st, iic=3, iif=2)                                # past the start-up branches of a step
    coef, indx1 = [], 1
    nsub = 140                                                   # ~3 periods of 447 s at dtfast = 10 s
    for _ in range(nsub):
        nxt = 3 - indx1
        st.update(predictor=1, kstp=3 - indx1, knew=3, krhs=indx1)
        r.side.call("step2d", st)
        st.update(predictor=0, knew=nxt, kstp=3 - nxt, krhs=3)
        r.side.call("step2d", st)
        indx1 = nxt
        st["indx1"] = indx1
        z = r.get("zeta")[nxt - 1]
        row = z[r.J(r.Mm // 2), r.I(1):r.I(r.Lm) + 1]
        coef.append(2.0 * np.dot(row, mode) / r.Lm)
        assert np.abs(r.interior(r.get("vbar")[nxt - 1:nxt])).max() < 1e-14      # nothing along eta
    c = np.array(coef)
    assert abs(c).max() < 1.05 * amp and abs(c).max() > 0.9 * amp                 # neither growing nor damped away
    t = dtfast * (1 + np.arange(nsub))
    sign = np.sign(c)
    k = np.nonzero(sign[1:] != sign[:-1])[0]
    tz = t[k] - c[k] * (t[k + 1] - t[k]) / (c[k + 1] - c[k])                       # zero crossings
    assert len(tz) >= 5
    period = 2.0 * np.mean(np.diff(tz))
    kw = 2.0 * np.pi / L
    om = 2.0 * np.sqrt(g * H) / dx * np.sin(0.5 * kw * dx)
    assert abs(period / (2.0 * np.pi / om) - 1.0) < 2e-3, (period, 2.0 * np.pi / om, 2.0 * np.pi / (kw * np.sqrt(g * H)))
    r.close()


@pytest.mark.parametrize("side_cls", SIDES)
@pytest.mark.parametrize("scheme", ["U3", "HSIMT", "MPDATA"])
def test_tracer_blob_rides_the_current_and_is_conserved(side_cls, scheme):
    """A salinity blob in a uniform along-channel current over a flat bottom (no rotation, friction, wind or mixing):
    after n steps its centre has moved U*n*dt (to a fraction of a cell), its content sum(Hz*S*area) is the initial
    one to round-off, and -- HSIMT and MPDATA being monotone -- no new extrema appear with those schemes."""
    hadv, vadv = {"U3": (("U3", "U3"), ("C4", "C4")), "HSIMT": (("HSIMT", "HSIMT"), ("HSIMT", "HSIMT")),
                  "MPDATA": (("MPDATA", "MPDATA"), ("MPDATA", "MPDATA"))}[scheme]
    r = Run(side_cls, hadv=hadv, vadv=vadv)
    H, U = 100.0, 0.2
    N = r.N
    r.put("h", 0.0 * r.get("h") + H)
    for n in ("f", "fomn", "rdrag", "visc2_r", "visc2_p", "diff2", "Akt", "Akv"):
        r.put(n, 0.0 * r.get(n))
    pm, pn = r.get("pm"), r.get("pn")
    dx = 1.0 / pm[0, r.J(1), r.I(1)]
    x = (np.arange(r.ni) + r.LBi - 0.5) * dx
    L = r.Lm * dx
    xc = 0.35 * L
    d = (x - xc + 0.5 * L) % L - 0.5 * L                          # periodic distance
    blob = np.exp(-(d / (2.0 * dx)) ** 2)[None, None, :] * np.ones((N, r.nj, 1))
    t = r.get("t")
    t[:3 * N] = r.cs["T0"]
    for lev in range(3):
        t[3 * N + lev * N:3 * N + (lev + 1) * N] = r.cs["S0"] + blob
    r.put("t", t)
    u = r.get("u")
    r.put("u", 0.0 * u + U)
    r.put("v", 0.0 * r.get("v"))
    ub = r.get("ubar")
    r.put("ubar", 0.0 * ub + U)
    r.put("vbar", 0.0 * r.get("vbar"))
    r.put("zeta", 0.0 * r.get("zeta"))
    r.put("Zt_avg1", 0.0 * r.get("Zt_avg1"))

    def content(run, lev):
        Hz = run.interior(run.get("Hz"))
        S = run.interior(run.get("t")[3 * N + lev * N:3 * N + (lev + 1) * N])
        area = 1.0 / (run.interior(pm) * run.interior(pn))
        return float((Hz * (S - run.cs["S0"]) * area).sum()), S

    def flat(run):
        # the set-up recomputes depths from h on the first pass (set_depth); keep the flow the test prescribes
        _no_wind(run)

    r.side.call("set_depth", r.st)
    c0, S0 = content(r, 0)
    nsteps = 10
    r.step(nsteps, after={"set_data": flat})
    newest = (1 + (r.st["iic"] - 1) % 2) - 1                       # nstp of the NEXT step holds the latest state
    c1, S1 = content(r, newest)
    assert abs(c1 - c0) <= 1e-11 * abs(c0), (c0, c1)
    prof0, prof1 = S0[N // 2, r.Mm // 2] - r.cs["S0"], S1[N // 2, r.Mm // 2] - r.cs["S0"]
    xi = x[r.I(1):r.I(r.Lm) + 1]
    ang0 = np.angle(np.sum(prof0 * np.exp(2j * np.pi * xi / L)))
    ang1 = np.angle(np.sum(prof1 * np.exp(2j * np.pi * xi / L)))
    moved = ((ang1 - ang0) % (2.0 * np.pi)) * L / (2.0 * np.pi)
    assert abs(moved - U * nsteps * r.cs["dt"]) < 0.015 * dx, (moved, U * nsteps * r.cs["dt"], dx)   # 600 m in 10 steps; U3 599.2, HSIMT 601.3, MPDATA 591.5
    uu = r.interior(r.get("u")[newest * N:(newest + 1) * N])
    assert np.abs(uu - U).max() < 1e-9                              # the current itself is a steady solution
    if scheme in ("HSIMT", "MPDATA"):
        assert prof1.min() > -1e-9 and prof1.max() < prof0.max() + 1e-9
    r.close()
