// k_prs4x.h -- the finite-volume pressure Jacobians with a reconstructed vertical density profile (Shchepetkin & McWilliams 2003):
//   prsgrd44.h:224-508 (PJ_GRADPQ4: quartic reconstruction, power-law reconciliation)   ROMS_PRSGRD44
//   prsgrd42.h:227-482 (PJ_GRADPQ2: parabolic WENO, PPM-style limiter, second pass)     ROMS_PRSGRD42
// prsgrd.F:16-19 selects them in front of every other scheme.  Not a BASELINE path: straight restatements, statement for statement
// (the bits of the reference; oracle/orc_prs4x.c is the same text in C), in three steps:
//   k_prs4x_col     one thread per rho column: the side limits aL, aR, dL, dR and (44) r1 of the column in work arrays, then
//                   r, (44) d, P, FX -- every recurrence of the scheme runs along k inside one thread
//   k_prs44_grad    one thread per velocity column: FC along k carried in a register, ru | rv
//   k_prs42_grad1   ... the first pass of prsgrd42 (ru on IstrU-1:Iend+1, rv on JstrV-1:Jend+1 as the reference computes them),
//                   stored in ru, rv AND in two work arrays
//   k_prs42_grad2   the second pass (:417-480) from those copies; rv(Iend+1,j,k) -- which no pass computes (prsgrd42.h:449) -- is
//                   read from rv itself, where nothing has written: a single tile only (roms_hip_create refuses the rest)
// Work arrays (roms_ctx.h: wrk3, all (0:N)): [1] P, [2] FX (and, until FX is formed, prsgrd44's r1), [3] r, [4] d, [11] aL |
// first-pass ru, [12] aR | first-pass rv, [5] dL, [0] dR -- none of them in use beside prsgrd in the reference-order schedule
// these options keep (wvelocity's [10] and uv3dmix2_s's [6..9] may run on the side stream at the same time).
// Neither file defines NEUMANN (#undef in their first line).
#pragma once
#include "k_diag3d.h"

KDEV double prs_ppm_rr(double deltaR, double deltaL) {          // prsgrd42.h:337-345 (= :367-375, :410-418)
  if ((deltaR * deltaL) < 0.0) return 0.0;
  if (fabs(deltaR) > (2.0 * fabs(deltaL))) return 3.0 * deltaL;
  if (fabs(deltaL) > (2.0 * fabs(deltaR))) return 3.0 * deltaR;
  return deltaR + deltaL;
}

// p1 = 42 | 44
THREAD_KERNEL(k_prs4x_col, KArgs) {
  (void)gz;
  const DGrid &G = a.G;
  const Fields &F = a.Fv;
  const TB &B = G.T;
  const bool q4 = a.p1 == 44;
  const int i = (q4 ? B.IstrU - 1 : B.IstrU - 2) + gx, j = (q4 ? B.JstrV - 1 : B.JstrV - 2) + gy, N = G.N;
  if (i > (q4 ? B.Iend : B.Iend + 1) || j > (q4 ? B.Jend : B.Jend + 1)) return;
  const double eps = 1.0E-8;
  const double *rho = F.rho, *Hz = F.Hz;
  double *P = F.wrk3[1], *FX = F.wrk3[2], *r = F.wrk3[3], *d = F.wrk3[4];
  double *aL = F.wrk3[11], *aR = F.wrk3[12], *dL = F.wrk3[5], *dR = F.wrk3[0], *r1 = q4 ? FX : r;     // (r1 is dead when FX is formed)
#define RH(k_) rho[X3(i, j, k_)]
#define HZ(k_) Hz[X3(i, j, k_)]
#define W_(A_, k_) A_[XW(i, j, k_)]
  // D(k) = (rho(k+1)-rho(k))/(Hz(k+1)+Hz(k)): prsgrd42's FC(i,k) :240, prsgrd44's d(i,j,k) :237 (FC*(...), FC = 1/(Hz+Hz))
#define DG(k_) (q4 ? (1.0 / (HZ((k_) + 1) + HZ(k_))) * (RH((k_) + 1) - RH(k_)) : (RH((k_) + 1) - RH(k_)) / (HZ((k_) + 1) + HZ(k_)))
  for (int k = 2; k <= N - 1; k++) {                                   // :246-284 of both
    const double Dk = DG(k), Dkm = DG(k - 1);
    double deltaR = HZ(k) * Dk;
    double deltaL = HZ(k) * Dkm;
    if ((deltaR * deltaL) < 0.0) { deltaR = 0.0; deltaL = 0.0; }
    double cff = HZ(k - 1) + 2.0 * HZ(k) + HZ(k + 1);
    const double cffR = cff * Dk;
    const double cffL = cff * Dkm;
    if (fabs(deltaR) > fabs(cffL)) deltaR = cffL;
    if (fabs(deltaL) > fabs(cffR)) deltaL = cffR;
    cff = (deltaR - deltaL) / (HZ(k - 1) + HZ(k) + HZ(k + 1));
    deltaR = deltaR - cff * HZ(k + 1);
    deltaL = deltaL + cff * HZ(k - 1);
    W_(aR, k) = RH(k) + deltaR;
    W_(aL, k) = RH(k) - deltaL;
    W_(dR, k) = (2.0 * deltaR - deltaL) * (2.0 * deltaR - deltaL);
    W_(dL, k) = (2.0 * deltaL - deltaR) * (2.0 * deltaL - deltaR);
  }
  {
    const double aLN = W_(aR, N - 1), aRN = 2.0 * RH(N) - aLN;
    W_(aL, N) = aLN; W_(aR, N) = aRN;
    { const double q = 2.0 * aRN + aLN - 3.0 * RH(N); W_(dR, N) = q * q; }
    { const double q = 3.0 * RH(N) - 2.0 * aLN - aRN; W_(dL, N) = q * q; }
    const double aR1 = W_(aL, 2), aL1 = 2.0 * RH(1) - aR1;
    W_(aR, 1) = aR1; W_(aL, 1) = aL1;
    { const double q = 2.0 * aR1 + aL1 - 3.0 * RH(1); W_(dR, 1) = q * q; }
    { const double q = 3.0 * RH(1) - 2.0 * aL1 - aR1; W_(dL, 1) = q * q; }
  }
  for (int k = 1; k <= N - 1; k++) {                                   // the WENO reconciliation: r (42) | r1 (44)
    const double deltaL = KMAX(W_(dL, k), eps);
    const double deltaR = KMAX(W_(dR, k + 1), eps);
    W_(r1, k) = (deltaR * W_(aR, k) + deltaL * W_(aL, k + 1)) / (deltaR + deltaL);
  }
  W_(r1, N) = 2.0 * RH(N) - W_(r1, N - 1);
  W_(r1, 0) = 2.0 * RH(1) - W_(r1, 1);
  if (q4) {
    for (int k = 1; k <= N; k++) {                                     // prsgrd44.h:316-348
      const double deltaR = W_(r1, k) - RH(k);
      const double deltaL = RH(k) - W_(r1, k - 1);
      double cff = deltaR * deltaL;
      if (cff > eps) cff = (deltaR + deltaL) / cff;
      else cff = 0.0;
      double cffL = cff * deltaL;
      double cffR = cff * deltaR;
      if (cffL > 3.0) {
        cffL = cffL * deltaL;
        cffR = 0.0;
      } else if (cffR > 3.0) {
        cffL = 0.0;
        cffR = cffR * deltaR;
      } else {
        cffL = 4.0 * deltaL - 2.0 * deltaR;
        cffR = 4.0 * deltaR - 2.0 * deltaL;
      }
      cff = 1.0 / HZ(k);
      W_(dR, k) = cff * cffR;
      W_(dL, k) = cff * cffL;
    }
    for (int k = N - 1; k >= 1; k--) {                                 // :361-391
      const double FCk = 1.0 / (HZ(k + 1) + HZ(k));
      double dk = FCk * (HZ(k + 1) * W_(dL, k + 1) + HZ(k) * W_(dR, k));
      const double cffR = 8.0 * (W_(dR, k) + 2.0 * W_(dL, k));
      const double cffL = 8.0 * (W_(dL, k + 1) + 2.0 * W_(dR, k + 1));
      if (fabs(dk) > fabs(cffR)) dk = cffR;
      if (fabs(dk) > fabs(cffL)) dk = cffL;
      W_(d, k) = dk;
      double Hdd, rr;
      if ((W_(dL, k + 1) - W_(dR, k)) * (RH(k + 1) - RH(k)) > 0.0) {
        Hdd = HZ(k) * (dk - W_(dR, k));
        rr = RH(k) - W_(r1, k - 1);
      } else {
        Hdd = HZ(k + 1) * (W_(dL, k + 1) - dk);
        rr = W_(r1, k + 1) - RH(k + 1);
      }
      rr = fabs(rr);
      double Ampl = 0.2 * Hdd * rr;
      Hdd = fabs(Hdd);
      const double cff = rr * rr + 0.0763636363636363636 * Hdd * (rr + 0.004329004329004329 * Hdd);
      if (cff > eps) Ampl = Ampl * (rr + 0.0363636363636363636 * Hdd) / cff;
      else Ampl = 0.0;
      W_(r, k) = W_(r1, k) + Ampl;
    }
    W_(r, 0) = 2.0 * RH(1) - W_(r, 1);
    W_(r, N) = 2.0 * RH(N) - W_(r, N - 1);
    W_(d, 0) = W_(d, 1);
    W_(d, N) = W_(d, N - 1);
  }
  // pressure and the lateral pressure force of the cell: prsgrd42.h:321-338, prsgrd44.h:422-431
  double Pk = 0.0;
  W_(P, N) = Pk;
  const double cff2 = 1.0 / 6.0, cff3 = 1.0 / 12.0;
  for (int k = N; k >= 1; k--) {
    const double Pk1 = Pk + HZ(k) * RH(k);
    W_(P, k - 1) = Pk1;
    if (q4) {
      W_(FX, k) = 0.5 * HZ(k) * (Pk + Pk1 + 0.2 * HZ(k) * (W_(r, k) - W_(r, k - 1) - cff3 * HZ(k) * (W_(d, k) + W_(d, k - 1))));
    } else {
      const double deltaR = W_(r, k) - RH(k);
      const double deltaL = RH(k) - W_(r, k - 1);
      const double rr = prs_ppm_rr(deltaR, deltaL);
      W_(FX, k) = 0.5 * HZ(k) * (Pk + Pk1 + cff2 * rr * HZ(k));
    }
    Pk = Pk1;
  }
#undef DG
#undef W_
#undef HZ
#undef RH
}
THREAD_GLOBAL(k_prs4x_col, KArgs)

// prsgrd44.h:436-506; grid.z = 0: ru on (IstrU:Iend, Jstr:Jend), 1: rv on (Istr:Iend, JstrV:Jend)
THREAD_KERNEL(k_prs44_grad, KArgs) {
  const DGrid &G = a.G;
  const Fields &F = a.Fv;
  const TB &B = G.T;
  const int dir = gz;
  const int i = (dir == 0 ? B.IstrU : B.Istr) + gx, j = (dir == 0 ? B.Jstr : B.JstrV) + gy;
  if (i > B.Iend || j > B.Jend) return;
  const int di = dir == 0 ? 1 : 0, dj = dir == 0 ? 0 : 1, N = G.N;
  const double eps = 1.0E-8;
  const double cff = 0.5 * G.g, cff1 = G.g / G.rho0, cff2 = 1.0 / 6.0, cff3 = 1.0 / 12.0;
  const double *z_w = F.z_w, *Hz = F.Hz;
  const double *P = F.wrk3[1], *FX = F.wrk3[2], *r = F.wrk3[3], *d = F.wrk3[4];
  double *rq = (dir == 0 ? F.ru : F.rv) + (size_t)(G.nrhs - 1) * G.nij * (N + 1);
  const double omn = (dir == 0 ? F.on_u : F.om_v)[X2(i, j)];
  const double dzeta = z_w[XW(i - di, j - dj, N)] - z_w[XW(i, j, N)];
  double FCk = 0.0;
  for (int k = N; k >= 1; k--) {
    const size_t xc = XW(i, j, k - 1), xm = XW(i - di, j - dj, k - 1);
    const double dh = z_w[xc] - z_w[xm];
    const double delP = P[xm] - P[xc];
    double rr = 0.5 * dh * (r[xc] + r[xm] - cff2 * dh * (d[xc] - d[xm]));
    double limtr = 2.0 * delP * rr;
    rr = rr * rr + delP * delP;
    if (limtr > eps * rr) limtr = limtr / rr;
    else limtr = 0.0;
    const double FC1 = 0.5 * dh * (P[xc] + P[xm] + limtr * 0.2 * dh * (r[xc] - r[xm] - cff3 * dh * (d[xc] + d[xm])));
    rq[XW(i, j, k)] = (cff * (Hz[X3(i - di, j - dj, k)] + Hz[X3(i, j, k)]) * dzeta +
                       cff1 * (FX[XW(i - di, j - dj, k)] - FX[XW(i, j, k)] + FCk - FC1)) * omn;
    FCk = FC1;
  }
}
THREAD_GLOBAL(k_prs44_grad, KArgs)

// prsgrd42.h:344-414, the first pass; grid.z = 0: ru on (IstrU-1:Iend+1, Jstr:Jend), 1: rv on (Istr:Iend, JstrV-1:Jend+1)
THREAD_KERNEL(k_prs42_grad1, KArgs) {
  const DGrid &G = a.G;
  const Fields &F = a.Fv;
  const TB &B = G.T;
  const int dir = gz;
  const int i = (dir == 0 ? B.IstrU - 1 : B.Istr) + gx, j = (dir == 0 ? B.Jstr : B.JstrV - 1) + gy;
  if (i > (dir == 0 ? B.Iend + 1 : B.Iend) || j > (dir == 0 ? B.Jend : B.Jend + 1)) return;
  const int di = dir == 0 ? 1 : 0, dj = dir == 0 ? 0 : 1, N = G.N;
  const double cff2 = 1.0 / 6.0;
  const double *z_w = F.z_w, *Hz = F.Hz;
  const double *P = F.wrk3[1], *FX = F.wrk3[2], *r = F.wrk3[3];
  double *rq = (dir == 0 ? F.ru : F.rv) + (size_t)(G.nrhs - 1) * G.nij * (N + 1);
  double *T = dir == 0 ? (double *)F.wrk3[11] : (double *)F.wrk3[12];
  const double msk = G.masking ? (dir == 0 ? G.umask : G.vmask)[X2(i, j)] : 1.0;
  double FCk = 0.0;
  for (int k = N; k >= 1; k--) {
    const size_t xc = XW(i, j, k - 1), xm = XW(i - di, j - dj, k - 1);
    const double delP = P[xm] - P[xc];
    const double dh = z_w[xc] - z_w[xm];
    const double deltaR = dh * r[xc] - delP;
    const double deltaL = delP - dh * r[xm];
    const double rr = prs_ppm_rr(deltaR, deltaL);
    const double FC1 = 0.5 * dh * (P[xc] + P[xm] + cff2 * rr);
    double v = 2.0 * (FX[XW(i - di, j - dj, k)] - FX[XW(i, j, k)] + FCk - FC1) / (Hz[X3(i - di, j - dj, k)] + Hz[X3(i, j, k)]);
    if (G.masking) v = v * msk;
    rq[XW(i, j, k)] = v;
    T[XW(i, j, k)] = v;
    FCk = FC1;
  }
}
THREAD_GLOBAL(k_prs42_grad1, KArgs)

// prsgrd42.h:417-480, the second pass; grid.z = 0: ru on (IstrU:Iend, Jstr:Jend), 1: rv on (Istr:Iend, JstrV:Jend)
THREAD_KERNEL(k_prs42_grad2, KArgs) {
  const DGrid &G = a.G;
  const Fields &F = a.Fv;
  const TB &B = G.T;
  const int dir = gz;
  const int i = (dir == 0 ? B.IstrU : B.Istr) + gx, j = (dir == 0 ? B.Jstr : B.JstrV) + gy;
  if (i > B.Iend || j > B.Jend) return;
  const int di = dir == 0 ? 1 : 0, dj = dir == 0 ? 0 : 1, N = G.N;
  const double rr = G.g / (24.0 * G.rho0), cff = 0.5 * G.g, cff1 = 0.5 * G.g / G.rho0;
  const double *z_w = F.z_w, *Hz = F.Hz;
  double *rq = (dir == 0 ? F.ru : F.rv) + (size_t)(G.nrhs - 1) * G.nij * (N + 1);
  const double *T = dir == 0 ? (const double *)F.wrk3[11] : (const double *)F.wrk3[12];
  // the first-pass values around the column: here (0), at i+1 (E), and "behind" = i-1 for ru, j-1 for rv (prsgrd42.h:423-426, :449-455)
  // -- rv(Iend+1,j,k): the array itself, no pass computes it
  const bool stale = dir == 1 && i + 1 > B.Iend;
#define T0(k_) T[XW(i, j, k_)]
#define TE(k_) (stale ? rq[XW(i + 1, j, k_)] : T[XW(i + 1, j, k_)])
#define TB_(k_) T[XW(i - di, j - dj, k_)]
  const double omn = (dir == 0 ? F.on_u : F.om_v)[X2(i, j)];
  const double dzeta = z_w[XW(i - di, j - dj, N)] - z_w[XW(i, j, N)];
  double FCm;                                                              // FC(k-1)
  {
    const double dh = rr * (z_w[XW(i, j, 0)] - z_w[XW(i - di, j - dj, 0)]);
    FCm = KMAX(dh, 0.0) * (T0(1) - TB_(1)) + KMIN(dh, 0.0) * (TE(1) - T0(1));
  }
  for (int k = 1; k <= N; k++) {
    double FCk = 0.0;
    if (k < N) {
      const double dh = rr * (z_w[XW(i, j, k)] - z_w[XW(i - di, j - dj, k)]);
      FCk = KMAX(dh, 0.0) * (T0(k + 1) + TE(k) - T0(k) - TB_(k + 1)) + KMIN(dh, 0.0) * (T0(k) + TE(k + 1) - T0(k + 1) - TB_(k));
    }
    rq[XW(i, j, k)] = (cff * dzeta + cff1 * T0(k)) * (Hz[X3(i - di, j - dj, k)] + Hz[X3(i, j, k)]) * omn + (FCk - FCm) * omn;
    FCm = FCk;
  }
#undef T0
#undef TE
#undef TB_
}
THREAD_GLOBAL(k_prs42_grad2, KArgs)
