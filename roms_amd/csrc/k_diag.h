// k_diag.h -- global diagnostics of diag_tile, ROMS/Nonlinear/diag.F:84-560: volume-averaged
// kinetic/potential energy, volume, maximum Courant number (with its location) and maximum speed.
//
// The reference sums column values first along eta (j) for every xi (i), then along xi, to limit
// round-off (:289-325).  The same order is kept so that the result is reproducible bit-for-bit:
//   k_diag_col  one thread per column (k = N..1 as in the reference)
//   k_diag_row  one thread per i: ordered sum over j, ordered max
//   k_diag_fin  one thread: ordered sum over i
#pragma once
#include "roms_ctx.h"
#include "k_diag3d.h"

struct DiagArgs {
  DGrid G;
  Fields F;
  double *col;   // 8 planes of nij doubles: ke2d pe2d C Cu Cv Cw kmax speed
  double *row;   // 12 rows of ni doubles
  double *out;   // 16 doubles
};

THREAD_KERNEL(k_diag_col, DiagArgs) {
  (void)gz;
  const DGrid &G = a.G;
  const Fields &F = a.F;
  const int i = G.T.Istr + gx, j = G.T.Jstr + gy, N = G.N, idia = G.nstp;
  const double g = G.g, dt = G.dt;
  const double *u = F.u + (size_t)(idia - 1) * G.nij * N, *v = F.v + (size_t)(idia - 1) * G.nij * N;
  double ke = 0.0, pe = 0.5 * g * F.z_w[XW(i, j, N)] * F.z_w[XW(i, j, N)];
  const double cff = g / G.rho0;
  double maxC = 0.0, mCu = 0.0, mCv = 0.0, mCw = 0.0, spd = 0.0;
  int mk = 0;
  for (int k = N; k >= 1; k--) {
    const double u2v2 = u[X3(i, j, k)] * u[X3(i, j, k)] + u[X3(i + 1, j, k)] * u[X3(i + 1, j, k)] +
                        v[X3(i, j, k)] * v[X3(i, j, k)] + v[X3(i, j + 1, k)] * v[X3(i, j + 1, k)];
    ke = ke + F.Hz[X3(i, j, k)] * 0.25 * u2v2;
    pe = pe + cff * F.Hz[X3(i, j, k)] * (F.rho[X3(i, j, k)] + 1000.0) * (F.z_r[X3(i, j, k)] - F.z_w[XW(i, j, 0)]);
    const double Cu = 0.5 * fabs(u[X3(i, j, k)] + u[X3(i + 1, j, k)]) * dt * F.pm[X2(i, j)];
    const double Cv = 0.5 * fabs(v[X3(i, j, k)] + v[X3(i, j + 1, k)]) * dt * F.pn[X2(i, j)];
    const double Cw = 0.5 * fabs(F.wvel[XW(i, j, k - 1)] + F.wvel[XW(i, j, k)]) * dt / F.Hz[X3(i, j, k)];
    const double C = Cu + Cv + Cw;
    if (C > maxC) { maxC = C; mCu = Cu; mCv = Cv; mCw = Cw; mk = k; }
    const double s = sqrt(0.5 * u2v2);
    if (s > spd) spd = s;
  }
  double *col = a.col;
  col[X2(i, j)] = ke;
  col[X2(i, j) + G.nij] = pe;
  col[X2(i, j) + 2 * G.nij] = maxC;
  col[X2(i, j) + 3 * G.nij] = mCu;
  col[X2(i, j) + 4 * G.nij] = mCv;
  col[X2(i, j) + 5 * G.nij] = mCw;
  col[X2(i, j) + 6 * G.nij] = (double)mk;
  col[X2(i, j) + 7 * G.nij] = spd;
}
THREAD_GLOBAL(k_diag_col, DiagArgs)

// "first maximum in the reference's loop order (j ascending, k descending, i ascending)"
KDEV bool diag_better(double C, int j, int k, int i, double C0, int j0, int k0, int i0) {
  if (C != C0) return C > C0;
  if (j != j0) return j < j0;
  if (k != k0) return k > k0;
  return i < i0;
}

// one thread per i of Istr:Iend (index space nx = Iend-Istr+1)
THREAD_KERNEL(k_diag_row, DiagArgs) {
  (void)gy; (void)gz;
  const DGrid &G = a.G;
  const Fields &F = a.F;
  const int i = G.T.Istr + gx, N = G.N;
  const double *col = a.col;
  double vol = 0.0, pes = 0.0, kes = 0.0, spd = 0.0;
  double bC = 0.0, bCu = 0.0, bCv = 0.0, bCw = 0.0;
  int bj = 0, bk = 0;
  for (int j = G.T.Jstr; j <= G.T.Jend; j++) {
    const double omn = F.omn[X2(i, j)];
    vol = vol + omn * (F.z_w[XW(i, j, N)] - F.z_w[XW(i, j, 0)]);
    pes = pes + omn * col[X2(i, j) + G.nij];
    kes = kes + omn * col[X2(i, j)];
    const double C = col[X2(i, j) + 2 * G.nij];
    const int k = (int)col[X2(i, j) + 6 * G.nij];
    if (C > 0.0 && (bC == 0.0 || diag_better(C, j, k, i, bC, bj, bk, i))) {
      bC = C; bCu = col[X2(i, j) + 3 * G.nij]; bCv = col[X2(i, j) + 4 * G.nij]; bCw = col[X2(i, j) + 5 * G.nij];
      bj = j; bk = k;
    }
    const double s = col[X2(i, j) + 7 * G.nij];
    if (s > spd) spd = s;
  }
  double *row = a.row;
  const int ni = G.ni, ii = i - G.LBi;
  row[ii] = vol; row[ii + ni] = pes; row[ii + 2 * ni] = kes; row[ii + 3 * ni] = spd;
  row[ii + 4 * ni] = bC; row[ii + 5 * ni] = bCu; row[ii + 6 * ni] = bCv; row[ii + 7 * ni] = bCw;
  row[ii + 8 * ni] = (double)bj; row[ii + 9 * ni] = (double)bk;
}
THREAD_GLOBAL(k_diag_row, DiagArgs)

THREAD_KERNEL(k_diag_fin, DiagArgs) {
  (void)gx; (void)gy; (void)gz;
  const DGrid &G = a.G;
  const double *row = a.row;
  const int ni = G.ni;
  double vol = 0.0, pes = 0.0, kes = 0.0, spd = -1.0E+20;
  double bC = 0.0, bCu = 0.0, bCv = 0.0, bCw = 0.0;
  int bi = 0, bj = 0, bk = 0;
  for (int i = G.T.Istr; i <= G.T.Iend; i++) {
    const int ii = i - G.LBi;
    vol = vol + row[ii];
    pes = pes + row[ii + ni];
    kes = kes + row[ii + 2 * ni];
    if (row[ii + 3 * ni] > spd) spd = row[ii + 3 * ni];
    const double C = row[ii + 4 * ni];
    const int j = (int)row[ii + 8 * ni], k = (int)row[ii + 9 * ni];
    if (C > 0.0 && (bC == 0.0 || diag_better(C, j, k, i, bC, bj, bk, bi))) {
      bC = C; bCu = row[ii + 5 * ni]; bCv = row[ii + 6 * ni]; bCw = row[ii + 7 * ni];
      bi = i; bj = j; bk = k;
    }
  }
  double *out = a.out;
  const double avgke = kes / vol, avgpe = pes / vol;
  out[0] = avgke; out[1] = avgpe; out[2] = avgke + avgpe; out[3] = vol; out[4] = spd;
  out[5] = bCu; out[6] = bCv; out[7] = bCw; out[8] = (double)bi; out[9] = (double)bj; out[10] = (double)bk;
  out[11] = bC;
  out[12] = kes; out[13] = pes;   // un-normalised sums: combined over tiles by the caller
}
THREAD_GLOBAL(k_diag_fin, DiagArgs)
