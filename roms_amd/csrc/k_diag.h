// k_diag.h -- global diagnostics of diag_tile, ROMS/Nonlinear/diag.F:84-560: volume-averaged
// kinetic/potential energy, volume, maximum Courant number (with its location) and maximum speed.
//
// The reference sums column values first along eta (j) for every xi (i), then along xi, to limit
// round-off (:289-325).  The same order is kept so that the result is reproducible bit-for-bit:
//   k_diag_col  one thread per column (k = N..1 as in the reference); writes the products
//               omn*ke, omn*pe, omn*dz the row sums need
//   k_diag_row  one block per 64 values of i: chunks of rows are staged in LDS by all threads,
//               then one thread per i adds them in j order; the maxima (order-free) use all threads
//   k_diag_fin  one block: row sums staged in LDS, one thread adds them in i order
#pragma once
#include "roms_ctx.h"
#include "k_diag3d.h"

struct DiagArgs {
  DGrid G;
  Fields Fv;         // the array pointers, by value (a table in device memory would cost every kernel one more dependent round trip)
  double *col;   // 8 planes of nij doubles: ke2d pe2d C Cu Cv Cw kmax speed
  double *row;   // 12 rows of ni doubles
  double *out;   // 16 doubles
};

THREAD_KERNEL(k_diag_col, DiagArgs) {
  (void)gz;
  const DGrid &G = a.G;
  const Fields &F = a.Fv;
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
  const double omn = F.omn[X2(i, j)];
  col[X2(i, j)] = omn * ke;
  col[X2(i, j) + G.nij] = omn * pe;
  col[X2(i, j) + 8 * G.nij] = omn * (F.z_w[XW(i, j, N)] - F.z_w[XW(i, j, 0)]);
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

// ---- row stage: block bx owns i = Istr + 64*bx .. +63 -------------------------------------
#define DIAG_IW 64
#define DIAG_JC 16
struct DiagBest { double C; int j, k; };
COOP_KERNEL(k_diag_row, DiagArgs) {
  (void)by; (void)bz;
  const DGrid &G = a.G;
  const TB &T = G.T;
  const double *col = a.col;
  const int i0 = T.Istr + DIAG_IW * bx;
  const int nw = KMIN(DIAG_IW, T.Iend - i0 + 1);
  double *sK = lds, *sP = lds + DIAG_IW * DIAG_JC, *sV = lds + 2 * DIAG_IW * DIAG_JC;
  double vol = 0.0, pes = 0.0, kes = 0.0;
  for (int j0 = T.Jstr; j0 <= T.Jend; j0 += DIAG_JC) {
    const int nj = KMIN(DIAG_JC, T.Jend - j0 + 1);
    KSYNC();
    KLOOP2(di, dj, 0, nw - 1, 0, nj - 1) {
      const size_t q = X2(i0 + di, j0 + dj);
      sK[dj * DIAG_IW + di] = col[q];
      sP[dj * DIAG_IW + di] = col[q + G.nij];
      sV[dj * DIAG_IW + di] = col[q + 8 * G.nij];
    }
    KSYNC();
    // ordered accumulation: thread di adds rows j0..j0+nj-1 of column i0+di
#ifdef ROMS_CPU_EMU
    // serial emulation: the accumulators of all 64 columns live in LDS
    double *acc = lds + 3 * DIAG_IW * DIAG_JC;
    if (j0 == T.Jstr) for (int di = 0; di < 3 * DIAG_IW; di++) acc[di] = 0.0;
    for (int di = 0; di < nw; di++)
      for (int dj = 0; dj < nj; dj++) {
        acc[di] = acc[di] + sV[dj * DIAG_IW + di];
        acc[DIAG_IW + di] = acc[DIAG_IW + di] + sP[dj * DIAG_IW + di];
        acc[2 * DIAG_IW + di] = acc[2 * DIAG_IW + di] + sK[dj * DIAG_IW + di];
      }
#else
    if (KTID < nw) {
      // all LDS reads of the chunk are issued before the (ordered, dependent) additions
      double rv[DIAG_JC], rp[DIAG_JC], rk[DIAG_JC];
#pragma unroll
      for (int dj = 0; dj < DIAG_JC; dj++) {
        const int d = dj < nj ? dj : 0;
        rv[dj] = sV[d * DIAG_IW + KTID]; rp[dj] = sP[d * DIAG_IW + KTID]; rk[dj] = sK[d * DIAG_IW + KTID];
      }
#pragma unroll
      for (int dj = 0; dj < DIAG_JC; dj++)
        if (dj < nj) { vol = vol + rv[dj]; pes = pes + rp[dj]; kes = kes + rk[dj]; }
    }
#endif
  }
  double *row = a.row;
  const int ni = G.ni;
#ifdef ROMS_CPU_EMU
  {
    const double *acc = lds + 3 * DIAG_IW * DIAG_JC;
    for (int di = 0; di < nw; di++) {
      const int ii = i0 + di - G.LBi;
      row[ii] = acc[di]; row[ii + ni] = acc[DIAG_IW + di]; row[ii + 2 * ni] = acc[2 * DIAG_IW + di];
    }
  }
#else
  if (KTID < nw) {
    const int ii = i0 + KTID - G.LBi;
    row[ii] = vol; row[ii + ni] = pes; row[ii + 2 * ni] = kes;
  }
#endif
  // maxima: every thread scans a strided subset of rows of one column, then the partial results of
  // a column are merged (the ordering relation diag_better is total, so the merge order is free)
  KSYNC();
  const int nsub = KMAX(1, KNT / DIAG_IW);
  double *mC = lds;                               // [nsub][64] C, spd ; ints packed as doubles
  double *mS = lds + 4 * DIAG_IW, *mJ = lds + 8 * DIAG_IW, *mK = lds + 12 * DIAG_IW;
  for (int t = KTID; t < nsub * DIAG_IW; t += KNT) {
    const int di = t % DIAG_IW, sub = t / DIAG_IW;
    double bC = 0.0, spd = 0.0;
    int bj = 0, bk = 0;
    if (di < nw) {
      const int i = i0 + di;
      for (int j = T.Jstr + sub; j <= T.Jend; j += nsub) {
        const double C = col[X2(i, j) + 2 * G.nij];
        const int k = (int)col[X2(i, j) + 6 * G.nij];
        if (C > 0.0 && (bC == 0.0 || diag_better(C, j, k, i, bC, bj, bk, i))) { bC = C; bj = j; bk = k; }
        const double s = col[X2(i, j) + 7 * G.nij];
        if (s > spd) spd = s;
      }
    }
    mC[sub * DIAG_IW + di] = bC; mS[sub * DIAG_IW + di] = spd;
    mJ[sub * DIAG_IW + di] = (double)bj; mK[sub * DIAG_IW + di] = (double)bk;
  }
  KSYNC();
  for (int di = KTID; di < nw; di += KNT) {
    const int i = i0 + di;
    double bC = 0.0, spd = 0.0;
    int bj = 0, bk = 0;
    for (int sub = 0; sub < nsub; sub++) {
      const double C = mC[sub * DIAG_IW + di];
      const int j = (int)mJ[sub * DIAG_IW + di], k = (int)mK[sub * DIAG_IW + di];
      if (C > 0.0 && (bC == 0.0 || diag_better(C, j, k, i, bC, bj, bk, i))) { bC = C; bj = j; bk = k; }
      if (mS[sub * DIAG_IW + di] > spd) spd = mS[sub * DIAG_IW + di];
    }
    const int ii = i - G.LBi;
    row[ii + 3 * ni] = spd;
    row[ii + 4 * ni] = bC;
    if (bC > 0.0) {
      row[ii + 5 * ni] = col[X2(i, bj) + 3 * G.nij]; row[ii + 6 * ni] = col[X2(i, bj) + 4 * G.nij];
      row[ii + 7 * ni] = col[X2(i, bj) + 5 * G.nij];
    } else {
      row[ii + 5 * ni] = 0.0; row[ii + 6 * ni] = 0.0; row[ii + 7 * ni] = 0.0;
    }
    row[ii + 8 * ni] = (double)bj; row[ii + 9 * ni] = (double)bk;
  }
}
COOP_GLOBAL(k_diag_row, DiagArgs)
#define DIAG_ROW_LDS (3 * DIAG_IW * DIAG_JC + 16 * DIAG_IW)

// ---- final stage: one block ----------------------------------------------------------------
#define DIAG_FC 1024
COOP_KERNEL(k_diag_fin, DiagArgs) {
  (void)bx; (void)by; (void)bz;
  const DGrid &G = a.G;
  const double *row = a.row;
  const int ni = G.ni;
  double vol = 0.0, pes = 0.0, kes = 0.0, spd = -1.0E+20;
  double bC = 0.0, bCu = 0.0, bCv = 0.0, bCw = 0.0;
  int bi = 0, bj = 0, bk = 0;
  double *sV = lds, *sP = lds + DIAG_FC, *sK = lds + 2 * DIAG_FC;
  for (int c0 = G.T.Istr; c0 <= G.T.Iend; c0 += DIAG_FC) {
    const int nc = KMIN(DIAG_FC, G.T.Iend - c0 + 1);
    KSYNC();
    KLOOP1(d, 0, nc - 1) {
      const int ii = c0 + d - G.LBi;
      sV[d] = row[ii]; sP[d] = row[ii + ni]; sK[d] = row[ii + 2 * ni];
    }
    KSYNC();
    if (KTID == 0) {
#ifdef ROMS_CPU_EMU
      for (int d = 0; d < nc; d++) {
        vol = vol + sV[d];
        pes = pes + sP[d];
        kes = kes + sK[d];
      }
#else
      // 16 values at a time: the LDS reads are issued together, the additions stay in order
      for (int d0 = 0; d0 < nc; d0 += 16) {
        double rv[16], rp[16], rk[16];
#pragma unroll
        for (int q = 0; q < 16; q++) {
          const int d = d0 + q < nc ? d0 + q : d0;
          rv[q] = sV[d]; rp[q] = sP[d]; rk[q] = sK[d];
        }
#pragma unroll
        for (int q = 0; q < 16; q++)
          if (d0 + q < nc) { vol = vol + rv[q]; pes = pes + rp[q]; kes = kes + rk[q]; }
      }
#endif
    }
  }
  // maxima (order-free): every thread scans a strided subset, thread 0 merges the partial results
  KSYNC();
  double *mC = lds, *mS = lds + 256, *mI = lds + 512, *mJ = lds + 768, *mK = lds + 1024;
  {
    double pC = 0.0, pS = -1.0E+20;
    int pi = 0, pj = 0, pk = 0;
    KLOOP1(i, G.T.Istr, G.T.Iend) {
      const int ii = i - G.LBi;
      if (row[ii + 3 * ni] > pS) pS = row[ii + 3 * ni];
      const double C = row[ii + 4 * ni];
      const int j = (int)row[ii + 8 * ni], k = (int)row[ii + 9 * ni];
      if (C > 0.0 && (pC == 0.0 || diag_better(C, j, k, i, pC, pj, pk, pi))) { pC = C; pi = i; pj = j; pk = k; }
    }
    if (KTID < 256) { mC[KTID] = pC; mS[KTID] = pS; mI[KTID] = (double)pi; mJ[KTID] = (double)pj; mK[KTID] = (double)pk; }
  }
  KSYNC();
#ifndef ROMS_CPU_EMU
  // tree merge of the 256 partial results (the ordering relation is total: any merge order)
  for (int stride = 128; stride >= 1; stride >>= 1) {
    if (KTID < stride) {
      const int o = KTID + stride;
      if (mS[o] > mS[KTID]) mS[KTID] = mS[o];
      const double C = mC[o];
      if (C > 0.0 && (mC[KTID] == 0.0 || diag_better(C, (int)mJ[o], (int)mK[o], (int)mI[o], mC[KTID], (int)mJ[KTID],
                                                     (int)mK[KTID], (int)mI[KTID]))) {
        mC[KTID] = C; mI[KTID] = mI[o]; mJ[KTID] = mJ[o]; mK[KTID] = mK[o];
      }
    }
    KSYNC();
  }
#endif
  if (KTID == 0) {
    if (mS[0] > spd) spd = mS[0];
    if (mC[0] > 0.0) { bC = mC[0]; bi = (int)mI[0]; bj = (int)mJ[0]; bk = (int)mK[0]; }
    if (bC > 0.0) {
      const int ii = bi - G.LBi;
      bCu = row[ii + 5 * ni]; bCv = row[ii + 6 * ni]; bCw = row[ii + 7 * ni];
    }
    double *out = a.out;
    const double avgke = kes / vol, avgpe = pes / vol;
    out[0] = avgke; out[1] = avgpe; out[2] = avgke + avgpe; out[3] = vol; out[4] = spd;
    out[5] = bCu; out[6] = bCv; out[7] = bCw; out[8] = (double)bi; out[9] = (double)bj; out[10] = (double)bk;
    out[11] = bC;
    out[12] = kes; out[13] = pes;   // un-normalised sums: combined over tiles by the caller
  }
}
COOP_GLOBAL(k_diag_fin, DiagArgs)
#define DIAG_FIN_LDS (3 * DIAG_FC)
