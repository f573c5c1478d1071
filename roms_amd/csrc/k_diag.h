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
// Eight waves: waves 0..2 each add one of the three column products (volume, pe, ke) over the rows in
// j order, one lane per i -- a row of 64 columns is one coalesced 512-byte read, the reads of the next
// 32 rows are in flight while the (ordered, dependent) additions of the current 32 run; waves 3..7
// scan interleaved subsets of the rows for the maxima (8 rows of reads at a time), merged through LDS.
#define DIAG_IW 64
#define DIAG_JC 32
#define DIAG_NSUB 5
COOP_KERNEL(k_diag_row, DiagArgs) {
  (void)by; (void)bz;
  const DGrid &G = a.G;
  const TB &T = G.T;
  const double *col = a.col;
  double *mC = lds, *mS = lds + DIAG_NSUB * DIAG_IW, *mJ = lds + 2 * DIAG_NSUB * DIAG_IW, *mK = lds + 3 * DIAG_NSUB * DIAG_IW;
  const int i0 = T.Istr + DIAG_IW * bx;
  const int nw = KMIN(DIAG_IW, T.Iend - i0 + 1);
  double *row = a.row;
  const int ni = G.ni;
  for (int t = KTID; t < (3 + DIAG_NSUB) * DIAG_IW; t += KNT) {
    const int w = t / DIAG_IW, di = t % DIAG_IW;
    if (di >= nw) continue;
    const int i = i0 + di, ii = i - G.LBi;
    if (w < 3) {
      const double *src = col + (w == 0 ? 8 : (w == 1 ? 1 : 0)) * G.nij;   // volume, pe, ke products
      double acc = 0.0, nx[DIAG_JC];
#pragma unroll
      for (int q = 0; q < DIAG_JC; q++) nx[q] = src[X2(i, KMIN(T.Jstr + q, T.Jend))];
      for (int j0 = T.Jstr; j0 <= T.Jend; j0 += DIAG_JC) {
        double cur[DIAG_JC];
#pragma unroll
        for (int q = 0; q < DIAG_JC; q++) cur[q] = nx[q];
        KSCHED_FENCE();
        if (j0 + DIAG_JC <= T.Jend) {
#pragma unroll
          for (int q = 0; q < DIAG_JC; q++) nx[q] = src[X2(i, KMIN(j0 + DIAG_JC + q, T.Jend))];
        }
        KSCHED_FENCE();
#pragma unroll
        for (int q = 0; q < DIAG_JC; q++)
          if (j0 + q <= T.Jend) acc = acc + cur[q];
      }
      row[ii + w * ni] = acc;
    } else {
      // maxima: wave w scans the rows j = Jstr + (w-3), step DIAG_NSUB; the partial results of a column
      // are merged below (the ordering relation diag_better is total, so the merge order is free)
      const int sub = w - 3;
      double bC = 0.0, spd = 0.0;
      int bj = 0, bk = 0;
      for (int j0 = T.Jstr + sub; j0 <= T.Jend; j0 += 8 * DIAG_NSUB) {
        double C[8], K[8], S[8];
#pragma unroll
        for (int q = 0; q < 8; q++) {
          const size_t x = X2(i, KMIN(j0 + q * DIAG_NSUB, T.Jend));
          C[q] = col[x + 2 * G.nij]; K[q] = col[x + 6 * G.nij]; S[q] = col[x + 7 * G.nij];
        }
#pragma unroll
        for (int q = 0; q < 8; q++) {
          const int j = j0 + q * DIAG_NSUB;
          if (j > T.Jend) break;
          const int k = (int)K[q];
          if (C[q] > 0.0 && (bC == 0.0 || diag_better(C[q], j, k, i, bC, bj, bk, i))) { bC = C[q]; bj = j; bk = k; }
          if (S[q] > spd) spd = S[q];
        }
      }
      mC[sub * DIAG_IW + di] = bC; mS[sub * DIAG_IW + di] = spd;
      mJ[sub * DIAG_IW + di] = (double)bj; mK[sub * DIAG_IW + di] = (double)bk;
    }
  }
  KSYNC();
  for (int di = KTID; di < nw; di += KNT) {
    const int i = i0 + di, ii = i - G.LBi;
    double bC = 0.0, spd = 0.0;
    int bj = 0, bk = 0;
    for (int sub = 0; sub < DIAG_NSUB; sub++) {
      const double C = mC[sub * DIAG_IW + di];
      const int j = (int)mJ[sub * DIAG_IW + di], k = (int)mK[sub * DIAG_IW + di];
      if (C > 0.0 && (bC == 0.0 || diag_better(C, j, k, i, bC, bj, bk, i))) { bC = C; bj = j; bk = k; }
      if (mS[sub * DIAG_IW + di] > spd) spd = mS[sub * DIAG_IW + di];
    }
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
COOP_GLOBAL_LB(k_diag_row, DiagArgs, 512)
#define DIAG_ROW_LDS (4 * DIAG_NSUB * DIAG_IW)


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
    // threads 0, 64 and 128 (three waves) each add one of the three sums in i order, 16 LDS reads at a time
    for (int t = KTID; t < 192; t += KNT) {
      if (t % 64 != 0) continue;
      const int w = t / 64;
      const double *sp = w == 0 ? sV : (w == 1 ? sP : sK);
      double acc = w == 0 ? vol : (w == 1 ? pes : kes);
      for (int d0 = 0; d0 < nc; d0 += 16) {
        double r[16];
#pragma unroll
        for (int q = 0; q < 16; q++) r[q] = sp[d0 + q < nc ? d0 + q : d0];
#pragma unroll
        for (int q = 0; q < 16; q++)
          if (d0 + q < nc) acc = acc + r[q];
      }
      if (w == 0) vol = acc; else if (w == 1) pes = acc; else kes = acc;
    }
  }
  // the three sums meet in thread 0 (through LDS, after the staging area is free)
  KSYNC();
  for (int t = KTID; t < 192; t += KNT)
    if (t % 64 == 0) lds[3 * DIAG_FC + t / 64] = t == 0 ? vol : (t == 64 ? pes : kes);
  KSYNC();
  vol = lds[3 * DIAG_FC]; pes = lds[3 * DIAG_FC + 1]; kes = lds[3 * DIAG_FC + 2];
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
  // merge of the 256 partial results with WAVEFRONT SHUFFLES (round 4; rounds 1-3: an LDS tree, eight barrier rounds): the
  // ordering relation is total and max() is exact, so any merge order gives the same bits.  Each of the four waves
  // folds its 64 candidates in six butterfly steps (__shfl_xor: lane l takes lane l^m's candidate and keeps the better
  // one -- both lanes of a pair end with the same winner), lane 0 of each wave publishes it, thread 0 folds the four.
  if (KTID < 256) {
    double wC = mC[KTID], wS = mS[KTID];
    int wi = (int)mI[KTID], wj = (int)mJ[KTID], wk = (int)mK[KTID];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
      const double oC = __shfl_xor(wC, m, 64), oS = __shfl_xor(wS, m, 64);
      const int oi = __shfl_xor(wi, m, 64), oj = __shfl_xor(wj, m, 64), ok = __shfl_xor(wk, m, 64);
      if (oS > wS) wS = oS;
      if (oC > 0.0 && (wC == 0.0 || diag_better(oC, oj, ok, oi, wC, wj, wk, wi))) { wC = oC; wi = oi; wj = oj; wk = ok; }
    }
    KSYNC();                           // (every thread < 256 has read its own entry; the block has 256 threads)
    if ((KTID & 63) == 0) { const int w = KTID >> 6; mC[w] = wC; mS[w] = wS; mI[w] = (double)wi; mJ[w] = (double)wj; mK[w] = (double)wk; }
    KSYNC();
    if (KTID == 0) {
      for (int o = 1; o < 4; o++) {
        if (mS[o] > mS[0]) mS[0] = mS[o];
        const double C = mC[o];
        if (C > 0.0 && (mC[0] == 0.0 || diag_better(C, (int)mJ[o], (int)mK[o], (int)mI[o], mC[0], (int)mJ[0], (int)mK[0], (int)mI[0]))) {
          mC[0] = C; mI[0] = mI[o]; mJ[0] = mJ[o]; mK[0] = mK[o];
        }
      }
    }
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
#define DIAG_FIN_LDS (3 * DIAG_FC + 8)
