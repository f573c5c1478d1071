// k_bench.h -- physics that only BENCHMARK-type applications switch on:
//
//   k_eos_nl        rho_eos_tile (NONLIN_EOS)   ROMS/Nonlinear/rho_eos.F:247-560, mod_eoscoef.F
//   k_t3dmix2_geo   t3dmix2_geo_tile            ROMS/Nonlinear/t3dmix2_geo.h:90-420
//   k_bulk_pt, k_bulk_str  bulk_flux_tile       ROMS/Nonlinear/bulk_flux.F:208-1595 (COARE 3.0)
//   k_set_data_bm   set_data_tile (analytic)    ana_cloud/tair/humid/srflux/winds/rain/pair .h
//   k_swdk          lmd_swfrac_tile             ROMS/Nonlinear/lmd_swfrac.F:6-140 (pre_step3d use)
//
// All point-wise / column-wise and HBM-bound except bulk_flux (2-D, transcendental-bound, tiny).
#pragma once
#include "roms_ctx.h"
#include "k_libm.h"
#include "k_diag3d.h"

// ------------------------------------------------------------------------------- nonlinear EOS
// one thread per column of (IstrT:IendT, JstrT:JendT); den1, bulk0/1/2 of level k+1 are carried
// in registers for the Brunt-Vaisala frequency, rhoA/rhoS are accumulated top-down.
// t3dmix4_geo.h / t3dmix4_iso.h, the first operator at closed western / eastern walls: the column outside is zero, the corner
// value the average of the two boundary values next to it -- both zero (LBC closed; open boundaries are refused with TS_DIF4)
#define T3D4_WE_WALLS(L_, ok_)                                                                                   \
  if (!G.ewp) {                                                                                                   \
    if (B.west && i == B.Istr) {                                                                                  \
      L_[(long)(ok_) - 1] = 0.0;                                                                                  \
      if (!G.nsp && B.south && j == B.Jstr) L_[(long)(ok_) - 1 - ni] = 0.5 * (0.0 + 0.0);                          \
      if (!G.nsp && B.north && j == B.Jend) L_[(long)(ok_) - 1 + ni] = 0.5 * (0.0 + 0.0);                          \
    }                                                                                                             \
    if (B.east && i == B.Iend) {                                                                                  \
      L_[(long)(ok_) + 1] = 0.0;                                                                                  \
      if (!G.nsp && B.south && j == B.Jstr) L_[(long)(ok_) + 1 - ni] = 0.5 * (0.0 + 0.0);                          \
      if (!G.nsp && B.north && j == B.Jend) L_[(long)(ok_) + 1 + ni] = 0.5 * (0.0 + 0.0);                          \
    }                                                                                                             \
  }

struct EosLevel { double den, den1, bulk, bulk0, bulk1, bulk2, DbulkDS, DbulkDT, Dden1DS, Dden1DT; };

KDEV EosLevel eos_level(double Tt_in, double Ts_in, double Tp) {
  const double A00 = +1.909256e+04, A01 = +2.098925e+02, A02 = -3.041638e+00, A03 = -1.852732e-03, A04 = -1.361629e-05,
               B00 = +1.044077e+02, B01 = -6.500517e+00, B02 = +1.553190e-01, B03 = +2.326469e-04, D00 = -5.587545e+00,
               D01 = +7.390729e-01, D02 = -1.909078e-02, E00 = +4.721788e-01, E01 = +1.028859e-02, E02 = -2.512549e-04,
               E03 = -5.939910e-07, F00 = -1.571896e-02, F01 = -2.598241e-04, F02 = +7.267926e-06, G00 = +2.042967e-03,
               G01 = +1.045941e-05, G02 = -5.782165e-10, G03 = +1.296821e-07, H00 = -2.595994e-07, H01 = -1.248266e-09,
               H02 = -3.508914e-09, Q00 = +9.99842594e+02, Q01 = +6.793952e-02, Q02 = -9.095290e-03, Q03 = +1.001685e-04,
               Q04 = -1.120083e-06, Q05 = +6.536332e-09, U00 = +8.24493e-01, U01 = -4.08990e-03, U02 = +7.64380e-05,
               U03 = -8.24670e-07, U04 = +5.38750e-09, V00 = -5.72466e-03, V01 = +1.02270e-04, V02 = -1.65460e-06,
               W00 = +4.8314e-04;
  EosLevel L;
  const double Tt = KMAX(-2.0, Tt_in), Ts = KMAX(0.0, Ts_in);
  const double sqrtTs = sqrt(Ts);
  const double Tpr10 = 0.1 * Tp;
  double C[10], dCdT[10];
  C[0] = Q00 + Tt * (Q01 + Tt * (Q02 + Tt * (Q03 + Tt * (Q04 + Tt * Q05))));
  C[1] = U00 + Tt * (U01 + Tt * (U02 + Tt * (U03 + Tt * U04)));
  C[2] = V00 + Tt * (V01 + Tt * V02);
  dCdT[0] = Q01 + Tt * (2.0 * Q02 + Tt * (3.0 * Q03 + Tt * (4.0 * Q04 + Tt * 5.0 * Q05)));
  dCdT[1] = U01 + Tt * (2.0 * U02 + Tt * (3.0 * U03 + Tt * 4.0 * U04));
  dCdT[2] = V01 + Tt * 2.0 * V02;
  L.den1 = C[0] + Ts * (C[1] + sqrtTs * C[2] + Ts * W00);
  L.Dden1DS = C[1] + 1.5 * C[2] * sqrtTs + 2.0 * W00 * Ts;
  L.Dden1DT = dCdT[0] + Ts * (dCdT[1] + sqrtTs * dCdT[2]);
  C[3] = A00 + Tt * (A01 + Tt * (A02 + Tt * (A03 + Tt * A04)));
  C[4] = B00 + Tt * (B01 + Tt * (B02 + Tt * B03));
  C[5] = D00 + Tt * (D01 + Tt * D02);
  C[6] = E00 + Tt * (E01 + Tt * (E02 + Tt * E03));
  C[7] = F00 + Tt * (F01 + Tt * F02);
  C[8] = G01 + Tt * (G02 + Tt * G03);
  C[9] = H00 + Tt * (H01 + Tt * H02);
  dCdT[3] = A01 + Tt * (2.0 * A02 + Tt * (3.0 * A03 + Tt * 4.0 * A04));
  dCdT[4] = B01 + Tt * (2.0 * B02 + Tt * 3.0 * B03);
  dCdT[5] = D01 + Tt * 2.0 * D02;
  dCdT[6] = E01 + Tt * (2.0 * E02 + Tt * 3.0 * E03);
  dCdT[7] = F01 + Tt * 2.0 * F02;
  dCdT[8] = G02 + Tt * 2.0 * G03;
  dCdT[9] = H01 + Tt * 2.0 * H02;
  L.bulk0 = C[3] + Ts * (C[4] + sqrtTs * C[5]);
  L.bulk1 = C[6] + Ts * (C[7] + sqrtTs * G00);
  L.bulk2 = C[8] + Ts * C[9];
  L.bulk = L.bulk0 - Tp * (L.bulk1 - Tp * L.bulk2);
  L.DbulkDS = C[4] + sqrtTs * 1.5 * C[5] - Tp * (C[7] + sqrtTs * 1.5 * G00 - Tp * C[9]);
  L.DbulkDT = dCdT[3] + Ts * (dCdT[4] + sqrtTs * dCdT[5]) - Tp * (dCdT[6] + Ts * dCdT[7] - Tp * (dCdT[8] + Ts * dCdT[9]));
  const double cff = 1.0 / (L.bulk + Tpr10);
  double den = L.den1 * L.bulk * cff;
  L.den = den - 1000.0;
  return L;
}

THREAD_KERNEL(k_eos_nl, KArgs) {
  (void)gz;
  const DGrid &G = a.G;
  const Fields &F = a.Fv;
  const int i = G.T.IstrT + gx, j = G.T.JstrT + gy, N = G.N, nrhs = G.nrhs;
  const double g = G.g;
  double rhoA = 0.0, rhoS = 0.0;
  const EmitPlan P = emit_plan(G, BC_NONE, i, j);
  EosLevel up = {};   // level k+1
  double zr_up = 0.0;   // z_r(k+1)
  // six levels are loaded at a time (the loads overlap), then the column recurrences run on registers
  for (int k0 = N; k0 >= 1; k0 -= 6) {
  double c_t1[6], c_t2[6], c_zr[6], c_hz[6], c_zw[6];
#pragma unroll
  for (int q = 0; q < 6; q++) {
    const int kk = KMAX(k0 - q, 1);
    c_t1[q] = F.t[XT(i, j, kk, nrhs, 1)]; c_t2[q] = F.t[XT(i, j, kk, nrhs, 2)];
    c_zr[q] = F.z_r[X3(i, j, kk)]; c_hz[q] = F.Hz[X3(i, j, kk)]; c_zw[q] = F.z_w[XW(i, j, kk)];
  }
#pragma unroll
  for (int q = 0; q < 6; q++) {
    const int k = k0 - q;
    if (k < 1) break;
    const double zr_k = c_zr[q];
    EosLevel L = eos_level(c_t1[q], c_t2[q], zr_k);
    if (G.masking) L.den = L.den * F.rmask[X2(i, j)];                                    // rho_eos.F:357
    emit_store(G, P, F.rho + (size_t)(k - 1) * G.nij, L.den);
    emit_store(G, P, F.pden + (size_t)(k - 1) * G.nij, G.masking ? (L.den1 - 1000.0) * F.rmask[X2(i, j)] : (L.den1 - 1000.0));   // :479
    const double Hzk = c_hz[q];
    const double cff1 = L.den * Hzk;
    if (k == N) {
      rhoS = 0.5 * cff1 * Hzk;
      rhoA = cff1;
      // thermal expansion / saline contraction at the surface :470-500
      const double Tpr10 = 0.1 * zr_k;
      const double cff = L.bulk + Tpr10;
      const double c1 = Tpr10 * L.den1;
      const double c2 = L.bulk * cff;
      const double wrk = (L.den + 1000.0) * cff * cff;
      const double Tcof = -(L.DbulkDT * c1 + L.Dden1DT * c2);
      const double Scof = (L.DbulkDS * c1 + L.Dden1DS * c2);
      const double o = 1.0 / wrk;
      emit_store(G, P, F.alpha, o * Tcof);
      emit_store(G, P, F.beta, o * Scof);
    } else {
      rhoS = rhoS + Hzk * (rhoA + 0.5 * cff1);
      rhoA = rhoA + cff1;
      // Brunt-Vaisala frequency at W-level k (between k and k+1)
      const double zw = c_zw[q];
      const double bulk_up = up.bulk0 - zw * (up.bulk1 - up.bulk2 * zw);
      const double bulk_dn = L.bulk0 - zw * (L.bulk1 - L.bulk2 * zw);
      const double c1 = 1.0 / (bulk_up + 0.1 * zw);
      const double c2 = 1.0 / (bulk_dn + 0.1 * zw);
      const double den_up = c1 * (up.den1 * bulk_up);
      const double den_dn = c2 * (L.den1 * bulk_dn);
      emit_store(G, P, F.bvf + (size_t)k * G.nij,
                 -g * (den_up - den_dn) / (0.5 * (den_up + den_dn) * (zr_up - zr_k)));
    }
    up = L;
    zr_up = zr_k;
  }
  }
  emit_store(G, P, F.bvf, 0.0);
  emit_store(G, P, F.bvf + (size_t)N * G.nij, 0.0);
  const double cff2 = 1.0 / G.rho0;
  const double cff1 = 1.0 / (F.z_w[XW(i, j, N)] - F.z_w[XW(i, j, 0)]);
  emit_store(G, P, F.rhoA, cff2 * cff1 * rhoA);
  emit_store(G, P, F.rhoS, 2.0 * cff1 * cff1 * cff2 * rhoS);
}
THREAD_GLOBAL(k_eos_nl, KArgs)

// The same routine for grids with too few columns to fill the chip (BENCHMARK1: 33 K columns = two waves per CU of a
// kernel that evaluates a 40-term polynomial per level): one thread per column and CHUNK of p0 levels (grid.z = chunk,
// counted from the surface), the level above the chunk evaluated once more for the chunk's uppermost Brunt-Vaisala
// value; the two vertical integrals rhoA, rhoS -- a recurrence over the whole column -- are k_eos_sum's.  Same bits.
THREAD_KERNEL(k_eos_nl_pt, KArgs) {
  const DGrid &G = a.G;
  const Fields &F = a.Fv;
  const int i = G.T.IstrT + gx, j = G.T.JstrT + gy, N = G.N, nrhs = G.nrhs, KC = a.p0;
  const int k1 = N - gz * KC, k0 = KMAX(k1 - KC + 1, 1);
  if (k1 < 1) return;
  const double g = G.g;
  const EmitPlan P = emit_plan(G, BC_NONE, i, j);
  EosLevel up = {};
  double zr_up = 0.0;
  if (k1 < N) {
    zr_up = F.z_r[X3(i, j, k1 + 1)];
    up = eos_level(F.t[XT(i, j, k1 + 1, nrhs, 1)], F.t[XT(i, j, k1 + 1, nrhs, 2)], zr_up);
  }
  for (int k = k1; k >= k0; k--) {
    const double zr_k = F.z_r[X3(i, j, k)];
    EosLevel L = eos_level(F.t[XT(i, j, k, nrhs, 1)], F.t[XT(i, j, k, nrhs, 2)], zr_k);
    if (G.masking) L.den = L.den * F.rmask[X2(i, j)];                                    // rho_eos.F:357
    emit_store(G, P, F.rho + (size_t)(k - 1) * G.nij, L.den);
    emit_store(G, P, F.pden + (size_t)(k - 1) * G.nij, G.masking ? (L.den1 - 1000.0) * F.rmask[X2(i, j)] : (L.den1 - 1000.0));   // :479
    if (k == N) {
      const double Tpr10 = 0.1 * zr_k;
      const double cff = L.bulk + Tpr10;
      const double c1 = Tpr10 * L.den1;
      const double c2 = L.bulk * cff;
      const double wrk = (L.den + 1000.0) * cff * cff;
      const double Tcof = -(L.DbulkDT * c1 + L.Dden1DT * c2);
      const double Scof = (L.DbulkDS * c1 + L.Dden1DS * c2);
      const double o = 1.0 / wrk;
      emit_store(G, P, F.alpha, o * Tcof);
      emit_store(G, P, F.beta, o * Scof);
      emit_store(G, P, F.bvf + (size_t)N * G.nij, 0.0);
    } else {
      const double zw = F.z_w[XW(i, j, k)];
      const double bulk_up = up.bulk0 - zw * (up.bulk1 - up.bulk2 * zw);
      const double bulk_dn = L.bulk0 - zw * (L.bulk1 - L.bulk2 * zw);
      const double c1 = 1.0 / (bulk_up + 0.1 * zw);
      const double c2 = 1.0 / (bulk_dn + 0.1 * zw);
      const double den_up = c1 * (up.den1 * bulk_up);
      const double den_dn = c2 * (L.den1 * bulk_dn);
      emit_store(G, P, F.bvf + (size_t)k * G.nij, -g * (den_up - den_dn) / (0.5 * (den_up + den_dn) * (zr_up - zr_k)));
    }
    up = L;
    zr_up = zr_k;
  }
  if (k0 == 1) emit_store(G, P, F.bvf, 0.0);
}
THREAD_GLOBAL(k_eos_nl_pt, KArgs)

// LMD_DDMIX: alfaobeta(i,j,k) = Tcof/Scof at every level (rho_eos.F:435-455; the surface level's pair is what alpha, beta are made
// of above) | the constant Tcoef/Scoef of the linear equation of state (:782-796) -- p1 = 1 | 0.  A launch of its own behind the
// density kernels (they stay as they are: not a BASELINE option); index space (IstrT:IendT, JstrT:JendT, N).
THREAD_KERNEL(k_eos_alfaobeta, KArgs) {
  const DGrid &G = a.G;
  const Fields &F = a.Fv;
  const int i = G.T.IstrT + gx, j = G.T.JstrT + gy, k = gz + 1, nrhs = G.nrhs;
  double v;
  if (a.p1) {
    const double zr_k = F.z_r[X3(i, j, k)];
    const EosLevel L = eos_level(F.t[XT(i, j, k, nrhs, 1)], F.t[XT(i, j, k, nrhs, 2)], zr_k);
    const double Tpr10 = 0.1 * zr_k;
    const double cff = L.bulk + Tpr10;
    const double c1 = Tpr10 * L.den1;
    const double c2 = L.bulk * cff;
    const double Tcof = -(L.DbulkDT * c1 + L.Dden1DT * c2);
    const double Scof = (L.DbulkDS * c1 + L.Dden1DS * c2);
    v = Tcof / Scof;
  } else {
    const double cff = G.Scoef == 0.0 ? 1.0 : 1.0 / G.Scoef;
    v = cff * G.Tcoef;
  }
  G.alfaobeta[XW(i, j, k)] = v;
}
THREAD_GLOBAL(k_eos_alfaobeta, KArgs)

// rhoA, rhoS (rho_eos.F:382-420) from the rho that k_eos_nl_pt stored: the recurrence from the surface down
THREAD_KERNEL(k_eos_sum, KArgs) {
  (void)gz;
  const DGrid &G = a.G;
  const Fields &F = a.Fv;
  const int i = G.T.IstrT + gx, j = G.T.JstrT + gy, N = G.N;
  const EmitPlan P = emit_plan(G, BC_NONE, i, j);
  double rhoA = 0.0, rhoS = 0.0;
  for (int k0 = N; k0 >= 1; k0 -= 6) {
    double c_r[6], c_hz[6];
#pragma unroll
    for (int q = 0; q < 6; q++) { const int kk = KMAX(k0 - q, 1); c_r[q] = F.rho[X3(i, j, kk)]; c_hz[q] = F.Hz[X3(i, j, kk)]; }
#pragma unroll
    for (int q = 0; q < 6; q++) {
      const int k = k0 - q;
      if (k < 1) break;
      const double Hzk = c_hz[q], cff1 = c_r[q] * Hzk;
      if (k == N) { rhoS = 0.5 * cff1 * Hzk; rhoA = cff1; }
      else { rhoS = rhoS + Hzk * (rhoA + 0.5 * cff1); rhoA = rhoA + cff1; }
    }
  }
  const double cff2 = 1.0 / G.rho0;
  const double cff1 = 1.0 / (F.z_w[XW(i, j, N)] - F.z_w[XW(i, j, 0)]);
  emit_store(G, P, F.rhoA, cff2 * cff1 * rhoA);
  emit_store(G, P, F.rhoS, 2.0 * cff1 * cff1 * cff2 * rhoS);
}
THREAD_GLOBAL(k_eos_sum, KArgs)

// --------------------------------------------------------------------------------- t3dmix2_geo
// Point-wise: a thread marches up a chunk of KCH levels of its column (grid.z = chunk + nchunk*(itrc-1),
// p0 = nchunk) with the reference's two-level rolling buffers (k1, k2) held in registers: the
// horizontal differences of z_r and t at the four faces of the cell for the rho levels k and k+1, the
// vertical difference dTdz at the five columns of the stencil for the W levels k-1 and k, and the
// vertical flux FS(k-1) carried from the level below.  Every expression is the reference's
// (t3dmix2_geo.h:196-400); the faces shared with neighbouring cells are recomputed, not exchanged.
struct GeoLev { double zc, zw, ze, zs, zn, tc, tw, te, ts, tn; };                 // z_r, t at (i,j), (i-1,j), (i+1,j), (i,j-1), (i,j+1)
struct GeoGrad { double zxi, zxp, txi, txp, zej, zep, tej, tep; };                // dZdx, dTdx at faces i, i+1; dZde, dTde at faces j, j+1
struct GeoTz { double c, w, e, s, n; };                                           // dTdz at the five columns
KDEV GeoLev geo_load(const double *z, const double *t, long ni) {
  GeoLev L;
  L.zc = z[0]; L.zw = z[-1]; L.ze = z[1]; L.zs = z[-ni]; L.zn = z[ni];
  L.tc = t[0]; L.tw = t[-1]; L.te = t[1]; L.ts = t[-ni]; L.tn = t[ni];
  return L;
}
KDEV GeoGrad geo_grad(const GeoLev &L, double cxi, double cxp, double cej, double cep) {
  GeoGrad D;
  D.zxi = cxi * (L.zc - L.zw); D.txi = cxi * (L.tc - L.tw);
  D.zxp = cxp * (L.ze - L.zc); D.txp = cxp * (L.te - L.tc);
  D.zej = cej * (L.zc - L.zs); D.tej = cej * (L.tc - L.ts);
  D.zep = cep * (L.zn - L.zc); D.tep = cep * (L.tn - L.tc);
  return D;
}
KDEV GeoTz geo_tz(const GeoLev &lo, const GeoLev &up, bool zero) {   // W level between rho levels lo and up
  GeoTz T;
  if (zero) { T.c = T.w = T.e = T.s = T.n = 0.0; return T; }
  { const double cff = 1.0 / (up.zc - lo.zc); T.c = cff * (up.tc - lo.tc); }
  { const double cff = 1.0 / (up.zw - lo.zw); T.w = cff * (up.tw - lo.tw); }
  { const double cff = 1.0 / (up.ze - lo.ze); T.e = cff * (up.te - lo.te); }
  { const double cff = 1.0 / (up.zs - lo.zs); T.s = cff * (up.ts - lo.ts); }
  { const double cff = 1.0 / (up.zn - lo.zn); T.n = cff * (up.tn - lo.tn); }
  return T;
}
// FS at the W level between rho levels k1 (D1) and k2 (D2) :333-372
KDEV double geo_fs(const GeoGrad &D1, const GeoGrad &D2, double tz, double cff) {
  double c1 = KMIN(D1.zxi, 0.0), c2 = KMIN(D2.zxp, 0.0), c3 = KMAX(D2.zxi, 0.0), c4 = KMAX(D1.zxp, 0.0);
  double FS = cff * (c1 * (c1 * tz - D1.txi) + c2 * (c2 * tz - D2.txp) + c3 * (c3 * tz - D2.txi) + c4 * (c4 * tz - D1.txp));
  c1 = KMIN(D1.zej, 0.0); c2 = KMIN(D2.zep, 0.0); c3 = KMAX(D2.zej, 0.0); c4 = KMAX(D1.zep, 0.0);
  FS = FS + cff * (c1 * (c1 * tz - D1.tej) + c2 * (c2 * tz - D2.tep) + c3 * (c3 * tz - D2.tej) + c4 * (c4 * tz - D1.tep));
  return FS;
}
// (a thread marches a.p1 levels: the two levels below its first one are read again by every chunk)
// a.p2: 0 = t3dmix2_geo.h, the sum added to t(nnew); 1 = ... stored in F.tmix (run ahead of pre_step3d: k_pre_new adds it);
// 2, 3 = the two rotated operators of t3dmix4_geo.h:262-470, :600-772 (TS_DIF4 + MIX_GEO_TS; sqrt(TNU4) in F.diff4):
// 2 = the first, without coefficient of time, on the range widened by one point (:228-245) into LapT = F.tmix, with the
// closed-wall rows (:521-556); 3 = the second, on LapT, subtracted from t(nnew)
THREAD_KERNEL(k_t3dmix2_geo, KArgs) {
  const DGrid &G = a.G;
  const Fields &F = a.Fv;
  const TB &B = G.T;
  const int mode = a.p2;
  const int nch = a.p0, gch = a.p1, itrc = gz / nch + 1, k0 = (gz - (itrc - 1) * nch) * gch + 1;
  const int N = G.N;
  // the first operator of the biharmonic form: Imin = Istr-1 (periodic or inside the domain), MAX(Istr-1,1) at a wall
  const int i0 = mode == 2 ? ((G.ewp || !B.west) ? B.Istr - 1 : KMAX(B.Istr - 1, 1)) : B.Istr;
  const int j0 = mode == 2 ? ((G.nsp || !B.south) ? B.Jstr - 1 : KMAX(B.Jstr - 1, 1)) : B.Jstr;
  const int i = i0 + gx, j = j0 + gy;
  if (mode == 2) {
    const int i1 = (G.ewp || !B.east) ? B.Iend + 1 : KMIN(B.Iend + 1, G.Lm), j1 = (G.nsp || !B.north) ? B.Jend + 1 : KMIN(B.Jend + 1, G.Mm);
    if (i > i1 || j > j1) return;
  }
  if (k0 > N) return;
  const int k1 = KMIN(k0 + gch - 1, N);
  const size_t nij = (size_t)G.nij;
  const long ni = G.ni, x = (long)X2(i, j);
  const double *pm = F.pm + x, *pn = F.pn + x;
  const double *d2 = (mode >= 2 ? F.diff4 : F.diff2) + (size_t)(itrc - 1) * nij + x;
  double cxi = 0.5 * (pm[0] + pm[-1]), cxp = 0.5 * (pm[1] + pm[0]);
  double cej = 0.5 * (pn[0] + pn[-ni]), cep = 0.5 * (pn[ni] + pn[0]);
  if (G.masking) {                                   // t3dmix2_geo.h:229,261
    cxi = cxi * F.umask[x]; cxp = cxp * F.umask[x + 1]; cej = cej * F.vmask[x]; cep = cep * F.vmask[x + ni];
  }
  if (G.wet_dry) {                                   // WET_DRY :232,264
    cxi = cxi * F.umask_wet[x]; cxp = cxp * F.umask_wet[x + 1]; cej = cej * F.vmask_wet[x]; cep = cep * F.vmask_wet[x + ni];
  }
  const double fxi = 0.25 * (d2[0] + d2[-1]) * F.on_u[x], fxp = 0.25 * (d2[1] + d2[0]) * F.on_u[x + 1];
  const double fej = 0.25 * (d2[0] + d2[-ni]) * F.om_v[x], fep = 0.25 * (d2[ni] + d2[0]) * F.om_v[x + ni];
  const double cS = 0.5 * d2[0];
  const double c = G.dt * pm[0] * pn[0];
  const double *z = F.z_r + x, *Hz = F.Hz + x;
  const double *t = (mode == 3 ? (const double *)F.tmix + (size_t)(itrc - 1) * nij * (size_t)N : F.t + XT(G.LBi, G.LBj, 1, G.nrhs, itrc)) + x;
  double *tnew = F.t + XT(G.LBi, G.LBj, 1, G.nnew, itrc) + x;
  // state below the first level: D(k0-1), Tz and FS at W level k0-1
  GeoLev Lk = geo_load(z + (size_t)(k0 - 1) * nij, t + (size_t)(k0 - 1) * nij, ni);
  GeoGrad Dk = geo_grad(Lk, cxi, cxp, cej, cep), Dm = Dk;
  GeoTz Tm = geo_tz(Lk, Lk, true);
  double FSm = 0.0;
  if (k0 > 1) {
    const GeoLev Lm = geo_load(z + (size_t)(k0 - 2) * nij, t + (size_t)(k0 - 2) * nij, ni);
    Dm = geo_grad(Lm, cxi, cxp, cej, cep);
    Tm = geo_tz(Lm, Lk, false);
    FSm = geo_fs(Dm, Dk, Tm.c, cS);
  }
  for (int k = k0; k <= k1; k++) {
    const size_t ok = (size_t)(k - 1) * nij;
    GeoLev Lp = Lk;
    GeoGrad Dp = Dk;
    GeoTz Tk = geo_tz(Lk, Lk, true);
    double FSk = 0.0;
    if (k < N) {
      Lp = geo_load(z + ok + nij, t + ok + nij, ni);
      Dp = geo_grad(Lp, cxi, cxp, cej, cep);
      Tk = geo_tz(Lk, Lp, false);
      FSk = geo_fs(Dk, Dp, Tk.c, cS);
    }
    // horizontal fluxes at rho level k :253-300 (Tm: W level k-1, Tk: W level k)
    const double hc = Hz[ok];
    const double FXi = fxi * (hc + Hz[ok - 1]) *
                       (Dk.txi - 0.5 * (KMIN(Dk.zxi, 0.0) * (Tm.w + Tk.c) + KMAX(Dk.zxi, 0.0) * (Tk.w + Tm.c)));
    const double FXp = fxp * (Hz[ok + 1] + hc) *
                       (Dk.txp - 0.5 * (KMIN(Dk.zxp, 0.0) * (Tm.c + Tk.e) + KMAX(Dk.zxp, 0.0) * (Tk.c + Tm.e)));
    const double FEj = fej * (hc + Hz[ok - ni]) *
                       (Dk.tej - 0.5 * (KMIN(Dk.zej, 0.0) * (Tm.s + Tk.c) + KMAX(Dk.zej, 0.0) * (Tk.s + Tm.c)));
    const double FEp = fep * (Hz[ok + ni] + hc) *
                       (Dk.tep - 0.5 * (KMIN(Dk.zep, 0.0) * (Tm.c + Tk.n) + KMAX(Dk.zep, 0.0) * (Tk.c + Tm.n)));
    if (mode == 2) {                                           // t3dmix4_geo.h:458-470
      const double cffh = pm[0] * pn[0];
      const double cff1h = 1.0 / hc;
      double *LapT = F.tmix + (size_t)(itrc - 1) * nij * (size_t)N + x;
      LapT[ok] = cff1h * (cffh * (FXp - FXi + FEp - FEj) + (FSk - FSm));
      // closed southern / northern wall :521-556 (LBC closed: the row outside is zero; gradient: the row inside)
      if (!G.nsp && B.south && j == B.Jstr) LapT[(long)ok - ni] = 0.0;
      if (!G.nsp && B.north && j == B.Jend) LapT[(long)ok + ni] = 0.0;
      // closed western / eastern wall :475-520 and the corner averages :558-600 (of two zeros) -- round 6, pinned in a closed basin
      T3D4_WE_WALLS(LapT, ok)
      Lk = Lp; Dm = Dk; Dk = Dp; Tm = Tk; FSm = FSk;
      continue;
    }
    const double cff1 = c * (FXp - FXi);
    const double cff2 = c * (FEp - FEj);
    const double cff3 = G.dt * (FSk - FSm);
    const double cff4 = cff1 + cff2 + cff3;
    if (mode == 3) tnew[ok] = tnew[ok] - cff4;                 // t3dmix4_geo.h:754-762
    else if (mode == 1) F.tmix[(size_t)(itrc - 1) * nij * (size_t)N + ok + x] = cff4;       // (run ahead of pre_step3d: k_pre_new adds it)
    else tnew[ok] = tnew[ok] + cff4;
    if (G.dia_ts) {                                            // DIAGNOSTICS_TS t3dmix2_geo.h:409-414 / t3dmix2_iso.h:428-433
      dia_wrk(G, F, DIA_XDIF, itrc)[ok + x] = cff1;
      dia_wrk(G, F, DIA_YDIF, itrc)[ok + x] = cff2;
      dia_wrk(G, F, DIA_SDIF, itrc)[ok + x] = cff3;
      dia_wrk(G, F, DIA_HDIF, itrc)[ok + x] = cff4;
    }
    Lk = Lp; Dm = Dk; Dk = Dp; Tm = Tk; FSm = FSk;
  }
}
THREAD_GLOBAL(k_t3dmix2_geo, KArgs)

// --------------------------------------------------------------------------------- t3dmix2_iso
// Harmonic tracer mixing along isopycnic surfaces (MIX_ISO_TS; t3dmix2_iso.h:200-440, its default slope treatment): the
// same marching kernel as k_t3dmix2_geo with the potential density in the place of the depth -- the gradients of pden at
// the four faces, dT/drho at the five columns (the stratification floored at eps = 0.5), the vertical flux scaled by the
// layer thickness over the density step -- and MAX / MIN exchanged in the slope selections (:348-400).
KDEV GeoTz iso_tr(const GeoLev &lo, const GeoLev &up, bool zero) {   // dTdr at the W level between rho levels lo and up (z*: pden)
  GeoTz T;
  if (zero) { T.c = T.w = T.e = T.s = T.n = 0.0; return T; }
  const double eps = 0.5;
  { const double cff = -1.0 / KMAX(lo.zc - up.zc, eps); T.c = cff * (up.tc - lo.tc); }
  { const double cff = -1.0 / KMAX(lo.zw - up.zw, eps); T.w = cff * (up.tw - lo.tw); }
  { const double cff = -1.0 / KMAX(lo.ze - up.ze, eps); T.e = cff * (up.te - lo.te); }
  { const double cff = -1.0 / KMAX(lo.zs - up.zs, eps); T.s = cff * (up.ts - lo.ts); }
  { const double cff = -1.0 / KMAX(lo.zn - up.zn, eps); T.n = cff * (up.tn - lo.tn); }
  return T;
}
// FS at the W level between rho levels k1 (D1) and k2 (D2) :386-415; fac = -(z_r(k2)-z_r(k1)) / max(pden(k1)-pden(k2), eps)
KDEV double iso_fs(const GeoGrad &D1, const GeoGrad &D2, double tr, double d2c, double fac) {
  double c1 = KMAX(D1.zxi, 0.0), c2 = KMAX(D2.zxp, 0.0), c3 = KMIN(D2.zxi, 0.0), c4 = KMIN(D1.zxp, 0.0);
  double cff = c1 * (c1 * tr - D1.txi) + c2 * (c2 * tr - D2.txp) + c3 * (c3 * tr - D2.txi) + c4 * (c4 * tr - D1.txp);
  c1 = KMAX(D1.zej, 0.0); c2 = KMAX(D2.zep, 0.0); c3 = KMIN(D2.zej, 0.0); c4 = KMIN(D1.zep, 0.0);
  cff = cff + c1 * (c1 * tr - D1.tej) + c2 * (c2 * tr - D2.tep) + c3 * (c3 * tr - D2.tej) + c4 * (c4 * tr - D1.tep);
  return 0.5 * cff * d2c * fac;
}
// ... of t3dmix4_iso.h:459-479, :764-784 (the coefficient inside: difx = dife = 0.5 sqrt(TNU4))
KDEV double iso_fs4(const GeoGrad &D1, const GeoGrad &D2, double tr, double d4c, double fac) {
  const double difx = 0.5 * d4c, dife = difx;
  double c1 = KMAX(D1.zxi, 0.0), c2 = KMAX(D2.zxp, 0.0), c3 = KMIN(D2.zxi, 0.0), c4 = KMIN(D1.zxp, 0.0);
  double cff = difx * (c1 * (c1 * tr - D1.txi) + c2 * (c2 * tr - D2.txp) + c3 * (c3 * tr - D2.txi) + c4 * (c4 * tr - D1.txp));
  c1 = KMAX(D1.zej, 0.0); c2 = KMAX(D2.zep, 0.0); c3 = KMIN(D2.zej, 0.0); c4 = KMIN(D1.zep, 0.0);
  cff = cff + dife * (c1 * (c1 * tr - D1.tej) + c2 * (c2 * tr - D2.tep) + c3 * (c3 * tr - D2.tej) + c4 * (c4 * tr - D1.tep));
  return cff * fac;
}
// a.p2: 0 = t3dmix2_iso.h; 2, 3 = the two rotated operators of t3dmix4_iso.h:270-499, :623-808 (TS_DIF4 + MIX_ISO_TS), as in
// k_t3dmix2_geo: the first into LapT = F.tmix on the widened range, the second on LapT, subtracted from t(nnew)
THREAD_KERNEL(k_t3dmix2_iso, KArgs) {
  const DGrid &G = a.G;
  const Fields &F = a.Fv;
  const TB &B = G.T;
  const int mode = a.p2;
  const int nch = a.p0, gch = a.p1, itrc = gz / nch + 1, k0 = (gz - (itrc - 1) * nch) * gch + 1;
  const int N = G.N;
  const int i0 = mode == 2 ? ((G.ewp || !B.west) ? B.Istr - 1 : KMAX(B.Istr - 1, 1)) : B.Istr;      // t3dmix4_iso.h:241-254
  const int j0 = mode == 2 ? ((G.nsp || !B.south) ? B.Jstr - 1 : KMAX(B.Jstr - 1, 1)) : B.Jstr;
  const int i = i0 + gx, j = j0 + gy;
  if (mode == 2) {
    const int i1 = (G.ewp || !B.east) ? B.Iend + 1 : KMIN(B.Iend + 1, G.Lm), j1 = (G.nsp || !B.north) ? B.Jend + 1 : KMIN(B.Jend + 1, G.Mm);
    if (i > i1 || j > j1) return;
  }
  if (k0 > N) return;
  const int k1 = KMIN(k0 + gch - 1, N);
  const size_t nij = (size_t)G.nij;
  const long ni = G.ni, x = (long)X2(i, j);
  const double *pm = F.pm + x, *pn = F.pn + x;
  const double *d2 = (mode >= 2 ? F.diff4 : F.diff2) + (size_t)(itrc - 1) * nij + x;
  double cxi = 0.5 * (pm[0] + pm[-1]), cxp = 0.5 * (pm[1] + pm[0]);
  double cej = 0.5 * (pn[0] + pn[-ni]), cep = 0.5 * (pn[ni] + pn[0]);
  if (G.masking) {
    cxi = cxi * F.umask[x]; cxp = cxp * F.umask[x + 1]; cej = cej * F.vmask[x]; cep = cep * F.vmask[x + ni];
  }
  if (G.wet_dry) {                                           // t3dmix2_iso.h:234-236, :266-268 (round 6)
    cxi = cxi * F.umask_wet[x]; cxp = cxp * F.umask_wet[x + 1]; cej = cej * F.vmask_wet[x]; cep = cep * F.vmask_wet[x + ni];
  }
  const double fxi = 0.25 * (d2[0] + d2[-1]) * F.on_u[x], fxp = 0.25 * (d2[1] + d2[0]) * F.on_u[x + 1];
  const double fej = 0.25 * (d2[0] + d2[-ni]) * F.om_v[x], fep = 0.25 * (d2[ni] + d2[0]) * F.om_v[x + ni];
  const double c = G.dt * pm[0] * pn[0], eps = 0.5;
  const double *r = F.pden + x, *zr = F.z_r + x, *Hz = F.Hz + x;
  const double *t = (mode == 3 ? (const double *)F.tmix + (size_t)(itrc - 1) * nij * (size_t)N : F.t + XT(G.LBi, G.LBj, 1, G.nrhs, itrc)) + x;
  double *tnew = F.t + XT(G.LBi, G.LBj, 1, G.nnew, itrc) + x;
#define ISO_FAC(klo) ((-1.0 / KMAX(r[(size_t)((klo) - 1) * nij] - r[(size_t)(klo) * nij], eps)) * (zr[(size_t)(klo) * nij] - zr[(size_t)((klo) - 1) * nij]))
  GeoLev Lk = geo_load(r + (size_t)(k0 - 1) * nij, t + (size_t)(k0 - 1) * nij, ni);
  GeoGrad Dk = geo_grad(Lk, cxi, cxp, cej, cep), Dm = Dk;
  GeoTz Tm = iso_tr(Lk, Lk, true);
  double FSm = 0.0;
  if (k0 > 1) {
    const GeoLev Lm = geo_load(r + (size_t)(k0 - 2) * nij, t + (size_t)(k0 - 2) * nij, ni);
    Dm = geo_grad(Lm, cxi, cxp, cej, cep);
    Tm = iso_tr(Lm, Lk, false);
    FSm = mode >= 2 ? iso_fs4(Dm, Dk, Tm.c, d2[0], ISO_FAC(k0 - 1)) : iso_fs(Dm, Dk, Tm.c, d2[0], ISO_FAC(k0 - 1));
  }
  for (int k = k0; k <= k1; k++) {
    const size_t ok = (size_t)(k - 1) * nij;
    GeoLev Lp = Lk;
    GeoGrad Dp = Dk;
    GeoTz Tk = iso_tr(Lk, Lk, true);
    double FSk = 0.0;
    if (k < N) {
      Lp = geo_load(r + ok + nij, t + ok + nij, ni);
      Dp = geo_grad(Lp, cxi, cxp, cej, cep);
      Tk = iso_tr(Lk, Lp, false);
      FSk = mode >= 2 ? iso_fs4(Dk, Dp, Tk.c, d2[0], ISO_FAC(k)) : iso_fs(Dk, Dp, Tk.c, d2[0], ISO_FAC(k));
    }
    const double hc = Hz[ok];
    const double FXi = fxi * (hc + Hz[ok - 1]) *
                       (Dk.txi - 0.5 * (KMAX(Dk.zxi, 0.0) * (Tm.w + Tk.c) + KMIN(Dk.zxi, 0.0) * (Tk.w + Tm.c)));
    const double FXp = fxp * (Hz[ok + 1] + hc) *
                       (Dk.txp - 0.5 * (KMAX(Dk.zxp, 0.0) * (Tm.c + Tk.e) + KMIN(Dk.zxp, 0.0) * (Tk.c + Tm.e)));
    const double FEj = fej * (hc + Hz[ok - ni]) *
                       (Dk.tej - 0.5 * (KMAX(Dk.zej, 0.0) * (Tm.s + Tk.c) + KMIN(Dk.zej, 0.0) * (Tk.s + Tm.c)));
    const double FEp = fep * (Hz[ok + ni] + hc) *
                       (Dk.tep - 0.5 * (KMAX(Dk.zep, 0.0) * (Tm.c + Tk.n) + KMIN(Dk.zep, 0.0) * (Tk.c + Tm.n)));
    if (mode == 2) {                                           // t3dmix4_iso.h:488-497
      const double cffh = pm[0] * pn[0];
      const double cff1h = 1.0 / hc;
      double *LapT = F.tmix + (size_t)(itrc - 1) * nij * (size_t)N + x;
      LapT[ok] = cff1h * (cffh * (FXp - FXi + FEp - FEj) + (FSk - FSm));
      // closed southern / northern wall :540-574 (LBC closed: the row outside is zero)
      if (!G.nsp && B.south && j == B.Jstr) LapT[(long)ok - ni] = 0.0;
      if (!G.nsp && B.north && j == B.Jend) LapT[(long)ok + ni] = 0.0;
      // closed western / eastern wall :504-539 and the corner averages :576-618 (of two zeros) -- round 6
      T3D4_WE_WALLS(LapT, ok)
      Lk = Lp; Dm = Dk; Dk = Dp; Tm = Tk; FSm = FSk;
      continue;
    }
    const double cff1 = c * (FXp - FXi);
    const double cff2 = c * (FEp - FEj);
    const double cff3 = G.dt * (FSk - FSm);
    const double cff4 = cff1 + cff2 + cff3;
    if (mode == 3) tnew[ok] = tnew[ok] - cff4;                 // t3dmix4_iso.h:791-798
    else tnew[ok] = tnew[ok] + cff4;
    if (G.dia_ts) {                                            // DIAGNOSTICS_TS t3dmix2_geo.h:409-414 / t3dmix2_iso.h:428-433
      dia_wrk(G, F, DIA_XDIF, itrc)[ok + x] = cff1;
      dia_wrk(G, F, DIA_YDIF, itrc)[ok + x] = cff2;
      dia_wrk(G, F, DIA_SDIF, itrc)[ok + x] = cff3;
      dia_wrk(G, F, DIA_HDIF, itrc)[ok + x] = cff4;
    }
    Lk = Lp; Dm = Dk; Dk = Dp; Tm = Tk; FSm = FSk;
  }
#undef ISO_FAC
}
THREAD_GLOBAL(k_t3dmix2_iso, KArgs)

// ------------------------------------------------------------------------------------ bulk_flux
KDEV double blk_psiu(double ZoL) {
  const double pi = 3.14159265358979323846, r3 = 1.0 / 3.0;
  if (ZoL < 0.0) {
    const double x = kpow(1.0 - 15.0 * ZoL, 0.25);
    const double psik = 2.0 * klog(0.5 * (1.0 + x)) + klog(0.5 * (1.0 + x * x)) - 2.0 * katan(x) + 0.5 * pi;
    double cff = sqrt(3.0);
    const double y = kpow(1.0 - 10.15 * ZoL, r3);
    const double psic = 1.5 * klog(r3 * (1.0 + y + y * y)) - cff * katan((1.0 + 2.0 * y) / cff) + pi / cff;
    cff = ZoL * ZoL;
    const double Fw = cff / (1.0 + cff);
    return (1.0 - Fw) * psik + Fw * psic;
  }
  const double cff = KMIN(50.0, 0.35 * ZoL);
  return -((1.0 + ZoL) + 0.6667 * (ZoL - 14.28) / kexp(cff) + 8.525);
}
KDEV double blk_psit(double ZoL) {
  const double pi = 3.14159265358979323846, r3 = 1.0 / 3.0;
  if (ZoL < 0.0) {
    const double x = kpow(1.0 - 15.0 * ZoL, 0.5);
    const double psik = 2.0 * klog(0.5 * (1.0 + x));
    double cff = sqrt(3.0);
    const double y = kpow(1.0 - 34.15 * ZoL, r3);
    const double psic = 1.5 * klog(r3 * (1.0 + y + y * y)) - cff * katan((1.0 + 2.0 * y) / cff) + pi / cff;
    cff = ZoL * ZoL;
    const double Fw = cff / (1.0 + cff);
    return (1.0 - Fw) * psik + Fw * psic;
  }
  const double cff = KMIN(50.0, 0.35 * ZoL);
  return -(kpow(1.0 + 2.0 * ZoL, 1.5) + 0.6667 * (ZoL - 14.28) / kexp(cff) + 8.525);
}

struct BulkArgs {
  Fields Fv;         // the array pointers, by value (a table in device memory would cost every kernel one more dependent round trip)
  DGrid G;
  double ZW, ZT, ZQ;
};

// point-wise over (Istr-1:IendR, Jstr-1:JendR): Taux -> wrk2[0], Tauy -> wrk2[1]; heat fluxes stored
// on (IstrR:IendR, JstrR:JendR)
THREAD_KERNEL(k_bulk_pt, BulkArgs) {
  (void)gz;
  const DGrid &G = a.G;
  const Fields &F = a.Fv;
  const TB &B = G.T;
  const int i = B.Istr - 1 + gx, j = B.Jstr - 1 + gy, N = G.N, nrhs = G.nrhs;
  const double StefBo = 5.67E-8, emmiss = 0.97, blk_Cpa = 1004.67, blk_Cpw = 4000.0, blk_Rgas = 287.1, blk_Zabl = 600.0,
               blk_beta = 1.2, vonKar = 0.41;
  const double eps = 1.0E-20, r3 = 1.0 / 3.0, g = G.g;
  const double ZW = a.ZW, ZT = a.ZT, ZQ = a.ZQ;
  const double Uair = F.Uwind[X2(i, j)], Vair = F.Vwind[X2(i, j)];
  const double Wmag = sqrt(Uair * Uair + Vair * Vair);
  const double PairM = F.Pair[X2(i, j)];
  const double TairC = F.Tair[X2(i, j)];
  const double TairK = TairC + 273.16;
  const double TseaC = F.t[XT(i, j, N, nrhs, 1)];
  const double TseaK = TseaC + 273.16;
  const double RH = F.Hair[X2(i, j)];
  double delTc = 0.0, delQc = 0.0;
  double cff, cff1, cff2;
  cff = (0.7859 + 0.03477 * TairC) / (1.0 + 0.00412 * TairC);
  const double e_sat = kpow(10.0, cff);
  const double vap_p = e_sat * RH;
  cff2 = TairK * TairK * TairK;
  cff1 = cff2 * TairK;
  const double cl = F.cloud[X2(i, j)];
  const double LRad = -emmiss * StefBo * (cff1 * (0.39 - 0.05 * sqrt(vap_p)) * (1.0 - 0.6823 * cl * cl) + cff2 * 4.0 * (TseaK - TairK));
  cff = (1.0007 + 3.46E-6 * PairM) * 6.1121 * kexp(17.502 * TairC / (240.97 + TairC));
  const double Qair = 0.62197 * (cff / (PairM - 0.378 * cff + eps));
  double Q;
  if (RH < 2.0) { cff = cff * RH; Q = 0.62197 * (cff / (PairM - 0.378 * cff + eps)); }
  else Q = RH / 1000.0;
  cff = (1.0007 + 3.46E-6 * PairM) * 6.1121 * kexp(17.502 * TseaC / (240.97 + TseaC));
  cff = cff * 0.98;
  const double Qsea = 0.62197 * (cff / (PairM - 0.378 * cff));
  const double rhoAir = PairM * 100.0 / (blk_Rgas * TairK * (1.0 + 0.61 * Q));
  const double VisAir = 1.326E-5 * (1.0 + TairC * (6.542E-3 + TairC * (8.301E-6 - 4.84E-9 * TairC)));
  const double Hlv = (2.501 - 0.00237 * TseaC) * 1.0E+6;
  double Wgus = 0.5;
  double delW = sqrt(Wmag * Wmag + Wgus * Wgus);
  const double delQ = Qsea - Q;
  const double delT = TseaC - TairC;
  double ZoW = 0.0001;
  const double u10 = delW * klog(10.0 / ZoW) / klog(ZW / ZoW);
  double Wstar = 0.035 * u10;
  const double Zo10 = 0.011 * Wstar * Wstar / g + 0.11 * VisAir / Wstar;
  double tmp = vonKar / klog(10.0 / Zo10);
  const double Cd10 = tmp * tmp;
  const double Ch10 = 0.00115;
  const double Ct10 = Ch10 / sqrt(Cd10);
  const double ZoT10 = 10.0 / kexp(vonKar / Ct10);
  tmp = vonKar / klog(ZW / Zo10);
  const double Cd = tmp * tmp;
  const double Ct = vonKar / klog(ZT / ZoT10);
  const double CC = vonKar * Ct / Cd;
  delTc = 0.0;
  const double Ribcu = -ZW / (blk_Zabl * 0.004 * (blk_beta * blk_beta * blk_beta));
  const double Ri = -g * ZW * ((delT - delTc) + 0.61 * TairK * delQ) / (TairK * delW * delW + eps);
  double Zetu;
  if (Ri < 0.0) Zetu = CC * Ri / (1.0 + Ri / Ribcu);
  else Zetu = CC * Ri / (1.0 + 3.0 * Ri / CC);
  const double L10 = ZW / Zetu;
  Wstar = delW * vonKar / (klog(ZW / Zo10) - blk_psiu(ZW / L10));
  double Tstar = -(delT - delTc) * vonKar / (klog(ZT / ZoT10) - blk_psit(ZT / L10));
  double Qstar = -(delQ - delQc) * vonKar / (klog(ZQ / ZoT10) - blk_psit(ZQ / L10));
  const double charn = KMIN(0.028, -0.005 + 0.0017 * delW);
  for (int Iter = 1; Iter <= 3; Iter++) {
    ZoW = charn * Wstar * Wstar / g + 0.11 * VisAir / (Wstar + eps);
    const double Rr = ZoW * Wstar / VisAir;
    const double ZoQ = KMIN(1.6e-4, 5.8e-5 / kpow(Rr, 0.72));
    const double ZoT = ZoQ;
    const double ZoL = vonKar * g * ZW * (Tstar * (1.0 + 0.61 * Q) + 0.61 * TairK * Qstar) /
                       (TairK * Wstar * Wstar * (1.0 + 0.61 * Q) + eps);
    const double L = ZW / (ZoL + eps);
    const double Wpsi = blk_psiu(ZoL);
    const double Tpsi = blk_psit(ZT / L);
    const double Qpsi = blk_psit(ZQ / L);
    Wstar = KMAX(eps, delW * vonKar / (klog(ZW / ZoW) - Wpsi));
    Tstar = -(delT - delTc) * vonKar / (klog(ZT / ZoT) - Tpsi);
    Qstar = -(delQ - delQc) * vonKar / (klog(ZQ / ZoQ) - Qpsi);
    const double Bf = -g / TairK * Wstar * (Tstar + 0.61 * TairK * Qstar);
    if (Bf > 0.0) Wgus = blk_beta * kpow(Bf * blk_Zabl, r3);
    else Wgus = 0.2;
    delW = sqrt(Wmag * Wmag + Wgus * Wgus);
  }
  const double Hs = -blk_Cpa * rhoAir * Wstar * Tstar;
  const double diffw = 2.11E-5 * kpow(TairK / 273.16, 1.94);
  const double diffh = 0.02411 * (1.0 + TairC * (3.309E-3 - 1.44E-6 * TairC)) / (rhoAir * blk_Cpa + eps);
  cff = Qair * Hlv / (blk_Rgas * TairK * TairK);
  const double wet_bulb = 1.0 / (1.0 + 0.622 * (cff * Hlv * diffw) / (blk_Cpa * diffh));
  const double rn = fabs(F.rain[X2(i, j)]);
  const double Hsr = rn * wet_bulb * blk_Cpw * ((TseaC - TairC) + (Qsea - Q) * Hlv / blk_Cpa);
  double SHeat = (Hs + Hsr);
  const double Hl = -Hlv * rhoAir * Wstar * Qstar;
  const double upvel = -1.61 * Wstar * Qstar - (1.0 + 1.61 * Q) * Wstar * Tstar / TairK;
  const double Hlw = rhoAir * Hlv * upvel * Q;
  double LHeat = (Hl + Hlw);
  const double Taur = 0.85 * rn * Wmag;
  cff = rhoAir * (Wstar * Wstar + Taur / rhoAir) / (Wmag + eps);
  double Taux = cff * Uair, Tauy = cff * Vair, LRadm = LRad;
  if (G.masking) {                                   // bulk_flux.F:635,977,1006,1030,1037
    const double rm = F.rmask[X2(i, j)];
    LRadm = LRadm * rm; SHeat = SHeat * rm; LHeat = LHeat * rm; Taux = Taux * rm; Tauy = Tauy * rm;
  }
  if (G.wet_dry) {                                   // WET_DRY bulk_flux.F:638,980,1009,1033,1040
    const double rw = F.rmask_wet[X2(i, j)];
    LRadm = LRadm * rw; SHeat = SHeat * rw; LHeat = LHeat * rw; Taux = Taux * rw; Tauy = Tauy * rw;
  }
  F.wrk2[0][X2(i, j)] = Taux;
  F.wrk2[1][X2(i, j)] = Tauy;
  if (i >= B.IstrR && j >= B.JstrR) {
    const double Hscale = 1.0 / (G.rho0 * G.Cp);
    const double lr = LRadm * Hscale, lh = -LHeat * Hscale, sh = -SHeat * Hscale;
    const EmitPlan P = emit_plan(G, BC_NONE, i, j);
    emit_store(G, P, F.lrflx, lr);
    emit_store(G, P, F.lhflx, lh);
    emit_store(G, P, F.shflx, sh);
    const double st = (F.srflx[X2(i, j)] + lr + lh + sh);
    double stm = G.masking ? st * F.rmask[X2(i, j)] : st;                                // :1259
    if (G.wet_dry) stm = stm * F.rmask_wet[X2(i, j)];                                    // WET_DRY :1262
    emit_store(G, P, F.stflux, stm);
  }
}
THREAD_GLOBAL(k_bulk_pt, BulkArgs)

// surface stresses at u,v points; index space (min(Istr,IstrR):IendR, min(Jstr,JstrR):JendR)
THREAD_KERNEL(k_bulk_str, BulkArgs) {
  (void)gz;
  const DGrid &G = a.G;
  const Fields &F = a.Fv;
  const TB &B = G.T;
  const int i = KMIN(B.Istr, B.IstrR) + gx, j = KMIN(B.Jstr, B.JstrR) + gy;
  const double cff = 0.5 / G.rho0;
  const EmitPlan P = emit_plan(G, BC_NONE, i, j);
  if (i >= B.Istr && j >= B.JstrR) {
    const double s = cff * (F.wrk2[0][X2(i - 1, j)] + F.wrk2[0][X2(i, j)]);
    double sm = G.masking ? s * F.umask[X2(i, j)] : s;                                   // :1295
    if (G.wet_dry) sm = sm * F.umask_wet[X2(i, j)];                                      // WET_DRY :1298
    emit_store(G, P, F.sustr, sm);
  }
  if (i >= B.IstrR && j >= B.Jstr) {
    const double s = cff * (F.wrk2[1][X2(i, j - 1)] + F.wrk2[1][X2(i, j)]);
    double sm = G.masking ? s * F.vmask[X2(i, j)] : s;                                   // :1310
    if (G.wet_dry) sm = sm * F.vmask_wet[X2(i, j)];                                      // WET_DRY :1313
    emit_store(G, P, F.svstr, sm);
  }
}
THREAD_GLOBAL(k_bulk_str, BulkArgs)

// -------------------------------------------------------------------------- set_data (BENCHMARK)
struct SetDataBmArgs {
  DGrid G;
  Fields Fv;         // the array pointers, by value (a table in device memory would cost every kernel one more dependent round trip)
  double Dangle, Hangle;   // solar declination and hour angle of this step (host: caldate)
};
// index space (IstrT:IendT, JstrT:JendT)
THREAD_KERNEL(k_set_data_bm, SetDataBmArgs) {
  (void)gz;
  const DGrid &G = a.G;
  const Fields &F = a.Fv;
  const int i = G.T.IstrT + gx, j = G.T.JstrT + gy;
  const double deg2rad = 3.14159265358979323846 / 180.0, Csolar = 1353.0, alb_w = 0.06;
  const double cl = 0.6, Ta = 4.0, Ha = 0.8;
  const EmitPlan P = emit_plan(G, BC_NONE, i, j);
  emit_store(G, P, F.cloud, cl);
  emit_store(G, P, F.Tair, Ta);
  emit_store(G, P, F.Hair, Ha);
  const double Rsolar = Csolar / (G.rho0 * G.Cp);
  const double LatRad = F.latr[X2(i, j)] * deg2rad;
  const double cff1 = ksin(LatRad) * ksin(a.Dangle);
  const double cff2 = kcos(LatRad) * kcos(a.Dangle);
  double sr = 0.0;
  const double zenith = cff1 + cff2 * kcos(a.Hangle - F.lonr[X2(i, j)] * deg2rad);
  if (zenith > 0.0) {
    const double cff = (0.7859 + 0.03477 * Ta) / (1.0 + 0.00412 * Ta);
    const double e_sat = kpow(10.0, cff);
    const double vap_p = e_sat * Ha;
    sr = Rsolar * zenith * zenith * (1.0 - 0.6 * (cl * cl * cl)) / ((zenith + 2.7) * vap_p * 1.0E-3 + 1.085 * zenith + 0.1);
  }
  emit_store(G, P, F.srflx, (1.0 - alb_w) * sr);
  const double cff = 0.2 * (60.0 + F.latr[X2(i, j)]);
  emit_store(G, P, F.Uwind, 15.0 * kexp(-cff * cff));
  emit_store(G, P, F.Vwind, 0.0);
  emit_store(G, P, F.rain, 0.0);
  F.btflux[X2T(i, j, 1)] = 0.0;
  emit_store(G, P, F.stflux + G.nij, 0.0);
  F.btflux[X2T(i, j, 2)] = 0.0;
  emit_store(G, P, F.Pair, 1025.0);
}
THREAD_GLOBAL(k_set_data_bm, SetDataBmArgs)

// ---------------------------------------------------------------- solar penetration (pre_step3d)
struct SwArgs {
  DGrid G;
  Fields Fv;         // the array pointers, by value (a table in device memory would cost every kernel one more dependent round trip)
  double fac1, fac2, fac3;   // Zscale/lmd_mu1(Jwt), Zscale/lmd_mu2(Jwt), lmd_r1(Jwt) with Zscale = -1
};
// swdk(i,j,k) into wrk3[5] for k = 1..N-1; index space (Istr:Iend, Jstr:Jend, N-1)
THREAD_KERNEL(k_swdk, SwArgs) {
  const DGrid &G = a.G;
  const Fields &F = a.Fv;
  const int i = G.T.Istr + gx, j = G.T.Jstr + gy, k = gz + 1, N = G.N;
  const double Z = F.z_w[XW(i, j, N)] - F.z_w[XW(i, j, k)];
  F.wrk3[5][XW(i, j, k)] = kexp(Z * a.fac1) * a.fac3 + kexp(Z * a.fac2) * (1.0 - a.fac3);
}
THREAD_GLOBAL(k_swdk, SwArgs)
