// k_lmd.h -- K-profile vertical mixing (Large, McWilliams and Doney 1994).
//
//   k_lmd_interior  lmd_vmix_tile    ROMS/Nonlinear/lmd_vmix.F:99-460  (LMD_RIMIX, RI_SPLINES)
//   k_lmd_skpp      lmd_skpp_tile    ROMS/Nonlinear/lmd_skpp.F:98-930  (LMD_SKPP, LMD_NONLOCAL)
//                   lmd_swfrac_tile  ROMS/Nonlinear/lmd_swfrac.F:6-140
//   (lmd_finish_tile, ROMS/Nonlinear/lmd_vmix.F:465-560, LMD_CONVEC: applied by k_lmd_skpp's last
//    sweep; the edge fills :560-760 by the storing thread or the halo kernel)
//
// Pure column physics: one thread per sigma-column; the per-column work arrays of the reference
// (FC, dR, dU, dV, Bflux: 0:N) live in 3-D work arrays so that every level access of a wave is
// coalesced.
#pragma once
#include "roms_ctx.h"
#include "k_libm.h"
#include "k_diag3d.h"

struct LmdArgs {
  Fields Fv;         // the array pointers, by value (a table in device memory would cost every kernel one more dependent round trip)
  DGrid G;
  double fac1, fac2, fac3;   // lmd_swfrac coefficients for Zscale = -1 and this Jerlov water type
  double lmd_Cg, Vtc;
};

#define LMD_CONSTS                                                                                             \
  const double lmd_Ri0 = 0.7, lmd_bvfcon = -2.0E-5, lmd_nu0c = 0.01, lmd_nu0m = 10.0E-4, lmd_nu0s = 10.0E-4;  \
  const double lmd_Ric = 0.3, lmd_am = 1.257, lmd_as = -28.86, lmd_cekman = 0.7, lmd_cmonob = 1.0,            \
               lmd_cm = 8.36, lmd_cs = 98.96, lmd_epsilon = 0.1, lmd_zetam = -0.2, lmd_zetas = -1.0,          \
               vonKar = 0.41;                                                                                  \
  (void)lmd_Ri0; (void)lmd_bvfcon; (void)lmd_nu0c; (void)lmd_nu0m; (void)lmd_nu0s; (void)lmd_Ric;             \
  (void)lmd_am; (void)lmd_as; (void)lmd_cekman; (void)lmd_cmonob; (void)lmd_cm; (void)lmd_cs;                  \
  (void)lmd_epsilon; (void)lmd_zetam; (void)lmd_zetas; (void)vonKar

// LMD_DDMIX, lmd_vmix.F:360-428: the double-diffusive contributions to the temperature | salinity coefficient at W-level k of
// the column at x = X2(i,j) -- salt fingering where 1 < Rrho and salinity increases upward, diffusive convection where
// 0 < Rrho < 1 and it decreases (Marmorino & Caldwell 1976) -- from alfaobeta of rho_eos (G.alfaobeta; k_eos_alfaobeta below)
KDEV void lmd_ddmix(const DGrid &G, const Fields &F, size_t x, int k, size_t ow, double &aktT, double &aktS) {
  const double lmd_Rrho0 = 1.9, lmd_nuf = 10.0E-4, lmd_fdd = 0.7, lmd_nu = 1.5E-6, lmd_tdd1 = 0.909, lmd_tdd2 = 4.6, lmd_tdd3 = 0.54,
               lmd_sdd1 = 0.15, lmd_sdd2 = 1.85, lmd_sdd3 = 0.85;
  const size_t nijN = (size_t)G.nij * (size_t)G.N;
  const size_t oT = x + (size_t)(k - 1) * (size_t)G.nij + (size_t)(G.nstp - 1) * nijN, oS = oT + 3 * nijN;      // t(i,j,k,nstp,itemp | isalt)
  const double ddDT = F.t[oT + (size_t)G.nij] - F.t[oT];
  double ddDS = F.t[oS + (size_t)G.nij] - F.t[oS];
  ddDS = copysign(1.0, ddDS) * KMAX(fabs(ddDS), 1.0E-14);
  double Rrho = G.alfaobeta[ow] * ddDT / ddDS;
  double nu_dds, nu_ddt;
  if ((Rrho > 1.0) && (ddDS > 0.0)) {
    Rrho = KMIN(Rrho, lmd_Rrho0);
    const double q = (Rrho - 1.0) / (lmd_Rrho0 - 1.0);
    nu_dds = 1.0 - q * q;
    nu_dds = lmd_nuf * nu_dds * nu_dds * nu_dds;
    nu_ddt = lmd_fdd * nu_dds;
  } else if ((0.0 < Rrho) && (Rrho < 1.0) && (ddDS < 0.0)) {
    nu_ddt = lmd_nu * lmd_tdd1 * kexp(lmd_tdd2 * kexp(-lmd_tdd3 * ((1.0 / Rrho) - 1.0)));
    if (Rrho < 0.5) nu_dds = nu_ddt * lmd_sdd1 * Rrho;
    else nu_dds = nu_ddt * (lmd_sdd2 * Rrho - lmd_sdd3);
  } else {
    nu_ddt = 0.0;
    nu_dds = 0.0;
  }
  aktT = aktT + nu_ddt;
  aktS = aktS + nu_dds;
}

// vertical parabolic-spline derivatives of R, U, V at W-points for column (i,j) :190-260.  Both
// sweeps take six levels at a time: the inputs of a chunk are loaded first (the loads overlap), then
// the recurrences run on registers.  with_uv = false computes dR only (FC is rebuilt; dU, dV of the
// same velocities are already in their work arrays: k_lmd_skpp after k_lmd_interior).
// Where a column's spline work arrays live: level k of the column is A[off + k*ks].  Global form: the 3-D
// work arrays wrk3[1..4] (ks = nij, off = the column's offset); COL form (k_lmd_col): LDS, ks = KLS, off = 0
// on a pointer already advanced to the thread's lane.  Bflux (wrk3[0]) stays in memory in both forms.
struct LmdWk {
  double *FC, *dR, *dU, *dV;
  size_t ks, off;
};
#define WKI(k) (w.off + (size_t)(k) * w.ks)

KDEV void lmd_col_splines(const DGrid &G, const Fields &F, int i, int j, const double *R, const LmdWk &w, bool with_uv) {
  double *FC = w.FC, *dR = w.dR, *dU = w.dU, *dV = w.dV;
  const int N = G.N;
  const size_t nij = (size_t)G.nij, x = X2(i, j);
  const long ni = G.ni;
  const double *Hz = F.Hz + x, *Rc = R + x;
  const double *u = F.u + (size_t)(G.nstp - 1) * nij * (size_t)N + x, *v = F.v + (size_t)(G.nstp - 1) * nij * (size_t)N + x;
  FC[WKI(0)] = 0.0; dR[WKI(0)] = 0.0;
  if (with_uv) { dU[WKI(0)] = 0.0; dV[WKI(0)] = 0.0; }
  double FCm = 0.0, dRm = 0.0, dUm = 0.0, dVm = 0.0;
  for (int k0 = 1; k0 <= N - 1; k0 += 6) {
    double hz[7], rr[7];      // level k0+q (clamped)
#pragma unroll
    for (int q = 0; q < 7; q++) {
      const size_t o = (size_t)(KMIN(k0 + q, N) - 1) * nij;
      hz[q] = Hz[o]; rr[q] = Rc[o];
    }
#pragma unroll
    for (int q = 0; q < 6; q++) {
      const int k = k0 + q;
      if (k > N - 1) break;
      const size_t ow = WKI(k);
      const double cff = 1.0 / (2.0 * hz[q + 1] + hz[q] * (2.0 - FCm));
      FCm = cff * hz[q + 1];
      dRm = cff * (6.0 * (rr[q + 1] - rr[q]) - hz[q] * dRm);
      FC[ow] = FCm; dR[ow] = dRm;
    }
  }
  if (with_uv) {
    // second pass over the column for the two velocity components (FC is recomputed: same values)
    double FCq = 0.0;
    for (int k0 = 1; k0 <= N - 1; k0 += 6) {
      double hz[7], ua[7], ub[7], va[7], vb[7];
#pragma unroll
      for (int q = 0; q < 7; q++) {
        const size_t o = (size_t)(KMIN(k0 + q, N) - 1) * nij;
        hz[q] = Hz[o]; ua[q] = u[o]; ub[q] = u[o + 1]; va[q] = v[o]; vb[q] = v[o + ni];
      }
#pragma unroll
      for (int q = 0; q < 6; q++) {
        const int k = k0 + q;
        if (k > N - 1) break;
        const size_t ow = WKI(k);
        const double cff = 1.0 / (2.0 * hz[q + 1] + hz[q] * (2.0 - FCq));
        FCq = cff * hz[q + 1];
        dUm = cff * (3.0 * (ua[q + 1] - ua[q] + ub[q + 1] - ub[q]) - hz[q] * dUm);
        dVm = cff * (3.0 * (va[q + 1] - va[q] + vb[q + 1] - vb[q]) - hz[q] * dVm);
        dU[ow] = dUm; dV[ow] = dVm;
      }
    }
  }
  const size_t oN = WKI(N);
  dR[oN] = 0.0;
  if (with_uv) { dU[oN] = 0.0; dV[oN] = 0.0; }
  double r1 = 0.0, u1 = 0.0, v1 = 0.0;
  for (int k0 = N - 1; k0 >= 1; k0 -= 6) {
    double fc[6], dr[6], du[6], dv[6];
#pragma unroll
    for (int q = 0; q < 6; q++) {
      const size_t ow = WKI(KMAX(k0 - q, 1));
      fc[q] = FC[ow]; dr[q] = dR[ow];
      if (with_uv) { du[q] = dU[ow]; dv[q] = dV[ow]; }
    }
    (void)du; (void)dv;
#pragma unroll
    for (int q = 0; q < 6; q++) {
      const int k = k0 - q;
      if (k < 1) break;
      const size_t ow = WKI(k);
      r1 = dr[q] - fc[q] * r1;
      dR[ow] = r1;
      if (with_uv) {
        u1 = du[q] - fc[q] * u1;
        v1 = dv[q] - fc[q] * v1;
        dU[ow] = u1; dV[ow] = v1;
      }
    }
  }
}

// interior mixing: index space (Istr:Iend, Jstr:Jend)
KDEV void lmd_interior_col(const LmdArgs &a, int i, int j, const LmdWk &w) {
  const DGrid &G = a.G;
  const Fields &F = a.Fv;
  LMD_CONSTS;
  const int N = G.N;
  const double eps = 1.0E-14;
  double *dU = w.dU, *dV = w.dV;
  lmd_col_splines(G, F, i, j, F.rho, w, true);
  const size_t nij = (size_t)G.nij, x = X2(i, j);
  const size_t oA = nij * (size_t)(N + 1);          // Akt: salt after temperature
  // the levels are independent: six at a time, loads first
  for (int k0 = 1; k0 <= N - 1; k0 += 6) {
    double du_[6], dv_[6], bv_[6];
#pragma unroll
    for (int q = 0; q < 6; q++) {
      const size_t ow = (size_t)KMIN(k0 + q, N - 1) * nij + x;
      du_[q] = dU[WKI(KMIN(k0 + q, N - 1))]; dv_[q] = dV[WKI(KMIN(k0 + q, N - 1))]; bv_[q] = F.bvf[ow];
    }
#pragma unroll
    for (int q = 0; q < 6; q++) {
      const int k = k0 + q;
      if (k > N - 1) break;
      const size_t ow = (size_t)k * nij + x;
      const double du = du_[q], dv = dv_[q];
      double shear2 = du * du + dv * dv;
      const double bv = bv_[q];
      const double Rig = bv / (shear2 + eps);
      double cff = KMIN(1.0, KMAX(0.0, Rig) / lmd_Ri0);
      double nu_sx = 1.0 - cff * cff;
      nu_sx = nu_sx * nu_sx * nu_sx;
      shear2 = bv / (Rig + eps);
      cff = shear2 * shear2 / (shear2 * shear2 + 16.0E-10);
      nu_sx = cff * nu_sx;
      cff = 1.0 / sqrt(KMAX(bv, 1.0E-7));
      const double lmd_iwm = 1.0E-6 * cff, lmd_iws = 1.0E-7 * cff;
      F.Akv[ow] = lmd_iwm + lmd_nu0m * nu_sx;
      const double akt = lmd_iws + lmd_nu0s * nu_sx;
      double aT = akt, aS = akt;
      if (G.ddmix) lmd_ddmix(G, F, x, k, ow, aT, aS);
      F.Akt[ow] = aT;
      F.Akt[ow + oA] = aS;
    }
  }
}
THREAD_KERNEL(k_lmd_interior, LmdArgs) {
  (void)gz;
  const DGrid &G = a.G;
  const int i = G.T.Istr + gx, j = G.T.Jstr + gy;
  const LmdWk w = {a.Fv.wrk3[1], a.Fv.wrk3[2], a.Fv.wrk3[3], a.Fv.wrk3[4], (size_t)G.nij, X2(i, j)};
  lmd_interior_col(a, i, j, w);
}
THREAD_GLOBAL(k_lmd_interior, LmdArgs)

KDEV void lmd_wscale(double Ustar, double zetahat, double Ustar3, double &wm, double &ws) {
  const double small = 1.0E-20, r3 = 1.0 / 3.0;
  const double vonKar = 0.41, lmd_zetam = -0.2, lmd_zetas = -1.0, lmd_am = 1.257, lmd_as = -28.86, lmd_cm = 8.36,
               lmd_cs = 98.96;
  const double zetapar = zetahat / (Ustar3 + small);
  if (zetahat >= 0.0) {
    wm = vonKar * Ustar / (1.0 + 5.0 * zetapar);
    ws = wm;
  } else {
    if (zetapar > lmd_zetam) wm = vonKar * Ustar * kpow(1.0 - 16.0 * zetapar, 0.25);
    else wm = vonKar * kpow(lmd_am * Ustar3 - lmd_cm * zetahat, r3);
    if (zetapar > lmd_zetas) ws = vonKar * Ustar * sqrt(1.0 - 16.0 * zetapar);   // **0.5_r8 = correctly rounded sqrt (as the reference build compiles it)
    else ws = vonKar * kpow(lmd_as * Ustar3 - lmd_cs * zetahat, r3);
  }
}

#define SWFRAC(Z) (kexp((Z) * a.fac1) * a.fac3 + kexp((Z) * a.fac2) * (1.0 - a.fac3))

// surface boundary layer: index space (Istr:Iend, Jstr:Jend).  hsbl is written on the interior;
// its boundary fill / exchange (bc_r2d_tile, lmd_skpp.F:608) follows in the halo kernel.
KDEV void lmd_skpp_col(const LmdArgs &a, int i, int j, const LmdWk &w) {
  const DGrid &G = a.G;
  const Fields &F = a.Fv;
  LMD_CONSTS;
  const int N = G.N;
  const double eps = 1.0E-10, g = G.g, gorho0 = G.g / G.rho0;
  const double *z_w = F.z_w, *Hz = F.Hz, *pden = F.pden, *bvf = F.bvf;
  const double *u = F.u + (size_t)(G.nstp - 1) * G.nij * N, *v = F.v + (size_t)(G.nstp - 1) * G.nij * N;
  double *FC = w.FC, *dR = w.dR, *dU = w.dU, *dV = w.dV, *Bflux = F.wrk3[0];
  const double zwN = z_w[XW(i, j, N)];
  double hsbl = F.hsbl[X2(i, j)];
  double sl_dpth = lmd_epsilon * (zwN - hsbl);
  double Ustar;
  {
    const double sa = 0.5 * (F.sustr[X2(i, j)] + F.sustr[X2(i + 1, j)]), sc = 0.5 * (F.svstr[X2(i, j)] + F.svstr[X2(i, j + 1)]);
    Ustar = sqrt(sqrt(sa * sa + sc * sc));
  }
  // MASKING (lmd_skpp.F:273-867): rmask of the column; every product below is the reference's statement
  const bool msk = G.masking != 0;
  const double rm = msk ? F.rmask[X2(i, j)] : 1.0;
  if (msk) Ustar = Ustar * rm;
  const double st1 = F.stflx[X2T(i, j, 1)], st2 = F.stflx[X2T(i, j, 2)], sr = F.srflx[X2(i, j)];
  const double Bo = g * (F.alpha[X2(i, j)] * (st1 - sr) - F.beta[X2(i, j)] * st2);
  const double Bosol = g * F.alpha[X2(i, j)] * sr;
  // (independent levels: four at a time, loads first)
  for (int k0 = 0; k0 <= N; k0 += 4) {
    double zw_[4];
#pragma unroll
    for (int q = 0; q < 4; q++) zw_[q] = z_w[XW(i, j, KMIN(k0 + q, N))];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int k = k0 + q;
      if (k > N) break;
      const double swdk = SWFRAC(zwN - zw_[q]);
      double bf = (Bo + Bosol * (1.0 - swdk));
      if (msk) bf = bf * rm;                                   // :317
      Bflux[XW(i, j, k)] = bf;
      const double cff = 1.0 - (0.5 + copysign(0.5, bf));
      F.ghats[XW4(i, j, k, 1)] = -cff * (st1 - sr + sr * (1.0 - swdk));
      F.ghats[XW4(i, j, k, 2)] = cff * st2;
    }
  }
  lmd_col_splines(G, F, i, j, pden, w, false);   // dU, dV: from the interior part (same velocities)
  const double c13 = 1.0 / 3.0, c16 = 1.0 / 6.0;
  const double Rref = pden[X3(i, j, N)] + Hz[X3(i, j, N)] * (c13 * dR[WKI(N)] + c16 * dR[WKI(N - 1)]);
  const double Uref = 0.5 * (u[X3(i, j, N)] + u[X3(i + 1, j, N)]) + Hz[X3(i, j, N)] * (c13 * dU[WKI(N)] + c16 * dU[WKI(N - 1)]);
  const double Vref = 0.5 * (v[X3(i, j, N)] + v[X3(i, j + 1, N)]) + Hz[X3(i, j, N)] * (c13 * dV[WKI(N)] + c16 * dV[WKI(N - 1)]);
  const double Ustar3 = Ustar * Ustar * Ustar;
  double wm = 0.0, ws = 0.0;
  // bulk Richardson number criterion; FC(k) overwrites the spline work array (as the reference)
  FC[WKI(N)] = 0.0;
  // (independent levels: three at a time, loads first)
  for (int k0 = N; k0 >= 1; k0 -= 3) {
    double zwm[3], bfm[3], pd[3], hz[3], drm[3], drk[3], uk[3], vk[3], dum[3], duk[3], dvm[3], dvk[3], bvm[3];
#pragma unroll
    for (int q = 0; q < 3; q++) {
      const int k = KMAX(k0 - q, 1);
      zwm[q] = z_w[XW(i, j, k - 1)]; bfm[q] = Bflux[XW(i, j, k - 1)]; pd[q] = pden[X3(i, j, k)]; hz[q] = Hz[X3(i, j, k)];
      drm[q] = dR[WKI(k - 1)]; drk[q] = dR[WKI(k)];
      uk[q] = 0.5 * (u[X3(i, j, k)] + u[X3(i + 1, j, k)]); vk[q] = 0.5 * (v[X3(i, j, k)] + v[X3(i, j + 1, k)]);
      dum[q] = dU[WKI(k - 1)]; duk[q] = dU[WKI(k)]; dvm[q] = dV[WKI(k - 1)]; dvk[q] = dV[WKI(k)];
      bvm[q] = bvf[XW(i, j, k - 1)];
    }
#pragma unroll
    for (int q = 0; q < 3; q++) {
      const int k = k0 - q;
      if (k < 1) break;
      const double depth = zwN - zwm[q];
      const double bf = bfm[q];
      const double sigma = (bf < 0.0) ? KMIN(sl_dpth, depth) : depth;
      const double zetahat = vonKar * sigma * bf;
      lmd_wscale(Ustar, zetahat, Ustar3, wm, ws);
      const double Rk = pd[q] - hz[q] * (c13 * drm[q] + c16 * drk[q]);
      const double Uk = uk[q] - hz[q] * (c13 * dum[q] + c16 * duk[q]);
      const double Vk = vk[q] - hz[q] * (c13 * dvm[q] + c16 * dvk[q]);
      const double Ritop = -gorho0 * (Rref - Rk) * depth;
      const double du_ = Uref - Uk, dv_ = Vref - Vk;
      const double Ribot = du_ * du_ + dv_ * dv_ + a.Vtc * depth * ws * sqrt(fabs(bvm[q]));
      FC[WKI(k - 1)] = Ritop - lmd_Ric * Ribot;
    }
  }
  int ksbl = 1;
  hsbl = z_w[XW(i, j, 1)];
  for (int k = N; k >= 2; k--) {
    const double fkm = FC[WKI(k - 1)];
    if (ksbl == 1 && fkm > 0.0) {
      const double fk = FC[WKI(k)];
      hsbl = (z_w[XW(i, j, k)] * fkm - z_w[XW(i, j, k - 1)] * fk) / (fkm - fk);
      ksbl = k;
    }
  }
  double Bfsfc = (Bo + Bosol * (1.0 - SWFRAC(msk ? (zwN - hsbl) * rm : zwN - hsbl)));   // zgrid*rmask :563
  if (msk) Bfsfc = Bfsfc * rm;                                 // :575
  if (Ustar > 0.0 && Bfsfc > 0.0) {
    const double hekman = lmd_cekman * Ustar / KMAX(fabs(F.f[X2(i, j)]), eps);
    const double hmonob = lmd_cmonob * Ustar * Ustar * Ustar / KMAX(vonKar * Bfsfc, eps);
    double m = KMIN(hekman, hmonob);
    m = KMIN(m, zwN - hsbl);
    hsbl = (zwN - m);
  }
  hsbl = KMIN(hsbl, zwN);
  hsbl = KMAX(hsbl, z_w[XW(i, j, 0)]);
  if (msk) hsbl = hsbl * rm;                                   // :596
  emit_store(G, emit_plan(G, BC_R, i, j), F.hsbl, hsbl);     // bc_r2d_tile + exchange lmd_skpp.F:608
  ksbl = 1;
  for (int k = N; k >= 2; k--)
    if (ksbl == 1 && z_w[XW(i, j, k - 1)] < hsbl) ksbl = k;
  Bfsfc = (Bo + Bosol * (1.0 - SWFRAC(msk ? (zwN - hsbl) * rm : zwN - hsbl)));          // :670
  if (msk) Bfsfc = Bfsfc * rm;                                 // :682
  sl_dpth = lmd_epsilon * (zwN - hsbl);
  {
    const double cff = (Bfsfc > 0.0) ? 1.0 : lmd_epsilon;
    const double sigma = cff * (zwN - hsbl);
    const double zetahat = vonKar * sigma * Bfsfc;
    lmd_wscale(Ustar, zetahat, Ustar3, wm, ws);
  }
  const double f1 = 5.0 * KMAX(0.0, Bfsfc) * vonKar / (Ustar * Ustar * Ustar * Ustar + eps);
  const double zbl = zwN - hsbl;
  double Gm1, Gt1, Gs1, dGm1dS, dGt1dS, dGs1dS;
  if (hsbl > z_w[XW(i, j, 1)]) {
    const int k = ksbl;
    const double cff = 1.0 / (z_w[XW(i, j, k)] - z_w[XW(i, j, k - 1)]);
    const double cff_dn = cff * (hsbl - z_w[XW(i, j, k - 1)]);
    const double cff_up = cff * (z_w[XW(i, j, k)] - hsbl);
    double K_bl = cff_dn * F.Akv[XW(i, j, k)] + cff_up * F.Akv[XW(i, j, k - 1)];
    double dK_bl = cff * (F.Akv[XW(i, j, k)] - F.Akv[XW(i, j, k - 1)]);
    Gm1 = K_bl / (zbl * wm + eps);
    if (msk) Gm1 = Gm1 * rm;                                   // :755,800
    dGm1dS = KMIN(0.0, -dK_bl / (wm + eps) - K_bl * f1);
    K_bl = cff_dn * F.Akt[XW4(i, j, k, 1)] + cff_up * F.Akt[XW4(i, j, k - 1, 1)];
    dK_bl = cff * (F.Akt[XW4(i, j, k, 1)] - F.Akt[XW4(i, j, k - 1, 1)]);
    Gt1 = K_bl / (zbl * ws + eps);
    if (msk) Gt1 = Gt1 * rm;                                   // :766,809
    dGt1dS = KMIN(0.0, -dK_bl / (ws + eps) - K_bl * f1);
    K_bl = cff_dn * F.Akt[XW4(i, j, k, 2)] + cff_up * F.Akt[XW4(i, j, k - 1, 2)];
    dK_bl = cff * (F.Akt[XW4(i, j, k, 2)] - F.Akt[XW4(i, j, k - 1, 2)]);
    Gs1 = K_bl / (zbl * ws + eps);
    if (msk) Gs1 = Gs1 * rm;                                   // :778
    dGs1dS = KMIN(0.0, -dK_bl / (ws + eps) - K_bl * f1);
  } else {
    ksbl = 0;
    const double ba = 0.5 * (F.bustr[X2(i, j)] + F.bustr[X2(i + 1, j)]), bc = 0.5 * (F.bvstr[X2(i, j)] + F.bvstr[X2(i, j + 1)]);
    double Ustarb = sqrt(sqrt(ba * ba + bc * bc));
    if (msk) Ustarb = Ustarb * rm;                             // :794
    const double dK_bl = vonKar * Ustarb;
    const double K_bl = dK_bl * (hsbl - z_w[XW(i, j, 0)]);
    Gm1 = K_bl / (zbl * wm + eps);
    if (msk) Gm1 = Gm1 * rm;                                   // :755,800
    dGm1dS = KMIN(0.0, -dK_bl / (wm + eps) - K_bl * f1);
    Gt1 = K_bl / (zbl * ws + eps);
    if (msk) Gt1 = Gt1 * rm;                                   // :766,809
    dGt1dS = KMIN(0.0, -dK_bl / (ws + eps) - K_bl * f1);
    Gs1 = Gt1;
    dGs1dS = dGt1dS;
  }
  // (independent levels: four at a time, loads first).  The convective adjustment of lmd_finish
  // (lmd_vmix.F:465-560: Akv, Akt += lmd_nu0c*nu_sxc(bvf), point-wise on the final values) is applied
  // here as the last operation on each level, and the boundary fill / exchange that follows it
  // (:560-760) is done by the storing thread in fused runs.
  const EmitPlan PA = emit_plan(G, BC_R, i, j);
  const size_t nij_ = (size_t)G.nij, x_ = X2(i, j), oA_ = nij_ * (size_t)(N + 1);
  for (int k0 = 1; k0 <= N - 1; k0 += 4) {
    double zw_[4], bf_[4], g1[4], g2[4], av[4], a1_[4], a2_[4], bv[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int k = KMIN(k0 + q, N - 1);
      const size_t ow = (size_t)k * nij_ + x_;
      zw_[q] = z_w[ow]; bf_[q] = Bflux[ow];
      g1[q] = F.ghats[ow]; g2[q] = F.ghats[ow + oA_];
      av[q] = F.Akv[ow]; a1_[q] = F.Akt[ow]; a2_[q] = F.Akt[ow + oA_]; bv[q] = bvf[ow];
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int k = k0 + q;
      if (k > N - 1) break;
      const size_t ow = (size_t)k * nij_;
      double akv = av[q], akt1 = a1_[q], akt2 = a2_[q];
      if (k > ksbl) {
        const double depth = zwN - zw_[q];
        const double bf = bf_[q];
        double sigma = (bf < 0.0) ? KMIN(sl_dpth, depth) : depth;
        const double zetahat = vonKar * sigma * bf;
        lmd_wscale(Ustar, zetahat, Ustar3, wm, ws);
        sigma = depth / (zbl + eps);
        if (msk) sigma = sigma * rm;                           // :867
        const double a1 = sigma - 2.0, a2 = 3.0 - 2.0 * sigma, a3 = sigma - 1.0;
        const double Gm = a1 + a2 * Gm1 + a3 * dGm1dS;
        const double Gt = a1 + a2 * Gt1 + a3 * dGt1dS;
        const double Gs = a1 + a2 * Gs1 + a3 * dGs1dS;
        akv = depth * wm * (1.0 + sigma * Gm);
        akt1 = depth * ws * (1.0 + sigma * Gt);
        akt2 = depth * ws * (1.0 + sigma * Gs);
        const double cff = a.lmd_Cg * (1.0 - (0.5 + copysign(0.5, bf))) / (zbl * ws + eps);
        F.ghats[ow + x_] = cff * g1[q];
        F.ghats[ow + x_ + oA_] = cff * g2[q];
      } else {
        F.ghats[ow + x_] = 0.0;
        F.ghats[ow + x_ + oA_] = 0.0;
      }
      if (G.bkpp) {     // LMD_BKPP: lmd_bkpp works on these values; k_lmd_bkpp applies lmd_finish behind it
        F.Akv[ow + x_] = akv; F.Akt[ow + x_] = akt1; F.Akt[ow + x_ + oA_] = akt2;
        continue;
      }
      // lmd_finish :500-540
      double cff = KMAX(bv[q], lmd_bvfcon);
      cff = KMIN(1.0, (lmd_bvfcon - cff) / lmd_bvfcon);
      double nu_sxc = 1.0 - cff * cff;
      nu_sxc = nu_sxc * nu_sxc * nu_sxc;
      emit_store(G, PA, F.Akv + ow, akv + lmd_nu0c * nu_sxc);
      emit_store(G, PA, F.Akt + ow, akt1 + lmd_nu0c * nu_sxc);
      emit_store(G, PA, F.Akt + ow + oA_, akt2 + lmd_nu0c * nu_sxc);
    }
  }
  if (G.bkpp) F.ksbl[x_] = (double)ksbl;        // MIXING(ng)%ksbl for lmd_bkpp.F:779
}

// ---- LMD_BKPP (round 6): the bottom boundary layer, lmd_bkpp_tile (lmd_bkpp.F:95-806), one thread per column, behind
// k_lmd_interior + k_lmd_skpp (lmd_vmix.F:86-88) and in front of lmd_finish, which this kernel applies as its last operation on
// every level (as k_lmd_skpp does without LMD_BKPP).  RI_SPLINES, SASHA (lmd_bkpp.F:3 defines it for the file), no LMD_SHAPIRO.
// The spline derivatives dR, dU, dV of the column are those k_lmd_interior / k_lmd_skpp left in the work arrays (the same
// recurrences on the same density and velocities, lmd_bkpp.F:311-348); FC is rebuilt as the critical function; the buoyancy
// flux profile (its own Bo: the bottom tracer fluxes, :273) is a function of z_w and is evaluated where the reference reads it.
KDEV void lmd_bkpp_col(const LmdArgs &a, int i, int j, const LmdWk &w) {
  const DGrid &G = a.G;
  const Fields &F = a.Fv;
  LMD_CONSTS;
  const int N = G.N;
  const double eps = 1.0E-10, g = G.g, gorho0 = G.g / G.rho0;
  const double *z_w = F.z_w, *Hz = F.Hz, *pden = F.pden, *bvf = F.bvf;
  const double *u = F.u + (size_t)(G.nstp - 1) * G.nij * N, *v = F.v + (size_t)(G.nstp - 1) * G.nij * N;
  double *FC = w.FC, *dR = w.dR, *dU = w.dU, *dV = w.dV;
  const size_t nij_ = (size_t)G.nij, x_ = X2(i, j), oA_ = nij_ * (size_t)(N + 1);
  const double zwN = z_w[XW(i, j, N)], zw0 = z_w[XW(i, j, 0)];
  double hbbl = F.hbbl[x_];
  double bl_dpth = lmd_epsilon * (hbbl - zw0);                                     // :244
  const bool msk = G.masking != 0;
  const double rm = msk ? F.rmask[x_] : 1.0;
  double Ustar;
  {
    const double ba = 0.5 * (F.bustr[X2(i, j)] + F.bustr[X2(i + 1, j)]), bc = 0.5 * (F.bvstr[X2(i, j)] + F.bvstr[X2(i, j + 1)]);
    Ustar = sqrt(sqrt(ba * ba + bc * bc));                                         // :256
    if (msk) Ustar = Ustar * rm;
  }
  const double Bo = g * (F.alpha[x_] * F.btflx[X2T(i, j, 1)] - F.beta[x_] * F.btflx[X2T(i, j, 2)]);    // :273
  const double Bosol = g * F.alpha[x_] * F.srflx[x_];
#define BKPP_BFLUX(k_, out_) { double bf_ = (Bo + Bosol * (1.0 - SWFRAC(zwN - z_w[XW(i, j, k_)]))); if (msk) bf_ = bf_ * rm; out_ = bf_; }   /* :285-303 */
  const double c13 = 1.0 / 3.0, c16 = 1.0 / 6.0;
  const double Rref = pden[X3(i, j, 1)] - Hz[X3(i, j, 1)] * (c13 * dR[WKI(0)] + c16 * dR[WKI(1)]);      // :410-417
  const double Uref = 0.5 * (u[X3(i, j, 1)] + u[X3(i + 1, j, 1)]) - Hz[X3(i, j, 1)] * (c13 * dU[WKI(0)] + c16 * dU[WKI(1)]);
  const double Vref = 0.5 * (v[X3(i, j, 1)] + v[X3(i, j + 1, 1)]) - Hz[X3(i, j, 1)] * (c13 * dV[WKI(0)] + c16 * dV[WKI(1)]);
  const double Ustar3 = Ustar * Ustar * Ustar;
  double wm = 0.0, ws = 0.0;
  FC[WKI(0)] = 0.0;
  for (int k = 1; k <= N; k++) {                                                   // :424-466
    const double depth = z_w[XW(i, j, k)] - zw0;
    double bf;
    BKPP_BFLUX(k, bf);
    const double sigma = (bf < 0.0) ? KMIN(bl_dpth, depth) : depth;
    const double zetahat = vonKar * sigma * bf;
    lmd_wscale(Ustar, zetahat, Ustar3, wm, ws);
    const double hz = Hz[X3(i, j, k)];
    const double Rk = pden[X3(i, j, k)] + hz * (c13 * dR[WKI(k)] + c16 * dR[WKI(k - 1)]);
    const double Uk = 0.5 * (u[X3(i, j, k)] + u[X3(i + 1, j, k)]) + hz * (c13 * dU[WKI(k)] + c16 * dU[WKI(k - 1)]);
    const double Vk = 0.5 * (v[X3(i, j, k)] + v[X3(i, j + 1, k)]) + hz * (c13 * dV[WKI(k)] + c16 * dV[WKI(k - 1)]);
    const double Ritop = -gorho0 * (Rk - Rref) * depth;
    const double du_ = Uk - Uref, dv_ = Vk - Vref;
    const double Ribot = du_ * du_ + dv_ * dv_ + a.Vtc * depth * ws * sqrt(fabs(bvf[XW(i, j, k)]));
    FC[WKI(k)] = Ritop - lmd_Ric * Ribot;                                          // SASHA :460
  }
  int kbbl = N;
  hbbl = zwN;
  for (int k = 1; k <= N - 1; k++) {                                               // SASHA :474-482
    const double fk = FC[WKI(k)];
    if (kbbl == N && fk > 0.0) {
      const double fkm = FC[WKI(k - 1)];
      hbbl = (z_w[XW(i, j, k)] * fkm - z_w[XW(i, j, k - 1)] * fk) / (fkm - fk);
      kbbl = k;
    }
  }
  if (Ustar >= 0.0) {                                                              // :525-538
    const double hekman = lmd_cekman * Ustar / KMAX(fabs(F.f[x_]), eps) - F.h[x_];
    hbbl = KMIN(hekman, hbbl);
  }
  hbbl = KMIN(hbbl, zwN);
  hbbl = KMAX(hbbl, zw0);
  if (msk) hbbl = hbbl * rm;
  F.hbbl[x_] = hbbl;                                                               // (bc_r2d_tile :577: the halo launch behind)
  kbbl = N;
  for (int k = 1; k <= N; k++)                                                     // :589-598
    if (kbbl == N && z_w[XW(i, j, k)] > hbbl) kbbl = k;
  double Bfbot = (Bo + Bosol * (1.0 - SWFRAC(msk ? (zwN - hbbl) * rm : zwN - hbbl)));   // :604-622
  if (msk) Bfbot = Bfbot * rm;
  bl_dpth = lmd_epsilon * (hbbl - zw0);                                            // :632-662
  {
    const double cff = (Bfbot > 0.0) ? 1.0 : lmd_epsilon;
    const double sigma = cff * (hbbl - zw0);
    const double zetahat = vonKar * sigma * Bfbot;
    lmd_wscale(Ustar, zetahat, Ustar3, wm, ws);
  }
  const double f1 = 5.0 * KMAX(0.0, Bfbot) * vonKar / (Ustar * Ustar * Ustar * Ustar + eps);   // :672-677
  const double zbl = hbbl - zw0;
  double Gm1, Gt1, Gs1, dGm1dS, dGt1dS, dGs1dS;
  {                                                                                // :679-722
    const int k = kbbl;
    const double cff = 1.0 / (z_w[XW(i, j, k)] - z_w[XW(i, j, k - 1)]);
    const double cff_dn = cff * (hbbl - z_w[XW(i, j, k - 1)]);
    const double cff_up = cff * (z_w[XW(i, j, k)] - hbbl);
    double K_bl = cff_dn * F.Akv[XW(i, j, k)] + cff_up * F.Akv[XW(i, j, k - 1)];
    double dK_bl = -cff * (F.Akv[XW(i, j, k)] - F.Akv[XW(i, j, k - 1)]);
    Gm1 = K_bl / (zbl * wm + eps);
    if (msk) Gm1 = Gm1 * rm;
    dGm1dS = KMIN(0.0, K_bl * f1 - dK_bl / (wm + eps));
    K_bl = cff_dn * F.Akt[XW4(i, j, k, 1)] + cff_up * F.Akt[XW4(i, j, k - 1, 1)];
    dK_bl = -cff * (F.Akt[XW4(i, j, k, 1)] - F.Akt[XW4(i, j, k - 1, 1)]);
    Gt1 = K_bl / (zbl * ws + eps);
    if (msk) Gt1 = Gt1 * rm;
    dGt1dS = KMIN(0.0, K_bl * f1 - dK_bl / (ws + eps));
    K_bl = cff_dn * F.Akt[XW4(i, j, k, 2)] + cff_up * F.Akt[XW4(i, j, k - 1, 2)];
    dK_bl = -cff * (F.Akt[XW4(i, j, k, 2)] - F.Akt[XW4(i, j, k - 1, 2)]);
    Gs1 = K_bl / (zbl * ws + eps);
    if (msk) Gs1 = Gs1 * rm;
    dGs1dS = KMIN(0.0, K_bl * f1 - dK_bl / (ws + eps));
  }
  const int ksbl = (int)F.ksbl[x_];
  const EmitPlan PA = emit_plan(G, BC_R, i, j);
  for (int k = 1; k <= N - 1; k++) {                                               // :728-800, then lmd_finish
    const size_t ow = (size_t)k * nij_;
    const double zwk = z_w[ow + x_];
    double akv = F.Akv[ow + x_], akt1 = F.Akt[ow + x_], akt2 = F.Akt[ow + x_ + oA_];
    if (zwk < hbbl) {
      const double depth = zwk - zw0;
      double bf;
      BKPP_BFLUX(k, bf);
      double sigma = (bf < 0.0) ? KMIN(bl_dpth, depth) : depth;
      const double zetahat = vonKar * sigma * bf;
      lmd_wscale(Ustar, zetahat, Ustar3, wm, ws);
      sigma = depth / (hbbl - zw0 + eps);
      if (msk) sigma = sigma * rm;
      const double a1 = sigma - 2.0, a2 = 3.0 - 2.0 * sigma, a3 = sigma - 1.0;
      const double Gm = a1 + a2 * Gm1 + a3 * dGm1dS;
      const double Gt = a1 + a2 * Gt1 + a3 * dGt1dS;
      const double Gs = a1 + a2 * Gs1 + a3 * dGs1dS;
      const double kv = depth * wm * (1.0 + sigma * Gm), kt = depth * ws * (1.0 + sigma * Gt), ks_ = depth * ws * (1.0 + sigma * Gs);
      if (k > ksbl) { akv = KMAX(akv, kv); akt1 = KMAX(akt1, kt); akt2 = KMAX(akt2, ks_); }
      else { akv = kv; akt1 = kt; akt2 = ks_; }
    }
    // lmd_finish (lmd_vmix.F:500-540)
    double cff = KMAX(bvf[ow + x_], lmd_bvfcon);
    cff = KMIN(1.0, (lmd_bvfcon - cff) / lmd_bvfcon);
    double nu_sxc = 1.0 - cff * cff;
    nu_sxc = nu_sxc * nu_sxc * nu_sxc;
    emit_store(G, PA, F.Akv + ow, akv + lmd_nu0c * nu_sxc);
    emit_store(G, PA, F.Akt + ow, akt1 + lmd_nu0c * nu_sxc);
    emit_store(G, PA, F.Akt + ow + oA_, akt2 + lmd_nu0c * nu_sxc);
  }
#undef BKPP_BFLUX
}
THREAD_KERNEL(k_lmd_bkpp, LmdArgs) {
  (void)gz;
  const DGrid &G = a.G;
  const int i = G.T.Istr + gx, j = G.T.Jstr + gy;
  const LmdWk w = {a.Fv.wrk3[1], a.Fv.wrk3[2], a.Fv.wrk3[3], a.Fv.wrk3[4], (size_t)G.nij, X2(i, j)};
  lmd_bkpp_col(a, i, j, w);
}
THREAD_GLOBAL(k_lmd_bkpp, LmdArgs)

THREAD_KERNEL(k_lmd_skpp, LmdArgs) {
  (void)gz;
  const DGrid &G = a.G;
  const int i = G.T.Istr + gx, j = G.T.Jstr + gy;
  const LmdWk w = {a.Fv.wrk3[1], a.Fv.wrk3[2], a.Fv.wrk3[3], a.Fv.wrk3[4], (size_t)G.nij, X2(i, j)};
  lmd_skpp_col(a, i, j, w);
}
THREAD_GLOBAL(k_lmd_skpp, LmdArgs)

// lmd_vmix as ONE column kernel (COL launch): interior mixing, then the surface boundary layer of the same
// column, with the spline work columns in LDS instead of four 3-D work arrays -- they were 2.2-2.6x the algorithmic
// traffic of the two kernels.  Round 4: THREE columns of LDS instead of four, and fewer passes over memory --
//   * the interior part needs no density spline at all (lmd_vmix.F forms dR, but with BV_FREQUENCY its Richardson number
//     takes bvf): one forward sweep builds FC, dU, dV together (FC is the same recurrence in every sweep);
//   * the boundary-layer part reads dU, dV only through the squared shear between the reference level and level k
//     (lmd_skpp.F:420-470): it is formed right behind the interior part and overwrites dU in place (descending: entry k
//     needs dU(k-1), dU(k)), which frees dV's column for the density spline dR;
//   * Bflux is a function of z_w (two exponentials): it is re-evaluated where the reference re-reads its work array,
//     and the preliminary ghats of :330-340 are formed where the final ones are, instead of being stored and read back.
// Three waves per CU instead of two (47.6 KB per wave at N = 30) and 28 instead of 37 array passes.  Every value is
// computed by the two-kernel form's expression on the same operands: same bits (tests: test_column_kernel_forms...,
// ROMS_HIP_LMDCOL=0 selects the two kernels, which tall columns -- 3*(N+1) doubles per column beyond 64 KB -- keep).
// (the three work columns: level k of a column at W0[k * ks], W1[...], W2[...] -- LDS with ks = KLS in the COL kernel,
//  three 3-D work arrays with ks = nij in the THREAD form that tall columns and the launches beside the barotropic loop take)
KDEV void lmd_col_fused(const LmdArgs &a, int i, int j, double *W0, double *W1, double *W2, size_t ks) {
  const DGrid &G = a.G;
  const Fields &F = a.Fv;
  LMD_CONSTS;
  const int N = G.N;
  double *FC = W0, *dU = W1, *dV = W2;        // dU -> SH (squared shear), dV -> dR later
#define LK(k) ((size_t)(k) * ks)
  const size_t nij = (size_t)G.nij, x = X2(i, j);
  const long ni = G.ni;
  const double *Hz = F.Hz + x;
  const double *u = F.u + (size_t)(G.nstp - 1) * nij * (size_t)N + x, *v = F.v + (size_t)(G.nstp - 1) * nij * (size_t)N + x;
  // ---------------------------------------------------------------- interior part (lmd_interior_col)
  {
    const double eps = 1.0E-14;
    FC[LK(0)] = 0.0; dU[LK(0)] = 0.0; dV[LK(0)] = 0.0;
    double FCq = 0.0, dUm = 0.0, dVm = 0.0;
    for (int k0 = 1; k0 <= N - 1; k0 += 6) {
      double hz[7], ua[7], ub[7], va[7], vb[7];
#pragma unroll
      for (int q = 0; q < 7; q++) {
        const size_t o = (size_t)(KMIN(k0 + q, N) - 1) * nij;
        hz[q] = Hz[o]; ua[q] = u[o]; ub[q] = u[o + 1]; va[q] = v[o]; vb[q] = v[o + ni];
      }
#pragma unroll
      for (int q = 0; q < 6; q++) {
        const int k = k0 + q;
        if (k > N - 1) break;
        const double cff = 1.0 / (2.0 * hz[q + 1] + hz[q] * (2.0 - FCq));
        FCq = cff * hz[q + 1];
        dUm = cff * (3.0 * (ua[q + 1] - ua[q] + ub[q + 1] - ub[q]) - hz[q] * dUm);
        dVm = cff * (3.0 * (va[q + 1] - va[q] + vb[q + 1] - vb[q]) - hz[q] * dVm);
        FC[LK(k)] = FCq; dU[LK(k)] = dUm; dV[LK(k)] = dVm;
      }
    }
    dU[LK(N)] = 0.0; dV[LK(N)] = 0.0;
    double u1 = 0.0, v1 = 0.0;
    for (int k = N - 1; k >= 1; k--) {
      const double fc = FC[LK(k)];
      u1 = dU[LK(k)] - fc * u1;
      v1 = dV[LK(k)] - fc * v1;
      dU[LK(k)] = u1; dV[LK(k)] = v1;
    }
    const size_t oA = nij * (size_t)(N + 1);
    for (int k0 = 1; k0 <= N - 1; k0 += 6) {
      double bv_[6];
#pragma unroll
      for (int q = 0; q < 6; q++) bv_[q] = F.bvf[(size_t)KMIN(k0 + q, N - 1) * nij + x];
#pragma unroll
      for (int q = 0; q < 6; q++) {
        const int k = k0 + q;
        if (k > N - 1) break;
        const size_t ow = (size_t)k * nij + x;
        const double du = dU[LK(k)], dv = dV[LK(k)];
        double shear2 = du * du + dv * dv;
        const double bv = bv_[q];
        const double Rig = bv / (shear2 + eps);
        double cff = KMIN(1.0, KMAX(0.0, Rig) / lmd_Ri0);
        double nu_sx = 1.0 - cff * cff;
        nu_sx = nu_sx * nu_sx * nu_sx;
        shear2 = bv / (Rig + eps);
        cff = shear2 * shear2 / (shear2 * shear2 + 16.0E-10);
        nu_sx = cff * nu_sx;
        cff = 1.0 / sqrt(KMAX(bv, 1.0E-7));
        const double lmd_iwm = 1.0E-6 * cff, lmd_iws = 1.0E-7 * cff;
        F.Akv[ow] = lmd_iwm + lmd_nu0m * nu_sx;
        const double akt = lmd_iws + lmd_nu0s * nu_sx;
        double aT = akt, aS = akt;
        if (G.ddmix) lmd_ddmix(G, F, x, k, ow, aT, aS);
        F.Akt[ow] = aT;
        F.Akt[ow + oA] = aS;
      }
    }
  }
  // ---------------------------------------------------------------- boundary-layer part (lmd_skpp_col)
  const double eps = 1.0E-10, g = G.g, gorho0 = G.g / G.rho0;
  const double *z_w = F.z_w, *pden = F.pden, *bvf = F.bvf;
  const double c13 = 1.0 / 3.0, c16 = 1.0 / 6.0;
  const double HzN = F.Hz[X3(i, j, N)];
  const double Uref = 0.5 * (u[(size_t)(N - 1) * nij] + u[(size_t)(N - 1) * nij + 1]) + HzN * (c13 * dU[LK(N)] + c16 * dU[LK(N - 1)]);
  const double Vref = 0.5 * (v[(size_t)(N - 1) * nij] + v[(size_t)(N - 1) * nij + ni]) + HzN * (c13 * dV[LK(N)] + c16 * dV[LK(N - 1)]);
  {   // squared shear between the reference level and level k, in the place of dU (descending: dU(k-1) is still intact)
    double *SH = dU;
    for (int k0 = N; k0 >= 1; k0 -= 3) {
      double hz[3], uk[3], vk[3];
#pragma unroll
      for (int q = 0; q < 3; q++) {
        const size_t o = (size_t)(KMAX(k0 - q, 1) - 1) * nij;
        hz[q] = Hz[o]; uk[q] = 0.5 * (u[o] + u[o + 1]); vk[q] = 0.5 * (v[o] + v[o + ni]);
      }
#pragma unroll
      for (int q = 0; q < 3; q++) {
        const int k = k0 - q;
        if (k < 1) break;
        const double Uk = uk[q] - hz[q] * (c13 * dU[LK(k - 1)] + c16 * dU[LK(k)]);
        const double Vk = vk[q] - hz[q] * (c13 * dV[LK(k - 1)] + c16 * dV[LK(k)]);
        const double du_ = Uref - Uk, dv_ = Vref - Vk;
        SH[LK(k)] = du_ * du_ + dv_ * dv_;
      }
    }
  }
  double *SH = dU, *dR = dV;
  const double zwN = z_w[XW(i, j, N)];
  double hsbl = F.hsbl[X2(i, j)];
  double sl_dpth = lmd_epsilon * (zwN - hsbl);
  double Ustar;
  {
    const double sa = 0.5 * (F.sustr[X2(i, j)] + F.sustr[X2(i + 1, j)]), sc = 0.5 * (F.svstr[X2(i, j)] + F.svstr[X2(i, j + 1)]);
    Ustar = sqrt(sqrt(sa * sa + sc * sc));
  }
  const bool msk = G.masking != 0;
  const double rm = msk ? F.rmask[X2(i, j)] : 1.0;
  if (msk) Ustar = Ustar * rm;
  const double st1 = F.stflx[X2T(i, j, 1)], st2 = F.stflx[X2T(i, j, 2)], sr = F.srflx[X2(i, j)];
  const double Bo = g * (F.alpha[X2(i, j)] * (st1 - sr) - F.beta[X2(i, j)] * st2);
  const double Bosol = g * F.alpha[X2(i, j)] * sr;
  // Bflux(k) and the preliminary ghats(k) of :300-340 as functions of z_w(k)
#define LMD_BF(zw_, swdk_, bf_)                                                                \
  const double swdk_ = SWFRAC(zwN - (zw_));                                                    \
  double bf_ = (Bo + Bosol * (1.0 - swdk_));                                                   \
  if (msk) bf_ = bf_ * rm
  for (int kk = 0; kk <= N; kk += N) {       // levels 0 and N keep the preliminary ghats (the last sweep covers 1 .. N-1)
    LMD_BF(z_w[XW(i, j, kk)], swdk, bf);
    const double cff = 1.0 - (0.5 + copysign(0.5, bf));
    F.ghats[XW4(i, j, kk, 1)] = -cff * (st1 - sr + sr * (1.0 - swdk));
    F.ghats[XW4(i, j, kk, 2)] = cff * st2;
  }
  {   // density spline: FC again (the same recurrence), dR in dV's place
    const double *Rc = pden + x;
    FC[LK(0)] = 0.0; dR[LK(0)] = 0.0;
    double FCm = 0.0, dRm = 0.0;
    for (int k0 = 1; k0 <= N - 1; k0 += 6) {
      double hz[7], rr[7];
#pragma unroll
      for (int q = 0; q < 7; q++) {
        const size_t o = (size_t)(KMIN(k0 + q, N) - 1) * nij;
        hz[q] = Hz[o]; rr[q] = Rc[o];
      }
#pragma unroll
      for (int q = 0; q < 6; q++) {
        const int k = k0 + q;
        if (k > N - 1) break;
        const double cff = 1.0 / (2.0 * hz[q + 1] + hz[q] * (2.0 - FCm));
        FCm = cff * hz[q + 1];
        dRm = cff * (6.0 * (rr[q + 1] - rr[q]) - hz[q] * dRm);
        FC[LK(k)] = FCm; dR[LK(k)] = dRm;
      }
    }
    dR[LK(N)] = 0.0;
    double r1 = 0.0;
    for (int k = N - 1; k >= 1; k--) {
      r1 = dR[LK(k)] - FC[LK(k)] * r1;
      dR[LK(k)] = r1;
    }
  }
  const double Rref = pden[X3(i, j, N)] + HzN * (c13 * dR[LK(N)] + c16 * dR[LK(N - 1)]);
  const double Ustar3 = Ustar * Ustar * Ustar;
  double wm = 0.0, ws = 0.0;
  FC[LK(N)] = 0.0;
  for (int k0 = N; k0 >= 1; k0 -= 3) {
    double zwm[3], pd[3], hz[3], bvm[3];
#pragma unroll
    for (int q = 0; q < 3; q++) {
      const int k = KMAX(k0 - q, 1);
      zwm[q] = z_w[XW(i, j, k - 1)]; pd[q] = pden[X3(i, j, k)]; hz[q] = Hz[(size_t)(k - 1) * nij];
      bvm[q] = bvf[XW(i, j, k - 1)];
    }
#pragma unroll
    for (int q = 0; q < 3; q++) {
      const int k = k0 - q;
      if (k < 1) break;
      const double depth = zwN - zwm[q];
      LMD_BF(zwm[q], swdk, bf);
      (void)swdk;
      const double sigma = (bf < 0.0) ? KMIN(sl_dpth, depth) : depth;
      const double zetahat = vonKar * sigma * bf;
      lmd_wscale(Ustar, zetahat, Ustar3, wm, ws);
      const double Rk = pd[q] - hz[q] * (c13 * dR[LK(k - 1)] + c16 * dR[LK(k)]);
      const double Ritop = -gorho0 * (Rref - Rk) * depth;
      const double Ribot = SH[LK(k)] + a.Vtc * depth * ws * sqrt(fabs(bvm[q]));
      FC[LK(k - 1)] = Ritop - lmd_Ric * Ribot;
    }
  }
  int ksbl = 1;
  hsbl = z_w[XW(i, j, 1)];
  for (int k = N; k >= 2; k--) {
    const double fkm = FC[LK(k - 1)];
    if (ksbl == 1 && fkm > 0.0) {
      const double fk = FC[LK(k)];
      hsbl = (z_w[XW(i, j, k)] * fkm - z_w[XW(i, j, k - 1)] * fk) / (fkm - fk);
      ksbl = k;
    }
  }
  double Bfsfc = (Bo + Bosol * (1.0 - SWFRAC(msk ? (zwN - hsbl) * rm : zwN - hsbl)));   // zgrid*rmask :563
  if (msk) Bfsfc = Bfsfc * rm;                                 // :575
  if (Ustar > 0.0 && Bfsfc > 0.0) {
    const double hekman = lmd_cekman * Ustar / KMAX(fabs(F.f[X2(i, j)]), eps);
    const double hmonob = lmd_cmonob * Ustar * Ustar * Ustar / KMAX(vonKar * Bfsfc, eps);
    double m = KMIN(hekman, hmonob);
    m = KMIN(m, zwN - hsbl);
    hsbl = (zwN - m);
  }
  hsbl = KMIN(hsbl, zwN);
  hsbl = KMAX(hsbl, z_w[XW(i, j, 0)]);
  if (msk) hsbl = hsbl * rm;                                   // :596
  emit_store(G, emit_plan(G, BC_R, i, j), F.hsbl, hsbl);     // bc_r2d_tile + exchange lmd_skpp.F:608
  ksbl = 1;
  for (int k = N; k >= 2; k--)
    if (ksbl == 1 && z_w[XW(i, j, k - 1)] < hsbl) ksbl = k;
  Bfsfc = (Bo + Bosol * (1.0 - SWFRAC(msk ? (zwN - hsbl) * rm : zwN - hsbl)));          // :670
  if (msk) Bfsfc = Bfsfc * rm;                                 // :682
  sl_dpth = lmd_epsilon * (zwN - hsbl);
  {
    const double cff = (Bfsfc > 0.0) ? 1.0 : lmd_epsilon;
    const double sigma = cff * (zwN - hsbl);
    const double zetahat = vonKar * sigma * Bfsfc;
    lmd_wscale(Ustar, zetahat, Ustar3, wm, ws);
  }
  const double f1 = 5.0 * KMAX(0.0, Bfsfc) * vonKar / (Ustar * Ustar * Ustar * Ustar + eps);
  const double zbl = zwN - hsbl;
  double Gm1, Gt1, Gs1, dGm1dS, dGt1dS, dGs1dS;
  if (hsbl > z_w[XW(i, j, 1)]) {
    const int k = ksbl;
    const double cff = 1.0 / (z_w[XW(i, j, k)] - z_w[XW(i, j, k - 1)]);
    const double cff_dn = cff * (hsbl - z_w[XW(i, j, k - 1)]);
    const double cff_up = cff * (z_w[XW(i, j, k)] - hsbl);
    double K_bl = cff_dn * F.Akv[XW(i, j, k)] + cff_up * F.Akv[XW(i, j, k - 1)];
    double dK_bl = cff * (F.Akv[XW(i, j, k)] - F.Akv[XW(i, j, k - 1)]);
    Gm1 = K_bl / (zbl * wm + eps);
    if (msk) Gm1 = Gm1 * rm;                                   // :755,800
    dGm1dS = KMIN(0.0, -dK_bl / (wm + eps) - K_bl * f1);
    K_bl = cff_dn * F.Akt[XW4(i, j, k, 1)] + cff_up * F.Akt[XW4(i, j, k - 1, 1)];
    dK_bl = cff * (F.Akt[XW4(i, j, k, 1)] - F.Akt[XW4(i, j, k - 1, 1)]);
    Gt1 = K_bl / (zbl * ws + eps);
    if (msk) Gt1 = Gt1 * rm;                                   // :766,809
    dGt1dS = KMIN(0.0, -dK_bl / (ws + eps) - K_bl * f1);
    K_bl = cff_dn * F.Akt[XW4(i, j, k, 2)] + cff_up * F.Akt[XW4(i, j, k - 1, 2)];
    dK_bl = cff * (F.Akt[XW4(i, j, k, 2)] - F.Akt[XW4(i, j, k - 1, 2)]);
    Gs1 = K_bl / (zbl * ws + eps);
    if (msk) Gs1 = Gs1 * rm;                                   // :778
    dGs1dS = KMIN(0.0, -dK_bl / (ws + eps) - K_bl * f1);
  } else {
    ksbl = 0;
    const double ba = 0.5 * (F.bustr[X2(i, j)] + F.bustr[X2(i + 1, j)]), bc = 0.5 * (F.bvstr[X2(i, j)] + F.bvstr[X2(i, j + 1)]);
    double Ustarb = sqrt(sqrt(ba * ba + bc * bc));
    if (msk) Ustarb = Ustarb * rm;                             // :794
    const double dK_bl = vonKar * Ustarb;
    const double K_bl = dK_bl * (hsbl - z_w[XW(i, j, 0)]);
    Gm1 = K_bl / (zbl * wm + eps);
    if (msk) Gm1 = Gm1 * rm;                                   // :755,800
    dGm1dS = KMIN(0.0, -dK_bl / (wm + eps) - K_bl * f1);
    Gt1 = K_bl / (zbl * ws + eps);
    if (msk) Gt1 = Gt1 * rm;                                   // :766,809
    dGt1dS = KMIN(0.0, -dK_bl / (ws + eps) - K_bl * f1);
    Gs1 = Gt1;
    dGs1dS = dGt1dS;
  }
  const EmitPlan PA = emit_plan(G, BC_R, i, j);
  const size_t oA_ = nij * (size_t)(N + 1);
  for (int k0 = 1; k0 <= N - 1; k0 += 4) {
    double zw_[4], av[4], a1_[4], a2_[4], bv[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int k = KMIN(k0 + q, N - 1);
      const size_t ow = (size_t)k * nij + x;
      zw_[q] = z_w[ow];
      av[q] = F.Akv[ow]; a1_[q] = F.Akt[ow]; a2_[q] = F.Akt[ow + oA_]; bv[q] = bvf[ow];
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int k = k0 + q;
      if (k > N - 1) break;
      const size_t ow = (size_t)k * nij;
      double akv = av[q], akt1 = a1_[q], akt2 = a2_[q];
      if (k > ksbl) {
        const double depth = zwN - zw_[q];
        LMD_BF(zw_[q], swdk, bf);
        // the preliminary ghats of :330-340 (what the two-kernel form stores first and reads back here)
        const double cffp = 1.0 - (0.5 + copysign(0.5, bf));
        const double g1 = -cffp * (st1 - sr + sr * (1.0 - swdk)), g2 = cffp * st2;
        double sigma = (bf < 0.0) ? KMIN(sl_dpth, depth) : depth;
        const double zetahat = vonKar * sigma * bf;
        lmd_wscale(Ustar, zetahat, Ustar3, wm, ws);
        sigma = depth / (zbl + eps);
        if (msk) sigma = sigma * rm;                           // :867
        const double a1 = sigma - 2.0, a2 = 3.0 - 2.0 * sigma, a3 = sigma - 1.0;
        const double Gm = a1 + a2 * Gm1 + a3 * dGm1dS;
        const double Gt = a1 + a2 * Gt1 + a3 * dGt1dS;
        const double Gs = a1 + a2 * Gs1 + a3 * dGs1dS;
        akv = depth * wm * (1.0 + sigma * Gm);
        akt1 = depth * ws * (1.0 + sigma * Gt);
        akt2 = depth * ws * (1.0 + sigma * Gs);
        const double cff = a.lmd_Cg * (1.0 - (0.5 + copysign(0.5, bf))) / (zbl * ws + eps);
        F.ghats[ow + x] = cff * g1;
        F.ghats[ow + x + oA_] = cff * g2;
      } else {
        F.ghats[ow + x] = 0.0;
        F.ghats[ow + x + oA_] = 0.0;
      }
      // lmd_finish :500-540
      double cff = KMAX(bv[q], lmd_bvfcon);
      cff = KMIN(1.0, (lmd_bvfcon - cff) / lmd_bvfcon);
      double nu_sxc = 1.0 - cff * cff;
      nu_sxc = nu_sxc * nu_sxc * nu_sxc;
      emit_store(G, PA, F.Akv + ow, akv + lmd_nu0c * nu_sxc);
      emit_store(G, PA, F.Akt + ow, akt1 + lmd_nu0c * nu_sxc);
      emit_store(G, PA, F.Akt + ow + oA_, akt2 + lmd_nu0c * nu_sxc);
    }
  }
#undef LMD_BF
#undef LK
}
COL_KERNEL(k_lmd_col, LmdArgs) {
  (void)gz;
  const DGrid &G = a.G;
  const size_t n1 = (size_t)(G.N + 1) * KLS;
  lmd_col_fused(a, G.T.Istr + gx, G.T.Jstr + gy, lds, lds + n1, lds + 2 * n1, (size_t)KLS);
}
COL_GLOBAL(k_lmd_col, LmdArgs)
// the same column function with its three work columns in the 3-D work arrays wrk3[1..3] (any N; many waves per CU, 14
// work-array passes on top of the 28): replaces k_lmd_interior + k_lmd_skpp (ROMS_HIP_LMDCOL=0 still selects those)
THREAD_KERNEL(k_lmd_fused, LmdArgs) {
  (void)gz;
  const DGrid &G = a.G;
  const int i = G.T.Istr + gx, j = G.T.Jstr + gy;
  const size_t x = X2(i, j);
  lmd_col_fused(a, i, j, a.Fv.wrk3[1] + x, a.Fv.wrk3[2] + x, a.Fv.wrk3[3] + x, (size_t)G.nij);
}
THREAD_GLOBAL(k_lmd_fused, LmdArgs)

// ---------------------------------------------------------------------------------------------- block form
// The same column physics with the work of a column spread over the lanes of a block.  k_lmd_col's time is the length of ONE
// column's chain of seven sweeps (12 000 VALU instructions and 700 loads per lane; on 512 x 64 columns 512 waves, two per CU,
// every one alone on its SIMD with VALU active in 27 % of the cycles; larger batches of loads make it slower: measured).
// Here a block of 512 threads owns 64 columns (one eta row) and the sweeps that have no recurrence in k -- the right-hand
// sides of the splines, the interior coefficients, the squared shear, the bulk Richardson criterion with its exponentials
// and cube roots, the last sweep -- run on (column, level) pairs, eight waves wide; the three spline recurrences run side by
// side on three waves, the per-column scalars on the fourth; the search for the boundary-layer depth on one wave.
// LDS: four columns of N+1 levels (FC -> squared shear, dU -> criterion, dV -> z_w, dR) and 21 scalars per column,
// [level][column]: 75.8 KB at N = 30 (two blocks per CU), up to 71 levels.  Measured alone (round 5): BENCHMARK1 91 -> 54 us,
// BENCHMARK3 1093 -> 842 us, config 5 (N = 50, where k_lmd_col does not fit: two kernels, 396 + 186 us) -> 477 us; the
// phases of a block in clock cycles (tools/gpu_debug/lmd_blk_prof.sh, BENCHMARK1): right-hand sides 8.7 K, forward sweeps
// 17.5 K, backward 4.3 K, shear + interior 14 K, criterion 13 K, depth 19.6 K, last sweep 9.5 K.
// Every value is computed by lmd_col_fused's expression on the same operands: same bits
// (tests/test_kernels_emu.py: test_kpp_block_form_bitwise; tests/test_gpu_parity.py: the forms test).
#define LMD_BLK_NS 24
#if defined(LMD_BLK_PROF) && !defined(ROMS_CPU_EMU)
// (test aid: the phases of one block timed with the 100 MHz counter, printed by its first thread; tools/gpu_debug/lmd_blk_prof.sh)
#define LMD_STAMP_DECL unsigned long long st_[8]; st_[0] = __builtin_amdgcn_s_memtime()
#define LMD_STAMP(n_) st_[n_] = __builtin_amdgcn_s_memtime()
#define LMD_STAMP_END                                                                                         \
  do {                                                                                                        \
    st_[7] = __builtin_amdgcn_s_memtime();                                                                    \
    if (KTID == 0 && bx == 1 && by == 7)                                                                      \
      printf("lmd_blk phases (10 ns): rhs %llu fwd %llu bwd %llu shear %llu crit %llu depth %llu last %llu\n", st_[1] - st_[0], st_[2] - st_[1], \
             st_[3] - st_[2], st_[4] - st_[3], st_[5] - st_[4], st_[6] - st_[5], st_[7] - st_[6]);          \
  } while (0)
#else
#define LMD_STAMP_DECL (void)0
#define LMD_STAMP(n_) (void)0
#define LMD_STAMP_END (void)0
#endif
COOP_KERNEL(k_lmd_blk, LmdArgs) {
  (void)bz;
  const DGrid &G = a.G;
  LMD_STAMP_DECL;
  const Fields &F = a.Fv;
  LMD_CONSTS;
  const int N = G.N;
  const int nx = G.T.Iend - G.T.Istr + 1;
  const int c0 = bx * 64, nc = KMIN(64, nx - c0);
  const int j = G.T.Jstr + by, i0 = G.T.Istr + c0;
  const size_t n1 = (size_t)(N + 1) * 64;
  double *FC = lds, *dU = lds + n1, *dV = lds + 2 * n1, *dR = lds + 3 * n1, *SC = lds + 4 * n1;
  double *SH = FC, *CR = dU, *ZW = dV;   // (the squared shear takes FC's place after the splines, the criterion dU's after the shear, z_w dV's)
  enum { S_USTAR, S_BO, S_BOSOL, S_ZWN, S_SLD, S_RM, S_ST1, S_ST2, S_SR, S_UREF, S_VREF, S_RREF, S_ZBL, S_GM1, S_DGM1, S_GT1,
         S_DGT1, S_GS1, S_DGS1, S_KSBL, S_HZN };
#define LK(k) ((size_t)(k) * 64 + (size_t)col)
#define SCV(s_) SC[(size_t)(s_) * 64 + (size_t)col]
  const size_t nij = (size_t)G.nij;
  const long ni = G.ni;
  const bool msk = G.masking != 0;
  const double g = G.g, gorho0 = G.g / G.rho0;
  const double c13 = 1.0 / 3.0, c16 = 1.0 / 6.0;
  const size_t oA = nij * (size_t)(N + 1);
  const double *uS = F.u + (size_t)(G.nstp - 1) * nij * (size_t)N, *vS = F.v + (size_t)(G.nstp - 1) * nij * (size_t)N;
#define LMD_BF(zw_, swdk_, bf_)                                                                \
  const double swdk_ = SWFRAC(zwN - (zw_));                                                    \
  double bf_ = (Bo + Bosol * (1.0 - swdk_));                                                   \
  if (msk) bf_ = bf_ * rm
  // ---- 0: the right-hand sides of the three splines, in the places of dU, dV, dR (no recurrence: all threads)
#pragma unroll 2
  for (int q = KTID; q < 64 * (N - 1); q += KNT) {
    const int col = q & 63, k = (q >> 6) + 1;
    if (col >= nc) continue;
    const size_t x = X2(i0 + col, j);
    const size_t o = (size_t)(k - 1) * nij + x;
    {
      const double pa0 = uS[o], pa1 = uS[o + nij], pb0 = uS[o + 1], pb1 = uS[o + nij + 1];
      dU[LK(k)] = 3.0 * (pa1 - pa0 + pb1 - pb0);
    }
    {
      const double pa0 = vS[o], pa1 = vS[o + nij], pb0 = vS[(long)o + ni], pb1 = vS[(long)(o + nij) + ni];
      dV[LK(k)] = 3.0 * (pa1 - pa0 + pb1 - pb0);
    }
    dR[LK(k)] = 6.0 * (F.pden[o + nij] - F.pden[o]);
  }
  KSYNC();
  LMD_STAMP(1);
  // ---- 1a: the forward sweeps of the three splines (FC stored by the first), and the scalars of the column
  for (int q = KTID; q < 4 * 64; q += KNT) {
    const int s = q >> 6, col = q & 63;
    if (col >= nc) continue;
    const int i = i0 + col;
    const size_t x = X2(i, j);
    const double *Hz = F.Hz + x;
    if (s == 3) {
      const double zwN = F.z_w[XW(i, j, N)];
      const double hsbl = F.hsbl[X2(i, j)];
      const double sl_dpth = lmd_epsilon * (zwN - hsbl);
      double Ustar;
      {
        const double sa = 0.5 * (F.sustr[X2(i, j)] + F.sustr[X2(i + 1, j)]), sc = 0.5 * (F.svstr[X2(i, j)] + F.svstr[X2(i, j + 1)]);
        Ustar = sqrt(sqrt(sa * sa + sc * sc));
      }
      const double rm = msk ? F.rmask[X2(i, j)] : 1.0;
      if (msk) Ustar = Ustar * rm;
      const double st1 = F.stflx[X2T(i, j, 1)], st2 = F.stflx[X2T(i, j, 2)], sr = F.srflx[X2(i, j)];
      const double Bo = g * (F.alpha[X2(i, j)] * (st1 - sr) - F.beta[X2(i, j)] * st2);
      const double Bosol = g * F.alpha[X2(i, j)] * sr;
      for (int kk = 0; kk <= N; kk += N) {       // levels 0 and N keep the preliminary ghats (the last sweep covers 1 .. N-1)
        LMD_BF(F.z_w[XW(i, j, kk)], swdk, bf);
        const double cff = 1.0 - (0.5 + copysign(0.5, bf));
        F.ghats[XW4(i, j, kk, 1)] = -cff * (st1 - sr + sr * (1.0 - swdk));
        F.ghats[XW4(i, j, kk, 2)] = cff * st2;
      }
      SCV(S_USTAR) = Ustar; SCV(S_BO) = Bo; SCV(S_BOSOL) = Bosol; SCV(S_ZWN) = zwN; SCV(S_SLD) = sl_dpth; SCV(S_RM) = rm;
      SCV(S_ST1) = st1; SCV(S_ST2) = st2; SCV(S_SR) = sr; SCV(S_HZN) = F.Hz[X3(i, j, N)];
      continue;
    }
    double *D = s == 0 ? dU : (s == 1 ? dV : dR);
    D[LK(0)] = 0.0;
    if (s == 0) FC[LK(0)] = 0.0;
    double FCq = 0.0, dm = 0.0;
    double hn[11];                                 // (the thicknesses of the next ten levels are under way while these ten are swept)
#pragma unroll
    for (int r = 0; r < 11; r++) hn[r] = Hz[(size_t)(KMIN(1 + r, N) - 1) * nij];
    for (int k0 = 1; k0 <= N - 1; k0 += 10) {
      double hz[11];
#pragma unroll
      for (int r = 0; r < 11; r++) hz[r] = hn[r];
      if (k0 + 10 <= N - 1) {
#pragma unroll
        for (int r = 0; r < 11; r++) hn[r] = Hz[(size_t)(KMIN(k0 + 10 + r, N) - 1) * nij];
      }
#pragma unroll
      for (int r = 0; r < 10; r++) {
        const int k = k0 + r;
        if (k > N - 1) break;
        const double cff = 1.0 / (2.0 * hz[r + 1] + hz[r] * (2.0 - FCq));
        FCq = cff * hz[r + 1];
        dm = cff * (D[LK(k)] - hz[r] * dm);
        if (s == 0) FC[LK(k)] = FCq;
        D[LK(k)] = dm;
      }
    }
    D[LK(N)] = 0.0;
  }
  KSYNC();
  LMD_STAMP(2);
  // ---- 1b: the backward sweeps, and the reference values of the surface level
  for (int q = KTID; q < 3 * 64; q += KNT) {
    const int s = q >> 6, col = q & 63;
    if (col >= nc) continue;
    const int i = i0 + col;
    const size_t x = X2(i, j);
    double *D = s == 0 ? dU : (s == 1 ? dV : dR);
    double x1 = 0.0;
    for (int k = N - 1; k >= 1; k--) {
      x1 = D[LK(k)] - FC[LK(k)] * x1;
      D[LK(k)] = x1;
    }
    const double HzN = F.Hz[X3(i, j, N)];
    if (s == 0) SCV(S_UREF) = 0.5 * (uS[x + (size_t)(N - 1) * nij] + uS[x + (size_t)(N - 1) * nij + 1]) + HzN * (c13 * D[LK(N)] + c16 * D[LK(N - 1)]);
    else if (s == 1) SCV(S_VREF) = 0.5 * (vS[x + (size_t)(N - 1) * nij] + vS[(long)(x + (size_t)(N - 1) * nij) + ni]) + HzN * (c13 * D[LK(N)] + c16 * D[LK(N - 1)]);
    else SCV(S_RREF) = F.pden[X3(i, j, N)] + HzN * (c13 * D[LK(N)] + c16 * D[LK(N - 1)]);
  }
  KSYNC();
  LMD_STAMP(3);
  // ---- 2: the interior coefficients (levels 1 .. N-1) and the squared shear against the reference level (1 .. N)
  for (int q = KTID; q < 64 * N; q += KNT) {
    const int col = q & 63, k = (q >> 6) + 1;
    if (col >= nc) continue;
    const int i = i0 + col;
    const size_t x = X2(i, j);
    const size_t o = (size_t)(k - 1) * nij + x;
    const double du_k = dU[LK(k)], dv_k = dV[LK(k)];
    if (k <= N - 1) {
      const double eps = 1.0E-14;
      const size_t ow = (size_t)k * nij + x;
      double shear2 = du_k * du_k + dv_k * dv_k;
      const double bv = F.bvf[ow];
      const double Rig = bv / (shear2 + eps);
      double cff = KMIN(1.0, KMAX(0.0, Rig) / lmd_Ri0);
      double nu_sx = 1.0 - cff * cff;
      nu_sx = nu_sx * nu_sx * nu_sx;
      shear2 = bv / (Rig + eps);
      cff = shear2 * shear2 / (shear2 * shear2 + 16.0E-10);
      nu_sx = cff * nu_sx;
      cff = 1.0 / sqrt(KMAX(bv, 1.0E-7));
      const double lmd_iwm = 1.0E-6 * cff, lmd_iws = 1.0E-7 * cff;
      F.Akv[ow] = lmd_iwm + lmd_nu0m * nu_sx;
      const double akt = lmd_iws + lmd_nu0s * nu_sx;
      double aT = akt, aS = akt;
      if (G.ddmix) lmd_ddmix(G, F, x, k, ow, aT, aS);
      F.Akt[ow] = aT;
      F.Akt[ow + oA] = aS;
    }
    const double hz = F.Hz[o];
    const double uk = 0.5 * (uS[o] + uS[o + 1]), vk = 0.5 * (vS[o] + vS[(long)o + ni]);
    const double Uk = uk - hz * (c13 * dU[LK(k - 1)] + c16 * du_k);
    const double Vk = vk - hz * (c13 * dV[LK(k - 1)] + c16 * dv_k);
    const double du_ = SCV(S_UREF) - Uk, dv_ = SCV(S_VREF) - Vk;
    SH[LK(k)] = du_ * du_ + dv_ * dv_;
  }
  KSYNC();
  LMD_STAMP(4);
  // ---- 3: the bulk Richardson criterion of level k-1 (lmd_skpp.F:420-470), in dU's place
  for (int q = KTID; q < 64 * N; q += KNT) {
    const int col = q & 63, k = (q >> 6) + 1;
    if (col >= nc) continue;
    const int i = i0 + col;
    const double zwN = SCV(S_ZWN), Bo = SCV(S_BO), Bosol = SCV(S_BOSOL), rm = SCV(S_RM), Ustar = SCV(S_USTAR), sl_dpth = SCV(S_SLD);
    const double Ustar3 = Ustar * Ustar * Ustar;
    const double zwm = F.z_w[XW(i, j, k - 1)], pd = F.pden[X3(i, j, k)], hz = F.Hz[X3(i, j, k)], bvm = F.bvf[XW(i, j, k - 1)];
    const double depth = zwN - zwm;
    LMD_BF(zwm, swdk, bf);
    (void)swdk;
    const double sigma = (bf < 0.0) ? KMIN(sl_dpth, depth) : depth;
    const double zetahat = vonKar * sigma * bf;
    double wm = 0.0, ws = 0.0;
    lmd_wscale(Ustar, zetahat, Ustar3, wm, ws);
    const double Rk = pd - hz * (c13 * dR[LK(k - 1)] + c16 * dR[LK(k)]);
    const double Ritop = -gorho0 * (SCV(S_RREF) - Rk) * depth;
    const double Ribot = SH[LK(k)] + a.Vtc * depth * ws * sqrt(fabs(bvm));
    CR[LK(k - 1)] = Ritop - lmd_Ric * Ribot;
    if (k == N) CR[LK(N)] = 0.0;
    ZW[LK(k - 1)] = zwm;                       // (z_w(0 .. N-1) for the search of the next sweep, in dV's place)
  }
  KSYNC();
  LMD_STAMP(5);
  // ---- 4: the depth of the boundary layer and the shape functions at its base, one thread per column
  for (int col = KTID; col < 64; col += KNT) {
    if (col >= nc) continue;
    const int i = i0 + col;
#define ZWL(k_) ((k_) == N ? zwN : ZW[LK(k_)])
    const double eps = 1.0E-10;
    const double zwN = SCV(S_ZWN), Bo = SCV(S_BO), Bosol = SCV(S_BOSOL), rm = SCV(S_RM), Ustar = SCV(S_USTAR);
    const double Ustar3 = Ustar * Ustar * Ustar;
    int ksbl = 1;
    double hsbl = ZWL(1);
    {   // (the criterion and the depths of both levels read every pass, so that the reads of several passes go out together)
      double fk = CR[LK(N)], zk = zwN;
#pragma unroll 4
      for (int k = N; k >= 2; k--) {
        const double fkm = CR[LK(k - 1)], zkm = ZW[LK(k - 1)];
        if (ksbl == 1 && fkm > 0.0) {
          hsbl = (zk * fkm - zkm * fk) / (fkm - fk);
          ksbl = k;
        }
        fk = fkm; zk = zkm;
      }
    }
    double Bfsfc = (Bo + Bosol * (1.0 - SWFRAC(msk ? (zwN - hsbl) * rm : zwN - hsbl)));   // zgrid*rmask :563
    if (msk) Bfsfc = Bfsfc * rm;                                 // :575
    if (Ustar > 0.0 && Bfsfc > 0.0) {
      const double hekman = lmd_cekman * Ustar / KMAX(fabs(F.f[X2(i, j)]), eps);
      const double hmonob = lmd_cmonob * Ustar * Ustar * Ustar / KMAX(vonKar * Bfsfc, eps);
      double m = KMIN(hekman, hmonob);
      m = KMIN(m, zwN - hsbl);
      hsbl = (zwN - m);
    }
    hsbl = KMIN(hsbl, zwN);
    hsbl = KMAX(hsbl, ZWL(0));
    if (msk) hsbl = hsbl * rm;                                   // :596
    emit_store(G, emit_plan(G, BC_R, i, j), F.hsbl, hsbl);     // bc_r2d_tile + exchange lmd_skpp.F:608
    ksbl = 1;
#pragma unroll 4
    for (int k = N; k >= 2; k--) {
      const double zkm = ZW[LK(k - 1)];
      if (ksbl == 1 && zkm < hsbl) ksbl = k;
    }
    Bfsfc = (Bo + Bosol * (1.0 - SWFRAC(msk ? (zwN - hsbl) * rm : zwN - hsbl)));          // :670
    if (msk) Bfsfc = Bfsfc * rm;                                 // :682
    const double sl_dpth = lmd_epsilon * (zwN - hsbl);
    double wm = 0.0, ws = 0.0;
    {
      const double cff = (Bfsfc > 0.0) ? 1.0 : lmd_epsilon;
      const double sigma = cff * (zwN - hsbl);
      const double zetahat = vonKar * sigma * Bfsfc;
      lmd_wscale(Ustar, zetahat, Ustar3, wm, ws);
    }
    const double f1 = 5.0 * KMAX(0.0, Bfsfc) * vonKar / (Ustar * Ustar * Ustar * Ustar + eps);
    const double zbl = zwN - hsbl;
    double Gm1, Gt1, Gs1, dGm1dS, dGt1dS, dGs1dS;
    if (hsbl > ZWL(1)) {
      const int k = ksbl;
      const double cff = 1.0 / (ZWL(k) - ZWL(k - 1));
      const double cff_dn = cff * (hsbl - ZWL(k - 1));
      const double cff_up = cff * (ZWL(k) - hsbl);
      double K_bl = cff_dn * F.Akv[XW(i, j, k)] + cff_up * F.Akv[XW(i, j, k - 1)];
      double dK_bl = cff * (F.Akv[XW(i, j, k)] - F.Akv[XW(i, j, k - 1)]);
      Gm1 = K_bl / (zbl * wm + eps);
      if (msk) Gm1 = Gm1 * rm;                                   // :755,800
      dGm1dS = KMIN(0.0, -dK_bl / (wm + eps) - K_bl * f1);
      K_bl = cff_dn * F.Akt[XW4(i, j, k, 1)] + cff_up * F.Akt[XW4(i, j, k - 1, 1)];
      dK_bl = cff * (F.Akt[XW4(i, j, k, 1)] - F.Akt[XW4(i, j, k - 1, 1)]);
      Gt1 = K_bl / (zbl * ws + eps);
      if (msk) Gt1 = Gt1 * rm;                                   // :766,809
      dGt1dS = KMIN(0.0, -dK_bl / (ws + eps) - K_bl * f1);
      K_bl = cff_dn * F.Akt[XW4(i, j, k, 2)] + cff_up * F.Akt[XW4(i, j, k - 1, 2)];
      dK_bl = cff * (F.Akt[XW4(i, j, k, 2)] - F.Akt[XW4(i, j, k - 1, 2)]);
      Gs1 = K_bl / (zbl * ws + eps);
      if (msk) Gs1 = Gs1 * rm;                                   // :778
      dGs1dS = KMIN(0.0, -dK_bl / (ws + eps) - K_bl * f1);
    } else {
      ksbl = 0;
      const double ba = 0.5 * (F.bustr[X2(i, j)] + F.bustr[X2(i + 1, j)]), bc = 0.5 * (F.bvstr[X2(i, j)] + F.bvstr[X2(i, j + 1)]);
      double Ustarb = sqrt(sqrt(ba * ba + bc * bc));
      if (msk) Ustarb = Ustarb * rm;                             // :794
      const double dK_bl = vonKar * Ustarb;
      const double K_bl = dK_bl * (hsbl - ZWL(0));
      Gm1 = K_bl / (zbl * wm + eps);
      if (msk) Gm1 = Gm1 * rm;                                   // :755,800
      dGm1dS = KMIN(0.0, -dK_bl / (wm + eps) - K_bl * f1);
      Gt1 = K_bl / (zbl * ws + eps);
      if (msk) Gt1 = Gt1 * rm;                                   // :766,809
      dGt1dS = KMIN(0.0, -dK_bl / (ws + eps) - K_bl * f1);
      Gs1 = Gt1;
      dGs1dS = dGt1dS;
    }
    SCV(S_SLD) = sl_dpth; SCV(S_ZBL) = zbl; SCV(S_KSBL) = (double)ksbl;
    SCV(S_GM1) = Gm1; SCV(S_DGM1) = dGm1dS; SCV(S_GT1) = Gt1; SCV(S_DGT1) = dGt1dS; SCV(S_GS1) = Gs1; SCV(S_DGS1) = dGs1dS;
#undef ZWL
  }
  KSYNC();
  LMD_STAMP(6);
  // ---- 5: the last sweep (boundary-layer coefficients above ksbl, nonlocal transport, lmd_finish), levels 1 .. N-1
  for (int q = KTID; q < 64 * (N - 1); q += KNT) {
    const int col = q & 63, k = (q >> 6) + 1;
    if (col >= nc) continue;
    const int i = i0 + col;
    const size_t x = X2(i, j);
    const double eps = 1.0E-10;
    const EmitPlan PA = emit_plan(G, BC_R, i, j);
    const double zwN = SCV(S_ZWN), Bo = SCV(S_BO), Bosol = SCV(S_BOSOL), rm = SCV(S_RM), Ustar = SCV(S_USTAR), sl_dpth = SCV(S_SLD);
    const double st1 = SCV(S_ST1), st2 = SCV(S_ST2), sr = SCV(S_SR), zbl = SCV(S_ZBL);
    const double Ustar3 = Ustar * Ustar * Ustar;
    const int ksbl = (int)SCV(S_KSBL);
    const size_t ow = (size_t)k * nij;
    const double zw_ = F.z_w[ow + x], bv = F.bvf[ow + x];
    double akv = F.Akv[ow + x], akt1 = F.Akt[ow + x], akt2 = F.Akt[ow + x + oA];
    if (k > ksbl) {
      const double depth = zwN - zw_;
      LMD_BF(zw_, swdk, bf);
      // the preliminary ghats of :330-340 (what the two-kernel form stores first and reads back here)
      const double cffp = 1.0 - (0.5 + copysign(0.5, bf));
      const double g1 = -cffp * (st1 - sr + sr * (1.0 - swdk)), g2 = cffp * st2;
      double sigma = (bf < 0.0) ? KMIN(sl_dpth, depth) : depth;
      const double zetahat = vonKar * sigma * bf;
      double wm = 0.0, ws = 0.0;
      lmd_wscale(Ustar, zetahat, Ustar3, wm, ws);
      sigma = depth / (zbl + eps);
      if (msk) sigma = sigma * rm;                           // :867
      const double a1 = sigma - 2.0, a2 = 3.0 - 2.0 * sigma, a3 = sigma - 1.0;
      const double Gm = a1 + a2 * SCV(S_GM1) + a3 * SCV(S_DGM1);
      const double Gt = a1 + a2 * SCV(S_GT1) + a3 * SCV(S_DGT1);
      const double Gs = a1 + a2 * SCV(S_GS1) + a3 * SCV(S_DGS1);
      akv = depth * wm * (1.0 + sigma * Gm);
      akt1 = depth * ws * (1.0 + sigma * Gt);
      akt2 = depth * ws * (1.0 + sigma * Gs);
      const double cff = a.lmd_Cg * (1.0 - (0.5 + copysign(0.5, bf))) / (zbl * ws + eps);
      F.ghats[ow + x] = cff * g1;
      F.ghats[ow + x + oA] = cff * g2;
    } else {
      F.ghats[ow + x] = 0.0;
      F.ghats[ow + x + oA] = 0.0;
    }
    // lmd_finish :500-540
    double cff = KMAX(bv, lmd_bvfcon);
    cff = KMIN(1.0, (lmd_bvfcon - cff) / lmd_bvfcon);
    double nu_sxc = 1.0 - cff * cff;
    nu_sxc = nu_sxc * nu_sxc * nu_sxc;
    emit_store(G, PA, F.Akv + ow, akv + lmd_nu0c * nu_sxc);
    emit_store(G, PA, F.Akt + ow, akt1 + lmd_nu0c * nu_sxc);
    emit_store(G, PA, F.Akt + ow + oA, akt2 + lmd_nu0c * nu_sxc);
  }
  LMD_STAMP_END;
#undef LMD_BF
#undef LK
#undef SCV
}
COOP_GLOBAL_LB(k_lmd_blk, LmdArgs, 512)
