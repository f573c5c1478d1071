/*
 * orc_lmd.c -- Large/McWilliams/Doney (1994) K-profile vertical mixing.
 * TEST INFRASTRUCTURE (see orc.h).
 *
 *   orc_lmd_swfrac   lmd_swfrac_tile   ROMS/Nonlinear/lmd_swfrac.F:6-140  (Paulson & Simpson 1977)
 *   lmd_interior     lmd_vmix_tile     ROMS/Nonlinear/lmd_vmix.F:99-460   (LMD_RIMIX, RI_SPLINES)
 *   lmd_skpp         lmd_skpp_tile     ROMS/Nonlinear/lmd_skpp.F:98-930   (LMD_SKPP, LMD_NONLOCAL)
 *   lmd_finish       lmd_finish_tile   ROMS/Nonlinear/lmd_vmix.F:465-760  (LMD_CONVEC)
 *   orc_lmd_vmix     lmd_vmix          ROMS/Nonlinear/lmd_vmix.F:45
 * Constants: ROMS/Modules/mod_scalars.F:1110-1215, lmd_Cg :2862.
 * PARITY: pinned (lmd_*.F build in oracle/_ref).
 */
#include "orc.h"
#include <math.h>
#include <stdlib.h>
#define MAX(a, b) ((a) > (b) ? (a) : (b))
#define MIN(a, b) ((a) < (b) ? (a) : (b))
#define CX(A, i, k) A[(size_t)((i) - LBi) + (size_t)(k) * ni]

static const double lmd_mu1[9] = {0.35, 0.6, 1.0, 1.5, 1.4, 0.42, 0.37, 0.33, 0.00468592};
static const double lmd_mu2[9] = {23.0, 20.0, 17.0, 14.0, 7.9, 5.13, 3.54, 2.34, 1.51};
static const double lmd_r1[9] = {0.58, 0.62, 0.67, 0.77, 0.78, 0.57, 0.57, 0.57, 0.55};
static const double lmd_Ri0 = 0.7, lmd_bvfcon = -2.0E-5, lmd_nu0c = 0.01, lmd_nu0m = 10.0E-4, lmd_nu0s = 10.0E-4;
static const double lmd_Cstar = 10.0, lmd_Cv = 1.25, lmd_Ric = 0.3, lmd_am = 1.257, lmd_as = -28.86,
                    lmd_betaT = -0.2, lmd_cekman = 0.7, lmd_cmonob = 1.0, lmd_cm = 8.36, lmd_cs = 98.96,
                    lmd_epsilon = 0.1, lmd_zetam = -0.2, lmd_zetas = -1.0, vonKar = 0.41;

/* swdk = fraction of solar shortwave flux penetrating to depth Z*Zscale */
void orc_lmd_swfrac(const orc_t *o, const orc_bounds *b, double Zscale, const double *Z, double *swdk) {
  ORC_LOCALS(o);
  const int Jindex = o->c.lmd_Jwt;
  const double fac1 = Zscale / lmd_mu1[Jindex - 1], fac2 = Zscale / lmd_mu2[Jindex - 1], fac3 = lmd_r1[Jindex - 1];
  for (int j = b->Jstr; j <= b->Jend; j++)
    for (int i = b->Istr; i <= b->Iend; i++)
      swdk[X2(i, j)] = exp(Z[X2(i, j)] * fac1) * fac3 + exp(Z[X2(i, j)] * fac2) * (1.0 - fac3);
}

/* turbulent velocity scales wm, ws (lmd_skpp.F "lmd_wscale" in-lined code) */
static void wscale(double Ustar, double zetahat, double Ustar3, double *wm, double *ws) {
  const double small = 1.0E-20, r3 = 1.0 / 3.0;
  const double zetapar = zetahat / (Ustar3 + small);
  if (zetahat >= 0.0) {
    *wm = vonKar * Ustar / (1.0 + 5.0 * zetapar);
    *ws = *wm;
  } else {
    if (zetapar > lmd_zetam) *wm = vonKar * Ustar * pow(1.0 - 16.0 * zetapar, 0.25);
    else *wm = vonKar * pow(lmd_am * Ustar3 - lmd_cm * zetahat, r3);
    if (zetapar > lmd_zetas) *ws = vonKar * Ustar * sqrt(1.0 - 16.0 * zetapar);   /* **0.5_r8: the correctly rounded sqrt, as the reference build compiles it; pow(x,0.5) is 1 ulp off in rare cases */
    else *ws = vonKar * pow(lmd_as * Ustar3 - lmd_cs * zetahat, r3);
  }
}

/* vertical parabolic-spline derivatives of (R, U, V) at W-points on row j for column i range */
static void col_splines(const orc_t *o, int j, int i0, int i1, const double *R, int nstp, double *FC, double *dR,
                        double *dU, double *dV) {
  ORC_LOCALS(o);
  const double *Hz = o->Hz, *u = o->u, *v = o->v;
  for (int i = i0; i <= i1; i++) { CX(FC, i, 0) = 0.0; CX(dR, i, 0) = 0.0; CX(dU, i, 0) = 0.0; CX(dV, i, 0) = 0.0; }
  for (int k = 1; k <= N - 1; k++)
    for (int i = i0; i <= i1; i++) {
      const double cff = 1.0 / (2.0 * Hz[X3(i, j, k + 1)] + Hz[X3(i, j, k)] * (2.0 - CX(FC, i, k - 1)));
      CX(FC, i, k) = cff * Hz[X3(i, j, k + 1)];
      CX(dR, i, k) = cff * (6.0 * (R[X3(i, j, k + 1)] - R[X3(i, j, k)]) - Hz[X3(i, j, k)] * CX(dR, i, k - 1));
      CX(dU, i, k) = cff * (3.0 * (u[X4(i, j, k + 1, nstp)] - u[X4(i, j, k, nstp)] + u[X4(i + 1, j, k + 1, nstp)] -
                                   u[X4(i + 1, j, k, nstp)]) -
                            Hz[X3(i, j, k)] * CX(dU, i, k - 1));
      CX(dV, i, k) = cff * (3.0 * (v[X4(i, j, k + 1, nstp)] - v[X4(i, j, k, nstp)] + v[X4(i, j + 1, k + 1, nstp)] -
                                   v[X4(i, j + 1, k, nstp)]) -
                            Hz[X3(i, j, k)] * CX(dV, i, k - 1));
    }
  for (int i = i0; i <= i1; i++) { CX(dR, i, N) = 0.0; CX(dU, i, N) = 0.0; CX(dV, i, N) = 0.0; }
  for (int k = N - 1; k >= 1; k--)
    for (int i = i0; i <= i1; i++) {
      CX(dR, i, k) = CX(dR, i, k) - CX(FC, i, k) * CX(dR, i, k + 1);
      CX(dU, i, k) = CX(dU, i, k) - CX(FC, i, k) * CX(dU, i, k + 1);
      CX(dV, i, k) = CX(dV, i, k) - CX(FC, i, k) * CX(dV, i, k + 1);
    }
}

static void lmd_interior(orc_t *o, const orc_bounds *b) {
  ORC_LOCALS(o);
  const int nstp = o->s.nstp;
  const double eps = 1.0E-14;
  double *bvf = o->bvf, *Akv = o->Akv, *Akt = o->Akt;
  const size_t cs = ni * (size_t)(N + 1);
  double *FC = (double *)calloc(4 * cs, sizeof(double)), *dR = FC + cs, *dU = FC + 2 * cs, *dV = FC + 3 * cs;
  double *Rig = (double *)calloc(nij * (size_t)(N + 1), sizeof(double));
  const int j0 = MAX(1, b->Jstr - 1), j1 = MIN(b->Jend + 1, o->c.Mm);
  const int i0 = MAX(1, b->Istr - 1), i1 = MIN(b->Iend + 1, o->c.Lm);
  for (int j = j0; j <= j1; j++) {
    col_splines(o, j, i0, i1, o->rho, nstp, FC, dR, dU, dV);
    for (int k = 1; k <= N - 1; k++)
      for (int i = i0; i <= i1; i++) {
        const double shear2 = CX(dU, i, k) * CX(dU, i, k) + CX(dV, i, k) * CX(dV, i, k);
        Rig[XW(i, j, k)] = bvf[XW(i, j, k)] / (shear2 + eps);
      }
  }
  for (int k = 1; k <= N - 1; k++)
    for (int j = b->Jstr; j <= b->Jend; j++)
      for (int i = b->Istr; i <= b->Iend; i++) {
        double cff = MIN(1.0, MAX(0.0, Rig[XW(i, j, k)]) / lmd_Ri0);
        double nu_sx = 1.0 - cff * cff;
        nu_sx = nu_sx * nu_sx * nu_sx;
        const double shear2 = bvf[XW(i, j, k)] / (Rig[XW(i, j, k)] + eps);
        cff = shear2 * shear2 / (shear2 * shear2 + 16.0E-10);
        nu_sx = cff * nu_sx;
        cff = 1.0 / sqrt(MAX(bvf[XW(i, j, k)], 1.0E-7));
        const double lmd_iwm = 1.0E-6 * cff, lmd_iws = 1.0E-7 * cff;
        Akv[XW(i, j, k)] = lmd_iwm + lmd_nu0m * nu_sx;
        Akt[XW4(i, j, k, 1)] = lmd_iws + lmd_nu0s * nu_sx;
        Akt[XW4(i, j, k, 2)] = Akt[XW4(i, j, k, 1)];
        if (o->ddmix) {
          /* LMD_DDMIX, lmd_vmix.F:360-428: double-diffusive mixing where the density gradient is stable and that of
             salinity (salt fingering) or temperature (diffusive convection) is not */
          const double lmd_Rrho0 = 1.9, lmd_nuf = 10.0E-4, lmd_fdd = 0.7, lmd_nu = 1.5E-6, lmd_tdd1 = 0.909, lmd_tdd2 = 4.6,
                       lmd_tdd3 = 0.54, lmd_sdd1 = 0.15, lmd_sdd2 = 1.85, lmd_sdd3 = 0.85;
          const double ddDT = o->t[XT(i, j, k + 1, nstp, 1)] - o->t[XT(i, j, k, nstp, 1)];
          double ddDS = o->t[XT(i, j, k + 1, nstp, 2)] - o->t[XT(i, j, k, nstp, 2)];
          ddDS = copysign(1.0, ddDS) * MAX(fabs(ddDS), 1.0E-14);
          double Rrho = o->alfaobeta[XW(i, j, k)] * ddDT / ddDS;
          double nu_dds, nu_ddt;
          if ((Rrho > 1.0) && (ddDS > 0.0)) {                                  /* salt fingering */
            Rrho = MIN(Rrho, lmd_Rrho0);
            const double q = (Rrho - 1.0) / (lmd_Rrho0 - 1.0);
            nu_dds = 1.0 - q * q;
            nu_dds = lmd_nuf * nu_dds * nu_dds * nu_dds;
            nu_ddt = lmd_fdd * nu_dds;
          } else if ((0.0 < Rrho) && (Rrho < 1.0) && (ddDS < 0.0)) {          /* diffusive convection */
            nu_ddt = lmd_nu * lmd_tdd1 * exp(lmd_tdd2 * exp(-lmd_tdd3 * ((1.0 / Rrho) - 1.0)));
            if (Rrho < 0.5) nu_dds = nu_ddt * lmd_sdd1 * Rrho;
            else nu_dds = nu_ddt * (lmd_sdd2 * Rrho - lmd_sdd3);
          } else {
            nu_ddt = 0.0;
            nu_dds = 0.0;
          }
          Akt[XW4(i, j, k, 1)] = Akt[XW4(i, j, k, 1)] + nu_ddt;
          Akt[XW4(i, j, k, 2)] = Akt[XW4(i, j, k, 2)] + nu_dds;
        }
      }
  free(FC);
  free(Rig);
}

static void lmd_skpp(orc_t *o, const orc_bounds *b) {
  ORC_LOCALS(o);
  const int msk = (o->c.options & ORC_MASKING) != 0;
  const int nstp = o->s.nstp;
  const int Istr = b->Istr, Iend = b->Iend, Jstr = b->Jstr, Jend = b->Jend;
  const double eps = 1.0E-10, g = o->c.g, gorho0 = o->c.g / o->c.rho0;
  const double lmd_Cg = lmd_Cstar * vonKar * pow(lmd_cs * vonKar * lmd_epsilon, 1.0 / 3.0);
  double *z_w = o->z_w, *Hz = o->Hz, *u = o->u, *v = o->v, *pden = o->pden, *bvf = o->bvf;
  double *Akv = o->Akv, *Akt = o->Akt, *ghats = o->ghats, *hsbl = o->hsbl;
  double *stflx = o->stflx, *srflx = o->srflx, *sustr = o->sustr, *svstr = o->svstr;
  int *ksbl = o->ksbl;
  const size_t cs = ni * (size_t)(N + 1);
  double *FC = (double *)calloc(4 * cs, sizeof(double)), *dR = FC + cs, *dU = FC + 2 * cs, *dV = FC + 3 * cs;
  double *Bflux = (double *)calloc(nij * (size_t)(N + 1), sizeof(double));
  double *S = (double *)calloc(17 * nij, sizeof(double));
  double *Bo = S, *Bosol = S + nij, *Bfsfc = S + 2 * nij, *Gm1 = S + 3 * nij, *Gt1 = S + 4 * nij, *Gs1 = S + 5 * nij,
         *Ustar = S + 6 * nij, *dGm1dS = S + 7 * nij, *dGt1dS = S + 8 * nij, *dGs1dS = S + 9 * nij, *f1 = S + 10 * nij,
         *sl_dpth = S + 11 * nij, *swdk = S + 12 * nij, *wm = S + 13 * nij, *ws = S + 14 * nij, *zgrid = S + 15 * nij;
  double *Rref = (double *)calloc(3 * ni, sizeof(double)), *Uref = Rref + ni, *Vref = Rref + 2 * ni;
  const double Vtc = lmd_Cv * sqrt(-lmd_betaT) / (sqrt(lmd_cs * lmd_epsilon) * lmd_Ric * vonKar * vonKar);
  double cff, cff1, cff2;

  for (int j = Jstr; j <= Jend; j++)
    for (int i = Istr; i <= Iend; i++) sl_dpth[X2(i, j)] = lmd_epsilon * (z_w[XW(i, j, N)] - hsbl[X2(i, j)]);
  for (int j = Jstr; j <= Jend; j++)
    for (int i = Istr; i <= Iend; i++) {
      const double a = 0.5 * (sustr[X2(i, j)] + sustr[X2(i + 1, j)]), c = 0.5 * (svstr[X2(i, j)] + svstr[X2(i, j + 1)]);
      Ustar[X2(i, j)] = sqrt(sqrt(a * a + c * c));
      if (msk) Ustar[X2(i, j)] = Ustar[X2(i, j)] * o->rmask[X2(i, j)];                    /* lmd_skpp.F:273 */
    }
  for (int j = Jstr; j <= Jend; j++)
    for (int i = Istr; i <= Iend; i++) {
      Bo[X2(i, j)] = g * (o->alpha[X2(i, j)] * (stflx[X2T(i, j, 1)] - srflx[X2(i, j)]) -
                          o->beta[X2(i, j)] * stflx[X2T(i, j, 2)]);
      Bosol[X2(i, j)] = g * o->alpha[X2(i, j)] * srflx[X2(i, j)];
    }
  for (int k = 0; k <= N; k++) {
    for (int j = Jstr; j <= Jend; j++)
      for (int i = Istr; i <= Iend; i++) zgrid[X2(i, j)] = z_w[XW(i, j, N)] - z_w[XW(i, j, k)];
    orc_lmd_swfrac(o, b, -1.0, zgrid, swdk);
    for (int j = Jstr; j <= Jend; j++)
      for (int i = Istr; i <= Iend; i++) {
        Bflux[XW(i, j, k)] = (Bo[X2(i, j)] + Bosol[X2(i, j)] * (1.0 - swdk[X2(i, j)]));
        if (msk) Bflux[XW(i, j, k)] = Bflux[XW(i, j, k)] * o->rmask[X2(i, j)];            /* :317 */
        cff = 1.0 - (0.5 + copysign(0.5, Bflux[XW(i, j, k)]));
        ghats[XW4(i, j, k, 1)] = -cff * (stflx[X2T(i, j, 1)] - srflx[X2(i, j)] + srflx[X2(i, j)] * (1.0 - swdk[X2(i, j)]));
        ghats[XW4(i, j, k, 2)] = cff * stflx[X2T(i, j, 2)];
      }
  }
  /* bulk Richardson number and boundary layer depth */
  for (int j = Jstr; j <= Jend; j++) {
    col_splines(o, j, Istr, Iend, pden, nstp, FC, dR, dU, dV);
    cff1 = 1.0 / 3.0;
    cff2 = 1.0 / 6.0;
    for (int i = Istr; i <= Iend; i++) {
      Rref[i - LBi] = pden[X3(i, j, N)] + Hz[X3(i, j, N)] * (cff1 * CX(dR, i, N) + cff2 * CX(dR, i, N - 1));
      Uref[i - LBi] = 0.5 * (u[X4(i, j, N, nstp)] + u[X4(i + 1, j, N, nstp)]) +
                      Hz[X3(i, j, N)] * (cff1 * CX(dU, i, N) + cff2 * CX(dU, i, N - 1));
      Vref[i - LBi] = 0.5 * (v[X4(i, j, N, nstp)] + v[X4(i, j + 1, N, nstp)]) +
                      Hz[X3(i, j, N)] * (cff1 * CX(dV, i, N) + cff2 * CX(dV, i, N - 1));
    }
    for (int i = Istr; i <= Iend; i++) {
      CX(FC, i, N) = 0.0;
      for (int k = N; k >= 1; k--) {
        const double depth = z_w[XW(i, j, N)] - z_w[XW(i, j, k - 1)];
        double sigma;
        if (Bflux[XW(i, j, k - 1)] < 0.0) sigma = MIN(sl_dpth[X2(i, j)], depth);
        else sigma = depth;
        const double Us = Ustar[X2(i, j)];
        const double Ustar3 = Us * Us * Us;
        const double zetahat = vonKar * sigma * Bflux[XW(i, j, k - 1)];
        wscale(Us, zetahat, Ustar3, &wm[X2(i, j)], &ws[X2(i, j)]);
        const double Rk = pden[X3(i, j, k)] - Hz[X3(i, j, k)] * (cff1 * CX(dR, i, k - 1) + cff2 * CX(dR, i, k));
        const double Uk = 0.5 * (u[X4(i, j, k, nstp)] + u[X4(i + 1, j, k, nstp)]) -
                          Hz[X3(i, j, k)] * (cff1 * CX(dU, i, k - 1) + cff2 * CX(dU, i, k));
        const double Vk = 0.5 * (v[X4(i, j, k, nstp)] + v[X4(i, j + 1, k, nstp)]) -
                          Hz[X3(i, j, k)] * (cff1 * CX(dV, i, k - 1) + cff2 * CX(dV, i, k));
        const double Ritop = -gorho0 * (Rref[i - LBi] - Rk) * depth;
        const double du_ = Uref[i - LBi] - Uk, dv_ = Vref[i - LBi] - Vk;
        const double Ribot = du_ * du_ + dv_ * dv_ + Vtc * depth * ws[X2(i, j)] * sqrt(fabs(bvf[XW(i, j, k - 1)]));
        CX(FC, i, k - 1) = Ritop - lmd_Ric * Ribot;
      }
    }
    for (int i = Istr; i <= Iend; i++) { ksbl[X2(i, j)] = 1; hsbl[X2(i, j)] = z_w[XW(i, j, 1)]; }
    for (int k = N; k >= 2; k--)
      for (int i = Istr; i <= Iend; i++)
        if (ksbl[X2(i, j)] == 1 && CX(FC, i, k - 1) > 0.0) {
          hsbl[X2(i, j)] = (z_w[XW(i, j, k)] * CX(FC, i, k - 1) - z_w[XW(i, j, k - 1)] * CX(FC, i, k)) /
                           (CX(FC, i, k - 1) - CX(FC, i, k));
          ksbl[X2(i, j)] = k;
        }
  }
  for (int j = Jstr; j <= Jend; j++)
    for (int i = Istr; i <= Iend; i++) {
      zgrid[X2(i, j)] = z_w[XW(i, j, N)] - hsbl[X2(i, j)];
      if (msk) zgrid[X2(i, j)] = zgrid[X2(i, j)] * o->rmask[X2(i, j)];                    /* :563,670 */
    }
  orc_lmd_swfrac(o, b, -1.0, zgrid, swdk);
  for (int j = Jstr; j <= Jend; j++)
    for (int i = Istr; i <= Iend; i++) {
      Bfsfc[X2(i, j)] = (Bo[X2(i, j)] + Bosol[X2(i, j)] * (1.0 - swdk[X2(i, j)]));
      if (msk) Bfsfc[X2(i, j)] = Bfsfc[X2(i, j)] * o->rmask[X2(i, j)];                    /* :575,682 */
    }
  /* stable-case limits: Ekman and Monin-Obukhov depths */
  for (int j = Jstr; j <= Jend; j++)
    for (int i = Istr; i <= Iend; i++) {
      const double Us = Ustar[X2(i, j)];
      if (Us > 0.0 && Bfsfc[X2(i, j)] > 0.0) {
        const double hekman = lmd_cekman * Us / MAX(fabs(o->f[X2(i, j)]), eps);
        const double hmonob = lmd_cmonob * Us * Us * Us / MAX(vonKar * Bfsfc[X2(i, j)], eps);
        double m = MIN(hekman, hmonob);
        m = MIN(m, z_w[XW(i, j, N)] - hsbl[X2(i, j)]);
        hsbl[X2(i, j)] = (z_w[XW(i, j, N)] - m);
      }
      hsbl[X2(i, j)] = MIN(hsbl[X2(i, j)], z_w[XW(i, j, N)]);
      hsbl[X2(i, j)] = MAX(hsbl[X2(i, j)], z_w[XW(i, j, 0)]);
      if (msk) hsbl[X2(i, j)] = hsbl[X2(i, j)] * o->rmask[X2(i, j)];                      /* :596 */
    }
  orc_bc_r2d(o, b, hsbl);
  for (int j = Jstr; j <= Jend; j++)
    for (int i = Istr; i <= Iend; i++) {
      ksbl[X2(i, j)] = 1;
      for (int k = N; k >= 2; k--)
        if (ksbl[X2(i, j)] == 1 && z_w[XW(i, j, k - 1)] < hsbl[X2(i, j)]) ksbl[X2(i, j)] = k;
    }
  for (int j = Jstr; j <= Jend; j++)
    for (int i = Istr; i <= Iend; i++) {
      zgrid[X2(i, j)] = z_w[XW(i, j, N)] - hsbl[X2(i, j)];
      if (msk) zgrid[X2(i, j)] = zgrid[X2(i, j)] * o->rmask[X2(i, j)];                    /* :563,670 */
    }
  orc_lmd_swfrac(o, b, -1.0, zgrid, swdk);
  for (int j = Jstr; j <= Jend; j++)
    for (int i = Istr; i <= Iend; i++) {
      Bfsfc[X2(i, j)] = (Bo[X2(i, j)] + Bosol[X2(i, j)] * (1.0 - swdk[X2(i, j)]));
      if (msk) Bfsfc[X2(i, j)] = Bfsfc[X2(i, j)] * o->rmask[X2(i, j)];                    /* :575,682 */
    }
  /* turbulent velocity scales at the boundary layer depth */
  for (int j = Jstr; j <= Jend; j++)
    for (int i = Istr; i <= Iend; i++) {
      sl_dpth[X2(i, j)] = lmd_epsilon * (z_w[XW(i, j, N)] - hsbl[X2(i, j)]);
      cff = (Bfsfc[X2(i, j)] > 0.0) ? 1.0 : lmd_epsilon;
      const double sigma = cff * (z_w[XW(i, j, N)] - hsbl[X2(i, j)]);
      const double Us = Ustar[X2(i, j)];
      const double Ustar3 = Us * Us * Us;
      const double zetahat = vonKar * sigma * Bfsfc[X2(i, j)];
      wscale(Us, zetahat, Ustar3, &wm[X2(i, j)], &ws[X2(i, j)]);
    }
  for (int j = Jstr; j <= Jend; j++)
    for (int i = Istr; i <= Iend; i++) {
      const double Us = Ustar[X2(i, j)];
      f1[X2(i, j)] = 5.0 * MAX(0.0, Bfsfc[X2(i, j)]) * vonKar / (Us * Us * Us * Us + eps);
    }
  /* shape-function coefficients at the boundary layer depth */
  for (int j = Jstr; j <= Jend; j++)
    for (int i = Istr; i <= Iend; i++) {
      const double zbl = z_w[XW(i, j, N)] - hsbl[X2(i, j)];
      double K_bl, dK_bl;
      if (hsbl[X2(i, j)] > z_w[XW(i, j, 1)]) {
        const int k = ksbl[X2(i, j)];
        cff = 1.0 / (z_w[XW(i, j, k)] - z_w[XW(i, j, k - 1)]);
        const double cff_dn = cff * (hsbl[X2(i, j)] - z_w[XW(i, j, k - 1)]);
        const double cff_up = cff * (z_w[XW(i, j, k)] - hsbl[X2(i, j)]);
        K_bl = cff_dn * Akv[XW(i, j, k)] + cff_up * Akv[XW(i, j, k - 1)];
        dK_bl = cff * (Akv[XW(i, j, k)] - Akv[XW(i, j, k - 1)]);
        Gm1[X2(i, j)] = K_bl / (zbl * wm[X2(i, j)] + eps);
        if (msk) Gm1[X2(i, j)] = Gm1[X2(i, j)] * o->rmask[X2(i, j)];                      /* :755,800 */
        dGm1dS[X2(i, j)] = MIN(0.0, -dK_bl / (wm[X2(i, j)] + eps) - K_bl * f1[X2(i, j)]);
        K_bl = cff_dn * Akt[XW4(i, j, k, 1)] + cff_up * Akt[XW4(i, j, k - 1, 1)];
        dK_bl = cff * (Akt[XW4(i, j, k, 1)] - Akt[XW4(i, j, k - 1, 1)]);
        Gt1[X2(i, j)] = K_bl / (zbl * ws[X2(i, j)] + eps);
        if (msk) Gt1[X2(i, j)] = Gt1[X2(i, j)] * o->rmask[X2(i, j)];                      /* :766,809 */
        dGt1dS[X2(i, j)] = MIN(0.0, -dK_bl / (ws[X2(i, j)] + eps) - K_bl * f1[X2(i, j)]);
        K_bl = cff_dn * Akt[XW4(i, j, k, 2)] + cff_up * Akt[XW4(i, j, k - 1, 2)];
        dK_bl = cff * (Akt[XW4(i, j, k, 2)] - Akt[XW4(i, j, k - 1, 2)]);
        Gs1[X2(i, j)] = K_bl / (zbl * ws[X2(i, j)] + eps);
        if (msk) Gs1[X2(i, j)] = Gs1[X2(i, j)] * o->rmask[X2(i, j)];                      /* :778 */
        dGs1dS[X2(i, j)] = MIN(0.0, -dK_bl / (ws[X2(i, j)] + eps) - K_bl * f1[X2(i, j)]);
      } else {
        ksbl[X2(i, j)] = 0;
        const double a = 0.5 * (o->bustr[X2(i, j)] + o->bustr[X2(i + 1, j)]),
                     c = 0.5 * (o->bvstr[X2(i, j)] + o->bvstr[X2(i, j + 1)]);
        double Ustarb = sqrt(sqrt(a * a + c * c));
        if (msk) Ustarb = Ustarb * o->rmask[X2(i, j)];                                    /* :794 */
        dK_bl = vonKar * Ustarb;
        K_bl = dK_bl * (hsbl[X2(i, j)] - z_w[XW(i, j, 0)]);
        Gm1[X2(i, j)] = K_bl / (zbl * wm[X2(i, j)] + eps);
        if (msk) Gm1[X2(i, j)] = Gm1[X2(i, j)] * o->rmask[X2(i, j)];                      /* :755,800 */
        dGm1dS[X2(i, j)] = MIN(0.0, -dK_bl / (wm[X2(i, j)] + eps) - K_bl * f1[X2(i, j)]);
        Gt1[X2(i, j)] = K_bl / (zbl * ws[X2(i, j)] + eps);
        if (msk) Gt1[X2(i, j)] = Gt1[X2(i, j)] * o->rmask[X2(i, j)];                      /* :766,809 */
        dGt1dS[X2(i, j)] = MIN(0.0, -dK_bl / (ws[X2(i, j)] + eps) - K_bl * f1[X2(i, j)]);
        Gs1[X2(i, j)] = Gt1[X2(i, j)];
        dGs1dS[X2(i, j)] = dGt1dS[X2(i, j)];
      }
    }
  /* boundary layer mixing coefficients and non-local transport */
  for (int k = 1; k <= N - 1; k++)
    for (int j = Jstr; j <= Jend; j++)
      for (int i = Istr; i <= Iend; i++) {
        const double zbl = z_w[XW(i, j, N)] - hsbl[X2(i, j)];
        if (k > ksbl[X2(i, j)]) {
          const double depth = z_w[XW(i, j, N)] - z_w[XW(i, j, k)];
          double sigma;
          if (Bflux[XW(i, j, k)] < 0.0) sigma = MIN(sl_dpth[X2(i, j)], depth);
          else sigma = depth;
          const double Us = Ustar[X2(i, j)];
          const double Ustar3 = Us * Us * Us;
          const double zetahat = vonKar * sigma * Bflux[XW(i, j, k)];
          wscale(Us, zetahat, Ustar3, &wm[X2(i, j)], &ws[X2(i, j)]);
          sigma = depth / (zbl + eps);
          if (msk) sigma = sigma * o->rmask[X2(i, j)];                                    /* :867 */
          const double a1 = sigma - 2.0, a2 = 3.0 - 2.0 * sigma, a3 = sigma - 1.0;
          const double Gm = a1 + a2 * Gm1[X2(i, j)] + a3 * dGm1dS[X2(i, j)];
          const double Gt = a1 + a2 * Gt1[X2(i, j)] + a3 * dGt1dS[X2(i, j)];
          const double Gs = a1 + a2 * Gs1[X2(i, j)] + a3 * dGs1dS[X2(i, j)];
          Akv[XW(i, j, k)] = depth * wm[X2(i, j)] * (1.0 + sigma * Gm);
          Akt[XW4(i, j, k, 1)] = depth * ws[X2(i, j)] * (1.0 + sigma * Gt);
          Akt[XW4(i, j, k, 2)] = depth * ws[X2(i, j)] * (1.0 + sigma * Gs);
          cff = lmd_Cg * (1.0 - (0.5 + copysign(0.5, Bflux[XW(i, j, k)]))) / (zbl * ws[X2(i, j)] + eps);
          ghats[XW4(i, j, k, 1)] = cff * ghats[XW4(i, j, k, 1)];
          ghats[XW4(i, j, k, 2)] = cff * ghats[XW4(i, j, k, 2)];
        } else {
          ghats[XW4(i, j, k, 1)] = 0.0;
          ghats[XW4(i, j, k, 2)] = 0.0;
        }
      }
  free(FC);
  free(Bflux);
  free(S);
  free(Rref);
}

/* lmd_bkpp_tile (lmd_bkpp.F:95-806; LMD_BKPP, round 6): the bottom boundary layer of the K-profile scheme, behind lmd_skpp
   (lmd_vmix.F:86-88).  RI_SPLINES, SASHA (the file defines it for itself, lmd_bkpp.F:3), no LMD_SHAPIRO -- the sub-options of the applications the
   oracle is pinned to.
   hbbl of the previous step gives the first guess of the layer's Monin-Obukhov fraction (:244); the scratch planes wm, ws are
   overwritten level by level as in the reference. */
static void lmd_bkpp(orc_t *o, const orc_bounds *b) {
  ORC_LOCALS(o);
  const int msk = (o->c.options & ORC_MASKING) != 0;
  const int nstp = o->s.nstp;
  const int Istr = b->Istr, Iend = b->Iend, Jstr = b->Jstr, Jend = b->Jend;
  const double eps = 1.0E-10, g = o->c.g, gorho0 = o->c.g / o->c.rho0;
  double *z_w = o->z_w, *Hz = o->Hz, *u = o->u, *v = o->v, *pden = o->pden, *bvf = o->bvf;
  double *Akv = o->Akv, *Akt = o->Akt, *hbbl = o->hbbl;
  double *btflx = o->btflx, *srflx = o->srflx, *bustr = o->bustr, *bvstr = o->bvstr;
  int *kbbl = o->kbbl, *ksbl = o->ksbl;
  const size_t cs = ni * (size_t)(N + 1);
  double *FC = (double *)calloc(4 * cs, sizeof(double)), *dR = FC + cs, *dU = FC + 2 * cs, *dV = FC + 3 * cs;
  double *Bflux = (double *)calloc(nij * (size_t)(N + 1), sizeof(double));
  double *S = (double *)calloc(17 * nij, sizeof(double));
  double *Bo = S, *Bosol = S + nij, *Bfbot = S + 2 * nij, *Gm1 = S + 3 * nij, *Gt1 = S + 4 * nij, *Gs1 = S + 5 * nij,
         *Ustar = S + 6 * nij, *dGm1dS = S + 7 * nij, *dGt1dS = S + 8 * nij, *dGs1dS = S + 9 * nij, *f1 = S + 10 * nij,
         *bl_dpth = S + 11 * nij, *swdk = S + 12 * nij, *wm = S + 13 * nij, *ws = S + 14 * nij, *zgrid = S + 15 * nij;
  double *Rref = (double *)calloc(3 * ni, sizeof(double)), *Uref = Rref + ni, *Vref = Rref + 2 * ni;
  const double Vtc = lmd_Cv * sqrt(-lmd_betaT) / (sqrt(lmd_cs * lmd_epsilon) * lmd_Ric * vonKar * vonKar);   /* :234 */
  double cff, cff1, cff2;

  for (int j = Jstr; j <= Jend; j++)
    for (int i = Istr; i <= Iend; i++) bl_dpth[X2(i, j)] = lmd_epsilon * (hbbl[X2(i, j)] - z_w[XW(i, j, 0)]);     /* :244 */
  for (int j = Jstr; j <= Jend; j++)
    for (int i = Istr; i <= Iend; i++) {                                                                        /* :256 */
      const double a = 0.5 * (bustr[X2(i, j)] + bustr[X2(i + 1, j)]), c = 0.5 * (bvstr[X2(i, j)] + bvstr[X2(i, j + 1)]);
      Ustar[X2(i, j)] = sqrt(sqrt(a * a + c * c));
      if (msk) Ustar[X2(i, j)] = Ustar[X2(i, j)] * o->rmask[X2(i, j)];
    }
  for (int j = Jstr; j <= Jend; j++)
    for (int i = Istr; i <= Iend; i++) {                                                                        /* :273 (SALINITY) */
      Bo[X2(i, j)] = g * (o->alpha[X2(i, j)] * btflx[X2T(i, j, 1)] - o->beta[X2(i, j)] * btflx[X2T(i, j, 2)]);
      Bosol[X2(i, j)] = g * o->alpha[X2(i, j)] * srflx[X2(i, j)];
    }
  for (int k = 0; k <= N; k++) {                                                                                /* :285-303 */
    for (int j = Jstr; j <= Jend; j++)
      for (int i = Istr; i <= Iend; i++) zgrid[X2(i, j)] = z_w[XW(i, j, N)] - z_w[XW(i, j, k)];
    orc_lmd_swfrac(o, b, -1.0, zgrid, swdk);
    for (int j = Jstr; j <= Jend; j++)
      for (int i = Istr; i <= Iend; i++) {
        Bflux[XW(i, j, k)] = (Bo[X2(i, j)] + Bosol[X2(i, j)] * (1.0 - swdk[X2(i, j)]));
        if (msk) Bflux[XW(i, j, k)] = Bflux[XW(i, j, k)] * o->rmask[X2(i, j)];
      }
  }
  /* bulk Richardson number and the depth of the layer :308-502 */
  for (int j = Jstr; j <= Jend; j++) {
    col_splines(o, j, Istr, Iend, pden, nstp, FC, dR, dU, dV);
    cff1 = 1.0 / 3.0;
    cff2 = 1.0 / 6.0;
    for (int i = Istr; i <= Iend; i++) {                                                                        /* :410-417 */
      Rref[i - LBi] = pden[X3(i, j, 1)] - Hz[X3(i, j, 1)] * (cff1 * CX(dR, i, 0) + cff2 * CX(dR, i, 1));
      Uref[i - LBi] = 0.5 * (u[X4(i, j, 1, nstp)] + u[X4(i + 1, j, 1, nstp)]) -
                      Hz[X3(i, j, 1)] * (cff1 * CX(dU, i, 0) + cff2 * CX(dU, i, 1));
      Vref[i - LBi] = 0.5 * (v[X4(i, j, 1, nstp)] + v[X4(i, j + 1, 1, nstp)]) -
                      Hz[X3(i, j, 1)] * (cff1 * CX(dV, i, 0) + cff2 * CX(dV, i, 1));
    }
    for (int i = Istr; i <= Iend; i++) {
      CX(FC, i, 0) = 0.0;
      for (int k = 1; k <= N; k++) {                                                                            /* :424-466 */
        const double depth = z_w[XW(i, j, k)] - z_w[XW(i, j, 0)];
        double sigma;
        if (Bflux[XW(i, j, k)] < 0.0) sigma = MIN(bl_dpth[X2(i, j)], depth);
        else sigma = depth;
        const double Us = Ustar[X2(i, j)];
        const double Ustar3 = Us * Us * Us;
        const double zetahat = vonKar * sigma * Bflux[XW(i, j, k)];
        wscale(Us, zetahat, Ustar3, &wm[X2(i, j)], &ws[X2(i, j)]);
        const double Rk = pden[X3(i, j, k)] + Hz[X3(i, j, k)] * (cff1 * CX(dR, i, k) + cff2 * CX(dR, i, k - 1));
        const double Uk = 0.5 * (u[X4(i, j, k, nstp)] + u[X4(i + 1, j, k, nstp)]) +
                          Hz[X3(i, j, k)] * (cff1 * CX(dU, i, k) + cff2 * CX(dU, i, k - 1));
        const double Vk = 0.5 * (v[X4(i, j, k, nstp)] + v[X4(i, j + 1, k, nstp)]) +
                          Hz[X3(i, j, k)] * (cff1 * CX(dV, i, k) + cff2 * CX(dV, i, k - 1));
        const double Ritop = -gorho0 * (Rk - Rref[i - LBi]) * depth;
        const double Ribot = (Uk - Uref[i - LBi]) * (Uk - Uref[i - LBi]) + (Vk - Vref[i - LBi]) * (Vk - Vref[i - LBi]) +
                             Vtc * depth * ws[X2(i, j)] * sqrt(fabs(bvf[XW(i, j, k)]));
        CX(FC, i, k) = Ritop - lmd_Ric * Ribot;                                                                 /* SASHA (defined at lmd_bkpp.F:3) :460 */
      }
    }
    for (int i = Istr; i <= Iend; i++) {
      kbbl[X2(i, j)] = N;
      hbbl[X2(i, j)] = z_w[XW(i, j, N)];
    }
    for (int k = 1; k <= N - 1; k++)                                                                            /* SASHA :474-482 */
      for (int i = Istr; i <= Iend; i++)
        if (kbbl[X2(i, j)] == N && CX(FC, i, k) > 0.0) {
          hbbl[X2(i, j)] = (z_w[XW(i, j, k)] * CX(FC, i, k - 1) - z_w[XW(i, j, k - 1)] * CX(FC, i, k)) / (CX(FC, i, k - 1) - CX(FC, i, k));
          kbbl[X2(i, j)] = k;
        }
  }
  /* (Bfbot at this depth, :501-523, is computed and not used before it is computed again at :600-622) */
  for (int j = Jstr; j <= Jend; j++)
    for (int i = Istr; i <= Iend; i++) {                                                                        /* :525-538 */
      if (Ustar[X2(i, j)] >= 0.0) {
        const double hekman = lmd_cekman * Ustar[X2(i, j)] / MAX(fabs(o->f[X2(i, j)]), eps) - o->h[X2(i, j)];
        hbbl[X2(i, j)] = MIN(hekman, hbbl[X2(i, j)]);
      }
      hbbl[X2(i, j)] = MIN(hbbl[X2(i, j)], z_w[XW(i, j, N)]);
      hbbl[X2(i, j)] = MAX(hbbl[X2(i, j)], z_w[XW(i, j, 0)]);
      if (msk) hbbl[X2(i, j)] = hbbl[X2(i, j)] * o->rmask[X2(i, j)];
    }
  orc_bc_r2d(o, b, hbbl);                                                                                       /* :577 */
  for (int j = Jstr; j <= Jend; j++)
    for (int i = Istr; i <= Iend; i++) {                                                                        /* :589-598 */
      kbbl[X2(i, j)] = N;
      for (int k = 1; k <= N; k++)
        if (kbbl[X2(i, j)] == N && z_w[XW(i, j, k)] > hbbl[X2(i, j)]) kbbl[X2(i, j)] = k;
    }
  for (int j = Jstr; j <= Jend; j++)
    for (int i = Istr; i <= Iend; i++) {                                                                        /* :604-610 */
      zgrid[X2(i, j)] = z_w[XW(i, j, N)] - hbbl[X2(i, j)];
      if (msk) zgrid[X2(i, j)] = zgrid[X2(i, j)] * o->rmask[X2(i, j)];
    }
  orc_lmd_swfrac(o, b, -1.0, zgrid, swdk);
  for (int j = Jstr; j <= Jend; j++)
    for (int i = Istr; i <= Iend; i++) {
      Bfbot[X2(i, j)] = (Bo[X2(i, j)] + Bosol[X2(i, j)] * (1.0 - swdk[X2(i, j)]));
      if (msk) Bfbot[X2(i, j)] = Bfbot[X2(i, j)] * o->rmask[X2(i, j)];
    }
  for (int j = Jstr; j <= Jend; j++)
    for (int i = Istr; i <= Iend; i++) {                                                                        /* :632-662 */
      bl_dpth[X2(i, j)] = lmd_epsilon * (hbbl[X2(i, j)] - z_w[XW(i, j, 0)]);
      cff = Bfbot[X2(i, j)] > 0.0 ? 1.0 : lmd_epsilon;
      const double sigma = cff * (hbbl[X2(i, j)] - z_w[XW(i, j, 0)]);
      const double Us = Ustar[X2(i, j)];
      const double Ustar3 = Us * Us * Us;
      const double zetahat = vonKar * sigma * Bfbot[X2(i, j)];
      wscale(Us, zetahat, Ustar3, &wm[X2(i, j)], &ws[X2(i, j)]);
    }
  for (int j = Jstr; j <= Jend; j++)
    for (int i = Istr; i <= Iend; i++) {                                                                        /* :672-677 */
      const double Us = Ustar[X2(i, j)];
      f1[X2(i, j)] = 5.0 * MAX(0.0, Bfbot[X2(i, j)]) * vonKar / (Us * Us * Us * Us + eps);
    }
  for (int j = Jstr; j <= Jend; j++)
    for (int i = Istr; i <= Iend; i++) {                                                                        /* :679-722 */
      const double zbl = hbbl[X2(i, j)] - z_w[XW(i, j, 0)];
      const int k = kbbl[X2(i, j)];
      cff = 1.0 / (z_w[XW(i, j, k)] - z_w[XW(i, j, k - 1)]);
      const double cff_dn = cff * (hbbl[X2(i, j)] - z_w[XW(i, j, k - 1)]);
      const double cff_up = cff * (z_w[XW(i, j, k)] - hbbl[X2(i, j)]);
      double K_bl = cff_dn * Akv[XW(i, j, k)] + cff_up * Akv[XW(i, j, k - 1)];
      double dK_bl = -cff * (Akv[XW(i, j, k)] - Akv[XW(i, j, k - 1)]);
      Gm1[X2(i, j)] = K_bl / (zbl * wm[X2(i, j)] + eps);
      if (msk) Gm1[X2(i, j)] = Gm1[X2(i, j)] * o->rmask[X2(i, j)];
      dGm1dS[X2(i, j)] = MIN(0.0, K_bl * f1[X2(i, j)] - dK_bl / (wm[X2(i, j)] + eps));
      K_bl = cff_dn * Akt[XW4(i, j, k, 1)] + cff_up * Akt[XW4(i, j, k - 1, 1)];
      dK_bl = -cff * (Akt[XW4(i, j, k, 1)] - Akt[XW4(i, j, k - 1, 1)]);
      Gt1[X2(i, j)] = K_bl / (zbl * ws[X2(i, j)] + eps);
      if (msk) Gt1[X2(i, j)] = Gt1[X2(i, j)] * o->rmask[X2(i, j)];
      dGt1dS[X2(i, j)] = MIN(0.0, K_bl * f1[X2(i, j)] - dK_bl / (ws[X2(i, j)] + eps));
      K_bl = cff_dn * Akt[XW4(i, j, k, 2)] + cff_up * Akt[XW4(i, j, k - 1, 2)];
      dK_bl = -cff * (Akt[XW4(i, j, k, 2)] - Akt[XW4(i, j, k - 1, 2)]);
      Gs1[X2(i, j)] = K_bl / (zbl * ws[X2(i, j)] + eps);
      if (msk) Gs1[X2(i, j)] = Gs1[X2(i, j)] * o->rmask[X2(i, j)];
      dGs1dS[X2(i, j)] = MIN(0.0, K_bl * f1[X2(i, j)] - dK_bl / (ws[X2(i, j)] + eps));
    }
  for (int k = 1; k <= N - 1; k++)                                                                              /* :728-800 */
    for (int j = Jstr; j <= Jend; j++)
      for (int i = Istr; i <= Iend; i++)
        if (z_w[XW(i, j, k)] < hbbl[X2(i, j)]) {
          const double depth = z_w[XW(i, j, k)] - z_w[XW(i, j, 0)];
          double sigma;
          if (Bflux[XW(i, j, k)] < 0.0) sigma = MIN(bl_dpth[X2(i, j)], depth);
          else sigma = depth;
          const double Us = Ustar[X2(i, j)];
          const double Ustar3 = Us * Us * Us;
          const double zetahat = vonKar * sigma * Bflux[XW(i, j, k)];
          wscale(Us, zetahat, Ustar3, &wm[X2(i, j)], &ws[X2(i, j)]);
          sigma = depth / (hbbl[X2(i, j)] - z_w[XW(i, j, 0)] + eps);
          if (msk) sigma = sigma * o->rmask[X2(i, j)];
          const double a1 = sigma - 2.0, a2 = 3.0 - 2.0 * sigma, a3 = sigma - 1.0;
          const double Gm = a1 + a2 * Gm1[X2(i, j)] + a3 * dGm1dS[X2(i, j)];
          const double Gt = a1 + a2 * Gt1[X2(i, j)] + a3 * dGt1dS[X2(i, j)];
          const double Gs = a1 + a2 * Gs1[X2(i, j)] + a3 * dGs1dS[X2(i, j)];
          if (k > ksbl[X2(i, j)]) {
            Akv[XW(i, j, k)] = MAX(Akv[XW(i, j, k)], depth * wm[X2(i, j)] * (1.0 + sigma * Gm));
            Akt[XW4(i, j, k, 1)] = MAX(Akt[XW4(i, j, k, 1)], depth * ws[X2(i, j)] * (1.0 + sigma * Gt));
            Akt[XW4(i, j, k, 2)] = MAX(Akt[XW4(i, j, k, 2)], depth * ws[X2(i, j)] * (1.0 + sigma * Gs));
          } else {
            Akv[XW(i, j, k)] = depth * wm[X2(i, j)] * (1.0 + sigma * Gm);
            Akt[XW4(i, j, k, 1)] = depth * ws[X2(i, j)] * (1.0 + sigma * Gt);
            Akt[XW4(i, j, k, 2)] = depth * ws[X2(i, j)] * (1.0 + sigma * Gs);
          }
        }
  free(FC); free(Bflux); free(S); free(Rref);
}

static void lmd_finish(orc_t *o, const orc_bounds *b) {
  ORC_LOCALS(o);
  const int Istr = b->Istr, Iend = b->Iend, Jstr = b->Jstr, Jend = b->Jend;
  double *bvf = o->bvf, *Akv = o->Akv, *Akt = o->Akt;
  for (int k = 1; k <= N - 1; k++)
    for (int j = Jstr; j <= Jend; j++)
      for (int i = Istr; i <= Iend; i++) {
        double cff = MAX(bvf[XW(i, j, k)], lmd_bvfcon);
        cff = MIN(1.0, (lmd_bvfcon - cff) / lmd_bvfcon);
        double nu_sxc = 1.0 - cff * cff;
        nu_sxc = nu_sxc * nu_sxc * nu_sxc;
        Akv[XW(i, j, k)] = Akv[XW(i, j, k)] + lmd_nu0c * nu_sxc;
        Akt[XW4(i, j, k, 1)] = Akt[XW4(i, j, k, 1)] + lmd_nu0c * nu_sxc;
        Akt[XW4(i, j, k, 2)] = Akt[XW4(i, j, k, 2)] + lmd_nu0c * nu_sxc;
      }
  /* edge replication irrespective of periodicity (lmd_vmix.F:560-700), then bc_w3d_tile */
  for (int k = 0; k <= N; k++)
    for (int f = 0; f < 3; f++) {
      double *A = f == 0 ? Akt + XW4(LBi, LBj, k, 1) : (f == 1 ? Akt + XW4(LBi, LBj, k, 2) : Akv + XW(LBi, LBj, k));
      if (b->west) for (int j = Jstr; j <= Jend; j++) A[X2(Istr - 1, j)] = A[X2(Istr, j)];
      if (b->east) for (int j = Jstr; j <= Jend; j++) A[X2(Iend + 1, j)] = A[X2(Iend, j)];
      if (b->south) for (int i = Istr; i <= Iend; i++) A[X2(i, Jstr - 1)] = A[X2(i, Jstr)];
      if (b->north) for (int i = Istr; i <= Iend; i++) A[X2(i, Jend + 1)] = A[X2(i, Jend)];
      if (b->sw) A[X2(Istr - 1, Jstr - 1)] = 0.5 * (A[X2(Istr, Jstr - 1)] + A[X2(Istr - 1, Jstr)]);
      if (b->se) A[X2(Iend + 1, Jstr - 1)] = 0.5 * (A[X2(Iend, Jstr - 1)] + A[X2(Iend + 1, Jstr)]);
      if (b->nw) A[X2(Istr - 1, Jend + 1)] = 0.5 * (A[X2(Istr, Jend + 1)] + A[X2(Istr - 1, Jend)]);
      if (b->ne) A[X2(Iend + 1, Jend + 1)] = 0.5 * (A[X2(Iend, Jend + 1)] + A[X2(Iend + 1, Jend)]);
    }
  orc_bc_w3d(o, b, Akv, N + 1);
  for (int it = 0; it < o->c.NAT; it++) orc_bc_w3d(o, b, Akt + (size_t)it * nij * (N + 1), N + 1);
}

void orc_lmd_vmix(orc_t *o, int tile) {
  const orc_bounds *b = &o->b[tile];
  lmd_interior(o, b);
  lmd_skpp(o, b);
  if (o->bkpp) lmd_bkpp(o, b);                      /* LMD_BKPP: lmd_vmix.F:86-88 */
  lmd_finish(o, b);
}
