/*
 * orc_rhs3d.c -- baroclinic right-hand-side: pre_step3d, prsgrd32, t3dmix2,
 * rhs3d_tile, uv3dmix2 (the sequence of rhs3d, ROMS/Nonlinear/rhs3d.F:25-193).
 * TEST INFRASTRUCTURE (see orc.h).
 *
 *   orc_pre_step3d  pre_step3d_tile  ROMS/Nonlinear/pre_step3d.F:126-1180  pinned (round 2)
 *   orc_prsgrd      prsgrd32_tile    ROMS/Nonlinear/prsgrd32.h:109-436     pinned
 *   orc_t3dmix2     t3dmix2_s_tile   ROMS/Nonlinear/t3dmix2_s.h:89         pinned
 *                   t3dmix2_geo_tile ROMS/Nonlinear/t3dmix2_geo.h:90 (orc_t3dmix_geo.c) pinned
 *   orc_uv3dmix2    uv3dmix2_s_tile  ROMS/Nonlinear/uv3dmix2_s.h:114       pinned
 *   orc_rhs3d_tile  rhs3d_tile       ROMS/Nonlinear/rhs3d.F:196-1921       pinned (round 2)
 * (pre_step3d.F and rhs3d.F USE mod_sources -> mod_netcdf: not buildable here.)
 */
#include "orc.h"
#include <math.h>
#include <stdlib.h>

#define MAX(a, b) ((a) > (b) ? (a) : (b))
#define MIN(a, b) ((a) < (b) ? (a) : (b))

void orc_t3dmix2_geo(orc_t *o, int tile);                       /* orc_t3dmix_geo.c */
void orc_t3dmix2_iso(orc_t *o, int tile);
void orc_lmd_swfrac(const orc_t *o, const orc_bounds *b, double Zscale, const double *Z,
                    double *swdk);                               /* orc_lmd.c */

/* column scratch (IminS:ImaxS,0:N) */
#define CX(A, i, k) A[(size_t)((i) - LBi) + (size_t)(k) * ni]

/* horizontal advective tracer fluxes FX,FE of field T (plane k) for the
   non-HSIMT schemes; shared by pre_step3d (T=t(nstp)) and step3d_t (T=t(3)).
   pre_step3d.F HADV_FLUX :357-534 == step3d_t.F HADV_FLUX :432-768 (same code). */
void orc_hadv_flux(const orc_t *o, const orc_bounds *b, int scheme, const double *T /*plane*/,
                   const double *Huon /*plane*/, const double *Hvom /*plane*/, double *FX, double *FE,
                   double *curv, double *grad) {
  ORC_LOCALS(o);
  const int Istr = b->Istr, Iend = b->Iend, Jstr = b->Jstr, Jend = b->Jend;
  const double eps = 1.0E-16;
  double cff, cff1, cff2;
  if (scheme == ORC_C2) {
    for (int j = Jstr; j <= Jend; j++)
      for (int i = Istr; i <= Iend + 1; i++)
        FX[X2(i, j)] = Huon[X2(i, j)] * 0.5 * (T[X2(i - 1, j)] + T[X2(i, j)]);
    for (int j = Jstr; j <= Jend + 1; j++)
      for (int i = Istr; i <= Iend; i++)
        FE[X2(i, j)] = Hvom[X2(i, j)] * 0.5 * (T[X2(i, j - 1)] + T[X2(i, j)]);
  } else if (scheme == ORC_MPDATA || scheme == ORC_HSIMT) {
    /* first-order upstream (predictor only) */
    for (int j = Jstr; j <= Jend; j++)
      for (int i = Istr; i <= Iend + 1; i++) {
        cff1 = MAX(Huon[X2(i, j)], 0.0);
        cff2 = MIN(Huon[X2(i, j)], 0.0);
        FX[X2(i, j)] = cff1 * T[X2(i - 1, j)] + cff2 * T[X2(i, j)];
      }
    for (int j = Jstr; j <= Jend + 1; j++)
      for (int i = Istr; i <= Iend; i++) {
        cff1 = MAX(Hvom[X2(i, j)], 0.0);
        cff2 = MIN(Hvom[X2(i, j)], 0.0);
        FE[X2(i, j)] = cff1 * T[X2(i, j - 1)] + cff2 * T[X2(i, j)];
      }
  } else {
    /* AKIMA4, CENTERED4, SPLIT_U3, UPSTREAM3 */
    for (int j = Jstr; j <= Jend; j++)
      for (int i = b->Istrm1; i <= b->Iendp2; i++) {
        FX[X2(i, j)] = T[X2(i, j)] - T[X2(i - 1, j)];
        if (o->c.options & ORC_MASKING) FX[X2(i, j)] = FX[X2(i, j)] * o->umask[X2(i, j)];   /* pre_step3d.F:411, step3d_t.F:646 */
      }
    if (!o->c.EWperiodic) {
      if (b->west) for (int j = Jstr; j <= Jend; j++) FX[X2(Istr - 1, j)] = FX[X2(Istr, j)];
      if (b->east) for (int j = Jstr; j <= Jend; j++) FX[X2(Iend + 2, j)] = FX[X2(Iend + 1, j)];
    }
    for (int j = Jstr; j <= Jend; j++)
      for (int i = Istr - 1; i <= Iend + 1; i++) {
        if (scheme == ORC_U3) {
          curv[X2(i, j)] = FX[X2(i + 1, j)] - FX[X2(i, j)];
        } else if (scheme == ORC_A4) {
          cff = 2.0 * FX[X2(i + 1, j)] * FX[X2(i, j)];
          if (cff > eps) grad[X2(i, j)] = cff / (FX[X2(i + 1, j)] + FX[X2(i, j)]);
          else grad[X2(i, j)] = 0.0;
        } else {
          grad[X2(i, j)] = 0.5 * (FX[X2(i + 1, j)] + FX[X2(i, j)]);
        }
      }
    cff1 = 1.0 / 6.0;
    cff2 = 1.0 / 3.0;
    for (int j = Jstr; j <= Jend; j++)
      for (int i = Istr; i <= Iend + 1; i++) {
        if (scheme == ORC_U3)
          FX[X2(i, j)] = Huon[X2(i, j)] * 0.5 * (T[X2(i - 1, j)] + T[X2(i, j)]) -
                         cff1 * (curv[X2(i - 1, j)] * MAX(Huon[X2(i, j)], 0.0) +
                                 curv[X2(i, j)] * MIN(Huon[X2(i, j)], 0.0));
        else
          FX[X2(i, j)] = Huon[X2(i, j)] * 0.5 *
                         (T[X2(i - 1, j)] + T[X2(i, j)] - cff2 * (grad[X2(i, j)] - grad[X2(i - 1, j)]));
      }
    for (int j = b->Jstrm1; j <= b->Jendp2; j++)
      for (int i = Istr; i <= Iend; i++) {
        FE[X2(i, j)] = T[X2(i, j)] - T[X2(i, j - 1)];
        if (o->c.options & ORC_MASKING) FE[X2(i, j)] = FE[X2(i, j)] * o->vmask[X2(i, j)];   /* pre_step3d.F:476, step3d_t.F:710 */
      }
    if (!o->c.NSperiodic) {
      if (b->south) for (int i = Istr; i <= Iend; i++) FE[X2(i, Jstr - 1)] = FE[X2(i, Jstr)];
      if (b->north) for (int i = Istr; i <= Iend; i++) FE[X2(i, Jend + 2)] = FE[X2(i, Jend + 1)];
    }
    for (int j = Jstr - 1; j <= Jend + 1; j++)
      for (int i = Istr; i <= Iend; i++) {
        if (scheme == ORC_U3) {
          curv[X2(i, j)] = FE[X2(i, j + 1)] - FE[X2(i, j)];
        } else if (scheme == ORC_A4) {
          cff = 2.0 * FE[X2(i, j + 1)] * FE[X2(i, j)];
          if (cff > eps) grad[X2(i, j)] = cff / (FE[X2(i, j + 1)] + FE[X2(i, j)]);
          else grad[X2(i, j)] = 0.0;
        } else {
          grad[X2(i, j)] = 0.5 * (FE[X2(i, j + 1)] + FE[X2(i, j)]);
        }
      }
    for (int j = Jstr; j <= Jend + 1; j++)
      for (int i = Istr; i <= Iend; i++) {
        if (scheme == ORC_U3)
          FE[X2(i, j)] = Hvom[X2(i, j)] * 0.5 * (T[X2(i, j - 1)] + T[X2(i, j)]) -
                         cff1 * (curv[X2(i, j - 1)] * MAX(Hvom[X2(i, j)], 0.0) +
                                 curv[X2(i, j)] * MIN(Hvom[X2(i, j)], 0.0));
        else
          FE[X2(i, j)] = Hvom[X2(i, j)] * 0.5 *
                         (T[X2(i, j - 1)] + T[X2(i, j)] - cff2 * (grad[X2(i, j)] - grad[X2(i, j - 1)]));
      }
  }
}

/* vertical advective flux FC(i,0:N) on row j for the schemes shared between
   pre_step3d (VADV_FLUX :634-809, SPLINES with 1.5/0.5/3/2 end conditions) and
   step3d_t (VADV_FLUX :936-1186, SPLINES with 2/1/2/1); HSIMT/MPDATA of
   step3d_t are handled by the caller.  T = pointer to level 1 of the tracer. */
void orc_vadv_flux(const orc_t *o, const orc_bounds *b, int scheme, int corrector, int j,
                   const double *T, double *FC, double *CF) {
  ORC_LOCALS(o);
  const int Istr = b->Istr, Iend = b->Iend;
  const double *W = o->W, *Hz = o->Hz;
  const double eps = 1.0E-16;
  double cff, cff1, cff2, cff3;
  if (scheme == ORC_SPLINES) {
    const double a0 = corrector ? 2.0 : 1.5, c1 = corrector ? 1.0 : 0.5;
    const double aN = corrector ? 2.0 : 3.0, dN = corrector ? 1.0 : 2.0;
    for (int i = Istr; i <= Iend; i++) {
      CX(FC, i, 0) = a0 * T[X3(i, j, 1)];
      CX(CF, i, 1) = c1;
    }
    for (int k = 1; k <= N - 1; k++)
      for (int i = Istr; i <= Iend; i++) {
        cff = 1.0 / (2.0 * Hz[X3(i, j, k)] + Hz[X3(i, j, k + 1)] * (2.0 - CX(CF, i, k)));
        CX(CF, i, k + 1) = cff * Hz[X3(i, j, k)];
        CX(FC, i, k) = cff * (3.0 * (Hz[X3(i, j, k)] * T[X3(i, j, k + 1)] + Hz[X3(i, j, k + 1)] * T[X3(i, j, k)]) -
                              Hz[X3(i, j, k + 1)] * CX(FC, i, k - 1));
      }
    for (int i = Istr; i <= Iend; i++)
      CX(FC, i, N) = (aN * T[X3(i, j, N)] - CX(FC, i, N - 1)) / (dN - CX(CF, i, N));
    for (int k = N - 1; k >= 0; k--)
      for (int i = Istr; i <= Iend; i++) {
        CX(FC, i, k) = CX(FC, i, k) - CX(CF, i, k + 1) * CX(FC, i, k + 1);
        CX(FC, i, k + 1) = W[XW(i, j, k + 1)] * CX(FC, i, k + 1);
      }
    for (int i = Istr; i <= Iend; i++) { CX(FC, i, N) = 0.0; CX(FC, i, 0) = 0.0; }
  } else if (scheme == ORC_A4) {
    for (int k = 1; k <= N - 1; k++)
      for (int i = Istr; i <= Iend; i++) CX(FC, i, k) = T[X3(i, j, k + 1)] - T[X3(i, j, k)];
    for (int i = Istr; i <= Iend; i++) { CX(FC, i, 0) = CX(FC, i, 1); CX(FC, i, N) = CX(FC, i, N - 1); }
    for (int k = 1; k <= N; k++)
      for (int i = Istr; i <= Iend; i++) {
        cff = 2.0 * CX(FC, i, k) * CX(FC, i, k - 1);
        if (cff > eps) CX(CF, i, k) = cff / (CX(FC, i, k) + CX(FC, i, k - 1));
        else CX(CF, i, k) = 0.0;
      }
    cff1 = 1.0 / 3.0;
    for (int k = 1; k <= N - 1; k++)
      for (int i = Istr; i <= Iend; i++)
        CX(FC, i, k) = W[XW(i, j, k)] * 0.5 *
                       (T[X3(i, j, k)] + T[X3(i, j, k + 1)] - cff1 * (CX(CF, i, k + 1) - CX(CF, i, k)));
    for (int i = Istr; i <= Iend; i++) { CX(FC, i, 0) = 0.0; CX(FC, i, N) = 0.0; }
  } else if (scheme == ORC_C2) {
    for (int k = 1; k <= N - 1; k++)
      for (int i = Istr; i <= Iend; i++)
        CX(FC, i, k) = W[XW(i, j, k)] * 0.5 * (T[X3(i, j, k)] + T[X3(i, j, k + 1)]);
    for (int i = Istr; i <= Iend; i++) { CX(FC, i, 0) = 0.0; CX(FC, i, N) = 0.0; }
  } else if (scheme == ORC_MPDATA || scheme == ORC_HSIMT) {
    /* first-order upstream (pre_step3d only) */
    for (int k = 1; k <= N - 1; k++)
      for (int i = Istr; i <= Iend; i++) {
        cff1 = MAX(W[XW(i, j, k)], 0.0);
        cff2 = MIN(W[XW(i, j, k)], 0.0);
        CX(FC, i, k) = cff1 * T[X3(i, j, k)] + cff2 * T[X3(i, j, k + 1)];
      }
    for (int i = Istr; i <= Iend; i++) { CX(FC, i, 0) = 0.0; CX(FC, i, N) = 0.0; }
  } else { /* CENTERED4, SPLIT_U3 */
    cff1 = 0.5;
    cff2 = 7.0 / 12.0;
    cff3 = 1.0 / 12.0;
    for (int k = 2; k <= N - 2; k++)
      for (int i = Istr; i <= Iend; i++)
        CX(FC, i, k) = W[XW(i, j, k)] * (cff2 * (T[X3(i, j, k)] + T[X3(i, j, k + 1)]) -
                                         cff3 * (T[X3(i, j, k - 1)] + T[X3(i, j, k + 2)]));
    for (int i = Istr; i <= Iend; i++) {
      CX(FC, i, 0) = 0.0;
      CX(FC, i, 1) = W[XW(i, j, 1)] * (cff1 * T[X3(i, j, 1)] + cff2 * T[X3(i, j, 2)] - cff3 * T[X3(i, j, 3)]);
      CX(FC, i, N - 1) = W[XW(i, j, N - 1)] * (cff1 * T[X3(i, j, N)] + cff2 * T[X3(i, j, N - 1)] -
                                               cff3 * T[X3(i, j, N - 2)]);
      CX(FC, i, N) = 0.0;
    }
  }
}

/* ------------------------------------------------------------ pre_step3d */
void orc_pre_step3d(orc_t *o, int tile) {
  ORC_LOCALS(o);
  const orc_bounds *b = &o->b[tile];
  const orc_cfg *c = &o->c;
  const int nrhs = o->s.nrhs, nstp = o->s.nstp, nnew = o->s.nnew, iic = o->s.iic;
  const int Istr = b->Istr, Iend = b->Iend, Jstr = b->Jstr, Jend = b->Jend;
  const int IstrU = b->IstrU, JstrV = b->JstrV;
  const double dt = c->dt, lambda = c->lambda;
  double *t = o->t, *u = o->u, *v = o->v, *Hz = o->Hz, *Huon = o->Huon, *Hvom = o->Hvom;
  double *z_r = o->z_r, *W = o->W, *pm = o->pm, *pn = o->pn, *Akt = o->Akt, *Akv = o->Akv;
  double *ru = o->ru, *rv = o->rv;
  double cff, cff1, cff2, cff3, cff4, Gamma;
  double *FX = (double *)calloc(4 * nij, sizeof(double));
  double *FE = FX + nij, *curv = FX + 2 * nij, *grad = FX + 3 * nij;
  double *CF = (double *)calloc(3 * ni * (size_t)(N + 1), sizeof(double));
  double *DC = CF + ni * (size_t)(N + 1), *FC = CF + 2 * ni * (size_t)(N + 1);
  double *swdk = NULL;

  if (c->options & ORC_SOLAR_SOURCE) {
    /* fraction of solar shortwave flux penetrating to W-levels :316-345 */
    swdk = (double *)calloc(nij * (size_t)(N + 1), sizeof(double));
    for (int k = 1; k <= N - 1; k++) {
      for (int j = Jstr; j <= Jend; j++)
        for (int i = Istr; i <= Iend; i++) FX[X2(i, j)] = o->z_w[XW(i, j, N)] - o->z_w[XW(i, j, k)];
      orc_lmd_swfrac(o, b, -1.0, FX, FE);
      for (int j = Jstr; j <= Jend; j++)
        for (int i = Istr; i <= Iend; i++) swdk[XW(i, j, k)] = FE[X2(i, j)];
    }
  }

  /* predictor tracer at n+1/2: horizontal part :357-625 */
  for (int itrc = 1; itrc <= c->NT; itrc++) {
    const int hs = c->hadv[itrc - 1];
    for (int k = 1; k <= N; k++) {
      orc_hadv_flux(o, b, hs, t + XT(LBi, LBj, k, nstp, itrc), Huon + X3(LBi, LBj, k),
                    Hvom + X3(LBi, LBj, k), FX, FE, curv, grad);
      if (hs == ORC_MPDATA || hs == ORC_HSIMT) Gamma = 0.5;
      else Gamma = 1.0 / 6.0;
      if (iic == c->ntfirst) {
        cff = 0.5 * dt;
        cff1 = 1.0;
        cff2 = 0.0;
      } else {
        cff = (1.0 - Gamma) * dt;
        cff1 = 0.5 + Gamma;
        cff2 = 0.5 - Gamma;
      }
      for (int j = Jstr; j <= Jend; j++)
        for (int i = Istr; i <= Iend; i++)
          t[XT(i, j, k, 3, itrc)] =
              Hz[X3(i, j, k)] * (cff1 * t[XT(i, j, k, nstp, itrc)] + cff2 * t[XT(i, j, k, nnew, itrc)]) -
              cff * pm[X2(i, j)] * pn[X2(i, j)] *
                  (FX[X2(i + 1, j)] - FX[X2(i, j)] + FE[X2(i, j + 1)] - FE[X2(i, j)]);
    }
  }

  /* vertical part :634-852 */
  for (int j = Jstr; j <= Jend; j++) {
    for (int itrc = 1; itrc <= c->NT; itrc++) {
      const int vs = c->vadv[itrc - 1];
      orc_vadv_flux(o, b, vs, 0, j, t + XT(LBi, LBj, 1, nstp, itrc), FC, CF);
      if (vs == ORC_MPDATA || vs == ORC_HSIMT) Gamma = 0.5;
      else Gamma = 1.0 / 6.0;
      if (iic == c->ntfirst) cff = 0.5 * dt;
      else cff = (1.0 - Gamma) * dt;
      for (int k = 1; k <= N; k++)
        for (int i = Istr; i <= Iend; i++)
          CX(DC, i, k) = 1.0 / (Hz[X3(i, j, k)] -
                                cff * pm[X2(i, j)] * pn[X2(i, j)] *
                                    (Huon[X3(i + 1, j, k)] - Huon[X3(i, j, k)] + Hvom[X3(i, j + 1, k)] -
                                     Hvom[X3(i, j, k)] + (W[XW(i, j, k)] - W[XW(i, j, k - 1)])));
      for (int k = 1; k <= N; k++)
        for (int i = Istr; i <= Iend; i++) {
          cff1 = cff * pm[X2(i, j)] * pn[X2(i, j)];
          t[XT(i, j, k, 3, itrc)] =
              CX(DC, i, k) * (t[XT(i, j, k, 3, itrc)] - cff1 * (CX(FC, i, k) - CX(FC, i, k - 1)));
        }
    }
  }

  /* start of t(nnew): explicit vertical diffusion, fluxes :855-935 */
  for (int j = Jstr; j <= Jend; j++) {
    cff3 = dt * (1.0 - lambda);
    for (int itrc = 1; itrc <= c->NT; itrc++) {
      const int ltrc = MIN(c->NAT, itrc);
      for (int k = 1; k <= N - 1; k++)
        for (int i = Istr; i <= Iend; i++) {
          cff = 1.0 / (z_r[X3(i, j, k + 1)] - z_r[X3(i, j, k)]);
          CX(FC, i, k) = cff3 * cff * Akt[XW4(i, j, k, ltrc)] *
                         (t[XT(i, j, k + 1, nstp, itrc)] - t[XT(i, j, k, nstp, itrc)]);
        }
      if ((c->options & ORC_LMD_MIXING) && itrc <= c->NAT) {
        /* LMD_NONLOCAL: non-local transport :878-886 */
        for (int k = 1; k <= N - 1; k++)
          for (int i = Istr; i <= Iend; i++)
            CX(FC, i, k) = CX(FC, i, k) - dt * Akt[XW4(i, j, k, itrc)] * o->ghats[XW4(i, j, k, itrc)];
      }
      if ((c->options & ORC_SOLAR_SOURCE) && itrc == 1) {
        /* SOLAR_SOURCE :890-900 */
        for (int k = 1; k <= N - 1; k++)
          for (int i = Istr; i <= Iend; i++)
            if (o->wet_dry) CX(FC, i, k) = CX(FC, i, k) + dt * o->srflx[X2(i, j)] * o->rmask_wet[X2(i, j)] * swdk[XW(i, j, k)];   /* pre_step3d.F:903 */
            else CX(FC, i, k) = CX(FC, i, k) + dt * o->srflx[X2(i, j)] * swdk[XW(i, j, k)];
      }
      for (int i = Istr; i <= Iend; i++) {
        CX(FC, i, 0) = dt * o->btflx[X2T(i, j, itrc)];
        CX(FC, i, N) = dt * o->stflx[X2T(i, j, itrc)];
      }
      for (int k = 1; k <= N; k++)
        for (int i = Istr; i <= Iend; i++) {
          cff1 = Hz[X3(i, j, k)] * t[XT(i, j, k, nstp, itrc)];
          cff2 = CX(FC, i, k) - CX(FC, i, k - 1);
          t[XT(i, j, k, nnew, itrc)] = cff1 + cff2;
          if (o->dia) {                                                 /* DIAGNOSTICS_TS pre_step3d.F:925-928 */
            orc_dia_wrk(o, ORC_DIA_RATE, itrc)[X3(i, j, k)] = cff1;
            orc_dia_wrk(o, ORC_DIA_VDIF, itrc)[X3(i, j, k)] = cff2;
          }
        }
    }
  }

  /* start of u,v(nnew) :943-1145 */
  for (int j = Jstr; j <= Jend; j++) {
    cff3 = dt * (1.0 - lambda);
    for (int k = 1; k <= N - 1; k++)
      for (int i = IstrU; i <= Iend; i++) {
        cff = 1.0 / (z_r[X3(i, j, k + 1)] + z_r[X3(i - 1, j, k + 1)] - z_r[X3(i, j, k)] - z_r[X3(i - 1, j, k)]);
        CX(FC, i, k) = cff3 * cff * (u[X4(i, j, k + 1, nstp)] - u[X4(i, j, k, nstp)]) *
                       (Akv[XW(i, j, k)] + Akv[XW(i - 1, j, k)]);
      }
    for (int i = IstrU; i <= Iend; i++) {
      CX(FC, i, 0) = dt * o->bustr[X2(i, j)];
      CX(FC, i, N) = dt * o->sustr[X2(i, j)];
    }
    cff = dt * 0.25;
    for (int i = IstrU; i <= Iend; i++)
      CX(DC, i, 0) = cff * (pm[X2(i, j)] + pm[X2(i - 1, j)]) * (pn[X2(i, j)] + pn[X2(i - 1, j)]);
    const int indx = 3 - nrhs;
    if (iic == c->ntfirst) {
      for (int k = 1; k <= N; k++)
        for (int i = IstrU; i <= Iend; i++) {
          cff1 = u[X4(i, j, k, nstp)] * 0.5 * (Hz[X3(i, j, k)] + Hz[X3(i - 1, j, k)]);
          cff2 = CX(FC, i, k) - CX(FC, i, k - 1);
          u[X4(i, j, k, nnew)] = cff1 + cff2;
          if (o->duv) {                                                          /* pre_step3d.F:979-984 */
            const orc_diauv *d = o->duv;
            for (int id = 1; id <= d->M3pgrd; id++) DU3(d->U3wrk, i, j, k, id) = 0.0;
            DU3(d->U3wrk, i, j, k, d->M3vvis) = cff2;
            DU3(d->U3wrk, i, j, k, d->M3rate) = cff1;
          }
        }
    } else if (iic == c->ntfirst + 1) {
      for (int k = 1; k <= N; k++)
        for (int i = IstrU; i <= Iend; i++) {
          cff1 = u[X4(i, j, k, nstp)] * 0.5 * (Hz[X3(i, j, k)] + Hz[X3(i - 1, j, k)]);
          cff2 = CX(FC, i, k) - CX(FC, i, k - 1);
          cff3 = 0.5 * CX(DC, i, 0);
          u[X4(i, j, k, nnew)] = cff1 - cff3 * ru[XW4(i, j, k, indx)] + cff2;
          if (o->duv) {                                                          /* :998-1007 */
            const orc_diauv *d = o->duv;
            for (int id = 1; id <= d->M3pgrd; id++) DU3(d->U3wrk, i, j, k, id) = -cff3 * DUR(d->RU, i, j, k, indx, id);
            DU3(d->U3wrk, i, j, k, d->M3vvis) = cff2;
            DU3(d->U3wrk, i, j, k, d->M3rate) = cff1;
          }
        }
    } else {
      cff1 = 5.0 / 12.0;
      cff2 = 16.0 / 12.0;
      for (int k = 1; k <= N; k++)
        for (int i = IstrU; i <= Iend; i++) {
          cff3 = u[X4(i, j, k, nstp)] * 0.5 * (Hz[X3(i, j, k)] + Hz[X3(i - 1, j, k)]);
          cff4 = CX(FC, i, k) - CX(FC, i, k - 1);
          u[X4(i, j, k, nnew)] =
              cff3 + CX(DC, i, 0) * (cff1 * ru[XW4(i, j, k, nrhs)] - cff2 * ru[XW4(i, j, k, indx)]) + cff4;
          if (o->duv) {                                                          /* :1022-1035 */
            const orc_diauv *d = o->duv;
            for (int id = 1; id <= d->M3pgrd; id++)
              DU3(d->U3wrk, i, j, k, id) = CX(DC, i, 0) * (cff1 * DUR(d->RU, i, j, k, nrhs, id) - cff2 * DUR(d->RU, i, j, k, indx, id));
            DU3(d->U3wrk, i, j, k, d->M3vvis) = cff4;
            DU3(d->U3wrk, i, j, k, d->M3rate) = cff3;
          }
        }
    }
    if (j >= JstrV) {
      cff3 = dt * (1.0 - lambda);
      for (int k = 1; k <= N - 1; k++)
        for (int i = Istr; i <= Iend; i++) {
          cff = 1.0 / (z_r[X3(i, j, k + 1)] + z_r[X3(i, j - 1, k + 1)] - z_r[X3(i, j, k)] - z_r[X3(i, j - 1, k)]);
          CX(FC, i, k) = cff3 * cff * (v[X4(i, j, k + 1, nstp)] - v[X4(i, j, k, nstp)]) *
                         (Akv[XW(i, j, k)] + Akv[XW(i, j - 1, k)]);
        }
      for (int i = Istr; i <= Iend; i++) {
        CX(FC, i, 0) = dt * o->bvstr[X2(i, j)];
        CX(FC, i, N) = dt * o->svstr[X2(i, j)];
      }
      cff = dt * 0.25;
      for (int i = Istr; i <= Iend; i++)
        CX(DC, i, 0) = cff * (pm[X2(i, j)] + pm[X2(i, j - 1)]) * (pn[X2(i, j)] + pn[X2(i, j - 1)]);
      if (iic == c->ntfirst) {
        for (int k = 1; k <= N; k++)
          for (int i = Istr; i <= Iend; i++) {
            cff1 = v[X4(i, j, k, nstp)] * 0.5 * (Hz[X3(i, j, k)] + Hz[X3(i, j - 1, k)]);
            cff2 = CX(FC, i, k) - CX(FC, i, k - 1);
            v[X4(i, j, k, nnew)] = cff1 + cff2;
            if (o->duv) {                                                          /* pre_step3d.F:979-984 */
              const orc_diauv *d = o->duv;
              for (int id = 1; id <= d->M3pgrd; id++) DU3(d->V3wrk, i, j, k, id) = 0.0;
              DU3(d->V3wrk, i, j, k, d->M3vvis) = cff2;
              DU3(d->V3wrk, i, j, k, d->M3rate) = cff1;
            }
          }
      } else if (iic == c->ntfirst + 1) {
        for (int k = 1; k <= N; k++)
          for (int i = Istr; i <= Iend; i++) {
            cff1 = v[X4(i, j, k, nstp)] * 0.5 * (Hz[X3(i, j, k)] + Hz[X3(i, j - 1, k)]);
            cff2 = CX(FC, i, k) - CX(FC, i, k - 1);
            cff3 = 0.5 * CX(DC, i, 0);
            v[X4(i, j, k, nnew)] = cff1 - cff3 * rv[XW4(i, j, k, indx)] + cff2;
            if (o->duv) {                                                          /* :998-1007 */
              const orc_diauv *d = o->duv;
              for (int id = 1; id <= d->M3pgrd; id++) DU3(d->V3wrk, i, j, k, id) = -cff3 * DUR(d->RV, i, j, k, indx, id);
              DU3(d->V3wrk, i, j, k, d->M3vvis) = cff2;
              DU3(d->V3wrk, i, j, k, d->M3rate) = cff1;
            }
          }
      } else {
        cff1 = 5.0 / 12.0;
        cff2 = 16.0 / 12.0;
        for (int k = 1; k <= N; k++)
          for (int i = Istr; i <= Iend; i++) {
            cff3 = v[X4(i, j, k, nstp)] * 0.5 * (Hz[X3(i, j, k)] + Hz[X3(i, j - 1, k)]);
            cff4 = CX(FC, i, k) - CX(FC, i, k - 1);
            v[X4(i, j, k, nnew)] =
                cff3 + CX(DC, i, 0) * (cff1 * rv[XW4(i, j, k, nrhs)] - cff2 * rv[XW4(i, j, k, indx)]) + cff4;
            if (o->duv) {                                                          /* :1022-1035 */
              const orc_diauv *d = o->duv;
              for (int id = 1; id <= d->M3pgrd; id++)
                DU3(d->V3wrk, i, j, k, id) = CX(DC, i, 0) * (cff1 * DUR(d->RV, i, j, k, nrhs, id) - cff2 * DUR(d->RV, i, j, k, indx, id));
              DU3(d->V3wrk, i, j, k, d->M3vvis) = cff4;
              DU3(d->V3wrk, i, j, k, d->M3rate) = cff3;
            }
          }
      }
    }
  }

  /* BCs and exchange of the predictor tracer :1157-1171 */
  for (int itrc = 1; itrc <= c->NT; itrc++) {
    orc_t3dbc(o, b, 3, itrc);
    orc_exchange3d(o, b, 'r', t + XT(LBi, LBj, 1, 3, itrc), N);
  }
  free(FX);
  free(CF);
  free(swdk);
}

/* ---------------------------------------------------------------- prsgrd32 */
/* prsgrd31_tile, prsgrd31.h:95-380: the standard density Jacobian (WJ_GRADP: weighted, Song 1998), RHO_SURF; the
   pressure-gradient scheme of an application that defines none of DJ_GRADPS, PJ_GRADP, PJ_GRADPQ2, PJ_GRADPQ4 */
static void orc_prsgrd31(orc_t *o, int tile) {
  ORC_LOCALS(o);
  const orc_bounds *b = &o->b[tile];
  const int nrhs = o->s.nrhs;
  const int Istr = b->Istr, Iend = b->Iend, Jstr = b->Jstr, Jend = b->Jend, IstrU = b->IstrU, JstrV = b->JstrV;
  const double g = o->c.g, rho0 = o->c.rho0;
  const double fac1 = 0.5 * g / rho0, fac2 = 1000.0 * g / rho0, fac3 = 0.25 * g / rho0;
  const int wj = (o->c.options & ORC_WJ_GRADP) != 0;
  double *rho = o->rho, *z_r = o->z_r, *z_w = o->z_w, *Hz = o->Hz, *ru = o->ru, *rv = o->rv;
  double *phi = (double *)malloc(sizeof(double) * o->ni);
  for (int j = Jstr; j <= Jend; j++)
    for (int dir = 0; dir < 2; dir++) {
      if (dir == 1 && j < JstrV) break;
      const int di = dir == 0 ? 1 : 0, dj = dir == 0 ? 0 : 1, i0 = dir == 0 ? IstrU : Istr;
      double *rq = dir == 0 ? ru : rv;
      const double *omn = dir == 0 ? o->on_u : o->om_v;
#define PH(i) phi[(i) - LBi]
#define Rm(k) rho[X3(i - di, j - dj, k)]
#define Rc(k) rho[X3(i, j, k)]
#define Zm(k) z_r[X3(i - di, j - dj, k)]
#define Zc(k) z_r[X3(i, j, k)]
      for (int i = i0; i <= Iend; i++) {
        double cff1 = z_w[XW(i, j, N)] - Zc(N) + z_w[XW(i - di, j - dj, N)] - Zm(N);
        PH(i) = fac1 * (Rc(N) - Rm(N)) * cff1;
        PH(i) = PH(i) + (fac2 + fac1 * (Rc(N) + Rm(N))) * (z_w[XW(i, j, N)] - z_w[XW(i - di, j - dj, N)]);    /* RHO_SURF */
        rq[XW4(i, j, N, nrhs)] = -0.5 * (Hz[X3(i, j, N)] + Hz[X3(i - di, j - dj, N)]) * PH(i) * omn[X2(i, j)];
      }
      for (int k = N - 1; k >= 1; k--)
        for (int i = i0; i <= Iend; i++) {
          double cff1, cff2, cff3, cff4;
          if (wj) {
            cff1 = 1.0 / ((Zc(k + 1) - Zc(k)) * (Zm(k + 1) - Zm(k)));
            cff2 = Zc(k) - Zm(k) + Zc(k + 1) - Zm(k + 1);
            cff3 = Zc(k + 1) - Zc(k) - Zm(k + 1) + Zm(k);
            const double gamma = 0.125 * cff1 * cff2 * cff3;
            cff1 = (1.0 + gamma) * (Rc(k + 1) - Rm(k + 1)) + (1.0 - gamma) * (Rc(k) - Rm(k));
            cff2 = Rc(k + 1) + Rm(k + 1) - Rc(k) - Rm(k);
            cff3 = Zc(k + 1) + Zm(k + 1) - Zc(k) - Zm(k);
            cff4 = (1.0 + gamma) * (Zc(k + 1) - Zm(k + 1)) + (1.0 - gamma) * (Zc(k) - Zm(k));
          } else {
            cff1 = Rc(k + 1) - Rm(k + 1) + Rc(k) - Rm(k);
            cff2 = Rc(k + 1) + Rm(k + 1) - Rc(k) - Rm(k);
            cff3 = Zc(k + 1) + Zm(k + 1) - Zc(k) - Zm(k);
            cff4 = Zc(k + 1) - Zm(k + 1) + Zc(k) - Zm(k);
          }
          PH(i) = PH(i) + fac3 * (cff1 * cff3 - cff2 * cff4);
          rq[XW4(i, j, k, nrhs)] = -0.5 * (Hz[X3(i, j, k)] + Hz[X3(i - di, j - dj, k)]) * PH(i) * omn[X2(i, j)];
        }
#undef PH
#undef Rm
#undef Rc
#undef Zm
#undef Zc
    }
  free(phi);
}

/* prsgrd40_tile, prsgrd40.h:186-290: finite-volume pressure gradient (Lin 1997), PJ_GRADP */
static void orc_prsgrd40(orc_t *o, int tile) {
  ORC_LOCALS(o);
  const orc_bounds *b = &o->b[tile];
  const int nrhs = o->s.nrhs;
  const int Istr = b->Istr, Iend = b->Iend, Jstr = b->Jstr, Jend = b->Jend, IstrU = b->IstrU, JstrV = b->JstrV;
  const double g = o->c.g, rho0 = o->c.rho0;
  double *rho = o->rho, *z_w = o->z_w, *Hz = o->Hz, *ru = o->ru, *rv = o->rv;
  double *P = (double *)calloc(nij * (size_t)(N + 1), sizeof(double)), *FX = (double *)calloc(nij * (size_t)(N + 1), sizeof(double));
  double *FC = (double *)calloc(ni * (size_t)(N + 1), sizeof(double));
  for (int j = JstrV - 1; j <= Jend; j++) {
    for (int i = IstrU - 1; i <= Iend; i++) P[XW(i, j, N)] = 0.0;
    for (int k = N; k >= 1; k--)
      for (int i = IstrU - 1; i <= Iend; i++) {
        P[XW(i, j, k - 1)] = P[XW(i, j, k)] + Hz[X3(i, j, k)] * rho[X3(i, j, k)];
        FX[XW(i, j, k)] = 0.5 * Hz[X3(i, j, k)] * (P[XW(i, j, k)] + P[XW(i, j, k - 1)]);
      }
    if (j >= Jstr) {
      for (int i = IstrU; i <= Iend; i++) CX(FC, i, N) = 0.0;
      const double cff = 0.5 * g, cff1 = g / rho0;
      for (int k = N; k >= 1; k--)
        for (int i = IstrU; i <= Iend; i++) {
          const double dh = z_w[XW(i, j, k - 1)] - z_w[XW(i - 1, j, k - 1)];
          CX(FC, i, k - 1) = 0.5 * dh * (P[XW(i, j, k - 1)] + P[XW(i - 1, j, k - 1)]);
          ru[XW4(i, j, k, nrhs)] = (cff * (Hz[X3(i - 1, j, k)] + Hz[X3(i, j, k)]) * (z_w[XW(i - 1, j, N)] - z_w[XW(i, j, N)]) +
                                    cff1 * (FX[XW(i - 1, j, k)] - FX[XW(i, j, k)] + CX(FC, i, k) - CX(FC, i, k - 1))) * o->on_u[X2(i, j)];
        }
    }
    if (j >= JstrV) {
      for (int i = Istr; i <= Iend; i++) CX(FC, i, N) = 0.0;
      const double cff = 0.5 * g, cff1 = g / rho0;
      for (int k = N; k >= 1; k--)
        for (int i = Istr; i <= Iend; i++) {
          const double dh = z_w[XW(i, j, k - 1)] - z_w[XW(i, j - 1, k - 1)];
          CX(FC, i, k - 1) = 0.5 * dh * (P[XW(i, j, k - 1)] + P[XW(i, j - 1, k - 1)]);
          rv[XW4(i, j, k, nrhs)] = (cff * (Hz[X3(i, j - 1, k)] + Hz[X3(i, j, k)]) * (z_w[XW(i, j - 1, N)] - z_w[XW(i, j, N)]) +
                                    cff1 * (FX[XW(i, j - 1, k)] - FX[XW(i, j, k)] + CX(FC, i, k) - CX(FC, i, k - 1))) * o->om_v[X2(i, j)];
        }
    }
  }
  free(P); free(FX); free(FC);
}

static void orc_prsgrd32(orc_t *o, int tile);
void orc_prsgrd(orc_t *o, int tile) {
  if (o->prs_scheme == 44) orc_prsgrd44(o, tile);              /* prsgrd.F:16-19: PJ_GRADPQ4, then PJ_GRADPQ2, come first */
  else if (o->prs_scheme == 42) orc_prsgrd42(o, tile);
  else if (o->c.options & ORC_PRSGRD40) orc_prsgrd40(o, tile);
  else if (o->c.options & ORC_PRSGRD31) orc_prsgrd31(o, tile);
  else orc_prsgrd32(o, tile);
  if (o->wet_dry && (o->prs_scheme == 44 || (o->c.options & (ORC_PRSGRD40 | ORC_PRSGRD31)))) {
    /* WET_DRY: ru, rv times the wet masks where these schemes assign them (prsgrd31.h:239,285,323,369; prsgrd40.h:251,281;
       prsgrd44.h:466,500) -- prsgrd32 does it inside (:362, :426); prsgrd42 masks its FIRST pass (not carried: refused) */
    ORC_LOCALS(o);
    const orc_bounds *b = &o->b[tile];
    const int nrhs = o->s.nrhs;
    for (int k = 1; k <= N; k++) {
      for (int j = b->Jstr; j <= b->Jend; j++)
        for (int i = b->IstrU; i <= b->Iend; i++) o->ru[XW4(i, j, k, nrhs)] = o->ru[XW4(i, j, k, nrhs)] * o->umask_wet[X2(i, j)];
      for (int j = b->JstrV; j <= b->Jend; j++)
        for (int i = b->Istr; i <= b->Iend; i++) o->rv[XW4(i, j, k, nrhs)] = o->rv[XW4(i, j, k, nrhs)] * o->vmask_wet[X2(i, j)];
    }
  }
  if (o->duv) {        /* DIAGNOSTICS_UV: DiaRU(i,j,k,nrhs,M3pgrd) = ru(i,j,k,nrhs) where every scheme assigns it (prsgrd32.h:364, :428) */
    ORC_LOCALS(o);
    const orc_bounds *b = &o->b[tile];
    const orc_diauv *d = o->duv;
    const int nrhs = o->s.nrhs;
    for (int k = 1; k <= N; k++) {
      for (int j = b->Jstr; j <= b->Jend; j++)
        for (int i = b->IstrU; i <= b->Iend; i++) DUR(d->RU, i, j, k, nrhs, d->M3pgrd) = o->ru[XW4(i, j, k, nrhs)];
      for (int j = b->JstrV; j <= b->Jend; j++)
        for (int i = b->Istr; i <= b->Iend; i++) DUR(d->RV, i, j, k, nrhs, d->M3pgrd) = o->rv[XW4(i, j, k, nrhs)];
    }
  }
}
static void orc_prsgrd32(orc_t *o, int tile) {
  ORC_LOCALS(o);
  const orc_bounds *b = &o->b[tile];
  const int nrhs = o->s.nrhs;
  const int Istr = b->Istr, Iend = b->Iend, Jstr = b->Jstr, Jend = b->Jend;
  const int IstrU = b->IstrU, JstrV = b->JstrV;
  const double OneFifth = 0.2, OneTwelfth = 1.0 / 12.0, eps = 1.0E-10;
  const double g = o->c.g;
  const double GRho = g / o->c.rho0;
  const double HalfGRho = 0.5 * GRho;
  double *rho = o->rho, *z_r = o->z_r, *z_w = o->z_w, *Hz = o->Hz, *ru = o->ru, *rv = o->rv;
  double cff, cff1, cff2;
  double *P = (double *)calloc(nij * (size_t)N, sizeof(double));
  double *dR = (double *)calloc(2 * ni * (size_t)(N + 1), sizeof(double));
  double *dZ = dR + ni * (size_t)(N + 1);
  double *FC = (double *)calloc(4 * nij, sizeof(double));
  double *aux = FC + nij, *dRx = FC + 2 * nij, *dZx = FC + 3 * nij;

  for (int j = JstrV - 1; j <= Jend; j++) {
    for (int k = 1; k <= N - 1; k++)
      for (int i = IstrU - 1; i <= Iend; i++) {
        CX(dR, i, k) = rho[X3(i, j, k + 1)] - rho[X3(i, j, k)];
        CX(dZ, i, k) = z_r[X3(i, j, k + 1)] - z_r[X3(i, j, k)];
      }
    for (int i = IstrU - 1; i <= Iend; i++) {
      CX(dR, i, N) = CX(dR, i, N - 1);
      CX(dZ, i, N) = CX(dZ, i, N - 1);
      CX(dR, i, 0) = CX(dR, i, 1);
      CX(dZ, i, 0) = CX(dZ, i, 1);
    }
    for (int k = N; k >= 1; k--)
      for (int i = IstrU - 1; i <= Iend; i++) {
        cff = 2.0 * CX(dR, i, k) * CX(dR, i, k - 1);
        if (cff > eps) CX(dR, i, k) = cff / (CX(dR, i, k) + CX(dR, i, k - 1));
        else CX(dR, i, k) = 0.0;
        CX(dZ, i, k) = 2.0 * CX(dZ, i, k) * CX(dZ, i, k - 1) / (CX(dZ, i, k) + CX(dZ, i, k - 1));
      }
    for (int i = IstrU - 1; i <= Iend; i++) {
      cff1 = 1.0 / (z_r[X3(i, j, N)] - z_r[X3(i, j, N - 1)]);
      cff2 = 0.5 * (rho[X3(i, j, N)] - rho[X3(i, j, N - 1)]) * (z_w[XW(i, j, N)] - z_r[X3(i, j, N)]) * cff1;
      P[X3(i, j, N)] = g * z_w[XW(i, j, N)] +
                       GRho * (rho[X3(i, j, N)] + cff2) * (z_w[XW(i, j, N)] - z_r[X3(i, j, N)]);
    }
    for (int k = N - 1; k >= 1; k--)
      for (int i = IstrU - 1; i <= Iend; i++)
        P[X3(i, j, k)] =
            P[X3(i, j, k + 1)] +
            HalfGRho * ((rho[X3(i, j, k + 1)] + rho[X3(i, j, k)]) * (z_r[X3(i, j, k + 1)] - z_r[X3(i, j, k)]) -
                        OneFifth * ((CX(dR, i, k + 1) - CX(dR, i, k)) *
                                        (z_r[X3(i, j, k + 1)] - z_r[X3(i, j, k)] -
                                         OneTwelfth * (CX(dZ, i, k + 1) + CX(dZ, i, k))) -
                                    (CX(dZ, i, k + 1) - CX(dZ, i, k)) *
                                        (rho[X3(i, j, k + 1)] - rho[X3(i, j, k)] -
                                         OneTwelfth * (CX(dR, i, k + 1) + CX(dR, i, k)))));
  }
  /* XI-component */
  for (int k = N; k >= 1; k--) {
    for (int j = Jstr; j <= Jend; j++)
      for (int i = IstrU - 1; i <= Iend + 1; i++) {
        aux[X2(i, j)] = z_r[X3(i, j, k)] - z_r[X3(i - 1, j, k)];
        FC[X2(i, j)] = rho[X3(i, j, k)] - rho[X3(i - 1, j, k)];
        if (o->c.options & ORC_MASKING) {                                                   /* prsgrd32.h:316,320 */
          aux[X2(i, j)] = aux[X2(i, j)] * o->umask[X2(i, j)];
          FC[X2(i, j)] = FC[X2(i, j)] * o->umask[X2(i, j)];
        }
      }
    for (int j = Jstr; j <= Jend; j++)
      for (int i = IstrU - 1; i <= Iend; i++) {
        cff = 2.0 * aux[X2(i, j)] * aux[X2(i + 1, j)];
        if (cff > eps) {
          cff1 = 1.0 / (aux[X2(i, j)] + aux[X2(i + 1, j)]);
          dZx[X2(i, j)] = cff * cff1;
        } else dZx[X2(i, j)] = 0.0;
        cff1 = 2.0 * FC[X2(i, j)] * FC[X2(i + 1, j)];
        if (cff1 > eps) {
          cff2 = 1.0 / (FC[X2(i, j)] + FC[X2(i + 1, j)]);
          dRx[X2(i, j)] = cff1 * cff2;
        } else dRx[X2(i, j)] = 0.0;
      }
    for (int j = Jstr; j <= Jend; j++)
      for (int i = IstrU; i <= Iend; i++) {
        ru[XW4(i, j, k, nrhs)] =
            o->on_u[X2(i, j)] * 0.5 * (Hz[X3(i, j, k)] + Hz[X3(i - 1, j, k)]) *
            (P[X3(i - 1, j, k)] - P[X3(i, j, k)] -
             HalfGRho * ((rho[X3(i, j, k)] + rho[X3(i - 1, j, k)]) * (z_r[X3(i, j, k)] - z_r[X3(i - 1, j, k)]) -
                         OneFifth * ((dRx[X2(i, j)] - dRx[X2(i - 1, j)]) *
                                         (z_r[X3(i, j, k)] - z_r[X3(i - 1, j, k)] -
                                          OneTwelfth * (dZx[X2(i, j)] + dZx[X2(i - 1, j)])) -
                                     (dZx[X2(i, j)] - dZx[X2(i - 1, j)]) *
                                         (rho[X3(i, j, k)] - rho[X3(i - 1, j, k)] -
                                          OneTwelfth * (dRx[X2(i, j)] + dRx[X2(i - 1, j)])))));
        if (o->wet_dry) ru[XW4(i, j, k, nrhs)] = ru[XW4(i, j, k, nrhs)] * o->umask_wet[X2(i, j)];   /* prsgrd32.h:362 */
      }
  }
  /* ETA-component */
  for (int k = N; k >= 1; k--) {
    for (int j = JstrV - 1; j <= Jend + 1; j++)
      for (int i = Istr; i <= Iend; i++) {
        aux[X2(i, j)] = z_r[X3(i, j, k)] - z_r[X3(i, j - 1, k)];
        FC[X2(i, j)] = rho[X3(i, j, k)] - rho[X3(i, j - 1, k)];
        if (o->c.options & ORC_MASKING) {                                                   /* prsgrd32.h:380,384 */
          aux[X2(i, j)] = aux[X2(i, j)] * o->vmask[X2(i, j)];
          FC[X2(i, j)] = FC[X2(i, j)] * o->vmask[X2(i, j)];
        }
      }
    for (int j = JstrV - 1; j <= Jend; j++)
      for (int i = Istr; i <= Iend; i++) {
        cff = 2.0 * aux[X2(i, j)] * aux[X2(i, j + 1)];
        if (cff > eps) {
          cff1 = 1.0 / (aux[X2(i, j)] + aux[X2(i, j + 1)]);
          dZx[X2(i, j)] = cff * cff1;
        } else dZx[X2(i, j)] = 0.0;
        cff1 = 2.0 * FC[X2(i, j)] * FC[X2(i, j + 1)];
        if (cff1 > eps) {
          cff2 = 1.0 / (FC[X2(i, j)] + FC[X2(i, j + 1)]);
          dRx[X2(i, j)] = cff1 * cff2;
        } else dRx[X2(i, j)] = 0.0;
      }
    for (int j = JstrV; j <= Jend; j++)
      for (int i = Istr; i <= Iend; i++) {
        rv[XW4(i, j, k, nrhs)] =
            o->om_v[X2(i, j)] * 0.5 * (Hz[X3(i, j, k)] + Hz[X3(i, j - 1, k)]) *
            (P[X3(i, j - 1, k)] - P[X3(i, j, k)] -
             HalfGRho * ((rho[X3(i, j, k)] + rho[X3(i, j - 1, k)]) * (z_r[X3(i, j, k)] - z_r[X3(i, j - 1, k)]) -
                         OneFifth * ((dRx[X2(i, j)] - dRx[X2(i, j - 1)]) *
                                         (z_r[X3(i, j, k)] - z_r[X3(i, j - 1, k)] -
                                          OneTwelfth * (dZx[X2(i, j)] + dZx[X2(i, j - 1)])) -
                                     (dZx[X2(i, j)] - dZx[X2(i, j - 1)]) *
                                         (rho[X3(i, j, k)] - rho[X3(i, j - 1, k)] -
                                          OneTwelfth * (dRx[X2(i, j)] + dRx[X2(i, j - 1)])))));
        if (o->wet_dry) rv[XW4(i, j, k, nrhs)] = rv[XW4(i, j, k, nrhs)] * o->vmask_wet[X2(i, j)];   /* prsgrd32.h:426 */
      }
  }
  free(P);
  free(dR);
  free(FC);
}

/* ------------------------------------------------------------- t3dmix2_s */
void orc_t3dmix2(orc_t *o, int tile) {
  if (!(o->c.options & ORC_TS_DIF2)) return;
  if (o->c.options & ORC_MIX_GEO_TS) { orc_t3dmix2_geo(o, tile); return; }
  if (o->c.options & ORC_MIX_ISO_TS) { orc_t3dmix2_iso(o, tile); return; }
  ORC_LOCALS(o);
  const orc_bounds *b = &o->b[tile];
  const int nrhs = o->s.nrhs, nnew = o->s.nnew;
  const int Istr = b->Istr, Iend = b->Iend, Jstr = b->Jstr, Jend = b->Jend;
  double *t = o->t, *Hz = o->Hz, *diff2 = o->diff2, *pm = o->pm, *pn = o->pn;
  double cff, cff1, cff2, cff3;
  double *FX = (double *)calloc(2 * nij, sizeof(double)), *FE = FX + nij;
  for (int itrc = 1; itrc <= o->c.NT; itrc++)
    for (int k = 1; k <= N; k++) {
      for (int j = Jstr; j <= Jend; j++)
        for (int i = Istr; i <= Iend + 1; i++) {
          cff = 0.25 * (diff2[X2T(i, j, itrc)] + diff2[X2T(i - 1, j, itrc)]) * o->pmon_u[X2(i, j)];
          FX[X2(i, j)] = cff * (Hz[X3(i, j, k)] + Hz[X3(i - 1, j, k)]) *
                         (t[XT(i, j, k, nrhs, itrc)] - t[XT(i - 1, j, k, nrhs, itrc)]);
          if (o->c.options & ORC_MASKING) FX[X2(i, j)] = FX[X2(i, j)] * o->umask[X2(i, j)];   /* t3dmix2_s.h:236 */
          if (o->wet_dry) FX[X2(i, j)] = FX[X2(i, j)] * o->umask_wet[X2(i, j)];               /* :239 */
        }
      for (int j = Jstr; j <= Jend + 1; j++)
        for (int i = Istr; i <= Iend; i++) {
          cff = 0.25 * (diff2[X2T(i, j, itrc)] + diff2[X2T(i, j - 1, itrc)]) * o->pnom_v[X2(i, j)];
          FE[X2(i, j)] = cff * (Hz[X3(i, j, k)] + Hz[X3(i, j - 1, k)]) *
                         (t[XT(i, j, k, nrhs, itrc)] - t[XT(i, j - 1, k, nrhs, itrc)]);
          if (o->c.options & ORC_MASKING) FE[X2(i, j)] = FE[X2(i, j)] * o->vmask[X2(i, j)];   /* t3dmix2_s.h:276 */
          if (o->wet_dry) FE[X2(i, j)] = FE[X2(i, j)] * o->vmask_wet[X2(i, j)];               /* :279 */
        }
      for (int j = Jstr; j <= Jend; j++)
        for (int i = Istr; i <= Iend; i++) {
          cff = o->c.dt * pm[X2(i, j)] * pn[X2(i, j)];
          cff1 = cff * (FX[X2(i + 1, j)] - FX[X2(i, j)]);
          cff2 = cff * (FE[X2(i, j + 1)] - FE[X2(i, j)]);
          cff3 = cff1 + cff2;
          t[XT(i, j, k, nnew, itrc)] = t[XT(i, j, k, nnew, itrc)] + cff3;
          if (o->dia) {                                                 /* DIAGNOSTICS_TS t3dmix2_s.h:293-297 */
            orc_dia_wrk(o, ORC_DIA_XDIF, itrc)[X3(i, j, k)] = cff1;
            orc_dia_wrk(o, ORC_DIA_YDIF, itrc)[X3(i, j, k)] = cff2;
            orc_dia_wrk(o, ORC_DIA_HDIF, itrc)[X3(i, j, k)] = cff3;
          }
        }
    }
  free(FX);
}

/* ------------------------------------------------------------ uv3dmix2_s */
void orc_uv3dmix2(orc_t *o, int tile) {
  if (!(o->c.options & ORC_UV_VIS2)) return;
  if (o->mix_geo_uv) { orc_uv3dmix2_geo(o, tile); return; }          /* uv3dmix.F: uv3dmix2_geo.h */
  ORC_LOCALS(o);
  const orc_bounds *b = &o->b[tile];
  const int nrhs = o->s.nrhs, nnew = o->s.nnew;
  const int Istr = b->Istr, Iend = b->Iend, Jstr = b->Jstr, Jend = b->Jend;
  const int IstrU = b->IstrU, JstrV = b->JstrV;
  const double dt = o->c.dt;
  double *u = o->u, *v = o->v, *Hz = o->Hz, *pm = o->pm, *pn = o->pn;
  double *om_r = o->om_r, *on_r = o->on_r, *om_p = o->om_p, *on_p = o->on_p;
  double cff, cff1, cff2, cff3;
  double *UFe = (double *)calloc(4 * nij, sizeof(double));
  double *VFe = UFe + nij, *UFx = UFe + 2 * nij, *VFx = UFe + 3 * nij;
  for (int k = 1; k <= N; k++) {
    for (int j = JstrV - 1; j <= Jend; j++)
      for (int i = IstrU - 1; i <= Iend; i++) {
        cff = Hz[X3(i, j, k)] * 0.5 *
              (o->pmon_r[X2(i, j)] * ((pn[X2(i, j)] + pn[X2(i + 1, j)]) * u[X4(i + 1, j, k, nrhs)] -
                                      (pn[X2(i - 1, j)] + pn[X2(i, j)]) * u[X4(i, j, k, nrhs)]) -
               o->pnom_r[X2(i, j)] * ((pm[X2(i, j)] + pm[X2(i, j + 1)]) * v[X4(i, j + 1, k, nrhs)] -
                                      (pm[X2(i, j - 1)] + pm[X2(i, j)]) * v[X4(i, j, k, nrhs)]));
        UFx[X2(i, j)] = on_r[X2(i, j)] * on_r[X2(i, j)] * o->visc2_r[X2(i, j)] * cff;
        VFe[X2(i, j)] = om_r[X2(i, j)] * om_r[X2(i, j)] * o->visc2_r[X2(i, j)] * cff;
      }
    for (int j = Jstr; j <= Jend + 1; j++)
      for (int i = Istr; i <= Iend + 1; i++) {
        cff = 0.125 * (Hz[X3(i - 1, j, k)] + Hz[X3(i, j, k)] + Hz[X3(i - 1, j - 1, k)] + Hz[X3(i, j - 1, k)]) *
              (o->pmon_p[X2(i, j)] * ((pn[X2(i, j - 1)] + pn[X2(i, j)]) * v[X4(i, j, k, nrhs)] -
                                      (pn[X2(i - 1, j - 1)] + pn[X2(i - 1, j)]) * v[X4(i - 1, j, k, nrhs)]) +
               o->pnom_p[X2(i, j)] * ((pm[X2(i - 1, j)] + pm[X2(i, j)]) * u[X4(i, j, k, nrhs)] -
                                      (pm[X2(i - 1, j - 1)] + pm[X2(i, j - 1)]) * u[X4(i, j - 1, k, nrhs)]));
        if (o->c.options & ORC_MASKING) cff = cff * o->pmask[X2(i, j)];                       /* uv3dmix2_s.h:273 */
        if (o->wet_dry) cff = cff * o->pmask_wet[X2(i, j)];                                   /* :276 */
        UFe[X2(i, j)] = om_p[X2(i, j)] * om_p[X2(i, j)] * o->visc2_p[X2(i, j)] * cff;
        VFx[X2(i, j)] = on_p[X2(i, j)] * on_p[X2(i, j)] * o->visc2_p[X2(i, j)] * cff;
      }
    for (int j = Jstr; j <= Jend; j++)
      for (int i = IstrU; i <= Iend; i++) {
        cff = dt * 0.25 * (pm[X2(i - 1, j)] + pm[X2(i, j)]) * (pn[X2(i - 1, j)] + pn[X2(i, j)]);
        cff1 = 0.5 * (pn[X2(i - 1, j)] + pn[X2(i, j)]) * (UFx[X2(i, j)] - UFx[X2(i - 1, j)]);
        cff2 = 0.5 * (pm[X2(i - 1, j)] + pm[X2(i, j)]) * (UFe[X2(i, j + 1)] - UFe[X2(i, j)]);
        cff3 = cff * (cff1 + cff2);
        o->rufrc[X2(i, j)] = o->rufrc[X2(i, j)] + cff1 + cff2;
        u[X4(i, j, k, nnew)] = u[X4(i, j, k, nnew)] + cff3;
        if (o->duv) {                                                            /* uv3dmix2_s.h:303-308 */
          const orc_diauv *d = o->duv;
          DUF(d->RUfrc, i, j, 3, d->M2hvis) = DUF(d->RUfrc, i, j, 3, d->M2hvis) + cff1 + cff2;
          DUF(d->RUfrc, i, j, 3, d->M2xvis) = DUF(d->RUfrc, i, j, 3, d->M2xvis) + cff1;
          DUF(d->RUfrc, i, j, 3, d->M2yvis) = DUF(d->RUfrc, i, j, 3, d->M2yvis) + cff2;
          DU3(d->U3wrk, i, j, k, d->M3hvis) = cff3;
          DU3(d->U3wrk, i, j, k, d->M3xvis) = cff * cff1;
          DU3(d->U3wrk, i, j, k, d->M3yvis) = cff * cff2;
        }
      }
    for (int j = JstrV; j <= Jend; j++)
      for (int i = Istr; i <= Iend; i++) {
        cff = dt * 0.25 * (pm[X2(i, j)] + pm[X2(i, j - 1)]) * (pn[X2(i, j)] + pn[X2(i, j - 1)]);
        cff1 = 0.5 * (pn[X2(i, j - 1)] + pn[X2(i, j)]) * (VFx[X2(i + 1, j)] - VFx[X2(i, j)]);
        cff2 = 0.5 * (pm[X2(i, j - 1)] + pm[X2(i, j)]) * (VFe[X2(i, j)] - VFe[X2(i, j - 1)]);
        cff3 = cff * (cff1 - cff2);
        o->rvfrc[X2(i, j)] = o->rvfrc[X2(i, j)] + cff1 - cff2;
        v[X4(i, j, k, nnew)] = v[X4(i, j, k, nnew)] + cff3;
        if (o->duv) {                                                            /* :321-326 */
          const orc_diauv *d = o->duv;
          DUF(d->RVfrc, i, j, 3, d->M2hvis) = DUF(d->RVfrc, i, j, 3, d->M2hvis) + cff1 - cff2;
          DUF(d->RVfrc, i, j, 3, d->M2xvis) = DUF(d->RVfrc, i, j, 3, d->M2xvis) + cff1;
          DUF(d->RVfrc, i, j, 3, d->M2yvis) = DUF(d->RVfrc, i, j, 3, d->M2yvis) - cff2;
          DU3(d->V3wrk, i, j, k, d->M3hvis) = cff3;
          DU3(d->V3wrk, i, j, k, d->M3xvis) = cff * cff1;
          DU3(d->V3wrk, i, j, k, d->M3yvis) = -cff * cff2;
        }
      }
  }
  free(UFe);
}

/* ------------------------------------------------------------ rhs3d_tile */
void orc_rhs3d_tile(orc_t *o, int tile) {
  ORC_LOCALS(o);
  const orc_bounds *b = &o->b[tile];
  const orc_cfg *c = &o->c;
  const int nrhs = o->s.nrhs;
  const int Istr = b->Istr, Iend = b->Iend, Jstr = b->Jstr, Jend = b->Jend;
  const int IstrU = b->IstrU, JstrV = b->JstrV;
  const double Gadv = -0.25;
  double *u = o->u, *v = o->v, *Hz = o->Hz, *Huon = o->Huon, *Hvom = o->Hvom, *W = o->W;
  double *ru = o->ru, *rv = o->rv;
  double cff, cff1, cff2;
  double *S = (double *)calloc(12 * nij, sizeof(double));
  double *Huee = S, *Huxx = S + nij, *Hvee = S + 2 * nij, *Hvxx = S + 3 * nij, *UFx = S + 4 * nij,
         *UFe = S + 5 * nij, *VFx = S + 6 * nij, *VFe = S + 7 * nij, *uee = S + 8 * nij,
         *uxx = S + 9 * nij, *vee = S + 10 * nij, *vxx = S + 11 * nij;
  double *FC = (double *)calloc(ni * (size_t)(N + 1), sizeof(double));
  const orc_diauv *d = o->duv;                      /* DIAGNOSTICS_UV: the stores of rhs3d.F under that option */
  const int curv = (c->options & ORC_CURVGRID) && (c->options & ORC_UV_ADV);
  double *Uwrk = d && curv ? (double *)calloc(2 * nij, sizeof(double)) : NULL, *Vwrk = Uwrk ? Uwrk + nij : NULL;
#define U(i, j, k) u[X4(i, j, k, nrhs)]
#define V(i, j, k) v[X4(i, j, k, nrhs)]
#define RU(i, j, k) ru[XW4(i, j, k, nrhs)]
#define RV(i, j, k) rv[XW4(i, j, k, nrhs)]
  for (int k = 1; k <= N; k++) {
    if (c->options & ORC_UV_COR) {
      /* Coriolis :500-560 */
      for (int j = JstrV - 1; j <= Jend; j++)
        for (int i = IstrU - 1; i <= Iend; i++) {
          cff = 0.5 * Hz[X3(i, j, k)] * o->fomn[X2(i, j)];
          UFx[X2(i, j)] = cff * (V(i, j, k) + V(i, j + 1, k));
          VFe[X2(i, j)] = cff * (U(i, j, k) + U(i + 1, j, k));
        }
      for (int j = Jstr; j <= Jend; j++)
        for (int i = IstrU; i <= Iend; i++) {
          cff1 = 0.5 * (UFx[X2(i, j)] + UFx[X2(i - 1, j)]);
          RU(i, j, k) = RU(i, j, k) + cff1;
          if (d) DUR(d->RU, i, j, k, nrhs, d->M3fcor) = cff1;                     /* :520 */
        }
      for (int j = JstrV; j <= Jend; j++)
        for (int i = Istr; i <= Iend; i++) {
          cff1 = 0.5 * (VFe[X2(i, j)] + VFe[X2(i, j - 1)]);
          RV(i, j, k) = RV(i, j, k) - cff1;
          if (d) DUR(d->RV, i, j, k, nrhs, d->M3fcor) = -cff1;                    /* :529 */
        }
    }
    if ((c->options & ORC_CURVGRID) && (c->options & ORC_UV_ADV)) {
      /* curvilinear terms :564-645 */
      for (int j = JstrV - 1; j <= Jend; j++)
        for (int i = IstrU - 1; i <= Iend; i++) {
          cff1 = 0.5 * (V(i, j, k) + V(i, j + 1, k));
          cff2 = 0.5 * (U(i, j, k) + U(i + 1, j, k));
          double cff3 = cff1 * o->dndx[X2(i, j)];
          double cff4 = cff2 * o->dmde[X2(i, j)];
          cff = Hz[X3(i, j, k)] * (cff3 - cff4);
          UFx[X2(i, j)] = cff * cff1;
          VFe[X2(i, j)] = cff * cff2;
          if (d) {                                                                /* :601-604 */
            cff = Hz[X3(i, j, k)] * cff4;
            Uwrk[X2(i, j)] = -cff * cff1;
            Vwrk[X2(i, j)] = -cff * cff2;
          }
        }
      for (int j = Jstr; j <= Jend; j++)
        for (int i = IstrU; i <= Iend; i++) {
          cff1 = 0.5 * (UFx[X2(i, j)] + UFx[X2(i - 1, j)]);
          RU(i, j, k) = RU(i, j, k) + cff1;
          if (d) {                                                                /* :617-625 */
            cff2 = 0.5 * (Uwrk[X2(i, j)] + Uwrk[X2(i - 1, j)]);
            DUR(d->RU, i, j, k, nrhs, d->M3xadv) = cff1 - cff2;
            DUR(d->RU, i, j, k, nrhs, d->M3yadv) = cff2;
            DUR(d->RU, i, j, k, nrhs, d->M3hadv) = cff1;
          }
        }
      for (int j = JstrV; j <= Jend; j++)
        for (int i = Istr; i <= Iend; i++) {
          cff1 = 0.5 * (VFe[X2(i, j)] + VFe[X2(i, j - 1)]);
          RV(i, j, k) = RV(i, j, k) - cff1;
          if (d) {                                                                /* :635-643 */
            cff2 = 0.5 * (Vwrk[X2(i, j)] + Vwrk[X2(i, j - 1)]);
            DUR(d->RV, i, j, k, nrhs, d->M3xadv) = -cff1 + cff2;
            DUR(d->RV, i, j, k, nrhs, d->M3yadv) = -cff2;
            DUR(d->RV, i, j, k, nrhs, d->M3hadv) = -cff1;
          }
        }
    }
    if (o->clima_flags & 1) {                                            /* nudging of 3-D momentum climatology :654-680 */
      for (int j = Jstr; j <= Jend; j++)
        for (int i = IstrU; i <= Iend; i++) {
          cff = 0.25 * (o->M3nudgcof[X3(i - 1, j, k)] + o->M3nudgcof[X3(i, j, k)]) * o->om_u[X2(i, j)] * o->on_u[X2(i, j)];
          RU(i, j, k) = RU(i, j, k) + cff * (Hz[X3(i - 1, j, k)] + Hz[X3(i, j, k)]) * (o->uclm[X3(i, j, k)] - u[X4(i, j, k, nrhs)]);
        }
      for (int j = JstrV; j <= Jend; j++)
        for (int i = Istr; i <= Iend; i++) {
          cff = 0.25 * (o->M3nudgcof[X3(i, j - 1, k)] + o->M3nudgcof[X3(i, j, k)]) * o->om_v[X2(i, j)] * o->on_v[X2(i, j)];
          RV(i, j, k) = RV(i, j, k) + cff * (Hz[X3(i, j - 1, k)] + Hz[X3(i, j, k)]) * (o->vclm[X3(i, j, k)] - v[X4(i, j, k, nrhs)]);
        }
    }
    if (!(c->options & ORC_UV_ADV)) continue;
    /* third-order upstream horizontal advection :679-1000 */
    for (int j = Jstr; j <= Jend; j++)
      for (int i = b->IstrUm1; i <= b->Iendp1; i++) {
        uxx[X2(i, j)] = U(i - 1, j, k) - 2.0 * U(i, j, k) + U(i + 1, j, k);
        Huxx[X2(i, j)] = Huon[X3(i - 1, j, k)] - 2.0 * Huon[X3(i, j, k)] + Huon[X3(i + 1, j, k)];
      }
    if (!c->EWperiodic) {
      if (b->west)
        for (int j = Jstr; j <= Jend; j++) {
          uxx[X2(Istr, j)] = uxx[X2(Istr + 1, j)];
          Huxx[X2(Istr, j)] = Huxx[X2(Istr + 1, j)];
        }
      if (b->east)
        for (int j = Jstr; j <= Jend; j++) {
          uxx[X2(Iend + 1, j)] = uxx[X2(Iend, j)];
          Huxx[X2(Iend + 1, j)] = Huxx[X2(Iend, j)];
        }
    }
    for (int j = Jstr; j <= Jend; j++)
      for (int i = IstrU - 1; i <= Iend; i++) {
        cff1 = U(i, j, k) + U(i + 1, j, k);
        if (cff1 > 0.0) cff = uxx[X2(i, j)];
        else cff = uxx[X2(i + 1, j)];
        UFx[X2(i, j)] = 0.25 * (cff1 + Gadv * cff) *
                        (Huon[X3(i, j, k)] + Huon[X3(i + 1, j, k)] +
                         Gadv * 0.5 * (Huxx[X2(i, j)] + Huxx[X2(i + 1, j)]));
      }
    for (int j = b->Jstrm1; j <= b->Jendp1; j++)
      for (int i = IstrU; i <= Iend; i++)
        uee[X2(i, j)] = U(i, j - 1, k) - 2.0 * U(i, j, k) + U(i, j + 1, k);
    if (!c->NSperiodic) {
      if (b->south) for (int i = IstrU; i <= Iend; i++) uee[X2(i, Jstr - 1)] = uee[X2(i, Jstr)];
      if (b->north) for (int i = IstrU; i <= Iend; i++) uee[X2(i, Jend + 1)] = uee[X2(i, Jend)];
    }
    for (int j = Jstr; j <= Jend + 1; j++)
      for (int i = IstrU - 1; i <= Iend; i++)
        Hvxx[X2(i, j)] = Hvom[X3(i - 1, j, k)] - 2.0 * Hvom[X3(i, j, k)] + Hvom[X3(i + 1, j, k)];
    for (int j = Jstr; j <= Jend + 1; j++)
      for (int i = IstrU; i <= Iend; i++) {
        cff1 = U(i, j, k) + U(i, j - 1, k);
        cff2 = Hvom[X3(i, j, k)] + Hvom[X3(i - 1, j, k)];
        if (cff2 > 0.0) cff = uee[X2(i, j - 1)];
        else cff = uee[X2(i, j)];
        UFe[X2(i, j)] = 0.25 * (cff1 + Gadv * cff) *
                        (cff2 + Gadv * 0.5 * (Hvxx[X2(i, j)] + Hvxx[X2(i - 1, j)]));
      }
    for (int j = JstrV; j <= Jend; j++)
      for (int i = b->Istrm1; i <= b->Iendp1; i++)
        vxx[X2(i, j)] = V(i - 1, j, k) - 2.0 * V(i, j, k) + V(i + 1, j, k);
    if (!c->EWperiodic) {
      if (b->west) for (int j = JstrV; j <= Jend; j++) vxx[X2(Istr - 1, j)] = vxx[X2(Istr, j)];
      if (b->east) for (int j = JstrV; j <= Jend; j++) vxx[X2(Iend + 1, j)] = vxx[X2(Iend, j)];
    }
    for (int j = JstrV - 1; j <= Jend; j++)
      for (int i = Istr; i <= Iend + 1; i++)
        Huee[X2(i, j)] = Huon[X3(i, j - 1, k)] - 2.0 * Huon[X3(i, j, k)] + Huon[X3(i, j + 1, k)];
    for (int j = JstrV; j <= Jend; j++)
      for (int i = Istr; i <= Iend + 1; i++) {
        cff1 = V(i, j, k) + V(i - 1, j, k);
        cff2 = Huon[X3(i, j, k)] + Huon[X3(i, j - 1, k)];
        if (cff2 > 0.0) cff = vxx[X2(i - 1, j)];
        else cff = vxx[X2(i, j)];
        VFx[X2(i, j)] = 0.25 * (cff1 + Gadv * cff) *
                        (cff2 + Gadv * 0.5 * (Huee[X2(i, j)] + Huee[X2(i, j - 1)]));
      }
    for (int j = b->JstrVm1; j <= b->Jendp1; j++)
      for (int i = Istr; i <= Iend; i++) {
        vee[X2(i, j)] = V(i, j - 1, k) - 2.0 * V(i, j, k) + V(i, j + 1, k);
        Hvee[X2(i, j)] = Hvom[X3(i, j - 1, k)] - 2.0 * Hvom[X3(i, j, k)] + Hvom[X3(i, j + 1, k)];
      }
    if (!c->NSperiodic) {
      if (b->south)
        for (int i = Istr; i <= Iend; i++) {
          vee[X2(i, Jstr)] = vee[X2(i, Jstr + 1)];
          Hvee[X2(i, Jstr)] = Hvee[X2(i, Jstr + 1)];
        }
      if (b->north)
        for (int i = Istr; i <= Iend; i++) {
          vee[X2(i, Jend + 1)] = vee[X2(i, Jend)];
          Hvee[X2(i, Jend + 1)] = Hvee[X2(i, Jend)];
        }
    }
    for (int j = JstrV - 1; j <= Jend; j++)
      for (int i = Istr; i <= Iend; i++) {
        cff1 = V(i, j, k) + V(i, j + 1, k);
        if (cff1 > 0.0) cff = vee[X2(i, j)];
        else cff = vee[X2(i, j + 1)];
        VFe[X2(i, j)] = 0.25 * (cff1 + Gadv * cff) *
                        (Hvom[X3(i, j, k)] + Hvom[X3(i, j + 1, k)] +
                         Gadv * 0.5 * (Hvee[X2(i, j)] + Hvee[X2(i, j + 1)]));
      }
    for (int j = Jstr; j <= Jend; j++)
      for (int i = IstrU; i <= Iend; i++) {
        cff1 = UFx[X2(i, j)] - UFx[X2(i - 1, j)];
        cff2 = UFe[X2(i, j + 1)] - UFe[X2(i, j)];
        cff = cff1 + cff2;
        RU(i, j, k) = RU(i, j, k) - cff;
        if (d) {                                                                  /* :971-980 */
          if (curv) {
            DUR(d->RU, i, j, k, nrhs, d->M3xadv) = DUR(d->RU, i, j, k, nrhs, d->M3xadv) - cff1;
            DUR(d->RU, i, j, k, nrhs, d->M3yadv) = DUR(d->RU, i, j, k, nrhs, d->M3yadv) - cff2;
            DUR(d->RU, i, j, k, nrhs, d->M3hadv) = DUR(d->RU, i, j, k, nrhs, d->M3hadv) - cff;
          } else {
            DUR(d->RU, i, j, k, nrhs, d->M3xadv) = -cff1;
            DUR(d->RU, i, j, k, nrhs, d->M3yadv) = -cff2;
            DUR(d->RU, i, j, k, nrhs, d->M3hadv) = -cff;
          }
        }
      }
    for (int j = JstrV; j <= Jend; j++)
      for (int i = Istr; i <= Iend; i++) {
        cff1 = VFx[X2(i + 1, j)] - VFx[X2(i, j)];
        cff2 = VFe[X2(i, j)] - VFe[X2(i, j - 1)];
        cff = cff1 + cff2;
        RV(i, j, k) = RV(i, j, k) - cff;
        if (d) {                                                                  /* :990-999 */
          if (curv) {
            DUR(d->RV, i, j, k, nrhs, d->M3xadv) = DUR(d->RV, i, j, k, nrhs, d->M3xadv) - cff1;
            DUR(d->RV, i, j, k, nrhs, d->M3yadv) = DUR(d->RV, i, j, k, nrhs, d->M3yadv) - cff2;
            DUR(d->RV, i, j, k, nrhs, d->M3hadv) = DUR(d->RV, i, j, k, nrhs, d->M3hadv) - cff;
          } else {
            DUR(d->RV, i, j, k, nrhs, d->M3xadv) = -cff1;
            DUR(d->RV, i, j, k, nrhs, d->M3yadv) = -cff2;
            DUR(d->RV, i, j, k, nrhs, d->M3hadv) = -cff;
          }
        }
      }
  }

  /* vertical advection (default 4th-order, 9/16 1/16) and vertical sums :1132-1918 */
  for (int j = Jstr; j <= Jend; j++) {
    if (c->options & ORC_UV_ADV) {
      cff1 = 9.0 / 16.0;
      cff2 = 1.0 / 16.0;
      for (int k = 2; k <= N - 2; k++)
        for (int i = IstrU; i <= Iend; i++)
          CX(FC, i, k) = (cff1 * (U(i, j, k) + U(i, j, k + 1)) - cff2 * (U(i, j, k - 1) + U(i, j, k + 2))) *
                         (cff1 * (W[XW(i, j, k)] + W[XW(i - 1, j, k)]) -
                          cff2 * (W[XW(i + 1, j, k)] + W[XW(i - 2, j, k)]));
      for (int i = IstrU; i <= Iend; i++) {
        CX(FC, i, N) = 0.0;
        CX(FC, i, N - 1) = (cff1 * (U(i, j, N - 1) + U(i, j, N)) - cff2 * (U(i, j, N - 2) + U(i, j, N))) *
                           (cff1 * (W[XW(i, j, N - 1)] + W[XW(i - 1, j, N - 1)]) -
                            cff2 * (W[XW(i + 1, j, N - 1)] + W[XW(i - 2, j, N - 1)]));
        CX(FC, i, 1) = (cff1 * (U(i, j, 1) + U(i, j, 2)) - cff2 * (U(i, j, 1) + U(i, j, 3))) *
                       (cff1 * (W[XW(i, j, 1)] + W[XW(i - 1, j, 1)]) -
                        cff2 * (W[XW(i + 1, j, 1)] + W[XW(i - 2, j, 1)]));
        CX(FC, i, 0) = 0.0;
      }
      for (int k = 1; k <= N; k++)
        for (int i = IstrU; i <= Iend; i++) {
          cff = CX(FC, i, k) - CX(FC, i, k - 1);
          RU(i, j, k) = RU(i, j, k) - cff;
          if (d) DUR(d->RU, i, j, k, nrhs, d->M3vadv) = -cff;                      /* :1173 */
        }
      if (j >= JstrV) {
        for (int k = 2; k <= N - 2; k++)
          for (int i = Istr; i <= Iend; i++)
            CX(FC, i, k) = (cff1 * (V(i, j, k) + V(i, j, k + 1)) - cff2 * (V(i, j, k - 1) + V(i, j, k + 2))) *
                           (cff1 * (W[XW(i, j, k)] + W[XW(i, j - 1, k)]) -
                            cff2 * (W[XW(i, j + 1, k)] + W[XW(i, j - 2, k)]));
        for (int i = Istr; i <= Iend; i++) {
          CX(FC, i, N) = 0.0;
          CX(FC, i, N - 1) = (cff1 * (V(i, j, N - 1) + V(i, j, N)) - cff2 * (V(i, j, N - 2) + V(i, j, N))) *
                             (cff1 * (W[XW(i, j, N - 1)] + W[XW(i, j - 1, N - 1)]) -
                              cff2 * (W[XW(i, j + 1, N - 1)] + W[XW(i, j - 2, N - 1)]));
          CX(FC, i, 1) = (cff1 * (V(i, j, 1) + V(i, j, 2)) - cff2 * (V(i, j, 1) + V(i, j, 3))) *
                         (cff1 * (W[XW(i, j, 1)] + W[XW(i, j - 1, 1)]) -
                          cff2 * (W[XW(i, j + 1, 1)] + W[XW(i, j - 2, 1)]));
          CX(FC, i, 0) = 0.0;
        }
        for (int k = 1; k <= N; k++)
          for (int i = Istr; i <= Iend; i++) {
            cff = CX(FC, i, k) - CX(FC, i, k - 1);
            RV(i, j, k) = RV(i, j, k) - cff;
            if (d) DUR(d->RV, i, j, k, nrhs, d->M3vadv) = -cff;                    /* :1323 */
          }
      }
    }
    if (o->wet_dry)                                      /* rhs3d.F:1709,1750 */
      for (int k = 1; k <= N; k++)
        for (int i = IstrU; i <= Iend; i++) RU(i, j, k) = RU(i, j, k) * o->umask_wet[X2(i, j)];
    for (int i = IstrU; i <= Iend; i++) o->rufrc[X2(i, j)] = RU(i, j, 1);
    for (int k = 2; k <= N; k++)
      for (int i = IstrU; i <= Iend; i++) o->rufrc[X2(i, j)] = o->rufrc[X2(i, j)] + RU(i, j, k);
    if (d) {                                            /* :1712-1790: the vertical sums of the terms, level 3 of DiaRUfrc */
      const int m3[5] = {d->M3pgrd, d->M3fcor, d->M3xadv, d->M3yadv, d->M3hadv};
      const int m2[5] = {d->M2pgrd, d->M2fcor, d->M2xadv, d->M2yadv, d->M2hadv};
      for (int q = 0; q < 5; q++) {
        if (!m3[q]) continue;
        for (int i = IstrU; i <= Iend; i++) DUF(d->RUfrc, i, j, 3, m2[q]) = DUR(d->RU, i, j, 1, nrhs, m3[q]);
        for (int k = 2; k <= N; k++)
          for (int i = IstrU; i <= Iend; i++)
            DUF(d->RUfrc, i, j, 3, m2[q]) = DUF(d->RUfrc, i, j, 3, m2[q]) + DUR(d->RU, i, j, k, nrhs, m3[q]);
      }
      if (d->M2hvis)
        for (int i = IstrU; i <= Iend; i++) {
          DUF(d->RUfrc, i, j, 3, d->M2xvis) = 0.0; DUF(d->RUfrc, i, j, 3, d->M2yvis) = 0.0; DUF(d->RUfrc, i, j, 3, d->M2hvis) = 0.0;
        }
    }
    for (int i = IstrU; i <= Iend; i++) {
      cff = o->om_u[X2(i, j)] * o->on_u[X2(i, j)];
      cff1 = o->sustr[X2(i, j)] * cff;
      cff2 = -o->bustr[X2(i, j)] * cff;
      o->rufrc[X2(i, j)] = o->rufrc[X2(i, j)] + cff1 + cff2;
      if (o->wet_dry) o->rufrc[X2(i, j)] = o->rufrc[X2(i, j)] * o->umask_wet[X2(i, j)];            /* :1804 */
      if (d) { DUF(d->RUfrc, i, j, 3, d->M2sstr) = cff1; DUF(d->RUfrc, i, j, 3, d->M2bstr) = cff2; }    /* :1807 */
    }
    if (j >= JstrV) {
      if (o->wet_dry)                                    /* :1815,1856 */
        for (int k = 1; k <= N; k++)
          for (int i = Istr; i <= Iend; i++) RV(i, j, k) = RV(i, j, k) * o->vmask_wet[X2(i, j)];
      for (int i = Istr; i <= Iend; i++) o->rvfrc[X2(i, j)] = RV(i, j, 1);
      for (int k = 2; k <= N; k++)
        for (int i = Istr; i <= Iend; i++) o->rvfrc[X2(i, j)] = o->rvfrc[X2(i, j)] + RV(i, j, k);
      if (d) {                                          /* :1819-1896 */
        const int m3[5] = {d->M3pgrd, d->M3fcor, d->M3xadv, d->M3yadv, d->M3hadv};
        const int m2[5] = {d->M2pgrd, d->M2fcor, d->M2xadv, d->M2yadv, d->M2hadv};
        for (int q = 0; q < 5; q++) {
          if (!m3[q]) continue;
          for (int i = Istr; i <= Iend; i++) DUF(d->RVfrc, i, j, 3, m2[q]) = DUR(d->RV, i, j, 1, nrhs, m3[q]);
          for (int k = 2; k <= N; k++)
            for (int i = Istr; i <= Iend; i++)
              DUF(d->RVfrc, i, j, 3, m2[q]) = DUF(d->RVfrc, i, j, 3, m2[q]) + DUR(d->RV, i, j, k, nrhs, m3[q]);
        }
        if (d->M2hvis)
          for (int i = Istr; i <= Iend; i++) {
            DUF(d->RVfrc, i, j, 3, d->M2xvis) = 0.0; DUF(d->RVfrc, i, j, 3, d->M2yvis) = 0.0; DUF(d->RVfrc, i, j, 3, d->M2hvis) = 0.0;
          }
      }
      for (int i = Istr; i <= Iend; i++) {
        cff = o->om_v[X2(i, j)] * o->on_v[X2(i, j)];
        cff1 = o->svstr[X2(i, j)] * cff;
        cff2 = -o->bvstr[X2(i, j)] * cff;
        o->rvfrc[X2(i, j)] = o->rvfrc[X2(i, j)] + cff1 + cff2;
        if (o->wet_dry) o->rvfrc[X2(i, j)] = o->rvfrc[X2(i, j)] * o->vmask_wet[X2(i, j)];          /* :1910 */
        if (d) { DUF(d->RVfrc, i, j, 3, d->M2sstr) = cff1; DUF(d->RVfrc, i, j, 3, d->M2bstr) = cff2; }  /* :1913 */
      }
    }
  }
  free(S);
  free(FC);
  free(Uwrk);
#undef U
#undef V
#undef RU
#undef RV
}

/* rhs3d driver rhs3d.F:80-181 */
void orc_rhs3d(orc_t *o, int tile) {
  orc_pre_step3d(o, tile);
  orc_prsgrd(o, tile);
  orc_t3dmix2(o, tile);
  orc_t3dmix4(o, tile);              /* rhs3d.F:141-153: TS_DIF4 behind TS_DIF2 */
  orc_rhs3d_tile(o, tile);
  orc_uv3dmix2(o, tile);
  orc_uv3dmix4(o, tile);             /* :170-178 */
}
