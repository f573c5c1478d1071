/*
 * orc_step2d.c -- barotropic (2-D) engine, LF-AM3 predictor/corrector.
 * TEST INFRASTRUCTURE (see orc.h).
 *
 * Follows step2d_tile, ROMS/Nonlinear/step2d_LF_AM3.h:163-3056 (serial branch,
 * options SOLVE3D, VAR_RHO_2D, UV_ADV (4th-order centred, the #else of
 * UV_C2ADVECTION :1246-1395), UV_COR, CURVGRID, UV_VIS2; no MASKING/WET_DRY/
 * NESTING/DIAGNOSTICS).  Section references in the body.
 *
 * PARITY: pinned bit for bit against step2d_tile of the reference build (first
 * predictor, correctors, last predictor on perturbed states; every call of 100
 * main3d passes): tests/test_oracle_vs_ref.py, tests/test_golden_reference.py.
 */
#include "orc.h"
#include <stdlib.h>
#include <string.h>

void orc_step2d(orc_t *o, int tile) {
  ORC_LOCALS(o);
  const orc_bounds *b = &o->b[tile];
  const orc_cfg *c = &o->c;
  const int krhs = o->s.krhs, kstp = o->s.kstp, knew = o->s.knew;
  const int nstp = o->s.nstp, nnew = o->s.nnew, iif = o->s.iif, iic = o->s.iic;
  const int PRED = o->s.predictor, CORR = !PRED;
  const int Istr = b->Istr, Iend = b->Iend, Jstr = b->Jstr, Jend = b->Jend;
  const int IstrU = b->IstrU, JstrV = b->JstrV, IstrR = b->IstrR, IendR = b->IendR,
            JstrR = b->JstrR, JendR = b->JendR;
  const int ptsk = 3 - kstp;
  const int msk = (c->options & ORC_MASKING) != 0;
  const double dtfast = c->dtfast, g = c->g, rho0 = c->rho0;
  double *zeta = o->zeta, *ubar = o->ubar, *vbar = o->vbar;
  double *rzeta = o->rzeta, *rubar = o->rubar, *rvbar = o->rvbar;
  double *h = o->h, *pm = o->pm, *pn = o->pn, *on_u = o->on_u, *om_v = o->om_v;
  double *rhoA = o->rhoA, *rhoS = o->rhoS;
  double *ru = o->ru, *rv = o->rv, *rufrc = o->rufrc, *rvfrc = o->rvfrc;
  double cff, cff1, cff2, cff3, cff4, cff5, fac, fac1;

  /* private scratch, (IminS:ImaxS,JminS:JmaxS) in the reference :585-610 */
  enum { NS = 20 };
  double *S = (double *)calloc((size_t)NS * nij, sizeof(double));
  double *Dgrad = S, *Dnew = S + nij, *Drhs = S + 2 * nij, *Drhs_p = S + 3 * nij,
         *Dstp = S + 4 * nij, *DUon = S + 5 * nij, *DVom = S + 6 * nij, *UFe = S + 7 * nij,
         *UFx = S + 8 * nij, *VFe = S + 9 * nij, *VFx = S + 10 * nij, *grad = S + 11 * nij,
         *gzeta = S + 12 * nij, *gzeta2 = S + 13 * nij, *gzetaSA = S + 14 * nij,
         *rhs_ubar = S + 15 * nij, *rhs_vbar = S + 16 * nij, *rhs_zeta = S + 17 * nij,
         *zeta_new = S + 18 * nij, *zwrk = S + 19 * nij;
  /* DIAGNOSTICS_UV: DiaU2rhs, DiaV2rhs(IminS:ImaxS,JminS:JmaxS,NDM2d-1), Uwrk, Vwrk :597-601 */
  const orc_diauv *d = o->duv;
  double *U2rhs = d ? (double *)calloc((size_t)(2 * d->NDM2d + 2) * nij, sizeof(double)) : NULL;
  double *V2rhs = d ? U2rhs + (size_t)d->NDM2d * nij : NULL, *Uwrk = d ? U2rhs + (size_t)(2 * d->NDM2d) * nij : NULL,
         *Vwrk = d ? Uwrk + nij : NULL;
#define Z(i, j, n) zeta[X2T(i, j, n)]
#define UB(i, j, n) ubar[X2T(i, j, n)]
#define VB(i, j, n) vbar[X2T(i, j, n)]

  /* total depth and mass fluxes :600-700 */
  for (int j = b->JstrVm2 - 1; j <= b->Jendp2; j++)
    for (int i = b->IstrUm2 - 1; i <= b->Iendp2; i++) Drhs[X2(i, j)] = Z(i, j, krhs) + h[X2(i, j)];
  for (int j = b->JstrVm2 - 1; j <= b->Jendp2; j++)
    for (int i = b->IstrUm2; i <= b->Iendp2; i++) {
      cff = 0.5 * on_u[X2(i, j)];
      cff1 = cff * (Drhs[X2(i, j)] + Drhs[X2(i - 1, j)]);
      DUon[X2(i, j)] = UB(i, j, krhs) * cff1;
    }
  for (int j = b->JstrVm2; j <= b->Jendp2; j++)
    for (int i = b->IstrUm2 - 1; i <= b->Iendp2; i++) {
      cff = 0.5 * om_v[X2(i, j)];
      cff1 = cff * (Drhs[X2(i, j)] + Drhs[X2(i, j - 1)]);
      DVom[X2(i, j)] = VB(i, j, krhs) * cff1;
    }

  /* VolCons: the mass fluxes along the open edges with the correction velocity of the previous call's obc_flux_tile taken off the
     inflow -- set_DUV_bc_tile, obc_volcons.F:236-370, called at step2d_LF_AM3.h:724; the ranges are obc_volcons.F:300-306's own
     (the shared-memory form of I_RANGE / J_RANGE: those of Drhs above except in a periodic direction, where they stop short) */
  if (c->volcons) {
    const double xs = o->ubar_xs;
    const int jlo = (JstrV - 1 > 2 ? JstrV - 1 : 2) - 2, jhi = (Jend + 1 < c->Mm ? Jend + 1 : c->Mm) + 1;
    const int ilo = (IstrU - 1 > 2 ? IstrU - 1 : 2) - 2, ihi = (Iend + 1 < c->Lm ? Iend + 1 : c->Lm) + 1;
    if ((c->volcons & (1 << ORC_IWEST)) && b->west)
      for (int j = jlo; j <= jhi; j++) {
        DUon[X2(Istr, j)] = 0.5 * (Drhs[X2(Istr, j)] + Drhs[X2(Istr - 1, j)]) * (UB(Istr, j, krhs) - xs) * on_u[X2(Istr, j)];
        if (msk) DUon[X2(Istr, j)] = DUon[X2(Istr, j)] * o->umask[X2(Istr, j)];
      }
    if ((c->volcons & (1 << ORC_IEAST)) && b->east)
      for (int j = jlo; j <= jhi; j++) {
        DUon[X2(Iend + 1, j)] = 0.5 * (Drhs[X2(Iend + 1, j)] + Drhs[X2(Iend, j)]) * (UB(Iend + 1, j, krhs) + xs) * on_u[X2(Iend + 1, j)];
        if (msk) DUon[X2(Iend + 1, j)] = DUon[X2(Iend + 1, j)] * o->umask[X2(Iend + 1, j)];
      }
    if ((c->volcons & (1 << ORC_ISOUTH)) && b->south)
      for (int i = ilo; i <= ihi; i++) {
        DVom[X2(i, Jstr)] = 0.5 * (Drhs[X2(i, Jstr)] + Drhs[X2(i, Jstr - 1)]) * (VB(i, Jstr, krhs) - xs) * om_v[X2(i, Jstr)];
        if (msk) DVom[X2(i, Jstr)] = DVom[X2(i, Jstr)] * o->vmask[X2(i, Jstr)];
      }
    if ((c->volcons & (1 << ORC_INORTH)) && b->north)
      for (int i = ilo; i <= ihi; i++) {
        DVom[X2(i, Jend + 1)] = 0.5 * (Drhs[X2(i, Jend + 1)] + Drhs[X2(i, Jend)]) * (VB(i, Jend + 1, krhs) + xs) * om_v[X2(i, Jend + 1)];
        if (msk) DVom[X2(i, Jend + 1)] = DVom[X2(i, Jend + 1)] * o->vmask[X2(i, Jend + 1)];
      }
  }

  /* fast-time averaging :739-880 */
  if (PRED) {
    if (iif == 1) {
      cff2 = (-1.0 / 12.0) * c->weight[1][iif + 1];
      for (int j = JstrR; j <= JendR; j++) {
        for (int i = IstrR; i <= IendR; i++) o->Zt_avg1[X2(i, j)] = 0.0;
        for (int i = Istr; i <= IendR; i++) {
          o->DU_avg1[X2(i, j)] = 0.0;
          o->DU_avg2[X2(i, j)] = cff2 * DUon[X2(i, j)];
        }
      }
      for (int j = Jstr; j <= JendR; j++)
        for (int i = IstrR; i <= IendR; i++) {
          o->DV_avg1[X2(i, j)] = 0.0;
          o->DV_avg2[X2(i, j)] = cff2 * DVom[X2(i, j)];
        }
    } else {
      cff1 = c->weight[0][iif - 1];
      cff2 = (8.0 / 12.0) * c->weight[1][iif] - (1.0 / 12.0) * c->weight[1][iif + 1];
      for (int j = JstrR; j <= JendR; j++) {
        for (int i = IstrR; i <= IendR; i++)
          o->Zt_avg1[X2(i, j)] = o->Zt_avg1[X2(i, j)] + cff1 * Z(i, j, krhs);
        for (int i = Istr; i <= IendR; i++) {
          o->DU_avg1[X2(i, j)] = o->DU_avg1[X2(i, j)] + cff1 * DUon[X2(i, j)];
          o->DU_avg2[X2(i, j)] = o->DU_avg2[X2(i, j)] + cff2 * DUon[X2(i, j)];
        }
      }
      for (int j = Jstr; j <= JendR; j++)
        for (int i = IstrR; i <= IendR; i++) {
          o->DV_avg1[X2(i, j)] = o->DV_avg1[X2(i, j)] + cff1 * DVom[X2(i, j)];
          o->DV_avg2[X2(i, j)] = o->DV_avg2[X2(i, j)] + cff2 * DVom[X2(i, j)];
        }
    }
  } else {
    if (iif == 1) cff2 = c->weight[1][iif];
    else cff2 = (5.0 / 12.0) * c->weight[1][iif];
    for (int j = JstrR; j <= JendR; j++)
      for (int i = Istr; i <= IendR; i++)
        o->DU_avg2[X2(i, j)] = o->DU_avg2[X2(i, j)] + cff2 * DUon[X2(i, j)];
    for (int j = Jstr; j <= JendR; j++)
      for (int i = IstrR; i <= IendR; i++)
        o->DV_avg2[X2(i, j)] = o->DV_avg2[X2(i, j)] + cff2 * DVom[X2(i, j)];
  }
  /* last (auxiliary) predictor call: finalise averages and return :821-883 */
  if (iif == c->nfast + 1 && PRED) {
    orc_exchange2d(o, b, 'r', o->Zt_avg1);
    orc_exchange2d(o, b, 'u', o->DU_avg1);
    orc_exchange2d(o, b, 'v', o->DV_avg1);
  }
  if (o->wet_dry) orc_wetdry_tile(o, tile);                              /* new wet/dry masks :863 */
  if (iif > c->nfast) { free(S); return; }

  /* free-surface step :886-1000 */
  fac = 1000.0 / rho0;
  if (iif == 1) {
    cff1 = dtfast;
    for (int j = JstrV - 1; j <= Jend; j++)
      for (int i = IstrU - 1; i <= Iend; i++) {
        rhs_zeta[X2(i, j)] = (DUon[X2(i, j)] - DUon[X2(i + 1, j)]) + (DVom[X2(i, j)] - DVom[X2(i, j + 1)]);
        zeta_new[X2(i, j)] = Z(i, j, kstp) + pm[X2(i, j)] * pn[X2(i, j)] * cff1 * rhs_zeta[X2(i, j)];
        if (msk) zeta_new[X2(i, j)] = zeta_new[X2(i, j)] * o->rmask[X2(i, j)];               /* :907,933 */
        Dnew[X2(i, j)] = zeta_new[X2(i, j)] + h[X2(i, j)];
        zwrk[X2(i, j)] = 0.5 * (Z(i, j, kstp) + zeta_new[X2(i, j)]);
        gzeta[X2(i, j)] = (fac + rhoS[X2(i, j)]) * zwrk[X2(i, j)];
        gzeta2[X2(i, j)] = gzeta[X2(i, j)] * zwrk[X2(i, j)];
        gzetaSA[X2(i, j)] = zwrk[X2(i, j)] * (rhoS[X2(i, j)] - rhoA[X2(i, j)]);
      }
  } else if (PRED) {
    cff1 = 2.0 * dtfast;
    cff4 = 4.0 / 25.0;
    cff5 = 1.0 - 2.0 * cff4;
    for (int j = JstrV - 1; j <= Jend; j++)
      for (int i = IstrU - 1; i <= Iend; i++) {
        rhs_zeta[X2(i, j)] = (DUon[X2(i, j)] - DUon[X2(i + 1, j)]) + (DVom[X2(i, j)] - DVom[X2(i, j + 1)]);
        zeta_new[X2(i, j)] = Z(i, j, kstp) + pm[X2(i, j)] * pn[X2(i, j)] * cff1 * rhs_zeta[X2(i, j)];
        if (msk) zeta_new[X2(i, j)] = zeta_new[X2(i, j)] * o->rmask[X2(i, j)];               /* :907,933 */
        Dnew[X2(i, j)] = zeta_new[X2(i, j)] + h[X2(i, j)];
        zwrk[X2(i, j)] = cff5 * Z(i, j, krhs) + cff4 * (Z(i, j, kstp) + zeta_new[X2(i, j)]);
        gzeta[X2(i, j)] = (fac + rhoS[X2(i, j)]) * zwrk[X2(i, j)];
        gzeta2[X2(i, j)] = gzeta[X2(i, j)] * zwrk[X2(i, j)];
        gzetaSA[X2(i, j)] = zwrk[X2(i, j)] * (rhoS[X2(i, j)] - rhoA[X2(i, j)]);
      }
  } else {
    cff1 = dtfast * 5.0 / 12.0;
    cff2 = dtfast * 8.0 / 12.0;
    cff3 = dtfast * 1.0 / 12.0;
    cff4 = 2.0 / 5.0;
    cff5 = 1.0 - cff4;
    for (int j = JstrV - 1; j <= Jend; j++)
      for (int i = IstrU - 1; i <= Iend; i++) {
        cff = cff1 * ((DUon[X2(i, j)] - DUon[X2(i + 1, j)]) + (DVom[X2(i, j)] - DVom[X2(i, j + 1)]));
        zeta_new[X2(i, j)] = Z(i, j, kstp) + pm[X2(i, j)] * pn[X2(i, j)] *
                                                 (cff + cff2 * rzeta[X2T(i, j, kstp)] -
                                                  cff3 * rzeta[X2T(i, j, ptsk)]);
        if (msk) zeta_new[X2(i, j)] = zeta_new[X2(i, j)] * o->rmask[X2(i, j)];               /* :964 */
        Dnew[X2(i, j)] = zeta_new[X2(i, j)] + h[X2(i, j)];
        zwrk[X2(i, j)] = cff5 * zeta_new[X2(i, j)] + cff4 * Z(i, j, krhs);
        gzeta[X2(i, j)] = (fac + rhoS[X2(i, j)]) * zwrk[X2(i, j)];
        gzeta2[X2(i, j)] = gzeta[X2(i, j)] * zwrk[X2(i, j)];
        gzetaSA[X2(i, j)] = zwrk[X2(i, j)] * (rhoS[X2(i, j)] - rhoA[X2(i, j)]);
      }
  }
  for (int j = Jstr; j <= Jend; j++)
    for (int i = Istr; i <= Iend; i++) {
      Z(i, j, knew) = zeta_new[X2(i, j)];
      if (o->wet_dry && msk) Z(i, j, knew) = Z(i, j, knew) + (o->Dcrit - h[X2(i, j)]) * (1.0 - o->rmask[X2(i, j)]);   /* :992 */
    }
  if (PRED) {
    for (int j = Jstr; j <= Jend; j++)
      for (int i = Istr; i <= Iend; i++) rzeta[X2T(i, j, krhs)] = rhs_zeta[X2(i, j)];
    orc_exchange2d(o, b, 'r', rzeta + (size_t)(krhs - 1) * nij);
  }
  orc_zetabc(o, b, knew);                                              /* :1057 */
  orc_exchange2d(o, b, 'r', zeta + (size_t)(knew - 1) * nij);         /* :1068 */

  /* pressure gradient with VAR_RHO_2D :1080-1200 */
  cff1 = 0.5 * g;
  cff2 = 1.0 / 3.0;
  for (int j = Jstr; j <= Jend; j++) {
    for (int i = IstrU; i <= Iend; i++)
      rhs_ubar[X2(i, j)] =
          cff1 * on_u[X2(i, j)] *
          ((h[X2(i - 1, j)] + h[X2(i, j)]) * (gzeta[X2(i - 1, j)] - gzeta[X2(i, j)]) +
           (h[X2(i - 1, j)] - h[X2(i, j)]) *
               (gzetaSA[X2(i - 1, j)] + gzetaSA[X2(i, j)] +
                cff2 * (rhoA[X2(i - 1, j)] - rhoA[X2(i, j)]) * (zwrk[X2(i - 1, j)] - zwrk[X2(i, j)])) +
           (gzeta2[X2(i - 1, j)] - gzeta2[X2(i, j)]));
    if (d) for (int i = IstrU; i <= Iend; i++) DU2(U2rhs, i, j, d->M2pgrd) = rhs_ubar[X2(i, j)];       /* :1122 */
    if (j >= JstrV)
      for (int i = Istr; i <= Iend; i++)
        rhs_vbar[X2(i, j)] =
            cff1 * om_v[X2(i, j)] *
            ((h[X2(i, j - 1)] + h[X2(i, j)]) * (gzeta[X2(i, j - 1)] - gzeta[X2(i, j)]) +
             (h[X2(i, j - 1)] - h[X2(i, j)]) *
                 (gzetaSA[X2(i, j - 1)] + gzetaSA[X2(i, j)] +
                  cff2 * (rhoA[X2(i, j - 1)] - rhoA[X2(i, j)]) * (zwrk[X2(i, j - 1)] - zwrk[X2(i, j)])) +
             (gzeta2[X2(i, j - 1)] - gzeta2[X2(i, j)]));
    if (d && j >= JstrV) for (int i = Istr; i <= Iend; i++) DU2(V2rhs, i, j, d->M2pgrd) = rhs_vbar[X2(i, j)];   /* :1180 */
  }

  if (c->options & ORC_UV_ADV) {
    /* 4th-order centred advection :1249-1425 */
    for (int j = Jstr; j <= Jend; j++)
      for (int i = b->IstrUm1; i <= b->Iendp1; i++) {
        grad[X2(i, j)] = UB(i - 1, j, krhs) - 2.0 * UB(i, j, krhs) + UB(i + 1, j, krhs);
        Dgrad[X2(i, j)] = DUon[X2(i - 1, j)] - 2.0 * DUon[X2(i, j)] + DUon[X2(i + 1, j)];
      }
    if (!c->EWperiodic) {
      if (b->west)
        for (int j = Jstr; j <= Jend; j++) {
          grad[X2(Istr, j)] = grad[X2(Istr + 1, j)];
          Dgrad[X2(Istr, j)] = Dgrad[X2(Istr + 1, j)];
        }
      if (b->east)
        for (int j = Jstr; j <= Jend; j++) {
          grad[X2(Iend + 1, j)] = grad[X2(Iend, j)];
          Dgrad[X2(Iend + 1, j)] = Dgrad[X2(Iend, j)];
        }
    }
    cff = 1.0 / 6.0;
    for (int j = Jstr; j <= Jend; j++)
      for (int i = IstrU - 1; i <= Iend; i++)
        UFx[X2(i, j)] = 0.25 *
                        (UB(i, j, krhs) + UB(i + 1, j, krhs) - cff * (grad[X2(i, j)] + grad[X2(i + 1, j)])) *
                        (DUon[X2(i, j)] + DUon[X2(i + 1, j)] - cff * (Dgrad[X2(i, j)] + Dgrad[X2(i + 1, j)]));
    for (int j = b->Jstrm1; j <= b->Jendp1; j++)
      for (int i = IstrU; i <= Iend; i++)
        grad[X2(i, j)] = UB(i, j - 1, krhs) - 2.0 * UB(i, j, krhs) + UB(i, j + 1, krhs);
    if (!c->NSperiodic) {
      if (b->south) for (int i = IstrU; i <= Iend; i++) grad[X2(i, Jstr - 1)] = grad[X2(i, Jstr)];
      if (b->north) for (int i = IstrU; i <= Iend; i++) grad[X2(i, Jend + 1)] = grad[X2(i, Jend)];
    }
    for (int j = Jstr; j <= Jend + 1; j++)
      for (int i = IstrU - 1; i <= Iend; i++)
        Dgrad[X2(i, j)] = DVom[X2(i - 1, j)] - 2.0 * DVom[X2(i, j)] + DVom[X2(i + 1, j)];
    for (int j = Jstr; j <= Jend + 1; j++)
      for (int i = IstrU; i <= Iend; i++)
        UFe[X2(i, j)] = 0.25 *
                        (UB(i, j, krhs) + UB(i, j - 1, krhs) - cff * (grad[X2(i, j)] + grad[X2(i, j - 1)])) *
                        (DVom[X2(i, j)] + DVom[X2(i - 1, j)] - cff * (Dgrad[X2(i, j)] + Dgrad[X2(i - 1, j)]));
    for (int j = JstrV; j <= Jend; j++)
      for (int i = b->Istrm1; i <= b->Iendp1; i++)
        grad[X2(i, j)] = VB(i - 1, j, krhs) - 2.0 * VB(i, j, krhs) + VB(i + 1, j, krhs);
    if (!c->EWperiodic) {
      if (b->west) for (int j = JstrV; j <= Jend; j++) grad[X2(Istr - 1, j)] = grad[X2(Istr, j)];
      if (b->east) for (int j = JstrV; j <= Jend; j++) grad[X2(Iend + 1, j)] = grad[X2(Iend, j)];
    }
    for (int j = JstrV - 1; j <= Jend; j++)
      for (int i = Istr; i <= Iend + 1; i++)
        Dgrad[X2(i, j)] = DUon[X2(i, j - 1)] - 2.0 * DUon[X2(i, j)] + DUon[X2(i, j + 1)];
    for (int j = JstrV; j <= Jend; j++)
      for (int i = Istr; i <= Iend + 1; i++)
        VFx[X2(i, j)] = 0.25 *
                        (VB(i, j, krhs) + VB(i - 1, j, krhs) - cff * (grad[X2(i, j)] + grad[X2(i - 1, j)])) *
                        (DUon[X2(i, j)] + DUon[X2(i, j - 1)] - cff * (Dgrad[X2(i, j)] + Dgrad[X2(i, j - 1)]));
    for (int j = b->JstrVm1; j <= b->Jendp1; j++)
      for (int i = Istr; i <= Iend; i++) {
        grad[X2(i, j)] = VB(i, j - 1, krhs) - 2.0 * VB(i, j, krhs) + VB(i, j + 1, krhs);
        Dgrad[X2(i, j)] = DVom[X2(i, j - 1)] - 2.0 * DVom[X2(i, j)] + DVom[X2(i, j + 1)];
      }
    if (!c->NSperiodic) {
      if (b->south)
        for (int i = Istr; i <= Iend; i++) {
          grad[X2(i, Jstr)] = grad[X2(i, Jstr + 1)];
          Dgrad[X2(i, Jstr)] = Dgrad[X2(i, Jstr + 1)];
        }
      if (b->north)
        for (int i = Istr; i <= Iend; i++) {
          grad[X2(i, Jend + 1)] = grad[X2(i, Jend)];
          Dgrad[X2(i, Jend + 1)] = Dgrad[X2(i, Jend)];
        }
    }
    for (int j = JstrV - 1; j <= Jend; j++)
      for (int i = Istr; i <= Iend; i++)
        VFe[X2(i, j)] = 0.25 *
                        (VB(i, j, krhs) + VB(i, j + 1, krhs) - cff * (grad[X2(i, j)] + grad[X2(i, j + 1)])) *
                        (DVom[X2(i, j)] + DVom[X2(i, j + 1)] - cff * (Dgrad[X2(i, j)] + Dgrad[X2(i, j + 1)]));
    for (int j = Jstr; j <= Jend; j++)
      for (int i = IstrU; i <= Iend; i++) {
        cff1 = UFx[X2(i, j)] - UFx[X2(i - 1, j)];
        cff2 = UFe[X2(i, j + 1)] - UFe[X2(i, j)];
        fac = cff1 + cff2;
        rhs_ubar[X2(i, j)] = rhs_ubar[X2(i, j)] - fac;
        if (d) { DU2(U2rhs, i, j, d->M2xadv) = -cff1; DU2(U2rhs, i, j, d->M2yadv) = -cff2; DU2(U2rhs, i, j, d->M2hadv) = -fac; }   /* :1405 */
      }
    for (int j = JstrV; j <= Jend; j++)
      for (int i = Istr; i <= Iend; i++) {
        cff1 = VFx[X2(i + 1, j)] - VFx[X2(i, j)];
        cff2 = VFe[X2(i, j)] - VFe[X2(i, j - 1)];
        fac = cff1 + cff2;
        rhs_vbar[X2(i, j)] = rhs_vbar[X2(i, j)] - fac;
        if (d) { DU2(V2rhs, i, j, d->M2xadv) = -cff1; DU2(V2rhs, i, j, d->M2yadv) = -cff2; DU2(V2rhs, i, j, d->M2hadv) = -fac; }   /* :1418 */
      }
  }

  if (c->options & ORC_UV_COR) {
    /* Coriolis :1429-1490 */
    for (int j = JstrV - 1; j <= Jend; j++)
      for (int i = IstrU - 1; i <= Iend; i++) {
        cff = 0.5 * Drhs[X2(i, j)] * o->fomn[X2(i, j)];
        UFx[X2(i, j)] = cff * (VB(i, j, krhs) + VB(i, j + 1, krhs));
        VFe[X2(i, j)] = cff * (UB(i, j, krhs) + UB(i + 1, j, krhs));
      }
    for (int j = Jstr; j <= Jend; j++)
      for (int i = IstrU; i <= Iend; i++) {
        fac1 = 0.5 * (UFx[X2(i, j)] + UFx[X2(i - 1, j)]);
        rhs_ubar[X2(i, j)] = rhs_ubar[X2(i, j)] + fac1;
        if (d) DU2(U2rhs, i, j, d->M2fcor) = fac1;                                    /* :1446 */
      }
    for (int j = JstrV; j <= Jend; j++)
      for (int i = Istr; i <= Iend; i++) {
        fac1 = 0.5 * (VFe[X2(i, j)] + VFe[X2(i, j - 1)]);
        rhs_vbar[X2(i, j)] = rhs_vbar[X2(i, j)] - fac1;
        if (d) DU2(V2rhs, i, j, d->M2fcor) = -fac1;                                   /* :1455 */
      }
  }

  if ((c->options & ORC_CURVGRID) && (c->options & ORC_UV_ADV)) {
    /* curvilinear metric terms :1494-1560 */
    for (int j = JstrV - 1; j <= Jend; j++)
      for (int i = IstrU - 1; i <= Iend; i++) {
        cff1 = 0.5 * (VB(i, j, krhs) + VB(i, j + 1, krhs));
        cff2 = 0.5 * (UB(i, j, krhs) + UB(i + 1, j, krhs));
        cff3 = cff1 * o->dndx[X2(i, j)];
        cff4 = cff2 * o->dmde[X2(i, j)];
        cff = Drhs[X2(i, j)] * (cff3 - cff4);
        UFx[X2(i, j)] = cff * cff1;
        VFe[X2(i, j)] = cff * cff2;
        if (d) {                                                                      /* :1529-1531 */
          cff = Drhs[X2(i, j)] * cff4;
          Uwrk[X2(i, j)] = -cff * cff1;
          Vwrk[X2(i, j)] = -cff * cff2;
        }
      }
    for (int j = Jstr; j <= Jend; j++)
      for (int i = IstrU; i <= Iend; i++) {
        fac1 = 0.5 * (UFx[X2(i, j)] + UFx[X2(i - 1, j)]);
        rhs_ubar[X2(i, j)] = rhs_ubar[X2(i, j)] + fac1;
        if (d) {                                                                      /* :1544-1547 */
          const double fac2 = 0.5 * (Uwrk[X2(i, j)] + Uwrk[X2(i - 1, j)]);
          DU2(U2rhs, i, j, d->M2xadv) = DU2(U2rhs, i, j, d->M2xadv) + fac1 - fac2;
          DU2(U2rhs, i, j, d->M2yadv) = DU2(U2rhs, i, j, d->M2yadv) + fac2;
          DU2(U2rhs, i, j, d->M2hadv) = DU2(U2rhs, i, j, d->M2hadv) + fac1;
        }
      }
    for (int j = JstrV; j <= Jend; j++)
      for (int i = Istr; i <= Iend; i++) {
        fac1 = 0.5 * (VFe[X2(i, j)] + VFe[X2(i, j - 1)]);
        rhs_vbar[X2(i, j)] = rhs_vbar[X2(i, j)] - fac1;
        if (d) {                                                                      /* :1556-1559 */
          const double fac2 = 0.5 * (Vwrk[X2(i, j)] + Vwrk[X2(i, j - 1)]);
          DU2(V2rhs, i, j, d->M2xadv) = DU2(V2rhs, i, j, d->M2xadv) - fac1 + fac2;
          DU2(V2rhs, i, j, d->M2yadv) = DU2(V2rhs, i, j, d->M2yadv) - fac2;
          DU2(V2rhs, i, j, d->M2hadv) = DU2(V2rhs, i, j, d->M2hadv) - fac1;
        }
      }
  }

  if (c->options & ORC_UV_VIS2) {
    /* harmonic viscosity :1567-1660 */
    double *om_r = o->om_r, *on_r = o->on_r, *om_p = o->om_p, *on_p = o->on_p;
    for (int j = Jstr; j <= Jend + 1; j++)
      for (int i = Istr; i <= Iend + 1; i++)
        Drhs_p[X2(i, j)] = 0.25 * (Drhs[X2(i, j)] + Drhs[X2(i - 1, j)] + Drhs[X2(i, j - 1)] +
                                   Drhs[X2(i - 1, j - 1)]);
    for (int j = JstrV - 1; j <= Jend; j++)
      for (int i = IstrU - 1; i <= Iend; i++) {
        cff = o->visc2_r[X2(i, j)] * Drhs[X2(i, j)] * 0.5 *
              (o->pmon_r[X2(i, j)] * ((pn[X2(i, j)] + pn[X2(i + 1, j)]) * UB(i + 1, j, krhs) -
                                      (pn[X2(i - 1, j)] + pn[X2(i, j)]) * UB(i, j, krhs)) -
               o->pnom_r[X2(i, j)] * ((pm[X2(i, j)] + pm[X2(i, j + 1)]) * VB(i, j + 1, krhs) -
                                      (pm[X2(i, j - 1)] + pm[X2(i, j)]) * VB(i, j, krhs)));
        UFx[X2(i, j)] = on_r[X2(i, j)] * on_r[X2(i, j)] * cff;
        VFe[X2(i, j)] = om_r[X2(i, j)] * om_r[X2(i, j)] * cff;
      }
    for (int j = Jstr; j <= Jend + 1; j++)
      for (int i = Istr; i <= Iend + 1; i++) {
        cff = o->visc2_p[X2(i, j)] * Drhs_p[X2(i, j)] * 0.5 *
              (o->pmon_p[X2(i, j)] * ((pn[X2(i, j - 1)] + pn[X2(i, j)]) * VB(i, j, krhs) -
                                      (pn[X2(i - 1, j - 1)] + pn[X2(i - 1, j)]) * VB(i - 1, j, krhs)) +
               o->pnom_p[X2(i, j)] * ((pm[X2(i - 1, j)] + pm[X2(i, j)]) * UB(i, j, krhs) -
                                      (pm[X2(i - 1, j - 1)] + pm[X2(i, j - 1)]) * UB(i, j - 1, krhs)));
        if (msk) cff = cff * o->pmask[X2(i, j)];                                            /* :1613 */
        if (o->wet_dry) cff = cff * o->pmask_wet[X2(i, j)];                                 /* :1617 */
        UFe[X2(i, j)] = om_p[X2(i, j)] * om_p[X2(i, j)] * cff;
        VFx[X2(i, j)] = on_p[X2(i, j)] * on_p[X2(i, j)] * cff;
      }
    for (int j = Jstr; j <= Jend; j++)
      for (int i = IstrU; i <= Iend; i++) {
        cff1 = 0.5 * (pn[X2(i - 1, j)] + pn[X2(i, j)]) * (UFx[X2(i, j)] - UFx[X2(i - 1, j)]);
        cff2 = 0.5 * (pm[X2(i - 1, j)] + pm[X2(i, j)]) * (UFe[X2(i, j + 1)] - UFe[X2(i, j)]);
        fac = cff1 + cff2;
        rhs_ubar[X2(i, j)] = rhs_ubar[X2(i, j)] + fac;
        if (d) { DU2(U2rhs, i, j, d->M2hvis) = fac; DU2(U2rhs, i, j, d->M2xvis) = cff1; DU2(U2rhs, i, j, d->M2yvis) = cff2; }      /* :1633 */
      }
    for (int j = JstrV; j <= Jend; j++)
      for (int i = Istr; i <= Iend; i++) {
        cff1 = 0.5 * (pn[X2(i, j - 1)] + pn[X2(i, j)]) * (VFx[X2(i + 1, j)] - VFx[X2(i, j)]);
        cff2 = 0.5 * (pm[X2(i, j - 1)] + pm[X2(i, j)]) * (VFe[X2(i, j)] - VFe[X2(i, j - 1)]);
        fac = cff1 - cff2;
        rhs_vbar[X2(i, j)] = rhs_vbar[X2(i, j)] + fac;
        if (d) { DU2(V2rhs, i, j, d->M2hvis) = fac; DU2(V2rhs, i, j, d->M2xvis) = cff1; DU2(V2rhs, i, j, d->M2yvis) = -cff2; }     /* :1646 */
      }
  }

  if (o->uv_vis4) orc_step2d_vis4(o, b, krhs, Drhs, rhs_ubar, rhs_vbar, U2rhs, V2rhs);   /* UV_VIS4 :1653-1920 (orc_mix4.c) */

  /* nudging of the 2-D momentum towards its climatology, LnudgeM2CLM :2179-2203 */
  if (o->clima_flags & 32) {
    for (int j = Jstr; j <= Jend; j++)
      for (int i = IstrU; i <= Iend; i++) {
        cff = 0.25 * (o->M2nudgcof[X2(i - 1, j)] + o->M2nudgcof[X2(i, j)]) * o->om_u[X2(i, j)] * o->on_u[X2(i, j)];
        rhs_ubar[X2(i, j)] = rhs_ubar[X2(i, j)] + cff * (Drhs[X2(i - 1, j)] + Drhs[X2(i, j)]) * (o->ubarclm[X2(i, j)] - ubar[X2T(i, j, krhs)]);
      }
    for (int j = JstrV; j <= Jend; j++)
      for (int i = Istr; i <= Iend; i++) {
        cff = 0.25 * (o->M2nudgcof[X2(i, j - 1)] + o->M2nudgcof[X2(i, j)]) * o->om_v[X2(i, j)] * o->on_v[X2(i, j)];
        rhs_vbar[X2(i, j)] = rhs_vbar[X2(i, j)] + cff * (Drhs[X2(i, j - 1)] + Drhs[X2(i, j)]) * (o->vbarclm[X2(i, j)] - vbar[X2T(i, j, krhs)]);
      }
  }

  if (o->wet_dry) {                                                      /* :2205-2222 */
    for (int j = Jstr; j <= Jend; j++)
      for (int i = IstrU; i <= Iend; i++) rhs_ubar[X2(i, j)] = rhs_ubar[X2(i, j)] * orc_wd_fac(o->umask_wet[X2(i, j)], rhs_ubar[X2(i, j)]);
    for (int j = JstrV; j <= Jend; j++)
      for (int i = Istr; i <= Iend; i++) rhs_vbar[X2(i, j)] = rhs_vbar[X2(i, j)] * orc_wd_fac(o->vmask_wet[X2(i, j)], rhs_vbar[X2(i, j)]);
  }
  /* coupling with the 3-D momentum forcing :2225-2460 */
  if (iif == 1 && PRED) {
    if (iic == c->ntfirst) {
      for (int j = Jstr; j <= Jend; j++)
        for (int i = IstrU; i <= Iend; i++) {
          rufrc[X2(i, j)] = rufrc[X2(i, j)] - rhs_ubar[X2(i, j)];
          rhs_ubar[X2(i, j)] = rhs_ubar[X2(i, j)] + rufrc[X2(i, j)];
          ru[XW4(i, j, 0, nstp)] = rufrc[X2(i, j)];
          if (d) {
            for (int id = 1; id <= d->M2pgrd; id++) {
              DUF(d->RUfrc, i, j, 3, id) = DUF(d->RUfrc, i, j, 3, id) - DU2(U2rhs, i, j, id);
              DU2(U2rhs, i, j, id) = DU2(U2rhs, i, j, id) + DUF(d->RUfrc, i, j, 3, id);
              DUF(d->RUfrc, i, j, nstp, id) = DUF(d->RUfrc, i, j, 3, id);
            }
            DU2(U2rhs, i, j, d->M2sstr) = DUF(d->RUfrc, i, j, 3, d->M2sstr);
            DUF(d->RUfrc, i, j, nstp, d->M2sstr) = DUF(d->RUfrc, i, j, 3, d->M2sstr);
            DU2(U2rhs, i, j, d->M2bstr) = DUF(d->RUfrc, i, j, 3, d->M2bstr);
            DUF(d->RUfrc, i, j, nstp, d->M2bstr) = DUF(d->RUfrc, i, j, 3, d->M2bstr);
          }
        }
      for (int j = JstrV; j <= Jend; j++)
        for (int i = Istr; i <= Iend; i++) {
          rvfrc[X2(i, j)] = rvfrc[X2(i, j)] - rhs_vbar[X2(i, j)];
          rhs_vbar[X2(i, j)] = rhs_vbar[X2(i, j)] + rvfrc[X2(i, j)];
          rv[XW4(i, j, 0, nstp)] = rvfrc[X2(i, j)];
          if (d) {
            for (int id = 1; id <= d->M2pgrd; id++) {
              DUF(d->RVfrc, i, j, 3, id) = DUF(d->RVfrc, i, j, 3, id) - DU2(V2rhs, i, j, id);
              DU2(V2rhs, i, j, id) = DU2(V2rhs, i, j, id) + DUF(d->RVfrc, i, j, 3, id);
              DUF(d->RVfrc, i, j, nstp, id) = DUF(d->RVfrc, i, j, 3, id);
            }
            DU2(V2rhs, i, j, d->M2sstr) = DUF(d->RVfrc, i, j, 3, d->M2sstr);
            DUF(d->RVfrc, i, j, nstp, d->M2sstr) = DUF(d->RVfrc, i, j, 3, d->M2sstr);
            DU2(V2rhs, i, j, d->M2bstr) = DUF(d->RVfrc, i, j, 3, d->M2bstr);
            DUF(d->RVfrc, i, j, nstp, d->M2bstr) = DUF(d->RVfrc, i, j, 3, d->M2bstr);
          }
        }
    } else if (iic == c->ntfirst + 1) {
      for (int j = Jstr; j <= Jend; j++)
        for (int i = IstrU; i <= Iend; i++) {
          rufrc[X2(i, j)] = rufrc[X2(i, j)] - rhs_ubar[X2(i, j)];
          rhs_ubar[X2(i, j)] = rhs_ubar[X2(i, j)] + 1.5 * rufrc[X2(i, j)] - 0.5 * ru[XW4(i, j, 0, nnew)];
          ru[XW4(i, j, 0, nstp)] = rufrc[X2(i, j)];
          if (d) {
            for (int id = 1; id <= d->M2pgrd; id++) {
              DUF(d->RUfrc, i, j, 3, id) = DUF(d->RUfrc, i, j, 3, id) - DU2(U2rhs, i, j, id);
              DU2(U2rhs, i, j, id) = DU2(U2rhs, i, j, id) + 1.5 * DUF(d->RUfrc, i, j, 3, id) - 0.5 * DUF(d->RUfrc, i, j, nnew, id);
              DUF(d->RUfrc, i, j, nstp, id) = DUF(d->RUfrc, i, j, 3, id);
            }
            DU2(U2rhs, i, j, d->M2sstr) = 1.5 * DUF(d->RUfrc, i, j, 3, d->M2sstr) - 0.5 * DUF(d->RUfrc, i, j, nnew, d->M2sstr);
            DUF(d->RUfrc, i, j, nstp, d->M2sstr) = DUF(d->RUfrc, i, j, 3, d->M2sstr);
            DU2(U2rhs, i, j, d->M2bstr) = 1.5 * DUF(d->RUfrc, i, j, 3, d->M2bstr) - 0.5 * DUF(d->RUfrc, i, j, nnew, d->M2bstr);
            DUF(d->RUfrc, i, j, nstp, d->M2bstr) = DUF(d->RUfrc, i, j, 3, d->M2bstr);
          }
        }
      for (int j = JstrV; j <= Jend; j++)
        for (int i = Istr; i <= Iend; i++) {
          rvfrc[X2(i, j)] = rvfrc[X2(i, j)] - rhs_vbar[X2(i, j)];
          rhs_vbar[X2(i, j)] = rhs_vbar[X2(i, j)] + 1.5 * rvfrc[X2(i, j)] - 0.5 * rv[XW4(i, j, 0, nnew)];
          rv[XW4(i, j, 0, nstp)] = rvfrc[X2(i, j)];
          if (d) {
            for (int id = 1; id <= d->M2pgrd; id++) {
              DUF(d->RVfrc, i, j, 3, id) = DUF(d->RVfrc, i, j, 3, id) - DU2(V2rhs, i, j, id);
              DU2(V2rhs, i, j, id) = DU2(V2rhs, i, j, id) + 1.5 * DUF(d->RVfrc, i, j, 3, id) - 0.5 * DUF(d->RVfrc, i, j, nnew, id);
              DUF(d->RVfrc, i, j, nstp, id) = DUF(d->RVfrc, i, j, 3, id);
            }
            DU2(V2rhs, i, j, d->M2sstr) = 1.5 * DUF(d->RVfrc, i, j, 3, d->M2sstr) - 0.5 * DUF(d->RVfrc, i, j, nnew, d->M2sstr);
            DUF(d->RVfrc, i, j, nstp, d->M2sstr) = DUF(d->RVfrc, i, j, 3, d->M2sstr);
            DU2(V2rhs, i, j, d->M2bstr) = 1.5 * DUF(d->RVfrc, i, j, 3, d->M2bstr) - 0.5 * DUF(d->RVfrc, i, j, nnew, d->M2bstr);
            DUF(d->RVfrc, i, j, nstp, d->M2bstr) = DUF(d->RVfrc, i, j, 3, d->M2bstr);
          }
        }
    } else {
      cff1 = 23.0 / 12.0;
      cff2 = 16.0 / 12.0;
      cff3 = 5.0 / 12.0;
      for (int j = Jstr; j <= Jend; j++)
        for (int i = IstrU; i <= Iend; i++) {
          rufrc[X2(i, j)] = rufrc[X2(i, j)] - rhs_ubar[X2(i, j)];
          rhs_ubar[X2(i, j)] = rhs_ubar[X2(i, j)] + cff1 * rufrc[X2(i, j)] -
                               cff2 * ru[XW4(i, j, 0, nnew)] + cff3 * ru[XW4(i, j, 0, nstp)];
          ru[XW4(i, j, 0, nstp)] = rufrc[X2(i, j)];
          if (d) {
            for (int id = 1; id <= d->M2pgrd; id++) {
              DUF(d->RUfrc, i, j, 3, id) = DUF(d->RUfrc, i, j, 3, id) - DU2(U2rhs, i, j, id);
              DU2(U2rhs, i, j, id) = DU2(U2rhs, i, j, id) + cff1 * DUF(d->RUfrc, i, j, 3, id) - cff2 * DUF(d->RUfrc, i, j, nnew, id) + cff3 * DUF(d->RUfrc, i, j, nstp, id);
              DUF(d->RUfrc, i, j, nstp, id) = DUF(d->RUfrc, i, j, 3, id);
            }
            DU2(U2rhs, i, j, d->M2sstr) = cff1 * DUF(d->RUfrc, i, j, 3, d->M2sstr) - cff2 * DUF(d->RUfrc, i, j, nnew, d->M2sstr) + cff3 * DUF(d->RUfrc, i, j, nstp, d->M2sstr);
            DUF(d->RUfrc, i, j, nstp, d->M2sstr) = DUF(d->RUfrc, i, j, 3, d->M2sstr);
            DU2(U2rhs, i, j, d->M2bstr) = cff1 * DUF(d->RUfrc, i, j, 3, d->M2bstr) - cff2 * DUF(d->RUfrc, i, j, nnew, d->M2bstr) + cff3 * DUF(d->RUfrc, i, j, nstp, d->M2bstr);
            DUF(d->RUfrc, i, j, nstp, d->M2bstr) = DUF(d->RUfrc, i, j, 3, d->M2bstr);
          }
        }
      for (int j = JstrV; j <= Jend; j++)
        for (int i = Istr; i <= Iend; i++) {
          rvfrc[X2(i, j)] = rvfrc[X2(i, j)] - rhs_vbar[X2(i, j)];
          rhs_vbar[X2(i, j)] = rhs_vbar[X2(i, j)] + cff1 * rvfrc[X2(i, j)] -
                               cff2 * rv[XW4(i, j, 0, nnew)] + cff3 * rv[XW4(i, j, 0, nstp)];
          rv[XW4(i, j, 0, nstp)] = rvfrc[X2(i, j)];
          if (d) {
            for (int id = 1; id <= d->M2pgrd; id++) {
              DUF(d->RVfrc, i, j, 3, id) = DUF(d->RVfrc, i, j, 3, id) - DU2(V2rhs, i, j, id);
              DU2(V2rhs, i, j, id) = DU2(V2rhs, i, j, id) + cff1 * DUF(d->RVfrc, i, j, 3, id) - cff2 * DUF(d->RVfrc, i, j, nnew, id) + cff3 * DUF(d->RVfrc, i, j, nstp, id);
              DUF(d->RVfrc, i, j, nstp, id) = DUF(d->RVfrc, i, j, 3, id);
            }
            DU2(V2rhs, i, j, d->M2sstr) = cff1 * DUF(d->RVfrc, i, j, 3, d->M2sstr) - cff2 * DUF(d->RVfrc, i, j, nnew, d->M2sstr) + cff3 * DUF(d->RVfrc, i, j, nstp, d->M2sstr);
            DUF(d->RVfrc, i, j, nstp, d->M2sstr) = DUF(d->RVfrc, i, j, 3, d->M2sstr);
            DU2(V2rhs, i, j, d->M2bstr) = cff1 * DUF(d->RVfrc, i, j, 3, d->M2bstr) - cff2 * DUF(d->RVfrc, i, j, nnew, d->M2bstr) + cff3 * DUF(d->RVfrc, i, j, nstp, d->M2bstr);
            DUF(d->RVfrc, i, j, nstp, d->M2bstr) = DUF(d->RVfrc, i, j, 3, d->M2bstr);
          }
        }
    }
  } else {
    for (int j = Jstr; j <= Jend; j++)
      for (int i = IstrU; i <= Iend; i++) {
        rhs_ubar[X2(i, j)] = rhs_ubar[X2(i, j)] + rufrc[X2(i, j)];
        if (d) {                                                                      /* :2430-2435 */
          for (int id = 1; id <= d->M2pgrd; id++) DU2(U2rhs, i, j, id) = DU2(U2rhs, i, j, id) + DUF(d->RUfrc, i, j, 3, id);
          DU2(U2rhs, i, j, d->M2sstr) = DUF(d->RUfrc, i, j, 3, d->M2sstr);
          DU2(U2rhs, i, j, d->M2bstr) = DUF(d->RUfrc, i, j, 3, d->M2bstr);
        }
      }
    for (int j = JstrV; j <= Jend; j++)
      for (int i = Istr; i <= Iend; i++) {
        rhs_vbar[X2(i, j)] = rhs_vbar[X2(i, j)] + rvfrc[X2(i, j)];
        if (d) {                                                                      /* :2447-2452 */
          for (int id = 1; id <= d->M2pgrd; id++) DU2(V2rhs, i, j, id) = DU2(V2rhs, i, j, id) + DUF(d->RVfrc, i, j, 3, id);
          DU2(V2rhs, i, j, d->M2sstr) = DUF(d->RVfrc, i, j, 3, d->M2sstr);
          DU2(V2rhs, i, j, d->M2bstr) = DUF(d->RVfrc, i, j, 3, d->M2bstr);
        }
      }
  }

  /* momentum time step :2488-2670 */
  for (int j = JstrV - 1; j <= Jend; j++)
    for (int i = IstrU - 1; i <= Iend; i++) Dstp[X2(i, j)] = Z(i, j, kstp) + h[X2(i, j)];
  if (iif == 1 || PRED) {
    cff1 = (iif == 1) ? 0.5 * dtfast : dtfast;
    for (int j = Jstr; j <= Jend; j++)
      for (int i = IstrU; i <= Iend; i++) {
        cff = (pm[X2(i, j)] + pm[X2(i - 1, j)]) * (pn[X2(i, j)] + pn[X2(i - 1, j)]);
        fac = 1.0 / (Dnew[X2(i, j)] + Dnew[X2(i - 1, j)]);
        UB(i, j, knew) = (UB(i, j, kstp) * (Dstp[X2(i, j)] + Dstp[X2(i - 1, j)]) +
                          cff * cff1 * rhs_ubar[X2(i, j)]) * fac;
        if (msk) UB(i, j, knew) = UB(i, j, knew) * o->umask[X2(i, j)];                      /* :2515,2578 */
        if (o->wet_dry) {                                                                   /* :2518-2529, 2581-2587 */
          const double cff7 = orc_wd_fac(o->umask_wet[X2(i, j)], UB(i, j, knew));
          UB(i, j, knew) = UB(i, j, knew) * cff7;
          rhs_ubar[X2(i, j)] = rhs_ubar[X2(i, j)] * cff7;
          if (iif == 1 && PRED) { rufrc[X2(i, j)] = rufrc[X2(i, j)] * cff7; ru[XW4(i, j, 0, nstp)] = rufrc[X2(i, j)]; }
        }
      }
    for (int j = JstrV; j <= Jend; j++)
      for (int i = Istr; i <= Iend; i++) {
        cff = (pm[X2(i, j)] + pm[X2(i, j - 1)]) * (pn[X2(i, j)] + pn[X2(i, j - 1)]);
        fac = 1.0 / (Dnew[X2(i, j)] + Dnew[X2(i, j - 1)]);
        VB(i, j, knew) = (VB(i, j, kstp) * (Dstp[X2(i, j)] + Dstp[X2(i, j - 1)]) +
                          cff * cff1 * rhs_vbar[X2(i, j)]) * fac;
        if (msk) VB(i, j, knew) = VB(i, j, knew) * o->vmask[X2(i, j)];                      /* :2544,2601 */
        if (o->wet_dry) {                                                                   /* :2547-2558, 2604-2610 */
          const double cff7 = orc_wd_fac(o->vmask_wet[X2(i, j)], VB(i, j, knew));
          VB(i, j, knew) = VB(i, j, knew) * cff7;
          rhs_vbar[X2(i, j)] = rhs_vbar[X2(i, j)] * cff7;
          if (iif == 1 && PRED) { rvfrc[X2(i, j)] = rvfrc[X2(i, j)] * cff7; rv[XW4(i, j, 0, nstp)] = rvfrc[X2(i, j)]; }
        }
      }
  } else if (CORR) {
    cff1 = 0.5 * dtfast * 5.0 / 12.0;
    cff2 = 0.5 * dtfast * 8.0 / 12.0;
    cff3 = 0.5 * dtfast * 1.0 / 12.0;
    for (int j = Jstr; j <= Jend; j++)
      for (int i = IstrU; i <= Iend; i++) {
        cff = (pm[X2(i, j)] + pm[X2(i - 1, j)]) * (pn[X2(i, j)] + pn[X2(i - 1, j)]);
        fac = 1.0 / (Dnew[X2(i, j)] + Dnew[X2(i - 1, j)]);
        UB(i, j, knew) = (UB(i, j, kstp) * (Dstp[X2(i, j)] + Dstp[X2(i - 1, j)]) +
                          cff * (cff1 * rhs_ubar[X2(i, j)] + cff2 * rubar[X2T(i, j, kstp)] -
                                 cff3 * rubar[X2T(i, j, ptsk)])) * fac;
        if (msk) UB(i, j, knew) = UB(i, j, knew) * o->umask[X2(i, j)];                      /* :2633 */
        if (o->wet_dry) {                                                                   /* :2636-2642 */
          const double cff7 = orc_wd_fac(o->umask_wet[X2(i, j)], UB(i, j, knew));
          UB(i, j, knew) = UB(i, j, knew) * cff7;
          rhs_ubar[X2(i, j)] = rhs_ubar[X2(i, j)] * cff7;
        }
      }
    for (int j = JstrV; j <= Jend; j++)
      for (int i = Istr; i <= Iend; i++) {
        cff = (pm[X2(i, j)] + pm[X2(i, j - 1)]) * (pn[X2(i, j)] + pn[X2(i, j - 1)]);
        fac = 1.0 / (Dnew[X2(i, j)] + Dnew[X2(i, j - 1)]);
        VB(i, j, knew) = (VB(i, j, kstp) * (Dstp[X2(i, j)] + Dstp[X2(i, j - 1)]) +
                          cff * (cff1 * rhs_vbar[X2(i, j)] + cff2 * rvbar[X2T(i, j, kstp)] -
                                 cff3 * rvbar[X2T(i, j, ptsk)])) * fac;
        if (msk) VB(i, j, knew) = VB(i, j, knew) * o->vmask[X2(i, j)];                      /* :2658 */
        if (o->wet_dry) {                                                                   /* :2661-2667 */
          const double cff7 = orc_wd_fac(o->vmask_wet[X2(i, j)], VB(i, j, knew));
          VB(i, j, knew) = VB(i, j, knew) * cff7;
          rhs_vbar[X2(i, j)] = rhs_vbar[X2(i, j)] * cff7;
        }
      }
  }
  if (d) {
    /* "Time step 2D momentum diagnostic terms" :2676-2743 (SOLVE3D): integrated over the fast steps with the corrector's
       weights, converted to mass-flux units and averaged with weight(1,iif) for the coupling with the 3-D terms */
    const int nd = d->NDM2d - 1;
    if (msk)
      for (int id = 1; id <= nd; id++) {
        for (int j = Jstr; j <= Jend; j++)
          for (int i = IstrU; i <= Iend; i++) DU2(U2rhs, i, j, id) = DU2(U2rhs, i, j, id) * o->umask[X2(i, j)];
        for (int j = JstrV; j <= Jend; j++)
          for (int i = Istr; i <= Iend; i++) DU2(V2rhs, i, j, id) = DU2(V2rhs, i, j, id) * o->vmask[X2(i, j)];
      }
    fac = c->weight[0][iif];
    if (iif == 1 && CORR) {
      cff1 = 0.5 * dtfast;
      for (int id = 1; id <= nd; id++) {
        for (int j = Jstr; j <= Jend; j++)
          for (int i = IstrU; i <= Iend; i++) {
            DU2(d->U2int, i, j, id) = cff1 * DU2(U2rhs, i, j, id);
            DU2(d->U2wrk, i, j, id) = DU2(d->U2int, i, j, id) * (pm[X2(i - 1, j)] + pm[X2(i, j)]) * fac;
          }
        for (int j = JstrV; j <= Jend; j++)
          for (int i = Istr; i <= Iend; i++) {
            DU2(d->V2int, i, j, id) = cff1 * DU2(V2rhs, i, j, id);
            DU2(d->V2wrk, i, j, id) = DU2(d->V2int, i, j, id) * (pn[X2(i, j)] + pn[X2(i, j - 1)]) * fac;
          }
      }
    } else if (CORR) {
      cff1 = 0.5 * dtfast * 5.0 / 12.0;
      cff2 = 0.5 * dtfast * 8.0 / 12.0;
      cff3 = 0.5 * dtfast * 1.0 / 12.0;
      for (int id = 1; id <= nd; id++) {
        for (int j = Jstr; j <= Jend; j++)
          for (int i = IstrU; i <= Iend; i++) {
            DU2(d->U2int, i, j, id) = DU2(d->U2int, i, j, id) +
                                      (cff1 * DU2(U2rhs, i, j, id) + cff2 * DUB(d->RUbar, i, j, kstp, id) -
                                       cff3 * DUB(d->RUbar, i, j, ptsk, id));
            DU2(d->U2wrk, i, j, id) = DU2(d->U2wrk, i, j, id) + DU2(d->U2int, i, j, id) * (pm[X2(i - 1, j)] + pm[X2(i, j)]) * fac;
          }
        for (int j = JstrV; j <= Jend; j++)
          for (int i = Istr; i <= Iend; i++) {
            DU2(d->V2int, i, j, id) = DU2(d->V2int, i, j, id) +
                                      (cff1 * DU2(V2rhs, i, j, id) + cff2 * DUB(d->RVbar, i, j, kstp, id) -
                                       cff3 * DUB(d->RVbar, i, j, ptsk, id));
            DU2(d->V2wrk, i, j, id) = DU2(d->V2wrk, i, j, id) + DU2(d->V2int, i, j, id) * (pn[X2(i, j)] + pn[X2(i, j - 1)]) * fac;
          }
      }
    }
    if (PRED)                                                                         /* :2852-2863 */
      for (int id = 1; id <= nd; id++) {
        for (int j = Jstr; j <= Jend; j++)
          for (int i = IstrU; i <= Iend; i++) DUB(d->RUbar, i, j, krhs, id) = DU2(U2rhs, i, j, id);
        for (int j = JstrV; j <= Jend; j++)
          for (int i = Istr; i <= Iend; i++) DUB(d->RVbar, i, j, krhs, id) = DU2(V2rhs, i, j, id);
      }
  }
  if (PRED) {
    for (int j = Jstr; j <= Jend; j++)
      for (int i = IstrU; i <= Iend; i++) rubar[X2T(i, j, krhs)] = rhs_ubar[X2(i, j)];
    for (int j = JstrV; j <= Jend; j++)
      for (int i = Istr; i <= Iend; i++) rvbar[X2T(i, j, krhs)] = rhs_vbar[X2(i, j)];
  }
  orc_u2dbc(o, b, knew);                                               /* :2871 */
  orc_v2dbc(o, b, knew);                                               /* :2876 */
  /* VolCons: cross-section and flux of the open edges, obc_flux_tile -- obc_volcons.F:60-233, called at step2d_LF_AM3.h:2885 with
     knew: this tile's sums in the reference's order (west, east, south, north; ascending index), added to the running sums in
     calling order; behind the last tile the correction velocity (the shared-memory form :192-227, NSUB tiles) */
  if (c->volcons) {
    double my_area = 0.0, my_flux = 0.0;
    if ((c->volcons & (1 << ORC_IWEST)) && b->west)
      for (int j = Jstr; j <= Jend; j++) {
        cff = 0.5 * (Z(Istr - 1, j, knew) + h[X2(Istr - 1, j)] + Z(Istr, j, knew) + h[X2(Istr, j)]) * on_u[X2(Istr, j)];
        if (msk) cff = cff * o->umask[X2(Istr, j)];
        my_area = my_area + cff;
        my_flux = my_flux + cff * UB(Istr, j, knew);
      }
    if ((c->volcons & (1 << ORC_IEAST)) && b->east)
      for (int j = Jstr; j <= Jend; j++) {
        cff = 0.5 * (Z(Iend, j, knew) + h[X2(Iend, j)] + Z(Iend + 1, j, knew) + h[X2(Iend + 1, j)]) * on_u[X2(Iend + 1, j)];
        if (msk) cff = cff * o->umask[X2(Iend + 1, j)];
        my_area = my_area + cff;
        my_flux = my_flux - cff * UB(Iend + 1, j, knew);
      }
    if ((c->volcons & (1 << ORC_ISOUTH)) && b->south)
      for (int i = Istr; i <= Iend; i++) {
        cff = 0.5 * (Z(i, Jstr - 1, knew) + h[X2(i, Jstr - 1)] + Z(i, Jstr, knew) + h[X2(i, Jstr)]) * om_v[X2(i, Jstr)];
        if (msk) cff = cff * o->vmask[X2(i, Jstr)];
        my_area = my_area + cff;
        my_flux = my_flux + cff * VB(i, JstrV - 1, knew);
      }
    if ((c->volcons & (1 << ORC_INORTH)) && b->north)
      for (int i = Istr; i <= Iend; i++) {
        cff = 0.5 * (Z(i, Jend, knew) + h[X2(i, Jend)] + Z(i, Jend + 1, knew) + h[X2(i, Jend + 1)]) * om_v[X2(i, Jend + 1)];
        if (msk) cff = cff * o->vmask[X2(i, Jend + 1)];
        my_area = my_area + cff;
        my_flux = my_flux - cff * VB(i, Jend + 1, knew);
      }
    if (o->vc_count == 0) { o->bc_flux = 0.0; o->bc_area = 0.0; }
    o->bc_area = o->bc_area + my_area;
    o->bc_flux = o->bc_flux + my_flux;
    o->vc_count = o->vc_count + 1;
    if (o->vc_count == c->NtileI * c->NtileJ) {
      o->vc_count = 0;
      o->ubar_xs = o->bc_flux / o->bc_area;
    }
  }
  orc_exchange2d(o, b, 'u', ubar + (size_t)(knew - 1) * nij);         /* :3043 */
  orc_exchange2d(o, b, 'v', vbar + (size_t)(knew - 1) * nij);
  free(S);
  free(U2rhs);
#undef Z
#undef UB
#undef VB
}
