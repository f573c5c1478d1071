/*
 * orc_wetdry.c -- WET_DRY: the time-dependent wet/dry masks.
 * TEST INFRASTRUCTURE (see orc.h).
 *
 * Follows ROMS/Nonlinear/wetdry.F (MASKING, SOLVE3D, no point sources):
 *   orc_wetdry_ini    wetdry_ini_tile        :355-490   (initial.F:467)
 *   orc_wetdry_tile   wetdry_tile            :93-351    (step2d_LF_AM3.h:863, every call of step2d)
 *   wd_mask           wetdry_mask_tile       :493-718
 *   wd_avg_mask       wetdry_avg_mask_tile   :723-900
 * The places that USE the masks carry their own references: orc_step2d.c, orc_rhs3d.c (prsgrd32, rhs3d_tile,
 * t3dmix2_s, uv3dmix2_s, pre_step3d), orc_step3d.c (step3d_uv), orc_diag3d.c (set_depth, set_vbc, ini_zeta, ini_fields).
 *
 * PARITY: pinned bit for bit against the reference built from oracle/ref/upwelling_wetdry.h (wetdry_ini, every mask
 * after each of the main3d passes, each routine on perturbed states): tests/test_oracle_vs_ref.py.
 */
#include "orc.h"
#include <math.h>
#include <stdlib.h>

void orc_set_wetdry(orc_t *o, double Dcrit) { o->wet_dry = 1; o->Dcrit = Dcrit; }

#define WD(i, j) wetdry[X2(i, j)]

/* PSI-point mask from the four surrounding rho values :545-600, :806-861 */
static double wd_psi(const double *wetdry, size_t ni, int LBi, int LBj, int i, int j) {
  const int a = WD(i - 1, j) > 0.5, b = WD(i, j) > 0.5, c = WD(i - 1, j - 1) > 0.5, d = WD(i, j - 1) > 0.5;
  const int al = WD(i - 1, j) < 0.5, bl = WD(i, j) < 0.5, cl = WD(i - 1, j - 1) < 0.5, dl = WD(i, j - 1) < 0.5;
  if (a && b && c && d) return 1.0;
  if (al && b && c && d) return 1.0;
  if (a && bl && c && d) return 1.0;
  if (a && b && cl && d) return 1.0;
  if (a && b && c && dl) return 1.0;
  if (a && bl && c && dl) return 2.0;
  if (al && b && cl && d) return 2.0;
  if (a && b && cl && dl) return 2.0;
  if (al && bl && c && d) return 2.0;
  return 0.0;
}

/* wetdry_mask_tile :493-718 */
static void wd_mask(orc_t *o, const orc_bounds *b, const double *wetdry) {
  ORC_LOCALS(o);
  for (int j = b->JstrR; j <= b->JendR; j++)
    for (int i = b->IstrR; i <= b->IendR; i++) o->rmask_wet[X2(i, j)] = WD(i, j);
  for (int j = b->JstrR; j <= b->JendR; j++)
    for (int i = b->Istr; i <= b->IendR; i++) {
      o->umask_wet[X2(i, j)] = WD(i - 1, j) + WD(i, j);
      if (o->umask_wet[X2(i, j)] == 1.0) o->umask_wet[X2(i, j)] = WD(i - 1, j) - WD(i, j);
    }
  for (int j = b->Jstr; j <= b->JendR; j++)
    for (int i = b->IstrR; i <= b->IendR; i++) {
      o->vmask_wet[X2(i, j)] = WD(i, j - 1) + WD(i, j);
      if (o->vmask_wet[X2(i, j)] == 1.0) o->vmask_wet[X2(i, j)] = WD(i, j - 1) - WD(i, j);
    }
  for (int j = b->Jstr; j <= b->JendR; j++)
    for (int i = b->Istr; i <= b->IendR; i++) o->pmask_wet[X2(i, j)] = wd_psi(wetdry, ni, LBi, LBj, i, j);
  orc_exchange2d(o, b, 'p', o->pmask_wet);
  orc_exchange2d(o, b, 'r', o->rmask_wet);
  orc_exchange2d(o, b, 'u', o->umask_wet);
  orc_exchange2d(o, b, 'v', o->vmask_wet);
}

/* wetdry_avg_mask_tile :723-900: the masks the 3-D step uses, from the rho mask averaged over the fast steps and the
   direction of the time-averaged barotropic transport */
static void wd_avg_mask(orc_t *o, const orc_bounds *b, const double *wetdry, const double *DU, const double *DV) {
  ORC_LOCALS(o);
  double cff1, cff5, cff6;
  for (int j = b->JstrR; j <= b->JendR; j++)
    for (int i = b->IstrR; i <= b->IendR; i++) o->rmask_wet[X2(i, j)] = WD(i, j);
  for (int j = b->JstrR; j <= b->JendR; j++)
    for (int i = b->Istr; i <= b->IendR; i++) {
      cff1 = WD(i - 1, j) + WD(i, j);
      if (cff1 == 1.0) cff1 = WD(i - 1, j) - WD(i, j);
      cff5 = fabs(fabs(cff1) - 1.0);
      cff6 = 0.5 + copysign(0.5, DU[X2(i, j)]) * cff1;
      o->umask_wet[X2(i, j)] = 0.5 * cff1 * cff5 + cff6 * (1.0 - cff5);
      if (DU[X2(i, j)] == 0.0)                                            /* "catch lone ponds" */
        if (WD(i - 1, j) + WD(i, j) <= 1.0) o->umask_wet[X2(i, j)] = 0.0;
    }
  for (int j = b->Jstr; j <= b->JendR; j++)
    for (int i = b->IstrR; i <= b->IendR; i++) {
      cff1 = WD(i, j - 1) + WD(i, j);
      if (cff1 == 1.0) cff1 = WD(i, j - 1) - WD(i, j);
      cff5 = fabs(fabs(cff1) - 1.0);
      cff6 = 0.5 + copysign(0.5, DV[X2(i, j)]) * cff1;
      o->vmask_wet[X2(i, j)] = 0.5 * cff1 * cff5 + cff6 * (1.0 - cff5);
      if (DV[X2(i, j)] == 0.0)
        if (WD(i, j - 1) + WD(i, j) <= 1.0) o->vmask_wet[X2(i, j)] = 0.0;
    }
  for (int j = b->Jstr; j <= b->JendR; j++)
    for (int i = b->Istr; i <= b->IendR; i++) o->pmask_wet[X2(i, j)] = wd_psi(wetdry, ni, LBi, LBj, i, j);
  orc_exchange2d(o, b, 'p', o->pmask_wet);
  orc_exchange2d(o, b, 'r', o->rmask_wet);
  orc_exchange2d(o, b, 'u', o->umask_wet);
  orc_exchange2d(o, b, 'v', o->vmask_wet);
}

/* "Set masks full time-dependent masks" :273-349, :428-488 */
static void wd_full(orc_t *o, const orc_bounds *b) {
  ORC_LOCALS(o);
  for (int j = b->JstrR; j <= b->JendR; j++)
    for (int i = b->IstrR; i <= b->IendR; i++) o->rmask_full[X2(i, j)] = o->rmask_wet[X2(i, j)] * o->rmask[X2(i, j)];
  for (int j = b->Jstr; j <= b->JendR; j++)
    for (int i = b->Istr; i <= b->IendR; i++) {
      const double v = o->pmask_wet[X2(i, j)] * o->pmask[X2(i, j)];
      o->pmask_full[X2(i, j)] = v > 2.0 ? v : 2.0;                        /* MAX(..., 2.0_r8) as written :287,443 */
    }
  for (int j = b->JstrR; j <= b->JendR; j++)
    for (int i = b->Istr; i <= b->IendR; i++) o->umask_full[X2(i, j)] = o->umask_wet[X2(i, j)] * o->umask[X2(i, j)];
  for (int j = b->Jstr; j <= b->JendR; j++)
    for (int i = b->IstrR; i <= b->IendR; i++) o->vmask_full[X2(i, j)] = o->vmask_wet[X2(i, j)] * o->vmask[X2(i, j)];
  orc_exchange2d(o, b, 'p', o->pmask_full);
  orc_exchange2d(o, b, 'r', o->rmask_full);
  orc_exchange2d(o, b, 'u', o->umask_full);
  orc_exchange2d(o, b, 'v', o->vmask_full);
}

/* the local rho mask from the free surface :186-204, :395-404 */
static void wd_local(const orc_t *o, const orc_bounds *b, const double *zeta, double *wetdry) {
  ORC_LOCALS(o);
  const double eps = 1.0E-10;
  const int msk = (o->c.options & ORC_MASKING) != 0;
  for (int j = b->Jstr - 1; j <= b->JendR; j++)
    for (int i = b->Istr - 1; i <= b->IendR; i++) {
      WD(i, j) = 1.0;
      if (msk) WD(i, j) = WD(i, j) * o->rmask[X2(i, j)];
      if ((zeta[X2(i, j)] + o->h[X2(i, j)]) <= (o->Dcrit + eps)) WD(i, j) = 0.0;
    }
}

void orc_wetdry_ini(orc_t *o, int tile) {
  ORC_LOCALS(o);
  const orc_bounds *b = &o->b[tile];
  const int kstp = o->s.kstp;
  double *wetdry = (double *)calloc(nij, sizeof(double));
  wd_local(o, b, o->zeta + (size_t)(kstp - 1) * nij, wetdry);
  /* SOLVE3D: the masks of the 3-D step's form, with ubar, vbar(kstp) for the direction of the flow :466-472 (the #else branch,
     wetdry_mask_tile, is the 2-D model's) */
  wd_avg_mask(o, b, wetdry, o->ubar + (size_t)(kstp - 1) * nij, o->vbar + (size_t)(kstp - 1) * nij);
  wd_full(o, b);
  free(wetdry);
}

void orc_wetdry(orc_t *o, int tile) { orc_wetdry_ini(o, tile); }   /* wetdry(ng, tile, Tindex, .TRUE.) :25-90 */

void orc_wetdry_tile(orc_t *o, int tile) {
  ORC_LOCALS(o);
  const orc_bounds *b = &o->b[tile];
  const orc_cfg *c = &o->c;
  const int kstp = o->s.kstp, iif = o->s.iif, PRED = o->s.predictor;
  double *wetdry = (double *)calloc(nij, sizeof(double));
  wd_local(o, b, o->zeta + (size_t)(kstp - 1) * nij, wetdry);
  if (iif <= c->nfast) {
    wd_mask(o, b, wetdry);                                                /* :208-214 */
    if (PRED && iif == 1) {                                               /* :220-226 */
      for (int j = b->JstrR; j <= b->JendR; j++)
        for (int i = b->IstrR; i <= b->IendR; i++) o->rmask_wet_avg[X2(i, j)] = WD(i, j);
    } else {
      for (int j = b->JstrR; j <= b->JendR; j++)
        for (int i = b->IstrR; i <= b->IendR; i++) o->rmask_wet_avg[X2(i, j)] = o->rmask_wet_avg[X2(i, j)] + WD(i, j);
    }
    orc_exchange2d(o, b, 'r', o->rmask_wet_avg);
  } else {
    const double cff = 1.0 / (double)(2 * c->nfast);                      /* :250-266 */
    for (int j = b->Jstr - 1; j <= b->JendR; j++)
      for (int i = b->Istr - 1; i <= b->IendR; i++) WD(i, j) = trunc(o->rmask_wet_avg[X2(i, j)] * cff);
    wd_avg_mask(o, b, wetdry, o->DU_avg1, o->DV_avg1);
    wd_full(o, b);                                                        /* :278 (SOLVE3D: iif > nfast) */
  }
  free(wetdry);
}
