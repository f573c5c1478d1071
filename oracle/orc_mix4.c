/*
 * orc_mix4.c -- biharmonic horizontal mixing along s-surfaces, the harmonic operator applied twice:
 *   orc_t3dmix4      t3dmix4_s_tile    ROMS/Nonlinear/t3dmix4_s.h:94-478     (TS_DIF4 + MIX_S_TS)
 *   orc_uv3dmix4     uv3dmix4_s_tile   ROMS/Nonlinear/uv3dmix4_s.h:119-627   (UV_VIS4 + MIX_S_UV)
 *   orc_step2d_vis4  the UV_VIS4 block of step2d_tile, ROMS/Nonlinear/step2d_LF_AM3.h:1653-1920
 * The coefficient arrays visc4_r, visc4_p, diff4 hold the SQUARE ROOTS of VISC4 / TNU4 (inp_par.F:634, read_phypar.F:7840,
 * ini_hmixcoef.F:270-296).  TEST INFRASTRUCTURE (see orc.h).  PARITY STATUS: pinned bit for bit against the reference built
 * from oracle/ref/upwelling_bih.h (tests/test_oracle_vs_ref.py: main3d, kernel by kernel, 2x2 tiles).  The DIAGNOSTICS_TS /
 * DIAGNOSTICS_UV statements of the three routines are included (t3dmix4_s.h:468-472, uv3dmix4_s.h:586-620,
 * step2d_LF_AM3.h:1903-1920).
 */
#include <stdlib.h>
#include <string.h>
#include "orc.h"

void orc_set_mix4(orc_t *o, int uv_vis4, int ts_dif4) { o->uv_vis4 = uv_vis4 != 0; o->ts_dif4 = ts_dif4 != 0; }

static int closed(const orc_t *o, int edge, int var) { return orc_lbc(o, edge, var) == ORC_LBC_CLO; }

void orc_t3dmix4_geo(orc_t *o, int tile);                       /* orc_t3dmix_geo.c */
void orc_t3dmix4_iso(orc_t *o, int tile);

void orc_t3dmix4(orc_t *o, int tile) {
  if (!o->ts_dif4) return;
  if (o->c.options & ORC_MIX_GEO_TS) { orc_t3dmix4_geo(o, tile); return; }   /* t3dmix.F: t3dmix4_geo.h */
  if (o->c.options & ORC_MIX_ISO_TS) { orc_t3dmix4_iso(o, tile); return; }   /* t3dmix4_iso.h */
  ORC_LOCALS(o);
  const orc_bounds *b = &o->b[tile];
  const orc_cfg *c = &o->c;
  const int nrhs = o->s.nrhs, nnew = o->s.nnew;
  const int Istr = b->Istr, Iend = b->Iend, Jstr = b->Jstr, Jend = b->Jend;
  const int msk = (c->options & ORC_MASKING) != 0;
  double *t = o->t, *Hz = o->Hz, *pm = o->pm, *pn = o->pn;
  double cff, cff1, cff2, cff3;
  double *FX = (double *)calloc(3 * nij, sizeof(double)), *FE = FX + nij, *LapT = FX + 2 * nij;
  int Imin, Imax, Jmin, Jmax;                                   /* :222-236 */
  if (c->EWperiodic) { Imin = Istr - 1; Imax = Iend + 1; }
  else { Imin = Istr - 1 > 1 ? Istr - 1 : 1; Imax = Iend + 1 < c->Lm ? Iend + 1 : c->Lm; }
  if (c->NSperiodic) { Jmin = Jstr - 1; Jmax = Jend + 1; }
  else { Jmin = Jstr - 1 > 1 ? Jstr - 1 : 1; Jmax = Jend + 1 < c->Mm ? Jend + 1 : c->Mm; }
  for (int itrc = 1; itrc <= c->NT; itrc++) {
    const double *d4 = o->diff4 + (size_t)(itrc - 1) * nij;
    double *Dx = orc_dia_wrk(o, ORC_DIA_XDIF, itrc), *Dy = orc_dia_wrk(o, ORC_DIA_YDIF, itrc), *Dh = orc_dia_wrk(o, ORC_DIA_HDIF, itrc);
    for (int k = 1; k <= N; k++) {
      for (int j = Jmin; j <= Jmax; j++)
        for (int i = Imin; i <= Imax + 1; i++) {
          cff = 0.25 * (d4[X2(i, j)] + d4[X2(i - 1, j)]) * o->pmon_u[X2(i, j)];
          if (msk) cff = cff * o->umask[X2(i, j)];
          FX[X2(i, j)] = cff * (Hz[X3(i, j, k)] + Hz[X3(i - 1, j, k)]) * (t[XT(i, j, k, nrhs, itrc)] - t[XT(i - 1, j, k, nrhs, itrc)]);
        }
      for (int j = Jmin; j <= Jmax + 1; j++)
        for (int i = Imin; i <= Imax; i++) {
          cff = 0.25 * (d4[X2(i, j)] + d4[X2(i, j - 1)]) * o->pnom_v[X2(i, j)];
          if (msk) cff = cff * o->vmask[X2(i, j)];
          FE[X2(i, j)] = cff * (Hz[X3(i, j, k)] + Hz[X3(i, j - 1, k)]) * (t[XT(i, j, k, nrhs, itrc)] - t[XT(i, j - 1, k, nrhs, itrc)]);
        }
      for (int j = Jmin; j <= Jmax; j++)                         /* first harmonic operator :341-349 */
        for (int i = Imin; i <= Imax; i++) {
          cff = 1.0 / Hz[X3(i, j, k)];
          LapT[X2(i, j)] = pm[X2(i, j)] * pn[X2(i, j)] * cff * (FX[X2(i + 1, j)] - FX[X2(i, j)] + FE[X2(i, j + 1)] - FE[X2(i, j)]);
        }
      /* closed or gradient conditions on the first operator :354-408 */
      if (!c->EWperiodic) {
        if (b->west) for (int j = Jmin; j <= Jmax; j++) LapT[X2(Istr - 1, j)] = closed(o, ORC_IWEST, ORC_ISTVAR + itrc - 1) ? 0.0 : LapT[X2(Istr, j)];
        if (b->east) for (int j = Jmin; j <= Jmax; j++) LapT[X2(Iend + 1, j)] = closed(o, ORC_IEAST, ORC_ISTVAR + itrc - 1) ? 0.0 : LapT[X2(Iend, j)];
      }
      if (!c->NSperiodic) {
        if (b->south) for (int i = Imin; i <= Imax; i++) LapT[X2(i, Jstr - 1)] = closed(o, ORC_ISOUTH, ORC_ISTVAR + itrc - 1) ? 0.0 : LapT[X2(i, Jstr)];
        if (b->north) for (int i = Imin; i <= Imax; i++) LapT[X2(i, Jend + 1)] = closed(o, ORC_INORTH, ORC_ISTVAR + itrc - 1) ? 0.0 : LapT[X2(i, Jend)];
      }
      for (int j = Jstr; j <= Jend; j++)                         /* :413-434 */
        for (int i = Istr; i <= Iend + 1; i++) {
          cff = 0.25 * (d4[X2(i, j)] + d4[X2(i - 1, j)]) * o->pmon_u[X2(i, j)];
          FX[X2(i, j)] = cff * (Hz[X3(i, j, k)] + Hz[X3(i - 1, j, k)]) * (LapT[X2(i, j)] - LapT[X2(i - 1, j)]);
          if (msk) FX[X2(i, j)] = FX[X2(i, j)] * o->umask[X2(i, j)];
        }
      for (int j = Jstr; j <= Jend + 1; j++)
        for (int i = Istr; i <= Iend; i++) {
          cff = 0.25 * (d4[X2(i, j)] + d4[X2(i, j - 1)]) * o->pnom_v[X2(i, j)];
          FE[X2(i, j)] = cff * (Hz[X3(i, j, k)] + Hz[X3(i, j - 1, k)]) * (LapT[X2(i, j)] - LapT[X2(i, j - 1)]);
          if (msk) FE[X2(i, j)] = FE[X2(i, j)] * o->vmask[X2(i, j)];
        }
      for (int j = Jstr; j <= Jend; j++)                         /* time step :461-474 */
        for (int i = Istr; i <= Iend; i++) {
          cff = c->dt * pm[X2(i, j)] * pn[X2(i, j)];
          cff1 = cff * (FX[X2(i + 1, j)] - FX[X2(i, j)]);
          cff2 = cff * (FE[X2(i, j + 1)] - FE[X2(i, j)]);
          cff3 = cff1 + cff2;
          t[XT(i, j, k, nnew, itrc)] = t[XT(i, j, k, nnew, itrc)] - cff3;
          if (Dx) { Dx[X3(i, j, k)] = -cff1; Dy[X3(i, j, k)] = -cff2; Dh[X3(i, j, k)] = -cff3; }
        }
    }
  }
  free(FX);
}

/* the conditions of the first harmonic operator of the momentum equations: uv3dmix4_s.h:335-470 (3-D: LBC of isUvel, isVvel)
   and step2d_LF_AM3.h:1722-1850 (2-D: isUbar, isVbar), on (iu0:iu1, ju0:ju1) for LapU and (iv0:iv1, jv0:jv1) for LapV */
void orc_lap_bc(const orc_t *o, const orc_bounds *b, double *LapU, double *LapV, int isu, int isv, int iu0, int iu1, int ju0,
                int ju1, int iv0, int iv1, int jv0, int jv1) {
  ORC_LOCALS(o);
  const orc_cfg *c = &o->c;
  const int Istr = b->Istr, Iend = b->Iend, Jstr = b->Jstr, Jend = b->Jend;
  const double gamma2 = c->gamma2;
  if (!c->EWperiodic) {
    if (b->west) {
      for (int j = ju0; j <= ju1; j++) LapU[X2(Istr, j)] = closed(o, ORC_IWEST, isu) ? 0.0 : LapU[X2(Istr + 1, j)];
      for (int j = jv0; j <= jv1; j++) LapV[X2(Istr - 1, j)] = closed(o, ORC_IWEST, isv) ? gamma2 * LapV[X2(Istr, j)] : 0.0;
    }
    if (b->east) {
      for (int j = ju0; j <= ju1; j++) LapU[X2(Iend + 1, j)] = closed(o, ORC_IEAST, isu) ? 0.0 : LapU[X2(Iend, j)];
      for (int j = jv0; j <= jv1; j++) LapV[X2(Iend + 1, j)] = closed(o, ORC_IEAST, isv) ? gamma2 * LapV[X2(Iend, j)] : 0.0;
    }
  }
  if (!c->NSperiodic) {
    if (b->south) {
      for (int i = iu0; i <= iu1; i++) LapU[X2(i, Jstr - 1)] = closed(o, ORC_ISOUTH, isu) ? gamma2 * LapU[X2(i, Jstr)] : 0.0;
      for (int i = iv0; i <= iv1; i++) LapV[X2(i, Jstr)] = closed(o, ORC_ISOUTH, isv) ? 0.0 : LapV[X2(i, Jstr + 1)];
    }
    if (b->north) {
      for (int i = iu0; i <= iu1; i++) LapU[X2(i, Jend + 1)] = closed(o, ORC_INORTH, isu) ? gamma2 * LapU[X2(i, Jend)] : 0.0;
      for (int i = iv0; i <= iv1; i++) LapV[X2(i, Jend + 1)] = closed(o, ORC_INORTH, isv) ? 0.0 : LapV[X2(i, Jend)];
    }
  }
  if (!(c->EWperiodic || c->NSperiodic)) {                      /* corners :472-520 */
    if (b->sw) {
      LapU[X2(Istr, Jstr - 1)] = 0.5 * (LapU[X2(Istr + 1, Jstr - 1)] + LapU[X2(Istr, Jstr)]);
      LapV[X2(Istr - 1, Jstr)] = 0.5 * (LapV[X2(Istr - 1, Jstr + 1)] + LapV[X2(Istr, Jstr)]);
    }
    if (b->se) {
      LapU[X2(Iend + 1, Jstr - 1)] = 0.5 * (LapU[X2(Iend, Jstr - 1)] + LapU[X2(Iend + 1, Jstr)]);
      LapV[X2(Iend + 1, Jstr)] = 0.5 * (LapV[X2(Iend, Jstr)] + LapV[X2(Iend + 1, Jstr + 1)]);
    }
    if (b->nw) {
      LapU[X2(Istr, Jend + 1)] = 0.5 * (LapU[X2(Istr + 1, Jend + 1)] + LapU[X2(Istr, Jend)]);
      LapV[X2(Istr - 1, Jend + 1)] = 0.5 * (LapV[X2(Istr, Jend + 1)] + LapV[X2(Istr - 1, Jend)]);
    }
    if (b->ne) {
      LapU[X2(Iend + 1, Jend + 1)] = 0.5 * (LapU[X2(Iend, Jend + 1)] + LapU[X2(Iend + 1, Jend)]);
      LapV[X2(Iend + 1, Jend + 1)] = 0.5 * (LapV[X2(Iend, Jend + 1)] + LapV[X2(Iend + 1, Jend)]);
    }
  }
}

void orc_uv3dmix4(orc_t *o, int tile) {
  if (!o->uv_vis4) return;
  if (o->mix_geo_uv) { orc_uv3dmix4_geo(o, tile); return; }          /* uv3dmix.F: MIX_GEO_UV -> uv3dmix4_geo.h */
  ORC_LOCALS(o);
  const orc_bounds *b = &o->b[tile];
  const orc_cfg *c = &o->c;
  const int nrhs = o->s.nrhs, nnew = o->s.nnew;
  const int Istr = b->Istr, Iend = b->Iend, Jstr = b->Jstr, Jend = b->Jend;
  const int IstrU = b->IstrU, JstrV = b->JstrV;
  const int msk = (c->options & ORC_MASKING) != 0;
  const double dt = c->dt;
  double *u = o->u, *v = o->v, *Hz = o->Hz, *pm = o->pm, *pn = o->pn;
  double *om_r = o->om_r, *on_r = o->on_r, *om_p = o->om_p, *on_p = o->on_p;
  double cff, cff1, cff2, cff3;
  double *UFe = (double *)calloc(6 * nij, sizeof(double));
  double *VFe = UFe + nij, *UFx = UFe + 2 * nij, *VFx = UFe + 3 * nij, *LapU = UFe + 4 * nij, *LapV = UFe + 5 * nij;
  int IminU, ImaxU, IminV, ImaxV, JminU, JmaxU, JminV, JmaxV;   /* :273-294 */
  if (c->EWperiodic) { IminU = Istr - 1; ImaxU = Iend + 1; IminV = Istr - 1; ImaxV = Iend + 1; }
  else {
    IminU = IstrU - 1 > 2 ? IstrU - 1 : 2; ImaxU = Iend + 1 < c->Lm ? Iend + 1 : c->Lm;
    IminV = Istr - 1 > 1 ? Istr - 1 : 1; ImaxV = Iend + 1 < c->Lm ? Iend + 1 : c->Lm;
  }
  if (c->NSperiodic) { JminU = Jstr - 1; JmaxU = Jend + 1; JminV = Jstr - 1; JmaxV = Jend + 1; }
  else {
    JminU = Jstr - 1 > 1 ? Jstr - 1 : 1; JmaxU = Jend + 1 < c->Mm ? Jend + 1 : c->Mm;
    JminV = JstrV - 1 > 2 ? JstrV - 1 : 2; JmaxV = Jend + 1 < c->Mm ? Jend + 1 : c->Mm;
  }
  const orc_diauv *d = o->duv;
  for (int k = 1; k <= N; k++) {
    for (int j = JminV - 1; j <= JmaxV; j++)
      for (int i = IminU - 1; i <= ImaxU; i++) {
        cff = 0.5 * (o->pmon_r[X2(i, j)] * ((pn[X2(i, j)] + pn[X2(i + 1, j)]) * u[X4(i + 1, j, k, nrhs)] -
                                            (pn[X2(i - 1, j)] + pn[X2(i, j)]) * u[X4(i, j, k, nrhs)]) -
                     o->pnom_r[X2(i, j)] * ((pm[X2(i, j)] + pm[X2(i, j + 1)]) * v[X4(i, j + 1, k, nrhs)] -
                                            (pm[X2(i, j - 1)] + pm[X2(i, j)]) * v[X4(i, j, k, nrhs)]));
        UFx[X2(i, j)] = on_r[X2(i, j)] * on_r[X2(i, j)] * o->visc4_r[X2(i, j)] * cff;
        VFe[X2(i, j)] = om_r[X2(i, j)] * om_r[X2(i, j)] * o->visc4_r[X2(i, j)] * cff;
      }
    for (int j = JminU; j <= JmaxU + 1; j++)
      for (int i = IminV; i <= ImaxV + 1; i++) {
        cff = 0.5 * (o->pmon_p[X2(i, j)] * ((pn[X2(i, j - 1)] + pn[X2(i, j)]) * v[X4(i, j, k, nrhs)] -
                                            (pn[X2(i - 1, j - 1)] + pn[X2(i - 1, j)]) * v[X4(i - 1, j, k, nrhs)]) +
                     o->pnom_p[X2(i, j)] * ((pm[X2(i - 1, j)] + pm[X2(i, j)]) * u[X4(i, j, k, nrhs)] -
                                            (pm[X2(i - 1, j - 1)] + pm[X2(i, j - 1)]) * u[X4(i, j - 1, k, nrhs)]));
        if (msk) cff = cff * o->pmask[X2(i, j)];
        UFe[X2(i, j)] = om_p[X2(i, j)] * om_p[X2(i, j)] * o->visc4_p[X2(i, j)] * cff;
        VFx[X2(i, j)] = on_p[X2(i, j)] * on_p[X2(i, j)] * o->visc4_p[X2(i, j)] * cff;
      }
    for (int j = JminU; j <= JmaxU; j++)                         /* first harmonic operator :313-333 */
      for (int i = IminU; i <= ImaxU; i++)
        LapU[X2(i, j)] = 0.125 * (pm[X2(i - 1, j)] + pm[X2(i, j)]) * (pn[X2(i - 1, j)] + pn[X2(i, j)]) *
                         ((pn[X2(i - 1, j)] + pn[X2(i, j)]) * (UFx[X2(i, j)] - UFx[X2(i - 1, j)]) +
                          (pm[X2(i - 1, j)] + pm[X2(i, j)]) * (UFe[X2(i, j + 1)] - UFe[X2(i, j)]));
    for (int j = JminV; j <= JmaxV; j++)
      for (int i = IminV; i <= ImaxV; i++)
        LapV[X2(i, j)] = 0.125 * (pm[X2(i, j)] + pm[X2(i, j - 1)]) * (pn[X2(i, j)] + pn[X2(i, j - 1)]) *
                         ((pn[X2(i, j - 1)] + pn[X2(i, j)]) * (VFx[X2(i + 1, j)] - VFx[X2(i, j)]) -
                          (pm[X2(i, j - 1)] + pm[X2(i, j)]) * (VFe[X2(i, j)] - VFe[X2(i, j - 1)]));
    orc_lap_bc(o, b, LapU, LapV, ORC_ISUVEL, ORC_ISVVEL, IminU, ImaxU, JminU, JmaxU, IminV, ImaxV, JminV, JmaxV);
    for (int j = JstrV - 1; j <= Jend; j++)                      /* second operator :526-575 */
      for (int i = IstrU - 1; i <= Iend; i++) {
        cff = Hz[X3(i, j, k)] * 0.5 *
              (o->pmon_r[X2(i, j)] * ((pn[X2(i, j)] + pn[X2(i + 1, j)]) * LapU[X2(i + 1, j)] - (pn[X2(i - 1, j)] + pn[X2(i, j)]) * LapU[X2(i, j)]) -
               o->pnom_r[X2(i, j)] * ((pm[X2(i, j)] + pm[X2(i, j + 1)]) * LapV[X2(i, j + 1)] - (pm[X2(i, j - 1)] + pm[X2(i, j)]) * LapV[X2(i, j)]));
        UFx[X2(i, j)] = on_r[X2(i, j)] * on_r[X2(i, j)] * o->visc4_r[X2(i, j)] * cff;
        VFe[X2(i, j)] = om_r[X2(i, j)] * om_r[X2(i, j)] * o->visc4_r[X2(i, j)] * cff;
      }
    for (int j = Jstr; j <= Jend + 1; j++)
      for (int i = Istr; i <= Iend + 1; i++) {
        cff = 0.125 * (Hz[X3(i - 1, j, k)] + Hz[X3(i, j, k)] + Hz[X3(i - 1, j - 1, k)] + Hz[X3(i, j - 1, k)]) *
              (o->pmon_p[X2(i, j)] * ((pn[X2(i, j - 1)] + pn[X2(i, j)]) * LapV[X2(i, j)] - (pn[X2(i - 1, j - 1)] + pn[X2(i - 1, j)]) * LapV[X2(i - 1, j)]) +
               o->pnom_p[X2(i, j)] * ((pm[X2(i - 1, j)] + pm[X2(i, j)]) * LapU[X2(i, j)] - (pm[X2(i - 1, j - 1)] + pm[X2(i, j - 1)]) * LapU[X2(i, j - 1)]));
        if (msk) cff = cff * o->pmask[X2(i, j)];
        UFe[X2(i, j)] = om_p[X2(i, j)] * om_p[X2(i, j)] * o->visc4_p[X2(i, j)] * cff;
        VFx[X2(i, j)] = on_p[X2(i, j)] * on_p[X2(i, j)] * o->visc4_p[X2(i, j)] * cff;
      }
    for (int j = Jstr; j <= Jend; j++)                           /* time step :581-622 */
      for (int i = IstrU; i <= Iend; i++) {
        cff = dt * 0.25 * (pm[X2(i - 1, j)] + pm[X2(i, j)]) * (pn[X2(i - 1, j)] + pn[X2(i, j)]);
        cff1 = 0.5 * (pn[X2(i - 1, j)] + pn[X2(i, j)]) * (UFx[X2(i, j)] - UFx[X2(i - 1, j)]);
        cff2 = 0.5 * (pm[X2(i - 1, j)] + pm[X2(i, j)]) * (UFe[X2(i, j + 1)] - UFe[X2(i, j)]);
        cff3 = cff * (cff1 + cff2);
        o->rufrc[X2(i, j)] = o->rufrc[X2(i, j)] - cff1 - cff2;
        u[X4(i, j, k, nnew)] = u[X4(i, j, k, nnew)] - cff3;
        if (d) {
          DUF(d->RUfrc, i, j, 3, d->M2hvis) = DUF(d->RUfrc, i, j, 3, d->M2hvis) - cff1 - cff2;
          DUF(d->RUfrc, i, j, 3, d->M2xvis) = DUF(d->RUfrc, i, j, 3, d->M2xvis) - cff1;
          DUF(d->RUfrc, i, j, 3, d->M2yvis) = DUF(d->RUfrc, i, j, 3, d->M2yvis) - cff2;
          DU3(d->U3wrk, i, j, k, d->M3hvis) = -cff3;
          DU3(d->U3wrk, i, j, k, d->M3xvis) = -cff * cff1;
          DU3(d->U3wrk, i, j, k, d->M3yvis) = -cff * cff2;
        }
      }
    for (int j = JstrV; j <= Jend; j++)
      for (int i = Istr; i <= Iend; i++) {
        cff = dt * 0.25 * (pm[X2(i, j)] + pm[X2(i, j - 1)]) * (pn[X2(i, j)] + pn[X2(i, j - 1)]);
        cff1 = 0.5 * (pn[X2(i, j - 1)] + pn[X2(i, j)]) * (VFx[X2(i + 1, j)] - VFx[X2(i, j)]);
        cff2 = 0.5 * (pm[X2(i, j - 1)] + pm[X2(i, j)]) * (VFe[X2(i, j)] - VFe[X2(i, j - 1)]);
        cff3 = cff * (cff1 - cff2);
        o->rvfrc[X2(i, j)] = o->rvfrc[X2(i, j)] - cff1 + cff2;
        v[X4(i, j, k, nnew)] = v[X4(i, j, k, nnew)] - cff3;
        if (d) {
          DUF(d->RVfrc, i, j, 3, d->M2hvis) = DUF(d->RVfrc, i, j, 3, d->M2hvis) - cff1 + cff2;
          DUF(d->RVfrc, i, j, 3, d->M2xvis) = DUF(d->RVfrc, i, j, 3, d->M2xvis) - cff1;
          DUF(d->RVfrc, i, j, 3, d->M2yvis) = DUF(d->RVfrc, i, j, 3, d->M2yvis) + cff2;
          DU3(d->V3wrk, i, j, k, d->M3hvis) = -cff3;
          DU3(d->V3wrk, i, j, k, d->M3xvis) = -cff * cff1;
          DU3(d->V3wrk, i, j, k, d->M3yvis) = cff * cff2;
        }
      }
  }
  free(UFe);
}

/* the UV_VIS4 block of step2d_tile (:1653-1920): rhs_ubar, rhs_vbar minus the biharmonic viscosity of ubar, vbar(krhs);
   U2rhs, V2rhs: the DIAGNOSTICS_UV terms of the call (or NULL) */
void orc_step2d_vis4(orc_t *o, const orc_bounds *b, int krhs, const double *Drhs, double *rhs_ubar, double *rhs_vbar, double *U2rhs,
                     double *V2rhs) {
  ORC_LOCALS(o);
  const orc_cfg *c = &o->c;
  const int Istr = b->Istr, Iend = b->Iend, Jstr = b->Jstr, Jend = b->Jend;
  const int IstrU = b->IstrU, JstrV = b->JstrV;
  const int msk = (c->options & ORC_MASKING) != 0;
  double *pm = o->pm, *pn = o->pn;
  double *om_r = o->om_r, *on_r = o->on_r, *om_p = o->om_p, *on_p = o->on_p;
  const double *ub = o->ubar + (size_t)(krhs - 1) * nij, *vb = o->vbar + (size_t)(krhs - 1) * nij;
  double cff, cff1, cff2, fac;
  double *UFe = (double *)calloc(7 * nij, sizeof(double));
  double *VFe = UFe + nij, *UFx = UFe + 2 * nij, *VFx = UFe + 3 * nij, *LapU = UFe + 4 * nij, *LapV = UFe + 5 * nij, *Drhs_p = UFe + 6 * nij;
  const orc_diauv *d = o->duv;
  for (int j = b->JstrVm2; j <= b->Jendp1; j++)                  /* :1667-1680 */
    for (int i = b->IstrUm2; i <= b->Iendp1; i++) {
      cff = o->visc4_r[X2(i, j)] * 0.5 *
            (o->pmon_r[X2(i, j)] * ((pn[X2(i, j)] + pn[X2(i + 1, j)]) * ub[X2(i + 1, j)] - (pn[X2(i - 1, j)] + pn[X2(i, j)]) * ub[X2(i, j)]) -
             o->pnom_r[X2(i, j)] * ((pm[X2(i, j)] + pm[X2(i, j + 1)]) * vb[X2(i, j + 1)] - (pm[X2(i, j - 1)] + pm[X2(i, j)]) * vb[X2(i, j)]));
      UFx[X2(i, j)] = on_r[X2(i, j)] * on_r[X2(i, j)] * cff;
      VFe[X2(i, j)] = om_r[X2(i, j)] * om_r[X2(i, j)] * cff;
    }
  for (int j = b->Jstrm1; j <= b->Jendp2; j++)                   /* :1681-1698 */
    for (int i = b->Istrm1; i <= b->Iendp2; i++) {
      cff = o->visc4_p[X2(i, j)] * 0.5 *
            (o->pmon_p[X2(i, j)] * ((pn[X2(i, j - 1)] + pn[X2(i, j)]) * vb[X2(i, j)] - (pn[X2(i - 1, j - 1)] + pn[X2(i - 1, j)]) * vb[X2(i - 1, j)]) +
             o->pnom_p[X2(i, j)] * ((pm[X2(i - 1, j)] + pm[X2(i, j)]) * ub[X2(i, j)] - (pm[X2(i - 1, j - 1)] + pm[X2(i, j - 1)]) * ub[X2(i, j - 1)]));
      if (msk) cff = cff * o->pmask[X2(i, j)];
      UFe[X2(i, j)] = om_p[X2(i, j)] * om_p[X2(i, j)] * cff;
      VFx[X2(i, j)] = on_p[X2(i, j)] * on_p[X2(i, j)] * cff;
    }
  for (int j = b->Jstrm1; j <= b->Jendp1; j++)                   /* first harmonic operator :1702-1722 */
    for (int i = b->IstrUm1; i <= b->Iendp1; i++)
      LapU[X2(i, j)] = 0.125 * (pm[X2(i - 1, j)] + pm[X2(i, j)]) * (pn[X2(i - 1, j)] + pn[X2(i, j)]) *
                       ((pn[X2(i - 1, j)] + pn[X2(i, j)]) * (UFx[X2(i, j)] - UFx[X2(i - 1, j)]) +
                        (pm[X2(i - 1, j)] + pm[X2(i, j)]) * (UFe[X2(i, j + 1)] - UFe[X2(i, j)]));
  for (int j = b->JstrVm1; j <= b->Jendp1; j++)
    for (int i = b->Istrm1; i <= b->Iendp1; i++)
      LapV[X2(i, j)] = 0.125 * (pm[X2(i, j)] + pm[X2(i, j - 1)]) * (pn[X2(i, j)] + pn[X2(i, j - 1)]) *
                       ((pn[X2(i, j - 1)] + pn[X2(i, j)]) * (VFx[X2(i + 1, j)] - VFx[X2(i, j)]) -
                        (pm[X2(i, j - 1)] + pm[X2(i, j)]) * (VFe[X2(i, j)] - VFe[X2(i, j - 1)]));
  orc_lap_bc(o, b, LapU, LapV, ORC_ISUBAR, ORC_ISVBAR, b->IstrUm1, b->Iendp1, b->Jstrm1, b->Jendp1, b->Istrm1, b->Iendp1, b->JstrVm1, b->Jendp1);   /* :1728-1853 */
  for (int j = Jstr; j <= Jend + 1; j++)                         /* :1856-1893 */
    for (int i = Istr; i <= Iend + 1; i++)
      Drhs_p[X2(i, j)] = 0.25 * (Drhs[X2(i, j)] + Drhs[X2(i - 1, j)] + Drhs[X2(i, j - 1)] + Drhs[X2(i - 1, j - 1)]);
  for (int j = JstrV - 1; j <= Jend; j++)
    for (int i = IstrU - 1; i <= Iend; i++) {
      cff = o->visc4_r[X2(i, j)] * Drhs[X2(i, j)] * 0.5 *
            (o->pmon_r[X2(i, j)] * ((pn[X2(i, j)] + pn[X2(i + 1, j)]) * LapU[X2(i + 1, j)] - (pn[X2(i - 1, j)] + pn[X2(i, j)]) * LapU[X2(i, j)]) -
             o->pnom_r[X2(i, j)] * ((pm[X2(i, j)] + pm[X2(i, j + 1)]) * LapV[X2(i, j + 1)] - (pm[X2(i, j - 1)] + pm[X2(i, j)]) * LapV[X2(i, j)]));
      UFx[X2(i, j)] = on_r[X2(i, j)] * on_r[X2(i, j)] * cff;
      VFe[X2(i, j)] = om_r[X2(i, j)] * om_r[X2(i, j)] * cff;
    }
  for (int j = Jstr; j <= Jend + 1; j++)
    for (int i = Istr; i <= Iend + 1; i++) {
      cff = o->visc4_p[X2(i, j)] * Drhs_p[X2(i, j)] * 0.5 *
            (o->pmon_p[X2(i, j)] * ((pn[X2(i, j - 1)] + pn[X2(i, j)]) * LapV[X2(i, j)] - (pn[X2(i - 1, j - 1)] + pn[X2(i - 1, j)]) * LapV[X2(i - 1, j)]) +
             o->pnom_p[X2(i, j)] * ((pm[X2(i - 1, j)] + pm[X2(i, j)]) * LapU[X2(i, j)] - (pm[X2(i - 1, j - 1)] + pm[X2(i, j - 1)]) * LapU[X2(i, j - 1)]));
      if (msk) cff = cff * o->pmask[X2(i, j)];
      UFe[X2(i, j)] = om_p[X2(i, j)] * om_p[X2(i, j)] * cff;
      VFx[X2(i, j)] = on_p[X2(i, j)] * on_p[X2(i, j)] * cff;
    }
  for (int j = Jstr; j <= Jend; j++)                             /* :1897-1921 */
    for (int i = IstrU; i <= Iend; i++) {
      cff1 = 0.5 * (pn[X2(i - 1, j)] + pn[X2(i, j)]) * (UFx[X2(i, j)] - UFx[X2(i - 1, j)]);
      cff2 = 0.5 * (pm[X2(i - 1, j)] + pm[X2(i, j)]) * (UFe[X2(i, j + 1)] - UFe[X2(i, j)]);
      fac = cff1 + cff2;
      rhs_ubar[X2(i, j)] = rhs_ubar[X2(i, j)] - fac;
      if (d && U2rhs) { DU2(U2rhs, i, j, d->M2hvis) = -fac; DU2(U2rhs, i, j, d->M2xvis) = -cff1; DU2(U2rhs, i, j, d->M2yvis) = -cff2; }
    }
  for (int j = JstrV; j <= Jend; j++)
    for (int i = Istr; i <= Iend; i++) {
      cff1 = 0.5 * (pn[X2(i, j - 1)] + pn[X2(i, j)]) * (VFx[X2(i + 1, j)] - VFx[X2(i, j)]);
      cff2 = 0.5 * (pm[X2(i, j - 1)] + pm[X2(i, j)]) * (VFe[X2(i, j)] - VFe[X2(i, j - 1)]);
      fac = cff1 - cff2;
      rhs_vbar[X2(i, j)] = rhs_vbar[X2(i, j)] - fac;
      if (d && V2rhs) { DU2(V2rhs, i, j, d->M2hvis) = -fac; DU2(V2rhs, i, j, d->M2xvis) = -cff1; DU2(V2rhs, i, j, d->M2yvis) = cff2; }
    }
  free(UFe);
}
