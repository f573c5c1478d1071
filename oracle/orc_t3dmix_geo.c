/*
 * orc_t3dmix_geo.c -- harmonic tracer mixing along geopotential surfaces, and along isopycnic surfaces.
 * TEST INFRASTRUCTURE (see orc.h).
 *
 * Follows t3dmix2_geo_tile, ROMS/Nonlinear/t3dmix2_geo.h:90-420 (rotated mixing tensor with the
 * two-level k1/k2 rolling buffers of the reference).
 * orc_t3dmix2_iso: t3dmix2_iso_tile, ROMS/Nonlinear/t3dmix2_iso.h:95-440 (MIX_ISO_TS; the default slope treatment:
 * none of TS_MIX_MAX_SLOPE, TS_MIX_MIN_STRAT, TS_MIX_STABILITY, TS_MIX_CLIMA).
 * orc_t3dmix4_geo: t3dmix4_geo_tile, ROMS/Nonlinear/t3dmix4_geo.h:98-780 (TS_DIF4 + MIX_GEO_TS: the rotated operator applied
 * twice, the first without coefficient and time step into LapT on the range widened by one point, closed / gradient
 * conditions and corner averages on LapT, the second on LapT); none of TS_MIX_STABILITY, TS_MIX_CLIMA, DIFF_3DCOEF.
 * orc_t3dmix4_iso: t3dmix4_iso_tile, ROMS/Nonlinear/t3dmix4_iso.h:98-812 (TS_DIF4 + MIX_ISO_TS, the same construction on the
 * density slopes; default slope treatment), pinned by oracle/ref/upwelling_bihiso.h in the periodic channel.
 * PARITY: pinned (t3dmix.F builds in oracle/_ref: BENCHMARK for the geopotential form, OVERFLOW for the isopycnic one,
 * oracle/ref/upwelling_bihgeo.h for the biharmonic geopotential form in the periodic channel: main3d 60 steps, 2x2 tiles,
 * rhs3d on a random state; the conditions on LapT at closed western / eastern walls and the corner averages (:475-600 for
 * iwest, ieast and the corners; t3dmix4_iso.h:504-618) between four walls since round 6: tests/test_oracle_vs_ref.py,
 * *_closed_small -- whole steps and rhs3d on a perturbed state, 2x2 tiles).
 */
#include "orc.h"
#include <stdlib.h>
#define MAX(a, b) ((a) > (b) ? (a) : (b))
#define MIN(a, b) ((a) < (b) ? (a) : (b))
/* two-level scratch (i,j,l), l = 1,2 */
#define L2(A, i, j, l) A[X2(i, j) + (size_t)((l) - 1) * nij]

void orc_t3dmix2_geo(orc_t *o, int tile) {
  ORC_LOCALS(o);
  const orc_bounds *b = &o->b[tile];
  const int nrhs = o->s.nrhs, nnew = o->s.nnew;
  const int Istr = b->Istr, Iend = b->Iend, Jstr = b->Jstr, Jend = b->Jend;
  const double dt = o->c.dt;
  double *t = o->t, *z_r = o->z_r, *Hz = o->Hz, *pm = o->pm, *pn = o->pn, *diff2 = o->diff2;
  double *S = (double *)calloc(14 * nij, sizeof(double));
  double *FE = S, *FX = S + nij, *FS = S + 2 * nij, *dTdz = S + 4 * nij, *dTdx = S + 6 * nij,
         *dTde = S + 8 * nij, *dZdx = S + 10 * nij, *dZde = S + 12 * nij;
  double cff, cff1, cff2, cff3, cff4;
  for (int itrc = 1; itrc <= o->c.NT; itrc++) {
    int k1, k2 = 1;
    for (int k = 0; k <= N; k++) {
      k1 = k2;
      k2 = 3 - k1;
      if (k < N) {
        for (int j = Jstr; j <= Jend; j++)
          for (int i = Istr; i <= Iend + 1; i++) {
            cff = 0.5 * (pm[X2(i, j)] + pm[X2(i - 1, j)]);
            if (o->c.options & ORC_MASKING) cff = cff * o->umask[X2(i, j)];                  /* t3dmix2_geo.h:229 */
            if (o->wet_dry) cff = cff * o->umask_wet[X2(i, j)];                              /* WET_DRY :232 */
            L2(dZdx, i, j, k2) = cff * (z_r[X3(i, j, k + 1)] - z_r[X3(i - 1, j, k + 1)]);
            L2(dTdx, i, j, k2) = cff * (t[XT(i, j, k + 1, nrhs, itrc)] - t[XT(i - 1, j, k + 1, nrhs, itrc)]);
          }
        for (int j = Jstr; j <= Jend + 1; j++)
          for (int i = Istr; i <= Iend; i++) {
            cff = 0.5 * (pn[X2(i, j)] + pn[X2(i, j - 1)]);
            if (o->c.options & ORC_MASKING) cff = cff * o->vmask[X2(i, j)];                  /* t3dmix2_geo.h:261 */
            if (o->wet_dry) cff = cff * o->vmask_wet[X2(i, j)];                              /* WET_DRY :264 */
            L2(dZde, i, j, k2) = cff * (z_r[X3(i, j, k + 1)] - z_r[X3(i, j - 1, k + 1)]);
            L2(dTde, i, j, k2) = cff * (t[XT(i, j, k + 1, nrhs, itrc)] - t[XT(i, j - 1, k + 1, nrhs, itrc)]);
          }
      }
      if (k == 0 || k == N) {
        for (int j = Jstr - 1; j <= Jend + 1; j++)
          for (int i = Istr - 1; i <= Iend + 1; i++) { L2(dTdz, i, j, k2) = 0.0; L2(FS, i, j, k2) = 0.0; }
      } else {
        for (int j = Jstr - 1; j <= Jend + 1; j++)
          for (int i = Istr - 1; i <= Iend + 1; i++) {
            cff = 1.0 / (z_r[X3(i, j, k + 1)] - z_r[X3(i, j, k)]);
            L2(dTdz, i, j, k2) = cff * (t[XT(i, j, k + 1, nrhs, itrc)] - t[XT(i, j, k, nrhs, itrc)]);
          }
      }
      if (k > 0) {
        for (int j = Jstr; j <= Jend; j++)
          for (int i = Istr; i <= Iend + 1; i++) {
            cff = 0.25 * (diff2[X2T(i, j, itrc)] + diff2[X2T(i - 1, j, itrc)]) * o->on_u[X2(i, j)];
            FX[X2(i, j)] = cff * (Hz[X3(i, j, k)] + Hz[X3(i - 1, j, k)]) *
                           (L2(dTdx, i, j, k1) -
                            0.5 * (MIN(L2(dZdx, i, j, k1), 0.0) * (L2(dTdz, i - 1, j, k1) + L2(dTdz, i, j, k2)) +
                                   MAX(L2(dZdx, i, j, k1), 0.0) * (L2(dTdz, i - 1, j, k2) + L2(dTdz, i, j, k1))));
          }
        for (int j = Jstr; j <= Jend + 1; j++)
          for (int i = Istr; i <= Iend; i++) {
            cff = 0.25 * (diff2[X2T(i, j, itrc)] + diff2[X2T(i, j - 1, itrc)]) * o->om_v[X2(i, j)];
            FE[X2(i, j)] = cff * (Hz[X3(i, j, k)] + Hz[X3(i, j - 1, k)]) *
                           (L2(dTde, i, j, k1) -
                            0.5 * (MIN(L2(dZde, i, j, k1), 0.0) * (L2(dTdz, i, j - 1, k1) + L2(dTdz, i, j, k2)) +
                                   MAX(L2(dZde, i, j, k1), 0.0) * (L2(dTdz, i, j - 1, k2) + L2(dTdz, i, j, k1))));
          }
        if (k < N) {
          for (int j = Jstr; j <= Jend; j++)
            for (int i = Istr; i <= Iend; i++) {
              cff = 0.5 * diff2[X2T(i, j, itrc)];
              cff1 = MIN(L2(dZdx, i, j, k1), 0.0);
              cff2 = MIN(L2(dZdx, i + 1, j, k2), 0.0);
              cff3 = MAX(L2(dZdx, i, j, k2), 0.0);
              cff4 = MAX(L2(dZdx, i + 1, j, k1), 0.0);
              L2(FS, i, j, k2) = cff * (cff1 * (cff1 * L2(dTdz, i, j, k2) - L2(dTdx, i, j, k1)) +
                                        cff2 * (cff2 * L2(dTdz, i, j, k2) - L2(dTdx, i + 1, j, k2)) +
                                        cff3 * (cff3 * L2(dTdz, i, j, k2) - L2(dTdx, i, j, k2)) +
                                        cff4 * (cff4 * L2(dTdz, i, j, k2) - L2(dTdx, i + 1, j, k1)));
              cff1 = MIN(L2(dZde, i, j, k1), 0.0);
              cff2 = MIN(L2(dZde, i, j + 1, k2), 0.0);
              cff3 = MAX(L2(dZde, i, j, k2), 0.0);
              cff4 = MAX(L2(dZde, i, j + 1, k1), 0.0);
              L2(FS, i, j, k2) = L2(FS, i, j, k2) +
                                 cff * (cff1 * (cff1 * L2(dTdz, i, j, k2) - L2(dTde, i, j, k1)) +
                                        cff2 * (cff2 * L2(dTdz, i, j, k2) - L2(dTde, i, j + 1, k2)) +
                                        cff3 * (cff3 * L2(dTdz, i, j, k2) - L2(dTde, i, j, k2)) +
                                        cff4 * (cff4 * L2(dTdz, i, j, k2) - L2(dTde, i, j + 1, k1)));
            }
        }
        for (int j = Jstr; j <= Jend; j++)
          for (int i = Istr; i <= Iend; i++) {
            cff = dt * pm[X2(i, j)] * pn[X2(i, j)];
            cff1 = cff * (FX[X2(i + 1, j)] - FX[X2(i, j)]);
            cff2 = cff * (FE[X2(i, j + 1)] - FE[X2(i, j)]);
            cff3 = dt * (L2(FS, i, j, k2) - L2(FS, i, j, k1));
            cff4 = cff1 + cff2 + cff3;
            t[XT(i, j, k, nnew, itrc)] = t[XT(i, j, k, nnew, itrc)] + cff4;
            if (o->dia) {                                               /* DIAGNOSTICS_TS t3dmix2_geo.h:409-414, t3dmix2_iso.h:428-433 */
              orc_dia_wrk(o, ORC_DIA_XDIF, itrc)[X3(i, j, k)] = cff1;
              orc_dia_wrk(o, ORC_DIA_YDIF, itrc)[X3(i, j, k)] = cff2;
              orc_dia_wrk(o, ORC_DIA_SDIF, itrc)[X3(i, j, k)] = cff3;
              orc_dia_wrk(o, ORC_DIA_HDIF, itrc)[X3(i, j, k)] = cff4;
            }
          }
      }
    }
  }
  free(S);
}

void orc_t3dmix2_iso(orc_t *o, int tile) {
  ORC_LOCALS(o);
  const orc_bounds *b = &o->b[tile];
  const int nrhs = o->s.nrhs, nnew = o->s.nnew;
  const int Istr = b->Istr, Iend = b->Iend, Jstr = b->Jstr, Jend = b->Jend;
  const double dt = o->c.dt, eps = 0.5;
  double *t = o->t, *z_r = o->z_r, *Hz = o->Hz, *pm = o->pm, *pn = o->pn, *diff2 = o->diff2, *pden = o->pden;
  double *S = (double *)calloc(14 * nij, sizeof(double));
  double *FE = S, *FX = S + nij, *FS = S + 2 * nij, *dTdr = S + 4 * nij, *dTdx = S + 6 * nij,
         *dTde = S + 8 * nij, *dRdx = S + 10 * nij, *dRde = S + 12 * nij;
  double cff, cff1, cff2, cff3, cff4;
  for (int itrc = 1; itrc <= o->c.NT; itrc++) {
    int k1, k2 = 1;
    for (int k = 0; k <= N; k++) {
      k1 = k2;
      k2 = 3 - k1;
      if (k < N) {
        for (int j = Jstr; j <= Jend; j++)
          for (int i = Istr; i <= Iend + 1; i++) {
            cff = 0.5 * (pm[X2(i, j)] + pm[X2(i - 1, j)]);
            if (o->c.options & ORC_MASKING) cff = cff * o->umask[X2(i, j)];
            L2(dRdx, i, j, k2) = cff * (pden[X3(i, j, k + 1)] - pden[X3(i - 1, j, k + 1)]);
            L2(dTdx, i, j, k2) = cff * (t[XT(i, j, k + 1, nrhs, itrc)] - t[XT(i - 1, j, k + 1, nrhs, itrc)]);
          }
        for (int j = Jstr; j <= Jend + 1; j++)
          for (int i = Istr; i <= Iend; i++) {
            cff = 0.5 * (pn[X2(i, j)] + pn[X2(i, j - 1)]);
            if (o->c.options & ORC_MASKING) cff = cff * o->vmask[X2(i, j)];
            L2(dRde, i, j, k2) = cff * (pden[X3(i, j, k + 1)] - pden[X3(i, j - 1, k + 1)]);
            L2(dTde, i, j, k2) = cff * (t[XT(i, j, k + 1, nrhs, itrc)] - t[XT(i, j - 1, k + 1, nrhs, itrc)]);
          }
      }
      if (k == 0 || k == N) {
        for (int j = Jstr - 1; j <= Jend + 1; j++)
          for (int i = Istr - 1; i <= Iend + 1; i++) { L2(dTdr, i, j, k2) = 0.0; L2(FS, i, j, k2) = 0.0; }
      } else {
        for (int j = Jstr - 1; j <= Jend + 1; j++)
          for (int i = Istr - 1; i <= Iend + 1; i++) {
            cff1 = MAX(pden[X3(i, j, k)] - pden[X3(i, j, k + 1)], eps);
            cff = -1.0 / cff1;
            L2(dTdr, i, j, k2) = cff * (t[XT(i, j, k + 1, nrhs, itrc)] - t[XT(i, j, k, nrhs, itrc)]);
            L2(FS, i, j, k2) = cff * (z_r[X3(i, j, k + 1)] - z_r[X3(i, j, k)]);
          }
      }
      if (k > 0) {
        for (int j = Jstr; j <= Jend; j++)
          for (int i = Istr; i <= Iend + 1; i++) {
            cff = 0.25 * (diff2[X2T(i, j, itrc)] + diff2[X2T(i - 1, j, itrc)]) * o->on_u[X2(i, j)];
            FX[X2(i, j)] = cff * (Hz[X3(i, j, k)] + Hz[X3(i - 1, j, k)]) *
                           (L2(dTdx, i, j, k1) -
                            0.5 * (MAX(L2(dRdx, i, j, k1), 0.0) * (L2(dTdr, i - 1, j, k1) + L2(dTdr, i, j, k2)) +
                                   MIN(L2(dRdx, i, j, k1), 0.0) * (L2(dTdr, i - 1, j, k2) + L2(dTdr, i, j, k1))));
          }
        for (int j = Jstr; j <= Jend + 1; j++)
          for (int i = Istr; i <= Iend; i++) {
            cff = 0.25 * (diff2[X2T(i, j, itrc)] + diff2[X2T(i, j - 1, itrc)]) * o->om_v[X2(i, j)];
            FE[X2(i, j)] = cff * (Hz[X3(i, j, k)] + Hz[X3(i, j - 1, k)]) *
                           (L2(dTde, i, j, k1) -
                            0.5 * (MAX(L2(dRde, i, j, k1), 0.0) * (L2(dTdr, i, j - 1, k1) + L2(dTdr, i, j, k2)) +
                                   MIN(L2(dRde, i, j, k1), 0.0) * (L2(dTdr, i, j - 1, k2) + L2(dTdr, i, j, k1))));
          }
        if (k < N) {
          for (int j = Jstr; j <= Jend; j++)
            for (int i = Istr; i <= Iend; i++) {
              cff1 = MAX(L2(dRdx, i, j, k1), 0.0);
              cff2 = MAX(L2(dRdx, i + 1, j, k2), 0.0);
              cff3 = MIN(L2(dRdx, i, j, k2), 0.0);
              cff4 = MIN(L2(dRdx, i + 1, j, k1), 0.0);
              cff = cff1 * (cff1 * L2(dTdr, i, j, k2) - L2(dTdx, i, j, k1)) +
                    cff2 * (cff2 * L2(dTdr, i, j, k2) - L2(dTdx, i + 1, j, k2)) +
                    cff3 * (cff3 * L2(dTdr, i, j, k2) - L2(dTdx, i, j, k2)) +
                    cff4 * (cff4 * L2(dTdr, i, j, k2) - L2(dTdx, i + 1, j, k1));
              cff1 = MAX(L2(dRde, i, j, k1), 0.0);
              cff2 = MAX(L2(dRde, i, j + 1, k2), 0.0);
              cff3 = MIN(L2(dRde, i, j, k2), 0.0);
              cff4 = MIN(L2(dRde, i, j + 1, k1), 0.0);
              cff = cff + cff1 * (cff1 * L2(dTdr, i, j, k2) - L2(dTde, i, j, k1)) +
                    cff2 * (cff2 * L2(dTdr, i, j, k2) - L2(dTde, i, j + 1, k2)) +
                    cff3 * (cff3 * L2(dTdr, i, j, k2) - L2(dTde, i, j, k2)) +
                    cff4 * (cff4 * L2(dTdr, i, j, k2) - L2(dTde, i, j + 1, k1));
              L2(FS, i, j, k2) = 0.5 * cff * diff2[X2T(i, j, itrc)] * L2(FS, i, j, k2);
            }
        }
        for (int j = Jstr; j <= Jend; j++)
          for (int i = Istr; i <= Iend; i++) {
            cff = dt * pm[X2(i, j)] * pn[X2(i, j)];
            cff1 = cff * (FX[X2(i + 1, j)] - FX[X2(i, j)]);
            cff2 = cff * (FE[X2(i, j + 1)] - FE[X2(i, j)]);
            cff3 = dt * (L2(FS, i, j, k2) - L2(FS, i, j, k1));
            cff4 = cff1 + cff2 + cff3;
            t[XT(i, j, k, nnew, itrc)] = t[XT(i, j, k, nnew, itrc)] + cff4;
            if (o->dia) {                                               /* DIAGNOSTICS_TS t3dmix2_geo.h:409-414, t3dmix2_iso.h:428-433 */
              orc_dia_wrk(o, ORC_DIA_XDIF, itrc)[X3(i, j, k)] = cff1;
              orc_dia_wrk(o, ORC_DIA_YDIF, itrc)[X3(i, j, k)] = cff2;
              orc_dia_wrk(o, ORC_DIA_SDIF, itrc)[X3(i, j, k)] = cff3;
              orc_dia_wrk(o, ORC_DIA_HDIF, itrc)[X3(i, j, k)] = cff4;
            }
          }
      }
    }
  }
  free(S);
}

/* one rotated harmonic operator of t3dmix4_geo.h on (i0:i1, j0:j1): A = the 3-D field it acts on (t(:,:,:,nrhs,itrc) | LapT),
   d4 = sqrt(TNU4) of the tracer; out == LapT: first operator :262-470, else the time step :600-770 on t(:,:,:,nnew,itrc) */
static void geo4_op(orc_t *o, const double *A, const double *d4, int i0, int i1, int j0, int j1, double *LapT, double *tnew,
                    double *S, int itrc) {
  ORC_LOCALS(o);
  const double dt = o->c.dt;
  const double *z_r = o->z_r, *Hz = o->Hz, *pm = o->pm, *pn = o->pn;
  double *FE = S, *FX = S + nij, *FS = S + 2 * nij, *dTdz = S + 4 * nij, *dTdx = S + 6 * nij, *dTde = S + 8 * nij,
         *dZdx = S + 10 * nij, *dZde = S + 12 * nij;
  double cff, cff1, cff2, cff3, cff4;
  int k1, k2 = 1;
  for (int k = 0; k <= N; k++) {
    k1 = k2;
    k2 = 3 - k1;
    if (k < N) {
      for (int j = j0; j <= j1; j++)
        for (int i = i0; i <= i1 + 1; i++) {
          cff = 0.5 * (pm[X2(i, j)] + pm[X2(i - 1, j)]);
          if (o->c.options & ORC_MASKING) cff = cff * o->umask[X2(i, j)];
          if (o->wet_dry) cff = cff * o->umask_wet[X2(i, j)];
          L2(dZdx, i, j, k2) = cff * (z_r[X3(i, j, k + 1)] - z_r[X3(i - 1, j, k + 1)]);
          L2(dTdx, i, j, k2) = cff * (A[X3(i, j, k + 1)] - A[X3(i - 1, j, k + 1)]);
        }
      for (int j = j0; j <= j1 + 1; j++)
        for (int i = i0; i <= i1; i++) {
          cff = 0.5 * (pn[X2(i, j)] + pn[X2(i, j - 1)]);
          if (o->c.options & ORC_MASKING) cff = cff * o->vmask[X2(i, j)];
          if (o->wet_dry) cff = cff * o->vmask_wet[X2(i, j)];
          L2(dZde, i, j, k2) = cff * (z_r[X3(i, j, k + 1)] - z_r[X3(i, j - 1, k + 1)]);
          L2(dTde, i, j, k2) = cff * (A[X3(i, j, k + 1)] - A[X3(i, j - 1, k + 1)]);
        }
    }
    if (k == 0 || k == N) {
      for (int j = j0 - 1; j <= j1 + 1; j++)
        for (int i = i0 - 1; i <= i1 + 1; i++) { L2(dTdz, i, j, k2) = 0.0; L2(FS, i, j, k2) = 0.0; }
    } else {
      for (int j = j0 - 1; j <= j1 + 1; j++)
        for (int i = i0 - 1; i <= i1 + 1; i++) {
          cff = 1.0 / (z_r[X3(i, j, k + 1)] - z_r[X3(i, j, k)]);
          L2(dTdz, i, j, k2) = cff * (A[X3(i, j, k + 1)] - A[X3(i, j, k)]);
        }
    }
    if (k > 0) {
      for (int j = j0; j <= j1; j++)
        for (int i = i0; i <= i1 + 1; i++) {
          cff = 0.25 * (d4[X2(i, j)] + d4[X2(i - 1, j)]) * o->on_u[X2(i, j)];
          FX[X2(i, j)] = cff * (Hz[X3(i, j, k)] + Hz[X3(i - 1, j, k)]) *
                         (L2(dTdx, i, j, k1) -
                          0.5 * (MIN(L2(dZdx, i, j, k1), 0.0) * (L2(dTdz, i - 1, j, k1) + L2(dTdz, i, j, k2)) +
                                 MAX(L2(dZdx, i, j, k1), 0.0) * (L2(dTdz, i - 1, j, k2) + L2(dTdz, i, j, k1))));
        }
      for (int j = j0; j <= j1 + 1; j++)
        for (int i = i0; i <= i1; i++) {
          cff = 0.25 * (d4[X2(i, j)] + d4[X2(i, j - 1)]) * o->om_v[X2(i, j)];
          FE[X2(i, j)] = cff * (Hz[X3(i, j, k)] + Hz[X3(i, j - 1, k)]) *
                         (L2(dTde, i, j, k1) -
                          0.5 * (MIN(L2(dZde, i, j, k1), 0.0) * (L2(dTdz, i, j - 1, k1) + L2(dTdz, i, j, k2)) +
                                 MAX(L2(dZde, i, j, k1), 0.0) * (L2(dTdz, i, j - 1, k2) + L2(dTdz, i, j, k1))));
        }
      if (k < N) {
        for (int j = j0; j <= j1; j++)
          for (int i = i0; i <= i1; i++) {
            const double difx = 0.5 * d4[X2(i, j)], dife = difx;
            cff1 = MIN(L2(dZdx, i, j, k1), 0.0);
            cff2 = MIN(L2(dZdx, i + 1, j, k2), 0.0);
            cff3 = MAX(L2(dZdx, i, j, k2), 0.0);
            cff4 = MAX(L2(dZdx, i + 1, j, k1), 0.0);
            L2(FS, i, j, k2) = difx * (cff1 * (cff1 * L2(dTdz, i, j, k2) - L2(dTdx, i, j, k1)) +
                                       cff2 * (cff2 * L2(dTdz, i, j, k2) - L2(dTdx, i + 1, j, k2)) +
                                       cff3 * (cff3 * L2(dTdz, i, j, k2) - L2(dTdx, i, j, k2)) +
                                       cff4 * (cff4 * L2(dTdz, i, j, k2) - L2(dTdx, i + 1, j, k1)));
            cff1 = MIN(L2(dZde, i, j, k1), 0.0);
            cff2 = MIN(L2(dZde, i, j + 1, k2), 0.0);
            cff3 = MAX(L2(dZde, i, j, k2), 0.0);
            cff4 = MAX(L2(dZde, i, j + 1, k1), 0.0);
            L2(FS, i, j, k2) = L2(FS, i, j, k2) +
                               dife * (cff1 * (cff1 * L2(dTdz, i, j, k2) - L2(dTde, i, j, k1)) +
                                       cff2 * (cff2 * L2(dTdz, i, j, k2) - L2(dTde, i, j + 1, k2)) +
                                       cff3 * (cff3 * L2(dTdz, i, j, k2) - L2(dTde, i, j, k2)) +
                                       cff4 * (cff4 * L2(dTdz, i, j, k2) - L2(dTde, i, j + 1, k1)));
          }
      }
      if (LapT) {                                                   /* :458-470 */
        for (int j = j0; j <= j1; j++)
          for (int i = i0; i <= i1; i++) {
            cff = pm[X2(i, j)] * pn[X2(i, j)];
            cff1 = 1.0 / Hz[X3(i, j, k)];
            LapT[X3(i, j, k)] = cff1 * (cff * (FX[X2(i + 1, j)] - FX[X2(i, j)] + FE[X2(i, j + 1)] - FE[X2(i, j)]) +
                                        (L2(FS, i, j, k2) - L2(FS, i, j, k1)));
          }
      } else {                                                      /* :754-772 */
        for (int j = j0; j <= j1; j++)
          for (int i = i0; i <= i1; i++) {
            cff = dt * pm[X2(i, j)] * pn[X2(i, j)];
            cff1 = cff * (FX[X2(i + 1, j)] - FX[X2(i, j)]);
            cff2 = cff * (FE[X2(i, j + 1)] - FE[X2(i, j)]);
            cff3 = dt * (L2(FS, i, j, k2) - L2(FS, i, j, k1));
            cff4 = cff1 + cff2 + cff3;
            tnew[X3(i, j, k)] = tnew[X3(i, j, k)] - cff4;
            if (o->dia) {                                           /* DIAGNOSTICS_TS :763-768 */
              orc_dia_wrk(o, ORC_DIA_XDIF, itrc)[X3(i, j, k)] = -cff1;
              orc_dia_wrk(o, ORC_DIA_YDIF, itrc)[X3(i, j, k)] = -cff2;
              orc_dia_wrk(o, ORC_DIA_SDIF, itrc)[X3(i, j, k)] = -cff3;
              orc_dia_wrk(o, ORC_DIA_HDIF, itrc)[X3(i, j, k)] = -cff4;
            }
          }
      }
    }
  }
}

/* the conditions on the first harmonic operator at the walls and corners (t3dmix4_geo.h:475-600, t3dmix4_iso.h:504-618) */
static void lap_walls(orc_t *o, const orc_bounds *b, double *L, int var, int Imin, int Imax, int Jmin, int Jmax) {
  ORC_LOCALS(o);
  const orc_cfg *c = &o->c;
  const int Istr = b->Istr, Iend = b->Iend, Jstr = b->Jstr, Jend = b->Jend;
  for (int k = 1; k <= N; k++) {
      if (!c->EWperiodic) {
        if (b->west) for (int j = Jmin; j <= Jmax; j++) L[X3(Istr - 1, j, k)] = orc_lbc(o, ORC_IWEST, var) == ORC_LBC_CLO ? 0.0 : L[X3(Istr, j, k)];
        if (b->east) for (int j = Jmin; j <= Jmax; j++) L[X3(Iend + 1, j, k)] = orc_lbc(o, ORC_IEAST, var) == ORC_LBC_CLO ? 0.0 : L[X3(Iend, j, k)];
      }
      if (!c->NSperiodic) {
        if (b->south) for (int i = Imin; i <= Imax; i++) L[X3(i, Jstr - 1, k)] = orc_lbc(o, ORC_ISOUTH, var) == ORC_LBC_CLO ? 0.0 : L[X3(i, Jstr, k)];
        if (b->north) for (int i = Imin; i <= Imax; i++) L[X3(i, Jend + 1, k)] = orc_lbc(o, ORC_INORTH, var) == ORC_LBC_CLO ? 0.0 : L[X3(i, Jend, k)];
      }
      if (!c->EWperiodic && !c->NSperiodic) {
        if (b->south && b->west) L[X3(Istr - 1, Jstr - 1, k)] = 0.5 * (L[X3(Istr, Jstr - 1, k)] + L[X3(Istr - 1, Jstr, k)]);
        if (b->south && b->east) L[X3(Iend + 1, Jstr - 1, k)] = 0.5 * (L[X3(Iend, Jstr - 1, k)] + L[X3(Iend + 1, Jstr, k)]);
        if (b->north && b->west) L[X3(Istr - 1, Jend + 1, k)] = 0.5 * (L[X3(Istr, Jend + 1, k)] + L[X3(Istr - 1, Jend, k)]);
        if (b->north && b->east) L[X3(Iend + 1, Jend + 1, k)] = 0.5 * (L[X3(Iend, Jend + 1, k)] + L[X3(Iend + 1, Jend, k)]);
      }
  }
}


/* one rotated harmonic operator of t3dmix4_iso.h (its default slope treatment: the stratification floored at eps) on
   (i0:i1, j0:j1): A = t(:,:,:,nrhs,itrc) | LapT; LapT != NULL: the first operator :270-499, else the time step :623-808 */
static void iso4_op(orc_t *o, const double *A, const double *d4, int i0, int i1, int j0, int j1, double *LapT, double *tnew,
                    double *S, int itrc) {
  ORC_LOCALS(o);
  const double dt = o->c.dt, eps = 0.5;
  const double *z_r = o->z_r, *Hz = o->Hz, *pm = o->pm, *pn = o->pn, *pden = o->pden;
  double *FE = S, *FX = S + nij, *FS = S + 2 * nij, *dTdr = S + 4 * nij, *dTdx = S + 6 * nij, *dTde = S + 8 * nij,
         *dRdx = S + 10 * nij, *dRde = S + 12 * nij;
  double cff, cff1, cff2, cff3, cff4;
  int k1, k2 = 1;
  for (int k = 0; k <= N; k++) {
    k1 = k2;
    k2 = 3 - k1;
    if (k < N) {
      for (int j = j0; j <= j1; j++)
        for (int i = i0; i <= i1 + 1; i++) {
          cff = 0.5 * (pm[X2(i, j)] + pm[X2(i - 1, j)]);
          if (o->c.options & ORC_MASKING) cff = cff * o->umask[X2(i, j)];
          if (o->wet_dry) cff = cff * o->umask_wet[X2(i, j)];
          L2(dRdx, i, j, k2) = cff * (pden[X3(i, j, k + 1)] - pden[X3(i - 1, j, k + 1)]);
          L2(dTdx, i, j, k2) = cff * (A[X3(i, j, k + 1)] - A[X3(i - 1, j, k + 1)]);
        }
      for (int j = j0; j <= j1 + 1; j++)
        for (int i = i0; i <= i1; i++) {
          cff = 0.5 * (pn[X2(i, j)] + pn[X2(i, j - 1)]);
          if (o->c.options & ORC_MASKING) cff = cff * o->vmask[X2(i, j)];
          if (o->wet_dry) cff = cff * o->vmask_wet[X2(i, j)];
          L2(dRde, i, j, k2) = cff * (pden[X3(i, j, k + 1)] - pden[X3(i, j - 1, k + 1)]);
          L2(dTde, i, j, k2) = cff * (A[X3(i, j, k + 1)] - A[X3(i, j - 1, k + 1)]);
        }
    }
    if (k == 0 || k == N) {
      for (int j = j0 - 1; j <= j1 + 1; j++)
        for (int i = i0 - 1; i <= i1 + 1; i++) { L2(dTdr, i, j, k2) = 0.0; L2(FS, i, j, k2) = 0.0; }
    } else {
      for (int j = j0 - 1; j <= j1 + 1; j++)
        for (int i = i0 - 1; i <= i1 + 1; i++) {
          cff1 = MAX(pden[X3(i, j, k)] - pden[X3(i, j, k + 1)], eps);
          cff = -1.0 / cff1;
          L2(dTdr, i, j, k2) = cff * (A[X3(i, j, k + 1)] - A[X3(i, j, k)]);
          L2(FS, i, j, k2) = cff * (z_r[X3(i, j, k + 1)] - z_r[X3(i, j, k)]);
        }
    }
    if (k > 0) {
      for (int j = j0; j <= j1; j++)
        for (int i = i0; i <= i1 + 1; i++) {
          cff = 0.25 * (d4[X2(i, j)] + d4[X2(i - 1, j)]) * o->on_u[X2(i, j)];
          FX[X2(i, j)] = cff * (Hz[X3(i, j, k)] + Hz[X3(i - 1, j, k)]) *
                         (L2(dTdx, i, j, k1) -
                          0.5 * (MAX(L2(dRdx, i, j, k1), 0.0) * (L2(dTdr, i - 1, j, k1) + L2(dTdr, i, j, k2)) +
                                 MIN(L2(dRdx, i, j, k1), 0.0) * (L2(dTdr, i - 1, j, k2) + L2(dTdr, i, j, k1))));
        }
      for (int j = j0; j <= j1 + 1; j++)
        for (int i = i0; i <= i1; i++) {
          cff = 0.25 * (d4[X2(i, j)] + d4[X2(i, j - 1)]) * o->om_v[X2(i, j)];
          FE[X2(i, j)] = cff * (Hz[X3(i, j, k)] + Hz[X3(i, j - 1, k)]) *
                         (L2(dTde, i, j, k1) -
                          0.5 * (MAX(L2(dRde, i, j, k1), 0.0) * (L2(dTdr, i, j - 1, k1) + L2(dTdr, i, j, k2)) +
                                 MIN(L2(dRde, i, j, k1), 0.0) * (L2(dTdr, i, j - 1, k2) + L2(dTdr, i, j, k1))));
        }
      if (k < N) {
        for (int j = j0; j <= j1; j++)
          for (int i = i0; i <= i1; i++) {
            const double difx = 0.5 * d4[X2(i, j)], dife = difx;
            cff1 = MAX(L2(dRdx, i, j, k1), 0.0);
            cff2 = MAX(L2(dRdx, i + 1, j, k2), 0.0);
            cff3 = MIN(L2(dRdx, i, j, k2), 0.0);
            cff4 = MIN(L2(dRdx, i + 1, j, k1), 0.0);
            cff = difx * (cff1 * (cff1 * L2(dTdr, i, j, k2) - L2(dTdx, i, j, k1)) +
                          cff2 * (cff2 * L2(dTdr, i, j, k2) - L2(dTdx, i + 1, j, k2)) +
                          cff3 * (cff3 * L2(dTdr, i, j, k2) - L2(dTdx, i, j, k2)) +
                          cff4 * (cff4 * L2(dTdr, i, j, k2) - L2(dTdx, i + 1, j, k1)));
            cff1 = MAX(L2(dRde, i, j, k1), 0.0);
            cff2 = MAX(L2(dRde, i, j + 1, k2), 0.0);
            cff3 = MIN(L2(dRde, i, j, k2), 0.0);
            cff4 = MIN(L2(dRde, i, j + 1, k1), 0.0);
            cff = cff + dife * (cff1 * (cff1 * L2(dTdr, i, j, k2) - L2(dTde, i, j, k1)) +
                                cff2 * (cff2 * L2(dTdr, i, j, k2) - L2(dTde, i, j + 1, k2)) +
                                cff3 * (cff3 * L2(dTdr, i, j, k2) - L2(dTde, i, j, k2)) +
                                cff4 * (cff4 * L2(dTdr, i, j, k2) - L2(dTde, i, j + 1, k1)));
            L2(FS, i, j, k2) = cff * L2(FS, i, j, k2);
          }
      }
      if (LapT) {                                                   /* :488-497 */
        for (int j = j0; j <= j1; j++)
          for (int i = i0; i <= i1; i++) {
            cff = pm[X2(i, j)] * pn[X2(i, j)];
            cff1 = 1.0 / Hz[X3(i, j, k)];
            LapT[X3(i, j, k)] = cff1 * (cff * (FX[X2(i + 1, j)] - FX[X2(i, j)] + FE[X2(i, j + 1)] - FE[X2(i, j)]) +
                                        (L2(FS, i, j, k2) - L2(FS, i, j, k1)));
          }
      } else {                                                      /* :791-806 */
        for (int j = j0; j <= j1; j++)
          for (int i = i0; i <= i1; i++) {
            cff = dt * pm[X2(i, j)] * pn[X2(i, j)];
            cff1 = cff * (FX[X2(i + 1, j)] - FX[X2(i, j)]);
            cff2 = cff * (FE[X2(i, j + 1)] - FE[X2(i, j)]);
            cff3 = dt * (L2(FS, i, j, k2) - L2(FS, i, j, k1));
            cff4 = cff1 + cff2 + cff3;
            tnew[X3(i, j, k)] = tnew[X3(i, j, k)] - cff4;
            if (o->dia) {                                           /* DIAGNOSTICS_TS :799-804 */
              orc_dia_wrk(o, ORC_DIA_XDIF, itrc)[X3(i, j, k)] = -cff1;
              orc_dia_wrk(o, ORC_DIA_YDIF, itrc)[X3(i, j, k)] = -cff2;
              orc_dia_wrk(o, ORC_DIA_SDIF, itrc)[X3(i, j, k)] = -cff3;
              orc_dia_wrk(o, ORC_DIA_HDIF, itrc)[X3(i, j, k)] = -cff4;
            }
          }
      }
    }
  }
}

static void t3dmix4_rot(orc_t *o, int tile, int iso) {
  ORC_LOCALS(o);
  const orc_bounds *b = &o->b[tile];
  const orc_cfg *c = &o->c;
  const int nrhs = o->s.nrhs, nnew = o->s.nnew;
  const int Istr = b->Istr, Iend = b->Iend, Jstr = b->Jstr, Jend = b->Jend;
  double *S = (double *)calloc(14 * nij + (size_t)nij * N, sizeof(double)), *LapT = S + 14 * nij;
  int Imin, Imax, Jmin, Jmax;                                      /* :228-245 */
  if (c->EWperiodic) { Imin = Istr - 1; Imax = Iend + 1; }
  else { Imin = Istr - 1 > 1 ? Istr - 1 : 1; Imax = Iend + 1 < c->Lm ? Iend + 1 : c->Lm; }
  if (c->NSperiodic) { Jmin = Jstr - 1; Jmax = Jend + 1; }
  else { Jmin = Jstr - 1 > 1 ? Jstr - 1 : 1; Jmax = Jend + 1 < c->Mm ? Jend + 1 : c->Mm; }
  for (int itrc = 1; itrc <= c->NT; itrc++) {
    const double *d4 = o->diff4 + (size_t)(itrc - 1) * nij;
    const int var = ORC_ISTVAR + itrc - 1;
    (iso ? iso4_op : geo4_op)(o, o->t + XT(LBi, LBj, 1, nrhs, itrc), d4, Imin, Imax, Jmin, Jmax, LapT, NULL, S, itrc);
    double *L = LapT;
    lap_walls(o, b, L, var, Imin, Imax, Jmin, Jmax);               /* :475-600 */
    (iso ? iso4_op : geo4_op)(o, L, d4, Istr, Iend, Jstr, Jend, NULL, o->t + XT(LBi, LBj, 1, nnew, itrc), S, itrc);
  }
  free(S);
}

void orc_t3dmix4_geo(orc_t *o, int tile) { t3dmix4_rot(o, tile, 0); }
void orc_t3dmix4_iso(orc_t *o, int tile) { t3dmix4_rot(o, tile, 1); }   /* t3dmix4_iso.h:98-812 */
