/*
 * orc_uvmix_geo.c -- harmonic viscosity along geopotential surfaces (UV_VIS2 + MIX_GEO_UV): the rotated stress tensor.
 * TEST INFRASTRUCTURE (see orc.h).
 *
 * orc_uv3dmix4_geo (round 6): uv3dmix4_geo_tile, ROMS/Nonlinear/uv3dmix4_geo.h:296-1478 (UV_VIS4 + MIX_GEO_UV) -- the same operator twice
 * (geo_uv_op, modes 1 and 2); pinned against the reference built from oracle/ref/upwelling_bihgeouv.h.
 * orc_uv3dmix2_geo follows uv3dmix2_geo_tile, ROMS/Nonlinear/uv3dmix2_geo.h:130-757, statement by statement (the two-level
 * k1/k2 rolling buffers of the reference; no VISC_3DCOEF; the DIAGNOSTICS_UV statements :706-714, :730-738 are not carried).
 * PARITY: pinned bit for bit against the reference built from oracle/ref/upwelling_geouv.h (MASKING: its rho- and psi-mask
 * statements included) -- tests/test_oracle_vs_ref.py.
 */
#include "orc.h"
#include <stdlib.h>
#define MAX(a, b) ((a) > (b) ? (a) : (b))
#define MIN(a, b) ((a) < (b) ? (a) : (b))
#define L2(A, i, j, l) A[X2(i, j) + (size_t)((l) - 1) * nij]

void orc_set_geouv(orc_t *o, int on) { o->mix_geo_uv = on != 0; }

/* One rotated operator of the stress tensor.  mode 0: uv3dmix2_geo_tile (harmonic: on u, v(nrhs) with visc2, added to rufrc /
   rvfrc and u, v(nnew)).  mode 1: the FIRST operator of uv3dmix4_geo_tile, uv3dmix4_geo.h:323-801 -- the same statements on the
   ranges widened by one point (Istrm1 ... Iendp2: the reference's names, clamped at the domain edges), without the thickness
   in the horizontal fluxes (:494-585), with visc4 (the square root of the biharmonic coefficient), its result the operator
   itself, LapU / LapV (:759-800).  mode 2: the SECOND operator, :982-1476 -- the harmonic statements again, on LapU / LapV
   with visc4, SUBTRACTED from rufrc / rvfrc and u, v(nnew).  (tests/test_oracle_vs_ref.py pins all three.) */
static void geo_uv_op(orc_t *o, int tile, int mode, const double *U, const double *V, const double *visc2_r, const double *visc2_p,
                      double *LapU, double *LapV) {
  ORC_LOCALS(o);
  const orc_bounds *b = &o->b[tile];
  const int nrhs = o->s.nrhs, nnew = o->s.nnew;
  const int Istr = b->Istr, Iend = b->Iend, Jstr = b->Jstr, Jend = b->Jend, IstrU = b->IstrU, JstrV = b->JstrV;
  const int msk = (o->c.options & ORC_MASKING) != 0, wet = o->wet_dry;
  const double dt = o->c.dt;
  double *u = o->u, *v = o->v, *Hz = o->Hz, *z_r = o->z_r, *pm = o->pm, *pn = o->pn;
  double *om_r = o->om_r, *on_r = o->on_r, *om_p = o->om_p, *on_p = o->on_p, *om_u = o->om_u, *on_u = o->on_u, *om_v = o->om_v,
         *on_v = o->on_v;
  const int x = mode == 1;      /* the widened ranges of the first biharmonic operator */
  /* lower bounds "X-1" | "X", upper bounds "Y+1" | "Y" of the harmonic loops and what uv3dmix4_geo.h:330-782 has in their place */
  const int jS1 = x ? b->Jstrm2 : Jstr - 1, jS0 = x ? b->Jstrm1 : Jstr, jV1 = x ? b->JstrVm2 : JstrV - 1, jV0 = x ? b->JstrVm1 : JstrV;
  const int iS1 = x ? b->Istrm2 : Istr - 1, iS0 = x ? b->Istrm1 : Istr, iU1 = x ? b->IstrUm2 : IstrU - 1, iU0 = x ? b->IstrUm1 : IstrU;
  const int jE1 = x ? b->Jendp2 : Jend + 1, jE0 = x ? b->Jendp1 : Jend, iE1 = x ? b->Iendp2 : Iend + 1, iE0 = x ? b->Iendp1 : Iend;
  double *S = (double *)calloc(32 * nij, sizeof(double));
  double *UFe = S, *VFe = S + nij, *UFx = S + 2 * nij, *VFx = S + 3 * nij;
  double *UFse = S + 4 * nij, *UFsx = S + 6 * nij, *VFse = S + 8 * nij, *VFsx = S + 10 * nij, *dmUde = S + 12 * nij, *dmVde = S + 14 * nij,
         *dnUdx = S + 16 * nij, *dnVdx = S + 18 * nij, *dUdz = S + 20 * nij, *dVdz = S + 22 * nij, *dZde_p = S + 24 * nij, *dZde_r = S + 26 * nij,
         *dZdx_p = S + 28 * nij, *dZdx_r = S + 30 * nij;
  double cff, fac1, fac2, pm_p, pn_p, cff1, cff2, cff3, cff4, cff5, cff6, cff7, cff8, dmUdz, dnUdz, dmVdz, dnVdz;
  int k1, k2 = 1;
  for (int k = 0; k <= N; k++) {                                   /* K_LOOP :293 */
    k1 = k2;
    k2 = 3 - k1;
    if (k < N) {
      for (int j = jS1; j <= jE1; j++)                   /* slopes at u- and v-points :301-326 */
        for (int i = iU1; i <= iE1; i++) {
          cff = 0.5 * (pm[X2(i - 1, j)] + pm[X2(i, j)]);
          if (msk) cff = cff * o->umask[X2(i, j)];
          if (wet) cff = cff * o->umask_wet[X2(i, j)];
          UFx[X2(i, j)] = cff * (z_r[X3(i, j, k + 1)] - z_r[X3(i - 1, j, k + 1)]);
        }
      for (int j = jV1; j <= jE1; j++)
        for (int i = iS1; i <= iE1; i++) {
          cff = 0.5 * (pn[X2(i, j - 1)] + pn[X2(i, j)]);
          if (msk) cff = cff * o->vmask[X2(i, j)];
          if (wet) cff = cff * o->vmask_wet[X2(i, j)];
          VFe[X2(i, j)] = cff * (z_r[X3(i, j, k + 1)] - z_r[X3(i, j - 1, k + 1)]);
        }
      for (int j = jS0; j <= jE1; j++)                       /* :328-334 */
        for (int i = iS0; i <= iE1; i++) {
          L2(dZdx_p, i, j, k2) = 0.5 * (UFx[X2(i, j - 1)] + UFx[X2(i, j)]);
          L2(dZde_p, i, j, k2) = 0.5 * (VFe[X2(i - 1, j)] + VFe[X2(i, j)]);
        }
      for (int j = jV1; j <= jE0; j++)                      /* :335-342 */
        for (int i = iU1; i <= iE0; i++) {
          L2(dZdx_r, i, j, k2) = 0.5 * (UFx[X2(i, j)] + UFx[X2(i + 1, j)]);
          L2(dZde_r, i, j, k2) = 0.5 * (VFe[X2(i, j)] + VFe[X2(i, j + 1)]);
        }
      for (int j = jV1; j <= jE0; j++)                      /* momentum gradients :346-412 */
        for (int i = iU1; i <= iE0; i++) {
          cff = 0.5 * pm[X2(i, j)];
          if (msk) cff = cff * o->rmask[X2(i, j)];
          if (wet) cff = cff * o->rmask_wet[X2(i, j)];
          L2(dnUdx, i, j, k2) = cff * ((pn[X2(i, j)] + pn[X2(i + 1, j)]) * U[X3(i + 1, j, k + 1)] -
                                      (pn[X2(i - 1, j)] + pn[X2(i, j)]) * U[X3(i, j, k + 1)]);
        }
      for (int j = jS0; j <= jE1; j++)
        for (int i = iS0; i <= iE1; i++) {
          cff = 0.125 * (pn[X2(i - 1, j)] + pn[X2(i, j)] + pn[X2(i - 1, j - 1)] + pn[X2(i, j - 1)]);
          if (msk) cff = cff * o->pmask[X2(i, j)];
          if (wet) cff = cff * o->pmask_wet[X2(i, j)];
          L2(dmUde, i, j, k2) = cff * ((pm[X2(i - 1, j)] + pm[X2(i, j)]) * U[X3(i, j, k + 1)] -
                                      (pm[X2(i - 1, j - 1)] + pm[X2(i, j - 1)]) * U[X3(i, j - 1, k + 1)]);
        }
      for (int j = jS0; j <= jE1; j++)
        for (int i = iS0; i <= iE1; i++) {
          cff = 0.125 * (pm[X2(i - 1, j)] + pm[X2(i, j)] + pm[X2(i - 1, j - 1)] + pm[X2(i, j - 1)]);
          if (msk) cff = cff * o->pmask[X2(i, j)];
          if (wet) cff = cff * o->pmask_wet[X2(i, j)];
          L2(dnVdx, i, j, k2) = cff * ((pn[X2(i, j - 1)] + pn[X2(i, j)]) * V[X3(i, j, k + 1)] -
                                      (pn[X2(i - 1, j - 1)] + pn[X2(i - 1, j)]) * V[X3(i - 1, j, k + 1)]);
        }
      for (int j = jV1; j <= jE0; j++)
        for (int i = iU1; i <= iE0; i++) {
          cff = 0.5 * pn[X2(i, j)];
          if (msk) cff = cff * o->rmask[X2(i, j)];
          if (wet) cff = cff * o->rmask_wet[X2(i, j)];
          L2(dmVde, i, j, k2) = cff * ((pm[X2(i, j)] + pm[X2(i, j + 1)]) * V[X3(i, j + 1, k + 1)] -
                                      (pm[X2(i, j - 1)] + pm[X2(i, j)]) * V[X3(i, j, k + 1)]);
        }
    }
    if (k == 0 || k == N) {                                        /* :415-438 */
      for (int j = jS1; j <= jE1; j++)
        for (int i = iU1; i <= iE1; i++) L2(dUdz, i, j, k2) = 0.0;
      for (int j = jV1; j <= jE1; j++)
        for (int i = iS1; i <= iE1; i++) L2(dVdz, i, j, k2) = 0.0;
      for (int j = jS0; j <= jE0; j++)
        for (int i = iU0; i <= iE0; i++) { L2(UFsx, i, j, k2) = 0.0; L2(UFse, i, j, k2) = 0.0; }
      for (int j = jV0; j <= jE0; j++)
        for (int i = iS0; i <= iE0; i++) { L2(VFsx, i, j, k2) = 0.0; L2(VFse, i, j, k2) = 0.0; }
    } else {                                                       /* :439-458 */
      for (int j = jS1; j <= jE1; j++)
        for (int i = iU1; i <= iE1; i++) {
          cff = 1.0 / (0.5 * (z_r[X3(i - 1, j, k + 1)] - z_r[X3(i - 1, j, k)] + z_r[X3(i, j, k + 1)] - z_r[X3(i, j, k)]));
          L2(dUdz, i, j, k2) = cff * (U[X3(i, j, k + 1)] - U[X3(i, j, k)]);
        }
      for (int j = jV1; j <= jE1; j++)
        for (int i = iS1; i <= iE1; i++) {
          cff = 1.0 / (0.5 * (z_r[X3(i, j - 1, k + 1)] - z_r[X3(i, j - 1, k)] + z_r[X3(i, j, k + 1)] - z_r[X3(i, j, k)]));
          L2(dVdz, i, j, k2) = cff * (V[X3(i, j, k + 1)] - V[X3(i, j, k)]);
        }
    }
    if (k > 0) {
      for (int j = jV1; j <= jE0; j++)                      /* rotated flux at rho-points :464-497 */
        for (int i = iU1; i <= iE0; i++) {
          cff1 = MIN(L2(dZdx_r, i, j, k1), 0.0);
          cff2 = MAX(L2(dZdx_r, i, j, k1), 0.0);
          cff3 = MIN(L2(dZde_r, i, j, k1), 0.0);
          cff4 = MAX(L2(dZde_r, i, j, k1), 0.0);
          cff = (on_r[X2(i, j)] * (L2(dnUdx, i, j, k1) -
                                   0.5 * pn[X2(i, j)] * (cff1 * (L2(dUdz, i, j, k1) + L2(dUdz, i + 1, j, k2)) +
                                                          cff2 * (L2(dUdz, i, j, k2) + L2(dUdz, i + 1, j, k1)))) -
                 om_r[X2(i, j)] * (L2(dmVde, i, j, k1) -
                                   0.5 * pm[X2(i, j)] * (cff3 * (L2(dVdz, i, j, k1) + L2(dVdz, i, j + 1, k2)) +
                                                          cff4 * (L2(dVdz, i, j, k2) + L2(dVdz, i, j + 1, k1)))));
          if (!x) cff = Hz[X3(i, j, k)] * cff;                          /* (the first biharmonic operator: no thickness, uv3dmix4_geo.h:501) */
          if (msk) cff = cff * o->rmask[X2(i, j)];
          if (wet) cff = cff * o->rmask_wet[X2(i, j)];
          UFx[X2(i, j)] = on_r[X2(i, j)] * on_r[X2(i, j)] * visc2_r[X2(i, j)] * cff;
          VFe[X2(i, j)] = om_r[X2(i, j)] * om_r[X2(i, j)] * visc2_r[X2(i, j)] * cff;
        }
      for (int j = jS0; j <= jE1; j++)                       /* ... at psi-points :499-543 */
        for (int i = iS0; i <= iE1; i++) {
          pm_p = 0.25 * (pm[X2(i - 1, j - 1)] + pm[X2(i - 1, j)] + pm[X2(i, j - 1)] + pm[X2(i, j)]);
          pn_p = 0.25 * (pn[X2(i - 1, j - 1)] + pn[X2(i - 1, j)] + pn[X2(i, j - 1)] + pn[X2(i, j)]);
          cff1 = MIN(L2(dZdx_p, i, j, k1), 0.0);
          cff2 = MAX(L2(dZdx_p, i, j, k1), 0.0);
          cff3 = MIN(L2(dZde_p, i, j, k1), 0.0);
          cff4 = MAX(L2(dZde_p, i, j, k1), 0.0);
          cff = (on_p[X2(i, j)] * (L2(dnVdx, i, j, k1) -
                                   0.5 * pn_p * (cff1 * (L2(dVdz, i - 1, j, k1) + L2(dVdz, i, j, k2)) +
                                                 cff2 * (L2(dVdz, i - 1, j, k2) + L2(dVdz, i, j, k1)))) +
                 om_p[X2(i, j)] * (L2(dmUde, i, j, k1) -
                                   0.5 * pm_p * (cff3 * (L2(dUdz, i, j - 1, k1) + L2(dUdz, i, j, k2)) +
                                                 cff4 * (L2(dUdz, i, j - 1, k2) + L2(dUdz, i, j, k1)))));
          if (!x) cff = 0.25 * (Hz[X3(i - 1, j, k)] + Hz[X3(i, j, k)] + Hz[X3(i - 1, j - 1, k)] + Hz[X3(i, j - 1, k)]) * cff;     /* (:540) */
          if (msk) cff = cff * o->pmask[X2(i, j)];
          if (wet) cff = cff * o->pmask_wet[X2(i, j)];
          UFe[X2(i, j)] = om_p[X2(i, j)] * om_p[X2(i, j)] * visc2_p[X2(i, j)] * cff;
          VFx[X2(i, j)] = on_p[X2(i, j)] * on_p[X2(i, j)] * visc2_p[X2(i, j)] * cff;
        }
      if (k < N) {                                                 /* vertical flux due to the sloping surfaces :548-690 */
        for (int j = jS0; j <= jE0; j++)
          for (int i = iU0; i <= iE0; i++) {
            cff = 0.25 * (visc2_r[X2(i - 1, j)] + visc2_r[X2(i, j)]);
            fac1 = cff * on_u[X2(i, j)];
            fac2 = cff * om_u[X2(i, j)];
            cff = 0.5 * (pn[X2(i - 1, j)] + pn[X2(i, j)]);
            dnUdz = cff * L2(dUdz, i, j, k2);
            dnVdz = cff * 0.25 * (L2(dVdz, i - 1, j + 1, k2) + L2(dVdz, i, j + 1, k2) + L2(dVdz, i - 1, j, k2) + L2(dVdz, i, j, k2));
            cff = 0.5 * (pm[X2(i - 1, j)] + pm[X2(i, j)]);
            dmUdz = cff * L2(dUdz, i, j, k2);
            dmVdz = cff * 0.25 * (L2(dVdz, i - 1, j + 1, k2) + L2(dVdz, i, j + 1, k2) + L2(dVdz, i - 1, j, k2) + L2(dVdz, i, j, k2));
            cff1 = MIN(L2(dZdx_r, i - 1, j, k1), 0.0);
            cff2 = MIN(L2(dZdx_r, i, j, k2), 0.0);
            cff3 = MAX(L2(dZdx_r, i - 1, j, k2), 0.0);
            cff4 = MAX(L2(dZdx_r, i, j, k1), 0.0);
            L2(UFsx, i, j, k2) = fac1 * (cff1 * (cff1 * dnUdz - L2(dnUdx, i - 1, j, k1)) + cff2 * (cff2 * dnUdz - L2(dnUdx, i, j, k2)) +
                                         cff3 * (cff3 * dnUdz - L2(dnUdx, i - 1, j, k2)) + cff4 * (cff4 * dnUdz - L2(dnUdx, i, j, k1)));
            cff1 = MIN(L2(dZde_p, i, j, k1), 0.0);
            cff2 = MIN(L2(dZde_p, i, j + 1, k2), 0.0);
            cff3 = MAX(L2(dZde_p, i, j, k2), 0.0);
            cff4 = MAX(L2(dZde_p, i, j + 1, k1), 0.0);
            L2(UFse, i, j, k2) = fac2 * (cff1 * (cff1 * dmUdz - L2(dmUde, i, j, k1)) + cff2 * (cff2 * dmUdz - L2(dmUde, i, j + 1, k2)) +
                                         cff3 * (cff3 * dmUdz - L2(dmUde, i, j, k2)) + cff4 * (cff4 * dmUdz - L2(dmUde, i, j + 1, k1)));
            cff1 = MIN(L2(dZde_p, i, j, k1), 0.0);
            cff2 = MIN(L2(dZde_p, i, j + 1, k2), 0.0);
            cff3 = MAX(L2(dZde_p, i, j, k2), 0.0);
            cff4 = MAX(L2(dZde_p, i, j + 1, k1), 0.0);
            cff5 = MIN(L2(dZdx_p, i, j, k1), 0.0);
            cff6 = MIN(L2(dZdx_p, i, j + 1, k2), 0.0);
            cff7 = MAX(L2(dZdx_p, i, j, k2), 0.0);
            cff8 = MAX(L2(dZdx_p, i, j + 1, k1), 0.0);
            L2(UFsx, i, j, k2) = L2(UFsx, i, j, k2) +
                                 fac1 * (cff1 * (cff5 * dnVdz - L2(dnVdx, i, j, k1)) + cff2 * (cff6 * dnVdz - L2(dnVdx, i, j + 1, k2)) +
                                         cff3 * (cff7 * dnVdz - L2(dnVdx, i, j, k2)) + cff4 * (cff8 * dnVdz - L2(dnVdx, i, j + 1, k1)));
            cff1 = MIN(L2(dZdx_r, i - 1, j, k1), 0.0);
            cff2 = MIN(L2(dZdx_r, i, j, k2), 0.0);
            cff3 = MAX(L2(dZdx_r, i - 1, j, k2), 0.0);
            cff4 = MAX(L2(dZdx_r, i, j, k1), 0.0);
            cff5 = MIN(L2(dZde_r, i - 1, j, k1), 0.0);
            cff6 = MIN(L2(dZde_r, i, j, k2), 0.0);
            cff7 = MAX(L2(dZde_r, i - 1, j, k2), 0.0);
            cff8 = MAX(L2(dZde_r, i, j, k1), 0.0);
            L2(UFse, i, j, k2) = L2(UFse, i, j, k2) -
                                 fac2 * (cff1 * (cff5 * dmVdz - L2(dmVde, i - 1, j, k1)) + cff2 * (cff6 * dmVdz - L2(dmVde, i, j, k2)) +
                                         cff3 * (cff7 * dmVdz - L2(dmVde, i - 1, j, k2)) + cff4 * (cff8 * dmVdz - L2(dmVde, i, j, k1)));
          }
        for (int j = jV0; j <= jE0; j++)
          for (int i = iS0; i <= iE0; i++) {
            cff = 0.25 * (visc2_r[X2(i, j - 1)] + visc2_r[X2(i, j)]);
            fac1 = cff * on_v[X2(i, j)];
            fac2 = cff * om_v[X2(i, j)];
            cff = 0.5 * (pn[X2(i, j - 1)] + pn[X2(i, j)]);
            dnUdz = cff * 0.25 * (L2(dUdz, i, j, k2) + L2(dUdz, i + 1, j, k2) + L2(dUdz, i, j - 1, k2) + L2(dUdz, i + 1, j - 1, k2));
            dnVdz = cff * L2(dVdz, i, j, k2);
            cff = 0.5 * (pm[X2(i, j - 1)] + pm[X2(i, j)]);
            dmUdz = cff * 0.25 * (L2(dUdz, i, j, k2) + L2(dUdz, i + 1, j, k2) + L2(dUdz, i, j - 1, k2) + L2(dUdz, i + 1, j - 1, k2));
            dmVdz = cff * L2(dVdz, i, j, k2);
            cff1 = MIN(L2(dZdx_p, i, j, k1), 0.0);
            cff2 = MIN(L2(dZdx_p, i + 1, j, k2), 0.0);
            cff3 = MAX(L2(dZdx_p, i, j, k2), 0.0);
            cff4 = MAX(L2(dZdx_p, i + 1, j, k1), 0.0);
            L2(VFsx, i, j, k2) = fac1 * (cff1 * (cff1 * dnVdz - L2(dnVdx, i, j, k1)) + cff2 * (cff2 * dnVdz - L2(dnVdx, i + 1, j, k2)) +
                                         cff3 * (cff3 * dnVdz - L2(dnVdx, i, j, k2)) + cff4 * (cff4 * dnVdz - L2(dnVdx, i + 1, j, k1)));
            cff1 = MIN(L2(dZde_r, i, j - 1, k1), 0.0);
            cff2 = MIN(L2(dZde_r, i, j, k2), 0.0);
            cff3 = MAX(L2(dZde_r, i, j - 1, k2), 0.0);
            cff4 = MAX(L2(dZde_r, i, j, k1), 0.0);
            L2(VFse, i, j, k2) = fac2 * (cff1 * (cff1 * dmVdz - L2(dmVde, i, j - 1, k1)) + cff2 * (cff2 * dmVdz - L2(dmVde, i, j, k2)) +
                                         cff3 * (cff3 * dmVdz - L2(dmVde, i, j - 1, k2)) + cff4 * (cff4 * dmVdz - L2(dmVde, i, j, k1)));
            cff1 = MIN(L2(dZde_r, i, j - 1, k1), 0.0);
            cff2 = MIN(L2(dZde_r, i, j, k2), 0.0);
            cff3 = MAX(L2(dZde_r, i, j - 1, k2), 0.0);
            cff4 = MAX(L2(dZde_r, i, j, k1), 0.0);
            cff5 = MIN(L2(dZdx_r, i, j - 1, k1), 0.0);
            cff6 = MIN(L2(dZdx_r, i, j, k2), 0.0);
            cff7 = MAX(L2(dZdx_r, i, j - 1, k2), 0.0);
            cff8 = MAX(L2(dZdx_r, i, j, k1), 0.0);
            L2(VFsx, i, j, k2) = L2(VFsx, i, j, k2) -
                                 fac1 * (cff1 * (cff5 * dnUdz - L2(dnUdx, i, j - 1, k1)) + cff2 * (cff6 * dnUdz - L2(dnUdx, i, j, k2)) +
                                         cff3 * (cff7 * dnUdz - L2(dnUdx, i, j - 1, k2)) + cff4 * (cff8 * dnUdz - L2(dnUdx, i, j, k1)));
            cff1 = MIN(L2(dZdx_p, i, j, k1), 0.0);
            cff2 = MIN(L2(dZdx_p, i + 1, j, k2), 0.0);
            cff3 = MAX(L2(dZdx_p, i, j, k2), 0.0);
            cff4 = MAX(L2(dZdx_p, i + 1, j, k1), 0.0);
            cff5 = MIN(L2(dZde_p, i, j, k1), 0.0);
            cff6 = MIN(L2(dZde_p, i + 1, j, k2), 0.0);
            cff7 = MAX(L2(dZde_p, i, j, k2), 0.0);
            cff8 = MAX(L2(dZde_p, i + 1, j, k1), 0.0);
            L2(VFse, i, j, k2) = L2(VFse, i, j, k2) +
                                 fac2 * (cff1 * (cff5 * dmUdz - L2(dmUde, i, j, k1)) + cff2 * (cff6 * dmUdz - L2(dmUde, i + 1, j, k2)) +
                                         cff3 * (cff7 * dmUdz - L2(dmUde, i, j, k2)) + cff4 * (cff8 * dmUdz - L2(dmUde, i + 1, j, k1)));
          }
      }
      if (x) {                                                     /* the first operator itself, uv3dmix4_geo.h:759-800 */
        for (int j = jS0; j <= jE0; j++)
          for (int i = iU0; i <= iE0; i++) {
            cff = 0.125 * (pm[X2(i - 1, j)] + pm[X2(i, j)]) * (pn[X2(i - 1, j)] + pn[X2(i, j)]);
            cff1 = 1.0 / (0.5 * (Hz[X3(i - 1, j, k)] + Hz[X3(i, j, k)]));
            LapU[X3(i, j, k)] = cff * ((pn[X2(i - 1, j)] + pn[X2(i, j)]) * (UFx[X2(i, j)] - UFx[X2(i - 1, j)]) +
                                       (pm[X2(i - 1, j)] + pm[X2(i, j)]) * (UFe[X2(i, j + 1)] - UFe[X2(i, j)])) +
                                cff1 * ((L2(UFsx, i, j, k2) + L2(UFse, i, j, k2)) - (L2(UFsx, i, j, k1) + L2(UFse, i, j, k1)));
            if (msk) LapU[X3(i, j, k)] = LapU[X3(i, j, k)] * o->umask[X2(i, j)];
            if (wet) LapU[X3(i, j, k)] = LapU[X3(i, j, k)] * o->umask_wet[X2(i, j)];
          }
        for (int j = jV0; j <= jE0; j++)
          for (int i = iS0; i <= iE0; i++) {
            cff = 0.125 * (pm[X2(i, j)] + pm[X2(i, j - 1)]) * (pn[X2(i, j)] + pn[X2(i, j - 1)]);
            cff1 = 1.0 / (0.5 * (Hz[X3(i, j - 1, k)] + Hz[X3(i, j, k)]));
            LapV[X3(i, j, k)] = cff * ((pn[X2(i, j - 1)] + pn[X2(i, j)]) * (VFx[X2(i + 1, j)] - VFx[X2(i, j)]) -
                                       (pm[X2(i, j - 1)] + pm[X2(i, j)]) * (VFe[X2(i, j)] - VFe[X2(i, j - 1)])) +
                                cff1 * ((L2(VFsx, i, j, k2) + L2(VFse, i, j, k2)) - (L2(VFsx, i, j, k1) + L2(VFse, i, j, k1)));
            if (msk) LapV[X3(i, j, k)] = LapV[X3(i, j, k)] * o->vmask[X2(i, j)];
            if (wet) LapV[X3(i, j, k)] = LapV[X3(i, j, k)] * o->vmask_wet[X2(i, j)];
          }
        continue;
      }
      for (int j = jS0; j <= jE0; j++)                             /* time step :693-740 (harmonic) | uv3dmix4_geo.h:1430-1475 (subtracted) */
        for (int i = iU0; i <= iE0; i++) {
          cff = dt * 0.25 * (pm[X2(i - 1, j)] + pm[X2(i, j)]) * (pn[X2(i - 1, j)] + pn[X2(i, j)]);
          cff1 = 0.5 * (pn[X2(i - 1, j)] + pn[X2(i, j)]) * (UFx[X2(i, j)] - UFx[X2(i - 1, j)]);
          cff2 = 0.5 * (pm[X2(i - 1, j)] + pm[X2(i, j)]) * (UFe[X2(i, j + 1)] - UFe[X2(i, j)]);
          cff3 = L2(UFsx, i, j, k2) - L2(UFsx, i, j, k1);
          cff4 = L2(UFse, i, j, k2) - L2(UFse, i, j, k1);
          cff5 = cff * (cff1 + cff2);
          cff6 = dt * (cff3 + cff4);
          if (mode == 2) {
            o->rufrc[X2(i, j)] = o->rufrc[X2(i, j)] - cff1 - cff2 - cff3 - cff4;
            u[X4(i, j, k, nnew)] = u[X4(i, j, k, nnew)] - cff5 - cff6;
          } else {
            o->rufrc[X2(i, j)] = o->rufrc[X2(i, j)] + cff1 + cff2 + cff3 + cff4;
            u[X4(i, j, k, nnew)] = u[X4(i, j, k, nnew)] + cff5 + cff6;
          }
        }
      for (int j = jV0; j <= jE0; j++)
        for (int i = iS0; i <= iE0; i++) {
          cff = dt * 0.25 * (pm[X2(i, j)] + pm[X2(i, j - 1)]) * (pn[X2(i, j)] + pn[X2(i, j - 1)]);
          cff1 = 0.5 * (pn[X2(i, j - 1)] + pn[X2(i, j)]) * (VFx[X2(i + 1, j)] - VFx[X2(i, j)]);
          cff2 = 0.5 * (pm[X2(i, j - 1)] + pm[X2(i, j)]) * (VFe[X2(i, j)] - VFe[X2(i, j - 1)]);
          cff3 = L2(VFsx, i, j, k2) - L2(VFsx, i, j, k1);
          cff4 = L2(VFse, i, j, k2) - L2(VFse, i, j, k1);
          cff5 = cff * (cff1 - cff2);
          cff6 = dt * (cff3 + cff4);
          if (mode == 2) {
            o->rvfrc[X2(i, j)] = o->rvfrc[X2(i, j)] - cff1 + cff2 - cff3 - cff4;
            v[X4(i, j, k, nnew)] = v[X4(i, j, k, nnew)] - cff5 - cff6;
          } else {
            o->rvfrc[X2(i, j)] = o->rvfrc[X2(i, j)] + cff1 - cff2 + cff3 + cff4;
            v[X4(i, j, k, nnew)] = v[X4(i, j, k, nnew)] + cff5 + cff6;
          }
        }
    }
  }
  free(S);
}

void orc_uv3dmix2_geo(orc_t *o, int tile) {
  ORC_LOCALS(o);
  const size_t oU = (size_t)(o->s.nrhs - 1) * nij * (size_t)N;
  geo_uv_op(o, tile, 0, o->u + oU, o->v + oU, o->visc2_r, o->visc2_p, NULL, NULL);
}

/* uv3dmix4_geo_tile, ROMS/Nonlinear/uv3dmix4_geo.h:296-1478 (UV_VIS4 + MIX_GEO_UV): the rotated operator twice -- the first into
   LapU, LapV on the tile widened by one point, their closed / gradient conditions and corner averages (:803-975, the
   statements of uv3dmix4_s.h: orc_lap_bc), the second on LapU, LapV.  visc4_r, visc4_p hold the square root of the coefficient. */
void orc_uv3dmix4_geo(orc_t *o, int tile) {
  ORC_LOCALS(o);
  const orc_bounds *b = &o->b[tile];
  const size_t oU = (size_t)(o->s.nrhs - 1) * nij * (size_t)N;
  double *LapU = (double *)calloc(2 * nij * (size_t)N, sizeof(double)), *LapV = LapU + nij * (size_t)N;
  geo_uv_op(o, tile, 1, o->u + oU, o->v + oU, o->visc4_r, o->visc4_p, LapU, LapV);
  for (int k = 1; k <= N; k++)
    orc_lap_bc(o, b, LapU + (size_t)(k - 1) * nij, LapV + (size_t)(k - 1) * nij, ORC_ISUVEL, ORC_ISVVEL, b->IstrUm1, b->Iendp1, b->Jstrm1, b->Jendp1,
               b->Istrm1, b->Iendp1, b->JstrVm1, b->Jendp1);
  geo_uv_op(o, tile, 2, LapU, LapV, o->visc4_r, o->visc4_p, NULL, NULL);
  free(LapU);
}
