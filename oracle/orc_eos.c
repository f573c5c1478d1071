/*
 * orc_eos.c -- nonlinear equation of state (Jackett & McDougall 1992).
 * TEST INFRASTRUCTURE (see orc.h).
 *
 * Follows rho_eos_tile, ROMS/Nonlinear/rho_eos.F:247-560 (NONLIN_EOS, BV_FREQUENCY,
 * EOS_TDERIVATIVE as derived for LMD_SKPP/BULK_FLUXES: alpha, beta at the surface level only,
 * :470-500); coefficients ROMS/Modules/mod_eoscoef.F.
 * PARITY: pinned (rho_eos.F builds in oracle/_ref).
 */
#include "orc.h"
#include <math.h>
#include <stdlib.h>
#define MAX(a, b) ((a) > (b) ? (a) : (b))
#define CX(A, i, k) A[(size_t)((i) - LBi) + (size_t)(k) * ni]

static const double A00 = +1.909256e+04, A01 = +2.098925e+02, A02 = -3.041638e+00, A03 = -1.852732e-03,
  A04 = -1.361629e-05, B00 = +1.044077e+02, B01 = -6.500517e+00, B02 = +1.553190e-01, B03 = +2.326469e-04,
  D00 = -5.587545e+00, D01 = +7.390729e-01, D02 = -1.909078e-02, E00 = +4.721788e-01, E01 = +1.028859e-02,
  E02 = -2.512549e-04, E03 = -5.939910e-07, F00 = -1.571896e-02, F01 = -2.598241e-04, F02 = +7.267926e-06,
  G00 = +2.042967e-03, G01 = +1.045941e-05, G02 = -5.782165e-10, G03 = +1.296821e-07, H00 = -2.595994e-07,
  H01 = -1.248266e-09, H02 = -3.508914e-09, Q00 = +9.99842594e+02, Q01 = +6.793952e-02, Q02 = -9.095290e-03,
  Q03 = +1.001685e-04, Q04 = -1.120083e-06, Q05 = +6.536332e-09, U00 = +8.24493e-01, U01 = -4.08990e-03,
  U02 = +7.64380e-05, U03 = -8.24670e-07, U04 = +5.38750e-09, V00 = -5.72466e-03, V01 = +1.02270e-04,
  V02 = -1.65460e-06, W00 = +4.8314e-04;

void orc_eos_nonlinear(orc_t *o, int tile) {
  const int msk = (o->c.options & ORC_MASKING) != 0;
  ORC_LOCALS(o);
  const orc_bounds *b = &o->b[tile];
  const int nrhs = o->s.nrhs;
  const double g = o->c.g;
  double *t = o->t, *z_r = o->z_r, *z_w = o->z_w, *Hz = o->Hz;
  double *rhoA = o->rhoA, *rhoS = o->rhoS, *bvf = o->bvf;
  const size_t cs = ni * (size_t)(N + 1);
  double *W = (double *)calloc(13 * cs, sizeof(double));
  double *DbulkDS = W, *DbulkDT = W + cs, *Dden1DS = W + 2 * cs, *Dden1DT = W + 3 * cs, *Scof = W + 4 * cs,
         *Tcof = W + 5 * cs, *wrk = W + 6 * cs, *bulk = W + 7 * cs, *bulk0 = W + 8 * cs, *bulk1 = W + 9 * cs,
         *bulk2 = W + 10 * cs, *den = W + 11 * cs, *den1 = W + 12 * cs;
  double C[10], dCdT[10], cff, cff1, cff2;
  for (int j = b->JstrT; j <= b->JendT; j++) {
    for (int k = 1; k <= N; k++)
      for (int i = b->IstrT; i <= b->IendT; i++) {
        const double Tt = MAX(-2.0, t[XT(i, j, k, nrhs, 1)]);
        const double Ts = MAX(0.0, t[XT(i, j, k, nrhs, 2)]);
        const double sqrtTs = sqrt(Ts);
        const double Tp = z_r[X3(i, j, k)];
        const double Tpr10 = 0.1 * Tp;
        C[0] = Q00 + Tt * (Q01 + Tt * (Q02 + Tt * (Q03 + Tt * (Q04 + Tt * Q05))));
        C[1] = U00 + Tt * (U01 + Tt * (U02 + Tt * (U03 + Tt * U04)));
        C[2] = V00 + Tt * (V01 + Tt * V02);
        dCdT[0] = Q01 + Tt * (2.0 * Q02 + Tt * (3.0 * Q03 + Tt * (4.0 * Q04 + Tt * 5.0 * Q05)));
        dCdT[1] = U01 + Tt * (2.0 * U02 + Tt * (3.0 * U03 + Tt * 4.0 * U04));
        dCdT[2] = V01 + Tt * 2.0 * V02;
        CX(den1, i, k) = C[0] + Ts * (C[1] + sqrtTs * C[2] + Ts * W00);
        CX(Dden1DS, i, k) = C[1] + 1.5 * C[2] * sqrtTs + 2.0 * W00 * Ts;
        CX(Dden1DT, i, k) = dCdT[0] + Ts * (dCdT[1] + sqrtTs * dCdT[2]);
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
        CX(bulk0, i, k) = C[3] + Ts * (C[4] + sqrtTs * C[5]);
        CX(bulk1, i, k) = C[6] + Ts * (C[7] + sqrtTs * G00);
        CX(bulk2, i, k) = C[8] + Ts * C[9];
        CX(bulk, i, k) = CX(bulk0, i, k) - Tp * (CX(bulk1, i, k) - Tp * CX(bulk2, i, k));
        CX(DbulkDS, i, k) = C[4] + sqrtTs * 1.5 * C[5] - Tp * (C[7] + sqrtTs * 1.5 * G00 - Tp * C[9]);
        CX(DbulkDT, i, k) = dCdT[3] + Ts * (dCdT[4] + sqrtTs * dCdT[5]) -
                            Tp * (dCdT[6] + Ts * dCdT[7] - Tp * (dCdT[8] + Ts * dCdT[9]));
        cff = 1.0 / (CX(bulk, i, k) + Tpr10);
        CX(den, i, k) = CX(den1, i, k) * CX(bulk, i, k) * cff;
        CX(den, i, k) = CX(den, i, k) - 1000.0;
        if (msk) CX(den, i, k) = CX(den, i, k) * o->rmask[X2(i, j)];                       /* rho_eos.F:357 */
      }
    for (int i = b->IstrT; i <= b->IendT; i++) {
      cff1 = CX(den, i, N) * Hz[X3(i, j, N)];
      rhoS[X2(i, j)] = 0.5 * cff1 * Hz[X3(i, j, N)];
      rhoA[X2(i, j)] = cff1;
    }
    for (int k = N - 1; k >= 1; k--)
      for (int i = b->IstrT; i <= b->IendT; i++) {
        cff1 = CX(den, i, k) * Hz[X3(i, j, k)];
        rhoS[X2(i, j)] = rhoS[X2(i, j)] + Hz[X3(i, j, k)] * (rhoA[X2(i, j)] + 0.5 * cff1);
        rhoA[X2(i, j)] = rhoA[X2(i, j)] + cff1;
      }
    cff2 = 1.0 / o->c.rho0;
    for (int i = b->IstrT; i <= b->IendT; i++) {
      cff1 = 1.0 / (z_w[XW(i, j, N)] - z_w[XW(i, j, 0)]);
      rhoA[X2(i, j)] = cff2 * cff1 * rhoA[X2(i, j)];
      rhoS[X2(i, j)] = 2.0 * cff1 * cff1 * cff2 * rhoS[X2(i, j)];
    }
    /* Brunt-Vaisala frequency at W-points */
    for (int k = 1; k <= N - 1; k++)
      for (int i = b->IstrT; i <= b->IendT; i++) {
        const double zw = z_w[XW(i, j, k)];
        const double bulk_up = CX(bulk0, i, k + 1) - zw * (CX(bulk1, i, k + 1) - CX(bulk2, i, k + 1) * zw);
        const double bulk_dn = CX(bulk0, i, k) - zw * (CX(bulk1, i, k) - CX(bulk2, i, k) * zw);
        cff1 = 1.0 / (bulk_up + 0.1 * zw);
        cff2 = 1.0 / (bulk_dn + 0.1 * zw);
        const double den_up = cff1 * (CX(den1, i, k + 1) * bulk_up);
        const double den_dn = cff2 * (CX(den1, i, k) * bulk_dn);
        bvf[XW(i, j, k)] = -g * (den_up - den_dn) / (0.5 * (den_up + den_dn) * (z_r[X3(i, j, k + 1)] - z_r[X3(i, j, k)]));
      }
    for (int i = b->IstrT; i <= b->IendT; i++) { bvf[XW(i, j, 0)] = 0.0; bvf[XW(i, j, N)] = 0.0; }
    /* thermal expansion / saline contraction at the surface; LMD_DDMIX: at every level :435-455, their ratio kept */
    for (int k = o->ddmix ? 1 : N; k <= N; k++)
    for (int i = b->IstrT; i <= b->IendT; i++) {
      const double Tpr10 = 0.1 * z_r[X3(i, j, k)];
      cff = CX(bulk, i, k) + Tpr10;
      cff1 = Tpr10 * CX(den1, i, k);
      cff2 = CX(bulk, i, k) * cff;
      CX(wrk, i, k) = (CX(den, i, k) + 1000.0) * cff * cff;
      CX(Tcof, i, k) = -(CX(DbulkDT, i, k) * cff1 + CX(Dden1DT, i, k) * cff2);
      CX(Scof, i, k) = (CX(DbulkDS, i, k) * cff1 + CX(Dden1DS, i, k) * cff2);
      if (o->ddmix) o->alfaobeta[XW(i, j, k)] = CX(Tcof, i, k) / CX(Scof, i, k);
    }
    for (int i = b->IstrT; i <= b->IendT; i++) {
      cff = 1.0 / CX(wrk, i, N);
      o->alpha[X2(i, j)] = cff * CX(Tcof, i, N);
      o->beta[X2(i, j)] = cff * CX(Scof, i, N);
    }
    for (int k = 1; k <= N; k++)
      for (int i = b->IstrT; i <= b->IendT; i++) {
        o->rho[X3(i, j, k)] = CX(den, i, k);
        o->pden[X3(i, j, k)] = (CX(den1, i, k) - 1000.0);
        if (msk) o->pden[X3(i, j, k)] = o->pden[X3(i, j, k)] * o->rmask[X2(i, j)];        /* :479 */
      }
  }
  free(W);
  orc_exchange3d(o, b, 'r', o->rho, N);
  orc_exchange3d(o, b, 'r', o->pden, N);
  orc_exchange2d(o, b, 'r', o->alpha);
  orc_exchange2d(o, b, 'r', o->beta);
  orc_exchange2d(o, b, 'r', rhoA);
  orc_exchange2d(o, b, 'r', rhoS);
  orc_exchange3d(o, b, 'w', bvf, N + 1);
  if (o->ddmix) orc_exchange3d(o, b, 'w', o->alfaobeta, N + 1);                            /* :499-502 */
}
