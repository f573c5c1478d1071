/*
 * orc_gls.c -- generic length-scale vertical turbulence closure (Umlauf and Burchard 2003; Warner et al. 2005).
 * TEST INFRASTRUCTURE (see orc.h).
 *
 *   orc_gls_prestep   gls_prestep_tile   ROMS/Nonlinear/gls_prestep.F:95-446   predictor of tke, gls at n+1/2
 *   orc_gls_corstep   gls_corstep_tile   ROMS/Nonlinear/gls_corstep.F:114-1257 corrector, stability functions,
 *                                                                              Akv, Akt, Akk, Akp, Lscale
 *   tkebc             tkebc_tile         ROMS/Nonlinear/tkebc_im.F:46-700      closed / gradient edges (the
 *                                                                              radiation condition is not restated)
 *   orc_gls_consts    initialize_scalars ROMS/Modules/mod_scalars.F:4715-4766  Canuto A/B, Kantha-Clayson, Galperin
 * Compile-time forms of the reference selected at run time by cfg.gls_flags: CANUTO_A | CANUTO_B | KANTHA_CLAYSON |
 * (none: Galperin); N2S2_HORAVG; RI_SPLINES; K_C2ADVECTION | K_C4ADVECTION | (none: third-order upstream);
 * CHARNOK; CRAIG_BANNER.  ZOS_HSIG and TKE_WAVEDISS need wave fields and are not restated.
 * PARITY: pinned (gls_prestep.F, gls_corstep.F, tkebc_im.F build in oracle/_ref: libromsref_upwelling_gls*.so).
 */
#include "orc.h"
#include <math.h>
#include <stdlib.h>
#define MAX(a, b) ((a) > (b) ? (a) : (b))
#define MIN(a, b) ((a) < (b) ? (a) : (b))
#define CX(A, i, k) A[(size_t)((i) - LBi) + (size_t)(k) * ni]

typedef struct {
  double Gh0, Ghcri, Ghmin, E2;
  double s0, s1, s2, s4, s5, s6, b0, b1, b2, b3, b4, b5;          /* Canuto */
  double my_B1pm1o3, my_Sh1, my_Sh2, my_Sm2, my_Sm3, my_Sm4;      /* Kantha-Clayson / Galperin */
} gls_consts;

/* mod_scalars.F:1764-1796 (parameters) and :4715-4766 (derived) */
static void orc_gls_consts(int flags, gls_consts *q) {
  double L1 = 0, L2 = 0, L3 = 0, L4 = 0, L5 = 0, L6 = 0, L7 = 0, L8 = 0;
  const gls_consts zero = {0};
  *q = zero;
  if (flags & ORC_GLS_CANUTO_A) {
    q->Gh0 = 0.0329; q->Ghcri = 0.03;
    L1 = 0.107; L2 = 0.0032; L3 = 0.0864; L4 = 0.12; L5 = 11.9; L6 = 0.4; L7 = 0.0; L8 = 0.48;
  } else if (flags & ORC_GLS_CANUTO_B) {
    q->Gh0 = 0.0444; q->Ghcri = 0.0414;
    L1 = 0.127; L2 = 0.00336; L3 = 0.0906; L4 = 0.101; L5 = 11.2; L6 = 0.4; L7 = 0.0; L8 = 0.318;
  } else {
    q->Gh0 = 0.028; q->Ghcri = 0.02;
  }
  q->Ghmin = -0.28;
  q->E2 = 1.33;
  if (flags & (ORC_GLS_CANUTO_A | ORC_GLS_CANUTO_B)) {
    q->s0 = 3.0 / 2.0 * L1 * (L5 * L5);
    q->s1 = -L4 * (L6 + L7) + 2.0 * L4 * L5 * (L1 - 1.0 / 3.0 * L2 - L3) + 3.0 / 2.0 * L1 * L5 * L8;
    q->s2 = -3.0 / 8.0 * L1 * (L6 * L6 - L7 * L7);
    q->s4 = 2.0 * L5;
    q->s5 = 2.0 * L4;
    q->s6 = 2.0 / 3.0 * L5 * (3.0 * (L3 * L3) - L2 * L2) - 1.0 / 2.0 * L5 * L1 * (3.0 * L3 - L2) + 3.0 / 4.0 * L1 * (L6 - L7);
    q->b0 = 3.0 * (L5 * L5);
    q->b1 = L5 * (7.0 * L4 + 3.0 * L8);
    q->b2 = L5 * L5 * (3.0 * (L3 * L3) - L2 * L2) - 3.0 / 4.0 * (L6 * L6 - L7 * L7);
    q->b3 = L4 * (4.0 * L4 + 3.0 * L8);
    q->b5 = 1.0 / 4.0 * (L2 * L2 - 3.0 * (L3 * L3)) * (L6 * L6 - L7 * L7);
    q->b4 = L4 * (L2 * L6 - 3.0 * L3 * L7 - L5 * (L2 * L2 - L3 * L3)) + L5 * L8 * (3.0 * (L3 * L3) - L2 * L2);
  }
  const double A1 = 0.92, A2 = 0.74, B1 = 16.6, B2 = 10.1, C1 = 0.08, C2 = 0.7, C3 = 0.2;
  q->my_B1pm1o3 = 1.0 / pow(B1, 1.0 / 3.0);
  q->my_Sm2 = 9.0 * A1 * A2;
  q->my_Sh1 = A2 * (1.0 - 6.0 * A1 / B1);
  if (flags & ORC_GLS_KANTHA_CLAYSON) {
    q->my_Sh2 = 3.0 * A2 * (6.0 * A1 + B2 * (1.0 - C3));
    q->my_Sm3 = 0.0;
    q->my_Sm4 = 18.0 * A1 * A1 + 9.0 * A1 * A2 * (1.0 - C2);
  } else {
    q->my_Sh2 = 3.0 * A2 * (6.0 * A1 + B2);
    q->my_Sm3 = A1 * (1.0 - 3.0 * C1 - 6.0 * A1 / B1);
    q->my_Sm4 = 18.0 * A1 * A1 + 9.0 * A1 * A2;
  }
}

/* tkebc_im.F: oracle/orc_obc.c (radiation, gradient and closed edges) */
static void tkebc(const orc_t *o, const orc_bounds *b, int nout) { orc_tkebc(o, b, nout); }

/* gls_prestep_tile, gls_prestep.F:95 */
void orc_gls_prestep(orc_t *o, int tile) {
  ORC_LOCALS(o);
  const orc_bounds *b = &o->b[tile];
  const orc_cfg *c = &o->c;
  const int Istr = b->Istr, Iend = b->Iend, Jstr = b->Jstr, Jend = b->Jend;
  const int nstp = o->s.nstp, nnew = o->s.nnew;
  const int c2 = (c->gls_flags & ORC_GLS_K_C2ADVECTION) != 0;
  const double Gamma = 1.0 / 6.0, dt = c->dt;
  double *tke = o->tke, *gls = o->gls;
  const double *Huon = o->Huon, *Hvom = o->Hvom, *Hz = o->Hz, *W = o->W, *pm = o->pm, *pn = o->pn,
               *umask = o->umask, *vmask = o->vmask;
  double *Hz_half = (double *)calloc(nij * (N + 1), sizeof(double));
  double *p2 = (double *)calloc(8 * nij, sizeof(double));
  double *XF = p2, *FX = p2 + nij, *FXL = p2 + 2 * nij, *EF = p2 + 3 * nij, *FE = p2 + 4 * nij, *FEL = p2 + 5 * nij,
         *grad = p2 + 6 * nij, *gradL = p2 + 7 * nij;
  double *CF = (double *)calloc(3 * ni * (N + 1), sizeof(double)), *FC = CF + ni * (N + 1), *FCL = CF + 2 * ni * (N + 1);
#define TK(i, j, k, n) tke[XW4(i, j, k, n)]
#define GL(i, j, k, n) gls[XW4(i, j, k, n)]
  for (int k = 1; k <= N - 1; k++) {
    if (c2) {                                                             /* :197-216 */
      for (int j = Jstr; j <= Jend; j++)
        for (int i = Istr; i <= Iend + 1; i++) {
          XF[X2(i, j)] = 0.5 * (Huon[X3(i, j, k)] + Huon[X3(i, j, k + 1)]);
          FX[X2(i, j)] = XF[X2(i, j)] * 0.5 * (TK(i, j, k, nstp) + TK(i - 1, j, k, nstp));
          FXL[X2(i, j)] = XF[X2(i, j)] * 0.5 * (GL(i, j, k, nstp) + GL(i - 1, j, k, nstp));
        }
      for (int j = Jstr; j <= Jend + 1; j++)
        for (int i = Istr; i <= Iend; i++) {
          EF[X2(i, j)] = 0.5 * (Hvom[X3(i, j, k)] + Hvom[X3(i, j, k + 1)]);
          FE[X2(i, j)] = EF[X2(i, j)] * 0.5 * (TK(i, j, k, nstp) + TK(i, j - 1, k, nstp));
          FEL[X2(i, j)] = EF[X2(i, j)] * 0.5 * (GL(i, j, k, nstp) + GL(i, j - 1, k, nstp));
        }
    } else {                                                              /* :218-302 */
      for (int j = Jstr; j <= Jend; j++)
        for (int i = b->Istrm1; i <= b->Iendp2; i++) {
          grad[X2(i, j)] = (TK(i, j, k, nstp) - TK(i - 1, j, k, nstp)) * umask[X2(i, j)];
          gradL[X2(i, j)] = (GL(i, j, k, nstp) - GL(i - 1, j, k, nstp)) * umask[X2(i, j)];
        }
      if (!c->EWperiodic) {
        if (b->west) for (int j = Jstr; j <= Jend; j++) { grad[X2(Istr - 1, j)] = grad[X2(Istr, j)]; gradL[X2(Istr - 1, j)] = gradL[X2(Istr, j)]; }
        if (b->east) for (int j = Jstr; j <= Jend; j++) { grad[X2(Iend + 2, j)] = grad[X2(Iend + 1, j)]; gradL[X2(Iend + 2, j)] = gradL[X2(Iend + 1, j)]; }
      }
      double cff = 1.0 / 6.0;
      for (int j = Jstr; j <= Jend; j++)
        for (int i = Istr; i <= Iend + 1; i++) {
          XF[X2(i, j)] = 0.5 * (Huon[X3(i, j, k)] + Huon[X3(i, j, k + 1)]);
          FX[X2(i, j)] = XF[X2(i, j)] * 0.5 * (TK(i - 1, j, k, nstp) + TK(i, j, k, nstp) - cff * (grad[X2(i + 1, j)] - grad[X2(i - 1, j)]));
          FXL[X2(i, j)] = XF[X2(i, j)] * 0.5 * (GL(i - 1, j, k, nstp) + GL(i, j, k, nstp) - cff * (gradL[X2(i + 1, j)] - gradL[X2(i - 1, j)]));
        }
      for (int j = b->Jstrm1; j <= b->Jendp2; j++)
        for (int i = Istr; i <= Iend; i++) {
          grad[X2(i, j)] = (TK(i, j, k, nstp) - TK(i, j - 1, k, nstp)) * vmask[X2(i, j)];
          gradL[X2(i, j)] = (GL(i, j, k, nstp) - GL(i, j - 1, k, nstp)) * vmask[X2(i, j)];
        }
      if (!c->NSperiodic) {
        if (b->south) for (int i = Istr; i <= Iend; i++) { grad[X2(i, Jstr - 1)] = grad[X2(i, Jstr)]; gradL[X2(i, Jstr - 1)] = gradL[X2(i, Jstr)]; }
        if (b->north) for (int i = Istr; i <= Iend; i++) { grad[X2(i, Jend + 2)] = grad[X2(i, Jend + 1)]; gradL[X2(i, Jend + 2)] = gradL[X2(i, Jend + 1)]; }
      }
      for (int j = Jstr; j <= Jend + 1; j++)
        for (int i = Istr; i <= Iend; i++) {
          EF[X2(i, j)] = 0.5 * (Hvom[X3(i, j, k)] + Hvom[X3(i, j, k + 1)]);
          FE[X2(i, j)] = EF[X2(i, j)] * 0.5 * (TK(i, j - 1, k, nstp) + TK(i, j, k, nstp) - cff * (grad[X2(i, j + 1)] - grad[X2(i, j - 1)]));
          FEL[X2(i, j)] = EF[X2(i, j)] * 0.5 * (GL(i, j - 1, k, nstp) + GL(i, j, k, nstp) - cff * (gradL[X2(i, j + 1)] - gradL[X2(i, j - 1)]));
        }
    }
    double cff1, cff2, cff3;                                              /* :306-335 */
    int indx;
    if (o->s.iic == c->ntfirst) { cff1 = 1.0; cff2 = 0.0; cff3 = 0.5 * dt; indx = nstp; }
    else { cff1 = 0.5 + Gamma; cff2 = 0.5 - Gamma; cff3 = (1.0 - Gamma) * dt; indx = 3 - nstp; }
    for (int j = Jstr; j <= Jend; j++)
      for (int i = Istr; i <= Iend; i++) {
        const double cff = 0.5 * (Hz[X3(i, j, k)] + Hz[X3(i, j, k + 1)]);
        const double cff4 = cff3 * pm[X2(i, j)] * pn[X2(i, j)];
        Hz_half[XW(i, j, k)] = cff - cff4 * (XF[X2(i + 1, j)] - XF[X2(i, j)] + EF[X2(i, j + 1)] - EF[X2(i, j)]);
        TK(i, j, k, 3) = cff * (cff1 * TK(i, j, k, nstp) + cff2 * TK(i, j, k, indx)) -
                         cff4 * (FX[X2(i + 1, j)] - FX[X2(i, j)] + FE[X2(i, j + 1)] - FE[X2(i, j)]);
        GL(i, j, k, 3) = cff * (cff1 * GL(i, j, k, nstp) + cff2 * GL(i, j, k, indx)) -
                         cff4 * (FXL[X2(i + 1, j)] - FXL[X2(i, j)] + FEL[X2(i, j + 1)] - FEL[X2(i, j)]);
        TK(i, j, k, nnew) = cff * TK(i, j, k, nstp);
        GL(i, j, k, nnew) = cff * GL(i, j, k, nstp);
      }
  }
  for (int j = Jstr; j <= Jend; j++) {                                    /* :339-417 */
    if (c2) {
      for (int k = 1; k <= N; k++)
        for (int i = Istr; i <= Iend; i++) {
          CX(CF, i, k) = 0.5 * (W[XW(i, j, k)] + W[XW(i, j, k - 1)]);
          CX(FC, i, k) = CX(CF, i, k) * 0.5 * (TK(i, j, k - 1, nstp) + TK(i, j, k, nstp));
          CX(FCL, i, k) = CX(CF, i, k) * 0.5 * (GL(i, j, k - 1, nstp) + GL(i, j, k, nstp));
        }
    } else {
      double cff1 = 7.0 / 12.0, cff2 = 1.0 / 12.0;
      for (int k = 2; k <= N - 1; k++)
        for (int i = Istr; i <= Iend; i++) {
          CX(CF, i, k) = 0.5 * (W[XW(i, j, k)] + W[XW(i, j, k - 1)]);
          CX(FC, i, k) = CX(CF, i, k) * (cff1 * (TK(i, j, k - 1, nstp) + TK(i, j, k, nstp)) - cff2 * (TK(i, j, k - 2, nstp) + TK(i, j, k + 1, nstp)));
          CX(FCL, i, k) = CX(CF, i, k) * (cff1 * (GL(i, j, k - 1, nstp) + GL(i, j, k, nstp)) - cff2 * (GL(i, j, k - 2, nstp) + GL(i, j, k + 1, nstp)));
        }
      cff1 = 1.0 / 3.0; cff2 = 5.0 / 6.0;
      const double cff3 = 1.0 / 6.0;
      for (int i = Istr; i <= Iend; i++) {
        CX(CF, i, 1) = 0.5 * (W[XW(i, j, 0)] + W[XW(i, j, 1)]);
        CX(FC, i, 1) = CX(CF, i, 1) * (cff1 * TK(i, j, 0, nstp) + cff2 * TK(i, j, 1, nstp) - cff3 * TK(i, j, 2, nstp));
        CX(FCL, i, 1) = CX(CF, i, 1) * (cff1 * GL(i, j, 0, nstp) + cff2 * GL(i, j, 1, nstp) - cff3 * GL(i, j, 2, nstp));
        CX(CF, i, N) = 0.5 * (W[XW(i, j, N)] + W[XW(i, j, N - 1)]);
        CX(FC, i, N) = CX(CF, i, N) * (cff1 * TK(i, j, N, nstp) + cff2 * TK(i, j, N - 1, nstp) - cff3 * TK(i, j, N - 2, nstp));
        CX(FCL, i, N) = CX(CF, i, N) * (cff1 * GL(i, j, N, nstp) + cff2 * GL(i, j, N - 1, nstp) - cff3 * GL(i, j, N - 2, nstp));
      }
    }
    const double cff3 = (o->s.iic == c->ntfirst) ? 0.5 * dt : (1.0 - Gamma) * dt;      /* :421-437 */
    for (int k = 1; k <= N - 1; k++)
      for (int i = Istr; i <= Iend; i++) {
        const double cff4 = cff3 * pm[X2(i, j)] * pn[X2(i, j)];
        Hz_half[XW(i, j, k)] = Hz_half[XW(i, j, k)] - cff4 * (CX(CF, i, k + 1) - CX(CF, i, k));
        const double cff1 = 1.0 / Hz_half[XW(i, j, k)];
        TK(i, j, k, 3) = cff1 * (TK(i, j, k, 3) - cff4 * (CX(FC, i, k + 1) - CX(FC, i, k)));
        GL(i, j, k, 3) = cff1 * (GL(i, j, k, 3) - cff4 * (CX(FCL, i, k + 1) - CX(FCL, i, k)));
      }
  }
  tkebc(o, b, 3);                                                          /* :441-456 */
  if (c->EWperiodic || c->NSperiodic) {
    orc_exchange3d(o, b, 'r', tke + 2 * nij * (N + 1), N + 1);
    orc_exchange3d(o, b, 'r', gls + 2 * nij * (N + 1), N + 1);
  }
  free(Hz_half); free(p2); free(CF);
}

#define TK(i, j, k, n) tke[XW4(i, j, k, n)]
#define GL(i, j, k, n) gls[XW4(i, j, k, n)]
#define SH(i, j, k) shear2[XW(i, j, k)]
#define BU(i, j, k) buoy2[XW(i, j, k)]
/* my25_corstep.F:585-750 for one row j: vertical mixing of the turbulent fields, production, dissipation, the two
   tridiagonal systems (boundary values inside them), length scale, stability functions, Akv, Akt, Akk, Lscale */
static void my25_column(orc_t *o, const orc_bounds *b, int j, const double *shear2, const double *buoy2, double *BCK, double *BCP,
                        double *CF, double *FCK) {
  ORC_LOCALS(o);
  const orc_cfg *c = &o->c;
  const int Istr = b->Istr, Iend = b->Iend, NAT = c->NAT, nstp = o->s.nstp, nnew = o->s.nnew;
  const int kc = (c->gls_flags & ORC_GLS_KANTHA_CLAYSON) != 0;
  const double eps = 1.0E-10, vonKar = 0.41, dt = c->dt;
  const double my_B1 = 16.6, my_E1 = 1.8, my_E2 = 1.33, my_Gh0 = 0.0233, my_Sq = 0.2, my_lmax = 0.53, my_qmin = 1.0E-8;
  const double my_B1p2o3 = pow(my_B1, 2.0 / 3.0);
  gls_consts q;
  orc_gls_consts(c->gls_flags, &q);
  double *tke = o->tke, *gls = o->gls, *Akv = o->Akv, *Akt = o->Akt, *Akk = o->Akk, *Lscale = o->Lscale;
  const double *Hz = o->Hz, *z_w = o->z_w, *sustr = o->sustr, *svstr = o->svstr, *bustr = o->bustr, *bvstr = o->bvstr;
  {
    const double cff = -0.5 * dt;
    for (int k = 1; k <= N; k++)
      for (int i = Istr; i <= Iend; i++) {
        CX(FCK, i, k) = cff * (Akk[XW(i, j, k)] + Akk[XW(i, j, k - 1)]) / Hz[X3(i, j, k)];
        CX(CF, i, k) = 0.0;
      }
  }
  const double cff3 = my_E2 / (vonKar * vonKar);
  for (int k = 1; k <= N - 1; k++)
    for (int i = Istr; i <= Iend; i++) {
      const double bu = BU(i, j, k);
      const double strat2 = (bu > -5.0E-5 && bu < 0.0) ? 0.0 : bu;
      const double Qprod = SH(i, j, k) * (Akv[XW(i, j, k)] - c->Akv_bak) - strat2 * (Akt[XW4(i, j, k, 1)] - c->Akt_bak[0]);
      const double Ls_unlmt = MAX(eps, GL(i, j, k, nstp) / (MAX(TK(i, j, k, nstp), eps)));
      const double cff1 = 0.5 * (Hz[X3(i, j, k)] + Hz[X3(i, j, k + 1)]);
      TK(i, j, k, nnew) = TK(i, j, k, nnew) + dt * cff1 * Qprod * 2.0;
      GL(i, j, k, nnew) = GL(i, j, k, nnew) + dt * cff1 * Qprod * my_E1 * Ls_unlmt;
      const double Qdiss = dt * sqrt(TK(i, j, k, nstp)) / (my_B1 * Ls_unlmt);
      const double cff = Ls_unlmt * (1.0 / (z_w[XW(i, j, N)] - z_w[XW(i, j, k)]) + 1.0 / (z_w[XW(i, j, k)] - z_w[XW(i, j, 0)]));
      const double Wscale = 1.0 + cff3 * cff * cff;
      CX(BCK, i, k) = cff1 * (1.0 + 2.0 * Qdiss) - CX(FCK, i, k) - CX(FCK, i, k + 1);
      CX(BCP, i, k) = cff1 * (1.0 + Wscale * Qdiss) - CX(FCK, i, k) - CX(FCK, i, k + 1);
    }
  for (int i = Istr; i <= Iend; i++) {
    TK(i, j, N, nnew) = my_B1p2o3 * 0.5 * sqrt((sustr[X2(i, j)] + sustr[X2(i + 1, j)]) * (sustr[X2(i, j)] + sustr[X2(i + 1, j)]) +
                                               (svstr[X2(i, j)] + svstr[X2(i, j + 1)]) * (svstr[X2(i, j)] + svstr[X2(i, j + 1)]));
    GL(i, j, N, nnew) = 0.0;
    TK(i, j, 0, nnew) = my_B1p2o3 * 0.5 * sqrt((bustr[X2(i, j)] + bustr[X2(i + 1, j)]) * (bustr[X2(i, j)] + bustr[X2(i + 1, j)]) +
                                               (bvstr[X2(i, j)] + bvstr[X2(i, j + 1)]) * (bvstr[X2(i, j)] + bvstr[X2(i, j + 1)]));
    GL(i, j, 0, nnew) = 0.0;
  }
  for (int f = 0; f < 2; f++) {          /* the tke system with BCK, the gls system with BCP; both with FCK */
    double *A = (f == 0 ? tke : gls) + (size_t)(nnew - 1) * nij * (N + 1);
    const double *BC = f == 0 ? BCK : BCP;
    for (int i = Istr; i <= Iend; i++) {
      const double cff = 1.0 / CX(BC, i, N - 1);
      CX(CF, i, N - 1) = cff * CX(FCK, i, N - 1);
      A[XW(i, j, N - 1)] = cff * (A[XW(i, j, N - 1)] - CX(FCK, i, N) * A[XW(i, j, N)]);
    }
    for (int k = N - 2; k >= 1; k--)
      for (int i = Istr; i <= Iend; i++) {
        const double cff = 1.0 / (CX(BC, i, k) - CX(CF, i, k + 1) * CX(FCK, i, k + 1));
        CX(CF, i, k) = cff * CX(FCK, i, k);
        A[XW(i, j, k)] = cff * (A[XW(i, j, k)] - CX(FCK, i, k + 1) * A[XW(i, j, k + 1)]);
      }
    for (int k = 1; k <= N - 1; k++)
      for (int i = Istr; i <= Iend; i++) A[XW(i, j, k)] = A[XW(i, j, k)] - CX(CF, i, k) * A[XW(i, j, k - 1)];
  }
  for (int k = 1; k <= N - 1; k++)
    for (int i = Istr; i <= Iend; i++) {
      TK(i, j, k, nnew) = MAX(TK(i, j, k, nnew), my_qmin);
      GL(i, j, k, nnew) = MAX(GL(i, j, k, nnew), my_qmin);
      const double Ls_unlmt = GL(i, j, k, nnew) / TK(i, j, k, nnew);
      const double Ls_lmt = MIN(Ls_unlmt, my_lmax * sqrt(TK(i, j, k, nnew) / (MAX(0.0, BU(i, j, k)) + eps)));
      const double Gh = MIN(my_Gh0, -BU(i, j, k) * Ls_lmt * Ls_lmt / TK(i, j, k, nnew));
      const double cff = 1.0 - q.my_Sh2 * Gh;
      const double Sh = q.my_Sh1 / cff;
      const double Sm = kc ? (q.my_B1pm1o3 + Sh * Gh * q.my_Sm4) / (1.0 - q.my_Sm2 * Gh) : (q.my_Sm3 + Sh * Gh * q.my_Sm4) / (1.0 - q.my_Sm2 * Gh);
      const double ql = 0.5 * (Ls_lmt * sqrt(TK(i, j, k, nnew)) + Lscale[XW(i, j, k)] * sqrt(TK(i, j, k, nstp)));
      Akv[XW(i, j, k)] = c->Akv_bak + ql * Sm;
      for (int it = 1; it <= NAT; it++) Akt[XW4(i, j, k, it)] = c->Akt_bak[it - 1] + ql * Sh;
      Akk[XW(i, j, k)] = c->Akk_bak + ql * my_Sq;
      Lscale[XW(i, j, k)] = Ls_lmt;
    }
}

/* gls_corstep_tile, gls_corstep.F:114; my25: my25_corstep_tile, my25_corstep.F:114 (the same shear, smoothing and
   advection code; its own production / dissipation, boundary values, stability functions and edge copies) */
static void corstep(orc_t *o, int tile, int my25) {
  ORC_LOCALS(o);
  const orc_bounds *b = &o->b[tile];
  const orc_cfg *c = &o->c;
  const int Istr = b->Istr, Iend = b->Iend, Jstr = b->Jstr, Jend = b->Jend;
  const int Lm = c->Lm, Mm = c->Mm, NAT = c->NAT;
  const int nstp = o->s.nstp, nnew = o->s.nnew, flags = c->gls_flags;
  const int c2 = (flags & ORC_GLS_K_C2ADVECTION) != 0, c4 = (flags & ORC_GLS_K_C4ADVECTION) != 0;
  const int canuto = (flags & (ORC_GLS_CANUTO_A | ORC_GLS_CANUTO_B)) != 0, kc = (flags & ORC_GLS_KANTHA_CLAYSON) != 0;
  const int crgban = (flags & ORC_GLS_CRAIG_BANNER) != 0;
  const double Gadv = 1.0 / 3.0, eps = 1.0E-10, vonKar = 0.41, dt = c->dt, g = c->g;
  const double gls_p = c->gls_p, gls_m = c->gls_m, gls_n = c->gls_n, gls_Kmin = c->gls_Kmin, gls_Pmin = c->gls_Pmin,
               gls_cmu0 = c->gls_cmu0, gls_c1 = c->gls_c1, gls_c2 = c->gls_c2, gls_c3m = c->gls_c3m, gls_c3p = c->gls_c3p,
               gls_sigk = c->gls_sigk, gls_sigp = c->gls_sigp, Akk_bak = c->Akk_bak, Akp_bak = c->Akp_bak,
               Akv_bak = c->Akv_bak, Akt_bak1 = c->Akt_bak[0];
  gls_consts q = {0};
  orc_gls_consts(flags, &q);
  double *tke = o->tke, *gls = o->gls, *Akv = o->Akv, *Akt = o->Akt, *Akk = o->Akk, *Akp = o->Akp, *Lscale = o->Lscale;
  const double *Huon = o->Huon, *Hvom = o->Hvom, *Hz = o->Hz, *W = o->W, *pm = o->pm, *pn = o->pn, *u = o->u, *v = o->v,
               *z_r = o->z_r, *z_w = o->z_w, *bvf = o->bvf, *umask = o->umask, *vmask = o->vmask,
               *sustr = o->sustr, *svstr = o->svstr, *bustr = o->bustr, *bvstr = o->bvstr;
  double *shear2 = (double *)calloc(2 * nij * (N + 1), sizeof(double)), *buoy2 = shear2 + nij * (N + 1);
  double *p2 = (double *)calloc(8 * nij, sizeof(double));
  double *FXK = p2, *FXP = p2 + nij, *FEK = p2 + 2 * nij, *FEP = p2 + 3 * nij, *gradK = p2 + 4 * nij, *gradP = p2 + 5 * nij,
         *curvK = p2 + 6 * nij, *curvP = p2 + 7 * nij;
  double *cw = (double *)calloc(7 * ni * (N + 1), sizeof(double));
  double *BCK = cw, *BCP = cw + ni * (N + 1), *CF = cw + 2 * ni * (N + 1), *FCK = cw + 3 * ni * (N + 1),
         *FCP = cw + 4 * ni * (N + 1), *dU = cw + 5 * ni * (N + 1), *dV = cw + 6 * ni * (N + 1);
  /* constants :250-340 */
  const double Zos_min = MAX(c->Zos, 0.0001);
  const double Zob_min = MAX(c->Zob, 0.0001);                   /* ZoBot = Zob (mod_grid.F:1380) */
  const int Lmy25 = (gls_p == 0.0) && (gls_n == 1.0) && (gls_m == 1.0);
  double L_sft, gls_sigp_cb;
  if (crgban) {
    const double cb_wallE = Lmy25 ? 1.25 : 1.0;
    L_sft = vonKar;
    const double cff1 = sqrt(1.5 * gls_sigk) * gls_cmu0 / L_sft;
    gls_sigp_cb = (L_sft * L_sft) / ((gls_cmu0 * gls_cmu0) * gls_c2 * cb_wallE) *
                  ((gls_n * gls_n) - cff1 * gls_n / 3.0 * (4.0 * gls_m + 1.0) + (cff1 * cff1) * gls_m / 9.0 * (2.0 + 4.0 * gls_m));
  } else {
    L_sft = vonKar;
    gls_sigp_cb = gls_sigp;
  }
  const double ogls_sigp = 1.0 / gls_sigp_cb;
  const double sqrt2 = sqrt(2.0);
  const double cmu_fac1 = pow(gls_cmu0, -gls_p / gls_n);
  const double cmu_fac2 = pow(gls_cmu0, 3.0 + gls_p / gls_n);
  const double cmu_fac3 = 1.0 / pow(gls_cmu0, 2.0);
  const double cmu_fac4 = pow(1.5 * gls_sigk, 1.0 / 3.0) / pow(gls_cmu0, 4.0 / 3.0);
  const double gls_fac2 = pow(gls_cmu0, gls_p) * gls_n * pow(vonKar, gls_n);
  const double gls_fac3 = pow(gls_cmu0, gls_p) * gls_n;
  const double gls_fac4 = pow(gls_cmu0, gls_p);
  const double gls_fac5 = pow(0.56, 0.5 * gls_n) * pow(gls_cmu0, gls_p);
  const double gls_fac6 = 8.0 / pow(gls_cmu0, 6.0);
  const double gls_exp1 = 1.0 / gls_n, tke_exp1 = gls_m / gls_n, tke_exp2 = 0.5 + gls_m / gls_n, tke_exp4 = gls_m + 0.5 * gls_n;
  /* vertical shear at W-points :345-400 */
  if (flags & ORC_GLS_RI_SPLINES) {
    for (int j = b->Jstrm1; j <= b->Jendp1; j++) {
      for (int i = b->Istrm1; i <= b->Iendp1; i++) { CX(CF, i, 0) = 0.0; CX(dU, i, 0) = 0.0; CX(dV, i, 0) = 0.0; }
      for (int k = 1; k <= N - 1; k++)
        for (int i = b->Istrm1; i <= b->Iendp1; i++) {
          const double cff = 1.0 / (2.0 * Hz[X3(i, j, k + 1)] + Hz[X3(i, j, k)] * (2.0 - CX(CF, i, k - 1)));
          CX(CF, i, k) = cff * Hz[X3(i, j, k + 1)];
          CX(dU, i, k) = cff * (3.0 * (u[X4(i, j, k + 1, nstp)] - u[X4(i, j, k, nstp)] + u[X4(i + 1, j, k + 1, nstp)] - u[X4(i + 1, j, k, nstp)]) -
                                Hz[X3(i, j, k)] * CX(dU, i, k - 1));
          CX(dV, i, k) = cff * (3.0 * (v[X4(i, j, k + 1, nstp)] - v[X4(i, j, k, nstp)] + v[X4(i, j + 1, k + 1, nstp)] - v[X4(i, j + 1, k, nstp)]) -
                                Hz[X3(i, j, k)] * CX(dV, i, k - 1));
        }
      for (int i = b->Istrm1; i <= b->Iendp1; i++) { CX(dU, i, N) = 0.0; CX(dV, i, N) = 0.0; }
      for (int k = N - 1; k >= 1; k--)
        for (int i = b->Istrm1; i <= b->Iendp1; i++) {
          CX(dU, i, k) = CX(dU, i, k) - CX(CF, i, k) * CX(dU, i, k + 1);
          CX(dV, i, k) = CX(dV, i, k) - CX(CF, i, k) * CX(dV, i, k + 1);
        }
      for (int k = 1; k <= N - 1; k++)
        for (int i = b->Istrm1; i <= b->Iendp1; i++) SH(i, j, k) = CX(dU, i, k) * CX(dU, i, k) + CX(dV, i, k) * CX(dV, i, k);
    }
  } else {
    for (int k = 1; k <= N - 1; k++)
      for (int j = b->Jstrm1; j <= b->Jendp1; j++)
        for (int i = b->Istrm1; i <= b->Iendp1; i++) {
          const double cff = 0.5 / (z_r[X3(i, j, k + 1)] - z_r[X3(i, j, k)]);
          const double a = cff * (u[X4(i, j, k + 1, nstp)] - u[X4(i, j, k, nstp)] + u[X4(i + 1, j, k + 1, nstp)] - u[X4(i + 1, j, k, nstp)]);
          const double e = cff * (v[X4(i, j, k + 1, nstp)] - v[X4(i, j, k, nstp)] + v[X4(i, j + 1, k + 1, nstp)] - v[X4(i, j + 1, k, nstp)]);
          SH(i, j, k) = a * a + e * e;
        }
  }
  for (int k = 1; k <= N - 1; k++)                                        /* :404-410 */
    for (int j = Jstr - 1; j <= Jend + 1; j++)
      for (int i = Istr - 1; i <= Iend + 1; i++) BU(i, j, k) = bvf[XW(i, j, k)];
  if (flags & ORC_GLS_N2S2_HORAVG) {                                      /* :418-475 */
    for (int k = 1; k <= N - 1; k++) {
      if (b->west) for (int j = MAX(1, Jstr - 1); j <= MIN(Jend + 1, Mm); j++) SH(Istr - 1, j, k) = SH(Istr, j, k);
      if (b->east) for (int j = MAX(1, Jstr - 1); j <= MIN(Jend + 1, Mm); j++) SH(Iend + 1, j, k) = SH(Iend, j, k);
      if (b->south) for (int i = MAX(1, Istr - 1); i <= MIN(Iend + 1, Lm); i++) SH(i, Jstr - 1, k) = SH(i, Jstr, k);
      if (b->north) for (int i = MAX(1, Istr - 1); i <= MIN(Iend + 1, Lm); i++) SH(i, Jend + 1, k) = SH(i, Jend, k);
      if (b->sw) SH(Istr - 1, Jstr - 1, k) = SH(Istr, Jstr, k);
      if (b->nw) SH(Istr - 1, Jend + 1, k) = SH(Istr, Jend, k);
      if (b->se) SH(Iend + 1, Jstr - 1, k) = SH(Iend, Jstr, k);
      if (b->ne) SH(Iend + 1, Jend + 1, k) = SH(Iend, Jend, k);
      for (int j = Jstr - 1; j <= Jend; j++)
        for (int i = Istr - 1; i <= Iend; i++) {
          BU(i, j, 0) = 0.25 * (BU(i, j, k) + BU(i + 1, j, k) + BU(i, j + 1, k) + BU(i + 1, j + 1, k));
          SH(i, j, 0) = 0.25 * (SH(i, j, k) + SH(i + 1, j, k) + SH(i, j + 1, k) + SH(i + 1, j + 1, k));
        }
      for (int j = Jstr; j <= Jend; j++)
        for (int i = Istr; i <= Iend; i++) {
          BU(i, j, k) = 0.25 * (BU(i, j, 0) + BU(i - 1, j, 0) + BU(i, j - 1, 0) + BU(i - 1, j - 1, 0));
          SH(i, j, k) = 0.25 * (SH(i, j, 0) + SH(i - 1, j, 0) + SH(i, j - 1, 0) + SH(i - 1, j - 1, 0));
        }
    }
  }
  /* horizontal advection :490-680 */
  for (int k = 1; k <= N - 1; k++) {
    if (c2) {
      for (int j = Jstr; j <= Jend; j++)
        for (int i = Istr; i <= Iend + 1; i++) {
          const double cff = 0.25 * (Huon[X3(i, j, k)] + Huon[X3(i, j, k + 1)]);
          FXK[X2(i, j)] = cff * (TK(i, j, k, 3) + TK(i - 1, j, k, 3));
          FXP[X2(i, j)] = cff * (GL(i, j, k, 3) + GL(i - 1, j, k, 3));
        }
      for (int j = Jstr; j <= Jend + 1; j++)
        for (int i = Istr; i <= Iend; i++) {
          const double cff = 0.25 * (Hvom[X3(i, j, k)] + Hvom[X3(i, j, k + 1)]);
          FEK[X2(i, j)] = cff * (TK(i, j, k, 3) + TK(i, j - 1, k, 3));
          FEP[X2(i, j)] = cff * (GL(i, j, k, 3) + GL(i, j - 1, k, 3));
        }
    } else {
      for (int j = Jstr; j <= Jend; j++)
        for (int i = b->Istrm1; i <= b->Iendp2; i++) {
          gradK[X2(i, j)] = (TK(i, j, k, 3) - TK(i - 1, j, k, 3)) * umask[X2(i, j)];
          gradP[X2(i, j)] = (GL(i, j, k, 3) - GL(i - 1, j, k, 3)) * umask[X2(i, j)];
        }
      if (!c->EWperiodic) {
        if (b->west) for (int j = Jstr; j <= Jend; j++) { gradK[X2(Istr - 1, j)] = gradK[X2(Istr, j)]; gradP[X2(Istr - 1, j)] = gradP[X2(Istr, j)]; }
        if (b->east) for (int j = Jstr; j <= Jend; j++) { gradK[X2(Iend + 2, j)] = gradK[X2(Iend + 1, j)]; gradP[X2(Iend + 2, j)] = gradP[X2(Iend + 1, j)]; }
      }
      if (c4) {
        const double cff1 = 1.0 / 6.0;
        for (int j = Jstr; j <= Jend; j++)
          for (int i = Istr; i <= Iend + 1; i++) {
            const double cff = 0.5 * (Huon[X3(i, j, k)] + Huon[X3(i, j, k + 1)]);
            FXK[X2(i, j)] = cff * 0.5 * (TK(i - 1, j, k, 3) + TK(i, j, k, 3) - cff1 * (gradK[X2(i + 1, j)] - gradK[X2(i - 1, j)]));
            FXP[X2(i, j)] = cff * 0.5 * (GL(i - 1, j, k, 3) + GL(i, j, k, 3) - cff1 * (gradP[X2(i + 1, j)] - gradP[X2(i - 1, j)]));
          }
      } else {
        for (int j = Jstr; j <= Jend; j++)
          for (int i = Istr - 1; i <= Iend + 1; i++) {
            curvK[X2(i, j)] = gradK[X2(i + 1, j)] - gradK[X2(i, j)];
            curvP[X2(i, j)] = gradP[X2(i + 1, j)] - gradP[X2(i, j)];
          }
        for (int j = Jstr; j <= Jend; j++)
          for (int i = Istr; i <= Iend + 1; i++) {
            const double cff = 0.5 * (Huon[X3(i, j, k)] + Huon[X3(i, j, k + 1)]);
            const double cff1 = cff > 0.0 ? curvK[X2(i - 1, j)] : curvK[X2(i, j)];
            const double cff2 = cff > 0.0 ? curvP[X2(i - 1, j)] : curvP[X2(i, j)];
            FXK[X2(i, j)] = cff * 0.5 * (TK(i - 1, j, k, 3) + TK(i, j, k, 3) - Gadv * cff1);
            FXP[X2(i, j)] = cff * 0.5 * (GL(i - 1, j, k, 3) + GL(i, j, k, 3) - Gadv * cff2);
          }
      }
      for (int j = b->Jstrm1; j <= b->Jendp2; j++)
        for (int i = Istr; i <= Iend; i++) {
          gradK[X2(i, j)] = (TK(i, j, k, 3) - TK(i, j - 1, k, 3)) * vmask[X2(i, j)];
          gradP[X2(i, j)] = (GL(i, j, k, 3) - GL(i, j - 1, k, 3)) * vmask[X2(i, j)];
        }
      if (!c->NSperiodic) {
        if (b->south) for (int i = Istr; i <= Iend; i++) { gradK[X2(i, Jstr - 1)] = gradK[X2(i, Jstr)]; gradP[X2(i, Jstr - 1)] = gradP[X2(i, Jstr)]; }
        if (b->north) for (int i = Istr; i <= Iend; i++) { gradK[X2(i, Jend + 2)] = gradK[X2(i, Jend + 1)]; gradP[X2(i, Jend + 2)] = gradP[X2(i, Jend + 1)]; }
      }
      if (c4) {
        const double cff1 = 1.0 / 6.0;
        for (int j = Jstr; j <= Jend + 1; j++)
          for (int i = Istr; i <= Iend; i++) {
            const double cff = 0.5 * (Hvom[X3(i, j, k)] + Hvom[X3(i, j, k + 1)]);
            FEK[X2(i, j)] = cff * 0.5 * (TK(i, j - 1, k, 3) + TK(i, j, k, 3) - cff1 * (gradK[X2(i, j + 1)] - gradK[X2(i, j - 1)]));
            FEP[X2(i, j)] = cff * 0.5 * (GL(i, j - 1, k, 3) + GL(i, j, k, 3) - cff1 * (gradP[X2(i, j + 1)] - gradP[X2(i, j - 1)]));
          }
      } else {
        for (int j = Jstr - 1; j <= Jend + 1; j++)
          for (int i = Istr; i <= Iend; i++) {
            curvK[X2(i, j)] = gradK[X2(i, j + 1)] - gradK[X2(i, j)];
            curvP[X2(i, j)] = gradP[X2(i, j + 1)] - gradP[X2(i, j)];
          }
        for (int j = Jstr; j <= Jend + 1; j++)
          for (int i = Istr; i <= Iend; i++) {
            const double cff = 0.5 * (Hvom[X3(i, j, k)] + Hvom[X3(i, j, k + 1)]);
            const double cff1 = cff > 0.0 ? curvK[X2(i, j - 1)] : curvK[X2(i, j)];
            const double cff2 = cff > 0.0 ? curvP[X2(i, j - 1)] : curvP[X2(i, j)];
            FEK[X2(i, j)] = cff * 0.5 * (TK(i, j - 1, k, 3) + TK(i, j, k, 3) - Gadv * cff1);
            FEP[X2(i, j)] = cff * 0.5 * (GL(i, j - 1, k, 3) + GL(i, j, k, 3) - Gadv * cff2);
          }
      }
    }
    for (int j = Jstr; j <= Jend; j++)                                    /* :664-678 */
      for (int i = Istr; i <= Iend; i++) {
        const double cff = dt * pm[X2(i, j)] * pn[X2(i, j)];
        TK(i, j, k, nnew) = TK(i, j, k, nnew) - cff * (FXK[X2(i + 1, j)] - FXK[X2(i, j)] + FEK[X2(i, j + 1)] - FEK[X2(i, j)]);
        if (!my25) TK(i, j, k, nnew) = MAX(TK(i, j, k, nnew), gls_Kmin);
        GL(i, j, k, nnew) = GL(i, j, k, nnew) - cff * (FXP[X2(i + 1, j)] - FXP[X2(i, j)] + FEP[X2(i, j + 1)] - FEP[X2(i, j)]);
        if (!my25) GL(i, j, k, nnew) = MAX(GL(i, j, k, nnew), gls_Pmin);
      }
  }
  double *Zos_eff = (double *)calloc(5 * ni, sizeof(double)), *tke_fluxt = Zos_eff + ni, *tke_fluxb = Zos_eff + 2 * ni,
         *gls_fluxt = Zos_eff + 3 * ni, *gls_fluxb = Zos_eff + 4 * ni;
#define V1(A, i) A[(i) - LBi]
  for (int j = Jstr; j <= Jend; j++) {
    /* vertical advection :684-760 */
    if (c2) {
      for (int k = 1; k <= N; k++)
        for (int i = Istr; i <= Iend; i++) {
          const double cff = 0.25 * (W[XW(i, j, k)] + W[XW(i, j, k - 1)]);
          CX(FCK, i, k) = cff * (TK(i, j, k, 3) + TK(i, j, k - 1, 3));
          CX(FCP, i, k) = cff * (GL(i, j, k, 3) + GL(i, j, k - 1, 3));
        }
    } else {
      double cff1 = 7.0 / 12.0, cff2 = 1.0 / 12.0;
      for (int k = 2; k <= N - 1; k++)
        for (int i = Istr; i <= Iend; i++) {
          const double cff = 0.5 * (W[XW(i, j, k)] + W[XW(i, j, k - 1)]);
          CX(FCK, i, k) = cff * (cff1 * (TK(i, j, k - 1, 3) + TK(i, j, k, 3)) - cff2 * (TK(i, j, k - 2, 3) + TK(i, j, k + 1, 3)));
          CX(FCP, i, k) = cff * (cff1 * (GL(i, j, k - 1, 3) + GL(i, j, k, 3)) - cff2 * (GL(i, j, k - 2, 3) + GL(i, j, k + 1, 3)));
        }
      cff1 = 1.0 / 3.0; cff2 = 5.0 / 6.0;
      const double cff3 = 1.0 / 6.0;
      for (int i = Istr; i <= Iend; i++) {
        double cff = 0.5 * (W[XW(i, j, 0)] + W[XW(i, j, 1)]);
        CX(FCK, i, 1) = cff * (cff1 * TK(i, j, 0, 3) + cff2 * TK(i, j, 1, 3) - cff3 * TK(i, j, 2, 3));
        CX(FCP, i, 1) = cff * (cff1 * GL(i, j, 0, 3) + cff2 * GL(i, j, 1, 3) - cff3 * GL(i, j, 2, 3));
        cff = 0.5 * (W[XW(i, j, N)] + W[XW(i, j, N - 1)]);
        CX(FCK, i, N) = cff * (cff1 * TK(i, j, N, 3) + cff2 * TK(i, j, N - 1, 3) - cff3 * TK(i, j, N - 2, 3));
        CX(FCP, i, N) = cff * (cff1 * GL(i, j, N, 3) + cff2 * GL(i, j, N - 1, 3) - cff3 * GL(i, j, N - 2, 3));
      }
    }
    for (int k = 1; k <= N - 1; k++)                                      /* :764-776 */
      for (int i = Istr; i <= Iend; i++) {
        const double cff = dt * pm[X2(i, j)] * pn[X2(i, j)];
        TK(i, j, k, nnew) = TK(i, j, k, nnew) - cff * (CX(FCK, i, k + 1) - CX(FCK, i, k));
        if (!my25) TK(i, j, k, nnew) = MAX(TK(i, j, k, nnew), gls_Kmin);
        GL(i, j, k, nnew) = GL(i, j, k, nnew) - cff * (CX(FCP, i, k + 1) - CX(FCP, i, k));
        if (!my25) GL(i, j, k, nnew) = MAX(GL(i, j, k, nnew), gls_Pmin);
      }
    if (my25) { my25_column(o, b, j, shear2, buoy2, BCK, BCP, CF, FCK); continue; }
    /* vertical mixing of the turbulent fields :786-800 */
    {
      const double cff = -0.5 * dt;
      for (int i = Istr; i <= Iend; i++) {
        for (int k = 2; k <= N - 1; k++) {
          CX(FCK, i, k) = cff * (Akk[XW(i, j, k)] + Akk[XW(i, j, k - 1)]) / Hz[X3(i, j, k)];
          CX(FCP, i, k) = cff * (Akp[XW(i, j, k)] + Akp[XW(i, j, k - 1)]) / Hz[X3(i, j, k)];
          CX(CF, i, k) = 0.0;
        }
        CX(FCP, i, 1) = 0.0; CX(FCP, i, N) = 0.0; CX(FCK, i, 1) = 0.0; CX(FCK, i, N) = 0.0;
      }
    }
    /* production and dissipation :804-900 */
    for (int i = Istr; i <= Iend; i++)
      for (int k = 1; k <= N - 1; k++) {
        const double strat2 = BU(i, j, k);
        const double gls_c3 = strat2 > 0.0 ? gls_c3m : gls_c3p;
        const double dAkt = Akt[XW4(i, j, k, 1)] - Akt_bak1, dAkv = Akv[XW(i, j, k)] - Akv_bak;
        double Kprod = SH(i, j, k) * dAkv - strat2 * dAkt;
        double Pprod = gls_c1 * SH(i, j, k) * dAkv - gls_c3 * strat2 * dAkt;
        double cff1 = 1.0;
        if (Kprod < 0.0) { Kprod = Kprod + strat2 * dAkt; cff1 = 0.0; }
        double cff2 = 1.0;
        if (Pprod < 0.0) { Pprod = Pprod + gls_c3 * strat2 * dAkt; cff2 = 0.0; }
        const double cff = 0.5 * (Hz[X3(i, j, k)] + Hz[X3(i, j, k + 1)]);
        TK(i, j, k, nnew) = TK(i, j, k, nnew) + dt * cff * Kprod;
        GL(i, j, k, nnew) = GL(i, j, k, nnew) + dt * cff * Pprod * GL(i, j, k, nstp) / MAX(TK(i, j, k, nstp), gls_Kmin);
        double wall_fac = 1.0;
        if (Lmy25) {
          const double a = pow(GL(i, j, k, nstp), gls_exp1) * cmu_fac1 * pow(TK(i, j, k, nstp), -tke_exp1) * (1.0 / (z_w[XW(i, j, k)] - z_w[XW(i, j, 0)]));
          const double e = pow(GL(i, j, k, nstp), gls_exp1) * cmu_fac1 * pow(TK(i, j, k, nstp), -tke_exp1) * (1.0 / (z_w[XW(i, j, N)] - z_w[XW(i, j, k)]));
          wall_fac = 1.0 + q.E2 / (vonKar * vonKar) * (a * a) + 0.25 / (vonKar * vonKar) * (e * e);
        }
        CX(BCK, i, k) = cff * (1.0 + dt * pow(GL(i, j, k, nstp), -gls_exp1) * cmu_fac2 * pow(TK(i, j, k, nstp), tke_exp2) +
                               dt * (1.0 - cff1) * strat2 * dAkt / TK(i, j, k, nstp)) - CX(FCK, i, k) - CX(FCK, i, k + 1);
        CX(BCP, i, k) = cff * (1.0 + dt * gls_c2 * wall_fac * pow(GL(i, j, k, nstp), -gls_exp1) * cmu_fac2 * pow(TK(i, j, k, nstp), tke_exp2) +
                               dt * (1.0 - cff2) * gls_c3 * strat2 * dAkt / TK(i, j, k, nstp)) - CX(FCP, i, k) - CX(FCP, i, k + 1);
      }
    /* surface and bottom conditions :912-955 */
    for (int i = Istr; i <= Iend; i++) {
      const double sstr = 0.5 * sqrt((sustr[X2(i, j)] + sustr[X2(i + 1, j)]) * (sustr[X2(i, j)] + sustr[X2(i + 1, j)]) +
                                     (svstr[X2(i, j)] + svstr[X2(i, j + 1)]) * (svstr[X2(i, j)] + svstr[X2(i, j + 1)]));
      const double bstr = 0.5 * sqrt((bustr[X2(i, j)] + bustr[X2(i + 1, j)]) * (bustr[X2(i, j)] + bustr[X2(i + 1, j)]) +
                                     (bvstr[X2(i, j)] + bvstr[X2(i, j + 1)]) * (bvstr[X2(i, j)] + bvstr[X2(i, j + 1)]));
      if (crgban) TK(i, j, N, nnew) = MAX(cmu_fac4 * sstr * pow(c->crgban_cw, 2.0 / 3.0), gls_Kmin);
      else TK(i, j, N, nnew) = MAX(cmu_fac3 * sstr, gls_Kmin);
      TK(i, j, 0, nnew) = MAX(cmu_fac3 * bstr, gls_Kmin);
      if (flags & ORC_GLS_CHARNOK) V1(Zos_eff, i) = MAX(c->charnok_alpha / g * sstr, Zos_min);
      else V1(Zos_eff, i) = Zos_min;
      GL(i, j, N, nnew) = MAX(pow(gls_cmu0, gls_p) * pow(TK(i, j, N, nnew), gls_m) * pow(L_sft * V1(Zos_eff, i), gls_n), gls_Pmin);
      const double cff = gls_fac4 * pow(vonKar * Zob_min, gls_n);
      GL(i, j, 0, nnew) = MAX(cff * pow(TK(i, j, 0, nnew), gls_m), gls_Pmin);
    }
    /* tridiagonal system for tke :959-990 */
    for (int i = Istr; i <= Iend; i++) {
      if (crgban) {
        const double sstr = 0.50 * sqrt((sustr[X2(i, j)] + sustr[X2(i + 1, j)]) * (sustr[X2(i, j)] + sustr[X2(i + 1, j)]) +
                                        (svstr[X2(i, j)] + svstr[X2(i, j + 1)]) * (svstr[X2(i, j)] + svstr[X2(i, j + 1)]));
        V1(tke_fluxt, i) = dt * c->crgban_cw * pow(sstr, 1.5);
      } else V1(tke_fluxt, i) = 0.0;
      V1(tke_fluxb, i) = 0.0;
      const double cff = 1.0 / CX(BCK, i, N - 1);
      CX(CF, i, N - 1) = cff * CX(FCK, i, N - 1);
      TK(i, j, N - 1, nnew) = cff * (TK(i, j, N - 1, nnew) + V1(tke_fluxt, i));
    }
    for (int i = Istr; i <= Iend; i++) {
      double cff = 1.0 / CX(BCK, i, N - 1);           /* the value left by the loop above for this i when N-2 < 1 */
      for (int k = N - 2; k >= 1; k--) {
        cff = 1.0 / (CX(BCK, i, k) - CX(CF, i, k + 1) * CX(FCK, i, k + 1));
        CX(CF, i, k) = cff * CX(FCK, i, k);
        TK(i, j, k, nnew) = cff * (TK(i, j, k, nnew) - CX(FCK, i, k + 1) * TK(i, j, k + 1, nnew));
      }
      TK(i, j, 1, nnew) = TK(i, j, 1, nnew) - cff * V1(tke_fluxb, i);
    }
    for (int k = 2; k <= N - 1; k++)
      for (int i = Istr; i <= Iend; i++) TK(i, j, k, nnew) = TK(i, j, k, nnew) - CX(CF, i, k) * TK(i, j, k - 1, nnew);
    /* tridiagonal system for gls :994-1050 */
    for (int i = Istr; i <= Iend; i++) {
      double cff = 0.5 * (TK(i, j, N, nnew) + TK(i, j, N - 1, nnew));
      V1(gls_fluxt, i) = dt * gls_fac3 * pow(cff, gls_m) * pow(L_sft, gls_n) * pow(V1(Zos_eff, i) + 0.5 * Hz[X3(i, j, N)], gls_n - 1.0) *
                         0.5 * (Akp[XW(i, j, N)] + Akp[XW(i, j, N - 1)]);
      if (crgban) {
        const double sstr = 0.5 * sqrt((sustr[X2(i, j)] + sustr[X2(i + 1, j)]) * (sustr[X2(i, j)] + sustr[X2(i + 1, j)]) +
                                       (svstr[X2(i, j)] + svstr[X2(i, j + 1)]) * (svstr[X2(i, j)] + svstr[X2(i, j + 1)]));
        V1(gls_fluxt, i) = V1(gls_fluxt, i) - dt * gls_m * pow(gls_cmu0, gls_p) * pow(cff, gls_m - 1.0) *
                           pow((V1(Zos_eff, i) + 0.5 * Hz[X3(i, j, N)]) * L_sft, gls_n) * gls_sigk * ogls_sigp * c->crgban_cw * pow(sstr, 1.5);
      }
      cff = 0.5 * (TK(i, j, 0, nnew) + TK(i, j, 1, nnew));
      V1(gls_fluxb, i) = dt * gls_fac2 * pow(cff, gls_m) * pow(0.5 * Hz[X3(i, j, 1)] + Zob_min, gls_n - 1.0) *
                         0.5 * (Akp[XW(i, j, 0)] + Akp[XW(i, j, 1)]);
      cff = 1.0 / CX(BCP, i, N - 1);
      CX(CF, i, N - 1) = cff * CX(FCP, i, N - 1);
      GL(i, j, N - 1, nnew) = cff * (GL(i, j, N - 1, nnew) - V1(gls_fluxt, i));
    }
    for (int i = Istr; i <= Iend; i++) {
      double cff = 1.0 / CX(BCP, i, N - 1);
      for (int k = N - 2; k >= 1; k--) {
        cff = 1.0 / (CX(BCP, i, k) - CX(CF, i, k + 1) * CX(FCP, i, k + 1));
        CX(CF, i, k) = cff * CX(FCP, i, k);
        GL(i, j, k, nnew) = cff * (GL(i, j, k, nnew) - CX(FCP, i, k + 1) * GL(i, j, k + 1, nnew));
      }
      GL(i, j, 1, nnew) = GL(i, j, 1, nnew) - cff * V1(gls_fluxb, i);
    }
    for (int k = 2; k <= N - 1; k++)
      for (int i = Istr; i <= Iend; i++) GL(i, j, k, nnew) = GL(i, j, k, nnew) - CX(CF, i, k) * GL(i, j, k - 1, nnew);
    /* vertical mixing coefficients :1058-1190 */
    for (int i = Istr; i <= Iend; i++) {
      for (int k = 1; k <= N - 1; k++) {
        TK(i, j, k, nnew) = MAX(TK(i, j, k, nnew), gls_Kmin);
        GL(i, j, k, nnew) = MAX(GL(i, j, k, nnew), gls_Pmin);
        const double lim = gls_fac5 * pow(TK(i, j, k, nnew), tke_exp4) * pow(sqrt(MAX(0.0, BU(i, j, k))) + eps, -gls_n);
        if (gls_n >= 0.0) GL(i, j, k, nnew) = MIN(GL(i, j, k, nnew), lim);
        else GL(i, j, k, nnew) = MAX(GL(i, j, k, nnew), lim);
        const double Ls_unlmt = MAX(eps, pow(GL(i, j, k, nnew), gls_exp1) * cmu_fac1 * pow(TK(i, j, k, nnew), -tke_exp1));
        double Ls_lmt;
        if (BU(i, j, k) > 0.0) Ls_lmt = MIN(Ls_unlmt, sqrt(0.56 * TK(i, j, k, nnew) / (MAX(0.0, BU(i, j, k)) + eps)));
        else Ls_lmt = Ls_unlmt;
        GL(i, j, k, nnew) = MAX(pow(gls_cmu0, gls_p) * pow(TK(i, j, k, nnew), gls_m) * pow(Ls_lmt, gls_n), gls_Pmin);
        double Gh = MIN(q.Gh0, -BU(i, j, k) * Ls_lmt * Ls_lmt / (2.0 * TK(i, j, k, nnew)));
        Gh = MIN(Gh, Gh - ((Gh - q.Ghcri) * (Gh - q.Ghcri)) / (Gh + q.Gh0 - 2.0 * q.Ghcri));
        Gh = MAX(Gh, q.Ghmin);
        double Sm, Sh;
        if (canuto) {
          double Gm = (q.b0 / gls_fac6 - q.b1 * Gh + q.b3 * gls_fac6 * (Gh * Gh)) / (q.b2 - q.b4 * gls_fac6 * Gh);
          Gm = MIN(Gm, SH(i, j, k) * Ls_lmt * Ls_lmt / (2.0 * TK(i, j, k, nnew)));
          const double cff = q.b0 - q.b1 * gls_fac6 * Gh + q.b2 * gls_fac6 * Gm + q.b3 * (gls_fac6 * gls_fac6) * (Gh * Gh) -
                             q.b4 * (gls_fac6 * gls_fac6) * Gh * Gm + q.b5 * (gls_fac6 * gls_fac6) * Gm * Gm;
          Sm = (q.s0 - q.s1 * gls_fac6 * Gh + q.s2 * gls_fac6 * Gm) / cff;
          Sh = (q.s4 - q.s5 * gls_fac6 * Gh + q.s6 * gls_fac6 * Gm) / cff;
          Sm = MAX(Sm, 0.0);
          Sh = MAX(Sh, 0.0);
          Sm = Sm * sqrt2 / (gls_cmu0 * gls_cmu0 * gls_cmu0);
          Sh = Sh * sqrt2 / (gls_cmu0 * gls_cmu0 * gls_cmu0);
        } else if (kc) {
          const double cff = 1.0 - q.my_Sh2 * Gh;
          Sh = q.my_Sh1 / cff;
          Sm = (q.my_B1pm1o3 + q.my_Sm4 * Sh * Gh) / (1.0 - q.my_Sm2 * Gh);
        } else {
          const double cff = 1.0 - q.my_Sh2 * Gh;
          Sh = q.my_Sh1 / cff;
          Sm = (q.my_Sm3 + Sh * Gh * q.my_Sm4) / (1.0 - q.my_Sm2 * Gh);
        }
        const double ql = sqrt2 * 0.5 * (Ls_lmt * sqrt(TK(i, j, k, nnew)) + Lscale[XW(i, j, k)] * sqrt(TK(i, j, k, nstp)));
        Akv[XW(i, j, k)] = Akv_bak + Sm * ql;
        for (int it = 1; it <= NAT; it++) Akt[XW4(i, j, k, it)] = c->Akt_bak[it - 1] + Sh * ql;
        Akk[XW(i, j, k)] = Akk_bak + Sm * ql / gls_sigk;
        if (crgban) {
          const double Pprod = gls_c1 * SH(i, j, k) * Akv[XW(i, j, k)];
          const double cff = cmu_fac2 * pow(TK(i, j, k, nnew), 1.5 + tke_exp1) * pow(GL(i, j, k, nnew), -1.0 / gls_n);
          const double cff2 = MIN(Pprod / cff, 1.0);
          const double sig_eff = cff2 * gls_sigp + (1.0 - cff2) * gls_sigp_cb;
          Akp[XW(i, j, k)] = Akp_bak + Sm * ql / sig_eff;
        } else Akp[XW(i, j, k)] = Akp_bak + Sm * ql * ogls_sigp;
        Lscale[XW(i, j, k)] = Ls_lmt;
      }
      Akv[XW(i, j, N)] = Akv_bak + L_sft * V1(Zos_eff, i) * gls_cmu0 * sqrt(TK(i, j, N, nnew));
      Akv[XW(i, j, 0)] = Akv_bak + vonKar * Zob_min * gls_cmu0 * sqrt(TK(i, j, 0, nnew));
      Akk[XW(i, j, N)] = Akk_bak + Akv[XW(i, j, N)] / gls_sigk;
      Akk[XW(i, j, 0)] = Akk_bak + Akv[XW(i, j, 0)] / gls_sigk;
      Akp[XW(i, j, N)] = Akp_bak + Akv[XW(i, j, N)] * ogls_sigp;
      Akp[XW(i, j, 0)] = Akp_bak + Akv[XW(i, j, 0)] / gls_sigp;
      for (int it = 1; it <= NAT; it++) { Akt[XW4(i, j, N, it)] = c->Akt_bak[it - 1]; Akt[XW4(i, j, 0, it)] = c->Akt_bak[it - 1]; }
    }
  }
  /* lateral conditions of Akt, Akv :1196-1280 (as lmd_vmix.F:560-700), tke and gls :1282 */
  for (int k = 0; k <= N; k++)
    for (int f = 0; f <= NAT; f++) {
      double *A = f == 0 ? Akv + XW(LBi, LBj, k) : Akt + XW4(LBi, LBj, k, f);
      if (b->west) for (int j = Jstr; j <= Jend; j++) A[X2(Istr - 1, j)] = A[X2(Istr, j)];
      /* my25_corstep.F:786-792 copies to Iend-1, an interior column, and leaves Iend+1 as it was: restated as it stands */
      if (b->east) for (int j = Jstr; j <= Jend; j++) A[X2(my25 ? Iend - 1 : Iend + 1, j)] = A[X2(Iend, j)];
      if (b->south) for (int i = Istr; i <= Iend; i++) A[X2(i, Jstr - 1)] = A[X2(i, Jstr)];
      if (b->north) for (int i = Istr; i <= Iend; i++) A[X2(i, Jend + 1)] = A[X2(i, Jend)];
      if (b->sw) A[X2(Istr - 1, Jstr - 1)] = 0.5 * (A[X2(Istr, Jstr - 1)] + A[X2(Istr - 1, Jstr)]);
      if (b->se) A[X2(Iend + 1, Jstr - 1)] = 0.5 * (A[X2(Iend, Jstr - 1)] + A[X2(Iend + 1, Jstr)]);
      if (b->nw) A[X2(Istr - 1, Jend + 1)] = 0.5 * (A[X2(Istr, Jend + 1)] + A[X2(Istr - 1, Jend)]);
      if (b->ne) A[X2(Iend + 1, Jend + 1)] = 0.5 * (A[X2(Iend, Jend + 1)] + A[X2(Iend + 1, Jend)]);
    }
  tkebc(o, b, nnew);
  if (c->EWperiodic || c->NSperiodic) {
    orc_exchange3d(o, b, 'r', tke + (size_t)(nnew - 1) * nij * (N + 1), N + 1);
    orc_exchange3d(o, b, 'r', gls + (size_t)(nnew - 1) * nij * (N + 1), N + 1);
    orc_exchange3d(o, b, 'r', Akv, N + 1);
    for (int it = 0; it < NAT; it++) orc_exchange3d(o, b, 'r', Akt + (size_t)it * nij * (N + 1), N + 1);
  }
  free(shear2); free(p2); free(cw); free(Zos_eff);
}

void orc_gls_corstep(orc_t *o, int tile) { corstep(o, tile, 0); }
/* MY25_MIXING: my25_prestep.F is gls_prestep.F word for word */
void orc_my25_prestep(orc_t *o, int tile) { orc_gls_prestep(o, tile); }
void orc_my25_corstep(orc_t *o, int tile) { corstep(o, tile, 1); }
