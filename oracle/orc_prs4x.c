/* orc_prs4x.c -- TEST INFRASTRUCTURE (CPU oracle; never part of the product path).
 *
 * The finite-volume pressure Jacobians of Shchepetkin & McWilliams (2003) with a reconstructed vertical density profile:
 *   orc_prsgrd42   ROMS/Nonlinear/prsgrd42.h:227-482  (PJ_GRADPQ2: parabolic WENO reconstruction, PPM-style limiter, and the
 *                                                      second pass over ru, rv)
 *   orc_prsgrd44   ROMS/Nonlinear/prsgrd44.h:224-508  (PJ_GRADPQ4: quartic reconstruction, power-law reconciliation)
 * selected by prsgrd.F:16-19; neither file defines NEUMANN (prsgrd42.h:1, prsgrd44.h:1: #undef).  Statement for statement in
 * the reference's loop order; pinned against the reference built with oracle/ref/upwelling_prs42.h / upwelling_prs44.h
 * (tests/test_oracle_vs_ref.py).
 *
 * prsgrd42's second pass reads rv(i+1,j,k,nrhs) at i = Iend (prsgrd42.h:449-455), a point its first pass does not compute
 * (:399, Istr:Iend): what is there is what the array held -- in a single tile nothing ever writes that column.  With more
 * than one tile the reference's result depends on the order in which tiles run (shared memory) or is the zero of a private
 * ghost column (distributed memory): the oracle, like the library, is pinned for one tile.
 */
#include "orc.h"
#include <math.h>
#include <stdlib.h>

#define CX(A, i, k) A[(size_t)((i) - LBi) + (size_t)(k) * ni]

static inline double d_max(double a, double b) { return a > b ? a : b; }
static inline double d_min(double a, double b) { return a < b ? a : b; }

/* the PPM-style limiter both passes of prsgrd42 use (prsgrd42.h:337-345, :367-375, :410-418) */
static inline double ppm_rr(double deltaR, double deltaL) {
  if ((deltaR * deltaL) < 0.0) return 0.0;
  if (fabs(deltaR) > (2.0 * fabs(deltaL))) return 3.0 * deltaL;
  if (fabs(deltaL) > (2.0 * fabs(deltaR))) return 3.0 * deltaR;
  return deltaR + deltaL;
}

/* the parabolic WENO side limits of one row (prsgrd42.h:246-284 = prsgrd44.h:246-284 with d in FC's place):
   D(i,k) = (rho(k+1)-rho(k))/(Hz(k+1)+Hz(k)), k = 1..N-1, is given */
#define WENO_SIDES(Dk, Dkm1, i0, i1)                                                                     \
  for (int k = 2; k <= N - 1; k++)                                                                       \
    for (int i = (i0); i <= (i1); i++) {                                                                 \
      double deltaR = Hz[X3(i, j, k)] * (Dk);                                                            \
      double deltaL = Hz[X3(i, j, k)] * (Dkm1);                                                          \
      if ((deltaR * deltaL) < 0.0) { deltaR = 0.0; deltaL = 0.0; }                                       \
      double cff = Hz[X3(i, j, k - 1)] + 2.0 * Hz[X3(i, j, k)] + Hz[X3(i, j, k + 1)];                    \
      const double cffR = cff * (Dk);                                                                    \
      const double cffL = cff * (Dkm1);                                                                  \
      if (fabs(deltaR) > fabs(cffL)) deltaR = cffL;                                                      \
      if (fabs(deltaL) > fabs(cffR)) deltaL = cffR;                                                      \
      cff = (deltaR - deltaL) / (Hz[X3(i, j, k - 1)] + Hz[X3(i, j, k)] + Hz[X3(i, j, k + 1)]);           \
      deltaR = deltaR - cff * Hz[X3(i, j, k + 1)];                                                       \
      deltaL = deltaL + cff * Hz[X3(i, j, k - 1)];                                                       \
      CX(aR, i, k) = rho[X3(i, j, k)] + deltaR;                                                          \
      CX(aL, i, k) = rho[X3(i, j, k)] - deltaL;                                                          \
      CX(dR, i, k) = (2.0 * deltaR - deltaL) * (2.0 * deltaR - deltaL);                                  \
      CX(dL, i, k) = (2.0 * deltaL - deltaR) * (2.0 * deltaL - deltaR);                                  \
    }                                                                                                    \
  for (int i = (i0); i <= (i1); i++) {                                                                   \
    CX(aL, i, N) = CX(aR, i, N - 1);                                                                     \
    CX(aR, i, N) = 2.0 * rho[X3(i, j, N)] - CX(aL, i, N);                                                \
    { const double q = 2.0 * CX(aR, i, N) + CX(aL, i, N) - 3.0 * rho[X3(i, j, N)]; CX(dR, i, N) = q * q; } \
    { const double q = 3.0 * rho[X3(i, j, N)] - 2.0 * CX(aL, i, N) - CX(aR, i, N); CX(dL, i, N) = q * q; } \
    CX(aR, i, 1) = CX(aL, i, 2);                                                                         \
    CX(aL, i, 1) = 2.0 * rho[X3(i, j, 1)] - CX(aR, i, 1);                                                \
    { const double q = 2.0 * CX(aR, i, 1) + CX(aL, i, 1) - 3.0 * rho[X3(i, j, 1)]; CX(dR, i, 1) = q * q; } \
    { const double q = 3.0 * rho[X3(i, j, 1)] - 2.0 * CX(aL, i, 1) - CX(aR, i, 1); CX(dL, i, 1) = q * q; } \
  }

/* prsgrd42_tile, prsgrd42.h:227-482 */
void orc_prsgrd42(orc_t *o, int tile) {
  ORC_LOCALS(o);
  const orc_bounds *b = &o->b[tile];
  const int nrhs = o->s.nrhs;
  const int Istr = b->Istr, Iend = b->Iend, Jstr = b->Jstr, Jend = b->Jend, IstrU = b->IstrU, JstrV = b->JstrV;
  const int masking = (o->c.options & ORC_MASKING) != 0;
  const double g = o->c.g, rho0 = o->c.rho0, eps = 1.0E-8;
  double *rho = o->rho, *z_w = o->z_w, *Hz = o->Hz, *ru = o->ru, *rv = o->rv;
  const size_t n3 = nij * (size_t)(N + 1), n2 = ni * (size_t)(N + 1);
  double *P = (double *)calloc(n3, sizeof(double)), *FX = (double *)calloc(n3, sizeof(double)), *r = (double *)calloc(n3, sizeof(double));
  double *FC = (double *)calloc(n2, sizeof(double)), *aL = (double *)calloc(n2, sizeof(double)), *aR = (double *)calloc(n2, sizeof(double));
  double *dL = (double *)calloc(n2, sizeof(double)), *dR = (double *)calloc(n2, sizeof(double));
  const double cff2 = 1.0 / 6.0;
  for (int j = JstrV - 2; j <= Jend + 1; j++) {
    for (int k = N - 1; k >= 1; k--)
      for (int i = IstrU - 2; i <= Iend + 1; i++)
        CX(FC, i, k) = (rho[X3(i, j, k + 1)] - rho[X3(i, j, k)]) / (Hz[X3(i, j, k + 1)] + Hz[X3(i, j, k)]);     /* :240 */
    WENO_SIDES(CX(FC, i, k), CX(FC, i, k - 1), IstrU - 2, Iend + 1)                                               /* :250-284 */
    for (int k = 1; k <= N - 1; k++)
      for (int i = IstrU - 2; i <= Iend + 1; i++) {                                                               /* :286-293 */
        const double deltaL = d_max(CX(dL, i, k), eps);
        const double deltaR = d_max(CX(dR, i, k + 1), eps);
        r[XW(i, j, k)] = (deltaR * CX(aR, i, k) + deltaL * CX(aL, i, k + 1)) / (deltaR + deltaL);
      }
    for (int i = IstrU - 2; i <= Iend + 1; i++) {                                                                 /* :300-301 (no NEUMANN) */
      r[XW(i, j, N)] = 2.0 * rho[X3(i, j, N)] - r[XW(i, j, N - 1)];
      r[XW(i, j, 0)] = 2.0 * rho[X3(i, j, 1)] - r[XW(i, j, 1)];
    }
    for (int i = IstrU - 2; i <= Iend + 1; i++) P[XW(i, j, N)] = 0.0;                                             /* :309 */
    for (int k = N; k >= 1; k--)
      for (int i = IstrU - 2; i <= Iend + 1; i++) {                                                               /* :321-338 */
        P[XW(i, j, k - 1)] = P[XW(i, j, k)] + Hz[X3(i, j, k)] * rho[X3(i, j, k)];
        const double deltaR = r[XW(i, j, k)] - rho[X3(i, j, k)];
        const double deltaL = rho[X3(i, j, k)] - r[XW(i, j, k - 1)];
        const double rr = ppm_rr(deltaR, deltaL);
        FX[XW(i, j, k)] = 0.5 * Hz[X3(i, j, k)] * (P[XW(i, j, k)] + P[XW(i, j, k - 1)] + cff2 * rr * Hz[X3(i, j, k)]);
      }
    if ((j >= Jstr) && (j <= Jend)) {                                                                             /* :344-376 */
      for (int i = IstrU - 1; i <= Iend + 1; i++) CX(FC, i, N) = 0.0;
      for (int k = N; k >= 1; k--)
        for (int i = IstrU - 1; i <= Iend + 1; i++) {
          const double delP = P[XW(i - 1, j, k - 1)] - P[XW(i, j, k - 1)];
          const double dh = z_w[XW(i, j, k - 1)] - z_w[XW(i - 1, j, k - 1)];
          const double deltaR = dh * r[XW(i, j, k - 1)] - delP;
          const double deltaL = delP - dh * r[XW(i - 1, j, k - 1)];
          const double rr = ppm_rr(deltaR, deltaL);
          CX(FC, i, k - 1) = 0.5 * dh * (P[XW(i, j, k - 1)] + P[XW(i - 1, j, k - 1)] + cff2 * rr);
          ru[XW4(i, j, k, nrhs)] = 2.0 * (FX[XW(i - 1, j, k)] - FX[XW(i, j, k)] + CX(FC, i, k) - CX(FC, i, k - 1)) /
                                   (Hz[X3(i - 1, j, k)] + Hz[X3(i, j, k)]);
          if (masking) ru[XW4(i, j, k, nrhs)] = ru[XW4(i, j, k, nrhs)] * o->umask[X2(i, j)];
        }
    }
    if (j >= JstrV - 1) {                                                                                         /* :382-414 */
      for (int i = Istr; i <= Iend; i++) CX(FC, i, N) = 0.0;
      for (int k = N; k >= 1; k--)
        for (int i = Istr; i <= Iend; i++) {
          const double delP = P[XW(i, j - 1, k - 1)] - P[XW(i, j, k - 1)];
          const double dh = z_w[XW(i, j, k - 1)] - z_w[XW(i, j - 1, k - 1)];
          const double deltaR = dh * r[XW(i, j, k - 1)] - delP;
          const double deltaL = delP - dh * r[XW(i, j - 1, k - 1)];
          const double rr = ppm_rr(deltaR, deltaL);
          CX(FC, i, k - 1) = 0.5 * dh * (P[XW(i, j, k - 1)] + P[XW(i, j - 1, k - 1)] + cff2 * rr);
          rv[XW4(i, j, k, nrhs)] = 2.0 * (FX[XW(i, j - 1, k)] - FX[XW(i, j, k)] + CX(FC, i, k) - CX(FC, i, k - 1)) /
                                   (Hz[X3(i, j - 1, k)] + Hz[X3(i, j, k)]);
          if (masking) rv[XW4(i, j, k, nrhs)] = rv[XW4(i, j, k, nrhs)] * o->vmask[X2(i, j)];
        }
    }
  }
  /* ---- the second pass :417-480 */
  {
    const double rr = g / (24.0 * rho0), cff = 0.5 * g, cff1 = 0.5 * g / rho0;
    for (int j = Jstr; j <= Jend; j++) {
      for (int k = N - 1; k >= 1; k--)
        for (int i = IstrU; i <= Iend; i++) {
          const double dh = rr * (z_w[XW(i, j, k)] - z_w[XW(i - 1, j, k)]);
          CX(FC, i, k) = d_max(dh, 0.0) * (ru[XW4(i, j, k + 1, nrhs)] + ru[XW4(i + 1, j, k, nrhs)] - ru[XW4(i, j, k, nrhs)] - ru[XW4(i - 1, j, k + 1, nrhs)]) +
                         d_min(dh, 0.0) * (ru[XW4(i, j, k, nrhs)] + ru[XW4(i + 1, j, k + 1, nrhs)] - ru[XW4(i, j, k + 1, nrhs)] - ru[XW4(i - 1, j, k, nrhs)]);
        }
      for (int i = IstrU; i <= Iend; i++) {
        CX(FC, i, N) = 0.0;
        const double dh = rr * (z_w[XW(i, j, 0)] - z_w[XW(i - 1, j, 0)]);
        CX(FC, i, 0) = d_max(dh, 0.0) * (ru[XW4(i, j, 1, nrhs)] - ru[XW4(i - 1, j, 1, nrhs)]) +
                       d_min(dh, 0.0) * (ru[XW4(i + 1, j, 1, nrhs)] - ru[XW4(i, j, 1, nrhs)]);
      }
      for (int k = 1; k <= N; k++)
        for (int i = IstrU; i <= Iend; i++)
          ru[XW4(i, j, k, nrhs)] = (cff * (z_w[XW(i - 1, j, N)] - z_w[XW(i, j, N)]) + cff1 * ru[XW4(i, j, k, nrhs)]) *
                                       (Hz[X3(i - 1, j, k)] + Hz[X3(i, j, k)]) * o->on_u[X2(i, j)] +
                                   (CX(FC, i, k) - CX(FC, i, k - 1)) * o->on_u[X2(i, j)];
    }
    for (int j = JstrV; j <= Jend; j++) {
      for (int k = N - 1; k >= 1; k--)
        for (int i = Istr; i <= Iend; i++) {
          const double dh = rr * (z_w[XW(i, j, k)] - z_w[XW(i, j - 1, k)]);
          FX[XW(i, j, k)] = d_max(dh, 0.0) * (rv[XW4(i, j, k + 1, nrhs)] + rv[XW4(i + 1, j, k, nrhs)] - rv[XW4(i, j, k, nrhs)] - rv[XW4(i, j - 1, k + 1, nrhs)]) +
                            d_min(dh, 0.0) * (rv[XW4(i, j, k, nrhs)] + rv[XW4(i + 1, j, k + 1, nrhs)] - rv[XW4(i, j, k + 1, nrhs)] - rv[XW4(i, j - 1, k, nrhs)]);
        }
      for (int i = Istr; i <= Iend; i++) {
        FX[XW(i, j, N)] = 0.0;
        const double dh = rr * (z_w[XW(i, j, 0)] - z_w[XW(i, j - 1, 0)]);
        FX[XW(i, j, 0)] = d_max(dh, 0.0) * (rv[XW4(i, j, 1, nrhs)] - rv[XW4(i, j - 1, 1, nrhs)]) +
                          d_min(dh, 0.0) * (rv[XW4(i + 1, j, 1, nrhs)] - rv[XW4(i, j, 1, nrhs)]);
      }
    }
    for (int j = JstrV; j <= Jend; j++)
      for (int k = 1; k <= N; k++)
        for (int i = Istr; i <= Iend; i++)
          rv[XW4(i, j, k, nrhs)] = (cff * (z_w[XW(i, j - 1, N)] - z_w[XW(i, j, N)]) + cff1 * rv[XW4(i, j, k, nrhs)]) *
                                       (Hz[X3(i, j - 1, k)] + Hz[X3(i, j, k)]) * o->om_v[X2(i, j)] +
                                   (FX[XW(i, j, k)] - FX[XW(i, j, k - 1)]) * o->om_v[X2(i, j)];
  }
  free(P); free(FX); free(r); free(FC); free(aL); free(aR); free(dL); free(dR);
}

/* prsgrd44_tile, prsgrd44.h:224-508 */
void orc_prsgrd44(orc_t *o, int tile) {
  ORC_LOCALS(o);
  const orc_bounds *b = &o->b[tile];
  const int nrhs = o->s.nrhs;
  const int Istr = b->Istr, Iend = b->Iend, Jstr = b->Jstr, Jend = b->Jend, IstrU = b->IstrU, JstrV = b->JstrV;
  const double g = o->c.g, rho0 = o->c.rho0, eps = 1.0E-8;
  double *rho = o->rho, *z_w = o->z_w, *Hz = o->Hz, *ru = o->ru, *rv = o->rv;
  const size_t n3 = nij * (size_t)(N + 1), n2 = ni * (size_t)(N + 1);
  double *P = (double *)calloc(n3, sizeof(double)), *FX = (double *)calloc(n3, sizeof(double)), *r = (double *)calloc(n3, sizeof(double));
  double *d = (double *)calloc(n3, sizeof(double));
  double *FC = (double *)calloc(n2, sizeof(double)), *aL = (double *)calloc(n2, sizeof(double)), *aR = (double *)calloc(n2, sizeof(double));
  double *dL = (double *)calloc(n2, sizeof(double)), *dR = (double *)calloc(n2, sizeof(double)), *r1 = (double *)calloc(n2, sizeof(double));
  for (int j = JstrV - 1; j <= Jend; j++) {
    for (int k = N - 1; k >= 1; k--)
      for (int i = IstrU - 1; i <= Iend; i++) {                                                                   /* :232-238 */
        CX(FC, i, k) = 1.0 / (Hz[X3(i, j, k + 1)] + Hz[X3(i, j, k)]);
        r[XW(i, j, k)] = CX(FC, i, k) * (rho[X3(i, j, k + 1)] * Hz[X3(i, j, k)] + rho[X3(i, j, k)] * Hz[X3(i, j, k + 1)]);
        d[XW(i, j, k)] = CX(FC, i, k) * (rho[X3(i, j, k + 1)] - rho[X3(i, j, k)]);
      }
    WENO_SIDES(d[XW(i, j, k)], d[XW(i, j, k - 1)], IstrU - 1, Iend)                                               /* :246-284 */
    for (int k = 1; k <= N - 1; k++)
      for (int i = IstrU - 1; i <= Iend; i++) {                                                                   /* :286-292 */
        const double deltaL = d_max(CX(dL, i, k), eps);
        const double deltaR = d_max(CX(dR, i, k + 1), eps);
        CX(r1, i, k) = (deltaR * CX(aR, i, k) + deltaL * CX(aL, i, k + 1)) / (deltaR + deltaL);
      }
    for (int i = IstrU - 1; i <= Iend; i++) {                                                                     /* :299-300 */
      CX(r1, i, N) = 2.0 * rho[X3(i, j, N)] - CX(r1, i, N - 1);
      CX(r1, i, 0) = 2.0 * rho[X3(i, j, 1)] - CX(r1, i, 1);
    }
    for (int k = 1; k <= N; k++)
      for (int i = IstrU - 1; i <= Iend; i++) {                                                                   /* :316-348: power-law reconciliation */
        const double deltaR = CX(r1, i, k) - rho[X3(i, j, k)];
        const double deltaL = rho[X3(i, j, k)] - CX(r1, i, k - 1);
        double cff = deltaR * deltaL;
        if (cff > eps) cff = (deltaR + deltaL) / cff;
        else cff = 0.0;
        double cffL = cff * deltaL;
        double cffR = cff * deltaR;
        if (cffL > 3.0) {
          cffL = cffL * deltaL;
          cffR = 0.0;
        } else if (cffR > 3.0) {
          cffL = 0.0;
          cffR = cffR * deltaR;
        } else {
          cffL = 4.0 * deltaL - 2.0 * deltaR;
          cffR = 4.0 * deltaR - 2.0 * deltaL;
        }
        cff = 1.0 / Hz[X3(i, j, k)];
        CX(dR, i, k) = cff * cffR;
        CX(dL, i, k) = cff * cffL;
      }
    for (int k = N - 1; k >= 1; k--)
      for (int i = IstrU - 1; i <= Iend; i++) {                                                                   /* :361-391 */
        double dk = CX(FC, i, k) * (Hz[X3(i, j, k + 1)] * CX(dL, i, k + 1) + Hz[X3(i, j, k)] * CX(dR, i, k));
        const double cffR = 8.0 * (CX(dR, i, k) + 2.0 * CX(dL, i, k));
        const double cffL = 8.0 * (CX(dL, i, k + 1) + 2.0 * CX(dR, i, k + 1));
        if (fabs(dk) > fabs(cffR)) dk = cffR;
        if (fabs(dk) > fabs(cffL)) dk = cffL;
        d[XW(i, j, k)] = dk;
        double Hdd, rr;
        if ((CX(dL, i, k + 1) - CX(dR, i, k)) * (rho[X3(i, j, k + 1)] - rho[X3(i, j, k)]) > 0.0) {
          Hdd = Hz[X3(i, j, k)] * (dk - CX(dR, i, k));
          rr = rho[X3(i, j, k)] - CX(r1, i, k - 1);
        } else {
          Hdd = Hz[X3(i, j, k + 1)] * (CX(dL, i, k + 1) - dk);
          rr = CX(r1, i, k + 1) - rho[X3(i, j, k + 1)];
        }
        rr = fabs(rr);
        double Ampl = 0.2 * Hdd * rr;
        Hdd = fabs(Hdd);
        const double cff = rr * rr + 0.0763636363636363636 * Hdd * (rr + 0.004329004329004329 * Hdd);
        if (cff > eps) Ampl = Ampl * (rr + 0.0363636363636363636 * Hdd) / cff;
        else Ampl = 0.0;
        r[XW(i, j, k)] = CX(r1, i, k) + Ampl;
      }
    for (int i = IstrU - 1; i <= Iend; i++) {                                                                     /* :399-402 (no NEUMANN) */
      r[XW(i, j, 0)] = 2.0 * rho[X3(i, j, 1)] - r[XW(i, j, 1)];
      r[XW(i, j, N)] = 2.0 * rho[X3(i, j, N)] - r[XW(i, j, N - 1)];
      d[XW(i, j, 0)] = d[XW(i, j, 1)];
      d[XW(i, j, N)] = d[XW(i, j, N - 1)];
    }
    for (int i = IstrU - 1; i <= Iend; i++) P[XW(i, j, N)] = 0.0;                                                 /* :410 */
    {
      const double cff3 = 1.0 / 12.0;
      for (int k = N; k >= 1; k--)
        for (int i = IstrU - 1; i <= Iend; i++) {                                                                 /* :422-431 */
          P[XW(i, j, k - 1)] = P[XW(i, j, k)] + Hz[X3(i, j, k)] * rho[X3(i, j, k)];
          FX[XW(i, j, k)] = 0.5 * Hz[X3(i, j, k)] *
                            (P[XW(i, j, k)] + P[XW(i, j, k - 1)] +
                             0.2 * Hz[X3(i, j, k)] * (r[XW(i, j, k)] - r[XW(i, j, k - 1)] - cff3 * Hz[X3(i, j, k)] * (d[XW(i, j, k)] + d[XW(i, j, k - 1)])));
        }
    }
    const double cff = 0.5 * g, cff1 = g / rho0, cff2 = 1.0 / 6.0, cff3 = 1.0 / 12.0;
    for (int dir = 0; dir < 2; dir++) {                                                                           /* :436-470 (xi), :472-506 (eta) */
      if (dir == 0 ? !(j >= Jstr) : !(j >= JstrV)) continue;
      const int di = dir == 0 ? 1 : 0, dj = dir == 0 ? 0 : 1, i0 = dir == 0 ? IstrU : Istr;
      double *rq = dir == 0 ? ru : rv;
      const double *omn = dir == 0 ? o->on_u : o->om_v;
      for (int i = i0; i <= Iend; i++) CX(FC, i, N) = 0.0;
      for (int k = N; k >= 1; k--)
        for (int i = i0; i <= Iend; i++) {
          const double dh = z_w[XW(i, j, k - 1)] - z_w[XW(i - di, j - dj, k - 1)];
          const double delP = P[XW(i - di, j - dj, k - 1)] - P[XW(i, j, k - 1)];
          double rr = 0.5 * dh * (r[XW(i, j, k - 1)] + r[XW(i - di, j - dj, k - 1)] - cff2 * dh * (d[XW(i, j, k - 1)] - d[XW(i - di, j - dj, k - 1)]));
          double limtr = 2.0 * delP * rr;
          rr = rr * rr + delP * delP;
          if (limtr > eps * rr) limtr = limtr / rr;
          else limtr = 0.0;
          CX(FC, i, k - 1) = 0.5 * dh *
                             (P[XW(i, j, k - 1)] + P[XW(i - di, j - dj, k - 1)] +
                              limtr * 0.2 * dh * (r[XW(i, j, k - 1)] - r[XW(i - di, j - dj, k - 1)] - cff3 * dh * (d[XW(i, j, k - 1)] + d[XW(i - di, j - dj, k - 1)])));
          rq[XW4(i, j, k, nrhs)] = (cff * (Hz[X3(i - di, j - dj, k)] + Hz[X3(i, j, k)]) * (z_w[XW(i - di, j - dj, N)] - z_w[XW(i, j, N)]) +
                                    cff1 * (FX[XW(i - di, j - dj, k)] - FX[XW(i, j, k)] + CX(FC, i, k) - CX(FC, i, k - 1))) * omn[X2(i, j)];
        }
    }
  }
  free(P); free(FX); free(r); free(d); free(FC); free(aL); free(aR); free(dL); free(dR); free(r1);
}
