/*
 * orc_main3d.c -- step sequencing (main3d), start-up and diagnostics.
 * TEST INFRASTRUCTURE (see orc.h).
 *
 *   orc_main3d_step  main3d     ROMS/Nonlinear/main3d.F:216-1148 (one STEP_LOOP pass;
 *                               LF-AM3 barotropic loop :810-918)       pinned: the reference's own kernels
 *                               called in main3d.F order (oracle/ref/ref_glue.F90:ref_main3d),
 *                               100 steps, every array every step (tests/test_oracle_vs_ref.py)
 *   orc_start        initial    ROMS/Nonlinear/initial.F:549-577 tail: set_massflux,
 *                               omega, rho_eos at iic=ntstart          (driver restated)
 *   orc_diag         diag_tile  ROMS/Nonlinear/diag.F:84-560           pinned (7 digits:
 *                               the reference only prints the values)
 */
#include "orc.h"
#include <math.h>
#include <stdlib.h>

/* Tile loops.  The reference runs them as OpenMP shared-memory tiles with a barrier after each loop
   (Drivers/nl_roms.h:304-310, main3d.F "!$OMP BARRIER"): a tile writes its own range (and its periodic
   images) and reads what the previous loop left, so the tiles of one loop are independent and any
   execution order -- ascending, descending, concurrent -- gives the same bits (tests/test_oracle.py).
   With nthreads > 1 (orc_set_threads, the cpu_baseline leg of bench.py) they run concurrently. */
#define ORC_PRAGMA(x) _Pragma(#x)
#define FWD(o, call)                                                                               \
  ORC_PRAGMA(omp parallel for schedule(static) num_threads((o)->nthreads) if ((o)->nthreads > 1))  \
  for (int tile = 0; tile < (o)->ntiles; tile++) call
#define REV(o, call)                                                                               \
  ORC_PRAGMA(omp parallel for schedule(static) num_threads((o)->nthreads) if ((o)->nthreads > 1))  \
  for (int tile = (o)->ntiles - 1; tile >= 0; tile--) call

/* diag.F:84 -- global kinetic/potential energy, volume, Courant numbers, max speed.
   out: 0 avgke 1 avgpe 2 avgkp 3 volume 4 maxspeed 5 max_Cu 6 max_Cv 7 max_Cw
        8 max_Ci 9 max_Cj 10 max_Ck 11 max_C */
void orc_diag(orc_t *o) {
  ORC_LOCALS(o);
  const orc_cfg *c = &o->c;
  const int idia = o->s.nstp;
  const double g = c->g, dt = c->dt;
  double *u = o->u, *v = o->v, *Hz = o->Hz, *z_w = o->z_w, *z_r = o->z_r, *rho = o->rho,
         *wvel = o->wvel, *pm = o->pm, *pn = o->pn, *omn = o->omn;
  double volume = 0.0, avgke = 0.0, avgpe = 0.0, maxspeed = -1.0E+20;
  double max_C = 0.0, max_Cu = 0.0, max_Cv = 0.0, max_Cw = 0.0;
  int max_Ci = 0, max_Cj = 0, max_Ck = 0;
  double *ke2d = (double *)calloc(2 * nij, sizeof(double)), *pe2d = ke2d + nij;
  for (int tile = 0; tile < o->ntiles; tile++) {
    const orc_bounds *b = &o->b[tile];
    const int Istr = b->Istr, Iend = b->Iend, Jstr = b->Jstr, Jend = b->Jend;
    double my_max_C = 0.0, my_max_Cu = 0.0, my_max_Cv = 0.0, my_max_Cw = 0.0, my_maxspeed = 0.0;
    int my_max_Ci = 0, my_max_Cj = 0, my_max_Ck = 0;
    for (int j = Jstr; j <= Jend; j++) {
      for (int i = Istr; i <= Iend; i++) {
        ke2d[X2(i, j)] = 0.0;
        pe2d[X2(i, j)] = 0.5 * g * z_w[XW(i, j, N)] * z_w[XW(i, j, N)];
      }
      double cff = g / c->rho0;
      for (int k = N; k >= 1; k--)
        for (int i = Istr; i <= Iend; i++) {
          double u2v2 = u[X4(i, j, k, idia)] * u[X4(i, j, k, idia)] +
                        u[X4(i + 1, j, k, idia)] * u[X4(i + 1, j, k, idia)] +
                        v[X4(i, j, k, idia)] * v[X4(i, j, k, idia)] +
                        v[X4(i, j + 1, k, idia)] * v[X4(i, j + 1, k, idia)];
          ke2d[X2(i, j)] = ke2d[X2(i, j)] + Hz[X3(i, j, k)] * 0.25 * u2v2;
          pe2d[X2(i, j)] = pe2d[X2(i, j)] + cff * Hz[X3(i, j, k)] * (rho[X3(i, j, k)] + 1000.0) *
                                                (z_r[X3(i, j, k)] - z_w[XW(i, j, 0)]);
          double my_Cu = 0.5 * fabs(u[X4(i, j, k, idia)] + u[X4(i + 1, j, k, idia)]) * dt * pm[X2(i, j)];
          double my_Cv = 0.5 * fabs(v[X4(i, j, k, idia)] + v[X4(i, j + 1, k, idia)]) * dt * pn[X2(i, j)];
          double my_Cw = 0.5 * fabs(wvel[XW(i, j, k - 1)] + wvel[XW(i, j, k)]) * dt / Hz[X3(i, j, k)];
          double my_C = my_Cu + my_Cv + my_Cw;
          if (my_C > my_max_C) {
            my_max_C = my_C; my_max_Cu = my_Cu; my_max_Cv = my_Cv; my_max_Cw = my_Cw;
            my_max_Ci = i; my_max_Cj = j; my_max_Ck = k;
          }
          double sp = sqrt(0.5 * u2v2);
          if (sp > my_maxspeed) my_maxspeed = sp;
        }
    }
    /* j-then-i summation to limit round-off :289-325 */
    for (int i = Istr; i <= Iend; i++) {
      pe2d[X2(i, Jend + 1)] = 0.0;
      pe2d[X2(i, Jstr - 1)] = 0.0;
      ke2d[X2(i, Jstr - 1)] = 0.0;
    }
    for (int j = Jstr; j <= Jend; j++)
      for (int i = Istr; i <= Iend; i++) {
        pe2d[X2(i, Jend + 1)] = pe2d[X2(i, Jend + 1)] + omn[X2(i, j)] * (z_w[XW(i, j, N)] - z_w[XW(i, j, 0)]);
        pe2d[X2(i, Jstr - 1)] = pe2d[X2(i, Jstr - 1)] + omn[X2(i, j)] * pe2d[X2(i, j)];
        ke2d[X2(i, Jstr - 1)] = ke2d[X2(i, Jstr - 1)] + omn[X2(i, j)] * ke2d[X2(i, j)];
      }
    double my_volume = 0.0, my_avgpe = 0.0, my_avgke = 0.0;
    for (int i = Istr; i <= Iend; i++) {
      my_volume = my_volume + pe2d[X2(i, Jend + 1)];
      my_avgpe = my_avgpe + pe2d[X2(i, Jstr - 1)];
      my_avgke = my_avgke + ke2d[X2(i, Jstr - 1)];
    }
    volume = volume + my_volume;
    avgke = avgke + my_avgke;
    avgpe = avgpe + my_avgpe;
    if (my_maxspeed > maxspeed) maxspeed = my_maxspeed;
    if (my_max_C == max_C) {
      if (my_max_Ci < max_Ci) max_Ci = my_max_Ci;
      if (my_max_Cj < max_Cj) max_Cj = my_max_Cj;
      if (my_max_Ck < max_Ck) max_Ck = my_max_Ck;
    } else if (my_max_C > max_C) {
      max_C = my_max_C; max_Cu = my_max_Cu; max_Cv = my_max_Cv; max_Cw = my_max_Cw;
      max_Ci = my_max_Ci; max_Cj = my_max_Cj; max_Ck = my_max_Ck;
    }
  }
  free(ke2d);
  avgke = avgke / volume;
  avgpe = avgpe / volume;
  o->diag[0] = avgke;
  o->diag[1] = avgpe;
  o->diag[2] = avgke + avgpe;
  o->diag[3] = volume;
  o->diag[4] = maxspeed;
  o->diag[5] = max_Cu;
  o->diag[6] = max_Cv;
  o->diag[7] = max_Cw;
  o->diag[8] = max_Ci;
  o->diag[9] = max_Cj;
  o->diag[10] = max_Ck;
  o->diag[11] = max_C;
}

/* initial.F tail: after the host filled grid + initial fields (nstp=1) */
void orc_start(orc_t *o) {
  orc_step *s = &o->s;
  s->iif = 1; s->indx1 = 1; s->kstp = 1; s->krhs = 1; s->knew = 1; s->predictor = 0;
  s->nstp = 1; s->nrhs = 1; s->nnew = 1;
  s->tdays = o->c.dstart;
  s->time = s->tdays * 86400.0;
  FWD(o, orc_set_massflux(o, tile));
  FWD(o, { orc_omega(o, tile); orc_rho_eos(o, tile); });
  s->iic = o->c.ntstart;
}

/* one pass of STEP_LOOP, main3d.F:216-1148 */
int orc_main3d_step(orc_t *o) {
  orc_step *s = &o->s;
  const orc_cfg *c = &o->c;
  s->nstp = 1 + (s->iic - c->ntstart) % 2;                              /* :220-231 */
  s->nnew = 3 - s->nstp;
  s->nrhs = s->nstp;
  s->tdays = s->time * (1.0 / 86400.0);
  FWD(o, orc_set_data(o, tile));                                        /* :258 */
  if (s->iic == c->ntstart) {                                           /* post_initial :335 */
    FWD(o, { orc_ini_zeta(o, tile); orc_set_depth(o, tile); });
    REV(o, orc_ini_fields(o, tile));
  }
  FWD(o, { orc_set_massflux(o, tile); orc_rho_eos(o, tile); });         /* :348-350 */
  orc_diag(o);                                                          /* :355 */
  if (c->options & ORC_BULK_FLUXES) FWD(o, orc_bulk_flux(o, tile));     /* :439 */
  FWD(o, orc_set_vbc(o, tile));                                         /* :445 */
  if (c->options & ORC_ANA_VMIX) { REV(o, orc_ana_vmix(o, tile)); }     /* :525 */
  else if (c->options & ORC_LMD_MIXING) { REV(o, orc_lmd_vmix(o, tile)); } /* :527 */
  REV(o, { orc_omega(o, tile); orc_wvelocity(o, tile, s->nstp); });     /* :534-535 */
  FWD(o, orc_set_zeta(o, tile));                                        /* :556 */
  if (o->avg) { FWD(o, orc_set_avg(o, tile)); }                         /* :562 (AVERAGES) */
  if (o->dia) { FWD(o, orc_set_diags(o, tile)); }                       /* :559 (DIAGNOSTICS) */
  REV(o, orc_rhs3d(o, tile));                                           /* :632 */
  if (c->options & ORC_MY25_MIXING) { REV(o, orc_my25_prestep(o, tile)); }  /* :634 */
  else if (c->options & ORC_GLS_MIXING) { REV(o, orc_gls_prestep(o, tile)); }  /* :636 */
  /* barotropic loop :810-918 */
  for (int my_iif = 1; my_iif <= c->nfast + 1; my_iif++) {
    int next_indx1 = 3 - s->indx1;
    if (!s->predictor && my_iif <= c->nfast + 1) {
      s->predictor = 1;
      s->iif = my_iif;
      if (s->iif == 1) s->kstp = s->indx1;
      else s->kstp = 3 - s->indx1;
      s->knew = 3;
      s->krhs = s->indx1;
    }
    REV(o, orc_step2d(o, tile));
    if (s->predictor) {
      s->predictor = 0;
      s->knew = next_indx1;
      s->kstp = 3 - s->knew;
      s->krhs = 3;
      if (s->iif < c->nfast + 1) s->indx1 = next_indx1;
    }
    if (s->iif < c->nfast + 1) FWD(o, orc_step2d(o, tile));
  }
  REV(o, orc_set_depth(o, tile));                                       /* :963 */
  REV(o, orc_step3d_uv(o, tile));                                       /* :990 */
  FWD(o, orc_omega(o, tile));                                           /* :1017 */
  if (c->options & ORC_MY25_MIXING) { FWD(o, orc_my25_corstep(o, tile)); } /* :1019 */
  else if (c->options & ORC_GLS_MIXING) { FWD(o, orc_gls_corstep(o, tile)); } /* :1021 */
  REV(o, orc_step3d_t(o, tile));                                        /* :1045 */
  s->iic = s->iic + 1;                                                  /* :1145-1148 */
  s->time = s->time + c->dt;
  return 0;
}

void orc_set_threads(orc_t *o, int n) { o->nthreads = n < 1 ? 1 : n; }

void orc_get_diag(orc_t *o, double *out) { for (int k = 0; k < 16; k++) out[k] = o->diag[k]; }
