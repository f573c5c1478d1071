/*
 * orc_avg.c -- time-averaged fields: set_avg_tile, ROMS/Nonlinear/set_avg.F:96-5210, for the fields the Aout
 * switches of ROMS/External/roms_upwelling.in ask for (zeta, ubar, vbar, u, v, omega, w, rho, the tracers, the
 * volume fluxes Huon/Hvom, and the quadratic terms zeta2, ubar2, vbar2, uu, vv, uv, <t*t>, <u*t>, <v*t>,
 * <Huon*t>, <Hvom*t>).  TEST INFRASTRUCTURE (see orc.h).  PARITY STATUS: pinned bit for bit against the
 * reference's set_avg.F compiled with the application header oracle/ref/upwelling_avg.h
 * (tests/test_oracle_vs_ref.py::test_set_avg_bitwise).
 *
 * Three phases per call, as in the reference: at the first step of an averaging window the arrays are SET to
 * the current fields (:251-1601), on the following steps the fields are ADDED (:1606-2954), and at the step that
 * closes the window the sums are multiplied by 1/nAVG and their periodic ghost points refilled (:2962-5210).
 * Fields are taken at the output time levels KOUT = kstp, NOUT = nrhs (globaldefs.h:500-516).
 */
#include <stdlib.h>
#include <string.h>
#include "orc.h"
#define MAX(a, b) ((a) > (b) ? (a) : (b))
#define MIN(a, b) ((a) < (b) ? (a) : (b))

enum { A_ZETA, A_UBAR, A_VBAR, A_U, A_V, A_OMEGA, A_W, A_RHO, A_T, A_ZZ, A_U2, A_V2, A_UU, A_VV, A_UV, A_HUON,
       A_HVOM, A_TT, A_UT, A_VT, A_HUT, A_HVT, A_NFIELDS };

typedef struct { const char *name; char grid; int k0, perT; int rng; } adesc;
/* rng: 0 (IstrR:IendR,JstrR:JendR)  1 (Istr:IendR,JstrR:JendR)  2 (IstrR:IendR,Jstr:JendR)
        3 (Istr:Iend,Jstr:Jend)      4 (Istr:Iend,JstrR:JendR)   5 (IstrR:IendR,Jstr:Jend);  k0 < 0: 2-D */
static const adesc AD[A_NFIELDS] = {
  {"avg_zeta", 'r', -1, 0, 0}, {"avg_ubar", 'u', -1, 0, 1}, {"avg_vbar", 'v', -1, 0, 2},
  {"avg_u", 'u', 1, 0, 1}, {"avg_v", 'v', 1, 0, 2}, {"avg_omega", 'r', 0, 0, 0}, {"avg_w", 'r', 0, 0, 0},
  {"avg_rho", 'r', 1, 0, 0}, {"avg_t", 'r', 1, 1, 0},
  {"avg_ZZ", 'r', -1, 0, 0}, {"avg_U2", 'u', -1, 0, 1}, {"avg_V2", 'v', -1, 0, 2},
  {"avg_UU", 'u', 1, 0, 1}, {"avg_VV", 'v', 1, 0, 2}, {"avg_UV", 'r', 1, 0, 3},
  {"avg_Huon", 'u', 1, 0, 1}, {"avg_Hvom", 'v', 1, 0, 2},
  {"avg_TT", 'r', 1, 1, 0}, {"avg_UT", 'u', 1, 1, 4}, {"avg_VT", 'v', 1, 1, 5},
  {"avg_HuonT", 'u', 1, 1, 4}, {"avg_HvomT", 'v', 1, 1, 5},
};

/* cnt: WET_DRY -- the wet-point counters GRID%rmask_avg, umask_avg, vmask_avg of set_avg.F:257-288, :1608-1645 (pmask_avg serves
   the vorticity fields only, which are not among the 22) */
typedef struct { int nAVG, ntsAVG, nrrec, ntstart; double *a[A_NFIELDS]; double avgtime; double *cnt[3]; } avg_state;

static size_t planes(const orc_t *o, int f) {
  const size_t N = (size_t)o->c.N;
  const size_t nk = AD[f].k0 < 0 ? 1 : (AD[f].k0 == 0 ? N + 1 : N);
  return nk * (AD[f].perT ? (size_t)o->c.NT : 1);
}

/* mod_average.F: allocate_average + the window parameters of mod_scalars (nAVG, ntsAVG, nrrec, ntstart) */
void orc_set_avg_window(orc_t *o, int nAVG, int ntsAVG, int nrrec, int ntstart) {
  avg_state *s = (avg_state *)o->avg;
  if (!s) {
    s = (avg_state *)calloc(1, sizeof(avg_state));
    for (int f = 0; f < A_NFIELDS; f++) s->a[f] = (double *)calloc(planes(o, f) * o->nij, sizeof(double));
    for (int m = 0; m < 3; m++) s->cnt[m] = (double *)calloc(o->nij, sizeof(double));
    o->avg = s;
  }
  s->nAVG = nAVG; s->ntsAVG = ntsAVG; s->nrrec = nrrec; s->ntstart = ntstart;
}
void orc_avg_free(orc_t *o) {
  avg_state *s = (avg_state *)o->avg;
  if (!s) return;
  for (int f = 0; f < A_NFIELDS; f++) free(s->a[f]);
  for (int m = 0; m < 3; m++) free(s->cnt[m]);
  free(s);
  o->avg = NULL;
}
double *orc_avg_field(orc_t *o, const char *name, long *nel) {
  avg_state *s = (avg_state *)o->avg;
  if (s)
    for (int f = 0; f < A_NFIELDS; f++)
      if (!strcmp(name, AD[f].name)) { if (nel) *nel = (long)(planes(o, f) * o->nij); return s->a[f]; }
  if (nel) *nel = -1;
  return NULL;
}
double orc_avg_time(const orc_t *o) { return o->avg ? ((const avg_state *)o->avg)->avgtime : 0.0; }

static void range(const orc_bounds *b, int rng, int *i0, int *i1, int *j0, int *j1) {
  *i0 = (rng == 1 || rng == 3 || rng == 4) ? b->Istr : b->IstrR;
  *i1 = (rng == 3 || rng == 4) ? b->Iend : b->IendR;
  *j0 = (rng == 2 || rng == 3 || rng == 5) ? b->Jstr : b->JstrR;
  *j1 = (rng == 3 || rng == 5) ? b->Jend : b->JendR;
}

/* the current value of field f at (i,j,k[,it]): the right-hand sides of set_avg.F:296-1601 */
static double value(const orc_t *o, int f, int i, int j, int k, int it) {
  ORC_LOCALS(o);
  const int Kout = o->s.kstp, Nout = o->s.nrhs;
  switch (f) {
    case A_ZETA: return o->zeta[X2T(i, j, Kout)];
    case A_UBAR: return o->ubar[X2T(i, j, Kout)];
    case A_VBAR: return o->vbar[X2T(i, j, Kout)];
    case A_U: return o->u[X4(i, j, k, Nout)];
    case A_V: return o->v[X4(i, j, k, Nout)];
    case A_OMEGA: return o->W[XW(i, j, k)] * o->pm[X2(i, j)] * o->pn[X2(i, j)];
    case A_W: return o->wvel[XW(i, j, k)];
    case A_RHO: return o->rho[X3(i, j, k)];
    case A_T: return o->t[XT(i, j, k, Nout, it)];
    case A_ZZ: return o->zeta[X2T(i, j, Kout)] * o->zeta[X2T(i, j, Kout)];
    case A_U2: return o->ubar[X2T(i, j, Kout)] * o->ubar[X2T(i, j, Kout)];
    case A_V2: return o->vbar[X2T(i, j, Kout)] * o->vbar[X2T(i, j, Kout)];
    case A_UU: return o->u[X4(i, j, k, Nout)] * o->u[X4(i, j, k, Nout)];
    case A_VV: return o->v[X4(i, j, k, Nout)] * o->v[X4(i, j, k, Nout)];
    case A_UV: return 0.25 * (o->u[X4(i, j, k, Nout)] + o->u[X4(i + 1, j, k, Nout)]) *
                      (o->v[X4(i, j, k, Nout)] + o->v[X4(i, j + 1, k, Nout)]);
    case A_HUON: return o->Huon[X3(i, j, k)];
    case A_HVOM: return o->Hvom[X3(i, j, k)];
    case A_TT: return o->t[XT(i, j, k, Nout, it)] * o->t[XT(i, j, k, Nout, it)];
    case A_UT: return 0.5 * o->u[X4(i, j, k, Nout)] * (o->t[XT(i - 1, j, k, Nout, it)] + o->t[XT(i, j, k, Nout, it)]);
    case A_VT: return 0.5 * o->v[X4(i, j, k, Nout)] * (o->t[XT(i, j - 1, k, Nout, it)] + o->t[XT(i, j, k, Nout, it)]);
    case A_HUT: return 0.5 * o->Huon[X3(i, j, k)] * (o->t[XT(i - 1, j, k, Nout, it)] + o->t[XT(i, j, k, Nout, it)]);
    case A_HVT: return 0.5 * o->Hvom[X3(i, j, k)] * (o->t[XT(i, j - 1, k, Nout, it)] + o->t[XT(i, j, k, Nout, it)]);
  }
  return 0.0;
}

void orc_set_avg(orc_t *o, int tile) {
  avg_state *s = (avg_state *)o->avg;
  if (!s || s->nAVG == 0) return;                                  /* :204 */
  ORC_LOCALS(o);
  const orc_bounds *b = &o->b[tile];
  const int iic = o->s.iic, nAVG = s->nAVG, ntsAVG = s->ntsAVG;
  const int init = ((iic > ntsAVG) && ((iic - 1) % nAVG == 1)) || ((iic >= ntsAVG) && nAVG == 1) ||
                   (s->nrrec > 0 && iic == s->ntstart);            /* :251-254 */
  const int accum = !init && iic > ntsAVG;                         /* :1606 */
  const int convert = ((iic > ntsAVG) && ((iic - 1) % nAVG == 0) && (iic != s->ntstart || s->nrrec == 0)) ||
                      ((iic >= ntsAVG) && nAVG == 1);              /* :2962-2965 */
  /* WET_DRY: every field times the full mask (land x wet) of its grid type where it is set :302, :403 ... and added :1652 ...; the
     sums are divided by the number of steps the point was wet, :2980-2988 */
  const int wet = o->wet_dry;
  const double *mfull[3] = { o->rmask_full, o->umask_full, o->vmask_full };
  if (wet && (init || accum))
    for (int m = 0; m < 3; m++) {                                  /* :257-288 | :1608-1645 */
      int i0, i1, j0, j1;
      range(b, m, &i0, &i1, &j0, &j1);                             /* rng 0 (rho), 1 (u: Istr:IendR), 2 (v: Jstr:JendR) */
      for (int j = j0; j <= j1; j++)
        for (int i = i0; i <= i1; i++) {
          const double c1 = MAX(0.0, MIN(mfull[m][X2(i, j)], 1.0));
          s->cnt[m][X2(i, j)] = init ? c1 : s->cnt[m][X2(i, j)] + c1;
        }
    }
  if (init || accum)
    for (int f = 0; f < A_NFIELDS; f++) {
      int i0, i1, j0, j1;
      range(b, AD[f].rng, &i0, &i1, &j0, &j1);
      const int ka = AD[f].k0 < 0 ? 1 : AD[f].k0, kb = AD[f].k0 < 0 ? 1 : N;
      const size_t np = AD[f].k0 < 0 ? 1 : (AD[f].k0 == 0 ? (size_t)N + 1 : (size_t)N);
      for (int it = 1; it <= (AD[f].perT ? o->c.NT : 1); it++)
        for (int k = ka; k <= kb; k++)
          for (int j = j0; j <= j1; j++)
            for (int i = i0; i <= i1; i++) {
              double *d = &s->a[f][X2(i, j) + (size_t)(k - ka) * nij + (size_t)(it - 1) * np * nij];
              double v = value(o, f, i, j, k, it);
              if (wet) v = v * mfull[AD[f].grid == 'u' ? 1 : AD[f].grid == 'v' ? 2 : 0][X2(i, j)];
              *d = init ? v : *d + v;
            }
    }
  if (convert) {
    if (tile == 0) s->avgtime = nAVG == 1 ? o->s.time : s->avgtime + (double)nAVG * o->c.dt;   /* :2966-2972 */
    const double fac = 1.0 / (double)nAVG;
    for (int f = 0; f < A_NFIELDS; f++) {
      int i0, i1, j0, j1;
      range(b, AD[f].rng, &i0, &i1, &j0, &j1);
      const size_t np = AD[f].k0 < 0 ? 1 : (AD[f].k0 == 0 ? (size_t)N + 1 : (size_t)N);
      const size_t ntr = AD[f].perT ? (size_t)o->c.NT : 1;
      for (size_t p = 0; p < np * ntr; p++)
        for (int j = j0; j <= j1; j++)
          for (int i = i0; i <= i1; i++) {
            const double fc = wet ? 1.0 / MAX(1.0, s->cnt[AD[f].grid == 'u' ? 1 : AD[f].grid == 'v' ? 2 : 0][X2(i, j)]) : fac;
            s->a[f][X2(i, j) + p * nij] = fc * s->a[f][X2(i, j) + p * nij];
          }
      if (o->c.EWperiodic || o->c.NSperiodic)
        for (size_t p = 0; p < np * ntr; p++) orc_exchange2d(o, b, AD[f].grid, s->a[f] + p * nij);
    }
  }
}
