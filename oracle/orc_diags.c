/*
 * orc_diags.c -- per-term tracer tendencies, DIAGNOSTICS_TS: the DiaTwrk stores of pre_step3d.F:925-928,
 * t3dmix2_s.h:293-297 (t3dmix2_geo.h:409, t3dmix2_iso.h:428), step3d_t.F:908-912, :1357-1362, :1716-1719, :1892-1904
 * (made by orc_rhs3d.c, orc_t3dmix_geo.c, orc_step3d.c through orc_dia_wrk) and set_diags_tile, ROMS/Utility/set_diags.F:
 * 60-735, for the tracer terms.  TEST INFRASTRUCTURE (see orc.h).  PARITY STATUS: pinned bit for bit against the reference
 * built from ROMS/Include/upwelling.h AS SHIPPED (oracle/ref/build_ref.sh upwelling_diag: AVERAGES, DIAGNOSTICS_TS,
 * DIAGNOSTICS_UV; tests/test_oracle_vs_ref.py::test_set_diags_bitwise).  Tracers advected with MPDATA are not covered
 * (their Dhadv / Dvadv work arrays, step3d_t.F:881-895, :1254): orc_set_dia_window refuses them.
 *
 * Term order (mod_scalars.F:4246-4262): iThadv 1, iTxadv 2, iTyadv 3, iTvadv 4, [TS_DIF2: iThdif 5, iTxdif 6, iTydif 7,
 * [MIX_GEO_TS | MIX_ISO_TS: iTsdif 8,]] iTvdif, iTrate.  DiaTwrk(i,j,k,itrc,idiag) as the reference lays it out.
 */
#include <stdlib.h>
#include <string.h>
#include "orc.h"

typedef struct {
  int nDIA, ntsDIA, nrrec, ntstart, NDT;
  int idx[ORC_DIA_NTERMS];              /* 1-based reference index of each term, 0 = absent */
  double *wrk, *trc, *avgzeta;
  double diatime;
} dia_state;

int orc_dia_ndt(const orc_t *o) {
  int n = 6;
  if (o->c.options & ORC_TS_DIF2) { n += 3; if (o->c.options & (ORC_MIX_GEO_TS | ORC_MIX_ISO_TS)) n += 1; }
  return n;
}

int orc_set_dia_window(orc_t *o, int nDIA, int ntsDIA, int nrrec, int ntstart) {
  for (int it = 0; it < o->c.NT; it++)
    if (o->c.hadv[it] == ORC_MPDATA || o->c.vadv[it] == ORC_MPDATA) return 5;
  dia_state *s = (dia_state *)o->dia;
  if (!s) {
    s = (dia_state *)calloc(1, sizeof(dia_state));
    s->NDT = orc_dia_ndt(o);
    int ic = 4;
    s->idx[ORC_DIA_HADV] = 1; s->idx[ORC_DIA_XADV] = 2; s->idx[ORC_DIA_YADV] = 3; s->idx[ORC_DIA_VADV] = 4;
    if (o->c.options & ORC_TS_DIF2) {
      s->idx[ORC_DIA_HDIF] = ic + 1; s->idx[ORC_DIA_XDIF] = ic + 2; s->idx[ORC_DIA_YDIF] = ic + 3; ic += 3;
      if (o->c.options & (ORC_MIX_GEO_TS | ORC_MIX_ISO_TS)) { s->idx[ORC_DIA_SDIF] = ic + 1; ic += 1; }
    }
    s->idx[ORC_DIA_VDIF] = ic + 1; s->idx[ORC_DIA_RATE] = ic + 2;
    const size_t n = o->nij * (size_t)o->c.N * (size_t)o->c.NT * (size_t)s->NDT;
    s->wrk = (double *)calloc(n, sizeof(double));
    s->trc = (double *)calloc(n, sizeof(double));
    s->avgzeta = (double *)calloc(o->nij, sizeof(double));
    o->dia = s;
  }
  s->nDIA = nDIA; s->ntsDIA = ntsDIA; s->nrrec = nrrec; s->ntstart = ntstart;
  return 0;
}
void orc_dia_free(orc_t *o) {
  dia_state *s = (dia_state *)o->dia;
  orc_diauv_free(o);
  if (!s) return;
  free(s->wrk); free(s->trc); free(s->avgzeta); free(s);
  o->dia = NULL;
}
/* DiaTwrk(:,:,:,itrc,term): level 1 of the term's block, NULL when diagnostics are off or the term is absent */
double *orc_dia_wrk(orc_t *o, int term, int itrc) {
  dia_state *s = (dia_state *)o->dia;
  if (!s || !s->idx[term]) return NULL;
  return s->wrk + ((size_t)(itrc - 1) + (size_t)o->c.NT * (size_t)(s->idx[term] - 1)) * (size_t)o->c.N * o->nij;
}
double *orc_dia_field(orc_t *o, const char *name, long *nel) {
  dia_state *s = (dia_state *)o->dia;
  if (s) {
    const long n = (long)(o->nij * (size_t)o->c.N * (size_t)o->c.NT * (size_t)s->NDT);
    if (!strcmp(name, "DiaTwrk")) { if (nel) *nel = n; return s->wrk; }
    if (!strcmp(name, "DiaTrc")) { if (nel) *nel = n; return s->trc; }
    if (!strcmp(name, "dia_zeta")) { if (nel) *nel = (long)o->nij; return s->avgzeta; }
  }
  return orc_diauv_field(o, name, nel);
}
double orc_dia_time(const orc_t *o) { return o->dia ? ((const dia_state *)o->dia)->diatime : 0.0; }

/* set_diags_tile, set_diags.F:60-735: the tracer terms and avgzeta.  First step of a window: DiaTrc = DiaTwrk (:174-196);
   following steps: DiaTrc += DiaTwrk (:300-322); step closing the window: DiaTrc *= 1/nDIA, time stamp (:478-520) */
void orc_set_diags(orc_t *o, int tile) {
  dia_state *s = (dia_state *)o->dia;
  if (!s || s->nDIA == 0) return;
  ORC_LOCALS(o);
  const orc_bounds *b = &o->b[tile];
  const int iic = o->s.iic, nDIA = s->nDIA, ntsDIA = s->ntsDIA, kout = o->s.kstp, NT = o->c.NT;
  const int init = (iic > ntsDIA && (iic - 1) % nDIA == 1) || (iic >= ntsDIA && nDIA == 1) || (s->nrrec > 0 && iic == s->ntstart);
  const int accum = !init && iic > ntsDIA;
  const int convert = (iic > ntsDIA && (iic - 1) % nDIA == 0 && (iic != s->ntstart || s->nrrec == 0)) || (iic >= ntsDIA && nDIA == 1);
  const size_t blk = (size_t)N * nij;
  orc_set_diags_uv(o, tile, init, accum, convert, 1.0 / (double)nDIA);       /* the momentum terms (orc_diags_uv.c) */
  if (init || accum) {
    for (int j = b->JstrR; j <= b->JendR; j++)
      for (int i = b->IstrR; i <= b->IendR; i++)
        s->avgzeta[X2(i, j)] = init ? o->zeta[X2T(i, j, kout)] : s->avgzeta[X2(i, j)] + o->zeta[X2T(i, j, kout)];
    for (int id = 0; id < s->NDT; id++)
      for (int it = 0; it < NT; it++) {
        double *T = s->trc + ((size_t)it + (size_t)NT * (size_t)id) * blk;
        const double *Wk = s->wrk + ((size_t)it + (size_t)NT * (size_t)id) * blk;
        for (int k = 1; k <= N; k++)
          for (int j = b->JstrR; j <= b->JendR; j++)
            for (int i = b->IstrR; i <= b->IendR; i++)
              T[X3(i, j, k)] = init ? Wk[X3(i, j, k)] : T[X3(i, j, k)] + Wk[X3(i, j, k)];
      }
  }
  if (convert) {
    const double fac = 1.0 / (double)nDIA;
    if (tile == 0) s->diatime = nDIA == 1 ? o->s.time : s->diatime + (double)nDIA * o->c.dt;
    for (int j = b->JstrR; j <= b->JendR; j++)
      for (int i = b->IstrR; i <= b->IendR; i++) s->avgzeta[X2(i, j)] = fac * s->avgzeta[X2(i, j)];
    for (int id = 0; id < s->NDT; id++)
      for (int it = 0; it < NT; it++) {
        double *T = s->trc + ((size_t)it + (size_t)NT * (size_t)id) * blk;
        for (int k = 1; k <= N; k++)
          for (int j = b->JstrR; j <= b->JendR; j++)
            for (int i = b->IstrR; i <= b->IendR; i++) T[X3(i, j, k)] = fac * T[X3(i, j, k)];
      }
    /* "Apply periodic or gradient boundary conditions for output purposes" :576-615: exchange_r2d of avgzeta, bc_r3d_tile
       (zero gradient at closed edges, corner means, periodic exchange: bc_3d.F:41) of every term */
    orc_exchange2d(o, b, 'r', s->avgzeta);
    for (int id = 0; id < s->NDT; id++)
      for (int it = 0; it < NT; it++) orc_bc_w3d(o, b, s->trc + ((size_t)it + (size_t)NT * (size_t)id) * blk, N);
  }
}
