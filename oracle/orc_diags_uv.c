/*
 * orc_diags_uv.c -- per-term momentum tendencies, DIAGNOSTICS_UV: the arrays of mod_diags.F:174-222, the term indices of
 * mod_scalars.F:4264-4377 and the momentum part of set_diags_tile (ROMS/Utility/set_diags.F:192-235, :319-360, :541-572,
 * :617-650).  The terms themselves are stored where the reference stores them: prsgrd*.h (M3pgrd), rhs3d.F (Coriolis,
 * advection, the vertical sums DiaRUfrc), uv3dmix2_s.h (viscosity), pre_step3d.F:979-1137 (DiaU3wrk from the two levels of
 * DiaRU), step2d_LF_AM3.h (the 2-D right-hand-side terms, their coupling with DiaRUfrc and their fast-time integration),
 * step3d_uv.F (the corrector, the implicit vertical viscosity, the coupling of the 2-D and 3-D terms) -- hooks in
 * orc_rhs3d.c, orc_step2d.c, orc_step3d.c behind `o->duv`.  TEST INFRASTRUCTURE (see orc.h).  PARITY STATUS: pinned bit for
 * bit against the reference built from ROMS/Include/upwelling.h AS SHIPPED (oracle/ref/build_ref.sh upwelling_diag;
 * tests/test_oracle_vs_ref.py::test_set_diags_uv_bitwise): UV_COR, UV_ADV, UV_VIS2 + MIX_S_UV, linear drag, no masks, no
 * curvilinear terms.  The CURVGRID and MASKING branches follow the reference's text and have no reference build behind
 * them; WEC_VF, BODYFORCE, UV_VIS4 and the geopotential viscosity are not covered.
 */
#include <stdlib.h>
#include <string.h>
#include "orc.h"

int orc_set_diauv(orc_t *o) {
  if (o->duv) return 0;
  orc_diauv *d = (orc_diauv *)calloc(1, sizeof(orc_diauv));
  const int cor = (o->c.options & ORC_UV_COR) != 0, adv = (o->c.options & ORC_UV_ADV) != 0, vis = (o->c.options & ORC_UV_VIS2) != 0;
  int ic = 0;                                                  /* mod_scalars.F:4272-4322 */
  if (cor) { d->M2fcor = ic + 1; ic += 1; }
  if (adv) { d->M2hadv = ic + 1; d->M2xadv = ic + 2; d->M2yadv = ic + 3; ic += 3; }
  if (vis) { d->M2hvis = ic + 1; d->M2xvis = ic + 2; d->M2yvis = ic + 3; ic += 3; }
  d->M2pgrd = ic + 1; d->M2sstr = ic + 2; d->M2bstr = ic + 3;
  d->NDM2d = 4 + (adv ? 3 : 0) + (cor ? 1 : 0) + (vis ? 3 : 0);   /* mod_param.F:1559-1584 */
  d->M2rate = d->NDM2d;
  ic = 0;                                                      /* mod_scalars.F:4331-4375 */
  if (cor) { d->M3fcor = ic + 1; ic += 1; }
  if (adv) { d->M3vadv = ic + 1; d->M3hadv = ic + 2; d->M3xadv = ic + 3; d->M3yadv = ic + 4; ic += 4; }
  d->M3pgrd = ic + 1; d->M3vvis = ic + 2;
  if (vis) { d->M3hvis = ic + 3; d->M3xvis = ic + 4; d->M3yvis = ic + 5; }
  d->NDM3d = 3 + (adv ? 4 : 0) + (cor ? 1 : 0) + (vis ? 3 : 0);   /* mod_param.F:1589-1603 */
  d->NDrhs = 1 + (adv ? 4 : 0) + (cor ? 1 : 0);
  d->M3rate = d->NDM3d;
  const size_t nij = o->nij, N = (size_t)o->c.N;
#define A_(n) (double *)calloc((n), sizeof(double))
  d->U2wrk = A_(nij * d->NDM2d); d->V2wrk = A_(nij * d->NDM2d);
  d->RUbar = A_(nij * 2 * (d->NDM2d - 1)); d->RVbar = A_(nij * 2 * (d->NDM2d - 1));
  d->U2int = A_(nij * d->NDM2d); d->V2int = A_(nij * d->NDM2d);
  d->RUfrc = A_(nij * 3 * (d->NDM2d - 1)); d->RVfrc = A_(nij * 3 * (d->NDM2d - 1));
  d->U3wrk = A_(nij * N * d->NDM3d); d->V3wrk = A_(nij * N * d->NDM3d);
  d->RU = A_(nij * N * 2 * d->NDrhs); d->RV = A_(nij * N * 2 * d->NDrhs);
  d->U2d = A_(nij * d->NDM2d); d->V2d = A_(nij * d->NDM2d);
  d->U3d = A_(nij * N * d->NDM3d); d->V3d = A_(nij * N * d->NDM3d);
#undef A_
  o->duv = d;
  return 0;
}
void orc_diauv_free(orc_t *o) {
  orc_diauv *d = o->duv;
  if (!d) return;
  double *all[] = {d->U2wrk, d->V2wrk, d->RUbar, d->RVbar, d->U2int, d->V2int, d->RUfrc, d->RVfrc, d->U3wrk, d->V3wrk, d->RU, d->RV,
                   d->U2d, d->V2d, d->U3d, d->V3d};
  for (size_t k = 0; k < sizeof(all) / sizeof(all[0]); k++) free(all[k]);
  free(d);
  o->duv = NULL;
}
double *orc_diauv_field(orc_t *o, const char *name, long *nel) {
  orc_diauv *d = o->duv;
  if (d) {
    const long nij = (long)o->nij, N = o->c.N;
    const struct { const char *n; double *p; long len; } T[] = {
        {"DiaU2wrk", d->U2wrk, nij * d->NDM2d}, {"DiaV2wrk", d->V2wrk, nij * d->NDM2d},
        {"DiaRUbar", d->RUbar, nij * 2 * (d->NDM2d - 1)}, {"DiaRVbar", d->RVbar, nij * 2 * (d->NDM2d - 1)},
        {"DiaU2int", d->U2int, nij * d->NDM2d}, {"DiaV2int", d->V2int, nij * d->NDM2d},
        {"DiaRUfrc", d->RUfrc, nij * 3 * (d->NDM2d - 1)}, {"DiaRVfrc", d->RVfrc, nij * 3 * (d->NDM2d - 1)},
        {"DiaU3wrk", d->U3wrk, nij * N * d->NDM3d}, {"DiaV3wrk", d->V3wrk, nij * N * d->NDM3d},
        {"DiaRU", d->RU, nij * N * 2 * d->NDrhs}, {"DiaRV", d->RV, nij * N * 2 * d->NDrhs},
        {"DiaU2d", d->U2d, nij * d->NDM2d}, {"DiaV2d", d->V2d, nij * d->NDM2d},
        {"DiaU3d", d->U3d, nij * N * d->NDM3d}, {"DiaV3d", d->V3d, nij * N * d->NDM3d}};
    for (size_t k = 0; k < sizeof(T) / sizeof(T[0]); k++)
      if (!strcmp(name, T[k].n)) { if (nel) *nel = T[k].len; return T[k].p; }
  }
  if (nel) *nel = -1;
  return NULL;
}

/* the momentum part of set_diags_tile: first step of a window DiaU2d = DiaU2wrk ... (:192-235); then += (:319-360); the
   closing step scales by 1/nDIA (:541-572) and fills the boundary / ghost points of every term (:617-650) */
void orc_set_diags_uv(orc_t *o, int tile, int init, int accum, int convert, double fac) {
  orc_diauv *d = o->duv;
  if (!d) return;
  ORC_LOCALS(o);
  const orc_bounds *b = &o->b[tile];
  if (init || accum) {
    for (int id = 1; id <= d->NDM2d; id++) {
      for (int j = b->JstrR; j <= b->JendR; j++)
        for (int i = b->Istr; i <= b->IendR; i++)
          DU2(d->U2d, i, j, id) = init ? DU2(d->U2wrk, i, j, id) : DU2(d->U2d, i, j, id) + DU2(d->U2wrk, i, j, id);
      for (int j = b->Jstr; j <= b->JendR; j++)
        for (int i = b->IstrR; i <= b->IendR; i++)
          DU2(d->V2d, i, j, id) = init ? DU2(d->V2wrk, i, j, id) : DU2(d->V2d, i, j, id) + DU2(d->V2wrk, i, j, id);
    }
    for (int id = 1; id <= d->NDM3d; id++)
      for (int k = 1; k <= N; k++) {
        for (int j = b->JstrR; j <= b->JendR; j++)
          for (int i = b->Istr; i <= b->IendR; i++)
            DU3(d->U3d, i, j, k, id) = init ? DU3(d->U3wrk, i, j, k, id) : DU3(d->U3d, i, j, k, id) + DU3(d->U3wrk, i, j, k, id);
        for (int j = b->Jstr; j <= b->JendR; j++)
          for (int i = b->IstrR; i <= b->IendR; i++)
            DU3(d->V3d, i, j, k, id) = init ? DU3(d->V3wrk, i, j, k, id) : DU3(d->V3d, i, j, k, id) + DU3(d->V3wrk, i, j, k, id);
      }
  }
  if (convert) {
    for (int id = 1; id <= d->NDM2d; id++) {
      for (int j = b->JstrR; j <= b->JendR; j++)
        for (int i = b->Istr; i <= b->IendR; i++) DU2(d->U2d, i, j, id) = fac * DU2(d->U2d, i, j, id);
      for (int j = b->Jstr; j <= b->JendR; j++)
        for (int i = b->IstrR; i <= b->IendR; i++) DU2(d->V2d, i, j, id) = fac * DU2(d->V2d, i, j, id);
    }
    for (int id = 1; id <= d->NDM3d; id++)
      for (int k = 1; k <= N; k++) {
        for (int j = b->JstrR; j <= b->JendR; j++)
          for (int i = b->Istr; i <= b->IendR; i++) DU3(d->U3d, i, j, k, id) = fac * DU3(d->U3d, i, j, k, id);
        for (int j = b->Jstr; j <= b->JendR; j++)
          for (int i = b->IstrR; i <= b->IendR; i++) DU3(d->V3d, i, j, k, id) = fac * DU3(d->V3d, i, j, k, id);
      }
    for (int id = 1; id <= d->NDM2d; id++) {
      orc_bc_u2d(o, b, d->U2d + (size_t)(id - 1) * nij);
      orc_bc_v2d(o, b, d->V2d + (size_t)(id - 1) * nij);
    }
    for (int id = 1; id <= d->NDM3d; id++) {
      orc_bc_u3d(o, b, d->U3d + (size_t)(id - 1) * (size_t)N * nij, N);
      orc_bc_v3d(o, b, d->V3d + (size_t)(id - 1) * (size_t)N * nij, N);
    }
  }
}
