/*
 * orc_core.c -- oracle state, tiling, periodic copies and boundary fills.
 * TEST INFRASTRUCTURE (see orc.h).
 *
 * Follows:
 *   ROMS/Utility/get_bounds.F   tile_bounds_2d :972-1042, get_bounds :32-287,
 *                               get_domain_edges :460-652, var_bounds :1044-1884
 *   ROMS/Nonlinear/exchange_2d.F exchange_{p,r,u,v}2d_tile :63-807
 *   ROMS/Nonlinear/exchange_3d.F exchange_{r,u,v,w}3d_tile :280-1126
 *   ROMS/Nonlinear/bc_2d.F      bc_{r,u,v}2d_tile :41-516
 *   ROMS/Nonlinear/bc_3d.F      bc_w3d_tile :588-723
 *   ROMS/Nonlinear/zetabc.F :60-650 (closed/periodic branches), u2dbc_im.F,
 *   v2dbc_im.F, t3dbc_im.F, u3dbc_im.F, v3dbc_im.F (closed/periodic branches)
 * PARITY: pinned (all of these reference routines build in oracle/_ref).
 */
#include "orc.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MAX(a, b) ((a) > (b) ? (a) : (b))
#define MIN(a, b) ((a) < (b) ? (a) : (b))

/* ------------------------------------------------------------------ tiling */

void orc_tile_bounds(const orc_cfg *c, int tile, orc_bounds *b) {
  const int Lm = c->Lm, Mm = c->Mm;
  /* tile_bounds_2d */
  int ChunkSizeI = (Lm + c->NtileI - 1) / c->NtileI;
  int ChunkSizeJ = (Mm + c->NtileJ - 1) / c->NtileJ;
  int MarginI = (c->NtileI * ChunkSizeI - Lm) / 2;
  int MarginJ = (c->NtileJ * ChunkSizeJ - Mm) / 2;
  int Jtile = tile / c->NtileI;
  int Itile = tile - Jtile * c->NtileI;
  int my_Istr = 1 + Itile * ChunkSizeI - MarginI;
  int my_Iend = my_Istr + ChunkSizeI - 1;
  my_Istr = MAX(my_Istr, 1);
  my_Iend = MIN(my_Iend, Lm);
  int my_Jstr = 1 + Jtile * ChunkSizeJ - MarginJ;
  int my_Jend = my_Jstr + ChunkSizeJ - 1;
  my_Jstr = MAX(my_Jstr, 1);
  my_Jend = MIN(my_Jend, Mm);
  /* get_domain_edges */
  b->west = Itile == 0;
  b->east = Itile == c->NtileI - 1;
  b->south = Jtile == 0;
  b->north = Jtile == c->NtileJ - 1;
  b->sw = b->west && b->south;
  b->se = b->east && b->south;
  b->nw = b->west && b->north;
  b->ne = b->east && b->north;
  /* var_bounds: a physical (non-periodic) edge changes the ranges */
  const int pw = b->west && !c->EWperiodic, pe = b->east && !c->EWperiodic;
  const int ps = b->south && !c->NSperiodic, pn = b->north && !c->NSperiodic;
  b->Istr = my_Istr;
  b->IstrP = my_Istr;
  b->IstrR = pw ? my_Istr - 1 : my_Istr;
  b->IstrT = b->IstrR;
  b->IstrU = pw ? my_Istr + 1 : my_Istr;
  b->IstrB = pw ? b->IstrT + 1 : my_Istr;
  b->IstrM = pw ? b->IstrP + 1 : b->IstrU;
  b->Istrm3 = pw ? MAX(0, my_Istr - 3) : my_Istr - 3;
  b->Istrm2 = pw ? MAX(0, my_Istr - 2) : my_Istr - 2;
  b->Istrm1 = pw ? MAX(1, my_Istr - 1) : my_Istr - 1;
  b->IstrUm2 = pw ? MAX(1, b->IstrU - 2) : b->IstrU - 2;
  b->IstrUm1 = pw ? MAX(2, b->IstrU - 1) : b->IstrU - 1;
  b->Iend = my_Iend;
  b->IendR = pe ? my_Iend + 1 : my_Iend;
  b->IendP = b->IendR;
  b->IendT = b->IendR;
  b->IendB = pe ? b->IendT - 1 : my_Iend;
  b->Iendp1 = pe ? MIN(my_Iend + 1, Lm) : my_Iend + 1;
  b->Iendp2i = pe ? MIN(my_Iend + 2, Lm) : my_Iend + 2;
  b->Iendp2 = pe ? MIN(my_Iend + 2, Lm + 1) : my_Iend + 2;
  b->Iendp3 = pe ? MIN(my_Iend + 3, Lm + 1) : my_Iend + 3;
  b->Jstr = my_Jstr;
  b->JstrP = my_Jstr;
  b->JstrR = ps ? my_Jstr - 1 : my_Jstr;
  b->JstrT = b->JstrR;
  b->JstrV = ps ? my_Jstr + 1 : my_Jstr;
  b->JstrB = ps ? b->JstrT + 1 : my_Jstr;
  b->JstrM = ps ? b->JstrP + 1 : b->JstrV;
  b->Jstrm3 = ps ? MAX(0, my_Jstr - 3) : my_Jstr - 3;
  b->Jstrm2 = ps ? MAX(0, my_Jstr - 2) : my_Jstr - 2;
  b->Jstrm1 = ps ? MAX(1, my_Jstr - 1) : my_Jstr - 1;
  b->JstrVm2 = ps ? MAX(1, b->JstrV - 2) : b->JstrV - 2;
  b->JstrVm1 = ps ? MAX(2, b->JstrV - 1) : b->JstrV - 1;
  b->Jend = my_Jend;
  b->JendR = pn ? my_Jend + 1 : my_Jend;
  b->JendP = b->JendR;
  b->JendT = b->JendR;
  b->JendB = pn ? b->JendT - 1 : my_Jend;
  b->Jendp1 = pn ? MIN(my_Jend + 1, Mm) : my_Jend + 1;
  b->Jendp2i = pn ? MIN(my_Jend + 2, Mm) : my_Jend + 2;
  b->Jendp2 = pn ? MIN(my_Jend + 2, Mm + 1) : my_Jend + 2;
  b->Jendp3 = pn ? MIN(my_Jend + 3, Mm + 1) : my_Jend + 3;
}

/* get_bounds (serial, non-DISTRIBUTE): allocation bounds of the global arrays;
   Im = Lm + padding, mod_param.F:1633-1636 */
static void alloc_bounds(orc_cfg *c) {
  int Im = c->Lm + ((c->Lm + 2) / 2 - (c->Lm + 1) / 2);
  int Jm = c->Mm + ((c->Mm + 2) / 2 - (c->Mm + 1) / 2);
  if (c->EWperiodic) { c->LBi = -c->Nghost; c->UBi = Im + c->Nghost; }
  else { c->LBi = 0; c->UBi = Im + 1; }
  if (c->NSperiodic) { c->LBj = -c->Nghost; c->UBj = Jm + c->Nghost; }
  else { c->LBj = 0; c->UBj = Jm + 1; }
}

/* ------------------------------------------------------------------- state */

void orc_set_clima(orc_t *o, int flags) { o->clima_flags = flags; }
void orc_set_prsgrd(orc_t *o, int scheme) { o->prs_scheme = scheme; }
void orc_set_ddmix(orc_t *o, int on) { o->ddmix = on != 0; }
void orc_set_bkpp(orc_t *o, int on) { o->bkpp = on != 0; }

static double *dalloc(size_t n) { return (double *)calloc(n ? n : 1, sizeof(double)); }

typedef struct { const char *name; size_t off; int kind; } fdesc;
/* kind: number of 2-D planes as a function of N, NT: see field_planes() */
enum { K2 = 0, KR, KW, KWx3, K2x3, K2x2, KRx2, KTR, KWx2, K2xNT, KWxNAT, KRxNT, KTAB_R, KTAB_W,
       KBJ, KBI, KBJN, KBIN, KBJT, KBIT };   /* boundary data: (LBj:UBj) / (LBi:UBi) [, N [, NT]] */
#define FD(nm, kind) { #nm, offsetof(orc_t, nm), kind }
static const fdesc fields[] = {
  FD(h, K2), FD(f, K2), FD(fomn, K2), FD(pm, K2), FD(pn, K2), FD(om_r, K2), FD(on_r, K2),
  FD(om_u, K2), FD(on_u, K2), FD(om_v, K2), FD(on_v, K2), FD(om_p, K2), FD(on_p, K2),
  FD(omn, K2), FD(pmon_r, K2), FD(pnom_r, K2), FD(pmon_p, K2), FD(pnom_p, K2),
  FD(pmon_u, K2), FD(pnom_u, K2), FD(pmon_v, K2), FD(pnom_v, K2), FD(dmde, K2), FD(dndx, K2),
  FD(angler, K2), FD(xr, K2), FD(yr, K2), FD(xp, K2), FD(yp, K2), FD(lonr, K2), FD(latr, K2), FD(rdrag, K2),
  FD(rdrag2, K2), FD(rmask, K2), FD(umask, K2), FD(vmask, K2), FD(pmask, K2),
  FD(rmask_wet, K2), FD(umask_wet, K2), FD(vmask_wet, K2), FD(pmask_wet, K2), FD(rmask_full, K2), FD(umask_full, K2),
  FD(vmask_full, K2), FD(pmask_full, K2), FD(rmask_wet_avg, K2),
  FD(Hz, KR), FD(z_r, KR), FD(z_w, KW), FD(Huon, KR), FD(Hvom, KR),
  FD(zeta, K2x3), FD(ubar, K2x3), FD(vbar, K2x3), FD(rzeta, K2x2), FD(rubar, K2x2),
  FD(rvbar, K2x2), FD(u, KRx2), FD(v, KRx2), FD(t, KTR), FD(W, KW), FD(wvel, KW),
  FD(rho, KR), FD(pden, KR), FD(ru, KWx2), FD(rv, KWx2),
  FD(rhoA, K2), FD(rhoS, K2), FD(rufrc, K2), FD(rvfrc, K2), FD(Zt_avg1, K2),
  FD(DU_avg1, K2), FD(DU_avg2, K2), FD(DV_avg1, K2), FD(DV_avg2, K2),
  FD(sustr, K2), FD(svstr, K2), FD(bustr, K2), FD(bvstr, K2), FD(stflx, K2xNT),
  FD(btflx, K2xNT), FD(stflux, K2xNT), FD(btflux, K2xNT), FD(srflx, K2),
  FD(Uwind, K2), FD(Vwind, K2), FD(Tair, K2), FD(Pair, K2), FD(Hair, K2), FD(rain, K2),
  FD(cloud, K2), FD(lhflx, K2), FD(shflx, K2), FD(lrflx, K2), FD(evap, K2),
  FD(Akv, KW), FD(Akt, KWxNAT), FD(visc2_r, K2), FD(visc2_p, K2), FD(diff2, K2xNT), FD(visc4_r, K2), FD(visc4_p, K2), FD(diff4, K2xNT),
  FD(tclm, KRxNT), FD(Tnudgcof, KRxNT), FD(uclm, KR), FD(vclm, KR), FD(M3nudgcof, KR),
  FD(ubarclm, K2), FD(vbarclm, K2), FD(M2nudgcof, K2),
  FD(bvf, KW), FD(alpha, K2), FD(beta, K2), FD(hsbl, K2), FD(hbbl, K2), FD(ghats, KWxNAT), FD(alfaobeta, KW),
  FD(tke, KWx3), FD(gls, KWx3), FD(Lscale, KW), FD(Akk, KW), FD(Akp, KW),
  FD(sc_r, KTAB_R), FD(Cs_r, KTAB_R), FD(sc_w, KTAB_W), FD(Cs_w, KTAB_W),
  FD(zeta_west, KBJ), FD(zeta_east, KBJ), FD(zeta_south, KBI), FD(zeta_north, KBI),
  FD(ubar_west, KBJ), FD(ubar_east, KBJ), FD(ubar_south, KBI), FD(ubar_north, KBI),
  FD(vbar_west, KBJ), FD(vbar_east, KBJ), FD(vbar_south, KBI), FD(vbar_north, KBI),
  FD(u_west, KBJN), FD(u_east, KBJN), FD(u_south, KBIN), FD(u_north, KBIN),
  FD(v_west, KBJN), FD(v_east, KBJN), FD(v_south, KBIN), FD(v_north, KBIN),
  FD(t_west, KBJT), FD(t_east, KBJT), FD(t_south, KBIT), FD(t_north, KBIT),
};
#define NFIELDS (sizeof(fields) / sizeof(fields[0]))

static size_t field_size(const orc_t *o, int kind) {
  const size_t N = o->c.N, NT = o->c.NT, NAT = o->c.NAT, p = o->nij;
  switch (kind) {
    case K2: return p;
    case KR: return p * N;
    case KW: return p * (N + 1);
    case KWx3: return p * (N + 1) * 3;
    case K2x3: return p * 3;
    case K2x2: return p * 2;
    case KRx2: return p * N * 2;
    case KTR: return p * N * 3 * NT;
    case KWx2: return p * (N + 1) * 2;
    case K2xNT: return p * NT;
    case KWxNAT: return p * (N + 1) * NAT;
    case KRxNT: return p * N * NT;
    case KTAB_R: return N;
    case KTAB_W: return N + 1;
    case KBJ: return o->nj;
    case KBI: return o->ni;
    case KBJN: return o->nj * N;
    case KBIN: return o->ni * N;
    case KBJT: return o->nj * N * NT;
    case KBIT: return o->ni * N * NT;
  }
  return 0;
}

orc_t *orc_create(const orc_cfg *cfg) {
  orc_t *o = (orc_t *)calloc(1, sizeof(orc_t));
  o->c = *cfg;
  alloc_bounds(&o->c);
  o->ni = (size_t)(o->c.UBi - o->c.LBi + 1);
  o->nj = (size_t)(o->c.UBj - o->c.LBj + 1);
  o->nij = o->ni * o->nj;
  o->ntiles = o->c.NtileI * o->c.NtileJ;
  o->b = (orc_bounds *)calloc((size_t)o->ntiles, sizeof(orc_bounds));
  for (int t = 0; t < o->ntiles; t++) orc_tile_bounds(&o->c, t, &o->b[t]);
  for (size_t k = 0; k < NFIELDS; k++)
    *(double **)((char *)o + fields[k].off) = dalloc(field_size(o, fields[k].kind));
  for (size_t k = 0; k < o->nij; k++) o->rmask[k] = o->umask[k] = o->vmask[k] = o->pmask[k] = 1.0;   /* all water */
  for (size_t k = 0; k < o->nij; k++) o->rmask_wet[k] = o->umask_wet[k] = o->vmask_wet[k] = o->pmask_wet[k] = 0.0;   /* IniVal, mod_grid.F:1428-1431 */
  o->ksbl = (int *)calloc(o->nij, sizeof(int));
  o->kbbl = (int *)calloc(o->nij, sizeof(int));
  /* the stepping indices of a state nobody has stepped yet: all 1 (what roms_hip_create sets); every routine checks
     them on entry (ORC_LOCALS -> orc_check_step), so a caller that forgot one reads a defined time level */
  o->s.iif = 1; o->s.indx1 = 1; o->s.kstp = o->s.krhs = o->s.knew = 1; o->s.nstp = o->s.nrhs = o->s.nnew = 1;
  return o;
}

void orc_destroy(orc_t *o) {
  if (!o) return;
  for (size_t k = 0; k < NFIELDS; k++) free(*(double **)((char *)o + fields[k].off));
  orc_avg_free(o);
  orc_dia_free(o);
  free(o->ksbl);
  free(o->kbbl);
  free(o->b);
  free(o);
}

double *orc_field(orc_t *o, const char *name, long *nel) {
  for (size_t k = 0; k < NFIELDS; k++)
    if (!strcmp(name, fields[k].name)) {
      if (nel) *nel = (long)field_size(o, fields[k].kind);
      return *(double **)((char *)o + fields[k].off);
    }
  { long n = -1; double *p = orc_avg_field(o, name, &n); if (p) { if (nel) *nel = n; return p; } }
  return orc_dia_field(o, name, nel);
}

orc_step *orc_stepping(orc_t *o) { return &o->s; }

/* time-level indices outside the arrays' extents (nstp, nnew, nrhs: 1..2, the tracer predictor level 3 is never a stepping
   index; kstp, knew, krhs: 1..3) would read or write before / behind an array: stop at once, with the routine's name */
void orc_check_step(const orc_t *o, const char *who) {
  const orc_step *s = &o->s;
  if (s->nstp < 1 || s->nstp > 2 || s->nnew < 1 || s->nnew > 2 || s->nrhs < 1 || s->nrhs > 2 || s->kstp < 1 || s->kstp > 3 ||
      s->knew < 1 || s->knew > 3 || s->krhs < 1 || s->krhs > 3 || s->indx1 < 1 || s->indx1 > 2) {
    fprintf(stderr, "oracle: %s called with a stepping index out of range (nstp %d nnew %d nrhs %d kstp %d knew %d krhs %d indx1 %d)\n",
            who, s->nstp, s->nnew, s->nrhs, s->kstp, s->knew, s->krhs, s->indx1);
    abort();
  }
}
orc_cfg *orc_config(orc_t *o) { return &o->c; }

void orc_get_bounds(orc_t *o, int tile, int *out) {
  const orc_bounds *b = &o->b[tile];
  const int v[] = { o->c.LBi, o->c.UBi, o->c.LBj, o->c.UBj, b->Istr, b->Iend, b->Jstr, b->Jend,
    b->IstrR, b->IendR, b->JstrR, b->JendR, b->IstrU, b->JstrV, b->IstrB, b->IendB, b->IstrM,
    b->JstrB, b->JendB, b->JstrM, b->IstrP, b->IendP, b->JstrP, b->JendP, b->IstrT, b->IendT,
    b->JstrT, b->JendT, b->Istrm3, b->Istrm2, b->Istrm1, b->IstrUm2, b->IstrUm1, b->Iendp1,
    b->Iendp2, b->Iendp2i, b->Iendp3, b->Jstrm3, b->Jstrm2, b->Jstrm1, b->JstrVm2, b->JstrVm1,
    b->Jendp1, b->Jendp2, b->Jendp2i, b->Jendp3, b->west, b->east, b->south, b->north,
    b->sw, b->se, b->nw, b->ne };
  memcpy(out, v, sizeof(v));
}

/* --------------------------------------------------------- periodic copies */
/* exchange_{p,r,u,v}2d_tile (serial: EW_exchange = NS_exchange = .TRUE.).
   The four grid types differ only in the transverse range. */
static void xrange(const orc_t *o, const orc_bounds *b, char grid, int *Jmin, int *Jmax,
                   int *Imin, int *Imax) {
  const int ewp = o->c.EWperiodic, nsp = o->c.NSperiodic;
  if (nsp) { *Jmin = b->Jstr; *Jmax = b->Jend; }
  else { *Jmin = (grid == 'r' || grid == 'u') ? b->JstrR : b->Jstr; *Jmax = b->JendR; }
  if (ewp) { *Imin = b->Istr; *Imax = b->Iend; }
  else { *Imin = (grid == 'r' || grid == 'v') ? b->IstrR : b->Istr; *Imax = b->IendR; }
}

static void exch_plane(const orc_t *o, const orc_bounds *b, char grid, double *A) {
  ORC_LOCALS(o);
  const int Lm = o->c.Lm, Mm = o->c.Mm, Ng = o->c.Nghost;
  int Jmin, Jmax, Imin, Imax;
  xrange(o, b, grid, &Jmin, &Jmax, &Imin, &Imax);
  if (o->c.EWperiodic) {
    if (b->west)
      for (int j = Jmin; j <= Jmax; j++) {
        A[X2(Lm + 1, j)] = A[X2(1, j)];
        A[X2(Lm + 2, j)] = A[X2(2, j)];
        if (Ng == 3) A[X2(Lm + 3, j)] = A[X2(3, j)];
      }
    if (b->east)
      for (int j = Jmin; j <= Jmax; j++) {
        A[X2(-2, j)] = A[X2(Lm - 2, j)];
        A[X2(-1, j)] = A[X2(Lm - 1, j)];
        A[X2(0, j)] = A[X2(Lm, j)];
      }
  }
  if (o->c.NSperiodic) {
    if (b->south)
      for (int i = Imin; i <= Imax; i++) {
        A[X2(i, Mm + 1)] = A[X2(i, 1)];
        A[X2(i, Mm + 2)] = A[X2(i, 2)];
        if (Ng == 3) A[X2(i, Mm + 3)] = A[X2(i, 3)];
      }
    if (b->north)
      for (int i = Imin; i <= Imax; i++) {
        A[X2(i, -2)] = A[X2(i, Mm - 2)];
        A[X2(i, -1)] = A[X2(i, Mm - 1)];
        A[X2(i, 0)] = A[X2(i, Mm)];
      }
  }
  if (o->c.EWperiodic && o->c.NSperiodic) {
    const int ne = Ng == 3 ? 3 : 2;
    if (b->sw)
      for (int dj = 1; dj <= ne; dj++)
        for (int di = 1; di <= ne; di++) A[X2(Lm + di, Mm + dj)] = A[X2(di, dj)];
    if (b->se)
      for (int dj = 1; dj <= ne; dj++)
        for (int di = -2; di <= 0; di++) A[X2(di, Mm + dj)] = A[X2(Lm + di, dj)];
    if (b->nw)
      for (int dj = -2; dj <= 0; dj++)
        for (int di = 1; di <= ne; di++) A[X2(Lm + di, dj)] = A[X2(di, Mm + dj)];
    if (b->ne)
      for (int dj = -2; dj <= 0; dj++)
        for (int di = -2; di <= 0; di++) A[X2(di, dj)] = A[X2(Lm + di, Mm + dj)];
  }
}

void orc_exchange2d(const orc_t *o, const orc_bounds *b, char grid, double *A) {
  if (!(o->c.EWperiodic || o->c.NSperiodic)) return;
  exch_plane(o, b, grid, A);
}

void orc_exchange3d(const orc_t *o, const orc_bounds *b, char grid, double *A, int nk) {
  if (!(o->c.EWperiodic || o->c.NSperiodic)) return;
  char g = grid == 'w' ? 'r' : grid;
  for (int k = 0; k < nk; k++) exch_plane(o, b, g, A + (size_t)k * o->nij);
}

/* ---------------------------------------------- gradient / closed edge fills */
/* bc_r2d_tile bc_2d.F:41 / bc_w3d_tile bc_3d.F:588: zero gradient + exchange */
static void bc_r_plane(const orc_t *o, const orc_bounds *b, double *A) {
  ORC_LOCALS(o);
  const int Istr = b->Istr, Iend = b->Iend, Jstr = b->Jstr, Jend = b->Jend;
  if (!o->c.EWperiodic) {
    if (b->east) for (int j = Jstr; j <= Jend; j++) A[X2(Iend + 1, j)] = A[X2(Iend, j)];
    if (b->west) for (int j = Jstr; j <= Jend; j++) A[X2(Istr - 1, j)] = A[X2(Istr, j)];
  }
  if (!o->c.NSperiodic) {
    if (b->north) for (int i = Istr; i <= Iend; i++) A[X2(i, Jend + 1)] = A[X2(i, Jend)];
    if (b->south) for (int i = Istr; i <= Iend; i++) A[X2(i, Jstr - 1)] = A[X2(i, Jstr)];
  }
  if (!(o->c.EWperiodic || o->c.NSperiodic)) {
    if (b->sw) A[X2(Istr - 1, Jstr - 1)] = 0.5 * (A[X2(Istr, Jstr - 1)] + A[X2(Istr - 1, Jstr)]);
    if (b->se) A[X2(Iend + 1, Jstr - 1)] = 0.5 * (A[X2(Iend, Jstr - 1)] + A[X2(Iend + 1, Jstr)]);
    if (b->nw) A[X2(Istr - 1, Jend + 1)] = 0.5 * (A[X2(Istr - 1, Jend)] + A[X2(Istr, Jend + 1)]);
    if (b->ne) A[X2(Iend + 1, Jend + 1)] = 0.5 * (A[X2(Iend + 1, Jend)] + A[X2(Iend, Jend + 1)]);
  }
}

void orc_bc_r2d(const orc_t *o, const orc_bounds *b, double *A) {
  bc_r_plane(o, b, A);
  orc_exchange2d(o, b, 'r', A);
}

void orc_bc_w3d(const orc_t *o, const orc_bounds *b, double *A, int nk) {
  for (int k = 0; k < nk; k++) bc_r_plane(o, b, A + (size_t)k * o->nij);
  orc_exchange3d(o, b, 'w', A, nk);
}

/* bc_u2d_tile bc_2d.F:164: LBC(:,isBu2d = isUbar)%closed, else zero gradient (:201-290) */
static void bc_u_plane(const orc_t *o, const orc_bounds *b, double *A, int ORC_ISUBAR_) {
  ORC_LOCALS(o);
  const int Istr = b->Istr, Iend = b->Iend, Jstr = b->Jstr, Jend = b->Jend;
  const double gamma2 = o->c.gamma2;
  if (!o->c.EWperiodic) {
    if (b->east) {
      if (orc_lbc(o, ORC_IEAST, ORC_ISUBAR_) == ORC_LBC_CLO) for (int j = Jstr; j <= Jend; j++) A[X2(Iend + 1, j)] = 0.0;
      else for (int j = Jstr; j <= Jend; j++) A[X2(Iend + 1, j)] = A[X2(Iend, j)];
    }
    if (b->west) {
      if (orc_lbc(o, ORC_IWEST, ORC_ISUBAR_) == ORC_LBC_CLO) for (int j = Jstr; j <= Jend; j++) A[X2(Istr, j)] = 0.0;
      else for (int j = Jstr; j <= Jend; j++) A[X2(Istr, j)] = A[X2(Istr + 1, j)];
    }
  }
  if (!o->c.NSperiodic) {
    int Imin = o->c.EWperiodic ? b->IstrU : b->Istr, Imax = o->c.EWperiodic ? b->Iend : b->IendR;
    const double *M = (o->c.options & ORC_MASKING) ? o->umask : NULL;   /* bc_2d.F:252,278 */
    if (b->north) {
      if (orc_lbc(o, ORC_INORTH, ORC_ISUBAR_) == ORC_LBC_CLO) for (int i = Imin; i <= Imax; i++) { A[X2(i, Jend + 1)] = gamma2 * A[X2(i, Jend)]; if (M) A[X2(i, Jend + 1)] = A[X2(i, Jend + 1)] * M[X2(i, Jend + 1)]; }
      else for (int i = b->IstrU; i <= Iend; i++) A[X2(i, Jend + 1)] = A[X2(i, Jend)];
    }
    if (b->south) {
      if (orc_lbc(o, ORC_ISOUTH, ORC_ISUBAR_) == ORC_LBC_CLO) for (int i = Imin; i <= Imax; i++) { A[X2(i, Jstr - 1)] = gamma2 * A[X2(i, Jstr)]; if (M) A[X2(i, Jstr - 1)] = A[X2(i, Jstr - 1)] * M[X2(i, Jstr - 1)]; }
      else for (int i = b->IstrU; i <= Iend; i++) A[X2(i, Jstr - 1)] = A[X2(i, Jstr)];
    }
  }
  if (!(o->c.EWperiodic || o->c.NSperiodic)) {
    if (b->sw) A[X2(Istr, Jstr - 1)] = 0.5 * (A[X2(Istr + 1, Jstr - 1)] + A[X2(Istr, Jstr)]);
    if (b->se) A[X2(Iend + 1, Jstr - 1)] = 0.5 * (A[X2(Iend, Jstr - 1)] + A[X2(Iend + 1, Jstr)]);
    if (b->nw) A[X2(Istr, Jend + 1)] = 0.5 * (A[X2(Istr, Jend)] + A[X2(Istr + 1, Jend + 1)]);
    if (b->ne) A[X2(Iend + 1, Jend + 1)] = 0.5 * (A[X2(Iend + 1, Jend)] + A[X2(Iend, Jend + 1)]);
  }
}

/* bc_v2d_tile bc_2d.F:342: LBC(:,isBv2d = isVbar)%closed, else zero gradient (:380-470) */
static void bc_v_plane(const orc_t *o, const orc_bounds *b, double *A, int ORC_ISVBAR_) {
  ORC_LOCALS(o);
  const int Istr = b->Istr, Iend = b->Iend, Jstr = b->Jstr, Jend = b->Jend;
  const double gamma2 = o->c.gamma2;
  if (!o->c.EWperiodic) {
    int Jmin = o->c.NSperiodic ? b->JstrV : b->Jstr, Jmax = o->c.NSperiodic ? b->Jend : b->JendR;
    const double *M = (o->c.options & ORC_MASKING) ? o->vmask : NULL;   /* bc_2d.F:392,418 */
    if (b->east) {
      if (orc_lbc(o, ORC_IEAST, ORC_ISVBAR_) == ORC_LBC_CLO) for (int j = Jmin; j <= Jmax; j++) { A[X2(Iend + 1, j)] = gamma2 * A[X2(Iend, j)]; if (M) A[X2(Iend + 1, j)] = A[X2(Iend + 1, j)] * M[X2(Iend + 1, j)]; }
      else for (int j = b->JstrV; j <= Jend; j++) A[X2(Iend + 1, j)] = A[X2(Iend, j)];
    }
    if (b->west) {
      if (orc_lbc(o, ORC_IWEST, ORC_ISVBAR_) == ORC_LBC_CLO) for (int j = Jmin; j <= Jmax; j++) { A[X2(Istr - 1, j)] = gamma2 * A[X2(Istr, j)]; if (M) A[X2(Istr - 1, j)] = A[X2(Istr - 1, j)] * M[X2(Istr - 1, j)]; }
      else for (int j = b->JstrV; j <= Jend; j++) A[X2(Istr - 1, j)] = A[X2(Istr, j)];
    }
  }
  if (!o->c.NSperiodic) {
    if (b->north) {
      if (orc_lbc(o, ORC_INORTH, ORC_ISVBAR_) == ORC_LBC_CLO) for (int i = Istr; i <= Iend; i++) A[X2(i, Jend + 1)] = 0.0;
      else for (int i = Istr; i <= Iend; i++) A[X2(i, Jend + 1)] = A[X2(i, Jend)];
    }
    if (b->south) {
      if (orc_lbc(o, ORC_ISOUTH, ORC_ISVBAR_) == ORC_LBC_CLO) for (int i = Istr; i <= Iend; i++) A[X2(i, Jstr)] = 0.0;
      else for (int i = Istr; i <= Iend; i++) A[X2(i, Jstr)] = A[X2(i, Jstr + 1)];
    }
  }
  if (!(o->c.EWperiodic || o->c.NSperiodic)) {
    if (b->sw) A[X2(Istr - 1, Jstr)] = 0.5 * (A[X2(Istr, Jstr)] + A[X2(Istr - 1, Jstr + 1)]);
    if (b->se) A[X2(Iend + 1, Jstr)] = 0.5 * (A[X2(Iend, Jstr)] + A[X2(Iend + 1, Jstr + 1)]);
    if (b->nw) A[X2(Istr - 1, Jend + 1)] = 0.5 * (A[X2(Istr - 1, Jend)] + A[X2(Istr, Jend + 1)]);
    if (b->ne) A[X2(Iend + 1, Jend + 1)] = 0.5 * (A[X2(Iend + 1, Jend)] + A[X2(Iend, Jend + 1)]);
  }
}

void orc_bc_u2d(const orc_t *o, const orc_bounds *b, double *A) { bc_u_plane(o, b, A, ORC_ISUBAR); orc_exchange2d(o, b, 'u', A); }
void orc_bc_v2d(const orc_t *o, const orc_bounds *b, double *A) { bc_v_plane(o, b, A, ORC_ISVBAR); orc_exchange2d(o, b, 'v', A); }
/* bc_u3d_tile, bc_v3d_tile bc_3d.F:164, :367: the same level by level with LBC(:,isBu3d = isUvel) / (:,isBv3d = isVvel) */
void orc_bc_u3d(const orc_t *o, const orc_bounds *b, double *A, int nk) {
  for (int k = 0; k < nk; k++) bc_u_plane(o, b, A + (size_t)k * o->nij, ORC_ISUVEL);
  orc_exchange3d(o, b, 'u', A, nk);
}
void orc_bc_v3d(const orc_t *o, const orc_bounds *b, double *A, int nk) {
  for (int k = 0; k < nk; k++) bc_v_plane(o, b, A + (size_t)k * o->nij, ORC_ISVVEL);
  orc_exchange3d(o, b, 'v', A, nk);
}

/* the lateral boundary conditions of the state (zetabc.F ... t3dbc_im.F), closed and open: orc_obc.c */
