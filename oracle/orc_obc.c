/*
 * orc_obc.c -- lateral boundary conditions of the state, closed and open.  TEST INFRASTRUCTURE (see orc.h).
 *
 * Follows (myroms/roms, ROMS/Nonlinear):
 *   zetabc.F    zetabc_tile :60-650      radiation (+nudging), Chapman explicit/implicit, clamped, gradient, closed
 *   u2dbc_im.F  u2dbc_tile  :51-1312     radiation (+nudging), Flather, Shchepetkin, clamped, gradient, closed;
 *   v2dbc_im.F  v2dbc_tile  :52-1362     tangential component under Flather/Shchepetkin: Chapman (:921-943)
 *   u3dbc_im.F  u3dbc_tile  :50-737      radiation (+nudging), clamped, gradient, closed
 *   v3dbc_im.F  v3dbc_tile  :50-737
 *   t3dbc_im.F  t3dbc_tile  :50-684
 * (not restated: reduced physics `Red`, nested `Nes`, the cpp variants SSH_TIDES, ATM_PRESS, IMPLICIT_NUDGING,
 *  WET_DRY, CELERITY_WRITE, nudging with climatology coefficients LnudgeM2CLM ...)
 *
 * Formulation.  The reference writes every condition four times, once per edge.  Here an edge is described by its
 * inward normal (di,dj) and the direction s runs along it (ti,tj); B is the boundary point, I1 = B + n and I2 = B + 2n
 * the first two interior points.  The radiation condition of every variable is then one function, rad_point():
 *
 *   dQdt = Q(I1,know) - Q(I1,kout);  dQdn = Q(I1,kout) - Q(I2,kout)
 *   lower(L) = Q(L,s) - Q(L,s-1), upper(L) = Q(L,s+1) - Q(L,s) at level know on the lines L = B and L = I1
 *     -- the reference's `grad` pairs: (grad(j), grad(j+1)) for rho-type and normal-velocity points, zetabc.F:122-133,
 *        u2dbc_im.F:146-151; (grad(i-1), grad(i)) for the staggered tangential velocity, u2dbc_im.F:851-856 --
 *   Cn = dQdt*dQdn, Ct = RADIATION_2D ? MIN(cff, MAX(dQdt*dQds, -cff)) : 0,  cff = MAX(dQdn^2 + dQds^2, eps)
 *   Q(B,kout) = (cff*Q(B,know) + Cn*Q(I1,kout) - MAX(Ct,0)*lower(B) - MIN(Ct,0)*upper(B)) / (cff + Cn)
 *
 * with the operand order of the reference's statements (sums of two squares and two-term averages commute in
 * IEEE arithmetic; the four-term sums h+zeta+h+zeta run in index order on every edge, u2dbc_im.F:252-255,603-606).
 *
 * PARITY: pinned -- every kind on every edge, routine by routine against zetabc_tile ... t3dbc_tile of oracle/_ref
 * (tests/test_oracle_vs_ref.py::test_open_boundary_routines_bitwise) and in whole main3d passes of the
 * reference's KELVIN application (Cha/Fla west, Rad east, RADIATION_2D).
 */
#include "orc.h"
#include <math.h>
#include <stdlib.h>

#define MAX(a, b) ((a) > (b) ? (a) : (b))
#define MIN(a, b) ((a) < (b) ? (a) : (b))

int orc_lbc(const orc_t *o, int edge, int var) {
  const int per = (edge == ORC_IWEST || edge == ORC_IEAST) ? o->c.EWperiodic : o->c.NSperiodic;
  if (per) return ORC_LBC_PER;
  const int k = o->c.lbc[edge][var];
  return k == ORC_LBC_DEFAULT ? ORC_LBC_CLO : k;
}

int orc_lbc_acquire(const orc_t *o, int edge, int var) {
  const int k = orc_lbc(o, edge, var);
  if (k == ORC_LBC_CLA || k == ORC_LBC_RADNUD || k == ORC_LBC_FLA || k == ORC_LBC_SHC) return 1;
  if (var == ORC_ISFSUR)      /* Flather / Shchepetkin momentum conditions need the free-surface data too */
    for (int v = ORC_ISUBAR; v <= ORC_ISVBAR; v++) {
      const int q = orc_lbc(o, edge, v);
      if (q == ORC_LBC_FLA || q == ORC_LBC_SHC) return 1;
    }
  return 0;
}

int orc_lbc_open(const orc_t *o) {
  for (int e = 0; e < 4; e++)
    for (int v = 0; v < ORC_ISTVAR + o->c.NT; v++) {
      const int k = orc_lbc(o, e, v);
      if (k != ORC_LBC_CLO && k != ORC_LBC_PER) return 1;
    }
  return 0;
}

/* one edge of one tile for one staggering ('r', 'u', 'v') */
typedef struct {
  int e;            /* ORC_IWEST ... */
  int di, dj;       /* inward normal */
  int ti, tj;       /* along the edge */
  int i0, j0;       /* boundary point at s = 0 */
  int s0, s1;       /* range of s the open conditions run over */
  int normal;       /* the variable is the velocity component normal to this edge */
} edge_t;

static int edge_setup(const orc_t *o, const orc_bounds *b, int e, char grid, edge_t *E) {
  const int we = (e == ORC_IWEST || e == ORC_IEAST);
  if (we ? o->c.EWperiodic : o->c.NSperiodic) return 0;
  if (!(e == ORC_IWEST ? b->west : e == ORC_IEAST ? b->east : e == ORC_ISOUTH ? b->south : b->north)) return 0;
  E->e = e;
  E->di = e == ORC_IWEST ? 1 : e == ORC_IEAST ? -1 : 0;
  E->dj = e == ORC_ISOUTH ? 1 : e == ORC_INORTH ? -1 : 0;
  E->ti = we ? 0 : 1;
  E->tj = we ? 1 : 0;
  E->normal = (we && grid == 'u') || (!we && grid == 'v');
  if (we) {
    E->i0 = e == ORC_IWEST ? (grid == 'u' ? b->Istr : b->Istr - 1) : b->Iend + 1;
    E->j0 = 0;
    E->s0 = grid == 'v' ? b->JstrV : b->Jstr;
    E->s1 = b->Jend;
  } else {
    E->j0 = e == ORC_ISOUTH ? (grid == 'v' ? b->Jstr : b->Jstr - 1) : b->Jend + 1;
    E->i0 = 0;
    E->s0 = grid == 'u' ? b->IstrU : b->Istr;
    E->s1 = b->Iend;
  }
  return 1;
}
#define EI(E, s) ((E)->i0 + (E)->ti * (s))
#define EJ(E, s) ((E)->j0 + (E)->tj * (s))

static const double eps = 1.0E-20;

/* implicit upstream radiation (+ nudging): zetabc.F:119-184, u2dbc_im.F:145-220,846-925, u3dbc_im.F:99-182, t3dbc_im.F:98-175.
   Qn, Qo: the planes of the levels know (nstp) and kout (nout); fm: face mask of the tangential differences (rho-type
   variables under MASKING) or NULL */
static double rad_point(const orc_t *o, const edge_t *E, const double *Qn, const double *Qo, int i, int j,
                        const double *fm, int nudging, double obc_in, double obc_out, double dtn, double bry, int inner) {
  ORC_LOCALS(o);
  const int di = E->di, dj = E->dj, ti = E->ti, tj = E->tj;
  const int i1 = i + di, j1 = j + dj, i2 = i + 2 * di, j2 = j + 2 * dj;
  double gBl = Qn[X2(i, j)] - Qn[X2(i - ti, j - tj)], gBu = Qn[X2(i + ti, j + tj)] - Qn[X2(i, j)];
  double gIl = Qn[X2(i1, j1)] - Qn[X2(i1 - ti, j1 - tj)], gIu = Qn[X2(i1 + ti, j1 + tj)] - Qn[X2(i1, j1)];
  if (fm) {
    gBl = gBl * fm[X2(i, j)];
    gBu = gBu * fm[X2(i + ti, j + tj)];
    gIl = gIl * fm[X2(i1, j1)];
    gIu = gIu * fm[X2(i1 + ti, j1 + tj)];
  }
  /* zetabc.F:486-487: at the SOUTHERN edge the free-surface condition advects with the differences of the first interior
     row, grad(i,Jstr) and grad(i+1,Jstr), where every other edge and routine takes the boundary row (`inner`: the two
     places where that one branch of the reference departs from the pattern of the other 23 -- reproduced, not repaired) */
  if (inner) { gBl = gIl; gBu = gIu; }
  double dQdt = Qn[X2(i1, j1)] - Qo[X2(i1, j1)];
  /* (zetabc.F:455: the southern free-surface condition differences towards the boundary row itself, zeta(i,Jstr,kout) -
     zeta(i,Jstr-1,kout), with the value the boundary point still holds at level kout) */
  const double dQdn = inner ? Qo[X2(i1, j1)] - Qo[X2(i, j)] : Qo[X2(i1, j1)] - Qo[X2(i2, j2)];
  double tau = 0.0;
  if (nudging) {
    tau = (dQdt * dQdn) < 0.0 ? obc_in : obc_out;
    tau = tau * dtn;
  }
  if ((dQdt * dQdn) < 0.0) dQdt = 0.0;
  const double dQds = (dQdt * (gIl + gIu)) > 0.0 ? gIl : gIu;
  const double cff = MAX(dQdn * dQdn + dQds * dQds, eps);
  const double Cn = dQdt * dQdn;
  const double Ct = (o->c.options & ORC_RADIATION_2D) ? MIN(cff, MAX(dQdt * dQds, -cff)) : 0.0;
  double val = (cff * Qn[X2(i, j)] + Cn * Qo[X2(i1, j1)] - MAX(Ct, 0.0) * gBl - MIN(Ct, 0.0) * gBu) / (cff + Cn);
  if (nudging) val = val + tau * (bry - Qn[X2(i, j)]);
  return val;
}

/* time level the conditions of the barotropic step take as "now", and its time step: zetabc.F:100-112 (LF-AM3 kernel) */
static void know_dt2d(const orc_t *o, int *know, double *dt2d) {
  if (o->s.iif == 1) { *know = o->s.krhs; *dt2d = o->c.dtfast; }
  else if (o->s.predictor) { *know = o->s.krhs; *dt2d = 2.0 * o->c.dtfast; }
  else { *know = o->s.kstp; *dt2d = o->c.dtfast; }
}

static double *bry2(const orc_t *o, int e, double *w, double *s, double *ea, double *n) {
  (void)o;
  return e == ORC_IWEST ? w : e == ORC_ISOUTH ? s : e == ORC_IEAST ? ea : n;
}

/* corners: the mean of the two neighbouring boundary values (zetabc.F:600-640 ...), where neither direction is periodic */
static void corners_r(const orc_t *o, const orc_bounds *b, double *A) {
  ORC_LOCALS(o);
  const int Istr = b->Istr, Iend = b->Iend, Jstr = b->Jstr, Jend = b->Jend;
  if (o->c.EWperiodic || o->c.NSperiodic) return;
  if (b->sw) A[X2(Istr - 1, Jstr - 1)] = 0.5 * (A[X2(Istr, Jstr - 1)] + A[X2(Istr - 1, Jstr)]);
  if (b->se) A[X2(Iend + 1, Jstr - 1)] = 0.5 * (A[X2(Iend, Jstr - 1)] + A[X2(Iend + 1, Jstr)]);
  if (b->nw) A[X2(Istr - 1, Jend + 1)] = 0.5 * (A[X2(Istr - 1, Jend)] + A[X2(Istr, Jend + 1)]);
  if (b->ne) A[X2(Iend + 1, Jend + 1)] = 0.5 * (A[X2(Iend + 1, Jend)] + A[X2(Iend, Jend + 1)]);
}
static void corners_u(const orc_t *o, const orc_bounds *b, double *A) {
  ORC_LOCALS(o);
  const int Istr = b->Istr, Iend = b->Iend, Jstr = b->Jstr, Jend = b->Jend;
  if (o->c.EWperiodic || o->c.NSperiodic) return;
  if (b->sw) A[X2(Istr, Jstr - 1)] = 0.5 * (A[X2(Istr + 1, Jstr - 1)] + A[X2(Istr, Jstr)]);
  if (b->se) A[X2(Iend + 1, Jstr - 1)] = 0.5 * (A[X2(Iend, Jstr - 1)] + A[X2(Iend + 1, Jstr)]);
  if (b->nw) A[X2(Istr, Jend + 1)] = 0.5 * (A[X2(Istr, Jend)] + A[X2(Istr + 1, Jend + 1)]);
  if (b->ne) A[X2(Iend + 1, Jend + 1)] = 0.5 * (A[X2(Iend + 1, Jend)] + A[X2(Iend, Jend + 1)]);
}
static void corners_v(const orc_t *o, const orc_bounds *b, double *A) {
  ORC_LOCALS(o);
  const int Istr = b->Istr, Iend = b->Iend, Jstr = b->Jstr, Jend = b->Jend;
  if (o->c.EWperiodic || o->c.NSperiodic) return;
  if (b->sw) A[X2(Istr - 1, Jstr)] = 0.5 * (A[X2(Istr, Jstr)] + A[X2(Istr - 1, Jstr + 1)]);
  if (b->se) A[X2(Iend + 1, Jstr)] = 0.5 * (A[X2(Iend, Jstr)] + A[X2(Iend + 1, Jstr + 1)]);
  if (b->nw) A[X2(Istr - 1, Jend + 1)] = 0.5 * (A[X2(Istr - 1, Jend)] + A[X2(Istr, Jend + 1)]);
  if (b->ne) A[X2(Iend + 1, Jend + 1)] = 0.5 * (A[X2(Iend + 1, Jend)] + A[X2(Iend, Jend + 1)]);
}

/* ------------------------------------------------------------------------------------------- zetabc_tile */
void orc_zetabc(const orc_t *o, const orc_bounds *b, int kout) {
  ORC_LOCALS(o);
  int know;
  double dt2d;
  know_dt2d(o, &know, &dt2d);
  double *Zo = o->zeta + (size_t)(kout - 1) * nij;
  const double *Zn = o->zeta + (size_t)(know - 1) * nij;
  const int msk = (o->c.options & ORC_MASKING) != 0;
  const double g = o->c.g;
  /* the reference's order: west, east, south, north */
  static const int order[4] = { ORC_IWEST, ORC_IEAST, ORC_ISOUTH, ORC_INORTH };
  for (int q = 0; q < 4; q++) {
    edge_t E;
    if (!edge_setup(o, b, order[q], 'r', &E)) continue;
    const int kind = orc_lbc(o, E.e, ORC_ISFSUR);
    const double *bry = bry2(o, E.e, o->zeta_west, o->zeta_south, o->zeta_east, o->zeta_north);
    const double *fm = msk ? (E.ti ? o->umask : o->vmask) : NULL;
    const double *pmn = E.ti ? o->pn : o->pm;       /* the metric across the edge */
    for (int s = E.s0; s <= E.s1; s++) {
      const int i = EI(&E, s), j = EJ(&E, s), i1 = i + E.di, j1 = j + E.dj;
      const double bv = bry[s - (E.ti ? LBi : LBj)];
      double val;
      if (kind == ORC_LBC_RAD || kind == ORC_LBC_RADNUD) {
        val = rad_point(o, &E, Zn, Zo, i, j, fm, kind == ORC_LBC_RADNUD, o->c.FSobc_in[E.e], o->c.FSobc_out[E.e], dt2d, bv, E.e == ORC_ISOUTH);
      } else if (kind == ORC_LBC_CHE) {               /* :186-204 */
        const double cff = dt2d * pmn[X2(i1, j1)];
        const double dep = o->h[X2(i1, j1)] + Zn[X2(i1, j1)];
        const double cff1 = sqrt(g * (o->wet_dry ? (dep > o->Dcrit ? dep : o->Dcrit) : dep));   /* WET_DRY: MAX(..., Dcrit) :190-192 */
        const double Cx = cff * cff1;
        val = (1.0 - Cx) * Zn[X2(i, j)] + Cx * Zn[X2(i1, j1)];
      } else if (kind == ORC_LBC_CHI) {               /* :208-227 */
        const double cff = dt2d * pmn[X2(i1, j1)];
        const double dep = o->h[X2(i1, j1)] + Zn[X2(i1, j1)];
        const double cff1 = sqrt(g * (o->wet_dry ? (dep > o->Dcrit ? dep : o->Dcrit) : dep));   /* WET_DRY: MAX(..., Dcrit) :190-192 */
        const double Cx = cff * cff1;
        const double cff2 = 1.0 / (1.0 + Cx);
        val = cff2 * (Zn[X2(i, j)] + Cx * Zo[X2(i1, j1)]);
      } else if (kind == ORC_LBC_CLA) {
        val = bv;
      } else {                                        /* gradient, closed: zero gradient */
        val = Zo[X2(i1, j1)];
      }
      if (msk) val = val * o->rmask[X2(i, j)];
      Zo[X2(i, j)] = val;
    }
  }
  corners_r(o, b, Zo);
  if (o->wet_dry) {
    /* "Ensure that water level on boundary cells is above bed elevation" zetabc.F:783-874 (every kind of condition) */
    const double cff = o->Dcrit - eps;
    const int Istr = b->Istr, Iend = b->Iend, Jstr = b->Jstr, Jend = b->Jend;
#define WDZ(i, j) if (Zo[X2(i, j)] <= (o->Dcrit - o->h[X2(i, j)])) Zo[X2(i, j)] = cff - o->h[X2(i, j)]
    if (!o->c.EWperiodic) {
      if (b->west) for (int j = Jstr; j <= Jend; j++) { WDZ(Istr - 1, j); }
      if (b->east) for (int j = Jstr; j <= Jend; j++) { WDZ(Iend + 1, j); }
    }
    if (!o->c.NSperiodic) {
      if (b->south) for (int i = Istr; i <= Iend; i++) { WDZ(i, Jstr - 1); }
      if (b->north) for (int i = Istr; i <= Iend; i++) { WDZ(i, Jend + 1); }
    }
    if (!(o->c.EWperiodic || o->c.NSperiodic)) {
      if (b->south && b->west) { WDZ(Istr - 1, Jstr - 1); }
      if (b->south && b->east) { WDZ(Iend + 1, Jstr - 1); }
      if (b->north && b->west) { WDZ(Istr - 1, Jend + 1); }
      if (b->north && b->east) { WDZ(Iend + 1, Jend + 1); }
    }
#undef WDZ
  }
}

/* ------------------------------------------------------------------------ u2dbc_tile / v2dbc_tile (one function) */
/* grid 'u': normal at W/E, tangential at S/N; grid 'v': the other way round */
static void uv2dbc(const orc_t *o, const orc_bounds *b, int kout, char grid) {
  ORC_LOCALS(o);
  int know;
  double dt2d;
  know_dt2d(o, &know, &dt2d);
  const int isU = grid == 'u';
  double *Q = isU ? o->ubar : o->vbar;
  double *Qo = Q + (size_t)(kout - 1) * nij;
  const double *Qn = Q + (size_t)(know - 1) * nij;
  const double *Zn = o->zeta + (size_t)(know - 1) * nij, *Zo = o->zeta + (size_t)(kout - 1) * nij;
  const int msk = (o->c.options & ORC_MASKING) != 0;
  const double *qmask = isU ? o->umask : o->vmask;
  const double g = o->c.g, gamma2 = o->c.gamma2;
  const double Co = 1.0 / (2.0 + sqrt(2.0));                   /* mod_scalars.F:4435 */
  const int var = isU ? ORC_ISUBAR : ORC_ISVBAR;
  /* u2dbc: west, east, south, north; v2dbc: south, north, west, east */
  static const int order_u[4] = { ORC_IWEST, ORC_IEAST, ORC_ISOUTH, ORC_INORTH };
  static const int order_v[4] = { ORC_ISOUTH, ORC_INORTH, ORC_IWEST, ORC_IEAST };
  for (int q = 0; q < 4; q++) {
    edge_t E;
    if (!edge_setup(o, b, (isU ? order_u : order_v)[q], grid, &E)) continue;
    const int kind = orc_lbc(o, E.e, var);
    const double *bry = isU ? bry2(o, E.e, o->ubar_west, o->ubar_south, o->ubar_east, o->ubar_north)
                            : bry2(o, E.e, o->vbar_west, o->vbar_south, o->vbar_east, o->vbar_north);
    const double *zbry = bry2(o, E.e, o->zeta_west, o->zeta_south, o->zeta_east, o->zeta_north);
    const int lb = E.ti ? LBi : LBj;
    if (kind == ORC_LBC_CLO) {
      if (E.normal) {                                           /* no flow through the wall */
        for (int s = E.s0; s <= E.s1; s++) Qo[X2(EI(&E, s), EJ(&E, s))] = 0.0;
      } else {                                                  /* free slip / no slip: u2dbc_im.F:966-985 */
        int s0, s1;
        if (E.ti) { s0 = o->c.EWperiodic ? b->IstrU : b->Istr; s1 = o->c.EWperiodic ? b->Iend : b->IendR; }
        else { s0 = o->c.NSperiodic ? b->JstrV : b->Jstr; s1 = o->c.NSperiodic ? b->Jend : b->JendR; }
        for (int s = s0; s <= s1; s++) {
          const int i = EI(&E, s), j = EJ(&E, s);
          Qo[X2(i, j)] = gamma2 * Qo[X2(i + E.di, j + E.dj)];
          if (msk) Qo[X2(i, j)] = Qo[X2(i, j)] * qmask[X2(i, j)];
        }
      }
      continue;
    }
    const double *pmn = E.ti ? o->pn : o->pm;                   /* metric across the edge */
    const double sgn = (E.e == ORC_IWEST || E.e == ORC_ISOUTH) ? -1.0 : 1.0;
    for (int s = E.s0; s <= E.s1; s++) {
      const int i = EI(&E, s), j = EJ(&E, s), i1 = i + E.di, j1 = j + E.dj;
      const double bv = bry[s - lb];
      double val;
      if (kind == ORC_LBC_RAD || kind == ORC_LBC_RADNUD) {
        double oin = o->c.M2obc_in[E.e], oout = o->c.M2obc_out[E.e];
        if (kind == ORC_LBC_RADNUD && (o->clima_flags & 32)) {     /* LnudgeM2CLM: u2dbc_im.F:158-162 ..., v2dbc_im.F (as u3dbc / v3dbc) */
          oout = 0.5 * (o->M2nudgcof[X2(i - (isU ? 1 : 0), j - (isU ? 0 : 1))] + o->M2nudgcof[X2(i, j)]);
          oin = o->c.obcfac * oout;
        }
        val = rad_point(o, &E, Qn, Qo, i, j, NULL, kind == ORC_LBC_RADNUD, oin, oout, dt2d, bv, 0);
      } else if (kind == ORC_LBC_CLA) {
        val = bv;
      } else if (kind == ORC_LBC_GRA) {
        val = Qo[X2(i1, j1)];
      } else if (E.normal) {
        /* the two rho points either side of the boundary velocity point, in index order */
        const int ilo = i - (isU ? 1 : 0), jlo = j - (isU ? 0 : 1);
        const size_t lo = X2(ilo, jlo), hi = X2(i, j);
        const size_t in = (E.e == ORC_IWEST || E.e == ORC_ISOUTH) ? hi : lo, out = (E.e == ORC_IWEST || E.e == ORC_ISOUTH) ? lo : hi;
        if (kind == ORC_LBC_FLA) {                              /* u2dbc_im.F:224-291,575-642 */
          const double cff = 1.0 / (0.5 * (o->h[lo] + Zn[lo] + o->h[hi] + Zn[hi]));
          const double Cx = sqrt(g * cff);
          val = bv + sgn * (Cx * (0.5 * (Zn[lo] + Zn[hi]) - zbry[s - lb]));
        } else {                                                /* Shchepetkin :296-369,647-720 */
          const double cff = o->wet_dry ? 0.5 * (o->h[lo] + Zn[lo] + o->h[hi] + Zn[hi]) : 0.5 * (o->h[lo] + o->h[hi]);   /* WET_DRY :339-347 */
          const double cff1 = sqrt(g / cff);
          const double Cx = dt2d * cff1 * cff * 0.5 * (pmn[lo] + pmn[hi]);
          double Zx = (0.5 + Cx) * Zn[in] + (0.5 - Cx) * Zn[out];
          if (Cx > Co) {
            const double r = 1.0 - Co / Cx;
            const double cff2 = r * r;
            const double cff3 = Zo[in] + Cx * Zn[out] - (1.0 + Cx) * Zn[in];
            Zx = Zx + cff2 * cff3;
          }
          if (sgn < 0.0) val = 0.5 * ((1.0 - Cx) * Qn[X2(i, j)] + Cx * Qn[X2(i1, j1)] + bv - cff1 * (Zx - zbry[s - lb]));
          else val = 0.5 * ((1.0 - Cx) * Qn[X2(i, j)] + Cx * Qn[X2(i1, j1)] + bv + cff1 * (Zx - zbry[s - lb]));
        }
      } else {
        /* tangential component under Flather / Shchepetkin: Chapman, u2dbc_im.F:921-943.  The two rho points of the first
           interior line either side of the velocity point */
        const size_t a = X2(i1 - E.ti, j1 - E.tj), c = X2(i1, j1);
        const double cff = dt2d * 0.5 * (pmn[a] + pmn[c]);
        const double cff1 = sqrt(g * 0.5 * (o->h[a] + Zn[a] + o->h[c] + Zn[c]));
        const double Ce = cff * cff1;
        const double cff2 = 1.0 / (1.0 + Ce);
        val = cff2 * (Qn[X2(i, j)] + Ce * Qo[X2(i1, j1)]);
      }
      if (msk) val = val * qmask[X2(i, j)];
      Qo[X2(i, j)] = val;
    }
  }
  if (isU) corners_u(o, b, Qo);
  else corners_v(o, b, Qo);
  if (o->wet_dry) {
    /* "Impose wetting and drying conditions" u2dbc_im.F:1190-1318, v2dbc_im.F:1239-1367, AS WRITTEN: the factor of the
       barotropic step from the wet mask and the value at one point (mi,mj) applied to the value at (ti,tj) -- the same
       point everywhere except v2dbc's western edge (mask and sign at Istr-1, product stored at Istr, v2dbc_im.F:1250-1255) */
    const double *mw = isU ? o->umask_wet : o->vmask_wet;
    const int Istr = b->Istr, Iend = b->Iend, Jstr = b->Jstr, Jend = b->Jend;
#define WDP(mi, mj, ti, tj) Qo[X2(ti, tj)] = Qo[X2(ti, tj)] * orc_wd_fac(mw[X2(mi, mj)], Qo[X2(mi, mj)])
    if (!o->c.EWperiodic) {
      if (b->west) {
        if (isU) for (int j = Jstr; j <= Jend; j++) WDP(Istr, j, Istr, j);
        else for (int j = b->JstrV; j <= Jend; j++) WDP(Istr - 1, j, Istr, j);
      }
      if (b->east) {
        if (isU) for (int j = Jstr; j <= Jend; j++) WDP(Iend + 1, j, Iend + 1, j);
        else for (int j = b->JstrV; j <= Jend; j++) WDP(Iend + 1, j, Iend + 1, j);
      }
    }
    if (!o->c.NSperiodic) {
      if (b->south) {
        if (isU) for (int i = b->IstrU; i <= Iend; i++) WDP(i, Jstr - 1, i, Jstr - 1);
        else for (int i = Istr; i <= Iend; i++) WDP(i, Jstr, i, Jstr);
      }
      if (b->north) for (int i = Istr; i <= Iend; i++) WDP(i, Jend + 1, i, Jend + 1);
    }
    if (!(o->c.EWperiodic || o->c.NSperiodic)) {
      if (isU) {
        if (b->south && b->west) WDP(Istr, Jstr - 1, Istr, Jstr - 1);
        if (b->south && b->east) WDP(Iend + 1, Jstr - 1, Iend + 1, Jstr - 1);
        if (b->north && b->west) WDP(Istr, Jend + 1, Istr, Jend + 1);
        if (b->north && b->east) WDP(Iend + 1, Jend + 1, Iend + 1, Jend + 1);
      } else {
        if (b->south && b->west) WDP(Istr - 1, Jstr, Istr - 1, Jstr);
        if (b->south && b->east) WDP(Iend + 1, Jstr, Iend + 1, Jstr);
        if (b->north && b->west) WDP(Istr - 1, Jend + 1, Istr - 1, Jend + 1);
        if (b->north && b->east) WDP(Iend + 1, Jend + 1, Iend + 1, Jend + 1);
      }
    }
#undef WDP
  }
}
void orc_u2dbc(const orc_t *o, const orc_bounds *b, int kout) { uv2dbc(o, b, kout, 'u'); }
void orc_v2dbc(const orc_t *o, const orc_bounds *b, int kout) { uv2dbc(o, b, kout, 'v'); }

/* ------------------------------------------------------------------------ u3dbc_tile / v3dbc_tile / t3dbc_tile */
/* grid 'u', 'v', or 'r' (tracer itrc); levels nstp ("now") and nout */
static void bc3d(const orc_t *o, const orc_bounds *b, int nout, char grid, int itrc) {
  ORC_LOCALS(o);
  const int nstp = o->s.nstp, NT = o->c.NT;
  const int msk = (o->c.options & ORC_MASKING) != 0;
  const double gamma2 = o->c.gamma2, dt = o->c.dt;
  const int var = grid == 'u' ? ORC_ISUVEL : grid == 'v' ? ORC_ISVVEL : ORC_ISTVAR + itrc - 1;
  const double *qmask = grid == 'u' ? o->umask : grid == 'v' ? o->vmask : o->rmask;
  static const int order_u[4] = { ORC_IWEST, ORC_IEAST, ORC_ISOUTH, ORC_INORTH };
  static const int order_v[4] = { ORC_ISOUTH, ORC_INORTH, ORC_IWEST, ORC_IEAST };
  (void)NT;
  for (int k = 1; k <= N; k++) {
    double *Qo;
    const double *Qn;
    if (grid == 'r') {
      Qo = o->t + (((size_t)(nout - 1) + 3 * (size_t)(itrc - 1)) * N + (size_t)(k - 1)) * nij;
      Qn = o->t + (((size_t)(nstp - 1) + 3 * (size_t)(itrc - 1)) * N + (size_t)(k - 1)) * nij;
    } else {
      double *Q = grid == 'u' ? o->u : o->v;
      Qo = Q + ((size_t)(nout - 1) * N + (size_t)(k - 1)) * nij;
      Qn = Q + ((size_t)(nstp - 1) * N + (size_t)(k - 1)) * nij;
    }
    for (int q = 0; q < 4; q++) {
      edge_t E;
      if (!edge_setup(o, b, (grid == 'v' ? order_v : order_u)[q], grid, &E)) continue;
      const int kind = orc_lbc(o, E.e, var);
      const int lb = E.ti ? LBi : LBj;
      const size_t nb = E.ti ? ni : o->nj;                      /* length of a boundary line */
      const double *bry;
      double obc_in, obc_out;
      if (grid == 'u') { bry = bry2(o, E.e, o->u_west, o->u_south, o->u_east, o->u_north) + (size_t)(k - 1) * nb; obc_in = o->c.M3obc_in[E.e]; obc_out = o->c.M3obc_out[E.e]; }
      else if (grid == 'v') { bry = bry2(o, E.e, o->v_west, o->v_south, o->v_east, o->v_north) + (size_t)(k - 1) * nb; obc_in = o->c.M3obc_in[E.e]; obc_out = o->c.M3obc_out[E.e]; }
      else {
        bry = bry2(o, E.e, o->t_west, o->t_south, o->t_east, o->t_north) + ((size_t)(itrc - 1) * N + (size_t)(k - 1)) * nb;
        obc_in = o->c.Tobc_in[itrc - 1][E.e]; obc_out = o->c.Tobc_out[itrc - 1][E.e];
      }
      if (kind == ORC_LBC_CLO && grid != 'r') {
        if (E.normal) {
          for (int s = E.s0; s <= E.s1; s++) Qo[X2(EI(&E, s), EJ(&E, s))] = 0.0;
        } else {
          int s0, s1;
          if (E.ti) { s0 = o->c.EWperiodic ? b->IstrU : b->Istr; s1 = o->c.EWperiodic ? b->Iend : b->IendR; }
          else { s0 = o->c.NSperiodic ? b->JstrV : b->Jstr; s1 = o->c.NSperiodic ? b->Jend : b->JendR; }
          for (int s = s0; s <= s1; s++) {
            const int i = EI(&E, s), j = EJ(&E, s);
            Qo[X2(i, j)] = gamma2 * Qo[X2(i + E.di, j + E.dj)];
            if (msk) Qo[X2(i, j)] = Qo[X2(i, j)] * qmask[X2(i, j)];
            if (o->wet_dry) Qo[X2(i, j)] = Qo[X2(i, j)] * (grid == 'u' ? o->umask_wet : o->vmask_wet)[X2(i, j)];   /* u3dbc_im.F:523 */
          }
        }
        continue;
      }
      const double *fm = (msk && grid == 'r') ? (E.ti ? o->umask : o->vmask) : NULL;
      for (int s = E.s0; s <= E.s1; s++) {
        const int i = EI(&E, s), j = EJ(&E, s);
        double val;
        if (kind == ORC_LBC_RAD || kind == ORC_LBC_RADNUD) {
          double oin = obc_in, oout = obc_out;
          if (kind == ORC_LBC_RADNUD) {
            /* climatology nudging: the time scales of the condition from the nudging coefficient arrays (round 6) --
               u3dbc_im.F:113-118, :255-260, :397-402, :555-560 (the mean of the two rho points either side of the u point: across
               the edge at the western / eastern boundary, along it on the boundary row at the southern / northern one),
               v3dbc_im.F likewise, t3dbc_im.F:120-122 ... (the boundary point itself); obc_in = obcfac * obc_out */
            if (grid != 'r' && (o->clima_flags & 1)) {
              const double *cf = o->M3nudgcof + (size_t)(k - 1) * nij;
              oout = 0.5 * (cf[X2(i - (grid == 'u'), j - (grid == 'v'))] + cf[X2(i, j)]);
              oin = o->c.obcfac * oout;
            } else if (grid == 'r' && (o->clima_flags & (1 << itrc))) {
              oout = o->Tnudgcof[X2(i, j) + ((size_t)(itrc - 1) * N + (size_t)(k - 1)) * nij];
              oin = o->c.obcfac * oout;
            }
          }
          val = rad_point(o, &E, Qn, Qo, i, j, fm, kind == ORC_LBC_RADNUD, oin, oout, dt, bry[s - lb], 0);
        }
        else if (kind == ORC_LBC_CLA) val = bry[s - lb];
        else val = Qo[X2(i + E.di, j + E.dj)];                  /* gradient; tracers: closed too (t3dbc_im.F:205-218) */
        if (msk) val = val * qmask[X2(i, j)];
        /* WET_DRY: u3dbc_im.F:174,193,212 ... -- every open kind on every edge but u's gradient condition at the southern edge,
           which the reference guards with "WET_MASK" (u3dbc_im.F:496), a name nothing defines */
        if (o->wet_dry && grid != 'r' && !(grid == 'u' && E.e == ORC_ISOUTH && kind == ORC_LBC_GRA))
          val = val * (grid == 'u' ? o->umask_wet : o->vmask_wet)[X2(i, j)];
        Qo[X2(i, j)] = val;
      }
    }
    if (grid == 'u') corners_u(o, b, Qo);
    else if (grid == 'v') corners_v(o, b, Qo);
    else corners_r(o, b, Qo);
  }
}
/* tkebc_tile, tkebc_im.F:46-700: turbulent kinetic energy and length-scale variable at level nout, W-points 0..N --
   radiation (:98-186 ...: the scheme of t3dbc without nudging, differences of level nstp), zero gradient at gradient and
   closed edges (:188-232), corners :640-700 */
void orc_tkebc(const orc_t *o, const orc_bounds *b, int nout) {
  ORC_LOCALS(o);
  const int nstp = o->s.nstp;
  const int msk = (o->c.options & ORC_MASKING) != 0;
  static const int order[4] = { ORC_IWEST, ORC_IEAST, ORC_ISOUTH, ORC_INORTH };
  for (int f = 0; f < 2; f++) {
    double *A = f == 0 ? o->tke : o->gls;
    for (int k = 0; k <= N; k++) {
      double *Qo = A + ((size_t)(nout - 1) * (N + 1) + (size_t)k) * nij;
      const double *Qn = A + ((size_t)(nstp - 1) * (N + 1) + (size_t)k) * nij;
      for (int q = 0; q < 4; q++) {
        edge_t E;
        if (!edge_setup(o, b, order[q], 'r', &E)) continue;
        const int kind = o->c.lbc_tke[E.e];
        const double *fm = msk ? (E.ti ? o->umask : o->vmask) : NULL;
        for (int s = E.s0; s <= E.s1; s++) {
          const int i = EI(&E, s), j = EJ(&E, s);
          double val;
          if (kind == ORC_LBC_RAD) val = rad_point(o, &E, Qn, Qo, i, j, fm, 0, 0.0, 0.0, o->c.dt, 0.0, 0);
          else val = Qo[X2(i + E.di, j + E.dj)];
          if (msk) val = val * o->rmask[X2(i, j)];
          Qo[X2(i, j)] = val;
        }
      }
      corners_r(o, b, Qo);
    }
  }
}

void orc_u3dbc(const orc_t *o, const orc_bounds *b, int nout) { bc3d(o, b, nout, 'u', 0); }
void orc_v3dbc(const orc_t *o, const orc_bounds *b, int nout) { bc3d(o, b, nout, 'v', 0); }
void orc_t3dbc(const orc_t *o, const orc_bounds *b, int nout, int itrc) { bc3d(o, b, nout, 'r', itrc); }

/* the call sequences of oracle/ref/ref_glue.F90:ref_bc2d / ref_bc3d (routine-level pinning) */
void orc_bc2d(orc_t *o, int tile, int kout) {
  const orc_bounds *b = &o->b[tile];
  orc_zetabc(o, b, kout);
  orc_u2dbc(o, b, kout);
  orc_v2dbc(o, b, kout);
}
void orc_bc3d(orc_t *o, int tile, int nout) {
  const orc_bounds *b = &o->b[tile];
  for (int it = 1; it <= o->c.NT; it++) orc_t3dbc(o, b, nout, it);
  if (nout <= 2) {
    orc_u3dbc(o, b, nout);
    orc_v3dbc(o, b, nout);
  }
}
