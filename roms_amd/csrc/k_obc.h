// k_obc.h -- lateral boundary conditions of the state where an edge is open.
//
//   k_obc    zetabc_tile  ROMS/Nonlinear/zetabc.F:60-650      radiation (+nudging), Chapman explicit / implicit,
//                                                            clamped, gradient, closed
//            u2dbc_tile   ROMS/Nonlinear/u2dbc_im.F:51-1312   radiation (+nudging), Flather, Shchepetkin, clamped,
//            v2dbc_tile   ROMS/Nonlinear/v2dbc_im.F:52-1362   gradient, closed; Chapman for the tangential component
//            u3dbc_tile   ROMS/Nonlinear/u3dbc_im.F:50-737    radiation (+nudging), clamped, gradient, closed
//            v3dbc_tile   ROMS/Nonlinear/v3dbc_im.F:50-737
//            t3dbc_tile   ROMS/Nonlinear/t3dbc_im.F:50-684
//
// A context with any open edge (DGrid::obc) applies ALL boundary conditions of the state variables here -- the closed
// edges of the same variable included -- and its halo launches keep the periodic copies / tile exchanges only
// (g_obc.cpp, roms_host.h:obc_bc).  One thread block per horizontal plane, as the halo kernel: the threads run along the
// edges (a few hundred to a few thousand points, each an independent expression), a barrier, the corner means.
// The boundary values of a closed basin stay where they were, fused into the producing kernels; nothing here is on
// the path of the BASELINE configurations.
//
// An edge is described by its inward normal n = (di,dj) and the direction (ti,tj) the loop index s runs along it; B is
// the boundary point, I1 = B + n, I2 = B + 2n.  With lower(L) = Q(L,s) - Q(L,s-1), upper(L) = Q(L,s+1) - Q(L,s) at the
// "now" level on the lines L = B and L = I1 (the reference's `grad` pairs) the radiation condition of every variable
// and edge is one expression (obc_radiate), written with the operand order of the reference's statements.  The
// reference's southern free-surface branch departs from the pattern in two places (zetabc.F:455, :486-487); `inner`
// reproduces it.
#pragma once
#include "roms_ctx.h"
#include "k_halo.h"          // wet_tail2

struct ObcItem {
  double *Qo;            // first plane of the level being set (kout / nout)
  const double *Qn;      // first plane of the "now" level (know / nstp)
  int nk;                // planes
  int grid;              // 'r', 'u', 'v'
  int is2d;              // barotropic variable: Chapman / Flather / Shchepetkin available
  int kind[4];           // ROMS_LBC_* per edge (ROMS_IWEST ...), defaults resolved
  double obc_in[4], obc_out[4];
  const double *bry[4];  // boundary data of plane 0 per edge
  long bstride[4];       // doubles from one plane's line to the next
  // climatology nudging (round 6): the radiation + nudging conditions take their time scales from the nudging coefficient array of the
  // variable -- plane 0 of "M2nudgcof" | "M3nudgcof" | "Tnudgcof"(itrc), rho points -- instead of obc_in / obc_out: the boundary point's
  // own value for a tracer (t3dbc_im.F:120-122), the mean of the two rho points either side of a velocity point (u3dbc_im.F:113-118,
  // u2dbc_im.F:158-162 ...), inflow = obcfac x that.  Null: the scales above
  const double *cof;
  double obcfac;
};
#define OBC_MAXITEMS 3
struct ObcArgs {
  DGrid G;
  int nitems;
  ObcItem it[OBC_MAXITEMS];
  double dtn;            // dt2d of the barotropic conditions (zetabc.F:100-112), dt of the 3-D ones
  // what the barotropic conditions read besides their own variable
  const double *zeta_n, *zeta_o, *zbry[4], *h, *pm, *pn;
};

struct ObcEdge { int e, di, dj, ti, tj, i0, j0, s0, s1, normal; };

KDEV bool obc_edge(const DGrid &G, int e, int grid, ObcEdge &E) {
  const TB &B = G.T;
  const bool we = e == ROMS_IWEST || e == ROMS_IEAST;
  if (we ? G.ewp : G.nsp) return false;
  if (!(e == ROMS_IWEST ? B.west : e == ROMS_IEAST ? B.east : e == ROMS_ISOUTH ? B.south : B.north)) return false;
  E.e = e;
  E.di = e == ROMS_IWEST ? 1 : e == ROMS_IEAST ? -1 : 0;
  E.dj = e == ROMS_ISOUTH ? 1 : e == ROMS_INORTH ? -1 : 0;
  E.ti = we ? 0 : 1;
  E.tj = we ? 1 : 0;
  E.normal = (we && grid == 'u') || (!we && grid == 'v');
  if (we) {
    E.i0 = e == ROMS_IWEST ? (grid == 'u' ? B.Istr : B.Istr - 1) : B.Iend + 1;
    E.j0 = 0;
    E.s0 = grid == 'v' ? B.JstrV : B.Jstr;
    E.s1 = B.Jend;
  } else {
    E.j0 = e == ROMS_ISOUTH ? (grid == 'v' ? B.Jstr : B.Jstr - 1) : B.Jend + 1;
    E.i0 = 0;
    E.s0 = grid == 'u' ? B.IstrU : B.Istr;
    E.s1 = B.Iend;
  }
  return true;
}

// implicit upstream radiation (+ nudging) at boundary point (i,j): zetabc.F:119-184, u2dbc_im.F:145-220,846-925,
// u3dbc_im.F:99-182, t3dbc_im.F:98-175
KDEV double obc_radiate(const DGrid &G, const ObcEdge &E, const double *Qn, const double *Qo, int i, int j, const double *fm,
                        bool nudging, double obc_in, double obc_out, double dtn, double bry, bool inner) {
  const int di = E.di, dj = E.dj, ti = E.ti, tj = E.tj;
  const int i1 = i + di, j1 = j + dj;
  const double qB = Qn[X2(i, j)], qI = Qn[X2(i1, j1)];
  double gBl = qB - Qn[X2(i - ti, j - tj)], gBu = Qn[X2(i + ti, j + tj)] - qB;
  double gIl = qI - Qn[X2(i1 - ti, j1 - tj)], gIu = Qn[X2(i1 + ti, j1 + tj)] - qI;
  if (fm) {      // MASKING, rho-type variables: each difference times the mask of the face it spans (zetabc.F:124-132)
    gBl = gBl * fm[X2(i, j)];
    gBu = gBu * fm[X2(i + ti, j + tj)];
    gIl = gIl * fm[X2(i1, j1)];
    gIu = gIu * fm[X2(i1 + ti, j1 + tj)];
  }
  if (inner) { gBl = gIl; gBu = gIu; }
  const double oI = Qo[X2(i1, j1)];
  double dQdt = qI - oI;
  const double dQdn = inner ? oI - Qo[X2(i, j)] : oI - Qo[X2(i1 + di, j1 + dj)];
  double tau = 0.0;
  if (nudging) {
    tau = (dQdt * dQdn) < 0.0 ? obc_in : obc_out;
    tau = tau * dtn;
  }
  if ((dQdt * dQdn) < 0.0) dQdt = 0.0;
  const double dQds = (dQdt * (gIl + gIu)) > 0.0 ? gIl : gIu;
  const double cff = KMAX(dQdn * dQdn + dQds * dQds, 1.0E-20);
  const double Cn = dQdt * dQdn;
  const double Ct = (G.options & ROMS_RADIATION_2D) ? KMIN(cff, KMAX(dQdt * dQds, -cff)) : 0.0;
  double val = (cff * qB + Cn * oI - KMAX(Ct, 0.0) * gBl - KMIN(Ct, 0.0) * gBu) / (cff + Cn);
  if (nudging) val = val + tau * (bry - qB);
  return val;
}

// the conditions of one edge of one plane
KDEV void obc_edge_fill(const ObcArgs &a, const ObcItem &it, const ObcEdge &E, double *Qo, const double *Qn, int plane) {
  const DGrid &G = a.G;
  const TB &B = G.T;
  const int kind = it.kind[E.e], grid = it.grid;
  const bool msk = G.masking != 0;
  const double *qmask = grid == 'u' ? G.umask : grid == 'v' ? G.vmask : G.rmask;
  const double g = G.g, gamma2 = G.gamma2, dtn = a.dtn;
  if (kind == ROMS_LBC_CLO && grid != 'r') {
    if (E.normal) {
      KLOOP1(s, E.s0, E.s1) Qo[X2(E.i0 + E.ti * s, E.j0 + E.tj * s)] = 0.0;
    } else {           // free slip / no slip, over the closed condition's own range (u2dbc_im.F:966-985)
      int s0, s1;
      if (E.ti) { s0 = G.ewp ? B.IstrU : B.Istr; s1 = G.ewp ? B.Iend : B.IendR; }
      else { s0 = G.nsp ? B.JstrV : B.Jstr; s1 = G.nsp ? B.Jend : B.JendR; }
      KLOOP1(s, s0, s1) {
        const int i = E.i0 + E.ti * s, j = E.j0 + E.tj * s;
        double v = gamma2 * Qo[X2(i + E.di, j + E.dj)];
        if (msk) v = v * qmask[X2(i, j)];
        if (G.wet_dry && !it.is2d) v = v * (grid == 'u' ? G.umask_wet : G.vmask_wet)[X2(i, j)];      // WET_DRY u3dbc_im.F:523,681
        Qo[X2(i, j)] = v;
      }
    }
    return;
  }
  const double *bry = it.bry[E.e] ? it.bry[E.e] + (size_t)plane * (size_t)it.bstride[E.e] : nullptr;
  const int lb = E.ti ? G.LBi : G.LBj;
  const bool rad = kind == ROMS_LBC_RAD || kind == ROMS_LBC_RADNUD;
  const double *fm = (msk && grid == 'r') ? (E.ti ? G.umask : G.vmask) : nullptr;
  const double *pmn = E.ti ? a.pn : a.pm;             // the metric across the edge
  const bool low = E.e == ROMS_IWEST || E.e == ROMS_ISOUTH;
  const double Co = 1.0 / (2.0 + sqrt(2.0));          // mod_scalars.F:4435
  KLOOP1(s, E.s0, E.s1) {
    const int i = E.i0 + E.ti * s, j = E.j0 + E.tj * s, i1 = i + E.di, j1 = j + E.dj;
    const double bv = bry ? bry[s - lb] : 0.0;
    double val;
    if (rad) {
      double oin = it.obc_in[E.e], oout = it.obc_out[E.e];
      if (kind == ROMS_LBC_RADNUD && it.cof) {
        const double *cf = it.cof + (size_t)plane * (size_t)G.nij;
        oout = grid == 'r' ? cf[X2(i, j)] : 0.5 * (cf[X2(i - (grid == 'u' ? 1 : 0), j - (grid == 'v' ? 1 : 0))] + cf[X2(i, j)]);
        oin = it.obcfac * oout;
      }
      val = obc_radiate(G, E, Qn, Qo, i, j, fm, kind == ROMS_LBC_RADNUD, oin, oout, dtn, bv,
                        it.is2d && grid == 'r' && E.e == ROMS_ISOUTH);
    } else if (kind == ROMS_LBC_CLA) {
      val = bv;
    } else if (kind == ROMS_LBC_CHE || kind == ROMS_LBC_CHI) {       // free surface, zetabc.F:186-227
      const double cff = dtn * pmn[X2(i1, j1)];
      const double dep = a.h[X2(i1, j1)] + Qn[X2(i1, j1)];
      const double cff1 = sqrt(g * (G.wet_dry ? (dep > G.Dcrit ? dep : G.Dcrit) : dep));            // WET_DRY: MAX(..., Dcrit) zetabc.F:190-192
      const double Cx = cff * cff1;
      if (kind == ROMS_LBC_CHE) val = (1.0 - Cx) * Qn[X2(i, j)] + Cx * Qn[X2(i1, j1)];
      else { const double cff2 = 1.0 / (1.0 + Cx); val = cff2 * (Qn[X2(i, j)] + Cx * Qo[X2(i1, j1)]); }
    } else if ((kind == ROMS_LBC_FLA || kind == ROMS_LBC_SHC) && E.normal) {
      // the two rho points either side of the boundary velocity point, in index order
      const size_t lo = X2(i - (grid == 'u' ? 1 : 0), j - (grid == 'u' ? 0 : 1)), hi = X2(i, j);
      const size_t in = low ? hi : lo, out = low ? lo : hi;
      const double *Zn = a.zeta_n, *Zo = a.zeta_o;
      const double zb = a.zbry[E.e][s - lb];
      if (kind == ROMS_LBC_FLA) {                                    // u2dbc_im.F:224-291, :575-642
        const double cff = 1.0 / (0.5 * (a.h[lo] + Zn[lo] + a.h[hi] + Zn[hi]));
        const double Cx = sqrt(g * cff);
        const double d = Cx * (0.5 * (Zn[lo] + Zn[hi]) - zb);
        val = low ? bv - d : bv + d;
      } else {                                                       // Shchepetkin :296-369, :647-720
        const double cff = G.wet_dry ? 0.5 * (a.h[lo] + Zn[lo] + a.h[hi] + Zn[hi]) : 0.5 * (a.h[lo] + a.h[hi]);   // WET_DRY u2dbc_im.F:339-347
        const double cff1 = sqrt(g / cff);
        const double Cx = dtn * cff1 * cff * 0.5 * (pmn[lo] + pmn[hi]);
        double Zx = (0.5 + Cx) * Zn[in] + (0.5 - Cx) * Zn[out];
        if (Cx > Co) {
          const double r = 1.0 - Co / Cx;
          const double cff2 = r * r;
          const double cff3 = Zo[in] + Cx * Zn[out] - (1.0 + Cx) * Zn[in];
          Zx = Zx + cff2 * cff3;
        }
        const double d = cff1 * (Zx - zb);
        val = low ? 0.5 * ((1.0 - Cx) * Qn[X2(i, j)] + Cx * Qn[X2(i1, j1)] + bv - d)
                  : 0.5 * ((1.0 - Cx) * Qn[X2(i, j)] + Cx * Qn[X2(i1, j1)] + bv + d);
      }
    } else if (kind == ROMS_LBC_FLA || kind == ROMS_LBC_SHC) {
      // tangential component under Flather / Shchepetkin: Chapman with the mean depth of the two rho points of the first
      // interior line either side of the velocity point, u2dbc_im.F:921-943
      const size_t p = X2(i1 - E.ti, j1 - E.tj), q = X2(i1, j1);
      const double cff = dtn * 0.5 * (pmn[p] + pmn[q]);
      const double cff1 = sqrt(g * 0.5 * (a.h[p] + a.zeta_n[p] + a.h[q] + a.zeta_n[q]));
      const double Ce = cff * cff1;
      const double cff2 = 1.0 / (1.0 + Ce);
      val = cff2 * (Qn[X2(i, j)] + Ce * Qo[X2(i1, j1)]);
    } else {                                                         // gradient; rho-type: closed too
      val = Qo[X2(i1, j1)];
    }
    if (msk) val = val * qmask[X2(i, j)];
    // WET_DRY: every open kind of u3dbc / v3dbc times the wet mask of the boundary point (u3dbc_im.F:174,193,212 ...), but u's
    // gradient condition at the southern edge: the reference guards that one with "WET_MASK" (u3dbc_im.F:496), a name nothing defines
    if (G.wet_dry && !it.is2d && grid != 'r' && !(grid == 'u' && E.e == ROMS_ISOUTH && kind == ROMS_LBC_GRA))
      val = val * (grid == 'u' ? G.umask_wet : G.vmask_wet)[X2(i, j)];
    Qo[X2(i, j)] = val;
  }
}

COOP_KERNEL(k_obc, ObcArgs) {
  (void)bx; (void)by; (void)lds;
  const DGrid &G = a.G;
  const TB &B = G.T;
  // item and plane of this block (no dynamic index into the argument struct: halo_plane, k_halo.h)
  int first = 0;
#pragma unroll
  for (int q = 0; q < OBC_MAXITEMS; q++) {
    const int nk = q < a.nitems ? a.it[q].nk : 0;
    if (bz >= first && bz < first + nk) {
      const ObcItem &it = a.it[q];
      const int plane = bz - first;
      double *Qo = it.Qo + (size_t)plane * (size_t)G.nij;
      const double *Qn = it.Qn + (size_t)plane * (size_t)G.nij;
      // the reference's edge order: west, east, south, north (v: south, north, west, east); the pairs are independent
      const int e0 = it.grid == 'v' ? ROMS_ISOUTH : ROMS_IWEST, e1 = it.grid == 'v' ? ROMS_INORTH : ROMS_IEAST;
      const int e2 = it.grid == 'v' ? ROMS_IWEST : ROMS_ISOUTH, e3 = it.grid == 'v' ? ROMS_IEAST : ROMS_INORTH;
      ObcEdge E;
      if (obc_edge(G, e0, it.grid, E)) obc_edge_fill(a, it, E, Qo, Qn, plane);
      if (obc_edge(G, e1, it.grid, E)) obc_edge_fill(a, it, E, Qo, Qn, plane);
      KSYNC();
      if (obc_edge(G, e2, it.grid, E)) obc_edge_fill(a, it, E, Qo, Qn, plane);
      if (obc_edge(G, e3, it.grid, E)) obc_edge_fill(a, it, E, Qo, Qn, plane);
      KSYNC();
      // corners: the mean of the two neighbouring boundary values, where neither direction is periodic
      if (!(G.ewp || G.nsp) && KTID == 0) {
        const int Istr = B.Istr, Iend = B.Iend, Jstr = B.Jstr, Jend = B.Jend;
        double *A = Qo;
        if (it.grid == 'r') {
          if (B.sw) A[X2(Istr - 1, Jstr - 1)] = 0.5 * (A[X2(Istr, Jstr - 1)] + A[X2(Istr - 1, Jstr)]);
          if (B.se) A[X2(Iend + 1, Jstr - 1)] = 0.5 * (A[X2(Iend, Jstr - 1)] + A[X2(Iend + 1, Jstr)]);
          if (B.nw) A[X2(Istr - 1, Jend + 1)] = 0.5 * (A[X2(Istr - 1, Jend)] + A[X2(Istr, Jend + 1)]);
          if (B.ne) A[X2(Iend + 1, Jend + 1)] = 0.5 * (A[X2(Iend + 1, Jend)] + A[X2(Iend, Jend + 1)]);
        } else if (it.grid == 'u') {
          if (B.sw) A[X2(Istr, Jstr - 1)] = 0.5 * (A[X2(Istr + 1, Jstr - 1)] + A[X2(Istr, Jstr)]);
          if (B.se) A[X2(Iend + 1, Jstr - 1)] = 0.5 * (A[X2(Iend, Jstr - 1)] + A[X2(Iend + 1, Jstr)]);
          if (B.nw) A[X2(Istr, Jend + 1)] = 0.5 * (A[X2(Istr, Jend)] + A[X2(Istr + 1, Jend + 1)]);
          if (B.ne) A[X2(Iend + 1, Jend + 1)] = 0.5 * (A[X2(Iend + 1, Jend)] + A[X2(Iend, Jend + 1)]);
        } else {
          if (B.sw) A[X2(Istr - 1, Jstr)] = 0.5 * (A[X2(Istr, Jstr)] + A[X2(Istr - 1, Jstr + 1)]);
          if (B.se) A[X2(Iend + 1, Jstr)] = 0.5 * (A[X2(Iend, Jstr)] + A[X2(Iend + 1, Jstr + 1)]);
          if (B.nw) A[X2(Istr - 1, Jend + 1)] = 0.5 * (A[X2(Istr - 1, Jend)] + A[X2(Istr, Jend + 1)]);
          if (B.ne) A[X2(Iend + 1, Jend + 1)] = 0.5 * (A[X2(Iend + 1, Jend)] + A[X2(Iend, Jend + 1)]);
        }
      }
      if (G.wet_dry && it.is2d) {       // WET_DRY: the wetting/drying conditions at the end of zetabc / u2dbc / v2dbc (k_halo.h)
        KSYNC();
        wet_tail2(G, Qo, it.grid == 'r' ? BC_R : it.grid == 'u' ? BC_U : BC_V);
      }
    }
    first += nk;
  }
}
COOP_GLOBAL(k_obc, ObcArgs)

// ---- VolCons: obc_flux_tile (obc_volcons.F:60-233; step2d_LF_AM3.h:2885, behind v2dbc of every barotropic call) -----------------
// Cross-section and mass flux of the open edges that conserve volume, summed in the reference's order -- west, east, south,
// north, each along ascending index, one running sum each -- and the correction velocity ubar_xs = bc_flux / bc_area that the
// next call's mass fluxes take off the inflow (k_step2d.h: set_DUV_bc_tile).  One block: the threads form the terms of a chunk
// in LDS, one thread adds them up in order (the order is the result: a tree would round differently).  One tile (the sum over
// tiles is a reduction across ranks the library has not got: roms_hip_create refuses).
struct ObcFluxArgs {
  DGrid G;
  const double *zeta, *ubar, *vbar;   // level knew
  const double *h, *on_u, *om_v;
};
#define OBCF_CHUNK 1024
COOP_KERNEL(k_obc_flux, ObcFluxArgs) {
  (void)bx; (void)by; (void)bz;
  const DGrid &G = a.G;
  const TB &B = G.T;
  const int nw = ((G.volcons & (1 << ROMS_IWEST)) && B.west) ? B.Jend - B.Jstr + 1 : 0;
  const int ne = ((G.volcons & (1 << ROMS_IEAST)) && B.east) ? B.Jend - B.Jstr + 1 : 0;
  const int ns = ((G.volcons & (1 << ROMS_ISOUTH)) && B.south) ? B.Iend - B.Istr + 1 : 0;
  const int nn = ((G.volcons & (1 << ROMS_INORTH)) && B.north) ? B.Iend - B.Istr + 1 : 0;
  const int ntot = nw + ne + ns + nn;
  double *sA = lds, *sF = lds + OBCF_CHUNK;
  double my_area = 0.0, my_flux = 0.0;
  for (int c0 = 0; c0 < ntot; c0 += OBCF_CHUNK) {
    const int c1 = KMIN(ntot, c0 + OBCF_CHUNK) - 1;
    KLOOP1(s, c0, c1) {
      double cff, p;
      if (s < nw) {
        const int i = B.Istr, j = B.Jstr + s;
        cff = 0.5 * (a.zeta[X2(i - 1, j)] + a.h[X2(i - 1, j)] + a.zeta[X2(i, j)] + a.h[X2(i, j)]) * a.on_u[X2(i, j)];
        if (G.masking) cff = cff * G.umask[X2(i, j)];
        p = cff * a.ubar[X2(i, j)];
      } else if (s < nw + ne) {
        const int i = B.Iend, j = B.Jstr + (s - nw);
        cff = 0.5 * (a.zeta[X2(i, j)] + a.h[X2(i, j)] + a.zeta[X2(i + 1, j)] + a.h[X2(i + 1, j)]) * a.on_u[X2(i + 1, j)];
        if (G.masking) cff = cff * G.umask[X2(i + 1, j)];
        p = -(cff * a.ubar[X2(i + 1, j)]);              // (my_flux - x == my_flux + (-x), bit for bit)
      } else if (s < nw + ne + ns) {
        const int i = B.Istr + (s - nw - ne), j = B.Jstr;
        cff = 0.5 * (a.zeta[X2(i, j - 1)] + a.h[X2(i, j - 1)] + a.zeta[X2(i, j)] + a.h[X2(i, j)]) * a.om_v[X2(i, j)];
        if (G.masking) cff = cff * G.vmask[X2(i, j)];
        p = cff * a.vbar[X2(i, B.JstrV - 1)];           // (obc_volcons.F:168 reads vbar(i,JstrV-1))
      } else {
        const int i = B.Istr + (s - nw - ne - ns), j = B.Jend;
        cff = 0.5 * (a.zeta[X2(i, j)] + a.h[X2(i, j)] + a.zeta[X2(i, j + 1)] + a.h[X2(i, j + 1)]) * a.om_v[X2(i, j + 1)];
        if (G.masking) cff = cff * G.vmask[X2(i, j + 1)];
        p = -(cff * a.vbar[X2(i, j + 1)]);
      }
      sA[s - c0] = cff;
      sF[s - c0] = p;
    }
    KSYNC();
    if (KTID == 0)
      for (int s = 0; s <= c1 - c0; s++) { my_area = my_area + sA[s]; my_flux = my_flux + sF[s]; }
    KSYNC();
  }
  if (KTID == 0) {
    const double bc_area = 0.0 + my_area, bc_flux = 0.0 + my_flux;      // (:203-208 with tile_count = 0 and one tile)
    G.vcons[0] = bc_area;
    G.vcons[1] = bc_flux;
    G.vcons[2] = bc_flux / bc_area;
  }
}
COOP_GLOBAL(k_obc_flux, ObcFluxArgs)
