// k_halo.h -- lateral boundary fills and periodic ghost copies on the GPU tile.
//
// Replaces, for closed/periodic edges (the LBC kinds of the BASELINE configs):
//   exchange_{p,r,u,v}2d_tile  ROMS/Nonlinear/exchange_2d.F:63-807
//   exchange_{r,u,v,w}3d_tile  ROMS/Nonlinear/exchange_3d.F:280-1126
//   bc_{r,u,v}2d_tile          ROMS/Nonlinear/bc_2d.F:41-516     bc_w3d_tile bc_3d.F:588
//   zetabc_tile zetabc.F:60 (closed :577-590), u2dbc_tile u2dbc_im.F:51, v2dbc_tile v2dbc_im.F:52,
//   t3dbc_tile t3dbc_im.F:50, u3dbc_tile u3dbc_im.F:50, v3dbc_tile v3dbc_im.F:50
//
// One thread block per horizontal plane; three barrier-separated phases
// (edge fills -> corner averages -> periodic copies), as the reference orders them.
#pragma once
#include "roms_ctx.h"


struct HaloItem {
  double *A;           // first plane
  int nk;              // number of consecutive planes
  int bc;              // BC_*
  int gtype;           // 'r','u','v','p' (exchange transverse ranges), 0 = no exchange
};
#define HALO_MAXITEMS 8
struct HaloArgs {
  DGrid G;
  int nitems;
  HaloItem it[HALO_MAXITEMS];
};

// the plane a block works on: item and plane of block bz, selected without indexing the
// kernel-argument array dynamically (a dynamic index makes the compiler copy the whole argument
// struct to scratch memory)
template <class ArgT>
KDEV double *halo_plane(const ArgT &a, int bz, int &bc, int &gtype) {
  double *A = nullptr;
  bc = BC_NONE; gtype = 0;
  int first = 0;
#pragma unroll
  for (int k = 0; k < HALO_MAXITEMS; k++) {
    const int nk = k < a.nitems ? a.it[k].nk : 0;
    if (bz >= first && bz < first + nk) {
      A = a.it[k].A + (size_t)(bz - first) * (size_t)a.G.nij;
      bc = a.it[k].bc;
      gtype = a.it[k].gtype;
    }
    first += nk;
  }
  return A;
}

// boundary fills and local periodic copies of one plane, by one thread block
KDEV void halo_fill(const DGrid &G, double *A, int bc, int gtype) {
  const TB &B = G.T;
  const int Istr = B.Istr, Iend = B.Iend, Jstr = B.Jstr, Jend = B.Jend;
  const int Lm = G.Lm, Mm = G.Mm;
  const double gamma2 = G.gamma2;
  // ---- phase 1: edges of closed boundaries
  if (bc == BC_R) {
    if (!G.ewp) {
      if (B.west) KLOOP1(j, Jstr, Jend) A[X2(Istr - 1, j)] = A[X2(Istr, j)];
      if (B.east) KLOOP1(j, Jstr, Jend) A[X2(Iend + 1, j)] = A[X2(Iend, j)];
    }
    if (!G.nsp) {
      if (B.south) KLOOP1(i, Istr, Iend) A[X2(i, Jstr - 1)] = A[X2(i, Jstr)];
      if (B.north) KLOOP1(i, Istr, Iend) A[X2(i, Jend + 1)] = A[X2(i, Jend)];
    }
  } else if (bc == BC_U) {
    if (!G.ewp) {
      if (B.west) KLOOP1(j, Jstr, Jend) A[X2(Istr, j)] = 0.0;
      if (B.east) KLOOP1(j, Jstr, Jend) A[X2(Iend + 1, j)] = 0.0;
    }
    KSYNC();
    if (!G.nsp) {
      const int Imin = G.ewp ? B.IstrU : B.Istr, Imax = G.ewp ? B.Iend : B.IendR;
      if (B.south) KLOOP1(i, Imin, Imax) A[X2(i, Jstr - 1)] = gamma2 * A[X2(i, Jstr)];
      if (B.north) KLOOP1(i, Imin, Imax) A[X2(i, Jend + 1)] = gamma2 * A[X2(i, Jend)];
    }
  } else if (bc == BC_V) {
    if (!G.ewp) {
      const int Jmin = G.nsp ? B.JstrV : B.Jstr, Jmax = G.nsp ? B.Jend : B.JendR;
      if (B.west) KLOOP1(j, Jmin, Jmax) A[X2(Istr - 1, j)] = gamma2 * A[X2(Istr, j)];
      if (B.east) KLOOP1(j, Jmin, Jmax) A[X2(Iend + 1, j)] = gamma2 * A[X2(Iend, j)];
    }
    KSYNC();
    if (!G.nsp) {
      if (B.south) KLOOP1(i, Istr, Iend) A[X2(i, Jstr)] = 0.0;
      if (B.north) KLOOP1(i, Istr, Iend) A[X2(i, Jend + 1)] = 0.0;
    }
  }
  KSYNC();
  // ---- phase 2: corners (only when neither direction is periodic)
  if (bc != BC_NONE && !(G.ewp || G.nsp) && KTID == 0) {
    if (bc == BC_R) {
      if (B.sw) A[X2(Istr - 1, Jstr - 1)] = 0.5 * (A[X2(Istr, Jstr - 1)] + A[X2(Istr - 1, Jstr)]);
      if (B.se) A[X2(Iend + 1, Jstr - 1)] = 0.5 * (A[X2(Iend, Jstr - 1)] + A[X2(Iend + 1, Jstr)]);
      if (B.nw) A[X2(Istr - 1, Jend + 1)] = 0.5 * (A[X2(Istr - 1, Jend)] + A[X2(Istr, Jend + 1)]);
      if (B.ne) A[X2(Iend + 1, Jend + 1)] = 0.5 * (A[X2(Iend + 1, Jend)] + A[X2(Iend, Jend + 1)]);
    } else if (bc == BC_U) {
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
  KSYNC();
  // ---- phase 3: periodic ghost copies (single tile in the periodic direction)
  if (gtype != 0 && (G.ewp || G.nsp)) {
    const int gt = gtype;
    int Jmin, Jmax, Imin, Imax;
    if (G.nsp) { Jmin = B.Jstr; Jmax = B.Jend; }
    else { Jmin = (gt == 'r' || gt == 'u') ? B.JstrR : B.Jstr; Jmax = B.JendR; }
    if (G.ewp) { Imin = B.Istr; Imax = B.Iend; }
    else { Imin = (gt == 'r' || gt == 'v') ? B.IstrR : B.Istr; Imax = B.IendR; }
    const int ng3 = G.Nghost == 3;
    if (G.ewp && B.west && B.east && G.xloc) {
      KLOOP1(j, Jmin, Jmax) {
        A[X2(Lm + 1, j)] = A[X2(1, j)];
        A[X2(Lm + 2, j)] = A[X2(2, j)];
        if (ng3) A[X2(Lm + 3, j)] = A[X2(3, j)];
        A[X2(-2, j)] = A[X2(Lm - 2, j)];
        A[X2(-1, j)] = A[X2(Lm - 1, j)];
        A[X2(0, j)] = A[X2(Lm, j)];
      }
    }
    if (G.nsp && B.south && B.north && G.yloc) {
      KLOOP1(i, Imin, Imax) {
        A[X2(i, Mm + 1)] = A[X2(i, 1)];
        A[X2(i, Mm + 2)] = A[X2(i, 2)];
        if (ng3) A[X2(i, Mm + 3)] = A[X2(i, 3)];
        A[X2(i, -2)] = A[X2(i, Mm - 2)];
        A[X2(i, -1)] = A[X2(i, Mm - 1)];
        A[X2(i, 0)] = A[X2(i, Mm)];
      }
    }
    if (G.ewp && G.nsp && B.sw && B.ne && G.xloc && G.yloc && KTID == 0) {
      const int ne = ng3 ? 3 : 2;
      for (int dj = 1; dj <= ne; dj++)
        for (int di = 1; di <= ne; di++) A[X2(Lm + di, Mm + dj)] = A[X2(di, dj)];
      for (int dj = 1; dj <= ne; dj++)
        for (int di = -2; di <= 0; di++) A[X2(di, Mm + dj)] = A[X2(Lm + di, dj)];
      for (int dj = -2; dj <= 0; dj++)
        for (int di = 1; di <= ne; di++) A[X2(Lm + di, dj)] = A[X2(di, Mm + dj)];
      for (int dj = -2; dj <= 0; dj++)
        for (int di = -2; di <= 0; di++) A[X2(di, dj)] = A[X2(Lm + di, Mm + dj)];
    }
  }
}

COOP_KERNEL(halo_kernel, HaloArgs) {
  (void)bx; (void)by; (void)lds;
  int bc, gtype;
  double *A = halo_plane(a, bz, bc, gtype);
  halo_fill(a.G, A, bc, gtype);
}
COOP_GLOBAL(halo_kernel, HaloArgs)

// ------------------------------------------------------------------------------------------
// Inter-tile halo strips (multi-GPU): the data movement of mp_exchange2d/3d/4d
// (ROMS/Utility/mp_exchange.F:28-2300) with the strip geometry of the periodic copies above --
// a tile's three west ghost columns {Istr-3..Istr-1} come from its west neighbour's last three
// interior columns {Iend-2..Iend}; its Nghost east ghost columns {Iend+1..Iend+Nghost} from the
// east neighbour's first Nghost interior columns; the same along eta.
//
// The reference moves full-height xi strips, then full-width eta strips which carry the corners
// (two dependent message phases).  Here ONE message phase gives the same ghost zone: full-height
// xi strips, full-width eta strips and the corner blocks from the four diagonal neighbours go out
// together; the receiver unpacks xi, then eta, then the corners, so the corner cells end up with
// the diagonal tile's interior values exactly as after the reference's second phase.  The
// boundary fills of the exchange point run in the pack launch (same block, same plane), so an
// exchange point costs two launches and one send/recv group.
//
// Directions d: 0 W, 1 E, 2 S, 3 N, 4 SW, 5 SE, 6 NW, 7 NE.
// Message layout: [plane][row][column] of the rectangle, columns fastest.
// ------------------------------------------------------------------------------------------
struct XchgArgs {
  DGrid G;
  int nitems;
  HaloItem it[HALO_MAXITEMS];
  int unpack;          // 0: (boundary fills and) pack own lines into send buffers, 1: unpack into ghosts
  int fill;            // pack launch: run halo_fill first
  double *buf[8];      // pack: message for neighbour d ; unpack: message received from neighbour d (null: none)
};

// rectangle of direction d: source lines of the message sent to neighbour d (unpack = 0) or ghost
// lines filled by the message from neighbour d (unpack = 1)
KDEV void xchg_rect(const DGrid &G, int d, int unpack, int &i0, int &i1, int &j0, int &j1) {
  const TB &B = G.T;
  const int ng = G.Nghost;
  const int dx = (d == 0 || d == 4 || d == 6) ? -1 : ((d == 1 || d == 5 || d == 7) ? 1 : 0);
  const int dy = (d == 2 || d == 4 || d == 5) ? -1 : ((d == 3 || d == 6 || d == 7) ? 1 : 0);
  if (dx == 0) { i0 = G.LBi; i1 = G.LBi + G.ni - 1; }
  else if (dx < 0) { i0 = unpack ? B.Istr - 3 : B.Istr; i1 = unpack ? B.Istr - 1 : B.Istr + ng - 1; }
  else { i0 = unpack ? B.Iend + 1 : B.Iend - 2; i1 = unpack ? B.Iend + ng : B.Iend; }
  if (dy == 0) { j0 = G.LBj; j1 = G.LBj + G.nj - 1; }
  else if (dy < 0) { j0 = unpack ? B.Jstr - 3 : B.Jstr; j1 = unpack ? B.Jstr - 1 : B.Jstr + ng - 1; }
  else { j0 = unpack ? B.Jend + 1 : B.Jend - 2; j1 = unpack ? B.Jend + ng : B.Jend; }
}

KDEV void xchg_move(const DGrid &G, double *A, double *msg, int bz, int d, int unpack) {
  if (!msg) return;
  int i0, i1, j0, j1;
  xchg_rect(G, d, unpack, i0, i1, j0, j1);
  const int w = i1 - i0 + 1;
  double *m = msg + (size_t)bz * (size_t)w * (size_t)(j1 - j0 + 1);
  if (unpack) KLOOP2(i, j, i0, i1, j0, j1) A[X2(i, j)] = m[(j - j0) * w + (i - i0)];
  else KLOOP2(i, j, i0, i1, j0, j1) m[(j - j0) * w + (i - i0)] = A[X2(i, j)];
}

COOP_KERNEL(xchg_kernel, XchgArgs) {
  (void)bx; (void)by; (void)lds;
  const DGrid &G = a.G;
  int bc, gtype;
  double *A = halo_plane(a, bz, bc, gtype);
  if (!a.unpack) {
    if (a.fill) { halo_fill(G, A, bc, gtype); KSYNC(); }
#pragma unroll
    for (int d = 0; d < 8; d++) xchg_move(G, A, a.buf[d], bz, d, 0);
  } else {
    xchg_move(G, A, a.buf[0], bz, 0, 1);
    xchg_move(G, A, a.buf[1], bz, 1, 1);
    KSYNC();
    xchg_move(G, A, a.buf[2], bz, 2, 1);
    xchg_move(G, A, a.buf[3], bz, 3, 1);
    KSYNC();
#pragma unroll
    for (int d = 4; d < 8; d++) xchg_move(G, A, a.buf[d], bz, d, 1);
  }
}
COOP_GLOBAL(xchg_kernel, XchgArgs)
