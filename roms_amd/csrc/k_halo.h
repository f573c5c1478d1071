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

COOP_KERNEL(halo_kernel, HaloArgs) {
  (void)bx; (void)by; (void)lds;
  const DGrid &G = a.G;
  const TB &B = G.T;
  // select this block's item without indexing the kernel-argument array dynamically (a dynamic
  // index makes the compiler copy the whole argument struct to scratch memory)
  double *A = nullptr;
  int bc = BC_NONE, gtype = 0;
  {
    int first = 0;
#pragma unroll
    for (int k = 0; k < HALO_MAXITEMS; k++) {
      const int nk = k < a.nitems ? a.it[k].nk : 0;
      if (bz >= first && bz < first + nk) {
        A = a.it[k].A + (size_t)(bz - first) * (size_t)G.nij;
        bc = a.it[k].bc;
        gtype = a.it[k].gtype;
      }
      first += nk;
    }
  }
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
COOP_GLOBAL(halo_kernel, HaloArgs)

// ------------------------------------------------------------------------------------------
// Inter-tile halo strips (multi-GPU): the data movement of mp_exchange2d/3d/4d
// (ROMS/Utility/mp_exchange.F:28-2300) with the strip geometry of the periodic copies above --
// a tile's three west ghost columns {Istr-3..Istr-1} come from its west neighbour's last three
// interior columns {Iend-2..Iend}; its Nghost east ghost columns {Iend+1..Iend+Nghost} from the
// east neighbour's first Nghost interior columns; the same along eta.  Phase 0 moves full-height
// xi strips, phase 1 full-width eta strips (which then carry the corners), as in the reference.
//
// Message layout: [plane][line][c], c = position across the strip.
// ------------------------------------------------------------------------------------------
struct StripArgs {
  DGrid G;
  int nitems;
  HaloItem it[HALO_MAXITEMS];
  int phase;           // 0: xi (west/east neighbours), 1: eta (south/north)
  int unpack;          // 0: pack own interior columns into send buffers, 1: unpack into ghosts
  double *lo;          // pack: strip sent to the west/south neighbour ; unpack: received from it
  double *hi;          // pack: strip sent to the east/north neighbour ; unpack: received from it
};

COOP_KERNEL(strip_kernel, StripArgs) {
  (void)bx; (void)by; (void)lds;
  const DGrid &G = a.G;
  const TB &B = G.T;
  double *A = nullptr;
  {
    int first = 0;
#pragma unroll
    for (int k = 0; k < HALO_MAXITEMS; k++) {
      const int nk = k < a.nitems ? a.it[k].nk : 0;
      if (bz >= first && bz < first + nk) A = a.it[k].A + (size_t)(bz - first) * (size_t)G.nij;
      first += nk;
    }
  }
  const int ng = G.Nghost;
  if (a.phase == 0) {
    const int nl = G.nj;                       // lines = all local rows
    double *lo = a.lo ? a.lo + (size_t)bz * (size_t)nl * (a.unpack ? 3 : ng) : nullptr;
    double *hi = a.hi ? a.hi + (size_t)bz * (size_t)nl * (a.unpack ? ng : 3) : nullptr;
    if (!a.unpack) {
      if (lo) KLOOP2(c, l, 0, ng - 1, 0, nl - 1) lo[l * ng + c] = A[X2(B.Istr + c, G.LBj + l)];
      if (hi) KLOOP2(c, l, 0, 2, 0, nl - 1) hi[l * 3 + c] = A[X2(B.Iend - 2 + c, G.LBj + l)];
    } else {
      if (lo) KLOOP2(c, l, 0, 2, 0, nl - 1) A[X2(B.Istr - 3 + c, G.LBj + l)] = lo[l * 3 + c];
      if (hi) KLOOP2(c, l, 0, ng - 1, 0, nl - 1) A[X2(B.Iend + 1 + c, G.LBj + l)] = hi[l * ng + c];
    }
  } else {
    const int nl = G.ni;                       // lines = all local columns
    double *lo = a.lo ? a.lo + (size_t)bz * (size_t)nl * (a.unpack ? 3 : ng) : nullptr;
    double *hi = a.hi ? a.hi + (size_t)bz * (size_t)nl * (a.unpack ? ng : 3) : nullptr;
    if (!a.unpack) {
      if (lo) KLOOP2(l, c, 0, nl - 1, 0, ng - 1) lo[c * nl + l] = A[X2(G.LBi + l, B.Jstr + c)];
      if (hi) KLOOP2(l, c, 0, nl - 1, 0, 2) hi[c * nl + l] = A[X2(G.LBi + l, B.Jend - 2 + c)];
    } else {
      if (lo) KLOOP2(l, c, 0, nl - 1, 0, 2) A[X2(G.LBi + l, B.Jstr - 3 + c)] = lo[c * nl + l];
      if (hi) KLOOP2(l, c, 0, nl - 1, 0, ng - 1) A[X2(G.LBi + l, B.Jend + 1 + c)] = hi[c * nl + l];
    }
  }
}
COOP_GLOBAL(strip_kernel, StripArgs)
