// k_haloblock.h -- boundary fills and periodic ghost copies done by the producing kernel.
//
// Same operations, in the same order, as halo_kernel (k_halo.h; reference: bc_2d.F, zetabc.F,
// u2dbc_im.F, v2dbc_im.F, exchange_2d.F), but executed by every thread block of a COOP kernel for
// the part of the boundary that mirrors ITS OWN sub-tile: the block that owns the last three
// interior columns writes the three west ghost columns, the block on the southern edge fills its
// piece of the boundary row, and so on.  No block reads another block's results, so no grid-wide
// synchronisation is needed and the separate halo launch disappears (single-tile runs only: with
// neighbouring tiles on other GPUs the strips have to travel, see roms_hip.cpp:exchange_phase).
//
// An item may carry an LDS tile T (sub-tile rectangle, S2 indexing) holding the block's new values:
// the fills then read the tile instead of re-reading global memory the block has just written, and
// update the tile where the target lies inside it.  Tile entries the kernel has not written must
// hold hb_sentinel(); such entries (boundary rows the kernel does not compute) are read from
// global memory, exactly as without a tile.
//
// All threads of the block must call halo_block (it contains barriers).  The arrays must not be
// read by other blocks of the same kernel.  Items are passed as separate by-value structs (not an
// indexed array) so that they stay in registers.
#pragma once
#include "roms_ctx.h"
#include <cstring>

struct HbItem {
  double *A;    // global array (one horizontal plane)
  double *T;    // LDS tile with the block's new values, or nullptr
  int bc;       // BC_*
  int gt;       // 'r','u','v','p': transverse ranges of the periodic copy; 0 = none
};

#define HB_SENTINEL_BITS 0x7FF8DEADBEEF0001ULL
KDEV double hb_sentinel() {
  const unsigned long long b = HB_SENTINEL_BITS;
  double d;
  memcpy(&d, &b, sizeof(d));
  return d;
}
KDEV double hb_get(const DGrid &G, const TB &B, const HbItem &I, int i, int j) {
  if (I.T) {
    const double t = I.T[S2(i, j)];
    unsigned long long b;
    memcpy(&b, &t, sizeof(b));
    if (b != HB_SENTINEL_BITS) return t;
  }
  return I.A[X2(i, j)];
}
KDEV void hb_put(const DGrid &G, const TB &B, const HbItem &I, int i, int j, double v) {
  if (I.T && i >= B.Istr - 3 && i <= B.Iend + 3 && j >= B.Jstr - 3 && j <= B.Jend + 3) I.T[S2(i, j)] = v;
  I.A[X2(i, j)] = v;
}
#define HBG(i, j) hb_get(G, B, I, i, j)
#define HBP(i, j, v) hb_put(G, B, I, i, j, v)

// ---- phase 1a: west/east edges (BC_R: all four edges)
KDEV void hb_phase1a(const DGrid &G, const TB &B, const HbItem &I) {
  const int Istr = B.Istr, Iend = B.Iend, Jstr = B.Jstr, Jend = B.Jend;
  if (I.bc == BC_R) {
    if (!G.ewp) {
      if (B.west) KLOOP1(j, Jstr, Jend) HBP(Istr - 1, j, HBG(Istr, j));
      if (B.east) KLOOP1(j, Jstr, Jend) HBP(Iend + 1, j, HBG(Iend, j));
    }
    if (!G.nsp) {
      if (B.south) KLOOP1(i, Istr, Iend) HBP(i, Jstr - 1, HBG(i, Jstr));
      if (B.north) KLOOP1(i, Istr, Iend) HBP(i, Jend + 1, HBG(i, Jend));
    }
  } else if (I.bc == BC_U) {
    if (!G.ewp) {
      if (B.west) KLOOP1(j, Jstr, Jend) HBP(Istr, j, 0.0);
      if (B.east) KLOOP1(j, Jstr, Jend) HBP(Iend + 1, j, 0.0);
    }
  } else if (I.bc == BC_V) {
    if (!G.ewp) {
      const int Jmin = G.nsp ? B.JstrV : B.Jstr, Jmax = G.nsp ? B.Jend : B.JendR;
      if (B.west) KLOOP1(j, Jmin, Jmax) HBP(Istr - 1, j, G.gamma2 * HBG(Istr, j));
      if (B.east) KLOOP1(j, Jmin, Jmax) HBP(Iend + 1, j, G.gamma2 * HBG(Iend, j));
    }
  }
}
// ---- phase 1b: south/north edges of the u- and v-type fills
KDEV void hb_phase1b(const DGrid &G, const TB &B, const HbItem &I) {
  const int Istr = B.Istr, Iend = B.Iend, Jstr = B.Jstr, Jend = B.Jend;
  if (G.nsp) return;
  if (I.bc == BC_U) {
    const int Imin = G.ewp ? B.IstrU : B.Istr, Imax = G.ewp ? B.Iend : B.IendR;
    if (B.south) KLOOP1(i, Imin, Imax) HBP(i, Jstr - 1, G.gamma2 * HBG(i, Jstr));
    if (B.north) KLOOP1(i, Imin, Imax) HBP(i, Jend + 1, G.gamma2 * HBG(i, Jend));
  } else if (I.bc == BC_V) {
    if (B.south) KLOOP1(i, Istr, Iend) HBP(i, Jstr, 0.0);
    if (B.north) KLOOP1(i, Istr, Iend) HBP(i, Jend + 1, 0.0);
  }
}
// ---- phase 2: corners (only when neither direction is periodic)
KDEV void hb_phase2(const DGrid &G, const TB &B, const HbItem &I) {
  const int Istr = B.Istr, Iend = B.Iend, Jstr = B.Jstr, Jend = B.Jend;
  if (G.ewp || G.nsp || KTID != 0) return;
  if (I.bc == BC_R) {
    if (B.sw) HBP(Istr - 1, Jstr - 1, 0.5 * (HBG(Istr, Jstr - 1) + HBG(Istr - 1, Jstr)));
    if (B.se) HBP(Iend + 1, Jstr - 1, 0.5 * (HBG(Iend, Jstr - 1) + HBG(Iend + 1, Jstr)));
    if (B.nw) HBP(Istr - 1, Jend + 1, 0.5 * (HBG(Istr - 1, Jend) + HBG(Istr, Jend + 1)));
    if (B.ne) HBP(Iend + 1, Jend + 1, 0.5 * (HBG(Iend + 1, Jend) + HBG(Iend, Jend + 1)));
  } else if (I.bc == BC_U) {
    if (B.sw) HBP(Istr, Jstr - 1, 0.5 * (HBG(Istr + 1, Jstr - 1) + HBG(Istr, Jstr)));
    if (B.se) HBP(Iend + 1, Jstr - 1, 0.5 * (HBG(Iend, Jstr - 1) + HBG(Iend + 1, Jstr)));
    if (B.nw) HBP(Istr, Jend + 1, 0.5 * (HBG(Istr, Jend) + HBG(Istr + 1, Jend + 1)));
    if (B.ne) HBP(Iend + 1, Jend + 1, 0.5 * (HBG(Iend + 1, Jend) + HBG(Iend, Jend + 1)));
  } else if (I.bc == BC_V) {
    if (B.sw) HBP(Istr - 1, Jstr, 0.5 * (HBG(Istr, Jstr) + HBG(Istr - 1, Jstr + 1)));
    if (B.se) HBP(Iend + 1, Jstr, 0.5 * (HBG(Iend, Jstr) + HBG(Iend + 1, Jstr + 1)));
    if (B.nw) HBP(Istr - 1, Jend + 1, 0.5 * (HBG(Istr - 1, Jend) + HBG(Istr, Jend + 1)));
    if (B.ne) HBP(Iend + 1, Jend + 1, 0.5 * (HBG(Iend + 1, Jend) + HBG(Iend, Jend + 1)));
  }
}
// ---- phase 3: periodic ghost copies; the block owning the source columns/rows writes them
KDEV void hb_phase3(const DGrid &G, const TB &B, const HbItem &I) {
  if (!(G.ewp || G.nsp) || I.gt == 0) return;
  const int Lm = G.Lm, Mm = G.Mm, gt = I.gt;
  const int ng3 = G.Nghost == 3;
  int Jmin, Jmax, Imin, Imax;
  if (G.nsp) { Jmin = B.Jstr; Jmax = B.Jend; }
  else { Jmin = (gt == 'r' || gt == 'u') ? B.JstrR : B.Jstr; Jmax = B.JendR; }
  if (G.ewp) { Imin = B.Istr; Imax = B.Iend; }
  else { Imin = (gt == 'r' || gt == 'v') ? B.IstrR : B.Istr; Imax = B.IendR; }
  if (G.ewp) {
    if (B.west) KLOOP1(j, Jmin, Jmax) {
      HBP(Lm + 1, j, HBG(1, j));
      HBP(Lm + 2, j, HBG(2, j));
      if (ng3) HBP(Lm + 3, j, HBG(3, j));
    }
    if (B.east) KLOOP1(j, Jmin, Jmax) {
      HBP(-2, j, HBG(Lm - 2, j));
      HBP(-1, j, HBG(Lm - 1, j));
      HBP(0, j, HBG(Lm, j));
    }
  }
  if (G.nsp) {
    if (B.south) KLOOP1(i, Imin, Imax) {
      HBP(i, Mm + 1, HBG(i, 1));
      HBP(i, Mm + 2, HBG(i, 2));
      if (ng3) HBP(i, Mm + 3, HBG(i, 3));
    }
    if (B.north) KLOOP1(i, Imin, Imax) {
      HBP(i, -2, HBG(i, Mm - 2));
      HBP(i, -1, HBG(i, Mm - 1));
      HBP(i, 0, HBG(i, Mm));
    }
  }
  if (G.ewp && G.nsp && KTID == 0) {
    const int ne = ng3 ? 3 : 2;
    if (B.sw)
      for (int dj = 1; dj <= ne; dj++)
        for (int di = 1; di <= ne; di++) HBP(Lm + di, Mm + dj, HBG(di, dj));
    if (B.se)
      for (int dj = 1; dj <= ne; dj++)
        for (int di = -2; di <= 0; di++) HBP(di, Mm + dj, HBG(Lm + di, dj));
    if (B.nw)
      for (int dj = -2; dj <= 0; dj++)
        for (int di = 1; di <= ne; di++) HBP(Lm + di, dj, HBG(di, Mm + dj));
    if (B.ne)
      for (int dj = -2; dj <= 0; dj++)
        for (int di = -2; di <= 0; di++) HBP(di, dj, HBG(Lm + di, Mm + dj));
  }
}
#undef HBG
#undef HBP

// up to four items; n is uniform over the block
KDEV void halo_block(const DGrid &G, const TB &B, int n, const HbItem &I0, const HbItem &I1, const HbItem &I2,
                     const HbItem &I3) {
  const bool edge = B.west || B.east || B.south || B.north;
  KSYNC();   // the block's own results (global or tile) are visible to all its threads
  if (!edge) return;   // uniform over the block: interior sub-tiles have nothing to fill
#define HB_ALL(phase)                                                                     \
  do {                                                                                    \
    phase(G, B, I0);                                                                      \
    if (n > 1) phase(G, B, I1);                                                           \
    if (n > 2) phase(G, B, I2);                                                           \
    if (n > 3) phase(G, B, I3);                                                           \
  } while (0)
  HB_ALL(hb_phase1a);
  KSYNC();
  HB_ALL(hb_phase1b);
  KSYNC();
  HB_ALL(hb_phase2);
  KSYNC();
  HB_ALL(hb_phase3);
#undef HB_ALL
}
