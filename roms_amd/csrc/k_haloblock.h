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
// All threads of the block must call halo_block (it contains barriers).  The arrays must not be
// read by other blocks of the same kernel.
#pragma once
#include "roms_ctx.h"

#define HALOBLOCK_MAX 4
struct HaloBlockItems {
  double *A[HALOBLOCK_MAX];
  int bc[HALOBLOCK_MAX];
  int gt[HALOBLOCK_MAX];
  int n;
};

KDEV void halo_block(const DGrid &G, const TB &B, const HaloBlockItems &H) {
  const int Istr = B.Istr, Iend = B.Iend, Jstr = B.Jstr, Jend = B.Jend;
  const int Lm = G.Lm, Mm = G.Mm;
  const double gamma2 = G.gamma2;
  KSYNC();   // the block's own results are visible to all its threads
  // ---- phase 1a: west/east edges (BC_R: all four edges)
  for (int q = 0; q < H.n; q++) {
    double *A = H.A[q];
    const int bc = H.bc[q];
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
    } else if (bc == BC_V) {
      if (!G.ewp) {
        const int Jmin = G.nsp ? B.JstrV : B.Jstr, Jmax = G.nsp ? B.Jend : B.JendR;
        if (B.west) KLOOP1(j, Jmin, Jmax) A[X2(Istr - 1, j)] = gamma2 * A[X2(Istr, j)];
        if (B.east) KLOOP1(j, Jmin, Jmax) A[X2(Iend + 1, j)] = gamma2 * A[X2(Iend, j)];
      }
    }
  }
  KSYNC();
  // ---- phase 1b: south/north edges of the u- and v-type fills
  for (int q = 0; q < H.n; q++) {
    double *A = H.A[q];
    const int bc = H.bc[q];
    if (bc == BC_U) {
      if (!G.nsp) {
        const int Imin = G.ewp ? B.IstrU : B.Istr, Imax = G.ewp ? B.Iend : B.IendR;
        if (B.south) KLOOP1(i, Imin, Imax) A[X2(i, Jstr - 1)] = gamma2 * A[X2(i, Jstr)];
        if (B.north) KLOOP1(i, Imin, Imax) A[X2(i, Jend + 1)] = gamma2 * A[X2(i, Jend)];
      }
    } else if (bc == BC_V) {
      if (!G.nsp) {
        if (B.south) KLOOP1(i, Istr, Iend) A[X2(i, Jstr)] = 0.0;
        if (B.north) KLOOP1(i, Istr, Iend) A[X2(i, Jend + 1)] = 0.0;
      }
    }
  }
  KSYNC();
  // ---- phase 2: corners (only when neither direction is periodic)
  if (!(G.ewp || G.nsp) && KTID == 0) {
    for (int q = 0; q < H.n; q++) {
      double *A = H.A[q];
      const int bc = H.bc[q];
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
      } else if (bc == BC_V) {
        if (B.sw) A[X2(Istr - 1, Jstr)] = 0.5 * (A[X2(Istr, Jstr)] + A[X2(Istr - 1, Jstr + 1)]);
        if (B.se) A[X2(Iend + 1, Jstr)] = 0.5 * (A[X2(Iend, Jstr)] + A[X2(Iend + 1, Jstr + 1)]);
        if (B.nw) A[X2(Istr - 1, Jend + 1)] = 0.5 * (A[X2(Istr - 1, Jend)] + A[X2(Istr, Jend + 1)]);
        if (B.ne) A[X2(Iend + 1, Jend + 1)] = 0.5 * (A[X2(Iend + 1, Jend)] + A[X2(Iend, Jend + 1)]);
      }
    }
  }
  KSYNC();
  // ---- phase 3: periodic ghost copies; the block owning the source columns/rows writes them
  if (G.ewp || G.nsp) {
    const int ng3 = G.Nghost == 3;
    for (int q = 0; q < H.n; q++) {
      double *A = H.A[q];
      const int gt = H.gt[q];
      if (gt == 0) continue;
      int Jmin, Jmax, Imin, Imax;
      if (G.nsp) { Jmin = B.Jstr; Jmax = B.Jend; }
      else { Jmin = (gt == 'r' || gt == 'u') ? B.JstrR : B.Jstr; Jmax = B.JendR; }
      if (G.ewp) { Imin = B.Istr; Imax = B.Iend; }
      else { Imin = (gt == 'r' || gt == 'v') ? B.IstrR : B.Istr; Imax = B.IendR; }
      if (G.ewp) {
        if (B.west) KLOOP1(j, Jmin, Jmax) {
          A[X2(Lm + 1, j)] = A[X2(1, j)];
          A[X2(Lm + 2, j)] = A[X2(2, j)];
          if (ng3) A[X2(Lm + 3, j)] = A[X2(3, j)];
        }
        if (B.east) KLOOP1(j, Jmin, Jmax) {
          A[X2(-2, j)] = A[X2(Lm - 2, j)];
          A[X2(-1, j)] = A[X2(Lm - 1, j)];
          A[X2(0, j)] = A[X2(Lm, j)];
        }
      }
      if (G.nsp) {
        if (B.south) KLOOP1(i, Imin, Imax) {
          A[X2(i, Mm + 1)] = A[X2(i, 1)];
          A[X2(i, Mm + 2)] = A[X2(i, 2)];
          if (ng3) A[X2(i, Mm + 3)] = A[X2(i, 3)];
        }
        if (B.north) KLOOP1(i, Imin, Imax) {
          A[X2(i, -2)] = A[X2(i, Mm - 2)];
          A[X2(i, -1)] = A[X2(i, Mm - 1)];
          A[X2(i, 0)] = A[X2(i, Mm)];
        }
      }
      if (G.ewp && G.nsp && KTID == 0) {
        const int ne = ng3 ? 3 : 2;
        if (B.sw)
          for (int dj = 1; dj <= ne; dj++)
            for (int di = 1; di <= ne; di++) A[X2(Lm + di, Mm + dj)] = A[X2(di, dj)];
        if (B.se)
          for (int dj = 1; dj <= ne; dj++)
            for (int di = -2; di <= 0; di++) A[X2(di, Mm + dj)] = A[X2(Lm + di, dj)];
        if (B.nw)
          for (int dj = -2; dj <= 0; dj++)
            for (int di = 1; di <= ne; di++) A[X2(Lm + di, dj)] = A[X2(di, Mm + dj)];
        if (B.ne)
          for (int dj = -2; dj <= 0; dj++)
            for (int di = -2; di <= 0; di++) A[X2(di, dj)] = A[X2(Lm + di, Mm + dj)];
      }
    }
  }
}
