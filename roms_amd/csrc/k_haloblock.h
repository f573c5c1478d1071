// k_haloblock.h -- boundary fills and periodic ghost copies written by the producing thread.
//
// The reference fills closed-boundary points from the adjacent interior point (bc_2d.F, zetabc.F,
// u2dbc_im.F, v2dbc_im.F: gradient / gamma2 slip / zero normal flow) and then copies the periodic
// ghost points (exchange_2d.F).  When at least one direction is periodic there are no corner
// averages, and every boundary or ghost value is a function of ONE interior value.  The thread
// that computes that value therefore stores all of its images itself: no extra kernel, no barrier,
// no re-read.  (Closed basins need the corner averages of two different edge values and keep the
// separate halo launch, as do multi-GPU runs whose strips travel between tiles.)
//
//   hb_emit(G, B, A, bc, i, j, v)   A(i,j) = v was computed at an interior point of sub-tile B;
//                                   stores A(i,j), the boundary values derived from it and the
//                                   periodic images of all of them
#pragma once
#include "roms_ctx.h"

// A(i,j) = v and its periodic images: east ghosts Lm+1..Lm+Nghost mirror columns 1..Nghost, west
// ghosts -2..0 mirror columns Lm-2..Lm (exchange_2d.F:100-160), the same along eta.
KDEV void hb_mirror(const DGrid &G, double *A, int i, int j, double v) {
  int xs[3] = {i, i, i}, ys[3] = {j, j, j}, nx = 1, ny = 1;
  if (G.ewp) {
    if (i >= 1 && i <= G.Nghost) xs[nx++] = G.Lm + i;
    if (i >= G.Lm - 2 && i <= G.Lm && nx < 3) xs[nx++] = i - G.Lm;
  }
  if (G.nsp) {
    if (j >= 1 && j <= G.Nghost) ys[ny++] = G.Mm + j;
    if (j >= G.Mm - 2 && j <= G.Mm && ny < 3) ys[ny++] = j - G.Mm;
  }
#pragma unroll
  for (int b = 0; b < 3; b++)
#pragma unroll
    for (int a = 0; a < 3; a++)
      if (a < nx && b < ny) A[X2(xs[a], ys[b])] = v;
}

KDEV void hb_emit(const DGrid &G, const TB &B, double *A, int bc, int i, int j, double v) {
  // points further than three lines from every domain edge have no image (Nghost <= 3; the boundary
  // fills read the first/last interior line): whole waves of interior sub-tiles take this exit
  if (i > 3 && i < G.Lm - 2 && j > 3 && j < G.Mm - 2) { A[X2(i, j)] = v; return; }
  hb_mirror(G, A, i, j, v);
  if (bc == BC_NONE) return;
  if (!G.nsp) {          // closed southern / northern edge
    if (bc == BC_R) {
      if (B.south && j == B.Jstr) hb_mirror(G, A, i, j - 1, v);
      if (B.north && j == B.Jend) hb_mirror(G, A, i, j + 1, v);
    } else if (bc == BC_U) {
      if (B.south && j == B.Jstr) hb_mirror(G, A, i, j - 1, G.gamma2 * v);
      if (B.north && j == B.Jend) hb_mirror(G, A, i, j + 1, G.gamma2 * v);
    } else if (bc == BC_V) {
      if (B.south && j == B.JstrV) hb_mirror(G, A, i, B.Jstr, 0.0);
      if (B.north && j == B.Jend) hb_mirror(G, A, i, j + 1, 0.0);
    }
  }
  if (!G.ewp) {          // closed western / eastern edge
    if (bc == BC_R) {
      if (B.west && i == B.Istr) hb_mirror(G, A, i - 1, j, v);
      if (B.east && i == B.Iend) hb_mirror(G, A, i + 1, j, v);
    } else if (bc == BC_U) {
      if (B.west && i == B.IstrU) hb_mirror(G, A, B.Istr, j, 0.0);
      if (B.east && i == B.Iend) hb_mirror(G, A, i + 1, j, 0.0);
    } else if (bc == BC_V) {
      if (B.west && i == B.Istr) hb_mirror(G, A, i - 1, j, G.gamma2 * v);
      if (B.east && i == B.Iend) hb_mirror(G, A, i + 1, j, G.gamma2 * v);
    }
  }
}
