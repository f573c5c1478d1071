// k_haloblock.h -- boundary fills and periodic ghost copies written by the producing thread.
//
// The reference fills closed-boundary points from the adjacent interior point (bc_2d.F, zetabc.F,
// u2dbc_im.F, v2dbc_im.F: gradient / gamma2 slip / zero normal flow) and then copies the periodic
// ghost points (exchange_2d.F).  When at least one direction is periodic there are no corner
// averages, and every boundary or ghost value is a function of ONE interior value.  The thread
// that computes that value therefore stores all of its images itself: no extra kernel, no barrier,
// no re-read.  A closed basin (neither direction periodic; round 6) adds the corner averages of two boundary values
// (zetabc.F:753-782, u2dbc_im.F:1159-1188, v2dbc_im.F:1208-1237) -- both of them derived from the ONE interior point next to
// the corner, or the zero of the wall: hb_corner, by the thread of that point.  (Multi-GPU runs, whose strips travel
// between tiles, and open boundaries keep the separate halo launch.)
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

// Closed basin: the corner value of A behind the interior point (i,j) that lies next to a corner of the domain.  As the
// reference writes it, 0.5*(boundary value along xi + boundary value along eta): for a rho-type field both are this point's
// value (times the mask of the boundary point); for ubar one is the slip value behind the southern/northern edge and the other
// the zero of the western/eastern wall (for vbar the other way round).  ST(x, y, value) stores.
#define HB_CORNERS(ST)                                                                                                       \
  if (bc == BC_R) {                                                                                                          \
    const bool w = B.west && i == B.Istr, e = B.east && i == B.Iend, s_ = B.south && j == B.Jstr, n = B.north && j == B.Jend; \
    if ((w || e) && (s_ || n)) {                                                                                             \
      const int ic = w ? i - 1 : i + 1, jc = s_ ? j - 1 : j + 1;                                                             \
      const double vx = M ? v * M[X2(ic, j)] : v, vy = M ? v * M[X2(i, jc)] : v;                                             \
      ST(ic, jc, 0.5 * (vy + vx));                                                                                           \
    }                                                                                                                        \
  } else if (bc == BC_U) {                                                                                                   \
    const bool w = B.west && i == B.IstrU, e = B.east && i == B.Iend, s_ = B.south && j == B.Jstr, n = B.north && j == B.Jend; \
    if (w && (s_ || n)) { const int jc = s_ ? j - 1 : j + 1; ST(B.Istr, jc, 0.5 * ((M ? G.gamma2 * v * M[X2(i, jc)] : G.gamma2 * v) + 0.0)); } \
    if (e && (s_ || n)) { const int jc = s_ ? j - 1 : j + 1; ST(i + 1, jc, 0.5 * ((M ? G.gamma2 * v * M[X2(i, jc)] : G.gamma2 * v) + 0.0)); } \
  } else if (bc == BC_V) {                                                                                                   \
    const bool w = B.west && i == B.Istr, e = B.east && i == B.Iend, s_ = B.south && j == B.JstrV, n = B.north && j == B.Jend; \
    if (s_ && (w || e)) { const int ic = w ? i - 1 : i + 1; ST(ic, B.Jstr, 0.5 * (0.0 + (M ? G.gamma2 * v * M[X2(ic, j)] : G.gamma2 * v))); } \
    if (n && (w || e)) { const int ic = w ? i - 1 : i + 1; ST(ic, j + 1, 0.5 * (0.0 + (M ? G.gamma2 * v * M[X2(ic, j)] : G.gamma2 * v))); } \
  }

// M (MASKING): the mask array of A's grid type, or null -- the gradient / slip value stored at a boundary point is
// multiplied by the mask of THAT point (zetabc.F:264, u2dbc_im.F:989, v2dbc_im.F:1048); zero values stay zero
KDEV void hb_emit(const DGrid &G, const TB &B, double *A, int bc, int i, int j, double v, const double *M = nullptr) {
  // points further than three lines from every domain edge have no image (Nghost <= 3; the boundary
  // fills read the first/last interior line): whole waves of interior sub-tiles take this exit
  if (i > 3 && i < G.Lm - 2 && j > 3 && j < G.Mm - 2) { A[X2(i, j)] = v; return; }
  hb_mirror(G, A, i, j, v);
  if (bc == BC_NONE) return;
  if (!G.nsp) {          // closed southern / northern edge
    if (bc == BC_R) {
      if (B.south && j == B.Jstr) hb_mirror(G, A, i, j - 1, M ? v * M[X2(i, j - 1)] : v);
      if (B.north && j == B.Jend) hb_mirror(G, A, i, j + 1, M ? v * M[X2(i, j + 1)] : v);
    } else if (bc == BC_U) {
      if (B.south && j == B.Jstr) hb_mirror(G, A, i, j - 1, M ? G.gamma2 * v * M[X2(i, j - 1)] : G.gamma2 * v);
      if (B.north && j == B.Jend) hb_mirror(G, A, i, j + 1, M ? G.gamma2 * v * M[X2(i, j + 1)] : G.gamma2 * v);
    } else if (bc == BC_V) {
      if (B.south && j == B.JstrV) hb_mirror(G, A, i, B.Jstr, 0.0);
      if (B.north && j == B.Jend) hb_mirror(G, A, i, j + 1, 0.0);
    }
  }
  if (!G.ewp) {          // closed western / eastern edge
    if (bc == BC_R) {
      if (B.west && i == B.Istr) hb_mirror(G, A, i - 1, j, M ? v * M[X2(i - 1, j)] : v);
      if (B.east && i == B.Iend) hb_mirror(G, A, i + 1, j, M ? v * M[X2(i + 1, j)] : v);
    } else if (bc == BC_U) {
      if (B.west && i == B.IstrU) hb_mirror(G, A, B.Istr, j, 0.0);
      if (B.east && i == B.Iend) hb_mirror(G, A, i + 1, j, 0.0);
    } else if (bc == BC_V) {
      if (B.west && i == B.Istr) hb_mirror(G, A, i - 1, j, M ? G.gamma2 * v * M[X2(i - 1, j)] : G.gamma2 * v);
      if (B.east && i == B.Iend) hb_mirror(G, A, i + 1, j, M ? G.gamma2 * v * M[X2(i + 1, j)] : G.gamma2 * v);
    }
  }
  if (!G.ewp && !G.nsp) {
#define HB_ST1_(x_, y_, v_) A[X2(x_, y_)] = (v_)
    HB_CORNERS(HB_ST1_)
#undef HB_ST1_
  }
}

// The same with the periodic images optional (k_step2d_pair.h: while the pairs follow each other nobody reads the
// periodic ghost points -- the kernel reads its rim at the wrapped own points -- so only the last launches store them;
// the boundary values derived at a closed edge are stored every time, the next launch reads them).  Straight-line: at
// most one image along each periodic direction (Lm, Mm >= 6), offsets instead of index lists.
// WT (k_step2d_loop.h): the stores are write-through (agent-scope relaxed atomic stores = `global_store ... sc1`): the values
// are read by OTHER workgroups of the same launch, behind an arrival word
#ifdef ROMS_CPU_EMU
#define HB_ST(WT_, A_, x_, v_) ((A_)[x_] = (v_))
#else
#define HB_ST(WT_, A_, x_, v_) do { if (WT_) __hip_atomic_store((unsigned long long *)((A_) + (x_)), (unsigned long long)__double_as_longlong(v_), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else (A_)[x_] = (v_); } while (0)
#endif
template <bool WT = false>
KDEV void hb_put(const DGrid &G, double *A, int i, int j, double v, bool images) {
  const int x = (int)X2(i, j);
  HB_ST(WT, A, x, v);
  if (!images) return;
  int dxo = 0, dyo = 0;
  if (G.ewp) { if (i >= 1 && i <= G.Nghost) dxo = G.Lm; else if (i >= G.Lm - 2 && i <= G.Lm) dxo = -G.Lm; }
  if (G.nsp) { if (j >= 1 && j <= G.Nghost) dyo = G.Mm * G.ni; else if (j >= G.Mm - 2 && j <= G.Mm) dyo = -G.Mm * G.ni; }
  if (dxo) HB_ST(WT, A, x + dxo, v);
  if (dyo) HB_ST(WT, A, x + dyo, v);
  if (dxo && dyo) HB_ST(WT, A, x + dxo + dyo, v);
}
template <bool WT = false>
KDEV void hb_emit2(const DGrid &G, const TB &B, double *A, int bc, int i, int j, double v, const double *M, bool images) {
  if (i > 3 && i < G.Lm - 2 && j > 3 && j < G.Mm - 2) { HB_ST(WT, A, (int)X2(i, j), v); return; }
  hb_put<WT>(G, A, i, j, v, images);
  if (bc == BC_NONE) return;
  if (!G.nsp) {          // closed southern / northern edge
    if (bc == BC_R) {
      if (B.south && j == B.Jstr) hb_put<WT>(G, A, i, j - 1, M ? v * M[X2(i, j - 1)] : v, images);
      if (B.north && j == B.Jend) hb_put<WT>(G, A, i, j + 1, M ? v * M[X2(i, j + 1)] : v, images);
    } else if (bc == BC_U) {
      if (B.south && j == B.Jstr) hb_put<WT>(G, A, i, j - 1, M ? G.gamma2 * v * M[X2(i, j - 1)] : G.gamma2 * v, images);
      if (B.north && j == B.Jend) hb_put<WT>(G, A, i, j + 1, M ? G.gamma2 * v * M[X2(i, j + 1)] : G.gamma2 * v, images);
    } else if (bc == BC_V) {
      if (B.south && j == B.JstrV) hb_put<WT>(G, A, i, B.Jstr, 0.0, images);
      if (B.north && j == B.Jend) hb_put<WT>(G, A, i, j + 1, 0.0, images);
    }
  }
  if (!G.ewp) {          // closed western / eastern edge
    if (bc == BC_R) {
      if (B.west && i == B.Istr) hb_put<WT>(G, A, i - 1, j, M ? v * M[X2(i - 1, j)] : v, images);
      if (B.east && i == B.Iend) hb_put<WT>(G, A, i + 1, j, M ? v * M[X2(i + 1, j)] : v, images);
    } else if (bc == BC_U) {
      if (B.west && i == B.IstrU) hb_put<WT>(G, A, B.Istr, j, 0.0, images);
      if (B.east && i == B.Iend) hb_put<WT>(G, A, i + 1, j, 0.0, images);
    } else if (bc == BC_V) {
      if (B.west && i == B.Istr) hb_put<WT>(G, A, i - 1, j, M ? G.gamma2 * v * M[X2(i - 1, j)] : G.gamma2 * v, images);
      if (B.east && i == B.Iend) hb_put<WT>(G, A, i + 1, j, M ? G.gamma2 * v * M[X2(i + 1, j)] : G.gamma2 * v, images);
    }
  }
  if (!G.ewp && !G.nsp) {
#define HB_ST2_(x_, y_, v_) HB_ST(WT, A, (int)X2(x_, y_), (v_))
    HB_CORNERS(HB_ST2_)
#undef HB_ST2_
  }
}

// Final stores of a point-wise kernel whose index space is the whole tile.  When the halo launch that
// would follow is fused (G.fuse3d: single tile, at least one periodic direction) the storing thread
// also writes the boundary value derived from its point and the periodic images of both.  Which
// array elements those are depends on (i,j) and the boundary kind only, so a thread works them out
// once (emit_plan: at most 4 targets = the point and its image along the periodic direction, the
// boundary point behind a closed edge and its image) and every level costs one store plus, on the few
// edge lanes, up to three more (emit_store).  Points that a kernel computes outside 1..Lm x 1..Mm
// (boundary rows, its own ghost columns) store their own value; where such a point coincides with an
// image, both writers store the same bits (the inputs of a point-wise kernel at a ghost point are the
// images of its inputs).
enum { EMIT_COPY = 0, EMIT_GAMMA2 = 1, EMIT_ZERO = 2 };
// Targets: o0 the point itself; with both directions periodic o1, o2, o3 = its xi image, eta image and
// corner image; with one periodic direction o1 = its image, o2 = the boundary point derived from it
// behind the closed edge (value by `kind`), o3 = the image of that.  m: bit q-1 set = target q exists.
// (Scalar members only: an array indexed at run time would put the plan in scratch memory.)
struct EmitPlan {
  int o0, o1, o2, o3;
  int m, kind;
};
KDEV EmitPlan emit_plan(const DGrid &G, int bc, int i, int j) {
  EmitPlan P;
  P.o0 = (int)X2(i, j); P.o1 = P.o0; P.o2 = P.o0; P.o3 = P.o0; P.m = 0; P.kind = EMIT_COPY;
  if (!G.fuse3d || (i > 3 && i < G.Lm - 2 && j > 3 && j < G.Mm - 2)) return P;
  const TB &B = G.T;
  int ix = i, jy = j;
  bool hx = false, hy = false;
  if (G.ewp) {
    if (i >= 1 && i <= G.Nghost) { ix = G.Lm + i; hx = true; }
    else if (i >= G.Lm - 2 && i <= G.Lm) { ix = i - G.Lm; hx = true; }
  }
  if (G.nsp) {
    if (j >= 1 && j <= G.Nghost) { jy = G.Mm + j; hy = true; }
    else if (j >= G.Mm - 2 && j <= G.Mm) { jy = j - G.Mm; hy = true; }
  }
  if (G.ewp && G.nsp) {
    if (hx) { P.o1 = (int)X2(ix, j); P.m |= 1; }
    if (hy) { P.o2 = (int)X2(i, jy); P.m |= 2; }
    if (hx && hy) { P.o3 = (int)X2(ix, jy); P.m |= 4; }
    return P;
  }
  if (hx) { P.o1 = (int)X2(ix, j); P.m |= 1; }
  if (hy) { P.o1 = (int)X2(i, jy); P.m |= 1; }
  if (bc == BC_NONE) return P;
  int di = i, dj = j;
  bool hd = false;
  if (!G.nsp) {          // closed southern / northern edge (the rules of hb_emit)
    if (bc == BC_R) {
      if (B.south && j == B.Jstr) { dj = j - 1; hd = true; P.kind = EMIT_COPY; }
      if (B.north && j == B.Jend) { dj = j + 1; hd = true; P.kind = EMIT_COPY; }
    } else if (bc == BC_U) {
      if (B.south && j == B.Jstr) { dj = j - 1; hd = true; P.kind = EMIT_GAMMA2; }
      if (B.north && j == B.Jend) { dj = j + 1; hd = true; P.kind = EMIT_GAMMA2; }
    } else if (bc == BC_V) {
      if (B.south && j == B.JstrV) { dj = B.Jstr; hd = true; P.kind = EMIT_ZERO; }
      if (B.north && j == B.Jend) { dj = j + 1; hd = true; P.kind = EMIT_ZERO; }
    }
    if (hd) {
      P.o2 = (int)X2(i, dj); P.m |= 2;
      if (hx) { P.o3 = (int)X2(ix, dj); P.m |= 4; }
    }
  } else {               // closed western / eastern edge (eta is the periodic direction)
    if (bc == BC_R) {
      if (B.west && i == B.Istr) { di = i - 1; hd = true; P.kind = EMIT_COPY; }
      if (B.east && i == B.Iend) { di = i + 1; hd = true; P.kind = EMIT_COPY; }
    } else if (bc == BC_U) {
      if (B.west && i == B.IstrU) { di = B.Istr; hd = true; P.kind = EMIT_ZERO; }
      if (B.east && i == B.Iend) { di = i + 1; hd = true; P.kind = EMIT_ZERO; }
    } else if (bc == BC_V) {
      if (B.west && i == B.Istr) { di = i - 1; hd = true; P.kind = EMIT_GAMMA2; }
      if (B.east && i == B.Iend) { di = i + 1; hd = true; P.kind = EMIT_GAMMA2; }
    }
    if (hd) {
      P.o2 = (int)X2(di, j); P.m |= 2;
      if (hy) { P.o3 = (int)X2(di, jy); P.m |= 4; }
    }
  }
  return P;
}
KDEV void emit_store(const DGrid &G, const EmitPlan &P, double *A, double v) {
  A[P.o0] = v;
  if (P.m) {
    const double d = P.kind == EMIT_COPY ? v : (P.kind == EMIT_GAMMA2 ? G.gamma2 * v : 0.0);
    if (P.m & 1) A[P.o1] = v;
    if (P.m & 2) A[P.o2] = d;
    if (P.m & 4) A[P.o3] = d;
  }
}
