// k_step2d_pair.h -- one barotropic predictor + corrector PAIR (fast step iif >= 2) as ONE kernel.
//
// Replaces two consecutive calls of step2d_tile (ROMS/Nonlinear/step2d_LF_AM3.h:163-3056; main3d.F:839 predictor and
// :888 corrector of the same my_iif) by one launch.  The reference exchanges ghost points 3-4 times per call
// (:714, :842, :1068, :3041-3043); a call consumes three lines of ghost points on the low side and two on the high
// side, so a pair needs the predictor on the sub-tile + (3 | 2) lines from inputs on + (5 | 4):
//
//   predictor   the per-call algorithm on the ENLARGED sub-tile E = [Istr-2,Iend+2] x [Jstr-2,Jend+2] (clipped at closed
//               domain edges): zeta(knew=3) on E's range [IstrU_E-1,Iend_E], ubar/vbar(3) on E -- all of it stays in LDS;
//               the block's own points are also stored (level 3, rzeta/rubar/rvbar(krhs), the fast-time averages)
//   boundary    closed domain edges: zetabc / u2dbc / v2dbc of the predictor's result (zetabc.F:577-590, u2dbc_im.F,
//               v2dbc_im.F; halo_fill of k_halo.h) applied to the LDS tiles
//   corrector   the per-call algorithm on the own sub-tile, its krhs level read from LDS
//
// Every value is computed with the per-call kernel's expression (k_step2d.h), and the rim is recomputed by every
// block that needs it: results are bit-identical to the two per-call launches (the reference's tiling invariance).
//
// Time levels.  The corrector's result belongs to level knew = 3 - indx1, which other blocks of the same launch still
// read as the predictor's kstp level on their rims.  It therefore goes to a STAGING level (physical levels 4 and 5 of
// zeta/ubar/vbar, internal to the library, alternating pair by pair); the next launch reads its krhs level from
// there and COMMITS it to the logical level (own points, and the tile's ghost points in a multi-tile run) -- nobody
// reads that level in that launch.  The auxiliary last predictor call (per-call kernel) commits the final one.
//
// Rim points beyond the tile: a tile that is alone in a periodic direction reads them from its own points on the
// other side (wrapx/wrapy: no ghost points are read at all); a tile with neighbours reads its ghost zone, which the
// exchange behind each pair fills 5 | 4 lines wide (roms_hip.cpp).
#pragma once
#include "k_step2d.h"

// ---- the rim across TILE edges inside a launch (round 6): k_step2d_loop.h (the persistent loop) and, for tiles the loop does not
// fit, the pair kernel below.  A rank's rim planes live in its mailbox slab (uncached memory the neighbours map over hipIpc /
// xGMI): four sets x {zeta, ubar, vbar}, indexed like its arrays, 16 bytes per point -- {low half of the value | number of
// the pair}, {high half | number} -- an 8-byte store is atomic on every path, so a point is polled for directly.
// multi-tile contexts: my rim planes and the neighbours' as mapped here
struct S2LPeer {
  int on;                       // 0: single tile (the kernels without MT never read this struct)
  int early;                    // 1: a value goes to the neighbours where it is computed, in front of the local drain (ROMS_HIP_LOOP_EARLY)
  int nbmask;                   // bit d: neighbour d (W, E, S, N, SW, SE, NW, NE) exists
  kword_t *rim;      // my rim planes [set][zeta | ubar | vbar][nij][2 words]: the neighbours' edge blocks write my ghost points
  kword_t *nrim[8];  // neighbour d's rim planes, as mapped in this process
  int noff[8], nni[8], nnij[8]; // my point (i,j) in neighbour d's planes: i + j * nni + noff (array origin + the shift across a periodic seam)
};

// ---- multi-tile: the value of field pf % 3 (0 zeta, 1 ubar, 2 vbar; pf = 3 * parity + field) at (i,j) -- an own point or a
// boundary point derived from one -- goes into the rim planes of every neighbour in whose ghost zone (i,j) lies, tagged
KDEV void s2l_ll_st(kword_t *q, double v, unsigned tag) {
  kword_t b;
  __builtin_memcpy(&b, &v, sizeof(b));
  const kword_t tg = (kword_t)tag << 32;
  ksys_st(q, (b & 0xffffffffull) | tg);
  ksys_st(q + 1, (b >> 32) | tg);
}
KDEV double s2l_ll_val(kword_t w0, kword_t w1) {
  const kword_t b = (w0 & 0xffffffffull) | (w1 << 32);
  double v;
  __builtin_memcpy(&v, &b, sizeof(v));
  return v;
}
KDEV void s2l_rput(const DGrid &G, const S2LPeer &P, int pf, int i, int j, double v, unsigned tag) {
  const TB &T = G.T;
  const bool w = i < T.Istr + B2D_GH, e = i > T.Iend - B2D_GL, s = j < T.Jstr + B2D_GH, n = j > T.Jend - B2D_GL;
  if (!(w || e || s || n)) return;
#pragma unroll
  for (int d = 0; d < 8; d++) {       // (static indices into the argument block)
    const bool hit = d == 0 ? w : d == 1 ? e : d == 2 ? s : d == 3 ? n : d == 4 ? (w && s) : d == 5 ? (e && s) : d == 6 ? (w && n) : (e && n);
    if (hit && P.nrim[d]) s2l_ll_st(P.nrim[d] + 2 * ((size_t)pf * (size_t)P.nnij[d] + (size_t)(i + j * P.nni[d] + P.noff[d])), v, tag);
  }
}
// the point and the boundary values a closed DOMAIN edge derives from it: the rules of hb_emit2 (k_haloblock.h)
KDEV void s2l_remit(const DGrid &G, const S2LPeer &P, const TB &B, int pf, int bc, int i, int j, double v, unsigned tag, const double *M = nullptr) {
  s2l_rput(G, P, pf, i, j, v, tag);
  if (i > 2 && i < G.Lm && j > 2 && j < G.Mm) return;
  if (!G.nsp) {
    if (bc == BC_R) {
      if (B.south && j == B.Jstr) s2l_rput(G, P, pf, i, j - 1, M ? v * M[X2(i, j - 1)] : v, tag);
      if (B.north && j == B.Jend) s2l_rput(G, P, pf, i, j + 1, M ? v * M[X2(i, j + 1)] : v, tag);
    } else if (bc == BC_U) {
      if (B.south && j == B.Jstr) s2l_rput(G, P, pf, i, j - 1, M ? G.gamma2 * v * M[X2(i, j - 1)] : G.gamma2 * v, tag);
      if (B.north && j == B.Jend) s2l_rput(G, P, pf, i, j + 1, M ? G.gamma2 * v * M[X2(i, j + 1)] : G.gamma2 * v, tag);
    } else if (bc == BC_V) {
      if (B.south && j == B.JstrV) s2l_rput(G, P, pf, i, B.Jstr, 0.0, tag);
      if (B.north && j == B.Jend) s2l_rput(G, P, pf, i, j + 1, 0.0, tag);
    }
  }
  if (!G.ewp) {
    if (bc == BC_R) {
      if (B.west && i == B.Istr) s2l_rput(G, P, pf, i - 1, j, M ? v * M[X2(i - 1, j)] : v, tag);
      if (B.east && i == B.Iend) s2l_rput(G, P, pf, i + 1, j, M ? v * M[X2(i + 1, j)] : v, tag);
    } else if (bc == BC_U) {
      if (B.west && i == B.IstrU) s2l_rput(G, P, pf, B.Istr, j, 0.0, tag);
      if (B.east && i == B.Iend) s2l_rput(G, P, pf, i + 1, j, 0.0, tag);
    } else if (bc == BC_V) {
      if (B.west && i == B.Istr) s2l_rput(G, P, pf, i - 1, j, M ? G.gamma2 * v * M[X2(i - 1, j)] : G.gamma2 * v, tag);
      if (B.east && i == B.Iend) s2l_rput(G, P, pf, i + 1, j, M ? G.gamma2 * v * M[X2(i + 1, j)] : G.gamma2 * v, tag);
    }
  }
}



struct Step2dPairArgs {
  S2Fields F;          // (first: DESIGN.md 6)
  DGrid G;             // stepping of the PREDICTOR call: iif >= 2, kstp = 3 - indx1, krhs = indx1, knew = 3
  double w1_m1;        // weight(1,iif-1)
  double w2_0, w2_p1;  // weight(2,iif), weight(2,iif+1)
  int lev_in;          // physical level holding zeta/ubar/vbar(krhs): G.krhs or a staging level
  int lev_out;         // staging level the corrector's result goes to
  int commit;          // lev_in is a staging level: copy it to the logical level G.krhs
  int wrapx, wrapy;    // rim indices beyond the tile wrap onto the tile's own points
  int tail;            // pairs still to follow this one: 0 = the last (iif = nfast).  What nobody reads before the loop ends is
                       // stored by the last launches only: level 3 (the predictor's result) by the last one; periodic images of
                       // the committed level and of the staged result by the last one, of rzeta(krhs) by the last two
  // ---- multi-tile contexts whose tiles are too large for the persistent loop (round 6, g_step2d.cpp:pair_rim_usable): the
  // corrector's result crosses the tile edges INSIDE the launches instead of through an exchange behind every pair
  int rim_in;          // the krhs level on the ghost points of the tile: from my rim planes (set tag_in & 3), polled until the
                       // points carry tag_in -- the neighbours' previous pair launch published them
  int rim_out;         // publish the corrector's result -- the own points in the neighbours' ghost zones and the boundary values a
                       // closed domain edge derives from them -- into their rim planes (set tag_out & 3), tagged tag_out; the
                       // boundary values go to the staging level as well (no fill launch follows)
  // FOUR sets of planes, not two.  The reading relation between blocks is not symmetric across a tile edge (a block two sub-tile
  // rows from the edge reads the neighbour's last line without publishing anything the neighbour's edge block waits for), and the
  // blocks of a launch are not all resident: with two sets a late block of launch n+1 could find its point already carrying tag
  // n+2 and wait for ever (seen with four processes on one device, one run in three).  What the launches DO guarantee: a rank
  // completes launch m only when every neighbour has published m-1, i.e. has STARTED m-1, i.e. has completed m-2 (stream order).
  // A block that publishes tag n+4 runs in launch n+4, so its rank has completed n+3 and every neighbour has completed n+1 --
  // the launch that reads tag n.  Launches without rim_in / rim_out (the ends of a fast loop) are ordered by their exchanges.
  unsigned tag_in, tag_out;
  unsigned long long *err;      // pinned host word: a wait that gave up (bounded like every wait of the library)
  long long timeout;
  S2LPeer P;
};

#define S2P_NLDS 19
#define S2P_RIM 5

// metrics of a momentum point (k_step2d.h: the W* register set)
struct S2Met {
  double onom, fomn0, fomn1, dndx0, dndx1, dmde0, dmde1, v2r0, v2r1, pmr0, pmr1, pnr0, pnr1, or0, or1;
  double v2p0, v2p1, pmp0, pmp1, pnp0, pnp1, op0, op1;
};
template <int ISV>
KDEV S2Met s2_metrics(const M2Rec *mr, const M2Rec *mp, int x, int x1, int q1) {
  const M2Rec R0 = mr[x], R1 = mr[x1], P0 = mp[x], P1 = mp[q1];
  S2Met m;
  m.onom = ISV ? P0.v[MP_OMV] : P0.v[MP_ONU];
  m.fomn0 = R0.v[MR_FOMN]; m.fomn1 = R1.v[MR_FOMN];
  m.dndx0 = R0.v[MR_DNDX]; m.dndx1 = R1.v[MR_DNDX];
  m.dmde0 = R0.v[MR_DMDE]; m.dmde1 = R1.v[MR_DMDE];
  m.v2r0 = R0.v[MR_V2]; m.v2r1 = R1.v[MR_V2];
  m.pmr0 = R0.v[MR_PMON]; m.pmr1 = R1.v[MR_PMON];
  m.pnr0 = R0.v[MR_PNOM]; m.pnr1 = R1.v[MR_PNOM];
  m.or0 = ISV ? R0.v[MR_OM] : R0.v[MR_ON]; m.or1 = ISV ? R1.v[MR_OM] : R1.v[MR_ON];
  m.v2p0 = P0.v[MP_V2]; m.v2p1 = P1.v[MP_V2];
  m.pmp0 = P0.v[MP_PMON]; m.pmp1 = P1.v[MP_PMON];
  m.pnp0 = P0.v[MP_PNOM]; m.pnp1 = P1.v[MP_PNOM];
  m.op0 = ISV ? P0.v[MP_ON] : P0.v[MP_OM]; m.op1 = ISV ? P1.v[MP_ON] : P1.v[MP_OM];
  return m;
}

// LDS tiles a momentum point reads (s = its index in the rectangle, TW = row length)
struct S2Tiles {
  const double *U, *V, *DU, *DV, *D, *PM, *PN, *H, *RA, *gz, *gz2, *gzSA, *zw;
  int TW;
};
// which lines of the (sub-)tile lie on a closed domain edge (4th-order advection: replicated gradients)
struct S2Edge { bool wfix, efix, sfix, nfix; int Istr, Iend, Jstr, Jend; };

// Right-hand side of one barotropic momentum point: pressure gradient (VAR_RHO_2D) :1080-1200, 4th-order centred
// advection :1246-1410, Coriolis :1429-1490, curvilinear terms :1494-1560, harmonic viscosity :1567-1660 -- the
// expressions of k_step2d.h stage 4, operand for operand.
template <int ISV>
KDEV double s2_rhs(const S2Tiles &T, const S2Met &M, const S2Edge &Eg, int s, int i, int j, double g, bool ADV, bool COR,
                   bool CURV, bool VIS, bool MSK, double pmk0, double pmk1) {
  const int TW = T.TW;
  const bool wfix = Eg.wfix, efix = Eg.efix, sfix = Eg.sfix, nfix = Eg.nfix;
  const int Istr = Eg.Istr, Iend = Eg.Iend, Jstr = Eg.Jstr, Jend = Eg.Jend;
  constexpr int isv = ISV;
  const int d1 = isv ? TW : 1;
  const double c6 = 1.0 / 6.0;
  const double pg1 = 0.5 * g, pg2 = 1.0 / 3.0;
#define TU(di, dj) T.U[s + (di) + (dj) * TW]
#define TV(di, dj) T.V[s + (di) + (dj) * TW]
#define TDU(di, dj) T.DU[s + (di) + (dj) * TW]
#define TDV(di, dj) T.DV[s + (di) + (dj) * TW]
#define TD(di, dj) T.D[s + (di) + (dj) * TW]
#define TPM(di, dj) T.PM[s + (di) + (dj) * TW]
#define TPN(di, dj) T.PN[s + (di) + (dj) * TW]
  double rhs = pg1 * M.onom *
               ((T.H[s - d1] + T.H[s]) * (T.gz[s - d1] - T.gz[s]) +
                (T.H[s - d1] - T.H[s]) * (T.gzSA[s - d1] + T.gzSA[s] +
                                         pg2 * (T.RA[s - d1] - T.RA[s]) * (T.zw[s - d1] - T.zw[s])) +
                (T.gz2[s - d1] - T.gz2[s]));
  if (ADV) {
    if (!isv) {
#define GUX(a_) ({ int q_ = (a_); if (wfix && i + q_ == Istr) q_ += 1; if (efix && i + q_ == Iend + 1) q_ -= 1; \
                   TU(q_ - 1, 0) - 2.0 * TU(q_, 0) + TU(q_ + 1, 0); })
#define GDX(a_) ({ int q_ = (a_); if (wfix && i + q_ == Istr) q_ += 1; if (efix && i + q_ == Iend + 1) q_ -= 1; \
                   TDU(q_ - 1, 0) - 2.0 * TDU(q_, 0) + TDU(q_ + 1, 0); })
#define UFX(a_) (0.25 * (TU(a_, 0) + TU((a_) + 1, 0) - c6 * (GUX(a_) + GUX((a_) + 1))) *                      \
                 (TDU(a_, 0) + TDU((a_) + 1, 0) - c6 * (GDX(a_) + GDX((a_) + 1))))
#define GUE(b_) ({ int q_ = (b_); if (sfix && j + q_ == Jstr - 1) q_ += 1; if (nfix && j + q_ == Jend + 1) q_ -= 1; \
                   TU(0, q_ - 1) - 2.0 * TU(0, q_) + TU(0, q_ + 1); })
#define GDE(a_, b_) (TDV((a_) - 1, b_) - 2.0 * TDV(a_, b_) + TDV((a_) + 1, b_))
#define UFE(b_) (0.25 * (TU(0, b_) + TU(0, (b_) - 1) - c6 * (GUE(b_) + GUE((b_) - 1))) *                      \
                 (TDV(0, b_) + TDV(-1, b_) - c6 * (GDE(0, b_) + GDE(-1, b_))))
      const double cff1 = UFX(0) - UFX(-1);
      const double cff2 = UFE(1) - UFE(0);
      const double fac = cff1 + cff2;
      rhs = rhs - fac;
#undef GUX
#undef GDX
#undef UFX
#undef GUE
#undef GDE
#undef UFE
    } else {
#define GVX(a_) ({ int q_ = (a_); if (wfix && i + q_ == Istr - 1) q_ += 1; if (efix && i + q_ == Iend + 1) q_ -= 1; \
                   TV(q_ - 1, 0) - 2.0 * TV(q_, 0) + TV(q_ + 1, 0); })
#define GDX(a_, b_) (TDU(a_, (b_) - 1) - 2.0 * TDU(a_, b_) + TDU(a_, (b_) + 1))
#define VFX(a_) (0.25 * (TV(a_, 0) + TV((a_) - 1, 0) - c6 * (GVX(a_) + GVX((a_) - 1))) *                      \
                 (TDU(a_, 0) + TDU(a_, -1) - c6 * (GDX(a_, 0) + GDX(a_, -1))))
#define GVE(b_) ({ int q_ = (b_); if (sfix && j + q_ == Jstr) q_ += 1; if (nfix && j + q_ == Jend + 1) q_ -= 1; \
                   TV(0, q_ - 1) - 2.0 * TV(0, q_) + TV(0, q_ + 1); })
#define GDE(b_) ({ int q_ = (b_); if (sfix && j + q_ == Jstr) q_ += 1; if (nfix && j + q_ == Jend + 1) q_ -= 1; \
                   TDV(0, q_ - 1) - 2.0 * TDV(0, q_) + TDV(0, q_ + 1); })
#define VFE(b_) (0.25 * (TV(0, b_) + TV(0, (b_) + 1) - c6 * (GVE(b_) + GVE((b_) + 1))) *                      \
                 (TDV(0, b_) + TDV(0, (b_) + 1) - c6 * (GDE(b_) + GDE((b_) + 1))))
      const double cff1 = VFX(1) - VFX(0);
      const double cff2 = VFE(0) - VFE(-1);
      const double fac = cff1 + cff2;
      rhs = rhs - fac;
#undef GVX
#undef GDX
#undef VFX
#undef GVE
#undef GDE
#undef VFE
    }
  }
  const int a1 = isv ? 0 : -1, b1 = isv ? -1 : 0;
  if (COR) {
    const double cf0 = 0.5 * TD(0, 0) * M.fomn0;
    const double cf1 = 0.5 * TD(a1, b1) * M.fomn1;
    if (!isv) {
      const double fac1 = 0.5 * (cf0 * (TV(0, 0) + TV(0, 1)) + cf1 * (TV(-1, 0) + TV(-1, 1)));
      rhs = rhs + fac1;
    } else {
      const double fac1 = 0.5 * (cf0 * (TU(0, 0) + TU(1, 0)) + cf1 * (TU(0, -1) + TU(1, -1)));
      rhs = rhs - fac1;
    }
  }
  if (CURV) {
    double t0, t1;
    {
      const double cff1 = 0.5 * (TV(0, 0) + TV(0, 1)), cff2 = 0.5 * (TU(0, 0) + TU(1, 0));
      const double cff3 = cff1 * M.dndx0, cff4 = cff2 * M.dmde0;
      const double cff = TD(0, 0) * (cff3 - cff4);
      t0 = isv ? cff * cff2 : cff * cff1;
    }
    {
      const double cff1 = 0.5 * (TV(a1, b1) + TV(a1, b1 + 1)), cff2 = 0.5 * (TU(a1, b1) + TU(a1 + 1, b1));
      const double cff3 = cff1 * M.dndx1, cff4 = cff2 * M.dmde1;
      const double cff = TD(a1, b1) * (cff3 - cff4);
      t1 = isv ? cff * cff2 : cff * cff1;
    }
    const double fac1 = 0.5 * (t0 + t1);
    if (!isv) rhs = rhs + fac1;
    else rhs = rhs - fac1;
  }
  if (VIS) {
#define STRESS_R(a_, b_, v2_, pmon_, pnom_)                                                                        \
  ((v2_) * TD(a_, b_) * 0.5 *                                                                                      \
   ((pmon_) * ((TPN(a_, b_) + TPN((a_) + 1, b_)) * TU((a_) + 1, b_) - (TPN((a_) - 1, b_) + TPN(a_, b_)) * TU(a_, b_)) - \
    (pnom_) * ((TPM(a_, b_) + TPM(a_, (b_) + 1)) * TV(a_, (b_) + 1) - (TPM(a_, (b_) - 1) + TPM(a_, b_)) * TV(a_, b_))))
#define DRHS_P(a_, b_) (0.25 * (TD(a_, b_) + TD((a_) - 1, b_) + TD(a_, (b_) - 1) + TD((a_) - 1, (b_) - 1)))
#define STRESS_P(a_, b_, v2_, pmon_, pnom_)                                                                        \
  ((v2_) * DRHS_P(a_, b_) * 0.5 *                                                                                  \
   ((pmon_) * ((TPN(a_, (b_) - 1) + TPN(a_, b_)) * TV(a_, b_) - (TPN((a_) - 1, (b_) - 1) + TPN((a_) - 1, b_)) * TV((a_) - 1, b_)) + \
    (pnom_) * ((TPM((a_) - 1, b_) + TPM(a_, b_)) * TU(a_, b_) - (TPM((a_) - 1, (b_) - 1) + TPM(a_, (b_) - 1)) * TU(a_, (b_) - 1))))
    const int qa = isv ? 1 : 0, qb = isv ? 0 : 1;
    const double sr0 = STRESS_R(0, 0, M.v2r0, M.pmr0, M.pnr0);
    const double sr1 = STRESS_R(a1, b1, M.v2r1, M.pmr1, M.pnr1);
    double sp0 = STRESS_P(0, 0, M.v2p0, M.pmp0, M.pnp0);
    double sp1 = STRESS_P(qa, qb, M.v2p1, M.pmp1, M.pnp1);
    if (MSK) { sp0 = sp0 * pmk0; sp1 = sp1 * pmk1; }
    const double or0 = M.or0, or1 = M.or1, op0 = M.op0, op1 = M.op1;
    if (!isv) {
      const double UFx0 = or0 * or0 * sr0;
      const double UFxm = or1 * or1 * sr1;
      const double UFe0 = op0 * op0 * sp0;
      const double UFep = op1 * op1 * sp1;
      const double cff1 = 0.5 * (TPN(-1, 0) + TPN(0, 0)) * (UFx0 - UFxm);
      const double cff2 = 0.5 * (TPM(-1, 0) + TPM(0, 0)) * (UFep - UFe0);
      const double fac = cff1 + cff2;
      rhs = rhs + fac;
    } else {
      const double VFx0 = op0 * op0 * sp0;
      const double VFxp = op1 * op1 * sp1;
      const double VFe0 = or0 * or0 * sr0;
      const double VFem = or1 * or1 * sr1;
      const double cff1 = 0.5 * (TPN(0, -1) + TPN(0, 0)) * (VFxp - VFx0);
      const double cff2 = 0.5 * (TPM(0, -1) + TPM(0, 0)) * (VFe0 - VFem);
      const double fac = cff1 - cff2;
      rhs = rhs + fac;
    }
#undef STRESS_R
#undef DRHS_P
#undef STRESS_P
  }
#undef TU
#undef TV
#undef TDU
#undef TDV
#undef TD
#undef TPM
#undef TPN
  return rhs;
}

#define INR(i, j, i0, i1, j0, j1) ((i) >= (i0) && (i) <= (i1) && (j) >= (j0) && (j) <= (j1))
// rectangle sweep (IT0:IT0+TW-1, JT0:JT0+TH-1): fixed point -> thread map q = KTID + m*NT, LDS index s0 = q; x0 = the
// point's index in the global arrays (wrapped where the tile closes a periodic direction on itself), ina = it exists
#define RLOOP(i, j)                                                                                         \
  _Pragma("unroll") for (int m = 0, q_ = KTID; FIXED ? m < PTS : q_ < NTILE; m++, q_ += NT)                \
    if (q_ < NTILE)                                                                                         \
      for (int s0 = q_, jj_ = q_ / TW, j = JT0 + jj_, i = IT0 + q_ - jj_ * TW, iw_ = WRAPI(i), jw_ = WRAPJ(j),  \
               ina = (iw_ >= G.LBi && iw_ <= UBi && jw_ >= G.LBj && jw_ <= UBj),                            \
               x0 = (iw_ - G.LBi) + (jw_ - G.LBj) * ni, once_ = 1; once_; once_ = 0)
#define WRAPI(i_) (wrapx ? ((i_) < 1 ? (i_) + G.Lm : ((i_) > G.Lm ? (i_) - G.Lm : (i_))) : (i_))
#define WRAPJ(j_) (wrapy ? ((j_) < 1 ? (j_) + G.Mm : ((j_) > G.Mm ? (j_) - G.Mm : (j_))) : (j_))
#define GIDX(i_, j_) ((WRAPI(i_) - G.LBi) + (WRAPJ(j_) - G.LBj) * ni)
#define PWDECL(name) double name[PTS > 0 ? PTS : 1]
#define PWSET(name, expr) do { if (FIXED) name[m] = (expr); } while (0)
#define PW(name, expr) (FIXED ? name[m] : (expr))
// momentum points: cell c of the enlarged sub-tile (row-major over EWD x EHT), isv = 0 its u-point, 1 its v-point
#define WLOOP(isv)                                                                                          \
  for (int m_ = 0, c = KTID - (isv) * VOFF; FIXED ? m_ < CPT : c < NE; m_++, c += NT)                        \
    if (c >= 0 && c < NE)
#define WSL(isv) (VOFF ? m_ : 2 * m_ + (isv))
#define WDECL(name) double name[WSLOTS]
#define WSET(name, isv, expr) do { if (FIXED) name[WSL(isv)] = (expr); } while (0)
#define WV(name, isv, expr) (FIXED ? name[WSL(isv)] : (expr))

#ifndef ROMS_CPU_EMU
#define S2P_TICK(n) do { if (a.G.dbg_stop == 99 && KTID == 0) F.xr[(bx + G.nbx2 * by) * 16 + (n)] = (double)wall_clock64(); } while (0)
#else
#define S2P_TICK(n) ((void)0)
#endif
template <int BWC, int BHC, int NTC, bool MK = (BWC == 0)>
COOP_KERNEL(k_step2d_pair_t, Step2dPairArgs) {
  (void)bz;
#ifndef ROMS_CPU_EMU
  __builtin_amdgcn_s_setprio(3);
#endif
  constexpr bool FIXED = BWC > 0;
  const bool MSK = MK && a.G.masking;
  const DGrid &G = a.G;
  const S2Fields &F = a.F;
#ifndef ROMS_CPU_EMU
  xcd_remap2(G, bx, by);
#endif
  const TB B = block_bounds2(G, bx, by);
  const bool wrapx = a.wrapx != 0, wrapy = a.wrapy != 0;
  // the enlarged sub-tile of the predictor phase
  int i0E = B.Istr - 2, i1E = B.Iend + 2, j0E = B.Jstr - 2, j1E = B.Jend + 2;
  if (!G.ewp) { i0E = KMAX(i0E, 1); i1E = KMIN(i1E, G.Lm); }
  if (!G.nsp) { j0E = KMAX(j0E, 1); j1E = KMIN(j1E, G.Mm); }
  const TB E = make_bounds(G.Lm, G.Mm, G.ewp, G.nsp, i0E, i1E, j0E, j1E, i0E <= 1, i1E >= G.Lm, j0E <= 1, j1E >= G.Mm);
  const int OW = FIXED ? BWC : G.bw2, OH = FIXED ? BHC : G.bh2;
  const int EWD = OW + 4, EHT = OH + 4, NE = EWD * EHT;
  const int TW = OW + 2 * S2P_RIM, TH = OH + 2 * S2P_RIM, NTILE = TW * TH, NT = FIXED ? NTC : KNT;
  constexpr int NEC = (BWC + 4) * (BHC + 4);                                         // cells of a full-size enlarged sub-tile
  constexpr int VOFF = (BWC > 0 && 2 * ((NEC + 63) / 64 * 64) <= NTC) ? (NEC + 63) / 64 * 64 : 0;
  constexpr int CPT = BWC > 0 ? (VOFF ? (NEC + NTC - 1) / NTC : (NEC + NTC - 1) / NTC) : 0;   // cells per thread and direction
  constexpr int WSLOTS = BWC > 0 ? (VOFF ? CPT : 2 * CPT) : 1;
  constexpr int PTS = BWC > 0 ? ((BWC + 2 * S2P_RIM) * (BHC + 2 * S2P_RIM) + NTC - 1) / NTC : 0;
  const size_t sz = (size_t)NTILE;
  double *D0 = lds, *U0 = lds + sz, *V0 = lds + 2 * sz, *sH = lds + 3 * sz, *sPm = lds + 4 * sz, *sPn = lds + 5 * sz;
  double *sRhoA = lds + 6 * sz, *sDstp = lds + 7 * sz, *DUon = lds + 8 * sz, *DVom = lds + 9 * sz, *D1 = lds + 10 * sz;
  double *zwrk = lds + 11 * sz, *gzeta = lds + 12 * sz, *gzeta2 = lds + 13 * sz, *gzetaSA = lds + 14 * sz;
  double *Z1 = lds + 15 * sz, *U1 = lds + 16 * sz, *V1 = lds + 17 * sz, *RZ1 = lds + 18 * sz;
  double *D2 = sDstp;                                  // the corrector's new depth: the predictor's Dstp tile is dead by then
  const int krhs = G.krhs, kstp = G.kstp, iif = G.iif;
  const double dtfast = G.dtfast, g = G.g;
  const int ni = G.ni, nij = (int)G.nij;
  const int UBi = G.LBi + G.ni - 1, UBj = G.LBj + G.nj - 1;
  const int o_in = (a.lev_in - 1) * nij, o_kstp = (kstp - 1) * nij;       // predictor: krhs (physical), kstp
  const int o_log = (krhs - 1) * nij;                                      // logical krhs level (commit target; corrector's kstp)
  double *zlog = F.zeta + (size_t)(krhs - 1) * G.nij, *ulog = F.ubar + (size_t)(krhs - 1) * G.nij,
         *vlog = F.vbar + (size_t)(krhs - 1) * G.nij;
  double *zn3 = F.zeta + 2 * G.nij, *un3 = F.ubar + 2 * G.nij, *vn3 = F.vbar + 2 * G.nij;
  double *zout = F.zeta + (size_t)(a.lev_out - 1) * G.nij, *uout = F.ubar + (size_t)(a.lev_out - 1) * G.nij,
         *vout = F.vbar + (size_t)(a.lev_out - 1) * G.nij;
  double *rz_k = F.rzeta + (size_t)(krhs - 1) * G.nij;                   // rzeta/rubar/rvbar(krhs of the predictor) = (kstp of the corrector)
  double *rub_k = F.rubar + (size_t)(krhs - 1) * G.nij, *rvb_k = F.rvbar + (size_t)(krhs - 1) * G.nij;
  const int o_ptc = (kstp - 1) * nij;                                      // corrector: ptsk = 3 - kstp_C = the predictor's kstp
  const M2Rec *mr = (const M2Rec *)(double *)F.m2r, *mp = (const M2Rec *)(double *)F.m2p;
  const bool fuse = G.fuse_halo != 0;
  const bool img0 = a.tail == 0, img1 = a.tail <= 1, store3 = a.tail == 0 || !fuse;
  const int IT0 = B.Istr - S2P_RIM, JT0 = B.Jstr - S2P_RIM;
  const bool ADV = (G.options & ROMS_UV_ADV) != 0, COR = (G.options & ROMS_UV_COR) != 0;
  const bool CURV = ADV && (G.options & ROMS_CURVGRID) != 0, VIS = (G.options & ROMS_UV_VIS2) != 0;
  const TB &T = G.T;
  (void)iif;

  PWDECL(r_zk); PWDECL(r_zs); PWDECL(r_on_u); PWDECL(r_om_v); PWDECL(r_rhoS);
  PWDECL(r_on_u1); PWDECL(r_om_v1);      // on_u(i+1,j), om_v(i,j+1): the free-surface stage forms the fluxes of its far faces itself
  PWDECL(r_Zt); PWDECL(r_DU1); PWDECL(r_DU2); PWDECL(r_DV1); PWDECL(r_DV2); PWDECL(r_rz_p);
  S2Met wm[WSLOTS];
  WDECL(w_s); WDECL(w_frc); WDECL(w_rp); WDECL(w_rP); WDECL(w_pk0); WDECL(w_pk1);

  S2P_TICK(0);
  // ---- stage 1: every global read of the kernel ------------------------------------------------
  RLOOP(i, j) {
    if (ina) {
      double zkv, ukv, vkv;
      const double hv = F.h[x0];
      // (multi-tile, rim_in) a ghost point of the tile the neighbours publish: 5 lines on my low side, 4 on my high side
      const bool remv = a.rim_in && !INR(iw_, jw_, T.IstrR, T.IendR, T.JstrR, T.JendR) &&
                        INR(iw_, jw_, (a.P.nbmask & 1) ? T.Istr - B2D_GL : T.IstrR, (a.P.nbmask & 2) ? T.Iend + B2D_GH : T.IendR,
                                      (a.P.nbmask & 4) ? T.Jstr - B2D_GL : T.JstrR, (a.P.nbmask & 8) ? T.Jend + B2D_GH : T.JendR);
      if (remv) {
        const kword_t *rq = a.P.rim + 2 * ((size_t)(3 * (int)(a.tag_in & 3u)) * (size_t)G.nij + (size_t)x0);
        const bool nu = G.ewp || iw_ >= 1, nv = G.nsp || jw_ >= 1;     // (no u-points west of a western wall, no v-points south of a southern one)
        const long long t0 = kclock();
        zkv = 0.0; ukv = 0.0; vkv = 0.0;
        for (;;) {
          kword_t w[6];
          for (int f = 0; f < 3; f++) { w[2 * f] = ksys_ld(rq + 2 * (size_t)f * (size_t)G.nij); w[2 * f + 1] = ksys_ld(rq + 2 * (size_t)f * (size_t)G.nij + 1); }
          const bool all = (unsigned)(w[0] >> 32) == a.tag_in && (unsigned)(w[1] >> 32) == a.tag_in &&
                           (!nu || ((unsigned)(w[2] >> 32) == a.tag_in && (unsigned)(w[3] >> 32) == a.tag_in)) &&
                           (!nv || ((unsigned)(w[4] >> 32) == a.tag_in && (unsigned)(w[5] >> 32) == a.tag_in));
          if (all) { zkv = s2l_ll_val(w[0], w[1]); if (nu) ukv = s2l_ll_val(w[2], w[3]); if (nv) vkv = s2l_ll_val(w[4], w[5]); break; }
          knap();
          if (kclock() - t0 > a.timeout) { *(volatile unsigned long long *)a.err = ((unsigned long long)a.tag_in << 32) | (unsigned long long)(bx + G.nbx2 * by + 1); break; }
        }
        // (the run-time form keeps no point in registers: its later stages read the level again -- from the staging level, where
        // this thread now leaves what arrived; every block whose rectangle holds the point stores the same bits)
        if (!FIXED) { F.zeta[x0 + o_in] = zkv; if (nu) F.ubar[x0 + o_in] = ukv; if (nv) F.vbar[x0 + o_in] = vkv; }
      } else {
        zkv = F.zeta[x0 + o_in]; ukv = F.ubar[x0 + o_in]; vkv = F.vbar[x0 + o_in];
      }
      D0[s0] = zkv + hv;
      U0[s0] = ukv; V0[s0] = vkv; sH[s0] = hv;
      sPm[s0] = F.pm[x0]; sPn[s0] = F.pn[x0];
      const double zsv = F.zeta[x0 + o_kstp];
      sDstp[s0] = zsv + hv;
      sRhoA[s0] = F.rhoA[x0];
      PWSET(r_zk, zkv); PWSET(r_zs, zsv);
      PWSET(r_on_u, F.on_u[x0]); PWSET(r_om_v, F.om_v[x0]); PWSET(r_rhoS, F.rhoS[x0]);
      if (INR(i, j, E.IstrU - 1, E.Iend, E.JstrV - 1, E.Jend)) { PWSET(r_on_u1, F.on_u[GIDX(i + 1, j)]); PWSET(r_om_v1, F.om_v[GIDX(i, j + 1)]); }
      const bool own = INR(i, j, B.Istr, B.Iend, B.Jstr, B.Jend);
      const bool ownR = INR(i, j, KMIN(B.IstrR, B.Istr), B.IendR, KMIN(B.JstrR, B.Jstr), B.JendR);
      if (ownR) {
        PWSET(r_Zt, F.Zt_avg1[x0]); PWSET(r_DU1, F.DU_avg1[x0]); PWSET(r_DU2, F.DU_avg2[x0]);
        PWSET(r_DV1, F.DV_avg1[x0]); PWSET(r_DV2, F.DV_avg2[x0]);
      }
      if (INR(i, j, B.IstrU - 1, B.Iend, B.JstrV - 1, B.Jend)) PWSET(r_rz_p, F.rzeta[x0 + o_ptc]);
      if (a.commit) {
        // the previous pair's result, staged: now the logical level krhs (nobody reads that level in this launch)
        if (fuse) {
          if (own) {
            hb_emit2(G, B, zlog, BC_R, i, j, zkv, MSK ? G.rmask : nullptr, img0);
            if (i >= B.IstrU) hb_emit2(G, B, ulog, BC_U, i, j, ukv, MSK ? G.umask : nullptr, img0);
            if (j >= B.JstrV) hb_emit2(G, B, vlog, BC_V, i, j, vkv, MSK ? G.vmask : nullptr, img0);
          }
        } else if ((own || !INR(i, j, T.Istr, T.Iend, T.Jstr, T.Jend)) && INR(i, j, G.LBi, UBi, G.LBj, UBj) &&
                   (!wrapx || (i >= -2 && i <= G.Lm + G.Nghost)) && (!wrapy || (j >= -2 && j <= G.Mm + G.Nghost))) {
          // own points; boundary and ghost points of the tile by its edge blocks -- at the point's own place in the array
          // (where the tile closes a periodic direction on itself the value was READ at the wrapped index: its image; the
          // reference's periodic ghost zone is -2:0 and Lm+1:Lm+Nghost, the padding column beyond is never written)
          const int xu = (i - G.LBi) + (j - G.LBj) * ni;
          zlog[xu] = zkv; ulog[xu] = ukv; vlog[xu] = vkv;
        }
      }
    } else {
      D0[s0] = 0.0; U0[s0] = 0.0; V0[s0] = 0.0; sH[s0] = 0.0; sPm[s0] = 0.0; sPn[s0] = 0.0; sRhoA[s0] = 0.0; sDstp[s0] = 0.0;
    }
  }
  if (FIXED) {
#pragma unroll
    for (int isv = 0; isv < 2; isv++) {
      WLOOP(isv) {
        const int jj = c / EWD, j = B.Jstr - 2 + jj, i = B.Istr - 2 + c - jj * EWD;
        if (i < i0E || i > i1E || j < j0E || j > j1E || !(isv ? (j >= E.JstrV) : (i >= E.IstrU))) continue;
        const int x = GIDX(i, j);
        const int x1 = isv ? GIDX(i, j - 1) : GIDX(i - 1, j);
        const int q1 = isv ? GIDX(i + 1, j) : GIDX(i, j + 1);
        if (FIXED) wm[WSL(isv)] = isv ? s2_metrics<1>(mr, mp, x, x1, q1) : s2_metrics<0>(mr, mp, x, x1, q1);
        WSET(w_s, isv, (isv ? F.vbar : F.ubar)[x + o_kstp]);
        WSET(w_frc, isv, (isv ? F.rvfrc : F.rufrc)[x]);
        if (MSK) { WSET(w_pk0, isv, G.pmask[x]); WSET(w_pk1, isv, G.pmask[q1]); }
        if (INR(i, j, B.Istr, B.Iend, B.Jstr, B.Jend)) WSET(w_rp, isv, (isv ? F.rvbar : F.rubar)[x + o_ptc]);
      }
    }
  }
  S2P_TICK(1);
  KSYNC();
  S2P_TICK(2);

  // ================================ PREDICTOR on the enlarged sub-tile ==========================
  // ---- stages 2+3: mass fluxes :600-700, fast-time averaging :739-880 (own points), free-surface step :886-1000
  //      (leap-frog, 2*dtfast).  A point forms the fluxes of its own west/south faces (kept in LDS for the momentum
  //      stage) AND those of its east/north faces -- the neighbour's expression, same bits -- so no barrier separates
  //      the fluxes from the free surface.
  {
    const double cA1 = a.w1_m1, cA2 = (8.0 / 12.0) * a.w2_0 - (1.0 / 12.0) * a.w2_p1;
    const double fac = 1000.0 / G.rho0;
    const double cff1z = 2.0 * dtfast, cff4 = 4.0 / 25.0, cff5 = 1.0 - 2.0 * cff4;
    RLOOP(i, j) {
      double du = 0.0, dv = 0.0;
      if (INR(i, j, E.IstrUm2 - 1, E.Iendp2, E.JstrVm2 - 1, E.Jendp2)) {
        if (i >= E.IstrUm2) {
          const double cff = 0.5 * PW(r_on_u, F.on_u[x0]);
          const double cff1 = cff * (D0[s0] + D0[(s0 - 1)]);
          du = U0[s0] * cff1;
          DUon[s0] = du;
        }
        if (j >= E.JstrVm2) {
          const double cff = 0.5 * PW(r_om_v, F.om_v[x0]);
          const double cff1 = cff * (D0[s0] + D0[(s0 - TW)]);
          dv = V0[s0] * cff1;
          DVom[s0] = dv;
        }
      }
      if (INR(i, j, KMIN(B.IstrR, B.Istr), B.IendR, KMIN(B.JstrR, B.Jstr), B.JendR)) {
        const bool pz = i >= B.IstrR && j >= B.JstrR, pu = i >= B.Istr && j >= B.JstrR, pv = i >= B.IstrR && j >= B.Jstr;
        if (pz) F.Zt_avg1[x0] = PW(r_Zt, F.Zt_avg1[x0]) + cA1 * PW(r_zk, F.zeta[x0 + o_in]);
        if (pu) {
          F.DU_avg1[x0] = PW(r_DU1, F.DU_avg1[x0]) + cA1 * du;
          const double v2 = PW(r_DU2, F.DU_avg2[x0]) + cA2 * du;
          F.DU_avg2[x0] = v2;
          PWSET(r_DU2, v2);
        }
        if (pv) {
          F.DV_avg1[x0] = PW(r_DV1, F.DV_avg1[x0]) + cA1 * dv;
          const double v2 = PW(r_DV2, F.DV_avg2[x0]) + cA2 * dv;
          F.DV_avg2[x0] = v2;
          PWSET(r_DV2, v2);
        }
      }
      if (INR(i, j, E.IstrU - 1, E.Iend, E.JstrV - 1, E.Jend)) {
        double du1, dv1;     // DUon(i+1,j), DVom(i,j+1)
        {
          const double cff = 0.5 * PW(r_on_u1, F.on_u[GIDX(i + 1, j)]);
          const double cff1 = cff * (D0[(s0 + 1)] + D0[s0]);
          du1 = U0[(s0 + 1)] * cff1;
        }
        {
          const double cff = 0.5 * PW(r_om_v1, F.om_v[GIDX(i, j + 1)]);
          const double cff1 = cff * (D0[(s0 + TW)] + D0[s0]);
          dv1 = V0[(s0 + TW)] * cff1;
        }
        const double rhs_zeta = (du - du1) + (dv - dv1);
        const double zsv = PW(r_zs, F.zeta[x0 + o_kstp]), zkv = PW(r_zk, F.zeta[x0 + o_in]);
        double zeta_new = zsv + sPm[s0] * sPn[s0] * cff1z * rhs_zeta;
        if (MSK) zeta_new = zeta_new * G.rmask[x0];
        const double zw = cff5 * zkv + cff4 * (zsv + zeta_new);
        const double rhoSv = PW(r_rhoS, F.rhoS[x0]);
        D1[s0] = zeta_new + sH[s0];
        Z1[s0] = zeta_new;
        RZ1[s0] = rhs_zeta;
        zwrk[s0] = zw;
        const double gz = (fac + rhoSv) * zw;
        gzeta[s0] = gz;
        gzeta2[s0] = gz * zw;
        gzetaSA[s0] = zw * (rhoSv - sRhoA[s0]);
        if (INR(i, j, B.Istr, B.Iend, B.Jstr, B.Jend)) {
          if (fuse) {
            if (store3) hb_emit2(G, B, zn3, BC_R, i, j, zeta_new, MSK ? G.rmask : nullptr, true);
            hb_emit2(G, B, rz_k, BC_NONE, i, j, rhs_zeta, nullptr, img1);
          } else {
            zn3[x0] = zeta_new;
            rz_k[x0] = rhs_zeta;
          }
        } else if (!fuse && ina && !INR(i, j, T.Istr, T.Iend, T.Jstr, T.Jend)) {
          rz_k[x0] = rhs_zeta;       // ghost points of the tile: what the neighbour computes there (:1030 exchanges it)
        }
      }
    }
  }
  KSYNC();
  S2P_TICK(3);
  S2P_TICK(4);
  // ---- stage 4: momentum on the enlarged sub-tile :1080-2670 ----------------------------------
  {
    const S2Tiles Tl = {U0, V0, DUon, DVom, D0, sPm, sPn, sH, sRhoA, gzeta, gzeta2, gzetaSA, zwrk, TW};
    // closed-edge replication of the advection gradients: by DOMAIN lines (a sub-tile that does not touch the edge can
    // still reach it with the stencil of a rim point; for one that does, Istr = 1 ... as in k_step2d.h)
    const S2Edge Eg = {!G.ewp, !G.ewp, !G.nsp, !G.nsp, 1, G.Lm, 1, G.Mm};
    const double c1 = dtfast;
#pragma unroll
    for (int isv = 0; isv < 2; isv++) {
      WLOOP(isv) {
        const int jj = c / EWD, j = B.Jstr - 2 + jj, i = B.Istr - 2 + c - jj * EWD;
        if (i < i0E || i > i1E || j < j0E || j > j1E || !(isv ? (j >= E.JstrV) : (i >= E.IstrU))) continue;
        const int s = (i - IT0) + (j - JT0) * TW;
        const int x = GIDX(i, j);
        const int d1 = isv ? TW : 1;
        S2Met M;
        if (FIXED) M = wm[WSL(isv)];
        else {
          const int x1 = isv ? GIDX(i, j - 1) : GIDX(i - 1, j), q1 = isv ? GIDX(i + 1, j) : GIDX(i, j + 1);
          M = isv ? s2_metrics<1>(mr, mp, x, x1, q1) : s2_metrics<0>(mr, mp, x, x1, q1);
        }
        double pk0 = 1.0, pk1 = 1.0;
        if (MSK) {
          pk0 = WV(w_pk0, isv, G.pmask[x]);
          pk1 = WV(w_pk1, isv, G.pmask[isv ? GIDX(i + 1, j) : GIDX(i, j + 1)]);
        }
        const double rhs = isv ? s2_rhs<1>(Tl, M, Eg, s, i, j, g, ADV, COR, CURV, VIS, MSK, pk0, pk1)
                               : s2_rhs<0>(Tl, M, Eg, s, i, j, g, ADV, COR, CURV, VIS, MSK, pk0, pk1);
        const double r = rhs + WV(w_frc, isv, (isv ? F.rvfrc : F.rufrc)[x]);
        const double cff = (sPm[s] + sPm[s - d1]) * (sPn[s] + sPn[s - d1]);
        const double fac = 1.0 / (D1[s] + D1[s - d1]);
        const double Dstp0 = sDstp[s], Dstp1 = sDstp[s - d1];
        const double sv = WV(w_s, isv, (isv ? F.vbar : F.ubar)[x + o_kstp]);
        double b = (sv * (Dstp0 + Dstp1) + cff * c1 * r) * fac;
        if (MSK) b = b * (isv ? G.vmask : G.umask)[x];
        (isv ? V1 : U1)[s] = b;
        WSET(w_rP, isv, r);
        const bool own = INR(i, j, B.Istr, B.Iend, B.Jstr, B.Jend) && (isv ? (j >= B.JstrV) : (i >= B.IstrU));
        if (own) {
          if (!isv) {
            if (fuse) { if (store3) hb_emit2(G, B, un3, BC_U, i, j, b, MSK ? G.umask : nullptr, true); }
            else un3[x] = b;
            rub_k[x] = r;
          } else {
            if (fuse) { if (store3) hb_emit2(G, B, vn3, BC_V, i, j, b, MSK ? G.vmask : nullptr, true); }
            else vn3[x] = b;
            rvb_k[x] = r;
          }
        }
      }
    }
  }
  KSYNC();
  S2P_TICK(5);
  // ---- closed domain edges: zetabc / u2dbc / v2dbc of the predictor's result, on the LDS tiles (halo_fill's phases) --
  {
    const bool cw = !G.ewp && E.west, ce = !G.ewp && E.east, cs = !G.nsp && E.south, cn = !G.nsp && E.north;
    if (cw || ce || cs || cn) {
#define LA(A_, i_, j_) A_[((i_) - IT0) + ((j_) - JT0) * TW]
#define MK_(M_, i_, j_) (MSK ? G.M_[GIDX(i_, j_)] : 1.0)
      const int Lm = G.Lm, Mm = G.Mm;
      const double gamma2 = G.gamma2;
      // rows / columns of the rectangle that hold the predictor's values
      const int zj0 = E.JstrV - 1, zj1 = E.Jend, zi0 = E.IstrU - 1, zi1 = E.Iend;
      // zeta: gradient condition
      if (cw) KLOOP1(j, zj0, zj1) { const double v = LA(Z1, 1, j) * MK_(rmask, 0, j); LA(Z1, 0, j) = v; LA(D1, 0, j) = v + LA(sH, 0, j); }
      if (ce) KLOOP1(j, zj0, zj1) { const double v = LA(Z1, Lm, j) * MK_(rmask, Lm + 1, j); LA(Z1, Lm + 1, j) = v; LA(D1, Lm + 1, j) = v + LA(sH, Lm + 1, j); }
      if (cs) KLOOP1(i, zi0, zi1) { const double v = LA(Z1, i, 1) * MK_(rmask, i, 0); LA(Z1, i, 0) = v; LA(D1, i, 0) = v + LA(sH, i, 0); }
      if (cn) KLOOP1(i, zi0, zi1) { const double v = LA(Z1, i, Mm) * MK_(rmask, i, Mm + 1); LA(Z1, i, Mm + 1) = v; LA(D1, i, Mm + 1) = v + LA(sH, i, Mm + 1); }
      // ubar: zero normal flow at the western/eastern walls, then gamma2 slip along the southern/northern ones
      if (cw) KLOOP1(j, j0E, j1E) LA(U1, 1, j) = 0.0;
      if (ce) KLOOP1(j, j0E, j1E) LA(U1, Lm + 1, j) = 0.0;
      // vbar: zero normal flow at the southern/northern walls (the reference's wall values are zero when its slip fill reads them)
      if (cs) KLOOP1(i, i0E, i1E) LA(V1, i, 1) = 0.0;
      if (cn) KLOOP1(i, i0E, i1E) LA(V1, i, Mm + 1) = 0.0;
      KSYNC();
      {
        const int ui0 = cw ? 1 : E.IstrU, ui1 = ce ? Lm + 1 : i1E;
        if (cs) KLOOP1(i, ui0, ui1) LA(U1, i, 0) = gamma2 * LA(U1, i, 1) * MK_(umask, i, 0);
        if (cn) KLOOP1(i, ui0, ui1) LA(U1, i, Mm + 1) = gamma2 * LA(U1, i, Mm) * MK_(umask, i, Mm + 1);
        const int vj0 = cs ? 1 : E.JstrV, vj1 = cn ? Mm + 1 : j1E;
        if (cw) KLOOP1(j, vj0, vj1) LA(V1, 0, j) = gamma2 * LA(V1, 1, j) * MK_(vmask, 0, j);
        if (ce) KLOOP1(j, vj0, vj1) LA(V1, Lm + 1, j) = gamma2 * LA(V1, Lm, j) * MK_(vmask, Lm + 1, j);
      }
      KSYNC();
      // corners of a closed basin (bc_2d.F / zetabc.F corner averages)
      if (!(G.ewp || G.nsp) && KTID == 0) {
        if (cw && cs) {
          { const double v = 0.5 * (LA(Z1, 1, 0) + LA(Z1, 0, 1)); LA(Z1, 0, 0) = v; LA(D1, 0, 0) = v + LA(sH, 0, 0); }
          LA(U1, 1, 0) = 0.5 * (LA(U1, 2, 0) + LA(U1, 1, 1));
          LA(V1, 0, 1) = 0.5 * (LA(V1, 1, 1) + LA(V1, 0, 2));
        }
        if (ce && cs) {
          { const double v = 0.5 * (LA(Z1, Lm, 0) + LA(Z1, Lm + 1, 1)); LA(Z1, Lm + 1, 0) = v; LA(D1, Lm + 1, 0) = v + LA(sH, Lm + 1, 0); }
          LA(U1, Lm + 1, 0) = 0.5 * (LA(U1, Lm, 0) + LA(U1, Lm + 1, 1));
          LA(V1, Lm + 1, 1) = 0.5 * (LA(V1, Lm, 1) + LA(V1, Lm + 1, 2));
        }
        if (cw && cn) {
          { const double v = 0.5 * (LA(Z1, 0, Mm) + LA(Z1, 1, Mm + 1)); LA(Z1, 0, Mm + 1) = v; LA(D1, 0, Mm + 1) = v + LA(sH, 0, Mm + 1); }
          LA(U1, 1, Mm + 1) = 0.5 * (LA(U1, 1, Mm) + LA(U1, 2, Mm + 1));
          LA(V1, 0, Mm + 1) = 0.5 * (LA(V1, 0, Mm) + LA(V1, 1, Mm + 1));
        }
        if (ce && cn) {
          { const double v = 0.5 * (LA(Z1, Lm + 1, Mm) + LA(Z1, Lm, Mm + 1)); LA(Z1, Lm + 1, Mm + 1) = v; LA(D1, Lm + 1, Mm + 1) = v + LA(sH, Lm + 1, Mm + 1); }
          LA(U1, Lm + 1, Mm + 1) = 0.5 * (LA(U1, Lm + 1, Mm) + LA(U1, Lm, Mm + 1));
          LA(V1, Lm + 1, Mm + 1) = 0.5 * (LA(V1, Lm + 1, Mm) + LA(V1, Lm, Mm + 1));
        }
      }
      KSYNC();
#undef LA
#undef MK_
    }
  }

  S2P_TICK(6);
  // ================================ CORRECTOR on the own sub-tile ===============================
  // krhs = 3: D1 = zeta(3)+h, U1, V1 (LDS); kstp = the predictor's krhs: D0, U0, V0; ptsk = the predictor's kstp
  // ---- stages 2+3: mass fluxes, the fast-time average of the corrector's fluxes, free-surface step (AM3) --------
  {
    const double cA2 = (5.0 / 12.0) * a.w2_0;
    const double fac = 1000.0 / G.rho0;
    const double cff1 = dtfast * 5.0 / 12.0, cff2 = dtfast * 8.0 / 12.0, cff3 = dtfast * 1.0 / 12.0, cff4 = 2.0 / 5.0, cff5 = 1.0 - cff4;
    RLOOP(i, j) {
      double du = 0.0, dv = 0.0;
      if (INR(i, j, B.IstrUm2 - 1, B.Iendp2, B.JstrVm2 - 1, B.Jendp2)) {
        if (i >= B.IstrUm2) {
          const double cff = 0.5 * PW(r_on_u, F.on_u[x0]);
          const double cff1f = cff * (D1[s0] + D1[(s0 - 1)]);
          du = U1[s0] * cff1f;
          DUon[s0] = du;
        }
        if (j >= B.JstrVm2) {
          const double cff = 0.5 * PW(r_om_v, F.om_v[x0]);
          const double cff1f = cff * (D1[s0] + D1[(s0 - TW)]);
          dv = V1[s0] * cff1f;
          DVom[s0] = dv;
        }
      }
      if (INR(i, j, KMIN(B.IstrR, B.Istr), B.IendR, KMIN(B.JstrR, B.Jstr), B.JendR)) {
        const bool pu = i >= B.Istr && j >= B.JstrR, pv = i >= B.IstrR && j >= B.Jstr;
        if (pu) F.DU_avg2[x0] = PW(r_DU2, F.DU_avg2[x0]) + cA2 * du;
        if (pv) F.DV_avg2[x0] = PW(r_DV2, F.DV_avg2[x0]) + cA2 * dv;
      }
      if (INR(i, j, B.IstrU - 1, B.Iend, B.JstrV - 1, B.Jend)) {
        double du1, dv1;
        {
          const double cff = 0.5 * PW(r_on_u1, F.on_u[GIDX(i + 1, j)]);
          const double cff1f = cff * (D1[(s0 + 1)] + D1[s0]);
          du1 = U1[(s0 + 1)] * cff1f;
        }
        {
          const double cff = 0.5 * PW(r_om_v1, F.om_v[GIDX(i, j + 1)]);
          const double cff1f = cff * (D1[(s0 + TW)] + D1[s0]);
          dv1 = V1[(s0 + TW)] * cff1f;
        }
        const double rhs_zeta = (du - du1) + (dv - dv1);
        const double zsv = PW(r_zk, F.zeta[x0 + o_in]), zkv = Z1[s0];
        const double cff = cff1 * rhs_zeta;
        double zeta_new = zsv + sPm[s0] * sPn[s0] * (cff + cff2 * RZ1[s0] - cff3 * PW(r_rz_p, F.rzeta[x0 + o_ptc]));
        if (MSK) zeta_new = zeta_new * G.rmask[x0];
        const double zw = cff5 * zeta_new + cff4 * zkv;
        const double rhoSv = PW(r_rhoS, F.rhoS[x0]);
        D2[s0] = zeta_new + sH[s0];
        zwrk[s0] = zw;
        const double gz = (fac + rhoSv) * zw;
        gzeta[s0] = gz;
        gzeta2[s0] = gz * zw;
        gzetaSA[s0] = zw * (rhoSv - sRhoA[s0]);
        if (i >= B.Istr && j >= B.Jstr) {
          if (fuse) hb_emit2(G, B, zout, BC_R, i, j, zeta_new, MSK ? G.rmask : nullptr, img0);
          else if (a.rim_out) {
            hb_emit2(G, B, zout, BC_R, i, j, zeta_new, MSK ? G.rmask : nullptr, false);
            s2l_remit(G, a.P, B, 3 * (int)(a.tag_out & 3u), BC_R, i, j, zeta_new, a.tag_out, MSK ? G.rmask : nullptr);
          } else zout[x0] = zeta_new;
        }
      }
    }
  }
  KSYNC();
  S2P_TICK(7);
  S2P_TICK(8);
  // ---- stage 4: momentum on the own sub-tile ----------------------------------------------------
  {
    const S2Tiles Tl = {U1, V1, DUon, DVom, D1, sPm, sPn, sH, sRhoA, gzeta, gzeta2, gzetaSA, zwrk, TW};
    // closed-edge replication of the advection gradients: by DOMAIN lines (a sub-tile that does not touch the edge can
    // still reach it with the stencil of a rim point; for one that does, Istr = 1 ... as in k_step2d.h)
    const S2Edge Eg = {!G.ewp, !G.ewp, !G.nsp, !G.nsp, 1, G.Lm, 1, G.Mm};
    const double k1 = 0.5 * dtfast * 5.0 / 12.0, k2 = 0.5 * dtfast * 8.0 / 12.0, k3 = 0.5 * dtfast * 1.0 / 12.0;
#pragma unroll
    for (int isv = 0; isv < 2; isv++) {
      WLOOP(isv) {
        const int jj = c / EWD, j = B.Jstr - 2 + jj, i = B.Istr - 2 + c - jj * EWD;
        if (!INR(i, j, B.Istr, B.Iend, B.Jstr, B.Jend) || !(isv ? (j >= B.JstrV) : (i >= B.IstrU))) continue;
        const int s = (i - IT0) + (j - JT0) * TW;
        const int x = (i - G.LBi) + (j - G.LBj) * ni;
        const int d1 = isv ? TW : 1;
        S2Met M;
        if (FIXED) M = wm[WSL(isv)];
        else {
          const int x1 = isv ? GIDX(i, j - 1) : GIDX(i - 1, j), q1 = isv ? GIDX(i + 1, j) : GIDX(i, j + 1);
          M = isv ? s2_metrics<1>(mr, mp, x, x1, q1) : s2_metrics<0>(mr, mp, x, x1, q1);
        }
        double pk0 = 1.0, pk1 = 1.0;
        if (MSK) {
          pk0 = WV(w_pk0, isv, G.pmask[x]);
          pk1 = WV(w_pk1, isv, G.pmask[isv ? GIDX(i + 1, j) : GIDX(i, j + 1)]);
        }
        const double rhs = isv ? s2_rhs<1>(Tl, M, Eg, s, i, j, g, ADV, COR, CURV, VIS, MSK, pk0, pk1)
                               : s2_rhs<0>(Tl, M, Eg, s, i, j, g, ADV, COR, CURV, VIS, MSK, pk0, pk1);
        const double r = rhs + WV(w_frc, isv, (isv ? F.rvfrc : F.rufrc)[x]);
        const double cff = (sPm[s] + sPm[s - d1]) * (sPn[s] + sPn[s - d1]);
        const double fac = 1.0 / (D2[s] + D2[s - d1]);
        const double Dstp0 = D0[s], Dstp1 = D0[s - d1];
        const double sv = (isv ? V0 : U0)[s];
        const double rs = WV(w_rP, isv, (isv ? rvb_k : rub_k)[x]);
        const double rp = WV(w_rp, isv, (isv ? F.rvbar : F.rubar)[x + o_ptc]);
        double b = (sv * (Dstp0 + Dstp1) + cff * (k1 * r + k2 * rs - k3 * rp)) * fac;
        if (MSK) b = b * (isv ? G.vmask : G.umask)[x];
        if (!isv) {
          if (fuse) hb_emit2(G, B, uout, BC_U, i, j, b, MSK ? G.umask : nullptr, img0);
          else if (a.rim_out) {
            hb_emit2(G, B, uout, BC_U, i, j, b, MSK ? G.umask : nullptr, false);
            s2l_remit(G, a.P, B, 3 * (int)(a.tag_out & 3u) + 1, BC_U, i, j, b, a.tag_out, MSK ? G.umask : nullptr);
          } else uout[x] = b;
        } else {
          if (fuse) hb_emit2(G, B, vout, BC_V, i, j, b, MSK ? G.vmask : nullptr, img0);
          else if (a.rim_out) {
            hb_emit2(G, B, vout, BC_V, i, j, b, MSK ? G.vmask : nullptr, false);
            s2l_remit(G, a.P, B, 3 * (int)(a.tag_out & 3u) + 2, BC_V, i, j, b, a.tag_out, MSK ? G.vmask : nullptr);
          } else vout[x] = b;
        }
      }
    }
  }
  S2P_TICK(9);
}
#undef RLOOP
#undef WRAPI
#undef WRAPJ
#undef GIDX
#undef PWDECL
#undef PWSET
#undef PW
#undef WLOOP
#undef WSL
#undef WDECL
#undef WSET
#undef WV
#undef INR

// entry points: sub-tiles up to 32x4 (640 threads: one rectangle point per thread; the u-points of the enlarged
// sub-tile on waves 0-4, its v-points on waves 5-9), and the generic form (any sub-tile shape; the CPU emulation)
COOP_KERNEL(k_step2d_pair_a, Step2dPairArgs) { k_step2d_pair_t_body<32, 4, 640>(a, bx, by, bz, lds); }
COOP_GLOBAL_LB(k_step2d_pair_a, Step2dPairArgs, 640)
COOP_KERNEL(k_step2d_pair, Step2dPairArgs) { k_step2d_pair_t_body<0, 0, 0>(a, bx, by, bz, lds); }
COOP_GLOBAL_LB(k_step2d_pair, Step2dPairArgs, 512)
