// k_step2d_loop.h -- the barotropic fast steps iif = 2 .. nfast (main3d.F:810-918) as ONE persistent launch.
//
// Replaces the 2*(nfast-1) calls of step2d_tile (ROMS/Nonlinear/step2d_LF_AM3.h:163-3056) that k_step2d_pair.h runs as
// nfast-1 launches.  On a tile of up to 64 K points (BENCHMARK1: 512x64 = 256 sub-tiles of 32x4 points, one block per
// CU) a pair launch is a latency chain: 3.8 us to bring the 42x14 rectangle of 8 fields and the metric records of its
// momentum points into LDS / registers, ~9 us of dependent stages, 3.5 us of launch + drain -- 28 times per baroclinic
// step, 58 % of it.  Here a block KEEPS its sub-tile: the time-invariant tiles (h, pm, pn, rhoA), the metric records, the
// forcing and the fast-time averages stay in LDS / registers for all pairs, the time levels rotate between two sets of
// LDS tiles the way the reference rotates kstp/krhs/knew, and per pair only the corrector's result on the 5-line rim
// (zeta, ubar, vbar of 460 points) crosses between blocks:
//
//   producer   stores its 32x4 own points (and the boundary values a closed edge derives from them, k_haloblock.h) of
//              the corrector's zeta/ubar/vbar WRITE-THROUGH (relaxed agent-scope stores = global_store sc1) into the
//              staging level of the pair (physical levels 4 | 5 of zeta/ubar/vbar, alternating -- the levels the pair
//              kernel stages its result in); every wave drains (s_waitcnt vmcnt(0)), barrier, ONE lane stores the
//              block's arrival word = pair number
//   consumer   one wave polls the arrival words of the blocks whose own points its rectangle touches (relaxed agent
//              loads, s_sleep), barrier, then every rim thread loads its point with sc1 loads (served by L2 / fabric:
//              the CU's L1 is bypassed, so no acquire fence is needed behind write-through stores --
//              MI355X guide "inter-workgroup visibility": {sc1 stores + drained flag | relaxed poll + sc1 loads})
//
// measured (tools/gpu_debug/rimx_probe.hip, this geometry): 2.2 us per exchange against 9.5 us with release / acquire
// fences around plain stores.  A level is overwritten two pairs after it was published; a block gets there only after it
// has consumed the NEXT pair's result of every block that reads its points (the reading relation is symmetric), so two
// staging levels suffice.  Every spin is bounded (Step2dLoopArgs::timeout ticks of the 100 MHz wall clock): a block
// that gives up reports through the pinned word `err`, stops waiting and keeps publishing, so the launch always ends;
// the host turns the word into exit_flag 2 (roms_hip.cpp: ctx_check).
//
// Every value is computed with the expressions of k_step2d_pair.h (which are those of k_step2d.h): bit-identical to
// the pair launches and to the per-call kernel (tests/test_gpu_parity.py: test_step2d_forms_...).  What the pair
// launches store to global memory only for a LATER launch to read is not stored at all; what is left behind when the
// loop ends is stored by the last pairs exactly as the last pair launches do (`tail`): the logical levels 1, 2 (the
// results of pairs nfast-2 and nfast-1), level 3 (the last predictor), rzeta/rubar/rvbar of both levels, the fast-time
// averages, the last result staged for the auxiliary call iif = nfast+1 (k_step2d_ac commits it).
//
// Single tile, fused boundary fills (at least one periodic direction), sub-tiles up to 32x4, no land mask: the
// configuration of BASELINE's 1-GPU headline.
//
// Multi-tile contexts (round 6, template parameter MT; mp_exchange2d of step2d_LF_AM3.h:714,842,1068,3041 inside the launch):
// the rim of an edge block reaches into the NEIGHBOURING RANK's tile.  Every rank keeps, in its mailbox slab (uncached
// memory the neighbours map over hipIpc / xGMI, roms_hip.cpp), four sets x {zeta, ubar, vbar} of RIM PLANES indexed like its
// own arrays, 16 bytes per point: two 8-byte words {low half of the value | pair number}, {high half | pair number}.  An
// 8-byte store is atomic on every path (HBM, xGMI), so a value can be polled for directly -- no arrival word, no drain
// between data and flag (the protocol RCCL calls LL).  Per pair
//   producer   behind its local publication (sc1 stores, drain, arrival word: the blocks of its own tile go on at once) an
//              edge block stores the own points that lie in a neighbour's ghost zone -- 4 lines towards the low side, 5
//              towards the high side, corners included; the boundary values a closed domain edge derives from them as
//              well -- into that neighbour's rim planes at the index the point has THERE (system-scope stores)
//   consumer   every thread whose rectangle point is a ghost point of the tile polls ITS point in the rim planes until both
//              words of all three fields carry this pair's number (bounded like the other waits), beside the wave that
//              polls the arrival words of the tile's own blocks
// FOUR sets of rim planes take turns (pair & 3).  Inside a tile two staging levels suffice because the reading relation between
// blocks is symmetric; across a tile edge it is not quite (a block two sub-tile rows from the edge reads the neighbour's last
// line, and the neighbour's edge block waits for nothing of its), so the set a block overwrites is the one of four pairs ago: the
// publisher of pair n+4 has consumed n+3 of the neighbour's edge blocks, those n+2 of every block beside them -- which therefore
// is through with pair n+1, the one that read n (k_step2d_pair.h: Step2dPairArgs has the same argument for launches).  Between two launches the 3-D exchanges of the baroclinic step order the ranks (a rank cannot start the next launch
// before every neighbour has left this one).  Tiles of equal size on every rank, at least one periodic direction.  What
// the pair launches exchange between the pairs is exchanged ONCE behind the launch (g_step2d.cpp).  Measured (one MI355X,
// the tile its own W/E neighbour through the mailbox): an arrival-word form -- stores, drain, word in the neighbour's
// ring, poll, loads: four uncached round trips -- cost 2.7 us per pair (442 us per launch against 363 single-tile).
// Everything else keeps the pair / per-call launches.
#pragma once
#include "k_step2d_pair.h"

#ifndef ROMS_CPU_EMU
#define S2L_NLDS 32
#define S2L_NLDS_MK 35           // + rmask, umask, vmask on the rectangle (MASKING)
#define INR(i, j, i0, i1, j0, j1) ((i) >= (i0) && (i) <= (i1) && (j) >= (j0) && (j) <= (j1))
#define S2L_FSTRIDE 16          // arrival words are 64 bytes apart
struct Step2dLoopArgs {
  S2Fields F;                   // (first: DESIGN.md 6)
  DGrid G;                      // stepping of the FIRST pair's predictor call: iif = 2, kstp = 3 - indx1, krhs = indx1, knew = 3
  const double *wts;            // [3 iif + ...], iif = 1 .. nfast+1: weight(1,iif-1), weight(2,iif), weight(2,iif+1)
  unsigned *flags;              // arrival words [sub-tile * S2L_FSTRIDE]: they only grow -- pair q of this launch is epoch + q + 1, the
                                // launches of a context count on from each other (no reset between them: g_step2d.cpp)
  unsigned epoch;
  unsigned long long *err;      // pinned host word: (pair << 32 | sub-tile + 1) of a wait that gave up
  long long timeout;            // ... after this many wall_clock64 ticks
  int npairs;                   // fast steps in this launch: nfast - 1 (iif = 2 .. nfast), or nfast with the first one
  int first;                    // the launch starts with iif = 1 (G: the stepping of ITS predictor call): forward-Euler start,
                                // conversion of the 3-D forcing (step2d_LF_AM3.h:2225-2460), reset of the fast-time averages
  int aux;                      // ... and ends with the auxiliary predictor call iif = nfast+1 (:821-883): final averages, commit
  int nstp, nnew, startup;      // (first) baroclinic time levels of ru, rv(:,:,0,:); 0 | 1 | 2 = iic - ntfirst capped (Euler, AB2, AB3)
  // uniform factors of the two calls, formed on the host with the kernels' expressions (IEEE division either way): in the
  // kernel they would be re-derived -- with f64 divisions -- in every pair, or held in vector registers across the loop
  double kfac;                  // 1000 / rho0
  double kz1, kz2, kz3;         // dtfast*5/12, dtfast*8/12, dtfast*1/12 (corrector, free surface)
  double km1, km2, km3;         // 0.5*dtfast*5/12, 0.5*dtfast*8/12, 0.5*dtfast*1/12 (corrector, momentum)
  int wrapx, wrapy;             // rim indices beyond the tile wrap onto the tile's own points
  int prio;                     // s_setprio of the block's waves (ROMS_HIP_LOOP_PRIO, default 3: as the pair kernel)
  S2LPeer P;                    // (last: the members above keep their places in the argument block)
};

KDEV double s2l_ld(const double *p) {
  return __longlong_as_double((long long)__hip_atomic_load((const unsigned long long *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
// own points (i0:i1, j0:j1) of sub-tile (bx,by): block_bounds_n without the loop-bound table
KDEV void s2l_range(const DGrid &G, int bx, int by, int &i0, int &i1, int &j0, int &j1) {
  const int LmT = G.T.Iend - G.T.Istr + 1, MmT = G.T.Jend - G.T.Jstr + 1;
  const int cI = (LmT + G.nbx2 - 1) / G.nbx2, cJ = (MmT + G.nby2 - 1) / G.nby2;
  const int mI = (G.nbx2 * cI - LmT) / 2, mJ = (G.nby2 * cJ - MmT) / 2;
  i0 = 1 + bx * cI - mI; i1 = i0 + cI - 1;
  j0 = 1 + by * cJ - mJ; j1 = j0 + cJ - 1;
  i0 = KMAX(i0, 1) + G.T.Istr - 1; i1 = KMIN(i1, LmT) + G.T.Istr - 1;
  j0 = KMAX(j0, 1) + G.T.Jstr - 1; j1 = KMIN(j1, MmT) + G.T.Jstr - 1;
}

// an SGPR zero the compiler cannot see through: indices derived from it are not hoisted out of the stage / the pair loop
#define S2L_OPQ(name) int name = 0; asm volatile("" : "+s"(name))
#define S2L_TICK(n) do { if (G.dbg_stop == 98 && p == 3 && t == 0) F.xr[me * 16 + (n)] = (double)wall_clock64(); } while (0)

template <int BWC, int BHC, int NTC, bool MT = false, bool MK = false>
static __device__ __forceinline__ void k_step2d_loop_body(const Step2dLoopArgs &a, int bx, int by, double *lds) {
  if (a.prio == 3) __builtin_amdgcn_s_setprio(3);
  else if (a.prio == 1) __builtin_amdgcn_s_setprio(1);
  const DGrid &G = a.G;
  const S2Fields &F = a.F;
  xcd_remap2(G, bx, by);
  const int me = bx + G.nbx2 * by;
  const TB B = block_bounds2(G, bx, by);
  const bool wrapx = a.wrapx != 0, wrapy = a.wrapy != 0;
  // the enlarged sub-tile of the predictor phase (k_step2d_pair.h)
  int i0E = B.Istr - 2, i1E = B.Iend + 2, j0E = B.Jstr - 2, j1E = B.Jend + 2;
  if (!G.ewp) { i0E = KMAX(i0E, 1); i1E = KMIN(i1E, G.Lm); }
  if (!G.nsp) { j0E = KMAX(j0E, 1); j1E = KMIN(j1E, G.Mm); }
  const TB E = make_bounds(G.Lm, G.Mm, G.ewp, G.nsp, i0E, i1E, j0E, j1E, i0E <= 1, i1E >= G.Lm, j0E <= 1, j1E >= G.Mm);
  constexpr int EWD = BWC + 4, EHT = BHC + 4, NE = EWD * EHT;
  constexpr int TW = BWC + 2 * S2P_RIM, TH = BHC + 2 * S2P_RIM, NTILE = TW * TH;
  constexpr int VOFF = (NE + 63) / 64 * 64;
  static_assert(NTILE <= NTC && 2 * VOFF <= NTC, "one rectangle point and one momentum point per thread");
  const size_t sz = (size_t)NTILE;
  // two sets of time-level tiles (zeta + h, ubar, vbar, zeta): P = the level the next predictor reads as krhs, Q = its
  // kstp level, dead behind the predictor's momentum stage -- the corrector's result goes there, and the sets swap
  double *DP = lds, *UP = lds + sz, *VP = lds + 2 * sz, *ZP = lds + 3 * sz;
  double *DQ = lds + 4 * sz, *UQ = lds + 5 * sz, *VQ = lds + 6 * sz, *ZQ = lds + 7 * sz;
  double *sH = lds + 8 * sz, *sPm = lds + 9 * sz, *sPn = lds + 10 * sz, *sRhoA = lds + 11 * sz;
  double *sRhoS = lds + 12 * sz, *sOnu = lds + 13 * sz, *sOmv = lds + 14 * sz;
  double *DUon = lds + 15 * sz, *DVom = lds + 16 * sz, *D1 = lds + 17 * sz;
  double *zwrk = lds + 18 * sz, *gzeta = lds + 19 * sz, *gzeta2 = lds + 20 * sz, *gzetaSA = lds + 21 * sz;
  double *Z1 = lds + 22 * sz, *U1 = lds + 23 * sz, *V1 = lds + 24 * sz;
  // values a rectangle point keeps for itself between the stages and the pairs, at its own slot (in registers they
  // would be live across the momentum stages, whose own demand fills the budget of three waves per SIMD): rhs_zeta of
  // this pair's predictor | of the previous one (= rzeta(ptsk)), and the five fast-time averages
  double *RZ1 = lds + 25 * sz, *RZP = lds + 26 * sz;
  double *aZt = lds + 27 * sz, *aDU1 = lds + 28 * sz, *aDU2 = lds + 29 * sz, *aDV1 = lds + 30 * sz, *aDV2 = lds + 31 * sz;
  // MASKING (template parameter MK; the masked statements of k_step2d_pair.h): the land/sea masks of the rectangle, the no-slip
  // factors pmask of the momentum point's two psi points in registers
  double *sRm = lds + (MK ? 32 : 0) * sz, *sUm = lds + (MK ? 33 : 0) * sz, *sVm = lds + (MK ? 34 : 0) * sz;
  const double *Mr = MK ? G.rmask : nullptr, *Mu = MK ? G.umask : nullptr, *Mv = MK ? G.vmask : nullptr;
  double pk0 = 1.0, pk1 = 1.0;
  const double dtfast = G.dtfast, g = G.g;
  const int ni = G.ni, nij = (int)G.nij;
  const int UBi = G.LBi + G.ni - 1, UBj = G.LBj + G.nj - 1;
  const M2Rec *mr = (const M2Rec *)(double *)F.m2r, *mp = (const M2Rec *)(double *)F.m2p;
  const int IT0 = B.Istr - S2P_RIM, JT0 = B.Jstr - S2P_RIM;
  const bool ADV = (G.options & ROMS_UV_ADV) != 0, COR = (G.options & ROMS_UV_COR) != 0;
  const bool CURV = ADV && (G.options & ROMS_CURVGRID) != 0, VIS = (G.options & ROMS_UV_VIS2) != 0;
  const int t = KTID;
#define WRAPI(i_) (wrapx ? ((i_) < 1 ? (i_) + G.Lm : ((i_) > G.Lm ? (i_) - G.Lm : (i_))) : (i_))
#define WRAPJ(j_) (wrapy ? ((j_) < 1 ? (j_) + G.Mm : ((j_) > G.Mm ? (j_) - G.Mm : (j_))) : (j_))
#define GIDX(i_, j_) ((WRAPI(i_) - G.LBi) + (WRAPJ(j_) - G.LBj) * ni)
  // ---- this thread's rectangle point ...
  const bool rp = t < NTILE;
  const int s0_ = t, jj0 = t / TW, j_ = JT0 + jj0, i_ = IT0 + t - jj0 * TW;
  const int iw = WRAPI(i_), jw = WRAPJ(j_);
  const bool ina = rp && iw >= G.LBi && iw <= UBi && jw >= G.LBj && jw <= UBj;
  const int x0_ = (iw - G.LBi) + (jw - G.LBj) * ni;
  const bool own = rp && INR(i_, j_, B.Istr, B.Iend, B.Jstr, B.Jend);
  const bool ownR = rp && INR(i_, j_, KMIN(B.IstrR, B.Istr), B.IendR, KMIN(B.JstrR, B.Jstr), B.JendR);
  const bool inEz = rp && INR(i_, j_, E.IstrU - 1, E.Iend, E.JstrV - 1, E.Jend);       // the predictor's free-surface points
  const bool inBz = rp && INR(i_, j_, B.IstrU - 1, B.Iend, B.JstrV - 1, B.Jend);       // the corrector's
  // (multi-tile) a rectangle point outside the tile and the boundary points of its closed domain edges: a ghost point, its
  // value comes from a neighbouring rank's block through my rim planes
  const bool rem = MT && rp && !INR(iw, jw, G.T.IstrR, G.T.IendR, G.T.JstrR, G.T.JendR);
  // ... and one the neighbours do write: inside the arrays, and inside the strips they publish (5 lines on my low side, 4 on my
  // high side, towards a side that has a neighbour)
  const bool remv = rem && ina && INR(iw, jw, (a.P.nbmask & 1) ? G.T.Istr - B2D_GL : G.T.IstrR, (a.P.nbmask & 2) ? G.T.Iend + B2D_GH : G.T.IendR,
                                      (a.P.nbmask & 4) ? G.T.Jstr - B2D_GL : G.T.JstrR, (a.P.nbmask & 8) ? G.T.Jend + B2D_GH : G.T.JendR);
  const bool IMG = !MT;                                                                 // periodic images: a single tile stores its own
  // ---- ... and momentum point: cell c of the enlarged sub-tile, its u-point (threads 0..VOFF-1) or v-point (VOFF..)
  const int isvt = t >= VOFF ? 1 : 0, c = t - isvt * VOFF;
  const int mjj = c / EWD, mj_ = B.Jstr - 2 + mjj, mi_ = B.Istr - 2 + c - mjj * EWD;
  const bool mE = c < NE && !(mi_ < i0E || mi_ > i1E || mj_ < j0E || mj_ > j1E) && (isvt ? (mj_ >= E.JstrV) : (mi_ >= E.IstrU));
  const bool mO = c < NE && INR(mi_, mj_, B.Istr, B.Iend, B.Jstr, B.Jend) && (isvt ? (mj_ >= B.JstrV) : (mi_ >= B.IstrU));
  const int ms_ = (mi_ - IT0) + (mj_ - JT0) * TW;
  const int mx_ = GIDX(mi_, mj_);

  S2Met wm;
  double w_frc = 0.0, w_rp = 0.0, w_rP = 0.0;

  // ---- prologue: every global read of the first pair (k_step2d_pair.h stage 1) ------------------
  {
    const int o_in = (G.krhs - 1) * nij, o_kstp = (G.kstp - 1) * nij, o_ptc = (G.kstp - 1) * nij;
    if (rp) {
      if (ina) {
        const double zkv = F.zeta[x0_ + o_in], hv = F.h[x0_];
        DP[s0_] = zkv + hv; ZP[s0_] = zkv;
        UP[s0_] = F.ubar[x0_ + o_in]; VP[s0_] = F.vbar[x0_ + o_in]; sH[s0_] = hv;
        sPm[s0_] = F.pm[x0_]; sPn[s0_] = F.pn[x0_];
        const double zsv = F.zeta[x0_ + o_kstp];
        DQ[s0_] = zsv + hv; ZQ[s0_] = zsv;
        UQ[s0_] = F.ubar[x0_ + o_kstp]; VQ[s0_] = F.vbar[x0_ + o_kstp];
        sRhoA[s0_] = F.rhoA[x0_]; sRhoS[s0_] = F.rhoS[x0_];
        sOnu[s0_] = F.on_u[x0_]; sOmv[s0_] = F.om_v[x0_];
        if (MK) { sRm[s0_] = G.rmask[x0_]; sUm[s0_] = G.umask[x0_]; sVm[s0_] = G.vmask[x0_]; }
        if (ownR) {
          aZt[s0_] = F.Zt_avg1[x0_]; aDU1[s0_] = F.DU_avg1[x0_]; aDU2[s0_] = F.DU_avg2[x0_];
          aDV1[s0_] = F.DV_avg1[x0_]; aDV2[s0_] = F.DV_avg2[x0_];
        }
        if (inBz) RZP[s0_] = F.rzeta[x0_ + o_ptc];
      } else {
        DP[s0_] = 0.0; UP[s0_] = 0.0; VP[s0_] = 0.0; ZP[s0_] = 0.0; sH[s0_] = 0.0; sPm[s0_] = 0.0; sPn[s0_] = 0.0; sRhoA[s0_] = 0.0;
        DQ[s0_] = 0.0; UQ[s0_] = 0.0; VQ[s0_] = 0.0; ZQ[s0_] = 0.0; sRhoS[s0_] = 0.0; sOnu[s0_] = 0.0; sOmv[s0_] = 0.0;
        if (MK) { sRm[s0_] = 0.0; sUm[s0_] = 0.0; sVm[s0_] = 0.0; }
      }
    }
    if (mE) {
      const int x1 = isvt ? GIDX(mi_, mj_ - 1) : GIDX(mi_ - 1, mj_);
      const int q1 = isvt ? GIDX(mi_ + 1, mj_) : GIDX(mi_, mj_ + 1);
      wm = isvt ? s2_metrics<1>(mr, mp, mx_, x1, q1) : s2_metrics<0>(mr, mp, mx_, x1, q1);
      if (MK) { pk0 = G.pmask[mx_]; pk1 = G.pmask[q1]; }
      w_frc = (isvt ? F.rvfrc : F.rufrc)[mx_];
      if (mO) w_rp = (isvt ? F.rvbar : F.rubar)[mx_ + o_ptc];
    }
  }
  // the blocks whose own points this block's rectangle touches (their results are its rim): lane q of wave 0 takes
  // the candidate at offset (q % 7 - 3, q / 7 - 3) -- sub-tiles are at least two points wide and high (g_step2d.cpp)
  int nbf = -1;
  if (t < 49) {
    const int dx = t % 7 - 3, dy = t / 7 - 3;
    int nx = bx + dx, ny = by + dy, sx = 0, sy = 0;
    bool ok = true;
    if (wrapx) { while (nx < 0) { nx += G.nbx2; sx -= G.Lm; } while (nx >= G.nbx2) { nx -= G.nbx2; sx += G.Lm; } }
    else if (nx < 0 || nx >= G.nbx2) ok = false;       // (multi-tile: beyond the tile the points arrive tagged, without a word)
    if (wrapy) { while (ny < 0) { ny += G.nby2; sy -= G.Mm; } while (ny >= G.nby2) { ny -= G.nby2; sy += G.Mm; } }
    else if (ny < 0 || ny >= G.nby2) ok = false;
    if (ok && !(nx == bx && ny == by)) {
      int p0, p1, q0, q1;
      s2l_range(G, nx, ny, p0, p1, q0, q1);
      if (p0 + sx <= B.Iend + S2P_RIM && p1 + sx >= B.Istr - S2P_RIM && q0 + sy <= B.Jend + S2P_RIM && q1 + sy >= B.Jstr - S2P_RIM)
        nbf = (nx + G.nbx2 * ny) * S2L_FSTRIDE;
    }
  }
  // (multi-tile) does any own point of this block lie in a neighbour's ghost zone?
  const bool edgeblk = MT && ((B.Istr < G.T.Istr + B2D_GH && (a.P.nbmask & 1)) || (B.Iend > G.T.Iend - B2D_GL && (a.P.nbmask & 2)) ||
                              (B.Jstr < G.T.Jstr + B2D_GH && (a.P.nbmask & 4)) || (B.Jend > G.T.Jend - B2D_GL && (a.P.nbmask & 8)));
  bool dead = false;
  KSYNC();

  const int np = a.npairs;
  for (int p = 0; p < np; p++) {
    const int tail = np - 1 - p;                      // pairs still to follow (Step2dPairArgs::tail)
    // (the indices the global stores are addressed with are re-derived in every pair: hoisted out of the loop, the
    // addresses and edge predicates of ~40 store sites would be live across it and spill)
    S2L_OPQ(oq);
    const int i = i_ + oq, j = j_ + oq, x0 = x0_ + oq, mi = mi_ + oq, mj = mj_ + oq, mx = mx_ + oq;
    // (likewise the products of the metric factors -- or0*or0 ... -- are formed where they are used)
#define OPD(x_) asm volatile("" : "+v"(x_))
    OPD(wm.onom); OPD(wm.fomn0); OPD(wm.fomn1); OPD(wm.dndx0); OPD(wm.dndx1); OPD(wm.dmde0); OPD(wm.dmde1); OPD(wm.v2r0); OPD(wm.v2r1);
    OPD(wm.pmr0); OPD(wm.pmr1); OPD(wm.pnr0); OPD(wm.pnr1); OPD(wm.or0); OPD(wm.or1); OPD(wm.v2p0); OPD(wm.v2p1); OPD(wm.pmp0); OPD(wm.pmp1);
    OPD(wm.pnp0); OPD(wm.pnp1); OPD(wm.op0); OPD(wm.op1); OPD(w_frc);
#undef OPD
    const bool img0 = tail == 0 && IMG, img1 = tail <= 1 && IMG, lst1 = tail <= 1, store3 = tail == 0;
    const int krhs = (p & 1) ? 3 - G.krhs : G.krhs;   // logical level of the predictor's krhs = the corrector's kstp
    const int lev_out = (p & 1) ? 5 : 4;
    double *zn3 = F.zeta + 2 * G.nij, *un3 = F.ubar + 2 * G.nij, *vn3 = F.vbar + 2 * G.nij;
    double *zout = F.zeta + (size_t)(lev_out - 1) * G.nij, *uout = F.ubar + (size_t)(lev_out - 1) * G.nij,
           *vout = F.vbar + (size_t)(lev_out - 1) * G.nij;
    double *rz_k = F.rzeta + (size_t)(krhs - 1) * G.nij, *rub_k = F.rubar + (size_t)(krhs - 1) * G.nij,
           *rvb_k = F.rvbar + (size_t)(krhs - 1) * G.nij;
    const int iif = G.iif + p;
    const bool f1 = iif == 1;                         // the first fast step: its own coefficients and branches (k_step2d.h: mode 0, `first`)
    const double w1_m1 = a.wts[3 * iif], w2_0 = a.wts[3 * iif + 1], w2_p1 = a.wts[3 * iif + 2];
    S2L_TICK(0);
    // the previous pair's result is this pair's krhs level: what the last two pair launches commit to the logical
    // levels (k_step2d_pair.h stage 1, `commit`)
    if (p >= 1 && lst1 && own) {
      const int s0 = s0_;
      double *zlog = F.zeta + (size_t)(krhs - 1) * G.nij, *ulog = F.ubar + (size_t)(krhs - 1) * G.nij,
             *vlog = F.vbar + (size_t)(krhs - 1) * G.nij;
      hb_emit2(G, B, zlog, BC_R, i, j, ZP[s0], Mr, img0);
      if (i >= B.IstrU) hb_emit2(G, B, ulog, BC_U, i, j, UP[s0], Mu, img0);
      if (j >= B.JstrV) hb_emit2(G, B, vlog, BC_V, i, j, VP[s0], Mv, img0);
    }
    // the converted forcing of the first fast step and its history level: the blocks around this one read rufrc, rvfrc and
    // ru, rv(:,:,0,nstp) of ITS points for the rim of their first predictor -- behind the first exchange they are through
    if (a.first && p == 1 && mO) {
      const size_t o_r0s = (size_t)(a.nstp - 1) * G.nij * (size_t)(G.N + 1);
      (isvt ? F.rvfrc : F.rufrc)[mx] = w_frc;
      (isvt ? F.rv : F.ru)[o_r0s + (size_t)mx] = w_frc;
    }
    // ================================ PREDICTOR on the enlarged sub-tile ==========================
    // ---- stages 2+3: mass fluxes :600-700, fast-time averaging :739-880 (own points), free-surface step :886-1000
    if (rp) {
      S2L_OPQ(lo);
      const int s0 = s0_ + lo;
      const double cA1 = w1_m1, cA2 = f1 ? (-1.0 / 12.0) * w2_p1 : (8.0 / 12.0) * w2_0 - (1.0 / 12.0) * w2_p1;
      const double fac = a.kfac;
      const double cff1z = f1 ? dtfast : 2.0 * dtfast, cff4 = 4.0 / 25.0, cff5 = 1.0 - 2.0 * cff4;
      double du = 0.0, dv = 0.0;
      if (INR(i, j, E.IstrUm2 - 1, E.Iendp2, E.JstrVm2 - 1, E.Jendp2)) {
        if (i >= E.IstrUm2) {
          const double cff = 0.5 * sOnu[s0];
          const double cff1 = cff * (DP[s0] + DP[(s0 - 1)]);
          du = UP[s0] * cff1;
          DUon[s0] = du;
        }
        if (j >= E.JstrVm2) {
          const double cff = 0.5 * sOmv[s0];
          const double cff1 = cff * (DP[s0] + DP[(s0 - TW)]);
          dv = VP[s0] * cff1;
          DVom[s0] = dv;
        }
      }
      if (ownR) {
        const bool pz = i >= B.IstrR && j >= B.JstrR, pu = i >= B.Istr && j >= B.JstrR, pv = i >= B.IstrR && j >= B.Jstr;
        if (f1) {                                       // :739-760: the averages start here
          if (pz) aZt[s0] = 0.0;
          if (pu) { aDU1[s0] = 0.0; aDU2[s0] = cA2 * du; }
          if (pv) { aDV1[s0] = 0.0; aDV2[s0] = cA2 * dv; }
        } else {
          if (pz) aZt[s0] = aZt[s0] + cA1 * ZP[s0];
          if (pu) {
            aDU1[s0] = aDU1[s0] + cA1 * du;
            aDU2[s0] = aDU2[s0] + cA2 * du;
          }
          if (pv) {
            aDV1[s0] = aDV1[s0] + cA1 * dv;
            aDV2[s0] = aDV2[s0] + cA2 * dv;
          }
        }
      }
      if (inEz) {
        double du1, dv1;     // DUon(i+1,j), DVom(i,j+1)
        {
          const double cff = 0.5 * sOnu[(s0 + 1)];
          const double cff1 = cff * (DP[(s0 + 1)] + DP[s0]);
          du1 = UP[(s0 + 1)] * cff1;
        }
        {
          const double cff = 0.5 * sOmv[(s0 + TW)];
          const double cff1 = cff * (DP[(s0 + TW)] + DP[s0]);
          dv1 = VP[(s0 + TW)] * cff1;
        }
        const double rhs_zeta = (du - du1) + (dv - dv1);
        const double zsv = ZQ[s0], zkv = ZP[s0];
        double zeta_new = zsv + sPm[s0] * sPn[s0] * cff1z * rhs_zeta;
        if (MK) zeta_new = zeta_new * sRm[s0];
        const double zw = f1 ? 0.5 * (zsv + zeta_new) : cff5 * zkv + cff4 * (zsv + zeta_new);
        const double rhoSv = sRhoS[s0];
        D1[s0] = zeta_new + sH[s0];
        Z1[s0] = zeta_new;
        RZ1[s0] = rhs_zeta;
        zwrk[s0] = zw;
        const double gz = (fac + rhoSv) * zw;
        gzeta[s0] = gz;
        gzeta2[s0] = gz * zw;
        gzetaSA[s0] = zw * (rhoSv - sRhoA[s0]);
        if (own) {
          if (store3) hb_emit2(G, B, zn3, BC_R, i, j, zeta_new, Mr, IMG);
          if (lst1) hb_emit2(G, B, rz_k, BC_NONE, i, j, rhs_zeta, nullptr, img1);
        }
      }
    }
    KSYNC();
    S2L_TICK(1);
    // ---- stage 4: momentum on the enlarged sub-tile :1080-2670 ----------------------------------
    {
      const S2Tiles Tl = {UP, VP, DUon, DVom, DP, sPm, sPn, sH, sRhoA, gzeta, gzeta2, gzetaSA, zwrk, TW};
      const S2Edge Eg = {!G.ewp, !G.ewp, !G.nsp, !G.nsp, 1, G.Lm, 1, G.Mm};
      const double c1 = f1 ? 0.5 * dtfast : dtfast;
#pragma unroll
      for (int isv = 0; isv < 2; isv++) {
        if (isvt == isv && mE) {
          S2L_OPQ(lo);
          const int s = ms_ + lo, x = mx;
          const int d1 = isv ? TW : 1;
          const double rhs = isv ? s2_rhs<1>(Tl, wm, Eg, s, mi, mj, g, ADV, COR, CURV, VIS, MK, pk0, pk1)
                                 : s2_rhs<0>(Tl, wm, Eg, s, mi, mj, g, ADV, COR, CURV, VIS, MK, pk0, pk1);
          double r = rhs;
          if (f1) {
            // coupling with the 3-D forcing :2225-2460: rufrc becomes the fast-time-constant forcing, its history kept in
            // ru, rv(:,:,0,nstp | nnew) (k_step2d.h `first`); every point of the enlarged sub-tile forms it, the own points store it
            double *r3 = isv ? F.rv : F.ru;
            const size_t o_r0s = (size_t)(a.nstp - 1) * G.nij * (size_t)(G.N + 1), o_r0n = (size_t)(a.nnew - 1) * G.nij * (size_t)(G.N + 1);
            const double fr = w_frc - r;
            if (a.startup == 0) r = r + fr;
            else if (a.startup == 1) r = r + 1.5 * fr - 0.5 * r3[o_r0n + (size_t)x];
            else r = r + (23.0 / 12.0) * fr - (16.0 / 12.0) * r3[o_r0n + (size_t)x] + (5.0 / 12.0) * r3[o_r0s + (size_t)x];
            w_frc = fr;                                   // (stored behind the first exchange: the neighbours read the old values on their rims)
          } else {
            r = r + w_frc;
          }
          const double cff = (sPm[s] + sPm[s - d1]) * (sPn[s] + sPn[s - d1]);
          const double fac = 1.0 / (D1[s] + D1[s - d1]);
          const double Dstp0 = DQ[s], Dstp1 = DQ[s - d1];
          const double sv = (isv ? VQ : UQ)[s];
          double b = (sv * (Dstp0 + Dstp1) + cff * c1 * r) * fac;
          if (MK) b = b * (isv ? sVm : sUm)[s];
          (isv ? V1 : U1)[s] = b;
          w_rP = r;
          if (mO) {
            if (!isv) {
              if (store3) hb_emit2(G, B, un3, BC_U, mi, mj, b, Mu, IMG);
              if (lst1) rub_k[x] = r;
            } else {
              if (store3) hb_emit2(G, B, vn3, BC_V, mi, mj, b, Mv, IMG);
              if (lst1) rvb_k[x] = r;
            }
          }
        }
      }
    }
    KSYNC();
    S2L_TICK(2);
    // ---- closed domain edges: zetabc / u2dbc / v2dbc of the predictor's result, on the LDS tiles (k_step2d_pair.h) --
    {
      const bool cw = !G.ewp && E.west, ce = !G.ewp && E.east, cs = !G.nsp && E.south, cn = !G.nsp && E.north;
      if (cw || ce || cs || cn) {
        S2L_OPQ(lo);
        const int IT0o = IT0 + lo;
#define LA(A_, i_, j_) A_[((i_) - IT0o) + ((j_) - JT0) * TW]
        const int Lm = G.Lm, Mm = G.Mm;
        const double gamma2 = G.gamma2;
        const int zj0 = E.JstrV - 1, zj1 = E.Jend, zi0 = E.IstrU - 1, zi1 = E.Iend;
#define MK_(T_, i_, j_) (MK ? LA(T_, i_, j_) : 1.0)
        if (cw) KLOOP1(jb, zj0, zj1) { const double v = LA(Z1, 1, jb) * MK_(sRm, 0, jb); LA(Z1, 0, jb) = v; LA(D1, 0, jb) = v + LA(sH, 0, jb); }
        if (ce) KLOOP1(jb, zj0, zj1) { const double v = LA(Z1, Lm, jb) * MK_(sRm, Lm + 1, jb); LA(Z1, Lm + 1, jb) = v; LA(D1, Lm + 1, jb) = v + LA(sH, Lm + 1, jb); }
        if (cs) KLOOP1(ib, zi0, zi1) { const double v = LA(Z1, ib, 1) * MK_(sRm, ib, 0); LA(Z1, ib, 0) = v; LA(D1, ib, 0) = v + LA(sH, ib, 0); }
        if (cn) KLOOP1(ib, zi0, zi1) { const double v = LA(Z1, ib, Mm) * MK_(sRm, ib, Mm + 1); LA(Z1, ib, Mm + 1) = v; LA(D1, ib, Mm + 1) = v + LA(sH, ib, Mm + 1); }
        if (cw) KLOOP1(jb, j0E, j1E) LA(U1, 1, jb) = 0.0;
        if (ce) KLOOP1(jb, j0E, j1E) LA(U1, Lm + 1, jb) = 0.0;
        if (cs) KLOOP1(ib, i0E, i1E) LA(V1, ib, 1) = 0.0;
        if (cn) KLOOP1(ib, i0E, i1E) LA(V1, ib, Mm + 1) = 0.0;
        KSYNC();
        {
          const int ui0 = cw ? 1 : E.IstrU, ui1 = ce ? Lm + 1 : i1E;
          if (cs) KLOOP1(ib, ui0, ui1) LA(U1, ib, 0) = gamma2 * LA(U1, ib, 1) * MK_(sUm, ib, 0);
          if (cn) KLOOP1(ib, ui0, ui1) LA(U1, ib, Mm + 1) = gamma2 * LA(U1, ib, Mm) * MK_(sUm, ib, Mm + 1);
          const int vj0 = cs ? 1 : E.JstrV, vj1 = cn ? Mm + 1 : j1E;
          if (cw) KLOOP1(jb, vj0, vj1) LA(V1, 0, jb) = gamma2 * LA(V1, 1, jb) * MK_(sVm, 0, jb);
          if (ce) KLOOP1(jb, vj0, vj1) LA(V1, Lm + 1, jb) = gamma2 * LA(V1, Lm, jb) * MK_(sVm, Lm + 1, jb);
        }
        KSYNC();
        // corners of a closed basin (zetabc.F:753-782, u2dbc_im.F:1159-1188, v2dbc_im.F:1208-1237; as k_step2d_pair.h)
        if (!(G.ewp || G.nsp)) {
          if (t == 0) {
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
        }
#undef MK_
#undef LA
      }
    }
    S2L_TICK(3);
    // ================================ CORRECTOR on the own sub-tile ===============================
    // krhs = 3: D1 = zeta(3)+h, U1, V1; kstp = the predictor's krhs: DP, UP, VP; ptsk = the predictor's kstp
    if (rp) {
      S2L_OPQ(lo);
      const int s0 = s0_ + lo;
      const double cA2 = f1 ? w2_0 : (5.0 / 12.0) * w2_0;
      const double fac = a.kfac;
      const double cff1 = a.kz1, cff2 = a.kz2, cff3 = a.kz3, cff4 = 2.0 / 5.0, cff5 = 1.0 - cff4;
      double du = 0.0, dv = 0.0;
      if (INR(i, j, B.IstrUm2 - 1, B.Iendp2, B.JstrVm2 - 1, B.Jendp2)) {
        if (i >= B.IstrUm2) {
          const double cff = 0.5 * sOnu[s0];
          const double cff1f = cff * (D1[s0] + D1[(s0 - 1)]);
          du = U1[s0] * cff1f;
          DUon[s0] = du;
        }
        if (j >= B.JstrVm2) {
          const double cff = 0.5 * sOmv[s0];
          const double cff1f = cff * (D1[s0] + D1[(s0 - TW)]);
          dv = V1[s0] * cff1f;
          DVom[s0] = dv;
        }
      }
      if (ownR) {
        const bool pu = i >= B.Istr && j >= B.JstrR, pv = i >= B.IstrR && j >= B.Jstr;
        if (pu) aDU2[s0] = aDU2[s0] + cA2 * du;
        if (pv) aDV2[s0] = aDV2[s0] + cA2 * dv;
      }
      if (inBz) {
        double du1, dv1;
        {
          const double cff = 0.5 * sOnu[(s0 + 1)];
          const double cff1f = cff * (D1[(s0 + 1)] + D1[s0]);
          du1 = U1[(s0 + 1)] * cff1f;
        }
        {
          const double cff = 0.5 * sOmv[(s0 + TW)];
          const double cff1f = cff * (D1[(s0 + TW)] + D1[s0]);
          dv1 = V1[(s0 + TW)] * cff1f;
        }
        const double rhs_zeta = (du - du1) + (dv - dv1);
        const double zsv = ZP[s0], zkv = Z1[s0];
        const double cff = cff1 * rhs_zeta;
        // (the corrector of the first fast step is a forward-Euler step too: k_step2d.h mode 0)
        double zeta_new = f1 ? zsv + sPm[s0] * sPn[s0] * dtfast * rhs_zeta
                             : zsv + sPm[s0] * sPn[s0] * (cff + cff2 * RZ1[s0] - cff3 * RZP[s0]);
        if (MK) zeta_new = zeta_new * sRm[s0];
        const double zw = f1 ? 0.5 * (zsv + zeta_new) : cff5 * zeta_new + cff4 * zkv;
        const double rhoSv = sRhoS[s0];
        DQ[s0] = zeta_new + sH[s0];
        zwrk[s0] = zw;
        const double gz = (fac + rhoSv) * zw;
        gzeta[s0] = gz;
        gzeta2[s0] = gz * zw;
        gzetaSA[s0] = zw * (rhoSv - sRhoA[s0]);
        if (own) {
          hb_emit2<true>(G, B, zout, BC_R, i, j, zeta_new, Mr, img0);
          if (MT && edgeblk && a.P.early) s2l_remit(G, a.P, B, 3 * (p & 3), BC_R, i, j, zeta_new, a.epoch + (unsigned)(p + 1), Mr);
          ZQ[s0] = zeta_new;
        }
      }
    }
    KSYNC();
    S2L_TICK(4);
    // ---- stage 4: momentum on the own sub-tile ----------------------------------------------------
    {
      const S2Tiles Tl = {U1, V1, DUon, DVom, D1, sPm, sPn, sH, sRhoA, gzeta, gzeta2, gzetaSA, zwrk, TW};
      const S2Edge Eg = {!G.ewp, !G.ewp, !G.nsp, !G.nsp, 1, G.Lm, 1, G.Mm};
      const double k1 = a.km1, k2 = a.km2, k3 = a.km3;
#pragma unroll
      for (int isv = 0; isv < 2; isv++) {
        if (isvt == isv && mO) {
          S2L_OPQ(lo);
          const int s = ms_ + lo;
          const int d1 = isv ? TW : 1;
          const double rhs = isv ? s2_rhs<1>(Tl, wm, Eg, s, mi, mj, g, ADV, COR, CURV, VIS, MK, pk0, pk1)
                                 : s2_rhs<0>(Tl, wm, Eg, s, mi, mj, g, ADV, COR, CURV, VIS, MK, pk0, pk1);
          const double r = rhs + w_frc;
          const double cff = (sPm[s] + sPm[s - d1]) * (sPn[s] + sPn[s - d1]);
          const double fac = 1.0 / (DQ[s] + DQ[s - d1]);
          const double Dstp0 = DP[s], Dstp1 = DP[s - d1];
          const double sv = (isv ? VP : UP)[s];
          const double rs = w_rP;
          const double rp_ = w_rp;
          double b = f1 ? (sv * (Dstp0 + Dstp1) + cff * (0.5 * dtfast) * r) * fac
                        : (sv * (Dstp0 + Dstp1) + cff * (k1 * r + k2 * rs - k3 * rp_)) * fac;
          if (MK) b = b * (isv ? sVm : sUm)[s];
          if (!isv) { hb_emit2<true>(G, B, uout, BC_U, mi, mj, b, Mu, img0); UQ[s] = b; }
          else { hb_emit2<true>(G, B, vout, BC_V, mi, mj, b, Mv, img0); VQ[s] = b; }
          if (MT && edgeblk && a.P.early) s2l_remit(G, a.P, B, 3 * (p & 3) + 1 + isv, isv ? BC_V : BC_U, mi, mj, b, a.epoch + (unsigned)(p + 1), isv ? Mv : Mu);
        }
      }
    }
    S2L_TICK(5);
    if (tail == 0 && !a.aux) break;
    // ================================ the rim of the next pair ======================================
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every wave: its write-through stores have left
    KSYNC();
    if (t == 0) __hip_atomic_store(a.flags + me * S2L_FSTRIDE, a.epoch + (unsigned)(p + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (MT && edgeblk && !a.P.early) {
      // the neighbouring ranks' ghost points, tagged with the pair (the values are in the Q tiles: the own points' threads stored them)
      const unsigned tag = a.epoch + (unsigned)(p + 1);
      if (own) s2l_remit(G, a.P, B, 3 * (p & 3), BC_R, i, j, ZQ[s0_], tag, Mr);
      if (mO) s2l_remit(G, a.P, B, 3 * (p & 3) + 1 + isvt, isvt ? BC_V : BC_U, mi, mj, (isvt ? VQ : UQ)[ms_], tag, isvt ? Mv : Mu);
    }
    if (t < 64) {
      if (nbf >= 0 && !dead) {
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(a.flags + nbf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < a.epoch + (unsigned)(p + 1)) {
          __builtin_amdgcn_s_sleep(1);
          if (wall_clock64() - t0 > a.timeout) {
            dead = true;
            *(volatile unsigned long long *)a.err = ((unsigned long long)(p + 1) << 32) | (unsigned long long)(me + 1);
            break;
          }
        }
      }
    }
    KSYNC();
    S2L_TICK(6);
    if (rp) {
      const int s0 = s0_;
      // own points keep what the block computed (stored to ZQ, DQ, UQ, VQ by their threads); every
      // other point of the rectangle -- the rim, and the boundary values behind a closed edge -- is some block's result
      const bool oz = own, ou = own && i >= B.IstrU, ov = own && j >= B.JstrV;
      if (!(MT && rem) && ina) {
        if (!oz) { const double z = s2l_ld(zout + x0); DQ[s0] = z + sH[s0]; ZQ[s0] = z; }
        if (!ou) UQ[s0] = s2l_ld(uout + x0);
        if (!ov) VQ[s0] = s2l_ld(vout + x0);
      }
    }
    // (multi-tile) the ghost points of the tile, behind the loads from the tile's own blocks: what the neighbouring ranks stored
    // has been on its way while this block waited for the arrival words and loaded its local rim
    if (MT && remv && !dead) {
      // my ghost point: until all six words carry this pair's number (two 16-byte loads per field would do as well: each
      // 8-byte word is checked by itself)
      const unsigned tag = a.epoch + (unsigned)(p + 1);
      const unsigned long long *rq = a.P.rim + 2 * ((size_t)(3 * (p & 3)) * (size_t)G.nij + (size_t)x0);
      const long long t0 = wall_clock64();
      for (;;) {
        unsigned long long w[6];
#pragma unroll
        for (int f = 0; f < 3; f++) {
          w[2 * f] = __hip_atomic_load(rq + 2 * (size_t)f * (size_t)G.nij, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          w[2 * f + 1] = __hip_atomic_load(rq + 2 * (size_t)f * (size_t)G.nij + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        // (no u-points west of a closed western edge's wall, no v-points south of a closed southern one: nobody writes them)
        const bool nu = G.ewp || iw >= 1, nv = G.nsp || jw >= 1;
        const bool all = (unsigned)(w[0] >> 32) == tag && (unsigned)(w[1] >> 32) == tag &&
                         (!nu || ((unsigned)(w[2] >> 32) == tag && (unsigned)(w[3] >> 32) == tag)) &&
                         (!nv || ((unsigned)(w[4] >> 32) == tag && (unsigned)(w[5] >> 32) == tag));
        if (all) {
          const double z = __longlong_as_double((long long)((w[0] & 0xffffffffull) | (w[1] << 32)));
          DQ[s0_] = z + sH[s0_]; ZQ[s0_] = z;
          UQ[s0_] = __longlong_as_double((long long)((w[2] & 0xffffffffull) | (w[3] << 32)));
          VQ[s0_] = __longlong_as_double((long long)((w[4] & 0xffffffffull) | (w[5] << 32)));
          break;
        }
        __builtin_amdgcn_s_sleep(1);
        if (wall_clock64() - t0 > a.timeout) {
          dead = true;
          *(volatile unsigned long long *)a.err = ((unsigned long long)(p + 1) << 32) | (unsigned long long)(me + 1);
          break;
        }
      }
    }
    w_rp = w_rP;
    { double *q; q = DP; DP = DQ; DQ = q; q = UP; UP = UQ; UQ = q; q = VP; VP = VQ; VQ = q; q = ZP; ZP = ZQ; ZQ = q; q = RZ1; RZ1 = RZP; RZP = q; }
    KSYNC();
    S2L_TICK(7);
    if (tail == 0) {
      // ================================ the auxiliary predictor call iif = nfast+1 (:821-883) =========================
      // its krhs level is the last corrector's result (P, rim included): commit it to the logical level (k_step2d_ac),
      // form the mass fluxes once more and close the fast-time averages, with their periodic images (:821-883 exchanges them)
      const int iifx = G.iif + np;
      const int kx = (np & 1) ? 3 - G.krhs : G.krhs;
      const double cA1 = a.wts[3 * iifx], cA2 = (8.0 / 12.0) * a.wts[3 * iifx + 1] - (1.0 / 12.0) * a.wts[3 * iifx + 2];
      if (rp) {
        const int s0 = s0_;
        if (own) {
          double *zlog = F.zeta + (size_t)(kx - 1) * G.nij, *ulog = F.ubar + (size_t)(kx - 1) * G.nij, *vlog = F.vbar + (size_t)(kx - 1) * G.nij;
          hb_emit2(G, B, zlog, BC_R, i, j, ZP[s0], Mr, IMG);
          if (i >= B.IstrU) hb_emit2(G, B, ulog, BC_U, i, j, UP[s0], Mu, IMG);
          if (j >= B.JstrV) hb_emit2(G, B, vlog, BC_V, i, j, VP[s0], Mv, IMG);
        }
        if (ownR && ina) {
          const bool pz = i >= B.IstrR && j >= B.JstrR, pu = i >= B.Istr && j >= B.JstrR, pv = i >= B.IstrR && j >= B.Jstr;
          double du = 0.0, dv = 0.0;
          if (i >= B.IstrUm2) {
            const double cff = 0.5 * sOnu[s0];
            const double cff1 = cff * (DP[s0] + DP[(s0 - 1)]);
            du = UP[s0] * cff1;
          }
          if (j >= B.JstrVm2) {
            const double cff = 0.5 * sOmv[s0];
            const double cff1 = cff * (DP[s0] + DP[(s0 - TW)]);
            dv = VP[s0] * cff1;
          }
          if (pz) hb_emit2(G, B, F.Zt_avg1, BC_NONE, i, j, aZt[s0] + cA1 * ZP[s0], nullptr, IMG);
          if (pu) {
            hb_emit2(G, B, F.DU_avg1, BC_NONE, i, j, aDU1[s0] + cA1 * du, nullptr, IMG);
            F.DU_avg2[x0] = aDU2[s0] + cA2 * du;
          }
          if (pv) {
            hb_emit2(G, B, F.DV_avg1, BC_NONE, i, j, aDV1[s0] + cA1 * dv, nullptr, IMG);
            F.DV_avg2[x0] = aDV2[s0] + cA2 * dv;
          }
        }
      }
      return;
    }
  }
  // ---- (without the auxiliary call) what the loop leaves behind: the fast-time averages (k_step2d_pair.h stores them in every launch)
  if (ownR && ina) {
    const bool pz = i_ >= B.IstrR && j_ >= B.JstrR, pu = i_ >= B.Istr && j_ >= B.JstrR, pv = i_ >= B.IstrR && j_ >= B.Jstr;
    if (pz) F.Zt_avg1[x0_] = aZt[s0_];
    if (pu) { F.DU_avg1[x0_] = aDU1[s0_]; F.DU_avg2[x0_] = aDU2[s0_]; }
    if (pv) { F.DV_avg1[x0_] = aDV1[s0_]; F.DV_avg2[x0_] = aDV2[s0_]; }
  }
#undef WRAPI
#undef WRAPJ
#undef GIDX
#undef INR
}

// sub-tiles up to 32x4 on 640 threads (the pair kernel's shape: 42x14 rectangle, 2 x 288 momentum points), and up to 16x8
// on 512 threads (26x18 rectangle = 468 points, 2 x 240 momentum points: 17-20 % less work per pair, eight neighbours
// instead of fourteen, two waves per SIMD -- a register budget of 256)
static __global__ void __launch_bounds__(640) k_step2d_loop_a(const Step2dLoopArgs a) {
  extern __shared__ double lds_dyn_[];
  k_step2d_loop_body<32, 4, 640>(a, (int)blockIdx.x, (int)blockIdx.y, lds_dyn_);
}
static __global__ void __launch_bounds__(512) k_step2d_loop_b(const Step2dLoopArgs a) {
  extern __shared__ double lds_dyn_[];
  k_step2d_loop_body<16, 8, 512>(a, (int)blockIdx.x, (int)blockIdx.y, lds_dyn_);
}
// ... with land/sea masks (MASKING, round 6)
static __global__ void __launch_bounds__(512) k_step2d_loop_bk(const Step2dLoopArgs a) {
  extern __shared__ double lds_dyn_[];
  k_step2d_loop_body<16, 8, 512, false, true>(a, (int)blockIdx.x, (int)blockIdx.y, lds_dyn_);
}
static __global__ void __launch_bounds__(512) k_step2d_loop_bmk(const Step2dLoopArgs a) {
  extern __shared__ double lds_dyn_[];
  k_step2d_loop_body<16, 8, 512, true, true>(a, (int)blockIdx.x, (int)blockIdx.y, lds_dyn_);
}
// the same in a multi-tile context: edge blocks hand their rim to the neighbouring ranks and take theirs (S2LPeer)
static __global__ void __launch_bounds__(512) k_step2d_loop_bm(const Step2dLoopArgs a) {
  extern __shared__ double lds_dyn_[];
  k_step2d_loop_body<16, 8, 512, true>(a, (int)blockIdx.x, (int)blockIdx.y, lds_dyn_);
}
#endif
