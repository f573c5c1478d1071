// k_step2d.h -- barotropic LF-AM3 predictor/corrector sub-step as ONE kernel.
//
// Replaces step2d_tile, ROMS/Nonlinear/step2d_LF_AM3.h:163-3056 (options SOLVE3D, VAR_RHO_2D,
// UV_ADV 4th-order centred :1246-1395, UV_COR, CURVGRID, UV_VIS2).
//
// One thread block = one ROMS sub-tile.  The kernel is a chain of dependent stencil stages on a
// small 2-D grid, so its cost is latency (global-load round trips and barriers), not bandwidth.
// It is therefore organised as FOUR stages with three barriers:
//
//   1  prologue   every global read of the kernel is issued here.  Fields needed at neighbouring
//                 points (zeta+h, ubar, vbar, h, pm, pn, rhoA, zeta(kstp)+h) go to LDS tiles over
//                 the sub-tile rectangle (Istr-3:Iend+3, Jstr-3:Jend+3); values needed only at
//                 a thread's own points stay in registers (fixed point -> thread mapping).
//   2  fluxes     DUon, DVom on the rectangle (LDS) + fast-time averaging at the own points
//   3  zeta       new free surface and the pressure-gradient work arrays on the rectangle (LDS)
//   4  momentum   each momentum point (u-point or v-point of an interior cell = one work item)
//                 evaluates its complete right-hand side -- pressure gradient, 4th-order
//                 advection, Coriolis, curvilinear terms, harmonic viscosity -- directly from the
//                 LDS tiles, recomputing the handful of neighbouring flux values it needs instead
//                 of exchanging them through LDS and barriers as the reference's loop nests
//                 (private arrays UFx, UFe, VFx, VFe, grad, Dgrad) would; then the coupling with
//                 the 3-D forcing and the time step.  Every flux value is computed with exactly
//                 the reference's expression, so results are bit-identical to the loop-nest form.
//
// In single-tile runs with a periodic direction every thread also stores the boundary and periodic
// images of its values (k_haloblock.h), so no halo launch follows.
//
// The kernel is instruction-issue bound (one block of 6-8 waves per CU, long dependent chains), so
// the body is written to keep integer/address work down: the sub-tile shape and thread count are
// template constants (LDS offsets become immediates, point <-> thread maps constant divisions), the
// u- and v-points are separate passes with the direction a compile-time constant, the 13 metric
// arrays a momentum point needs are read from two packed 64-byte records (k_pack_m2d), and time
// levels of one array are addressed by index offsets instead of separate base pointers (SGPRs).
#pragma once
#include "roms_ctx.h"
#include "k_haloblock.h"


// The arrays of the barotropic step, by value in the kernel arguments: the kernel is latency bound and
// a pointer taken from the device-resident table costs one more dependent memory round trip before
// the first data load can be issued.
struct S2Fields {
  GPtr zeta, ubar, vbar, h, on_u, om_v, pm, pn, rhoA, rhoS, Zt_avg1, DU_avg1, DU_avg2, DV_avg1, DV_avg2,
      rzeta, rubar, rvbar, rufrc, rvfrc, ru, rv, m2r, m2p, xr, duv;
};
#define S2F_FILL(dst, src)                                                                               \
  do {                                                                                                   \
    (dst).zeta = (src).zeta; (dst).ubar = (src).ubar; (dst).vbar = (src).vbar; (dst).h = (src).h;        \
    (dst).on_u = (src).on_u; (dst).om_v = (src).om_v; (dst).pm = (src).pm; (dst).pn = (src).pn;          \
    (dst).rhoA = (src).rhoA; (dst).rhoS = (src).rhoS; (dst).Zt_avg1 = (src).Zt_avg1;                     \
    (dst).DU_avg1 = (src).DU_avg1; (dst).DU_avg2 = (src).DU_avg2; (dst).DV_avg1 = (src).DV_avg1;         \
    (dst).DV_avg2 = (src).DV_avg2; (dst).rzeta = (src).rzeta; (dst).rubar = (src).rubar;                 \
    (dst).rvbar = (src).rvbar; (dst).rufrc = (src).rufrc; (dst).rvfrc = (src).rvfrc; (dst).ru = (src).ru; \
    (dst).rv = (src).rv; (dst).m2r = (src).m2r; (dst).m2p = (src).m2p; (dst).xr = (src).xr;              \
    (dst).duv = (src).duv;                                                                               \
  } while (0)
struct Step2dArgs {
  S2Fields F;          // (first: its offsets in the argument block do not move when DGrid grows)
  DGrid G;
  double w1_m1;        // weight(1,iif-1)
  double w1_0;         // weight(1,iif): the fast-time average of the momentum diagnostics (DIAGNOSTICS_UV, :2707)
  double w2_0, w2_p1;  // weight(2,iif), weight(2,iif+1)
  int lev_in;          // physical level of zeta/ubar/vbar(krhs): G.krhs, or the staging level the pair kernel left its result in
  int commit;          // (the auxiliary last call behind a pair) copy that level to the logical level G.krhs: k_step2d_pair.h
  // LnudgeM2CLM (round 6; the NDG instantiation only, behind everything the other kernels read): nudging of the 2-D momentum
  // towards its climatology, step2d_LF_AM3.h:2179-2203
  const double *om_u, *on_v, *ubarclm, *vbarclm, *M2nudgcof;
};

#define STEP2D_NLDS 15

// Packed time-invariant metrics of the momentum stage, one 64-byte record per grid point (a wave reads
// 4 KB of consecutive records).  Built by k_pack_m2d from the individual arrays, which stay the
// interface (roms_hip_upload) and the source of truth.
#ifdef ROMS_CPU_EMU
struct M2Rec { double v[8]; };
#else
struct alignas(64) M2Rec { double v[8]; };   // hipMalloc'ed base, 64-byte records: dwordx4 loads
#endif
enum { MR_FOMN = 0, MR_DNDX, MR_DMDE, MR_V2, MR_PMON, MR_PNOM, MR_ON, MR_OM };          // rho points (Fields::m2r)
enum { MP_V2 = 0, MP_PMON, MP_PNOM, MP_OM, MP_ON, MP_ONU, MP_OMV, MP_SPARE };           // psi points (Fields::m2p)

struct PackArgs {
  DGrid G;
  Fields Fv;         // the array pointers, by value (a table in device memory would cost every kernel one more dependent round trip)
};
THREAD_KERNEL(k_pack_m2d, PackArgs) {
  (void)gz;
  const DGrid &G = a.G;
  const Fields &F = a.Fv;
  const size_t x = (size_t)gx + (size_t)gy * (size_t)G.ni;
  double *r = F.m2r + 8 * x, *p = F.m2p + 8 * x;
  r[MR_FOMN] = F.fomn[x]; r[MR_DNDX] = F.dndx[x]; r[MR_DMDE] = F.dmde[x]; r[MR_V2] = G.uv_vis4 ? F.visc4_r[x] : F.visc2_r[x];   // (UV_VIS4: the biharmonic coefficient in its place)
  r[MR_PMON] = F.pmon_r[x]; r[MR_PNOM] = F.pnom_r[x]; r[MR_ON] = F.on_r[x]; r[MR_OM] = F.om_r[x];
  p[MP_V2] = G.uv_vis4 ? F.visc4_p[x] : F.visc2_p[x]; p[MP_PMON] = F.pmon_p[x]; p[MP_PNOM] = F.pnom_p[x]; p[MP_OM] = F.om_p[x];
  p[MP_ON] = F.on_p[x]; p[MP_ONU] = F.on_u[x]; p[MP_OMV] = F.om_v[x]; p[MP_SPARE] = 0.0;
}
THREAD_GLOBAL(k_pack_m2d, PackArgs)

#define INR(i, j, i0, i1, j0, j1) ((i) >= (i0) && (i) <= (i1) && (j) >= (j0) && (j) <= (j1))

// Template constants: BWC x BHC = largest sub-tile (the actual one may be smaller: edge sub-tiles),
// NTC threads, PTS rectangle points per thread.  BWC = 0 is the generic form (any sub-tile shape and
// thread count, run-time loops, no register-resident own-point values); it is also the form the
// serial CPU emulation (tests/emu) runs.
//
// Sweep over the rectangle (IT0:IT0+TW-1, JT0:JT0+TH-1) of a full-size sub-tile: fixed
// point -> thread mapping q = KTID + m*NT, LDS index s0 = q.
#define RLOOP(i, j)                                                                                         \
  _Pragma("unroll") for (int m = 0, q_ = KTID; FIXED ? m < PTS : q_ < NTILE; m++, q_ += NT)                \
    if (q_ < NTILE)                                                                                         \
      for (int s0 = q_, jj_ = q_ / TW, j = JT0 + jj_, i = IT0 + q_ - jj_ * TW,                              \
               x0 = (i - G.LBi) + (j - G.LBj) * ni, once_ = 1; once_; once_ = 0)
// (PWR: rectangle points per thread whose own-point values stay in registers; the others re-read them where a stage needs
// them -- the 32x8 form has 532 rectangle points on 512 threads: a second register set for 20 threads made it spill)
#define PWDECL(name) double name[PWR > 0 ? PWR : 1]
#define PWLOAD(name, expr) do { if (FIXED && m < PWR) name[m] = (expr); } while (0)
#define PW(name, expr) ((FIXED && m < PWR) ? name[m < PWR ? m : 0] : (expr))
// Momentum points: c = interior cell (row-major over BW x BH), isv = 0 its u-point, 1 its v-point.
// The u-points use threads 0..NOWN-1, the v-points threads VOFF..VOFF+NOWN-1 (VOFF = NOWN when the
// block has the threads, else 0: then every thread does a u- and a v-point in turn).
#define WLOOP(isv)                                                                                          \
  for (int m_ = 0, c = KTID - (isv) * VOFF; FIXED ? m_ < 1 : c < NOWN; m_++, c += NT)                       \
    if (c >= 0 && c < NOWN)
#define WDECL(name) double name[WSLOTS]
#define WLOAD(name, isv, expr) do { if (FIXED) name[VOFF ? 0 : (isv)] = (expr); } while (0)
#define WV(name, isv, expr) (FIXED ? name[VOFF ? 0 : (isv)] : (expr))

#ifndef ROMS_CPU_EMU
#define S2D_TICK(n) do { if (a.G.dbg_stop == 99 && KTID == 0) F.xr[(bx + G.nbx2 * by) * 8 + (n)] = (double)wall_clock64(); } while (0)
#else
#define S2D_TICK(n) ((void)0)
#endif
// VIS4 (generic form only): the UV_VIS4 block of step2d_LF_AM3.h:1653-1920 -- two more LDS tiles hold the first harmonic
// operator LapU, LapV of ubar, vbar(krhs) on the sub-tile + 1 with its closed / gradient conditions and corner values; the
// momentum stage then forms the same stress tensor of (LapU, LapV) times the total depth.  The packed metric records carry
// visc4 in the place of visc2 (g_step2d.cpp:pack_metrics).
template <int BWC, int BHC, int NTC, int PTS, bool MK = (BWC == 0), bool CM = (BWC == 0), int PWR = PTS, bool DUV = false, bool VIS4 = false, bool WD = false, bool NDG = false>
COOP_KERNEL(k_step2d_t, Step2dArgs) {
  (void)bz;
#ifndef ROMS_CPU_EMU
  __builtin_amdgcn_s_setprio(3);   // beside other kernels (main3d_late) these waves are the step's critical path
#endif
  constexpr bool FIXED = BWC > 0;
  // MASKING runs take the generic form or k_step2d_am (g_step2d.cpp): the other fixed-shape instantiations compile the
  // mask products away (with them k_step2d_c spilled 36 VGPRs instead of 14 and an unmasked 512x512x50 step took 7 % longer)
  const bool MSK = MK && a.G.masking;                 // MK: this instantiation carries the mask products
  const DGrid &G = a.G;
  const S2Fields &F = a.F;
#ifndef ROMS_CPU_EMU
  xcd_remap2(G, bx, by);
#endif
  const TB B = block_bounds2(G, bx, by);
  const int OW = FIXED ? BWC : G.bw2, OH = FIXED ? BHC : G.bh2, NOWN = OW * OH;
  const int TW = OW + 6, TH = OH + 6, NTILE = TW * TH, NT = FIXED ? NTC : KNT;
  constexpr int VOFF = (BWC > 0 && 2 * BWC * BHC <= NTC) ? (BWC * BHC + 63) / 64 * 64 : 0;
  constexpr int WSLOTS = (BWC > 0 && VOFF == 0) ? 2 : 1;
  const size_t sz = (size_t)NTILE;
  double *Drhs = lds, *DUon = lds + sz, *DVom = lds + 2 * sz, *Dnew = lds + 3 * sz;
  double *zwrk = lds + 4 * sz, *gzeta = lds + 5 * sz, *gzeta2 = lds + 6 * sz, *gzetaSA = lds + 7 * sz;
  double *sUk = lds + 8 * sz, *sVk = lds + 9 * sz, *sH = lds + 10 * sz, *sPm = lds + 11 * sz, *sPn = lds + 12 * sz,
         *sRhoA = lds + 13 * sz, *sDstp = lds + 14 * sz;
  double *sLU = VIS4 ? lds + 15 * sz : nullptr, *sLV = VIS4 ? lds + 16 * sz : nullptr;
  const int krhs = G.krhs, kstp = G.kstp, knew = G.knew, nstp = G.nstp, nnew = G.nnew, iif = G.iif, iic = G.iic;
  const bool PRED = G.predictor != 0;
  const int Istr = B.Istr, Iend = B.Iend, Jstr = B.Jstr, Jend = B.Jend, IstrU = B.IstrU, JstrV = B.JstrV;
  const int IstrR = B.IstrR, IendR = B.IendR, JstrR = B.JstrR, JendR = B.JendR;
  const int ptsk = 3 - kstp;
  const double dtfast = G.dtfast, g = G.g;
  const int ni = G.ni, nij = (int)G.nij;
  // time levels of the 2-D state: index offsets into one array
  const int o_krhs = (a.lev_in - 1) * nij, o_kstp = (kstp - 1) * nij, o_ptsk = (ptsk - 1) * nij;
  double *zn = F.zeta + (size_t)(knew - 1) * G.nij, *un = F.ubar + (size_t)(knew - 1) * G.nij,
         *vn = F.vbar + (size_t)(knew - 1) * G.nij;
  double *rz_k = F.rzeta + (size_t)(krhs - 1) * G.nij;
  double *rub_k = F.rubar + (size_t)(krhs - 1) * G.nij, *rvb_k = F.rvbar + (size_t)(krhs - 1) * G.nij;
  const size_t o_r0s = (size_t)(nstp - 1) * G.nij * (size_t)(G.N + 1);   // ru(:,:,0,nstp)
  const size_t o_r0n = (size_t)(nnew - 1) * G.nij * (size_t)(G.N + 1);
  const M2Rec *mr = (const M2Rec *)(double *)F.m2r, *mp = (const M2Rec *)(double *)F.m2p;
  const bool last = iif > G.nfast;                 // auxiliary last predictor call :883
  const bool fuse = G.fuse_halo != 0, fuse_last = fuse && last && PRED;
  const int first = (iif == 1 && PRED);
  const int corr = (!PRED && iif != 1);
  const int startup = (iic == G.ntfirst) ? 0 : ((iic == G.ntfirst + 1) ? 1 : 2);
  const bool wfix = !G.ewp && B.west, efix = !G.ewp && B.east, sfix = !G.nsp && B.south, nfix = !G.nsp && B.north;
  const int IT0 = Istr - 3, JT0 = Jstr - 3;
  const int UBi = G.LBi + G.ni - 1, UBj = G.LBj + G.nj - 1;

  // stage 2-3 own-point values
  PWDECL(r_zk); PWDECL(r_zs); PWDECL(r_on_u); PWDECL(r_om_v); PWDECL(r_rhoS);
  PWDECL(r_Zt); PWDECL(r_DU1); PWDECL(r_DU2); PWDECL(r_DV1); PWDECL(r_DV2);
  PWDECL(r_rz_s); PWDECL(r_rz_p);
  // stage 4 values of a momentum point: metrics at its two rho points (P0 = own cell, P1 = the cell on
  // the other side of the momentum point) and its two psi points (Q0 = (i,j), Q1 = next along the face)
  WDECL(w_onom); WDECL(w_fomn0); WDECL(w_fomn1); WDECL(w_dndx0); WDECL(w_dndx1); WDECL(w_dmde0); WDECL(w_dmde1);
  WDECL(w_v2r0); WDECL(w_v2r1); WDECL(w_pmr0); WDECL(w_pmr1); WDECL(w_pnr0); WDECL(w_pnr1);
  WDECL(w_or0); WDECL(w_or1);     // on_r (u-points) | om_r (v-points) at P0, P1
  WDECL(w_v2p0); WDECL(w_v2p1); WDECL(w_pmp0); WDECL(w_pmp1); WDECL(w_pnp0); WDECL(w_pnp1);
  WDECL(w_op0); WDECL(w_op1);     // om_p (u-points) | on_p (v-points) at Q0, Q1
  WDECL(w_s); WDECL(w_frc); WDECL(w_rs); WDECL(w_rp); WDECL(w_r0n); WDECL(w_r0s);

  S2D_TICK(0);
  // ---- stage 1: every global read of the kernel ----------------------------------------------
  RLOOP(i, j) {
    if (INR(i, j, G.LBi, UBi, G.LBj, UBj)) {
      // each array is read only as far from the sub-tile as some stage below needs it
      const bool ring2 = INR(i, j, Istr - 2, Iend + 2, Jstr - 2, Jend + 2);
      const bool ring1 = INR(i, j, Istr - 1, Iend + 1, Jstr - 1, Jend + 1);
      const double zkv = F.zeta[x0 + o_krhs], hv = F.h[x0];
      Drhs[s0] = zkv + hv;                                    // total depth :600
      const double ukv = F.ubar[x0 + o_krhs], vkv = F.vbar[x0 + o_krhs];
      sUk[s0] = ukv; sVk[s0] = vkv; sH[s0] = hv;
      if (CM && a.commit) {      // (CM: only the instantiations a pair launch can precede carry this code)
        // the last pair's result, staged by k_step2d_pair: now the logical level krhs (own points; the tile's boundary and
        // ghost points by its edge blocks)
        double *zl = F.zeta + (size_t)(krhs - 1) * G.nij, *ul = F.ubar + (size_t)(krhs - 1) * G.nij, *vl = F.vbar + (size_t)(krhs - 1) * G.nij;
        const bool own = INR(i, j, Istr, Iend, Jstr, Jend);
        if (fuse) {
          if (own) {
            hb_emit(G, B, zl, BC_R, i, j, zkv, MSK ? G.rmask : nullptr);
            if (i >= IstrU) hb_emit(G, B, ul, BC_U, i, j, ukv, MSK ? G.umask : nullptr);
            if (j >= JstrV) hb_emit(G, B, vl, BC_V, i, j, vkv, MSK ? G.vmask : nullptr);
          }
        } else if (own || !INR(i, j, G.T.Istr, G.T.Iend, G.T.Jstr, G.T.Jend)) {
          zl[x0] = zkv; ul[x0] = ukv; vl[x0] = vkv;
        }
      }
      PWLOAD(r_zk, zkv);
      PWLOAD(r_on_u, F.on_u[x0]); PWLOAD(r_om_v, F.om_v[x0]);
      if (ring2 || VIS4) { sPm[s0] = F.pm[x0]; sPn[s0] = F.pn[x0]; }      // (UV_VIS4: the first operator on the sub-tile + 1 reads pm, pn three points out)
      if (ring1) {
        const double zsv = F.zeta[x0 + o_kstp];
        sDstp[s0] = zsv + hv;
        PWLOAD(r_zs, zsv);
        sRhoA[s0] = F.rhoA[x0];
        PWLOAD(r_Zt, F.Zt_avg1[x0]); PWLOAD(r_DU1, F.DU_avg1[x0]); PWLOAD(r_DU2, F.DU_avg2[x0]);
        PWLOAD(r_DV1, F.DV_avg1[x0]); PWLOAD(r_DV2, F.DV_avg2[x0]);
        if (!last) {
          PWLOAD(r_rhoS, F.rhoS[x0]);
          if (corr) { PWLOAD(r_rz_s, F.rzeta[x0 + o_kstp]); PWLOAD(r_rz_p, F.rzeta[x0 + o_ptsk]); }
        }
      }
    } else {
      Drhs[s0] = 0.0; sDstp[s0] = 0.0; sUk[s0] = 0.0; sVk[s0] = 0.0; sH[s0] = 0.0; sPm[s0] = 0.0; sPn[s0] = 0.0; sRhoA[s0] = 0.0;
    }
  }
  if (FIXED && !last) {
#pragma unroll
    for (int isv = 0; isv < 2; isv++) {
      WLOOP(isv) {
        const int jj = c / OW, j = Jstr + jj, i = Istr + c - jj * OW;
        if (i > Iend || j > Jend || !(isv ? (j >= JstrV) : (i >= IstrU))) continue;
        const int x = (i - G.LBi) + (j - G.LBj) * ni;
        const int x1 = isv ? x - ni : x - 1;        // P1: (i,j-1) for a v-point, (i-1,j) for a u-point
        const int q1 = isv ? x + 1 : x + ni;        // Q1: (i+1,j) for a v-point, (i,j+1) for a u-point
        const M2Rec R0 = mr[x], R1 = mr[x1], P0 = mp[x], P1 = mp[q1];
        WLOAD(w_onom, isv, isv ? P0.v[MP_OMV] : P0.v[MP_ONU]);
        WLOAD(w_fomn0, isv, R0.v[MR_FOMN]); WLOAD(w_fomn1, isv, R1.v[MR_FOMN]);
        WLOAD(w_dndx0, isv, R0.v[MR_DNDX]); WLOAD(w_dndx1, isv, R1.v[MR_DNDX]);
        WLOAD(w_dmde0, isv, R0.v[MR_DMDE]); WLOAD(w_dmde1, isv, R1.v[MR_DMDE]);
        WLOAD(w_v2r0, isv, R0.v[MR_V2]); WLOAD(w_v2r1, isv, R1.v[MR_V2]);
        WLOAD(w_pmr0, isv, R0.v[MR_PMON]); WLOAD(w_pmr1, isv, R1.v[MR_PMON]);
        WLOAD(w_pnr0, isv, R0.v[MR_PNOM]); WLOAD(w_pnr1, isv, R1.v[MR_PNOM]);
        WLOAD(w_or0, isv, isv ? R0.v[MR_OM] : R0.v[MR_ON]); WLOAD(w_or1, isv, isv ? R1.v[MR_OM] : R1.v[MR_ON]);
        WLOAD(w_v2p0, isv, P0.v[MP_V2]); WLOAD(w_v2p1, isv, P1.v[MP_V2]);
        WLOAD(w_pmp0, isv, P0.v[MP_PMON]); WLOAD(w_pmp1, isv, P1.v[MP_PMON]);
        WLOAD(w_pnp0, isv, P0.v[MP_PNOM]); WLOAD(w_pnp1, isv, P1.v[MP_PNOM]);
        WLOAD(w_op0, isv, isv ? P0.v[MP_ON] : P0.v[MP_OM]); WLOAD(w_op1, isv, isv ? P1.v[MP_ON] : P1.v[MP_OM]);
        WLOAD(w_s, isv, (isv ? F.vbar : F.ubar)[x + o_kstp]);
        WLOAD(w_frc, isv, (isv ? F.rvfrc : F.rufrc)[x]);
        if (corr) {
          WLOAD(w_rs, isv, (isv ? F.rvbar : F.rubar)[x + o_kstp]);
          WLOAD(w_rp, isv, (isv ? F.rvbar : F.rubar)[x + o_ptsk]);
        }
        if (first && startup >= 1) WLOAD(w_r0n, isv, (isv ? F.rv : F.ru)[o_r0n + (size_t)x]);
        if (first && startup >= 2) WLOAD(w_r0s, isv, (isv ? F.rv : F.ru)[o_r0s + (size_t)x]);
      }
    }
  }
  S2D_TICK(1);
  KSYNC();
  S2D_TICK(2);

  // ---- stage 2: mass fluxes :600-700 and fast-time averaging :739-880 -------------------------
  {
    double cA1 = 0.0, cA2;   // weights of DU_avg1 / DU_avg2 for this call
    int amode;               // 0: first predictor (reset), 1: predictor, 2: corrector
    if (PRED) {
      if (iif == 1) { amode = 0; cA2 = (-1.0 / 12.0) * a.w2_p1; }
      else { amode = 1; cA1 = a.w1_m1; cA2 = (8.0 / 12.0) * a.w2_0 - (1.0 / 12.0) * a.w2_p1; }
    } else {
      amode = 2;
      cA2 = (iif == 1) ? a.w2_0 : (5.0 / 12.0) * a.w2_0;
    }
    RLOOP(i, j) {
      double du = 0.0, dv = 0.0;
      if (INR(i, j, B.IstrUm2 - 1, B.Iendp2, B.JstrVm2 - 1, B.Jendp2)) {
        if (i >= B.IstrUm2) {
          const double cff = 0.5 * PW(r_on_u, F.on_u[x0]);
          const double cff1 = cff * (Drhs[s0] + Drhs[(s0 - 1)]);
          du = sUk[s0] * cff1;
          if (BWC == 0 && G.volcons) {     // VolCons (the generic form only): set_DUV_bc_tile, obc_volcons.F:310-335, over its own line range
            const bool w = (G.volcons & (1 << ROMS_IWEST)) && B.west && i == B.Istr, e = (G.volcons & (1 << ROMS_IEAST)) && B.east && i == B.Iend + 1;
            if ((w || e) && j >= KMAX(2, B.JstrV - 1) - 2 && j <= KMIN(B.Jend + 1, G.Mm) + 1) {
              const double xs = G.vcons[2];
              du = 0.5 * (Drhs[s0] + Drhs[(s0 - 1)]) * (w ? sUk[s0] - xs : sUk[s0] + xs) * PW(r_on_u, F.on_u[x0]);
              if (MSK) du = du * G.umask[x0];
            }
          }
          DUon[s0] = du;
        }
        if (j >= B.JstrVm2) {
          const double cff = 0.5 * PW(r_om_v, F.om_v[x0]);
          const double cff1 = cff * (Drhs[s0] + Drhs[(s0 - TW)]);
          dv = sVk[s0] * cff1;
          if (BWC == 0 && G.volcons) {     // obc_volcons.F:337-362
            const bool sb = (G.volcons & (1 << ROMS_ISOUTH)) && B.south && j == B.Jstr, nb = (G.volcons & (1 << ROMS_INORTH)) && B.north && j == B.Jend + 1;
            if ((sb || nb) && i >= KMAX(2, B.IstrU - 1) - 2 && i <= KMIN(B.Iend + 1, G.Lm) + 1) {
              const double xs = G.vcons[2];
              dv = 0.5 * (Drhs[s0] + Drhs[(s0 - TW)]) * (sb ? sVk[s0] - xs : sVk[s0] + xs) * PW(r_om_v, F.om_v[x0]);
              if (MSK) dv = dv * G.vmask[x0];
            }
          }
          DVom[s0] = dv;
        }
      }
      if (INR(i, j, KMIN(IstrR, Istr), IendR, KMIN(JstrR, Jstr), JendR)) {
        const bool pz = i >= IstrR && j >= JstrR, pu = i >= Istr && j >= JstrR, pv = i >= IstrR && j >= Jstr;
        if (amode == 0) {
          if (pz) F.Zt_avg1[x0] = 0.0;
          if (pu) { F.DU_avg1[x0] = 0.0; F.DU_avg2[x0] = cA2 * du; }
          if (pv) { F.DV_avg1[x0] = 0.0; F.DV_avg2[x0] = cA2 * dv; }
        } else if (amode == 1) {
          if (pz) {
            const double v = PW(r_Zt, F.Zt_avg1[x0]) + cA1 * PW(r_zk, F.zeta[x0 + o_krhs]);
            if (fuse_last) hb_emit(G, B, F.Zt_avg1, BC_NONE, i, j, v);   // final averages: exchange :821-883
            else F.Zt_avg1[x0] = v;
          }
          if (pu) {
            const double v = PW(r_DU1, F.DU_avg1[x0]) + cA1 * du;
            if (fuse_last) hb_emit(G, B, F.DU_avg1, BC_NONE, i, j, v);
            else F.DU_avg1[x0] = v;
            F.DU_avg2[x0] = PW(r_DU2, F.DU_avg2[x0]) + cA2 * du;
          }
          if (pv) {
            const double v = PW(r_DV1, F.DV_avg1[x0]) + cA1 * dv;
            if (fuse_last) hb_emit(G, B, F.DV_avg1, BC_NONE, i, j, v);
            else F.DV_avg1[x0] = v;
            F.DV_avg2[x0] = PW(r_DV2, F.DV_avg2[x0]) + cA2 * dv;
          }
        } else {
          if (pu) F.DU_avg2[x0] = PW(r_DU2, F.DU_avg2[x0]) + cA2 * du;
          if (pv) F.DV_avg2[x0] = PW(r_DV2, F.DV_avg2[x0]) + cA2 * dv;
        }
      }
    }
  }
  if (last) return;            // auxiliary last predictor call :883 (uniform over the grid)
  KSYNC();
  S2D_TICK(3);

  // ---- stage 3: free-surface step :886-1000 ---------------------------------------------------
  {
    const double fac = 1000.0 / G.rho0;
    double cff1, cff2 = 0.0, cff3 = 0.0, cff4, cff5;
    int mode;
    if (iif == 1) { mode = 0; cff1 = dtfast; cff4 = 0.0; cff5 = 0.0; }
    else if (PRED) { mode = 1; cff1 = 2.0 * dtfast; cff4 = 4.0 / 25.0; cff5 = 1.0 - 2.0 * cff4; }
    else { mode = 2; cff1 = dtfast * 5.0 / 12.0; cff2 = dtfast * 8.0 / 12.0; cff3 = dtfast * 1.0 / 12.0; cff4 = 2.0 / 5.0; cff5 = 1.0 - cff4; }
    RLOOP(i, j) {
      if (INR(i, j, IstrU - 1, Iend, JstrV - 1, Jend)) {
        const double rhs_zeta = (DUon[s0] - DUon[(s0 + 1)]) + (DVom[s0] - DVom[(s0 + TW)]);
        const double zsv = PW(r_zs, F.zeta[x0 + o_kstp]), zkv = PW(r_zk, F.zeta[x0 + o_krhs]);
        double zeta_new, zw;
        if (mode == 0 || mode == 1) {
          zeta_new = zsv + sPm[s0] * sPn[s0] * cff1 * rhs_zeta;
        } else {
          const double cff = cff1 * rhs_zeta;
          zeta_new = zsv + sPm[s0] * sPn[s0] * (cff + cff2 * PW(r_rz_s, F.rzeta[x0 + o_kstp]) - cff3 * PW(r_rz_p, F.rzeta[x0 + o_ptsk]));
        }
        if (MSK) zeta_new = zeta_new * G.rmask[x0];                      // :907,933,964
        if (mode == 0) zw = 0.5 * (zsv + zeta_new);
        else if (mode == 1) zw = cff5 * zkv + cff4 * (zsv + zeta_new);
        else zw = cff5 * zeta_new + cff4 * zkv;
        const double rhoSv = PW(r_rhoS, F.rhoS[x0]);
        Dnew[s0] = zeta_new + sH[s0];
        zwrk[s0] = zw;
        const double gz = (fac + rhoSv) * zw;
        gzeta[s0] = gz;
        gzeta2[s0] = gz * zw;
        gzetaSA[s0] = zw * (rhoSv - sRhoA[s0]);
        if (i >= Istr && j >= Jstr) {
          // zetabc :1057 + exchange :1068, rzeta exchange :1030
          if (fuse) {
            hb_emit(G, B, zn, BC_R, i, j, zeta_new, MSK ? G.rmask : nullptr);
            if (PRED) hb_emit(G, B, rz_k, BC_NONE, i, j, rhs_zeta);
          } else {
            // WET_DRY: "depth > Dcrit for masked cells" :992-995 (the shared level only; Dnew keeps zeta_new)
            zn[x0] = (WD && MSK) ? zeta_new + (G.Dcrit - sH[s0]) * (1.0 - G.rmask[x0]) : zeta_new;
            if (PRED) rz_k[x0] = rhs_zeta;
          }
        }
      }
    }
  }
  KSYNC();
  S2D_TICK(4);

  if (VIS4) {
    // ---- stage 3b (UV_VIS4): first harmonic operator of ubar, vbar(krhs) :1667-1722, no thickness in it
    const int clu = (int)((G.lbc_closed >> (4 * ROMS_ISUBAR)) & 15ull), clv = (int)((G.lbc_closed >> (4 * ROMS_ISVBAR)) & 15ull);
    const double gamma2 = G.gamma2;
#define SU_(o) sUk[s0 + (o)]
#define SV_(o) sVk[s0 + (o)]
#define SPM_(o) sPm[s0 + (o)]
#define SPN_(o) sPn[s0 + (o)]
    // stress at the rho point at tile offset o (record r), at the psi point at tile offset o (record p)
#define S4R(o, r) ((r).v[MR_V2] * 0.5 *                                                                              \
                   ((r).v[MR_PMON] * ((SPN_(o) + SPN_((o) + 1)) * SU_((o) + 1) - (SPN_((o) - 1) + SPN_(o)) * SU_(o)) -      \
                    (r).v[MR_PNOM] * ((SPM_(o) + SPM_((o) + TW)) * SV_((o) + TW) - (SPM_((o) - TW) + SPM_(o)) * SV_(o))))
#define S4P(o, p) ((p).v[MP_V2] * 0.5 *                                                                              \
                   ((p).v[MP_PMON] * ((SPN_((o) - TW) + SPN_(o)) * SV_(o) - (SPN_((o) - 1 - TW) + SPN_((o) - 1)) * SV_((o) - 1)) +   \
                    (p).v[MP_PNOM] * ((SPM_((o) - 1) + SPM_(o)) * SU_(o) - (SPM_((o) - 1 - TW) + SPM_((o) - TW)) * SU_((o) - TW))))
    RLOOP(i, j) {
      const bool inU = INR(i, j, IstrU - 1, Iend + 1, Jstr - 1, Jend + 1), inV = INR(i, j, Istr - 1, Iend + 1, JstrV - 1, Jend + 1);
      if (inU || inV) {
        const M2Rec r0 = mr[x0], p0 = mp[x0];
        const double cR0 = S4R(0, r0);
        double cP0 = S4P(0, p0);
        if (MSK) cP0 = cP0 * G.pmask[x0];
        if (inU) {
          const M2Rec rw = mr[x0 - 1], pn_ = mp[x0 + ni];
          const double cRw = S4R(-1, rw);
          double cPn = S4P(TW, pn_);
          if (MSK) cPn = cPn * G.pmask[x0 + ni];
          const double UFx1 = r0.v[MR_ON] * r0.v[MR_ON] * cR0, UFx0 = rw.v[MR_ON] * rw.v[MR_ON] * cRw;
          const double UFe1 = pn_.v[MP_OM] * pn_.v[MP_OM] * cPn, UFe0 = p0.v[MP_OM] * p0.v[MP_OM] * cP0;
          sLU[s0] = 0.125 * (SPM_(-1) + SPM_(0)) * (SPN_(-1) + SPN_(0)) *
                    ((SPN_(-1) + SPN_(0)) * (UFx1 - UFx0) + (SPM_(-1) + SPM_(0)) * (UFe1 - UFe0));
        }
        if (inV) {
          const M2Rec rs = mr[x0 - ni], pe = mp[x0 + 1];
          const double cRs = S4R(-TW, rs);
          double cPe = S4P(1, pe);
          if (MSK) cPe = cPe * G.pmask[x0 + 1];
          const double VFx1 = pe.v[MP_ON] * pe.v[MP_ON] * cPe, VFx0 = p0.v[MP_ON] * p0.v[MP_ON] * cP0;
          const double VFe1 = r0.v[MR_OM] * r0.v[MR_OM] * cR0, VFe0 = rs.v[MR_OM] * rs.v[MR_OM] * cRs;
          sLV[s0] = 0.125 * (SPM_(0) + SPM_(-TW)) * (SPN_(0) + SPN_(-TW)) *
                    ((SPN_(-TW) + SPN_(0)) * (VFx1 - VFx0) - (SPM_(-TW) + SPM_(0)) * (VFe1 - VFe0));
        }
      }
    }
#undef S4R
#undef S4P
#undef SU_
#undef SV_
#undef SPM_
#undef SPN_
    KSYNC();
    // conditions of the first operator at the edges of the domain :1728-1810 (sources: values no condition touches) ...
    if (wfix || efix || sfix || nfix) {
      RLOOP(i, j) {
        const bool cu = ((wfix && i == Istr) || (efix && i == Iend + 1)) && ((sfix && j == Jstr - 1) || (nfix && j == Jend + 1));
        const bool cv = ((wfix && i == Istr - 1) || (efix && i == Iend + 1)) && ((sfix && j == Jstr) || (nfix && j == Jend + 1));
        if (!cu && INR(i, j, IstrU - 1, Iend + 1, Jstr - 1, Jend + 1)) {
          if (wfix && i == Istr) sLU[s0] = (clu & (1 << ROMS_IWEST)) ? 0.0 : sLU[s0 + 1];
          else if (efix && i == Iend + 1) sLU[s0] = (clu & (1 << ROMS_IEAST)) ? 0.0 : sLU[s0 - 1];
          else if (sfix && j == Jstr - 1) sLU[s0] = (clu & (1 << ROMS_ISOUTH)) ? gamma2 * sLU[s0 + TW] : 0.0;
          else if (nfix && j == Jend + 1) sLU[s0] = (clu & (1 << ROMS_INORTH)) ? gamma2 * sLU[s0 - TW] : 0.0;
        }
        if (!cv && INR(i, j, Istr - 1, Iend + 1, JstrV - 1, Jend + 1)) {
          if (wfix && i == Istr - 1) sLV[s0] = (clv & (1 << ROMS_IWEST)) ? gamma2 * sLV[s0 + 1] : 0.0;
          else if (efix && i == Iend + 1) sLV[s0] = (clv & (1 << ROMS_IEAST)) ? gamma2 * sLV[s0 - 1] : 0.0;
          else if (sfix && j == Jstr) sLV[s0] = (clv & (1 << ROMS_ISOUTH)) ? 0.0 : sLV[s0 + TW];
          else if (nfix && j == Jend + 1) sLV[s0] = (clv & (1 << ROMS_INORTH)) ? 0.0 : sLV[s0 - TW];
        }
      }
      KSYNC();
      // ... and the corner values :1812-1853
      if (!(G.ewp || G.nsp)) {
        RLOOP(i, j) {
          const int di = (wfix && i == Istr) ? 1 : ((efix && i == Iend + 1) ? -1 : 0), dj = (sfix && j == Jstr - 1) ? 1 : ((nfix && j == Jend + 1) ? -1 : 0);
          if (di && dj) sLU[s0] = 0.5 * (sLU[s0 + di] + sLU[s0 + dj * TW]);
          const int ei = (wfix && i == Istr - 1) ? 1 : ((efix && i == Iend + 1) ? -1 : 0), ej = (sfix && j == Jstr) ? 1 : ((nfix && j == Jend + 1) ? -1 : 0);
          if (ei && ej) sLV[s0] = 0.5 * (sLV[s0 + ej * TW] + sLV[s0 + ei]);
        }
        KSYNC();
      }
    }
  }

  // ---- stage 4: right-hand sides and the momentum step, one work item per momentum point ------
  // Tile accessors relative to s = S2(i,j); the flux helpers below are the reference's expressions
  // for one entry of its private arrays (UFx, UFe, VFx, VFe, ...), edge replication included.
#define TU(di, dj) sUk[s + (di) + (dj) * TW]
#define TV(di, dj) sVk[s + (di) + (dj) * TW]
#define TDU(di, dj) DUon[s + (di) + (dj) * TW]
#define TDV(di, dj) DVom[s + (di) + (dj) * TW]
#define TD(di, dj) Drhs[s + (di) + (dj) * TW]
#define TPM(di, dj) sPm[s + (di) + (dj) * TW]
#define TPN(di, dj) sPn[s + (di) + (dj) * TW]
  {
    const double c6 = 1.0 / 6.0;
    const bool ADV = (G.options & ROMS_UV_ADV) != 0, COR = (G.options & ROMS_UV_COR) != 0;
    const bool CURV = ADV && (G.options & ROMS_CURVGRID) != 0, VIS = (G.options & ROMS_UV_VIS2) != 0;
    const double c1 = (iif == 1) ? 0.5 * dtfast : dtfast;
    const double k1 = 0.5 * dtfast * 5.0 / 12.0, k2 = 0.5 * dtfast * 8.0 / 12.0, k3 = 0.5 * dtfast * 1.0 / 12.0;
    const double pg1 = 0.5 * g, pg2 = 1.0 / 3.0;
#pragma unroll
    for (int isv = 0; isv < 2; isv++) {
    WLOOP(isv) {
      const int jj = c / OW, j = Jstr + jj, i = Istr + c - jj * OW;
      if (i > Iend || j > Jend || !(isv ? (j >= JstrV) : (i >= IstrU))) continue;
      const int s = (i - IT0) + (j - JT0) * TW;
      const int x = (i - G.LBi) + (j - G.LBj) * ni;
      const int x1 = isv ? x - ni : x - 1, q1 = isv ? x + 1 : x + ni;
      const int d1 = isv ? TW : 1;                   // tile offset from P1 to P0 (across the momentum point)
      // -- pressure gradient (VAR_RHO_2D) :1080-1200
      double rhs = pg1 * WV(w_onom, isv, isv ? mp[x].v[MP_OMV] : mp[x].v[MP_ONU]) *
                   ((sH[s - d1] + sH[s]) * (gzeta[s - d1] - gzeta[s]) +
                    (sH[s - d1] - sH[s]) * (gzetaSA[s - d1] + gzetaSA[s] +
                                           pg2 * (sRhoA[s - d1] - sRhoA[s]) * (zwrk[s - d1] - zwrk[s])) +
                    (gzeta2[s - d1] - gzeta2[s]));
      // DIAGNOSTICS_UV (DUV instantiation only): DiaU2rhs | DiaV2rhs(i,j,1:NDM2d-1) of this point, step2d_LF_AM3.h:1122-2472
      double dr[12];
      if (DUV) { for (int q_ = 0; q_ < 12; q_++) dr[q_] = 0.0; dr[G.m2[M2PGRD]] = rhs; }
      if (ADV) {
        // 4th-order centred advection :1246-1410.  G*(a) = second difference at offset a along the
        // flux direction, with the closed-edge replication of the reference.
        if (!isv) {
          // UFx(i) and UFx(i-1): grad/Dgrad along xi of ubar, DUon
#define GUX(a_) ({ int q_ = (a_); if (wfix && i + q_ == Istr) q_ += 1; if (efix && i + q_ == Iend + 1) q_ -= 1; \
                   TU(q_ - 1, 0) - 2.0 * TU(q_, 0) + TU(q_ + 1, 0); })
#define GDX(a_) ({ int q_ = (a_); if (wfix && i + q_ == Istr) q_ += 1; if (efix && i + q_ == Iend + 1) q_ -= 1; \
                   TDU(q_ - 1, 0) - 2.0 * TDU(q_, 0) + TDU(q_ + 1, 0); })
#define UFX(a_) (0.25 * (TU(a_, 0) + TU((a_) + 1, 0) - c6 * (GUX(a_) + GUX((a_) + 1))) *                      \
                 (TDU(a_, 0) + TDU((a_) + 1, 0) - c6 * (GDX(a_) + GDX((a_) + 1))))
          // UFe(j) and UFe(j+1): grad along eta of ubar (replicated at closed S/N edges), Dgrad along xi of DVom
#define GUE(b_) ({ int q_ = (b_); if (sfix && j + q_ == Jstr - 1) q_ += 1; if (nfix && j + q_ == Jend + 1) q_ -= 1; \
                   TU(0, q_ - 1) - 2.0 * TU(0, q_) + TU(0, q_ + 1); })
#define GDE(a_, b_) (TDV((a_) - 1, b_) - 2.0 * TDV(a_, b_) + TDV((a_) + 1, b_))
#define UFE(b_) (0.25 * (TU(0, b_) + TU(0, (b_) - 1) - c6 * (GUE(b_) + GUE((b_) - 1))) *                      \
                 (TDV(0, b_) + TDV(-1, b_) - c6 * (GDE(0, b_) + GDE(-1, b_))))
          const double cff1 = UFX(0) - UFX(-1);
          const double cff2 = UFE(1) - UFE(0);
          const double fac = cff1 + cff2;
          rhs = rhs - fac;
          if (DUV) { dr[G.m2[M2XADV]] = -cff1; dr[G.m2[M2YADV]] = -cff2; dr[G.m2[M2HADV]] = -fac; }       // :1405
#undef GUX
#undef GDX
#undef UFX
#undef GUE
#undef GDE
#undef UFE
        } else {
          // VFx(i) and VFx(i+1): grad along xi of vbar (replicated at closed W/E edges), Dgrad along eta of DUon
#define GVX(a_) ({ int q_ = (a_); if (wfix && i + q_ == Istr - 1) q_ += 1; if (efix && i + q_ == Iend + 1) q_ -= 1; \
                   TV(q_ - 1, 0) - 2.0 * TV(q_, 0) + TV(q_ + 1, 0); })
#define GDX(a_, b_) (TDU(a_, (b_) - 1) - 2.0 * TDU(a_, b_) + TDU(a_, (b_) + 1))
#define VFX(a_) (0.25 * (TV(a_, 0) + TV((a_) - 1, 0) - c6 * (GVX(a_) + GVX((a_) - 1))) *                      \
                 (TDU(a_, 0) + TDU(a_, -1) - c6 * (GDX(a_, 0) + GDX(a_, -1))))
          // VFe(j) and VFe(j-1): grad/Dgrad along eta of vbar, DVom
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
          if (DUV) { dr[G.m2[M2XADV]] = -cff1; dr[G.m2[M2YADV]] = -cff2; dr[G.m2[M2HADV]] = -fac; }       // :1418
#undef GVX
#undef GDX
#undef VFX
#undef GVE
#undef GDE
#undef VFE
        }
      }
      // values at the two rho points of this momentum point: P0 = (i,j), P1 = (i-1,j) | (i,j-1)
      const int a1 = isv ? 0 : -1, b1 = isv ? -1 : 0;
      if (COR) {
        // Coriolis :1429-1490: UFx = cff*(vbar(j)+vbar(j+1)), VFe = cff*(ubar(i)+ubar(i+1)) at rho points
        const double cf0 = 0.5 * TD(0, 0) * WV(w_fomn0, isv, mr[x].v[MR_FOMN]);
        const double cf1 = 0.5 * TD(a1, b1) * WV(w_fomn1, isv, mr[x1].v[MR_FOMN]);
        if (!isv) {
          const double fac1 = 0.5 * (cf0 * (TV(0, 0) + TV(0, 1)) + cf1 * (TV(-1, 0) + TV(-1, 1)));
          rhs = rhs + fac1;
          if (DUV) dr[G.m2[M2FCOR]] = fac1;                                                                // :1446
        } else {
          const double fac1 = 0.5 * (cf0 * (TU(0, 0) + TU(1, 0)) + cf1 * (TU(0, -1) + TU(1, -1)));
          rhs = rhs - fac1;
          if (DUV) dr[G.m2[M2FCOR]] = -fac1;                                                               // :1455
        }
      }
      if (CURV) {
        // curvilinear metric terms :1494-1560
        double t0, t1, w0 = 0.0, w1 = 0.0;               // w: Uwrk | Vwrk at P0, P1 (:1529-1531)
        {
          const double cff1 = 0.5 * (TV(0, 0) + TV(0, 1)), cff2 = 0.5 * (TU(0, 0) + TU(1, 0));
          const double cff3 = cff1 * WV(w_dndx0, isv, mr[x].v[MR_DNDX]), cff4 = cff2 * WV(w_dmde0, isv, mr[x].v[MR_DMDE]);
          const double cff = TD(0, 0) * (cff3 - cff4);
          t0 = isv ? cff * cff2 : cff * cff1;
          if (DUV) { const double c_ = TD(0, 0) * cff4; w0 = isv ? -c_ * cff2 : -c_ * cff1; }
        }
        {
          const double cff1 = 0.5 * (TV(a1, b1) + TV(a1, b1 + 1)), cff2 = 0.5 * (TU(a1, b1) + TU(a1 + 1, b1));
          const double cff3 = cff1 * WV(w_dndx1, isv, mr[x1].v[MR_DNDX]), cff4 = cff2 * WV(w_dmde1, isv, mr[x1].v[MR_DMDE]);
          const double cff = TD(a1, b1) * (cff3 - cff4);
          t1 = isv ? cff * cff2 : cff * cff1;
          if (DUV) { const double c_ = TD(a1, b1) * cff4; w1 = isv ? -c_ * cff2 : -c_ * cff1; }
        }
        const double fac1 = 0.5 * (t0 + t1);
        if (!isv) rhs = rhs + fac1;
        else rhs = rhs - fac1;
        if (DUV) {                                        // :1544-1559
          const double fac2 = 0.5 * (w0 + w1);
          if (!isv) { dr[G.m2[M2XADV]] = dr[G.m2[M2XADV]] + fac1 - fac2; dr[G.m2[M2YADV]] = dr[G.m2[M2YADV]] + fac2; dr[G.m2[M2HADV]] = dr[G.m2[M2HADV]] + fac1; }
          else { dr[G.m2[M2XADV]] = dr[G.m2[M2XADV]] - fac1 + fac2; dr[G.m2[M2YADV]] = dr[G.m2[M2YADV]] - fac2; dr[G.m2[M2HADV]] = dr[G.m2[M2HADV]] - fac1; }
        }
      }
      if (VIS4) {
        // biharmonic viscosity, second operator :1856-1921: the stress tensor of (LapU, LapV) times the total depth
#define TLU(di, dj) sLU[s + (di) + (dj) * TW]
#define TLV(di, dj) sLV[s + (di) + (dj) * TW]
#define STRESS_R4(a_, b_, v2_, pmon_, pnom_)                                                                       \
  ((v2_) * TD(a_, b_) * 0.5 *                                                                                      \
   ((pmon_) * ((TPN(a_, b_) + TPN((a_) + 1, b_)) * TLU((a_) + 1, b_) - (TPN((a_) - 1, b_) + TPN(a_, b_)) * TLU(a_, b_)) - \
    (pnom_) * ((TPM(a_, b_) + TPM(a_, (b_) + 1)) * TLV(a_, (b_) + 1) - (TPM(a_, (b_) - 1) + TPM(a_, b_)) * TLV(a_, b_))))
#define DRHS_P4(a_, b_) (0.25 * (TD(a_, b_) + TD((a_) - 1, b_) + TD(a_, (b_) - 1) + TD((a_) - 1, (b_) - 1)))
#define STRESS_P4(a_, b_, v2_, pmon_, pnom_)                                                                       \
  ((v2_) * DRHS_P4(a_, b_) * 0.5 *                                                                                 \
   ((pmon_) * ((TPN(a_, (b_) - 1) + TPN(a_, b_)) * TLV(a_, b_) - (TPN((a_) - 1, (b_) - 1) + TPN((a_) - 1, b_)) * TLV((a_) - 1, b_)) + \
    (pnom_) * ((TPM((a_) - 1, b_) + TPM(a_, b_)) * TLU(a_, b_) - (TPM((a_) - 1, (b_) - 1) + TPM(a_, (b_) - 1)) * TLU(a_, (b_) - 1))))
        const int qa = isv ? 1 : 0, qb = isv ? 0 : 1;
        const double sr0 = STRESS_R4(0, 0, mr[x].v[MR_V2], mr[x].v[MR_PMON], mr[x].v[MR_PNOM]);
        const double sr1 = STRESS_R4(a1, b1, mr[x1].v[MR_V2], mr[x1].v[MR_PMON], mr[x1].v[MR_PNOM]);
        double sp0 = STRESS_P4(0, 0, mp[x].v[MP_V2], mp[x].v[MP_PMON], mp[x].v[MP_PNOM]);
        double sp1 = STRESS_P4(qa, qb, mp[q1].v[MP_V2], mp[q1].v[MP_PMON], mp[q1].v[MP_PNOM]);
        if (MSK) { sp0 = sp0 * G.pmask[x]; sp1 = sp1 * G.pmask[q1]; }
        const double or0 = isv ? mr[x].v[MR_OM] : mr[x].v[MR_ON], or1 = isv ? mr[x1].v[MR_OM] : mr[x1].v[MR_ON];
        const double op0 = isv ? mp[x].v[MP_ON] : mp[x].v[MP_OM], op1 = isv ? mp[q1].v[MP_ON] : mp[q1].v[MP_OM];
        if (!isv) {
          const double UFx0 = or0 * or0 * sr0, UFxm = or1 * or1 * sr1, UFe0 = op0 * op0 * sp0, UFep = op1 * op1 * sp1;
          const double cff1 = 0.5 * (TPN(-1, 0) + TPN(0, 0)) * (UFx0 - UFxm);
          const double cff2 = 0.5 * (TPM(-1, 0) + TPM(0, 0)) * (UFep - UFe0);
          const double fac = cff1 + cff2;
          rhs = rhs - fac;
        } else {
          const double VFx0 = op0 * op0 * sp0, VFxp = op1 * op1 * sp1, VFe0 = or0 * or0 * sr0, VFem = or1 * or1 * sr1;
          const double cff1 = 0.5 * (TPN(0, -1) + TPN(0, 0)) * (VFxp - VFx0);
          const double cff2 = 0.5 * (TPM(0, -1) + TPM(0, 0)) * (VFe0 - VFem);
          const double fac = cff1 - cff2;
          rhs = rhs - fac;
        }
#undef TLU
#undef TLV
#undef STRESS_R4
#undef DRHS_P4
#undef STRESS_P4
      } else if (VIS) {
        // harmonic viscosity :1567-1660: stress at the rho points P0, P1 and the psi points Q0, Q1
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
        const double sr0 = STRESS_R(0, 0, WV(w_v2r0, isv, mr[x].v[MR_V2]), WV(w_pmr0, isv, mr[x].v[MR_PMON]), WV(w_pnr0, isv, mr[x].v[MR_PNOM]));
        const double sr1 = STRESS_R(a1, b1, WV(w_v2r1, isv, mr[x1].v[MR_V2]), WV(w_pmr1, isv, mr[x1].v[MR_PMON]), WV(w_pnr1, isv, mr[x1].v[MR_PNOM]));
        double sp0 = STRESS_P(0, 0, WV(w_v2p0, isv, mp[x].v[MP_V2]), WV(w_pmp0, isv, mp[x].v[MP_PMON]), WV(w_pnp0, isv, mp[x].v[MP_PNOM]));
        double sp1 = STRESS_P(qa, qb, WV(w_v2p1, isv, mp[q1].v[MP_V2]), WV(w_pmp1, isv, mp[q1].v[MP_PMON]), WV(w_pnp1, isv, mp[q1].v[MP_PNOM]));
        if (MSK) { sp0 = sp0 * G.pmask[x]; sp1 = sp1 * G.pmask[q1]; }   // :1613
        if (WD) { sp0 = sp0 * G.pmask_wet[x]; sp1 = sp1 * G.pmask_wet[q1]; }   // :1617
        // on_r (u) | om_r (v) at P0, P1; om_p (u) | on_p (v) at Q0, Q1
        const double or0 = WV(w_or0, isv, isv ? mr[x].v[MR_OM] : mr[x].v[MR_ON]);
        const double or1 = WV(w_or1, isv, isv ? mr[x1].v[MR_OM] : mr[x1].v[MR_ON]);
        const double op0 = WV(w_op0, isv, isv ? mp[x].v[MP_ON] : mp[x].v[MP_OM]);
        const double op1 = WV(w_op1, isv, isv ? mp[q1].v[MP_ON] : mp[q1].v[MP_OM]);
        if (!isv) {
          const double UFx0 = or0 * or0 * sr0;
          const double UFxm = or1 * or1 * sr1;
          const double UFe0 = op0 * op0 * sp0;
          const double UFep = op1 * op1 * sp1;
          const double cff1 = 0.5 * (TPN(-1, 0) + TPN(0, 0)) * (UFx0 - UFxm);
          const double cff2 = 0.5 * (TPM(-1, 0) + TPM(0, 0)) * (UFep - UFe0);
          const double fac = cff1 + cff2;
          rhs = rhs + fac;
          if (DUV) { dr[G.m2[M2HVIS]] = fac; dr[G.m2[M2XVIS]] = cff1; dr[G.m2[M2YVIS]] = cff2; }           // :1633
        } else {
          const double VFx0 = op0 * op0 * sp0;
          const double VFxp = op1 * op1 * sp1;
          const double VFe0 = or0 * or0 * sr0;
          const double VFem = or1 * or1 * sr1;
          const double cff1 = 0.5 * (TPN(0, -1) + TPN(0, 0)) * (VFxp - VFx0);
          const double cff2 = 0.5 * (TPM(0, -1) + TPM(0, 0)) * (VFe0 - VFem);
          const double fac = cff1 - cff2;
          rhs = rhs + fac;
          if (DUV) { dr[G.m2[M2HVIS]] = fac; dr[G.m2[M2XVIS]] = cff1; dr[G.m2[M2YVIS]] = -cff2; }          // :1646
        }
#undef STRESS_R
#undef DRHS_P
#undef STRESS_P
      }
      // -- coupling with the 3-D forcing :2225-2460, then the momentum step :2488-2670
      double *frc = isv ? F.rvfrc : F.rufrc;
      double *r3 = isv ? F.rv : F.ru;
      const double *rb = isv ? F.rvbar : F.rubar;
      double r = rhs;
      if (NDG) {
        // nudging towards the 2-D momentum climatology :2179-2203 (Drhs at the point's two rho points, ubar / vbar(krhs))
        const int dn = isv ? TW : 1;
        const int xm = isv ? x - ni : x - 1;
        const double c0 = 0.25 * (a.M2nudgcof[xm] + a.M2nudgcof[x]);
        const double cffn = isv ? c0 * F.om_v[x] * a.on_v[x] : c0 * a.om_u[x] * F.on_u[x];
        r = r + cffn * (Drhs[s - dn] + Drhs[s]) * ((isv ? a.vbarclm : a.ubarclm)[x] - (isv ? sVk : sUk)[s]);
      }
      const double mwet = WD ? (isv ? G.vmask_wet : G.umask_wet)[x] : 1.0;
      if (WD) r = r * wd_fac(mwet, r);                                    // :2205-2222
      double fr = 0.0;
      if (first) {
        fr = WV(w_frc, isv, frc[x]) - r;
        frc[x] = fr;
        if (startup == 0) r = r + fr;
        else if (startup == 1) r = r + 1.5 * fr - 0.5 * WV(w_r0n, isv, r3[o_r0n + (size_t)x]);
        else r = r + (23.0 / 12.0) * fr - (16.0 / 12.0) * WV(w_r0n, isv, r3[o_r0n + (size_t)x]) +
                 (5.0 / 12.0) * WV(w_r0s, isv, r3[o_r0s + (size_t)x]);
        r3[o_r0s + (size_t)x] = fr;
      } else {
        r = r + WV(w_frc, isv, frc[x]);
      }
      if (DUV) {
        // coupling of the terms with their 3-D vertical sums DiaRUfrc (:2249-2455), then their fast-time integration
        // (:2676-2743) and the predictor's copy for the next corrector (:2852-2863)
        const int dir = isv, Mp = G.m2[M2PGRD], Ms = G.m2[M2SSTR], Mb = G.m2[M2BSTR], nd = G.ndm2 - 1;
        if (first) {
          const double c1_ = startup == 0 ? 1.0 : (startup == 1 ? 1.5 : 23.0 / 12.0);
          const double c2_ = startup == 1 ? 0.5 : 16.0 / 12.0, c3_ = 5.0 / 12.0;
          for (int id = 1; id <= nd; id++) {
            if (id > Mp && id != Ms && id != Mb) continue;
            double *f3 = duv_rfrc(G, F, dir, 3, id) + x, *fn = duv_rfrc(G, F, dir, nnew, id) + x, *fs = duv_rfrc(G, F, dir, nstp, id) + x;
            double v3 = *f3;
            if (id <= Mp) { v3 = v3 - dr[id]; *f3 = v3; }
            // (the reference's left-to-right sums: DiaU2rhs + cff1*frc(3) - cff2*frc(nnew) + cff3*frc(nstp))
            if (id <= Mp) {
              if (startup == 0) dr[id] = dr[id] + v3;
              else if (startup == 1) dr[id] = dr[id] + c1_ * v3 - c2_ * *fn;
              else dr[id] = dr[id] + c1_ * v3 - c2_ * *fn + c3_ * *fs;
            } else {
              if (startup == 0) dr[id] = v3;
              else if (startup == 1) dr[id] = c1_ * v3 - c2_ * *fn;
              else dr[id] = c1_ * v3 - c2_ * *fn + c3_ * *fs;
            }
            *fs = v3;
          }
        } else {
          for (int id = 1; id <= Mp; id++) dr[id] = dr[id] + duv_rfrc(G, F, dir, 3, id)[x];
          dr[Ms] = duv_rfrc(G, F, dir, 3, Ms)[x];
          dr[Mb] = duv_rfrc(G, F, dir, 3, Mb)[x];
        }
        if (MSK) { const double m_ = (isv ? G.vmask : G.umask)[x]; for (int id = 1; id <= nd; id++) dr[id] = dr[id] * m_; }
        if (!PRED) {
          const double facw = a.w1_0;
          const double pmn = isv ? (sPn[s] + sPn[s - d1]) : (sPm[s - d1] + sPm[s]);
          for (int id = 1; id <= nd; id++) {
            double *I = duv_2int(G, F, dir, id) + x, *W = duv_2wrk(G, F, dir, id) + x;
            if (iif == 1) {
              const double v = 0.5 * dtfast * dr[id];
              *I = v;
              *W = v * pmn * facw;
            } else {
              const double v = *I + (k1 * dr[id] + k2 * duv_rbar(G, F, dir, kstp, id)[x] - k3 * duv_rbar(G, F, dir, ptsk, id)[x]);
              *I = v;
              *W = *W + v * pmn * facw;
            }
          }
        } else {
          for (int id = 1; id <= nd; id++) duv_rbar(G, F, dir, krhs, id)[x] = dr[id];
        }
      }
      const double cff = (sPm[s] + sPm[s - d1]) * (sPn[s] + sPn[s - d1]);
      const double fac = 1.0 / (Dnew[s] + Dnew[s - d1]);
      const double Dstp0 = sDstp[s], Dstp1 = sDstp[s - d1];
      const double sv = WV(w_s, isv, (isv ? F.vbar : F.ubar)[x + o_kstp]);
      double b;
      if (!corr) b = (sv * (Dstp0 + Dstp1) + cff * c1 * r) * fac;
      else b = (sv * (Dstp0 + Dstp1) +
                cff * (k1 * r + k2 * WV(w_rs, isv, rb[x + o_kstp]) - k3 * WV(w_rp, isv, rb[x + o_ptsk]))) * fac;
      if (MSK) b = b * (isv ? G.vmask : G.umask)[x];                      // :2515-2660
      if (WD) {                                                           // :2518-2529 ... :2661-2667
        const double cff7 = wd_fac(mwet, b);
        b = b * cff7;
        r = r * cff7;
        if (first) { const double f2 = fr * cff7; frc[x] = f2; r3[o_r0s + (size_t)x] = f2; }
      }
      // u2dbc/v2dbc :2871-2876 + exchange :3043
      if (!isv) {
        if (fuse) hb_emit(G, B, un, BC_U, i, j, b, MSK ? G.umask : nullptr);
        else un[x] = b;
        if (PRED) rub_k[x] = r;
      } else {
        if (fuse) hb_emit(G, B, vn, BC_V, i, j, b, MSK ? G.vmask : nullptr);
        else vn[x] = b;
        if (PRED) rvb_k[x] = r;
      }
    }
    }
  }
  S2D_TICK(5);
#undef TU
#undef TV
#undef TDU
#undef TDV
#undef TD
#undef TPM
#undef TPN
}
#undef RLOOP
#undef PWDECL
#undef PWLOAD
#undef PW
#undef WLOOP
#undef WDECL
#undef WLOAD
#undef WV
#undef INR

// entry points: sub-tiles up to 32x4 (384 threads: one rectangle point per thread, u- and v-points on
// different waves), up to 64x8 (1024 threads likewise, or 512 threads with two points each), up to
// 32x8, and the generic form
COOP_KERNEL(k_step2d_a, Step2dArgs) { k_step2d_t_body<32, 4, 384, 1>(a, bx, by, bz, lds); }
COOP_GLOBAL_LB(k_step2d_a, Step2dArgs, 384)
COOP_KERNEL(k_step2d_ac, Step2dArgs) { k_step2d_t_body<32, 4, 384, 1, false, true>(a, bx, by, bz, lds); }   // ... behind a pair launch: commits the staged level
COOP_GLOBAL_LB(k_step2d_ac, Step2dArgs, 384)
COOP_KERNEL(k_step2d_am, Step2dArgs) { k_step2d_t_body<32, 4, 384, 1, true>(a, bx, by, bz, lds); }   // the same with land/sea masks
COOP_GLOBAL_LB(k_step2d_am, Step2dArgs, 384)
COOP_KERNEL(k_step2d_b, Step2dArgs) { k_step2d_t_body<64, 8, 512, 2>(a, bx, by, bz, lds); }
COOP_GLOBAL_LB(k_step2d_b, Step2dArgs, 512)
COOP_KERNEL(k_step2d_d, Step2dArgs) { k_step2d_t_body<64, 8, 1024, 1>(a, bx, by, bz, lds); }
COOP_GLOBAL_LB(k_step2d_d, Step2dArgs, 1024)
COOP_KERNEL(k_step2d_c, Step2dArgs) { k_step2d_t_body<32, 8, 512, 2, false, false, 1>(a, bx, by, bz, lds); }
COOP_GLOBAL_LB2(k_step2d_c, Step2dArgs, 512, 4)   // two blocks of 8 waves per CU: at most 128 VGPRs
COOP_KERNEL(k_step2d, Step2dArgs) { k_step2d_t_body<0, 0, 0, 0>(a, bx, by, bz, lds); }
COOP_GLOBAL_LB(k_step2d, Step2dArgs, 512)
COOP_KERNEL(k_step2d_vis4, Step2dArgs) { k_step2d_t_body<0, 0, 0, 0, true, true, 0, false, true>(a, bx, by, bz, lds); }   // ... with the biharmonic viscosity (UV_VIS4)
COOP_GLOBAL_LB(k_step2d_vis4, Step2dArgs, 512)
COOP_KERNEL(k_step2d_wd, Step2dArgs) { k_step2d_t_body<0, 0, 0, 0, true, true, 0, false, false, true>(a, bx, by, bz, lds); }   // ... with wetting and drying (WET_DRY)
COOP_GLOBAL_LB(k_step2d_wd, Step2dArgs, 512)
COOP_KERNEL(k_step2d_ndg, Step2dArgs) { k_step2d_t_body<0, 0, 0, 0, true, true, 0, false, false, false, true>(a, bx, by, bz, lds); }   // ... with LnudgeM2CLM
COOP_GLOBAL_LB(k_step2d_ndg, Step2dArgs, 512)
COOP_KERNEL(k_step2d_duv, Step2dArgs) { k_step2d_t_body<0, 0, 0, 0, true, true, 0, true>(a, bx, by, bz, lds); }   // ... with the momentum diagnostics (DIAGNOSTICS_UV)
COOP_GLOBAL_LB(k_step2d_duv, Step2dArgs, 512)
