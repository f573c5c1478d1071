// k_tadv_lds.h -- the two tracer-advection point kernels of the north-star pair with the horizontal neighbourhood
// of a level staged through LDS:
//   MODE 0  k_pre_t3   pre_step3d_tile, tracer predictor t(3)   (ROMS/Nonlinear/pre_step3d.F:357-852)
//   MODE 1  k_s3t_hv   step3d_t_tile, corrector advection of t(3) into t(nnew)   (step3d_t.F:633-1340)
// The point-wise forms read the advected tracer at nine neighbours of every point and level and run one thread per
// TRACER and chunk, so the mass fluxes, Hz and W are fetched once per tracer.  Here a block of 64x4 points marches a
// chunk of levels for ALL tracers: per level the (64+4)x(4+4) rectangle of each tracer is loaded once per block
// (2.1 loads per point instead of 9), double-buffered through registers with one barrier per level as in
// k_rhs3d_lds.h; Huon, Hvom, Hz, W of a point are read once for all tracers; the vertical flux of a column is
// carried in registers (window t(k-1..k+2), the flux through the interface below).  Every expression is the
// point-wise kernel's, operand for operand (the face fluxes are hadv4_core on the tile): same bits
// (tests/test_gpu_parity.py::test_column_kernel_forms_agree_bitwise, ROMS_HIP_TADV_LDS=0 selects the point-wise
// forms, which the serial CPU emulation and tracers on the column paths -- HSIMT, SPLINES, MPDATA -- keep).
//
// Round 4, HS = true (MODE 1 only): tracers with Hadvection = HSIMT and / or Vadvection = HSIMT are advected HERE too
// instead of by k_s3t_h (one block per sub-tile and LEVEL, its own loads of Hz, Huon, Hvom, pm, pn: 324 us at 512x512x50
// for one tracer) and the HSIMT sweep of k_s3t_col (+227 us).  Besides the tracers' rectangles a block stages 1/Hz, Huon
// and Hvom of the level; a point forms the gradient and the KaX / KaE of the four xi faces i-1 .. i+2 and the four eta
// faces j-1 .. j+2 it needs from the tiles -- step3d_t.F:472-632's expressions, the metric factor of a face kept in
// registers for the whole march -- and the four limited face fluxes as k_s3t_h does; the vertical HSIMT flux
// (:1069-1150) comes from the column window (t at k-1 .. k+2, z_r, W) with KaZ of three interfaces carried upward.
// Same operations on the same operands as k_s3t_h / k_s3t_col (ROMS_HIP_HSIMT_LDS=0 selects those; the forms test
// compares the bits).
#pragma once
#include "roms_ctx.h"

#ifndef TL_BX
#define TL_BX 64                        // interior of a block: TL_BX x TL_BY points, 256 threads
#define TL_BY 4
#endif
#define TL_TW (TL_BX + 4)
#define TL_TH (TL_BY + 4)
#define TL_NLD ((TL_TW * TL_TH + 255) / 256)   // staged loads per thread, tracer and level
#define TL_NT (TL_TW * TL_TH)           // 544 values per tracer and level
#define TL_MAXT 2                       // tracers per block (NT <= 2: temperature and salinity)
#define TL_LDS_DOUBLES (2 * TL_MAXT * TL_NT)
#define TL_LDS_DOUBLES_HS (2 * (TL_MAXT + 3) * TL_NT)      // + 1/Hz, Huon, Hvom

// KaZ of the interface kk (between rho levels kk and kk+1), step3d_t.F:1080-1090: zero at the bottom and the surface
KDEV double hsimt_kaz(int kk, int N, double cK, double w, double zlo, double zhi) {
  return (kk <= 0 || kk >= N) ? 0.0 : 1.0 - fabs(cK * w / (zhi - zlo));
}
// HSIMT vertical flux FC(k), 1 <= k <= N (:1092-1150): t at k-1 .. k+2, KaZ at k-1, k, k+1, w = W(k)
KDEV double hsimt_vflux(int k, int N, double w, double tkm1, double tk, double tkp1, double tkp2, double KAm, double KA0,
                        double KAp) {
  if (k >= N) return 0.0;
  if (k == 1 && w >= 0.0) return w * tk;
  if (k == N - 1 && w < 0.0) return w * tkp1;
  const double Ka = KA0, oKa = 1.0 / Ka;
  const double GZ0 = tkp1 - tk;
  const double GZm = (k - 1 >= 1) ? tk - tkm1 : 0.0;
  const double GZp = (k + 1 < N) ? tkp2 - tkp1 : 0.0;
  double sw;
  if (w >= 0.0) sw = tk + hsimt_lim(GZ0, GZm, Ka, KAm, oKa);
  else sw = tkp1 - hsimt_lim(GZ0, GZp, Ka, KAp, oKa);
  return w * sw;
}

template <int MODE, int MINW, bool HS = false>
static __global__ void __launch_bounds__(256, MINW) k_tadv_lds(const KArgs a, int nx, int ny, int nz) {
  static_assert(!HS || MODE == 1, "the HSIMT paths belong to the corrector");
  constexpr int NA = HS ? TL_MAXT + 3 : TL_MAXT;          // staged arrays per level
  extern __shared__ double lds_dyn_[];
  const int nby_ = (ny + TL_BY - 1) / TL_BY, nt_ = ((nx + TL_BX - 1) / TL_BX) * nby_, seg_ = (nt_ + 7) / 8;
  const int r_ = (int)(blockIdx.x >> 3), xcd_ = (int)(blockIdx.x & 7);
  const int gz = r_ % nz;
  const int t_ = xcd_ * seg_ + r_ / nz;
  if (t_ >= nt_) return;
  KTILE_XY(t_, (nx + TL_BX - 1) / TL_BX, nby_, tbx, tby);
  const DGrid &G = a.G;
  if (G.region) {            // rim / interior split: a block belongs to the rim if any of its points does (block-uniform: barriers follow)
    const bool rim = tbx * TL_BX < G.rimw || tbx * TL_BX + TL_BX - 1 >= nx - G.rimw || tby * TL_BY < G.rimw || tby * TL_BY + TL_BY - 1 >= ny - G.rimw;
    if ((G.region == 1) != rim) return;
  }
  const Fields &F = a.Fv;
  const TB &B = G.T;
  const int KC = a.p0, N = G.N, NT = G.NT;
  const int k0 = gz * KC + 1, k1 = KMIN(N, k0 + KC - 1);
  if (k0 > N) return;
  const int tx = (int)threadIdx.x, ty = (int)threadIdx.y, tid = tx + TL_BX * ty;
  const int I0 = B.Istr + tbx * TL_BX, J0 = B.Jstr + tby * TL_BY;
  const int i = I0 + tx, j = J0 + ty;
  const bool inside = i <= B.Iend && j <= B.Jend;
  const size_t nij = (size_t)G.nij;
  const long ni = G.ni;
  const int UBi = G.LBi + G.ni - 1, UBj = G.LBj + G.nj - 1;
  // the advected tracer: t(nstp) in the predictor, t(3) in the corrector
  const double *Tsrc[TL_MAXT];
#pragma unroll
  for (int it = 0; it < TL_MAXT; it++) Tsrc[it] = F.t + XT(G.LBi, G.LBj, 1, MODE == 0 ? G.nstp : 3, KMIN(it + 1, NT));

  long gofs[TL_NLD];
  bool gok[TL_NLD];
#pragma unroll
  for (int m = 0; m < TL_NLD; m++) {
    const int e = tid + m * 256;
    const int row = e / TL_TW, col = e - row * TL_TW;
    const int gi = I0 - 2 + col, gj = J0 - 2 + row;
    gok[m] = e < TL_NT && gi >= G.LBi && gi <= UBi && gj >= G.LBj && gj <= UBj;
    gofs[m] = gok[m] ? (long)X2(gi, gj) : 0;
  }
  double st[NA][TL_NLD];
  auto stage_load = [&](int k) {
    const size_t ok = (size_t)(k - 1) * nij;
#pragma unroll
    for (int it = 0; it < TL_MAXT; it++)
#pragma unroll
      for (int m = 0; m < TL_NLD; m++) st[it][m] = (gok[m] && it < NT) ? Tsrc[it][ok + gofs[m]] : 0.0;
    if (HS) {
#pragma unroll
      for (int m = 0; m < TL_NLD; m++) {
        st[TL_MAXT][m] = gok[m] ? F.Hz[ok + gofs[m]] : 1.0;
        st[TL_MAXT + 1][m] = gok[m] ? F.Huon[ok + gofs[m]] : 0.0;
        st[TL_MAXT + 2][m] = gok[m] ? F.Hvom[ok + gofs[m]] : 0.0;
      }
    }
  };
  auto stage_store = [&](double *buf) {
#pragma unroll
    for (int m = 0; m < TL_NLD; m++)
      if (tid + m * 256 < TL_NT) {
#pragma unroll
        for (int it = 0; it < TL_MAXT; it++) buf[it * TL_NT + tid + m * 256] = st[it][m];
        if (HS) {
          buf[TL_MAXT * TL_NT + tid + m * 256] = 1.0 / st[TL_MAXT][m];          // 1/Hz (:481, :556)
          buf[(TL_MAXT + 1) * TL_NT + tid + m * 256] = st[TL_MAXT + 1][m];
          buf[(TL_MAXT + 2) * TL_NT + tid + m * 256] = st[TL_MAXT + 2][m];
        }
      }
  };

  // ---- own column
  const long x = inside ? (long)X2(i, j) : (long)X2(B.Istr, B.Jstr);
  const double pmv = F.pm[x], pnv = F.pn[x];
  const double cffc = G.dt * pmv * pnv;                  // corrector: cff = dt*pm*pn
  int hs[TL_MAXT], vs[TL_MAXT];
  bool vert[TL_MAXT];
  double cffp[TL_MAXT], cff1p[TL_MAXT], cff2p[TL_MAXT], cfv[TL_MAXT];     // predictor constants per tracer
#pragma unroll
  for (int it = 0; it < TL_MAXT; it++) {
    const int itc = KMIN(it, NT - 1);
    hs[it] = G.hadv[itc]; vs[it] = G.vadv[itc];
    // (HS: the HSIMT vertical flux is formed here too; a tracer with an HSIMT horizontal step and a column scheme in the
    //  vertical -- SPLINES -- still leaves the vertical step to k_s3t_col)
    vert[it] = MODE == 0 ? true : ((vs[it] != ROMS_HSIMT || HS) && vs[it] != ROMS_MPDATA && vs[it] != ROMS_SPLINES);
    const double GammaH = (hs[it] == ROMS_MPDATA || hs[it] == ROMS_HSIMT) ? 0.5 : 1.0 / 6.0;
    if (G.iic == G.ntfirst) { cffp[it] = 0.5 * G.dt; cff1p[it] = 1.0; cff2p[it] = 0.0; }
    else { cffp[it] = (1.0 - GammaH) * G.dt; cff1p[it] = 0.5 + GammaH; cff2p[it] = 0.5 - GammaH; }
    const double GammaV = (vs[it] == ROMS_MPDATA || vs[it] == ROMS_HSIMT) ? 0.5 : 1.0 / 6.0;
    cfv[it] = (G.iic == G.ntfirst) ? 0.5 * G.dt : (1.0 - GammaV) * G.dt;
  }
  // HS: metric factor and validity of the xi faces i-1 .. i+2 and the eta faces j-1 .. j+2 of this point (:476-480, :551-555;
  // faces beyond a closed edge carry a zero gradient and KaX = 0, :533-547 / :610-624, as in k_s3t_h)
  double cffx[4] = {0, 0, 0, 0}, cffe[4] = {0, 0, 0, 0};
  // (validity of a face is re-derived level by level -- one subtraction and one unsigned compare -- rather than kept in
  //  eight lane masks: the kernel is short of scalar registers)
  const int fxlo = B.IstrU - 1, fxn = B.Iendp2 - fxlo, felo = B.JstrV - 1, fen = B.Jendp2 - felo;
#define TL_FXV(q) ((unsigned)(i - 1 + (q) - fxlo) <= (unsigned)fxn)
#define TL_FEV(q) ((unsigned)(j - 1 + (q) - felo) <= (unsigned)fen)
  const double cK = pmv * pnv * G.dt;                    // the column kernels' order of pm*pn*dt (k_s3t_col: cK)
  if (HS && inside) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int fi = i - 1 + q, fj = j - 1 + q;
      if (TL_FXV(q)) cffx[q] = 0.125 * (F.pm[X2(fi - 1, j)] + F.pm[X2(fi, j)]) * (F.pn[X2(fi - 1, j)] + F.pn[X2(fi, j)]) * G.dt;
      if (TL_FEV(q)) cffe[q] = 0.125 * (F.pn[X2(i, fj)] + F.pn[X2(i, fj - 1)]) * (F.pm[X2(i, fj)] + F.pm[X2(i, fj - 1)]) * G.dt;
    }
  }
  EmitPlan P3;
  if (MODE == 0 && inside) P3 = emit_plan(G, BC_R, i, j);
  // vertical window of the advected tracer: levels k-1 .. k+2 (clamped to 1..N); flux through the interface below
#define TL_Q(p3, kk) (p3)[(size_t)(KMIN(KMAX((kk), 1), N) - 1) * nij + x]
  double tq[TL_MAXT][4], FCm[TL_MAXT];
  double wm = 0.0;                                        // W(k-1)
  // HS: KaZ of the interfaces k-1 and k (k+1 is formed level by level), z_r(k+1)
  double kam = 0.0, ka0 = 0.0, zc1 = 0.0;
  bool hsv = false;                                       // some tracer of this block takes the HSIMT vertical flux
  if (HS) {
#pragma unroll
    for (int it = 0; it < TL_MAXT; it++) hsv |= it < NT && ((a.p1 >> it) & 1) && vs[it] == ROMS_HSIMT;
  }
  if (inside) {
    wm = F.W[x + (size_t)KMIN(k0 - 1, N) * nij];
    double ka_m2 = 0.0;
    if (HS && hsv) {
#define TL_Z(kk) F.z_r[(size_t)(KMIN(KMAX((kk), 1), N) - 1) * nij + x]
#define TL_W(kk) F.W[(size_t)KMIN(KMAX((kk), 0), N) * nij + x]
      ka_m2 = hsimt_kaz(k0 - 2, N, cK, TL_W(k0 - 2), TL_Z(k0 - 2), TL_Z(k0 - 1));
      kam = hsimt_kaz(k0 - 1, N, cK, wm, TL_Z(k0 - 1), TL_Z(k0));
      ka0 = hsimt_kaz(k0, N, cK, TL_W(k0), TL_Z(k0), TL_Z(k0 + 1));
      zc1 = TL_Z(k0 + 1);
    }
#pragma unroll
    for (int it = 0; it < TL_MAXT; it++) {
      FCm[it] = 0.0;
      if (it < NT) {
#pragma unroll
        for (int q = 0; q < 4; q++) tq[it][q] = TL_Q(Tsrc[it], k0 - 1 + q);
        if (vert[it]) {
          const double tm2 = TL_Q(Tsrc[it], k0 - 2);
          if (HS && vs[it] == ROMS_HSIMT) {
            if (k0 - 1 >= 1) FCm[it] = hsimt_vflux(k0 - 1, N, wm, tm2, tq[it][0], tq[it][1], tq[it][2], ka_m2, kam, ka0);
          } else VFLUX_REL(FCm[it], vs[it], k0 - 1, N, tm2, tq[it][0], tq[it][1], tq[it][2], wm);
        }
      } else {
#pragma unroll
        for (int q = 0; q < 4; q++) tq[it][q] = 0.0;
      }
    }
  } else {
#pragma unroll
    for (int it = 0; it < TL_MAXT; it++) {
      FCm[it] = 0.0;
#pragma unroll
      for (int q = 0; q < 4; q++) tq[it][q] = 0.0;
    }
  }

  // ---- march
  stage_load(k0);
  stage_store(lds_dyn_);
  __syncthreads();
  const int s = (ty + 2) * TL_TW + (tx + 2);
  // own-point values of a level are loaded one level ahead, like the tiles (pn_*: level k+1 while k is evaluated)
  double pn_hu0 = 0, pn_hup = 0, pn_hv0 = 0, pn_hvp = 0, pn_Hz = 0, pn_w = 0, pn_tq3[TL_MAXT], pn_told[TL_MAXT];
  double pn_wp = 0, pn_z2 = 0;                            // HS: W(k+1), z_r(k+2)
  auto own_load = [&](int k) {
    const size_t ok = (size_t)(k - 1) * nij;
    if (!HS) {        // (HS: the staged tiles of Huon, Hvom and 1/Hz hold these)
      pn_hu0 = F.Huon[ok + x]; pn_hup = F.Huon[ok + x + 1]; pn_hv0 = F.Hvom[ok + x]; pn_hvp = F.Hvom[ok + x + ni];
      pn_Hz = F.Hz[ok + x];
    }
    pn_w = F.W[x + (size_t)KMIN(k, N) * nij];
    if (HS && hsv) { pn_wp = F.W[x + (size_t)KMIN(k + 1, N) * nij]; pn_z2 = F.z_r[(size_t)(KMIN(k + 2, N) - 1) * nij + x]; }
#pragma unroll
    for (int it = 0; it < TL_MAXT; it++) {
      pn_tq3[it] = it < NT ? TL_Q(Tsrc[it], k + 3) : 0.0;
      pn_told[it] = it < NT ? (F.t + XT(G.LBi, G.LBj, 1, G.nnew, it + 1))[ok + x] : 0.0;     // t(nnew): read by both modes
    }
  };
  if (inside) own_load(k0);
  for (int k = k0; k <= k1; k++) {
    const double *cur = lds_dyn_ + ((k - k0) & 1) * (NA * TL_NT);
    double *nxt = lds_dyn_ + ((k - k0 + 1) & 1) * (NA * TL_NT);
    double hu0 = pn_hu0, hup = pn_hup, hv0 = pn_hv0, hvp = pn_hvp, oHzk = 0.0;
    const double Hzk = pn_Hz, w0 = pn_w;
    const double wp = pn_wp, z2 = pn_z2;
    if (HS) {
      const double *O = cur + TL_MAXT * TL_NT, *U = cur + (TL_MAXT + 1) * TL_NT, *V = cur + (TL_MAXT + 2) * TL_NT;
      hu0 = U[s]; hup = U[s + 1]; hv0 = V[s]; hvp = V[s + TL_TW]; oHzk = O[s];       // O[s] = 1.0 / Hz(i,j,k), the same division
    }
    double c_tq3[TL_MAXT], c_told[TL_MAXT];
#pragma unroll
    for (int it = 0; it < TL_MAXT; it++) { c_tq3[it] = pn_tq3[it]; c_told[it] = pn_told[it]; }
    if (k < k1) {
      stage_load(k + 1);
      if (inside) own_load(k + 1);
    }
    if (inside) {
      const size_t ok = (size_t)(k - 1) * nij;
      double kap = 0.0;
      if (HS && hsv) kap = hsimt_kaz(k + 1, N, cK, wp, zc1, z2);       // KaZ(k+1)
#pragma unroll
      for (int it = 0; it < TL_MAXT; it++) {
        if (it >= NT) break;
        if (!((a.p1 >> it) & 1)) continue;                            // a tracer of another kernel (MPDATA; HSIMT unless HS)
        const double tq3 = c_tq3[it];                                 // enters the window after this level
        double FX0, FXp, FE0, FEp;
        if (HS && hs[it] == ROMS_HSIMT) {
          // step3d_t.F:472-632: gradient and KaX / KaE of the faces i-1 .. i+2 | j-1 .. j+2, then the limited fluxes
          const double *Tt = cur + it * TL_NT, *O = cur + TL_MAXT * TL_NT, *U = cur + (TL_MAXT + 1) * TL_NT, *V = cur + (TL_MAXT + 2) * TL_NT;
          double gX[4], KX[4], gE[4], KE[4];
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const int sx = s + q - 1, se = s + (q - 1) * TL_TW;
            const bool vx = TL_FXV(q), ve = TL_FEV(q);
            gX[q] = vx ? Tt[sx] - Tt[sx - 1] : 0.0;
            KX[q] = vx ? 1.0 - fabs(U[sx] * (cffx[q] * (O[sx - 1] + O[sx]))) : 0.0;
            gE[q] = ve ? Tt[se] - Tt[se - TL_TW] : 0.0;
            KE[q] = ve ? 1.0 - fabs(V[se] * (cffe[q] * (O[se] + O[se - TL_TW]))) : 0.0;
          }
          const double tc = Tt[s];
          FX0 = hsimt_flux(hu0, Tt[s - 1], tc, gX[1], gX[0], gX[2], KX[1], KX[0], KX[2]);
          FXp = hsimt_flux(hup, tc, Tt[s + 1], gX[2], gX[1], gX[3], KX[2], KX[1], KX[3]);
          FE0 = hsimt_flux(hv0, Tt[s - TL_TW], tc, gE[1], gE[0], gE[2], KE[1], KE[0], KE[2]);
          FEp = hsimt_flux(hvp, tc, Tt[s + TL_TW], gE[2], gE[1], gE[3], KE[2], KE[1], KE[3]);
        } else
        hadv4_core(G, hs[it], cur + it * TL_NT + s, (long)TL_TW, hu0, hup, hv0, hvp, i, j, FX0, FXp, FE0, FEp);
        double FCk = 0.0;
        if (vert[it]) {
          if (HS && vs[it] == ROMS_HSIMT) FCk = hsimt_vflux(k, N, w0, tq[it][0], tq[it][1], tq[it][2], tq[it][3], kam, ka0, kap);
          else VFLUX_REL(FCk, vs[it], k, N, tq[it][0], tq[it][1], tq[it][2], tq[it][3], w0);
        }
        if (MODE == 0) {
          double *t3 = F.t + XT(G.LBi, G.LBj, 1, 3, it + 1);
          const double t3h = Hzk * (cff1p[it] * tq[it][1] + cff2p[it] * c_told[it]) - cffp[it] * pmv * pnv * (FXp - FX0 + FEp - FE0);
          const double cfv1 = cfv[it] * pmv * pnv;       // cff*pm*pn in the reference's order (:830, :845)
          const double DC = 1.0 / (Hzk - cfv1 * (hup - hu0 + hvp - hv0 + (w0 - wm)));
          emit_store(G, P3, t3 + ok, DC * (t3h - cfv1 * (FCk - FCm[it])));      // t3dbc + exchange :1157-1171
        } else {
          double *tn = F.t + XT(G.LBi, G.LBj, 1, G.nnew, it + 1) + x;
          const double cff1 = cffc * (FXp - FX0);
          const double cff2 = cffc * (FEp - FE0);
          const double cff3 = cff1 + cff2;
          double tt = c_told[it] - cff3;
          if (vert[it]) {
            const double cv = cffc * (FCk - FCm[it]);
            tt = tt - cv;
            tt = tt * (HS ? oHzk : 1.0 / Hzk);
          }
          tn[ok] = tt;
        }
        FCm[it] = FCk;
        tq[it][0] = tq[it][1]; tq[it][1] = tq[it][2]; tq[it][2] = tq[it][3]; tq[it][3] = tq3;
      }
      wm = w0;
      if (HS && hsv) { kam = ka0; ka0 = kap; zc1 = z2; }
    }
    if (k < k1) {
      stage_store(nxt);
      __syncthreads();
    }
  }
#undef TL_Q
#undef TL_Z
#undef TL_W
#undef TL_FXV
#undef TL_FEV
}
