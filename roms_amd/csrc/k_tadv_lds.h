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

template <int MODE, int MINW>
static __global__ void __launch_bounds__(256, MINW) k_tadv_lds(const KArgs a, int nx, int ny, int nz) {
  extern __shared__ double lds_dyn_[];
  const int nby_ = (ny + TL_BY - 1) / TL_BY, nt_ = ((nx + TL_BX - 1) / TL_BX) * nby_, seg_ = (nt_ + 7) / 8;
  const int r_ = (int)(blockIdx.x >> 3), xcd_ = (int)(blockIdx.x & 7);
  const int gz = r_ % nz;
  const int t_ = xcd_ * seg_ + r_ / nz;
  if (t_ >= nt_) return;
  KTILE_XY(t_, (nx + TL_BX - 1) / TL_BX, nby_, tbx, tby);
  const DGrid &G = a.G;
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
  double st[TL_MAXT][TL_NLD];
  auto stage_load = [&](int k) {
    const size_t ok = (size_t)(k - 1) * nij;
#pragma unroll
    for (int it = 0; it < TL_MAXT; it++)
#pragma unroll
      for (int m = 0; m < TL_NLD; m++) st[it][m] = (gok[m] && it < NT) ? Tsrc[it][ok + gofs[m]] : 0.0;
  };
  auto stage_store = [&](double *buf) {
#pragma unroll
    for (int m = 0; m < TL_NLD; m++)
      if (tid + m * 256 < TL_NT) {
#pragma unroll
        for (int it = 0; it < TL_MAXT; it++) buf[it * TL_NT + tid + m * 256] = st[it][m];
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
    vert[it] = MODE == 0 ? true : (vs[it] != ROMS_HSIMT && vs[it] != ROMS_MPDATA && vs[it] != ROMS_SPLINES);
    const double GammaH = (hs[it] == ROMS_MPDATA || hs[it] == ROMS_HSIMT) ? 0.5 : 1.0 / 6.0;
    if (G.iic == G.ntfirst) { cffp[it] = 0.5 * G.dt; cff1p[it] = 1.0; cff2p[it] = 0.0; }
    else { cffp[it] = (1.0 - GammaH) * G.dt; cff1p[it] = 0.5 + GammaH; cff2p[it] = 0.5 - GammaH; }
    const double GammaV = (vs[it] == ROMS_MPDATA || vs[it] == ROMS_HSIMT) ? 0.5 : 1.0 / 6.0;
    cfv[it] = (G.iic == G.ntfirst) ? 0.5 * G.dt : (1.0 - GammaV) * G.dt;
  }
  EmitPlan P3;
  if (MODE == 0 && inside) P3 = emit_plan(G, BC_R, i, j);
  // vertical window of the advected tracer: levels k-1 .. k+2 (clamped to 1..N); flux through the interface below
#define TL_Q(p3, kk) (p3)[(size_t)(KMIN(KMAX((kk), 1), N) - 1) * nij + x]
  double tq[TL_MAXT][4], FCm[TL_MAXT];
  double wm = 0.0;                                        // W(k-1)
  if (inside) {
    wm = F.W[x + (size_t)KMIN(k0 - 1, N) * nij];
#pragma unroll
    for (int it = 0; it < TL_MAXT; it++) {
      FCm[it] = 0.0;
      if (it < NT) {
#pragma unroll
        for (int q = 0; q < 4; q++) tq[it][q] = TL_Q(Tsrc[it], k0 - 1 + q);
        if (vert[it]) {
          const double tm2 = TL_Q(Tsrc[it], k0 - 2);
          VFLUX_REL(FCm[it], vs[it], k0 - 1, N, tm2, tq[it][0], tq[it][1], tq[it][2], wm);
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
  auto own_load = [&](int k) {
    const size_t ok = (size_t)(k - 1) * nij;
    pn_hu0 = F.Huon[ok + x]; pn_hup = F.Huon[ok + x + 1]; pn_hv0 = F.Hvom[ok + x]; pn_hvp = F.Hvom[ok + x + ni];
    pn_Hz = F.Hz[ok + x];
    pn_w = F.W[x + (size_t)KMIN(k, N) * nij];
#pragma unroll
    for (int it = 0; it < TL_MAXT; it++) {
      pn_tq3[it] = it < NT ? TL_Q(Tsrc[it], k + 3) : 0.0;
      pn_told[it] = it < NT ? (F.t + XT(G.LBi, G.LBj, 1, G.nnew, it + 1))[ok + x] : 0.0;     // t(nnew): read by both modes
    }
  };
  if (inside) own_load(k0);
  for (int k = k0; k <= k1; k++) {
    const double *cur = lds_dyn_ + ((k - k0) & 1) * (TL_MAXT * TL_NT);
    double *nxt = lds_dyn_ + ((k - k0 + 1) & 1) * (TL_MAXT * TL_NT);
    const double hu0 = pn_hu0, hup = pn_hup, hv0 = pn_hv0, hvp = pn_hvp, Hzk = pn_Hz, w0 = pn_w;
    double c_tq3[TL_MAXT], c_told[TL_MAXT];
#pragma unroll
    for (int it = 0; it < TL_MAXT; it++) { c_tq3[it] = pn_tq3[it]; c_told[it] = pn_told[it]; }
    if (k < k1) {
      stage_load(k + 1);
      if (inside) own_load(k + 1);
    }
    if (inside) {
      const size_t ok = (size_t)(k - 1) * nij;
#pragma unroll
      for (int it = 0; it < TL_MAXT; it++) {
        if (it >= NT) break;
        if (!((a.p1 >> it) & 1)) continue;                            // a tracer of another kernel (MPDATA, HSIMT)
        const double tq3 = c_tq3[it];                                 // enters the window after this level
        double FX0, FXp, FE0, FEp;
        hadv4_core(G, hs[it], cur + it * TL_NT + s, (long)TL_TW, hu0, hup, hv0, hvp, i, j, FX0, FXp, FE0, FEp);
        double FCk = 0.0;
        if (vert[it]) VFLUX_REL(FCk, vs[it], k, N, tq[it][0], tq[it][1], tq[it][2], tq[it][3], w0);
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
            tt = tt * (1.0 / Hzk);
          }
          tn[ok] = tt;
        }
        FCm[it] = FCk;
        tq[it][0] = tq[it][1]; tq[it][1] = tq[it][2]; tq[it][2] = tq[it][3]; tq[it][3] = tq3;
      }
      wm = w0;
    }
    if (k < k1) {
      stage_store(nxt);
      __syncthreads();
    }
  }
#undef TL_Q
}
