// k_uv3dmix2_col.h -- uv3dmix2 (harmonic viscosity along s-surfaces, ROMS/Nonlinear/uv3dmix2_s.h:90-262) and the
// 2-D/3-D coupling sums that close rhs3d_tile (rhs3d.F:1700-1918) as ONE kernel for large grids.
//
// The chunked form needs four 3-D work arrays: k_uv3dmix2_s stores the two terms every momentum point adds to
// rufrc/rvfrc, and a column kernel adds them in level order behind the vertical sum of ru/rv -- the reference
// accumulates rufrc inside its k loop, and the additions have to be made in that order to give its bits.
// That is 8 of the 17 array passes of the two kernels.  Here a thread owns a whole column: it first forms
// the vertical sum of ru (rv) and the surface/bottom stress terms exactly as k_rhs3d_sum does, then marches
// k = 1..N evaluating the stress divergence of level k and adding its two terms to that running value at
// once -- same additions, same order, no work arrays.  The (64+2)x(4+2) rectangle of Hz, u(nrhs), v(nrhs) of
// a level is loaded once per block into LDS (the point-wise form issues 20 neighbour loads per point and
// level), double-buffered through registers, one barrier per level.  Used from 128 K columns up (a column per
// thread has to fill the chip); smaller grids and the CPU emulation keep the chunked kernels.  Bit-identical
// to them (tests/test_gpu_parity.py::test_column_kernel_forms_agree_bitwise, ROMS_HIP_UVCOL=0).
#pragma once
#include "roms_ctx.h"

#define UC_TW 66
#define UC_TH 6
#define UC_NT (UC_TW * UC_TH)          // 396 values per array and level
#define UC_LDS_DOUBLES (2 * 3 * UC_NT)

static __global__ void __launch_bounds__(256, 3) k_uv3dmix2_col(const KArgs a, int nx, int ny) {
  extern __shared__ double lds_dyn_[];
  const int nby_ = (ny + 3) / 4, nt_ = ((nx + 63) / 64) * nby_, seg_ = (nt_ + 7) / 8;
  const int r_ = (int)(blockIdx.x >> 3), xcd_ = (int)(blockIdx.x & 7);
  const int t_ = xcd_ * seg_ + r_;
  if (t_ >= nt_) return;
  KTILE_XY(t_, (nx + 63) / 64, nby_, tbx, tby);
  const DGrid &G = a.G;
  const Fields &F = a.Fv;
  const TB &B = G.T;
  const int N = G.N, nrhs = G.nrhs, nnew = G.nnew;
  const int tx = (int)threadIdx.x, ty = (int)threadIdx.y, tid = tx + 64 * ty;
  const int I0 = B.Istr + tbx * 64, J0 = B.Jstr + tby * 4;
  const int i = I0 + tx, j = J0 + ty;
  const bool inside = i <= B.Iend && j <= B.Jend;
  const bool do_u = inside && i >= B.IstrU, do_v = inside && j >= B.JstrV;
  const size_t nij = (size_t)G.nij;
  const long ni = G.ni;
  const int UBi = G.LBi + G.ni - 1, UBj = G.LBj + G.nj - 1;
  const double *u3 = F.u + (size_t)(nrhs - 1) * nij * (size_t)N, *v3 = F.v + (size_t)(nrhs - 1) * nij * (size_t)N;
  const double *pm = F.pm, *pn = F.pn;

  // ---- tile staging: e = tid + m*256 < 396
  long gofs[2];
  bool gok[2];
#pragma unroll
  for (int m = 0; m < 2; m++) {
    const int e = tid + m * 256;
    const int row = e / UC_TW, col = e - row * UC_TW;
    const int gi = I0 - 1 + col, gj = J0 - 1 + row;
    gok[m] = e < UC_NT && gi >= G.LBi && gi <= UBi && gj >= G.LBj && gj <= UBj;
    gofs[m] = gok[m] ? (long)X2(gi, gj) : 0;
  }
  double st[3][2];
  auto stage_load = [&](int k) {
    const size_t ok = (size_t)(k - 1) * nij;
#pragma unroll
    for (int m = 0; m < 2; m++) {
      if (gok[m]) { st[0][m] = F.Hz[ok + gofs[m]]; st[1][m] = u3[ok + gofs[m]]; st[2][m] = v3[ok + gofs[m]]; }
      else { st[0][m] = 0.0; st[1][m] = 0.0; st[2][m] = 0.0; }
    }
  };
  auto stage_store = [&](double *buf) {
#pragma unroll
    for (int m = 0; m < 2; m++)
      if (tid + m * 256 < UC_NT) {
        buf[0 * UC_NT + tid + m * 256] = st[0][m];
        buf[1 * UC_NT + tid + m * 256] = st[1][m];
        buf[2 * UC_NT + tid + m * 256] = st[2][m];
      }
  };
  stage_load(1);

  // ---- level-independent metric products at the thread's cell (k_uv3dmix2_t, same association order)
  const long x = inside ? (long)X2(i, j) : (long)X2(B.Istr, B.Jstr);
#define RSET(c, o)                                                                                \
  const double c##0 = F.pmon_r[o], c##1 = pn[o] + pn[(o) + 1], c##2 = pn[(o) - 1] + pn[o],         \
               c##3 = F.pnom_r[o], c##4 = pm[o] + pm[(o) + ni], c##5 = pm[(o) - ni] + pm[o]
#define PSET(c, o)                                                                                \
  const double c##0 = F.pmon_p[o], c##1 = pn[(o) - ni] + pn[o], c##2 = pn[(o) - 1 - ni] + pn[(o) - 1], \
               c##3 = F.pnom_p[o], c##4 = pm[(o) - 1] + pm[o], c##5 = pm[(o) - 1 - ni] + pm[(o) - ni]
  RSET(r0_, x); PSET(p0_, x);
  const long xw = do_u ? x - 1 : x, xs = do_v ? x - ni : x;
  RSET(rw_, xw); PSET(pn_, x + ni);
  RSET(rs_, xs); PSET(pe_, x + 1);
#undef RSET
#undef PSET
  const double fur1 = F.on_r[x] * F.on_r[x] * F.visc2_r[x], fur0 = F.on_r[x - 1] * F.on_r[x - 1] * F.visc2_r[x - 1];
  const double fup1 = F.om_p[x + ni] * F.om_p[x + ni] * F.visc2_p[x + ni], fup0 = F.om_p[x] * F.om_p[x] * F.visc2_p[x];
  const double fvp1 = F.on_p[x + 1] * F.on_p[x + 1] * F.visc2_p[x + 1], fvp0 = F.on_p[x] * F.on_p[x] * F.visc2_p[x];
  const double fvr1 = F.om_r[x] * F.om_r[x] * F.visc2_r[x], fvr0 = F.om_r[x - ni] * F.om_r[x - ni] * F.visc2_r[x - ni];
  const double ucff = G.dt * 0.25 * (pm[x - 1] + pm[x]) * (pn[x - 1] + pn[x]);
  const double uc1 = 0.5 * (pn[x - 1] + pn[x]), uc2 = 0.5 * (pm[x - 1] + pm[x]);
  const double vcff = G.dt * 0.25 * (pm[x] + pm[x - ni]) * (pn[x] + pn[x - ni]);
  const double vc1 = 0.5 * (pn[x - ni] + pn[x]), vc2 = 0.5 * (pm[x - ni] + pm[x]);

  // ---- vertical sums of ru, rv and the stress terms: rhs3d.F:1700-1918 as in k_rhs3d_sum
  double ruf = 0.0, rvf = 0.0;
  {
    const double *ru = F.ru + (size_t)(nrhs - 1) * nij * (size_t)(N + 1) + x;
    const double *rv = F.rv + (size_t)(nrhs - 1) * nij * (size_t)(N + 1) + x;
#define COLSUM(sum, A)                                                                     \
  do {                                                                                     \
    sum = 0.0;                                                                             \
    for (int k0 = 1; k0 <= N; k0 += 8) {                                                   \
      double r_[8];                                                                        \
      _Pragma("unroll") for (int q = 0; q < 8; q++) r_[q] = A[(size_t)KMIN(k0 + q, N) * nij];  \
      _Pragma("unroll") for (int q = 0; q < 8; q++)                                        \
        if (k0 + q <= N) sum = (k0 + q == 1) ? r_[q] : sum + r_[q];                        \
    }                                                                                      \
  } while (0)
    if (do_u) {
      double sum;
      COLSUM(sum, ru);
      const double cff = F.om_u[x] * F.on_u[x];
      const double c1 = F.sustr[x] * cff;
      const double c2 = -F.bustr[x] * cff;
      ruf = sum + c1 + c2;
    }
    if (do_v) {
      double sum;
      COLSUM(sum, rv);
      const double cff = F.om_v[x] * F.on_v[x];
      const double c1 = F.svstr[x] * cff;
      const double c2 = -F.bvstr[x] * cff;
      rvf = sum + c1 + c2;
    }
#undef COLSUM
  }

  // ---- march over the levels
  stage_store(lds_dyn_);
  __syncthreads();
  const int s = (ty + 1) * UC_TW + (tx + 1);
  double *un3 = F.u + (size_t)(nnew - 1) * nij * (size_t)N + x, *vn3 = F.v + (size_t)(nnew - 1) * nij * (size_t)N + x;
#define LHZ(di, dj) cur[0 * UC_NT + s + (di) + (dj) * UC_TW]
#define LU(di, dj) cur[1 * UC_NT + s + (di) + (dj) * UC_TW]
#define LV(di, dj) cur[2 * UC_NT + s + (di) + (dj) * UC_TW]
#define CFFR(c, di, dj)                                                                             \
  (LHZ(di, dj) * 0.5 * (c##0 * (c##1 * LU((di) + 1, dj) - c##2 * LU(di, dj)) - c##3 * (c##4 * LV(di, (dj) + 1) - c##5 * LV(di, dj))))
#define CFFP(c, di, dj)                                                                             \
  (0.125 * (LHZ((di) - 1, dj) + LHZ(di, dj) + LHZ((di) - 1, (dj) - 1) + LHZ(di, (dj) - 1)) *        \
   (c##0 * (c##1 * LV(di, dj) - c##2 * LV((di) - 1, dj)) + c##3 * (c##4 * LU(di, dj) - c##5 * LU(di, (dj) - 1))))
  for (int k = 1; k <= N; k++) {
    const double *cur = lds_dyn_ + ((k - 1) & 1) * (3 * UC_NT);
    double *nxt = lds_dyn_ + (k & 1) * (3 * UC_NT);
    if (k < N) stage_load(k + 1);
    if (inside) {
      const size_t ok = (size_t)(k - 1) * nij;
      const double cR = CFFR(r0_, 0, 0), cP = CFFP(p0_, 0, 0);
      if (do_u) {
        const double cRw = CFFR(rw_, -1, 0), cPn = CFFP(pn_, 0, 1);
        const double UFx1 = fur1 * cR;
        const double UFx0 = fur0 * cRw;
        const double UFe1 = fup1 * cPn;
        const double UFe0 = fup0 * cP;
        const double u1 = uc1 * (UFx1 - UFx0);
        const double u2 = uc2 * (UFe1 - UFe0);
        const double cff3 = ucff * (u1 + u2);
        un3[ok] = un3[ok] + cff3;
        ruf = ruf + u1 + u2;
      }
      if (do_v) {
        const double cRs = CFFR(rs_, 0, -1), cPe = CFFP(pe_, 1, 0);
        const double VFx1 = fvp1 * cPe;
        const double VFx0 = fvp0 * cP;
        const double VFe1 = fvr1 * cR;
        const double VFe0 = fvr0 * cRs;
        const double v1 = vc1 * (VFx1 - VFx0);
        const double v2 = vc2 * (VFe1 - VFe0);
        const double cff3 = vcff * (v1 - v2);
        vn3[ok] = vn3[ok] + cff3;
        rvf = rvf + v1 - v2;
      }
    }
    if (k < N) {
      stage_store(nxt);
      __syncthreads();
    }
  }
#undef LHZ
#undef LU
#undef LV
#undef CFFR
#undef CFFP
  if (do_u) F.rufrc[x] = ruf;
  if (do_v) F.rvfrc[x] = rvf;
}
