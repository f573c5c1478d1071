// k_mix4.h -- biharmonic horizontal mixing along s-surfaces: the harmonic operator applied twice.
//   k_t3dmix4     t3dmix4_s_tile    ROMS/Nonlinear/t3dmix4_s.h:94-478     (TS_DIF4 + MIX_S_TS)
//   k_uv4_lap     uv3dmix4_s_tile   ROMS/Nonlinear/uv3dmix4_s.h:296-524   first harmonic operator LapU, LapV with its closed /
//                                   gradient conditions and corner values (UV_VIS4 + MIX_S_UV)
//   the second operator (:526-622) is k_uv3dmix2_t's template form VIS4 (k_rhs3d.h): the same stress tensor of (LapU, LapV)
//   with visc4 in place of visc2, its terms stored NEGATED where uv3dmix2 stores its own -- the update of u, v(nnew) and the
//   sums into rufrc, rvfrc are then the harmonic path's, bit for bit (a - b == a + (-b))
// The barotropic part (step2d_LF_AM3.h:1653-1920) is k_step2d_vis4 in k_step2d.h.
// Coefficients: visc4_r, visc4_p, diff4 hold the square roots of VISC4 / TNU4 (inp_par.F:634).  Every thread evaluates the
// first operator where it needs it from the reference's expressions (a point's value is the same whoever forms it): the
// tracer kernel needs no work array and no second launch.
#pragma once
#include "roms_ctx.h"
#include "k_diag3d.h"

// ---- t3dmix4: first harmonic operator LapT(i,j) of level k as t3dmix4_s.h:248-349 forms it, then its conditions :354-408
KDEV double t4_lap_raw(const DGrid &G, const Fields &F, const double *d4, const double *T /* t(:,:,k,nrhs,itrc) */, const double *Hz /* level k */,
                       int i, int j) {
  const bool msk = G.masking != 0;
  double cff = 0.25 * (d4[X2(i, j)] + d4[X2(i - 1, j)]) * F.pmon_u[X2(i, j)];
  if (msk) cff = cff * F.umask[X2(i, j)];
  const double FX0 = cff * (Hz[X2(i, j)] + Hz[X2(i - 1, j)]) * (T[X2(i, j)] - T[X2(i - 1, j)]);
  cff = 0.25 * (d4[X2(i + 1, j)] + d4[X2(i, j)]) * F.pmon_u[X2(i + 1, j)];
  if (msk) cff = cff * F.umask[X2(i + 1, j)];
  const double FX1 = cff * (Hz[X2(i + 1, j)] + Hz[X2(i, j)]) * (T[X2(i + 1, j)] - T[X2(i, j)]);
  cff = 0.25 * (d4[X2(i, j)] + d4[X2(i, j - 1)]) * F.pnom_v[X2(i, j)];
  if (msk) cff = cff * F.vmask[X2(i, j)];
  const double FE0 = cff * (Hz[X2(i, j)] + Hz[X2(i, j - 1)]) * (T[X2(i, j)] - T[X2(i, j - 1)]);
  cff = 0.25 * (d4[X2(i, j + 1)] + d4[X2(i, j)]) * F.pnom_v[X2(i, j + 1)];
  if (msk) cff = cff * F.vmask[X2(i, j + 1)];
  const double FE1 = cff * (Hz[X2(i, j + 1)] + Hz[X2(i, j)]) * (T[X2(i, j + 1)] - T[X2(i, j)]);
  cff = 1.0 / Hz[X2(i, j)];
  return F.pm[X2(i, j)] * F.pn[X2(i, j)] * cff * (FX1 - FX0 + FE1 - FE0);
}
KDEV double t4_lap(const DGrid &G, const Fields &F, const double *d4, const double *T, const double *Hz, int i, int j, int clo /* bit e: edge e closed */) {
  const TB &B = G.T;
  if (!G.ewp) {
    if (B.west && i == B.Istr - 1) return (clo & (1 << ROMS_IWEST)) ? 0.0 : t4_lap_raw(G, F, d4, T, Hz, B.Istr, j);
    if (B.east && i == B.Iend + 1) return (clo & (1 << ROMS_IEAST)) ? 0.0 : t4_lap_raw(G, F, d4, T, Hz, B.Iend, j);
  }
  if (!G.nsp) {
    if (B.south && j == B.Jstr - 1) return (clo & (1 << ROMS_ISOUTH)) ? 0.0 : t4_lap_raw(G, F, d4, T, Hz, i, B.Jstr);
    if (B.north && j == B.Jend + 1) return (clo & (1 << ROMS_INORTH)) ? 0.0 : t4_lap_raw(G, F, d4, T, Hz, i, B.Jend);
  }
  return t4_lap_raw(G, F, d4, T, Hz, i, j);
}
// grid (Iend-Istr+1, Jend-Jstr+1, N*NT)
THREAD_KERNEL(k_t3dmix4, KArgs) {
  const DGrid &G = a.G;
  const Fields &F = a.Fv;
  const TB &B = G.T;
  const int N = G.N, k = gz % N + 1, itrc = gz / N + 1;
  const int i = B.Istr + gx, j = B.Jstr + gy;
  const bool msk = G.masking != 0;
  const double *d4 = F.diff4 + (size_t)(itrc - 1) * G.nij;
  const double *T = F.t + XT(G.LBi, G.LBj, k, G.nrhs, itrc), *Hz = F.Hz + X3(G.LBi, G.LBj, k);
  // LBC(edge,isTvar(itrc))%closed: everything but an open kind (lbc_closed holds the closed edges of every state variable)
  const int clo = (int)((G.lbc_closed >> (4 * (ROMS_ISTVAR + itrc - 1))) & 15ull);
  const double L0 = t4_lap(G, F, d4, T, Hz, i, j, clo), Lw = t4_lap(G, F, d4, T, Hz, i - 1, j, clo), Le = t4_lap(G, F, d4, T, Hz, i + 1, j, clo),
               Ls = t4_lap(G, F, d4, T, Hz, i, j - 1, clo), Ln = t4_lap(G, F, d4, T, Hz, i, j + 1, clo);
  double cff = 0.25 * (d4[X2(i, j)] + d4[X2(i - 1, j)]) * F.pmon_u[X2(i, j)];                                   // :413-434
  double FX0 = cff * (Hz[X2(i, j)] + Hz[X2(i - 1, j)]) * (L0 - Lw);
  if (msk) FX0 = FX0 * F.umask[X2(i, j)];
  cff = 0.25 * (d4[X2(i + 1, j)] + d4[X2(i, j)]) * F.pmon_u[X2(i + 1, j)];
  double FX1 = cff * (Hz[X2(i + 1, j)] + Hz[X2(i, j)]) * (Le - L0);
  if (msk) FX1 = FX1 * F.umask[X2(i + 1, j)];
  cff = 0.25 * (d4[X2(i, j)] + d4[X2(i, j - 1)]) * F.pnom_v[X2(i, j)];
  double FE0 = cff * (Hz[X2(i, j)] + Hz[X2(i, j - 1)]) * (L0 - Ls);
  if (msk) FE0 = FE0 * F.vmask[X2(i, j)];
  cff = 0.25 * (d4[X2(i, j + 1)] + d4[X2(i, j)]) * F.pnom_v[X2(i, j + 1)];
  double FE1 = cff * (Hz[X2(i, j + 1)] + Hz[X2(i, j)]) * (Ln - L0);
  if (msk) FE1 = FE1 * F.vmask[X2(i, j + 1)];
  cff = G.dt * F.pm[X2(i, j)] * F.pn[X2(i, j)];                                                                   // :461-467
  const double cff1 = cff * (FX1 - FX0), cff2 = cff * (FE1 - FE0);
  const double cff3 = cff1 + cff2;
  double *tn = F.t + XT(i, j, k, G.nnew, itrc);
  *tn = *tn - cff3;
}
THREAD_GLOBAL(k_t3dmix4, KArgs)

// ---- uv3dmix4, first harmonic operator.  Stresses of (u,v)(nrhs) WITHOUT the thickness (:296-331):
KDEV double uv4_sr(const DGrid &G, const Fields &F, const double *u, const double *v, int i, int j) {   // rho point: cff of :299-305 times visc4_r
  const double *pm = F.pm, *pn = F.pn;
  return 0.5 * (F.pmon_r[X2(i, j)] * ((pn[X2(i, j)] + pn[X2(i + 1, j)]) * u[X2(i + 1, j)] - (pn[X2(i - 1, j)] + pn[X2(i, j)]) * u[X2(i, j)]) -
                F.pnom_r[X2(i, j)] * ((pm[X2(i, j)] + pm[X2(i, j + 1)]) * v[X2(i, j + 1)] - (pm[X2(i, j - 1)] + pm[X2(i, j)]) * v[X2(i, j)]));
}
KDEV double uv4_sp(const DGrid &G, const Fields &F, const double *u, const double *v, int i, int j) {   // psi point :312-318 (+ mask)
  const double *pm = F.pm, *pn = F.pn;
  double cff = 0.5 * (F.pmon_p[X2(i, j)] * ((pn[X2(i, j - 1)] + pn[X2(i, j)]) * v[X2(i, j)] - (pn[X2(i - 1, j - 1)] + pn[X2(i - 1, j)]) * v[X2(i - 1, j)]) +
                      F.pnom_p[X2(i, j)] * ((pm[X2(i - 1, j)] + pm[X2(i, j)]) * u[X2(i, j)] - (pm[X2(i - 1, j - 1)] + pm[X2(i, j - 1)]) * u[X2(i, j - 1)]));
  if (G.masking) cff = cff * F.pmask[X2(i, j)];
  return cff;
}
KDEV double uv4_lapu_raw(const DGrid &G, const Fields &F, const double *u, const double *v, int i, int j) {
  const double *pm = F.pm, *pn = F.pn;
  const double UFx1 = F.on_r[X2(i, j)] * F.on_r[X2(i, j)] * F.visc4_r[X2(i, j)] * uv4_sr(G, F, u, v, i, j);
  const double UFx0 = F.on_r[X2(i - 1, j)] * F.on_r[X2(i - 1, j)] * F.visc4_r[X2(i - 1, j)] * uv4_sr(G, F, u, v, i - 1, j);
  const double UFe1 = F.om_p[X2(i, j + 1)] * F.om_p[X2(i, j + 1)] * F.visc4_p[X2(i, j + 1)] * uv4_sp(G, F, u, v, i, j + 1);
  const double UFe0 = F.om_p[X2(i, j)] * F.om_p[X2(i, j)] * F.visc4_p[X2(i, j)] * uv4_sp(G, F, u, v, i, j);
  return 0.125 * (pm[X2(i - 1, j)] + pm[X2(i, j)]) * (pn[X2(i - 1, j)] + pn[X2(i, j)]) *
         ((pn[X2(i - 1, j)] + pn[X2(i, j)]) * (UFx1 - UFx0) + (pm[X2(i - 1, j)] + pm[X2(i, j)]) * (UFe1 - UFe0));
}
KDEV double uv4_lapv_raw(const DGrid &G, const Fields &F, const double *u, const double *v, int i, int j) {
  const double *pm = F.pm, *pn = F.pn;
  const double VFx1 = F.on_p[X2(i + 1, j)] * F.on_p[X2(i + 1, j)] * F.visc4_p[X2(i + 1, j)] * uv4_sp(G, F, u, v, i + 1, j);
  const double VFx0 = F.on_p[X2(i, j)] * F.on_p[X2(i, j)] * F.visc4_p[X2(i, j)] * uv4_sp(G, F, u, v, i, j);
  const double VFe1 = F.om_r[X2(i, j)] * F.om_r[X2(i, j)] * F.visc4_r[X2(i, j)] * uv4_sr(G, F, u, v, i, j);
  const double VFe0 = F.om_r[X2(i, j - 1)] * F.om_r[X2(i, j - 1)] * F.visc4_r[X2(i, j - 1)] * uv4_sr(G, F, u, v, i, j - 1);
  return 0.125 * (pm[X2(i, j)] + pm[X2(i, j - 1)]) * (pn[X2(i, j)] + pn[X2(i, j - 1)]) *
         ((pn[X2(i, j - 1)] + pn[X2(i, j)]) * (VFx1 - VFx0) - (pm[X2(i, j - 1)] + pm[X2(i, j)]) * (VFe1 - VFe0));
}
// one-edge conditions :335-470 (closed: LBC(edge,isUvel | isVvel)%closed), then the corner values :472-524
KDEV double uv4_lapu_edge(const DGrid &G, const Fields &F, const double *u, const double *v, int i, int j, int clo) {
  const TB &B = G.T;
  if (!G.ewp) {
    if (B.west && i == B.Istr) return (clo & (1 << ROMS_IWEST)) ? 0.0 : uv4_lapu_raw(G, F, u, v, B.Istr + 1, j);
    if (B.east && i == B.Iend + 1) return (clo & (1 << ROMS_IEAST)) ? 0.0 : uv4_lapu_raw(G, F, u, v, B.Iend, j);
  }
  if (!G.nsp) {
    if (B.south && j == B.Jstr - 1) return (clo & (1 << ROMS_ISOUTH)) ? G.gamma2 * uv4_lapu_raw(G, F, u, v, i, B.Jstr) : 0.0;
    if (B.north && j == B.Jend + 1) return (clo & (1 << ROMS_INORTH)) ? G.gamma2 * uv4_lapu_raw(G, F, u, v, i, B.Jend) : 0.0;
  }
  return uv4_lapu_raw(G, F, u, v, i, j);
}
KDEV double uv4_lapv_edge(const DGrid &G, const Fields &F, const double *u, const double *v, int i, int j, int clo) {
  const TB &B = G.T;
  if (!G.nsp) {
    if (B.south && j == B.Jstr) return (clo & (1 << ROMS_ISOUTH)) ? 0.0 : uv4_lapv_raw(G, F, u, v, i, B.Jstr + 1);
    if (B.north && j == B.Jend + 1) return (clo & (1 << ROMS_INORTH)) ? 0.0 : uv4_lapv_raw(G, F, u, v, i, B.Jend);
  }
  if (!G.ewp) {
    if (B.west && i == B.Istr - 1) return (clo & (1 << ROMS_IWEST)) ? G.gamma2 * uv4_lapv_raw(G, F, u, v, B.Istr, j) : 0.0;
    if (B.east && i == B.Iend + 1) return (clo & (1 << ROMS_IEAST)) ? G.gamma2 * uv4_lapv_raw(G, F, u, v, B.Iend, j) : 0.0;
  }
  return uv4_lapv_raw(G, F, u, v, i, j);
}
KDEV double uv4_lapu(const DGrid &G, const Fields &F, const double *u, const double *v, int i, int j, int clo) {
  const TB &B = G.T;
  if (!(G.ewp || G.nsp)) {
    const bool wi = i == B.Istr, ei = i == B.Iend + 1, sj = j == B.Jstr - 1, nj = j == B.Jend + 1;
    if (B.sw && wi && sj) return 0.5 * (uv4_lapu_edge(G, F, u, v, i + 1, j, clo) + uv4_lapu_edge(G, F, u, v, i, j + 1, clo));
    if (B.se && ei && sj) return 0.5 * (uv4_lapu_edge(G, F, u, v, i - 1, j, clo) + uv4_lapu_edge(G, F, u, v, i, j + 1, clo));
    if (B.nw && wi && nj) return 0.5 * (uv4_lapu_edge(G, F, u, v, i + 1, j, clo) + uv4_lapu_edge(G, F, u, v, i, j - 1, clo));
    if (B.ne && ei && nj) return 0.5 * (uv4_lapu_edge(G, F, u, v, i - 1, j, clo) + uv4_lapu_edge(G, F, u, v, i, j - 1, clo));
  }
  return uv4_lapu_edge(G, F, u, v, i, j, clo);
}
KDEV double uv4_lapv(const DGrid &G, const Fields &F, const double *u, const double *v, int i, int j, int clo) {
  const TB &B = G.T;
  if (!(G.ewp || G.nsp)) {
    const bool wi = i == B.Istr - 1, ei = i == B.Iend + 1, sj = j == B.Jstr, nj = j == B.Jend + 1;
    if (B.sw && wi && sj) return 0.5 * (uv4_lapv_edge(G, F, u, v, i, j + 1, clo) + uv4_lapv_edge(G, F, u, v, i + 1, j, clo));
    if (B.se && ei && sj) return 0.5 * (uv4_lapv_edge(G, F, u, v, i - 1, j, clo) + uv4_lapv_edge(G, F, u, v, i, j + 1, clo));
    if (B.nw && wi && nj) return 0.5 * (uv4_lapv_edge(G, F, u, v, i + 1, j, clo) + uv4_lapv_edge(G, F, u, v, i, j - 1, clo));
    if (B.ne && ei && nj) return 0.5 * (uv4_lapv_edge(G, F, u, v, i, j - 1, clo) + uv4_lapv_edge(G, F, u, v, i - 1, j, clo));
  }
  return uv4_lapv_edge(G, F, u, v, i, j, clo);
}
// grid (Iend+1-(Istr-1)+1, Jend+1-(Jstr-1)+1, N): LapU, LapV -> lap4 wherever the second operator reads them:
// LapU on (IstrU-1:Iend+1, Jstr-1:Jend+1), LapV on (Istr-1:Iend+1, JstrV-1:Jend+1)
THREAD_KERNEL(k_uv4_lap, KArgs) {
  const DGrid &G = a.G;
  const Fields &F = a.Fv;
  const TB &B = G.T;
  const int N = G.N, k = gz + 1;
  const int i = B.Istr - 1 + gx, j = B.Jstr - 1 + gy;
  const double *u = F.u + X3(G.LBi, G.LBj, k) + (size_t)(G.nrhs - 1) * G.nij * (size_t)N;
  const double *v = F.v + X3(G.LBi, G.LBj, k) + (size_t)(G.nrhs - 1) * G.nij * (size_t)N;
  double *LU = (double *)F.lap4 + X3(G.LBi, G.LBj, k), *LV = LU + (size_t)N * G.nij;
  const int clu = (int)((G.lbc_closed >> (4 * ROMS_ISUVEL)) & 15ull), clv = (int)((G.lbc_closed >> (4 * ROMS_ISVVEL)) & 15ull);
  if (i >= B.IstrU - 1) LU[X2(i, j)] = uv4_lapu(G, F, u, v, i, j, clu);
  if (j >= B.JstrV - 1) LV[X2(i, j)] = uv4_lapv(G, F, u, v, i, j, clv);
}
THREAD_GLOBAL(k_uv4_lap, KArgs)
