// k_duv.h -- DIAGNOSTICS_UV: the kernels of the per-term momentum tendencies that are not stores inside a kernel of the
// step (those: k_pre_new, k_rhs3d_pt_duv in k_rhs3d.h; k_step2d_duv in k_step2d.h; k_s3uv_couple in k_step3d.h):
//   k_duv_pgrd   DiaRU | DiaRV(i,j,k,nrhs,M3pgrd) = ru | rv(i,j,k,nrhs) behind prsgrd (prsgrd32.h:364, :428; 31, 40 alike)
//   k_duv_frc    the vertical sums DiaRUfrc | DiaRVfrc(i,j,3,:) of rhs3d.F:1712-1915 and uv3dmix2_s.h:303-326, and the
//                viscous terms of DiaU3wrk | DiaV3wrk, from the per-level terms k_uv3dmix2_s leaves in wrk3[6..9]
//   k_duv_s3uv   step3d_uv_tile's first J loop (step3d_uv.F:345-791 for u, :812-1258 for v) WITH its diagnostic
//                statements: time step of the r.h.s., implicit vertical viscosity (parabolic splines), replacement of the
//                vertical mean by the barotropic one -- one thread per column, the reference's loops as they stand.  It
//                runs INSTEAD of k_s3uv_col when the diagnostics are on and leaves the same u, v(nnew) (both are the
//                reference's operations in the reference's order)
//   k_duv_acc, k_duv_scale   the momentum part of set_diags_tile (set_diags.F:192-235, :319-360, :541-572)
// The arrays live in Fields::duv (roms_ctx.h: duv_*).  Pinned through the oracle (oracle/orc_diags_uv.c, bit for bit against
// the reference built from upwelling.h as shipped): tests/test_kernels_emu.py::test_momentum_diagnostics_bitwise.
#pragma once
#include "roms_ctx.h"
#include "k_diag3d.h"
#include "k_step3d.h"       // ROMS_NPRIV, emit_plan / emit_store

// grid (Iend-Istr+1, Jend-Jstr+1, N)
THREAD_KERNEL(k_duv_pgrd, KArgs) {
  const DGrid &G = a.G;
  const Fields &F = a.Fv;
  const TB &B = G.T;
  const int i = B.Istr + gx, j = B.Jstr + gy, k = gz + 1, nrhs = G.nrhs, N = G.N;
  const size_t o = (size_t)(nrhs - 1) * G.nij * (size_t)(N + 1);
  if (i >= B.IstrU) duv_r3(G, F, 0, nrhs, G.m3[M3PGRD])[X3(i, j, k)] = F.ru[o + XW(i, j, k)];
  if (j >= B.JstrV) duv_r3(G, F, 1, nrhs, G.m3[M3PGRD])[X3(i, j, k)] = F.rv[o + XW(i, j, k)];
}
THREAD_GLOBAL(k_duv_pgrd, KArgs)

// grid (Iend-Istr+1, Jend-Jstr+1, 2): one thread per column and direction
THREAD_KERNEL(k_duv_frc, KArgs) {
  const DGrid &G = a.G;
  const Fields &F = a.Fv;
  const TB &B = G.T;
  const int dir = gz, i = B.Istr + gx, j = B.Jstr + gy, N = G.N, nrhs = G.nrhs;
  if (dir == 0 ? (i < B.IstrU) : (j < B.JstrV)) return;
  const size_t x = X2(i, j), nij = (size_t)G.nij;
  const int m3[5] = {G.m3[M3PGRD], G.m3[M3FCOR], G.m3[M3XADV], G.m3[M3YADV], G.m3[M3HADV]};
  const int m2[5] = {G.m2[M2PGRD], G.m2[M2FCOR], G.m2[M2XADV], G.m2[M2YADV], G.m2[M2HADV]};
  for (int q = 0; q < 5; q++) {
    if (!m3[q]) continue;
    const double *R = duv_r3(G, F, dir, nrhs, m3[q]) + x;
    double s = R[0];
    for (int k = 2; k <= N; k++) s = s + R[(size_t)(k - 1) * nij];
    duv_rfrc(G, F, dir, 3, m2[q])[x] = s;
  }
  {                                                       // surface and bottom stress rhs3d.F:1801-1808, :1907-1914
    const double cff = dir == 0 ? F.om_u[x] * F.on_u[x] : F.om_v[x] * F.on_v[x];
    const double cff1 = (dir == 0 ? F.sustr : F.svstr)[x] * cff;
    const double cff2 = -(dir == 0 ? F.bustr : F.bvstr)[x] * cff;
    duv_rfrc(G, F, dir, 3, G.m2[M2SSTR])[x] = cff1;
    duv_rfrc(G, F, dir, 3, G.m2[M2BSTR])[x] = cff2;
  }
  if (G.m3[M3HVIS]) {                                     // uv3dmix2_s.h:296-327 from the terms of k_uv3dmix2_s (cff1, cff2 per level)
    const long ni = G.ni;
    const double *pm = F.pm, *pn = F.pn;
    const long xl = (long)x;
    const double cff = dir == 0 ? G.dt * 0.25 * (pm[xl - 1] + pm[xl]) * (pn[xl - 1] + pn[xl])
                                : G.dt * 0.25 * (pm[xl] + pm[xl - ni]) * (pn[xl] + pn[xl - ni]);
    const double *A1 = F.wrk3[dir == 0 ? 6 : 8] + x, *A2 = F.wrk3[dir == 0 ? 7 : 9] + x;
    double *Wh = duv_3wrk(G, F, dir, G.m3[M3HVIS]) + x, *Wx = duv_3wrk(G, F, dir, G.m3[M3XVIS]) + x, *Wy = duv_3wrk(G, F, dir, G.m3[M3YVIS]) + x;
    double sh = 0.0, sx = 0.0, sy = 0.0;
    for (int k = 1; k <= N; k++) {
      const size_t ok = (size_t)(k - 1) * nij;
      const double cff1 = A1[ok], cff2 = A2[ok];
      if (dir == 0) {
        const double cff3 = cff * (cff1 + cff2);
        sh = sh + cff1 + cff2; sx = sx + cff1; sy = sy + cff2;
        Wh[ok] = cff3; Wx[ok] = cff * cff1; Wy[ok] = cff * cff2;
      } else {
        const double cff3 = cff * (cff1 - cff2);
        sh = sh + cff1 - cff2; sx = sx + cff1; sy = sy - cff2;
        Wh[ok] = cff3; Wx[ok] = cff * cff1; Wy[ok] = -cff * cff2;
      }
    }
    duv_rfrc(G, F, dir, 3, G.m2[M2HVIS])[x] = sh;
    duv_rfrc(G, F, dir, 3, G.m2[M2XVIS])[x] = sx;
    duv_rfrc(G, F, dir, 3, G.m2[M2YVIS])[x] = sy;
  }
}
THREAD_GLOBAL(k_duv_frc, KArgs)

// grid (Iend-Istr+1, Jend-Jstr+1, 2)
THREAD_KERNEL(k_duv_s3uv, KArgs) {
  const DGrid &G = a.G;
  const Fields &F = a.Fv;
  const TB &B = G.T;
  const int dir = gz;
  const int i = (dir == 0 ? B.IstrU : B.Istr) + gx, j = (dir == 0 ? B.Jstr : B.JstrV) + gy;
  if (i > B.Iend || j > B.Jend) return;
  const int di = dir == 0 ? 1 : 0, dj = dir == 0 ? 0 : 1;
  const int N = G.N, nrhs = G.nrhs, nnew = G.nnew;
  const double dt = G.dt;
  const size_t nij = (size_t)G.nij, x = X2(i, j);
  double *q = (dir == 0 ? F.u : F.v) + (size_t)(nnew - 1) * nij * (size_t)N + x;
  const double *rq = (dir == 0 ? F.ru : F.rv) + (size_t)(nrhs - 1) * nij * (size_t)(N + 1) + x;
  double AK[ROMS_NPRIV], Hzk[ROMS_NPRIV], oHz[ROMS_NPRIV], CF[ROMS_NPRIV], DC[ROMS_NPRIV], qv[ROMS_NPRIV];
  AK[0] = 0.5 * (F.Akv[XW(i - di, j - dj, 0)] + F.Akv[XW(i, j, 0)]);
  for (int k = 1; k <= N; k++) {
    AK[k] = 0.5 * (F.Akv[XW(i - di, j - dj, k)] + F.Akv[XW(i, j, k)]);
    Hzk[k] = 0.5 * (F.Hz[X3(i - di, j - dj, k)] + F.Hz[X3(i, j, k)]);
    oHz[k] = 1.0 / Hzk[k];
  }
  double cff, cff1;
  if (G.iic == G.ntfirst) cff = 0.25 * dt;
  else if (G.iic == G.ntfirst + 1) cff = 0.25 * dt * 3.0 / 2.0;
  else cff = 0.25 * dt * 23.0 / 12.0;
  const double DC0 = cff * (F.pm[X2(i, j)] + F.pm[X2(i - di, j - dj)]) * (F.pn[X2(i, j)] + F.pn[X2(i - di, j - dj)]);
  const int Mp = G.m3[M3PGRD], Mvv = G.m3[M3VVIS], Mhv = G.m3[M3HVIS], Mxv = G.m3[M3XVIS], Myv = G.m3[M3YVIS], Mrt = G.m3[M3RATE];
#define W3(id, k) duv_3wrk(G, F, dir, id)[x + (size_t)((k) - 1) * nij]
  for (int k = 1; k <= N; k++) {                          // :358-385
    qv[k] = (q[(size_t)(k - 1) * nij] + DC0 * rq[(size_t)k * nij]) * oHz[k];
    for (int id = 1; id <= Mp; id++) W3(id, k) = (W3(id, k) + DC0 * duv_r3(G, F, dir, nrhs, id)[x + (size_t)(k - 1) * nij]) * oHz[k];
    if (Mhv) { W3(Mxv, k) = W3(Mxv, k) * oHz[k]; W3(Myv, k) = W3(Myv, k) * oHz[k]; W3(Mhv, k) = W3(Mhv, k) * oHz[k]; }
    W3(Mvv, k) = W3(Mvv, k) * oHz[k];
    W3(Mrt, k) = W3(Mrt, k) * oHz[k];
  }
  {                                                       // implicit vertical viscosity, parabolic splines :392-437
    double FC[ROMS_NPRIV], BC;
    cff1 = 1.0 / 6.0;
    for (int k = 1; k <= N - 1; k++) {
      FC[k] = cff1 * Hzk[k] - dt * AK[k - 1] * oHz[k];
      CF[k] = cff1 * Hzk[k + 1] - dt * AK[k + 1] * oHz[k + 1];
    }
    CF[0] = 0.0; DC[0] = 0.0;
    cff1 = 1.0 / 3.0;
    for (int k = 1; k <= N - 1; k++) {
      BC = cff1 * (Hzk[k] + Hzk[k + 1]) + dt * AK[k] * (oHz[k] + oHz[k + 1]);
      cff = 1.0 / (BC - FC[k] * CF[k - 1]);
      CF[k] = cff * CF[k];
      DC[k] = cff * (qv[k + 1] - qv[k] - FC[k] * DC[k - 1]);
    }
    DC[N] = 0.0;
    for (int k = N - 1; k >= 1; k--) DC[k] = DC[k] - CF[k] * DC[k + 1];
    for (int k = 1; k <= N; k++) {
      DC[k] = DC[k] * AK[k];
      cff = dt * oHz[k] * (DC[k] - DC[k - 1]);
      qv[k] = qv[k] + cff;
      W3(Mvv, k) = W3(Mvv, k) + cff;
    }
  }
  // vertical mean :594-708: CF(i,0), DC(i,0) and Dwrk(i,M2...) of the terms with a 2-D counterpart
  int m2[9], m3[9], nm = 0;
  m2[nm] = G.m2[M2PGRD]; m3[nm++] = Mp;
  m2[nm] = G.m2[M2BSTR]; m3[nm++] = Mvv;
  if (G.m3[M3FCOR]) { m2[nm] = G.m2[M2FCOR]; m3[nm++] = G.m3[M3FCOR]; }
  if (Mhv) { m2[nm] = G.m2[M2XVIS]; m3[nm++] = Mxv; m2[nm] = G.m2[M2YVIS]; m3[nm++] = Myv; m2[nm] = G.m2[M2HVIS]; m3[nm++] = Mhv; }
  if (G.m3[M3HADV]) { m2[nm] = G.m2[M2XADV]; m3[nm++] = G.m3[M3XADV]; m2[nm] = G.m2[M2YADV]; m3[nm++] = G.m3[M3YADV]; m2[nm] = G.m2[M2HADV]; m3[nm++] = G.m3[M3HADV]; }
  double Dwrk[12];
  for (int id = 0; id < 12; id++) Dwrk[id] = 0.0;
  double CF0 = Hzk[1], DCs = qv[1] * Hzk[1];
  for (int k = 2; k <= N; k++) { CF0 = CF0 + Hzk[k]; DCs = DCs + qv[k] * Hzk[k]; }
  for (int t = 0; t < nm; t++) {
    double s = W3(m3[t], 1) * Hzk[1];
    for (int k = 2; k <= N; k++) s = s + W3(m3[t], k) * Hzk[k];
    Dwrk[m2[t]] = s;
  }
  const double omn1 = (dir == 0 ? F.on_u : F.om_v)[x];
  const double Davg = (dir == 0 ? F.DU_avg1 : F.DV_avg1)[x];
  cff1 = 1.0 / (CF0 * omn1);
  const double corr = (DCs * omn1 - Davg) * cff1;
  for (int id = 1; id <= G.m2[M2PGRD]; id++) Dwrk[id] = (Dwrk[id] * omn1 - duv_2wrk(G, F, dir, id)[x]) * cff1;
  Dwrk[G.m2[M2BSTR]] = (Dwrk[G.m2[M2BSTR]] * omn1 - duv_2wrk(G, F, dir, G.m2[M2BSTR])[x] - duv_2wrk(G, F, dir, G.m2[M2SSTR])[x]) * cff1;
  const double qmask = G.masking ? (dir == 0 ? F.umask : F.vmask)[x] : 1.0;
  const EmitPlan PQ = emit_plan(G, dir == 0 ? BC_U : BC_V, i, j);
  for (int k = 1; k <= N; k++) {                          // couple and update :712-791
    double r = qv[k] - corr;
    if (G.masking) r = r * qmask;
    emit_store(G, PQ, q - x + (size_t)(k - 1) * nij, r);
    for (int t = 0; t < nm; t++) W3(m3[t], k) = W3(m3[t], k) - Dwrk[m2[t]];
    if (G.masking) for (int id = 1; id <= G.ndm3; id++) W3(id, k) = W3(id, k) * qmask;
  }
#undef W3
}
THREAD_GLOBAL(k_duv_s3uv, KArgs)

struct DuvArgs {
  DGrid G;
  Fields Fv;
  int init;
  double fac;
};
// the momentum part of set_diags_tile.  grid.z = plane of the wrk arrays: the 2 x NDM2d planes of DiaU2wrk | DiaV2wrk, then the
// 2 x NDM3d x N planes of DiaU3wrk | DiaV3wrk; U on (Istr:IendR, JstrR:JendR), V on (IstrR:IendR, Jstr:JendR)
THREAD_KERNEL(k_duv_acc, DuvArgs) {
  const DGrid &G = a.G;
  const Fields &F = a.Fv;
  const TB &B = G.T;
  const int i = B.IstrR + gx, j = B.JstrR + gy;
  const int n2 = 2 * G.ndm2;
  const size_t x = X2(i, j);
  const double *W;
  double *D;
  int dir;
  if (gz < n2) { dir = gz / G.ndm2; const int id = gz - dir * G.ndm2 + 1; W = duv_2wrk(G, F, dir, id) + x; D = duv_2d(G, F, dir, id) + x; }
  else {
    const int p = gz - n2, blk = p / G.N, k = p - blk * G.N;
    dir = blk / G.ndm3;
    const int id = blk - dir * G.ndm3 + 1;
    W = duv_3wrk(G, F, dir, id) + x + (size_t)k * G.nij; D = duv_3d(G, F, dir, id) + x + (size_t)k * G.nij;
  }
  if (dir == 0 ? (i < B.Istr) : (j < B.Jstr)) return;
  if (a.init == 2) *D = a.fac * *D;                       // conversion :541-572
  else *D = a.init ? *W : *D + *W;
}
THREAD_GLOBAL(k_duv_acc, DuvArgs)
