// g_rhs3d.cpp -- launch sequences of pre_step3d, prsgrd, t3dmix2, rhs3d_tile, uv3dmix2.
#include "roms_host.h"
#include <cstdlib>
#include "k_rhs3d.h"
#include "k_mix4.h"
#include "k_prs4x.h"
#ifndef ROMS_CPU_EMU
#include "k_rhs3d_lds.h"
#include "k_tadv_lds.h"
#include "k_uv3dmix2_col.h"
#endif
#include "k_uvmix_geo.h"

static inline KArgs mk(roms_hip_ctx *c, int p0 = 0, int p1 = 0, int p2 = 0) {
  KArgs a;
  a.G = c->G;
  a.Fv = c->F;
  a.p0 = p0; a.p1 = p1; a.p2 = p2;
  return a;
}
static inline size_t lds_sz(const DGrid &G) { return (size_t)(G.bw + 6) * (size_t)(G.bh + 6); }

int run_swdk(roms_hip_ctx *c);            // g_lmd.cpp: solar penetration fractions into wrk3[5]
int run_t3dmix2_geo(roms_hip_ctx *c);     // g_geo.cpp

// The LDS-tiled form of the two tracer-advection point kernels (k_tadv_lds.h): mode 0 = k_pre_t3, 1 = k_s3t_hv.
// Applies when every tracer takes the point path of that kernel and NT <= 2; false: the caller launches the
// point-wise form (ROMS_HIP_TADV_LDS=0 forces that; the CPU emulation has it only).
bool launch_tadv_lds(roms_hip_ctx *c, int mode) {
#ifdef ROMS_CPU_EMU
  (void)c; (void)mode;
  return false;
#else
  const DGrid &G = c->G;
  const TB &B = G.T;
  static const char *el = getenv("ROMS_HIP_TADV_LDS"), *ek = getenv("ROMS_HIP_TADV_KC");
  if (G.NT > TL_MAXT) return false;
  if (G.dia_ts) return false;                 // DIAGNOSTICS_TS: the point kernels carry the DiaTwrk stores
  const int nx = B.Iend - B.Istr + 1, ny = B.Jend - B.Jstr + 1;
  if (el ? el[0] == '0' : (long)nx * ny < 64L * 1024L) return false;
  // mode 0: k_pre_t3's tracers are all those without a spline vertical flux (MPDATA/HSIMT tracers take the
  // first-order upstream predictor there); mode 1: the tracers of k_s3t_hv (every horizontal scheme but MPDATA and
  // HSIMT) -- the others are left to k_mpdata.h / k_s3t_h and skipped here (a.p1 = mask of the block's tracers)
  // mode 1, round 4: tracers with an HSIMT horizontal and / or vertical step ride along (template HS: 1/Hz, Huon, Hvom staged
  // too; k_s3t_h and the HSIMT sweep of k_s3t_col are then not needed for them) -- not with MASKING (k_s3t_h carries
  // the masked forms); ROMS_HIP_HSIMT_LDS=0 keeps the round-3 kernels
  static const char *ehs = getenv("ROMS_HIP_HSIMT_LDS");
  bool any_hs = false;
  for (int it = 0; it < G.NT; it++) any_hs |= G.hadv[it] == ROMS_HSIMT || G.vadv[it] == ROMS_HSIMT;
  const bool HS = mode == 1 && any_hs && !G.masking && !(ehs && ehs[0] == '0');
  int mask = 0;
  c->tadv_hdone = 0; c->tadv_vdone = 0;
  for (int it = 0; it < G.NT; it++) {
    const int hs = G.hadv[it], vs = G.vadv[it];
    if (mode == 0 && vs == ROMS_SPLINES) return false;
    if (mode == 0 || (hs != ROMS_MPDATA && (hs != ROMS_HSIMT || HS))) mask |= 1 << it;
    if (HS && ((mask >> it) & 1)) {
      if (hs == ROMS_HSIMT) c->tadv_hdone |= 1 << it;
      // the kernel's vertical step covers every local scheme (HSIMT included); what k_s3t_col would otherwise do for a
      // tracer off the point path
      if ((hs == ROMS_HSIMT || vs == ROMS_HSIMT) && vs != ROMS_MPDATA && vs != ROMS_SPLINES) c->tadv_vdone |= 1 << it;
    }
  }
  if (!mask) return false;
  KArgs a = mk(c);
  a.p1 = mask;
  const int nt = ((nx + TL_BX - 1) / TL_BX) * ((ny + TL_BY - 1) / TL_BY);
  int nz = KMAX(1, (2048 + nt - 1) / nt);
  int kc = KMAX(5, (G.N + nz - 1) / nz);
  if (ek && atoi(ek) > 0) kc = atoi(ek);
  kc = KMIN(kc, G.N);
  nz = (G.N + kc - 1) / kc;
  a.p0 = kc;
  const dim3 grid((unsigned)(8 * ((nt + 7) / 8) * nz), 1, 1), block(TL_BX, TL_BY, 1);
  const size_t lds = (size_t)(HS ? TL_LDS_DOUBLES_HS : TL_LDS_DOUBLES) * sizeof(double);
  static const char *ew = getenv("ROMS_HIP_TADV_W");
  const int w = ew ? atoi(ew) : ((mode == 0 || HS) ? 2 : 3);      // (HS: 256 VGPRs at two waves per SIMD, 22 spilled; three waves spill 123)
  if (HS) {
    if (w == 2) KPROF_WRAP(k_s3t_hv, c->stream, ROMS_LAUNCH((k_tadv_lds<1, 2, true>), grid, block, lds, c->stream, a, nx, ny, nz));
    else KPROF_WRAP(k_s3t_hv, c->stream, ROMS_LAUNCH((k_tadv_lds<1, 3, true>), grid, block, lds, c->stream, a, nx, ny, nz));
  }
  else if (mode == 0 && w == 2) KPROF_WRAP(k_pre_t3, c->stream, ROMS_LAUNCH((k_tadv_lds<0, 2>), grid, block, lds, c->stream, a, nx, ny, nz));
  else if (mode == 0) KPROF_WRAP(k_pre_t3, c->stream, ROMS_LAUNCH((k_tadv_lds<0, 3>), grid, block, lds, c->stream, a, nx, ny, nz));
  else if (w == 2) KPROF_WRAP(k_s3t_hv, c->stream, ROMS_LAUNCH((k_tadv_lds<1, 2>), grid, block, lds, c->stream, a, nx, ny, nz));
  else KPROF_WRAP(k_s3t_hv, c->stream, ROMS_LAUNCH((k_tadv_lds<1, 3>), grid, block, lds, c->stream, a, nx, ny, nz));
  return true;
#endif
}

// The tracer predictor of pre_step3d (pre_step3d.F:357-852: t(3) from t(nstp), t(nnew), Hz, Huon, Hvom, W) is
// independent of the surface forcing and the vertical mixing; on small grids the fused main3d sequence
// launches it early, on the side stream behind omega, beside the bulk-flux / KPP chain (c->pre_t3_ready).
int run_pre_t3(roms_hip_ctx *c) {
  const DGrid &G = c->G;
  const TB &B = G.T;
  KArgs a = mk(c);
  bool any_col = false;      // tracers with a spline vertical flux keep the two-kernel column path
  for (int it = 0; it < G.NT; it++) any_col |= G.vadv[it] == ROMS_SPLINES;
  auto point_part = [&]() {
    if (!launch_tadv_lds(c, 0)) {
      KArgs b = mk(c);
      b.p0 = (G.N + KCH - 1) / KCH;
      LAUNCH_THREAD(k_pre_t3, B.Iend - B.Istr + 1, B.Jend - B.Jstr + 1, b.p0 * G.NT, c->stream, b);
    }
  };
  HaloSpec sp[ROMS_MAXT];
  for (int it = 1; it <= G.NT; it++) sp[it - 1] = {t_lev(c, 3, it), G.N, obc_bc(c, bc_rstate(c)), 'r'};   // t3dbc + exchange :1157-1171
  if (c->rim_split && !any_col && !G.obc && !G.fuse3d) {
    // multi-tile, round 4: the points the strip exchange packs first, the exchange on its own stream behind them, the
    // interior beside it (nothing reads t(3) before step3d_t: FG_T3)
    c->G.region = 1; point_part();
    c->G.region = 0; launch_halo_tail(c, sp, G.NT);
    c->G.region = 2; point_part();
    c->G.region = 0;
    return 0;
  }
  point_part();
  if (any_col) {
    a.p0 = (G.N + KCH - 1) / KCH;
    LAUNCH_COOP(k_pre_t3h, G.nbx, G.nby, G.N * G.NT, 256, 3 * lds_sz(G), c->stream, a);
    LAUNCH_THREAD(k_pre_t3v, B.Iend - B.Istr + 1, B.Jend - B.Jstr + 1, G.NT, c->stream, a);
  }
  if (G.fuse3d && !any_col) return 0;   // k_pre_t3 stored the boundary values and images (pt_emit)
  if (G.obc) for (int it = 1; it <= G.NT; it++) { int r = run_obc3d_t(c, 3, it); if (r) return r; }
  launch_halo_multi(c, sp, G.NT);
  return 0;
}

int run_pre_step3d(roms_hip_ctx *c) {
  const DGrid &G = c->G;
  const TB &B = G.T;
  if ((G.options & ROMS_SOLAR_SOURCE) && !c->swdk_ready) { int r = run_swdk(c); if (r) return r; }
  if (!c->pre_t3_ready) { int r = run_pre_t3(c); if (r) return r; }     // (k_pre_new overwrites the t(nnew) it reads)
  KArgs a = mk(c);
  const bool fold_uv = c->late_pre && c->fold_uvmix && (G.options & ROMS_UV_VIS2);
  a.p1 = (c->late_pre ? 1 : 0) | (fold_uv ? 2 : 0) | (c->tmix_ready ? 4 : 0);
  // large grids: the marching form (every level read once; the chunked form re-reads two levels per chunk of five)
  static const char *epm = getenv("ROMS_HIP_PRENEW_MARCH");
  const long cols = (long)(B.Iend - B.Istr + 1) * (B.Jend - B.Jstr + 1);
  if (!G.dia_ts && !G.dia_uv && (epm ? epm[0] != '0' : cols >= 128L * 1024L)) {      // (DIAGNOSTICS_TS | _UV: the chunked form stores the terms)
    static const char *epp = getenv("ROMS_HIP_PRENEW_PARTS");       // (test aid: parts of the column per thread)
    const int parts = epp ? KMAX(1, atoi(epp)) : 1;
    a.p2 = (G.N + parts - 1) / parts;
    a.p2 = (a.p2 + KCH - 1) / KCH * KCH;                     // whole groups
    if (G.NT <= 2) LAUNCH_THREAD_AS(k_pre_new, k_pre_new_m, B.Iend - B.Istr + 1, B.Jend - B.Jstr + 1, (G.N + a.p2 - 1) / a.p2, c->stream, a);
    else LAUNCH_THREAD_AS(k_pre_new, k_pre_new_m4, B.Iend - B.Istr + 1, B.Jend - B.Jstr + 1, (G.N + a.p2 - 1) / a.p2, c->stream, a);
  } else
  LAUNCH_THREAD(k_pre_new, B.Iend - B.Istr + 1, B.Jend - B.Jstr + 1, (G.N + KCH - 1) / KCH, c->stream, a);
  if (c->late_pre && (G.options & ROMS_UV_VIS2) && !fold_uv)     // the update of u,v(nnew) k_uv3dmix2_s left out
    LAUNCH_THREAD(k_uv3dmix2_apply, B.Iend - B.Istr + 1, B.Jend - B.Jstr + 1, (G.N + KCH - 1) / KCH, c->stream, a);
  return 0;
}

int run_prsgrd(roms_hip_ctx *c) {
  const DGrid &G = c->G;
  const TB &B = G.T;
  KArgs a = mk(c);
  if (G.prs4x) {     // PJ_GRADPQ4, PJ_GRADPQ2 (prsgrd.F:16-19: in front of every other scheme)
    const bool q4 = G.prs4x == 44;
    a.p1 = q4 ? 44 : 42;
    const int e = q4 ? 0 : 1;          // prsgrd42 reconstructs one more column each way (:238: JstrV-2:Jend+1, IstrU-2:Iend+1)
    LAUNCH_THREAD(k_prs4x_col, (B.Iend + e) - (B.IstrU - 1 - e) + 1, (B.Jend + e) - (B.JstrV - 1 - e) + 1, 1, c->stream, a);
    if (q4) {
      LAUNCH_THREAD(k_prs44_grad, B.Iend - B.Istr + 1, B.Jend - B.Jstr + 1, 2, c->stream, a);
      if (G.wet_dry) { int r = run_wd_scale3(c); if (r) return r; }     // WET_DRY: ru, rv times the wet masks (prsgrd44.h:466,500)
    } else {
      LAUNCH_THREAD(k_prs42_grad1, (B.Iend + 1) - (B.IstrU - 1) + 1, (B.Jend + 1) - (B.JstrV - 1) + 1, 2, c->stream, a);
      LAUNCH_THREAD(k_prs42_grad2, B.Iend - B.Istr + 1, B.Jend - B.Jstr + 1, 2, c->stream, a);
    }
    return run_duv_pgrd(c);
  }
  if (G.options & ROMS_PRSGRD40) {     // PJ_GRADP: prsgrd40.h
    LAUNCH_THREAD(k_prs40, B.Iend - B.Istr + 1, B.Jend - B.Jstr + 1, 2, c->stream, a);
    if (G.wet_dry) { int r = run_wd_scale3(c); if (r) return r; }       // WET_DRY: ru, rv times the wet masks (prsgrd40.h:251,281)
    return run_duv_pgrd(c);
  }
  if (G.options & ROMS_PRSGRD31) {     // no DJ_GRADPS: prsgrd31.h (the reference order of main3d: roms_hip.cpp keeps the late-predictor schedule off)
    LAUNCH_THREAD(k_prs31, B.Iend - B.Istr + 1, B.Jend - B.Jstr + 1, 2, c->stream, a);
    if (G.wet_dry) { int r = run_wd_scale3(c); if (r) return r; }       // WET_DRY: ... (prsgrd31.h:239,285,323,369)
    return run_duv_pgrd(c);
  }
  a.p1 = c->late_pre ? 1 : 0;
  LAUNCH_THREAD(k_prs_P, B.Iend - (B.IstrU - 1) + 1, B.Jend - (B.JstrV - 1) + 1, 1, c->stream, a);
  {
    static const char *ept = getenv("ROMS_HIP_PRS_TILE");   // (experiment: 0 = 64 x 4, 1 = 32 x 8, 2 = 16 x 16 points per block)
    const int shape = ept ? atoi(ept) : 0;
    if (shape == 1) LAUNCH_THREAD_S(k_prs_grad, k_prs_grad, 32, 8, B.Iend - B.Istr + 1, B.Jend - B.Jstr + 1, (G.N + KCH - 1) / KCH, c->stream, a);
    else if (shape == 2) LAUNCH_THREAD_S(k_prs_grad, k_prs_grad, 16, 16, B.Iend - B.Istr + 1, B.Jend - B.Jstr + 1, (G.N + KCH - 1) / KCH, c->stream, a);
    else LAUNCH_THREAD(k_prs_grad, B.Iend - B.Istr + 1, B.Jend - B.Jstr + 1, (G.N + KCH - 1) / KCH, c->stream, a);
  }
  if (G.wet_dry) { int r = run_wd_scale3(c); if (r) return r; }     // WET_DRY: ru, rv times the wet masks (prsgrd32.h:362,426)
  return run_duv_pgrd(c);                          // DIAGNOSTICS_UV: DiaRU(M3pgrd) = ru as prsgrd leaves it
}

// biharmonic operators (k_mix4.h): t3dmix4 behind t3dmix2 (rhs3d.F:107-115); uv3dmix4 in uv3dmix2's place (:181-189; with
// UV_VIS4 the harmonic coefficients are zero and its kernel is skipped: its terms would be exact zeros)
static void launch_t3dmix4(roms_hip_ctx *c) {
  const DGrid &G = c->G;
  const TB &B = G.T;
  KArgs a = mk(c);
  LAUNCH_THREAD(k_t3dmix4, B.Iend - B.Istr + 1, B.Jend - B.Jstr + 1, G.N * G.NT, c->stream, a);
}
static void launch_uv3dmix4(roms_hip_ctx *c, int defer) {
  const DGrid &G = c->G;
  const TB &B = G.T;
  KArgs a = mk(c);
  LAUNCH_THREAD(k_uv4_lap, B.Iend - B.Istr + 3, B.Jend - B.Jstr + 3, G.N, c->stream, a);
  a.p1 = defer;
  LAUNCH_THREAD_AS(k_uv3dmix2_s, k_uv3dmix4_s, B.Iend - B.Istr + 1, B.Jend - B.Jstr + 1, (G.N + KCH - 1) / KCH, c->stream, a);
}
int run_t3dmix2(roms_hip_ctx *c) {
  const DGrid &G = c->G;
  const TB &B = G.T;
  if (!(G.options & ROMS_TS_DIF2)) return 0;
  if (c->tmix_ready) return 0;                         // (done ahead of pre_step3d as terms: k_pre_new added them)
  if (G.ts_dif4 && (G.options & (ROMS_MIX_GEO_TS | ROMS_MIX_ISO_TS))) return run_t3dmix2_geo(c);     // t3dmix4_geo.h, t3dmix4_iso.h: the marching kernel, twice
  if (G.ts_dif4) { launch_t3dmix4(c); return 0; }      // (TS_DIF4: diff2 is zero, t3dmix2 would add exact zeros)
  if (G.options & (ROMS_MIX_GEO_TS | ROMS_MIX_ISO_TS)) return run_t3dmix2_geo(c);     // (the isopycnic form: the same marching kernel on pden)
  KArgs a = mk(c);
  a.p2 = c->tmix_terms ? 1 : 0;
  static const char *et = getenv("ROMS_HIP_T3CH");
  // large grids: a thread loops over the column (512x512x50: KCH 228, 10: 216, 25: 211, 50: 210 us)
  a.p1 = et ? atoi(et) : ((long)(B.Iend - B.Istr + 1) * (B.Jend - B.Jstr + 1) >= 128L * 1024L ? G.N : 0);
  if (a.p1 > 0) {      // (both forms store the DIAGNOSTICS_TS terms: one template)
    a.p0 = (G.N + a.p1 - 1) / a.p1;
    LAUNCH_THREAD_AS(k_t3dmix2_s, k_t3dmix2_m, B.Iend - B.Istr + 1, B.Jend - B.Jstr + 1, a.p0 * G.NT, c->stream, a);
    return 0;
  }
  a.p0 = (G.N + KCH - 1) / KCH;
  LAUNCH_THREAD(k_t3dmix2_s, B.Iend - B.Istr + 1, B.Jend - B.Jstr + 1, a.p0 * G.NT, c->stream, a);
  return 0;
}

// uv3dmix2_geo.h: every loop nest of the routine as a point-wise kernel over all levels (k_uvmix_geo.h); mode 0 harmonic,
// 1 | 2 the first | second operator of uv3dmix4_geo.h (the first on the ranges widened by one point: ug_ranges)
static int run_uvmix_geo_op(roms_hip_ctx *c, int mode) {
  const DGrid &G = c->G;
  KArgs a = mk(c);
  a.p0 = mode;
  const int N = G.N;
  const UgR R = ug_ranges(G.T, mode);
  LAUNCH_THREAD(k_uvg_slopes, R.iE1 - KMIN(R.iU1, R.iS1) + 1, R.jE1 - KMIN(R.jS1, R.jV1) + 1, N + 1, c->stream, a);
  LAUNCH_THREAD(k_uvg_grads, R.iE1 - KMIN(R.iU1, R.iS0) + 1, R.jE1 - KMIN(R.jV1, R.jS0) + 1, N, c->stream, a);
  LAUNCH_THREAD(k_uvg_flux, R.iE1 - KMIN(R.iU1, R.iS0) + 1, R.jE1 - KMIN(R.jV1, R.jS0) + 1, N, c->stream, a);
  LAUNCH_THREAD(k_uvg_vflux, R.iE0 - R.iS0 + 1, R.jE0 - R.jS0 + 1, N + 1, c->stream, a);
  LAUNCH_THREAD(k_uvg_step, R.iE0 - R.iS0 + 1, R.jE0 - R.jS0 + 1, 1, c->stream, a);
  return 0;
}
static int run_uv3dmix2_geo(roms_hip_ctx *c) { return run_uvmix_geo_op(c, 0); }
// uv3dmix4_geo.h (UV_VIS4 + MIX_GEO_UV, round 6): the operator twice, the conditions on LapU, LapV between
static int run_uv3dmix4_geo(roms_hip_ctx *c) {
  const DGrid &G = c->G;
  const TB &B = G.T;
  int r = run_uvmix_geo_op(c, 1);
  if (r) return r;
  KArgs a = mk(c);
  const int nx = (B.Iendp1 + 1) - (B.Istrm1 - 1) + 1, ny = (B.Jendp1 + 1) - (B.Jstrm1 - 1) + 1;
  a.p1 = 0;
  LAUNCH_THREAD(k_uvg_lapbc, nx, ny, G.N, c->stream, a);
  if (!(G.ewp || G.nsp)) { a.p1 = 1; LAUNCH_THREAD(k_uvg_lapbc, nx, ny, G.N, c->stream, a); }
  return run_uvmix_geo_op(c, 2);
}

int run_uv3dmix2(roms_hip_ctx *c) {
  const DGrid &G = c->G;
  const TB &B = G.T;
  if (!(G.options & ROMS_UV_VIS2)) return 0;
  if (G.mix_geo_uv) return G.uv_vis4 ? run_uv3dmix4_geo(c) : run_uv3dmix2_geo(c);      // (uv3dmix.F: one of the two)
  KArgs a = mk(c);
  if (G.uv_vis4) launch_uv3dmix4(c, c->late_pre ? 1 : 0);
  else
  { // levels per thread: KCH unrolled on small grids; on large ones a thread marches the column in
    // ceil(N/30) equal parts (measured, us: 512x512x50 KCH 349, 17: 318, 25: 296, 50: 330;
    // 2048x256x30 KCH 425, 15: 375, 30: 365; 512x64x30 KCH 29, 10: 30, 25: 45).  ROMS_HIP_UVCH overrides.
    static const char *eu = getenv("ROMS_HIP_UVCH");
    const long cols = (long)(B.Iend - B.Istr + 1) * (B.Jend - B.Jstr + 1);
    const int parts = (G.N + 29) / 30;
    a.p2 = eu ? atoi(eu) : (cols >= 128L * 1024L ? (G.N + parts - 1) / parts : 0);
    a.p1 = c->late_pre ? 1 : 0;
    if (G.wet_dry) { a.p2 = 0; LAUNCH_THREAD_AS(k_uv3dmix2_s, k_uv3dmix2_wd, B.Iend - B.Istr + 1, B.Jend - B.Jstr + 1, (G.N + KCH - 1) / KCH, c->stream, a); }
    else if (a.p2 > 0) LAUNCH_THREAD_AS(k_uv3dmix2_s, k_uv3dmix2_m, B.Iend - B.Istr + 1, B.Jend - B.Jstr + 1, (G.N + a.p2 - 1) / a.p2, c->stream, a);
    else LAUNCH_THREAD(k_uv3dmix2_s, B.Iend - B.Istr + 1, B.Jend - B.Jstr + 1, (G.N + KCH - 1) / KCH, c->stream, a); }
  LAUNCH_THREAD(k_uv3dmix2_sum, B.Iend - B.Istr + 1, B.Jend - B.Jstr + 1, 1, c->stream, a);
  return run_duv_frc(c);                           // DIAGNOSTICS_UV: the vertical sums of the terms and the viscous terms
}

// rhs3d_tile's point-wise part.  Default: the LDS-tiled form (k_rhs3d_lds.h), a block of 64x4 points marching a
// chunk of levels; ROMS_HIP_RHS3D_LDS=0 (and the serial CPU emulation) take the point-wise form k_rhs3d_pt.
// Both are reported to the per-kernel timers as k_rhs3d_pt.
static int launch_rhs3d_point_part(roms_hip_ctx *c) {
  const DGrid &G = c->G;
  const TB &B = G.T;
  KArgs a = mk(c);
  const int nx = B.Iend - B.Istr + 1, ny = B.Jend - B.Jstr + 1;
#ifndef ROMS_CPU_EMU
  static const char *el = getenv("ROMS_HIP_RHS3D_LDS"), *ek = getenv("ROMS_HIP_RHS3D_KC");
  if (!(el && el[0] == '0') && !G.dia_uv) {
    // levels per chunk: whole columns once the (xi,eta) blocks alone fill the chip (>= 2048 blocks), chunks of
    // at least 5 levels on smaller grids so that grid.z supplies the blocks
    const int nt = ((nx + 63) / 64) * ((ny + 3) / 4);
    int nz = KMAX(1, (2048 + nt - 1) / nt);
    int kc = KMAX(5, (G.N + nz - 1) / nz);
    if (ek && atoi(ek) > 0) kc = atoi(ek);
    kc = KMIN(kc, G.N);
    nz = (G.N + kc - 1) / kc;
    a.p0 = kc;
    static const char *ew = getenv("ROMS_HIP_RHS3D_W");
    const int w = ew ? atoi(ew) : 2;
    const dim3 grid((unsigned)(8 * ((nt + 7) / 8) * nz), 1, 1), block(64, 4, 1);
    const size_t lds = (size_t)RL_LDS_DOUBLES * sizeof(double);
    // compiled for 2 waves per SIMD (198 VGPRs with the own-point values of the next level in flight): 300 us at
    // 512x512x50; 3 waves (168 VGPRs) spill 33 of them: 521 us; before the own-point loads were taken a level
    // ahead: 366 (3 waves) / 383 (2 waves); the point-wise form: 478
    if (w == 2) KPROF_WRAP(k_rhs3d_pt, c->stream, ROMS_LAUNCH(k_rhs3d_lds<2>, grid, block, lds, c->stream, a, nx, ny, nz));
    else KPROF_WRAP(k_rhs3d_pt, c->stream, ROMS_LAUNCH(k_rhs3d_lds<3>, grid, block, lds, c->stream, a, nx, ny, nz));
    return 0;
  }
#endif
  a.p0 = (G.N + KCH - 1) / KCH;
  if (G.dia_uv) LAUNCH_THREAD_AS(k_rhs3d_pt, k_rhs3d_pt_duv, nx, ny, 2 * a.p0, c->stream, a);   // DIAGNOSTICS_UV: with the term stores
  else LAUNCH_THREAD(k_rhs3d_pt, nx, ny, 2 * a.p0, c->stream, a);
  return 0;
}

int run_rhs3d_tile(roms_hip_ctx *c) {
  const DGrid &G = c->G;
  const TB &B = G.T;
  int r = launch_rhs3d_point_part(c);
  if (r) return r;
  if (G.wet_dry) { r = run_wd_scale3(c); if (r) return r; }         // WET_DRY: ru, rv(k) times the wet masks, then their sums (rhs3d.F:1709,1750)
  KArgs a = mk(c);
  a.p1 = 0;
  LAUNCH_THREAD(k_rhs3d_sum, B.Iend - B.Istr + 1, B.Jend - B.Jstr + 1, 1, c->stream, a);
  if (!(G.options & ROMS_UV_VIS2)) return run_duv_frc(c);      // (with UV_VIS2: behind uv3dmix2, whose per-level terms it sums)
  return 0;
}

// The same two routines as the fused main3d sequence launches them: the point-wise kernels of
// rhs3d_tile and uv3dmix2 are independent (the latter may run on the side stream), and ONE column
// kernel then forms rufrc/rvfrc from both, in the reference's order of additions.
int run_rhs3d_pt(roms_hip_ctx *c) { return launch_rhs3d_point_part(c); }
int run_uv3dmix2_s(roms_hip_ctx *c) {
  const DGrid &G = c->G;
  const TB &B = G.T;
  if (!(G.options & ROMS_UV_VIS2)) return 0;
  KArgs a = mk(c);
  if (G.uv_vis4) { launch_uv3dmix4(c, 0); return 0; }
  // in the late-predictor schedules only the terms are stored: the update of u,v(nnew) is k_pre_new's (folded) or
  // k_uv3dmix2_apply's, behind the predictor -- an update here would be overwritten by it (round 5: it was made, 35 MB of
  // reads and writes per BENCHMARK1 step for nothing; ROMS_HIP_UVDEFER=0 makes it again)
  static const char *edf = getenv("ROMS_HIP_UVDEFER");
  a.p1 = (c->late_pre && !(edf && edf[0] == '0')) ? 1 : 0;
  { // levels per thread: KCH unrolled on small grids; on large ones a thread marches the column in
    // ceil(N/30) equal parts (measured, us: 512x512x50 KCH 349, 17: 318, 25: 296, 50: 330;
    // 2048x256x30 KCH 425, 15: 375, 30: 365; 512x64x30 KCH 29, 10: 30, 25: 45).  ROMS_HIP_UVCH overrides.
    static const char *eu = getenv("ROMS_HIP_UVCH");
    const long cols = (long)(B.Iend - B.Istr + 1) * (B.Jend - B.Jstr + 1);
    const int parts = (G.N + 29) / 30;
    a.p2 = eu ? atoi(eu) : (cols >= 128L * 1024L ? (G.N + parts - 1) / parts : 0);
    if (G.wet_dry) { a.p2 = 0; LAUNCH_THREAD_AS(k_uv3dmix2_s, k_uv3dmix2_wd, B.Iend - B.Istr + 1, B.Jend - B.Jstr + 1, (G.N + KCH - 1) / KCH, c->stream, a); }
    else if (a.p2 > 0) LAUNCH_THREAD_AS(k_uv3dmix2_s, k_uv3dmix2_m, B.Iend - B.Istr + 1, B.Jend - B.Jstr + 1, (G.N + a.p2 - 1) / a.p2, c->stream, a);
    else LAUNCH_THREAD(k_uv3dmix2_s, B.Iend - B.Istr + 1, B.Jend - B.Jstr + 1, (G.N + KCH - 1) / KCH, c->stream, a); }
  return 0;
}
// uv3dmix2 and the coupling sums of rhs3d_tile as one column-marching kernel (k_uv3dmix2_col.h): large grids
// only.  Returns -1 when the form does not apply (the caller then takes k_uv3dmix2_s + k_rhs3d_sum).
int run_uv3dmix2_col(roms_hip_ctx *c) {
#ifdef ROMS_CPU_EMU
  (void)c;
  return -1;
#else
  const DGrid &G = c->G;
  const TB &B = G.T;
  if (!(G.options & ROMS_UV_VIS2) || G.masking || G.dia_uv || G.uv_vis4 || G.mix_geo_uv) return -1;   // (the column form carries no land/sea masks and leaves no per-level terms)
  static const char *e = getenv("ROMS_HIP_UVCOL");
  const int nx = B.Iend - B.Istr + 1, ny = B.Jend - B.Jstr + 1;
  const bool big = (long)nx * ny >= 128L * 1024L;
  if (e ? e[0] == '0' : !big) return -1;
  KArgs a = mk(c);
  const int nt = ((nx + 63) / 64) * ((ny + 3) / 4);
  KPROF_WRAP(k_uv3dmix2_col, c->stream,
             ROMS_LAUNCH(k_uv3dmix2_col, dim3((unsigned)(8 * ((nt + 7) / 8)), 1, 1), dim3(64, 4, 1),
                                (size_t)UC_LDS_DOUBLES * sizeof(double), c->stream, a, nx, ny));
  return 0;
#endif
}
int run_rufrc_sums(roms_hip_ctx *c) {
  const DGrid &G = c->G;
  const TB &B = G.T;
  if (G.wet_dry) { int r = run_wd_scale3(c); if (r) return r; }
  KArgs a = mk(c);
  a.p1 = ((G.options & ROMS_UV_VIS2) && !G.mix_geo_uv) ? 1 : 0;     // (MIX_GEO_UV: k_uvg_step adds its terms to the sums itself)
  LAUNCH_THREAD(k_rhs3d_sum, B.Iend - B.Istr + 1, B.Jend - B.Jstr + 1, 1, c->stream, a);
  return run_duv_frc(c);
}
