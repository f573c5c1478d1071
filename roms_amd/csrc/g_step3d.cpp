// g_step3d.cpp -- launch sequences of step3d_uv and step3d_t.
#include "roms_host.h"
#include <cstdlib>
#include "k_step3d.h"
#include "k_mpdata.h"

static inline KArgs mk(roms_hip_ctx *c, int p0 = 0) {
  KArgs a;
  a.G = c->G;
  a.Fv = c->F;
  a.p0 = p0; a.p1 = 0; a.p2 = 0;
  return a;
}
// column kernels with their per-column state in LDS (COL launches): 2*(N+1) doubles per column must
// fit the 64 KB a block gets without opting in; ROMS_HIP_COLLDS=0 selects the private-memory forms
static inline bool col_lds(const DGrid &G) {
  static const char *e = getenv("ROMS_HIP_COLLDS");
  return !(e && e[0] == '0') && 2 * (G.N + 1) * 64 * sizeof(double) < 64 * 1024;
}
static inline size_t lds_sz(const DGrid &G) { return (size_t)(G.bw + 6) * (size_t)(G.bh + 6); }


int run_step3d_uv(roms_hip_ctx *c) {
  const DGrid &G = c->G;
  const TB &B = G.T;
  const int N = G.N, nnew = G.nnew;
  KArgs a = mk(c);
  if (G.wet_dry) {
    // WET_DRY: the new velocity times the land mask, then times the wet mask (step3d_uv.F:717-720 ...): the kernels get the
    // product of the two as their mask (k_wd_eff: the same bits)
    int r = run_wd_eff(c);
    if (r) return r;
    a.Fv.umask = (double *)c->F.wd_eff; a.Fv.vmask = (double *)c->F.wd_eff + G.nij;
  }
  static const char *ech = getenv("ROMS_HIP_COLCH"), *ereg = getenv("ROMS_HIP_UVREG");
  if (G.dia_uv) { int r = run_duv_s3uv(c); if (r) return r; }          // DIAGNOSTICS_UV: the column kernel with the diagnostic statements (k_duv.h)
  else if (G.options & ROMS_PLAIN_VVISC) LAUNCH_THREAD(k_s3uv_col_p, B.Iend - B.Istr + 1, B.Jend - B.Jstr + 1, 2, c->stream, a);   // without SPLINES_VVISC
  else if (col_lds(G) && !(ereg && ereg[0] == '0') && N <= 32) LAUNCH_COL_AS(k_s3uv_col, k_s3uv_col_r32, B.Iend - B.Istr + 1, B.Jend - B.Jstr + 1, 2, 2 * (N + 1), c->stream, a);
  else if (col_lds(G) && !(ereg && ereg[0] == '0') && N <= 52) LAUNCH_COL_AS(k_s3uv_col, k_s3uv_col_r52, B.Iend - B.Istr + 1, B.Jend - B.Jstr + 1, 2, 2 * (N + 1), c->stream, a);
  else if (col_lds(G) && (ech ? ech[0] == '1' : N > 40)) LAUNCH_COL_AS(k_s3uv_col, k_s3uv_col_l10, B.Iend - B.Istr + 1, B.Jend - B.Jstr + 1, 2, 2 * (N + 1), c->stream, a);
  else if (col_lds(G)) LAUNCH_COL_AS(k_s3uv_col, k_s3uv_col_l, B.Iend - B.Istr + 1, B.Jend - B.Jstr + 1, 2, 2 * (N + 1), c->stream, a);
  else LAUNCH_THREAD(k_s3uv_col, B.Iend - B.Istr + 1, B.Jend - B.Jstr + 1, 2, c->stream, a);
  if (G.wet_dry) { int r = run_wd_scale3(c); if (r) return r; }        // ... and ru, rv(nrhs) times the wet mask (:721, :1188)
  if (!G.fuse3d) {   // (fused: the kernels store the boundary values and periodic images themselves, pt_emit)
    if (G.obc) { int r = run_obc3d_uv(c, nnew); if (r) return r; }
    else {
      const int wet = G.wet_dry ? BC_WET3 : 0;
      HaloSpec sp[2] = {{uv_lev(c, c->F.u, nnew), N, BC_U | wet, 0}, {uv_lev(c, c->F.v, nnew), N, BC_V | wet, 0}};   // u3dbc/v3dbc :1266,1271
      // (multi-tile: the coupling below reads the tile's own columns and these boundary values only, and exchanges u, v(nnew)
      // itself: the conditions without the strips)
      if (ghost_compute(c, 32)) launch_fill_only(c, sp, 2, G.T);
      else launch_halo_multi(c, sp, 2);
    }
  }
  auto couple = [&]() {
    if (col_lds(G) && !G.dia_uv) LAUNCH_COL_AS(k_s3uv_couple, k_s3uv_couple_l, B.IendT - KMIN(B.IstrP, B.IstrT) + 1, B.JendT - KMIN(B.JstrT, B.Jstr) + 1, 2, 2 * (N + 1), c->stream, a);
    else LAUNCH_THREAD(k_s3uv_couple, B.IendT - KMIN(B.IstrP, B.IstrT) + 1, B.JendT - KMIN(B.JstrT, B.Jstr) + 1, 2, c->stream, a);
  };
  HaloSpec sp[6] = {{uv_lev(c, c->F.u, nnew), N, BC_NONE, 'u'}, {uv_lev(c, c->F.v, nnew), N, BC_NONE, 'v'},
                    {c->F.Huon, N, BC_NONE, 'u'},               {c->F.Hvom, N, BC_NONE, 'v'},
                    {c->F.ubar, 2, BC_NONE, 'u'},               {c->F.vbar, 2, BC_NONE, 'v'}};   // :1763-1830
  if (c->rim_split && !G.fuse3d) {
    // multi-tile, round 4: the columns the exchange packs first, the exchange (u, v, Huon, Hvom and ubar, vbar(1:2): every
    // reader of the 2-D state fences FG_2D) on its own stream behind them, the interior columns beside it
    a.G.region = 1; couple();
    c->x_2d_ok = true; launch_halo_tail(c, sp, 6); c->x_2d_ok = false;
    a.G.region = 2; couple();
    a.G.region = 0;
    return 0;
  }
  couple();
  if (!G.fuse3d) launch_halo_multi(c, sp, 6);
  return 0;
}

int run_step3d_t(roms_hip_ctx *c) {
  const DGrid &G = c->G;
  const TB &B = G.T;
  const int N = G.N, nnew = G.nnew;
  bool any_mp = false;
  for (int it = 0; it < G.NT; it++) {
    const bool hm = G.hadv[it] == ROMS_MPDATA, vm = G.vadv[it] == ROMS_MPDATA;
    if (hm != vm) { set_error("step3d_t: MPDATA must be selected for both Hadvection and Vadvection of a tracer"); return 8; }
    any_mp |= hm;
  }
  if (any_mp && !c->F.mp3[0]) { set_error("step3d_t: MPDATA work arrays missing"); return 8; }
  if (any_mp) {
    // exchange of t(nnew) :420: the extended range below reads it at ghost points
    HaloSpec sp[ROMS_MAXT];
    int n = 0;
    for (int it = 1; it <= G.NT; it++)
      if (G.hadv[it - 1] == ROMS_MPDATA) sp[n++] = {t_lev(c, nnew, it), N, BC_NONE, 'r'};
    launch_halo_multi(c, sp, n);
  }
  KArgs a = mk(c);
  // exchange of t(nnew) for HSIMT tracers (:420) only refreshes ghost points that are not read here
  bool any_pt = false, any_hsimt = false;   // point kernel: every horizontal scheme but MPDATA and HSIMT (LDS tiles)
  for (int it = 0; it < G.NT; it++) {
    any_pt |= G.hadv[it] != ROMS_MPDATA && G.hadv[it] != ROMS_HSIMT;
    any_hsimt |= G.hadv[it] == ROMS_HSIMT;
  }
  a.p0 = (N + KCH - 1) / KCH;
  const bool plain = (G.options & ROMS_PLAIN_VDIFF) != 0;    // without SPLINES_VDIFF: the straightforward kernel forms, then k_mp_vdiff
  c->tadv_hdone = 0; c->tadv_vdone = 0;
  if ((any_pt || any_hsimt) && (plain || !launch_tadv_lds(c, 1))) {
    c->tadv_hdone = 0; c->tadv_vdone = 0;
    if (any_pt) LAUNCH_THREAD(k_s3t_hv, B.Iend - B.Istr + 1, B.Jend - B.Jstr + 1, a.p0 * G.NT, c->stream, a);
  }
  // (k_tadv_lds with HS has done the HSIMT tracers: c->tadv_hdone; all of them or none)
  if (any_hsimt && !c->tadv_hdone) LAUNCH_COOP(k_s3t_h, G.nbx, G.nby, N * G.NT, 256, S3T_NLDS * lds_sz(G), c->stream, a);
  a.p1 = c->tadv_vdone;                                      // k_s3t_col: tracers whose vertical advection is done
  bool any_col = false;        // (a launch whose tracers are all MPDATA's would return at once: k_mpdata.h does their column work)
  for (int it = 0; it < G.NT; it++) any_col |= G.vadv[it] != ROMS_MPDATA;
  auto column_part = [&]() {
  if (any_col) {
    const int nx = B.Iend - B.Istr + 1, ny = B.Jend - B.Jstr + 1;
#ifdef ROMS_CPU_EMU
    if (col_lds(G) && !G.dia_ts) LAUNCH_COL_AS(k_s3t_col, k_s3t_col_l, nx, ny, G.NT, 2 * (N + 1), c->stream, a);
    else LAUNCH_THREAD(k_s3t_col, nx, ny, G.NT, c->stream, a);
#else
    const char *er = getenv("ROMS_HIP_COLREGS");
    const bool regs = !(er && er[0] == '0');
    static const char *el = getenv("ROMS_HIP_S3TLDS");
    bool hsimt_v = false;
    for (int it = 0; it < G.NT; it++) hsimt_v |= G.vadv[it] == ROMS_HSIMT && !((c->tadv_vdone >> it) & 1);
    const bool ldsform = col_lds(G) && (el ? el[0] == '1' : (N != 30 || hsimt_v));
    // chunks of 10 levels on tall columns: 542 -> 499 us at N = 50 (ROMS_HIP_S3TCH=0/1 forces a form)
    static const char *e10 = getenv("ROMS_HIP_S3TCH");
    if (plain || G.dia_ts) LAUNCH_THREAD(k_s3t_col, nx, ny, G.NT, c->stream, a);      // (DIAGNOSTICS_TS: the form that stores the terms)
    else if (ldsform && (e10 ? e10[0] == '1' : N > 40)) LAUNCH_COL_AS(k_s3t_col, k_s3t_col_l10, nx, ny, G.NT, 2 * (N + 1), c->stream, a);
    else if (ldsform) LAUNCH_COL_AS(k_s3t_col, k_s3t_col_l, nx, ny, G.NT, 2 * (N + 1), c->stream, a);
    else if (regs && N == 30) LAUNCH_THREAD_AS(k_s3t_col, k_s3t_col_n30, nx, ny, G.NT, c->stream, a);
    else LAUNCH_THREAD(k_s3t_col, nx, ny, G.NT, c->stream, a);
#endif
  }
  };
  HaloSpec spt[ROMS_MAXT];
  for (int it = 1; it <= G.NT; it++) spt[it - 1] = {t_lev(c, nnew, it), N, obc_bc(c, bc_rstate(c, true)), 'r'};   // t3dbc :1858 + exchange :1920
  if (c->rim_split && any_col && !any_mp && !plain && !G.obc && !G.fuse3d && !G.masking && !G.dia_ts && !(G.clima & ~1)) {
    // multi-tile, round 4: the columns the exchange packs first, the exchange of t(nnew) on its own stream, the rest beside it
    // (not with MASKING: its fill multiplies the WHOLE plane by rmask, step3d_t.F:1880-1890, interior included)
    a.G.region = 1; column_part();
    launch_halo_tail(c, spt, G.NT);
    a.G.region = 2; column_part();
    a.G.region = 0;
    return 0;
  }
  column_part();
  for (int it = 1; it <= G.NT && any_mp; it++) {
    if (G.hadv[it - 1] != ROMS_MPDATA) continue;
    MpArgs m;
    m.G = G; m.Fv = c->F; m.itrc = it;
    const int LmT = B.Iend - B.Istr + 1, MmT = B.Jend - B.Jstr + 1;
    LAUNCH_THREAD(k_mp_ta, B.Iendp2i - B.IstrUm2 + 1, B.Jendp2i - B.JstrVm2 + 1, N, c->stream, m);
    static const char *euw = getenv("ROMS_HIP_MP_UVWA");          // 0: Ua|Va and Wa as two launches
    if (!(euw && euw[0] == '0')) {
      LAUNCH_THREAD(k_mp_uvwa, B.Iendp2 - (B.IstrU - 1) + 1, B.Jendp2 - (B.JstrV - 1) + 1, N + 1, c->stream, m);
    } else {
      LAUNCH_THREAD(k_mp_uva, B.Iendp2 - (B.IstrU - 1) + 1, B.Jendp2 - KMIN(B.JstrV - 1, B.JstrVm1) + 1, 2 * N, c->stream, m);
      LAUNCH_THREAD(k_mp_wa, B.Iendp1 - (B.IstrU - 1) + 1, B.Jendp1 - (B.JstrV - 1) + 1, N + 1, c->stream, m);
    }
    static const char *elim = getenv("ROMS_HIP_MPFUSE");
    LAUNCH_THREAD(k_mp_beta, B.Iendp1 - (B.IstrU - 1) + 1, B.Jendp1 - (B.JstrV - 1) + 1, N, c->stream, m);
    if (elim && elim[0] == '0') {
      LAUNCH_THREAD(k_mp_limit, LmT + 1, MmT + 1, N, c->stream, m);
      LAUNCH_THREAD(k_mp_apply, LmT, MmT, N, c->stream, m);
    } else {
      LAUNCH_THREAD_AS(k_mp_apply, k_mp_limapply, LmT, MmT, N, c->stream, m);
    }
    if (col_lds(G)) LAUNCH_COL_AS(k_mp_vdiff, k_mp_vdiff_l, LmT, MmT, 1, 2 * (N + 1), c->stream, m);
    else LAUNCH_THREAD(k_mp_vdiff, LmT, MmT, 1, c->stream, m);
  }
  for (int it = 1; it <= G.NT && plain; it++) {        // step3d_t.F:1722-1790 for every tracer that is not MPDATA's (done above)
    if (G.hadv[it - 1] == ROMS_MPDATA) continue;
    MpArgs m;
    m.G = G; m.Fv = c->F; m.itrc = it;
    const int LmT = B.Iend - B.Istr + 1, MmT = B.Jend - B.Jstr + 1;
    if (col_lds(G)) LAUNCH_COL_AS(k_mp_vdiff, k_mp_vdiff_l, LmT, MmT, 1, 2 * (N + 1), c->stream, m);
    else LAUNCH_THREAD(k_mp_vdiff, LmT, MmT, 1, c->stream, m);
  }
  if (G.fuse3d && !any_mp && !plain) return run_dia_rate(c);   // k_s3t_col stored the boundary values and images (pt_emit); DIAGNOSTICS_TS: the rate term :1892-1904
  HaloSpec sp[ROMS_MAXT];
  if (G.obc) for (int it = 1; it <= G.NT; it++) { int r = run_obc3d_t(c, nnew, it); if (r) return r; }
  if (G.clima & ~1) {
    // nudging towards the tracer climatology :1866-1878 sits between t3dbc and the land/sea mask + exchange: the boundary
    // values first (closed walls here; open edges: run_obc3d_t above has set them), the nudging on the whole (IstrR:IendR,
    // JstrR:JendR) range, then the mask of the whole plane and the exchange
    if (!G.obc) {
      for (int it = 1; it <= G.NT; it++) sp[it - 1] = {t_lev(c, nnew, it), N, bc_rstate(c, false), 'r'};
      launch_halo_multi(c, sp, G.NT);
    }
    KArgs an;
    an.G = G; an.Fv = c->F; an.p0 = an.p1 = an.p2 = 0;
    LAUNCH_THREAD(k_tnudge, B.IendR - B.IstrR + 1, B.JendR - B.JstrR + 1, N * G.NT, c->stream, an);
    for (int it = 1; it <= G.NT; it++) sp[it - 1] = {t_lev(c, nnew, it), N, bc_rstate(c, true) & ~BC_KIND, 'r'};
    launch_halo_tail(c, sp, G.NT);
    if (G.dia_ts) { halo_fence(c, FG_T); return run_dia_rate(c); }
    return 0;
  }
  for (int it = 1; it <= G.NT; it++) sp[it - 1] = {t_lev(c, nnew, it), N, obc_bc(c, bc_rstate(c, true)), 'r'};   // t3dbc :1858 + exchange :1920
  launch_halo_tail(c, sp, G.NT);
  if (G.dia_ts) { halo_fence(c, FG_T); return run_dia_rate(c); }     // the rate term reads the boundary values t3dbc has just set
  return 0;
}
