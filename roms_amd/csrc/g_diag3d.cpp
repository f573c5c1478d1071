// g_diag3d.cpp -- launch sequences of the diagnostic 3-D kernels (k_diag3d.h): the bodies of the
// reference wrappers set_depth, set_massflux, rho_eos, set_vbc, ana_vmix, set_data, omega,
// wvelocity, set_zeta, ini_zeta, ini_fields, each followed by the same boundary fills / periodic
// exchanges the reference issues at the tail of the _tile routine.
#include "roms_host.h"
#include <cstdlib>
#include "k_diag3d.h"

static inline KArgs mk(roms_hip_ctx *c, int p0 = 0, int p1 = 0, int p2 = 0) {
  KArgs a;
  a.G = c->G;
  a.Fv = c->F;
  a.p0 = p0; a.p1 = p1; a.p2 = p2;
  return a;
}

int run_eos_nonlinear(roms_hip_ctx *c);   // g_eos.cpp
int run_set_data_benchmark(roms_hip_ctx *c);

int run_set_depth(roms_hip_ctx *c) {
  const TB &B = c->G.T;
  const int N = c->G.N;
  KArgs a = mk(c);
  if (ghost_compute(c, 1)) {
    // Zt_avg1 and h carry valid ghost lines (the exchange behind the fast steps; ini_zeta): depths and thicknesses there are
    // functions of the column's own Zt_avg1 and h -- computed with the tile, no exchange (set_depth.F:417-440)
    if (!c->h_ghost_done) {      // (the reference exchanges the time-invariant h with every call: once is enough)
      const HaloSpec hh = {c->F.h, 1, BC_NONE, 'r'};
      launch_halo_multi(c, &hh, 1);
      c->h_ghost_done = true;
    }
    const TB X = ghost_tb(c, 3, c->G.Nghost);
    a.G.T = X;
    LAUNCH_THREAD(k_set_depth, X.IendT - X.IstrT + 1, X.JendT - X.JstrT + 1, N, c->stream, a);
    return 0;
  }
  LAUNCH_THREAD(k_set_depth, B.IendT - B.IstrT + 1, B.JendT - B.JstrT + 1, N, c->stream, a);
  if (c->G.fuse3d) return 0;   // the kernel stored the periodic images itself (emit_store); h is time-invariant
  const HaloSpec hs1[] = {
      {c->F.h, 1, BC_NONE, 'r'},
      {c->F.z_w, N + 1, BC_NONE, 'r'},
      {c->F.z_r, N, BC_NONE, 'r'},
      {c->F.Hz, N, BC_NONE, 'r'},
  };
  launch_halo_tail(c, hs1, 4);
  return 0;
}

int run_set_massflux(roms_hip_ctx *c) {
  const TB &B = c->G.T;
  const int N = c->G.N;
  KArgs a = mk(c);
  if (ghost_compute(c, 8) && c->G.Nghost == 2) {
    // Huon(i) = f(Hz(i-1), Hz(i), u(i)): with Hz and u valid on 3 | 2 ghost lines the fluxes on the two ghost lines the
    // reference's exchange fills (NghostPoints = 2) are computed with the tile (set_massflux.F:139-160)
    const TB X = ghost_tb(c, 2, 2);
    a.G.T = X;
    const int x0 = KMIN(X.IstrP, X.IstrT), y0 = KMIN(X.JstrT, X.JstrP);
    LAUNCH_THREAD(k_set_massflux, X.IendT - x0 + 1, X.JendT - y0 + 1, N, c->stream, a);
    return 0;
  }
  const int i0 = KMIN(B.IstrP, B.IstrT), j0 = KMIN(B.JstrT, B.JstrP);
  LAUNCH_THREAD(k_set_massflux, B.IendT - i0 + 1, B.JendT - j0 + 1, N, c->stream, a);
  if (c->G.fuse3d) return 0;   // the kernel stored the periodic images itself (pt_emit)
  const HaloSpec hs2[] = {
      {c->F.Huon, N, BC_NONE, 'u'},
      {c->F.Hvom, N, BC_NONE, 'v'},
  };
  launch_halo_tail(c, hs2, 2);
  return 0;
}

int run_eos_alfaobeta(roms_hip_ctx *c);   // g_bench.cpp
static int run_rho_eos_only(roms_hip_ctx *c);
int run_rho_eos(roms_hip_ctx *c) {
  const int r = run_rho_eos_only(c);
  return (r || !c->G.ddmix) ? r : run_eos_alfaobeta(c);       // LMD_DDMIX: alfaobeta (rho_eos.F:454, :794)
}
static int run_rho_eos_only(roms_hip_ctx *c) {
  if (c->G.options & ROMS_NONLIN_EOS) return run_eos_nonlinear(c);
  const TB &B = c->G.T;
  const int N = c->G.N;
  KArgs a = mk(c);
  if (ghost_compute(c, 4)) {      // (the ghost columns with the tile, no exchange: run_eos_nonlinear)
    const TB X = ghost_tb(c, 3, c->G.Nghost);
    a.G.T = X;
    LAUNCH_THREAD(k_rho_eos_lin, X.IendT - X.IstrT + 1, X.JendT - X.JstrT + 1, 1, c->stream, a);
    return 0;
  }
  LAUNCH_THREAD(k_rho_eos_lin, B.IendT - B.IstrT + 1, B.JendT - B.JstrT + 1, 1, c->stream, a);
  if (c->G.fuse3d) return 0;   // the kernel stored the periodic images itself (pt_emit)
  const HaloSpec hs3[] = {
      {c->F.rho, N, BC_NONE, 'r'},
      {c->F.pden, N, BC_NONE, 'r'},
      {c->F.rhoA, 1, BC_NONE, 'r'},
      {c->F.rhoS, 1, BC_NONE, 'r'},
      {c->F.bvf, N + 1, BC_NONE, 'r'},    // BV_FREQUENCY: LMD_MIXING, GLS_MIXING (:751-764)
      {c->F.alpha, 1, BC_NONE, 'r'},      // LMD_MIXING only (:766-780)
      {c->F.beta, 1, BC_NONE, 'r'},
  };
  launch_halo_tail(c, hs3, (c->G.options & ROMS_LMD_MIXING) ? 7 : ((c->G.options & (ROMS_GLS_MIXING | ROMS_MY25_MIXING)) ? 5 : 4));
  return 0;
}

int run_set_vbc(roms_hip_ctx *c) {
  const TB &B = c->G.T;
  KArgs a = mk(c);
  LAUNCH_THREAD(k_set_vbc, B.IendR - B.IstrR + 1, B.JendR - B.JstrR + 1, 1, c->stream, a);
  if (c->G.fuse3d) return 0;   // the kernel stored the boundary values and images itself (emit_store)
  // bc_u2d_tile / bc_v2d_tile: closed, or zero gradient where LBC(:,isUbar / isVbar) is not (bc_2d.F:201-290; BC_LBC2D)
  const HaloSpec hs4[] = {
      {c->F.bustr, 1, BC_U | (c->G.obc ? BC_LBC2D : 0), 'u'},
      {c->F.bvstr, 1, BC_V | (c->G.obc ? BC_LBC2D : 0), 'v'},
  };
  launch_halo_multi(c, hs4, 2);
  return 0;
}

int run_ana_vmix(roms_hip_ctx *c) {
  const TB &B = c->G.T;
  const int N = c->G.N;
  KArgs a = mk(c);
  LAUNCH_THREAD(k_ana_vmix, B.IendT - B.IstrT + 1, B.JendT - B.JstrT + 1, N - 1, c->stream, a);
  const HaloSpec hs5[] = {
      {c->F.Akv, N + 1, BC_NONE, 'r'},
      {c->F.Akt, (N + 1) * c->G.NAT, BC_NONE, 'r'},
  };
  launch_halo_tail(c, hs5, 2);
  return 0;
}

int run_set_data(roms_hip_ctx *c) {
  if (c->G.options & ROMS_APP_BENCHMARK) return run_set_data_benchmark(c);
  const TB &B = c->G.T;
  KArgs a = mk(c);
  {
    // ana_smflux.h:306-318 (UPWELLING; KELVIN and the others: the default branch, no wind): uniform in space, so the
    // host forms it -- with the libm the reference itself calls -- and the kernel only stores it
    const DGrid &G = c->G;
    const double pi = 3.14159265358979323846;
    if (!(G.options & ROMS_APP_UPWELLING)) a.d0 = 0.0;
    else if ((G.tdays - G.dstart) <= 2.0) a.d0 = -0.1 * sin(pi * (G.tdays - G.dstart) / 4.0) / G.rho0;
    else a.d0 = -0.1 / G.rho0;
  }
  const int i0 = KMIN(B.IstrP, B.IstrT), j0 = KMIN(B.JstrP, B.JstrT);
  LAUNCH_THREAD(k_set_data_upw, B.IendT - i0 + 1, B.JendT - j0 + 1, 1, c->stream, a);
  const HaloSpec hs6[] = {
      {c->F.stflux, 2, BC_NONE, 'r'},
      {c->F.sustr, 1, BC_NONE, 'u'},
      {c->F.svstr, 1, BC_NONE, 'v'},
      {c->F.srflx, 1, BC_NONE, 'r'},
  };
  launch_halo_multi(c, hs6, (c->G.options & ROMS_SOLAR_SOURCE) ? 4 : 3);
  if (c->G.options & ROMS_APP_KELVIN) {     // ANA_FSOBC, ANA_M2OBC: the boundary data of this step (other applications upload theirs)
    auto acquire = [&](int e, int v) {
      const int k = lbc_kind(c->cfg, e, v);
      if (k == ROMS_LBC_CLA || k == ROMS_LBC_RADNUD || k == ROMS_LBC_FLA || k == ROMS_LBC_SHC) return true;
      if (v == ROMS_ISFSUR)
        for (int q = ROMS_ISUBAR; q <= ROMS_ISVBAR; q++) {
          const int kq = lbc_kind(c->cfg, e, q);
          if (kq == ROMS_LBC_FLA || kq == ROMS_LBC_SHC) return true;
        }
      return false;
    };
    KArgs k = mk(c, (acquire(ROMS_IWEST, ROMS_ISFSUR) ? 1 : 0) | (acquire(ROMS_IEAST, ROMS_ISFSUR) ? 2 : 0) |
                       (acquire(ROMS_IWEST, ROMS_ISUBAR) && acquire(ROMS_IWEST, ROMS_ISVBAR) ? 4 : 0) |
                       (acquire(ROMS_IEAST, ROMS_ISUBAR) && acquire(ROMS_IEAST, ROMS_ISVBAR) ? 8 : 0));
    LAUNCH_THREAD(k_set_data_kelvin, B.JendT - KMIN(B.JstrP, B.JstrT) + 1, 1, 1, c->stream, k);
  }
  return 0;
}

int run_omega(roms_hip_ctx *c) {
  const TB &B = c->G.T;
  KArgs a = mk(c);
  // (round 6, tried and dropped: W on the two ghost lines below and the one above the tile computed with the tile from Huon,
  // Hvom there -- every line rhs3d.F reads -- instead of exchanged.  It leaves the OUTER ghost lines of a periodic domain edge,
  // W(-2,:) and W(Lm+2,:), unwritten: nobody reads them, but a serial run defines them, and the tiled runs are held to every
  // point a serial run defines.)
  {
    static const char *e = getenv("ROMS_HIP_COLLDS");
    const bool l = !(e && e[0] == '0') && 2 * (c->G.N + 1) * 64 * sizeof(double) < 64 * 1024;
    if (l) LAUNCH_COL_AS(k_omega, k_omega_l, B.Iend - B.Istr + 1, B.Jend - B.Jstr + 1, 1, 2 * (c->G.N + 1), c->stream, a);
    else LAUNCH_THREAD(k_omega, B.Iend - B.Istr + 1, B.Jend - B.Jstr + 1, 1, c->stream, a);
  }
  if (!c->G.fuse3d) {   // bc_w3d_tile (fused: pt_emit in the kernel)
    const HaloSpec hw = {c->F.W, c->G.N + 1, BC_R, 'r'};
    launch_halo_tail(c, &hw, 1);
  }
  return 0;
}

int run_wvelocity(roms_hip_ctx *c, int ninp) {
  const TB &B = c->G.T;
  const int N = c->G.N;
  const HaloSpec hs7[] = {
      {c->F.DU_avg1, 1, BC_NONE, 'u'},
      {c->F.DV_avg1, 1, BC_NONE, 'v'},
  };
  // (a single tile whose barotropic kernels store the periodic images of what they write -- DGrid::fuse_halo -- has left
  // DU_avg1, DV_avg1 complete: the fast-time averages are closed by the last call with their images)
  // (... and so has a multi-tile context inside roms_hip_main3d: the exchange behind the fast steps carried them)
  if (!c->G.fuse_halo && !ghost_compute(c, 64)) launch_halo_multi(c, hs7, 2);
  KArgs a = mk(c, ninp);
  static const char *ef = getenv("ROMS_HIP_WVELF");
  if (ef && ef[0] == '0') {
    LAUNCH_THREAD(k_wvel_vert, B.Iend - B.Istr + 1, B.Jend - B.Jstr + 1, N, c->stream, a);
    LAUNCH_THREAD(k_wvel, B.Iend - B.Istr + 1, B.Jend - B.Jstr + 1, N + 1, c->stream, a);
    launch_halo(c, c->F.wvel, N + 1, BC_R, 'r');     // bc_w3d_tile
    return 0;
  }
  static const char *ew = getenv("ROMS_HIP_WVELCH");
  // w-levels per thread (each chunk forms three levels of vert again): measured at 512x512x50 5: 216,
  // 17: 177, 51: 162 us; 2048x256x30 5: 275, 17: 235, 31: 223; 512x64x30 5: 21.6, 10: 20.5, 17: 25.7
  const long cols = (long)(B.Iend - B.Istr + 1) * (B.Jend - B.Jstr + 1);
  a.p1 = ew ? atoi(ew) : (cols >= 128L * 1024L ? N + 1 : 10);
  if (a.p1 < 1) a.p1 = KCH;
  LAUNCH_THREAD_AS(k_wvel, k_wvel_f, B.Iend - B.Istr + 1, B.Jend - B.Jstr + 1, (N + a.p1) / a.p1, c->stream, a);
  if (!c->G.fuse3d) {   // bc_w3d_tile (fused: emit_store in the kernel)
    const HaloSpec hw = {c->F.wvel, N + 1, BC_R, 'r'};
    launch_halo_tail(c, &hw, 1);
  }
  return 0;
}

int run_set_zeta(roms_hip_ctx *c) {
  const TB &B = c->G.T;
  KArgs a = mk(c);
  static const char *ezx = getenv("ROMS_HIP_SETZETA_X");
  if (c->has_exchange && !(ezx && ezx[0] == '0')) {
    // Zt_avg1 carries valid ghost lines (exchanged with the final fast-time averages; ini_zeta at the first step): copy
    // them with the tile -- three lines on the low side, Nghost on the high side, the boundary line of a closed edge --
    // and no exchange follows (set_zeta.F:100-118 exchanges zeta(1:2), whose ghost lines then hold the same values)
    const DGrid &G = c->G;
    const int i0 = (B.west && !G.ewp) ? B.IstrR : B.Istr - 3, i1 = (B.east && !G.ewp) ? B.IendR : B.Iend + G.Nghost;
    const int j0 = (B.south && !G.nsp) ? B.JstrR : B.Jstr - 3, j1 = (B.north && !G.nsp) ? B.JendR : B.Jend + G.Nghost;
    a.p0 = i0; a.p1 = j0;
    LAUNCH_THREAD(k_set_zeta_x, i1 - i0 + 1, j1 - j0 + 1, 1, c->stream, a);
    return 0;
  }
  LAUNCH_THREAD(k_set_zeta, B.IendR - B.IstrR + 1, B.JendR - B.JstrR + 1, 1, c->stream, a);
  if (!c->G.fuse3d) launch_halo(c, c->F.zeta, 2, BC_NONE, 'r');   // (fused: emit_store in the kernel)
  return 0;
}

int run_ini_zeta(roms_hip_ctx *c) {
  const TB &B = c->G.T;
  const int kstp = c->G.kstp;
  // radiation / Chapman conditions need the boundary values of the initial state: the load then covers them and no
  // condition is applied (ini_fields.F:830-871)
  bool keep = false;
  for (int e = 0; e < 4; e++) {
    const int k = lbc_kind(c->cfg, e, ROMS_ISFSUR);
    keep |= k == ROMS_LBC_RAD || k == ROMS_LBC_RADNUD || k == ROMS_LBC_CHE || k == ROMS_LBC_CHI;
  }
  if (c->G.masking) {
    KArgs m = mk(c, keep ? 3 : 0);
    if (keep) LAUNCH_THREAD(k_ini_mask, B.IendT - B.IstrT + 1, B.JendT - B.JstrT + 1, 1, c->stream, m);
    else LAUNCH_THREAD(k_ini_mask, B.IendB - KMIN(B.IstrM, B.IstrB) + 1, B.JendB - B.JstrB + 1, 1, c->stream, m);
  }
  if (c->G.obc && !keep) { int r = run_obc2d(c, kstp, 1); if (r) return r; }
  launch_halo(c, lev2d(c, c->F.zeta, kstp), 1, (c->G.obc || keep) ? BC_NONE : (bc_rstate(c) | (c->G.wet_dry ? BC_WET2 : 0)), 'r');   // zetabc_tile + exchange
  KArgs a = mk(c, kstp);
  LAUNCH_THREAD(k_copy_zt, B.IendT - B.IstrT + 1, B.JendT - B.JstrT + 1, 1, c->stream, a);
  launch_halo(c, c->F.Zt_avg1, 1, BC_NONE, 'r');
  return 0;
}

int run_ini_fields(roms_hip_ctx *c) {
  const TB &B = c->G.T;
  const int N = c->G.N, nstp = c->G.nstp, kstp = c->G.kstp;
  if (c->G.masking) {
    KArgs m = mk(c, 1);
    LAUNCH_THREAD(k_ini_mask, B.IendB - KMIN(B.IstrM, B.IstrB) + 1, B.JendB - B.JstrB + 1, N, c->stream, m);
  }
  if (c->G.obc) { int r = run_obc3d_uv(c, nstp); if (r) return r; }
  const HaloSpec hs8[] = {
      {uv_lev(c, c->F.u, nstp), N, obc_bc(c, BC_U) | (c->G.wet_dry ? BC_WET3 : 0), 'u'},     // u3dbc_tile + exchange_u3d
      {uv_lev(c, c->F.v, nstp), N, obc_bc(c, BC_V) | (c->G.wet_dry ? BC_WET3 : 0), 'v'},
  };
  launch_halo_multi(c, hs8, 2);
  KArgs a = mk(c);
  const int i0 = KMIN(B.IstrM, B.IstrB);
  LAUNCH_THREAD(k_ini_bar, B.IendB - i0 + 1, B.JendB - B.JstrB + 1, 1, c->stream, a);
  {
    // not with radiation or Flather conditions on the barotropic momentum (ini_fields.F:412-425)
    bool keep = false;
    for (int e = 0; e < 4; e++)
      for (int v = ROMS_ISUBAR; v <= ROMS_ISVBAR; v++) {
        const int k = lbc_kind(c->cfg, e, v);
        keep |= k == ROMS_LBC_RAD || k == ROMS_LBC_RADNUD || k == ROMS_LBC_FLA;
      }
    if (c->G.obc && !keep) { int r = run_obc2d(c, kstp, 6); if (r) return r; }
    const HaloSpec hs9[] = {
        {lev2d(c, c->F.ubar, kstp), 1, (c->G.obc || keep) ? BC_NONE : (BC_U | (c->G.wet_dry ? BC_WET2 : 0)), 'u'},   // u2dbc_tile + exchange
        {lev2d(c, c->F.vbar, kstp), 1, (c->G.obc || keep) ? BC_NONE : (BC_V | (c->G.wet_dry ? BC_WET2 : 0)), 'v'},
    };
    launch_halo_multi(c, hs9, 2);
  }
  if (c->G.masking) {
    KArgs m = mk(c, 2);
    LAUNCH_THREAD(k_ini_mask, B.IendB - KMIN(B.IstrM, B.IstrB) + 1, B.JendB - B.JstrB + 1, N * c->G.NT, c->stream, m);
  }
  {
    HaloSpec ht[ROMS_MAXT];
    if (c->G.obc) for (int it = 1; it <= c->G.NT; it++) { int r = run_obc3d_t(c, nstp, it); if (r) return r; }
    for (int it = 1; it <= c->G.NT; it++) ht[it - 1] = HaloSpec{t_lev(c, nstp, it), N, obc_bc(c, bc_rstate(c)), 'r'};   // t3dbc + exchange
    launch_halo_multi(c, ht, c->G.NT);
  }
  return 0;
}

// copy `planes` horizontal planes wrk3[1] -> wrk3[2] `reps` times; the caller times it (kprof / events)
int run_copy_probe(roms_hip_ctx *c, int reps) {
  CopyArgs a;
  a.src = c->F.wrk3[1];
  a.dst = c->F.wrk3[2];
  a.n = (long)c->G.nij * (long)(c->G.N + 1);
  for (int r = 0; r < reps; r++) LAUNCH_THREAD(k_copy_probe, 64 * 4096, 1, 1, c->stream, a);
  return 0;
}
