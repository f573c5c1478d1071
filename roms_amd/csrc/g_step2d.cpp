// g_step2d.cpp -- step2d(ng,tile): one k_step2d launch + one halo launch.
#include "roms_host.h"
#include <cstdlib>
#include <cstring>
#include "k_step2d.h"
#include "k_step2d_pair.h"
#include "k_step2d_loop.h"

// (Re)build the packed metric records after the grid arrays were uploaded.  A multi-tile context that runs the pair
// kernel first fills the WIDE ghost lines of the time-invariant 2-D fields the barotropic kernels read (the caller's
// windows carry 3 | Nghost lines): every rank reaches its first step2d call with m2d_dirty set, so the exchanges match.
static void pack_metrics(roms_hip_ctx *c) {
  const DGrid &G = c->G;
  if (c->static_wide_dirty && c->pair_mt) {
    Fields &F = c->F;
    const HaloSpec a[8] = {{F.h, 1, BC_NONE, 'r'}, {F.pm, 1, BC_NONE, 'r'}, {F.pn, 1, BC_NONE, 'r'}, {F.on_u, 1, BC_NONE, 'u'},
                           {F.om_v, 1, BC_NONE, 'v'}, {F.fomn, 1, BC_NONE, 'r'}, {F.dndx, 1, BC_NONE, 'r'}, {F.dmde, 1, BC_NONE, 'r'}};
    const HaloSpec b[8] = {{F.visc2_r, 1, BC_NONE, 'r'}, {F.pmon_r, 1, BC_NONE, 'r'}, {F.pnom_r, 1, BC_NONE, 'r'}, {F.on_r, 1, BC_NONE, 'r'},
                           {F.om_r, 1, BC_NONE, 'r'}, {F.visc2_p, 1, BC_NONE, 'p'}, {F.pmon_p, 1, BC_NONE, 'p'}, {F.pnom_p, 1, BC_NONE, 'p'}};
    const HaloSpec d[6] = {{F.om_p, 1, BC_NONE, 'p'}, {F.on_p, 1, BC_NONE, 'p'}, {F.rmask, 1, BC_NONE, 'r'}, {F.umask, 1, BC_NONE, 'u'},
                           {F.vmask, 1, BC_NONE, 'v'}, {F.pmask, 1, BC_NONE, 'p'}};
    launch_halo_wide(c, a, 8);
    launch_halo_wide(c, b, 8);
    launch_halo_wide(c, d, G.masking ? 6 : 2);
  }
  c->static_wide_dirty = false;
  PackArgs pa;
  pa.G = G;
  pa.Fv = c->F;
  LAUNCH_THREAD(k_pack_m2d, G.ni, G.nj, 1, c->stream, pa);
  c->m2d_dirty = false;
}

int run_step2d(roms_hip_ctx *c) {
  const DGrid &G = c->G;
  const roms_hip_config &cf = c->cfg;
  Step2dArgs a;
  a.G = G;
  S2F_FILL(a.F, c->F);
  const int iif = G.iif;
  a.w1_m1 = (iif >= 2) ? cf.weight[0][iif - 1] : 0.0;
  a.w1_0 = cf.weight[0][iif];
  a.w2_0 = cf.weight[1][iif];
  a.w2_p1 = (iif + 1 <= ROMS_MAXW) ? cf.weight[1][iif + 1] : 0.0;
  // behind a pair launch the krhs level is still staged: read it there and commit it (k_step2d_pair.h)
  a.lev_in = c->b2_stage ? c->b2_stage : G.krhs;
  a.commit = c->b2_stage ? 1 : 0;
  a.om_u = c->F.om_u; a.on_v = c->F.on_v; a.ubarclm = c->F.ubarclm; a.vbarclm = c->F.vbarclm; a.M2nudgcof = c->F.M2nudgcof;
  if (c->b2_stage && !(G.predictor && G.krhs != 3)) { set_error("step2d: a staged pair result can only be followed by a predictor call"); return 8; }
  c->b2_stage = 0;
  if (c->m2d_dirty) pack_metrics(c);
  // kernel variant by sub-tile size: up to 32x4, up to 64x8, generic (ROMS_HIP_TILE2D overrides)
  int variant = (G.bw2 <= 32 && G.bh2 <= 4) ? 0 : (G.bw2 <= 32 && G.bh2 <= 8) ? 3 : (G.bw2 <= 64 && G.bh2 <= 8) ? 1 : 2;
  if (getenv("ROMS_HIP_S2D_GENERIC") || (G.masking && variant != 0)) variant = 2;   // (masks: k_step2d_am or the generic form)
  if (a.commit && (variant != 0 || G.masking)) variant = 2;       // behind a pair launch: k_step2d_ac or the generic form commit the staged level
  if (G.wet_dry) {                                                // WET_DRY: the masks of this call first (step2d_LF_AM3.h:863; behind the last fast
    variant = 2;                                                  // step: after the averages, below); the generic form carries the branches (k_step2d_wd)
    if (iif <= G.nfast) { int r = run_wetdry(c, 0); if (r) return r; }
  }
  if (G.dia_uv || G.uv_vis4 || (G.clima & 32) || G.volcons) variant = 2;                         // (UV_VIS4: the generic form carries the biharmonic block, k_step2d_vis4)                                      // DIAGNOSTICS_UV: the generic form carries the term stores (k_step2d_duv)
  // 64x8 sub-tiles: 1024 threads (one rectangle point and one momentum point per thread) measure 3 %
  // faster than 512 threads with two each; ROMS_HIP_S2D_1024=0 selects the latter
  const char *e1024 = getenv("ROMS_HIP_S2D_1024");
  const bool wide = variant == 1 && !(e1024 && e1024[0] == '0');
#ifdef ROMS_CPU_EMU
  variant = 2;   // the serial emulation has one "thread": only the generic form applies
#endif
  const int tw = variant == 0 || variant == 3 ? 38 : variant == 1 ? 70 : G.bw2 + 6;
  const int th = variant == 0 ? 10 : variant == 1 || variant == 3 ? 14 : G.bh2 + 6;
  const size_t lds = (size_t)(STEP2D_NLDS + (G.uv_vis4 ? 2 : 0)) * (size_t)tw * (size_t)th;   // (+ LapU, LapV)
  if (variant == 0) {
    if (G.masking) LAUNCH_COOP_AS(k_step2d, k_step2d_am, G.nbx2, G.nby2, 1, 384, lds, c->stream, a);
    else if (a.commit) LAUNCH_COOP_AS(k_step2d, k_step2d_ac, G.nbx2, G.nby2, 1, 384, lds, c->stream, a);
    else LAUNCH_COOP_AS(k_step2d, k_step2d_a, G.nbx2, G.nby2, 1, 384, lds, c->stream, a);
  } else if (variant == 3) {
    LAUNCH_COOP_AS(k_step2d, k_step2d_c, G.nbx2, G.nby2, 1, 512, lds, c->stream, a);
  } else if (variant == 1) {
#ifndef ROMS_CPU_EMU
    static bool big_lds = false;
    if (!big_lds) {   // 118 KB: more than the default dynamic LDS limit
      if (hipFuncSetAttribute((const void *)k_step2d_b, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
        set_error("k_step2d: cannot raise the dynamic LDS limit");
        return 2;
      }
      big_lds = true;
    }
#endif
    if (wide) {
#ifndef ROMS_CPU_EMU
      static bool big_lds_d = false;
      if (!big_lds_d) {
        if (hipFuncSetAttribute((const void *)k_step2d_d, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
          set_error("k_step2d: cannot raise the dynamic LDS limit");
          return 2;
        }
        big_lds_d = true;
      }
#endif
      LAUNCH_COOP_AS(k_step2d, k_step2d_d, G.nbx2, G.nby2, 1, 1024, lds, c->stream, a);
    } else {
      LAUNCH_COOP_AS(k_step2d, k_step2d_b, G.nbx2, G.nby2, 1, 512, lds, c->stream, a);
    }
  } else {
    int nthreads = 512;
    if (getenv("ROMS_HIP_S2D_THREADS")) nthreads = atoi(getenv("ROMS_HIP_S2D_THREADS"));
    nthreads = KMAX(64, KMIN(512, nthreads / 64 * 64));   // the kernel is compiled for at most 512 threads
#ifndef ROMS_CPU_EMU
    static bool big_lds_g = false;
    if (lds * sizeof(double) > 64 * 1024 && !big_lds_g) {
      if (hipFuncSetAttribute((const void *)k_step2d, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
        set_error("k_step2d: cannot raise the dynamic LDS limit");
        return 2;
      }
      big_lds_g = true;
    }
#endif
    if (G.uv_vis4) {
#ifndef ROMS_CPU_EMU
      static bool big_lds_4 = false;
      if (lds * sizeof(double) > 64 * 1024 && !big_lds_4) {
        if (hipFuncSetAttribute((const void *)k_step2d_vis4, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
          set_error("k_step2d: cannot raise the dynamic LDS limit");
          return 2;
        }
        big_lds_4 = true;
      }
#endif
      LAUNCH_COOP_AS(k_step2d, k_step2d_vis4, G.nbx2, G.nby2, 1, nthreads, lds, c->stream, a);
    } else if (G.dia_uv) {
#ifndef ROMS_CPU_EMU
      static bool big_lds_u = false;
      if (lds * sizeof(double) > 64 * 1024 && !big_lds_u) {
        if (hipFuncSetAttribute((const void *)k_step2d_duv, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
          set_error("k_step2d: cannot raise the dynamic LDS limit");
          return 2;
        }
        big_lds_u = true;
      }
#endif
      LAUNCH_COOP_AS(k_step2d, k_step2d_duv, G.nbx2, G.nby2, 1, nthreads, lds, c->stream, a);
    } else if (G.clima & 32) {
      if (lds * sizeof(double) > 64 * 1024) { set_error("k_step2d_ndg: the sub-tile needs more than 64 KB of LDS (ROMS_HIP_TILE2D)"); return 5; }
      LAUNCH_COOP_AS(k_step2d, k_step2d_ndg, G.nbx2, G.nby2, 1, nthreads, lds, c->stream, a);
    } else if (G.wet_dry) {
#ifndef ROMS_CPU_EMU
      static bool big_lds_w = false;
      if (lds * sizeof(double) > 64 * 1024 && !big_lds_w) {
        if (hipFuncSetAttribute((const void *)k_step2d_wd, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
          set_error("k_step2d: cannot raise the dynamic LDS limit");
          return 2;
        }
        big_lds_w = true;
      }
#endif
      LAUNCH_COOP_AS(k_step2d, k_step2d_wd, G.nbx2, G.nby2, 1, nthreads, lds, c->stream, a);
    } else
    LAUNCH_COOP(k_step2d, G.nbx2, G.nby2, 1, nthreads, lds, c->stream, a);
  }
  if (G.fuse_halo) return 0;   // the kernel filled the boundary and periodic ghost points itself
  if (iif == G.nfast + 1 && G.predictor) {
    // final fast-time averages :821-883
    HaloSpec sp[3] = {{c->F.Zt_avg1, 1, BC_NONE, 'r'}, {c->F.DU_avg1, 1, BC_NONE, 'u'}, {c->F.DV_avg1, 1, BC_NONE, 'v'}};
    launch_halo_multi(c, sp, 3);
    if (G.wet_dry) { int r = run_wetdry(c, 1); if (r) return r; }     // the time-averaged masks of the 3-D step (wetdry.F:250-349)
  }
  if (iif > G.nfast) return 0;
  if (G.obc) { int r = run_obc2d(c, G.knew); if (r) return r; }                  // zetabc, u2dbc, v2dbc with open edges (k_obc.h)
  if (G.volcons) { int r = run_obc_flux(c, G.knew); if (r) return r; }          // VolCons: obc_flux_tile :2885
  HaloSpec sp[8];
  int n = 0;
  const int wet = G.wet_dry ? BC_WET2 : 0;                                             // (+ the wetting/drying conditions at the end of zetabc / u2dbc / v2dbc)
  sp[n++] = {lev2d(c, c->F.zeta, G.knew), 1, obc_bc(c, bc_rstate(c)) | wet, 'r'};       // zetabc :1057 + exchange :1068
  if (G.predictor) sp[n++] = {lev2d(c, c->F.rzeta, G.krhs), 1, BC_NONE, 'r'};   // :1030
  sp[n++] = {lev2d(c, c->F.ubar, G.knew), 1, obc_bc(c, BC_U) | wet, 'u'};       // u2dbc :2871 + exchange :3043
  sp[n++] = {lev2d(c, c->F.vbar, G.knew), 1, obc_bc(c, BC_V) | wet, 'v'};       // v2dbc :2876
  if (c->pair_on && c->pair_mt && !G.predictor && iif == 1 && G.nfast >= 2) {
    // the pairs follow (k_step2d_pair.h): this exchange carries the wide strips, and with them what the first pair reads
    // on its rim besides the level just written -- its kstp level (this call's kstp) and the fast-time-constant forcing
    // the first predictor left in rufrc, rvfrc
    sp[n++] = {lev2d(c, c->F.zeta, G.kstp), 1, BC_NONE, 'r'};
    sp[n++] = {lev2d(c, c->F.ubar, G.kstp), 1, BC_NONE, 'u'};
    sp[n++] = {lev2d(c, c->F.vbar, G.kstp), 1, BC_NONE, 'v'};
    sp[n++] = {c->F.rufrc, 1, BC_NONE, 'u'};
    sp[n++] = {c->F.rvfrc, 1, BC_NONE, 'v'};
    launch_halo_wide(c, sp, n);
    return 0;
  }
  launch_halo_multi(c, sp, n);
  return 0;
}


// ---- predictor + corrector of one fast step as one launch (k_step2d_pair.h) ----------------------------------
static size_t pair_lds_doubles(int bw2, int bh2) { return (size_t)S2P_NLDS * (size_t)(bw2 + 2 * S2P_RIM) * (size_t)(bh2 + 2 * S2P_RIM); }
// ROMS_HIP_PAIR=0/1 forces; default: tiles whose barotropic sub-tiles are the 32x4 ones (up to 64 K points)
bool step2d_pair_usable(const roms_hip_ctx *c) {
  const DGrid &G = c->G;
  const char *e = getenv("ROMS_HIP_PAIR");
  if (e && e[0] == '0') return false;
  if (G.obc) return false;                                               // open boundaries: zetabc/u2dbc/v2dbc between the two calls (k_obc.h)
  if (G.dia_uv) return false;                                            // DIAGNOSTICS_UV: the per-call kernel carries the term stores
  if (G.uv_vis4) return false;                                           // UV_VIS4: the per-call generic kernel carries the biharmonic block
  if (G.wet_dry) return false;                                           // WET_DRY: new masks in front of every call
  if (G.clima & 32) return false;                                        // LnudgeM2CLM: the per-call kernel carries the nudging term
  const int LmT = G.T.Iend - G.T.Istr + 1, MmT = G.T.Jend - G.T.Jstr + 1;
  if (LmT < 8 || MmT < 8) return false;                                  // the rim (5 | 4 lines) comes from the neighbour's / the tile's own points
  if (pair_lds_doubles(G.bw2, G.bh2) * sizeof(double) > 160 * 1024) return false;
  if (c->has_exchange && !c->pair_mt) return false;                      // multi-tile: needs the wide ghost zone (roms_hip_create)
  if (e && e[0] == '1') return true;
  return G.bw2 <= 32 && G.bh2 <= 4;
}

#ifndef ROMS_CPU_EMU
// my rim planes and the neighbours' as mapped here; my point (i,j) in neighbour d's planes is (i + sx, j + sy) of ITS arrays, sx / sy
// the shift across the periodic seam where my tile lies on that domain edge (k_step2d_pair.h: S2LPeer)
static void fill_peer(roms_hip_ctx *c, S2LPeer &P) {
  const TileComm &m = c->comm;
  const roms_hip_config &cf = c->cfg;
  const DGrid &G = c->G;
  static const int ddx[8] = {-1, 1, 0, 0, -1, 1, -1, 1}, ddy[8] = {0, 0, -1, 1, -1, -1, 1, 1};
  memset(&P, 0, sizeof(P));
  P.on = 1;
  P.rim = (kword_t *)((char *)m.peer_slab + m.loop_rim_off);
  for (int d = 0; d < 8; d++) {
    if (m.nbr[d] < 0) continue;
    const TileComm::PeerGeom &g = m.ngeom[d];
    P.nbmask |= 1 << d;
    P.nrim[d] = (kword_t *)((char *)m.peer_map[d] + g.rim_off);
    const int sx = ddx[d] < 0 && cf.west_edge ? G.Lm : (ddx[d] > 0 && cf.east_edge ? -G.Lm : 0);
    const int sy = ddy[d] < 0 && cf.south_edge ? G.Mm : (ddy[d] > 0 && cf.north_edge ? -G.Mm : 0);
    P.nni[d] = g.ni; P.nnij[d] = g.ni * g.nj;
    P.noff[d] = (sx - g.LBi) + (sy - g.LBj) * g.ni;
  }
}
// Multi-tile contexts whose tiles are too large for the persistent loop (more sub-tiles than compute units: the 1024x64 tiles of
// BASELINE's 8-GPU partition): the pair launches hand the corrector's result across the tile edges themselves -- published into
// the neighbours' rim planes at the end of a launch, polled for at the start of the next (k_step2d_pair.h: rim_in / rim_out) --
// instead of one exchange launch behind every pair.  A launch waits only for the neighbours' PREVIOUS launch, so nothing has to be
// resident at the same time; ranks that share a device still keep the exchanges (a device full of waiting blocks would keep the
// neighbour's launch from starting).  Same conditions on the partition as the loop: every rank decides alike.
// ROMS_HIP_PAIR_RIM=0/1 forces.
static bool pair_rim_usable(roms_hip_ctx *c) {
  if (c->rim_refused) return false;                   // (roms_hip_rim_disable: the self-check of the rim planes failed on some rank)
  if (c->pair_rim_state) return c->pair_rim_state > 0;
  c->pair_rim_state = -1;
  const TileComm &m = c->comm;
  const roms_hip_config &cf = c->cfg;
  const char *e = getenv("ROMS_HIP_PAIR_RIM");        // (decided once per context)
  if (e && e[0] == '0') return false;
  if (!c->has_exchange || !c->pair_mt || !c->pair_on || !m.peer_on || !m.loop_rim_off || c->G.obc) return false;
  if (m.peer_shared && !(e && e[0] == '1')) return false;
  if (cf.Lm % cf.NtileI || cf.Mm % cf.NtileJ) return false;
  for (int d = 0; d < 8; d++)
    if (m.nbr[d] >= 0 && !m.ngeom[d].rim_off) return false;
  if (step2d_loop_usable(c)) return false;
  hipDeviceProp_t prop;
  int d0 = 0;
  if (hipGetDevice(&d0) != hipSuccess || hipGetDeviceProperties(&prop, d0) != hipSuccess) return false;
  if (strncmp(prop.gcnArchName, "gfx942", 6) && strncmp(prop.gcnArchName, "gfx950", 6)) return false;
  if (!c->loop_err) {
    if (hipHostMalloc((void **)&c->loop_err, sizeof(unsigned long long), hipHostMallocDefault) != hipSuccess) { c->loop_err = nullptr; return false; }
    *c->loop_err = 0;
  }
  c->pair_rim_state = 1;
  return true;
}
#endif

int run_step2d_pair(roms_hip_ctx *c) {
  const DGrid &G = c->G;
  const roms_hip_config &cf = c->cfg;
  const int iif = G.iif;
  if (!G.predictor || iif < 2 || iif > G.nfast || G.knew != 3 || G.krhs == 3) { set_error("step2d_pair: needs the stepping of a predictor call, 2 <= iif <= nfast"); return 8; }
  if (G.dia_uv) { set_error("step2d_pair: the momentum diagnostics (DIAGNOSTICS_UV) are carried by the per-call kernel"); return 8; }
  Step2dPairArgs a;
  a.G = G;
  S2F_FILL(a.F, c->F);
  a.w1_m1 = cf.weight[0][iif - 1];
  a.w2_0 = cf.weight[1][iif];
  a.w2_p1 = (iif + 1 <= ROMS_MAXW) ? cf.weight[1][iif + 1] : 0.0;
  a.lev_in = c->b2_stage ? c->b2_stage : G.krhs;
  a.commit = c->b2_stage ? 1 : 0;
  a.lev_out = c->b2_stage == 4 ? 5 : 4;
  a.wrapx = G.ewp && G.xloc;
  a.wrapy = G.nsp && G.yloc;
  a.tail = G.nfast - iif;
  a.rim_in = a.rim_out = 0; a.tag_in = a.tag_out = 0; a.err = nullptr; a.timeout = 0;
  memset(&a.P, 0, sizeof(a.P));
#ifndef ROMS_CPU_EMU
  if (pair_rim_usable(c)) {
    // (the last two pairs keep their exchanges: they carry what the loop leaves behind -- level 3, rzeta of both levels)
    static const double tmo = getenv("ROMS_HIP_LOOP_TIMEOUT") ? atof(getenv("ROMS_HIP_LOOP_TIMEOUT")) : 2.0;
    fill_peer(c, a.P);
    a.err = c->loop_err; a.timeout = (long long)(tmo * 1e8);
    a.rim_in = (a.commit && c->b2_rim) ? 1 : 0; a.tag_in = c->pair_epoch;
    a.rim_out = a.tail > 1 ? 1 : 0;
    if (a.rim_out) a.tag_out = ++c->pair_epoch;
  }
#endif
  if (c->m2d_dirty) pack_metrics(c);
  const bool fixed = G.bw2 <= 32 && G.bh2 <= 4 && !G.masking && !getenv("ROMS_HIP_S2D_GENERIC");
#ifdef ROMS_CPU_EMU
  const size_t lds = pair_lds_doubles(G.bw2, G.bh2);
  LAUNCH_COOP(k_step2d_pair, G.nbx2, G.nby2, 1, 1, lds, c->stream, a);
#else
  static bool big_a = false, big_g = false;
  if (fixed) {
    const size_t lds = pair_lds_doubles(32, 4);
    if (!big_a) {
      if (hipFuncSetAttribute((const void *)k_step2d_pair_a, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
        set_error("k_step2d_pair: cannot raise the dynamic LDS limit");
        return 2;
      }
      big_a = true;
    }
    LAUNCH_COOP_AS(k_step2d_pair, k_step2d_pair_a, G.nbx2, G.nby2, 1, 640, lds, c->stream, a);
  } else {
    const size_t lds = pair_lds_doubles(G.bw2, G.bh2);
    if (!big_g) {
      if (hipFuncSetAttribute((const void *)k_step2d_pair, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
        set_error("k_step2d_pair: cannot raise the dynamic LDS limit");
        return 2;
      }
      big_g = true;
    }
    LAUNCH_COOP(k_step2d_pair, G.nbx2, G.nby2, 1, 512, lds, c->stream, a);
  }
#endif
  c->b2_stage = a.lev_out;
  c->b2_rim = a.rim_out != 0;
  if (G.fuse_halo) return 0;        // the kernel stored the boundary values and periodic images itself
  if (a.rim_out) return 0;          // ... or handed its rim to the neighbouring ranks itself (pair_rim_usable)
  // closed basin: zetabc / u2dbc / v2dbc of the corrector's result (staged); the predictor's level 3 and rzeta(krhs) as
  // the per-call sequence leaves them
  const size_t nij = (size_t)G.nij;
  HaloSpec sp[7];
  int n = 0;
  sp[n++] = {c->F.zeta + (size_t)(a.lev_out - 1) * nij, 1, bc_rstate(c), 'r'};
  sp[n++] = {c->F.ubar + (size_t)(a.lev_out - 1) * nij, 1, BC_U, 'u'};
  sp[n++] = {c->F.vbar + (size_t)(a.lev_out - 1) * nij, 1, BC_V, 'v'};
  if (iif >= G.nfast - 1) {
    sp[n++] = {lev2d(c, c->F.zeta, 3), 1, bc_rstate(c), 'r'};
    sp[n++] = {lev2d(c, c->F.rzeta, G.krhs), 1, BC_NONE, 'r'};
    sp[n++] = {lev2d(c, c->F.ubar, 3), 1, BC_U, 'u'};
    sp[n++] = {lev2d(c, c->F.vbar, 3), 1, BC_V, 'v'};
  }
  launch_halo_wide(c, sp, n);       // (multi-tile: strips B2D_GL | B2D_GH lines wide; a closed basin on one tile: boundary fills only)
  return 0;
}


// ---- the fast steps 2 .. nfast as ONE persistent launch (k_step2d_loop.h) --------------------------------------
// ROMS_HIP_LOOP=0/1 forces (1: wherever the kernel is built for); default: on where the pair engine runs 32x4 sub-tiles
// on a single tile with fused boundary fills and every sub-tile gets a compute unit of its own.  The loop has its own
// decomposition of the tile (ROMS_HIP_LOOP_TILE=32x4 | 16x8): the per-call kernels in front of it and behind it leave
// and find the whole state in global memory.
#ifndef ROMS_CPU_EMU
static int loop_shape() {     // 0: 32x4 sub-tiles on 640 threads, 1: 16x8 on 512
  static const char *e = getenv("ROMS_HIP_LOOP_TILE");
  return (e && !strcmp(e, "32x4")) ? 0 : 1;
}
static void loop_grid(const DGrid &G, DGrid &L) {
  L = G;
  const int LmT = G.T.Iend - G.T.Istr + 1, MmT = G.T.Jend - G.T.Jstr + 1;
  const int bw = loop_shape() ? 16 : 32, bh = loop_shape() ? 8 : 4;
  L.nbx2 = KMAX(1, (LmT + bw - 1) / bw); L.nby2 = KMAX(1, (MmT + bh - 1) / bh);
  L.bw2 = (LmT + L.nbx2 - 1) / L.nbx2; L.bh2 = (MmT + L.nby2 - 1) / L.nby2;
  { const char *ex = getenv("ROMS_HIP_S2D_XCD"); L.xmap2 = ((L.nbx2 * L.nby2) % 8 == 0 && L.nbx2 * L.nby2 >= 16 && !(ex && ex[0] == '0')) ? 1 : 0; }
  { static const char *ep = getenv("ROMS_HIP_LOOP_XCD");
    if (L.xmap2 && ep && !strcmp(ep, "patch") && L.nbx2 % 4 == 0 && L.nby2 % 2 == 0 && (L.nbx2 / 4) * (L.nby2 / 2) * 8 == L.nbx2 * L.nby2) L.xmap2 = 2; }
}
static size_t loop_lds_doubles(bool masked = false) { return (size_t)(masked ? S2L_NLDS_MK : S2L_NLDS) * (loop_shape() ? (16 + 2 * S2P_RIM) * (8 + 2 * S2P_RIM) : (32 + 2 * S2P_RIM) * (4 + 2 * S2P_RIM)); }
// A multi-tile context (round 6): the loop crosses the tile edges through the mailbox slab (k_step2d_loop.h, S2LPeer).  Every
// rank must take the same decision, so it depends on the partition, the transport and the environment only: equal tiles
// (the neighbours' sub-tile grids continue mine), the mailbox installed with a loop region on every neighbour, no rank
// sharing its device with another (the kernels of all ranks must be resident at the same time: ROMS_HIP_LOOP=1 forces it
// for test set-ups that know their blocks fit side by side), no open boundaries, the 16x8 shape.
static bool loop_mt_usable(roms_hip_ctx *c, const DGrid &L, bool forced) {
  const DGrid &G = c->G;
  const roms_hip_config &cf = c->cfg;
  const TileComm &m = c->comm;
  if (!c->pair_mt || !m.peer_on || !m.loop_rim_off || c->rim_refused) return false;
  if (m.peer_shared && !forced) return false;
  if (G.obc || !loop_shape()) return false;
  if (cf.Lm % cf.NtileI || cf.Mm % cf.NtileJ) return false;
  if (m.loop_nb2[0] != L.nbx2 || m.loop_nb2[1] != L.nby2) return false;
  const char *ewh = getenv("ROMS_HIP_LOOP_WHOLE");
  if (ewh && ewh[0] == '0') return false;                  // (only the whole loop -- first fast step and auxiliary call inside -- crosses tiles)
  for (int d = 0; d < 8; d++) {
    if (m.nbr[d] < 0) continue;
    const TileComm::PeerGeom &g = m.ngeom[d];
    if (!g.rim_off || g.nbx2 != L.nbx2 || g.nby2 != L.nby2) return false;
  }
  return true;
}
// The launch.  Its blocks wait for each other, so all of them must be resident at once.  hipLaunchCooperativeKernel would make
// the runtime guarantee that (or refuse the launch) -- but the runtime routes cooperative launches through a queue of their
// own and orders it against the stream with barrier packets on both sides: measured on the MI355X (round 6,
// tools/gpu_debug/ab_env.sh ROMS_HIP_LOOP_COOP "1 0"), a BENCHMARK1 step takes 1.53-1.58 ms with it against 0.82 ms with the
// ordinary launch (one-step trace: 70 us of idle queue in front of the kernel, 30 us behind it, and the four streams of the
// schedule serialised against the cooperative queue).  So the ordinary launch stays the default, guarded by the occupancy
// query at set-up (every block resident on an otherwise idle device) and by bounded waits: a launch that meets a busy
// device ends with exit_flag 2 and a message that names ROMS_HIP_LOOP=0 (ctx_check; the step's state is lost, as with any
// other device error).  Ranks
// that share a device (test set-ups) do not take the loop at all; ROMS_HIP_LOOP=0 is the switch for any other shared use.
// ROMS_HIP_LOOP_COOP=1 selects the cooperative launch.
template <class K>
static int loop_launch(roms_hip_ctx *c, K kern, const char *label, const Step2dLoopArgs &a, int nthr, size_t lds_doubles) {
  static const char *ec = getenv("ROMS_HIP_LOOP_COOP");
  const bool coop = ec && ec[0] == '1' && !g_kp_start && !g_kprof_mode;
  const dim3 grid((unsigned)a.G.nbx2, (unsigned)a.G.nby2, 1), block((unsigned)nthr);
  if (coop) {
    void *args[1] = {(void *)&a};
    const hipError_t e = hipLaunchCooperativeKernel((const void *)kern, grid, block, args, lds_doubles * sizeof(double), c->stream);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      set_error(std::string("step2d_loop: the cooperative launch was refused (") + hipGetErrorString(e) +
                "): the device cannot hold every block of the persistent barotropic loop at once -- ROMS_HIP_LOOP=0 runs the pair launches");
      return 2;
    }
    return 0;
  }
  (void)label;
  KPROF_WRAP(k_step2d_loop, c->stream, ROMS_LAUNCH(kern, grid, block, lds_doubles * sizeof(double), c->stream, a));
  return 0;
}
#endif
void step2d_loop_dims(const roms_hip_ctx *c, int &nbx2, int &nby2) {
#ifdef ROMS_CPU_EMU
  (void)c; nbx2 = nby2 = 0;
#else
  DGrid L;
  loop_grid(c->G, L);
  nbx2 = L.nbx2; nby2 = L.nby2;
#endif
}
bool step2d_loop_usable(roms_hip_ctx *c) {
#ifdef ROMS_CPU_EMU
  (void)c;
  return false;       // blocks that wait for each other cannot run one after the other
#else
  if (c->loop_state) return c->loop_state > 0;
  c->loop_state = -1;
  const DGrid &G = c->G;
  const char *e = getenv("ROMS_HIP_LOOP");
  if (e && e[0] == '0') return false;
  if (!c->pair_on) return false;
  // MASKING (round 6): the 16x8 form carries the masked statements (three more LDS tiles); not with WET_DRY (new masks every call)
  if (G.masking && (!loop_shape() || G.wet_dry)) return false;
  if (!c->has_exchange && !G.fuse_halo) return false;
  if (G.bw2 > 32 || G.bh2 > 4 || getenv("ROMS_HIP_S2D_GENERIC")) return false;
  if (c->cfg.nfast < 3) return false;
  DGrid L;
  loop_grid(G, L);
  if (L.bw2 < 2 || L.bh2 < 2) return false;             // (the neighbour window of the kernel: 3 sub-tiles each way)
  if (c->has_exchange && !loop_mt_usable(c, L, e && e[0] == '1')) return false;
  {
    // the rim hand-off relies on what gfx942 / gfx950 do with write-through (sc1) stores, a drained vmcnt and sc1 loads (the
    // MI355X guide's form, outside the HIP memory model): no other architecture takes this kernel
    hipDeviceProp_t prop;
    int d0 = 0;
    if (hipGetDevice(&d0) != hipSuccess || hipGetDeviceProperties(&prop, d0) != hipSuccess) return false;
    if (strncmp(prop.gcnArchName, "gfx942", 6) && strncmp(prop.gcnArchName, "gfx950", 6)) return false;
  }
  // every block must be resident at once: they wait for each other
  const void *kern = c->has_exchange ? (G.masking ? (const void *)k_step2d_loop_bmk : (const void *)k_step2d_loop_bm)
                                     : G.masking ? (const void *)k_step2d_loop_bk : loop_shape() ? (const void *)k_step2d_loop_b : (const void *)k_step2d_loop_a;
  const int nthr = loop_shape() ? 512 : 640;
  if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return false;
  int dev = 0, ncu = 0, per = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, kern, nthr, loop_lds_doubles(G.masking != 0) * sizeof(double)) != hipSuccess) return false;
  if ((long)L.nbx2 * L.nby2 > (long)ncu * per) return false;
  if (c->loop_flags) {                                          // (decided again after a configuration call: the buffers exist)
    c->loop_state = 1;
    if (c->has_exchange && !getenv("ROMS_HIP_XASYNC")) { halo_fence(c, FG_ALL); c->x_async = false; c->rim_split = false; }
    return true;
  }
  void *pf = nullptr, *pw = nullptr;
  if (hipMalloc(&pf, (size_t)L.nbx2 * L.nby2 * S2L_FSTRIDE * sizeof(unsigned)) != hipSuccess) return false;
  c->allocs.push_back(pf);
  if (hipMemset(pf, 0, (size_t)L.nbx2 * L.nby2 * S2L_FSTRIDE * sizeof(unsigned)) != hipSuccess) return false;
  c->loop_epoch = 0;
  if (hipMalloc(&pw, 3 * (size_t)(ROMS_MAXW + 1) * sizeof(double)) != hipSuccess) return false;   // (iif <= nfast+1 <= 2 ndtfast <= ROMS_MAXW)
  c->allocs.push_back(pw);
  if (!c->loop_err) {
    if (hipHostMalloc((void **)&c->loop_err, sizeof(unsigned long long), hipHostMallocDefault) != hipSuccess) { c->loop_err = nullptr; return false; }
    *c->loop_err = 0;
  }
  // the weights of a predictor + corrector pair, by iif = 1 .. nfast+1: weight(1,iif-1), weight(2,iif), weight(2,iif+1)
  std::vector<double> w(3 * (size_t)(ROMS_MAXW + 1), 0.0);
  for (int iif = 1; iif <= c->cfg.nfast + 1 && iif <= ROMS_MAXW; iif++) {
    w[3 * iif] = c->cfg.weight[0][iif - 1];
    w[3 * iif + 1] = c->cfg.weight[1][iif];
    w[3 * iif + 2] = (iif + 1 <= ROMS_MAXW) ? c->cfg.weight[1][iif + 1] : 0.0;
  }
  if (hipMemcpy(pw, w.data(), w.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return false;
  c->loop_flags = (unsigned *)pf;
  c->loop_wts = (double *)pw;
  c->loop_state = 1;
  if (c->has_exchange && !getenv("ROMS_HIP_XASYNC")) {
    // the step is arranged around the loop on four streams (roms_hip.cpp:main3d_around_loop): every exchange stays in the
    // stream of its producer, on that stream's mailbox channel -- the other lanes are what it overlaps with
    halo_fence(c, FG_ALL);
    c->x_async = false;
    c->rim_split = false;
  }
  return true;
#endif
}

// What the first fast step of a multi-tile loop reads beyond the tile (k_step2d_loop.h, prologue), exchanged on the current
// stream -- the schedule around the loop issues both early, off the critical path; run_step2d_loop_n does what is left:
//   1  the 3-D forcing rufrc, rvfrc and its AB3 history ru, rv(:,:,0,nnew | nstp) on the enlarged sub-tiles (two lines)
//   2  the kstp (= krhs) level of zeta, ubar, vbar, 5 | 4 lines wide
int step2d_loop_pre(roms_hip_ctx *c, int what) {
  const DGrid &G = c->G;
  if (what == 1) {
    const int startup = (G.iic == G.ntfirst) ? 0 : ((G.iic == G.ntfirst + 1) ? 1 : 2);
    const size_t o_r0s = (size_t)(G.nstp - 1) * G.nij * (size_t)(G.N + 1), o_r0n = (size_t)(G.nnew - 1) * G.nij * (size_t)(G.N + 1);
    HaloSpec s1[6] = {{c->F.rufrc, 1, BC_NONE, 'u'}, {c->F.rvfrc, 1, BC_NONE, 'v'},
                      {c->F.ru + o_r0n, 1, BC_NONE, 'u'}, {c->F.rv + o_r0n, 1, BC_NONE, 'v'},
                      {c->F.ru + o_r0s, 1, BC_NONE, 'u'}, {c->F.rv + o_r0s, 1, BC_NONE, 'v'}};
    launch_halo_multi(c, s1, startup == 0 ? 2 : (startup == 1 ? 4 : 6));
    c->loop_pre_frc = true;
  } else {
    const int k = c->s.indx1;            // kstp = krhs of the first fast step (main3d.F:824-830)
    HaloSpec s2[3] = {{lev2d(c, c->F.zeta, k), 1, BC_NONE, 'r'}, {lev2d(c, c->F.ubar, k), 1, BC_NONE, 'u'}, {lev2d(c, c->F.vbar, k), 1, BC_NONE, 'v'}};
    launch_halo_wide(c, s2, 3);
    c->loop_pre_state = true;
  }
  return c->comm_failed ? 2 : 0;
}

// whole = 1: the launch starts with the first fast step and ends with the auxiliary call (c->G = the stepping of the predictor
// call of iif = 1); 0: fast steps 2 .. nfast only (c->G = that of iif = 2), the per-call kernel in front and behind
int run_step2d_loop_n(roms_hip_ctx *c, int whole) {
#ifdef ROMS_CPU_EMU
  (void)c; (void)whole;
  set_error("step2d_loop: not part of the emulated build");
  return 8;
#else
  const DGrid &G = c->G;
  if (!step2d_loop_usable(c)) { set_error("step2d_loop: this context does not run the persistent barotropic loop"); return 8; }
  if (!G.predictor || G.iif != (whole ? 1 : 2) || G.knew != 3 || G.krhs == 3 || c->b2_stage) { set_error("step2d_loop: needs the stepping of the predictor call of iif = 2 (of iif = 1 for the whole loop)"); return 8; }
  const bool mt = c->has_exchange;
  if (mt && !whole) { set_error("step2d_loop: a multi-tile context runs the whole loop only (the stepping of the predictor call of iif = 1)"); return 8; }
  if (c->m2d_dirty) pack_metrics(c);
  Step2dLoopArgs a;
  loop_grid(G, a.G);
  S2F_FILL(a.F, c->F);
  a.wts = c->loop_wts;
  a.flags = c->loop_flags;
  a.err = c->loop_err;
  static const double tmo = getenv("ROMS_HIP_LOOP_TIMEOUT") ? atof(getenv("ROMS_HIP_LOOP_TIMEOUT")) : 2.0;    // seconds
  a.timeout = (long long)(tmo * 1e8);                 // wall_clock64: 100 MHz
  a.npairs = whole ? c->cfg.nfast : c->cfg.nfast - 1;
  a.first = whole ? 1 : 0;
  a.aux = whole ? 1 : 0;
  a.nstp = G.nstp; a.nnew = G.nnew;
  a.startup = (G.iic == G.ntfirst) ? 0 : ((G.iic == G.ntfirst + 1) ? 1 : 2);
  {
    const double dtfast = G.dtfast;
    a.kfac = 1000.0 / G.rho0;
    a.kz1 = dtfast * 5.0 / 12.0; a.kz2 = dtfast * 8.0 / 12.0; a.kz3 = dtfast * 1.0 / 12.0;
    a.km1 = 0.5 * dtfast * 5.0 / 12.0; a.km2 = 0.5 * dtfast * 8.0 / 12.0; a.km3 = 0.5 * dtfast * 1.0 / 12.0;
  }
  a.wrapx = G.ewp && G.xloc;
  a.wrapy = G.nsp && G.yloc;
  static const int prio = getenv("ROMS_HIP_LOOP_PRIO") ? atoi(getenv("ROMS_HIP_LOOP_PRIO")) : 3;
  a.prio = prio;
  memset(&a.P, 0, sizeof(a.P));
  if (mt) {
    fill_peer(c, a.P);
    { static const char *ee = getenv("ROMS_HIP_LOOP_EARLY"); a.P.early = ee && ee[0] == '1' ? 1 : 0; }
    if (!c->loop_pre_frc && step2d_loop_pre(c, 1)) return 2;
    if (!c->loop_pre_state && step2d_loop_pre(c, 2)) return 2;
    c->loop_pre_frc = c->loop_pre_state = false;
  }
  // the arrival words count on from launch to launch (a reset would be one more operation in the stream, 5 us + a boundary);
  // zero again long before they wrap (a multi-tile context: its ring is written by the neighbours -- 2^32 pairs are 70 million
  // baroclinic steps of BENCHMARK1: the run ends with exit_flag 8 instead)
  if (c->loop_epoch > 0xF0000000u) {
    if (mt) { set_error("step2d_loop: the arrival counters of a multi-tile context are exhausted"); return 8; }
    const size_t nflag = (size_t)a.G.nbx2 * a.G.nby2 * S2L_FSTRIDE * sizeof(unsigned);
    if (hipMemsetAsync(c->loop_flags, 0, nflag, c->stream) != hipSuccess) { set_error("step2d_loop: hipMemsetAsync"); return 2; }
    c->loop_epoch = 0;
  }
  a.epoch = c->loop_epoch;
  c->loop_epoch += (unsigned)a.npairs;
  const size_t lds = loop_lds_doubles(G.masking != 0);
  int r;
  if (mt && G.masking) r = loop_launch(c, k_step2d_loop_bmk, "k_step2d_loop", a, 512, lds);
  else if (mt) r = loop_launch(c, k_step2d_loop_bm, "k_step2d_loop", a, 512, lds);
  else if (G.masking) r = loop_launch(c, k_step2d_loop_bk, "k_step2d_loop", a, 512, lds);
  else if (loop_shape()) r = loop_launch(c, k_step2d_loop_b, "k_step2d_loop", a, 512, lds);
  else r = loop_launch(c, k_step2d_loop_a, "k_step2d_loop", a, 640, lds);
  if (r) return r;
  c->b2_stage = whole ? 0 : (((a.npairs - 1) & 1) ? 5 : 4);     // the last pair's result: committed by the kernel | staged for the auxiliary call
  if (mt) {
    // what the pair launches exchange between the pairs, once: the three levels of the state (the logical levels 1, 2 and the
    // last predictor's level 3), rzeta of both levels, the fast-time averages -- 5 | 4 lines wide, with the boundary fills of
    // zetabc / u2dbc / v2dbc and the exchanges of step2d_LF_AM3.h:842-883,1030,1068,3041
    HaloSpec sp[7] = {{c->F.zeta, 3, bc_rstate(c), 'r'}, {c->F.ubar, 3, BC_U, 'u'}, {c->F.vbar, 3, BC_V, 'v'}, {c->F.rzeta, 2, BC_NONE, 'r'},
                      {c->F.Zt_avg1, 1, BC_NONE, 'r'}, {c->F.DU_avg1, 1, BC_NONE, 'u'}, {c->F.DV_avg1, 1, BC_NONE, 'v'}};
    launch_halo_wide(c, sp, 7);
    if (c->comm_failed) return 2;
  }
  return 0;
#endif
}
int run_step2d_loop(roms_hip_ctx *c) { return run_step2d_loop_n(c, c->G.iif == 1 ? 1 : 0); }    // (the stepping of the call says which)


// ---- self-check of the rim planes (round 6): before a multi-tile run trusts the rim hand-off inside its barotropic launches
// (k_step2d_loop.h, k_step2d_pair.h: rim_in / rim_out), every rank publishes index-coded values of its own points into the
// neighbours' rim planes -- every direction, corners included, all twelve planes over four repetitions, the sets taking turns
// -- and verifies on the device that every ghost point of its own tile receives the code of the point it images: the address of
// a point in a neighbour's planes (array origin, the shift across a periodic seam), the mapping of the slabs and the visibility
// of tagged 8-byte words across devices, in the geometry of THIS run.  The tags live above 0xC0000000: the pairs of a run
// count up from one and never meet them.  0, or exit_flag 2 with the first wrong point; a context without the mailbox or
// without rim planes returns 0 (nothing to check).
#ifndef ROMS_CPU_EMU
struct RimProbeArgs {
  DGrid G;
  S2LPeer P;
  unsigned long long *bad;     // [0] wrong or missing points, [1] packed (i,j) of the first, [2] its plane, [3] 1 = never arrived
  int rep;
  int skip;                    // test aid (ROMS_HIP_RIM_PROBE_BREAK): this plane is not published -- the check has to fail
  long long timeout;
};
static __device__ __forceinline__ double rim_code(const DGrid &G, int i, int j, int pf, int rep) {
  if (i < 1) i += G.Lm; else if (i > G.Lm) i -= G.Lm;
  if (j < 1) j += G.Mm; else if (j > G.Mm) j -= G.Mm;
  return ((double)(rep & 0xFFFF) * 8.0 + (double)pf) * 67108864.0 + (double)j * 8192.0 + (double)i;
}
static __global__ void k_rim_publish(const RimProbeArgs a, int nx, int ny) {
  const int gx = (int)(blockIdx.x * blockDim.x + threadIdx.x), gy = (int)blockIdx.y;
  if (gx >= nx || gy >= ny) return;
  const DGrid &G = a.G;
  const int i = G.T.Istr + gx, j = G.T.Jstr + gy;
  for (int f = 0; f < 3; f++) {
    const int pf = 3 * (a.rep & 3) + f;
    if (pf == a.skip) continue;
    s2l_rput(G, a.P, pf, i, j, rim_code(G, i, j, pf, a.rep), 0xC0000000u + (unsigned)a.rep);
  }
}
static __global__ void k_rim_verify(const RimProbeArgs a, int nx, int ny) {
  const int gx = (int)(blockIdx.x * blockDim.x + threadIdx.x), gy = (int)blockIdx.y;
  if (gx >= nx || gy >= ny) return;
  const DGrid &G = a.G;
  const TB &T = G.T;
  const int i = G.LBi + gx, j = G.LBj + gy;
  // a ghost point of the tile that images an OWN point of a neighbouring rank (the strips they publish: 5 | 4 lines)
  const int dx = i < T.Istr ? -1 : (i > T.Iend ? 1 : 0), dy = j < T.Jstr ? -1 : (j > T.Jend ? 1 : 0);
  if (dx == 0 && dy == 0) return;
  if ((dx < 0 && i < T.Istr - B2D_GL) || (dx > 0 && i > T.Iend + B2D_GH) || (dy < 0 && j < T.Jstr - B2D_GL) || (dy > 0 && j > T.Jend + B2D_GH)) return;
  const int d = dy == 0 ? (dx < 0 ? 0 : 1) : (dx == 0 ? (dy < 0 ? 2 : 3) : (dy < 0 ? (dx < 0 ? 4 : 5) : (dx < 0 ? 6 : 7)));
  if (!(a.P.nbmask & (1 << d))) return;
  const unsigned tag = 0xC0000000u + (unsigned)a.rep;
  const size_t x = X2(i, j);
  if (ksys_ld((const kword_t *)a.bad)) return;           // (already failed: no more waiting)
  for (int f = 0; f < 3; f++) {
    const int pf = 3 * (a.rep & 3) + f;
    const kword_t *q = a.P.rim + 2 * ((size_t)pf * (size_t)G.nij + x);
    const long long t0 = kclock();
    kword_t w0 = 0, w1 = 0;
    bool here = false;
    for (;;) {
      w0 = ksys_ld(q); w1 = ksys_ld(q + 1);
      if ((unsigned)(w0 >> 32) == tag && (unsigned)(w1 >> 32) == tag) { here = true; break; }
      knap();
      if (kclock() - t0 > a.timeout) break;
    }
    if (!here || s2l_ll_val(w0, w1) != rim_code(G, i, j, pf, a.rep)) {
      if (atomicAdd(a.bad, 1ull) == 0) {
        a.bad[1] = ((unsigned long long)(unsigned)(i + 4096) << 32) | (unsigned)(j + 4096);
        a.bad[2] = (unsigned long long)pf;
        a.bad[3] = here ? 0ull : 1ull;
      }
    }
  }
}
#endif
int run_rim_probe(roms_hip_ctx *c, int reps) {
#ifdef ROMS_CPU_EMU
  (void)c; (void)reps;
  return 0;
#else
  const TileComm &m = c->comm;
  if (!c->has_exchange || !m.peer_on || !m.loop_rim_off || !c->pair_mt) return 0;
  for (int d = 0; d < 8; d++)
    if (m.nbr[d] >= 0 && !m.ngeom[d].rim_off) return 0;
  const DGrid &G = c->G;
  RimProbeArgs a;
  a.G = G;
  fill_peer(c, a.P);
  void *pb = nullptr;
  if (hipMalloc(&pb, 4 * sizeof(unsigned long long)) != hipSuccess) { set_error("rim probe: hipMalloc"); return 2; }
  (void)hipMemsetAsync(pb, 0, 4 * sizeof(unsigned long long), c->stream);
  a.bad = (unsigned long long *)pb;
  const char *brk = getenv("ROMS_HIP_RIM_PROBE_BREAK");
  a.skip = brk && *brk ? atoi(brk) : -1;
  a.timeout = a.skip >= 0 ? (long long)(2.0e7) : (long long)(5.0e8);   // 5 s of the 100 MHz clock: the neighbour may still be setting up
  const int ox = G.T.Iend - G.T.Istr + 1, oy = G.T.Jend - G.T.Jstr + 1;
  for (int rep = 1; rep <= reps; rep++) {
    a.rep = rep;
    hipLaunchKernelGGL(k_rim_publish, dim3((unsigned)((ox + 255) / 256), (unsigned)oy, 1), dim3(256), 0, c->stream, a, ox, oy);
    hipLaunchKernelGGL(k_rim_verify, dim3((unsigned)((G.ni + 255) / 256), (unsigned)G.nj, 1), dim3(256), 0, c->stream, a, G.ni, G.nj);
  }
  unsigned long long bad[4] = {0, 0, 0, 0};
  const bool ok = hipMemcpyAsync(bad, pb, sizeof(bad), hipMemcpyDeviceToHost, c->stream) == hipSuccess && hipStreamSynchronize(c->stream) == hipSuccess;
  (void)hipFree(pb);
  if (!ok) { set_error("rim probe: the device reported an error"); return 2; }
  if (bad[0]) {
    char msg[320];
    snprintf(msg, sizeof(msg), "rim probe: tile %d: %llu ghost points of the rim planes wrong or missing in %d repetitions; first: point (%d,%d), plane %llu, %s",
             c->cfg.tile, bad[0], reps, (int)(bad[1] >> 32) - 4096, (int)(bad[1] & 0xFFFFFFFFu) - 4096, bad[2], bad[3] ? "never arrived" : "wrong value");
    set_error(msg);
    return 2;
  }
  return 0;
#endif
}
// the caller's decision after the probe (every rank the same one): no rim hand-off inside the barotropic launches
void rim_disable(roms_hip_ctx *c) {
  c->rim_refused = true;                 // (sticky: the states below are decided again by later calls, this is not)
  if (c->has_exchange) c->loop_state = 0;
  c->pair_rim_state = 0;
}
