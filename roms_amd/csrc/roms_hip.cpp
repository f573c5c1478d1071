// roms_hip.cpp -- context, memory, C ABI and the main3d step sequence of libroms_hip.so.
//
// Sequencing follows main3d (ROMS/Nonlinear/main3d.F:216-1148) and the LF-AM3 barotropic index
// state machine (:810-918); see roms_hip_main3d below.
#include "roms_host.h"
#include "k_halo.h"
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <string>
#include <utility>

static thread_local std::string g_last_error;
void set_error(const std::string &msg) { g_last_error = msg; }
extern "C" const char *roms_hip_last_error(void) { return g_last_error.c_str(); }
extern "C" int roms_hip_abi_version(void) { return ROMS_HIP_ABI_VERSION; }

// ------------------------------------------------------------------------------ device memory
#ifdef ROMS_CPU_EMU
int g_emu_reverse = 0;
static int dmalloc(void **p, size_t bytes) { *p = calloc(bytes ? bytes : 8, 1); return *p ? 0 : 2; }
static void dfree(void *p) { free(p); }
static int h2d(void *d, const void *h, size_t bytes, kstream_t) { memcpy(d, h, bytes); return 0; }
static int d2h(void *h, const void *d, size_t bytes, kstream_t) { memcpy(h, d, bytes); return 0; }
static int d2d(void *d, const void *s, size_t bytes, kstream_t) { memcpy(d, s, bytes); return 0; }
static int dsync(kstream_t) { return 0; }
static int dpoison(void *p, size_t bytes, kstream_t) { memset(p, 0xFF, bytes); return 0; }
int ctx_check(roms_hip_ctx *c, const char *) { return c->comm_failed ? 2 : 0; }
#else
static int hipfail(hipError_t e, const char *what) {
  if (e == hipSuccess) return 0;
  set_error(std::string(what) + ": " + hipGetErrorString(e));
  return 2;
}
static int dmalloc(void **p, size_t bytes) {
  int r = hipfail(hipMalloc(p, bytes ? bytes : 8), "hipMalloc");
  if (r) return r;
  // The fill runs asynchronously on the null stream, which the contexts' non-blocking streams do
  // not wait for: finish it here, or it may land after a later upload into the same buffer.
  r = hipfail(hipMemset(*p, 0, bytes ? bytes : 8), "hipMemset");
  if (r) return r;
  return hipfail(hipDeviceSynchronize(), "hipDeviceSynchronize");
}
static void dfree(void *p) { (void)hipFree(p); }
static int h2d(void *d, const void *h, size_t bytes, kstream_t s) {
  int r = hipfail(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s), "hipMemcpy H2D");
  if (r) return r;
  return hipfail(hipStreamSynchronize(s), "hipStreamSynchronize");
}
static int d2h(void *h, const void *d, size_t bytes, kstream_t s) {
  int r = hipfail(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s), "hipMemcpy D2H");
  if (r) return r;
  return hipfail(hipStreamSynchronize(s), "hipStreamSynchronize");
}
static int d2d(void *d, const void *src, size_t bytes, kstream_t s) {
  return hipfail(hipMemcpyAsync(d, src, bytes, hipMemcpyDeviceToDevice, s), "hipMemcpy D2D");
}
static int dsync(kstream_t s) { return hipfail(hipStreamSynchronize(s), "hipStreamSynchronize"); }
static int dpoison(void *p, size_t bytes, kstream_t s) { return hipfail(hipMemsetAsync(p, 0xFF, bytes, s), "hipMemsetAsync"); }
int ctx_check(roms_hip_ctx *c, const char *what) {
  if (c->comm_failed) return 2;
  if (c->comm.peer_err && *(volatile unsigned long long *)c->comm.peer_err) {
    set_error("mailbox transport: a neighbour's strips did not arrive in time (exchange " +
              std::to_string(*(volatile unsigned long long *)c->comm.peer_err) + "); ROMS_HIP_PEER_TIMEOUT sets the limit in seconds");
    return 2;
  }
  if (c->loop_err && *(volatile unsigned long long *)c->loop_err) {
    const unsigned long long w = *(volatile unsigned long long *)c->loop_err;
    set_error("barotropic loop (k_step2d_loop): sub-tile " + std::to_string((w & 0xffffffffull) - 1) + " gave up waiting for a neighbouring block in pair " +
              std::to_string(w >> 32) + " (not every block of the launch was resident?); ROMS_HIP_LOOP=0 runs the pair launches, ROMS_HIP_LOOP_TIMEOUT sets the limit in seconds");
    // reported once: the state of this step is lost (exit_flag 2), and the context leaves the persistent loop for good -- a caller
    // that goes on from a restart record in the same context gets the pair launches
    (void)hipDeviceSynchronize();
    *(volatile unsigned long long *)c->loop_err = 0;
    c->loop_state = -1;
    return 2;
  }
  return hipfail(hipGetLastError(), what);
}
#endif

// ------------------------------------------------------------------------------- field table
#define FD(nm, kind) { #nm, offsetof(Fields, nm), kind }
// ROMS_HIP_POISON=1 (test aid, device and emulated build alike): every WORK array -- the private scratch of the reference's
// routines (wrk3, wrk2, the MPDATA arrays, the staging levels of the pair kernel) -- holds NaN (all bits set) from
// roms_hip_create on and again at the start of every roms_hip_main3d step, instead of the zeros of dmalloc: a kernel that
// reads scratch nobody wrote in this step, or scratch another thread of the launch is still writing, shows as NaN in
// the state instead of passing on the zeros the serial emulation happens to hold (tests: test_poisoned_work_arrays).
static bool poison_on() { const char *e = getenv("ROMS_HIP_POISON"); return e && e[0] == '1'; }     // (read per call: a test may switch it between contexts)
struct WorkSpan { void *p; size_t bytes; };
static void work_spans(roms_hip_ctx *c, std::vector<WorkSpan> &out);
static int poison_work(roms_hip_ctx *c) {
  if (!poison_on()) return 0;
  std::vector<WorkSpan> w;
  work_spans(c, w);
#ifndef ROMS_CPU_EMU
  if (hipfail(hipDeviceSynchronize(), "hipDeviceSynchronize")) return 2;    // (side streams of the previous step may still read scratch)
#endif
  static const char *only = getenv("ROMS_HIP_POISON_ONLY");      // (debugging aid: the one span of work_spans' list with this index)
  int q = 0;
  for (const WorkSpan &x : w) {
    if (only && atoi(only) != q++) continue;
    int r = dpoison(x.p, x.bytes, c->stream);
    if (r) return r;
  }
  return dsync(c->stream);
}

static const FieldDesc g_fields[] = {
    FD(h, FK_2D), FD(f, FK_2D), FD(fomn, FK_2D), FD(pm, FK_2D), FD(pn, FK_2D), FD(om_r, FK_2D), FD(on_r, FK_2D),
    FD(om_u, FK_2D), FD(on_u, FK_2D), FD(om_v, FK_2D), FD(on_v, FK_2D), FD(om_p, FK_2D), FD(on_p, FK_2D),
    FD(omn, FK_2D), FD(pmon_r, FK_2D), FD(pnom_r, FK_2D), FD(pmon_p, FK_2D), FD(pnom_p, FK_2D), FD(pmon_u, FK_2D),
    FD(pnom_u, FK_2D), FD(pmon_v, FK_2D), FD(pnom_v, FK_2D), FD(dmde, FK_2D), FD(dndx, FK_2D), FD(angler, FK_2D),
    FD(xr, FK_2D), FD(yr, FK_2D), FD(xp, FK_2D), FD(yp, FK_2D), FD(lonr, FK_2D), FD(latr, FK_2D), FD(rdrag, FK_2D), FD(rdrag2, FK_2D),
    FD(rmask, FK_2D), FD(umask, FK_2D), FD(vmask, FK_2D), FD(pmask, FK_2D),
    FD(Hz, FK_R), FD(z_r, FK_R), FD(z_w, FK_W), FD(Huon, FK_R), FD(Hvom, FK_R),
    FD(zeta, FK_2Dx3), FD(ubar, FK_2Dx3), FD(vbar, FK_2Dx3), FD(rzeta, FK_2Dx2), FD(rubar, FK_2Dx2),
    FD(rvbar, FK_2Dx2), FD(u, FK_Rx2), FD(v, FK_Rx2), FD(t, FK_T), FD(W, FK_W), FD(wvel, FK_W), FD(rho, FK_R),
    FD(pden, FK_R), FD(ru, FK_Wx2), FD(rv, FK_Wx2),
    FD(rhoA, FK_2D), FD(rhoS, FK_2D), FD(rufrc, FK_2D), FD(rvfrc, FK_2D), FD(Zt_avg1, FK_2D), FD(DU_avg1, FK_2D),
    FD(DU_avg2, FK_2D), FD(DV_avg1, FK_2D), FD(DV_avg2, FK_2D),
    FD(sustr, FK_2D), FD(svstr, FK_2D), FD(bustr, FK_2D), FD(bvstr, FK_2D), FD(stflx, FK_2DxNT),
    FD(btflx, FK_2DxNT), FD(stflux, FK_2DxNT), FD(btflux, FK_2DxNT), FD(srflx, FK_2D),
    FD(Uwind, FK_2D), FD(Vwind, FK_2D), FD(Tair, FK_2D), FD(Pair, FK_2D), FD(Hair, FK_2D), FD(rain, FK_2D),
    FD(cloud, FK_2D), FD(lhflx, FK_2D), FD(shflx, FK_2D), FD(lrflx, FK_2D), FD(evap, FK_2D),
    FD(Akv, FK_W), FD(Akt, FK_WxNAT), FD(visc2_r, FK_2D), FD(visc2_p, FK_2D), FD(diff2, FK_2DxNT), FD(bvf, FK_W),
    FD(alpha, FK_2D), FD(beta, FK_2D), FD(hsbl, FK_2D), FD(ghats, FK_WxNAT),
    FD(tke, FK_Wx3), FD(gls, FK_Wx3), FD(Lscale, FK_W), FD(Akk, FK_W), FD(Akp, FK_W),
    FD(sc_r, FK_TABR), FD(Cs_r, FK_TABR), FD(sc_w, FK_TABW), FD(Cs_w, FK_TABW),
    // wvelocity's result at the output point of a step (roms_hip_output_point); download only
    {"w_out", offsetof(Fields, wrk3) + 12 * sizeof(GPtr), FK_W},
    // open-boundary data (mod_boundary.F), inputs like the forcing
#define FB(nm, idx, kind) {nm, offsetof(Fields, bry) + (idx) * sizeof(GPtr), kind}
    FB("zeta_west", 0, FK_BJ), FB("zeta_east", 1, FK_BJ), FB("zeta_south", 2, FK_BI), FB("zeta_north", 3, FK_BI),
    FB("ubar_west", 4, FK_BJ), FB("ubar_east", 5, FK_BJ), FB("ubar_south", 6, FK_BI), FB("ubar_north", 7, FK_BI),
    FB("vbar_west", 8, FK_BJ), FB("vbar_east", 9, FK_BJ), FB("vbar_south", 10, FK_BI), FB("vbar_north", 11, FK_BI),
    FB("u_west", 12, FK_BJN), FB("u_east", 13, FK_BJN), FB("u_south", 14, FK_BIN), FB("u_north", 15, FK_BIN),
    FB("v_west", 16, FK_BJN), FB("v_east", 17, FK_BJN), FB("v_south", 18, FK_BIN), FB("v_north", 19, FK_BIN),
    FB("t_west", 20, FK_BJT), FB("t_east", 21, FK_BJT), FB("t_south", 22, FK_BIT), FB("t_north", 23, FK_BIT),
#undef FB
    // (arrays of later rounds at the END of the table: the table's order is the order of allocation, and the layout of the
    // arrays of the hot path relative to one another is part of what profiles/ measured)
    FD(visc4_r, FK_2D), FD(visc4_p, FK_2D), FD(diff4, FK_2DxNT),                     // UV_VIS4 / TS_DIF4
    FD(rmask_wet, FK_2D), FD(umask_wet, FK_2D), FD(vmask_wet, FK_2D), FD(pmask_wet, FK_2D), FD(rmask_full, FK_2D), FD(umask_full, FK_2D),
    FD(vmask_full, FK_2D), FD(pmask_full, FK_2D), FD(rmask_wet_avg, FK_2D),      // WET_DRY (wetdry.F)
    // climatology nudging (mod_clima.F; option bits ROMS_NUDGE_M3CLM, ROMS_NUDGE_TCLM): tclm, Tnudgcof hold N planes per tracer
    FD(tclm, FK_RxNT), FD(Tnudgcof, FK_RxNT), FD(uclm, FK_R), FD(vclm, FK_R), FD(M3nudgcof, FK_R),
    FD(ubarclm, FK_2D), FD(vbarclm, FK_2D), FD(M2nudgcof, FK_2D),
    FD(hbbl, FK_2D), FD(ksbl, FK_2D),                                                   // LMD_BKPP (lmd_bkpp.F)
};
static const int g_nfields = (int)(sizeof(g_fields) / sizeof(g_fields[0]));

int field_planes(const roms_hip_ctx *c, int kind) {
  const int N = c->G.N, NT = c->G.NT, NAT = c->G.NAT;
  switch (kind) {
    case FK_2D: return 1;
    case FK_R: return N;
    case FK_W: return N + 1;
    case FK_2Dx3: return 3;
    case FK_2Dx2: return 2;
    case FK_Rx2: return N * 2;
    case FK_T: return N * 3 * NT;
    case FK_Wx2: return (N + 1) * 2;
    case FK_Wx3: return (N + 1) * 3;
    case FK_2DxNT: return NT;
    case FK_WxNAT: return (N + 1) * NAT;
    case FK_RxNT: return N * NT;
  }
  return -1;
}
static long table_elems(const roms_hip_ctx *c, int kind) {
  const long N = c->G.N, NT = c->G.NT, nj = c->cnj, ni = c->cni;
  switch (kind) {
    case FK_TABR: return N;
    case FK_TABW: return N + 1;
    case FK_BJ: return nj;
    case FK_BI: return ni;
    case FK_BJN: return nj * N;
    case FK_BIN: return ni * N;
    case FK_BJT: return nj * N * NT;
    case FK_BIT: return ni * N * NT;
  }
  return -1;
}
long field_elems(const roms_hip_ctx *c, int kind) {
  const int np = field_planes(c, kind);
  return np < 0 ? table_elems(c, kind) : (long)np * c->G.nij;
}
long field_elems_caller(const roms_hip_ctx *c, int kind) {
  const int np = field_planes(c, kind);
  return np < 0 ? table_elems(c, kind) : (long)np * (long)c->cni * (long)c->cnj;
}

const FieldDesc *find_field(const char *name) {
  for (int k = 0; k < g_nfields; k++)
    if (!strcmp(name, g_fields[k].name)) return &g_fields[k];
  return nullptr;
}

// -------------------------------------------------------------------------------- life cycle
static void split_tile(int LmT, int MmT, int bw, int bh, int &nbx, int &nby, int &w, int &h) {
  nbx = (LmT + bw - 1) / bw;
  nby = (MmT + bh - 1) / bh;
  if (nbx < 1) nbx = 1;
  if (nby < 1) nby = 1;
  w = (LmT + nbx - 1) / nbx;
  h = (MmT + nby - 1) / nby;
}
static bool env_tile(const char *name, int &bw, int &bh, long maxpts) {
  const char *e = getenv(name);
  int a = 0, b = 0;
  if (e && sscanf(e, "%dx%d", &a, &b) == 2 && a >= 4 && b >= 2 && (a + 6) * (b + 6) <= (int)maxpts) { bw = a; bh = b; return true; }
  return false;
}
static void choose_blocks(DGrid &G) {
  // Sub-tile size of the COOP kernels; LDS scratch per array is (bw+6)*(bh+6) doubles (at most 12
  // arrays, 64 KB).  The 3-D kernels launch sub-tiles x N blocks and keep 32x8 sub-tiles (lanes
  // along xi).  The 2-D barotropic kernel is a chain of ~25 barrier-separated phases whose cost is
  // latency, not bandwidth: on small grids it gets small sub-tiles so that every CU holds several
  // blocks whose phases overlap.  ROMS_HIP_TILE3D / ROMS_HIP_TILE2D ("WxH") override for tuning.
  const int LmT = G.T.Iend - G.T.Istr + 1, MmT = G.T.Jend - G.T.Jstr + 1;
  int bw = 32, bh = 8;
  env_tile("ROMS_HIP_TILE3D", bw, bh, 672);              // 12 LDS arrays, 64 KB
  split_tile(LmT, MmT, bw, bh, G.nbx, G.nby, G.bw, G.bh);
  // 64x8 (one block of 1024 threads per CU, 118 KB LDS) on 64 K .. 256 K points; above that 32x8 with
  // TWO blocks per CU (64 KB LDS, 512 threads compiled for <= 128 VGPRs): a block's load phase overlaps
  // the other's compute phases -- 2048x256: 74 -> 64 us per launch, 512x512: 38.4 -> 37.1, but
  // 1024x128: 21 -> 24 (a single round of blocks either way)
  int bw2 = 64, bh2 = 8;
  if ((long)LmT * MmT <= 64L * 1024L) { bw2 = 32; bh2 = 4; }
  else if ((long)LmT * MmT >= 256L * 1024L) { bw2 = 32; bh2 = 8; }
  env_tile("ROMS_HIP_TILE2D", bw2, bh2, 1024);            // 19 LDS arrays < 160 KB; 2 points x 512 threads
  split_tile(LmT, MmT, bw2, bh2, G.nbx2, G.nby2, G.bw2, G.bh2);
  { const char *ex = getenv("ROMS_HIP_S2D_XCD"); G.xmap2 = ((G.nbx2 * G.nby2) % 8 == 0 && G.nbx2 * G.nby2 >= 16 && !(ex && ex[0] == '0')) ? 1 : 0; }
}
// narrowest first/last sub-tile of a tile_bounds_2d partition of n points into nb pieces
static int edge_subtile(int n, int nb) {
  const int c = (n + nb - 1) / nb, m = (nb * c - n) / 2;
  const int first = KMIN(c - m, n), last = n - KMAX(1 + (nb - 1) * c - m, 1) + 1;
  return KMIN(first, last);
}

static void comm_destroy(roms_hip_ctx *c);
// the options of ABI version 4's upper bits (below, behind the entries that use them)
static int mix4_config(roms_hip_ctx *c, int uv_vis4, int ts_dif4);
static int wetdry_config(roms_hip_ctx *c, double Dcrit);
static int diauv_config(roms_hip_ctx *c);
extern "C" int roms_hip_create(const roms_hip_config *cfg, roms_hip_ctx **out) {
#ifdef ROMS_CPU_EMU
  { const char *eo = getenv("ROMS_EMU_ORDER"); g_emu_reverse = eo && eo[0] == 'r'; }
#endif
  if (!cfg || !out) { set_error("null argument"); return 8; }
  if (cfg->abi_version != ROMS_HIP_ABI_VERSION) { set_error("ABI version mismatch"); return 5; }
  if (cfg->N < 4 || cfg->N > 127 || cfg->NT < 1 || cfg->NT > ROMS_MAXT || 2 * cfg->ndtfast > ROMS_MAXW) {
    set_error("unsupported dimensions (need 4 <= N <= 127, NT <= 4, 2*ndtfast <= 512)");
    return 5;
  }
  if (cfg->NtileI < 1 || cfg->NtileJ < 1 || cfg->tile < 0 || cfg->tile >= cfg->NtileI * cfg->NtileJ) {
    set_error("bad tile partition (NtileI, NtileJ, tile)");
    return 5;
  }
  if ((cfg->options & ROMS_MIX_GEO_TS) && (cfg->options & ROMS_MIX_ISO_TS)) { set_error("MIX_GEO_TS and MIX_ISO_TS exclude each other"); return 5; }
  // (MIX_ISO_TS: pinned through OVERFLOW; round 6: with MASKING + WET_DRY, oracle/ref/upwelling_wetdry_iso.h, and with the nonlinear
  // equation of state, oracle/ref/benchmark_iso.h -- no combination of it is refused any more)
  if ((cfg->options & ROMS_GLS_MIXING) && (cfg->options & ROMS_MY25_MIXING)) { set_error("GLS_MIXING and MY25_MIXING exclude each other"); return 5; }
  if (cfg->options & (ROMS_GLS_MIXING | ROMS_MY25_MIXING)) {     // gls_prestep.F, gls_corstep.F (my25_*.F): one closure, one form of it, sane parameters
    const int st = cfg->gls_flags & (ROMS_GLS_CANUTO_A | ROMS_GLS_CANUTO_B | ROMS_GLS_KANTHA_CLAYSON);
    const int ad = cfg->gls_flags & (ROMS_GLS_K_C2ADVECTION | ROMS_GLS_K_C4ADVECTION);
    if ((cfg->options & (ROMS_ANA_VMIX | ROMS_LMD_MIXING)) || (st & (st - 1)) || (ad & (ad - 1))) {
      set_error("GLS_MIXING: choose one vertical closure, at most one of CANUTO_A / CANUTO_B / KANTHA_CLAYSON and at most one of K_C2ADVECTION / K_C4ADVECTION");
      return 5;
    }
    if (!(cfg->options & ROMS_MY25_MIXING) && (!(cfg->gls_n != 0.0) || !(cfg->gls_cmu0 > 0.0) || !(cfg->gls_Kmin > 0.0) || !(cfg->gls_Pmin > 0.0) || !(cfg->gls_sigk > 0.0) ||
        !(cfg->gls_sigp > 0.0))) {
      set_error("GLS_MIXING: GLS_N must not be zero; GLS_CMU0, GLS_Kmin, GLS_Pmin, GLS_SIGK, GLS_SIGP must be positive");
      return 5;
    }
    for (int e = 0; e < 4; e++) {
      const int k = cfg->lbc_tke[e];
      if (!(k == ROMS_LBC_DEFAULT || k == ROMS_LBC_CLO || k == ROMS_LBC_GRA || k == ROMS_LBC_PER || k == ROMS_LBC_RAD)) {
        set_error("LBC(isMtke): closed, gradient, radiation and periodic are the kinds of tkebc_im.F");
        return 5;
      }
      if (k == ROMS_LBC_RAD && (cfg->options & ROMS_MY25_MIXING)) {
        set_error("LBC(isMtke) radiation with MY25_MIXING: my25_corstep.F's own edge copies are built for its closed / gradient form only");
        return 5;
      }
      if ((k == ROMS_LBC_PER) != (((e == ROMS_IWEST || e == ROMS_IEAST) ? cfg->EWperiodic : cfg->NSperiodic) != 0) && k != ROMS_LBC_DEFAULT) {
        set_error("LBC(isMtke): periodic where, and only where, the direction is");
        return 5;
      }
    }
  }
  if (cfg->Nghost != 2 && cfg->Nghost != 3) {
    // get_bounds.F / inp_par.F:210-216: NghostPoints is 2, or 3 with MPDATA/HSIMT; the strip buffers of
    // the halo exchange (3 lines per side, 9-point corner blocks) are sized for that
    set_error("Nghost must be 2 or 3 (NghostPoints of the reference)");
    return 5;
  }
  if (cfg->NtileI * cfg->NtileJ > 1 &&
      (cfg->Iend - cfg->Istr + 1 < 3 || cfg->Jend - cfg->Jstr + 1 < 3)) {
    // a strip of three lines is packed from a tile's own points: narrower tiles would send foreign lines
    set_error("a tile of a multi-tile run must be at least 3 points wide and high");
    return 5;
  }
  {  // array bounds must hold the ghost zone the kernels and the strip exchange assume
    const int pw = cfg->west_edge && !cfg->EWperiodic, pe = cfg->east_edge && !cfg->EWperiodic;
    const int ps = cfg->south_edge && !cfg->NSperiodic, pn = cfg->north_edge && !cfg->NSperiodic;
    const bool ok = cfg->LBi <= (pw ? cfg->Istr - 1 : cfg->Istr - 3) && cfg->UBi >= (pe ? cfg->Iend + 1 : cfg->Iend + cfg->Nghost) &&
                    cfg->LBj <= (ps ? cfg->Jstr - 1 : cfg->Jstr - 3) && cfg->UBj >= (pn ? cfg->Jend + 1 : cfg->Jend + cfg->Nghost);
    if (!ok) { set_error("array bounds LBi:UBi,LBj:UBj too small for the tile's ghost zone"); return 5; }
  }
#ifndef ROMS_CPU_EMU
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
    set_error("no HIP device: libroms_hip.so needs an AMD GPU (there is no CPU fallback)");
    return 2;
  }
  if (hipfail(hipSetDevice(cfg->device), "hipSetDevice")) return 2;
#endif
  roms_hip_ctx *c = new roms_hip_ctx();
  c->cfg = *cfg;
  DGrid &G = c->G;
  memset(&G, 0, sizeof(G));
  // (array bounds: below, once the neighbours are known)
  G.N = cfg->N; G.NT = cfg->NT; G.NAT = cfg->NAT; G.Lm = cfg->Lm; G.Mm = cfg->Mm; G.Nghost = cfg->Nghost;
  G.ewp = cfg->EWperiodic; G.nsp = cfg->NSperiodic; G.options = (int)(cfg->options & 0x7fffffffull);   // (the kernels read the lower word; the upper bits become DGrid members below)
  for (int i = 0; i < ROMS_MAXT; i++) { G.hadv[i] = cfg->hadv[i]; G.vadv[i] = cfg->vadv[i]; G.Akt_bak[i] = cfg->Akt_bak[i]; }
  G.T = make_bounds(cfg->Lm, cfg->Mm, cfg->EWperiodic, cfg->NSperiodic, cfg->Istr, cfg->Iend, cfg->Jstr, cfg->Jend,
                    cfg->west_edge, cfg->east_edge, cfg->south_edge, cfg->north_edge);
  choose_blocks(G);
  {  // lateral boundary conditions: what is built, and the two summaries the kernels read
    G.obc = 0; G.lbc_closed = 0;
    for (int e = 0; e < 4; e++)
      for (int v = 0; v < ROMS_ISTVAR + cfg->NT; v++) {
        const int k = lbc_kind(*cfg, e, v);
        bool ok = k >= ROMS_LBC_CLO && k <= ROMS_LBC_SHC;
        if (k == ROMS_LBC_PER && cfg->lbc[e][v] != ROMS_LBC_DEFAULT && !((e == ROMS_IWEST || e == ROMS_IEAST) ? cfg->EWperiodic : cfg->NSperiodic)) ok = false;
        if ((k == ROMS_LBC_CHE || k == ROMS_LBC_CHI) && v != ROMS_ISFSUR) ok = false;               // Chapman: free surface only
        if ((k == ROMS_LBC_FLA || k == ROMS_LBC_SHC) && v != ROMS_ISUBAR && v != ROMS_ISVBAR) ok = false;   // 2-D momentum only
        if (!ok) {
          set_error("lateral boundary condition not built: lbc[" + std::to_string(e) + "][" + std::to_string(v) + "] = " + std::to_string(cfg->lbc[e][v]) +
                    " (built: Clo Per Gra Cla Rad RadNud, Che/Cha for the free surface, Fla/Shc for ubar/vbar)");
          return 5;
        }
        if (k == ROMS_LBC_CLO) G.lbc_closed |= 1ull << (4 * v + e);
        if (k != ROMS_LBC_CLO && k != ROMS_LBC_PER) G.obc = 1;
      }
  }
  if (G.obc) {
    // mpdata_adiff.F:696-760,1158-1220 gives the anti-diffusive velocities a zero-gradient value at an open edge; the
    // MPDATA kernels carry the closed-wall form only
    for (int it = 0; it < cfg->NT; it++)
      if (cfg->hadv[it] == ROMS_MPDATA || cfg->vadv[it] == ROMS_MPDATA)
        for (int e = 0; e < 4; e++)
          if (!(G.lbc_closed & (1ull << (4 * ((e == ROMS_IWEST || e == ROMS_IEAST) ? ROMS_ISUVEL : ROMS_ISVVEL) + e))) &&
              lbc_kind(*cfg, e, ROMS_ISUVEL) != ROMS_LBC_PER) {
            set_error("MPDATA is not built together with an open boundary of the 3-D momentum (mpdata_adiff.F:696-760)");
            return 5;
          }
  }
  G.dbg_stop = getenv("ROMS_HIP_DBG_STOP") ? atoi(getenv("ROMS_HIP_DBG_STOP")) : 0;
  {  // neighbours in the reference's tile numbering; a periodic direction wraps around.
     // ROMS_HIP_SELF_EXCHANGE=1 (test aid): a tile that is alone in a periodic direction exchanges with
     // ITSELF through the transport instead of copying locally -- the message pattern of a 2-tile
     // periodic partition (both neighbours the same rank, two messages per pair matched by issue order)
     // on one GPU, so that the RCCL send/recv path can be exercised without a second device.
    memset(&c->comm, 0, sizeof(c->comm));
    const int NI = cfg->NtileI, NJ = cfg->NtileJ, it = cfg->tile % NI, jt = cfg->tile / NI;
    const char *es = getenv("ROMS_HIP_SELF_EXCHANGE");
    const bool self = es && es[0] == '1';
    int *nb = c->comm.nbr;
    // neighbour of tile t one step along xi (dx) and/or eta (dy); -1: none (domain edge, or a
    // periodic direction this tile closes by a local copy)
    auto step = [&](int t, int dx, int dy) -> int {
      if (t < 0) return -1;
      int i = t % NI, j = t / NI;
      if (dx) {
        i += dx;
        if (i < 0 || i >= NI) { if (!(cfg->EWperiodic && (NI > 1 || self))) return -1; i = (i + NI) % NI; }
      }
      if (dy) {
        j += dy;
        if (j < 0 || j >= NJ) { if (!(cfg->NSperiodic && (NJ > 1 || self))) return -1; j = (j + NJ) % NJ; }
      }
      return i + j * NI;
    };
    (void)it; (void)jt;
    static const int ddx[8] = {-1, 1, 0, 0, -1, 1, -1, 1}, ddy[8] = {0, 0, -1, 1, -1, -1, 1, 1};
    for (int d = 0; d < 8; d++) nb[d] = step(cfg->tile, ddx[d], ddy[d]);
    G.xloc = nb[0] < 0 && nb[1] < 0;
    G.yloc = nb[2] < 0 && nb[3] < 0;
    c->has_exchange = nb[0] >= 0 || nb[1] >= 0 || nb[2] >= 0 || nb[3] >= 0;
    // Array bounds.  The caller's LBi:UBi x LBj:UBj (the reference layout) is what upload / download move.  A multi-tile
    // context that runs the barotropic steps as predictor+corrector pairs (k_step2d_pair.h) keeps a wider ghost zone
    // towards every neighbouring tile -- B2D_GL | B2D_GH lines -- in ALL its arrays (one index rule for every kernel);
    // the 2-D exchanges behind the pairs fill it, every other exchange fills the inner 3 | Nghost lines as before.
    // Every rank takes the same decision: it depends on the partition only (the narrowest tile must own the lines it sends).
    c->cLBi = cfg->LBi; c->cLBj = cfg->LBj;
    c->cni = cfg->UBi - cfg->LBi + 1; c->cnj = cfg->UBj - cfg->LBj + 1;
    int LB_i = cfg->LBi, UB_i = cfg->UBi, LB_j = cfg->LBj, UB_j = cfg->UBj;
    {
      const char *ep = getenv("ROMS_HIP_PAIR");
      const bool shape_ok = ep && ep[0] == '1' ? true : (G.bw2 <= 32 && G.bh2 <= 4);
      const int wmin = KMIN(edge_subtile(cfg->Lm, NI), (cfg->Lm + NI - 1) / NI), hmin = KMIN(edge_subtile(cfg->Mm, NJ), (cfg->Mm + NJ - 1) / NJ);
      c->pair_mt = c->has_exchange && !(ep && ep[0] == '0') && shape_ok && wmin >= 8 && hmin >= 8 && !G.obc;   // (open boundaries: per-call kernels, k_obc.h)
      if (c->pair_mt) {
        if (nb[0] >= 0) LB_i = KMIN(LB_i, cfg->Istr - B2D_GL);
        if (nb[1] >= 0) UB_i = KMAX(UB_i, cfg->Iend + B2D_GH);
        if (nb[2] >= 0) LB_j = KMIN(LB_j, cfg->Jstr - B2D_GL);
        if (nb[3] >= 0) UB_j = KMAX(UB_j, cfg->Jend + B2D_GH);
      }
    }
    G.LBi = LB_i; G.LBj = LB_j;
    G.ni = UB_i - LB_i + 1; G.nj = UB_j - LB_j + 1;
    G.nij = (long)G.ni * G.nj;
    c->wide = G.LBi != c->cLBi || G.LBj != c->cLBj || G.ni != c->cni || G.nj != c->cnj;
    c->stage_buf = nullptr; c->stage_cap = 0;
    c->static_wide_dirty = c->pair_mt;
    G.xgl = 3; G.xgh = cfg->Nghost;
    // rim of a split 3-D producer (DGrid::rimw): the lines its strip exchange packs and its boundary fills read -- the widest
    // narrow strip, max(xgl, xgh) <= 3 -- counted from the edge of the LAUNCH's index space, whose origin lies up to two
    // lines outside the tile's first point (IstrT, KMIN(IstrP, IstrT) ...): launch_halo_tail checks what it is handed
    G.region = 0; G.rimw = KMAX(G.xgl, G.xgh) + 2;
  }
  {  // producer-side halo fills need the whole domain on this GPU and edge sub-tiles that own the
     // three source lines of a periodic copy
    const int LmT = cfg->Iend - cfg->Istr + 1, MmT = cfg->Jend - cfg->Jstr + 1;
    const char *e = getenv("ROMS_HIP_FUSE_HALO");
    // (round 6: a closed basin too -- the corner averages by the thread of the point next to the corner, k_haloblock.h:
    // HB_CORNERS; ROMS_HIP_FUSE_CLOSED=0 keeps its separate halo launches)
    const char *ecb = getenv("ROMS_HIP_FUSE_CLOSED");
    const bool closed_ok = !(ecb && ecb[0] == '0');
    G.fuse_halo = cfg->NtileI * cfg->NtileJ == 1 && !c->has_exchange && (cfg->EWperiodic || cfg->NSperiodic || closed_ok) && LmT >= 6 && MmT >= 6 &&
                  !(e && e[0] == '0') && !G.obc;
    const char *e3 = getenv("ROMS_HIP_FUSE3D");
    // (a masked run: the barotropic kernel's boundary stores carry the mask of the boundary point, hb_emit; the 3-D
    // producers take the separate halo launches -- their emit_plan knows no mask)
    G.fuse3d = G.fuse_halo && (cfg->EWperiodic || cfg->NSperiodic) && !(e3 && e3[0] == '0') && !(cfg->options & ROMS_MASKING);
  }
  G.ntfirst = cfg->ntfirst; G.nfast = cfg->nfast;
  G.dt = cfg->dt; G.dtfast = cfg->dtfast; G.rho0 = cfg->rho0; G.g = cfg->g; G.lambda = cfg->lambda;
  G.gamma2 = cfg->gamma2; G.Cp = cfg->Cp; G.R0 = cfg->R0; G.T0 = cfg->T0; G.S0 = cfg->S0; G.Tcoef = cfg->Tcoef;
  G.Scoef = cfg->Scoef; G.hc = cfg->hc; G.dstart = cfg->dstart; G.Akv_bak = cfg->Akv_bak; G.Zob = cfg->Zob;
  G.Vtransform = cfg->Vtransform;
  c->profile = false;
  memset(c->regions, 0, sizeof(c->regions));
  c->comm_failed = false;
  c->m2d_dirty = true;
  c->b2_stage = 0;
  c->pair_on = step2d_pair_usable(c);
  c->loop_state = 0;
  c->swdk_ready = false;
  c->pre_t3_ready = false;
  c->stream3 = nullptr;
  c->G.dia_ts = 0;
  c->G.dia_uv = 0; c->G.ndm2 = 0; c->G.ndm3 = 0; c->G.ndrhs = 0;
  c->G.uv_vis4 = 0; c->G.ts_dif4 = 0; c->F.lap4 = nullptr; c->F.wd_eff = nullptr;
  for (int k = 0; k < 12; k++) { c->G.m2[k] = 0; c->G.m3[k] = 0; }
  c->F.duv = nullptr;
  for (int k = 0; k < 10; k++) c->G.dia_idx[k] = 0;
  for (int f = 0; f < 24; f++) c->avg[f] = nullptr;
  c->avg_nAVG = 0; c->avg_ntsAVG = 1; c->avg_nrrec = 0; c->avg_ntstart = 1; c->avg_mask = 0;
  c->avg_time = 0.0; c->avg_done_iic = -1;
  c->late_pre = false;
  c->diag_join_pending = false;
  for (int e = 0; e < 12; e++) c->ev_lane[e] = nullptr;
#ifdef ROMS_CPU_EMU
  c->stream = nullptr;
  c->stream0 = nullptr;
  c->stream2 = nullptr;
  c->overlap = false;
#else
  int prio_lo = 0, prio_hi = 0;
  if (getenv("ROMS_HIP_PRIO")) (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);   // (least, greatest)
  if (hipfail(hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio_hi), "hipStreamCreate")) { delete c; return 2; }
  c->stream0 = c->stream;
  if (hipfail(hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, prio_hi), "hipStreamCreate")) { delete c; return 2; }
  (void)hipEventCreate(&c->ev0);
  (void)hipEventCreate(&c->ev1);
  (void)hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming | hipEventDisableSystemFence);
  (void)hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming | hipEventDisableSystemFence);
  (void)hipEventCreateWithFlags(&c->ev_point, hipEventDisableTiming | hipEventDisableSystemFence);
  if (hipfail(hipStreamCreateWithPriority(&c->stream3, hipStreamNonBlocking, prio_lo), "hipStreamCreate")) { delete c; return 2; }
  if (hipfail(hipStreamCreateWithPriority(&c->stream4, hipStreamNonBlocking, prio_lo), "hipStreamCreate")) { delete c; return 2; }
  for (int e = 0; e < 12; e++) (void)hipEventCreateWithFlags(&c->ev_lane[e], hipEventDisableTiming | hipEventDisableSystemFence);
  {
    const char *e = getenv("ROMS_HIP_OVERLAP");
    c->overlap = !(e && e[0] == '0');
  }
  c->xstream = nullptr;
  c->x_async = false;
  c->rim_split = false;
  c->x_2d_ok = false;
  c->x_tail = false;
  c->x_wide = false;
  c->x_pending = 0;
  c->ev_x_next = 0;
  for (int k = 0; k < 16; k++) c->x_event_of[k] = -1;
  if (c->has_exchange) {
    // Default: on when the neighbours are other ranks (an xGMI transfer to hide), off on the one-GPU
    // self-exchange test path, where the "transfer" is a local copy and the four cross-stream hops of an
    // asynchronous exchange only cost (measured: BENCHMARK1 2.92 -> 3.24, 512x512x50 10.48 -> 10.65,
    // BENCHMARK3 13.06 -> 13.19 ms per step).  ROMS_HIP_XASYNC=0/1 forces it (the parity tests run both).
    const char *e = getenv("ROMS_HIP_XASYNC"), *ep = getenv("ROMS_HIP_XASYNC_PLANES"), *es2 = getenv("ROMS_HIP_SELF_EXCHANGE");
    const bool selfx = es2 && es2[0] == '1';
    c->x_async = e ? e[0] != '0' : !selfx;
    // rim / interior split of the 3-D producers (g_rhs3d.cpp:run_pre_t3, g_step3d.cpp): by default from 128 K columns up.
    // Measured on one MI355X through the mailbox (every exchange device-local, so there is nothing to hide and the
    // split shows only its cost -- one more launch and two cross-stream hops per exchange point): 512x64x30 tile
    // 1.49 -> 1.69 ms per step, 512x512x50 7.84 -> 7.91; a 3-D exchange point of the large tile moves ~5 MB (xGMI:
    // tens of microseconds), of the small one 0.4 MB.  ROMS_HIP_RIM=0/1 forces it; the tile tests run it on.
    { const char *er = getenv("ROMS_HIP_RIM");
      const long cols = (long)(G.T.Iend - G.T.Istr + 1) * (G.T.Jend - G.T.Jstr + 1);
      c->rim_split = c->x_async && (er ? er[0] != '0' : cols >= 128L * 1024L); }
    c->x_min_planes = ep ? atoi(ep) : 8;
    if (hipfail(hipStreamCreateWithFlags(&c->xstream, hipStreamNonBlocking), "hipStreamCreate")) { delete c; return 2; }
    (void)hipEventCreateWithFlags(&c->ev_xprod, hipEventDisableTiming | hipEventDisableSystemFence);
    for (int k = 0; k < 32; k++) (void)hipEventCreateWithFlags(&c->ev_x[k], hipEventDisableTiming | hipEventDisableSystemFence);
  }
#endif
  // state arrays
  memset(&c->F, 0, sizeof(c->F));
  for (int k = 0; k < g_nfields; k++) {
    void *p = nullptr;
    // (the climatology and nudging-coefficient arrays -- seven 3-D arrays' worth -- only in a context that nudges;
    // ROMS_HIP_CLIMA_ALLOC=1 allocates them regardless: no difference in the step time was measured either way)
    {
      const size_t off = g_fields[k].offset;
      const bool clima_t = off == offsetof(Fields, tclm) || off == offsetof(Fields, Tnudgcof);
      const bool clima_m = off == offsetof(Fields, uclm) || off == offsetof(Fields, vclm) || off == offsetof(Fields, M3nudgcof);
      const char *eca = getenv("ROMS_HIP_CLIMA_ALLOC");
      if (!(eca && eca[0] == '1') && ((clima_t && !(cfg->options & ROMS_NUDGE_TCLM_ALL)) || (clima_m && !(cfg->options & ROMS_NUDGE_M3CLM)))) continue;
    }
    // (zeta, ubar, vbar: two more levels than the caller sees -- the staging levels of the pair kernel, k_step2d_pair.h)
    const size_t ne = (size_t)field_elems(c, g_fields[k].kind) + (g_fields[k].kind == FK_2Dx3 ? 2 * (size_t)G.nij : 0);
    if (dmalloc(&p, ne * sizeof(double))) { roms_hip_destroy(c); return 2; }
    c->allocs.push_back(p);
    *(double **)((char *)&c->F + g_fields[k].offset) = (double *)p;
  }
  {  // land/sea masks: all water until uploaded
    std::vector<double> ones((size_t)G.nij, 1.0);
    for (double *m : {(double *)c->F.rmask, (double *)c->F.umask, (double *)c->F.vmask, (double *)c->F.pmask})
      if (h2d(m, ones.data(), ones.size() * sizeof(double), c->stream)) { roms_hip_destroy(c); return 2; }
    c->G.rmask = c->F.rmask; c->G.umask = c->F.umask; c->G.vmask = c->F.vmask; c->G.pmask = c->F.pmask;
    c->G.masking = (cfg->options & ROMS_MASKING) != 0;
    c->G.wet_dry = 0; c->G.Dcrit = 0.0; c->G.hbath = c->F.h;
    c->G.rmask_wet = c->F.rmask_wet; c->G.umask_wet = c->F.umask_wet; c->G.vmask_wet = c->F.vmask_wet; c->G.pmask_wet = c->F.pmask_wet;
  }
  for (int k = 0; k < 13; k++) {
    void *p = nullptr;
    size_t n = (size_t)G.nij * (size_t)(G.N + 1) * (k == 0 || k == 3 || k == 4 ? (size_t)G.NT : 1);   // [3], [4]: spline scratch per tracer
    if (dmalloc(&p, n * sizeof(double))) { roms_hip_destroy(c); return 2; }
    c->allocs.push_back(p);
    c->F.wrk3[k] = (double *)p;
  }
  for (int k = 0; k < 4; k++) {
    void *p = nullptr;
    if (dmalloc(&p, (size_t)G.nij * sizeof(double))) { roms_hip_destroy(c); return 2; }
    c->allocs.push_back(p);
    c->F.wrk2[k] = (double *)p;
  }
  c->F.tmix = nullptr;
  if (cfg->options & ROMS_TS_DIF2) {                        // the terms of t3dmix2 where it runs ahead of pre_step3d
    void *p = nullptr;
    if (dmalloc(&p, (size_t)G.nij * (size_t)G.N * (size_t)G.NT * sizeof(double))) { roms_hip_destroy(c); return 2; }
    c->allocs.push_back(p);
    c->F.tmix = (double *)p;
  }
  for (int k = 0; k < 2; k++) {   // packed barotropic metrics, 8 doubles per point
    void *p = nullptr;
    if (dmalloc(&p, 8 * (size_t)G.nij * sizeof(double))) { roms_hip_destroy(c); return 2; }
    c->allocs.push_back(p);
    (k == 0 ? c->F.m2r : c->F.m2p) = (double *)p;
  }
  {  // MPDATA work arrays
    bool any_mp = false;
    for (int it = 0; it < cfg->NT; it++) any_mp |= cfg->hadv[it] == ROMS_MPDATA || cfg->vadv[it] == ROMS_MPDATA;
    const bool plain = (cfg->options & ROMS_PLAIN_VDIFF) != 0;   // k_mp_vdiff's two work arrays (the form without LDS)
    for (int k = 0; k < 6; k++) {
      if (!(any_mp || (plain && k >= 4))) continue;
      void *p = nullptr;
      if (dmalloc(&p, (size_t)G.nij * (size_t)(G.N + 1) * (k == 0 ? (size_t)G.NT : 1) * sizeof(double))) { roms_hip_destroy(c); return 2; }
      c->allocs.push_back(p);
      c->F.mp3[k] = (double *)p;
    }
  }
  {  // pointer table for the kernels
#ifdef ROMS_CPU_EMU
    c->d_F = &c->F;
#else
    void *p = nullptr;
    if (dmalloc(&p, sizeof(Fields))) { roms_hip_destroy(c); return 2; }
    c->allocs.push_back(p);
    c->d_F = (Fields *)p;
    if (h2d(c->d_F, &c->F, sizeof(Fields), c->stream)) { roms_hip_destroy(c); return 2; }
#endif
  }
  // s-coordinate tables
  if (h2d(c->F.sc_r, cfg->sc_r, sizeof(double) * (size_t)G.N, c->stream) ||
      h2d(c->F.Cs_r, cfg->Cs_r, sizeof(double) * (size_t)G.N, c->stream) ||
      h2d(c->F.sc_w, cfg->sc_w, sizeof(double) * (size_t)(G.N + 1), c->stream) ||
      h2d(c->F.Cs_w, cfg->Cs_w, sizeof(double) * (size_t)(G.N + 1), c->stream)) { roms_hip_destroy(c); return 2; }
  // diag scratch
  c->nblk_diag = 256;
  {
    void *p = nullptr;
    if (dmalloc(&p, sizeof(double) * 16 * (size_t)(G.nj + 8))) { roms_hip_destroy(c); return 2; }
    c->allocs.push_back(p);
    c->d_diag = (double *)p;
    c->h_diag = (double *)calloc(16 * (size_t)(G.nj + 8), sizeof(double));
    void *q = nullptr;
    if (dmalloc(&q, sizeof(double) * (9 * (size_t)G.nij + 12 * (size_t)G.ni))) { roms_hip_destroy(c); return 2; }
    c->allocs.push_back(q);
    c->d_diagwork = (double *)q;
  }
#ifndef ROMS_CPU_EMU
  // the zero fills above ran on the null stream; the context's stream is non-blocking and would
  // not wait for them
  if (hipfail(hipDeviceSynchronize(), "hipDeviceSynchronize")) { roms_hip_destroy(c); return 2; }
#endif
  memset(&c->s, 0, sizeof(c->s));
  c->s.iif = 1; c->s.indx1 = 1; c->s.kstp = 1; c->s.krhs = 1; c->s.knew = 1;
  c->s.nstp = 1; c->s.nrhs = 1; c->s.nnew = 1;
  c->s.time = cfg->dstart * 86400.0;
  ctx_sync_stepping(c);
  if (poison_work(c)) { roms_hip_destroy(c); return 2; }
  // ABI version 4: the options of the upper word (configuration calls of their own until round 4)
  if (cfg->options & (ROMS_UV_VIS4 | ROMS_TS_DIF4)) {
    const int r = mix4_config(c, (cfg->options & ROMS_UV_VIS4) != 0, (cfg->options & ROMS_TS_DIF4) != 0);
    if (r) { roms_hip_destroy(c); return r; }
  }
  if (cfg->options & ROMS_WET_DRY) {
    const int r = wetdry_config(c, cfg->Dcrit);
    if (r) { roms_hip_destroy(c); return r; }
  }
  if (cfg->options & (ROMS_NUDGE_M3CLM | ROMS_NUDGE_TCLM_ALL | ROMS_NUDGE_M2CLM)) {       // climatology nudging: rhs3d.F:654-680, step3d_t.F:1866-1878, step2d_LF_AM3.h:2179
    // (open boundaries, round 6: the radiation + nudging conditions read the coefficient arrays, k_obc.h: ObcItem::cof; OBCFAC in the
    // configuration since ABI version 5)
    if (cfg->options & ROMS_DIAGNOSTICS_UV) { set_error("climatology nudging with DIAGNOSTICS_UV: not built"); roms_hip_destroy(c); return 5; }
    // (the rule of this library: an option set without a reference-written fixture or a pinned oracle run is refused)
    // (round 6: nudging of the 3-D momentum and the tracers together with WET_DRY, TS_DIF4 / UV_VIS4 and MIX_GEO_UV is pinned --
    // tests/test_oracle_vs_ref.py, clima=39 on the wetting/drying, biharmonic and geopotential cases.  What stays refused is the
    // nudging of the 2-D momentum where the barotropic step runs a kernel without the nudging term: WET_DRY and UV_VIS4)
    if ((cfg->options & ROMS_NUDGE_M2CLM) && (cfg->options & (ROMS_WET_DRY | ROMS_UV_VIS4))) {
      set_error("LnudgeM2CLM together with WET_DRY or UV_VIS4: the barotropic kernels of those options carry no nudging term"); roms_hip_destroy(c); return 5;
    }
    c->G.clima = ((cfg->options & ROMS_NUDGE_M3CLM) ? 1 : 0) | ((cfg->options & ROMS_NUDGE_M2CLM) ? 32 : 0);
    for (int it = 1; it <= c->G.NT; it++) if (cfg->options & ROMS_NUDGE_TCLM(it)) c->G.clima |= 1 << it;
    if (c->G.clima & 30) c->G.fuse3d = 0;
    if (c->G.clima & 32) { c->pair_on = step2d_pair_usable(c); c->loop_state = 0; }     // (LnudgeM2CLM: the per-call barotropic kernel)                    // (the nudging sits between t3dbc and the exchange: separate launches)
  }
  c->G.bkpp = 0;
  if (cfg->options & ROMS_LMD_BKPP) {                       // lmd_bkpp.F (round 6)
    if (!(cfg->options & ROMS_LMD_MIXING)) { set_error("LMD_BKPP without LMD_MIXING"); roms_hip_destroy(c); return 5; }
    if (c->G.NT < 2 || !(cfg->options & ROMS_SALINITY)) { set_error("LMD_BKPP: built with salinity (the reference builds the oracle is pinned to)"); roms_hip_destroy(c); return 5; }
    // (pinned: benchmark.h and upwelling_kpp.h with -DLMD_BKPP, one tile and 2x2; not with the options below)
    if (cfg->options & (ROMS_MASKING | ROMS_WET_DRY | ROMS_LMD_DDMIX)) { set_error("LMD_BKPP together with MASKING, WET_DRY or LMD_DDMIX: not pinned against the reference, not built"); roms_hip_destroy(c); return 5; }
    c->G.bkpp = 1;
    c->G.fuse3d = 0;                                        // (hbbl's boundary values and the order skpp -> bkpp -> finish: separate launches)
  }
  c->G.volcons = 0; c->G.vcons = nullptr;
  if (cfg->volcons & 15) {                                   // VolCons: obc_volcons.F (round 6)
    const int vc = cfg->volcons & 15;
    if (!c->G.obc) { set_error("VolCons without an open boundary"); roms_hip_destroy(c); return 5; }
    for (int e = 0; e < 4; e++)
      if ((vc & (1 << e)) && ((e == ROMS_IWEST || e == ROMS_IEAST) ? cfg->EWperiodic : cfg->NSperiodic)) { set_error("VolCons on a periodic edge"); roms_hip_destroy(c); return 5; }
    if (cfg->NtileI * cfg->NtileJ > 1) {
      set_error("VolCons on more than one tile: obc_flux_tile sums over the tiles (mp_reduce), a reduction across ranks this library has not got");
      roms_hip_destroy(c); return 5;
    }
    // (pinned: the KELVIN cases with and without masks, all four edges; not with the options whose barotropic kernels are forms of their own)
    if (cfg->options & (ROMS_WET_DRY | ROMS_UV_VIS4 | ROMS_DIAGNOSTICS_UV | ROMS_NUDGE_M2CLM)) {
      set_error("VolCons together with WET_DRY, UV_VIS4, DIAGNOSTICS_UV or LnudgeM2CLM: not pinned against the reference, not built"); roms_hip_destroy(c); return 5;
    }
    void *p = nullptr;
    if (dmalloc(&p, 4 * sizeof(double))) { roms_hip_destroy(c); return 2; }
    c->allocs.push_back(p);
    // (zeroed by dmalloc: mod_scalars.F:1460-1462, bc_area = bc_flux = ubar_xs = 0)
    c->G.vcons = (double *)p;
    c->G.volcons = vc;
  }
  c->G.ddmix = 0; c->G.alfaobeta = nullptr;
  if (cfg->options & ROMS_LMD_DDMIX) {                      // lmd_vmix.F:360-428
    if (!(cfg->options & ROMS_LMD_MIXING)) { set_error("LMD_DDMIX without LMD_MIXING"); roms_hip_destroy(c); return 5; }
    if (c->G.NT < 2 || !(cfg->options & ROMS_SALINITY)) { set_error("LMD_DDMIX needs salinity (the double-diffusive density ratio)"); roms_hip_destroy(c); return 5; }
    void *p = nullptr;
    const size_t nb = (size_t)c->G.nij * (size_t)(c->G.N + 1) * sizeof(double);
    if (dmalloc(&p, nb)) { roms_hip_destroy(c); return 2; }
    c->allocs.push_back(p);
    c->G.alfaobeta = (double *)p;
    c->G.ddmix = 1;
  }
  c->G.prs4x = (cfg->options & ROMS_PRSGRD44) ? 44 : ((cfg->options & ROMS_PRSGRD42) ? 42 : 0);     // prsgrd.F:16-19
  if (c->G.prs4x) {
    if (c->G.prs4x == 42 && cfg->NtileI * cfg->NtileJ != 1) {
      set_error("PJ_GRADPQ2 (prsgrd42.h) on more than one tile: its second pass reads rv(Iend+1,j,k) (prsgrd42.h:449), a column no tile computes -- "
                "the reference's own result depends on the partition; a single tile, or PJ_GRADPQ4");
      roms_hip_destroy(c); return 5;
    }
    if ((cfg->options & ROMS_WET_DRY) && c->G.prs4x == 42) { set_error("WET_DRY with PJ_GRADPQ2: prsgrd42.h masks its FIRST pass (:355-361), which is not built"); roms_hip_destroy(c); return 5; }
    if (c->G.N < 3) { set_error("PJ_GRADPQ2 / PJ_GRADPQ4: the reconstruction needs three levels"); roms_hip_destroy(c); return 5; }
  }
  if (cfg->options & ROMS_MIX_GEO_UV) {                     // uv3dmix2_geo.h | uv3dmix4_geo.h (k_uvmix_geo.h): twenty-two 3-D work arrays
    if (!(cfg->options & ROMS_UV_VIS2) && !c->G.uv_vis4) { set_error("MIX_GEO_UV without UV_VIS2 or UV_VIS4"); roms_hip_destroy(c); return 5; }
    // (open boundaries: pinned with oracle/ref/kelvin_geouv.h since round 6)
    if (cfg->options & ROMS_DIAGNOSTICS_UV) { set_error("MIX_GEO_UV: the DIAGNOSTICS_UV statements of uv3dmix2_geo.h are not built"); roms_hip_destroy(c); return 5; }
    if ((cfg->options & ROMS_WET_DRY) && c->G.uv_vis4) { set_error("UV_VIS4 with WET_DRY: harmonic mixing only"); roms_hip_destroy(c); return 5; }
    void *p = nullptr;
    if (dmalloc(&p, (size_t)22 * (size_t)c->G.nij * (size_t)(c->G.N + 1) * sizeof(double))) { roms_hip_destroy(c); return 2; }    // (UG_NARR)
    c->allocs.push_back(p);
    c->F.gwrk = (double *)p;
    c->G.mix_geo_uv = 1;
  }
  *out = c;
  return 0;
}

static void work_spans(roms_hip_ctx *c, std::vector<WorkSpan> &out) {
  const DGrid &G = c->G;
  const size_t plane = (size_t)G.nij * sizeof(double), col = plane * (size_t)(G.N + 1);
  for (int k = 0; k < 13; k++) out.push_back({(void *)(double *)c->F.wrk3[k], col * (k == 0 || k == 3 || k == 4 ? (size_t)G.NT : 1)});
  for (int k = 0; k < 4; k++) out.push_back({(void *)(double *)c->F.wrk2[k], plane});
  if ((double *)c->F.tmix) out.push_back({(void *)(double *)c->F.tmix, (size_t)G.nij * (size_t)G.N * (size_t)G.NT * sizeof(double)});
  for (int k = 0; k < 6; k++)
    if ((double *)c->F.mp3[k]) out.push_back({(void *)(double *)c->F.mp3[k], col * (k == 0 ? (size_t)G.NT : 1)});
  // the two staging levels behind zeta, ubar, vbar (k_step2d_pair.h)
  for (double *f : {(double *)c->F.zeta, (double *)c->F.ubar, (double *)c->F.vbar}) out.push_back({(void *)(f + 3 * (size_t)G.nij), 2 * plane});
}

extern "C" int roms_hip_destroy(roms_hip_ctx *c) {
  if (!c) return 0;
  (void)dsync(c->stream);
#ifndef ROMS_CPU_EMU
  if (c->stream2) (void)hipStreamSynchronize(c->stream2);   // side-stream kernels may still read the arrays
  if (c->stream3) (void)hipStreamSynchronize(c->stream3);
  if (c->stream4) (void)hipStreamSynchronize(c->stream4);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->ev_join) (void)hipEventDestroy(c->ev_join);
  if (c->ev_point) (void)hipEventDestroy(c->ev_point);
  for (int e = 0; e < 12; e++) if (c->ev_lane[e]) (void)hipEventDestroy(c->ev_lane[e]);
  if (c->xstream) {
    (void)hipStreamSynchronize(c->xstream);
    (void)hipEventDestroy(c->ev_xprod);
    for (int k = 0; k < 32; k++) (void)hipEventDestroy(c->ev_x[k]);
    (void)hipStreamDestroy(c->xstream);
  }
#endif
  for (void *p : c->allocs) dfree(p);
  for (ktimer_t e : c->step_ev) ktimer_destroy(e);
  c->step_ev.clear();
#ifndef ROMS_CPU_EMU
  if (c->loop_err) (void)hipHostFree(c->loop_err);
#endif
  if (c->stage_buf) dfree(c->stage_buf);
  for (int k = 0; k < 8; k++) { if (c->comm.sbuf[k]) dfree(c->comm.sbuf[k]); if (c->comm.rbuf[k]) dfree(c->comm.rbuf[k]); }
  comm_destroy(c);
  free(c->h_diag);
#ifndef ROMS_CPU_EMU
  if (c->stream) (void)hipStreamDestroy(c->stream);
  if (c->stream2) (void)hipStreamDestroy(c->stream2);
  if (c->stream3) (void)hipStreamDestroy(c->stream3);
  if (c->stream4) (void)hipStreamDestroy(c->stream4);
#endif
  delete c;
  return 0;
}

void ctx_sync_stepping(roms_hip_ctx *c) {
  DGrid &G = c->G;
  const roms_hip_stepping &s = c->s;
  G.iic = s.iic; G.iif = s.iif; G.nstp = s.nstp; G.nnew = s.nnew; G.nrhs = s.nrhs;
  G.kstp = s.kstp; G.knew = s.knew; G.krhs = s.krhs; G.predictor = s.predictor;
  G.time = s.time;
  G.tdays = s.time * (1.0 / 86400.0);
}

extern "C" int roms_hip_set_stepping(roms_hip_ctx *c, const roms_hip_stepping *s) {
  if (!c || !s) return 8;
  c->s = *s;
  ctx_sync_stepping(c);
  return 0;
}
extern "C" int roms_hip_get_stepping(roms_hip_ctx *c, roms_hip_stepping *s) {
  if (!c || !s) return 8;
  *s = c->s;
  return 0;
}

// DIAGNOSTICS_TS arrays ("DiaTwrk", "DiaTrc": N*NT*NDT planes; "dia_zeta": one): device pointer and planes, or nullptr
static double *dia_field(roms_hip_ctx *c, const char *name, int *np) {
  if (!c->G.dia_ts) return nullptr;
  const int n = c->G.N * c->G.NT * c->G.dia_ts;
  if (!strcmp(name, "DiaTwrk")) { *np = n; return (double *)c->F.DiaTwrk; }
  if (!strcmp(name, "DiaTrc")) { *np = n; return (double *)c->F.DiaTrc; }
  if (!strcmp(name, "dia_zeta")) { *np = 1; return (double *)c->F.dia_zeta; }
  return duv_field(c, name, np);                     // DIAGNOSTICS_UV: "DiaU2wrk" ... "DiaV3d" (g_duv.cpp)
}
extern "C" long roms_hip_field_size(roms_hip_ctx *c, const char *name) {
  const FieldDesc *f = find_field(name);
  if (!f) {
    int dnp = 0;
    if (dia_field(c, name, &dnp)) return (long)dnp * c->cni * c->cnj;
    const int a = avg_field_index(name);
    return (a >= 0 && c->avg[a]) ? avg_field_elems(c, a) / c->G.nij * ((long)c->cni * c->cnj) : -1;
  }
  return field_elems_caller(c, f->kind);
}
// Caller layout <-> library layout (c->wide): `np` planes through a device staging buffer in the caller's layout.
// to_lib: the caller's window is copied into the library's arrays (the ghost lines beyond it keep their contents);
// otherwise the window is cut out of them.
static int relayout(roms_hip_ctx *c, double *lib, int np, bool to_lib, const double *host_in, double *host_out) {
  const size_t cn = (size_t)c->cni * (size_t)c->cnj * (size_t)np;
  if (cn > c->stage_cap) {
    if (c->stage_buf) { (void)dsync(c->stream); dfree(c->stage_buf); c->stage_buf = nullptr; c->stage_cap = 0; }
    void *p = nullptr;
    if (dmalloc(&p, cn * sizeof(double))) return 2;
    c->stage_buf = (double *)p; c->stage_cap = cn;
  }
  RelayoutArgs a;
  a.lib = lib; a.win = c->stage_buf; a.to_lib = to_lib ? 1 : 0;
  a.ni = c->G.ni; a.nij = c->G.nij; a.cni = c->cni; a.cnij = (long)c->cni * c->cnj;
  a.di = c->cLBi - c->G.LBi; a.dj = c->cLBj - c->G.LBj;
  if (to_lib) {
    int r = h2d(c->stage_buf, host_in, cn * sizeof(double), c->stream);
    if (r) return r;
    LAUNCH_THREAD(k_relayout, c->cni, c->cnj, np, c->stream, a);
    return dsync(c->stream);
  }
  LAUNCH_THREAD(k_relayout, c->cni, c->cnj, np, c->stream, a);
  return d2h(host_out, c->stage_buf, cn * sizeof(double), c->stream);
}
extern "C" int roms_hip_upload(roms_hip_ctx *c, const char *name, const double *host, long n) {
  const FieldDesc *f = find_field(name);
  if (!f) { set_error(std::string("unknown field ") + name); return 8; }
  if (n != field_elems_caller(c, f->kind)) { set_error(std::string("size mismatch for field ") + name); return 8; }
  if (f->kind == FK_2D) { c->m2d_dirty = true; c->static_wide_dirty = c->pair_mt; }
  halo_fence(c, FG_ALL);
  double *dst = *(double **)((char *)&c->F + f->offset);
  if (!dst) { set_error(std::string("field ") + name + " is not allocated in this context (climatology arrays: with ROMS_NUDGE_M3CLM / ROMS_NUDGE_TCLM)"); return 8; }
  const int np = field_planes(c, f->kind);
  if (c->wide && np > 0) return relayout(c, dst, np, true, host, nullptr);
  return h2d(dst, host, (size_t)n * sizeof(double), c->stream);
}
extern "C" int roms_hip_download(roms_hip_ctx *c, const char *name, double *host, long n) {
  const FieldDesc *f = find_field(name);
  if (!f) {
    int dnp = 0;
    double *dp = dia_field(c, name, &dnp);             // per-term tracer tendencies: "DiaTwrk", "DiaTrc", "dia_zeta"
    if (dp) {
      if (n != (long)dnp * c->cni * c->cnj) { set_error(std::string("size mismatch for field ") + name); return 8; }
      halo_fence(c, FG_ALL);
      if (c->wide) return relayout(c, dp, dnp, false, nullptr, host);
      return d2h(host, dp, (size_t)n * sizeof(double), c->stream);
    }
    const int a = avg_field_index(name);               // time-averaged fields: "avg_zeta" ... "avg_HvomT"
    if (a >= 0 && c->avg[a]) {
      const int np = (int)(avg_field_elems(c, a) / c->G.nij);
      if (n != (long)np * c->cni * c->cnj) { set_error(std::string("size mismatch for field ") + name); return 8; }
      halo_fence(c, FG_ALL);
      if (c->wide) return relayout(c, c->avg[a], np, false, nullptr, host);
      return d2h(host, c->avg[a], (size_t)n * sizeof(double), c->stream);
    }
  }
  if (!f) { set_error(std::string("unknown field ") + name); return 8; }
  if (n != field_elems_caller(c, f->kind)) { set_error(std::string("size mismatch for field ") + name); return 8; }
  halo_fence(c, FG_ALL);
  double *src = *(double **)((char *)&c->F + f->offset);
  if (!src) { set_error(std::string("field ") + name + " is not allocated in this context"); return 8; }
  const int np = field_planes(c, f->kind);
  if (c->wide && np > 0) return relayout(c, src, np, false, nullptr, host);
  return d2h(host, src, (size_t)n * sizeof(double), c->stream);
}
extern "C" int roms_hip_sync(roms_hip_ctx *c) {
  halo_fence(c, FG_ALL);
  int r = dsync(c->stream);
#ifndef ROMS_CPU_EMU
  if (!r && c->loop_err && *(volatile unsigned long long *)c->loop_err) r = ctx_check(c, "sync");   // (a wait inside the persistent barotropic loop gave up)
#endif
  return r;
}
int run_rhs3d_pt(roms_hip_ctx *c);
int run_uv3dmix2_s(roms_hip_ctx *c);
int run_uv3dmix2(roms_hip_ctx *c);
int run_rufrc_sums(roms_hip_ctx *c);
int run_uv3dmix2_col(roms_hip_ctx *c);
int run_swdk(roms_hip_ctx *c);
int run_pre_t3(roms_hip_ctx *c);
int fetch_diag(roms_hip_ctx *c, const double *d_out, double *out) { return d2h(out, d_out, 16 * sizeof(double), c->stream); }
int run_diag_async(roms_hip_ctx *c, double *d_out, int part);   // g_diag.cpp

// ------------------------------------------------------------------------------- side stream
#ifdef ROMS_CPU_EMU
void side_mark(roms_hip_ctx *) {}
void side_begin(roms_hip_ctx *) {}
void side_end(roms_hip_ctx *) {}
void side_join(roms_hip_ctx *) {}
void side_point(roms_hip_ctx *) {}
void side_join_point(roms_hip_ctx *) {}
bool lanes_on(roms_hip_ctx *) { return false; }
void lane_record(roms_hip_ctx *, int) {}
void lane_wait(roms_hip_ctx *, int) {}
#else
static bool side_on(roms_hip_ctx *c) { return c->overlap && !c->profile && g_kprof_mode != 1 && g_kprof_mode != 4; }
// side_mark: the fork point on the main stream.  The host then enqueues the main-stream work that
// follows the fork FIRST and the side-stream work (side_begin .. side_end) after it: the main queue
// is the critical path and must not wait for the host to finish enqueuing the side chain.
void side_mark(roms_hip_ctx *c) {
  if (!side_on(c)) return;
  (void)hipEventRecord(c->ev_fork, c->stream);
}
void side_begin(roms_hip_ctx *c) {
  if (!side_on(c)) return;
  (void)hipStreamWaitEvent(c->stream2, c->ev_fork, 0);
  std::swap(c->stream, c->stream2);
}
void side_end(roms_hip_ctx *c) {
  if (!side_on(c)) return;
  (void)hipEventRecord(c->ev_join, c->stream);   // c->stream is the side stream here
  std::swap(c->stream, c->stream2);
}
void side_join(roms_hip_ctx *c) {
  if (!side_on(c)) return;
  (void)hipStreamWaitEvent(c->stream, c->ev_join, 0);
}
// a join point in the middle of a side region: side_point (called between side_begin and side_end) marks it,
// side_join_point makes the main stream wait for the side work enqueued before the mark only
void side_point(roms_hip_ctx *c) {
  if (!side_on(c)) return;
  (void)hipEventRecord(c->ev_point, c->stream);   // c->stream is the side stream here
}
void side_join_point(roms_hip_ctx *c) {
  if (!side_on(c)) return;
  (void)hipStreamWaitEvent(c->stream, c->ev_point, 0);
}
// Lanes (main3d_one's late-predictor schedule): the caller points c->stream at the stream a kernel is to run on;
// lane_record marks the current position of that stream with event e, lane_wait makes it wait for event e.
// Without side streams (profiling, per-kernel timing, ROMS_HIP_OVERLAP=0) c->stream never changes and both do
// nothing: the host's enqueue order is a valid serial order.
bool lanes_on(roms_hip_ctx *c) { return side_on(c); }
void lane_record(roms_hip_ctx *c, int e) {
  if (!side_on(c)) return;
  (void)hipEventRecord(c->ev_lane[e], c->stream);
}
void lane_wait(roms_hip_ctx *c, int e) {
  if (!side_on(c)) return;
  (void)hipStreamWaitEvent(c->stream, c->ev_lane[e], 0);
}
#endif

// ------------------------------------------------------------------------------ region timing
RegionTimer::RegionTimer(roms_hip_ctx *c_, int id_) : c(c_), id(id_) {
#ifndef ROMS_CPU_EMU
  if (c->profile) (void)hipEventRecord(c->ev0, c->stream);
#endif
}
RegionTimer::~RegionTimer() {
#ifndef ROMS_CPU_EMU
  if (c->profile) {
    (void)hipEventRecord(c->ev1, c->stream);
    (void)hipEventSynchronize(c->ev1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, c->ev0, c->ev1);
    c->regions[id].seconds += 1.0e-3 * (double)ms;
  }
#endif
  c->regions[id].calls += 1;
}
extern "C" int roms_hip_profile(roms_hip_ctx *c, int enable) {
  c->profile = enable != 0;
  memset(c->regions, 0, sizeof(c->regions));
  return 0;
}
extern "C" int roms_hip_region_seconds(roms_hip_ctx *c, int region, double *seconds, long *calls) {
  if (region < 0 || region >= 96) return 8;
  if (seconds) *seconds = c->regions[region].seconds;
  if (calls) *calls = c->regions[region].calls;
  return 0;
}

// -------------------------------------------------------------------- per-kernel timing
// Process-wide table keyed by the kernel's name (the LAUNCH_* macros pass it).  Mode 1 brackets
// every launch with an event pair and waits for it (a breakdown pass; slows the step down).
// Mode 2 brackets only launches of the selected kernel with event pairs taken from a pool and
// resolves them when the pool is full or when the table is read: no host synchronisation inside
// the timed region.  roms_hip_kprof_batch(n): one event pair around n consecutive launches of the
// selected kernel (back-to-back launches of the barotropic loop), so that the two event markers do
// not inflate a 10-microsecond kernel; a run interrupted by another launch is discarded, and if that
// happens before any run has completed the mode falls back to one pair per launch.
#ifndef ROMS_CPU_EMU
int g_kprof_mode = 0;
thread_local hipEvent_t g_kp_start = nullptr, g_kp_stop = nullptr;
namespace {
struct KSlot { char name[48]; double seconds; long calls; };
const int KMAXSLOT = 128, KPOOL = 8192;
KSlot g_kslot[KMAXSLOT];
int g_nkslot = 0;
char g_kselect[48] = "";
hipEvent_t g_kev[2 * KPOOL];
int g_kev_slot[KPOOL], g_kev_n[KPOOL];
int g_kev_made = 0, g_kev_used = 0;
int g_kbatch = 1, g_kb_open = -1, g_kb_count = 0, g_kb_done = 0;
bool g_kb_broken = false;
int kslot_of(const char *name) {
  for (int i = 0; i < g_nkslot; i++)
    if (!strcmp(g_kslot[i].name, name)) return i;
  if (g_nkslot >= KMAXSLOT) return -1;
  KSlot &k = g_kslot[g_nkslot];
  strncpy(k.name, name, sizeof(k.name) - 1); k.name[sizeof(k.name) - 1] = 0;
  k.seconds = 0.0; k.calls = 0;
  return g_nkslot++;
}
void kprof_resolve() {
  if (g_kb_open >= 0) { g_kev_used = g_kb_open; g_kb_open = -1; }   // an unfinished run is dropped
  for (int i = 0; i < g_kev_used; i++) {
    (void)hipEventSynchronize(g_kev[2 * i + 1]);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, g_kev[2 * i], g_kev[2 * i + 1]);
    g_kslot[g_kev_slot[i]].seconds += 1.0e-3 * (double)ms;
    g_kslot[g_kev_slot[i]].calls += g_kev_n[i];
  }
  g_kev_used = 0;
}
}  // namespace
static int g_kstride = 1, g_kcount = 0, g_kwin = 1;
extern "C" int roms_hip_kprof_stride(int every) { g_kstride = every > 0 ? every : 1; g_kcount = 0; g_kwin = 1; return 0; }
extern "C" int roms_hip_kprof_window(int on, int period) { g_kstride = period > 0 ? period : 1; g_kwin = on > 0 ? on : 1; g_kcount = 0; return 0; }
extern "C" int roms_hip_kprof_batch(int n) { kprof_resolve(); g_kbatch = n > 0 ? n : 1; g_kb_done = 0; return 0; }
int kprof_begin(const char *name, hipStream_t stream) {
  if (g_kprof_mode == 3) {
    if (strcmp(name, g_kselect) || (g_kcount++ % g_kstride) >= g_kwin) return -1;
  }
  // mode 4: EVERY launch carries a start / stop pair filled by the launch itself (asynchronous; the side streams are
  // off as in mode 1, so a kernel's duration is its own)
  if (g_kprof_mode == 2) {
    if (strcmp(name, g_kselect)) {
      if (g_kb_open >= 0) g_kb_broken = true;
      return -1;
    }
    if (g_kbatch > 1) {
      if (g_kb_open >= 0 && g_kb_broken) {   // interrupted run: dropped; if no run ever completed, per-launch pairs from now on
        g_kev_used = g_kb_open; g_kb_open = -1;
        if (!g_kb_done) g_kbatch = 1;
      }
      else if (g_kb_open >= 0) return g_kb_open;   // inside a run
    }
    if (g_kcount++ % g_kstride) return -1;     // sample every g_kstride-th launch (or run) of the selected kernel
  }
  static const bool trace = getenv("ROMS_HIP_TRACE") != nullptr;   // debugging aid: name every launch
  if (trace) { fprintf(stderr, "launch %s\n", name); fflush(stderr); }
  int slot = kslot_of(name);
  if (slot < 0) return -1;
  if (g_kev_used >= KPOOL) kprof_resolve();
  if (g_kev_used >= g_kev_made) {
    (void)hipEventCreateWithFlags(&g_kev[2 * g_kev_made], hipEventDisableSystemFence);
    (void)hipEventCreateWithFlags(&g_kev[2 * g_kev_made + 1], hipEventDisableSystemFence);
    g_kev_made++;
  }
  int e = g_kev_used++;
  g_kev_slot[e] = slot;
  g_kev_n[e] = 1;
  if (g_kprof_mode == 3 || g_kprof_mode == 4) { g_kp_start = g_kev[2 * e]; g_kp_stop = g_kev[2 * e + 1]; return e; }   // the launch itself fills the pair
  (void)hipEventRecord(g_kev[2 * e], stream);
  if (g_kprof_mode == 2 && g_kbatch > 1) { g_kb_open = e; g_kb_count = 0; g_kb_broken = false; }
  return e;
}
void kprof_end(int e, hipStream_t stream) {
  if (g_kprof_mode == 3 || g_kprof_mode == 4) { g_kp_start = nullptr; g_kp_stop = nullptr; return; }
  if (g_kprof_mode == 2 && g_kb_open == e) {
    if (++g_kb_count < g_kbatch) return;       // the run goes on
    g_kev_n[e] = g_kbatch;
    g_kb_open = -1;
    g_kb_done++;
  }
  (void)hipEventRecord(g_kev[2 * e + 1], stream);
  if (g_kprof_mode == 1) kprof_resolve();
}
extern "C" int roms_hip_kprof(int mode, const char *kernel) {
  kprof_resolve();
  g_kbatch = 1;
  g_kprof_mode = mode;
  g_nkslot = 0;
  g_kselect[0] = 0;
  if (kernel) { strncpy(g_kselect, kernel, sizeof(g_kselect) - 1); g_kselect[sizeof(g_kselect) - 1] = 0; }
  return 0;
}
extern "C" int roms_hip_kprof_get(int index, char *name, int name_len, double *seconds, long *calls) {
  kprof_resolve();
  if (index < 0 || index >= g_nkslot) return 8;
  if (name && name_len > 0) { strncpy(name, g_kslot[index].name, (size_t)name_len - 1); name[name_len - 1] = 0; }
  if (seconds) *seconds = g_kslot[index].seconds;
  if (calls) *calls = g_kslot[index].calls;
  return 0;
}
#else
extern "C" int roms_hip_kprof(int, const char *) { return 0; }
extern "C" int roms_hip_kprof_stride(int) { return 0; }
extern "C" int roms_hip_kprof_window(int, int) { return 0; }
extern "C" int roms_hip_kprof_batch(int) { return 0; }
extern "C" int roms_hip_kprof_get(int, char *, int, double *, long *) { return 8; }
#endif

// ------------------------------------------------------------------------------------- halo
void launch_halo(roms_hip_ctx *c, double *A, int nk, int bc, char gtype) {
  HaloSpec sp = {A, nk, bc, gtype};
  launch_halo_multi(c, &sp, 1);
}
// ---- transports ---------------------------------------------------------------------------
#ifndef ROMS_CPU_EMU
#include <dlfcn.h>
#include <unistd.h>
#include <rccl/rccl.h>
namespace {
struct RcclApi {
  void *lib;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *);
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int);
  ncclResult_t (*CommDestroy)(ncclComm_t);
  ncclResult_t (*CommCount)(const ncclComm_t, int *);
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
  ncclResult_t (*GroupStart)();
  ncclResult_t (*GroupEnd)();
  const char *(*GetErrorString)(ncclResult_t);
} g_rccl = {};
// RCCL is opened on first use so that single-GPU runs do not depend on it; a copy that the
// process already loaded (PyTorch ships one) is reused.
int rccl_load() {
  if (g_rccl.lib) return 0;
  const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void *h = nullptr;
  for (const char *n : names) { h = dlopen(n, RTLD_NOW | RTLD_NOLOAD); if (h) break; }
  if (!h) for (const char *n : names) { h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
  if (!h) { set_error(std::string("cannot load RCCL: ") + dlerror()); return 2; }
#define RSYM(f) *(void **)(&g_rccl.f) = dlsym(h, "nccl" #f); if (!g_rccl.f) { set_error("RCCL symbol nccl" #f " missing"); return 2; }
  RSYM(GetUniqueId) RSYM(CommInitRank) RSYM(CommDestroy) RSYM(CommCount) RSYM(Send) RSYM(Recv) RSYM(GroupStart) RSYM(GroupEnd)
  RSYM(GetErrorString)
#undef RSYM
  g_rccl.lib = h;
  return 0;
}
int rcclfail(ncclResult_t r, const char *what) {
  if (r == ncclSuccess) return 0;
  set_error(std::string(what) + ": " + g_rccl.GetErrorString(r));
  return 2;
}
}  // namespace
extern "C" int roms_hip_rccl_unique_id(void *id128) {
  if (!id128) return 8;
  if (rccl_load()) return 2;
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId size");
  return rcclfail(g_rccl.GetUniqueId((ncclUniqueId *)id128), "ncclGetUniqueId");
}
extern "C" int roms_hip_comm_rccl(roms_hip_ctx *c, const void *id128, int nranks, int rank) {
  if (!c || !id128) return 8;
  if (nranks != c->cfg.NtileI * c->cfg.NtileJ || rank != c->cfg.tile) {
    set_error("roms_hip_comm_rccl: need nranks == NtileI*NtileJ and rank == tile");
    return 5;
  }
  if (rccl_load()) return 2;
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  ncclComm_t comm = nullptr;
  if (rcclfail(g_rccl.CommInitRank(&comm, nranks, id, rank), "ncclCommInitRank")) return 2;
  c->comm.nccl = (void *)comm;
  return 0;
}
extern "C" long roms_hip_rccl_ranks(roms_hip_ctx *c) {
  if (!c || !c->comm.nccl || !g_rccl.lib) return 0;
  int n = 0;
  if (g_rccl.CommCount((ncclComm_t)c->comm.nccl, &n) != ncclSuccess) return -1;
  return n;
}
// ---- mailbox transport: receive slots in uncached device memory, mapped by the neighbours ----
namespace {
struct PeerBlob {              // 128 bytes, what roms_hip_peer_export hands out
  hipIpcMemHandle_t handle;    // 64 bytes
  long long pid;
  unsigned long long raw;      // the slab's address in the exporting process (neighbour contexts of the same process)
  unsigned long long bytes;
  int device, planes;
  unsigned magic;
  // round 6: what the persistent barotropic loop of a neighbour needs to address my rim planes and ring (k_step2d_loop.h),
  // and the physical identity of my device (two ranks on one GPU are recognised by it, not by the ordinal)
  short LBi, LBj, ni, nj, nbx2, nby2;
  unsigned rim_off256, ring_off256;    // offsets in the slab in units of 256 bytes (0: none)
  unsigned busid;
  unsigned pad;
};
static_assert(sizeof(PeerBlob) == 128, "PeerBlob size");
const unsigned PEER_MAGIC = 0x524d5042u;
const size_t PEER_TABLE = 0, PEER_WORDS = 1024;   // slot table; arrival words [channel][direction][plane]; slots from the next 4 KB boundary
const int g_opp[8] = {1, 0, 3, 2, 7, 6, 5, 4};
}  // namespace
extern "C" int roms_hip_peer_export(roms_hip_ctx *c, void *blob128) {
  if (!c || !blob128) return 8;
  TileComm &m = c->comm;
  if (!m.peer_slab) {
    const DGrid &G = c->G;
    const char *ep = getenv("ROMS_HIP_PEER_PLANES");
    m.peer_planes = ep ? atoi(ep) : 8 * (G.N + 1);
    size_t off = (PEER_WORDS + (size_t)(8 * PEER_NCH) * m.peer_planes * sizeof(unsigned long long) + 4095) & ~(size_t)4095;
    for (int ch = 0; ch < PEER_NCH; ch++)
      for (int par = 0; par < 2; par++)
        for (int d = 0; d < 8; d++) {
          m.peer_off[ch][par][d] = off;
          if (m.nbr[d] < 0) continue;
          const size_t gw = c->pair_mt ? (size_t)B2D_GL : 3;       // widest strip an exchange point carries
          const size_t w = d < 2 ? gw * G.nj : (d < 4 ? gw * G.ni : gw * gw);
          off += ((size_t)m.peer_planes * w * sizeof(double) + 255) & ~(size_t)255;
        }
    m.loop_rim_off = m.loop_ring_off = 0;
    if (c->pair_mt) {                                             // rim planes [4][3][nij] x 16 bytes: the persistent barotropic loop uses two sets (k_step2d_loop.h), the pair launches four (k_step2d_pair.h: Step2dPairArgs)
      step2d_loop_dims(c, m.loop_nb2[0], m.loop_nb2[1]);
      m.loop_rim_off = off;
      off += ((size_t)12 * (size_t)G.nij * 16 + 255) & ~(size_t)255;
    }
    m.peer_bytes = off;
    {
      char bus[64] = {0};
      unsigned dom = 0, b = 0, dv = 0, fn = 0;
      m.busid = 0xFFFFFFFFu;
      if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), c->cfg.device) == hipSuccess && sscanf(bus, "%x:%x:%x.%x", &dom, &b, &dv, &fn) >= 3)
        m.busid = (dom << 16) | (b << 8) | (dv << 3) | fn;
    }
    if (hipfail(hipExtMallocWithFlags(&m.peer_slab, m.peer_bytes, hipDeviceMallocUncached), "hipExtMallocWithFlags (mailbox slab)")) return 2;
    if (hipfail(hipMemset(m.peer_slab, 0, m.peer_bytes), "hipMemset")) return 2;
    if (hipfail(hipMemcpy((char *)m.peer_slab + PEER_TABLE, m.peer_off, sizeof(m.peer_off), hipMemcpyHostToDevice), "hipMemcpy")) return 2;
    if (hipfail(hipHostMalloc((void **)&m.peer_err, sizeof(unsigned long long), hipHostMallocDefault), "hipHostMalloc")) return 2;
    *m.peer_err = 0;
    if (hipfail(hipDeviceSynchronize(), "hipDeviceSynchronize")) return 2;
  }
  PeerBlob b;
  memset(&b, 0, sizeof(b));
  if (hipfail(hipIpcGetMemHandle(&b.handle, m.peer_slab), "hipIpcGetMemHandle")) return 2;
  b.pid = (long long)getpid();
  b.raw = (unsigned long long)(uintptr_t)m.peer_slab;
  b.bytes = m.peer_bytes;
  b.device = c->cfg.device; b.planes = m.peer_planes; b.magic = PEER_MAGIC;
  b.LBi = (short)c->G.LBi; b.LBj = (short)c->G.LBj; b.ni = (short)c->G.ni; b.nj = (short)c->G.nj;
  b.nbx2 = (short)m.loop_nb2[0]; b.nby2 = (short)m.loop_nb2[1];
  b.rim_off256 = (unsigned)(m.loop_rim_off >> 8); b.ring_off256 = 0;
  b.busid = m.busid;
  memcpy(blob128, &b, sizeof(b));
  return 0;
}
extern "C" int roms_hip_comm_peer(roms_hip_ctx *c, const void *blobs128, int nranks, int rank) {
  if (!c || !blobs128) return 8;
  TileComm &m = c->comm;
  if (nranks != c->cfg.NtileI * c->cfg.NtileJ || rank != c->cfg.tile) {
    set_error("roms_hip_comm_peer: need nranks == NtileI*NtileJ and rank == tile");
    return 5;
  }
  if (!m.peer_slab) { set_error("roms_hip_comm_peer: call roms_hip_peer_export on every rank first"); return 8; }
  const PeerBlob *B = (const PeerBlob *)blobs128;
  for (int d = 0; d < 8; d++) {
    m.peer_map[d] = nullptr; m.peer_opened[d] = false;
    const int r = m.nbr[d];
    if (r < 0) continue;
    if (r >= nranks || B[r].magic != PEER_MAGIC) { set_error("roms_hip_comm_peer: bad blob for a neighbour rank"); return 8; }
    if (B[r].planes != m.peer_planes) { set_error("roms_hip_comm_peer: ranks differ in slot capacity"); return 5; }
    // a neighbour on the SAME physical device (ranks sharing one GPU: test set-ups), whatever ordinal each process knows it by
    if (B[r].pid != (long long)getpid() && (m.busid != 0xFFFFFFFFu && B[r].busid != 0xFFFFFFFFu ? B[r].busid == m.busid : B[r].device == c->cfg.device)) m.peer_shared = true;
    m.ngeom[d] = {B[r].LBi, B[r].LBj, B[r].ni, B[r].nj, B[r].nbx2, B[r].nby2, (size_t)B[r].rim_off256 << 8, (size_t)B[r].ring_off256 << 8, B[r].busid};
    for (int e = 0; e < d; e++)
      if (m.nbr[e] == r) { m.peer_map[d] = m.peer_map[e]; break; }
    if (!m.peer_map[d]) {
      if (B[r].pid == (long long)getpid()) m.peer_map[d] = (void *)(uintptr_t)B[r].raw;   // a context of this process (or myself)
      else {
        if (hipfail(hipIpcOpenMemHandle(&m.peer_map[d], B[r].handle, hipIpcMemLazyEnablePeerAccess), "hipIpcOpenMemHandle (neighbour's mailbox)")) return 2;
        m.peer_opened[d] = true;
      }
    }
    size_t table[PEER_NCH][2][8];
    if (hipfail(hipMemcpy(table, (char *)m.peer_map[d] + PEER_TABLE, sizeof(table), hipMemcpyDeviceToHost), "hipMemcpy (neighbour's slot table)")) return 2;
    for (int ch = 0; ch < PEER_NCH; ch++)
      for (int par = 0; par < 2; par++) m.peer_noff[d][ch][par] = table[ch][par][g_opp[d]];
  }
  for (int ch = 0; ch < PEER_NCH; ch++) m.peer_seq[ch] = 0;
  // (the arrival words of my slots and the elements of my rim planes carry the numbers of the exchanges that wrote them: a transport installed again
  // counts from one again, so whatever an earlier installation left there goes -- the caller's barrier behind this call keeps
  // the neighbours from writing before everyone is through, roms_amd/tiling.py:_install_peer)
  {
    if (m.peer_bytes > PEER_WORDS && hipfail(hipMemset((char *)m.peer_slab + PEER_WORDS, 0, m.peer_bytes - PEER_WORDS), "hipMemset (mailbox slab)")) return 2;
    if (hipfail(hipDeviceSynchronize(), "hipDeviceSynchronize")) return 2;
    c->loop_epoch = 0;
    if (c->loop_flags) {
      int nb2x = 0, nb2y = 0;
      step2d_loop_dims(c, nb2x, nb2y);
      (void)hipMemset(c->loop_flags, 0, (size_t)nb2x * nb2y * 16 * sizeof(unsigned));
    }
  }
  m.peer_on = true;
  c->rim_refused = false;
  c->loop_state = 0;                    // (the persistent barotropic loop of a multi-tile context needs the mailbox: decided again)
  // the step is arranged on four streams, every exchange in the stream of its producer on that stream's channel (main3d_around_loop)
  if (c->pair_mt && !getenv("ROMS_HIP_XASYNC") && !(getenv("ROMS_HIP_MT_LANES") && getenv("ROMS_HIP_MT_LANES")[0] == '0')) { halo_fence(c, FG_ALL); c->x_async = false; c->rim_split = false; }
  // (ranks sharing one device: smaller blocks, so that the waiting unpack blocks of ALL of them fit the device beside the
  // pack blocks they wait for -- 8 ranks x 200 planes x 1024 threads exceed an MI355X's 512 K resident threads)
  { const char *et = getenv("ROMS_HIP_PEER_THREADS"); c->peer_threads = et ? atoi(et) : (m.peer_shared ? 256 : 1024); }
  return 0;
}
static void comm_destroy(roms_hip_ctx *c) {
  if (c->comm.nccl && g_rccl.lib) (void)g_rccl.CommDestroy((ncclComm_t)c->comm.nccl);
  c->comm.nccl = nullptr;
  TileComm &m = c->comm;
  for (int d = 0; d < 8; d++)
    if (m.peer_opened[d]) { (void)hipIpcCloseMemHandle(m.peer_map[d]); m.peer_opened[d] = false; }
  if (m.peer_slab) (void)hipFree(m.peer_slab);
  if (m.peer_err) (void)hipHostFree(m.peer_err);
  m.peer_slab = nullptr; m.peer_err = nullptr; m.peer_on = false;
}
#else
extern "C" int roms_hip_peer_export(roms_hip_ctx *, void *) { set_error("the mailbox transport is not part of the CPU-emulated test build"); return 2; }
extern "C" int roms_hip_comm_peer(roms_hip_ctx *, const void *, int, int) {
  set_error("the mailbox transport is not part of the CPU-emulated test build");
  return 2;
}
extern "C" long roms_hip_rccl_ranks(roms_hip_ctx *) { return 0; }
extern "C" int roms_hip_rccl_unique_id(void *) { set_error("RCCL is not part of the CPU-emulated test build"); return 2; }
extern "C" int roms_hip_comm_rccl(roms_hip_ctx *, const void *, int, int) {
  set_error("RCCL is not part of the CPU-emulated test build");
  return 2;
}
static void comm_destroy(roms_hip_ctx *) {}
#endif
extern "C" int roms_hip_set_exchange(roms_hip_ctx *c, roms_hip_exchange_fn fn, void *user) {
  if (!c) return 8;
  c->comm.fn = fn;
  c->comm.user = user;
  return 0;
}
extern "C" long roms_hip_exchange_count(roms_hip_ctx *c) { return c ? c->comm.nexchanges : 0; }

// ---- overlap of halo exchanges with compute ------------------------------------------------------
// field group of a device array (any plane of it)
static unsigned group_of(const roms_hip_ctx *c, const double *p) {
  static const struct { const char *name; unsigned g; } map[] = {
      {"sustr", FG_FLUX}, {"svstr", FG_FLUX}, {"bustr", FG_FLUX}, {"bvstr", FG_FLUX}, {"stflx", FG_FLUX}, {"btflx", FG_FLUX},
      {"stflux", FG_FLUX}, {"btflux", FG_FLUX}, {"srflx", FG_FLUX}, {"Uwind", FG_FLUX}, {"Vwind", FG_FLUX}, {"Tair", FG_FLUX},
      {"Pair", FG_FLUX}, {"Hair", FG_FLUX}, {"rain", FG_FLUX}, {"cloud", FG_FLUX}, {"lhflx", FG_FLUX}, {"shflx", FG_FLUX},
      {"lrflx", FG_FLUX}, {"evap", FG_FLUX},
      {"rho", FG_RHO}, {"pden", FG_RHO}, {"rhoA", FG_RHO}, {"rhoS", FG_RHO}, {"bvf", FG_RHO}, {"alpha", FG_RHO}, {"beta", FG_RHO},
      {"Huon", FG_MF}, {"Hvom", FG_MF}, {"W", FG_W}, {"wvel", FG_WVEL},
      {"Akv", FG_AK}, {"Akt", FG_AK}, {"ghats", FG_AK}, {"hsbl", FG_AK}, {"tke", FG_AK}, {"gls", FG_AK},
      {"t", FG_T}, {"u", FG_UV}, {"v", FG_UV},
      {"zeta", FG_2D}, {"ubar", FG_2D}, {"vbar", FG_2D}, {"rzeta", FG_2D}, {"rubar", FG_2D}, {"rvbar", FG_2D},
      {"Zt_avg1", FG_AVG}, {"DU_avg1", FG_AVG}, {"DU_avg2", FG_AVG}, {"DV_avg1", FG_AVG}, {"DV_avg2", FG_AVG},
      {"Hz", FG_HZ}, {"z_r", FG_HZ}, {"z_w", FG_HZ},
      {"ru", FG_R}, {"rv", FG_R}, {"rufrc", FG_R}, {"rvfrc", FG_R}};
  for (int k = 0; k < g_nfields; k++) {
    const double *base = *(double *const *)((const char *)&c->F + g_fields[k].offset);
    if (!base || p < base || p >= base + field_elems(c, g_fields[k].kind)) continue;
    if (!strcmp(g_fields[k].name, "t")) {                   // the predictor level t(:,:,:,3,:) is a group of its own
      const size_t lev = (size_t)(p - base) / ((size_t)c->G.nij * (size_t)c->G.N) % 3;
      return lev == 2 ? FG_T3 : FG_T;
    }
    for (const auto &m : map)
      if (!strcmp(m.name, g_fields[k].name)) return m.g;
    return FG_OTHER;
  }
  return FG_OTHER;
}
void halo_fence(roms_hip_ctx *c, unsigned groups) {
#ifndef ROMS_CPU_EMU
  const unsigned need = groups & c->x_pending;
  if (!need) return;
  bool waited[32] = {};
  for (int g = 0; g < FG_NGROUPS; g++) {
    if (!(need & (1u << g))) continue;
    const int e = c->x_event_of[g];
    if (e >= 0 && !waited[e]) {
      // both compute streams: the side stream's kernels are ordered behind the main stream's fork only
      (void)hipStreamWaitEvent(c->stream, c->ev_x[e], 0);
      (void)hipStreamWaitEvent(c->stream2, c->ev_x[e], 0);
      waited[e] = true;
    }
  }
  c->x_pending &= ~need;
#else
  (void)c; (void)groups;
#endif
}

// Boundary fills and strip exchange of the fields of one exchange point (multi-tile contexts):
// fill + pack launch, one group of sends/receives with the up to eight neighbours, unpack launch.
static int exchange_all(roms_hip_ctx *c, const HaloArgs &h, int planes) {
  TileComm &m = c->comm;
  if (!m.fn && !m.nccl && !m.peer_on) {
    set_error("multi-tile context without a transport: call roms_hip_comm_rccl, roms_hip_comm_peer or roms_hip_set_exchange first");
    return 8;
  }
  const DGrid &G = h.G;                    // (c->G with the strip widths of this exchange point)
  const size_t lines = (size_t)(G.ni > G.nj ? G.ni : G.nj);
  const size_t gw = c->pair_mt ? (size_t)B2D_GL : 3;
  const size_t need = (size_t)planes * lines * gw;
  if (m.peer_on && planes > m.peer_planes) {
    set_error("mailbox transport: an exchange point carries more planes than a slot holds (ROMS_HIP_PEER_PLANES)");
    return 8;
  }
  if (need > m.cap && !m.peer_on) {
    (void)dsync(c->stream);
#ifndef ROMS_CPU_EMU
    if (c->xstream) (void)hipStreamSynchronize(c->xstream);   // an exchange in flight still uses the old buffers
#endif
    for (int k = 0; k < 8; k++) {
      if (m.sbuf[k]) dfree(m.sbuf[k]);
      if (m.rbuf[k]) dfree(m.rbuf[k]);
      m.sbuf[k] = m.rbuf[k] = nullptr;
      if (m.nbr[k] < 0) continue;
      const size_t n = k < 4 ? need : (size_t)planes * gw * gw;
      void *p = nullptr, *q = nullptr;
      if (dmalloc(&p, n * sizeof(double)) || dmalloc(&q, n * sizeof(double))) return 2;
      m.sbuf[k] = (double *)p; m.rbuf[k] = (double *)q;
    }
#ifndef ROMS_CPU_EMU
    (void)hipDeviceSynchronize();   // zero fills of the new buffers (null stream) before use on c->stream
#endif
    m.cap = need;
  }
  // message sizes (k_halo.h:xchg_rect): planes x width along xi x width along eta, a width being all
  // local lines across the direction of travel, Nghost lines for a message going to the low side
  // (the receiver's high ghost zone) and three for one going to the high side (its low ghost zone)
  const int gl = G.xgl, gh = G.xgh;
  auto count = [&](int d, bool sending) -> long {
    const int dx = (d == 0 || d == 4 || d == 6) ? -1 : ((d == 1 || d == 5 || d == 7) ? 1 : 0);
    const int dy = (d == 2 || d == 4 || d == 5) ? -1 : ((d == 3 || d == 6 || d == 7) ? 1 : 0);
    // sending to the low side: my first gh lines (the receiver's HIGH ghost zone); receiving from the low side: gl lines
    const long wx = dx == 0 ? G.ni : ((dx < 0) == sending ? gh : gl);
    const long wy = dy == 0 ? G.nj : ((dy < 0) == sending ? gh : gl);
    return (long)planes * wx * wy;
  };
  XchgArgs a;
  a.G = G; a.nitems = h.nitems;
  for (int k = 0; k < HALO_MAXITEMS; k++) a.it[k] = h.it[k];
  // Stream of the exchange.  Asynchronous mode: its own stream, ordered behind everything enqueued on the
  // compute stream so far (the producing kernel); kernels enqueued afterwards run beside it until one of them
  // touches a field group of this exchange (halo_fence).  The exchange kernels read the strips of the tile's own
  // points and write ghost points, boundary points and the buffers only.
  kstream_t xs = c->stream;
  unsigned groups = 0;
  for (int k = 0; k < h.nitems; k++) groups |= group_of(c, h.it[k].A);
#ifndef ROMS_CPU_EMU
  // Only exchanges that have something to hide behind go to the exchange stream: the 3-D fields (a cross-stream
  // dependency costs ~3 us per hop, four hops per exchange; measured on the self-exchange path, BENCHMARK1:
  // every exchange asynchronous 2.90 -> 3.85 ms per step).  The barotropic exchanges -- each sub-step needs the
  // previous one's ghost points at once -- stay on the compute stream, behind whatever the exchange stream
  // still holds (the send/receive buffers and the communicator are shared).
  const bool async = c->x_async && c->x_tail && planes >= c->x_min_planes && !(groups & ((c->x_2d_ok ? 0 : FG_2D) | FG_AVG));
  if (async) {
    halo_fence(c, groups);            // an earlier exchange of the same fields: keep the two in order on both sides
    xs = c->xstream;
    (void)hipEventRecord(c->ev_xprod, c->stream);
    (void)hipStreamWaitEvent(xs, c->ev_xprod, 0);
  } else if (c->x_async) {
    halo_fence(c, FG_ALL);
  }
#else
  const bool async = false;
#endif
#ifndef ROMS_CPU_EMU
  if (m.peer_on) {
    // two launches: the pack kernel stores into the neighbours' slots and releases their arrival words, the
    // unpack kernel waits for mine (k_halo.h:xchg_peer_pack/unpack)
    // (the channel of the issuing stream: a lane of the schedule around the barotropic loop has its own)
    const int ch = async ? 1 : (xs == c->stream2 ? 2 : (xs == c->stream3 ? 3 : (xs == c->stream4 ? 4 : 0)));
    const unsigned long long seq = ++m.peer_seq[ch];
    const int par = (int)(seq & 1);
    static long long timeout = 0;
    if (!timeout) { const char *et = getenv("ROMS_HIP_PEER_TIMEOUT"); timeout = (long long)((et ? atof(et) : 20.0) * 1.0e8); }
    static const char *eboth = getenv("ROMS_HIP_PEER_BOTH");
    // Pack and unpack in one launch (k_halo.h:xchg_peer_both) needs the neighbours' kernels RESIDENT while this one waits:
    // true with one GPU per rank (a launch is at most peer_planes = 8 (N+1) blocks).  Ranks that SHARE a device (test
    // set-ups: 4-8 processes on one GPU) can fill it with blocks that wait for blocks the device has no room to start
    // -- round 5, config 5 in 2x4 on one MI355X: waits ran into the time limit, garbage in the ghost zones -- so they take
    // the two launches (the pack kernel never waits).  ROMS_HIP_PEER_BOTH=0/1 forces.
    if (eboth ? eboth[0] != '0' : !m.peer_shared) {
      XchgPeerBothArgs ba;
      ba.x = a;
      ba.x.unpack = 0; ba.x.fill = 1;
      ba.sp.seq = seq; ba.sp.err = m.peer_err; ba.sp.timeout = timeout;
      ba.su = ba.sp;
      for (int d = 0; d < 8; d++) {
        const bool on = m.nbr[d] >= 0;
        ba.x.buf[d] = on ? (double *)((char *)m.peer_map[d] + m.peer_noff[d][ch][par]) : nullptr;
        ba.sp.word[d] = on ? (unsigned long long *)((char *)m.peer_map[d] + PEER_WORDS) + (size_t)(ch * 8 + g_opp[d]) * m.peer_planes : nullptr;
        ba.ubuf[d] = on ? (double *)((char *)m.peer_slab + m.peer_off[ch][par][d]) : nullptr;
        ba.su.word[d] = on ? (unsigned long long *)((char *)m.peer_slab + PEER_WORDS) + (size_t)(ch * 8 + d) * m.peer_planes : nullptr;
      }
      KPROF_WRAP(xchg_peer_both, xs, ROMS_LAUNCH(xchg_peer_both, dim3(1, 1, (unsigned)planes), dim3(c->peer_threads), 0, xs, ba));
    } else {
    XchgPeerArgs pa;
    pa.x = a;
    pa.s.seq = seq; pa.s.err = m.peer_err; pa.s.timeout = timeout;
    pa.x.unpack = 0; pa.x.fill = 1;
    for (int d = 0; d < 8; d++) {
      const bool on = m.nbr[d] >= 0;
      pa.x.buf[d] = on ? (double *)((char *)m.peer_map[d] + m.peer_noff[d][ch][par]) : nullptr;
      pa.s.word[d] = on ? (unsigned long long *)((char *)m.peer_map[d] + PEER_WORDS) + (size_t)(ch * 8 + g_opp[d]) * m.peer_planes : nullptr;
    }
    KPROF_WRAP(xchg_peer_pack, xs, ROMS_LAUNCH(xchg_peer_pack, dim3(1, 1, (unsigned)planes), dim3(c->peer_threads), 0, xs, pa));
    pa.x.unpack = 1; pa.x.fill = 0;
    for (int d = 0; d < 8; d++) {
      const bool on = m.nbr[d] >= 0;
      pa.x.buf[d] = on ? (double *)((char *)m.peer_slab + m.peer_off[ch][par][d]) : nullptr;
      pa.s.word[d] = on ? (unsigned long long *)((char *)m.peer_slab + PEER_WORDS) + (size_t)(ch * 8 + d) * m.peer_planes : nullptr;
    }
    KPROF_WRAP(xchg_peer_unpack, xs, ROMS_LAUNCH(xchg_peer_unpack, dim3(1, 1, (unsigned)planes), dim3(c->peer_threads), 0, xs, pa));
    }
  } else {
#endif
  a.unpack = 0; a.fill = 1;
  for (int d = 0; d < 8; d++) a.buf[d] = m.nbr[d] >= 0 ? m.sbuf[d] : nullptr;
  LAUNCH_COOP(xchg_kernel, 1, 1, planes, 256, 0, xs, a);
  // Messages are issued in the order of their tag = direction of travel (0 eastward, 1 westward,
  // 2 northward, 3 southward, 4 NE, 5 NW, 6 SE, 7 SW): several messages between the same pair of
  // ranks (small periodic partitions) then match by issue order as well as by tag.
  static const int opp[8] = {1, 0, 3, 2, 7, 6, 5, 4};
  int sp[8], st[8], rp[8], rt[8], ns = 0, nr = 0;
  double *sb[8], *rb[8];
  long sc[8], rc[8];
  for (int tag = 0; tag < 8; tag++) {
    const int to = opp[tag];        // a message travelling in direction `tag` goes to that neighbour ...
    if (m.nbr[to] >= 0) { sp[ns] = m.nbr[to]; sb[ns] = m.sbuf[to]; sc[ns] = count(to, true); st[ns] = tag; ns++; }
    const int from = tag;           // ... and arrives from the neighbour on the opposite side
    if (m.nbr[from] >= 0) { rp[nr] = m.nbr[from]; rb[nr] = m.rbuf[from]; rc[nr] = count(from, false); rt[nr] = tag; nr++; }
  }
  if (m.fn) {
    int r = dsync(xs);
    if (r) return r;
    if (m.fn(m.user, ns, sp, sb, sc, st, nr, rp, rb, rc, rt)) { set_error("halo exchange transport failed"); return 2; }
  } else {
#ifndef ROMS_CPU_EMU
    ncclComm_t comm = (ncclComm_t)m.nccl;
    if (rcclfail(g_rccl.GroupStart(), "ncclGroupStart")) return 2;
    for (int k = 0; k < ns; k++)
      if (rcclfail(g_rccl.Send(sb[k], (size_t)sc[k], ncclDouble, sp[k], comm, xs), "ncclSend")) return 2;
    for (int k = 0; k < nr; k++)
      if (rcclfail(g_rccl.Recv(rb[k], (size_t)rc[k], ncclDouble, rp[k], comm, xs), "ncclRecv")) return 2;
    if (rcclfail(g_rccl.GroupEnd(), "ncclGroupEnd")) return 2;
#endif
  }
  a.unpack = 1; a.fill = 0;
  for (int d = 0; d < 8; d++) a.buf[d] = m.nbr[d] >= 0 ? m.rbuf[d] : nullptr;
  LAUNCH_COOP(xchg_kernel, 1, 1, planes, 256, 0, xs, a);
#ifndef ROMS_CPU_EMU
  }
  if (async) {
    const int e = c->ev_x_next;
    c->ev_x_next = (e + 1) % 32;
    (void)hipEventRecord(c->ev_x[e], xs);
    for (int g = 0; g < FG_NGROUPS; g++)
      if (groups & (1u << g)) c->x_event_of[g] = e;
    c->x_pending |= groups;
  }
#endif
  m.nexchanges++;
  return 0;
}

void launch_halo_multi(roms_hip_ctx *c, const HaloSpec *sp, int n) {
  HaloArgs a;
  a.G = c->G;
  if (c->x_wide && c->pair_mt) { a.G.xgl = B2D_GL; a.G.xgh = B2D_GH; }
  a.nitems = n;
  int planes = 0;
  for (int k = 0; k < HALO_MAXITEMS; k++) { a.it[k].A = nullptr; a.it[k].nk = 0; a.it[k].bc = BC_NONE; a.it[k].gtype = 0; }
  for (int k = 0; k < n; k++) {
    a.it[k].A = sp[k].A; a.it[k].nk = sp[k].nk; a.it[k].bc = sp[k].bc; a.it[k].gtype = (int)sp[k].gtype;
    planes += sp[k].nk;
  }
  if (c->has_exchange) {
    // neighbouring tiles on other GPUs
    if (exchange_all(c, a, planes)) c->comm_failed = true;
  } else {
    LAUNCH_COOP(halo_kernel, 1, 1, planes, 256, 0, c->stream, a);
  }
}

void launch_fill_only(roms_hip_ctx *c, const HaloSpec *sp, int n, const TB &X) {
  HaloArgs a;
  a.G = c->G;
  a.G.T = X;
  a.nitems = n;
  int planes = 0;
  for (int k = 0; k < HALO_MAXITEMS; k++) { a.it[k].A = nullptr; a.it[k].nk = 0; a.it[k].bc = BC_NONE; a.it[k].gtype = 0; }
  for (int k = 0; k < n; k++) {
    a.it[k].A = sp[k].A; a.it[k].nk = sp[k].nk; a.it[k].bc = sp[k].bc; a.it[k].gtype = (int)sp[k].gtype;
    planes += sp[k].nk;
  }
  if (!(X.west || X.east || X.south || X.north)) return;     // (no edge of the domain on this tile: nothing to fill)
  LAUNCH_COOP(halo_kernel, 1, 1, planes, 256, 0, c->stream, a);
}

// Self-check of the installed transport: `reps` exchanges of a work plane whose own points carry a code of their
// global indices and of the repetition; afterwards every ghost point that comes from a neighbour's own points must
// hold the code of the point it images (tile arrays are indexed globally, so that is its own index, wrapped where the
// domain is periodic).  Multi-GPU runs call it once after installing a transport (roms_amd/tiling.py): a mapping that
// does not carry the data, a stale cache line or a lost message shows up here, with a message, not in the fields.
// The probe covers what production uses (several planes per exchange: the slot / arrival-word indexing of the mailbox;
// both strip widths; the exchange stream's channel where asynchronous exchanges are on).
static int probe_once(roms_hip_ctx *c, int rep, int planes, bool wide, bool tail) {
  const DGrid &G = c->G;
  const TB &B = G.T;
  const TileComm &m = c->comm;
  const size_t n = (size_t)G.ni * (size_t)G.nj;
  std::vector<double> h(n * planes), g(n * planes);
  // a buffer of its own, `planes` consecutive planes (round 3 borrowed wrk3[0], whose (N+1)*NT planes a shallow grid or a
  // large ROMS_HIP_XASYNC_PLANES overran)
  struct Buf { void *p = nullptr; ~Buf() { if (p) { dfree(p); } } } buf;
  if (dmalloc(&buf.p, n * (size_t)planes * sizeof(double))) return 2;
  double *A = (double *)buf.p;
  auto at = [&](int i, int j) -> size_t { return (size_t)(i - G.LBi) + (size_t)(j - G.LBj) * (size_t)G.ni; };
  auto code = [&](int i, int j, int k) -> double {
    if (i < 1) i += G.Lm; else if (i > G.Lm) i -= G.Lm;
    if (j < 1) j += G.Mm; else if (j > G.Mm) j -= G.Mm;
    return ((double)rep * 64.0 + (double)k) * 67108864.0 + (double)j * 8192.0 + (double)i;
  };
  const int gl = wide ? B2D_GL : 3, gh = wide ? B2D_GH : G.Nghost;
  for (int k = 0; k < planes; k++) {
    for (size_t q = 0; q < n; q++) h[q + n * k] = -1.0;
    for (int j = B.Jstr; j <= B.Jend; j++)
      for (int i = B.Istr; i <= B.Iend; i++) h[at(i, j) + n * k] = code(i, j, k);
  }
  int r = h2d(A, h.data(), n * planes * sizeof(double), c->stream);
  if (r) return r;
  HaloSpec sp = {A, planes, BC_NONE, 'r'};
  if (wide) launch_halo_wide(c, &sp, 1);
  else if (tail) launch_halo_tail(c, &sp, 1);
  else launch_halo_multi(c, &sp, 1);
  halo_fence(c, FG_ALL);
  r = ctx_check(c, "exchange probe");
  if (r) return r;
  r = d2h(g.data(), A, n * planes * sizeof(double), c->stream);
  if (r) return r;
  r = ctx_check(c, "exchange probe");
  if (r) return r;
  for (int k = 0; k < planes; k++)
    for (int d = 0; d < 8; d++) {
      if (m.nbr[d] < 0) continue;
      const int dx = (d == 0 || d == 4 || d == 6) ? -1 : ((d == 1 || d == 5 || d == 7) ? 1 : 0);
      const int dy = (d == 2 || d == 4 || d == 5) ? -1 : ((d == 3 || d == 6 || d == 7) ? 1 : 0);
      const int i0 = dx == 0 ? B.Istr : (dx < 0 ? B.Istr - gl : B.Iend + 1), i1 = dx == 0 ? B.Iend : (dx < 0 ? B.Istr - 1 : B.Iend + gh);
      const int j0 = dy == 0 ? B.Jstr : (dy < 0 ? B.Jstr - gl : B.Jend + 1), j1 = dy == 0 ? B.Jend : (dy < 0 ? B.Jstr - 1 : B.Jend + gh);
      for (int j = j0; j <= j1; j++)
        for (int i = i0; i <= i1; i++)
          if (g[at(i, j) + n * k] != code(i, j, k)) {
            char msg[320];
            snprintf(msg, sizeof(msg), "exchange probe: tile %d, repetition %d (%d planes, strips %d|%d lines%s), plane %d, ghost point (%d,%d) from neighbour %d (rank %d) holds %.17g, expected %.17g",
                     c->cfg.tile, rep, planes, gl, gh, tail ? ", exchange stream" : "", k, i, j, d, m.nbr[d], g[at(i, j) + n * k], code(i, j, k));
            set_error(msg);
            return 2;
          }
    }
  return 0;
}
extern "C" int roms_hip_exchange_probe(roms_hip_ctx *c, int reps) {
  if (!c) return 8;
  if (!c->has_exchange) return 0;
  const int np3 = KMAX(c->x_min_planes, KMIN(c->G.N, 12));      // enough planes for the exchange stream's channel
  for (int rep = 1; rep <= reps; rep++) {
    int r;
    if ((r = probe_once(c, rep, 1, false, false))) return r;                   // one 2-D plane, compute stream
    if ((r = probe_once(c, rep, 4, false, false))) return r;                   // a barotropic exchange point: four planes
    if ((r = probe_once(c, rep, np3, false, true))) return r;                  // a 3-D exchange point (exchange stream where it is on)
    if (c->pair_mt && (r = probe_once(c, rep, 3, true, false))) return r;      // the wide strips behind a pair
    if (c->pair_mt && (r = probe_once(c, rep, 8, true, false))) return r;
  }
  return 0;
}

// Soak test of the installed transport (round 4): `reps` exchange points issued back to back WITHOUT a host synchronisation
// in between, as a run issues them -- planes coded with (i, j, plane, repetition) by a kernel, exchanged (narrow strips,
// every third repetition as a tail exchange on the exchange stream, every fourth with the wide strips where the pair
// engine is on), verified by a kernel that counts wrong ghost points on the device.  What the index-coded probe above
// cannot see -- a slot read before this repetition's strips have landed, a reordered store, a stale cache line under a
// stream of exchanges -- shows as the previous repetition's code.  0, or exit_flag 2 with the first wrong point.
extern "C" int roms_hip_exchange_soak(roms_hip_ctx *c, int reps) {
  if (!c) return 8;
  if (!c->has_exchange) return 0;
  const DGrid &G = c->G;
  const int planes = KMAX(c->x_min_planes, KMIN(G.N, 12));
  struct Buf { void *p = nullptr; ~Buf() { if (p) { dfree(p); } } } buf, cnt;
  if (dmalloc(&buf.p, (size_t)G.nij * (size_t)planes * sizeof(double)) || dmalloc(&cnt.p, 4 * sizeof(unsigned long long))) return 2;
  SoakArgs a;
  a.G = G; a.A = (double *)buf.p; a.bad = (unsigned long long *)cnt.p; a.planes = planes;
  for (int d = 0; d < 8; d++) a.nbr[d] = c->comm.nbr[d];
  for (int rep = 1; rep <= reps; rep++) {
    const bool wide = c->pair_mt && rep % 4 == 0, tail = !wide && rep % 3 == 0;
    const int np = wide ? KMIN(planes, 8) : planes;
    a.rep = rep;
    a.gl = wide ? B2D_GL : 3; a.gh = wide ? B2D_GH : G.Nghost;
    halo_fence(c, FG_ALL);                     // (the planes are reused: the previous tail exchange must have unpacked)
    LAUNCH_THREAD(k_soak_fill, G.ni, G.nj, np, c->stream, a);
    HaloSpec sp = {a.A, np, BC_NONE, 'r'};
    static const char *efault = getenv("ROMS_HIP_SOAK_FAULT");          // (test of the test: one exchange is left out)
    if (efault && efault[0] == '1' && rep == (reps + 1) / 2) { /* nothing arrives in this repetition */ }
    else if (wide) launch_halo_wide(c, &sp, 1);
    else if (tail) launch_halo_tail(c, &sp, 1);
    else launch_halo_multi(c, &sp, 1);
    halo_fence(c, FG_ALL);                     // a stream dependency, not a host synchronisation
    LAUNCH_THREAD(k_soak_check, G.ni, G.nj, np, c->stream, a);
    if (c->comm_failed) break;
  }
  int r = ctx_check(c, "exchange soak");
  if (r) return r;
  unsigned long long bad[4] = {0, 0, 0, 0};
  r = d2h(bad, cnt.p, sizeof(bad), c->stream);
  if (r) return r;
  if (bad[0]) {
    char msg[256];
    snprintf(msg, sizeof(msg), "exchange soak: tile %d: %llu wrong ghost points in %d back-to-back exchanges; first: repetition %llu, point (%d,%d), plane %llu",
             c->cfg.tile, bad[0], reps, bad[1], (int)(bad[2] >> 32) - 4096, (int)(bad[2] & 0xFFFFFFFFu) - 4096, bad[3]);
    set_error(msg);
    return 2;
  }
  return 0;
}

// Self-check of the rim planes the barotropic launches hand their rim through in a multi-tile run (g_step2d.cpp:run_rim_probe);
// roms_hip_rim_disable: the caller's verdict when the check failed on ANY rank -- every rank then keeps the exchanges.
extern "C" int roms_hip_rim_probe(roms_hip_ctx *c, int reps) {
  if (!c || reps < 1) return 8;
  int r = run_rim_probe(c, reps);
  return r ? r : ctx_check(c, "rim probe");
}
extern "C" int roms_hip_rim_disable(roms_hip_ctx *c) {
  if (!c) return 8;
  rim_disable(c);
  return 0;
}

// Back to no transport (a failed probe: the caller installs another one).  The mailbox slab stays allocated.
extern "C" int roms_hip_comm_reset(roms_hip_ctx *c) {
  if (!c) return 8;
  (void)dsync(c->stream);
#ifndef ROMS_CPU_EMU
  if (c->xstream) (void)hipStreamSynchronize(c->xstream);
  if (c->comm.peer_err) *c->comm.peer_err = 0;
#endif
  c->comm.peer_on = false;
  c->comm.fn = nullptr;
  c->comm_failed = false;
  c->loop_state = 0;
  return 0;
}

void launch_halo_wide(roms_hip_ctx *c, const HaloSpec *sp, int n) {
  c->x_wide = true;                       // the exchange being launched carries B2D_GL | B2D_GH lines
  launch_halo_multi(c, sp, n);
  c->x_wide = false;
}
void launch_halo_tail(roms_hip_ctx *c, const HaloSpec *sp, int n) {
  // a tail exchange may follow a producer that ran its rim first (c->rim_split): every line it packs must lie inside that
  // rim.  The wide strips of the barotropic pair kernel (B2D_GL lines) never do -- they carry 2-D state, no split producer
  if (c->rim_split && KMAX(c->G.xgl, c->G.xgh) + 2 > c->G.rimw) {
    set_error("launch_halo_tail: the strips of this exchange are wider than the rim a split producer ran first (DGrid::rimw)");
    c->comm_failed = true;                 // (reported as exit_flag 2 by the entry's ctx_check)
    return;
  }
  c->x_tail = true;
  launch_halo_multi(c, sp, n);
  c->x_tail = false;
}

// ---------------------------------------------------------------------- per-kernel C entries
// `groups`: every field group the routine reads or writes (over-approximated from the reference routine's
// argument list); an exchange in flight that carries one of them is waited for first (halo_fence), all
// others keep running beside the routine's kernels.
#define ENTRY(name, region, groups)                                     \
  extern "C" int roms_hip_##name(roms_hip_ctx *c) {                     \
    if (!c) return 8;                                                   \
    RegionTimer rt(c, region);                                          \
    halo_fence(c, (groups));                                            \
    int r = run_##name(c);                                              \
    return r ? r : ctx_check(c, #name);                                 \
  }
ENTRY(set_depth, 12, FG_HZ | FG_AVG)                                        // set_depth.F:76: Zt_avg1 -> Hz, z_r, z_w
ENTRY(set_massflux, 12, FG_MF | FG_UV | FG_HZ)                              // set_massflux.F:25
ENTRY(rho_eos, 14, FG_RHO | FG_T | FG_HZ)                                   // rho_eos.F:70
ENTRY(set_vbc, 6, FG_FLUX | FG_UV | FG_HZ | FG_2D | FG_T)                          // set_vbc.F:27
ENTRY(ana_vmix, 18, FG_AK | FG_HZ)
ENTRY(set_data, 4, FG_FLUX)
ENTRY(omega, 13, FG_W | FG_MF | FG_HZ)                                      // omega.F:96
ENTRY(set_zeta, 12, FG_2D | FG_AVG)
ENTRY(ini_zeta, 2, FG_ALL)
ENTRY(ini_fields, 2, FG_ALL)
ENTRY(pre_step3d, 22, FG_T | FG_T3 | FG_UV | FG_MF | FG_W | FG_AK | FG_HZ | FG_FLUX | FG_R)       // pre_step3d.F:126
ENTRY(prsgrd, 23, FG_R | FG_RHO | FG_HZ)                                    // prsgrd32.h
ENTRY(t3dmix2, 24, FG_T | FG_HZ | FG_RHO)
ENTRY(uv3dmix2, 30, FG_UV | FG_HZ | FG_R)
ENTRY(rhs3d_tile, 21, FG_R | FG_UV | FG_MF | FG_W | FG_HZ | FG_FLUX)        // rhs3d.F:196
ENTRY(step2d, 9, FG_2D | FG_AVG | FG_R | FG_RHO)                            // step2d_LF_AM3.h:163
ENTRY(step2d_pair, 9, FG_2D | FG_AVG | FG_R | FG_RHO)                       // step2d_LF_AM3.h:163, predictor + corrector of one fast step
ENTRY(step2d_loop, 9, FG_2D | FG_AVG | FG_R | FG_RHO)                       // ... of the fast steps 2 .. nfast (main3d.F:810-918), one persistent launch
ENTRY(step3d_uv, 34, FG_UV | FG_MF | FG_2D | FG_AVG | FG_AK | FG_HZ | FG_R | FG_FLUX)     // step3d_uv.F:134
ENTRY(step3d_t, 35, FG_T | FG_T3 | FG_MF | FG_W | FG_AK | FG_HZ | FG_FLUX)          // step3d_t.F:120
ENTRY(lmd_vmix, 18, FG_AK | FG_RHO | FG_UV | FG_HZ | FG_FLUX | FG_T)        // lmd_vmix.F:45
ENTRY(bulk_flux, 17, FG_FLUX | FG_T | FG_UV | FG_RHO | FG_HZ)               // bulk_flux.F:100
ENTRY(gls_prestep, 18, FG_AK | FG_MF | FG_W | FG_HZ)                         // gls_prestep.F:42
ENTRY(gls_corstep, 18, FG_AK | FG_MF | FG_W | FG_HZ | FG_UV | FG_RHO | FG_FLUX)   // gls_corstep.F:52

extern "C" int roms_hip_wvelocity(roms_hip_ctx *c, int ninp) {
  RegionTimer rt(c, 12);
  halo_fence(c, FG_WVEL | FG_W | FG_UV | FG_HZ | FG_AVG | FG_2D);           // wvelocity.F:30
  int r = run_wvelocity(c, ninp);
  return r ? r : ctx_check(c, "wvelocity");
}
extern "C" int roms_hip_rhs3d(roms_hip_ctx *c) {
  int r;
  if ((r = roms_hip_pre_step3d(c))) return r;
  if ((r = roms_hip_prsgrd(c))) return r;
  if ((r = roms_hip_t3dmix2(c))) return r;
  if ((r = roms_hip_rhs3d_tile(c))) return r;
  return roms_hip_uv3dmix2(c);
}
extern "C" int roms_hip_copy_probe(roms_hip_ctx *c, int reps, long *bytes_per_launch) {
  if (!c) return 8;
  if (bytes_per_launch) *bytes_per_launch = 2L * 8L * (long)c->G.nij * (long)(c->G.N + 1);   // read + write
  int r = run_copy_probe(c, reps);
  return r ? r : ctx_check(c, "copy_probe");
}
extern "C" int roms_hip_diag(roms_hip_ctx *c, double *out) {
  RegionTimer rt(c, 7);
  halo_fence(c, FG_UV | FG_RHO | FG_HZ | FG_WVEL | FG_2D);
  return run_diag(c, out);
}

// initial.F tail
extern "C" int roms_hip_start(roms_hip_ctx *c) {
  roms_hip_stepping &s = c->s;
  s.iif = 1; s.indx1 = 1; s.kstp = 1; s.krhs = 1; s.knew = 1; s.predictor = 0;
  s.nstp = 1; s.nrhs = 1; s.nnew = 1;
  s.time = c->cfg.dstart * 86400.0;
  ctx_sync_stepping(c);
  int r;
  if ((r = roms_hip_set_massflux(c))) return r;
  if ((r = roms_hip_omega(c))) return r;
  if ((r = roms_hip_rho_eos(c))) return r;
  s.iic = c->cfg.ntstart;
  ctx_sync_stepping(c);
  return 0;
}

// the barotropic loop (main3d.F:810-918), then set_depth, step3d_uv, omega, step3d_t (:963-1045) and the
// step counters (:1145-1148); join_late >= 0: the main stream waits for lane event join_late in between
static int baro_and_corrector(roms_hip_ctx *c, int join_late = -1, bool pre_behind = false, int ev_mix = -1) {
  roms_hip_stepping &s = c->s;
  const roms_hip_config &cf = c->cfg;
  int r;
#define DO(call) do { if ((r = (call))) return r; } while (0)
  for (int my_iif = 1; my_iif <= cf.nfast + 1; my_iif++) {  // :810-918
    const int next_indx1 = 3 - s.indx1;
    if (!s.predictor && my_iif <= cf.nfast + 1) {
      s.predictor = 1;
      s.iif = my_iif;
      s.kstp = (s.iif == 1) ? s.indx1 : 3 - s.indx1;
      s.knew = 3;
      s.krhs = s.indx1;
    }
    ctx_sync_stepping(c);
    // fast steps 2 .. nfast: predictor and corrector in one launch (k_step2d_pair.h); the first step (forward Euler start,
    // conversion of the 3-D forcing) and the auxiliary last call keep the per-call kernel
    static const int pair_from = getenv("ROMS_HIP_PAIR_FROM") ? atoi(getenv("ROMS_HIP_PAIR_FROM")) : 2;     // (debugging aids)
    static const int pair_to = getenv("ROMS_HIP_PAIR_TO") ? atoi(getenv("ROMS_HIP_PAIR_TO")) : 1 << 30;
    const bool pair = c->pair_on && s.predictor && s.iif >= 2 && s.iif <= cf.nfast && s.iif >= pair_from && s.iif <= pair_to;
    // ... or all of them in ONE persistent launch (k_step2d_loop.h) -- with the first fast step and the auxiliary call
    // iif = nfast+1 inside it (ROMS_HIP_LOOP_WHOLE=0: without; the per-call kernel then runs them) -- afterwards the indices
    // stand where the calls it replaces leave them
    static const char *ewh = getenv("ROMS_HIP_LOOP_WHOLE");
    const bool loop_ok = c->pair_on && s.predictor && pair_from == 2 && pair_to == (1 << 30) && step2d_loop_usable(c);
    if (loop_ok && s.iif == 1 && !(ewh && ewh[0] == '0')) {
      DO(roms_hip_step2d_loop(c));
      const int indx1 = (cf.nfast & 1) ? 3 - s.indx1 : s.indx1;      // one flip per fast step 1 .. nfast
      s.predictor = 0;
      s.iif = cf.nfast + 1;
      s.indx1 = indx1;
      s.knew = 3 - indx1;                                  // (the auxiliary call leaves next_indx1, and indx1 as it is)
      s.kstp = indx1;
      s.krhs = 3;
      ctx_sync_stepping(c);
      break;
    }
    if (loop_ok && pair && s.iif == 2) {
      DO(roms_hip_step2d_loop(c));
      const int flips = cf.nfast - 1;                      // one per pair
      const int indx1 = (flips & 1) ? 3 - s.indx1 : s.indx1;
      s.predictor = 0;
      s.iif = cf.nfast;
      s.indx1 = indx1;
      s.knew = indx1;
      s.kstp = 3 - s.knew;
      s.krhs = 3;
      ctx_sync_stepping(c);
      my_iif = cf.nfast;
      continue;
    }
    if (pair) DO(roms_hip_step2d_pair(c));
    else DO(roms_hip_step2d(c));
    if (s.predictor) {
      s.predictor = 0;
      s.knew = next_indx1;
      s.kstp = 3 - s.knew;
      s.krhs = 3;
      if (s.iif < cf.nfast + 1) s.indx1 = next_indx1;
    }
    ctx_sync_stepping(c);
    if (s.iif < cf.nfast + 1 && !pair) DO(roms_hip_step2d(c));
  }
  if (join_late >= 0) lane_wait(c, join_late);
  if (pre_behind) {
    // main3d_around_loop, form 2: the momentum predictor of pre_step3d and t3dmix2 (which adds to the t(nnew) it starts)
    // BEHIND the loop -- they touch no barotropic array; both read the old Hz, z_r: in front of set_depth
    (void)ev_mix;
    DO(roms_hip_pre_step3d(c));                             // k_pre_new (+ k_uv3dmix2_apply)
    DO(roms_hip_t3dmix2(c));
  }
  DO(roms_hip_set_depth(c));                                // :963
  DO(roms_hip_step3d_uv(c));                                // :990
  DO(roms_hip_omega(c));                                    // :1017
  if (cf.options & (ROMS_GLS_MIXING | ROMS_MY25_MIXING)) DO(roms_hip_gls_corstep(c));   // :1019-1021
  DO(roms_hip_step3d_t(c));                                 // :1045
  s.iic = s.iic + 1;                                        // :1145-1148
  s.time = s.time + cf.dt;
  ctx_sync_stepping(c);
#undef DO
  return 0;
}

#ifndef ROMS_CPU_EMU
thread_local size_t g_thread_ballast = 0;
#endif

// diag (main3d.F:355) as a device-side reduction into c->d_diag; the caller has placed it on a stream
static int enqueue_diag(roms_hip_ctx *c, int part = 0) {       // (part: run_diag_async)
  if (part != 2) {
    halo_fence(c, FG_UV | FG_RHO | FG_HZ | FG_WVEL | FG_2D);
    c->diag_ran = true;
    c->diag_step = c->s.iic - 1;
  }
  return run_diag_async(c, c->d_diag, part);
}

// The rest of a step from rho_eos on, single tile, small and medium grids (below the size from which uv3dmix2 is
// the column-marching kernel): the LATE-PREDICTOR schedule.  On such grids the step is a chain of kernels that
// each leave most of the chip idle, 59 of them the barotropic sub-steps.  What the barotropic loop needs from the
// 3-D part is only rufrc/rvfrc (prsgrd -> rhs3d_tile, the uv3dmix2 terms, the surface/bottom stresses), rhoA/rhoS
// (rho_eos) and zeta (set_zeta); the vertical mixing coefficients, pre_step3d (which needs them) and t3dmix2 feed
// step3d_uv/step3d_t, which follow the loop.  So three streams:
//   main   rho_eos, bulk_flux, set_vbc, uv3dmix2 (terms only), the rufrc sums, THE LOOP, set_depth ... step3d_t
//   S      set_massflux, omega, set_zeta, prsgrd, rhs3d_tile (point part)
//   X      diag, wvelocity (it reads DU_avg1, which the loop resets: the loop waits for it); then, BESIDE the loop:
//          ana_vmix | lmd_vmix, pre_step3d, t3dmix2
// pre_step3d therefore runs after prsgrd/rhs3d_tile/uv3dmix2 instead of before them (main3d.F:632 -> rhs3d.F
// calls them in that order).  Two read-before-overwrite dependences of the reference order are kept explicitly:
// pre_step3d.F:1008,1110 reads ru,rv(nrhs) of two steps ago, which prsgrd overwrites (k_prs_grad keeps the
// bracket they enter, wrk3[11..12]); uv3dmix2_s.h:226-262 adds to the u,v(nnew) pre_step3d sets
// (k_uv3dmix2_s stores its terms only, k_uv3dmix2_apply adds them behind k_pre_new).  Same operations on the same
// operands: bit-identical to the reference order (tests/test_gpu_parity.py, ROMS_HIP_LATE_PRE=0 is that order).
// Without side streams the enqueue order below is itself a valid serial order.
static int main3d_late(roms_hip_ctx *c, bool do_diag) {
  roms_hip_stepping &s = c->s;
  const roms_hip_config &cf = c->cfg;
  int r;
  enum { E_FORK = 0, E_EOS, E_VBC, E_W, E_D, E_X, E_UV, E_L, E_Z };
  const bool avg = c->avg_nAVG > 0 && c->avg_done_iic != s.iic;      // set_avg :562 (not when the output point ran it)
  kstream_t M = c->stream, S = c->stream2, X = c->stream3;
  const bool on = lanes_on(c);
  struct Back {
    roms_hip_ctx *c; kstream_t m;
    ~Back() { c->stream = m; c->late_pre = false; c->swdk_ready = false; c->pre_t3_ready = false; c->fold_uvmix = false; }
  } back{c, M};
  auto to = [&](kstream_t q) { if (on) c->stream = q; };
#define DO(call) do { if ((r = (call))) return r; } while (0)
  c->late_pre = true;
  { const char *efold = getenv("ROMS_HIP_FOLD"); c->fold_uvmix = !(efold && efold[0] == '0'); }   // (k_pre_new adds the uv3dmix2 terms)
  lane_record(c, E_FORK);
  DO(roms_hip_rho_eos(c));                                  // :350
  lane_record(c, E_EOS);
  if (cf.options & ROMS_BULK_FLUXES) DO(roms_hip_bulk_flux(c));   // :439
  DO(roms_hip_set_vbc(c));                                  // :445
  lane_record(c, E_VBC);
  to(S);
  lane_wait(c, E_FORK);
  DO(roms_hip_set_massflux(c));                             // :348
  DO(roms_hip_omega(c));                                    // :534
  lane_record(c, E_W);
  DO(roms_hip_set_zeta(c));                                 // :556
  lane_record(c, E_Z);
  lane_wait(c, E_EOS);
  DO(roms_hip_prsgrd(c));                                   // rhs3d.F: prsgrd, rhs3d_tile
  DO(run_rhs3d_pt(c));
  lane_record(c, E_D);
  to(X);
  lane_wait(c, E_EOS);
  if (do_diag) DO(enqueue_diag(c));                         // :355
  lane_wait(c, E_W);
  DO(roms_hip_wvelocity(c, s.nstp));                        // :535
  if (avg) {                                                // set_avg :562: what the loop overwrites, before it
    lane_wait(c, E_Z);
    DO(run_set_avg(c, c->G.wet_dry ? 0 : 1));                // (WET_DRY: the whole routine here -- the barotropic steps change the wet masks it multiplies with)
  }
  lane_record(c, E_X);
  to(M);
  DO(run_uv3dmix2_s(c));                                    // the terms of uv3dmix2 (c->late_pre)
  lane_record(c, E_UV);
  lane_wait(c, E_D);
  DO(run_rufrc_sums(c));
  lane_wait(c, E_X);
  to(X);                                                    // beside the barotropic loop
#ifndef ROMS_CPU_EMU
  static const char *eb = getenv("ROMS_HIP_LATE_BALLAST");
  struct Ballast { ~Ballast() { g_thread_ballast = 0; } } ballast;
  if (on) g_thread_ballast = eb ? (size_t)atol(eb) : 0;
#endif
  lane_wait(c, E_VBC);
  lane_wait(c, E_D);                                        // (the two-kernel KPP form reuses prsgrd's work array)
  if (avg && !c->G.wet_dry) DO(run_set_avg(c, 2));          // the rest of set_avg: rho, u, v, t, W, wvel, Huon, Hvom stay put
  if (cf.options & ROMS_ANA_VMIX) DO(roms_hip_ana_vmix(c));        // :525
  else if (cf.options & ROMS_LMD_MIXING) DO(roms_hip_lmd_vmix(c)); // :527
  if (cf.options & ROMS_SOLAR_SOURCE) { DO(run_swdk(c)); c->swdk_ready = true; }
  DO(run_pre_t3(c));
  c->pre_t3_ready = true;
  lane_wait(c, E_UV);
  DO(roms_hip_pre_step3d(c));                               // k_pre_new (+ the uv3dmix2 terms)
  DO(roms_hip_t3dmix2(c));
  lane_record(c, E_L);
  to(M);
#ifndef ROMS_CPU_EMU
  g_thread_ballast = 0;
#endif
#undef DO
  return baro_and_corrector(c, E_L);
}


// The same for a context whose fast steps 2 .. nfast are ONE persistent launch (k_step2d_loop.h).  That kernel fills every
// compute unit's registers and LDS for ~0.3 ms: a kernel placed beside it gets a fifth of its usual speed and costs the
// loop 10-15 % (measured: k_lmd_skpp 82 -> 441 us, k_pre_new 37 -> 252 us, the loop 316 -> 363 us), so NOTHING runs beside
// the loop here.  What the loop does not need is split around it instead:
//   in front, on their own streams beside the chain that feeds the loop (rho_eos -> prsgrd -> rhs3d_tile -> sums):
//       t3dmix2 and uv3dmix2_s AS TERMS in the first 100 us of the step (stream Y; they read only the state the step starts
//       from), then the vertical-mixing closure; diag, swdk, the tracer predictor of pre_step3d, wvelocity (stream X)
//   k_pre_new (pre_step3d's t, u, v(nnew); it adds the uv3dmix2 and t3dmix2 terms to the values it sets -- ROMS_HIP_FOLD=0:
//       k_uv3dmix2_apply and t3dmix2 as launches of their own behind it): form 1 (default) in front too, on stream Y behind the
//       closure; form 2 behind the loop on the main stream, in front of set_depth (it reads the old Hz, z_r)
// Tried and dropped (round 5): the chain rho_eos -> bulk_flux -> set_vbc -> closure -> k_pre_new -> loop on ONE stream, to save
// the 10-20 us a kernel behind another stream's event starts late -- no gain (0.862 either way): beside the tracer predictor
// and rhs3d_tile the closure's single-wave blocks wait for LDS whichever stream they come from (k_lmd_col 90 us alone, 110-170
// in the step); what is in front of the loop is ~530 us of kernel time on a chip it saturates.
// Same kernels on the same operands as main3d_late: bit-identical to the reference order.
static int main3d_around_loop(roms_hip_ctx *c, bool do_diag, int form, bool with_set_data) {
  roms_hip_stepping &s = c->s;
  const roms_hip_config &cf = c->cfg;
  int r;
  enum { E_FORK = 0, E_EOS, E_VBC, E_W, E_D, E_X, E_UV, E_Z, E_AK, E_T3, E_SD, E_MIX };
  const bool avg = c->avg_nAVG > 0 && c->avg_done_iic != s.iic;
  kstream_t M = c->stream, S = c->stream2, X = c->stream3, Y = c->stream4;
  const bool on = lanes_on(c);
  struct Back {
    roms_hip_ctx *c; kstream_t m;
    ~Back() { c->stream = m; c->late_pre = false; c->kpp_col_ok = false; c->swdk_ready = false; c->pre_t3_ready = false;
              c->tmix_terms = false; c->tmix_ready = false; c->fold_uvmix = false; }
  } back{c, M};
  auto to = [&](kstream_t q) { if (on) c->stream = q; };
#define DO(call) do { if ((r = (call))) return r; } while (0)
  c->late_pre = true;
  // KPP as one kernel with its columns in LDS leaves prsgrd's and the spline fluxes' work arrays alone; the two-kernel
  // form (columns taller than 41 levels) shares wrk3[1..4] with them and keeps its place behind both
  const bool kpp = (cf.options & ROMS_LMD_MIXING) != 0;
  // (round 5: the block form k_lmd_blk keeps its columns in LDS up to 71 levels)
  const char *eblk = getenv("ROMS_HIP_LMDBLK");
  const bool blk = !(eblk && eblk[0] == '0') && c->G.region == 0 && (size_t)(4 * (c->G.N + 1) + 24) * 64 * sizeof(double) <= 160 * 1024;
  // (LMD_BKPP: k_lmd_bkpp works on the spline columns the two-kernel form leaves in wrk3[1..4])
  const bool kpp_col = kpp && !c->G.bkpp && (blk || (size_t)3 * (size_t)(c->G.N + 1) * 64 * sizeof(double) < 64 * 1024) && !getenv("ROMS_HIP_LMDCOL");
  c->kpp_col_ok = kpp_col;
  // (ROMS_HIP_FOLD=0: k_uv3dmix2_apply and t3dmix2 as launches of their own behind k_pre_new -- the form the bit-identity
  // of the folded one is tested against)
  const char *efold = getenv("ROMS_HIP_FOLD");
  const bool fold = !(efold && efold[0] == '0');
  c->fold_uvmix = fold;
  c->tmix_terms = fold && (cf.options & ROMS_TS_DIF2) && !(cf.options & ROMS_MIX_ISO_TS) && !c->G.ts_dif4 && (double *)c->F.tmix;
  lane_record(c, E_FORK);
  if (with_set_data) {                                      // set_data (:258) beside rho_eos: bulk_flux / set_vbc are its first readers
    to(S);
    lane_wait(c, E_FORK);
    DO(roms_hip_set_data(c));
    lane_record(c, E_SD);
    to(M);
  }
  DO(roms_hip_rho_eos(c));                                  // :350
  lane_record(c, E_EOS);
  // Round 6: the chain that decides when the loop can start is set_data -> bulk_flux -> set_vbc -> vertical mixing -> k_pre_new
  // (one-step traces: 265 of the 304 us in front of the loop); bulk_flux and set_vbc read nothing rho_eos writes, so the chain
  // starts on stream Y beside rho_eos instead of behind it on this stream, and what ran on Y in the first 100 us -- t3dmix2 and
  // uv3dmix2_s as terms, which read only the state the step starts from -- runs here behind rho_eos
  // -- measured, BENCHMARK1, interleaved on one box (tools/gpu_debug/ab_env.sh ROMS_HIP_BULK_LANE "1 0"): 0.830 against 0.820 ms per
  // step with the order of round 5: the ~530 us of kernels in front of the loop saturate the chip whichever stream they come from
  // (DESIGN.md 4), a shorter dependency chain does not shorten them.  Kept as a switch (ROMS_HIP_BULK_LANE=1), not the default.
  static const char *ebl = getenv("ROMS_HIP_BULK_LANE");
  const bool bulk_y = on && ebl && ebl[0] == '1';
  auto surface = [&]() -> int {
    if (with_set_data) lane_wait(c, E_SD);
    if (cf.options & ROMS_BULK_FLUXES) DO(roms_hip_bulk_flux(c));   // :439
    DO(roms_hip_set_vbc(c));                                  // :445
    lane_record(c, E_VBC);
    return 0;
  };
  auto mixing_terms = [&]() -> int {
    if (c->tmix_terms) { DO(run_t3dmix2(c)); c->tmix_ready = true; }   // t3dmix2 as terms (k_pre_new adds them)
    DO(run_uv3dmix2_s(c));                                    // the terms of uv3dmix2 (c->late_pre)
    lane_record(c, E_UV);
    return 0;
  };
  if (bulk_y) {
    DO(mixing_terms());
    to(Y);
    lane_wait(c, E_FORK);
    DO(surface());
    to(M);
  } else DO(surface());
  // (the main stream idles from here to the sums of rufrc: the place to join the reductions of the previous step's diag)
  if (c->diag_join_pending) { lane_wait(c, E_MIX); c->diag_join_pending = false; }
  to(S);
  if (!with_set_data) lane_wait(c, E_FORK);
  DO(roms_hip_set_massflux(c));                             // :348
  DO(roms_hip_omega(c));                                    // :534
  lane_record(c, E_W);
  DO(roms_hip_set_zeta(c));                                 // :556
  lane_record(c, E_Z);
  // (multi-tile: the level the loop's first fast step starts from, 5 | 4 lines wide -- here, 300 us ahead of the launch)
  if (c->has_exchange && step2d_loop_usable(c)) DO(step2d_loop_pre(c, 2));
  lane_wait(c, E_EOS);
  DO(roms_hip_prsgrd(c));                                   // rhs3d.F: prsgrd, rhs3d_tile
  DO(run_rhs3d_pt(c));
  lane_record(c, E_D);
  if (!bulk_y) {
    to(Y);                                                  // what reads only the state the step starts from: in the first
    lane_wait(c, E_FORK);                                   // 100 us, while the chains above are short kernels waiting on each other
    DO(mixing_terms());
  }
  to(X);
  lane_wait(c, E_W);                                        // (neither reads a surface flux: k_pre_new does, behind the closure's wait)
  if (cf.options & ROMS_SOLAR_SOURCE) { DO(run_swdk(c)); c->swdk_ready = true; }
  DO(run_pre_t3(c));                                        // the tracer predictor: k_pre_new (form 1) waits for it
  c->pre_t3_ready = true;
  lane_record(c, E_T3);
  // diag (:355) and wvelocity (:535; diag reads the PREVIOUS step's wvel: in this order) feed nothing in the step: behind the
  // predictor on stream X (0.851 against 0.862 ms per BENCHMARK1 step with diag in front of it).  ROMS_HIP_DIAG_LANE=1: on the
  // main stream in the gap between set_vbc and the sums of rufrc (0.856).  A FIFTH stream for them: 0.937 -- the runtime
  // maps streams onto four hardware queues, the fifth shares one and serialises against it.
  const char *edl = getenv("ROMS_HIP_DIAG_LANE");
  const bool diag_main = edl && edl[0] == '1';
  if (diag_main) to(M); else lane_wait(c, E_EOS);
  // diag's two reductions read the column sums only: behind wvelocity and not awaited by the loop (two small launches beside
  // its first fast steps; the main stream joins them where it idles in the next step, or when the call ends: a wait in front
  // of set_depth costs 6 us there) -- ROMS_HIP_DIAG_SPLIT=0: all three in front
  static const char *eds = getenv("ROMS_HIP_DIAG_SPLIT");
  const bool diag_split = do_diag && !diag_main && on && !(eds && eds[0] == '0');
  if (do_diag) DO(enqueue_diag(c, diag_split ? 1 : 0));
  if (diag_main) lane_wait(c, E_W);
  DO(roms_hip_wvelocity(c, s.nstp));
  if (avg) {                                                // set_avg :562: what the loop overwrites, before it
    lane_wait(c, E_Z);
    DO(run_set_avg(c, c->G.wet_dry ? 0 : 1));                // (WET_DRY: the whole routine here -- the barotropic steps change the wet masks it multiplies with)
  }
  if (avg && !c->G.wet_dry) DO(run_set_avg(c, 2));          // the rest of set_avg: rho, u, v, t, W, wvel, Huon, Hvom stay put
  if (!diag_main) lane_record(c, E_X);
  if (diag_split) { DO(enqueue_diag(c, 2)); lane_record(c, E_MIX); c->diag_join_pending = true; }   // (joined in the next step, or when the call ends)
  to(Y);
  lane_wait(c, E_VBC);
  lane_wait(c, E_EOS);
  if (!kpp_col) { lane_wait(c, E_D); lane_wait(c, E_T3); }
  if (cf.options & ROMS_ANA_VMIX) DO(roms_hip_ana_vmix(c));        // :525
  else if (kpp) DO(roms_hip_lmd_vmix(c));                          // :527
  if (form == 1) {
    lane_wait(c, E_T3);
    lane_wait(c, E_UV);                                     // (the mixing terms k_pre_new adds: from the main stream since round 6)
    lane_wait(c, E_D);                                      // (the old ru/rv bracket k_prs_grad kept)
    DO(roms_hip_pre_step3d(c));                             // k_pre_new (+ the uv3dmix2 and t3dmix2 terms)
    DO(roms_hip_t3dmix2(c));                                // (a launch of its own only where it did not run ahead)
  }
  lane_record(c, E_AK);
  to(M);
  lane_wait(c, E_UV);
  lane_wait(c, E_D);
  DO(run_rufrc_sums(c));                                    // (the predictor on this stream, the loop behind it in-stream: 0.849 against 0.828 ms)
  if (c->has_exchange && step2d_loop_usable(c)) DO(step2d_loop_pre(c, 1));   // (multi-tile: the forcing on the enlarged sub-tiles, while the other lanes finish)
  if (!diag_main) lane_wait(c, E_X);
  lane_wait(c, E_T3);                                       // (nothing beside the loop)
  lane_wait(c, E_AK);
#undef DO
  return baro_and_corrector(c, -1, form == 2, E_MIX);
}

// may this context take the late-predictor schedules (main3d_late, main3d_around_loop)?
static bool late_schedule_ok(roms_hip_ctx *c) {
  const roms_hip_config &cf = c->cfg;
  static const char *elate = getenv("ROMS_HIP_LATE_PRE"), *euc = getenv("ROMS_HIP_UVCOL");
  const long cols = (long)(c->G.T.Iend - c->G.T.Istr + 1) * (c->G.T.Jend - c->G.T.Jstr + 1);
#ifdef ROMS_CPU_EMU
  const bool uvcol = false;
  (void)euc; (void)cols;
#else
  const bool uvcol = (cf.options & ROMS_UV_VIS2) && (euc ? euc[0] != '0' : cols >= 128L * 1024L);   // run_uv3dmix2_col's rule
#endif
  // (a masked run keeps the reference order -- its boundary fills are separate launches on the compute stream -- unless it
  // takes the persistent barotropic loop (round 6): the schedule around the loop then, 1.015 -> 0.985 ms on BENCHMARK1 with land)
  static const char *elm = getenv("ROMS_HIP_LATE_MASK");
  // (prsgrd31: the late schedule relies on k_prs_grad keeping the previous ru/rv bracket for the deferred predictor)
  // (a multi-tile context, round 6: the schedule around the persistent loop only, with the mailbox -- every lane has its own
  // channel -- and every exchange in the stream of its producer: the lanes are what overlaps an exchange with compute)
  // ... or, where the loop does not fit (tiles above 32 K columns: BASELINE's 8-GPU partition), with the pair launches in its
  // place: ROMS_HIP_MT_LANES=0 keeps those contexts in the reference order
  static const char *eml = getenv("ROMS_HIP_MT_LANES");
  const bool tiles_ok = !c->has_exchange || (c->comm.peer_on && !c->x_async && (step2d_loop_usable(c) || (c->pair_mt && c->pair_on && !(eml && eml[0] == '0'))));
  return tiles_ok && !uvcol && !c->G.dia_ts && !c->G.dia_uv && !c->G.uv_vis4 && !c->G.ts_dif4 && !c->G.mix_geo_uv && (!c->G.masking || (elm ? elm[0] == '1' : step2d_loop_usable(c))) &&
         !(elate && elate[0] == '0') && !(cf.options & (ROMS_PRSGRD31 | ROMS_PRSGRD40 | ROMS_PRSGRD42 | ROMS_PRSGRD44 | ROMS_GLS_MIXING | ROMS_MY25_MIXING));
}
// ... and the one around the persistent barotropic loop: 0 = no, else its form (ROMS_HIP_LOOP_SCHED: 0 = the loop inside the
// late-predictor schedule, 1 = pre_step3d / t3dmix2 in front of the loop: the default since their mixing terms are folded
// into k_pre_new (0.862 against 0.877 ms per BENCHMARK1 step), 2 = behind it)
static int around_loop_form(roms_hip_ctx *c) {
  static const char *esch = getenv("ROMS_HIP_LOOP_SCHED");
  const int form = esch ? atoi(esch) : 1;
  return (form > 0 && late_schedule_ok(c) && (step2d_loop_usable(c) || c->has_exchange)) ? form : 0;
}

// one pass of STEP_LOOP, main3d.F:216-1148
static int main3d_one(roms_hip_ctx *c) {
  roms_hip_stepping &s = c->s;
  const roms_hip_config &cf = c->cfg;
  int r;
#define DO(call) do { if ((r = (call))) return r; } while (0)
  s.nstp = 1 + (s.iic - cf.ntstart) % 2;                    // :220-231
  s.nnew = 3 - s.nstp;
  s.nrhs = s.nstp;
  ctx_sync_stepping(c);
  DO(poison_work(c));                                       // (ROMS_HIP_POISON=1 only)
  c->loop_pre_frc = c->loop_pre_state = false;
  c->ghost_ok = s.iic != cf.ntstart;                        // (the first step: behind post_initial, below)
  if (c->diag_join_pending && !(late_schedule_ok(c) && around_loop_form(c) > 0)) { lane_wait(c, 11); c->diag_join_pending = false; }
  // set_data (:258) feeds bulk_flux / set_vbc only: the schedule around the persistent loop places it on a side stream
  // beside rho_eos (main3d_around_loop); post_initial reads none of its fields
  const bool data_late = around_loop_form(c) > 0 && s.iic != cf.ntstart;
  if (!data_late) DO(roms_hip_set_data(c));                 // :258
  if (s.iic == cf.ntstart) {                                // post_initial :335
    DO(roms_hip_ini_zeta(c));
    DO(roms_hip_set_depth(c));
    DO(roms_hip_ini_fields(c));
  }
  c->ghost_ok = true;                                       // (from here on every field the point-wise producers read has been exchanged)
  // diag (:355): device-side reduction every ninfo steps; the blow-up test is made on the host
  // when roms_hip_main3d returns (no per-step host synchronisation)
  const bool do_diag = cf.ninfo > 0 && (s.iic - 1) % cf.ninfo == 0;
  {
    if (late_schedule_ok(c)) {
      const int form = around_loop_form(c);
      if (form > 0) return main3d_around_loop(c, do_diag, form, data_late);
      if (!c->has_exchange) return main3d_late(c, do_diag);
    }   // (GLS: its two routines keep the reference's places)
  }
  DO(roms_hip_rho_eos(c));                                  // :350
  // Two independent chains follow: set_massflux (:348) -> omega (:534) -> wvelocity (:535), and the
  // surface forcing / vertical mixing (:439-527), which needs neither Huon/Hvom nor W.  In a
  // single-tile run (no halo transport to order) the first chain, behind diag, goes to the side
  // stream; on a small grid neither chain fills the chip.
  const bool side_chain = !c->has_exchange;
  side_mark(c);
  if (side_chain) {          // main stream first (see side_mark)
    if (cf.options & ROMS_BULK_FLUXES) DO(roms_hip_bulk_flux(c));   // :439
    DO(roms_hip_set_vbc(c));                                  // :445
    if (cf.options & ROMS_ANA_VMIX) DO(roms_hip_ana_vmix(c));       // :525
    else if (cf.options & ROMS_LMD_MIXING) DO(roms_hip_lmd_vmix(c)); // :527
  }
  side_begin(c);
  r = 0;
  auto diag_now = [&]() { return do_diag ? enqueue_diag(c) : 0; };   // reads u, v, rho, wvel ... of this point of the step
  if (side_chain) {
    // what pre_step3d waits for comes first; diag and wvelocity -- nothing on the main stream reads their
    // results before the next step -- follow behind the partial join point (side_point)
    r = roms_hip_set_massflux(c);
    if (!r) r = roms_hip_omega(c);
    if (!r) r = roms_hip_set_zeta(c);              // :556
    if (!r && (cf.options & ROMS_SOLAR_SOURCE)) { r = run_swdk(c); c->swdk_ready = r == 0; }   // pre_step3d's first kernel
    // small grids: the tracer predictor of pre_step3d too (it needs W, not the mixing coefficients) -- the
    // bulk-flux / KPP chain on the main stream is a string of column kernels that leave most of the chip idle
    static const char *ept = getenv("ROMS_HIP_EARLY_T3");
    const bool early = ept ? ept[0] != '0' : (long)(c->G.T.Iend - c->G.T.Istr + 1) * (c->G.T.Jend - c->G.T.Jstr + 1) <= 64L * 1024L;
    if (!r && early) { r = run_pre_t3(c); c->pre_t3_ready = r == 0; }
    // set_diags :559 reads the terms of the PREVIOUS step in DiaTwrk, which pre_step3d (main stream, behind the partial
    // join) starts to overwrite: it runs ahead of the join point, not beside it like set_avg
    if (!r && c->G.dia_ts && c->dia_nDIA > 0 && c->dia_done_iic != s.iic) r = run_set_diags(c);
    side_point(c);
    if (!r) r = diag_now();
    if (!r) r = roms_hip_wvelocity(c, s.nstp);     // overwrites wvel, which diag reads: same stream, in order
    if (!r && c->avg_nAVG > 0 && c->avg_done_iic != s.iic) r = run_set_avg(c, 0);      // :562
  } else {
    r = diag_now();
  }
  side_end(c);
  if (r) return r;
  if (!side_chain) {
    DO(roms_hip_set_massflux(c));
    if (cf.options & ROMS_BULK_FLUXES) DO(roms_hip_bulk_flux(c));
    DO(roms_hip_set_vbc(c));
    if (cf.options & ROMS_ANA_VMIX) DO(roms_hip_ana_vmix(c));
    else if (cf.options & ROMS_LMD_MIXING) DO(roms_hip_lmd_vmix(c));
    DO(roms_hip_omega(c));
    side_join(c);
    DO(roms_hip_wvelocity(c, s.nstp));
    DO(roms_hip_set_zeta(c));                               // :556
    if (c->avg_nAVG > 0 && c->avg_done_iic != s.iic) DO(run_set_avg(c, 0));           // :562
    if (c->G.dia_ts && c->dia_nDIA > 0 && c->dia_done_iic != s.iic) DO(run_set_diags(c));   // :559
  } else {
    side_join_point(c);               // (diag and wvelocity are picked up by the next join, before the barotropic loop)
  }
  // rhs3d :632 -- t3dmix2 only touches t(nnew): it overlaps prsgrd and rhs3d_tile
  DO(roms_hip_pre_step3d(c));
  c->swdk_ready = false;
  c->pre_t3_ready = false;
  side_mark(c);
  DO(roms_hip_prsgrd(c));
  halo_fence(c, FG_R | FG_UV | FG_MF | FG_W | FG_HZ | FG_FLUX);
  DO(run_rhs3d_pt(c));
  halo_fence(c, FG_UV | FG_HZ | FG_R | FG_T | FG_RHO);
  // large grids: uv3dmix2 and the coupling sums as one column-marching kernel behind rhs3d_tile's point part
  // (it starts from the vertical sums of ru, rv); otherwise uv3dmix2's point-wise part on the side stream
  // beside prsgrd/rhs3d_tile and ONE column kernel for the sums of both.  t3dmix2: side stream either way.
  halo_fence(c, FG_FLUX);
  const int rcol = run_uv3dmix2_col(c);
  if (rcol > 0) return rcol;
  side_begin(c);
  r = roms_hip_t3dmix2(c);
  if (!r && rcol < 0 && !c->G.mix_geo_uv) r = run_uv3dmix2_s(c);
  side_end(c);
  if (r) return r;
  side_join(c);
  if (rcol < 0) {
    halo_fence(c, FG_R | FG_FLUX);
    DO(run_rufrc_sums(c));            // rufrc/rvfrc of rhs3d_tile and uv3dmix2 in one kernel
    if (c->G.mix_geo_uv) DO(run_uv3dmix2(c));     // uv3dmix2_geo.h: its kernels add to the sums and to u,v(nnew) level by level
  }
  if (cf.options & (ROMS_GLS_MIXING | ROMS_MY25_MIXING)) DO(roms_hip_gls_prestep(c));   // :634-636
#undef DO
  return baro_and_corrector(c);
}

extern "C" int roms_hip_last_diag(roms_hip_ctx *c, double *out) {
  if (!c || !out) return 8;
  for (int k = 0; k < 16; k++) out[k] = 0.0;
  out[14] = -1.0;
  if (c->diag_step < 0) return 0;
  int r = fetch_diag(c, c->d_diag, out);
  if (r) return r;
  out[14] = (double)c->diag_step;
  return 0;
}

// Time-averaged fields (AVERAGES): the window of set_avg.F and which fields it accumulates (bit f of mask = field f
// of the list in include/roms_hip.h).  nAVG = 0 switches averaging off and frees nothing; the arrays are allocated
// at the first call that switches it on.
extern "C" int roms_hip_avg_config(roms_hip_ctx *c, int nAVG, int ntsAVG, int nrrec, int ntstart, unsigned mask) {
  if (!c || nAVG < 0) return 8;
  // (WET_DRY is an option of the context: refused here as the reverse is in roms_hip_create)
  // (WET_DRY, round 6: every field times the full mask of its grid type, the sums divided by the wet-point counters -- k_avg.h)
  if (nAVG > 0 && c->G.wet_dry)
    for (int m = 0; m < 3; m++)
      if (!c->avg_cnt[m]) {
        void *p = nullptr;
        if (dmalloc(&p, (size_t)c->G.nij * sizeof(double))) return 2;
        c->allocs.push_back(p);
        c->avg_cnt[m] = (double *)p;
      }
  // (MASKING: the 22 fields built carry no mask arithmetic of their own -- set_avg.F masks the rotated and vorticity
  // fields only -- and accumulate the masked state; pinned with oracle/ref/upwelling_avg_mask.h)
  if (nAVG > 0)
    for (int f = 0; f < 22; f++)
      if (((mask >> f) & 1u) && !c->avg[f]) {
        void *p = nullptr;
        if (dmalloc(&p, (size_t)avg_field_elems(c, f) * sizeof(double))) return 2;
        c->allocs.push_back(p);
        c->avg[f] = (double *)p;
      }
  c->avg_nAVG = nAVG; c->avg_ntsAVG = ntsAVG; c->avg_nrrec = nrrec; c->avg_ntstart = ntstart; c->avg_mask = mask;
  c->avg_done_iic = -1;
  return 0;
}
// set_avg(ng,tile), main3d.F:562
extern "C" int roms_hip_set_avg(roms_hip_ctx *c) {
  if (!c) return 8;
  if (c->avg_nAVG <= 0 || c->avg_done_iic == c->s.iic) return 0;     // off, or roms_hip_output_point ran it for this step
  halo_fence(c, FG_ALL);
  c->avg_done_iic = c->s.iic;
  return run_set_avg(c, 0);
}
// Per-term tracer tendencies (DIAGNOSTICS_TS of the application header): allocates DIAGS(ng)%DiaTwrk / DiaTrc / avgzeta
// (mod_diags.F:104-228) and sets the window of set_diags.F (nDIA, ntsDIA; nrrec, ntstart of a restart).  From then on the
// kernels of pre_step3d, t3dmix2 and step3d_t store their terms, roms_hip_main3d calls set_diags where main3d.F:559 does,
// and "DiaTwrk", "DiaTrc", "dia_zeta" can be downloaded.  nDIA = 0 switches the accumulation off (the terms stay on).
// Refused (exit_flag 5): MPDATA tracers (their Dhadv / Dvadv path, step3d_t.F:881-895, :1254), the plain vertical diffusion.
extern "C" int roms_hip_dia_config(roms_hip_ctx *c, int nDIA, int ntsDIA, int nrrec, int ntstart) {
  if (!c || nDIA < 0 || ntsDIA < 1) return 8;
  DGrid &G = c->G;
  if (G.uv_vis4 || G.ts_dif4) { set_error("DIAGNOSTICS_TS: the diagnostic statements of the biharmonic operators are not built"); return 5; }
  if (G.wet_dry) { set_error("DIAGNOSTICS_TS with WET_DRY: the wet/dry masks of set_diags.F are not built"); return 5; }
  for (int it = 0; it < G.NT; it++)
    if (G.hadv[it] == ROMS_MPDATA || G.vadv[it] == ROMS_MPDATA) { set_error("DIAGNOSTICS_TS: tracers advected with MPDATA are not built (step3d_t.F:881-895)"); return 5; }
  if (G.options & ROMS_PLAIN_VDIFF) { set_error("DIAGNOSTICS_TS: built for SPLINES_VDIFF only"); return 5; }
  if (!G.dia_ts) {
    int ic = 4, ndt = 6;
    for (int k = 0; k < 10; k++) G.dia_idx[k] = 0;
    G.dia_idx[DIA_HADV] = 1; G.dia_idx[DIA_XADV] = 2; G.dia_idx[DIA_YADV] = 3; G.dia_idx[DIA_VADV] = 4;      // mod_scalars.F:4246-4262
    if (G.options & ROMS_TS_DIF2) {
      G.dia_idx[DIA_HDIF] = ic + 1; G.dia_idx[DIA_XDIF] = ic + 2; G.dia_idx[DIA_YDIF] = ic + 3; ic += 3; ndt += 3;
      if (G.options & (ROMS_MIX_GEO_TS | ROMS_MIX_ISO_TS)) { G.dia_idx[DIA_SDIF] = ic + 1; ic += 1; ndt += 1; }
    }
    G.dia_idx[DIA_VDIF] = ic + 1; G.dia_idx[DIA_RATE] = ic + 2;
    const size_t n = (size_t)G.nij * (size_t)G.N * (size_t)G.NT * (size_t)ndt;
    void *p = nullptr, *q = nullptr, *z = nullptr;
    if (dmalloc(&p, n * sizeof(double))) return 2;
    c->allocs.push_back(p);                           // (each as soon as it exists: roms_hip_destroy frees what a failed call leaves)
    if (dmalloc(&q, n * sizeof(double))) return 2;
    c->allocs.push_back(q);
    if (dmalloc(&z, (size_t)G.nij * sizeof(double))) return 2;
    c->allocs.push_back(z);
    c->F.DiaTwrk = (double *)p; c->F.DiaTrc = (double *)q; c->F.dia_zeta = (double *)z;
#ifndef ROMS_CPU_EMU
    if (hipfail(hipDeviceSynchronize(), "hipDeviceSynchronize")) return 2;
#endif
    G.dia_ts = ndt;
  }
  c->dia_nDIA = nDIA; c->dia_ntsDIA = ntsDIA; c->dia_nrrec = nrrec; c->dia_ntstart = ntstart; c->dia_done_iic = -1;
  if (c->cfg.options & ROMS_DIAGNOSTICS_UV) return diauv_config(c);     // the momentum terms beside them (ABI version 4: an option bit)
  return 0;
}
// set_diags(ng,tile), main3d.F:559
// Biharmonic horizontal mixing along s-surfaces (UV_VIS4 + MIX_S_UV, TS_DIF4 + MIX_S_TS of the application header): the harmonic
// operators applied twice -- uv3dmix4_s.h, the UV_VIS4 block of step2d_LF_AM3.h:1653-1920, t3dmix4_s.h.  Between
// roms_hip_create and roms_hip_start; the caller uploads "visc4_r", "visc4_p", "diff4" = the SQUARE ROOTS of VISC4 / TNU4
// (inp_par.F:634) and leaves the harmonic coefficient arrays at zero (the harmonic operators then add exact zeros).
// Refused (exit_flag 5): the geopotential / isopycnic tracer operators outside a periodic channel, open boundaries, the per-term
// diagnostics (their biharmonic statements are not built).
static int mix4_config(roms_hip_ctx *c, int uv_vis4, int ts_dif4) {
  if (!c) return 8;
  DGrid &G = c->G;
  if (!uv_vis4 && !ts_dif4) { G.uv_vis4 = G.ts_dif4 = 0; return 0; }
  // (round 6: TS_DIF4 + MIX_GEO_TS / MIX_ISO_TS between western / eastern walls -- the conditions of t3dmix4_geo.h:475-600, t3dmix4_iso.h:504-618
  // on the first operator there and at the corners -- is pinned in a closed basin and no longer refused)
  if (G.obc) { set_error("UV_VIS4 / TS_DIF4 with open boundaries: the gradient conditions of the first harmonic operator are not built on the device"); return 5; }
  if (G.dia_ts || G.dia_uv) { set_error("UV_VIS4 / TS_DIF4: the per-term diagnostics of the biharmonic operators are not built"); return 5; }
  if (G.wet_dry) { set_error("UV_VIS4 / TS_DIF4 with WET_DRY: harmonic mixing along s-surfaces only (the barotropic kernel of a WET_DRY run carries no biharmonic block)"); return 5; }
  if (G.Nghost < 3 && uv_vis4) { set_error("UV_VIS4 needs three ghost points (inp_par.F:214-223)"); return 5; }
  if (uv_vis4 && !c->F.lap4) {
    void *p = nullptr;
    if (dmalloc(&p, (size_t)2 * G.N * G.nij * sizeof(double))) return 2;
    c->allocs.push_back(p);
    c->F.lap4 = (double *)p;
  }
  G.uv_vis4 = uv_vis4 != 0; G.ts_dif4 = ts_dif4 != 0;
  c->m2d_dirty = true;                               // (the packed barotropic metrics carry visc4 in place of visc2)
  c->pair_on = step2d_pair_usable(c);
  c->loop_state = 0;
  return 0;
}
// Wetting and drying (WET_DRY, ROMS/Nonlinear/wetdry.F): switches on the time-dependent masks "rmask_wet", "umask_wet",
// "vmask_wet", "pmask_wet" (set per fast step by k_wetdry in front of the barotropic kernel, time-averaged for the 3-D step
// behind the last one), "rmask_full" ... (wet x land, for output) and every WET_DRY branch of the path: step2d_LF_AM3.h
// (:992, :1617, :2205-2222, :2518-2667), prsgrd32.h:362,426, rhs3d.F:1709-1910, t3dmix2_s.h:239,279, uv3dmix2_s.h:276,
// step3d_uv.F:720-721,1187-1188,1359,1579 and its boundary rows, set_vbc.F:307-308,397 and LIMIT_BSTRESS (globaldefs.h:160),
// ini_fields.F:294-403,850, zetabc.F:783-874, u2dbc_im.F:1190-1318, v2dbc_im.F:1239-1367, u3dbc_im.F:523,681.  Dcrit = DCRIT of
// roms.in.  roms_hip_wetdry_ini sets the initial masks (initial.F:467).  Call between roms_hip_create and roms_hip_start.
// Open boundaries: the WET_DRY forms of zetabc.F:190 (Chapman), u2dbc_im.F:339 (Shchepetkin), u3dbc_im.F:174 ... in k_obc.h.
// Round 5: the WET_DRY statements of mpdata_adiff.F (:463,686,927,1117,1132,1148), bulk_flux.F (:637-1312), pre_step3d.F:903
// (solar source) and t3dmix2_geo.h (:231,263); KPP has none of its own (pinned by oracle/ref/benchmark_wetdry.h).
// Refused (exit_flag 5) where the reference's WET_DRY statements are not built on the device: no MASKING (globaldefs.h:152
// switches it on), the closures GLS / MY2.5, isopycnic / biharmonic mixing, the pressure Jacobians other than prsgrd32,
// averages and diagnostics.
static int wetdry_config(roms_hip_ctx *c, double Dcrit) {
  if (!c) return 8;
  DGrid &G = c->G;
  const int opt = G.options;
  if (!G.masking) { set_error("WET_DRY: needs MASKING (globaldefs.h:152-154 defines it with WET_DRY)"); return 5; }
  // (round 6: GLS_MIXING / MY25_MIXING -- their routines carry no WET_DRY statement --, MIX_GEO_UV and the Jacobians prsgrd31 / 40 / 44
  // are pinned under WET_DRY: oracle/ref/upwelling_wetdry_*.h)
  if (G.uv_vis4 || G.ts_dif4) { set_error("WET_DRY: harmonic mixing only (t3dmix2_s.h, t3dmix2_geo.h, t3dmix2_iso.h, uv3dmix2_s.h, uv3dmix2_geo.h)"); return 5; }
  if (G.prs4x == 42) { set_error("WET_DRY with PJ_GRADPQ2: prsgrd42.h masks its FIRST pass (:355-361), which is not built"); return 5; }
  if (opt & ROMS_PRSGRD40) { set_error("WET_DRY with PJ_GRADP: the reference does not compile with the two together (prsgrd40.h:98 passes umask_wet, vmask_wet undeclared) -- nothing to pin against"); return 5; }
  if (opt & (ROMS_PLAIN_VVISC)) { set_error("WET_DRY: SPLINES_VVISC only"); return 5; }
  if (G.dia_ts || G.dia_uv) { set_error("WET_DRY: the wet/dry masks of set_diags.F are not built"); return 5; }
  if (!c->F.wd_eff) {
    void *p = nullptr;
    if (dmalloc(&p, (size_t)2 * G.nij * sizeof(double))) return 2;
    c->allocs.push_back(p);
    c->F.wd_eff = (double *)p;
  }
  G.wet_dry = 1; G.Dcrit = Dcrit;
  G.fuse_halo = 0; G.fuse3d = 0;                        // (the wetting/drying conditions follow the boundary fills: separate launches)
  c->pair_on = step2d_pair_usable(c);
  c->loop_state = 0;
  return 0;
}
extern "C" int roms_hip_wetdry_ini(roms_hip_ctx *c) {
  if (!c) return 8;
  if (!c->G.wet_dry) { set_error("roms_hip_wetdry_ini: create the context with ROMS_WET_DRY"); return 8; }
  ctx_sync_stepping(c);
  return run_wetdry(c, 2);
}
// Per-term momentum tendencies (DIAGNOSTICS_UV): allocates DIAGS(ng)%DiaU2wrk ... DiaV3d (mod_diags.F:174-222) and switches
// the term stores of prsgrd, rhs3d, uv3dmix2, pre_step3d, step2d and step3d_uv on; the window is roms_hip_dia_config's (call
// that first), set_diags accumulates "DiaU2d", "DiaV2d", "DiaU3d", "DiaV3d".  Refused (exit_flag 5): no SPLINES_VVISC.
static int diauv_config(roms_hip_ctx *c) {
  if (!c) return 8;
  if (!c->G.dia_ts) { set_error("DIAGNOSTICS_UV: needs the tracer terms (roms_hip_dia_config allocates both)"); return 8; }
  if (c->G.wet_dry) { set_error("DIAGNOSTICS_UV with WET_DRY: the wet/dry masks of set_diags.F are not built"); return 5; }
  halo_fence(c, FG_ALL);
  if (c->G.dia_uv) return 0;
  int r = duv_config(c);
  if (r) return r;
  void *p = nullptr;
  if (dmalloc(&p, duv_planes(c->G) * (size_t)c->G.nij * sizeof(double))) return 2;
  c->allocs.push_back(p);
  c->F.duv = (double *)p;
  c->G.dia_uv = 1;
  c->pair_on = step2d_pair_usable(c);
  c->loop_state = 0;                // (the per-call barotropic kernel carries the term stores)
  return 0;
}
extern "C" int roms_hip_set_diags(roms_hip_ctx *c) {
  if (!c) return 8;
  if (!c->G.dia_ts || c->dia_nDIA <= 0 || c->dia_done_iic == c->s.iic) return 0;
  c->dia_done_iic = c->s.iic;
  return run_set_diags(c);
}
extern "C" int roms_hip_dia_time(roms_hip_ctx *c, double *t) {
  if (!c || !t) return 8;
  *t = c->dia_time;
  return 0;
}
extern "C" int roms_hip_avg_time(roms_hip_ctx *c, double *t) {
  if (!c || !t) return 8;
  *t = c->avg_time;
  return 0;
}

// main3d.F:591: output (wrt_his, wrt_rst) sits in the MIDDLE of a step, behind set_zeta.  roms_hip_main3d
// returns between steps, so a caller that writes history or restart records first brings the derived fields to
// that point of the step about to be taken: set_data, set_massflux, rho_eos, the surface forcing, the vertical
// mixing coefficients, omega, wvelocity, set_zeta (:258-556) with that step's time indices.  Every one of them
// only recomputes derived fields from the prognostic state, so the step itself, which repeats them, is not
// changed by this call -- with one exception that is undone here: diag (:355) reads the wvel of the PREVIOUS
// step's wvelocity, so the new one is parked in the field "w_out" and wvel restored.
extern "C" int roms_hip_output_point(roms_hip_ctx *c) {
  if (!c) return 8;
  roms_hip_stepping &s = c->s;
  const roms_hip_config &cf = c->cfg;
  int r;
#define DO(call) do { if ((r = (call))) return r; } while (0)
  s.nstp = 1 + (s.iic - cf.ntstart) % 2;
  s.nnew = 3 - s.nstp;
  s.nrhs = s.nstp;
  ctx_sync_stepping(c);
  DO(roms_hip_set_data(c));
  DO(roms_hip_set_massflux(c));
  DO(roms_hip_rho_eos(c));
  if (cf.options & ROMS_BULK_FLUXES) DO(roms_hip_bulk_flux(c));
  DO(roms_hip_set_vbc(c));
  if (cf.options & ROMS_ANA_VMIX) DO(roms_hip_ana_vmix(c));
  else if (cf.options & ROMS_LMD_MIXING) DO(roms_hip_lmd_vmix(c));
  DO(roms_hip_omega(c));
  const size_t wbytes = (size_t)c->G.nij * (size_t)(c->G.N + 1) * sizeof(double);
  halo_fence(c, FG_ALL);
  DO(d2d((double *)c->F.wrk3[11], (double *)c->F.wvel, wbytes, c->stream));
  DO(roms_hip_wvelocity(c, s.nstp));
  halo_fence(c, FG_ALL);
  DO(d2d((double *)c->F.wrk3[12], (double *)c->F.wvel, wbytes, c->stream));
  DO(d2d((double *)c->F.wvel, (double *)c->F.wrk3[11], wbytes, c->stream));
  DO(roms_hip_set_zeta(c));
  if (c->avg_nAVG > 0 && c->avg_done_iic != s.iic) {     // set_avg :562 precedes output :591; the step skips it then
    DO(d2d((double *)c->F.wrk3[11], (double *)c->F.wvel, wbytes, c->stream));        // (it averages the NEW wvel)
    DO(d2d((double *)c->F.wvel, (double *)c->F.wrk3[12], wbytes, c->stream));
    DO(run_set_avg(c, 0));
    DO(d2d((double *)c->F.wvel, (double *)c->F.wrk3[11], wbytes, c->stream));
    c->avg_done_iic = s.iic;
  }
  if (c->G.dia_ts && c->dia_nDIA > 0 && c->dia_done_iic != s.iic) {   // set_diags :559 precedes output too
    DO(run_set_diags(c));
    c->dia_done_iic = s.iic;
  }
#undef DO
  return dsync(c->stream);
}

extern "C" int roms_hip_get_bounds(roms_hip_ctx *c, int *out) {
  if (!c || !out) return 8;
  const TB &b = c->G.T;
  const int v[54] = {c->cLBi, c->cLBi + c->cni - 1, c->cLBj, c->cLBj + c->cnj - 1,     // (the caller's bounds)
    b.Istr, b.Iend, b.Jstr, b.Jend, b.IstrR, b.IendR, b.JstrR, b.JendR, b.IstrU, b.JstrV, b.IstrB, b.IendB, b.IstrM,
    b.JstrB, b.JendB, b.JstrM, b.IstrP, b.IendP, b.JstrP, b.JendP, b.IstrT, b.IendT, b.JstrT, b.JendT,
    b.Istrm3, b.Istrm2, b.Istrm1, b.IstrUm2, b.IstrUm1, b.Iendp1, b.Iendp2, b.Iendp2i, b.Iendp3,
    b.Jstrm3, b.Jstrm2, b.Jstrm1, b.JstrVm2, b.JstrVm1, b.Jendp1, b.Jendp2, b.Jendp2i, b.Jendp3,
    b.west, b.east, b.south, b.north, b.sw, b.se, b.nw, b.ne};
  for (int k = 0; k < 54; k++) out[k] = v[k];
  return 0;
}

extern "C" int roms_hip_step_timing(roms_hip_ctx *c, int nmax) {
  if (!c || nmax < 0) return 8;
  for (ktimer_t e : c->step_ev) ktimer_destroy(e);
  c->step_ev.clear();
  c->step_ev_n = 0;
  for (int k = 0; k < (nmax > 0 ? nmax + 1 : 0); k++) {
    ktimer_t e;
    if (!ktimer_create(&e)) { set_error("roms_hip_step_timing: no event timers (hipEventCreate)"); return 2; }
    c->step_ev.push_back(e);
  }
  return 0;
}
extern "C" int roms_hip_step_times(roms_hip_ctx *c, double *ms, int cap) {
  if (!c || !ms || c->step_ev_n < 2 || !ktimer_sync(c->step_ev[c->step_ev_n - 1])) return 0;
  int n = 0;
  for (int k = 1; k < c->step_ev_n && n < cap; k++)
    if (!ktimer_elapsed_ms(c->step_ev[k - 1], c->step_ev[k], &ms[n++])) return n - 1;
  return n;
}

extern "C" int roms_hip_main3d(roms_hip_ctx *c, int nsteps) {
  if (!c) return 8;
  static const bool host_trace = getenv("ROMS_HIP_TRACE_HOST") != nullptr;   // measurement aid: host time to enqueue the steps
  const auto t0 = std::chrono::steady_clock::now();
  for (int n = 0; n < nsteps; n++) {
    if (!c->step_ev.empty() && c->step_ev_n == 0) { ktimer_record(c->step_ev[0], c->stream0); c->step_ev_n = 1; }
    int r = main3d_one(c);
    if (r) return r;
    if (c->step_ev_n > 0 && c->step_ev_n < (int)c->step_ev.size()) ktimer_record(c->step_ev[c->step_ev_n++], c->stream0);
  }
  if (c->diag_join_pending) { lane_wait(c, 11); c->diag_join_pending = false; }   // (main3d_around_loop: diag's reductions on the side stream)
  c->ghost_ok = false;
  if (host_trace && nsteps > 0) {
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    fprintf(stderr, "roms_hip_main3d: %d steps enqueued in %.1f us of host time (%.1f us per step)\n", nsteps, us, us / nsteps);
  }
  // The reference tests the GLOBALLY reduced energies (diag.F:419-437 mp_reduce, then :510-540).  A tile
  // of a multi-tile run therefore does not stop on its local numbers -- one rank leaving the step loop
  // while its neighbours wait in the next exchange would hang them -- the caller reduces roms_hip_diag's
  // raw sums over the ranks and stops all of them together (roms_amd/tiling.py:TiledRun.check).
  if (c->diag_ran && c->cfg.NtileI * c->cfg.NtileJ == 1) {   // d_diag holds the report of this call's last NINFO point
    c->diag_ran = false;
    double out[16];
    int r = fetch_diag(c, c->d_diag, out);
    if (r) return r;
#ifndef ROMS_CPU_EMU
    if (c->loop_err && *(volatile unsigned long long *)c->loop_err) return ctx_check(c, "main3d");   // (rather than the blow-up its garbage causes)
#endif
    if (!(std::isfinite(out[0]) && std::isfinite(out[1])) || !(out[4] <= 20.0)) {   // diag.F:510-540
      set_error("blow-up: KE/PE not finite or MaxSpeed > 20 m/s");
      return 1;
    }
  }
  return 0;
}

