// roms_host.h -- host-side context of libroms_hip.so (shared by the translation units).
#pragma once
#include "roms_ctx.h"
#include <cstdlib>
#include <string>
#include <vector>

#ifdef ROMS_CPU_EMU
typedef void *kevent_t;
#else
typedef hipEvent_t kevent_t;
#endif

struct FieldDesc {
  const char *name;
  size_t offset;   // offsetof(Fields, name)
  int kind;        // plane count rule, see field_planes()
};

enum { FK_2D = 0, FK_R, FK_W, FK_2Dx3, FK_2Dx2, FK_Rx2, FK_T, FK_Wx2, FK_2DxNT, FK_WxNAT, FK_RxNT, FK_TABR, FK_TABW,
       FK_BJ, FK_BI, FK_BJN, FK_BIN, FK_BJT, FK_BIT, FK_Wx3 };   // boundary lines (LBj:UBj) / (LBi:UBi) [, N [, NT]] in the caller's bounds

struct Region { double seconds; long calls; };

// channels of the mailbox transport: exchanges of one channel are ordered by the stream they are issued on, and every
// rank issues the same sequence per channel -- 0 the main compute stream, 1 the exchange stream (asynchronous 3-D
// exchanges), 2..4 the side streams of the schedule around the barotropic loop (round 6: stream2, stream3, stream4)
#define PEER_NCH 5
// inter-tile communication state (multi-GPU runs: one tile per process/GPU)
struct TileComm {
  int nbr[8];                   // ranks of the W, E, S, N, SW, SE, NW, NE neighbours (-1: none)
  double *sbuf[8], *rbuf[8];    // device staging buffers per neighbour
  size_t cap;                   // doubles per staging buffer
  roms_hip_exchange_fn fn;      // user transport (MPI, gloo ...) or null
  void *user;
  void *nccl;                   // ncclComm_t of the built-in RCCL transport or null
  // mailbox transport (roms_hip_comm_peer): my slab, the neighbours' slabs as mapped here, slot tables
  void *peer_slab;              // uncached device memory: slot table, arrival words [channel][direction][plane], then the receive buffers [channel][parity][direction]
  size_t peer_bytes;
  int peer_planes;              // capacity of a slot in planes
  size_t peer_off[PEER_NCH][2][8];   // byte offsets of MY slots [channel][parity][direction] (same table in the slab's header for the neighbours)
  void *peer_map[8];            // neighbour d's slab in this process's address space (null: none)
  size_t peer_noff[8][PEER_NCH][2];  // offset in neighbour d's slab of the slot my message to it goes into (its direction opp[d])
  bool peer_opened[8];          // mapped with hipIpcOpenMemHandle (closed at destroy)
  unsigned long long peer_seq[PEER_NCH];   // exchanges issued per channel
  unsigned long long *peer_err; // pinned host word: set by an unpack kernel whose message did not arrive in time
  bool peer_on;
  bool peer_shared;             // a neighbour rank runs on the SAME device (several ranks sharing one GPU: test set-ups)
  long nexchanges;
  // the persistent barotropic loop across tiles (k_step2d_loop.h, S2LPeer): my rim planes inside the slab, and the
  // neighbours' as their blobs describe them
  size_t loop_rim_off, loop_ring_off;   // byte offsets in MY slab (0: the slab has no such region)
  int loop_nb2[2];                      // the loop's sub-tile grid the region was sized for
  struct PeerGeom { int LBi, LBj, ni, nj, nbx2, nby2; size_t rim_off, ring_off; unsigned busid; } ngeom[8];
  unsigned busid;                       // PCI bus id of my device (domain << 16 | bus << 8 | device << 3 | function)
};

struct roms_hip_ctx {
  roms_hip_config cfg;
  TileComm comm;
  int peer_threads;             // block size of the mailbox pack/unpack kernels
  bool comm_failed;             // a halo exchange failed (reported by the next ctx_check)
  bool has_exchange;            // some neighbour is reached through the transport (multi-tile, or the self-exchange test aid)
  bool swdk_ready;              // main3d_one has launched k_swdk already (side stream): run_pre_step3d skips it
  int tadv_hdone = 0, tadv_vdone = 0;   // step3d_t: tracers whose HSIMT horizontal step / whose vertical advection the LDS-tiled kernel did (k_tadv_lds.h, HS)
  bool pre_t3_ready;            // main3d_one has launched the tracer predictor of pre_step3d already (side stream)
  bool tmix_terms = false;      // run_t3dmix2 stores what it adds to t(nnew) in F.tmix instead (it runs ahead of pre_step3d)
  bool tmix_ready = false;      // ... and has done so for this step: k_pre_new adds the terms, the later t3dmix2 call is a no-op
  bool fold_uvmix = false;      // k_pre_new adds the uv3dmix2 terms to the u,v(nnew) it sets (no k_uv3dmix2_apply launch)
  // time-averaged fields (set_avg.F; g_avg.cpp): off until roms_hip_avg_config
  double *avg[24];
  int avg_nAVG, avg_ntsAVG, avg_nrrec, avg_ntstart;
  unsigned avg_mask;
  double avg_time;              // AVGtime (mod_scalars.F): time stamp of the record wrt_avg writes
  int avg_done_iic;             // roms_hip_output_point has run set_avg for this step already
  // per-term tracer tendencies (DIAGNOSTICS_TS; g_dia.cpp): off until roms_hip_dia_config
  int dia_nDIA = 0, dia_ntsDIA = 1, dia_nrrec = 0, dia_ntstart = 1, dia_done_iic = -1;
  double dia_time = 0.0;        // DIAtime (mod_scalars.F)
  bool late_pre;                // main3d_one runs pre_step3d BEHIND prsgrd/rhs3d_tile/uv3dmix2 (beside the barotropic loop):
                                // k_prs_grad keeps the old ru/rv bracket, k_uv3dmix2_s only stores its terms, k_pre_new uses both
  bool m2d_dirty;               // grid arrays uploaded since Fields::m2r/m2p were packed (g_step2d.cpp)
  // barotropic predictor+corrector pairs as one launch (k_step2d_pair.h)
  // Multi-tile runs with the pair kernel keep their arrays with a WIDER ghost zone than the caller's LBi:UBi x LBj:UBj
  // (B2D_GL | B2D_GH lines towards every neighbouring tile): c->G describes the library's layout, cLB*/cn* the caller's;
  // upload / download repack (roms_hip.cpp:relayout)
  int cLBi, cLBj, cni, cnj;
  bool wide;                    // the two layouts differ
  bool pair_mt;                 // multi-tile context set up for the pair kernel (wide 2-D exchanges)
  double *stage_buf;            // device staging of upload / download in the caller's layout
  size_t stage_cap;
  bool static_wide_dirty;       // the wide ghost lines of the time-invariant 2-D fields have not been exchanged yet
  bool pair_on;                 // roms_hip_main3d runs the fast steps iif >= 2 as pairs
  int b2_stage;                 // physical level (4 | 5) of zeta/ubar/vbar holding the last pair's result, not yet committed
                                // to its logical level; 0: none
  // the fast steps 2 .. nfast as one persistent launch (k_step2d_loop.h)
  int loop_state;               // 0: not decided yet, 1: roms_hip_main3d uses it, -1: it does not
  unsigned *loop_flags;         // arrival words of the sub-tiles (device)
  unsigned loop_epoch;          // ... hold at most this value (the pairs of all launches so far)
  double *loop_wts;             // weights per pair (device)
  unsigned long long *loop_err; // pinned host word: a wait for a neighbouring block gave up (ctx_check reports it)
  std::vector<ktimer_t> step_ev;   // roms_hip_step_timing: events at the step boundaries (main stream)
  int step_ev_n = 0;
  bool ghost_ok = false;        // inside roms_hip_main3d, behind post_initial: every input of the point-wise producers carries valid ghost lines (ghost_compute)
  bool h_ghost_done = false;    // (multi-tile) the ghost lines of h have been exchanged once (run_set_depth computes the ghost columns itself)
  // the pair launches handing their rim across tile edges themselves (tiles too large for the loop; g_step2d.cpp:pair_rim_usable)
  double *avg_cnt[3] = {nullptr, nullptr, nullptr};   // AVERAGES with WET_DRY: the wet-point counters of set_avg.F (rho, u, v)
  int pair_rim_state = 0;       // 0: not decided, 1: on, -1: off
  bool rim_refused = false;     // roms_hip_rim_disable: no rim hand-off inside the barotropic launches (until another transport is installed)
  bool b2_rim = false;          // the staged result of the last pair launch was published into the neighbours' rim planes (not exchanged)
  unsigned pair_epoch = 0;      // number of the last published pair (the tag its points carry)
  bool loop_pre_frc = false;    // (multi-tile) this step's schedule has already exchanged what the loop's first fast step reads beyond the tile:
  bool loop_pre_state = false;  // the 3-D forcing and its history | the kstp level of zeta, ubar, vbar (step2d_loop_pre)
  bool diag_ran;                // a diag report was enqueued since the last blow-up test (roms_hip_main3d)
  int diag_step = -1;           // step count (iic-1) of the report in d_diag, -1: none yet
  DGrid G;
  Fields F;                     // host copy of the pointer table
  Fields *d_F;                  // the same table in device memory: kernels take it by pointer, which
                                // keeps the by-value kernel arguments small
  kstream_t stream;
  kstream_t stream0;   // the main compute stream (`stream` is switched between the lanes of a schedule; this one is not)
  std::vector<void *> allocs;
  roms_hip_stepping s;
  bool profile;
  Region regions[96];
  kevent_t ev0, ev1;
  double *d_diag;      // device scratch for diag reductions
  double *d_diagwork;  // column/row partial results of diag (own buffer: diag overlaps other kernels)
  kstream_t stream2;   // side stream: kernels of a step that do not depend on each other overlap
  kstream_t stream4;   // third side stream (main3d_around_loop: the vertical-mixing closure beside the chain in front of the barotropic loop)
  bool diag_join_pending;   // diag's reductions of the last step run on a side stream and the main stream has not joined them yet (lane event 11)
  bool kpp_col_ok;     // the schedule keeps KPP away from the barotropic loop: its one-kernel LDS form may be used (g_bench.cpp)
  kstream_t stream3;   // second side stream (main3d_one, small grids: diag/wvelocity, then the kernels that run beside the barotropic loop)
  kevent_t ev_fork, ev_join, ev_point;
  kevent_t ev_lane[12];
  bool overlap;        // use the side stream (single-GPU latency hiding on small grids)
  double *h_diag;      // pinned host mirror
  int nblk_diag;
  // Halo exchange overlapped with compute (multi-tile contexts, built-in RCCL transport): the exchange of a
  // group of fields runs on its own stream behind the producing kernel; a later kernel waits for it only if
  // it touches that group (roms_hip.cpp: halo_fence, entry_groups).
  kstream_t xstream;
  kevent_t ev_xprod;            // "producer done" marker: compute stream -> exchange stream
  kevent_t ev_x[32];            // completion events of exchanges, a rotating pool
  int ev_x_next;
  int x_event_of[16];           // per field group: index into ev_x of the last exchange that carried it, -1 none
  unsigned x_pending;           // field groups with an exchange possibly still in flight
  bool x_async;
  bool rim_split;               // 3-D producers in front of an asynchronous exchange run rim first, interior beside the exchange (round 4)
  bool x_tail;                  // the exchange being launched is the last operation of its routine
  bool x_2d_ok;                 // ... may go to the exchange stream although it carries 2-D state (step3d_uv: every reader of ubar, vbar fences FG_2D)
  bool x_wide;                  // ... carries the wide strips of the barotropic pair kernel (launch_halo_wide)
  int x_min_planes;             // exchanges with fewer planes stay on the compute stream
};

// Field groups of the halo-exchange dependency tracking (a kernel entry names the groups it reads or writes)
enum {
  FG_FLUX = 1 << 0,   // surface/bottom forcing and fluxes: sustr .. lrflx, stflx, btflx, srflx, the bulk atmosphere
  FG_RHO = 1 << 1,    // rho pden rhoA rhoS bvf alpha beta
  FG_MF = 1 << 2,     // Huon Hvom
  FG_W = 1 << 3,      // W
  FG_WVEL = 1 << 4,   // wvel
  FG_AK = 1 << 5,     // Akv Akt ghats hsbl
  FG_T = 1 << 6,      // t
  FG_UV = 1 << 7,     // u v
  FG_2D = 1 << 8,     // zeta ubar vbar rzeta rubar rvbar
  FG_AVG = 1 << 9,    // Zt_avg1 DU_avg1 DU_avg2 DV_avg1 DV_avg2
  FG_HZ = 1 << 10,    // Hz z_r z_w
  FG_R = 1 << 11,     // ru rv rufrc rvfrc
  FG_OTHER = 1 << 12, // anything else (grid metrics, work arrays)
  FG_T3 = 1 << 13,    // t(:,:,:,3,:), the tracer predictor: exchanged behind pre_step3d, read by step3d_t only (round 4)
  FG_NGROUPS = 14,
  FG_ALL = (1 << 14) - 1
};
// make the compute stream(s) wait for the exchanges in flight that carry any of `groups`
void halo_fence(roms_hip_ctx *c, unsigned groups);

// LBC(edge, variable) with the default resolved: periodic where the direction is, else closed unless the caller chose a kind
inline int lbc_kind(const roms_hip_config &cf, int edge, int var) {
  if ((edge == ROMS_IWEST || edge == ROMS_IEAST) ? cf.EWperiodic : cf.NSperiodic) return ROMS_LBC_PER;
  return cf.lbc[edge][var] == ROMS_LBC_DEFAULT ? ROMS_LBC_CLO : cf.lbc[edge][var];
}
int run_obc_flux(roms_hip_ctx *c, int kinp);     // obc_flux_tile (VolCons)
int run_obc2d(roms_hip_ctx *c, int kout, unsigned vars = 7);     // zetabc (1), u2dbc (2), v2dbc (4) of level kout (g_obc.cpp)
int run_obc3d_uv(roms_hip_ctx *c, int nout);                     // u3dbc, v3dbc
int run_obc3d_t(roms_hip_ctx *c, int nout, int itrc);            // t3dbc of tracer itrc (1-based)

// helpers (roms_hip.cpp)
void ctx_sync_stepping(roms_hip_ctx *c);           // copy c->s into c->G
// side-stream helpers (roms_hip.cpp): between side_begin and side_end launches go to the side stream,
// ordered after everything launched so far; side_join makes the main stream wait for them
void side_mark(roms_hip_ctx *c);
void side_begin(roms_hip_ctx *c);
void side_end(roms_hip_ctx *c);
void side_join(roms_hip_ctx *c);
void side_point(roms_hip_ctx *c);
void side_join_point(roms_hip_ctx *c);
bool lanes_on(roms_hip_ctx *c);
void lane_record(roms_hip_ctx *c, int e);
void lane_wait(roms_hip_ctx *c, int e);
int ctx_check(roms_hip_ctx *c, const char *what);  // hipGetLastError -> exit_flag style code
void set_error(const std::string &msg);
long field_elems(const roms_hip_ctx *c, int kind);          // in the library's layout (allocation)
long field_elems_caller(const roms_hip_ctx *c, int kind);   // in the caller's layout (what upload / download move)
int field_planes(const roms_hip_ctx *c, int kind);          // horizontal planes of a field, -1: a table without horizontal extent
const FieldDesc *find_field(const char *name);

// halo / BC launcher (k_halo.h): nk planes starting at A
struct HaloSpec { double *A; int nk; int bc; char gtype; };
// boundary kind of the state variables zeta and t (zetabc.F, t3dbc_im.F): the gradient value carries rmask of the boundary point
// in a masked run; `all`: and the whole plane is multiplied by rmask afterwards (step3d_t.F:1880)
inline int bc_rstate(const roms_hip_ctx *c, bool all = false) {
  return c->G.masking ? (BC_R | BC_MASKF | (all ? BC_MASKALL : 0)) : BC_R;
}
// a context with open edges applies the state's boundary conditions in k_obc launches (g_obc.cpp): its halo launches
// keep the flags (BC_MASKALL) and the exchange, not the fill
inline int obc_bc(const roms_hip_ctx *c, int bc) { return c->G.obc ? (bc & ~BC_KIND) : bc; }
void launch_halo(roms_hip_ctx *c, double *A, int nk, int bc, char gtype);
// the same as the LAST operation of a routine: nothing enqueued later in that routine depends on it, so in a
// multi-tile run the exchange may go to the exchange stream and overlap the routines that follow (halo_fence)
void launch_halo_tail(roms_hip_ctx *c, const HaloSpec *sp, int n);
void launch_halo_multi(roms_hip_ctx *c, const HaloSpec *sp, int n);   // n <= 8 fields in one launch
void launch_halo_wide(roms_hip_ctx *c, const HaloSpec *sp, int n);    // ... with strips B2D_GL | B2D_GH lines wide (pair kernel)
// the boundary fills of an exchange point WITHOUT the exchange (multi-tile contexts, round 6: the ghost lines of these fields are
// computed with the tile or nobody reads them before the next exchange of the same field), on the rectangle X -- the tile, or the
// tile with the ghost lines a ghost-computing producer has just written (ghost_tb)
void launch_fill_only(roms_hip_ctx *c, const HaloSpec *sp, int n, const TB &X);
int run_set_avg(roms_hip_ctx *c, int part = 0);   // g_avg.cpp
int run_set_diags(roms_hip_ctx *c);
// WET_DRY (g_wetdry.cpp)
int run_obc_tke(roms_hip_ctx *c, int nout);                                // g_obc.cpp
int run_wetdry(roms_hip_ctx *c, int mode);
int run_wd_scale3(roms_hip_ctx *c);
int run_wd_eff(roms_hip_ctx *c);
// DIAGNOSTICS_UV (g_duv.cpp)
int duv_config(roms_hip_ctx *c);
double *duv_field(roms_hip_ctx *c, const char *name, int *np);
int run_duv_pgrd(roms_hip_ctx *c);
int run_duv_frc(roms_hip_ctx *c);
int run_duv_s3uv(roms_hip_ctx *c);
int run_set_diags_uv(roms_hip_ctx *c, int phase, double fac);               // g_dia.cpp
int run_dia_rate(roms_hip_ctx *c);
bool launch_tadv_lds(roms_hip_ctx *c, int mode);  // g_rhs3d.cpp
int avg_field_index(const char *name);
long avg_field_elems(const roms_hip_ctx *c, int f);

// region timing
struct RegionTimer {
  roms_hip_ctx *c; int id;
  RegionTimer(roms_hip_ctx *c_, int id_);
  ~RegionTimer();
};

// kernel groups (one translation unit each)
int run_set_depth(roms_hip_ctx *c);
int run_set_massflux(roms_hip_ctx *c);
int run_rho_eos(roms_hip_ctx *c);
int run_set_vbc(roms_hip_ctx *c);
int run_ana_vmix(roms_hip_ctx *c);
int run_set_data(roms_hip_ctx *c);
int run_omega(roms_hip_ctx *c);
int run_wvelocity(roms_hip_ctx *c, int ninp);
int run_set_zeta(roms_hip_ctx *c);
int run_ini_zeta(roms_hip_ctx *c);
int run_ini_fields(roms_hip_ctx *c);
int run_pre_step3d(roms_hip_ctx *c);
int run_prsgrd(roms_hip_ctx *c);
int run_t3dmix2(roms_hip_ctx *c);
int run_uv3dmix2(roms_hip_ctx *c);
int run_rhs3d_tile(roms_hip_ctx *c);
int run_step2d(roms_hip_ctx *c);
int run_step2d_pair(roms_hip_ctx *c);     // predictor (c->G = its stepping) + corrector of one fast step
bool step2d_pair_usable(const roms_hip_ctx *c);
bool step2d_loop_usable(roms_hip_ctx *c);   // fast steps 2 .. nfast as one persistent launch (k_step2d_loop.h)
int run_rim_probe(roms_hip_ctx *c, int reps);      // self-check of the rim planes (g_step2d.cpp)
void rim_disable(roms_hip_ctx *c);
int step2d_loop_pre(roms_hip_ctx *c, int what);   // (multi-tile) 1: exchange rufrc, rvfrc and the AB3 history; 2: the kstp level of the barotropic state, wide
void step2d_loop_dims(const roms_hip_ctx *c, int &nbx2, int &nby2);   // its sub-tile grid (the mailbox slab holds a ring of arrival words around it)
int run_step2d_loop(roms_hip_ctx *c);       // c->G = the stepping of the predictor call of iif = 2, or of iif = 1: then the first fast
                                            // step and the auxiliary call iif = nfast+1 run inside the launch too
int run_step3d_uv(roms_hip_ctx *c);
int run_step3d_t(roms_hip_ctx *c);
int run_lmd_vmix(roms_hip_ctx *c);
int run_bulk_flux(roms_hip_ctx *c);
int run_gls_prestep(roms_hip_ctx *c);
int run_gls_corstep(roms_hip_ctx *c);
int run_diag(roms_hip_ctx *c, double *out);
int run_copy_probe(roms_hip_ctx *c, int reps);

// A multi-tile context: the tile's bounds extended by gl | gh ghost lines towards every side that is not a physical edge.
// A POINT-WISE producer whose inputs are valid on those lines computes them itself -- the same expression on the same
// operands the neighbour evaluates at its own points, so the same bits -- and the strip exchange the reference issues behind
// it (exchange_r3d_tile + mp_exchange3d at the tail of set_depth, ana_* ...) is not needed.  Only inside roms_hip_main3d, behind
// the first step's post_initial (roms_hip_ctx::ghost_ok): roms_hip_start and the per-routine entries work on whatever the caller
// uploaded, whose ghost lines the reference does not promise either.  ROMS_HIP_GHOSTCOMP=0: exchange.
// (a mask: 1 set_depth, 2 set_data, 4 rho_eos, 8 set_massflux, 32 the boundary values of u, v(nnew) in step3d_uv without their
// exchange, 64 wvelocity without a second exchange of DU_avg1, DV_avg1; default all)
inline bool ghost_compute(const roms_hip_ctx *c, int which) {
  static const char *e = getenv("ROMS_HIP_GHOSTCOMP");
  static const int mask = e ? atoi(e) : 127;
  return c->has_exchange && c->ghost_ok && (mask & which);
}
inline TB ghost_tb(const roms_hip_ctx *c, int gl, int gh) {
  const roms_hip_config &cf = c->cfg;
  // every side that is not a physical edge of the domain: a neighbouring rank, or the tile's own periodic image
  const bool xw = !(cf.west_edge && !cf.EWperiodic), xe = !(cf.east_edge && !cf.EWperiodic);
  const bool xs = !(cf.south_edge && !cf.NSperiodic), xn = !(cf.north_edge && !cf.NSperiodic);
  return make_bounds(cf.Lm, cf.Mm, cf.EWperiodic, cf.NSperiodic, cf.Istr - (xw ? gl : 0), cf.Iend + (xe ? gh : 0),
                     cf.Jstr - (xs ? gl : 0), cf.Jend + (xn ? gh : 0), cf.west_edge, cf.east_edge, cf.south_edge, cf.north_edge);
}

// pointer helpers for time levels
static inline double *t_lev(roms_hip_ctx *c, int n, int itrc) {
  return c->F.t + ((size_t)(n - 1) + 3 * (size_t)(itrc - 1)) * (size_t)c->G.nij * (size_t)c->G.N;
}
static inline double *uv_lev(roms_hip_ctx *c, double *q, int n) {
  return q + (size_t)(n - 1) * (size_t)c->G.nij * (size_t)c->G.N;
}
static inline double *lev2d(roms_hip_ctx *c, double *q, int n) { return q + (size_t)(n - 1) * (size_t)c->G.nij; }
