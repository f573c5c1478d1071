// kdefs.h -- kernel definition / launch macros for the HIP kernels of libroms_hip.so.
//
// Two kernel shapes are used throughout:
//
//  COOP   a thread block works cooperatively on one horizontal sub-tile of the GPU's
//         (xi,eta) tile -- exactly a ROMS "tile" with its private scratch arrays
//         (IminS:ImaxS,JminS:JmaxS) held in LDS.  Loop nests of the algorithm become
//         block-strided loops (KLOOP2) separated by __syncthreads() (KSYNC).
//  THREAD one thread per (i,j[,k]) point with no inter-thread communication:
//         point-wise updates and sigma-column recurrences (tridiagonal solves,
//         vertical integrals) with the column held in registers / private memory.
//
// The same kernel bodies can also be compiled by a host C++ compiler with
// -DROMS_CPU_EMU (tests/emu/): blocks and threads are then executed serially.  That
// build exists ONLY so that kernel logic can be unit-tested on machines without a GPU
// (tests/, not-gpu marker); it is never built by __graft_entry__.build(), never loaded
// by the roms_amd package and is not a fallback of the product library.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cmath>

#ifdef ROMS_CPU_EMU
#include <cstdlib>
#include <cstring>
#include <vector>
#define KDEV inline
#define KTID 0
#define KNT 1
#define KSYNC() ((void)0)
#define KSCHED_FENCE() ((void)0)
typedef void *kstream_t;
// (tagged 8-byte words another rank's kernel polls: the serial emulation has one rank per call, plain accesses)
typedef unsigned long long kword_t;
static inline void ksys_st(kword_t *p, kword_t v) { *p = v; }
static inline kword_t ksys_ld(const kword_t *p) { return *p; }
static inline long long kclock() { return 0; }
static inline void knap() {}
// (step-boundary events of roms_hip_step_timing: the serial emulation has no device clock)
typedef void *ktimer_t;
static inline bool ktimer_create(ktimer_t *e) { *e = nullptr; return false; }
static inline void ktimer_destroy(ktimer_t) {}
static inline void ktimer_record(ktimer_t, kstream_t) {}
static inline bool ktimer_sync(ktimer_t) { return false; }
static inline bool ktimer_elapsed_ms(ktimer_t, ktimer_t, double *) { return false; }
struct kdim3 { int x, y, z; };

#define COOP_KERNEL(name, ArgT) static inline void name##_body(const ArgT &a, int bx, int by, int bz, double *lds)
#define COOP_GLOBAL(name, ArgT)
#define COOP_GLOBAL_LB(name, ArgT, maxthreads)
#define COOP_GLOBAL_LB2(name, ArgT, maxthreads, minwaves)
// ROMS_EMU_ORDER=reverse (test aid): the blocks / threads of every launch run in the opposite order.  A launch whose
// threads communicate through global memory without a kernel boundary in between (a race on the device) then gives
// different bits; tests/test_kernels_emu.py requires the two orders to agree (round 4: the spline scratch shared
// between the tracers of a launch was such a race, invisible to one fixed serial order).
extern int g_emu_reverse;
#define EMU_IDX(q, n) (g_emu_reverse ? (n) - 1 - (q) : (q))
#define LAUNCH_COOP(name, gx, gy, gz, nthreads, lds_doubles, stream, args)               \
  do {                                                                                   \
    std::vector<double> lds_((size_t)(lds_doubles) + 8);                                 \
    for (int bz_ = 0; bz_ < (gz); bz_++)                                                 \
      for (int by_ = 0; by_ < (gy); by_++)                                               \
        for (int bx_ = 0; bx_ < (gx); bx_++) name##_body(args, EMU_IDX(bx_, gx), EMU_IDX(by_, gy), EMU_IDX(bz_, gz), lds_.data()); \
  } while (0)

#define LAUNCH_COOP_AS(label, name, gx, gy, gz, nthreads, lds_doubles, stream, args) \
  LAUNCH_COOP(name, gx, gy, gz, nthreads, lds_doubles, stream, args)

#define LAUNCH_THREAD_AS(label, name, nx, ny, nz, stream, args) LAUNCH_THREAD(name, nx, ny, nz, stream, args)
#define THREAD_KERNEL(name, ArgT) static inline void name##_body(const ArgT &a, int gx, int gy, int gz)
#define THREAD_GLOBAL(name, ArgT)
#define THREAD_GLOBAL_W(name, ArgT, minwaves)
#define THREAD_GLOBAL_S(name, ArgT, TXS, TYS)
#define LAUNCH_THREAD_S(label, name, TXS, TYS, nx, ny, nz, stream, args) LAUNCH_THREAD(name, nx, ny, nz, stream, args)
#define LAUNCH_THREAD(name, nx, ny, nz, stream, args)                                    \
  do {                                                                                   \
    for (int gz_ = 0; gz_ < (nz); gz_++)                                                 \
      for (int gy_ = 0; gy_ < (ny); gy_++)                                               \
        for (int gx_ = 0; gx_ < (nx); gx_++) name##_body(args, EMU_IDX(gx_, nx), EMU_IDX(gy_, ny), EMU_IDX(gz_, nz)); \
  } while (0)

// COL: one thread per sigma column like THREAD, plus `per_thread` doubles of block-shared (LDS)
// storage per column, addressed lds[k * KLS]
#define KLS 1
#define COL_KERNEL(name, ArgT) static inline void name##_body(const ArgT &a, int gx, int gy, int gz, double *lds)
#define COL_GLOBAL(name, ArgT)
#define LAUNCH_COL(name, nx, ny, nz, per_thread, stream, args)                           \
  do {                                                                                   \
    std::vector<double> lds_((size_t)(per_thread) + 8);                                  \
    for (int gz_ = 0; gz_ < (nz); gz_++)                                                 \
      for (int gy_ = 0; gy_ < (ny); gy_++)                                               \
        for (int gx_ = 0; gx_ < (nx); gx_++) name##_body(args, EMU_IDX(gx_, nx), EMU_IDX(gy_, ny), EMU_IDX(gz_, nz), lds_.data()); \
  } while (0)
#define LAUNCH_COL_AS(label, name, nx, ny, nz, per_thread, stream, args) LAUNCH_COL(name, nx, ny, nz, per_thread, stream, args)

#else  // ------------------------------------------------------------------ HIP / gfx950
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#define KDEV __device__ __forceinline__
#define KTID ((int)threadIdx.x)
#define KNT ((int)blockDim.x)
#define KSYNC() __syncthreads()
// nothing is scheduled across this point: keeps the loads of a later chunk of an unrolled column sweep
// from being hoisted above the earlier chunks (register pressure)
#define KSCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
typedef hipStream_t kstream_t;
// tagged 8-byte words another rank's kernel polls: system-scope relaxed accesses (write-through stores, loads past the caches)
typedef unsigned long long kword_t;
static __device__ __forceinline__ void ksys_st(kword_t *p, kword_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
static __device__ __forceinline__ kword_t ksys_ld(const kword_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
static __device__ __forceinline__ long long kclock() { return (long long)wall_clock64(); }
static __device__ __forceinline__ void knap() { __builtin_amdgcn_s_sleep(1); }
typedef hipEvent_t ktimer_t;
static inline bool ktimer_create(ktimer_t *e) { return hipEventCreate(e) == hipSuccess; }
static inline void ktimer_destroy(ktimer_t e) { (void)hipEventDestroy(e); }
static inline void ktimer_record(ktimer_t e, kstream_t s) { (void)hipEventRecord(e, s); }
static inline bool ktimer_sync(ktimer_t e) { return hipEventSynchronize(e) == hipSuccess; }
static inline bool ktimer_elapsed_ms(ktimer_t a, ktimer_t b, double *ms) { float t = 0.0f; if (hipEventElapsedTime(&t, a, b) != hipSuccess) return false; *ms = (double)t; return true; }

#define COOP_KERNEL(name, ArgT) static __device__ __forceinline__ void name##_body(const ArgT &a, int bx, int by, int bz, double *lds)
// the __global__ entry: dynamic LDS, block index -> sub-tile (with the XCD-aware remap done
// by the body through DGrid, see roms_ctx.h:block_rect)
// All COOP launches use 256 threads unless the kernel is declared with COOP_GLOBAL_LB; telling the
// compiler (instead of its default assumption of 1024) doubles the VGPR budget per thread.
#define COOP_GLOBAL_LB(name, ArgT, maxthreads)                                           \
  static __global__ void __launch_bounds__(maxthreads) name(const ArgT a) {              \
    extern __shared__ double lds_dyn_[];                                                 \
    name##_body(a, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, lds_dyn_);         \
  }
// ... with a minimum number of waves per SIMD as well (caps the VGPRs so that that many fit)
#define COOP_GLOBAL_LB2(name, ArgT, maxthreads, minwaves)                                \
  static __global__ void __launch_bounds__(maxthreads, minwaves) name(const ArgT a) {    \
    extern __shared__ double lds_dyn_[];                                                 \
    name##_body(a, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, lds_dyn_);         \
  }
#define COOP_GLOBAL(name, ArgT) COOP_GLOBAL_LB(name, ArgT, 256)
// per-kernel HIP-event timing (roms_hip_kprof, roms_hip.cpp): mode 0 off, 1 every launch
// (synchronous), 2 launches of one selected kernel (asynchronous event pairs)
extern int g_kprof_mode;
// every launch goes through hipExtLaunchKernelGGL: with the two events null it IS hipLaunchKernelGGL; kprof mode 3 hands
// the selected kernel a start / stop event pair, which the runtime fills from the dispatch packet's own begin / end
// timestamps -- the kernel's duration as rocprofv3 reports it, no marker packets in the queue, no gaps between launches
// counted (bench.py: roofline.avg_launch_us)
extern thread_local hipEvent_t g_kp_start, g_kp_stop;
#define ROMS_LAUNCH(kern, grid, block, lds, stream, ...) \
  hipExtLaunchKernelGGL(kern, grid, block, lds, stream, g_kp_start, g_kp_stop, 0, __VA_ARGS__)
int kprof_begin(const char *name, hipStream_t stream);
void kprof_end(int slot, hipStream_t stream);
#define KPROF_WRAP(name, stream, launch)                                                 \
  do {                                                                                   \
    int kp_ = g_kprof_mode ? kprof_begin(#name, stream) : -1;                            \
    launch;                                                                              \
    if (kp_ >= 0) kprof_end(kp_, stream);                                                \
  } while (0)
#define LAUNCH_COOP(name, gx, gy, gz, nthreads, lds_doubles, stream, args)               \
  KPROF_WRAP(name, stream,                                                               \
  ROMS_LAUNCH(name, dim3((unsigned)(gx), (unsigned)(gy), (unsigned)(gz)), dim3((unsigned)(nthreads)), \
                     (size_t)(lds_doubles) * sizeof(double), stream, args))

// variant `name` of a kernel reported to the profiler hooks as `label`
#define LAUNCH_COOP_AS(label, name, gx, gy, gz, nthreads, lds_doubles, stream, args)     \
  KPROF_WRAP(label, stream,                                                              \
  ROMS_LAUNCH(name, dim3((unsigned)(gx), (unsigned)(gy), (unsigned)(gz)), dim3((unsigned)(nthreads)), \
                     (size_t)(lds_doubles) * sizeof(double), stream, args))

#define THREAD_KERNEL(name, ArgT) static __device__ __forceinline__ void name##_body(const ArgT &a, int gx, int gy, int gz)
// XCD-aware block order.  MI355X hands consecutive workgroups to its 8 XCDs round-robin (block b ->
// XCD b % 8), each XCD with its own L2.  The 64x4-point blocks of a launch are numbered eta-fastest
// (t = xi-block * nby + eta-block) and that list is cut into 8 contiguous segments, one per XCD; the
// linear block index is decoded as xcd = b % 8, then (fastest to slowest) level gz, position in the
// XCD's segment.  All levels of a block column and (up to the segment ends) its eta-neighbours run on
// the same XCD and close in time, so the vertical (k-1..k+2) and eta (j-2..j+2) neighbours a
// point-wise kernel re-reads hit that XCD's L2, and every XCD gets the same number of blocks whatever
// the grid shape.  (Measured with rocprofv3 FETCH_SIZE: profiles/, DESIGN.md.)
#ifndef KTY
#define KTY 4   // eta rows per THREAD block (64 x KTY threads)
#endif
#ifndef KMINW
#define KMINW 1   // minimum waves per SIMD the THREAD kernels are compiled for (experiment knob)
#endif
// tile number -> (xi block, eta block): KXIFAST 1 (default since round 4) = xi-fastest: an XCD's segment is a band of whole
// rows of blocks, so the xi-neighbours of a block -- whose stencil reads share the 128-byte lines at the block's ends --
// and its eta-neighbours run on the same XCD, and a row of single-wave COL blocks reads whole array rows; 0 = eta-fastest
// (rounds 1-3: an XCD's segment was a 64-point-wide strip, every stencil array fetched 6 lines per row for 4).  Measured
// (rocprofv3, 512x512x50): HBM traffic of the LDS-tiled kernels 1.46-1.53 -> 1.23-1.41 x algorithmic, of the point and
// column kernels 1.2 -> 1.05; serial step 90.4 -> 88.2 ms / 13 steps, BENCHMARK3 115.3 -> 108.9 ms / 11 steps.
#ifndef KXIFAST
#define KXIFAST 1
#endif
#if KXIFAST
// ... in column groups of KXIGROUP blocks: tiles are numbered xi-fastest inside a group of KXIGROUP xi blocks, then eta,
// then group by group.  On a grid 8 blocks wide (512 points) that IS xi-fastest; on a wider one (2048 points = 32
// blocks) the ~64 blocks an XCD runs at a time form an 8 x 8 patch instead of two full rows, so that the eta
// neighbours -- whose staged rectangles overlap by half in the LDS-tiled kernels -- are in flight together too
// (BENCHMARK3, k_rhs3d_lds: traffic 1.67 x with plain rows against 1.46 x eta-fastest).
#ifndef KXIGROUP
#define KXIGROUP 8
#endif
KDEV void ktile_xy(int t, int nbx, int nby, int &tx, int &ty) {
  const int full = nbx / KXIGROUP, per = KXIGROUP * nby;
  int g = t / per;
  if (g >= full) g = full;                       // the last, narrower group
  const int w = g < full ? KXIGROUP : nbx - full * KXIGROUP;
  const int r = t - g * per;
  ty = r / w;
  tx = g * KXIGROUP + (r - ty * w);
}
#define KTILE_XY(t_, nbx_, nby_, tx_, ty_) int tx_, ty_; ktile_xy((t_), (nbx_), (nby_), tx_, ty_)
#else
#define KTILE_XY(t_, nbx_, nby_, tx_, ty_) const int tx_ = (t_) / (nby_), ty_ = (t_) - tx_ * (nby_)
#endif
// rim / interior split (DGrid::region): does the point (gx,gy) of an nx x ny index space belong to this launch?
// (argument structs without a DGrid -- the streaming copy, the relayout kernel -- always run whole)
template <class A> KDEV auto kregion_of(const A &a, int) -> decltype((void)a.G.region, int()) { return a.G.region; }
template <class A> KDEV int kregion_of(const A &, long) { return 0; }
template <class A> KDEV auto krimw_of(const A &a, int) -> decltype((void)a.G.rimw, int()) { return a.G.rimw; }
template <class A> KDEV int krimw_of(const A &, long) { return 0; }
#define KREGION_RIM(w_, gx_, gy_, nx_, ny_) ((gx_) < (w_) || (gx_) >= (nx_) - (w_) || (gy_) < (w_) || (gy_) >= (ny_) - (w_))
#define KREGION_OK(a_, gx_, gy_, nx_, ny_) (kregion_of(a_, 0) == 0 || ((kregion_of(a_, 0) == 1) == KREGION_RIM(krimw_of(a_, 0), gx_, gy_, nx_, ny_)))
#define THREAD_GLOBAL(name, ArgT) THREAD_GLOBAL_W(name, ArgT, KMINW)
// ... compiled for at least `minwaves` waves per SIMD (caps the VGPRs)
#define THREAD_GLOBAL_W(name, ArgT, minwaves)                                            \
  static __global__ void __launch_bounds__(256, minwaves) name(const ArgT a, int nx, int ny, int nz) {    \
    const int nby_ = (ny + KTY - 1) / KTY, nt_ = ((nx + 63) / 64) * nby_, seg_ = (nt_ + 7) / 8;  \
    const int r_ = (int)(blockIdx.x >> 3), xcd_ = (int)(blockIdx.x & 7);                 \
    const int gz = r_ % nz;                                                              \
    const int t_ = xcd_ * seg_ + r_ / nz;                                                \
    if (t_ >= nt_) return;                                                               \
    KTILE_XY(t_, (nx + 63) / 64, nby_, tx_, ty_);                                        \
    const int gx = tx_ * 64 + (int)threadIdx.x;                                          \
    const int gy = ty_ * KTY + (int)threadIdx.y;                                           \
    if (gx < nx && gy < ny && KREGION_OK(a, gx, gy, nx, ny)) name##_body(a, gx, gy, gz); \
  }
// ... the same with blocks of TXS x TYS points (TXS * TYS = 256): a stencil kernel that reads eta-neighbours fetches
// (TYS + halo) / TYS rows per row of its own -- 1.75 with 64 x 4 and two rows below / one above, 1.19 with 16 x 16
#define THREAD_GLOBAL_S(name, ArgT, TXS, TYS)                                            \
  static __global__ void __launch_bounds__(256, KMINW) name##_t##TXS##x##TYS(const ArgT a, int nx, int ny, int nz) { \
    const int nbx_ = (nx + TXS - 1) / TXS, nby_ = (ny + TYS - 1) / TYS, nt_ = nbx_ * nby_, seg_ = (nt_ + 7) / 8;  \
    const int r_ = (int)(blockIdx.x >> 3), xcd_ = (int)(blockIdx.x & 7);                 \
    const int gz = r_ % nz;                                                              \
    const int t_ = xcd_ * seg_ + r_ / nz;                                                \
    if (t_ >= nt_) return;                                                               \
    KTILE_XY(t_, nbx_, nby_, tx_, ty_);                                                  \
    const int gx = tx_ * TXS + (int)threadIdx.x;                                         \
    const int gy = ty_ * TYS + (int)threadIdx.y;                                         \
    if (gx < nx && gy < ny && KREGION_OK(a, gx, gy, nx, ny)) name##_body(a, gx, gy, gz); \
  }
#define LAUNCH_THREAD_S(label, name, TXS, TYS, nx, ny, nz, stream, args)                 \
  KPROF_WRAP(label, stream,                                                              \
  ROMS_LAUNCH(name##_t##TXS##x##TYS, dim3((unsigned)(8 * ((((((nx) + TXS - 1) / TXS) * (((ny) + TYS - 1) / TYS)) + 7) / 8) * (nz)), 1, 1), \
                     dim3(TXS, TYS, 1), g_thread_ballast, stream, args, (int)(nx), (int)(ny), (int)(nz)))
// 64 lanes along xi (coalesced), 4 rows of eta per block; 1-D grid of 8*ceil(blocks/8)*nz workgroups
// g_thread_ballast: bytes of (unused) dynamic LDS a THREAD launch asks for -- caps its blocks per CU, so that
// kernels placed beside the barotropic loop leave every CU room for a block of k_step2d (main3d_late)
extern thread_local size_t g_thread_ballast;   // (per host thread: several contexts may be driven from one process)
#define LAUNCH_THREAD(name, nx, ny, nz, stream, args)                                    \
  KPROF_WRAP(name, stream,                                                               \
  ROMS_LAUNCH(name, dim3((unsigned)(8 * ((((((nx) + 63) / 64) * (((ny) + KTY - 1) / KTY)) + 7) / 8) * (nz)), 1, 1), \
                     dim3(64, KTY, 1), g_thread_ballast, stream, args, (int)(nx), (int)(ny), (int)(nz)))
#define LAUNCH_THREAD_AS(label, name, nx, ny, nz, stream, args)                           \
  KPROF_WRAP(label, stream,                                                              \
  ROMS_LAUNCH(name, dim3((unsigned)(8 * ((((((nx) + 63) / 64) * (((ny) + KTY - 1) / KTY)) + 7) / 8) * (nz)), 1, 1), \
                     dim3(64, KTY, 1), g_thread_ballast, stream, args, (int)(nx), (int)(ny), (int)(nz)))
// COL kernels: one thread per sigma column with `per_thread` doubles of LDS each (the elimination
// coefficients of a tridiagonal solve, a column kept between sweeps).  Blocks are single waves
// (64 columns along xi, one eta row) so that the LDS footprint of a block stays small and several fit
// a CU; lds[k * KLS] is level k of the thread's column (consecutive lanes, consecutive banks).  Same
// XCD-aware block order as THREAD launches.
#define KLS 64
#define COL_KERNEL(name, ArgT) static __device__ __forceinline__ void name##_body(const ArgT &a, int gx, int gy, int gz, double *lds)
#define COL_GLOBAL(name, ArgT)                                                           \
  static __global__ void __launch_bounds__(64) name(const ArgT a, int nx, int ny, int nz) {     \
    extern __shared__ double lds_dyn_[];                                                 \
    const int nt_ = ((nx + 63) / 64) * ny, seg_ = (nt_ + 7) / 8;                         \
    const int r_ = (int)(blockIdx.x >> 3), xcd_ = (int)(blockIdx.x & 7);                 \
    const int gz = r_ % nz;                                                              \
    const int t_ = xcd_ * seg_ + r_ / nz;                                                \
    if (t_ >= nt_) return;                                                               \
    KTILE_XY(t_, (nx + 63) / 64, ny, tx_, gy);                                           \
    const int gx = tx_ * 64 + (int)threadIdx.x;                                          \
    if (gx < nx && KREGION_OK(a, gx, gy, nx, ny)) name##_body(a, gx, gy, gz, lds_dyn_ + threadIdx.x); \
  }
#define LAUNCH_COL_AS(label, name, nx, ny, nz, per_thread, stream, args)                 \
  KPROF_WRAP(label, stream,                                                              \
  ROMS_LAUNCH(name, dim3((unsigned)(8 * (((((nx) + 63) / 64) * (ny) + 7) / 8) * (nz)), 1, 1), \
                     dim3(64, 1, 1), (size_t)(per_thread) * 64 * sizeof(double), stream, args, (int)(nx), (int)(ny), (int)(nz)))
#define LAUNCH_COL(name, nx, ny, nz, per_thread, stream, args) LAUNCH_COL_AS(name, name, nx, ny, nz, per_thread, stream, args)
#endif

// block-strided loop nest over the rectangle [ilo,ihi] x [jlo,jhi] (inclusive), i fastest
#define KLOOP2(i, j, ilo, ihi, jlo, jhi)                                                 \
  for (int w_ = (ihi) - (ilo) + 1, n_ = (w_ > 0 && (jhi) >= (jlo)) ? w_ * ((jhi) - (jlo) + 1) : 0, \
           q_ = KTID, i = 0, j = 0;                                                      \
       q_ < n_ && ((j = (jlo) + q_ / w_), (i = (ilo) + q_ - (j - (jlo)) * w_), true); q_ += KNT)
// block-strided 1-D loop
#define KLOOP1(i, ilo, ihi) for (int i = (ilo) + KTID; i <= (ihi); i += KNT)

#define KMAX(a, b) ((a) > (b) ? (a) : (b))
#define KMIN(a, b) ((a) < (b) ? (a) : (b))
