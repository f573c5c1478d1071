// g_bench.cpp -- launch sequences of the BENCHMARK-only physics: nonlinear EOS, geopotential
// tracer mixing, KPP vertical mixing, COARE bulk fluxes, analytic atmospheric forcing.
#include "roms_host.h"
#include <cstdlib>
#include "k_bench.h"
#include "k_lmd.h"
#include <cmath>

static inline size_t lds_sz(const DGrid &G) { return (size_t)(G.bw + 6) * (size_t)(G.bh + 6); }

static inline KArgs mk(roms_hip_ctx *c) {
  KArgs a;
  a.G = c->G;
  a.Fv = c->F;
  a.p0 = a.p1 = a.p2 = 0;
  return a;
}

// Jerlov water type tables, mod_scalars.F:1110-1118
static const double k_mu1[9] = {0.35, 0.6, 1.0, 1.5, 1.4, 0.42, 0.37, 0.33, 0.00468592};
static const double k_mu2[9] = {23.0, 20.0, 17.0, 14.0, 7.9, 5.13, 3.54, 2.34, 1.51};
static const double k_r1[9] = {0.58, 0.62, 0.67, 0.77, 0.78, 0.57, 0.57, 0.57, 0.55};

int run_eos_nonlinear(roms_hip_ctx *c) {
  const int N = c->G.N;
  KArgs a = mk(c);
  // (a multi-tile context: the ghost columns with the tile -- t, z_r, z_w, Hz are valid there, every output is a function of its
  // own column -- and no exchange behind it: roms_host.h:ghost_tb)
  const bool gc = ghost_compute(c, 4);
  if (gc) a.G.T = ghost_tb(c, 3, c->G.Nghost);
  const TB B = a.G.T;
  const int nxe = B.IendT - B.IstrT + 1, nye = B.JendT - B.JstrT + 1;
  static const char *ept = getenv("ROMS_HIP_EOSPT");
  if (ept ? ept[0] != '0' : (long)nxe * nye <= 64L * 1024L) {       // few columns: chunks of five levels per thread + the column sums
    static const char *ekc = getenv("ROMS_HIP_EOSKC");
    a.p0 = ekc && atoi(ekc) > 0 ? atoi(ekc) : 5;
    LAUNCH_THREAD(k_eos_nl_pt, nxe, nye, (N + a.p0 - 1) / a.p0, c->stream, a);
    LAUNCH_THREAD(k_eos_sum, nxe, nye, 1, c->stream, a);
  } else LAUNCH_THREAD(k_eos_nl, nxe, nye, 1, c->stream, a);
  if (c->G.fuse3d || gc) return 0;   // the kernel stored the periodic images itself (pt_emit) | computed the ghost columns
  HaloSpec sp[7] = {{c->F.rho, N, BC_NONE, 'r'},  {c->F.pden, N, BC_NONE, 'r'}, {c->F.alpha, 1, BC_NONE, 'r'},
                    {c->F.beta, 1, BC_NONE, 'r'}, {c->F.rhoA, 1, BC_NONE, 'r'}, {c->F.rhoS, 1, BC_NONE, 'r'},
                    {c->F.bvf, N + 1, BC_NONE, 'r'}};
  launch_halo_tail(c, sp, 7);
  return 0;
}

// LMD_DDMIX: alfaobeta behind either equation of state (k_bench.h:k_eos_alfaobeta), on the columns the density kernel took
int run_eos_alfaobeta(roms_hip_ctx *c) {
  KArgs a = mk(c);
  if (ghost_compute(c, 4)) a.G.T = ghost_tb(c, 3, c->G.Nghost);
  const TB B = a.G.T;
  a.p1 = (c->G.options & ROMS_NONLIN_EOS) ? 1 : 0;
  LAUNCH_THREAD(k_eos_alfaobeta, B.IendT - B.IstrT + 1, B.JendT - B.JstrT + 1, c->G.N, c->stream, a);
  return 0;
}

int run_t3dmix2_geo(roms_hip_ctx *c) {
  const DGrid &G = c->G;
  KArgs a = mk(c);
  static const char *eg = getenv("ROMS_HIP_GEOCH");
  // levels per thread (every chunk re-reads the two levels below its first one): measured on 2048x256x30
  // 5: 438, 10: 382, 15: 363, 30: 348 us; on 512x64x30 5: 31, 10: 28, 15: 32, 30: 40 us
  const long cols = (long)(G.T.Iend - G.T.Istr + 1) * (G.T.Jend - G.T.Jstr + 1);
  a.p1 = eg ? atoi(eg) : (cols >= 128L * 1024L ? 15 : 10);
  if (a.p1 < 1) a.p1 = KCH;
  if (a.p1 > G.N) a.p1 = G.N;
  a.p0 = (G.N + a.p1 - 1) / a.p1;
  a.p2 = (c->tmix_terms && !(G.options & ROMS_MIX_ISO_TS)) ? 1 : 0;
  if (G.ts_dif4) {          // t3dmix4_geo.h: the rotated operator twice (first on the range widened by one point, into LapT = F.tmix)
    const TB &B = G.T;
    const int i0 = (G.ewp || !B.west) ? B.Istr - 1 : KMAX(B.Istr - 1, 1), i1 = (G.ewp || !B.east) ? B.Iend + 1 : KMIN(B.Iend + 1, G.Lm);
    const int j0 = (G.nsp || !B.south) ? B.Jstr - 1 : KMAX(B.Jstr - 1, 1), j1 = (G.nsp || !B.north) ? B.Jend + 1 : KMIN(B.Jend + 1, G.Mm);
    const bool iso = (G.options & ROMS_MIX_ISO_TS) != 0;          // t3dmix4_iso.h: the same, on the density slopes
    a.p2 = 2;
    if (iso) LAUNCH_THREAD(k_t3dmix2_iso, i1 - i0 + 1, j1 - j0 + 1, a.p0 * G.NT, c->stream, a);
    else LAUNCH_THREAD(k_t3dmix2_geo, i1 - i0 + 1, j1 - j0 + 1, a.p0 * G.NT, c->stream, a);
    a.p2 = 3;
    if (iso) LAUNCH_THREAD(k_t3dmix2_iso, B.Iend - B.Istr + 1, B.Jend - B.Jstr + 1, a.p0 * G.NT, c->stream, a);
    else LAUNCH_THREAD(k_t3dmix2_geo, B.Iend - B.Istr + 1, B.Jend - B.Jstr + 1, a.p0 * G.NT, c->stream, a);
    return 0;
  }
  if (G.options & ROMS_MIX_ISO_TS) LAUNCH_THREAD(k_t3dmix2_iso, G.T.Iend - G.T.Istr + 1, G.T.Jend - G.T.Jstr + 1, a.p0 * G.NT, c->stream, a);
  else LAUNCH_THREAD(k_t3dmix2_geo, G.T.Iend - G.T.Istr + 1, G.T.Jend - G.T.Jstr + 1, a.p0 * G.NT, c->stream, a);
  return 0;
}

int run_swdk(roms_hip_ctx *c) {
  const DGrid &G = c->G;
  const TB &B = G.T;
  const int J = c->cfg.lmd_Jwt;
  if (J < 1 || J > 9) { set_error("lmd_Jwt out of range"); return 5; }
  SwArgs a;
  a.G = G; a.Fv = c->F;
  a.fac1 = -1.0 / k_mu1[J - 1]; a.fac2 = -1.0 / k_mu2[J - 1]; a.fac3 = k_r1[J - 1];
  LAUNCH_THREAD(k_swdk, B.Iend - B.Istr + 1, B.Jend - B.Jstr + 1, G.N - 1, c->stream, a);
  return 0;
}

int run_lmd_vmix(roms_hip_ctx *c) {
  const DGrid &G = c->G;
  const TB &B = G.T;
  const int N = G.N, J = c->cfg.lmd_Jwt;
  if (J < 1 || J > 9) { set_error("lmd_Jwt out of range"); return 5; }
  LmdArgs a;
  a.G = G; a.Fv = c->F;
  a.fac1 = -1.0 / k_mu1[J - 1]; a.fac2 = -1.0 / k_mu2[J - 1]; a.fac3 = k_r1[J - 1];
  const double lmd_Cstar = 10.0, lmd_Cv = 1.25, lmd_Ric = 0.3, lmd_betaT = -0.2, lmd_cs = 98.96, lmd_epsilon = 0.1,
               vonKar = 0.41;
  a.lmd_Cg = lmd_Cstar * vonKar * pow(lmd_cs * vonKar * lmd_epsilon, 1.0 / 3.0);          // mod_scalars.F:2862
  a.Vtc = lmd_Cv * sqrt(-lmd_betaT) / (sqrt(lmd_cs * lmd_epsilon) * lmd_Ric * vonKar * vonKar);
  const int nx = B.Iend - B.Istr + 1, ny = B.Jend - B.Jstr + 1;
  static const char *elc = getenv("ROMS_HIP_LMDCOL");
  // ROMS_HIP_LMDCOL: 1 = one COL kernel with the three spline columns in LDS (47.6 KB per wave at N = 30) -- the default when
  // the chain waits for KPP; beside the barotropic loop (c->late_pre) its LDS would keep k_step2d's blocks off the CUs, and
  // columns taller than 41 levels would leave two waves per CU: 0 = the two kernels of rounds 1-3 (the default there);
  // 2 = the new column function as ONE THREAD kernel with its three columns in 3-D work arrays -- measured (round 4):
  // config 5 572 us against 389 + 187, BENCHMARK1 step 1.008 against 1.004 ms: no gain, the recurrences wait for their
  // own work-array round trips whichever way the passes are arranged; kept as a tested form, not the default
  const bool fits = (size_t)3 * (N + 1) * 64 * sizeof(double) < 64 * 1024;
  // 3 (round 5, the default where the chain waits for KPP): one block of 512 threads per 64 columns, k_lmd_blk -- BENCHMARK1
  // 91 -> 54 us, BENCHMARK3 1093 -> 842 us, config 5 (N = 50: two kernels, 396 + 186 us) -> 477 us
  const size_t blk_lds = (size_t)(4 * (N + 1) + LMD_BLK_NS) * 64 * sizeof(double);
  static const char *eblk = getenv("ROMS_HIP_LMDBLK");             // (0: the forms of round 4)
  const bool blk_ok = !(eblk && eblk[0] == '0') && (!c->late_pre || c->kpp_col_ok) && G.region == 0 && blk_lds <= 160 * 1024;
  int form = elc ? atoi(elc) : (blk_ok ? 3 : (((!c->late_pre || c->kpp_col_ok) && fits) ? 1 : 0));
  if (G.bkpp) form = 0;      // LMD_BKPP: the two THREAD kernels leave Akv, Akt without lmd_finish and the spline columns in the work arrays for k_lmd_bkpp
  if (form == 3 && G.region == 0 && blk_lds <= 160 * 1024) {        // the block form (k_lmd.h: k_lmd_blk): 64 columns per block of 512 threads
    const size_t ldsd = (size_t)(4 * (N + 1) + LMD_BLK_NS) * 64;
#ifndef ROMS_CPU_EMU
    static bool big_lds[64] = {};          // (the attribute is per device: a process may drive contexts on several)
    const int dev = c->cfg.device & 63;
    if (ldsd * sizeof(double) > 64 * 1024 && !big_lds[dev]) {
      if (hipFuncSetAttribute((const void *)k_lmd_blk, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
        set_error("k_lmd_blk: cannot raise the dynamic LDS limit"); return 2;
      }
      big_lds[dev] = true;
    }
#endif
    static const char *ebt = getenv("ROMS_HIP_LMDBT");
    const int nth = ebt ? atoi(ebt) : 512;
    LAUNCH_COOP(k_lmd_blk, (nx + 63) / 64, ny, 1, nth, ldsd, c->stream, a);
  } else if (form == 1 && fits) {
    LAUNCH_COL(k_lmd_col, nx, ny, 1, 3 * (N + 1), c->stream, a);
  } else if (form != 0) {
    LAUNCH_THREAD(k_lmd_fused, nx, ny, 1, c->stream, a);
  } else {
    LAUNCH_THREAD(k_lmd_interior, nx, ny, 1, c->stream, a);
    LAUNCH_THREAD(k_lmd_skpp, nx, ny, 1, c->stream, a);
  }
  if (G.bkpp) LAUNCH_THREAD(k_lmd_bkpp, nx, ny, 1, c->stream, a);    // lmd_bkpp_tile + lmd_finish (lmd_vmix.F:86-90)
  // (lmd_finish: fused into k_lmd_skpp's last sweep)
  if (G.fuse3d) return 0;   // k_lmd_skpp stored the boundary values and images (emit_store)
  if (G.bkpp) {
    HaloSpec sb[4] = {{c->F.hsbl, 1, BC_R, 'r'}, {c->F.hbbl, 1, BC_R, 'r'},        // bc_r2d_tile lmd_skpp.F:608, lmd_bkpp.F:577
                      {c->F.Akv, N + 1, BC_R, 'r'}, {c->F.Akt, (N + 1) * G.NAT, BC_R, 'r'}};
    launch_halo_tail(c, sb, 4);
    return 0;
  }
  HaloSpec sp[3] = {{c->F.hsbl, 1, BC_R, 'r'},                       // bc_r2d_tile lmd_skpp.F:608
                    {c->F.Akv, N + 1, BC_R, 'r'},                    // bc_w3d_tile lmd_vmix.F:740-760
                    {c->F.Akt, (N + 1) * G.NAT, BC_R, 'r'}};
  launch_halo_tail(c, sp, 3);
  return 0;
}

int run_bulk_flux(roms_hip_ctx *c) {
  const DGrid &G = c->G;
  const TB &B = G.T;
  BulkArgs a;
  a.G = G; a.Fv = c->F;
  a.ZW = c->cfg.blk_ZW; a.ZT = c->cfg.blk_ZT; a.ZQ = c->cfg.blk_ZQ;
  LAUNCH_THREAD(k_bulk_pt, B.IendR - (B.Istr - 1) + 1, B.JendR - (B.Jstr - 1) + 1, 1, c->stream, a);
  LAUNCH_THREAD(k_bulk_str, B.IendR - KMIN(B.Istr, B.IstrR) + 1, B.JendR - KMIN(B.Jstr, B.JstrR) + 1, 1, c->stream, a);
  if (G.fuse3d) return 0;   // the kernels stored the periodic images themselves (emit_store)
  HaloSpec sp[6] = {{c->F.lrflx, 1, BC_NONE, 'r'},  {c->F.lhflx, 1, BC_NONE, 'r'}, {c->F.shflx, 1, BC_NONE, 'r'},
                    {c->F.stflux, 1, BC_NONE, 'r'}, {c->F.sustr, 1, BC_NONE, 'u'}, {c->F.svstr, 1, BC_NONE, 'v'}};
  launch_halo_multi(c, sp, 6);
  return 0;
}

// ---- calendar of the reference for time_ref = 0 (caldate/datevec/datenum, dateclock.F; tolerant
//      ROUND, round.F): day of the year (fractional) and hour of the day of model time tdays.
static double k_ufloor(double X) { return X - fmod(X, 1.0) - fmod(2.0 + copysign(1.0, X), 3.0); }
static double k_tfloor(double X, double CT) {
  double Q = 1.0;
  if (X < 0.0) Q = 1.0 - CT;
  const double RMAX = Q / (2.0 - CT), EPS5 = CT / Q;
  const double Y = k_ufloor(X + fmax(CT, fmin(RMAX, EPS5 * fabs(1.0 + k_ufloor(X)))));
  if (X <= 0.0 || (Y - X) < RMAX) return Y;
  return Y - 1.0;
}
void roms_caldate(double tdays, double *yday, double *hour) {
  const double DateNumber = 367.0 + tdays;            // datenum(0001-01-01) = 367
  const double DayFraction = fabs(DateNumber - trunc(DateNumber));
  double D = DateNumber;
  D = (D < 61.0) ? D - 61.0 + 1.0 : D - 61.0;
  int yr = (int)((10000.0 * trunc(D) + 14780.0) / 3652425.0);
  int dy = (int)D - ((int)(365.0 * yr) + (int)(0.25 * yr) - (int)(0.01 * yr) + (int)(0.0025 * yr));
  if (dy < 0) {
    yr -= 1;
    dy = (int)D - ((int)(365.0 * yr) + (int)(0.25 * yr) - (int)(0.01 * yr) + (int)(0.0025 * yr));
  }
  const int mo = (int)((100.0 * dy + 52.0) / 3060.0);
  const int month = (mo + 2) % 12 + 1;
  const int year = yr + (int)((mo + 2.0) / 12.0);
  const int day = dy - (int)(0.1 * (mo * 306.0 + 5.0)) + 1;
  double seconds = DayFraction * 86400.0;
  seconds = k_tfloor(seconds + 0.5, 3.0 * 2.220446049250313e-16);
  *hour = seconds / 3600.0;
  const int fac = (((year % 4 == 0) && (year % 100 != 0)) || (year % 400 == 0)) ? 1 : 2;
  const int yd = (int)((275.0f * (float)month) / 9.0f) - fac * ((month + 9) / 12) + day - 30;
  *yday = (double)yd + DayFraction;
}

int run_set_data_benchmark(roms_hip_ctx *c) {
  const DGrid &G = c->G;
  const TB &B = G.T;
  const double pi = 3.14159265358979323846, deg2rad = pi / 180.0;
  SetDataBmArgs a;
  a.G = G; a.Fv = c->F;
  double yday, hour;
  roms_caldate(G.tdays, &yday, &hour);
  double Dangle = 23.44 * cos((172.0 - yday) * 2.0 * pi / 365.2425);   // ana_srflux.h:215-220
  a.Dangle = Dangle * deg2rad;
  a.Hangle = (12.0 - hour) * pi / 12.0;
  if (ghost_compute(c, 2)) {     // analytic fields of the position: the ghost points with the tile, no exchange (ana_*.h: exchange_r2d_tile)
    const TB X = ghost_tb(c, 3, G.Nghost);
    a.G.T = X;
    LAUNCH_THREAD(k_set_data_bm, X.IendT - X.IstrT + 1, X.JendT - X.JstrT + 1, 1, c->stream, a);
    return 0;
  }
  LAUNCH_THREAD(k_set_data_bm, B.IendT - B.IstrT + 1, B.JendT - B.JstrT + 1, 1, c->stream, a);
  if (G.fuse3d) return 0;   // the kernel stored the periodic images itself (emit_store)
  HaloSpec s1[8] = {{c->F.cloud, 1, BC_NONE, 'r'}, {c->F.Tair, 1, BC_NONE, 'r'},  {c->F.Hair, 1, BC_NONE, 'r'},
                    {c->F.srflx, 1, BC_NONE, 'r'}, {c->F.Uwind, 1, BC_NONE, 'r'}, {c->F.Vwind, 1, BC_NONE, 'r'},
                    {c->F.rain, 1, BC_NONE, 'r'},  {c->F.Pair, 1, BC_NONE, 'r'}};
  launch_halo_multi(c, s1, 8);
  launch_halo(c, c->F.stflux + G.nij, 1, BC_NONE, 'r');
  return 0;
}
