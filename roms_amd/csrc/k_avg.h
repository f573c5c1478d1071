// k_avg.h -- time-averaged fields: set_avg_tile, ROMS/Nonlinear/set_avg.F:96-5210, for the fields the Aout switches
// of ROMS/External/roms_upwelling.in ask for: zeta, ubar, vbar, u, v, omega (W*pm*pn), w, rho, the tracers, Huon,
// Hvom and the quadratic terms zeta2, ubar2, vbar2, uu, vv, uv, <t*t>, <u*t>, <v*t>, <Huon*t>, <Hvom*t>.
//
//   k_avg_acc    the SET phase of the first step of a window (:251-1601) and the ADD phase of the following ones
//                (:1606-2954): one thread per (i,j) of (IstrR:IendR, JstrR:JendR) and chunk of KCH levels; the
//                thread of chunk 0 also does the 2-D fields and interface 0 of the w-type ones.  Every field keeps
//                the index range set_avg.F gives it (u-type from Istr, v-type from Jstr, the products that
//                average two neighbours on the interior only).  HBM-bound by construction: 22 arrays are
//                read-modify-written, the state arrays they are formed from are read once.
//   k_avg_scale  the step that closes a window (:2962-5210): sums times 1/nAVG on the same ranges; the periodic
//                ghost points are refilled by the halo launcher, as the exchange_*_tile calls there do.
// Time levels: KOUT = kstp, NOUT = nrhs (globaldefs.h:500-516).  a.mask: bit f set = field f is averaged.
#pragma once
#include "roms_ctx.h"
#include "k_diag3d.h"

enum { AV_ZETA = 0, AV_UBAR, AV_VBAR, AV_U, AV_V, AV_OMEGA, AV_W, AV_RHO, AV_T, AV_ZZ, AV_U2, AV_V2, AV_UU, AV_VV,
       AV_UV, AV_HUON, AV_HVOM, AV_TT, AV_UT, AV_VT, AV_HUT, AV_HVT, AV_NFIELDS };

struct AvgFields { double *a[AV_NFIELDS]; };
struct AvgArgs {
  DGrid G;
  Fields Fv;
  AvgFields A;
  unsigned mask;
  int init;          // 1: set, 0: add
  int gz0;           // k_avg_acc: first chunk of this launch
  double fac;        // k_avg_scale: 1/nAVG
  double *cnt[3];    // WET_DRY (round 6): the wet-point counters rmask_avg, umask_avg, vmask_avg of set_avg.F:257-288, :1608-1645
};

#define AV_ON(f) ((a.mask >> (f)) & 1u)
// WET_DRY: the value times the full mask (land x wet) of the field's grid type, set_avg.F:302 ... :1652 ... (mr, mu, mv below; 1 otherwise)
#define AV_PUTM(f, off, val, m_)                              \
  do {                                                        \
    double *d_ = a.A.a[f] + (off);                            \
    double v_ = (val);                                        \
    if (wet) v_ = v_ * (m_);                                  \
    *d_ = a.init ? v_ : *d_ + v_;                             \
  } while (0)
#define AV_PUT(f, off, val) AV_PUTM(f, off, val, mr)
#define AV_PUTU(f, off, val) AV_PUTM(f, off, val, mu)
#define AV_PUTV(f, off, val) AV_PUTM(f, off, val, mv)

THREAD_KERNEL(k_avg_acc, AvgArgs) {
  gz += a.gz0;
  const DGrid &G = a.G;
  const Fields &F = a.Fv;
  const TB &B = G.T;
  const int i = B.IstrR + gx, j = B.JstrR + gy, N = G.N, NT = G.NT;
  const int Kout = G.kstp, Nout = G.nrhs;
  const size_t nij = (size_t)G.nij, x = X2(i, j);
  const long ni = G.ni;
  const bool uu = i >= B.Istr, vv = j >= B.Jstr;                       // u-type / v-type ranges
  const bool ii = i >= B.Istr && i <= B.Iend, jj = j >= B.Jstr && j <= B.Jend;
  const double *u = F.u + (size_t)(Nout - 1) * nij * (size_t)N + x, *v = F.v + (size_t)(Nout - 1) * nij * (size_t)N + x;
  const bool wet = G.wet_dry != 0;
  const double mr = wet ? F.rmask_full[x] : 1.0, mu = wet ? F.umask_full[x] : 1.0, mv = wet ? F.vmask_full[x] : 1.0;
  if (gz == 0 && wet) {                                                // the wet-point counters :257-288 | :1608-1645
    const double cr = KMAX(0.0, KMIN(mr, 1.0)), cu = KMAX(0.0, KMIN(mu, 1.0)), cv = KMAX(0.0, KMIN(mv, 1.0));
    a.cnt[0][x] = a.init ? cr : a.cnt[0][x] + cr;
    if (uu) a.cnt[1][x] = a.init ? cu : a.cnt[1][x] + cu;
    if (vv) a.cnt[2][x] = a.init ? cv : a.cnt[2][x] + cv;
  }
  if (gz == 0) {
    const double z = F.zeta[X2T(i, j, Kout)];
    if (AV_ON(AV_ZETA)) AV_PUT(AV_ZETA, x, z);
    if (AV_ON(AV_ZZ)) AV_PUT(AV_ZZ, x, z * z);
    if (uu) {
      const double ub = F.ubar[X2T(i, j, Kout)];
      if (AV_ON(AV_UBAR)) AV_PUTU(AV_UBAR, x, ub);
      if (AV_ON(AV_U2)) AV_PUTU(AV_U2, x, ub * ub);
    }
    if (vv) {
      const double vb = F.vbar[X2T(i, j, Kout)];
      if (AV_ON(AV_VBAR)) AV_PUTV(AV_VBAR, x, vb);
      if (AV_ON(AV_V2)) AV_PUTV(AV_V2, x, vb * vb);
    }
    if (AV_ON(AV_OMEGA)) AV_PUT(AV_OMEGA, x, F.W[x] * F.pm[x] * F.pn[x]);         // interface 0
    if (AV_ON(AV_W)) AV_PUT(AV_W, x, F.wvel[x]);
  }
  const double pm = F.pm[x], pn = F.pn[x];
#pragma unroll
  for (int q = 0; q < KCH; q++) {
    const int k = gz * KCH + 1 + q;
    if (k > N) break;
    const size_t o3 = (size_t)(k - 1) * nij, ow = (size_t)k * nij;
    if (AV_ON(AV_OMEGA)) AV_PUT(AV_OMEGA, ow + x, F.W[ow + x] * pm * pn);
    if (AV_ON(AV_W)) AV_PUT(AV_W, ow + x, F.wvel[ow + x]);
    if (AV_ON(AV_RHO)) AV_PUT(AV_RHO, o3 + x, F.rho[o3 + x]);
    const double uk = uu ? u[o3] : 0.0, vk = vv ? v[o3] : 0.0;
    if (uu) {
      if (AV_ON(AV_U)) AV_PUTU(AV_U, o3 + x, uk);
      if (AV_ON(AV_UU)) AV_PUTU(AV_UU, o3 + x, uk * uk);
      if (AV_ON(AV_HUON)) AV_PUTU(AV_HUON, o3 + x, F.Huon[o3 + x]);
    }
    if (vv) {
      if (AV_ON(AV_V)) AV_PUTV(AV_V, o3 + x, vk);
      if (AV_ON(AV_VV)) AV_PUTV(AV_VV, o3 + x, vk * vk);
      if (AV_ON(AV_HVOM)) AV_PUTV(AV_HVOM, o3 + x, F.Hvom[o3 + x]);
    }
    if (ii && jj && AV_ON(AV_UV)) AV_PUT(AV_UV, o3 + x, 0.25 * (uk + u[o3 + 1]) * (vk + v[o3 + ni]));
    for (int it = 1; it <= NT; it++) {
      const double *t = F.t + XT(G.LBi, G.LBj, 1, Nout, it) + x + o3;
      const size_t ot = (size_t)(it - 1) * nij * (size_t)N + o3 + x;
      const double tk = t[0];
      if (AV_ON(AV_T)) AV_PUT(AV_T, ot, tk);
      if (AV_ON(AV_TT)) AV_PUT(AV_TT, ot, tk * tk);
      if (ii) {
        const double ts = t[-1] + tk;
        if (AV_ON(AV_UT)) AV_PUTU(AV_UT, ot, 0.5 * uk * ts);
        if (AV_ON(AV_HUT)) AV_PUTU(AV_HUT, ot, 0.5 * F.Huon[o3 + x] * ts);
      }
      if (jj) {
        const double ts = t[-ni] + tk;
        if (AV_ON(AV_VT)) AV_PUTV(AV_VT, ot, 0.5 * vk * ts);
        if (AV_ON(AV_HVT)) AV_PUTV(AV_HVT, ot, 0.5 * F.Hvom[o3 + x] * ts);
      }
    }
  }
}
THREAD_GLOBAL(k_avg_acc, AvgArgs)

// planes of field f: 1 (2-D), N (rho levels), N+1 (w levels), times NT for the tracer terms
KHD int avg_planes(int f, int N, int NT) {
  switch (f) {
    case AV_ZETA: case AV_UBAR: case AV_VBAR: case AV_ZZ: case AV_U2: case AV_V2: return 1;
    case AV_OMEGA: case AV_W: return N + 1;
    case AV_T: case AV_TT: case AV_UT: case AV_VT: case AV_HUT: case AV_HVT: return N * NT;
    default: return N;
  }
}
// range class of field f: 0 (IstrR:IendR,JstrR:JendR) 1 (Istr:IendR,JstrR:JendR) 2 (IstrR:IendR,Jstr:JendR)
// 3 (Istr:Iend,Jstr:Jend) 4 (Istr:Iend,JstrR:JendR) 5 (IstrR:IendR,Jstr:Jend)
KHD int avg_range(int f) {
  switch (f) {
    case AV_UBAR: case AV_U2: case AV_U: case AV_UU: case AV_HUON: return 1;
    case AV_VBAR: case AV_V2: case AV_V: case AV_VV: case AV_HVOM: return 2;
    case AV_UV: return 3;
    case AV_UT: case AV_HUT: return 4;
    case AV_VT: case AV_HVT: return 5;
    default: return 0;
  }
}

// one thread per (i,j) of (IstrR:IendR, JstrR:JendR) and plane chunk of KCH; gz = field * chunks + chunk
THREAD_KERNEL(k_avg_scale, AvgArgs) {
  const DGrid &G = a.G;
  const TB &B = G.T;
  const int N = G.N, NT = G.NT;
  const int i = B.IstrR + gx, j = B.JstrR + gy;
  // fields are laid out over gz by their plane counts in chunks of KCH (the launcher uses the same walk)
  int f = 0, c0 = gz;
  for (; f < AV_NFIELDS; f++) {
    const int nc = (avg_planes(f, N, NT) + KCH - 1) / KCH;
    if (c0 < nc) break;
    c0 -= nc;
  }
  if (f >= AV_NFIELDS || !AV_ON(f)) return;
  const int r = avg_range(f);
  const int i0 = (r == 1 || r == 3 || r == 4) ? B.Istr : B.IstrR, i1 = (r == 3 || r == 4) ? B.Iend : B.IendR;
  const int j0 = (r == 2 || r == 3 || r == 5) ? B.Jstr : B.JstrR, j1 = (r == 3 || r == 5) ? B.Jend : B.JendR;
  if (i < i0 || i > i1 || j < j0 || j > j1) return;
  const int np = avg_planes(f, N, NT);
  double *d = a.A.a[f] + X2(i, j);
  // WET_DRY: the sums divided by the number of steps the point was wet, :2980-2988 (else 1/nAVG)
  const double fac = G.wet_dry ? 1.0 / KMAX(1.0, a.cnt[(r == 1 || r == 4) ? 1 : (r == 2 || r == 5) ? 2 : 0][X2(i, j)]) : a.fac;
#pragma unroll
  for (int q = 0; q < KCH; q++) {
    const int p = c0 * KCH + q;
    if (p >= np) break;
    d[(size_t)p * (size_t)G.nij] = fac * d[(size_t)p * (size_t)G.nij];
  }
}
THREAD_GLOBAL(k_avg_scale, AvgArgs)
