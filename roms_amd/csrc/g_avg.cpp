// g_avg.cpp -- set_avg(ng,tile), ROMS/Nonlinear/set_avg.F:51 (main3d.F:562): which phase a step is in, the two
// kernels of k_avg.h, the periodic refill / tile exchange of the converted averages.
#include "roms_host.h"
#include <cstdlib>
#include <cstring>
#include "k_avg.h"

static const char *const avg_names[AV_NFIELDS] = {
    "avg_zeta", "avg_ubar", "avg_vbar", "avg_u", "avg_v", "avg_omega", "avg_w", "avg_rho", "avg_t", "avg_ZZ", "avg_U2",
    "avg_V2", "avg_UU", "avg_VV", "avg_UV", "avg_Huon", "avg_Hvom", "avg_TT", "avg_UT", "avg_VT", "avg_HuonT", "avg_HvomT"};

int avg_field_index(const char *name) {
  for (int f = 0; f < AV_NFIELDS; f++)
    if (!strcmp(name, avg_names[f])) return f;
  return -1;
}
long avg_field_elems(const roms_hip_ctx *c, int f) { return (long)avg_planes(f, c->G.N, c->G.NT) * (long)c->G.nij; }

static char avg_grid(int f) {
  const int r = avg_range(f);
  return (r == 1 || r == 4) ? 'u' : (r == 2 || r == 5) ? 'v' : 'r';
}

// 2-D part first (it reads zeta, ubar, vbar of the output time level, which the barotropic loop overwrites):
// `part` 1 = chunk 0 only (2-D fields, interface 0), 2 = the remaining chunks, 3 = everything
static int launch_acc(roms_hip_ctx *c, int init, int part) {
  const DGrid &G = c->G;
  const TB &B = G.T;
  AvgArgs a;
  a.G = G;
  a.Fv = c->F;
  for (int f = 0; f < AV_NFIELDS; f++) a.A.a[f] = c->avg[f];
  a.mask = c->avg_mask;
  a.init = init;
  a.fac = 0.0;
  for (int m = 0; m < 3; m++) a.cnt[m] = c->avg_cnt[m];
  const int nch = (G.N + KCH - 1) / KCH;
  a.gz0 = part == 2 ? 1 : 0;
  const int nz = part == 1 ? 1 : part == 2 ? nch - 1 : nch;
  if (nz > 0) LAUNCH_THREAD(k_avg_acc, B.IendR - B.IstrR + 1, B.JendR - B.JstrR + 1, nz, c->stream, a);
  return 0;
}

static int launch_scale(roms_hip_ctx *c) {
  const DGrid &G = c->G;
  const TB &B = G.T;
  AvgArgs a;
  a.G = G;
  a.Fv = c->F;
  for (int f = 0; f < AV_NFIELDS; f++) a.A.a[f] = c->avg[f];
  a.mask = c->avg_mask;
  a.init = 0;
  a.gz0 = 0;
  a.fac = 1.0 / (double)c->avg_nAVG;
  for (int m = 0; m < 3; m++) a.cnt[m] = c->avg_cnt[m];
  int nchunks = 0;
  for (int f = 0; f < AV_NFIELDS; f++) nchunks += (avg_planes(f, G.N, G.NT) + KCH - 1) / KCH;
  LAUNCH_THREAD(k_avg_scale, B.IendR - B.IstrR + 1, B.JendR - B.JstrR + 1, nchunks, c->stream, a);
  // exchange_{r,u,v}{2,3}d_tile + mp_exchange of set_avg.F:3011-5200
  if (G.ewp || G.nsp || c->has_exchange) {
    HaloSpec sp[8];
    int n = 0;
    for (int f = 0; f < AV_NFIELDS; f++) {
      if (!((c->avg_mask >> f) & 1u)) continue;
      sp[n++] = {c->avg[f], avg_planes(f, G.N, G.NT), BC_NONE, avg_grid(f)};
      if (n == 8) { launch_halo_multi(c, sp, n); n = 0; }
    }
    if (n) launch_halo_multi(c, sp, n);
  }
  return 0;
}

// the phases of set_avg_tile for step iic (set_avg.F:251-254, :1606, :2962-2965).  part 0: the whole routine;
// 1: chunk 0 of the set/add phase only (the 2-D fields -- zeta, ubar, vbar of the output time level, which the
// barotropic loop overwrites -- and the lowest levels); 2: the other chunks and the conversion (main3d_late runs
// these beside the barotropic loop: their inputs are not touched before step3d_uv)
int run_set_avg(roms_hip_ctx *c, int part) {
  const int nAVG = c->avg_nAVG, ntsAVG = c->avg_ntsAVG, iic = c->s.iic;
  if (nAVG <= 0) return 0;                                                                  // :204
  const bool init = (iic > ntsAVG && (iic - 1) % nAVG == 1) || (iic >= ntsAVG && nAVG == 1) ||
                    (c->avg_nrrec > 0 && iic == c->avg_ntstart);
  const bool accum = !init && iic > ntsAVG;
  const bool convert = (iic > ntsAVG && (iic - 1) % nAVG == 0 && (iic != c->avg_ntstart || c->avg_nrrec == 0)) ||
                       (iic >= ntsAVG && nAVG == 1);
  int r = 0;
  if (init || accum) r = launch_acc(c, init ? 1 : 0, part);
  if (!r && convert && part != 1) {
    c->avg_time = nAVG == 1 ? c->s.time : c->avg_time + (double)nAVG * c->cfg.dt;           // :2966-2972
    r = launch_scale(c);
  }
  return r ? r : ctx_check(c, "set_avg");
}
