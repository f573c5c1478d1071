// g_wetdry.cpp -- WET_DRY: launches of k_wetdry.h (ROMS/Nonlinear/wetdry.F; the call sites: step2d_LF_AM3.h:863 per
// barotropic call, initial.F:467) and of the post-scaling of ru, rv (prsgrd32.h:362,426, step3d_uv.F:721,1188).
#include "roms_host.h"
#include "k_wetdry.h"

// mode 0: a fast step (ahead of the barotropic kernel), 1: behind the last fast step (after the exchange of DU_avg1, DV_avg1),
// 2: the initial masks
int run_wetdry(roms_hip_ctx *c, int mode) {
  const DGrid &G = c->G;
  WdArgs a;
  a.G = G; a.Fv = c->F; a.mode = mode;
  a.init = (mode == 0 && G.predictor && G.iif == 1) ? 1 : 0;
  LAUNCH_THREAD(k_wetdry, G.ni, G.nj, 1, c->stream, a);
  return 0;
}

int run_wd_scale3(roms_hip_ctx *c) {
  const DGrid &G = c->G;
  const TB &B = G.T;
  KArgs a;
  a.G = G; a.Fv = c->F; a.p0 = 0; a.p1 = 0; a.p2 = 0;
  LAUNCH_THREAD(k_wd_scale3, B.Iend - B.Istr + 1, B.Jend - B.Jstr + 1, G.N, c->stream, a);
  return 0;
}

int run_wd_eff(roms_hip_ctx *c) {
  const DGrid &G = c->G;
  KArgs a;
  a.G = G; a.Fv = c->F; a.p0 = 0; a.p1 = 0; a.p2 = 0;
  LAUNCH_THREAD(k_wd_eff, G.ni, G.nj, 1, c->stream, a);
  return 0;
}
