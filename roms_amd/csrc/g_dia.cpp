// g_dia.cpp -- DIAGNOSTICS_TS: set_diags(ng,tile), ROMS/Utility/set_diags.F (main3d.F:559), and the rate term of
// step3d_t.F:1892-1904; kernels in k_dia.h.
#include "roms_host.h"
#include <cstring>
#include "k_dia.h"

static DiaArgs mk(roms_hip_ctx *c) {
  DiaArgs a;
  a.G = c->G; a.Fv = c->F; a.init = 0; a.fac = 0.0; a.kout = c->G.kstp;
  return a;
}

int run_dia_rate(roms_hip_ctx *c) {
  const DGrid &G = c->G;
  if (!G.dia_ts) return 0;
  const TB &B = G.T;
  DiaArgs a = mk(c);
  LAUNCH_THREAD(k_dia_rate, B.IendR - B.IstrR + 1, B.JendR - B.JstrR + 1, G.N * G.NT, c->stream, a);
  return 0;
}

// the phases of set_diags_tile for step iic (set_diags.F:121-124, :244, :369-372)
int run_set_diags(roms_hip_ctx *c) {
  const DGrid &G = c->G;
  const int nDIA = c->dia_nDIA, ntsDIA = c->dia_ntsDIA, iic = c->s.iic;
  if (!G.dia_ts || nDIA <= 0) return 0;
  const TB &B = G.T;
  const bool init = (iic > ntsDIA && (iic - 1) % nDIA == 1) || (iic >= ntsDIA && nDIA == 1) || (c->dia_nrrec > 0 && iic == c->dia_ntstart);
  const bool accum = !init && iic > ntsDIA;
  const bool convert = (iic > ntsDIA && (iic - 1) % nDIA == 0 && (iic != c->dia_ntstart || c->dia_nrrec == 0)) || (iic >= ntsDIA && nDIA == 1);
  const int np = G.N * G.NT * G.dia_ts;
  DiaArgs a = mk(c);
  halo_fence(c, FG_2D | FG_T);
  if (init || accum) { int r = run_set_diags_uv(c, init ? 1 : 0, 0.0); if (r) return r; }
  if (init || accum) {
    a.init = init ? 1 : 0;
    LAUNCH_THREAD(k_dia_acc, B.IendR - B.IstrR + 1, B.JendR - B.JstrR + 1, np + 1, c->stream, a);
  }
  if (convert) {
    c->dia_time = nDIA == 1 ? c->s.time : c->dia_time + (double)nDIA * c->cfg.dt;     // DIAtime :379-383
    a.fac = 1.0 / (double)nDIA;
    { int r = run_set_diags_uv(c, 2, a.fac); if (r) return r; }
    LAUNCH_THREAD(k_dia_scale, B.IendR - B.IstrR + 1, B.JendR - B.JstrR + 1, np + 1, c->stream, a);
    // "periodic or gradient boundary conditions for output purposes" :576-615: exchange of avgzeta, bc_r3d_tile of every term
    HaloSpec z = {c->F.dia_zeta, 1, BC_NONE, 'r'};
    launch_halo_multi(c, &z, 1);
    for (int p0 = 0; p0 < np; p0 += G.N * G.NT) {
      HaloSpec sp = {(double *)c->F.DiaTrc + (size_t)p0 * G.nij, G.N * G.NT, BC_R, 'r'};
      launch_halo_multi(c, &sp, 1);
    }
  }
  return ctx_check(c, "set_diags");
}
