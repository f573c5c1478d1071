// k_dia.h -- DIAGNOSTICS_TS: the time rate of change that closes a step's tracer terms (step3d_t.F:1892-1904) and
// set_diags_tile, ROMS/Utility/set_diags.F:60-735, for the tracer terms and the free surface: the set / accumulate phase
// and the conversion at the end of a window.  The terms themselves are stored by the kernels that compute them
// (k_pre_new, k_t3dmix2_*, k_s3t_hv, k_s3t_h, k_s3t_col: `if (G.dia_ts)`, roms_ctx.h:dia_wrk).
#pragma once
#include "roms_ctx.h"
#include "k_diag3d.h"

struct DiaArgs {
  DGrid G;
  Fields Fv;
  int init;        // set phase (first step of a window) instead of add
  double fac;      // 1/nDIA of the conversion
  int kout;        // time level of zeta (KOUT = kstp)
};

// DiaTwrk(iTrate) = t(nnew) - DiaTwrk(iTrate) on (IstrR:IendR, JstrR:JendR, N*NT) -- behind t3dbc and the land mask
THREAD_KERNEL(k_dia_rate, DiaArgs) {
  const DGrid &G = a.G;
  const Fields &F = a.Fv;
  const int N = G.N, k = gz % N + 1, itrc = gz / N + 1;
  const int i = G.T.IstrR + gx, j = G.T.JstrR + gy;
  double *D = dia_wrk(G, F, DIA_RATE, itrc) + X3(i, j, k);
  *D = F.t[XT(i, j, k, G.nnew, itrc)] - *D;
}
THREAD_GLOBAL(k_dia_rate, DiaArgs)

// set / add phase: grid.z = plane of DiaTwrk (N*NT*NDT of them), plus one plane for the free surface
THREAD_KERNEL(k_dia_acc, DiaArgs) {
  const DGrid &G = a.G;
  const Fields &F = a.Fv;
  const int i = G.T.IstrR + gx, j = G.T.JstrR + gy;
  const size_t x = X2(i, j);
  const int np = G.N * G.NT * G.dia_ts;
  if (gz == np) {
    const double z = F.zeta[x + (size_t)(a.kout - 1) * G.nij];
    double *A = (double *)F.dia_zeta + x;
    *A = a.init ? z : *A + z;
    return;
  }
  const size_t at = (size_t)gz * G.nij + x;
  const double w = ((const double *)F.DiaTwrk)[at];
  double *T = (double *)F.DiaTrc + at;
  *T = a.init ? w : *T + w;
}
THREAD_GLOBAL(k_dia_acc, DiaArgs)

THREAD_KERNEL(k_dia_scale, DiaArgs) {
  const DGrid &G = a.G;
  const Fields &F = a.Fv;
  const int i = G.T.IstrR + gx, j = G.T.JstrR + gy;
  const size_t x = X2(i, j);
  const int np = G.N * G.NT * G.dia_ts;
  double *A = gz == np ? (double *)F.dia_zeta + x : (double *)F.DiaTrc + (size_t)gz * G.nij + x;
  *A = a.fac * *A;
}
THREAD_GLOBAL(k_dia_scale, DiaArgs)
