// g_stubs.cpp -- entry points whose kernels are not written yet return exit_flag 8 (loudly).
#include "roms_host.h"
#define STUB(name) int name(roms_hip_ctx *) { set_error(#name ": not implemented in this build"); return 8; }
#if 0
STUB(run_step2d)
#endif
#ifndef HAVE_MPDATA
STUB(run_step3d_t_mpdata)
#endif
#if 0
int run_diag(roms_hip_ctx *, double *) { set_error("run_diag: not implemented in this build"); return 8; }
#endif
