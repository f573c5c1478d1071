// g_stubs.cpp -- entry points whose kernels are not written yet return exit_flag 8 (loudly).
#include "roms_host.h"
#define STUB(name) int name(roms_hip_ctx *) { set_error(#name ": not implemented in this build"); return 8; }
#ifndef HAVE_RHS3D
STUB(run_pre_step3d) STUB(run_prsgrd) STUB(run_t3dmix2) STUB(run_uv3dmix2) STUB(run_rhs3d_tile)
#endif
#if 0
STUB(run_step2d)
#endif
#ifndef HAVE_STEP3D
STUB(run_step3d_uv) STUB(run_step3d_t)
#endif
#ifndef HAVE_LMD
STUB(run_lmd_vmix)
#endif
#ifndef HAVE_BULK
STUB(run_bulk_flux) STUB(run_set_data_benchmark)
#endif
#ifndef HAVE_EOS
STUB(run_eos_nonlinear)
#endif
#ifndef HAVE_DIAG
int run_diag(roms_hip_ctx *, double *) { set_error("run_diag: not implemented in this build"); return 8; }
#endif
