/*
** Custom application header: the UPWELLING test case with the finite-volume pressure Jacobian of Shchepetkin & McWilliams (2003) with quartic reconstruction of density, prsgrd44.h (PJ_GRADPQ4).
** TEST INFRASTRUCTURE: used by build_ref.sh (makefile:235-236 mechanism) with the application flag UPWELLING.
*/
/* momentum */
#define UV_ADV
#define UV_COR
#define UV_LDRAG
#define UV_VIS2
#define MIX_S_UV
#define SPLINES_VVISC
#define PJ_GRADPQ4
/* tracers */
#define SOLVE3D
#define SALINITY
#define TS_DIF2
#define MIX_S_TS
#define SPLINES_VDIFF
/* analytic grid, initial state, forcing and vertical mixing */
#define ANA_GRID
#define ANA_INITIAL
#define ANA_SMFLUX
#define ANA_STFLUX
#define ANA_SSFLUX
#define ANA_BTFLUX
#define ANA_BSFLUX
#define ANA_VMIX
/* double-precision output, no averages/diagnostics (as PERFECT_RESTART does for upwelling.h) */
#define PERFECT_RESTART
#define OUT_DOUBLE
