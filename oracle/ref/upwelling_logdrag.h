/*
** Custom application header: the UPWELLING test case with the logarithmic bottom drag (UV_LOGDRAG) in place of
** the linear one -- the form most realistic ROMS applications use.  TEST INFRASTRUCTURE: used by build_ref.sh
** through the reference makefile's MY_HEADER_DIR mechanism (makefile:235-236) with the application flag
** UPWELLING, to pin the UV_LOGDRAG branch of oracle/orc_diag3d.c:orc_set_vbc against set_vbc.F:591-635.
*/
/* momentum */
#define UV_ADV
#define UV_COR
#define UV_LOGDRAG
#define UV_VIS2
#define MIX_S_UV
#define SPLINES_VVISC
#define DJ_GRADPS
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
