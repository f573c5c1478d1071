/*
** Custom application header: the UPWELLING test case as shipped WITH its time-averaged output (AVERAGES), without
** the per-term diagnostics (DIAGNOSTICS_TS/UV need mod_diags in every kernel and are not built here).  TEST
** INFRASTRUCTURE: used by build_ref.sh through the reference makefile's MY_HEADER_DIR mechanism (makefile:235-236)
** with the application flag UPWELLING, to pin oracle/orc_avg.c against the reference's set_avg.F.  The stock
** upwelling.h cannot be used for this: the PERFECT_RESTART build of the other pins undefines AVERAGES (upwelling.h:71-77).
*/
/* momentum */
#define UV_ADV
#define UV_COR
#define UV_LDRAG
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
/* time-averaged output */
#define AVERAGES
/* analytic grid, initial state, forcing and vertical mixing */
#define ANA_GRID
#define ANA_INITIAL
#define ANA_SMFLUX
#define ANA_STFLUX
#define ANA_SSFLUX
#define ANA_BTFLUX
#define ANA_BSFLUX
#define ANA_VMIX
#define OUT_DOUBLE
