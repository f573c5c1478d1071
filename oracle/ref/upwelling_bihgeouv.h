/*
** Custom application header: the UPWELLING test case with BIHARMONIC viscosity along GEOPOTENTIAL surfaces (UV_VIS4 + MIX_GEO_UV:
** uv3dmix4_geo.h, the rotated stress tensor twice) and land/sea masking (MASKING): an island and a headland
** set by the test through the glue (the reference reads masks from its grid file).  TEST INFRASTRUCTURE: used by build_ref.sh
** through the reference makefile's MY_HEADER_DIR mechanism (makefile:235-236) with the application flag
** UPWELLING, to pin uv3dmix4_geo.h with its rho-, psi-, u- and v-mask statements.
*/
/* land/sea masking */
#define MASKING
/* momentum */
#define UV_ADV
#define UV_COR
#define UV_LDRAG
#define UV_VIS4
#define MIX_GEO_UV
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
