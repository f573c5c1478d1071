/*
** Custom application header: the cpp options of the UPWELLING test case with the analytic vertical
** mixing (ANA_VMIX) replaced by the Mellor-Yamada level 2.5 closure (MY25_MIXING) with Galperin's
** stability functions, K_C4ADVECTION, the plain shear and no smoothing.  TEST INFRASTRUCTURE: used by build_ref.sh through the reference makefile's
** MY_HEADER_DIR mechanism (makefile:235-236), with the application flag UPWELLING on the command
** line so that the reference's ana_*.h pick their UPWELLING branches.  (The shipped upwelling.h
** built with -DGLS_MIXING -- KANTHA_CLAYSON, N2S2_HORAVG, RI_SPLINES -- is the library "upwelling_gls".)
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
/* analytic grid, initial state and forcing */
#define ANA_GRID
#define ANA_INITIAL
#define ANA_SMFLUX
#define ANA_STFLUX
#define ANA_SSFLUX
#define ANA_BTFLUX
#define ANA_BSFLUX
/* vertical mixing: Mellor and Yamada (1982) level 2.5 closure, Galperin stability functions, centred fourth-order advection */
#define MY25_MIXING
#define K_C4ADVECTION
/* double-precision output, no averages/diagnostics (as PERFECT_RESTART does for upwelling.h) */
#define PERFECT_RESTART
#define OUT_DOUBLE
