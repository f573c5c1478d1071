/*
** Custom application header: the UPWELLING case with AVERAGES (as oracle/ref/upwelling_avg.h) and MASKING.
** TEST INFRASTRUCTURE: pins set_avg.F on a masked run (its 22 fields of roms_upwelling.in carry no mask arithmetic of
** their own; the masked state they accumulate does).
*/
#define MASKING
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
