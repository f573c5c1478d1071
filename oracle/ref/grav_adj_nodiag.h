/*
** Custom application header: the reference's GRAV_ADJ test case (ROMS/Include/grav_adj.h, the lock exchange) without its
** output options AVERAGES, DIAGNOSTICS_TS and DIAGNOSTICS_UV (accumulators for the averages / diagnostics files; the
** state they are computed from is the same).  TEST INFRASTRUCTURE: used by build_ref.sh (makefile:235-236 mechanism)
** with the application flag GRAV_ADJ.
*/
#define UV_ADV
#define UV_VIS2
#define UV_LDRAG
#define MIX_S_UV
#define DJ_GRADPS
#define SPLINES_VDIFF
#define SPLINES_VVISC
#define TS_DIF2
#define MIX_S_TS
#define SOLVE3D
#define ANA_GRID
#define ANA_INITIAL
#define ANA_SMFLUX
#define ANA_STFLUX
#define ANA_BTFLUX
#define OUT_DOUBLE
