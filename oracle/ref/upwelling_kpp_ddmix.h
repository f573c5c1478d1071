/*
** Custom application header for BASELINE config 5 ("UPWELLING + LMD vertical mixing + MPDATA"):
** the cpp options of the UPWELLING test case with the analytic vertical mixing (ANA_VMIX)
** replaced by the LMD/KPP closure of the BENCHMARK application.  TEST INFRASTRUCTURE: used
** by build_ref.sh through the reference makefile's MY_HEADER_DIR mechanism
** (makefile:235-236: ROMS_HEADER="$(MY_HEADER_DIR)/$(HEADER)"), with the application flag
** UPWELLING on the command line so that the reference's ana_*.h pick their UPWELLING branches.
** MPDATA is a run-time choice (Hadvection/Vadvection in roms.in), not a cpp option.
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
#define ANA_SRFLUX
/* vertical mixing: Large, McWilliams and Doney (1994) surface KPP */
#define LMD_MIXING
#define LMD_RIMIX
#define LMD_CONVEC
#define LMD_SKPP
#define LMD_NONLOCAL
#define LMD_DDMIX
#define RI_SPLINES
#define SOLAR_SOURCE
/* double-precision output, no averages/diagnostics (as PERFECT_RESTART does for upwelling.h) */
#define PERFECT_RESTART
#define OUT_DOUBLE
