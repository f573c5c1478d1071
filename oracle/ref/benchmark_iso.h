/*
** (round 6 variant of benchmark_mask.h without MASKING: tracer mixing along ISOPYCNIC surfaces with the nonlinear equation of state, t3dmix2_iso.h)
** Custom application header: the BENCHMARK test case (its cpp options as SURVEY.md Appendix A lists them:
** quadratic drag, geopotential tracer mixing, curvilinear spherical grid, nonlinear equation of state, KPP,
** COARE bulk fluxes with the analytic atmosphere) with land/sea masking (MASKING) added.  TEST INFRASTRUCTURE:
** used by build_ref.sh through the reference makefile's MY_HEADER_DIR mechanism (makefile:235-236) with the
** application flag BENCHMARK, to pin the MASKING branches of rho_eos.F (nonlinear), lmd_skpp.F, bulk_flux.F
** and t3dmix2_geo.h.  The masks are data set by the test (tests/refdrive.py), see functionals/ana_mask.h.
*/
/* land/sea masking */

/* momentum */
#define UV_ADV
#define UV_COR
#define UV_QDRAG
#define UV_VIS2
#define MIX_S_UV
#define SPLINES_VVISC
#define DJ_GRADPS
/* tracers */
#define SOLVE3D
#define SALINITY
#define NONLIN_EOS
#define TS_DIF2
#define MIX_ISO_TS
#define SPLINES_VDIFF
#define SOLAR_SOURCE
/* grid */
#define CURVGRID
#define SPHERICAL
#define ANA_GRID
#define ANA_INITIAL
/* vertical mixing: Large, McWilliams and Doney (1994) surface KPP */
#define LMD_MIXING
#define LMD_RIMIX
#define LMD_CONVEC
#define LMD_SKPP
#define LMD_NONLOCAL
#define RI_SPLINES
/* air-sea fluxes: COARE bulk formulae with the analytic atmosphere */
#define BULK_FLUXES
#define LONGWAVE
#define ALBEDO
#define ANA_WINDS
#define ANA_TAIR
#define ANA_PAIR
#define ANA_HUMIDITY
#define ANA_RAIN
#define ANA_CLOUD
#define ANA_SRFLUX
#define ANA_SSFLUX
#define ANA_BSFLUX
#define ANA_BTFLUX
