/*
** (round 6 variant of kelvin_splines.h: the harmonic viscosity along geopotential surfaces, uv3dmix2_geo.h, beside open boundaries)
** Custom application header: the reference's KELVIN test case (ROMS/Include/kelvin.h: a Kelvin wave entering through
** the open western boundary of a flat channel -- Chapman / Flather conditions west, radiation east, RADIATION_2D) with the
** parabolic-spline vertical solvers SPLINES_VDIFF and SPLINES_VVISC that UPWELLING and BENCHMARK use (the plain
** tridiagonal forms of step3d_t.F:1507-1660 / step3d_uv.F:361-470 are not restated).  TEST INFRASTRUCTURE: used by
** build_ref.sh through the reference makefile's MY_HEADER_DIR mechanism (makefile:235-236) with the application flag
** KELVIN, to pin the open-boundary conditions in whole main3d passes.
*/
#define UV_ADV
#define UV_COR
#define UV_QDRAG
#define UV_VIS2
#define MIX_GEO_UV
#define SPLINES_VVISC
#define DJ_GRADPS
#define TS_DIF2
#define MIX_S_TS
#define SPLINES_VDIFF
#define SOLVE3D
#define RADIATION_2D
#define ANA_GRID
#define ANA_INITIAL
#define ANA_FSOBC
#define ANA_M2OBC
#define ANA_SMFLUX
#define ANA_STFLUX
#define ANA_SRFLUX
#define ANA_BTFLUX
