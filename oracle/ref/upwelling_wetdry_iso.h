/*
** (round 6 variant of upwelling_wetdry.h: harmonic tracer mixing along isopycnic surfaces, t3dmix2_iso.h with its WET_DRY statements)
** Custom application header: the UPWELLING test case with land/sea masking and wetting and drying (MASKING + WET_DRY: wetdry.F, the WET_DRY branches of step2d_LF_AM3.h, rhs3d.F, prsgrd32.h, t3dmix2_s.h, uv3dmix2_s.h, step3d_uv.F, set_vbc.F): an island and a headland
** set by the test through the glue (the reference reads masks from its grid file).  TEST INFRASTRUCTURE: used by build_ref.sh
** through the reference makefile's MY_HEADER_DIR mechanism (makefile:235-236) with the application flag
** UPWELLING, to pin every MASKING branch of the oracle against the reference routines built with -DMASKING.
*/
/* land/sea masking */
#define MASKING
#define WET_DRY
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
#define MIX_ISO_TS
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
