/*
** Custom application header: the reference's OVERFLOW test case (ROMS/Include/overflow.h: a dense-water overflow down a
** slope, isopycnic tracer mixing) without its output option AVERAGES (accumulators for the averages file; the state they
** are computed from is the same).  TEST INFRASTRUCTURE: used by build_ref.sh (makefile:235-236 mechanism) with the
** application flag OVERFLOW.
*/
#define UV_ADV
#define UV_COR
#define UV_QDRAG
#define UV_VIS2
#define MIX_S_UV
#define DJ_GRADPS
#define SPLINES_VDIFF
#define SPLINES_VVISC
#define TS_DIF2
#define MIX_ISO_TS
#define SOLVE3D
#define ANA_GRID
#define ANA_INITIAL
#define ANA_SMFLUX
#define ANA_STFLUX
#define ANA_BTFLUX
#define OUT_DOUBLE
