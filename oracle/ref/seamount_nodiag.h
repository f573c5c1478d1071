/*
** Custom application header: the reference's SEAMOUNT test case (ROMS/Include/seamount.h) without ANA_DIAG -- the user
** diagnostics hook (Functionals/ana_diag.h writes a text file of its own; its OPEN statement uses an undeclared `io_err`
** and does not compile under IMPLICIT NONE with this compiler).  Nothing on the time-stepping path depends on it.
** TEST INFRASTRUCTURE: used by build_ref.sh (makefile:235-236 mechanism) with the application flag SEAMOUNT.
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
#define MIX_GEO_TS
#define SOLVE3D
#define ANA_GRID
#define ANA_INITIAL
#define ANA_SMFLUX
#define ANA_STFLUX
#define ANA_BTFLUX
