#!/bin/bash
# Build oracle/_ref/libromsref_<app>.so from the reference's own Fortran
# sources WHERE THEY LIE under /root/reference (nothing is copied into the
# repo; intermediates live in a temp dir and are deleted).
#
# TEST INFRASTRUCTURE.  Recipe mirrors the reference's makefile
# (makefile:219,230-238,397-423: cpp -P -traditional, then ROMS/Bin/cpp_clean,
# then the Fortran compiler) with amdflang in place of gfortran.
#
# Only reference code that compiles WITHOUT the NetCDF Fortran module is built
# (this image has no NetCDF; no stand-in is written for it, nothing is stubbed).
#
# mod_sources.F -- USEd by step2d, omega, pre_step3d, step3d_uv, step3d_t -- touches
# NetCDF in two places: allocate_sources reads the number of point sources from the
# river file unless the reference's own cpp option ANA_PSOURCE is defined, and
# check_sources (an input-file inquiry called only by the NetCDF reader get_data)
# USEs mod_netcdf unconditionally.  The recipe therefore pre-processes THAT ONE FILE
# with -DANA_PSOURCE and leaves check_sources out of the pre-processed text (sed,
# below) -- the same "leave out what needs NetCDF" rule the file list applies to
# whole files, at subroutine granularity.  Every statement that is compiled is the
# reference's own text; the BASELINE applications have no point sources
# (LuvSrc = LwSrc = LtracerSrc = .FALSE., mod_scalars.F:4512-4517), so neither
# routine is ever executed and the six kernels compile to exactly what a NetCDF
# build of the same application gives.  main3d.F itself (USEs the NetCDF readers
# and writers) stays out; ref_glue.F90 calls the reference kernels in its order.
#
# usage: build_ref.sh upwelling|benchmark|upwelling_kpp|...|kelvin|kelvin_splines
# (upwelling_kpp = the custom application header oracle/ref/upwelling_kpp.h of BASELINE config 5)
set -e
APP=${1:-upwelling}
REF=${ROMS_REF:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=$HERE/../_ref
if [ ! -d "$REF/ROMS" ]; then echo "build_ref: no reference tree at $REF -- skipped"; exit 0; fi
FC=${FC:-amdflang}
command -v $FC >/dev/null || { echo "build_ref: $FC not found -- skipped"; exit 0; }
UP=$(echo $APP | tr a-z A-Z)
HDR=$APP
EXTRA=""
[ "$APP" = upwelling ] && EXTRA="-DPERFECT_RESTART"
HDRPATH="$HDR.h"
if [ "$APP" = upwelling_kpp ]; then
  # custom application header (oracle/ref/upwelling_kpp.h) through the makefile's MY_HEADER_DIR
  # mechanism (makefile:235-236); application flag UPWELLING for the ana_*.h branches
  UP=UPWELLING; HDR=upwelling_kpp; HDRPATH="$HERE/upwelling_kpp.h"
  EXTRA=""
fi
if [ "$APP" = upwelling_kpp_ddmix ]; then
  # ... with double-diffusive mixing in the interior scheme (oracle/ref/upwelling_kpp_ddmix.h: LMD_DDMIX, lmd_vmix.F:360-428; linear EOS)
  UP=UPWELLING; HDR=upwelling_kpp_ddmix; HDRPATH="$HERE/upwelling_kpp_ddmix.h"
  EXTRA=""
fi
if [ "$APP" = benchmark_wetdry_ddmix ]; then
  # oracle/ref/benchmark_wetdry.h (MASKING + WET_DRY, bulk fluxes, KPP) with LMD_DDMIX on top
  UP=BENCHMARK; HDR=benchmark_wetdry; HDRPATH="$HERE/benchmark_wetdry.h"
  EXTRA="-I$HERE/functionals -DLMD_DDMIX"
fi
if [ "$APP" = benchmark_bkpp ]; then
  # the shipped benchmark.h with the bottom boundary layer of the K-profile scheme on top (LMD_BKPP: lmd_bkpp.F), round 6
  UP=BENCHMARK; HDR=benchmark; HDRPATH="benchmark.h"
  EXTRA="-DLMD_BKPP"
fi
if [ "$APP" = upwelling_kpp_bkpp ]; then
  # ... and on UPWELLING with KPP (oracle/ref/upwelling_kpp.h: linear EOS, no surface heat flux)
  UP=UPWELLING; HDR=upwelling_kpp; HDRPATH="$HERE/upwelling_kpp.h"
  EXTRA="-DLMD_BKPP"
fi
if [ "$APP" = benchmark_ddmix ]; then
  # the shipped benchmark.h with LMD_DDMIX switched on as a user does (nonlinear EOS: alfaobeta of rho_eos.F:435-455)
  UP=BENCHMARK; HDR=benchmark; HDRPATH="benchmark.h"
  EXTRA="-DLMD_DDMIX"
fi
if [ "$APP" = upwelling_logdrag ]; then
  # the UPWELLING case with UV_LOGDRAG (oracle/ref/upwelling_logdrag.h): pins the logarithmic bottom stress
  UP=UPWELLING; HDR=upwelling_logdrag; HDRPATH="$HERE/upwelling_logdrag.h"
  EXTRA=""
fi
if [ "$APP" = upwelling_noadv ]; then
  # the UPWELLING case with WINDBASIN's option set: no UV_ADV, no UV_VIS2 / TS_DIF2 (oracle/ref/upwelling_noadv.h)
  UP=UPWELLING; HDR=upwelling_noadv; HDRPATH="$HERE/upwelling_noadv.h"
  EXTRA=""
fi
if [ "$APP" = upwelling_mask ]; then
  # the UPWELLING case with MASKING (oracle/ref/upwelling_mask.h): pins the land/sea mask branches
  UP=UPWELLING; HDR=upwelling_mask; HDRPATH="$HERE/upwelling_mask.h"
  EXTRA="-I$HERE/functionals"     # the user analytical file ana_mask.h of this application
fi
if [ "$APP" = upwelling_bihgeouv ]; then
  # ... with the BIHARMONIC viscosity along geopotential surfaces (oracle/ref/upwelling_bihgeouv.h: UV_VIS4 + MIX_GEO_UV, uv3dmix4_geo.h)
  UP=UPWELLING; HDR=upwelling_bihgeouv; HDRPATH="$HERE/upwelling_bihgeouv.h"
  EXTRA="-I$HERE/functionals"
fi
if [ "$APP" = upwelling_geouv ]; then
  # UPWELLING + MASKING with the viscosity along geopotential surfaces (oracle/ref/upwelling_geouv.h: MIX_GEO_UV, uv3dmix2_geo.h)
  UP=UPWELLING; HDR=upwelling_geouv; HDRPATH="$HERE/upwelling_geouv.h"
  EXTRA="-I$HERE/functionals"
fi
if [ "$APP" = benchmark_mask ]; then
  # the BENCHMARK case with MASKING (oracle/ref/benchmark_mask.h): pins the masked KPP / bulk-flux / EOS / geopotential-mixing branches
  UP=BENCHMARK; HDR=benchmark_mask; HDRPATH="$HERE/benchmark_mask.h"
  EXTRA="-I$HERE/functionals"
fi
if [ "$APP" = benchmark_wetdry ]; then
  # ... with MASKING + WET_DRY (oracle/ref/benchmark_wetdry.h): the WET_DRY branches of bulk_flux.F, pre_step3d.F, t3dmix2_geo.h, mpdata_adiff.F
  UP=BENCHMARK; HDR=benchmark_wetdry; HDRPATH="$HERE/benchmark_wetdry.h"
  EXTRA="-I$HERE/functionals"
fi
if [ "$APP" = upwelling_wetdry_gls ] || [ "$APP" = upwelling_wetdry_my25 ] || [ "$APP" = upwelling_wetdry_geouv ] || [ "$APP" = upwelling_wetdry_prs31 ] || [ "$APP" = upwelling_wetdry_prs44 ] || [ "$APP" = upwelling_wetdry_iso ]; then
  # WET_DRY with the closures, the viscosity along geopotentials and the other pressure Jacobians (round 6: oracle/ref/upwelling_wetdry_*.h;
  # PJ_GRADP does not compile with WET_DRY in the reference itself: prsgrd40.h:98 passes umask_wet, vmask_wet without declaring them)
  UP=UPWELLING; HDR=$APP; HDRPATH="$HERE/$APP.h"
  EXTRA="-I$HERE/functionals"
fi
if [ "$APP" = upwelling_wetdry_avg ]; then
  # MASKING + WET_DRY + AVERAGES (oracle/ref/upwelling_wetdry_avg.h): the wet masks and wet-point counters of set_avg.F
  UP=UPWELLING; HDR=upwelling_wetdry_avg; HDRPATH="$HERE/upwelling_wetdry_avg.h"
  EXTRA="-I$HERE/functionals"
fi
if [ "$APP" = upwelling_avg_mask ]; then
  # AVERAGES + MASKING (oracle/ref/upwelling_avg_mask.h)
  UP=UPWELLING; HDR=upwelling_avg_mask; HDRPATH="$HERE/upwelling_avg_mask.h"
  EXTRA="-I$HERE/functionals"
fi
if [ "$APP" = upwelling_prs31 ] || [ "$APP" = upwelling_wjgradp ] || [ "$APP" = upwelling_prs40 ] || [ "$APP" = upwelling_prs42 ] || [ "$APP" = upwelling_prs44 ]; then
  # UPWELLING with the standard density Jacobian prsgrd31.h (no DJ_GRADPS; _wjgradp: WJ_GRADP, its weighted form); _prs40: PJ_GRADP, prsgrd40.h;
  # _prs42: PJ_GRADPQ2, prsgrd42.h; _prs44: PJ_GRADPQ4, prsgrd44.h
  UP=UPWELLING; HDR=$APP; HDRPATH="$HERE/$APP.h"
  EXTRA=""
fi
if [ "$APP" = upwelling_bih ] || [ "$APP" = upwelling_bihgeo ] || [ "$APP" = upwelling_bihiso ]; then
  # (_bihgeo: the tracers along geopotentials, t3dmix4_geo.h; _bihiso: along isopycnals, t3dmix4_iso.h)
  # UPWELLING with biharmonic mixing (oracle/ref/upwelling_bih.h: UV_VIS4, TS_DIF4 along s-surfaces)
  UP=UPWELLING; HDR=$APP; HDRPATH="$HERE/$APP.h"
  EXTRA=""
fi
if [ "$APP" = upwelling_wetdry ]; then
  # UPWELLING with MASKING + WET_DRY (oracle/ref/upwelling_wetdry.h): pins wetdry.F and the WET_DRY branches
  UP=UPWELLING; HDR=$APP; HDRPATH="$HERE/$APP.h"
  EXTRA="-I$HERE/functionals"
fi
if [ "$APP" = upwelling_gls ]; then
  # the shipped upwelling.h with the generic length-scale closure switched on as a user does (-DGLS_MIXING: upwelling.h
  # then selects KANTHA_CLAYSON, N2S2_HORAVG, RI_SPLINES)
  UP=UPWELLING; HDR=upwelling; HDRPATH="upwelling.h"
  EXTRA="-DPERFECT_RESTART -DGLS_MIXING"
fi
if [ "$APP" = upwelling_my25 ]; then
  # the shipped upwelling.h with the Mellor-Yamada 2.5 closure switched on (-DMY25_MIXING: KANTHA_CLAYSON, N2S2_HORAVG, RI_SPLINES)
  UP=UPWELLING; HDR=upwelling; HDRPATH="upwelling.h"
  EXTRA="-DPERFECT_RESTART -DMY25_MIXING"
fi
if [ "$APP" = upwelling_my25_gal ]; then
  # MY25_MIXING with Galperin's stability functions, K_C4ADVECTION, plain shear, no smoothing (oracle/ref/upwelling_my25_gal.h)
  UP=UPWELLING; HDR=$APP; HDRPATH="$HERE/$APP.h"
  EXTRA=""
fi
if [ "$APP" = upwelling_gls_ca ] || [ "$APP" = upwelling_gls_cb ] || [ "$APP" = upwelling_gls_gal ]; then
  # GLS_MIXING in its other compile-time forms (oracle/ref/upwelling_gls_*.h; _ca is masked)
  UP=UPWELLING; HDR=$APP; HDRPATH="$HERE/$APP.h"
  EXTRA="-I$HERE/functionals"
fi
if [ "$APP" = seamount ]; then
  # the SEAMOUNT case without the user diagnostics hook ANA_DIAG (oracle/ref/seamount_nodiag.h)
  UP=SEAMOUNT; HDR=seamount_nodiag; HDRPATH="$HERE/seamount_nodiag.h"
  EXTRA=""
fi
if [ "$APP" = overflow ]; then
  # the OVERFLOW case (MIX_ISO_TS: t3dmix2_iso.h) without its output option AVERAGES (oracle/ref/overflow_noavg.h)
  UP=OVERFLOW; HDR=overflow_noavg; HDRPATH="$HERE/overflow_noavg.h"
  EXTRA=""
fi
if [ "$APP" = grav_adj ]; then
  # the GRAV_ADJ case without its output options AVERAGES / DIAGNOSTICS_TS / DIAGNOSTICS_UV (oracle/ref/grav_adj_nodiag.h)
  UP=GRAV_ADJ; HDR=grav_adj_nodiag; HDRPATH="$HERE/grav_adj_nodiag.h"
  EXTRA=""
fi
if [ "$APP" = kelvin_geouv ]; then
  # KELVIN (open boundaries) with the viscosity along geopotentials (oracle/ref/kelvin_geouv.h: MIX_GEO_UV; round 6)
  UP=KELVIN; HDR=kelvin_geouv; HDRPATH="$HERE/kelvin_geouv.h"
  EXTRA=""
fi
if [ "$APP" = benchmark_iso ]; then
  # BENCHMARK with tracer mixing along isopycnals and the nonlinear EOS (oracle/ref/benchmark_iso.h: MIX_ISO_TS; round 6)
  UP=BENCHMARK; HDR=benchmark_iso; HDRPATH="$HERE/benchmark_iso.h"
  EXTRA=""
fi
if [ "$APP" = kelvin_gls ]; then
  # KELVIN (open boundaries, spline solvers) with GLS_MIXING (oracle/ref/kelvin_gls.h): tkebc next to radiating edges
  UP=KELVIN; HDR=kelvin_gls; HDRPATH="$HERE/kelvin_gls.h"
  EXTRA=""
fi
if [ "$APP" = kelvin_splines ]; then
  # the KELVIN case (open boundaries) with the spline vertical solvers (oracle/ref/kelvin_splines.h)
  UP=KELVIN; HDR=kelvin_splines; HDRPATH="$HERE/kelvin_splines.h"
  EXTRA=""
fi
if [ "$APP" = upwelling_diag ]; then
  # ROMS/Include/upwelling.h AS SHIPPED (no PERFECT_RESTART): AVERAGES, DIAGNOSTICS_TS, DIAGNOSTICS_UV -- pins the per-term
  # tendencies of mod_diags.F / set_diags.F
  UP=UPWELLING; HDR=upwelling; HDRPATH="upwelling.h"
  EXTRA=""
fi
if [ "$APP" = upwelling_avg ]; then
  # the UPWELLING case with AVERAGES (oracle/ref/upwelling_avg.h): pins set_avg.F
  UP=UPWELLING; HDR=upwelling_avg; HDRPATH="$HERE/upwelling_avg.h"
  EXTRA=""
fi
WORK=$(mktemp -d /tmp/romsref_${APP}_XXXX)
[ -n "$KEEP_WORK" ] && echo "build_ref: keeping $WORK" || trap 'rm -rf "$WORK"' EXIT
mkdir -p "$OUT"
cd "$WORK"   # cpp must run from a writable cwd with absolute input paths

pp () {  # pp <abs .F path> -> $WORK/<base>.f90
  local b; b=$(basename "$1"); b=${b%.*}
  /usr/bin/cpp -P -traditional -w -D$UP -D"ROMS_HEADER=\"$HDRPATH\"" -D"HEADER=\"$HDR.h\"" \
    -DLINUX -DX86_64 -DGFORTRAN -DNestedGrids=1 \
    -D"ROOT_DIR=\"$REF\"" -D"ANALYTICAL_DIR=\"$REF/ROMS/Functionals\"" -D"HEADER_DIR=\"$REF/ROMS/Include\"" \
    -D'GIT_URL="x"' -D'GIT_REV="x"' -D'MY_OS="Linux"' -D'MY_CPU="x86_64"' -D'MY_FORT="gfortran"' \
    -D'MY_FC="flang"' -D'MY_FFLAGS="-O2"' $EXTRA $XDEF \
    -I$REF/ROMS/Include -I$REF/ROMS/Nonlinear -I$REF/ROMS/Functionals -I$REF/ROMS/Utility \
    -I$REF/ROMS/Drivers -I$REF/Master "$1" > $b.$2
  perl $REF/ROMS/Bin/cpp_clean $b.$2
}

# Reference files wanted in the library (those that are inactive for the
# application pre-process to nothing and are skipped).  Compiled by repeated
# passes until every file's modules are available (no hand-kept order).
FILES="mod_kinds mod_param mod_scalars mod_stepping mod_strings mod_iounits mod_parallel mod_eoscoef
  mod_clima mod_coupling mod_forces mod_grid mod_mixing mod_ocean mod_ncparam mod_boundary
  round dateclock strings timers yaml_parser get_env get_metadata stdout_mod destroy
  get_hash stats erf exchange_2d exchange_3d exchange_4d get_bounds tile_indices set_scoord set_weights
  metrics ini_hmixcoef stiffness mp_routines ntimestep
  bc_2d bc_3d zetabc u2dbc_im v2dbc_im t3dbc_im u3dbc_im v3dbc_im obc_volcons
  set_depth set_massflux rho_eos prsgrd t3dmix uv3dmix set_vbc set_zeta wvelocity diag ini_fields
  mod_sources uv_var_change wetdry step2d omega pre_step3d rhs3d step3d_uv step3d_t
  mpdata_adiff lmd_swfrac lmd_skpp lmd_bkpp lmd_vmix gls_prestep gls_corstep my25_prestep my25_corstep tkebc_im bulk_flux analytical
  mod_average uv_rotate vorticity set_masks set_avg mod_diags set_diags"
TODO=""
for m in $FILES; do
  src=""
  for d in Modules Utility Nonlinear Functionals; do
    [ -f $REF/ROMS/$d/$m.F ] && src=$REF/ROMS/$d/$m.F
  done
  [ -z "$src" ] && continue
  XDEF=""; [ $m = mod_sources ] && XDEF="-DANA_PSOURCE"
  pp $src f90
  [ $m = mod_sources ] && sed -i '/SUBROUTINE check_sources/,/END SUBROUTINE check_sources/d' $m.f90
  XDEF=""
  [ $(wc -c < $m.f90) -lt 20 ] && continue      # inactive for this application
  TODO="$TODO $m"
done
OBJS=""
while [ -n "$TODO" ]; do
  NEXT=""; PROG=0
  for m in $TODO; do
    if $FC -c -O2 -fPIC $m.f90 -o $m.o 2> $m.err; then OBJS="$OBJS $m.o"; PROG=1; else NEXT="$NEXT $m"; fi
  done
  TODO=$NEXT
  if [ $PROG -eq 0 ]; then echo "build_ref: cannot compile:$TODO"; for m in $TODO; do head -5 $m.err; done; exit 1; fi
done
pp $HERE/ref_glue.F90 f90 && mv ref_glue.f90 ref_glue_pp.F90
$FC -c -O2 -fPIC -ffree-form ref_glue_pp.F90 -o ref_glue.o
$FC -shared -o $OUT/libromsref_$APP.so $OBJS ref_glue.o
echo "build_ref: wrote $OUT/libromsref_$APP.so"
