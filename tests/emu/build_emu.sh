#!/bin/bash
# tests/emu/build_emu.sh -- DEVELOPMENT/TEST ONLY.
# Compiles the HIP kernel sources of roms_amd/csrc with a host C++ compiler and
# -DROMS_CPU_EMU (thread blocks and threads executed serially) into
# tests/emu/libroms_hip_emu.so, so that kernel LOGIC can be unit-tested against the
# oracle on machines without a GPU.  Not built by __graft_entry__.build(), never
# loaded by the roms_amd package: the product has no CPU fallback.
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
exec 9>"$HERE/.build.lock"      # one build at a time (pytest-xdist workers each run this from a module fixture)
flock 9
SRC=$HERE/../../roms_amd/csrc
OUT=$HERE/libroms_hip_emu.so
DEFS="${ROMS_DEFS:-}"
mkdir -p $HERE/obj
objs=""
for f in $SRC/*.cpp; do
  o=$HERE/obj/$(basename ${f%.cpp}).o
  if [ ! -f $o ] || [ $f -nt $o ] || [ -n "$(find $SRC -name '*.h' -newer $o)" ] || [ $HERE/../../include/roms_hip.h -nt $o ]; then
    g++ -O2 -g -fPIC -std=c++17 -ffp-contract=off -DROMS_CPU_EMU $DEFS -Wall -Wno-unused-variable -Wno-unused-but-set-variable -c $f -o $o &
  fi
  objs="$objs $o"
done
wait
# link only when something changed, into a temporary name that replaces the library in one step: other test processes may
# be loading it meanwhile
newer () { for o in "${@:2}"; do [ "$o" -nt "$1" ] && return 0; done; [ ! -f "$1" ]; }
if newer $OUT $objs; then
  g++ -shared -Wl,-Bsymbolic -o $OUT.tmp.$$ $objs && mv -f $OUT.tmp.$$ $OUT
fi
echo "built $OUT"

# Fortran host driver linked against the emulated library (multi-tile CPU tests)
FOBJ=$HERE/obj_f
mkdir -p $FOBJ
HOSTSRC=$HERE/../../roms_amd/host
if [ ! -f $FOBJ/nc3.o ] || [ $HOSTSRC/nc3.c -nt $FOBJ/nc3.o ]; then
  gcc -O2 -fPIC -D_FILE_OFFSET_BITS=64 -Wall -c $HOSTSRC/nc3.c -o $FOBJ/nc3.o
fi
rebuild=""
for f in roms_hip_mod roms_host roms_output roms_host_api; do
  if [ -n "$rebuild" ] || [ ! -f $FOBJ/$f.o ] || [ $HOSTSRC/$f.f90 -nt $FOBJ/$f.o ]; then
    /opt/rocm/bin/amdflang -O2 -fPIC -ffp-contract=off -module-dir $FOBJ -c $HOSTSRC/$f.f90 -o $FOBJ/$f.o
    rebuild=1
  fi
done
HOBJS="$FOBJ/roms_hip_mod.o $FOBJ/roms_host.o $FOBJ/nc3.o $FOBJ/roms_output.o $FOBJ/roms_host_api.o"
if newer $HERE/libroms_host_emu.so $HOBJS $OUT; then
  /opt/rocm/bin/amdflang -shared -o $HERE/libroms_host_emu.so.tmp.$$ $HOBJS -L$HERE -lroms_hip_emu -Wl,-rpath,'$ORIGIN' && \
    mv -f $HERE/libroms_host_emu.so.tmp.$$ $HERE/libroms_host_emu.so
fi
echo "built $HERE/libroms_host_emu.so"

# the stand-alone driver against the emulated library (its run report is checked on CPU too)
if [ ! -f $FOBJ/romsM.o ] || [ $HOSTSRC/romsM.f90 -nt $FOBJ/romsM.o ] || [ $FOBJ/roms_host.o -nt $FOBJ/romsM.o ]; then
  /opt/rocm/bin/amdflang -O2 -fPIC -ffp-contract=off -module-dir $FOBJ -c $HOSTSRC/romsM.f90 -o $FOBJ/romsM.o
fi
if newer $HERE/romsM_emu $FOBJ/romsM.o $HERE/libroms_host_emu.so $OUT; then
  /opt/rocm/bin/amdflang -o $HERE/romsM_emu.tmp.$$ $FOBJ/romsM.o -L$HERE -lroms_host_emu -lroms_hip_emu -Wl,-rpath,'$ORIGIN' && \
    mv -f $HERE/romsM_emu.tmp.$$ $HERE/romsM_emu
fi
echo "built $HERE/romsM_emu"
