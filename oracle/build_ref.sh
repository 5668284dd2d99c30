#!/usr/bin/env bash
# oracle/build_ref.sh -- TEST INFRASTRUCTURE.
#
# Builds oracle/_ref/libezref.so: the reference's OWN EZ-interpolation sources, compiled where
# they lie under /root/reference (nothing is copied), with gcc (C front-end) and AMD flang
# (Fortran leaf kernels).  Only runs where /root/reference exists (the build container);
# the GPU box uses the prebuilt .so that travels with the snapshot.
#
# What is and is not in this build (see DESIGN.md "Oracle"):
#   * src/interp/*.c, src/interp/f_ezscint.F90 (all 77 .inc leaf kernels), and the src/base
#     Fortran files the path calls (grll grps permut dgauss ordleg llfxy xyfll mxm valide igaxg95)
#     compile unmodified, with NO stub headers.
#   * src/base/igaxg.f90 and xgaig.f90 `use app` (module of the absent App submodule) and are
#     therefore unbuildable; cigaxg_/cxgaig_ are resolved to the oracle's restatement
#     (oracle/orc_igaxg.c, -DORC_FORTRAN_ABI).  This is the one non-reference piece of the .so.
#   * The packers / compressor (src/packers, src/compresseur) #include <App.h> from the same
#     absent submodule: unbuildable here, NOT part of this build ("parity unpinned" for them).
#   * src/interpv (vertical interpolation): Interp1D_Constants, Interp1D_FindPos, Interp1D_NearestNeighbour,
#     Extrap1D_Fixed and Extrap1D_LapseRate compile unmodified into oracle/_ref/libinterpvref.so.  Interp1D_Linear,
#     Interp1D_CubicLagrange, Interp1D_CubicWithDerivs, Extrap1D_Abort, Extrap1D_Surface, Extrap1D_SurfaceWind
#     `use app` (same absent submodule): unbuildable, not part of the build.
#   * FST file-I/O symbols (c_fstinf, fstluk_, fnom_ ...) referenced by grid-from-file code are
#     left UNDEFINED; the library is loaded with lazy binding and those paths are never called.
set -euo pipefail
R=${EZ_REFERENCE:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=$HERE/_ref
B=$HERE/_build/ref
FL=${FLANG:-/opt/rocm/lib/llvm/bin/flang}
if [ ! -d "$R/src/interp" ]; then
  echo "build_ref: $R not present, keeping prebuilt $OUT (if any)"; exit 0
fi
mkdir -p "$OUT" "$B"
INC="-I $R/include -I $R/src/PUBLIC_INCLUDES -I $R/src/PUBLIC_INCLUDES/rmn -I $R/src -I $R/src/interp"
CFLAGS="-std=gnu99 -O2 -fPIC -D_GNU_SOURCE -w -ffp-contract=off"
for f in "$R"/src/interp/*.c "$R"/src/base/ftnStrLen.c; do
  o=$B/$(basename "$f" .c).o
  [ "$o" -nt "$f" ] || gcc $CFLAGS $INC -c "$f" -o "$o"
done
FFLAGS="-O2 -fPIC -cpp -w -ffp-contract=off"
o=$B/f_ezscint.o
[ "$o" -nt "$R/src/interp/f_ezscint.F90" ] || \
  "$FL" $FFLAGS -I "$R/src/interp" -I "$R/src/PUBLIC_INCLUDES" -I "$R/src" -c "$R/src/interp/f_ezscint.F90" -o "$o" 2>/dev/null
for f in grll.f grps.f permut.f dgauss.F ordleg.F llfxy.F xyfll.F mxm.F90 valide.f igaxg95.F; do
  o=$B/base_${f%.*}.o
  [ "$o" -nt "$R/src/base/$f" ] || \
    "$FL" $FFLAGS -I "$R/src/base" -I "$R/src/PUBLIC_INCLUDES" -I "$R/src" -c "$R/src/base/$f" -o "$o" 2>/dev/null
done
"$FL" $FFLAGS -c "$R/src/primitives/up2low.f" -o "$B/up2low.o" 2>/dev/null
gcc $CFLAGS -DORC_FORTRAN_ABI -c "$HERE/orc_igaxg.c" -o "$B/orc_igaxg_fabi.o"
"$FL" -shared -Wl,-z,lazy -Wl,-Bsymbolic -o "$OUT/libezref.so" "$B"/*.o -lm
echo "build_ref: wrote $OUT/libezref.so"
BV=$HERE/_build/refv
mkdir -p "$BV"
V=$R/src/interpv
for f in Interp1D_Constants Interp1D_FindPos Interp1D_NearestNeighbour Extrap1D_Fixed Extrap1D_LapseRate; do
  o=$BV/$f.o
  [ "$o" -nt "$V/$f.F90" ] || ( cd "$BV" && "$FL" $FFLAGS -I "$V" -c "$V/$f.F90" -o "$o" 2>/dev/null )
done
"$FL" -shared -o "$OUT/libinterpvref.so" "$BV"/*.o -lm
echo "build_ref: wrote $OUT/libinterpvref.so"
nm -D --undefined-only "$OUT/libezref.so" | grep -v -E "GLIBC|_Fortran|__cxa|__gmon|_ITM|__deregister|__register" | awk '{print "  undefined (never called):", $2}'
