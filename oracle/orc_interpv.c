/* orc_interpv.c -- TEST INFRASTRUCTURE (CPU oracle).
 *
 * Plain-C restatement of the reference's 1-D (vertical) interpolation package, src/interpv:
 * Interp1D_FindPos, Interp1D_NearestNeighbour, Interp1D_Linear, Interp1D_CubicLagrange, Interp1D_CubicWithDerivs,
 * Extrap1D_Fixed, Extrap1D_LapseRate, Extrap1D_Abort, each in REAL (no suffix) and REAL*8 (suffix 8).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it; the product never does.
 *
 * Pins:
 *   FindPos, NearestNeighbour, Extrap1D_Fixed, Extrap1D_LapseRate compile from the reference's own files
 *   (oracle/build_ref.sh -> oracle/_ref/libinterpvref.so) and are compared bit for bit (tests/test_oracle_interpv.py).
 *   Linear, CubicLagrange, CubicWithDerivs, Extrap1D_Abort `use app` (the absent App submodule's module, for the
 *   error log only) and are unbuildable here: pinned by the checks of the reference's own test program
 *   (src/interpv/test/Test_Interp1D.F90: its data, its clamp / sin / tan / derivative criteria and its literal
 *   lapse-rate answers) and by hand-derived known answers.
 * Extrap1D_Surface / Extrap1D_SurfaceWind take a host callback (`external flux`) from the physics library per
 * call: not restated (out of scope).
 */
#include <stddef.h>
#include <stdlib.h>
#include "orc_interpv.h"

#define REAL float
#define FN(x) x
#include "orc_interpv_tmpl.h"
#undef REAL
#undef FN

#define REAL double
#define FN(x) x##8
#include "orc_interpv_tmpl.h"
#undef REAL
#undef FN
