/* TEST INFRASTRUCTURE.  The C library's own REAL sinf / cosf / asinf / atanf / atan2f over an array: what the reference's rotated-frame locate calls through
 * the Fortran intrinsics (ez_lac.inc:31-47, ez_cal.inc:22-47), and what librmn_amd/csrc/libm_exact.h restates for the device (compared in tests/). */
#include <math.h>
#include <stddef.h>
void orc_libm_apply(int fn, const float *a, const float *b, float *out, size_t n)
{
#pragma omp parallel for schedule(static)
    for (size_t k = 0; k < n; k++)
        out[k] = fn == 0 ? sinf(a[k]) : fn == 1 ? cosf(a[k]) : fn == 2 ? asinf(a[k]) : fn == 3 ? atanf(a[k]) : atan2f(a[k], b[k]);
}
