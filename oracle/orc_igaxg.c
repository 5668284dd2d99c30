/*
 * oracle/orc_igaxg.c -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * Restatement of the grid-descriptor codecs
 *   CIGAXG  (reference: src/base/igaxg.f90:5-143)   integer descriptors -> real
 *   CXGAIG  (reference: src/base/xgaig.f90:4-195)   real descriptors -> integer
 * for the grid types the hot path uses ('L', 'E', 'A', 'B', 'G', and old-style 'N'/'S' as
 * needed by the polar-wind helper grids of ez_calcnpolarwind.c:86-91).
 *
 * Why this file exists twice in the build: igaxg.f90 / xgaig.f90 `use app`, a module
 * of the un-vendored App submodule (empty in /root/reference), so those two files are
 * UNBUILDABLE here.  With -DORC_FORTRAN_ABI this file also exports the Fortran-ABI
 * symbols cigaxg_ / cxgaig_ so that oracle/_ref/libezref.so (everything else compiled
 * from the reference's own sources) can link.  DESIGN.md states this hybrid explicitly.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>

/* Fortran NINT for default REAL: round half away from zero. */
static int nint_f(float v) { return (int)lroundf(v); }

/* src/base/igaxg.f90:51-135 */
void orc_cigaxg(char grtyp, float *xg1, float *xg2, float *xg3, float *xg4,
                int ig1, int ig2, int ig3, int ig4)
{
    switch (grtyp) {
    case 'N': case 'S':                           /* igaxg.f90:53-58 : "ancien style" only */
        if (ig4 < 32768) {
            *xg1 = (float)ig2 * 0.1f;
            *xg2 = (float)ig1 * 0.1f;
            *xg3 = (float)ig4 * 100.f;
            *xg4 = (float)ig3 * 0.01f;
        } else {
            fprintf(stderr, "orc_cigaxg: new-style N/S descriptors not restated\n");
            *xg1 = *xg2 = *xg3 = *xg4 = 0.f;
        }
        break;
    case 'A': case 'B': case 'G':                 /* igaxg.f90:87-91 */
        *xg1 = (float)ig1; *xg2 = (float)ig2; *xg3 = 0.f; *xg4 = 0.f;
        break;
    case 'L':                                     /* igaxg.f90:93-97 : REAL arithmetic */
        *xg1 = (float)ig3 * 0.01f - 90.f;
        *xg2 = (float)ig4 * 0.01f;
        *xg3 = (float)ig1 * 0.01f;
        *xg4 = (float)ig2 * 0.01f;
        break;
    case 'E': {                                   /* igaxg.f90:106-122 : REAL*8 then rounded */
        int i2b = ig3 & 3;
        int lg3 = ig3 >> 2;
        int lg1 = (ig1 << 2) | i2b;
        i2b = ig4 & 3;
        int lg4 = ig4 >> 2;
        int lg2 = (ig2 << 2) | i2b;
        if (lg2 > 3600) lg2 -= 7201;
        /* (LG1 - 3600.0D0) / 40.0 : double / (real 40.0 widened) */
        *xg1 = (float)(((double)lg1 - 3600.0) / 40.0);
        if (lg3 < 3559) lg3 += 16384;
        *xg2 = (float)(((double)lg3 - 3600.0) / 40.0);
        *xg3 = (float)((double)lg2 / 40.0);
        *xg4 = (float)((double)lg4 / 40.0);
        break;
    }
    default:
        fprintf(stderr, "orc_cigaxg: grid type '%c' not restated\n", grtyp);
        *xg1 = *xg2 = *xg3 = *xg4 = 0.f;
    }
}

/* src/base/xgaig.f90:57-192 */
void orc_cxgaig(char grtyp, int *ig1, int *ig2, int *ig3, int *ig4,
                float xg1, float xg2, float xg3, float xg4)
{
    switch (grtyp) {
    case 'N': case 'S':                           /* xgaig.f90:57-66 : the in-range ("old style") branch */
        *ig1 = nint_f(xg2 * 10.f);
        *ig2 = nint_f(xg1 * 10.f);
        *ig3 = nint_f(xg4 * 100.f);
        *ig4 = nint_f(xg3 * 0.01f);
        while (*ig3 < 0) *ig3 += 36000;
        if (*ig1 < 0 || *ig2 < 0 || *ig1 > 2047 || *ig2 > 2047 || *ig4 > 32000)
            fprintf(stderr, "orc_cxgaig: N/S descriptor outside the old-style range is not restated\n");
        break;
    case 'A': case 'B': case 'G':                 /* xgaig.f90:98-104 : truncation */
        *ig1 = (int)xg1; *ig2 = (int)xg2; *ig3 = 0; *ig4 = 0;
        break;
    case 'L':                                     /* xgaig.f90:129-139 */
        *ig1 = nint_f(xg3 * 100.f);
        *ig2 = nint_f(xg4 * 100.f);
        *ig3 = nint_f((90.f + xg1) * 100.f);
        *ig4 = nint_f(xg2 * 100.f);
        while (*ig4 < 0) *ig4 += 36000;
        break;
    case 'E': {                                   /* xgaig.f90:141-168 */
        float xxg2 = xg2, xxg4 = xg4;
        while (xxg2 < 0) xxg2 += 360.f;
        while (xxg4 < 0) xxg4 += 360.f;
        int j1 = nint_f((xg1 + 90.f) * 40.f);
        int j2 = nint_f(xg3 * 40.f);
        int j3 = nint_f((xxg2 + 90.f) * 40.f);
        if (j3 >= 16384) j3 -= 16384;
        int j4 = nint_f(xxg4 * 40.f);
        int i2b = j1 & 3;
        j1 = j1 >> 2;
        j3 = (j3 << 2) | i2b;
        if (j2 < 0) j2 += 7201;
        i2b = j2 & 3;
        j2 = j2 >> 2;
        j4 = (j4 << 2) | i2b;
        *ig1 = j1; *ig2 = j2; *ig3 = j3; *ig4 = j4;
        break;
    }
    default:
        fprintf(stderr, "orc_cxgaig: grid type '%c' not restated\n", grtyp);
        *ig1 = *ig2 = *ig3 = *ig4 = 0;
    }
}

#ifdef ORC_FORTRAN_ABI
/* Fortran-ABI entry points consumed by oracle/_ref/libezref.so (see header comment).
 * Hidden trailing CHARACTER length is ignored. */
void cigaxg_(const char *grtyp, float *xg1, float *xg2, float *xg3, float *xg4,
             const int *ig1, const int *ig2, const int *ig3, const int *ig4)
{
    orc_cigaxg(grtyp[0], xg1, xg2, xg3, xg4, *ig1, *ig2, *ig3, *ig4);
}
void cxgaig_(const char *grtyp, int *ig1, int *ig2, int *ig3, int *ig4,
             const float *xg1, const float *xg2, const float *xg3, const float *xg4)
{
    orc_cxgaig(grtyp[0], ig1, ig2, ig3, ig4, *xg1, *xg2, *xg3, *xg4);
}
#endif
