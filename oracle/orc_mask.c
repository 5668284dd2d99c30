/*
 * oracle/orc_mask.c -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * Masked interpolation helpers of src/interp/ez_mask.c: c_ezsint_mask (:184-227 -> qqq_ezsint_mask,
 * ezget_mask_zones.inc:61-105), c_ezget_mask_zones (:231-264 -> qqq_ezget_mask_zones, ezget_mask_zones.inc:1-59),
 * c_ezsint_mdm / c_ezuvint_mdm (:127-181) with lorenzo_mask_fill method 2 (lorenzo_mask_fill.inc:30-37).
 *
 * The Fortran indexes mask_in(ix+1, iy+1) and mask_in(nint(x), nint(y)) without bounds: for points in the last
 * source column the reference reads the first element of the next row (column-major storage), and past the last row it
 * reads beyond the array.  The restatement keeps the linear addressing and clamps the address to the array.
 */
#include <math.h>
#include <stdlib.h>
#include "orc_ez.h"

static int mask_at(const int *m, int ni, int nj, int i1, int j1)      /* 1-based (i1, j1), Fortran linear addressing */
{
    long k = (long)(j1 - 1) * ni + (i1 - 1);
    long n = (long)ni * nj;
    if (k < 0) k = 0;
    if (k >= n) k = n - 1;
    return m[k];
}
static int nint_f(float v) { return (int)lroundf(v); }

int orc_ezsint_mask(orc_gridset *gs, int cloud_linear, int *mask_out, const int *mask_in)
{
    orc_calcxy(gs);
    const orc_grid *gi = gs->gdin, *go = gs->gdout;
    int ni_in = gi->ni, nj_in = gi->nj, ni_out = go->ni, nj_out = go->nj;
    const float *x = gs->x, *y = gs->y;
    for (int k = 0; k < ni_out * nj_out; k++) {
        int ix = (int)x[k], iy = (int)y[k], nix = nint_f(x[k]), niy = nint_f(y[k]);
        mask_out[k] = 1;
        if (ix < 1 || ix > ni_in || iy < 1 || iy > nj_in) mask_out[k] = 0;
        else if (mask_at(mask_in, ni_in, nj_in, nix, niy) == 0) mask_out[k] = 0;
    }
    if (cloud_linear)
        for (int j = 0; j < nj_out - 1; j++)
            for (int i = 0; i < ni_out - 1; i++) {
                int k = j * ni_out + i;
                if (mask_out[k] != 1) continue;
                int ix = (int)x[k], iy = (int)y[k];
                if (mask_at(mask_in, ni_in, nj_in, ix + 1, iy) == 0 || mask_at(mask_in, ni_in, nj_in, ix, iy + 1) == 0 ||
                    mask_at(mask_in, ni_in, nj_in, ix + 1, iy + 1) == 0) mask_out[k] = 0;
            }
    return 0;
}

int orc_ezget_mask_zones(orc_gridset *gs, int *mask_zones, const int *mask_in)
{
    orc_calcxy(gs);
    const orc_grid *gi = gs->gdin, *go = gs->gdout;
    int ni_in = gi->ni, nj_in = gi->nj;
    for (int k = 0; k < go->ni * go->nj; k++) {
        int ix = (int)gs->x[k], iy = (int)gs->y[k];
        if (ix < 1 || ix > ni_in || iy < 1 || iy > nj_in) { mask_zones[k] = 7; continue; }   /* outside_src_grid */
        int nmissing = 0;                          /* nearest_point_missing (5) is always overwritten by the select case */
        for (int kk = 1; kk <= 2; kk++) for (int l = 1; l <= 2; l++) if (mask_at(mask_in, ni_in, nj_in, ix + kk - 1, iy + l - 1) == 0) nmissing++;
        mask_zones[k] = 4 - nmissing;              /* 4 all present ... 0 all missing */
    }
    return 0;
}

/* lorenzo_mask_fill method 2: masked points take the field's minimum (computed over ALL points) */
void orc_mask_fill2(float *fld, const int *mask, int n)
{
    float rmin = fld[0];
    for (int k = 1; k < n; k++) if (fld[k] < rmin) rmin = fld[k];
    for (int k = 0; k < n; k++) if (mask[k] == 0) fld[k] = rmin;
}
