/*
 * oracle/orc_ez.c -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * Plain-C restatement of librmn's EZ scalar interpolation path: grid definition, Gaussian
 * latitudes, Newton coefficient tables, target lat/lon, locate, the 11 point kernels, polar
 * zones and polar/extrapolation correction.  Straight per-point loops in the reference's own
 * operation order and precision (REAL = float, REAL*8 = double; Fortran evaluates an all-REAL
 * right-hand side in float even when the left-hand side is REAL*8).  Build with
 * -ffp-contract=off.  Each function cites the reference file:line it follows.
 *
 * Scope: grid types L, A, B, G (global, ig1=0, ig2=0), E, and Z on L / E reference.
 * Hemispheric expansion, Y clouds, U supergrids, PS / Lambert grids are out of scope
 *  and rejected.
 */
#include "orc_ez.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

void orc_cigaxg(char grtyp, float *xg1, float *xg2, float *xg3, float *xg4, int ig1, int ig2, int ig3, int ig4);
void orc_cxgaig(char grtyp, int *ig1, int *ig2, int *ig3, int *ig4, float xg1, float xg2, float xg3, float xg4);
void orc_gfxyfll(float *lon_rot, float *lat_rot, const float *lon, const float *lat, int n,
                 float xlat1, float xlon1, float xlat2, float xlon2);
void orc_gfllfxy(float *lon_true, float *lat_true, const float *lon_rot, const float *lat_rot, int n,
                 float xlat1, float xlon1, float xlat2, float xlon2);

static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

void orc_default_opts(orc_opts *o)   /* src/interp/ez_statics.c:17 */
{
    o->degre_interp = ORC_CUBIC;
    o->degre_extrap = ORC_EXTRAP_MAXIMUM;
    o->polar_correction = 1;
    o->vecteur = 0;
    o->valeur_extrap = 0.0f;
}

/* ------------------------------------------------------------------------------------------
 * Gaussian latitudes.  All REAL (float) arithmetic, float libm.
 * ---------------------------------------------------------------------------------------- */

/* ORDLEG, src/base/ordleg.F:50-86 */
static float ordleg(float coa, int ir)
{
    const float sqr2 = sqrtf(2.f);
    int irpp = ir + 1;
    int irppm = irpp - 1;
    float delta = acosf(coa);
    float theta = delta;
    float c1 = sqr2;
    float fn = 0.f, fn2 = 0.f, fn2sq;
    for (int n = 1; n <= irppm; n++) {
        fn = (float)n;
        fn2 = 2.f * fn;
        fn2sq = fn2 * fn2;
        c1 = c1 * sqrtf(1.0f - 1.0f / fn2sq);
    }
    int n = irppm;
    float ang = fn * theta;
    float s1 = 0.0f, c4 = 1.0f, a = -1.0f, b = 0.0f;
    int n1 = n + 1;
    for (int kk = 1; kk <= n1; kk += 2) {
        int k = kk - 1;
        if (k == n) c4 = 0.5f * c4;
        s1 = s1 + c4 * cosf(ang);
        a = a + 2.0f;
        b = b + 1.0f;
        float fk = (float)k;
        ang = theta * (fn - fk - 2.0f);
        c4 = (a * (fn - b + 1.0f) / (b * (fn2 - a))) * c4;
    }
    return s1 * c1;
}

/* DGAUSS (KASE = GLOBAL), src/base/dgauss.F:90-131; pie from src/base/pi.cdk */
static void dgauss_global(int n, float *roots /* 1-based: roots[1..n] */)
{
    const float pie = 3.1415926535898f;
    const float tol = 1.0e-6f;
    float normn = sqrtf(2.0f / (2.0f * (float)n + 1.0f));
    float normnm = sqrtf(2.0f / (2.0f * (float)n - 1.0f));
    int l = n / 2;
    for (int i = 1; i <= l; i++) {
        int j = i;
        float t = (float)(4 * j - 1) * pie / (float)(4 * n + 2);
        roots[i] = cosf(t + 1.0f / (8.0f * (float)(n * n) * tanf(t)));
    }
    for (int i = 1; i <= l; i++) {
        float delta;
        do {
            float g = ordleg(roots[i], n);
            float gm = ordleg(roots[i], n - 1);
            float pn = normn * g;
            float pnm = normnm * gm;
            float rdpdx = (roots[i] * roots[i] - 1.0f) / ((float)n * (roots[i] * pn - pnm));
            delta = -pn * rdpdx;
            roots[i] = roots[i] + delta;
        } while (fabsf(delta) > tol);
        roots[n + 1 - i] = -roots[i];
    }
    if (n % 2 != 0) roots[l + 1] = 0.0f;
}

/* ez_glat (hem = GLOBAL), src/interp/ez_glat.inc:26-67; rdtodg from src/interp/pi.cdk */
void orc_gauss_lat(float *lat, int nj)
{
    const float rdtodg = 57.295779513082f;
    float *g = (float *)malloc((size_t)(nj + 2) * sizeof(float));
    dgauss_global(nj, g);
    for (int j = 1; j <= nj / 2; j++) {
        float t = g[j]; g[j] = g[nj + 1 - j]; g[nj + 1 - j] = t;
    }
    for (int j = 1; j <= nj; j++) lat[j - 1] = 90.f - rdtodg * acosf(g[j]);
    free(g);
}

/* ------------------------------------------------------------------------------------------
 * ez_nwtncof, src/interp/ez_nwtncof.inc:20-178.  cx is (ni,6) column-major, cy is (j1:j2,6).
 * ---------------------------------------------------------------------------------------- */
static void six_recips(float *c, int stride, float x1, float x2, float x3, float x4)
{
    c[0 * stride] = 1.f / (x2 - x1);
    c[1 * stride] = 1.f / (x3 - x1);
    c[2 * stride] = 1.f / (x3 - x2);
    c[3 * stride] = 1.f / (x4 - x1);
    c[4 * stride] = 1.f / (x4 - x2);
    c[5 * stride] = 1.f / (x4 - x3);
}

void orc_nwtncof(float *cx, float *cy, const float *ax, const float *ay, int ni, int nj,
                 int j1, int j2, int extension)
{
#define AX(i) ax[(i) - 1]
#define AY(j) ay[(j) - j1]
    int nnj = j2 - j1 + 1;
    for (int i = 1; i <= ni - 1; i++)
        if (AX(i + 1) <= AX(i)) { fprintf(stderr, "orc_nwtncof: x axis not monotonic\n"); exit(13); }
    for (int j = 1; j <= nj - 1; j++)
        if (ay[j] <= ay[j - 1]) { fprintf(stderr, "orc_nwtncof: y axis not monotonic\n"); exit(13); }
    for (int k = 0; k < 6 * ni; k++) cx[k] = 1.0f;
    for (int k = 0; k < 6 * nnj; k++) cy[k] = 1.0f;
    for (int i = 2; i <= ni - 2; i++)
        six_recips(&cx[i - 1], ni, AX(i - 1), AX(i), AX(i + 1), AX(i + 2));
    for (int j = j1 + 1; j <= j2 - 2; j++)
        six_recips(&cy[j - j1], nnj, AY(j - 1), AY(j), AY(j + 1), AY(j + 2));
    if (extension == 1) {            /* :117-141 */
        six_recips(&cx[0], ni, AX(1) - (AX(ni) - AX(ni - 1)), AX(1), AX(2), AX(3));
        six_recips(&cx[ni - 2], ni, AX(ni - 2), AX(ni - 1), AX(ni), AX(ni) + (AX(2) - AX(1)));
    }
    if (extension == 2) {            /* :143-177 */
        six_recips(&cx[0], ni, AX(1) - (360.0f - AX(ni)), AX(1), AX(2), AX(3));
        six_recips(&cx[ni - 2], ni, AX(ni - 2), AX(ni - 1), AX(ni), AX(1) + 360.0f);
        six_recips(&cx[ni - 1], ni, AX(ni - 1), AX(ni), AX(1) + 360.0f, AX(2) + 360.0f);
    }
#undef AX
#undef AY
}

/* ------------------------------------------------------------------------------------------
 * ez_xpncof, src/interp/ez_xpncof.c:48-226 (global grids only: ig1 == 0)
 * ---------------------------------------------------------------------------------------- */
static void xpncof(orc_grid *g)
{
    float first_lat, first_lon, last_lat, last_lon, extra_lon, dlat, dlon;
    int ni = g->ni, nj = g->nj;
    g->i1 = 1; g->i2 = ni; g->j1 = 1; g->j2 = nj;
    switch (g->grtyp) {
    case 'L':                                                   /* :69-106 */
        g->extension = 0;
        orc_cigaxg('L', &first_lat, &first_lon, &dlat, &dlon, g->ig[0], g->ig[1], g->ig[2], g->ig[3]);
        if ((double)(first_lat - dlat) > (-90.0 + 0.01 * (double)dlat)) break;
        last_lat = first_lat + (float)(nj - 1) * dlat;
        if ((double)(last_lat + dlat) < (90.0 - 0.01 * (double)dlat)) break;
        if (first_lon < 0.0) first_lon += 360.0;
        last_lon = first_lon + (dlon * (float)(ni - 1));
        if ((double)(last_lon - first_lon) > (360 - 0.01 * (double)dlon)) { g->extension = 1; break; }
        extra_lon = last_lon + dlon;
        if ((double)(extra_lon - first_lon) > (360.0 - (double)dlon * 0.01)) g->extension = 2;
        break;
    case 'A': case 'G': g->extension = 2;                       /* :107-125 */
        if (g->ig[0] == 1) { g->j1 = -nj + 1; g->j2 = nj; } else if (g->ig[0] == 2) { g->j1 = 1; g->j2 = 2 * nj; }
        break;
    case 'B': g->extension = 1;                                 /* :127-145 */
        if (g->ig[0] == 1) { g->j1 = -nj + 2; g->j2 = nj; } else if (g->ig[0] == 2) { g->j1 = 1; g->j2 = 2 * nj - 1; }
        break;
    case 'E': break;                                            /* :152-155 extension untouched (0) */
    case 'Z':
        if (g->grref == 'E') {                                  /* :161-171 */
            g->extension = ((g->ax[ni - 1] - g->ax[0]) < 359.0) ? 0 : 1;
        } else if (g->grref == 'L') {                           /* :173-207 */
            g->extension = 0;
            first_lat = g->ay[0]; last_lat = g->ay[nj - 1];
            dlat = g->ay[1] - g->ay[0];
            if ((double)(first_lat - dlat) > (-90.0 + 0.01 * (double)dlat)) break;
            dlat = g->ay[nj - 1] - g->ay[nj - 2];
            if ((double)(last_lat + dlat) < (90.0 - 0.01 * (double)dlat)) break;
            first_lon = g->ax[0]; last_lon = g->ax[ni - 1];
            dlon = g->ax[ni - 1] - g->ax[ni - 2];
            if ((double)(last_lon - first_lon) > (360.0 - 0.01 * (double)dlon)) g->extension = 1;
            else {
                extra_lon = last_lon + dlon;
                if ((double)(extra_lon - first_lon) > (360.0 - 0.01 * (double)dlon)) g->extension = 2;
            }
        } else g->extension = 0;
        break;
    default: g->extension = 0;
    }
}

/* ------------------------------------------------------------------------------------------
 * Grid definition: c_ezgdef_fmem (ezgdef_fmem.c:37-93) -> c_ezdefxg (ez_defxg.c:28-170),
 * c_ezdefaxes (ez_defaxes.c:31-114), ez_calcxpncof, ez_calcntncof (ez_calcntncof.c:26-50)
 * ---------------------------------------------------------------------------------------- */
orc_grid *orc_grid_define(int ni, int nj, char grtyp, char grref,
                          int ig1, int ig2, int ig3, int ig4, const float *ax, const float *ay)
{
    orc_grid *g = (orc_grid *)calloc(1, sizeof(orc_grid));
    g->grtyp = grtyp; g->grref = grref; g->ni = ni; g->nj = nj;
    g->ig[0] = ig1; g->ig[1] = ig2; g->ig[2] = ig3; g->ig[3] = ig4;
    switch (grtyp) {
    case 'A': case 'B':
        if (ig1 < 0 || ig1 > 2 || ig2 < 0 || ig2 > 1) { fprintf(stderr, "orc_grid_define: bad ig1 / ig2\n"); free(g); return NULL; }
        break;
    case 'G':
        if (ig1 < 0 || ig1 > 2 || ig2 < 0 || ig2 > 1) { fprintf(stderr, "orc_grid_define: bad ig1 / ig2\n"); free(g); return NULL; }
        break;
    case 'L': case 'E': case 'Z': case 'N': case 'S': break;
    case 'Y': if (grref != 'L') { fprintf(stderr, "orc_grid_define: 'Y' grids on '%c' out of scope\n", grref); free(g); return NULL; } break;   /* target only */
    default: fprintf(stderr, "orc_grid_define: grid type '%c' out of scope\n", grtyp); free(g); return NULL;
    }
    /* c_ezdefxg */
    switch (grtyp) {
    case 'A': case 'G':                                         /* ez_defxg.c:39-62, double -> float */
        g->xg[3] = (float)(360. / ni); g->xg[1] = 0.0f;
        if (ig1 == 0) { g->xg[2] = (float)(180. / nj); g->xg[0] = (float)(-90. + 0.5 * (double)g->xg[2]); }
        else if (ig1 == 1) { g->xg[2] = (float)(90. / nj); g->xg[0] = (float)(0.5 * (double)g->xg[2]); }
        else { g->xg[2] = (float)(90. / nj); g->xg[0] = (float)(-90. + 0.5 * (double)g->xg[2]); }
        break;
    case 'B':                                                   /* ez_defxg.c:77-103 */
        g->xg[3] = (float)(360. / (ni - 1)); g->xg[1] = 0.0f;
        if (ig1 == 0) { g->xg[2] = (float)(180. / (nj - 1)); g->xg[0] = -90.f; }
        else if (ig1 == 1) { g->xg[2] = (float)(90. / (nj - 1)); g->xg[0] = 0.f; }
        else { g->xg[2] = (float)(90. / (nj - 1)); g->xg[0] = -90.f; }
        break;
    case 'E': case 'L':                                         /* ez_defxg.c:111-147 */
    case 'N': case 'S':                                         /* ez_defxg.c:150-160: xg = (pi, pj, d60, dgrw) */
        orc_cigaxg(grtyp, &g->xg[0], &g->xg[1], &g->xg[2], &g->xg[3], ig1, ig2, ig3, ig4);
        break;
    }
    if (grtyp == 'Z') {                                         /* ez_defaxes.c:43-54, ez_identifygrid.c:171-177 */
        for (int k = 0; k < 4; k++) g->igref[k] = g->ig[k];
        orc_cigaxg(grref, &g->xgref[0], &g->xgref[1], &g->xgref[2], &g->xgref[3], ig1, ig2, ig3, ig4);
        g->ax = (float *)malloc(sizeof(float) * ni); memcpy(g->ax, ax, sizeof(float) * ni);
        g->ay = (float *)malloc(sizeof(float) * nj); memcpy(g->ay, ay, sizeof(float) * nj);
    }
    if (grtyp == 'Y') {                                         /* ez_defaxes.c:58-65: ax = lon, ay = lat of every point */
        size_t n = (size_t)ni * nj;
        g->ax = (float *)malloc(sizeof(float) * n); memcpy(g->ax, ax, sizeof(float) * n);
        g->ay = (float *)malloc(sizeof(float) * n); memcpy(g->ay, ay, sizeof(float) * n);
    }
    if (grtyp == 'G') {                                         /* ez_defaxes.c:65-91 */
        g->grref = 'L';
        g->xgref[0] = 0.0f; g->xgref[1] = 0.0f; g->xgref[2] = 1.0f; g->xgref[3] = 1.0f;
        orc_cxgaig('L', &g->igref[0], &g->igref[1], &g->igref[2], &g->igref[3], 0.0f, 0.0f, 1.0f, 1.0f);
        g->ax = (float *)malloc(sizeof(float) * ni);
        float dlon = (float)(360. / (float)ni);
        for (int i = 0; i < ni; i++) g->ax[i] = (float)i * dlon;
        const int nlat = ig1 == 0 ? nj : 2 * nj;              /* ez_defaxes.c:86-99: a hemisphere keeps the latitudes of the whole globe */
        g->ay = (float *)malloc(sizeof(float) * nlat);
        orc_gauss_lat(g->ay, nlat);
    }
    xpncof(g);
    if (grtyp == 'Z' || grtyp == 'G') {
        int nnj = g->j2 - g->j1 + 1;
        g->ncx = (float *)malloc(sizeof(float) * 6 * ni);
        g->ncy = (float *)malloc(sizeof(float) * 6 * nnj);
        orc_nwtncof(g->ncx, g->ncy, g->ax, g->ay, ni, nj, g->j1, g->j2, g->extension);
    }
    return g;
}

void orc_grid_free(orc_grid *g)
{
    if (!g) return;
    free(g->ax); free(g->ay); free(g->ncx); free(g->ncy); free(g->lat); free(g->lon); free(g);
}

/* ------------------------------------------------------------------------------------------
 * Target lat/lon: ez_calclatlon.c:30-284; GRLL src/base/grll.f:57-64; c_llfgr ez_llfgr.c:25-33
 * ---------------------------------------------------------------------------------------- */
static void grll(float *lat, float *lon, int ni, int nj, float xla0, float xlo0, float dla0, float dlo0)
{
    for (int j = 1; j <= nj; j++) {
        float xla = xla0 + (float)(j - 1) * dla0;
        for (int i = 1; i <= ni; i++) {
            lat[(j - 1) * ni + i - 1] = xla;
            lon[(j - 1) * ni + i - 1] = fmodf(xlo0 + (float)(i - 1) * dlo0, 360.0f);
        }
    }
}

/* PERMUT, src/base/permut.f:30-46: the rows of z(ni, nj) in reverse order */
void orc_permut(float *z, int ni, int nj)
{
    for (int j = 0; j < nj / 2; j++)
        for (int i = 0; i < ni; i++) { float t = z[(size_t)(nj - 1 - j) * ni + i]; z[(size_t)(nj - 1 - j) * ni + i] = z[(size_t)j * ni + i]; z[(size_t)j * ni + i] = t; }
}

/* LLFXY, src/base/llfxy.F:21-40: all REAL arithmetic (RDTODG from pi.cdk) */
static void llfxy(float *dlat, float *dlon, float x, float y, float d60, float dgrw, int nhem)
{
    const float rdtodg = 57.295779513082f;
    float re = 1.866025f * 6.371E+6f / d60;
    float re2 = re * re;
    *dlat = 90.f; *dlon = 0.f;
    if (!(x == 0.f && y == 0.f)) {
        if (x == 0.f) *dlon = copysignf(90.f, y);
        if (x != 0.f) *dlon = atanf(y / x) * rdtodg;
        if (x < 0.f) *dlon = *dlon + copysignf(180.f, y);
        *dlon = *dlon - dgrw;
        if (*dlon > +180.f) *dlon = *dlon - 360.f;
        if (*dlon < -180.f) *dlon = *dlon + 360.f;
        float r2 = x * x + y * y;
        *dlat = (re2 - r2) / (re2 + r2);
        *dlat = asinf(*dlat) * rdtodg;
    }
    if (nhem == 2) *dlat = -*dlat;
    if (nhem == 2) *dlon = -*dlon;
}
/* GRPS, src/base/grps.f:43-58 */
static void grps(float *xlat, float *xlon, int ni, int nj, float pi, float pj, float d60, float dgrw, int hem)
{
    for (int j = 1; j <= nj; j++) {
        float y = (float)j - pj;
        for (int i = 1; i <= ni; i++) {
            float xla, xlo;
            llfxy(&xla, &xlo, (float)i - pi, y, d60, dgrw, hem);
            xlat[(j - 1) * ni + (i - 1)] = xla;
            if (xlo < 0) xlo = xlo + 360.f;
            xlon[(j - 1) * ni + (i - 1)] = xlo;
        }
    }
}

int orc_calclatlon(orc_grid *g)
{
    if (g->lat) return 0;
    int ni = g->ni, nj = g->nj, npts = ni * nj;
    g->lat = (float *)malloc(sizeof(float) * npts);
    g->lon = (float *)malloc(sizeof(float) * npts);
    switch (g->grtyp) {
    case 'A': case 'B': case 'L':                               /* :47-52, :73-77 */
        grll(g->lat, g->lon, ni, nj, g->xg[0], g->xg[1], g->xg[2], g->xg[3]);
        break;
    case 'E': {                                                 /* :54-71 */
        float dlon = (float)(360. / (ni - 1));
        float dlat = (float)(180. / (nj));
        float xlon00 = 0.0f;
        float xlat00 = (float)(-90. + 0.5 * (double)dlat);
        grll(g->lat, g->lon, ni, nj, xlat00, xlon00, dlat, dlon);
        float *latp = (float *)malloc(sizeof(float) * npts), *lonp = (float *)malloc(sizeof(float) * npts);
        orc_gfllfxy(lonp, latp, g->lon, g->lat, npts, g->xg[0], g->xg[1], g->xg[2], g->xg[3]);
        memcpy(g->lat, latp, sizeof(float) * npts); memcpy(g->lon, lonp, sizeof(float) * npts);
        free(latp); free(lonp);
        break;
    }
    case 'Y':                                                   /* :127-143, grref 'L' */
        memcpy(g->lon, g->ax, sizeof(float) * npts); memcpy(g->lat, g->ay, sizeof(float) * npts);
        break;
    case 'N': case 'S':                                         /* :83-97 -> GRPS (src/base/grps.f:43-58) */
        grps(g->lat, g->lon, ni, nj, g->xg[0], g->xg[1], g->xg[2], g->xg[3], g->grtyp == 'N' ? 1 : 2);
        break;
    case 'Z': case 'G':                                         /* :169-226 */
        for (int j = 0; j < nj; j++)
            for (int i = 0; i < ni; i++) { g->lat[j * ni + i] = g->ay[j]; g->lon[j * ni + i] = g->ax[i]; }
        if (g->grtyp == 'G' && g->ig[0] == 1)                  /* ez_calclatlon.c:167-176 */
            for (int j = 0; j < nj; j++) for (int i = 0; i < ni; i++) g->lat[j * ni + i] = g->ay[j + nj];
        if (g->grref == 'L') {
            for (int k = 0; k < npts; k++) { g->lat[k] += 1.0; g->lon[k] += 1.0; }
            for (int k = 0; k < npts; k++) {                    /* c_llfgr(lat, lon, x=lon, y=lat, ...) */
                float x = g->lon[k], y = g->lat[k];
                float lo = (float)((double)g->xgref[1] + (double)g->xgref[3] * ((double)x - 1.0));
                lo = (float)fmod(fmod((double)lo, 360.0) + 360.0, 360.0);
                g->lon[k] = lo;
                g->lat[k] = (float)((double)g->xgref[0] + (double)g->xgref[2] * ((double)y - 1.0));
            }
        } else if (g->grref == 'E') {
            float *latp = (float *)malloc(sizeof(float) * npts), *lonp = (float *)malloc(sizeof(float) * npts);
            orc_gfllfxy(lonp, latp, g->lon, g->lat, npts, g->xgref[0], g->xgref[1], g->xgref[2], g->xgref[3]);
            memcpy(g->lat, latp, sizeof(float) * npts); memcpy(g->lon, lonp, sizeof(float) * npts);
            free(latp); free(lonp);
        }
        break;
    default: return -1;
    }
    if ((g->grtyp == 'A' || g->grtyp == 'B' || g->grtyp == 'G') && g->ig[1] == 1) orc_permut(g->lat, ni, nj);      /* :268-279 */
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Locate
 * ---------------------------------------------------------------------------------------- */

/* ez_cherche, src/interp/ez_cherche.inc:53-69 (tab 1-based) */
static int cherche(float val, const float *tab, int n)
{
    int debut = 1, fin = n;
    int milieu = (int)((float)(debut + fin) * 0.5f);
    while (milieu != debut) {
        if (val <= tab[milieu - 1]) fin = milieu; else debut = milieu;
        milieu = (int)((float)(debut + fin) * 0.5f);
    }
    return milieu;
}

/* ez_llll2gd, src/interp/ez_llll2gd.inc:22-53 (mutates dlon) */
static void llll2gd(float *x, float *y, const float *dlat, float *dlon, int npts,
                    float xlat0, float xlon0, float dellat, float dellon, float lonref)
{
    if (lonref == -180.0f) { for (int i = 0; i < npts; i++) if (dlon[i] > 180.0f) dlon[i] = dlon[i] - 360.0f; }
    else                   { for (int i = 0; i < npts; i++) if (dlon[i] < 0.0f) dlon[i] = dlon[i] + 360.0f; }
    for (int i = 0; i < npts; i++) {
        x[i] = (dlon[i] - xlon0) / dellon + 1.0f;
        y[i] = (dlat[i] - xlat0) / dellat + 1.0f;
    }
}

/* ez_ll2igd, src/interp/ez_ll2igd.inc:22-86 (coordflag = RELATIF) */
/* nj: the number of entries of ay the search runs over -- gr.nj in ez_calcxy.c:49, gr.j2 in gdxyfll.c:143 (they differ for a hemispheric
 * 'G' grid, whose table holds the 2 nj latitudes of the globe); a northern hemisphere then has nj subtracted (ez_calcxy.c:90-96,
 * gdxyfll.c:149-153): its points are sought among the SOUTHERN latitudes and land beyond the end of that half, extrapolated */
static void ll2igd(float *px, float *py, const float *xlat, float *xlon, int npts, const orc_grid *g, int nj)
{
    int ni = g->ni;
    if (g->grref == 'L') {
        float xlat0, xlon0, dlat, dlon;
        orc_cigaxg('L', &xlat0, &xlon0, &dlat, &dlon, g->igref[0], g->igref[1], g->igref[2], g->igref[3]);
        float lonref = (g->ax[0] < 0.0f) ? -180.0f : 0.0f;
        llll2gd(px, py, xlat, xlon, npts, xlat0, xlon0, dlat, dlon, lonref);
        for (int i = 0; i < npts; i++) { px[i] = px[i] - 1.0f; py[i] = py[i] - 1.0f; }
    } else if (g->grref == 'E') {
        float xlat1, xlon1, xlat2, xlon2;
        orc_cigaxg('E', &xlat1, &xlon1, &xlat2, &xlon2, g->igref[0], g->igref[1], g->igref[2], g->igref[3]);
        orc_gfxyfll(px, py, xlon, xlat, npts, xlat1, xlon1, xlat2, xlon2);
    }
    for (int i = 0; i < npts; i++) {                            /* :74-85 */
        int indx = cherche(px[i], g->ax, ni);
        int indy = cherche(py[i], g->ay, nj);
        if (indx >= ni) indx = ni - 1;
        if (indy >= nj) indy = nj - 1;
        px[i] = (float)indx + (px[i] - g->ax[indx - 1]) / (g->ax[indx] - g->ax[indx - 1]);
        py[i] = (float)indy + (py[i] - g->ay[indy - 1]) / (g->ay[indy] - g->ay[indy - 1]);
    }
    if (g->grtyp == 'G' && g->ig[0] == 1) for (int i = 0; i < npts; i++) py[i] = py[i] - (float)nj;
}

/* ez_ll2rgd, src/interp/ez_ll2rgd.inc:22-166 (A, B, L, E branches; mutates xlon) */
/* ez_vxyfll, src/interp/ez_vxyfll.inc:20-99.  dgtord is REAL: `dgtord*(dlon+dgrw)` and `dgtord*dlat` are REAL products
 * widened on assignment; in the SUD branch rlon is already double when it meets dgtord. */
static void vxyfll(float *x, float *y, const float *dlat, const float *dlon, int npts, float d60, float dgrw, float pi, float pj, int nhem)
{
    const float dgtord = 1.7453292519943e-2f;
    double re = 1.866025 * 6.371e+6 / (double)d60;
    for (int i = 0; i < npts; i++) {
        double rlon, rlat;
        if (nhem == 1) {
            rlon = (double)(float)(dgtord * (float)(dlon[i] + dgrw));
            rlat = (double)(float)(dgtord * dlat[i]);
        } else {
            rlon = (double)dlon[i];
            if (rlon > 180.0) rlon = rlon - 360.0;
            rlon = (double)dgtord * (-rlon + (double)dgrw);
            rlat = (double)(float)(dgtord * (-dlat[i]));
        }
        double sinlat = sin(rlat);
        double r = re * sqrt((1.0 - sinlat) / (1.0 + sinlat));
        x[i] = (float)(r * cos(rlon) + (double)pi);
        y[i] = (float)(r * sin(rlon) + (double)pj);
    }
}

static int ll2rgd(float *px, float *py, const float *xlat, float *xlon, int npts, const orc_grid *g)
{
    int ni = g->ni, nj = g->nj;
    float dellat, dellon, xlat0, xlon0;
    switch (g->grtyp) {
    case 'A':                                                   /* :58-78 */
        dellon = 360.0f / (float)ni; xlon0 = 0.0f;
        if (g->ig[0] == 1) { dellat = 90.0f / (float)nj; xlat0 = dellat * 0.5f; }
        else if (g->ig[0] == 2) { dellat = 90.0f / (float)nj; xlat0 = -90.0f + dellat * 0.5f; }
        else { dellat = 180.0f / (float)nj; xlat0 = -90.0f + dellat * 0.5f; }
        for (int i = 0; i < npts; i++) if (xlon[i] < 0.0f) xlon[i] = xlon[i] + 360.0f;
        llll2gd(px, py, xlat, xlon, npts, xlat0, xlon0, dellat, dellon, 0.0f);
        return 0;
    case 'B':                                                   /* :80-100 */
        dellon = 360.0f / (float)(ni - 1); xlon0 = 0.0f;
        if (g->ig[0] == 1) { dellat = 90.0f / (float)(nj - 1); xlat0 = 0.0f; }
        else if (g->ig[0] == 2) { dellat = 90.0f / (float)(nj - 1); xlat0 = -90.0f; }
        else { dellat = 180.0f / (float)(nj - 1); xlat0 = -90.0f; }
        for (int i = 0; i < npts; i++) if (xlon[i] < 0.0f) xlon[i] = xlon[i] + 360.0f;
        llll2gd(px, py, xlat, xlon, npts, xlat0, xlon0, dellat, dellon, 0.0f);
        return 0;
    case 'L':                                                   /* :134-149 */
        orc_cigaxg('L', &xlat0, &xlon0, &dellat, &dellon, g->ig[0], g->ig[1], g->ig[2], g->ig[3]);
        for (int i = 0; i < npts; i++) {
            if (xlon[i] < xlon0) xlon[i] = xlon[i] + 360.0f;
            if (xlon[i] > (xlon0 + (float)ni * dellon)) xlon[i] = xlon[i] - 360.0f;
        }
        llll2gd(px, py, xlat, xlon, npts, xlat0, xlon0, dellat, dellon, 0.0f);
        return 0;
    case 'N': case 'S': {                                       /* ez_ll2rgd.inc:43-55 */
        float pi, pj, d60, dgrw;
        orc_cigaxg(g->grtyp, &pi, &pj, &d60, &dgrw, g->ig[0], g->ig[1], g->ig[2], g->ig[3]);
        vxyfll(px, py, xlat, xlon, npts, d60, dgrw, pi, pj, g->grtyp == 'N' ? 1 : 2);
        return 0;
    }
    case 'E': {                                                 /* ez_ll2ergd.inc:37-48 */
        float xlat1, xlon1, xlat2, xlon2;
        orc_cigaxg('E', &xlat1, &xlon1, &xlat2, &xlon2, g->ig[0], g->ig[1], g->ig[2], g->ig[3]);
        float *latgf = (float *)malloc(sizeof(float) * npts), *longf = (float *)malloc(sizeof(float) * npts);
        orc_gfxyfll(longf, latgf, xlon, xlat, npts, xlat1, xlon1, xlat2, xlon2);
        dellon = 360.0f / (float)(ni - 1); xlon0 = 0.0f;
        dellat = 180.0f / (float)nj; xlat0 = -90.f + 0.5f * dellat;
        llll2gd(px, py, latgf, longf, npts, xlat0, xlon0, dellat, dellon, 0.0f);
        free(latgf); free(longf);
        return 0;
    }
    }
    return -1;
}

/* c_gdxyfll_orig, src/interp/gdxyfll.c:90-139.  A/B/E/L work on a copy of lon; G/Z mutate it. */
int orc_gdxyfll(const orc_grid *g, float *x, float *y, const float *lat, float *lon, int n)
{
    switch (g->grtyp) {
    case 'A': case 'B': case 'E': case 'L': case 'N': case 'S': {
        float *tmplons = (float *)malloc(sizeof(float) * n);
        memcpy(tmplons, lon, sizeof(float) * n);
        int rc = ll2rgd(x, y, lat, tmplons, n, g);
        free(tmplons);
        return rc;
    }
    case 'Z': case 'G':
        ll2igd(x, y, lat, lon, n, g, g->j2);
        return 0;
    }
    return -1;
}

/* ------------------------------------------------------------------------------------------
 * Point kernels.  z is (ni, j1:j2) column-major, 1-based i.
 * ---------------------------------------------------------------------------------------- */
#define Z(i, j) z[(size_t)((j) - j1) * ni + ((i) - 1)]
#define AX(i) ax[(i) - 1]
#define AY(j) ay[(j) - j1]

/* src/interp/ez_rgdint_0.inc:20-35 */
void orc_rgdint_0(float *zo, const float *px, const float *py, int npts, const float *z, int ni, int j1, int j2)
{
    for (int n = 0; n < npts; n++) {
        int i = imin(ni, imax(1, (int)lroundf(px[n])));
        int j = imin(j2, imax(j1, (int)lroundf(py[n])));
        zo[n] = Z(i, j);
    }
}

static inline double zlin(double zz1, double zz2, double zdx) { return zz1 + (zz2 - zz1) * zdx; }  /* zlin8.cdk:1-3 */

/* src/interp/ez_rgdint_1_nw.inc:20-44 */
void orc_rgdint_1_nw(float *zo, const float *px, const float *py, int npts, const float *z, int ni, int j1, int j2)
{
    for (int n = 0; n < npts; n++) {
        int i = imin(ni - 1, imax(1, (int)px[n]));
        int j = imin(j2 - 1, imax(j1, (int)py[n]));
        double dx = (double)(px[n] - (float)i);
        double dy = (double)(py[n] - (float)j);
        double y2 = zlin((double)Z(i, j), (double)Z(i + 1, j), dx);
        double y3 = zlin((double)Z(i, j + 1), (double)Z(i + 1, j + 1), dx);
        zo[n] = (float)zlin(y2, y3, dy);
    }
}

/* src/interp/ez_rgdint_1_w.inc:20-51 */
void orc_rgdint_1_w(float *zo, const float *px, const float *py, int npts, const float *z, int ni, int j1, int j2, int wrap)
{
    int limite = ni + 2 - wrap;
    for (int n = 0; n < npts; n++) {
        int i = imin(ni - 2 + wrap, imax(1, (int)px[n]));
        int j = imin(j2 - 1, imax(j1, (int)py[n]));
        int iplus1 = i + 1;
        if (wrap > 0 && (i == (ni - 2 + wrap))) iplus1 = (limite + i + 1) % limite;
        double dx = (double)(px[n] - (float)i);
        double dy = (double)(py[n] - (float)j);
        double y2 = zlin((double)Z(i, j), (double)Z(iplus1, j), dx);
        double y3 = zlin((double)Z(i, j + 1), (double)Z(iplus1, j + 1), dx);
        zo[n] = (float)zlin(y2, y3, dy);
    }
}

/* src/interp/ez_irgdint_1_nw.inc:20-50 (j1 = 1, j2 = nj) */
void orc_irgdint_1_nw(float *zo, const float *px, const float *py, int npts, const float *ax, const float *ay, const float *z, int ni, int nj)
{
    const int j1 = 1;
    for (int n = 0; n < npts; n++) {
        int i = imin(ni - 1, imax(1, (int)px[n]));
        int j = imin(nj - 1, imax(1, (int)py[n]));
        double x1 = AX(i), x2 = AX(i + 1);
        double x = (double)AX(i) + (x2 - x1) * (double)(px[n] - (float)i);
        double y = (double)(AY(j) + (AY(j + 1) - AY(j)) * (py[n] - (float)j));
        double dx = (x - x1) / (x2 - x1);
        double dy = (y - (double)AY(j)) / (double)(AY(j + 1) - AY(j));
        double y1 = zlin((double)Z(i, j), (double)Z(i + 1, j), dx);
        double y2 = zlin((double)Z(i, j + 1), (double)Z(i + 1, j + 1), dx);
        zo[n] = (float)zlin(y1, y2, dy);
    }
}

/* src/interp/ez_irgdint_1_w.inc:20-64 */
void orc_irgdint_1_w(float *zo, const float *px, const float *py, int npts, const float *ax, const float *ay, const float *z, int ni, int j1, int j2, int wrap)
{
    double x2 = 0.0;   /* the reference leaves x2 from the previous point when iplus1 > ni and wrap == 0 */
    for (int n = 0; n < npts; n++) {
        int limite = ni + 2 - wrap;
        int i = imin(ni - 2 + wrap, imax(1, (int)px[n]));
        int j = imin(j2 - 1, imax(j1 + 1, (int)py[n]));
        if (j < 0) j = j - 1;
        int iplus1 = i + 1;
        double x1 = AX(i);
        if (iplus1 <= ni) x2 = AX(iplus1);
        if (wrap > 0 && (i == (ni - 2 + wrap))) {
            iplus1 = (limite + i + 1) % limite;
            x2 = (double)(AX(2) + AX(ni));
        }
        double x = x1 + (x2 - x1) * (double)(px[n] - (float)i);
        double y = (double)(AY(j) + (AY(j + 1) - AY(j)) * (py[n] - (float)j));
        double dx = (x - x1) / (x2 - x1);
        double dy = (y - (double)AY(j)) / (double)(AY(j + 1) - AY(j));
        double y1 = zlin((double)Z(i, j), (double)Z(iplus1, j), dx);
        double y2 = zlin((double)Z(i, j + 1), (double)Z(iplus1, j + 1), dx);
        zo[n] = (float)zlin(y1, y2, dy);
    }
}

/* cubic8.cdk:1-3.  The literals 0.1666666666666 / 0.3333333333333 are default-REAL constants in
 * Fortran, i.e. float values widened to double. */
static inline double cubic(double z1, double z2, double z3, double z4, double dx)
{
    const double c6 = (double)0.1666666666666f, c3 = (double)0.3333333333333f;
    return ((((z4 - z1) * c6 + 0.5 * (z2 - z3)) * dx + 0.5 * (z1 + z3) - z2) * dx + z3 - c6 * z4 - 0.5 * z2 - c3 * z1) * dx + z2;
}

#define CUBIC_ROWS(im1, i, ip1, ip2)                                                                    \
    double y1 = cubic((double)Z(im1, j - 1), (double)Z(i, j - 1), (double)Z(ip1, j - 1), (double)Z(ip2, j - 1), dx); \
    double y2 = cubic((double)Z(im1, j), (double)Z(i, j), (double)Z(ip1, j), (double)Z(ip2, j), dx);                 \
    double y3 = cubic((double)Z(im1, j + 1), (double)Z(i, j + 1), (double)Z(ip1, j + 1), (double)Z(ip2, j + 1), dx); \
    double y4 = cubic((double)Z(im1, j + 2), (double)Z(i, j + 2), (double)Z(ip1, j + 2), (double)Z(ip2, j + 2), dx); \
    zo[n] = (float)cubic(y1, y2, y3, y4, dy);

/* src/interp/ez_rgdint_3_nw.inc:20-77 */
void orc_rgdint_3_nw(float *zo, const float *px, const float *py, int npts, const float *z, int ni, int j1, int j2)
{
    for (int n = 0; n < npts; n++) {
        int i = imin(ni - 2, imax(2, (int)px[n]));
        int j = imin(j2 - 2, imax(j1 + 1, (int)py[n]));
        double dx = (double)(px[n] - (float)i);
        double dy = (double)(py[n] - (float)j);
        CUBIC_ROWS(i - 1, i, i + 1, i + 2)
    }
}

static inline void wrap_cols_regular(int ni, int wrap, int limite, int *i, int *im1, int *ip1, int *ip2)
{   /* ez_rgdint_3_w.inc:72-90 */
    *im1 = (limite + *i - 1) % limite;
    *ip1 = (limite + *i + 1) % limite;
    *ip2 = (limite + *i + 2) % limite;
    if (*im1 == 0) *im1 = ni;
    if (*i == 0) *i = ni;
    if (*ip1 == 0) *ip1 = ni;
    if (*ip2 == 0) *ip2 = ni;
    if (wrap == 1) {
        if (*ip2 == ni) *ip2 = 2;
        if (*im1 == ni) *im1 = ni - 1;
    }
}

/* src/interp/ez_rgdint_3_w.inc:20-108 */
void orc_rgdint_3_w(float *zo, const float *px, const float *py, int npts, const float *z, int ni, int j1, int j2, int wrap)
{
    int limite = ni + 2 - wrap;
    for (int n = 0; n < npts; n++) {
        int i = imin(ni - 2 + wrap, imax(1, imax(2 - wrap, (int)px[n])));
        int j = imin(j2 - 2, imax(j1 + 1, (int)py[n]));
        int im1, ip1, ip2;
        if (wrap > 0) wrap_cols_regular(ni, wrap, limite, &i, &im1, &ip1, &ip2);
        else { im1 = i - 1; ip1 = i + 1; ip2 = i + 2; }
        double dx = (double)(px[n] - (float)i);
        double dy = (double)(py[n] - (float)j);
        CUBIC_ROWS(im1, i, ip1, ip2)
    }
}

/* src/interp/ez_rgdint_3_wnnc.inc:20-107 */
void orc_rgdint_3_wnnc(float *zo, const float *px, const float *py, int npts, const float *z, int ni, int j1, int j2, int wrap)
{
    int limite = ni + 2 - wrap;
    for (int n = 0; n < npts; n++) {
        int i = imin(ni - 2 + wrap, imax(1, imax(2 - wrap, (int)px[n])));
        int j = imin(j2 - 2, imax(j1 + 1, (int)py[n]));
        int im1, ip1, ip2;
        if ((wrap > 0 && i <= 1) || i >= (ni - 1)) wrap_cols_regular(ni, wrap, limite, &i, &im1, &ip1, &ip2);
        else { im1 = i - 1; ip1 = i + 1; ip2 = i + 2; }
        double dx = (double)(px[n] - (float)i);
        double dy = (double)(py[n] - (float)j);
        CUBIC_ROWS(im1, i, ip1, ip2)
    }
}

/* Newton-form statement functions, ez_irgdint_3_w.inc:86-89 */
static inline double fa(double a1, double a2, double a3, double a4, double x, double x1, double x2, double x3)
{ return a1 + (x - x1) * (a2 + (x - x2) * (a3 + a4 * (x - x3))); }
static inline double fa2(double c1, double a1, double a2) { return c1 * (a2 - a1); }
static inline double fa3(double c1, double c2, double c3, double a1, double a2, double a3)
{ return c2 * (c3 * (a3 - a2) - c1 * (a2 - a1)); }
static inline double fa4(double c1, double c2, double c3, double c4, double c5, double c6, double a1, double a2, double a3, double a4)
{ return c4 * (c5 * (c6 * (a4 - a3) - c3 * (a3 - a2)) - c2 * (c3 * (a3 - a2) - c1 * (a2 - a1))); }

/* src/interp/ez_irgdint_3_nw.inc:20-168.  fa..fa4 are declared REAL there: every statement-function
 * result is rounded to float (:32). */
void orc_irgdint_3_nw(float *zo, const float *px, const float *py, int npts, const float *ax, const float *ay,
                      const float *cx, const float *cy, const float *z, int i1, int i2, int j1, int j2)
{
    const int ni = i2 - i1 + 1, nnj = j2 - j1 + 1;
#define AXI(i) ax[(i) - i1]
#define ZI(i, j) z[(size_t)((j) - j1) * ni + ((i) - i1)]
#define CXI(i, k) (double)cx[((k) - 1) * ni + ((i) - i1)]
#define CYJ(j, k) (double)cy[((k) - 1) * nnj + ((j) - j1)]
#define R(e) ((double)(float)(e))
    for (int n = 0; n < npts; n++) {
        int i = imin(i2 - 2, imax(i1 + 1, (int)px[n]));
        int j = imin(j2 - 2, imax(j1 + 1, (int)py[n]));
        double x = (double)(AXI(i) + (AXI(i + 1) - AXI(i)) * (px[n] - (float)i));
        double y = (double)(AY(j) + (AY(j + 1) - AY(j)) * (py[n] - (float)j));
        double x1 = AXI(i - 1), x2 = AXI(i), x3 = AXI(i + 1);
        double y1 = AY(j - 1), y2 = AY(j), y3 = AY(j + 1);
        double b[4];
        for (int r = 0; r < 4; r++) {
            int jj = j - 1 + r;
            double z1 = ZI(i - 1, jj), z2 = ZI(i, jj), z3 = ZI(i + 1, jj), z4 = ZI(i + 2, jj);
            double a1 = z1;
            double a2 = R(fa2(CXI(i, 1), z1, z2));
            double a3 = R(fa3(CXI(i, 1), CXI(i, 2), CXI(i, 3), z1, z2, z3));
            double a4 = R(fa4(CXI(i, 1), CXI(i, 2), CXI(i, 3), CXI(i, 4), CXI(i, 5), CXI(i, 6), z1, z2, z3, z4));
            b[r] = R(fa(a1, a2, a3, a4, x, x1, x2, x3));
        }
        double b11 = b[0];
        double b12 = R(fa2(CYJ(j, 1), b[0], b[1]));
        double b13 = R(fa3(CYJ(j, 1), CYJ(j, 2), CYJ(j, 3), b[0], b[1], b[2]));
        double b14 = R(fa4(CYJ(j, 1), CYJ(j, 2), CYJ(j, 3), CYJ(j, 4), CYJ(j, 5), CYJ(j, 6), b[0], b[1], b[2], b[3]));
        zo[n] = (float)fa(b11, b12, b13, b14, y, y1, y2, y3);
    }
#undef AXI
#undef ZI
#undef CXI
#undef R
}

/* seam handling shared by ez_irgdint_3_w.inc:100-156 and ez_irgdint_3_wnnc.inc:100-156.
 * Returns the four abscissae as FLOAT values (each is computed by an all-REAL expression).
 * ip2_wrap1 is the column used for iplus2 at (wrap==1, i==ni-1): 2 in _3_w, 1 in _3_wnnc. */
static inline void irr_cols(const float *ax, int ni, int wrap, int i, int ip2_wrap1,
                            int *im1, int *ip1, int *ip2, float *x1, float *x2, float *x3, float *x4)
{
    *im1 = i - 1; *ip1 = i + 1; *ip2 = i + 2;
    if (wrap == 1 && (i <= 1 || i >= (ni - wrap))) {
        if (i == 1) { *im1 = ni - 1; *ip1 = 2; *ip2 = 3; *x1 = AX(ni - 1) - 360.0f; *x2 = AX(1); *x3 = AX(2); *x4 = AX(3); }
        if (i == (ni - 1)) { *im1 = ni - 2; *ip1 = ni; *ip2 = ip2_wrap1; *x1 = AX(ni - 2); *x2 = AX(ni - 1); *x3 = AX(ni); *x4 = AX(2) + 360.0f; }
    } else if (wrap == 2 && (i <= 1 || i > (ni - wrap))) {
        if (i == 1) { *im1 = ni; *ip1 = 2; *ip2 = 3; *x1 = AX(ni) - 360.0f; *x2 = AX(1); *x3 = AX(2); *x4 = AX(3); }
        if (i == (ni - 1)) { *im1 = ni - 2; *ip1 = ni; *ip2 = 1; *x1 = AX(ni - 2); *x2 = AX(ni - 1); *x3 = AX(ni); *x4 = AX(1) + 360.0f; }
        if (i == ni) { *im1 = ni - 1; *ip1 = 1; *ip2 = 2; *x1 = AX(ni - 1); *x2 = AX(ni); *x3 = AX(1) + 360.0f; *x4 = AX(2) + 360.0f; }
    } else {
        *x1 = AX(*im1); *x2 = AX(i); *x3 = AX(*ip1); *x4 = AX(*ip2);
    }
}

/* src/interp/ez_irgdint_3_w.inc:20-235 -- THE cfg2 kernel.  x1..x4, y1..y4 are REAL here (:75). */
void orc_irgdint_3_w(float *zo, const float *px, const float *py, int npts, const float *ax, const float *ay,
                     const float *cx, const float *cy, const float *z, int ni, int j1, int j2, int wrap)
{
    const int nnj = j2 - j1 + 1;
#define CX(i, k) (double)cx[((k) - 1) * ni + ((i) - 1)]
    float x1 = 0, x2 = 0, x3 = 0, x4 = 0;
    for (int n = 0; n < npts; n++) {
        int i = imin(ni - 2 + wrap, imax(1, imax(2 - wrap, (int)px[n])));
        int j = imin(j2 - 2, imax(j1 + 1, (int)py[n]));
        int im1, ip1, ip2;
        irr_cols(ax, ni, wrap, i, 2, &im1, &ip1, &ip2, &x1, &x2, &x3, &x4);
        double x = (double)(x2 + (x3 - x2) * (px[n] - (float)i));
        double y = (double)(AY(j) + (AY(j + 1) - AY(j)) * (py[n] - (float)j));
        float y1 = AY(j - 1), y2 = AY(j), y3 = AY(j + 1);
        double b[4];
        for (int r = 0; r < 4; r++) {
            int jj = j - 1 + r;
            double z1 = Z(im1, jj), z2 = Z(i, jj), z3 = Z(ip1, jj), z4 = Z(ip2, jj);
            double a2 = fa2(CX(i, 1), z1, z2);
            double a3 = fa3(CX(i, 1), CX(i, 2), CX(i, 3), z1, z2, z3);
            double a4 = fa4(CX(i, 1), CX(i, 2), CX(i, 3), CX(i, 4), CX(i, 5), CX(i, 6), z1, z2, z3, z4);
            b[r] = fa(z1, a2, a3, a4, x, (double)x1, (double)x2, (double)x3);
        }
        double b12 = fa2(CYJ(j, 1), b[0], b[1]);
        double b13 = fa3(CYJ(j, 1), CYJ(j, 2), CYJ(j, 3), b[0], b[1], b[2]);
        double b14 = fa4(CYJ(j, 1), CYJ(j, 2), CYJ(j, 3), CYJ(j, 4), CYJ(j, 5), CYJ(j, 6), b[0], b[1], b[2], b[3]);
        zo[n] = (float)fa(b[0], b12, b13, b14, y, (double)y1, (double)y2, (double)y3);
    }
#undef CX
}
#undef CYJ

/* src/interp/ez_irgdint_3_wnnc.inc:20-246.  x1..x4, y1..y4 are REAL*8 here (:76); the six reciprocals
 * are recomputed per point in double (:165-170, :231-236). */
void orc_irgdint_3_wnnc(float *zo, const float *px, const float *py, int npts, const float *ax, const float *ay,
                        const float *z, int ni, int j1, int j2, int wrap)
{
    float fx1 = 0, fx2 = 0, fx3 = 0, fx4 = 0;
    for (int n = 0; n < npts; n++) {
        int i = imin(ni - 2 + wrap, imax(1, imax(2 - wrap, (int)px[n])));
        int j = imin(j2 - 2, imax(j1 + 1, (int)py[n]));
        int im1, ip1, ip2;
        irr_cols(ax, ni, wrap, i, 1, &im1, &ip1, &ip2, &fx1, &fx2, &fx3, &fx4);
        double x1 = fx1, x2 = fx2, x3 = fx3, x4 = fx4;
        double x = x2 + (x3 - x2) * (double)(px[n] - (float)i);
        double y = (double)(AY(j) + (AY(j + 1) - AY(j)) * (py[n] - (float)j));
        double c1 = 1.0 / (x2 - x1), c2 = 1.0 / (x3 - x1), c3 = 1.0 / (x3 - x2);
        double c4 = 1.0 / (x4 - x1), c5 = 1.0 / (x4 - x2), c6 = 1.0 / (x4 - x3);
        double y1 = AY(j - 1), y2 = AY(j), y3 = AY(j + 1), y4 = AY(j + 2);
        double b[4];
        for (int r = 0; r < 4; r++) {
            int jj = j - 1 + r;
            double z1 = Z(im1, jj), z2 = Z(i, jj), z3 = Z(ip1, jj), z4 = Z(ip2, jj);
            double a2 = fa2(c1, z1, z2);
            double a3 = fa3(c1, c2, c3, z1, z2, z3);
            double a4 = fa4(c1, c2, c3, c4, c5, c6, z1, z2, z3, z4);
            b[r] = fa(z1, a2, a3, a4, x, x1, x2, x3);
        }
        double d1 = 1.0 / (y2 - y1), d2 = 1.0 / (y3 - y1), d3 = 1.0 / (y3 - y2);
        double d4 = 1.0 / (y4 - y1), d5 = 1.0 / (y4 - y2), d6 = 1.0 / (y4 - y3);
        double b12 = fa2(d1, b[0], b[1]);
        double b13 = fa3(d1, d2, d3, b[0], b[1], b[2]);
        double b14 = fa4(d1, d2, d3, d4, d5, d6, b[0], b[1], b[2], b[3]);
        zo[n] = (float)fa(b[0], b12, b13, b14, y, y1, y2, y3);
    }
}
#undef Z
#undef AX
#undef AY

/* ------------------------------------------------------------------------------------------
 * interp_degree = "average" (degre_interp 4, gdinterp.c:181-187 / :283-289 -> ez_avg, src/interp/ez_avg.inc:1-190): every target
 * cell -- bounded half way to its neighbours in the located coordinates of the FIRST target row / column (x = xx(:,1), y = yy(1,:),
 * :55-56) -- averages the source cells it covers, weighted by the covered fraction; REAL arithmetic, cells added in row-major order.
 * The source is addressed through ztmp, the field with its columns repeated over [-ni, 2 ni] according to the grid's wrap (:19-52);
 * outside that (a target that leaves a regional source) the reference indexes beyond the array: clamped here.
 * ---------------------------------------------------------------------------------------- */
static float avg_src(const float *zin, int ni, int nj, int ext, int ii, int jj)
{
    if (jj < 1) jj = 1;
    if (jj > nj) jj = nj;
    if (ext == 1) { int k = ii; while (k < 1) k += ni - 1; while (k > ni - 1) k -= ni - 1; ii = k; }
    else if (ext == 2) { int k = ii; while (k < 1) k += ni; while (k > ni) k -= ni; ii = k; }
    else { if (ii < 1) ii = 1; if (ii > ni) ii = ni; }
    return zin[(size_t)(jj - 1) * ni + (ii - 1)];
}
/* the bounds of the target cells (:55-78): x_low / x_high [ni_dst], y_low / y_high [nj_dst] */
void orc_avg_bounds(float *x_low, float *x_high, float *y_low, float *y_high, const float *xx, const float *yy, int nid, int njd, int nis, int njs)
{
    float *x = (float *)malloc(sizeof(float) * nid), *y = (float *)malloc(sizeof(float) * njd);
    for (int i = 0; i < nid; i++) x[i] = xx[i];
    for (int j = 0; j < njd; j++) y[j] = yy[(size_t)j * nid];
    if (x[0] > (float)(nis - 1)) x[0] = 1.0f;
    x_low[0] = x[0] - 0.5f * (x[1] - x[0]);
    for (int i = 1; i < nid; i++) x_low[i] = x[i] - 0.5f * (x[i] - x[i - 1]);
    y_low[0] = fmaxf(1.0f, y[0] - 0.5f * (y[1] - y[0]));
    for (int j = 1; j < njd; j++) y_low[j] = y[j] - 0.5f * (y[j] - y[j - 1]);
    x_high[nid - 1] = x[nid - 1] + 0.5f * (x[nid - 1] - x[nid - 2]);
    for (int i = 0; i < nid - 1; i++) x_high[i] = x[i] + 0.5f * (x[i + 1] - x[i]);
    y_high[njd - 1] = fminf(1.0f * (float)njs, y[njd - 1] + 0.5f * (y[njd - 1] - y[njd - 2]));
    for (int j = 0; j < njd - 1; j++) y_high[j] = y[j] + 0.5f * (y[j + 1] - y[j]);
    free(x); free(y);
}
/* one target cell; row: 0 first row (:121-166), 1 middle rows (:81-117), 2 last row (:171-211).  ylast = y(nj_dst) (the last row's jstart) */
float orc_avg_cell(const float *zin, int nis, int njs, int ext, int row, float xl, float xh, float yl, float yh, float ylast)
{
    int jstart, jend = (int)lroundf(yh), istart = (int)xl, iend = (int)lroundf(xh);
    if (row == 2) jstart = (int)ylast;
    else {
        jstart = (int)yl;
        if (row == 1) { if ((0.5f + (float)jstart) < yl) jstart = jstart + 1; }
        else { if ((float)jstart > yl) jstart = jstart - 1; }
    }
    if ((0.5f + (float)istart) < xl) istart = istart + 1;
    if (row == 0 && (float)iend < xh) iend = iend + 1;
    float z = 0.0f, total = 0.0f;
    for (int jj = jstart; jj <= jend; jj++) {
        float ymin = (float)jj - 0.5f, ymax = (float)jj + 0.5f, yfrac = 1.0f;
        if (row == 0) { if (jj == 1) ymin = 1.0f; yfrac = ymax - ymin; }
        if (row == 2) { if (jj == njs) ymax = (float)njs; yfrac = ymax - ymin; }
        if (ymin < yl) yfrac = ymax - yl;
        if (ymax > yh) yfrac = yh - ymin;
        for (int ii = istart; ii <= iend; ii++) {
            float xmin = (float)ii - 0.5f, xmax = (float)ii + 0.5f, xfrac = 1.0f;
            if (xmin < xl) xfrac = xmax - xl;
            if (xmax > xh) xfrac = xh - xmin;
            float area = xfrac * yfrac;
            total = total + area;
            float prod = avg_src(zin, nis, njs, ext, ii, jj) * area;
            z = z + prod;
        }
    }
    if (total != 0.0f) z = z / total;
    return z;
}
/* interp_degree = "sph_average" (ez_avg_sph.inc): ez_avg with the x bounds of every target row widened by 1 / cos(latitude of the row) */
void orc_ez_avg_sph(float *zout, const float *xx, const float *yy, const float *lats_dst, int nid, int njd, const float *zin, int nis, int njs, int ext)
{
    float *xl = (float *)malloc(sizeof(float) * nid), *xh = (float *)malloc(sizeof(float) * nid);
    float *yl = (float *)malloc(sizeof(float) * njd), *yh = (float *)malloc(sizeof(float) * njd), *x = (float *)malloc(sizeof(float) * nid);
    orc_avg_bounds(xl, xh, yl, yh, xx, yy, nid, njd, nis, njs);               /* y bounds as ez_avg (:83-86, :95-98) */
    for (int i = 0; i < nid; i++) x[i] = xx[i];
    if (x[0] > (float)(nis - 1)) x[0] = 1.0f;
    const float ylast = yy[(size_t)(njd - 1) * nid];
    const float degre_a_radian = 0.017453295199f;
    for (int j = 0; j < njd; j++) {
        const float amplif = 1.0f / cosf(lats_dst[j] * degre_a_radian);      /* :63-65 */
        const int row = j == 0 ? 0 : (j == njd - 1 ? 2 : 1);
        for (int i = 0; i < nid; i++) {
            const float lo = i == 0 ? x[0] - 0.5f * (x[1] - x[0]) * amplif : x[i] - 0.5f * (x[i] - x[i - 1]) * amplif;                 /* :71-76 */
            const float hi = i == nid - 1 ? x[nid - 1] + 0.5f * (x[nid - 1] - x[nid - 2]) * amplif : x[i] + 0.5f * (x[i + 1] - x[i]) * amplif;   /* :88-93 */
            zout[(size_t)j * nid + i] = orc_avg_cell(zin, nis, njs, ext, row, lo, hi, yl[j], yh[j], ylast);
        }
    }
    free(xl); free(xh); free(yl); free(yh); free(x);
}
void orc_ez_avg(float *zout, const float *xx, const float *yy, int nid, int njd, const float *zin, int nis, int njs, int ext)
{
    float *xl = (float *)malloc(sizeof(float) * nid), *xh = (float *)malloc(sizeof(float) * nid);
    float *yl = (float *)malloc(sizeof(float) * njd), *yh = (float *)malloc(sizeof(float) * njd);
    orc_avg_bounds(xl, xh, yl, yh, xx, yy, nid, njd, nis, njs);
    const float ylast = yy[(size_t)(njd - 1) * nid];
    for (int j = 0; j < njd; j++) {
        const int row = j == 0 ? 0 : (j == njd - 1 ? 2 : 1);
        for (int i = 0; i < nid; i++) zout[(size_t)j * nid + i] = orc_avg_cell(zin, nis, njs, ext, row, xl[i], xh[i], yl[j], yh[j], ylast);
    }
    free(xl); free(xh); free(yl); free(yh);
}

/* ------------------------------------------------------------------------------------------
 * c_gdinterp dispatch, src/interp/gdinterp.c:133-309
 * ---------------------------------------------------------------------------------------- */
int orc_gdinterp(const orc_grid *g, int degree, float *zout, const float *zin,
                 const float *x, const float *y, int npts)
{
    int ni = g->ni, nj = g->nj, j1 = g->j1, j2 = g->j2, ext = g->extension;
    if (g->grtyp == 'Z' || g->grtyp == 'G') {
        switch (degree) {
        case ORC_NEAREST: orc_rgdint_0(zout, x, y, npts, zin, ni, j1, j2); return 0;
        case ORC_LINEAR:
            if (ext == 0) orc_irgdint_1_nw(zout, x, y, npts, g->ax, g->ay, zin, ni, nj);
            else orc_irgdint_1_w(zout, x, y, npts, g->ax, g->ay, zin, ni, j1, j2, ext);
            return 0;
        case ORC_CUBIC:
            if (ext == 0) orc_irgdint_3_nw(zout, x, y, npts, g->ax, g->ay, g->ncx, g->ncy, zin, g->i1, g->i2, j1, j2);
            else orc_irgdint_3_w(zout, x, y, npts, g->ax, g->ay, g->ncx, g->ncy, zin, ni, j1, j2, ext);
            return 0;
        }
        return -1;
    }
    switch (degree) {
    case ORC_NEAREST: orc_rgdint_0(zout, x, y, npts, zin, ni, j1, j2); return 0;
    case ORC_LINEAR:
        if (ext == 2) orc_rgdint_1_w(zout, x, y, npts, zin, ni, j1, j2, ext);
        else orc_rgdint_1_nw(zout, x, y, npts, zin, ni, j1, j2);
        return 0;
    case ORC_CUBIC:
        if (ext == 0) orc_rgdint_3_nw(zout, x, y, npts, zin, ni, j1, j2);
        else orc_rgdint_3_w(zout, x, y, npts, zin, ni, j1, j2, ext);
        return 0;
    }
    return -1;
}

/* the public c_gdxyfll (gdxyfll.c:29-104, every grid but 'U'): c_gdxyfll_orig, and on a y-inverted Gaussian grid y := nj_in + 1 - y (:88-94) */
int orc_gdxyfll_public(const orc_grid *g, float *x, float *y, const float *lat, float *lon, int n)
{
    int rc = orc_gdxyfll(g, x, y, lat, lon, n);
    if (rc >= 0 && g->grtyp == 'G' && g->ig[1] == 1)
        for (int k = 0; k < n; k++) y[k] = (float)(((double)g->j2 + 1.0) - (double)y[k]);
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * Gridset, locate cache, zones
 * ---------------------------------------------------------------------------------------- */
orc_gridset *orc_defset(orc_grid *gdout, orc_grid *gdin)   /* src/interp/ezdefset.c:38-171 */
{
    orc_gridset *gs = (orc_gridset *)calloc(1, sizeof(orc_gridset));
    gs->gdin = gdin; gs->gdout = gdout;
    return gs;
}

void orc_gridset_free(orc_gridset *gs)
{
    if (!gs) return;
    free(gs->x); free(gs->y);
    for (int k = 0; k < 5; k++) { free(gs->zones[k].x); free(gs->zones[k].y); free(gs->zones[k].idx); }
    free(gs);
}

int orc_calcxy(orc_gridset *gs)   /* src/interp/ez_calcxy.c:28-137 */
{
    if (gs->have_xy) return 0;
    orc_grid *gi = gs->gdin, *go = gs->gdout;
    int npts = go->ni * go->nj;
    orc_calclatlon(go);
    gs->x = (float *)malloc(sizeof(float) * npts);
    gs->y = (float *)malloc(sizeof(float) * npts);
    int rc;
    switch (gi->grtyp) {   /* note: operates on the target grid's cached lon IN PLACE */
    case 'A': case 'B': case 'E': case 'L': case 'N': case 'S': rc = ll2rgd(gs->x, gs->y, go->lat, go->lon, npts, gi); break;
    case 'Z': case 'G': ll2igd(gs->x, gs->y, go->lat, go->lon, npts, gi, gi->nj); rc = 0; break;
    default: rc = -1;
    }
    gs->have_xy = 1;
    return rc;
}

typedef int (*zone_pred)(float x, float y, const void *ctx);
static void collect_zone(orc_zone *zn, const float *x, const float *y, int npts, zone_pred pred, const void *ctx)
{
    int nh = 0;
    for (int i = 0; i < npts; i++) if (pred(x[i], y[i], ctx)) nh++;
    zn->npts = nh;
    if (!nh) return;
    zn->x = (float *)malloc(sizeof(float) * nh); zn->y = (float *)malloc(sizeof(float) * nh);
    zn->idx = (int *)malloc(sizeof(int) * nh);
    nh = 0;
    for (int i = 0; i < npts; i++) if (pred(x[i], y[i], ctx)) { zn->x[nh] = x[i]; zn->y[nh] = y[i]; zn->idx[nh] = i; nh++; }
}
typedef struct { int jlim; float ypole; int ni, nj; } zctx;
static int pred_nord(float x, float y, const void *c) { (void)x; return (int)y > ((const zctx *)c)->jlim; }          /* ez_defzone_nord.c:41-49 */
static int pred_sud(float x, float y, const void *c) { (void)x; return (int)y < ((const zctx *)c)->jlim; }           /* ez_defzone_sud.c:42-50 */
static int pred_pole(float x, float y, const void *c) { (void)x; return fabs(y - ((const zctx *)c)->ypole) < 1.0e-3; } /* ez_defzone_polenord.c:54-62 */
static int pred_dehors(float x, float y, const void *c)                                                              /* ez_defzone_dehors.c:63-74 */
{
    int ix = (int)(x + 0.5), iy = (int)(y + 0.5);
    const zctx *z = (const zctx *)c;
    return ix < 1 || iy < 1 || ix > z->ni || iy > z->nj;
}

int orc_defzones(orc_gridset *gs)   /* src/interp/ez_defzones.c:25-113 */
{
    if (gs->have_zones) return 0;
    orc_grid *gi = gs->gdin, *go = gs->gdout;
    int npts = go->ni * go->nj;
    int extrap = 0;
    switch (gi->grtyp) {
    case 'N': case 'S': extrap = 1; break;                        /* :46-50 */
    case 'L':
        extrap = (gi->extension == 0);
        /* falls through into the Z case in the reference; grref is 0 for 'L' so nothing happens */
        break;
    case 'Z':
        if (gi->grref == 'E' || gi->grref == 'L')
            if (358.0 > (gi->ax[gi->ni - 1] - gi->ax[0])) extrap = 1;
        break;
    }
    zctx c; c.ni = gi->ni; c.nj = gi->nj;
    if (extrap) {
        collect_zone(&gs->zones[ORC_Z_DEHORS], gs->x, gs->y, npts, pred_dehors, &c);
    } else {
        float xp, yp, latp, lonp;
        if (gi->grtyp == 'Z' && gi->grref == 'E') { yp = (float)(gi->nj + 0.5); }      /* ez_defzone_polenord.c:46-49 */
        else { latp = 90.0f; lonp = 0.0f; orc_gdxyfll(gi, &xp, &yp, &latp, &lonp, 1); }
        c.ypole = yp;
        collect_zone(&gs->zones[ORC_Z_POLE_NORD], gs->x, gs->y, npts, pred_pole, &c);
        if (gi->grtyp == 'Z' && gi->grref == 'E') { yp = 0.5f; }
        else { latp = -90.0f; lonp = 0.0f; orc_gdxyfll(gi, &xp, &yp, &latp, &lonp, 1); }
        c.ypole = yp;
        collect_zone(&gs->zones[ORC_Z_POLE_SUD], gs->x, gs->y, npts, pred_pole, &c);
        c.jlim = gi->j1 + 1;
        collect_zone(&gs->zones[ORC_Z_AU_SUD], gs->x, gs->y, npts, pred_sud, &c);
        c.jlim = gi->j2 - 2;
        collect_zone(&gs->zones[ORC_Z_AU_NORD], gs->x, gs->y, npts, pred_nord, &c);
    }
    gs->have_zones = 1;
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Polar / extrapolation correction
 * ---------------------------------------------------------------------------------------- */

/* ez_calcpoleval, src/interp/ez_calcpoleval.inc:21-48 (sequential REAL sum) */
static float calcpoleval(const float *z, int ni, const float *ax, char grtyp, char grref)
{
    float poleval = 0.0f;
    if (grtyp == 'Z' && grref == 'E') {
        for (int i = 1; i <= ni - 1; i++) poleval = poleval + z[i - 1] * (ax[i] - ax[i - 1]);
        if ((ax[ni - 1] - ax[0]) != 0.0f) poleval = poleval / (ax[ni - 1] - ax[0]);
        return poleval;
    }
    for (int i = 0; i < ni; i++) poleval = poleval + z[i];
    return poleval / (1.0f * (float)ni);
}

/* ez_corrval_aunord.c:28-117 */
static void corrval_aunord(orc_gridset *gs, int degree, float *zout, const float *zin)
{
    orc_grid *g = gs->gdin;
    orc_zone *zn = &gs->zones[ORC_Z_AU_NORD];
    int npts = zn->npts, ni = g->ni, nj = g->j2 - g->j1 + 1;
    if (npts <= 0) return;
    int j1 = g->j2 - 2, j2 = j1 + 3;
    float *temp = (float *)malloc(sizeof(float) * 4 * ni), *vals = (float *)malloc(sizeof(float) * npts);
    float poleval = calcpoleval(&zin[(size_t)(nj - 1) * ni], ni, g->ax, g->grtyp, g->grref);
    /* ez_fillnpole.inc:26-34: rows j2-2, j2-1, j2, then the pole row */
    for (int r = 0; r < 3; r++) memcpy(&temp[r * ni], &zin[(size_t)(g->j2 - 2 + r - g->j1) * ni], sizeof(float) * ni);   /* z(ni, j1:j2) */
    for (int i = 0; i < ni; i++) temp[3 * ni + i] = poleval;
    if (degree == ORC_CUBIC) {
        if (g->grtyp == 'Z' || g->grtyp == 'E' || g->grtyp == 'G') {
            float ay[4];   /* an 'E' source has ay == NULL here: the reference crashes */
            if (!g->ay) { fprintf(stderr, "orc: regular 'E' source + polar_correction dereferences NULL ay in the reference\n"); exit(13); }
            if (g->ay[g->j2 - 1] == 90.0) { ay[0] = g->ay[g->j2 - 4]; ay[1] = g->ay[g->j2 - 3]; ay[2] = g->ay[g->j2 - 2]; ay[3] = g->ay[g->j2 - 1]; }
            else { ay[0] = g->ay[g->j2 - 3]; ay[1] = g->ay[g->j2 - 2]; ay[2] = g->ay[g->j2 - 1]; ay[3] = 90.0f; }
            orc_irgdint_3_wnnc(vals, zn->x, zn->y, npts, g->ax, ay, temp, ni, j1, j2, g->extension);
        } else {
            orc_rgdint_3_wnnc(vals, zn->x, zn->y, npts, temp, ni, j1, j2, g->extension);
        }
    } else {
        float *ty = (float *)malloc(sizeof(float) * npts);
        for (int i = 0; i < npts; i++) ty[i] = (float)((double)zn->y[i] - (1.0 * (g->j2 - 3)));
        if (degree == ORC_LINEAR) orc_rgdint_1_w(vals, zn->x, ty, npts, temp, ni, 1, 4, g->extension);
        else orc_rgdint_0(vals, zn->x, ty, npts, temp, ni, 1, 4);
        free(ty);
    }
    for (int i = 0; i < npts; i++) zout[zn->idx[i]] = vals[i];
    free(vals); free(temp);
}

/* ez_corrval_ausud.c:30-137 */
static void corrval_ausud(orc_gridset *gs, int degree, float *zout, const float *zin)
{
    orc_grid *g = gs->gdin;
    orc_zone *zn = &gs->zones[ORC_Z_AU_SUD];
    int npts = zn->npts, ni = g->ni;
    if (npts <= 0) return;
    int j1 = g->j1 - 1, j2 = j1 + 3;
    float *temp = (float *)malloc(sizeof(float) * 4 * ni), *vals = (float *)malloc(sizeof(float) * npts);
    float vpolesud = calcpoleval(zin, ni, g->ax, g->grtyp, g->grref);
    /* ez_fillspole.inc:26-35: the pole row, then rows j1, j1+1, j1+2 */
    for (int i = 0; i < ni; i++) temp[i] = vpolesud;
    for (int r = 0; r < 3; r++) memcpy(&temp[(r + 1) * ni], &zin[(size_t)r * ni], sizeof(float) * ni);
    if (degree == ORC_CUBIC) {
        if (g->grtyp == 'Z' || g->grtyp == 'E' || g->grtyp == 'G') {
            float ay[4];
            if (!g->ay) { fprintf(stderr, "orc: regular 'E' source + polar_correction dereferences NULL ay in the reference\n"); exit(13); }
            if (g->ay[g->j1 - 1] == -90.0) { ay[0] = g->ay[0]; ay[1] = g->ay[1]; ay[2] = g->ay[2]; ay[3] = g->ay[3]; }
            else { ay[0] = -90.0f; ay[1] = g->ay[0]; ay[2] = g->ay[1]; ay[3] = g->ay[2]; }
            orc_irgdint_3_wnnc(vals, zn->x, zn->y, npts, g->ax, ay, temp, ni, j1, j2, g->extension);
        } else {
            orc_rgdint_3_wnnc(vals, zn->x, zn->y, npts, temp, ni, j1, j2, g->extension);
        }
    } else if (degree == ORC_LINEAR) {
        orc_rgdint_1_w(vals, zn->x, zn->y, npts, temp, ni, j1, j2, g->extension);
    } else {
        orc_rgdint_0(vals, zn->x, zn->y, npts, temp, ni, j1, j2);
    }
    for (int i = 0; i < npts; i++) zout[zn->idx[i]] = vals[i];
    free(vals); free(temp);
}

/* ez_corrval, src/interp/ez_corrval.c:29-151 */
static int corrval(orc_gridset *gs, const orc_opts *o, float *zout, const float *zin)
{
    orc_grid *g = gs->gdin;
    int ierc = 0;
    int nj = g->j2 - g->j1 + 1;
    orc_zone *dz = &gs->zones[ORC_Z_DEHORS];
    if (dz->npts > 0) {
        ierc = 2;
        if (o->degre_extrap == ORC_EXTRAP_ABORT) return -1;
        float valmin = zin[0], valmax = zin[0];                 /* ez_aminmax.inc: MINVAL / MAXVAL */
        for (size_t k = 1; k < (size_t)g->ni * nj; k++) { if (zin[k] < valmin) valmin = zin[k]; if (zin[k] > valmax) valmax = zin[k]; }
        if (o->degre_extrap >= ORC_EXTRAP_MAXIMUM) {
            float fudgeval = 0.0f;
            if (o->vecteur != 1) {
                switch (o->degre_extrap) {                      /* float + double*float -> double -> float */
                case ORC_EXTRAP_MAXIMUM: fudgeval = (float)((double)valmax + 0.05 * (double)(valmax - valmin)); break;
                case ORC_EXTRAP_MINIMUM: fudgeval = (float)((double)valmin - 0.05 * (double)(valmax - valmin)); break;
                case ORC_EXTRAP_VALUE: fudgeval = o->valeur_extrap; break;
                }
            }
            for (int i = 0; i < dz->npts; i++) zout[dz->idx[i]] = fudgeval;
        } else {
            float *temp = (float *)malloc(sizeof(float) * dz->npts);
            orc_gdinterp(g, o->degre_extrap, temp, zin, dz->x, dz->y, dz->npts);
            for (int i = 0; i < dz->npts; i++) zout[dz->idx[i]] = temp[i];
            free(temp);
        }
    }
    if (o->vecteur == 1) return ierc;
    if (o->degre_interp == 4) {
        /* interp_degree = average: ez_corrval_aunord / _ausud have no case for it (ez_corrval_aunord.c:52-110) and scatter an
         * uninitialised array over their zones (:112-114).  Defined only where the pole values overwrite every such point */
        for (int k = 0; k < 2; k++) {
            const orc_zone *st = &gs->zones[k == 0 ? ORC_Z_AU_NORD : ORC_Z_AU_SUD], *po = &gs->zones[k == 0 ? ORC_Z_POLE_NORD : ORC_Z_POLE_SUD];
            for (int i = 0; i < st->npts; i++) {
                int found = 0;
                for (int q = 0; q < po->npts && !found; q++) found = po->idx[q] == st->idx[i];
                if (!found) return -3;
            }
        }
    } else {
        corrval_aunord(gs, o->degre_interp, zout, zin);
        corrval_ausud(gs, o->degre_interp, zout, zin);
    }
    orc_zone *pn = &gs->zones[ORC_Z_POLE_NORD], *ps = &gs->zones[ORC_Z_POLE_SUD];
    if (pn->npts > 0 || ps->npts > 0) {
        float vpolnor = calcpoleval(&zin[(size_t)(nj - 1) * g->ni], g->ni, g->ax, g->grtyp, g->grref);
        for (int i = 0; i < pn->npts; i++) zout[pn->idx[i]] = vpolnor;
        float vpolsud = calcpoleval(zin, g->ni, g->ax, g->grtyp, g->grref);
        for (int i = 0; i < ps->npts; i++) zout[ps->idx[i]] = vpolsud;
    }
    /* ez_corrval.c:146-148 -> ez_corrbgd.inc:20-55: Z- or #-on-E source, 'B' target: the rows at the poles (row 1 unless the target is the northern
     * hemisphere, row nj unless it is the southern one: hem = the target's ig1) are replaced by their mean, a sequential REAL sum divided by ni * 1.0 */
    if ((g->grtyp == 'Z' || g->grtyp == '#') && g->grref == 'E' && gs->gdout->grtyp == 'B') {
        const orc_grid *t = gs->gdout;
        const int hem = t->ig[0];
        for (int pass = 0; pass < 2; pass++) {
            if (pass == 0 ? !(hem == 0 || hem == 2) : !(hem == 0 || hem == 1)) continue;
            float *row = zout + (pass == 0 ? 0 : (size_t)(t->nj - 1) * t->ni);
            float somme = 0.0f;
            for (int i = 0; i < t->ni; i++) somme = somme + row[i];
            const float moyenne = somme / ((float)t->ni * 1.0f);
            for (int i = 0; i < t->ni; i++) row[i] = moyenne;
        }
    }
    return ierc;
}

/* c_ezsint_orig, src/interp/ezsint.c:75-135 */
/* ez_xpnsrcgd -> ez_xpngdag2 / ez_xpngdb2 (ez_xpngdag2.inc:20-60, ez_xpngdb2.inc:20-60): a hemispheric field mirrored
 * into the other hemisphere; zout(ni, j1:j2), row (j - j1) of the C array */
void orc_xpnsrcgd(const orc_grid *g, float *zout, const float *zi, int symetrie)
{
    const int ni = g->ni, nj = g->nj, j1 = g->j1;
    const float sign = symetrie == 0 ? -1.0f : 1.0f;
#define ZO(i, j) zout[(size_t)((j) - j1) * ni + (i)]
#define ZI(i, j) zi[(size_t)((j) - 1) * ni + (i)]
    for (int j = 1; j <= nj; j++) for (int i = 0; i < ni; i++) ZO(i, j) = ZI(i, j);
    if (g->grtyp == 'B') {
        if (g->ig[0] == 1) { for (int j = 2; j <= nj; j++) for (int i = 0; i < ni; i++) ZO(i, 2 - j) = sign * ZI(i, j); }
        else { for (int j = 2; j <= nj; j++) for (int i = 0; i < ni; i++) ZO(i, nj + j - 1) = sign * ZI(i, nj - j + 1); }
    } else {
        if (g->ig[0] == 1) { for (int j = 1; j <= nj; j++) for (int i = 0; i < ni; i++) ZO(i, -j + 1) = sign * ZI(i, j); }
        else { for (int j = 1; j <= nj; j++) for (int i = 0; i < ni; i++) ZO(i, nj + j) = sign * ZI(i, nj - j + 1); }
    }
#undef ZO
#undef ZI
}

int orc_ezsint(orc_gridset *gs, const orc_opts *o, float *zout, const float *zin)
{
    orc_grid *gi = gs->gdin, *go = gs->gdout;
    if (gi == go) { memcpy(zout, zin, sizeof(float) * gi->ni * gi->nj); return 1; }
    float *xz = NULL, *pz = NULL;
    if ((gi->grtyp == 'A' || gi->grtyp == 'B' || gi->grtyp == 'G') && gi->ig[1] == 1) {      /* ezsint.c:100-106: axe_y_inverse */
        pz = (float *)malloc(sizeof(float) * (size_t)gi->ni * gi->nj);
        memcpy(pz, zin, sizeof(float) * (size_t)gi->ni * gi->nj);
        orc_permut(pz, gi->ni, gi->nj);
        zin = pz;
    }
    if ((gi->grtyp == 'A' || gi->grtyp == 'B' || gi->grtyp == 'G') && gi->ig[0] != 0) {      /* ezsint.c:108-113 -> ez_xpnsrcgd */
        xz = (float *)malloc(sizeof(float) * 2 * (size_t)gi->ni * gi->nj);
        orc_xpnsrcgd(gi, xz, zin, o->vecteur == 2 ? 0 : 1);
        zin = xz;
    }
    orc_calclatlon(go);
    orc_calcxy(gs);
    if (o->degre_interp == 4) orc_ez_avg(zout, gs->x, gs->y, go->ni, go->nj, zin, gi->ni, gi->nj, gi->extension);      /* gdinterp.c:181-187 */
    else orc_gdinterp(gi, o->degre_interp, zout, zin, gs->x, gs->y, go->ni * go->nj);
    int ierc = 0;
    if (o->polar_correction == 1) {
        orc_defzones(gs);
        ierc = corrval(gs, o, zout, zin);
    }
    free(xz); free(pz);
    return ierc;
}
