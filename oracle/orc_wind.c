/*
 * oracle/orc_wind.c -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * Restatement of the rotated-frame helpers and of the vector (wind) interpolation path of
 * librmn's EZ interpolator: ez_crot / ez_lac / ez_cal / mxm,
 * ez_gfxyfll / ez_gfllfxy, wind <-> speed/direction conversions, rotated-wind transforms, the
 * synthetic polar wind rows and c_ezuvint.  REAL = float, REAL*8 = double, float libm for REAL
 * intrinsics, double libm for REAL*8 ones.  Build with -ffp-contract=off.
 *
 * Scope: source grids L, A, B, G, Z-on-L, Z-on-E, (regular E without polar correction);
 * target grids L, A, B, G.  Rotated *targets* are out of scope (the reference's c_gduvfwd 'E'
 * branch reads uninitialised memory, src/interp/gduvfwd.c:44-47).
 */
#include "orc_ez.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

void orc_cigaxg(char grtyp, float *xg1, float *xg2, float *xg3, float *xg4, int ig1, int ig2, int ig3, int ig4);
void orc_cxgaig(char grtyp, int *ig1, int *ig2, int *ig3, int *ig4, float xg1, float xg2, float xg3, float xg4);

static const float RDTODG = 57.295779513082f;     /* src/interp/pi.cdk */
static const float DGTORD = 1.7453292519943e-2f;

/* ez_lac, src/interp/ez_lac.inc:31-47 */
static void lac(float *xyz, const float *lon, const float *lat, int n)
{
    float dar = acosf(-1.0f) / 180.0f;
    for (int i = 0; i < n; i++) {
        float cosdar = cosf(dar * lat[i]);
        xyz[3 * i + 0] = cosdar * cosf(dar * lon[i]);
        xyz[3 * i + 1] = cosdar * sinf(dar * lon[i]);
        xyz[3 * i + 2] = sinf(dar * lat[i]);
    }
}

/* ez_cal, src/interp/ez_cal.inc:22-47 */
static void cal(float *lon, float *lat, const float *xyz, int n)
{
    float rad = 180.f / acosf(-1.00f);
    for (int i = 0; i < n; i++) {
        lat[i] = asinf(fmaxf(-1.00f, fminf(1.0f, xyz[3 * i + 2]))) * rad;
        lon[i] = atan2f(xyz[3 * i + 1], xyz[3 * i + 0]) * rad;
        lon[i] = fmodf(lon[i], 360.0f);
        if (lon[i] < 0.0f) lon[i] = lon[i] + 360.0f;
    }
}

/* ez_crot, src/interp/ez_crot.inc:44-97.  r, ri are Fortran (3,3): r(i,j) = r[(j-1)*3 + (i-1)] */
static void crot(float *r, float *ri, float lon1, float lat1, float lon2, float lat2)
{
#define RR(i, j) r[((j) - 1) * 3 + ((i) - 1)]
    float p[3], q[3];
    lac(p, &lon1, &lat1, 1);
    lac(q, &lon2, &lat2, 1);
    float a = (p[0] * q[0]) + (p[1] * q[1]) + (p[2] * q[2]);
    float t1 = (p[1] * q[2]) - (q[1] * p[2]);
    float t2 = (q[0] * p[2]) - (p[0] * q[2]);
    float t3 = (p[0] * q[1]) - (q[0] * p[1]);
    float b = sqrtf(t1 * t1 + t2 * t2 + t3 * t3);
    float c = sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
    float e1 = ((a * p[0]) - q[0]) / b, e2 = ((a * p[1]) - q[1]) / b, e3 = ((a * p[2]) - q[2]) / b;
    float d = sqrtf(e1 * e1 + e2 * e2 + e3 * e3);
    RR(1, 1) = -p[0] / c; RR(1, 2) = -p[1] / c; RR(1, 3) = -p[2] / c;
    RR(2, 1) = e1 / d;    RR(2, 2) = e2 / d;    RR(2, 3) = e3 / d;
    RR(3, 1) = t1 / b;    RR(3, 2) = t2 / b;    RR(3, 3) = t3 / b;
    for (int i = 1; i <= 3; i++)
        for (int j = 1; j <= 3; j++) ri[(j - 1) * 3 + (i - 1)] = RR(j, i);
#undef RR
}

/* mxm(a,3,b,3,c,n), src/base/mxm.F90:23-40: c(i,j) = ((0 + a(i,1)b(1,j)) + a(i,2)b(2,j)) + a(i,3)b(3,j) */
static void mxm3(const float *a, const float *b, float *c, int n)
{
    for (int j = 0; j < n; j++)
        for (int i = 0; i < 3; i++) {
            float s = 0.0f;
            for (int k = 0; k < 3; k++) s = s + a[k * 3 + i] * b[3 * j + k];
            c[3 * j + i] = s;
        }
}

/* ez_gfxyfll / c_ezgfxyfll, src/interp/ez_gfxyfll.c:38-57: true (lon,lat) -> rotated (lon,lat) */
void orc_gfxyfll(float *lon_rot, float *lat_rot, const float *lon, const float *lat, int n,
                 float xlat1, float xlon1, float xlat2, float xlon2)
{
    float r[9], ri[9];
    float *cart = (float *)malloc(sizeof(float) * 3 * n), *carot = (float *)malloc(sizeof(float) * 3 * n);
    crot(r, ri, xlon1, xlat1, xlon2, xlat2);
    lac(cart, lon, lat, n);
    mxm3(r, cart, carot, n);
    cal(lon_rot, lat_rot, carot, n);
    free(cart); free(carot);
}

/* ez_gfllfxy / c_ezgfllfxy, src/interp/ez_gfllfxy.c:38-55: rotated -> true */
void orc_gfllfxy(float *lon_true, float *lat_true, const float *lon_rot, const float *lat_rot, int n,
                 float xlat1, float xlon1, float xlat2, float xlon2)
{
    float r[9], ri[9];
    float *cart = (float *)malloc(sizeof(float) * 3 * n), *carot = (float *)malloc(sizeof(float) * 3 * n);
    crot(r, ri, xlon1, xlat1, xlon2, xlat2);
    lac(cart, lon_rot, lat_rot, n);
    mxm3(ri, cart, carot, n);
    cal(lon_true, lat_true, carot, n);
    free(cart); free(carot);
}

/* ez_llwfgdw, src/interp/ez_llwfgdw.inc:22-170: grid components -> speed, direction (in place) */
static void llwfgdw(float *z1, float *z2, const float *xlon, int n, char grtyp, int ig1, int ig2, int ig3, int ig4)
{
    float xg1, xg2, xg3, xg4 = 0.f;
    if (grtyp == 'N' || grtyp == 'S') orc_cigaxg(grtyp, &xg1, &xg2, &xg3, &xg4, ig1, ig2, ig3, ig4);
    for (int i = 0; i < n; i++) {
        float uu = z1[i], vv = z2[i];
        float spd0 = sqrtf(uu * uu + vv * vv), dir0;
        if (spd0 == 0.0f) dir0 = 0.0f;
        else if (grtyp == 'N') {                                 /* :52-70 */
            if (uu == 0.0f) dir0 = (vv >= 0.0f) ? xlon[i] + xg4 - 90.0f : xlon[i] + xg4 + 90.0f;
            else dir0 = xlon[i] + xg4 - RDTODG * atan2f(vv, uu);
        } else if (grtyp == 'S') {                               /* :80-98 */
            if (uu == 0.0f) dir0 = (vv >= 0.0f) ? 90.0f - xlon[i] + xg4 : 270.0f - xlon[i] + xg4;
            else dir0 = 180.0f - xlon[i] + xg4 - RDTODG * atan2f(vv, uu);
        } else {                                                 /* A, B, G, L :108-127 */
            if (uu == 0.0f) dir0 = (vv >= 0.0f) ? 180.0f : 0.0f;
            else dir0 = 270.0f - RDTODG * atan2f(vv, uu);
        }
        dir0 = fmodf(fmodf(dir0, 360.0f) + 360.0f, 360.0f);
        z1[i] = spd0; z2[i] = dir0;
    }
}

/* ez_gdwfllw, src/interp/ez_gdwfllw.inc:23-137: speed, direction -> grid components (in place) */
static void gdwfllw(float *z1, float *z2, const float *xlon, int n, char grtyp, int ig1, int ig2, int ig3, int ig4)
{
    float xg1, xg2, xg3, xg4 = 0.f;
    if (grtyp == 'N' || grtyp == 'S') orc_cigaxg(grtyp, &xg1, &xg2, &xg3, &xg4, ig1, ig2, ig3, ig4);
    for (int i = 0; i < n; i++) {
        float psi;
        if (grtyp == 'N') psi = xlon[i] + xg4 - z2[i];
        else if (grtyp == 'S') psi = 180.0f - xlon[i] + xg4 - z2[i];
        else if (grtyp == 'A' || grtyp == 'B' || grtyp == 'G' || grtyp == 'L') psi = 270.0f - z2[i];
        else continue;                                                /* any other type (e.g. 'Z'): left as it is (:123-134) */
        float u = cosf(psi * DGTORD) * z1[i];
        float v = sinf(psi * DGTORD) * z1[i];
        z1[i] = u; z2[i] = v;
    }
}

/* c_ezllwfgfw, src/interp/ez_llwfgfw.c:38-73 with ez_uvacart.inc:24-62, ez_cartauv.inc:23-62 */
static void llwfgfw(float *uu, float *vv, const float *latin, const float *lonin,
                    const float *latgf, const float *longf, int n, int ig1, int ig2, int ig3, int ig4)
{
    float r[9], ri[9], xlat1, xlon1, xlat2, xlon2;
    float *xyz = (float *)malloc(sizeof(float) * 3 * n), *uvcart = (float *)malloc(sizeof(float) * 3 * n);
    orc_cigaxg('E', &xlat1, &xlon1, &xlat2, &xlon2, ig1, ig2, ig3, ig4);
    crot(r, ri, xlon1, xlat1, xlon2, xlat2);
    const double dar = (double)(acosf(-1.f) / 180.f);            /* REAL expression stored in REAL*8 */
    for (int k = 0; k < n; k++) {                                /* ez_uvacart at ROTATED lon/lat */
        double a = sin(dar * (double)longf[k]), b = cos(dar * (double)longf[k]);
        double c = sin(dar * (double)latgf[k]), d = cos(dar * (double)latgf[k]);
        xyz[3 * k + 0] = (float)(-((double)uu[k] * a) - ((double)vv[k] * b * c));
        xyz[3 * k + 1] = (float)(((double)uu[k] * b) - ((double)vv[k] * a * c));
        xyz[3 * k + 2] = (float)((double)vv[k] * d);
    }
    mxm3(ri, xyz, uvcart, n);
    for (int k = 0; k < n; k++) {                                /* ez_cartauv at TRUE lon/lat */
        double a = cos(dar * (double)lonin[k]), b = sin(dar * (double)lonin[k]);
        double e = cos(dar * (double)latin[k]), f = sin(dar * (double)latin[k]);
        float c1 = uvcart[3 * k + 0], c2 = uvcart[3 * k + 1], c3 = uvcart[3 * k + 2];
        uu[k] = (float)(((double)c2 * a) - ((double)c1 * b));
        double c = ((double)c1 * a) + ((double)c2 * b);
        double d = sqrt(c * c + (double)(c3 * c3));
        double s = ((double)c3 * e) - (c * f);
        vv[k] = (float)(s >= 0.0 ? fabs(d) : -fabs(d));           /* SIGN(D, s) */
    }
    llwfgdw(uu, vv, longf, n, 'L', 0, 0, 0, 0);
    free(xyz); free(uvcart);
}

/* c_ezgfwfllw, src/interp/ez_gfwfllw.c:38-79: (speed, direction) at true lat/lon -> components on the rotated grid:
 * ez_gdwfllw 'L', ez_uvacart at TRUE lon/lat, mxm with r, ez_cartauv at ROTATED lon/lat */
static void gfwfllw(float *uu, float *vv, const float *latin, const float *lonin,
                    const float *latgf, const float *longf, int n, int ig1, int ig2, int ig3, int ig4)
{
    float r[9], ri[9], xlat1, xlon1, xlat2, xlon2;
    float *xyz = (float *)malloc(sizeof(float) * 3 * n), *uvcart = (float *)malloc(sizeof(float) * 3 * n);
    orc_cigaxg('E', &xlat1, &xlon1, &xlat2, &xlon2, ig1, ig2, ig3, ig4);
    crot(r, ri, xlon1, xlat1, xlon2, xlat2);
    gdwfllw(uu, vv, lonin, n, 'L', 0, 0, 0, 0);
    const double dar = (double)(acosf(-1.f) / 180.f);
    for (int k = 0; k < n; k++) {                                /* ez_uvacart at TRUE lon/lat */
        double a = sin(dar * (double)lonin[k]), b = cos(dar * (double)lonin[k]);
        double c = sin(dar * (double)latin[k]), d = cos(dar * (double)latin[k]);
        xyz[3 * k + 0] = (float)(-((double)uu[k] * a) - ((double)vv[k] * b * c));
        xyz[3 * k + 1] = (float)(((double)uu[k] * b) - ((double)vv[k] * a * c));
        xyz[3 * k + 2] = (float)((double)vv[k] * d);
    }
    mxm3(r, xyz, uvcart, n);
    for (int k = 0; k < n; k++) {                                /* ez_cartauv at ROTATED lon/lat */
        double a = cos(dar * (double)longf[k]), b = sin(dar * (double)longf[k]);
        double e = cos(dar * (double)latgf[k]), f = sin(dar * (double)latgf[k]);
        float c1 = uvcart[3 * k + 0], c2 = uvcart[3 * k + 1], c3 = uvcart[3 * k + 2];
        uu[k] = (float)(((double)c2 * a) - ((double)c1 * b));
        double c = ((double)c1 * a) + ((double)c2 * b);
        double d = sqrt(c * c + (double)(c3 * c3));
        double s = ((double)c3 * e) - (c * f);
        vv[k] = (float)(s >= 0.0 ? fabs(d) : -fabs(d));
    }
    free(xyz); free(uvcart);
}

/* c_gdwdfuv_orig, src/interp/gdwdfuv.c:29-100 */
int orc_gdwdfuv(orc_grid *g, float *spd, float *dir, const float *uu, const float *vv,
                const float *lat, const float *lon, int npts)
{
    memcpy(spd, uu, sizeof(float) * npts);
    memcpy(dir, vv, sizeof(float) * npts);
    int is_e = (g->grtyp == 'E'), is_ze = (g->grtyp == 'Z' && g->grref == 'E');
    if (is_e || is_ze) {
        const float *xg = is_e ? g->xg : g->xgref;
        const int *ig = is_e ? g->ig : g->igref;
        float *lat_rot = (float *)malloc(sizeof(float) * npts), *lon_rot = (float *)malloc(sizeof(float) * npts);
        orc_gfxyfll(lon_rot, lat_rot, lon, lat, npts, xg[0], xg[1], xg[2], xg[3]);
        llwfgfw(spd, dir, lat, lon, lat_rot, lon_rot, npts, ig[0], ig[1], ig[2], ig[3]);
        free(lat_rot); free(lon_rot);
        return 0;
    }
    if (g->grtyp == 'Z') llwfgdw(spd, dir, lon, npts, g->grref, g->igref[0], g->igref[1], g->igref[2], g->igref[3]);
    else llwfgdw(spd, dir, lon, npts, g->grtyp, g->ig[0], g->ig[1], g->ig[2], g->ig[3]);
    return 0;
}

/* c_gduvfwd_orig, src/interp/gduvfwd.c:29-96; a Z target falls through the missing `break` at :81-92 into a second
 * ez_gdwfllw call with grtyp 'Z', which converts nothing */
int orc_gduvfwd(orc_grid *g, float *uu, float *vv, const float *spd, const float *dir,
                const float *lat, const float *lon, int npts)
{
    memcpy(uu, spd, sizeof(float) * npts);
    memcpy(vv, dir, sizeof(float) * npts);
    if (g->grtyp == 'E') {
        fprintf(stderr, "orc_gduvfwd: regular 'E' target grids are out of scope (the reference reads uninitialised memory there)\n");
        return -1;
    }
    if (g->grtyp == 'Z' && g->grref == 'E') {                   /* :62-80: ez_gfxyfll + c_ezgfwfllw, then return */
        float *lat_rot = (float *)malloc(sizeof(float) * npts), *lon_rot = (float *)malloc(sizeof(float) * npts);
        orc_gfxyfll(lon_rot, lat_rot, lon, lat, npts, g->xgref[0], g->xgref[1], g->xgref[2], g->xgref[3]);
        gfwfllw(uu, vv, lat, lon, lat_rot, lon_rot, npts, g->igref[0], g->igref[1], g->igref[2], g->igref[3]);
        free(lat_rot); free(lon_rot);
        return 0;
    }
    if (g->grtyp == 'Z' || g->grtyp == 'Y') gdwfllw(uu, vv, lon, npts, g->grref, g->igref[0], g->igref[1], g->igref[2], g->igref[3]);      /* :58-61, :81-87 */
    gdwfllw(uu, vv, lon, npts, g->grtyp, g->ig[0], g->ig[1], g->ig[2], g->ig[3]);
    return 0;
}

/* c_gdllfxy_orig for one source row, src/interp/gdllfxy.c:245-390 */
static void gdllfxy_row(const orc_grid *g, float *lat, float *lon, const float *x, const float *y, int n)
{
    switch (g->grtyp) {
    case 'A': case 'B': case 'L':
        for (int i = 0; i < n; i++) {
            lat[i] = (float)(((double)y[i] - 1.0) * (double)g->xg[2] + (double)g->xg[0]);
            lon[i] = (float)(((double)x[i] - 1.0) * (double)g->xg[3] + (double)g->xg[1]);
            lon[i] = (float)fmod((double)(lon[i] + 360.0), 360.0);
        }
        break;
    case 'E': {
        float *tx = (float *)malloc(sizeof(float) * n), *ty = (float *)malloc(sizeof(float) * n);
        for (int i = 0; i < n; i++) {
            float dlat = (float)(180.0 / g->nj), dlon = (float)(360.0 / (g->ni - 1));
            float swlat = (float)(-90.0 + 0.5 * (double)dlat), swlon = 0.0f;
            tx[i] = (float)(((double)x[i] - 1.0) * (double)dlon + (double)swlon);
            ty[i] = (float)(((double)y[i] - 1.0) * (double)dlat + (double)swlat);
        }
        orc_gfllfxy(lon, lat, tx, ty, n, g->xg[0], g->xg[1], g->xg[2], g->xg[3]);
        free(tx); free(ty);
        break;
    }
    case 'Z': case 'G': {
        float *tx = (float *)malloc(sizeof(float) * n), *ty = (float *)malloc(sizeof(float) * n);
        for (int i = 0; i < n; i++) {
            int indx = (int)x[i] - 1, indy = (int)y[i] - 1;
            indx = indx < 0 ? 0 : indx; indy = indy < 0 ? 0 : indy;
            indx = indx > g->ni - 2 ? g->ni - 2 : indx;
            indy = indy > g->j2 - 2 ? g->j2 - 2 : indy;
            float delxx = g->ax[indx + 1] - g->ax[indx];
            tx[i] = (float)((double)g->ax[indx] + (((double)x[i] - 1.0 - indx) * (double)delxx));
            float delyy = g->ay[indy + 1] - g->ay[indy];
            ty[i] = (float)((double)g->ay[indy] + (((double)y[i] - 1.0 - indy) * (double)delyy));
        }
        if (g->grref == 'E') orc_gfllfxy(lon, lat, tx, ty, n, g->xgref[0], g->xgref[1], g->xgref[2], g->xgref[3]);
        else for (int i = 0; i < n; i++) {                      /* 'L' :372-379 */
            lat[i] = ty[i] * g->xgref[2] + g->xgref[0];
            lon[i] = tx[i] * g->xgref[3] + g->xgref[1];
            lon[i] = (float)fmod((double)(lon[i] + 360.0), 360.0);
        }
        free(tx); free(ty);
        break;
    }
    }
}

static float calcpoleval_w(const float *z, int ni, const float *ax, char grtyp, char grref)
{   /* ez_calcpoleval.inc:21-48 */
    float p = 0.0f;
    if (grtyp == 'Z' && grref == 'E') {
        for (int i = 1; i <= ni - 1; i++) p = p + z[i - 1] * (ax[i] - ax[i - 1]);
        if ((ax[ni - 1] - ax[0]) != 0.0f) p = p / (ax[ni - 1] - ax[0]);
        return p;
    }
    for (int i = 0; i < ni; i++) p = p + z[i];
    return p / (1.0f * (float)ni);
}

/* ez_calcnpolarwind.c:28-138 (north = 1) / ez_calcspolarwind.c (north = 0): builds the 4-row strip */
static void calcpolarwind(float *polar_uu_in, float *polar_vv_in, const float *uuin, const float *vvin,
                          int ni, int nj, orc_grid *g, int north)
{
    float *puu = (float *)malloc(sizeof(float) * ni), *pvv = (float *)malloc(sizeof(float) * ni);
    float *pwd = (float *)malloc(sizeof(float) * ni), *pspd = (float *)malloc(sizeof(float) * ni);
    float *plat = (float *)malloc(sizeof(float) * ni), *plon = (float *)malloc(sizeof(float) * ni);
    float *px = (float *)malloc(sizeof(float) * ni), *py = (float *)malloc(sizeof(float) * ni);
    for (int i = 0; i < ni; i++) { px[i] = (float)(1.0 * (i + 1)); py[i] = north ? (float)(1.0 * nj) : 1.0f; }
    gdllfxy_row(g, plat, plon, px, py, ni);
    if (g->grtyp == 'Z' && g->grref == 'E') {
        float *glat = (float *)malloc(sizeof(float) * ni), *glon = (float *)malloc(sizeof(float) * ni);
        memcpy(glat, plat, sizeof(float) * ni); memcpy(glon, plon, sizeof(float) * ni);
        orc_gfxyfll(plon, plat, glon, glat, ni, g->xgref[0], g->xgref[1], g->xgref[2], g->xgref[3]);
        free(glat); free(glon);
    }
    const float *urow = north ? &uuin[(size_t)(nj - 1) * ni] : uuin;
    const float *vrow = north ? &vvin[(size_t)(nj - 1) * ni] : vvin;
    /* gda = 'A' 24x12: c_gdwdfuv -> ez_llwfgdw('A') */
    memcpy(pspd, urow, sizeof(float) * ni); memcpy(pwd, vrow, sizeof(float) * ni);
    llwfgdw(pspd, pwd, plon, ni, 'A', 0, 0, 0, 0);
    /* gdps = 'N' (or 'S') ni x 1 with (pi,pj,d60,dgrw) = (0,0,1000,0) */
    char hs = north ? 'N' : 'S';
    int ig1n, ig2n, ig3n, ig4n;
    orc_cxgaig(hs, &ig1n, &ig2n, &ig3n, &ig4n, 0.0f, 0.0f, 1000.0f, 0.0f);
    memcpy(puu, pspd, sizeof(float) * ni); memcpy(pvv, pwd, sizeof(float) * ni);
    gdwfllw(puu, pvv, plon, ni, hs, ig1n, ig2n, ig3n, ig4n);
    float uupole = calcpoleval_w(puu, ni, g->ax, g->grtyp, g->grref);
    float vvpole = calcpoleval_w(pvv, ni, g->ax, g->grtyp, g->grref);
    float zero = 0.0f;
    float spd0 = uupole, wd0 = vvpole;
    llwfgdw(&spd0, &wd0, &zero, 1, hs, ig1n, ig2n, ig3n, ig4n);
    pspd[0] = spd0; pwd[0] = wd0;
    plat[0] = north ? 90.0f : -90.0f;
    for (int i = 1; i < ni; i++) {
        pwd[i] = north ? pwd[0] + plon[i] : pwd[0] - plon[i];
        pspd[i] = pspd[0];
        plat[i] = plat[0];
    }
    pwd[0] = pwd[0] + plon[0];    /* both hemispheres ADD for element 0 (ez_calcspolarwind.c:90) */
    memcpy(puu, pspd, sizeof(float) * ni); memcpy(pvv, pwd, sizeof(float) * ni);
    gdwfllw(puu, pvv, plon, ni, 'A', 0, 0, 0, 0);
    if (north) {
        for (int j = 0; j < 3; j++) {
            memcpy(&polar_uu_in[j * ni], &uuin[(size_t)(nj - 3 + j) * ni], sizeof(float) * ni);
            memcpy(&polar_vv_in[j * ni], &vvin[(size_t)(nj - 3 + j) * ni], sizeof(float) * ni);
        }
        memcpy(&polar_uu_in[3 * ni], puu, sizeof(float) * ni);
        memcpy(&polar_vv_in[3 * ni], pvv, sizeof(float) * ni);
    } else {
        for (int j = 0; j < 3; j++) {
            memcpy(&polar_uu_in[(j + 1) * ni], &uuin[(size_t)j * ni], sizeof(float) * ni);
            memcpy(&polar_vv_in[(j + 1) * ni], &vvin[(size_t)j * ni], sizeof(float) * ni);
        }
        memcpy(polar_uu_in, puu, sizeof(float) * ni);
        memcpy(polar_vv_in, pvv, sizeof(float) * ni);
    }
    free(puu); free(pvv); free(pwd); free(pspd); free(plat); free(plon); free(px); free(py);
}

/* ez_corrvec_aunord.c:29-139 (north = 1), ez_corrvec_ausud.c:28-125 (north = 0) */
static void corrvec_strip(orc_gridset *gs, int degree, float *uuout, float *vvout,
                          const float *uuin, const float *vvin, int north)
{
    orc_grid *g = gs->gdin;
    orc_zone *zn = &gs->zones[north ? ORC_Z_AU_NORD : ORC_Z_AU_SUD];
    int npts = zn->npts, ni = g->ni, nj = g->j2 - g->j1 + 1;
    int j1 = north ? g->j2 - 2 : g->j1 - 1, j2 = j1 + 3;
    float *pu = (float *)malloc(sizeof(float) * 4 * ni), *pv = (float *)malloc(sizeof(float) * 4 * ni);
    float *cu = (float *)malloc(sizeof(float) * (npts + 1)), *cv = (float *)malloc(sizeof(float) * (npts + 1));
    calcpolarwind(pu, pv, uuin, vvin, ni, nj, g, north);
    if (degree == ORC_CUBIC) {
        if (g->grtyp == 'Z' || g->grtyp == 'E' || g->grtyp == 'G') {
            float ay[4];
            if (!g->ay) { fprintf(stderr, "orc: regular 'E' source + polar_correction dereferences NULL ay in the reference\n"); exit(13); }
            if (north) {
                if (g->ay[g->j2 - 1] == 90.0) { ay[0] = g->ay[g->j2 - 4]; ay[1] = g->ay[g->j2 - 3]; ay[2] = g->ay[g->j2 - 2]; ay[3] = g->ay[g->j2 - 1]; }
                else { ay[0] = g->ay[g->j2 - 3]; ay[1] = g->ay[g->j2 - 2]; ay[2] = g->ay[g->j2 - 1]; ay[3] = 90.0f; }
            } else {
                if (g->ay[0] == -90.0) { ay[0] = g->ay[0]; ay[1] = g->ay[1]; ay[2] = g->ay[2]; ay[3] = g->ay[3]; }
                else { ay[0] = -90.f; ay[1] = g->ay[0]; ay[2] = g->ay[1]; ay[3] = g->ay[2]; }
            }
            orc_irgdint_3_wnnc(cu, zn->x, zn->y, npts, g->ax, ay, pu, ni, j1, j2, g->extension);
            orc_irgdint_3_wnnc(cv, zn->x, zn->y, npts, g->ax, ay, pv, ni, j1, j2, g->extension);
        } else {
            orc_rgdint_3_wnnc(cu, zn->x, zn->y, npts, pu, ni, j1, j2, g->extension);
            orc_rgdint_3_wnnc(cv, zn->x, zn->y, npts, pv, ni, j1, j2, g->extension);
        }
    } else if (north) {
        float *ty = (float *)malloc(sizeof(float) * (npts + 1));
        for (int i = 0; i < npts; i++) ty[i] = (float)((double)zn->y[i] - (1.0 * (g->j2 - 3)));
        if (degree == ORC_LINEAR) { orc_rgdint_1_w(cu, zn->x, ty, npts, pu, ni, 1, 4, g->extension); orc_rgdint_1_w(cv, zn->x, ty, npts, pv, ni, 1, 4, g->extension); }
        else { orc_rgdint_0(cu, zn->x, ty, npts, pu, ni, 1, 4); orc_rgdint_0(cv, zn->x, ty, npts, pv, ni, 1, 4); }
        free(ty);
    } else {
        if (degree == ORC_LINEAR) { orc_rgdint_1_w(cu, zn->x, zn->y, npts, pu, ni, j1, j2, g->extension); orc_rgdint_1_w(cv, zn->x, zn->y, npts, pv, ni, j1, j2, g->extension); }
        else { orc_rgdint_0(cu, zn->x, zn->y, npts, pu, ni, j1, j2); orc_rgdint_0(cv, zn->x, zn->y, npts, pv, ni, j1, j2); }
    }
    for (int i = 0; i < npts; i++) { uuout[zn->idx[i]] = cu[i]; vvout[zn->idx[i]] = cv[i]; }
    free(pu); free(pv); free(cu); free(cv);
}

/* c_ezuvint_orig, src/interp/ezuvint.c:51-94 */
int orc_ezuvint(orc_gridset *gs, const orc_opts *o, float *uuout, float *vvout,
                const float *uuin, const float *vvin)
{
    orc_grid *gi = gs->gdin, *go = gs->gdout;
    int npts = go->ni * go->nj;
    orc_calclatlon(go);
    orc_opts ov = *o;
    ov.vecteur = 1;
    int ierc = 0;
    int ierc1 = orc_ezsint(gs, &ov, uuout, uuin);
    int ierc2 = orc_ezsint(gs, &ov, vvout, vvin);
    if (ierc1 == 2 || ierc2 == 2) ierc = 2;
    if (o->polar_correction == 1 && gi != go) {                  /* ez_corrvec.c:24-48; one grid on both sides: c_ezsint returned before ez_defzones (ezsint.c), the set has no zones */
        orc_defzones(gs);
        if (gs->zones[ORC_Z_AU_NORD].npts > 0) corrvec_strip(gs, o->degre_interp, uuout, vvout, uuin, vvin, 1);
        if (gs->zones[ORC_Z_AU_SUD].npts > 0) corrvec_strip(gs, o->degre_interp, uuout, vvout, uuin, vvin, 0);
        if (gs->zones[ORC_Z_POLE_NORD].npts > 0) corrvec_strip(gs, o->degre_interp, uuout, vvout, uuin, vvin, 1);
        if (gs->zones[ORC_Z_POLE_SUD].npts > 0) corrvec_strip(gs, o->degre_interp, uuout, vvout, uuin, vvin, 0);
    }
    float *spd = (float *)malloc(sizeof(float) * npts), *dir = (float *)malloc(sizeof(float) * npts);
    orc_gdwdfuv(gi, spd, dir, uuout, vvout, go->lat, go->lon, npts);
    int rc = orc_gduvfwd(go, uuout, vvout, spd, dir, go->lat, go->lon, npts);
    free(spd); free(dir);
    return rc < 0 ? rc : ierc;
}

/* c_ezwdint_orig, ezwdint.c:62-113: the vector interpolation of c_ezuvint_orig, but the result is left as wind
 * speed (uuout) and meteorological direction (vvout) on the target grid: c_gdwdfuv only, no c_gduvfwd. */
int orc_ezwdint(orc_gridset *gs, const orc_opts *o, float *uuout, float *vvout,
                const float *uuin, const float *vvin)
{
    orc_grid *gi = gs->gdin, *go = gs->gdout;
    int npts = go->ni * go->nj;
    orc_calclatlon(go);
    orc_opts ov = *o;
    ov.vecteur = 1;
    int ierc = 0;
    int ierc1 = orc_ezsint(gs, &ov, uuout, uuin);
    int ierc2 = orc_ezsint(gs, &ov, vvout, vvin);
    if (ierc1 == 2 || ierc2 == 2) ierc = 2;
    if (o->polar_correction == 1 && gi != go) {                  /* ez_corrvec.c:24-48; one grid on both sides: c_ezsint returned before ez_defzones (ezsint.c), the set has no zones */
        orc_defzones(gs);
        if (gs->zones[ORC_Z_AU_NORD].npts > 0) corrvec_strip(gs, o->degre_interp, uuout, vvout, uuin, vvin, 1);
        if (gs->zones[ORC_Z_AU_SUD].npts > 0) corrvec_strip(gs, o->degre_interp, uuout, vvout, uuin, vvin, 0);
        if (gs->zones[ORC_Z_POLE_NORD].npts > 0) corrvec_strip(gs, o->degre_interp, uuout, vvout, uuin, vvin, 1);
        if (gs->zones[ORC_Z_POLE_SUD].npts > 0) corrvec_strip(gs, o->degre_interp, uuout, vvout, uuin, vvin, 0);
    }
    float *spd = (float *)malloc(sizeof(float) * npts), *dir = (float *)malloc(sizeof(float) * npts);
    int rc = orc_gdwdfuv(gi, spd, dir, uuout, vvout, go->lat, go->lon, npts);
    memcpy(uuout, spd, sizeof(float) * npts); memcpy(vvout, dir, sizeof(float) * npts);
    free(spd); free(dir);
    return rc < 0 ? rc : ierc;
}
