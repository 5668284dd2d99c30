/*
 * oracle/orc_yy.c -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * Yin-Yang 'U' source grids -> one ordinary target grid: c_ezgdef_supergrid
 * (src/interp/ezgdef_supergrid.c:40-150), c_ezgdef_yymask (ezgdef_yymask.c:25-52), c_ezyymint (ezyymint.c:27-95),
 * c_ezyy_calcxy (ezyy_calcxy.c:25-179, the yyout == 0 branch), c_ezyysint (ezyysint.c:27-170) and
 * c_ezyyuvint (ezyyuvint.c, the yyin == 1 && yyout == 0 branch).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "orc_ez.h"

/* c_ezgdef_yymask: the part of a subgrid with 45 <= ax <= 315 and -45 <= ay <= 45, as a grid of its own */
static orc_grid *yymask(const orc_grid *sub)
{
    int k = 0, i0 = 0, j0 = 0;
    for (int i = 0; i < sub->ni; i++) if (sub->ax[i] >= 45.0 && sub->ax[i] <= 315.0) { k++; if (k == 1) i0 = i; }
    int yni = k;
    k = 0;
    for (int j = 0; j < sub->nj; j++) if (sub->ay[j] >= -45.0 && sub->ay[j] <= 45.0) { k++; if (k == 1) j0 = j; }
    int ynj = k;
    return orc_grid_define(yni, ynj, sub->grtyp, sub->grref, sub->igref[0], sub->igref[1], sub->igref[2], sub->igref[3], &sub->ax[i0], &sub->ay[j0]);
}

orc_supergrid *orc_supergrid_define(orc_grid *yin, orc_grid *yan)
{
    orc_supergrid *sg = (orc_supergrid *)calloc(1, sizeof(orc_supergrid));
    sg->sub[0] = yin; sg->sub[1] = yan;
    sg->mask[0] = yymask(yin); sg->mask[1] = yymask(yan);
    sg->ni = yin->ni; sg->nj = 2 * yin->nj;
    return sg;
}

/* c_ezyymint + the x,y part of c_ezyy_calcxy for a single target grid.  mask[k] == 1: point k is served by Yang. */
static int yy_plan(orc_supergrid *sg, orc_grid *gdout, const orc_opts *o)
{
    if (sg->planned_for == gdout) return 0;
    int npts = gdout->ni * gdout->nj;
    orc_calclatlon(gdout);
    float *lat = (float *)malloc(sizeof(float) * npts), *lon = (float *)malloc(sizeof(float) * npts);
    memcpy(lat, gdout->lat, sizeof(float) * npts); memcpy(lon, gdout->lon, sizeof(float) * npts);      /* c_gdll(yin_gdout, yinlat, yinlon) */
    /* ezyymint.c:44-66: a zero field on the Yin mask grid, nearest neighbour, extrapolation value 1 */
    orc_opts mo = *o;
    mo.degre_interp = ORC_NEAREST; mo.degre_extrap = ORC_EXTRAP_VALUE; mo.valeur_extrap = 1.0f;
    orc_gridset *gs = orc_defset(gdout, sg->mask[0]);
    float *zero = (float *)calloc((size_t)sg->mask[0]->ni * sg->mask[0]->nj, sizeof(float));
    float *mask = (float *)malloc(sizeof(float) * npts);
    orc_ezsint(gs, &mo, mask, zero);
    free(zero); orc_gridset_free(gs);
    int nyin = 0, nyan = 0;
    for (int k = 0; k < npts; k++) if (mask[k] == 1.0f) nyan++; else nyin++;
    float *la[2], *lo[2];
    for (int s = 0; s < 2; s++) {
        int n = s ? nyan : nyin;
        la[s] = (float *)malloc(sizeof(float) * (n + 1)); lo[s] = (float *)malloc(sizeof(float) * (n + 1));
        sg->x[s] = (float *)malloc(sizeof(float) * (n + 1)); sg->y[s] = (float *)malloc(sizeof(float) * (n + 1));
        sg->lat[s] = la[s]; sg->lon[s] = lo[s];
    }
    nyin = nyan = 0;
    for (int k = 0; k < npts; k++) {
        if (mask[k] == 1.0f) { la[1][nyan] = lat[k]; lo[1][nyan] = lon[k]; nyan++; }
        else { la[0][nyin] = lat[k]; lo[0][nyin] = lon[k]; nyin++; }
    }
    sg->count[0] = nyin; sg->count[1] = nyan; sg->mask_out = mask;
    orc_gdxyfll(sg->sub[0], sg->x[0], sg->y[0], la[0], lo[0], nyin);       /* c_gdxyfll_orig: the list's lon may be modified */
    orc_gdxyfll(sg->sub[1], sg->x[1], sg->y[1], la[1], lo[1], nyan);
    sg->full_lat = lat; sg->full_lon = lon;
    sg->planned_for = gdout;
    return 0;
}

static void yy_merge(const orc_supergrid *sg, int npts, float *zout, const float *vyin, const float *vyan)
{
    int a = 0, b = 0;
    for (int k = 0; k < npts; k++) zout[k] = (sg->mask_out[k] == 1.0f) ? vyan[b++] : vyin[a++];
}

/* c_ezyysint, yyin == 1 && yyout == 0 (ezyysint.c:127-160): c_gdxysval on each subgrid, no polar correction */
int orc_ezyysint(orc_supergrid *sg, orc_grid *gdout, const orc_opts *o, float *zout, const float *zin)
{
    yy_plan(sg, gdout, o);
    int npts = gdout->ni * gdout->nj;
    size_t nsub = (size_t)sg->sub[0]->ni * sg->sub[0]->nj;
    float *v0 = (float *)malloc(sizeof(float) * (sg->count[0] + 1)), *v1 = (float *)malloc(sizeof(float) * (sg->count[1] + 1));
    orc_gdinterp(sg->sub[0], o->degre_interp, v0, zin, sg->x[0], sg->y[0], sg->count[0]);
    orc_gdinterp(sg->sub[1], o->degre_interp, v1, zin + nsub, sg->x[1], sg->y[1], sg->count[1]);
    yy_merge(sg, npts, zout, v0, v1);
    free(v0); free(v1);
    return 0;
}

/* c_ezyyuvint, yyin == 1 && yyout == 0 (ezyyuvint.c:164-215): c_gdxyvval + c_gdwdfuv per subgrid on its list, merge of
 * speed / direction, then ONE c_gduvfwd on the target with its full lat/lon */
static int yy_uv(orc_supergrid *sg, orc_grid *gdout, const orc_opts *o, float *uuout, float *vvout, const float *uuin, const float *vvin, int wd_only)
{
    yy_plan(sg, gdout, o);
    int npts = gdout->ni * gdout->nj;
    size_t nsub = (size_t)sg->sub[0]->ni * sg->sub[0]->nj;
    float *spd[2], *dir[2];
    for (int s = 0; s < 2; s++) {
        int n = sg->count[s];
        float *u = (float *)malloc(sizeof(float) * (n + 1)), *v = (float *)malloc(sizeof(float) * (n + 1));
        spd[s] = (float *)malloc(sizeof(float) * (n + 1)); dir[s] = (float *)malloc(sizeof(float) * (n + 1));
        orc_gdinterp(sg->sub[s], o->degre_interp, u, uuin + s * nsub, sg->x[s], sg->y[s], n);
        orc_gdinterp(sg->sub[s], o->degre_interp, v, vvin + s * nsub, sg->x[s], sg->y[s], n);
        orc_gdwdfuv(sg->sub[s], spd[s], dir[s], u, v, sg->lat[s], sg->lon[s], n);
        free(u); free(v);
    }
    float *fs = (float *)malloc(sizeof(float) * npts), *fd = (float *)malloc(sizeof(float) * npts);
    yy_merge(sg, npts, fs, spd[0], spd[1]);
    yy_merge(sg, npts, fd, dir[0], dir[1]);
    if (wd_only) { memcpy(uuout, fs, sizeof(float) * npts); memcpy(vvout, fd, sizeof(float) * npts); }      /* c_ezyywdint (ezyywdint.c:157-205): speed / direction are the result */
    else orc_gduvfwd(gdout, uuout, vvout, fs, fd, sg->full_lat, sg->full_lon, npts);
    for (int s = 0; s < 2; s++) { free(spd[s]); free(dir[s]); }
    free(fs); free(fd);
    return 0;
}

int orc_ezyyuvint(orc_supergrid *sg, orc_grid *gdout, const orc_opts *o, float *uuout, float *vvout, const float *uuin, const float *vvin)
{ return yy_uv(sg, gdout, o, uuout, vvout, uuin, vvin, 0); }
int orc_ezyywdint(orc_supergrid *sg, orc_grid *gdout, const orc_opts *o, float *spdout, float *dirout, const float *uuin, const float *vvin)
{ return yy_uv(sg, gdout, o, spdout, dirout, uuin, vvin, 1); }
