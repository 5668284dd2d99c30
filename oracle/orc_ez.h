/*
 * oracle/orc_ez.h -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * Plain-C restatement of librmn's EZ interpolation hot path (SURVEY.md section 8a rows
 * a1-a24).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 * Every function cites the reference file:line it follows.  Parity of this oracle is pinned
 * against oracle/_ref/libezref.so (the reference's own sources, see oracle/build_ref.sh) by
 * tests/test_oracle_vs_ref.py and against the committed fixtures in tests/golden/.
 */
#ifndef ORC_EZ_H
#define ORC_EZ_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* interpolation / extrapolation degrees: src/interp/ez_def.h:44-61 */
enum { ORC_NEAREST = 0, ORC_LINEAR = 1, ORC_CUBIC = 3 };
enum { ORC_EXTRAP_MAXIMUM = 4, ORC_EXTRAP_MINIMUM = 5, ORC_EXTRAP_VALUE = 6, ORC_EXTRAP_ABORT = 13 };
enum { ORC_Z_DEHORS = 0, ORC_Z_AU_NORD = 1, ORC_Z_AU_SUD = 2, ORC_Z_POLE_NORD = 3, ORC_Z_POLE_SUD = 4 };

/* subset of _Grille (src/interp/ez_def.h:204-222) that the hot path reads */
typedef struct {
    char grtyp, grref;
    int ni, nj;
    int ig[4], igref[4];
    float xg[4], xgref[4];
    int i1, i2, j1, j2, extension;
    float *ax, *ay;          /* ax[ni], ay[nj]; NULL for regular grids */
    float *ncx, *ncy;        /* Newton coefficient tables, column-major (ni,6) / (nj,6) */
    float *lat, *lon;        /* cached by orc_calclatlon; locate mutates lon like the reference */
} orc_grid;

/* subset of _groptions (src/interp/ez_def.h:225-243); defaults src/interp/ez_statics.c:17 */
typedef struct {
    int degre_interp;        /* 3 */
    int degre_extrap;        /* 4 = MAXIMUM */
    int polar_correction;    /* 1 */
    int vecteur;             /* 0 */
    float valeur_extrap;     /* 0 */
} orc_opts;

typedef struct { int npts; float *x, *y; int *idx; } orc_zone;

/* subset of _gridset (src/interp/ez_def.h:162-183) */
typedef struct {
    orc_grid *gdin, *gdout;
    float *x, *y;            /* source-grid fractional indices of every target point */
    int have_xy, have_zones;
    orc_zone zones[5];
} orc_gridset;

void orc_default_opts(orc_opts *o);

/* c_ezgdef_fmem / c_ezqkdef (src/interp/ezgdef_fmem.c:37-93) */
orc_grid *orc_grid_define(int ni, int nj, char grtyp, char grref,
                          int ig1, int ig2, int ig3, int ig4, const float *ax, const float *ay);
void orc_grid_free(orc_grid *g);

/* ez_glat -> dgauss -> ordleg (src/interp/ez_glat.inc, src/base/dgauss.F, ordleg.F) */
void orc_gauss_lat(float *lat, int nj);
/* ez_nwtncof (src/interp/ez_nwtncof.inc:20-178) */
void orc_nwtncof(float *cx, float *cy, const float *ax, const float *ay, int ni, int nj,
                 int j1, int j2, int extension);
/* ez_calclatlon (src/interp/ez_calclatlon.c:30-284); c_gdll */
int orc_calclatlon(orc_grid *g);
/* c_gdxyfll_orig (src/interp/gdxyfll.c:90-139): lon[] may be modified in place */
int orc_gdxyfll(const orc_grid *g, float *x, float *y, const float *lat, float *lon, int n);
void orc_ez_avg(float *zout, const float *xx, const float *yy, int nid, int njd, const float *zin, int nis, int njs, int ext);      /* ez_avg.inc */
int orc_gdxyfll_public(const orc_grid *g, float *x, float *y, const float *lat, float *lon, int n);      /* c_gdxyfll; orc_gdxyfll is c_gdxyfll_orig */
/* c_gdinterp (src/interp/gdinterp.c:96-313) with an explicit degree */
int orc_gdinterp(const orc_grid *gdin, int degree, float *zout, const float *zin,
                 const float *x, const float *y, int npts);

/* the 11 leaf kernels, exposed for differential tests against the reference's Fortran */
void orc_rgdint_0(float *zo, const float *px, const float *py, int npts, const float *z, int ni, int j1, int j2);
void orc_rgdint_1_nw(float *zo, const float *px, const float *py, int npts, const float *z, int ni, int j1, int j2);
void orc_rgdint_1_w(float *zo, const float *px, const float *py, int npts, const float *z, int ni, int j1, int j2, int wrap);
void orc_rgdint_3_nw(float *zo, const float *px, const float *py, int npts, const float *z, int ni, int j1, int j2);
void orc_rgdint_3_w(float *zo, const float *px, const float *py, int npts, const float *z, int ni, int j1, int j2, int wrap);
void orc_rgdint_3_wnnc(float *zo, const float *px, const float *py, int npts, const float *z, int ni, int j1, int j2, int wrap);
void orc_irgdint_1_nw(float *zo, const float *px, const float *py, int npts, const float *ax, const float *ay, const float *z, int ni, int nj);
void orc_irgdint_1_w(float *zo, const float *px, const float *py, int npts, const float *ax, const float *ay, const float *z, int ni, int j1, int j2, int wrap);
void orc_irgdint_3_nw(float *zo, const float *px, const float *py, int npts, const float *ax, const float *ay, const float *cx, const float *cy, const float *z, int i1, int i2, int j1, int j2);
void orc_irgdint_3_w(float *zo, const float *px, const float *py, int npts, const float *ax, const float *ay, const float *cx, const float *cy, const float *z, int ni, int j1, int j2, int wrap);
void orc_irgdint_3_wnnc(float *zo, const float *px, const float *py, int npts, const float *ax, const float *ay, const float *z, int ni, int j1, int j2, int wrap);

/* c_ezdefset (src/interp/ezdefset.c:38-171) */
orc_gridset *orc_defset(orc_grid *gdout, orc_grid *gdin);
void orc_gridset_free(orc_gridset *gs);
/* ez_calcxy (src/interp/ez_calcxy.c:28-137) */
int orc_calcxy(orc_gridset *gs);
/* ez_defzones (src/interp/ez_defzones.c:25-113) */
int orc_defzones(orc_gridset *gs);
void orc_permut(float *z, int ni, int nj);      /* src/base/permut.f */
/* ez_xpnsrcgd (src/interp/ez_xpnsrcgd.c:28-52): hemispheric A / B source -> its global expansion zout(ni, j1:j2) */
void orc_xpnsrcgd(const orc_grid *g, float *zout, const float *zi, int symetrie);
/* c_ezsint_orig (src/interp/ezsint.c:75-135) */
int orc_ezsint(orc_gridset *gs, const orc_opts *o, float *zout, const float *zin);
/* c_ezuvint_orig (src/interp/ezuvint.c:51-94) */
int orc_ezuvint(orc_gridset *gs, const orc_opts *o, float *uuout, float *vvout,
                const float *uuin, const float *vvin);
int orc_ezwdint(orc_gridset *gs, const orc_opts *o, float *uuout, float *vvout, const float *uuin, const float *vvin);

/* wind conversions (src/interp/gdwdfuv.c, gduvfwd.c) exposed for tests */
int orc_gdwdfuv(orc_grid *g, float *spd, float *dir, const float *uu, const float *vv,
                const float *lat, const float *lon, int npts);
int orc_gduvfwd(orc_grid *g, float *uu, float *vv, const float *spd, const float *dir,
                const float *lat, const float *lon, int npts);

/* masked interpolation helpers (oracle/orc_mask.c; src/interp/ez_mask.c) */
int orc_ezsint_mask(orc_gridset *gs, int cloud_linear, int *mask_out, const int *mask_in);
int orc_ezget_mask_zones(orc_gridset *gs, int *mask_zones, const int *mask_in);
void orc_mask_fill2(float *fld, const int *mask, int n);

/* Yin-Yang 'U' source grids (oracle/orc_yy.c): c_ezgdef_supergrid + c_ezgdef_yymask, c_ezyysint / c_ezyyuvint towards one
 * ordinary target grid */
typedef struct {
    orc_grid *sub[2], *mask[2];
    int ni, nj;
    orc_grid *planned_for;            /* target grid of the cached plan (the reference caches it in the grid set) */
    float *mask_out;                  /* 1.0: the point is served by Yang */
    int count[2];
    float *x[2], *y[2], *lat[2], *lon[2];
    float *full_lat, *full_lon;
} orc_supergrid;
orc_supergrid *orc_supergrid_define(orc_grid *yin, orc_grid *yan);
int orc_ezyysint(orc_supergrid *sg, orc_grid *gdout, const orc_opts *o, float *zout, const float *zin);
int orc_ezyyuvint(orc_supergrid *sg, orc_grid *gdout, const orc_opts *o, float *uuout, float *vvout, const float *uuin, const float *vvin);
int orc_ezyywdint(orc_supergrid *sg, orc_grid *gdout, const orc_opts *o, float *spdout, float *dirout, const float *uuin, const float *vvin);

#ifdef __cplusplus
}
#endif
#endif
