/*
 * ez_host.c -- C host front-end of the MI355X-native EZ interpolator (librmn_ez_hip.so).
 *
 * Mirrors the reference's EZSCINT front-end (src/interp/ez*.c, gd*.c): a global grid table, per
 * (gdout, gdin) grid sets, thread-local "current set" and options, string-keyed option setters,
 * and the c_ezsint / c_ezuvint / c_gdxysint entry points with the reference's names, argument
 * meaning and return codes (include/ezscint_hip.h cites the reference line of each).
 *
 * What is different by design: the O(npts) loops run on the GPU through the HIP shim
 * (ezhip_shim.h).  The host only does O(ni + nj) setup: grid descriptors, Gaussian latitudes,
 * Newton coefficient tables, and -- for rectilinear source/target pairs -- the per-column /
 * per-row tap tables of the separable kernel.  There is NO CPU fallback for the interpolation
 * itself: without a usable HIP device the interpolation entry points fail loudly (-1).
 *
 * Scope: grid types L, A, B, G (global), E, Z on L / E, polar-stereographic N / S, and 'Y' point clouds on 'L' as TARGETS.  Other types are rejected (-1).
 */
#include <ctype.h>
#include <math.h>
#include <pthread.h>
#include <unistd.h>
#include "libm_exact.h"
#include <sys/mman.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ezhip_shim.h"
#include "../../include/ezscint_hip.h"

/* ------------------------------------------------------------------------------------------ */
/* types                                                                                        */
/* ------------------------------------------------------------------------------------------ */
enum { DEG_NEAREST = 0, DEG_LINEAR = 1, DEG_CUBIC = 3 };
enum { XT_MAXIMUM = 4, XT_MINIMUM = 5, XT_VALUE = 6, XT_ABORT = 13 };

typedef struct {            /* src/interp/ez_def.h:225-243, defaults src/interp/ez_statics.c:17 */
    int degre_interp, degre_extrap, use_1subgrid, valeur_1subgrid, verbose, polar_correction;
    int wgt_num, msg_pt_tol, cld_interp_alg;
    float msg_gridpt_dist, msg_dist_thresh, valeur_extrap;
} ezh_opts;

typedef struct {            /* one compiled separable plan: device tables + the descriptor */
    int built;
    ezhip_sep_plan p;
    void *dev[40];          /* owned device allocations */
    /* host copy of the k_sepx row geometry (host-pointer ABI, sepx_host_chunks): per valid row-block its 16-row target block and the end of
     * its source window; per target row the special flag */
    int h_nvb, *h_vb_by, *h_vb_send;
    unsigned char *h_rflag;
} ezh_sepplan;

typedef struct ezh_set {
    int gdin, gdout;
    struct ezh_set *next;
    int extrap;             /* ez_defzones: 1 = EZ_EXTRAP (DEHORS zone), 0 = EZ_NO_EXTRAP (polar zones) */
    int have_dehors;        /* EZ_EXTRAP and at least one target point lies outside */
    int sep_capable;        /* geometry allows the separable kernel */
    float *x1d, *y1d;       /* host: located x per target column, y per target row (separable) */
    float ypole_n, ypole_s;
    ezh_sepplan sep[3][2];  /* [degree index][vector_mode] */
    float *d_x, *d_y;       /* per-point located coordinates (generic mode) */
    float *d_scratch;       /* float[8]: fill value + keys, pole values */
    float *d_poles_batch; int poles_cap;   /* pole values of a c_ezsint_batch_dev batch (2 per field) */
    float *d_prow;          /* 2 * ni_src synthetic polar wind rows (vector mode), u then v: [u_n, u_s, v_n, v_s] */
    int prow_pairs;         /* pairs d_prow holds rows for (1; more after a c_ezuvint_batch_dev) */
    float *d_avg[2];        /* interp_degree = average / sph_average: [x | row widening | y_low | y_high] of the target cells (ez_avg.inc:55-78, ez_avg_sph.inc:63-98) */
    void *d_uvt_tiles; int uvt_shape, uvt_cap, uvt_stats[4];
    void *d_st_tiles[2], *d_st_streams[2]; int st_key[2], st_cap[2], st_stats[2][4];      /* k_st1 / k_st (slot 0: bilinear, 1: bicubic): the tile table (scalar zone rules) and the tile-ordered {x, y} copy; st_key: the zone options they were built under, -1: tried, not worth it */
    size_t cache_bytes;                    /* HBM this set's staged-tile caches hold (tile tables + tile-ordered stream copies): counted against g_cache_budget */
    void *d_uvt_streams; const void *uvt_streams_M;      /* the tile-ordered {x, y, a, b} copy and the matrix array it was packed from (another matrix: the plain arrays are read) */      /* k_uvt's tile table over the set's x, y (built with the special-point cache, under the same key) */
    int *d_sspec_list; float *d_sspec_xy; int sspec_count, sspec_key;      /* the same for the bicubic SCALAR staged-tile launch (k_st): pole points and polar strips under the zone options of st_key[1]; 0: none kept */
    int *d_cspec_list; float *d_cspec_xy; int cspec_count, cspec_key;      /* the special points (polar strips, re-interpolated extrapolation) of the wind-pair launch, kept after its first call: index, x, y; key = the zone options they were listed under (0: none yet) */
    float *d_lamb_cs[2];    /* c_ezuvint from / towards a Lambert '!' grid: {cos, sin} of its rotation angle at the target points' longitudes (source leg, target leg) */
    void *d_windM;          /* c_ezuvint through a rotated frame: the wind chain of this grid pair as a 2 x 2 matrix per target point (built on first use) */
    int windM_half;         /* every point's matrix is a pure rotation to rounding (ezhip_wind_matrix's max_dev <= 4e-6): the packed rotation (one word per point) behind the float4 form is used */
    float windM_dev;
    /* Yin-Yang 'U' source (c_ezyy_calcxy, ezyy_calcxy.c): per subgrid the list of target points it serves */
    int yy_ready, yy_count[2];
    float *d_yy_x[2], *d_yy_y[2], *d_yy_lat[2], *d_yy_lon[2], *d_yy_tmp[4];
    int *d_yy_idx[2];
} ezh_set;

typedef struct {
    int used, refcount;
    char grtyp, grref;
    char shown;             /* the type the caller gave when it differs from the one computed with: '#' (a regional tile through c_ezgdef_fmem is a 'Z' grid, ez_defaxes.c:44-55) */
    int ni, nj, ig[4], igref[4];
    float xg[4], xgref[4];
    int i1, i2, j1, j2, extension;
    float *ax, *ay, *ncx, *ncy;            /* host */
    float *d_ax, *d_ay, *d_ncx, *d_ncy;    /* device mirrors */
    float *d_ncx8, *d_ncy8;                /* Newton coefficients laid out [index][8] for the per-point kernel */
    double *d_xrec10, *d_yrec10;           /* k_uvt: REAL*8 {ax(i-1), ax(i), ax(i+1), c1 .. c6, c5 + c2} per column i (index i - 1), the same per row */
    float *d_xrec8, *d_yrec8;              /* the wind pair kernels: REAL {ax(i-1 .. i+2), the four Lagrange denominators' reciprocals} per column i (index i - 1), the same per row */
    /* this grid as a TARGET: its lat/lon.  Separable grids keep 1-D arrays. */
    int coords_ready, separable;
    float *lat1d, *lon1d, *lat2d, *lon2d;
    float *d_lat, *d_lon; int d_coords_valid;
    double *d_windtrig;                    /* this grid as a separable wind TARGET: {cos,sin}(lon) per column, then {cos,sin}(lat) per row */
    float *d_plon2; float xg4_pole[2];     /* this grid as a wind SOURCE: longitudes of its last / first row, PS-frame xg4 (N, S) */
    ezh_set *sets;                          /* sets having this grid as gdout */
    int *mask;                              /* c_gdsetmask */
    int nsub, sub[2], maskgrid[2];          /* 'U' (Yin-Yang): the two Z-on-E subgrids and their core ("mask") grids */
    float lamb[14];                         /* '!' (Lambert conformal): the xglst of igaxg95 (qgblukig, igaxg95.F:121-150) from the gribtable entry of ig1 */
} ezh_grid;

/* The grid table never moves: one lazily backed virtual reservation of the reference's capacity (1024 chunks of 128 grids,
 * gd_key2rowcol.h:24-27).  A thread that interpolates holds &G[gd] while another thread defines grids (the reference has
 * chunked tables for the same reason, ez_statics.c:6-10); a table grown by realloc() moved under them. */
#define EZH_MAX_GRIDS (1024 * 128)
static ezh_grid *G = NULL;
static int nG = 0;
static int grid_table_init(void)
{
    if (G) return 0;
    void *p = mmap(NULL, sizeof(ezh_grid) * (size_t)EZH_MAX_GRIDS, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (p == MAP_FAILED) { fprintf(stderr, "<ezscint> cannot reserve the grid table\n"); return -1; }
    G = (ezh_grid *)p;
    return 0;
}
/* a new slot (under g_mtx): the first released slot, else the next one; -1 when the table is full */
static int grid_slot(void)
{
    if (grid_table_init()) return -1;
    int gd;
    for (gd = 0; gd < nG; gd++) if (!G[gd].used) return gd;
    if (nG >= EZH_MAX_GRIDS) { fprintf(stderr, "<ezscint> grid table full (%d grids)\n", EZH_MAX_GRIDS); return -1; }
    memset(&G[nG], 0, sizeof(ezh_grid));
    __atomic_store_n(&nG, nG + 1, __ATOMIC_RELEASE);        /* readers (grid_ok) take no lock: entries never move, `used` is set last */
    return gd;
}
static pthread_mutex_t g_mtx = PTHREAD_MUTEX_INITIALIZER;

static __thread int cur_gdin = -1, cur_gdout = -1;
static __thread ezh_opts O = {DEG_CUBIC, XT_MAXIMUM, 0, -1, 0, 1, 16, 0, 4, 0.5f, 3.0f, 0.0f};

static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

/* ------------------------------------------------------------------------------------------ */
/* descriptor codecs: CIGAXG / CXGAIG (src/base/igaxg.f90:51-135, src/base/xgaig.f90:57-192)     */
/* ------------------------------------------------------------------------------------------ */
static void h_cigaxg(char t, float *x1, float *x2, float *x3, float *x4, int i1, int i2, int i3, int i4)
{
    *x1 = *x2 = *x3 = *x4 = 0.f;
    if (t == 'A' || t == 'B' || t == 'G') { *x1 = (float)i1; *x2 = (float)i2; }
    else if (t == 'L') {
        *x1 = (float)i3 * 0.01f - 90.f; *x2 = (float)i4 * 0.01f;
        *x3 = (float)i1 * 0.01f;        *x4 = (float)i2 * 0.01f;
    } else if (t == 'E') {
        int b = i3 & 3, l3 = i3 >> 2, l1 = (i1 << 2) | b;
        b = i4 & 3;
        int l4 = i4 >> 2, l2 = (i2 << 2) | b;
        if (l2 > 3600) l2 -= 7201;
        *x1 = (float)(((double)l1 - 3600.0) / 40.0);
        if (l3 < 3559) l3 += 16384;
        *x2 = (float)(((double)l3 - 3600.0) / 40.0);
        *x3 = (float)((double)l2 / 40.0);
        *x4 = (float)((double)l4 / 40.0);
    } else if ((t == 'N' || t == 'S') && i4 < 32768) {
        *x1 = (float)i2 * 0.1f; *x2 = (float)i1 * 0.1f; *x3 = (float)i4 * 100.f; *x4 = (float)i3 * 0.01f;
    }
}

static int nintf_(float v) { return (int)lroundf(v); }

static void h_cxgaig(char t, int *i1, int *i2, int *i3, int *i4, float x1, float x2, float x3, float x4)
{
    *i1 = *i2 = *i3 = *i4 = 0;
    if (t == 'A' || t == 'B' || t == 'G') { *i1 = (int)x1; *i2 = (int)x2; }
    else if (t == 'L') {
        *i1 = nintf_(x3 * 100.f); *i2 = nintf_(x4 * 100.f);
        *i3 = nintf_((90.f + x1) * 100.f); *i4 = nintf_(x2 * 100.f);
        while (*i4 < 0) *i4 += 36000;
    } else if (t == 'N' || t == 'S') {
        *i1 = nintf_(x2 * 10.f); *i2 = nintf_(x1 * 10.f); *i3 = nintf_(x4 * 100.f); *i4 = nintf_(x3 * 0.01f);
        while (*i3 < 0) *i3 += 36000;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* Gaussian latitudes: ez_glat.inc:26-67 -> DGAUSS (dgauss.F:90-131) -> ORDLEG (ordleg.F:50-86)   */
/* all REAL arithmetic with the float libm, as the reference                                    */
/* ------------------------------------------------------------------------------------------ */
/* ORDLEG's leading coefficient: a product over n = 1 .. ir that depends on ir alone (the reference recomputes it in every call: the same REAL operations, the same value) */
static float legendre_c1(int ir)
{
    float c1 = sqrtf(2.f);
    for (int n = 1; n <= ir; n++) {
        const float fn2 = 2.f * (float)n, q = fn2 * fn2;
        c1 = c1 * sqrtf(1.0f - 1.0f / q);
    }
    return c1;
}
static float legendre_norm(float coa, int ir, float c1)
{
    const float theta = acosf(coa), fn = ir >= 1 ? (float)ir : 0.f, fn2 = ir >= 1 ? 2.f * fn : 0.f;
    float ang = fn * theta, s1 = 0.0f, c4 = 1.0f, a = -1.0f, b = 0.0f;
    for (int kk = 1; kk <= ir + 1; kk += 2) {
        int k = kk - 1;
        if (k == ir) c4 = 0.5f * c4;
        s1 = s1 + c4 * cosf(ang);
        a = a + 2.0f; b = b + 1.0f;
        ang = theta * (fn - (float)k - 2.0f);
        c4 = (a * (fn - b + 1.0f) / (b * (fn2 - a))) * c4;
    }
    return s1 * c1;
}

/* the Newton iterations of the roots are independent of each other (dgauss.F:107-125): slices of them on host threads, every root by the reference's own operations */
typedef struct { float *rt; int nj, i0, i1; float normn, normnm, c1n, c1nm; } glat_job;
static void *glat_worker(void *arg)
{
    glat_job *j = (glat_job *)arg;
    const float tol = 1.0e-6f;
    float *rt = j->rt;
    const int nj = j->nj;
    for (int i = j->i0; i < j->i1; i++) {
        float delta;
        do {
            float pn = j->normn * legendre_norm(rt[i], nj, j->c1n);
            float pnm = j->normnm * legendre_norm(rt[i], nj - 1, j->c1nm);
            float rdpdx = (rt[i] * rt[i] - 1.0f) / ((float)nj * (rt[i] * pn - pnm));
            delta = -pn * rdpdx;
            rt[i] = rt[i] + delta;
        } while (fabsf(delta) > tol);
        rt[nj + 1 - i] = -rt[i];
    }
    return NULL;
}
static void h_gauss_lat(float *lat, int nj)
{
    const float pie = 3.1415926535898f, rdtodg = 57.295779513082f;
    float *rt = (float *)calloc((size_t)nj + 2, sizeof(float));
    float normn = sqrtf(2.0f / (2.0f * (float)nj + 1.0f)), normnm = sqrtf(2.0f / (2.0f * (float)nj - 1.0f));
    int half = nj / 2;
    for (int i = 1; i <= half; i++) {
        float t = (float)(4 * i - 1) * pie / (float)(4 * nj + 2);
        rt[i] = cosf(t + 1.0f / (8.0f * (float)(nj * nj) * tanf(t)));
    }
    {
        long ncpu = sysconf(_SC_NPROCESSORS_ONLN);
        int nt = half >= 256 ? (int)(ncpu > 16 ? 16 : ncpu < 1 ? 1 : ncpu) : 1;
        if (getenv("EZHIP_GLAT_THREADS")) nt = atoi(getenv("EZHIP_GLAT_THREADS"));
        if (nt < 1) nt = 1;
        if (nt > 16) nt = 16;
        glat_job job[16]; pthread_t th[16];
        const float c1n = legendre_c1(nj), c1nm = legendre_c1(nj - 1);
        for (int t = 0; t < nt; t++) {
            job[t] = (glat_job){rt, nj, 1 + (int)((long long)half * t / nt), 1 + (int)((long long)half * (t + 1) / nt), normn, normnm, c1n, c1nm};
            if (t == nt - 1 || pthread_create(&th[t], NULL, glat_worker, &job[t])) { glat_worker(&job[t]); th[t] = 0; }
        }
        for (int t = 0; t < nt - 1; t++) if (th[t]) pthread_join(th[t], NULL);
    }
    if (nj % 2) rt[half + 1] = 0.0f;
    /* ez_glat: reverse to south->north, then degrees */
    for (int j = 1; j <= nj; j++) lat[j - 1] = 90.f - rdtodg * acosf(rt[nj + 1 - j]);
    free(rt);
}

/* ------------------------------------------------------------------------------------------ */
/* Newton coefficient tables: ez_nwtncof.inc:20-178                                             */
/* ------------------------------------------------------------------------------------------ */
static void recip6(float *c, int ld, float a, float b, float cc, float d)
{
    c[0] = 1.f / (b - a); c[ld] = 1.f / (cc - a); c[2 * ld] = 1.f / (cc - b);
    c[3 * ld] = 1.f / (d - a); c[4 * ld] = 1.f / (d - b); c[5 * ld] = 1.f / (d - cc);
}

/* rows of the latitude table: a hemispheric Gaussian grid keeps the 2 nj latitudes of the globe (ez_defaxes.c:93-99), its field is expanded
 * into rows j1 .. j2 of that table */
static int ay_rows(const ezh_grid *g) { return (g->grtyp == 'G' && g->ig[0] != 0) ? 2 * g->nj : g->nj; }
/* the tables themselves: cx[6][ni], cy[6][nr] (nr rows of the latitude table: ay(j1:j2) of the reference) */
static void nwtncof_tables(float *cx, float *cy, const float *ax, const float *ay, int ni, int nr, int ext)
{
    for (int i = 0; i + 1 < ni; i++) if (ax[i + 1] <= ax[i]) { fprintf(stderr, "ez_nwtncof: x axis must be strictly increasing\n"); exit(13); }
    for (int j = 0; j + 1 < nr; j++) if (ay[j + 1] <= ay[j]) { fprintf(stderr, "ez_nwtncof: y axis must be strictly increasing\n"); exit(13); }
    for (int k = 0; k < 6 * ni; k++) cx[k] = 1.0f;
    for (int k = 0; k < 6 * nr; k++) cy[k] = 1.0f;
    for (int i = 1; i + 2 < ni; i++) recip6(&cx[i], ni, ax[i - 1], ax[i], ax[i + 1], ax[i + 2]);
    for (int j = 1; j + 2 < nr; j++) recip6(&cy[j], nr, ay[j - 1], ay[j], ay[j + 1], ay[j + 2]);
    if (ext == 1) {
        recip6(&cx[0], ni, ax[0] - (ax[ni - 1] - ax[ni - 2]), ax[0], ax[1], ax[2]);
        recip6(&cx[ni - 2], ni, ax[ni - 3], ax[ni - 2], ax[ni - 1], ax[ni - 1] + (ax[1] - ax[0]));
    } else if (ext == 2) {
        recip6(&cx[0], ni, ax[0] - (360.0f - ax[ni - 1]), ax[0], ax[1], ax[2]);
        recip6(&cx[ni - 2], ni, ax[ni - 3], ax[ni - 2], ax[ni - 1], ax[0] + 360.0f);
        recip6(&cx[ni - 1], ni, ax[ni - 2], ax[ni - 1], ax[0] + 360.0f, ax[1] + 360.0f);
    }
}
static int h_nwtncof(ezh_grid *g)
{
    const int ni = g->ni, nr = ay_rows(g);
    g->ncx = (float *)malloc(sizeof(float) * 6 * ni);
    g->ncy = (float *)malloc(sizeof(float) * 6 * nr);
    nwtncof_tables(g->ncx, g->ncy, g->ax, g->ay, ni, nr, g->extension);
    return 0;
}
/* the reference's own entry point (f_ezscint.F90 / ez_nwtncof.inc:20: subroutine ez_nwtncof(cx, cy, ax, ay, ni, nj, i1, i2, j1, j2, extension), called from
 * ez_calcntncof.c:44 through f77name): the same routine the grid table above is filled by.  Host only. */
void ez_nwtncof_(float *cx, float *cy, const float *ax, const float *ay, const int32_t *ni, const int32_t *nj, const int32_t *i1, const int32_t *i2,
                 const int32_t *j1, const int32_t *j2, const int32_t *extension)
{
    (void)nj; (void)i1; (void)i2;
    nwtncof_tables(cx, cy, ax, ay, *ni, *j2 - *j1 + 1, *extension);
}

/* ------------------------------------------------------------------------------------------ */
/* ez_xpncof.c:48-226 : i1,i2,j1,j2 and the longitude "extension"                               */
/* ------------------------------------------------------------------------------------------ */
static void h_xpncof(ezh_grid *g)
{
    int ni = g->ni, nj = g->nj;
    g->i1 = 1; g->i2 = ni; g->j1 = 1; g->j2 = nj; g->extension = 0;
    if (g->grtyp == 'A' || g->grtyp == 'G') {              /* ez_xpncof.c:107-145: hemispheres are expanded into (j1:j2) */
        g->extension = 2;
        if (g->ig[0] == 1) { g->j1 = -nj + 1; g->j2 = nj; } else if (g->ig[0] == 2) { g->j1 = 1; g->j2 = 2 * nj; }
    } else if (g->grtyp == 'B') {
        g->extension = 1;
        if (g->ig[0] == 1) { g->j1 = -nj + 2; g->j2 = nj; } else if (g->ig[0] == 2) { g->j1 = 1; g->j2 = 2 * nj - 1; }
    }
    else if (g->grtyp == 'L') {
        float lat0, lon0, dlat, dlon;
        h_cigaxg('L', &lat0, &lon0, &dlat, &dlon, g->ig[0], g->ig[1], g->ig[2], g->ig[3]);
        if ((double)(lat0 - dlat) > (-90.0 + 0.01 * (double)dlat)) return;
        float lastlat = lat0 + (float)(nj - 1) * dlat;
        if ((double)(lastlat + dlat) < (90.0 - 0.01 * (double)dlat)) return;
        if (lon0 < 0.0) lon0 += 360.0;
        float lastlon = lon0 + (dlon * (float)(ni - 1));
        if ((double)(lastlon - lon0) > (360 - 0.01 * (double)dlon)) { g->extension = 1; return; }
        float extra = lastlon + dlon;
        if ((double)(extra - lon0) > (360.0 - (double)dlon * 0.01)) g->extension = 2;
    } else if (g->grtyp == 'Z' && g->grref == 'E') {
        g->extension = ((g->ax[ni - 1] - g->ax[0]) < 359.0) ? 0 : 1;
    } else if (g->grtyp == 'Z' && g->grref == 'L') {
        const float *ax = g->ax, *ay = g->ay;
        float dlat = ay[1] - ay[0];
        if ((double)(ay[0] - dlat) > (-90.0 + 0.01 * (double)dlat)) return;
        dlat = ay[nj - 1] - ay[nj - 2];
        if ((double)(ay[nj - 1] + dlat) < (90.0 - 0.01 * (double)dlat)) return;
        float dlon = ax[ni - 1] - ax[ni - 2];
        if ((double)(ax[ni - 1] - ax[0]) > (360.0 - 0.01 * (double)dlon)) g->extension = 1;
        else if ((double)((ax[ni - 1] + dlon) - ax[0]) > (360.0 - 0.01 * (double)dlon)) g->extension = 2;
    }
}

/* the reference's own entry point (ez_xpncof.c:48, declared in ez_funcdef.h:67): bounds of the source rows / columns and the longitude extension of a grid
 * given by its descriptors.  sym is unused there as here ('!' grids: no extension).  Host only. */
void ez_xpncof(int32_t *i1, int32_t *i2, int32_t *j1, int32_t *j2, int32_t *extension, int32_t ni, int32_t nj, char grtyp, char grref,
               int32_t ig1, int32_t ig2, int32_t ig3, int32_t ig4, int32_t sym, float *ax, float *ay)
{
    ezh_grid g;
    (void)sym;
    memset(&g, 0, sizeof(g));
    g.ni = ni; g.nj = nj; g.grtyp = grtyp; g.grref = grref; g.ig[0] = ig1; g.ig[1] = ig2; g.ig[2] = ig3; g.ig[3] = ig4; g.ax = ax; g.ay = ay;
    h_xpncof(&g);
    *i1 = g.i1; *i2 = g.i2; *j1 = g.j1; *j2 = g.j2; *extension = g.extension;
}

/* ------------------------------------------------------------------------------------------ */
/* rotated-frame helpers: ez_crot.inc, ez_lac.inc, ez_cal.inc, mxm.F90, ez_gfxyfll.c, ez_gfllfxy.c */
/* ------------------------------------------------------------------------------------------ */
static void h_lac(float *xyz, const float *lon, const float *lat, int n)
{
    float dar = acosf(-1.0f) / 180.0f;
    for (int i = 0; i < n; i++) {
        float cd = cosf(dar * lat[i]);
        xyz[3 * i] = cd * cosf(dar * lon[i]); xyz[3 * i + 1] = cd * sinf(dar * lon[i]); xyz[3 * i + 2] = sinf(dar * lat[i]);
    }
}
static void h_cal(float *lon, float *lat, const float *xyz, int n)
{
    float rad = 180.f / acosf(-1.00f);
    for (int i = 0; i < n; i++) {
        lat[i] = asinf(fmaxf(-1.00f, fminf(1.0f, xyz[3 * i + 2]))) * rad;
        lon[i] = atan2f(xyz[3 * i + 1], xyz[3 * i]) * rad;
        lon[i] = fmodf(lon[i], 360.0f);
        if (lon[i] < 0.0f) lon[i] = lon[i] + 360.0f;
    }
}
static void h_crot(float *r, float *ri, float lon1, float lat1, float lon2, float lat2)
{
    float p[3], q[3];
    h_lac(p, &lon1, &lat1, 1); h_lac(q, &lon2, &lat2, 1);
    float a = (p[0] * q[0]) + (p[1] * q[1]) + (p[2] * q[2]);
    float t1 = (p[1] * q[2]) - (q[1] * p[2]), t2 = (q[0] * p[2]) - (p[0] * q[2]), t3 = (p[0] * q[1]) - (q[0] * p[1]);
    float b = sqrtf(t1 * t1 + t2 * t2 + t3 * t3);
    float c = sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
    float e1 = ((a * p[0]) - q[0]) / b, e2 = ((a * p[1]) - q[1]) / b, e3 = ((a * p[2]) - q[2]) / b;
    float d = sqrtf(e1 * e1 + e2 * e2 + e3 * e3);
    /* Fortran r(i,j) -> r[(j-1)*3 + i-1] */
    r[0] = -p[0] / c; r[3] = -p[1] / c; r[6] = -p[2] / c;
    r[1] = e1 / d;    r[4] = e2 / d;    r[7] = e3 / d;
    r[2] = t1 / b;    r[5] = t2 / b;    r[8] = t3 / b;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) ri[j * 3 + i] = r[i * 3 + j];
}
static void h_mxm3(const float *a, const float *b, float *c, int n)
{
    for (int j = 0; j < n; j++) for (int i = 0; i < 3; i++) {
        float s = 0.0f;
        for (int k = 0; k < 3; k++) s = s + a[k * 3 + i] * b[3 * j + k];
        c[3 * j + i] = s;
    }
}
/* true -> rotated (forward = 1, ez_gfxyfll) or rotated -> true (forward = 0, ez_gfllfxy) */
static void h_rotate(float *lon_o, float *lat_o, const float *lon, const float *lat, int n, const float *xg4, int forward)
{
    float r[9], ri[9];
    h_crot(r, ri, xg4[1], xg4[0], xg4[3], xg4[2]);
    /* in pieces of 2048 points through two stack buffers: as whole-array temporaries (2 x 12 bytes per point from malloc, fresh pages every call) the first touch of
     * those pages -- serialised in the kernel whatever the number of threads -- was most of the 60 ms the set's first call spent locating cfg3's 8 M points */
    enum { PIECE = 2048 };
    float cart[3 * PIECE], rot[3 * PIECE];
    for (int o = 0; o < n; o += PIECE) {
        const int c = n - o < PIECE ? n - o : PIECE;
        h_lac(cart, lon + o, lat + o, c);
        h_mxm3(forward ? r : ri, cart, rot, c);
        h_cal(lon_o + o, lat_o + o, rot, c);
    }
}

/* ------------------------------------------------------------------------------------------ */
/* host locate (exact restatement; used for the 1-D tables and for c_gdxyfll)                    */
/* ez_ll2rgd.inc:22-166, ez_ll2igd.inc:22-86, ez_llll2gd.inc:22-53, ez_cherche.inc:53-69          */
/* ------------------------------------------------------------------------------------------ */
static int h_cherche(float v, const float *tab, int n)
{
    int lo = 1, hi = n, mid = (int)((float)(lo + hi) * 0.5f);
    while (mid != lo) {
        if (v <= tab[mid - 1]) hi = mid; else lo = mid;
        mid = (int)((float)(lo + hi) * 0.5f);
    }
    return mid;
}
static void h_llll2gd(float *x, float *y, const float *lat, float *lon, int n, float lat0, float lon0, float dlat, float dlon, float lonref)
{
    if (lonref == -180.0f) { for (int i = 0; i < n; i++) if (lon[i] > 180.0f) lon[i] = lon[i] - 360.0f; }
    else { for (int i = 0; i < n; i++) if (lon[i] < 0.0f) lon[i] = lon[i] + 360.0f; }
    for (int i = 0; i < n; i++) { x[i] = (lon[i] - lon0) / dlon + 1.0f; y[i] = (lat[i] - lat0) / dlat + 1.0f; }
}
/* lon[] is modified in place exactly where the reference modifies it */
/* polar-stereographic grids: LLFXY (src/base/llfxy.F:21-40, REAL arithmetic), GRPS (src/base/grps.f:43-58),
 * ez_vxyfll (src/interp/ez_vxyfll.inc:20-99; dgtord is REAL, so `dgtord*(dlon+dgrw)` is a REAL product widened on
 * assignment, while the SUD branch multiplies in double) */
static void h_llfxy(float *dlat, float *dlon, float x, float y, float d60, float dgrw, int nhem)
{
    const float rdtodg = 57.295779513082f;
    float re = 1.866025f * 6.371E+6f / d60, re2 = re * re;
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
    if (nhem == 2) { *dlat = -*dlat; *dlon = -*dlon; }
}
static void h_grps(float *xlat, float *xlon, int ni, int nj, float pi, float pj, float d60, float dgrw, int hem)
{
    for (int j = 1; j <= nj; j++) {
        float y = (float)j - pj;
        for (int i = 1; i <= ni; i++) {
            float xla, xlo;
            h_llfxy(&xla, &xlo, (float)i - pi, y, d60, dgrw, hem);
            if (xlo < 0) xlo = xlo + 360.f;
            xlat[(size_t)(j - 1) * ni + (i - 1)] = xla; xlon[(size_t)(j - 1) * ni + (i - 1)] = xlo;
        }
    }
}
static void h_vxyfll(float *x, float *y, const float *dlat, const float *dlon, int npts, float d60, float dgrw, float pi, float pj, int nhem)
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

/* ------------------------------------------------------------------------------------------ */
/* Lambert conformal '!' grids                                                 */
/* ------------------------------------------------------------------------------------------ */
/* The four descriptors of a '!' grid name an entry of the GRIB grid dictionary `gribtable` (igaxg95.F:63-110 -> qqqgbld :201-268, qgblukig :113-180):
 * a text file looked for in ./, $CMCCONST/ and $ARMNLIB/data/, read once; lines that start with '#' are comments, every other line is a list-directed
 * record  gribcode centercode subcentercode projcode grtyp lat(1,1) lon(1,1) dx[km] dy[km] yaxislon latin1 latin2 ni nj lat(ni,nj) lon(ni,nj) gds-text.
 * The FIRST line whose gribcode equals ig1 is the grid (ig2 is compared with the centre code but the entry is used either way, :164-171). */
#define LAMB_MAXGRIDS 1024
static struct { int n, state; int code[LAMB_MAXGRIDS]; char gty[LAMB_MAXGRIDS]; float xg[LAMB_MAXGRIDS][14]; } GBT;       /* state: 0 not read, 1 read, -1 no file */
static pthread_mutex_t g_gbt_mtx = PTHREAD_MUTEX_INITIALIZER;
static int gbt_tokens(char *line, char **tok, int maxtok)
{   /* list-directed input: blanks and commas separate, character values may be quoted */
    int n = 0;
    char *p = line;
    while (*p && n < maxtok) {
        while (*p == ' ' || *p == ',' || *p == '\t' || *p == '\r' || *p == '\n') p++;
        if (!*p) break;
        if (*p == '\'' || *p == '"') { const char q = *p++; tok[n++] = p; while (*p && *p != q) p++; if (*p) *p++ = 0; }
        else { tok[n++] = p; while (*p && *p != ' ' && *p != ',' && *p != '\t' && *p != '\r' && *p != '\n') p++; if (*p) *p++ = 0; }
    }
    return n;
}
static int gbt_load(void)
{
    pthread_mutex_lock(&g_gbt_mtx);
    if (GBT.state == 0) {
        char path[1024];
        FILE *f = fopen("gribtable", "r");
        if (!f && getenv("CMCCONST")) { snprintf(path, sizeof(path), "%s/gribtable", getenv("CMCCONST")); f = fopen(path, "r"); }
        if (!f && getenv("ARMNLIB")) { snprintf(path, sizeof(path), "%s/data/gribtable", getenv("ARMNLIB")); f = fopen(path, "r"); }
        GBT.state = f ? 1 : -1;
        char line[1024];
        while (f && fgets(line, sizeof(line), f) && GBT.n < LAMB_MAXGRIDS) {
            if (line[0] == '#') continue;
            char *t[20];
            if (gbt_tokens(line, t, 20) < 16) continue;
            float *x = GBT.xg[GBT.n];
            GBT.code[GBT.n] = atoi(t[0]); GBT.gty[GBT.n] = t[4][0];
            x[0] = strtof(t[5], NULL); x[1] = strtof(t[6], NULL); x[2] = strtof(t[7], NULL); x[3] = strtof(t[8], NULL);      /* lat(1,1) lon(1,1) dx dy */
            x[4] = strtof(t[9], NULL); x[5] = strtof(t[10], NULL); x[6] = strtof(t[11], NULL);                              /* yaxislon latin1 latin2 */
            x[7] = (float)atoi(t[12]) * 1.0f; x[8] = (float)atoi(t[13]) * 1.0f;                                             /* ni nj */
            x[9] = strtof(t[14], NULL); x[10] = strtof(t[15], NULL);                                                         /* lat(ni,nj) lon(ni,nj) */
            x[11] = (float)atoi(t[1]) * 1.0f; x[12] = (float)atoi(t[2]) * 1.0f; x[13] = (float)atoi(t[3]) * 1.0f;           /* centre, sub-centre, projection */
            GBT.n++;
        }
        if (f) fclose(f);
    }
    const int st = GBT.state;
    pthread_mutex_unlock(&g_gbt_mtx);
    return st;
}
/* the xglst of grid ig1; 0, or -1 (no table / no such code / not a Lambert entry) with the reference's message */
static int lamb_lookup(float *xg14, int ig1)
{
    if (gbt_load() < 0) { fprintf(stderr, " <qqqgbld> no \"gribtable\" file found (./gribtable, $CMCCONST/gribtable, $ARMNLIB/data/gribtable)\n"); return -1; }
    for (int k = 0; k < GBT.n; k++)
        if (GBT.code[k] == ig1) {
            if (GBT.gty[k] != 'H') { fprintf(stderr, " <igaxg95> GRIB code %d is not a Lambert conformal ('H') entry\n", ig1); return -1; }
            memcpy(xg14, GBT.xg[k], sizeof(float) * 14);
            return 0;
        }
    fprintf(stderr, " <qgblukig> GRIB code %d not found...\n", ig1);
    return -1;
}
/* ez_lambxyfll99.inc:20-68 / ez_lambllfxy99.inc:20-58: the projection in REAL, operation by operation */
typedef struct { float rn, rf, rhozero; } lamb_cone;
static lamb_cone h_lamb_cone(float latin1, float latin2, float yaxislat)
{
    const float pisur4 = atanf(1.0f), pi = 4.0f * pisur4, d2r = pi / 180.0f, r = 6370997.0f;
    const float rphi1 = d2r * latin1, rphi2 = d2r * latin2;
    lamb_cone c;
    if (rphi1 == rphi2) c.rn = sinf(rphi1);
    else { c.rn = logf(cosf(rphi1) / cosf(rphi2)); c.rn = c.rn / logf((tanf(pisur4 + 0.5f * rphi2)) / tanf(pisur4 + 0.5f * rphi1)); }
    const float rtan = tanf(pisur4 + rphi1 * 0.5f);
    const float f = (cosf(rphi1) * powf(rtan, c.rn)) / c.rn;
    c.rf = r * f;
    c.rhozero = r * f / powf(tanf(pisur4 + yaxislat * d2r * .5f), c.rn);
    return c;
}
static void h_lambxyfll99(float *x, float *y, const float *lat, const float *lon, int n, float latin1, float latin2, float yaxislat, float yaxislon)
{
    const float pisur4 = atanf(1.0f), pi = 4.0f * pisur4, d2r = pi / 180.0f;
    const lamb_cone c = h_lamb_cone(latin1, latin2, yaxislat);
    for (int i = 0; i < n; i++) {
        float tmplat = lat[i];
        if (tmplat > 90.0f) tmplat = 89.95f;
        const float rho = c.rf / powf(tanf(pisur4 + tmplat * 0.5f * d2r), c.rn);
        float dlon = lon[i] - yaxislon;
        if (dlon < -180.0f) dlon = dlon + 360.f; else if (dlon > 180.f) dlon = dlon - 360.f;
        const float theta = c.rn * (d2r * dlon);
        x[i] = rho * sinf(theta);
        y[i] = c.rhozero - rho * cosf(theta);
    }
}
static void h_lambllfxy99(float *lat, float *lon, const float *x, const float *y, int n, float latin1, float latin2, float yaxislat, float yaxislon)
{
    const float pisur4 = atanf(1.0f), pi = 4.0f * pisur4, d2r = pi / 180.0f;
    const lamb_cone c = h_lamb_cone(latin1, latin2, yaxislat);
    for (int i = 0; i < n; i++) {
        const float dy = c.rhozero - y[i];
        const float rho = copysignf(1.0f, c.rn) * sqrtf(x[i] * x[i] + (dy * dy));
        const float theta = atanf(x[i] / (c.rhozero - y[i]));
        lat[i] = (2.0f * atanf(powf(c.rf / rho, 1.0f / c.rn)) - 0.5f * pi) / d2r;
        lon[i] = theta / (d2r * c.rn) + yaxislon;
    }
}
/* ez_lambfll.inc:20-70 (lat, lon -> grid coordinates) and ez_llflamb.inc:20-62 (grid coordinates -> lat, lon) of a '!' grid */
static void h_lambfll(const ezh_grid *g, float *x, float *y, const float *lat, const float *lon, int n)
{
    const float *xg = g->lamb;
    const float xlat11 = xg[0], xlon11 = xg[1], yaxislat = 0.5f * (xg[0] + xg[9]), yaxislon = xg[4], dx = xg[2] * 1000.0f, dy = xg[3] * 1000.0f;
    float x11, y11;
    h_lambxyfll99(&x11, &y11, &xlat11, &xlon11, 1, xg[5], xg[6], yaxislat, yaxislon);
    h_lambxyfll99(x, y, lat, lon, n, xg[5], xg[6], yaxislat, yaxislon);
    for (int i = 0; i < n; i++) { x[i] = 1.0f + (x[i] - x11) / dx; y[i] = 1.0f + (y[i] - y11) / dy; }
}
static void h_llflamb(const ezh_grid *g, float *lat, float *lon, float *x, float *y, int n)          /* (x, y are overwritten with projection coordinates, as in the reference) */
{
    const float *xg = g->lamb;
    const float xlat11 = xg[0], xlon11 = xg[1], yaxislat = 0.5f * (xg[0] + xg[9]), yaxislon = xg[4], dx = xg[2] * 1000.0f, dy = xg[3] * 1000.0f;
    float x11, y11;
    h_lambxyfll99(&x11, &y11, &xlat11, &xlon11, 1, xg[5], xg[6], yaxislat, yaxislon);
    for (int i = 0; i < n; i++) { x[i] = x11 + dx * (x[i] - 1.0f); y[i] = y11 + dy * (y[i] - 1.0f); }
    h_lambllfxy99(lat, lon, x, y, n, xg[5], xg[6], yaxislat, yaxislon);
}

static void *upload(const void *h, size_t nbytes);
/* winds on a '!' grid (ez_lamb_llwfgdw.inc:41-56, ez_lamb_gdwfllw.inc:41-56): the grid is turned against true north by the angle of the meridian through
 * the point -- the direction from (45 N, lon) to (50 N, lon) in grid coordinates; to_wd: alpha = pi / 2 - atan2(dely, delx) (grid -> true components), else
 * atan2(dely, delx) - pi / 2 (true -> grid).  cs[2 i] = cos(alpha_i), cs[2 i + 1] = sin(alpha_i), all REAL as in the reference */
static int h_lamb_wind_cs(const ezh_grid *g, const float *lon, int n, int to_wd, float *cs)
{
    const float pie = 3.1415926535898f;
    float *w = (float *)calloc(5 * (size_t)(n > 0 ? n : 1), sizeof(float));
    if (!w) return -1;
    float *la = w, *x1 = w + n, *y1 = w + 2 * (size_t)n, *x2 = w + 3 * (size_t)n, *y2 = w + 4 * (size_t)n;
    for (int i = 0; i < n; i++) la[i] = 45.0f;
    h_lambfll(g, x1, y1, la, lon, n);
    for (int i = 0; i < n; i++) la[i] = 50.0f;
    h_lambfll(g, x2, y2, la, lon, n);
    for (int i = 0; i < n; i++) {
        const float delx = x2[i] - x1[i], dely = y2[i] - y1[i];
        const float alpha = to_wd ? pie * 0.50f - atan2f(dely, delx) : atan2f(dely, delx) - 0.5f * pie;
        cs[2 * i] = cosf(alpha); cs[2 * i + 1] = sinf(alpha);
    }
    free(w);
    return 0;
}
/* the same as a device array (NULL on failure) */
static float *lamb_wind_cs_dev(const ezh_grid *g, const float *lon, int n, int to_wd)
{
    float *cs = (float *)malloc(sizeof(float) * 2 * (size_t)(n > 0 ? n : 1));
    if (!cs) return NULL;
    if (h_lamb_wind_cs(g, lon, n, to_wd, cs)) { free(cs); return NULL; }
    float *d = (float *)upload(cs, sizeof(float) * 2 * (size_t)(n > 0 ? n : 1));
    ezhip_sync();
    free(cs);
    return d;
}

static __thread int t_locate_j2;       /* 1 inside c_gdxyfll_orig: the search length of gdxyfll.c (see the 'G' / 'Z' branch) */
static int h_locate(const ezh_grid *g, float *x, float *y, const float *lat, float *lon, int n)
{
    int ni = g->ni, nj = g->nj;
    float lat0, lon0, dlat, dlon;
    switch (g->grtyp) {
    case 'A':
        dlon = 360.0f / (float)ni; lon0 = 0.0f;
        if (g->ig[0] == 1) { dlat = 90.0f / (float)nj; lat0 = dlat * 0.5f; }
        else if (g->ig[0] == 2) { dlat = 90.0f / (float)nj; lat0 = -90.0f + dlat * 0.5f; }
        else { dlat = 180.0f / (float)nj; lat0 = -90.0f + dlat * 0.5f; }
        for (int i = 0; i < n; i++) if (lon[i] < 0.0f) lon[i] = lon[i] + 360.0f;
        h_llll2gd(x, y, lat, lon, n, lat0, lon0, dlat, dlon, 0.0f);
        return 0;
    case 'B':
        dlon = 360.0f / (float)(ni - 1); lon0 = 0.0f;
        if (g->ig[0] == 1) { dlat = 90.0f / (float)(nj - 1); lat0 = 0.0f; }
        else if (g->ig[0] == 2) { dlat = 90.0f / (float)(nj - 1); lat0 = -90.0f; }
        else { dlat = 180.0f / (float)(nj - 1); lat0 = -90.0f; }
        for (int i = 0; i < n; i++) if (lon[i] < 0.0f) lon[i] = lon[i] + 360.0f;
        h_llll2gd(x, y, lat, lon, n, lat0, lon0, dlat, dlon, 0.0f);
        return 0;
    case 'L':
        h_cigaxg('L', &lat0, &lon0, &dlat, &dlon, g->ig[0], g->ig[1], g->ig[2], g->ig[3]);
        for (int i = 0; i < n; i++) {
            if (lon[i] < lon0) lon[i] = lon[i] + 360.0f;
            if (lon[i] > (lon0 + (float)ni * dlon)) lon[i] = lon[i] - 360.0f;
        }
        h_llll2gd(x, y, lat, lon, n, lat0, lon0, dlat, dlon, 0.0f);
        return 0;
    case 'N': case 'S':                                     /* ez_ll2rgd.inc:43-55 */
        h_vxyfll(x, y, lat, lon, n, g->xg[2], g->xg[3], g->xg[0], g->xg[1], g->grtyp == 'N' ? 1 : 2);
        return 0;
    case '!':                                               /* ez_ll2rgd.inc:156-159 */
        h_lambfll(g, x, y, lat, lon, n);
        return 0;
    case 'E': {
        float *la = (float *)malloc(sizeof(float) * (size_t)n), *lo = (float *)malloc(sizeof(float) * (size_t)n);
        h_rotate(lo, la, lon, lat, n, g->xg, 1);
        dlon = 360.0f / (float)(ni - 1); lon0 = 0.0f; dlat = 180.0f / (float)nj; lat0 = -90.f + 0.5f * dlat;
        h_llll2gd(x, y, la, lo, n, lat0, lon0, dlat, dlon, 0.0f);
        free(la); free(lo);
        return 0;
    }
    case 'G': case 'Z':
        if (g->grref == 'L') {
            h_cigaxg('L', &lat0, &lon0, &dlat, &dlon, g->igref[0], g->igref[1], g->igref[2], g->igref[3]);
            h_llll2gd(x, y, lat, lon, n, lat0, lon0, dlat, dlon, (g->ax[0] < 0.0f) ? -180.0f : 0.0f);
            for (int i = 0; i < n; i++) { x[i] = x[i] - 1.0f; y[i] = y[i] - 1.0f; }
        } else {
            h_rotate(x, y, lon, lat, n, g->xgref, 1);
        }
        {
            /* entries of ay the search runs over: gr.nj for the x, y of a set (ez_calcxy.c:49), gr.j2 for c_gdxyfll (gdxyfll.c:143); they
             * differ on a hemispheric 'G' grid.  A northern hemisphere then has that count subtracted (ez_calcxy.c:90-96, gdxyfll.c:149-153):
             * its points were sought among the SOUTHERN latitudes of the table and lie beyond the end of that half, extrapolated */
            const int njs = t_locate_j2 ? g->j2 : nj;
            for (int i = 0; i < n; i++) {
                int ix = h_cherche(x[i], g->ax, ni), iy = h_cherche(y[i], g->ay, njs);
                if (ix >= ni) ix = ni - 1;
                if (iy >= njs) iy = njs - 1;
                x[i] = (float)ix + (x[i] - g->ax[ix - 1]) / (g->ax[ix] - g->ax[ix - 1]);
                y[i] = (float)iy + (y[i] - g->ay[iy - 1]) / (g->ay[iy] - g->ay[iy - 1]);
            }
            if (g->grtyp == 'G' && g->ig[0] == 1) for (int i = 0; i < n; i++) y[i] = y[i] - (float)njs;
        }
        return 0;
    }
    return -1;
}

/* h_locate over many points: slices on host threads (the point loops are independent; first-call work such as the
 * Yin-Yang point lists locates 3 x 25.9 M points) */
typedef struct { const ezh_grid *g; float *x, *y; const float *lat; float *lon; int n, rc; const float *lat1d, *lon1d; int ni_t; size_t first; } locate_job;
static void *locate_worker(void *p)
{
    locate_job *j = (locate_job *)p;
    if (j->lat1d) {      /* a separable target: the slice's latitudes and longitudes are written here, by the thread that reads them (first touch of fresh pages in parallel) */
        float *la = (float *)j->lat;
        for (int k = 0; k < j->n; k++) { const size_t q = j->first + (size_t)k; la[k] = j->lat1d[q / (size_t)j->ni_t]; j->lon[k] = j->lon1d[q % (size_t)j->ni_t]; }
    }
    j->rc = h_locate(j->g, j->x, j->y, j->lat, j->lon, j->n);
    return NULL;
}
static int h_locate_mt2(const ezh_grid *g, float *x, float *y, const float *lat, float *lon, int n, const float *lat1d, const float *lon1d, int ni_t);
static int h_locate_mt(const ezh_grid *g, float *x, float *y, const float *lat, float *lon, int n) { return h_locate_mt2(g, x, y, lat, lon, n, NULL, NULL, 0); }
/* lat1d / lon1d != NULL: lat[] and lon[] are OUTPUT too -- the points are the rows of a separable ni_t-wide target, filled slice by slice by the workers */
static int h_locate_mt2(const ezh_grid *g, float *x, float *y, const float *lat, float *lon, int n, const float *lat1d, const float *lon1d, int ni_t)
{
    long ncpu = sysconf(_SC_NPROCESSORS_ONLN);
    /* up to 128 threads, at least 32 768 points each (round 5: 32 threads took 61 ms for cfg3's 8 M points on a 256-thread host -- the set's first call) */
    int nt = (int)(ncpu > 128 ? 128 : ncpu);
    if (getenv("EZHIP_LOCATE_THREADS")) { const int v = atoi(getenv("EZHIP_LOCATE_THREADS")); if (v >= 1 && v <= 128) nt = v; }
    if (nt > n / 32768) nt = n / 32768;
    if (n < 400000 || nt < 2) { locate_job one = {g, x, y, lat, lon, n, 0, lat1d, lon1d, ni_t, 0}; locate_worker(&one); return one.rc; }
    pthread_t th[128]; locate_job job[128];
    int per = (n + nt - 1) / nt, used = 0, rc = 0;
    for (int t = 0; t < nt; t++) {
        int o = t * per, c = n - o < per ? n - o : per;
        if (c <= 0) break;
        job[t] = (locate_job){g, x + o, y + o, lat + o, lon + o, c, 0, lat1d, lon1d, ni_t, (size_t)o};
        if (pthread_create(&th[t], NULL, locate_worker, &job[t])) { locate_worker(&job[t]); th[t] = 0; }
        used = t + 1;
    }
    for (int t = 0; t < used; t++) { if (th[t]) pthread_join(th[t], NULL); if (job[t].rc) rc = job[t].rc; }
    return rc;
}

/* ------------------------------------------------------------------------------------------ */
/* grid table                                                                                   */
/* ------------------------------------------------------------------------------------------ */
int32_t c_gdxyfll_orig(int32_t gd, float *x, float *y, float *lat, float *lon, int32_t n);
static int grid_ok(int gd) { return gd >= 0 && gd < __atomic_load_n(&nG, __ATOMIC_ACQUIRE) && __atomic_load_n(&G[gd].used, __ATOMIC_ACQUIRE); }

static int type_supported(char t, char ref, int ig1, int ig2)
{
    if (t == 'L' || t == 'E' || t == 'N' || t == 'S' || t == '!') return 1;     /* '!': scalars and coordinates (its gribtable entry is looked up when the grid is defined) */
    if (t == 'A' || t == 'B') return ig1 >= 0 && ig1 <= 2 && ig2 >= 0 && ig2 <= 1;   /* hemispheric / y-inverted: scalars only */
    if (t == 'G') return ig1 >= 0 && ig1 <= 2 && ig2 >= 0 && ig2 <= 1;   /* hemispheres: scalars; the northern one reproduces the reference's locate (ez_calcxy.c:90-96) literally */
    if (t == 'Z') return ref == 'L' || ref == 'E';
    if (t == 'Y') return ref == 'L';                                      /* as a target; c_ezdefset refuses it as a source */
    return 0;
}

static int find_grid(int ni, int nj, char t, char ref, const int *ig, const float *ax, const float *ay, char shown)
{
    for (int k = 0; k < nG; k++) {
        ezh_grid *g = &G[k];
        if (!g->used || g->ni != ni || g->nj != nj || g->grtyp != t || g->shown != shown) continue;
        if (memcmp(g->ig, ig, sizeof(int) * 4)) continue;
        if (t == 'Z') {
            if (g->grref != ref) continue;
            if (memcmp(g->ax, ax, sizeof(float) * ni) || memcmp(g->ay, ay, sizeof(float) * nj)) continue;
        }
        if (t == 'Y') {
            if (g->grref != ref) continue;
            if (memcmp(g->ax, ax, sizeof(float) * (size_t)ni * nj) || memcmp(g->ay, ay, sizeof(float) * (size_t)ni * nj)) continue;
        }
        return k;
    }
    return -1;
}

int32_t c_ezgdef_fmem(int32_t ni, int32_t nj, char *grtyp, char *grref, int32_t ig1, int32_t ig2, int32_t ig3, int32_t ig4, float *ax, float *ay)
{
    char t = grtyp[0], ref = grref ? grref[0] : ' ', shown = 0;
    if (t == '#') {
        /* a tile defined from memory carries its own ni / nj axes: every routine of the reference groups it with 'Z' (ez_defaxes.c:44-55,
         * ez_calcxy.c:78-80, ez_defzones.c:62-63 ...) except the polar pieces (pole position, weighted polar wind, strip kernels), which
         * a regional grid never reaches.  Regional tiles only; grids read from a file are out of scope */
        if ((ref != 'L' && ref != 'E') || !ax || !ay || ni < 2 || !(358.0 > (ax[ni - 1] - ax[0]))) {
            fprintf(stderr, "<c_ezgdef_fmem> '#' grids: regional tiles on an 'L' or 'E' reference only (a global '#' grid takes other polar kernels than 'Z' in the reference)\n");
            return -1;
        }
        t = 'Z'; shown = '#';
    }
    if (!type_supported(t, ref, ig1, ig2)) {
        fprintf(stderr, "<c_ezgdef_fmem> grid type '%c' (ref '%c') is outside the MI355X hot-path scope\n", t, ref);
        return -1;
    }
    int ig[4] = {ig1, ig2, ig3, ig4};
    pthread_mutex_lock(&g_mtx);
    int gd = find_grid(ni, nj, t, ref, ig, ax, ay, shown);
    if (gd >= 0) { G[gd].refcount++; pthread_mutex_unlock(&g_mtx); return gd; }
    gd = grid_slot();
    if (gd < 0) { pthread_mutex_unlock(&g_mtx); return -1; }
    ezh_grid *g = &G[gd];
    memset(g, 0, sizeof(*g));
    g->used = 1; g->refcount = 1; g->grtyp = t; g->shown = shown; g->grref = (t == 'Z' || t == 'Y') ? ref : 0;
    g->ni = ni; g->nj = nj; memcpy(g->ig, ig, sizeof(ig));
    switch (t) {                                            /* c_ezdefxg, ez_defxg.c:28-170 */
    case 'A': case 'G':                                     /* ez_defxg.c:39-62 */
        g->xg[3] = (float)(360. / ni); g->xg[1] = 0.0f;
        if (ig1 == 0) { g->xg[2] = (float)(180. / nj); g->xg[0] = (float)(-90. + 0.5 * (double)g->xg[2]); }
        else if (ig1 == 1) { g->xg[2] = (float)(90. / nj); g->xg[0] = (float)(0.5 * (double)g->xg[2]); }
        else { g->xg[2] = (float)(90. / nj); g->xg[0] = (float)(-90. + 0.5 * (double)g->xg[2]); }
        break;
    case 'B':                                               /* ez_defxg.c:77-103 */
        g->xg[3] = (float)(360. / (ni - 1)); g->xg[1] = 0.0f;
        if (ig1 == 0) { g->xg[2] = (float)(180. / (nj - 1)); g->xg[0] = -90.f; }
        else if (ig1 == 1) { g->xg[2] = (float)(90. / (nj - 1)); g->xg[0] = 0.f; }
        else { g->xg[2] = (float)(90. / (nj - 1)); g->xg[0] = -90.f; }
        break;
    case 'L': case 'E':
    case 'N': case 'S':                                     /* ez_defxg.c:150-160: xg = (pi, pj, d60, dgrw) */
        h_cigaxg(t, &g->xg[0], &g->xg[1], &g->xg[2], &g->xg[3], ig1, ig2, ig3, ig4);
        break;
    case '!':                                               /* ez_defxg.c:130-132 keeps nothing; the dictionary entry is read here so that a grid without one fails at its definition */
        if (lamb_lookup(g->lamb, ig1)) { g->used = 0; pthread_mutex_unlock(&g_mtx); return -1; }
        break;
    }
    if (t == 'Z') {                                         /* c_ezdefaxes, ez_defaxes.c:43-54 */
        memcpy(g->igref, ig, sizeof(ig));
        h_cigaxg(ref, &g->xgref[0], &g->xgref[1], &g->xgref[2], &g->xgref[3], ig1, ig2, ig3, ig4);
        g->ax = (float *)malloc(sizeof(float) * ni); memcpy(g->ax, ax, sizeof(float) * ni);
        g->ay = (float *)malloc(sizeof(float) * nj); memcpy(g->ay, ay, sizeof(float) * nj);
    } else if (t == 'Y') {                                  /* ez_defaxes.c:58-65: ax = lon, ay = lat of every point */
        size_t n = (size_t)ni * nj;
        g->ax = (float *)malloc(sizeof(float) * n); memcpy(g->ax, ax, sizeof(float) * n);
        g->ay = (float *)malloc(sizeof(float) * n); memcpy(g->ay, ay, sizeof(float) * n);
    } else if (t == 'G') {                                  /* ez_defaxes.c:65-91 */
        g->grref = 'L';
        g->xgref[0] = 0.0f; g->xgref[1] = 0.0f; g->xgref[2] = 1.0f; g->xgref[3] = 1.0f;
        h_cxgaig('L', &g->igref[0], &g->igref[1], &g->igref[2], &g->igref[3], 0.0f, 0.0f, 1.0f, 1.0f);
        g->ax = (float *)malloc(sizeof(float) * ni);
        float dlon = (float)(360. / (float)ni);
        for (int i = 0; i < ni; i++) g->ax[i] = (float)i * dlon;
        g->ay = (float *)malloc(sizeof(float) * (ig1 ? 2 * nj : nj));
        h_gauss_lat(g->ay, ig1 ? 2 * nj : nj);
    }
    h_xpncof(g);
    if (t == 'Z' || t == 'G') h_nwtncof(g);
    pthread_mutex_unlock(&g_mtx);
    if (O.verbose > 0) printf("Gdid = %02d  grtyp '%c' ni %d nj %d ig %d %d %d %d\n", gd, t, ni, nj, ig1, ig2, ig3, ig4);
    return gd;
}

int32_t c_ezqkdef(int32_t ni, int32_t nj, char *grtyp, int32_t ig1, int32_t ig2, int32_t ig3, int32_t ig4, int32_t iunit)
{
    (void)iunit;
    char t = grtyp[0];
    if (t == '#' || t == 'Y' || t == 'Z' || t == 'U' || t == ' ') {
        fprintf(stderr, "<c_ezqkdef> grid descriptors read from an FST file are outside the MI355X hot-path scope; use c_ezgdef_fmem\n");
        return -1;
    }
    char ref[2] = " ";
    return c_ezgdef_fmem(ni, nj, grtyp, ref, ig1, ig2, ig3, ig4, NULL, NULL);
}

static void free_sepplan(ezh_sepplan *sp)
{
    for (int k = 0; k < 40; k++) { ezhip_free(sp->dev[k]); sp->dev[k] = NULL; }
    free(sp->h_vb_by); free(sp->h_vb_send); free(sp->h_rflag);
    sp->h_vb_by = sp->h_vb_send = NULL; sp->h_rflag = NULL; sp->h_nvb = 0;
    sp->built = 0;
}
static pthread_mutex_t g_plan_mtx = PTHREAD_MUTEX_INITIALIZER;
/* The staged-tile kernels (k_st, k_st1, k_uvt) keep, per grid set, a tile table and a tile-ordered copy of the set's streams: 8 bytes per target point and
 * degree for scalars, 12 for wind pairs (a 7200 x 3601 target: 207 MB + 311 MB), held until c_gdrls.  All sets together stay under a byte budget; a set whose
 * caches would not fit keeps the gathering kernels (same results, slower).  Default 4 GiB; EZHIP_CACHE_MB or ezhip_set_cache_budget_mb() change it (0: no caches). */
static size_t g_cache_bytes = 0;
static long long g_cache_budget = -1;
static size_t cache_budget(void)
{
    if (g_cache_budget < 0) { const char *e = getenv("EZHIP_CACHE_MB"); g_cache_budget = (e ? atoll(e) : 4096LL) << 20; if (g_cache_budget < 0) g_cache_budget = 0; }
    return (size_t)g_cache_budget;
}
/* under g_plan_mtx: reserve `bytes` for set s (0 when the budget is spent) / give them back */
static int cache_reserve(ezh_set *s, size_t bytes)
{
    if (g_cache_bytes + bytes > cache_budget()) return 0;
    g_cache_bytes += bytes; s->cache_bytes += bytes;
    return 1;
}
static void cache_unreserve(ezh_set *s, size_t bytes)
{
    if (bytes > s->cache_bytes) bytes = s->cache_bytes;
    s->cache_bytes -= bytes; g_cache_bytes -= bytes <= g_cache_bytes ? bytes : g_cache_bytes;
}
void ezhip_set_cache_budget_mb(int32_t mb)
{
    pthread_mutex_lock(&g_plan_mtx);
    g_cache_budget = mb > 0 ? (long long)mb << 20 : 0;
    pthread_mutex_unlock(&g_plan_mtx);
}
long long ezhip_cache_bytes(void) { pthread_mutex_lock(&g_plan_mtx); const long long b = (long long)g_cache_bytes; pthread_mutex_unlock(&g_plan_mtx); return b; }

static void free_set(ezh_set *s)
{
    for (int d = 0; d < 3; d++) for (int v = 0; v < 2; v++) free_sepplan(&s->sep[d][v]);
    free(s->x1d); free(s->y1d);
    ezhip_free(s->d_x); ezhip_free(s->d_y); ezhip_free(s->d_scratch); ezhip_free(s->d_prow); ezhip_free(s->d_poles_batch); ezhip_free(s->d_windM); ezhip_free(s->d_lamb_cs[0]); ezhip_free(s->d_lamb_cs[1]); ezhip_free(s->d_cspec_list); ezhip_free(s->d_cspec_xy); ezhip_free(s->d_sspec_list); ezhip_free(s->d_sspec_xy); pthread_mutex_lock(&g_plan_mtx); cache_unreserve(s, s->cache_bytes); pthread_mutex_unlock(&g_plan_mtx); ezhip_free(s->d_uvt_tiles); ezhip_free(s->d_uvt_streams); ezhip_free(s->d_st_tiles[0]); ezhip_free(s->d_st_streams[0]); ezhip_free(s->d_st_tiles[1]); ezhip_free(s->d_st_streams[1]); ezhip_free(s->d_avg[0]); ezhip_free(s->d_avg[1]);
    for (int k = 0; k < 2; k++) { ezhip_free(s->d_yy_x[k]); ezhip_free(s->d_yy_y[k]); ezhip_free(s->d_yy_lat[k]); ezhip_free(s->d_yy_lon[k]); ezhip_free(s->d_yy_idx[k]); }
    for (int k = 0; k < 4; k++) ezhip_free(s->d_yy_tmp[k]);
    free(s);
}

int32_t c_gdrls(int32_t gd)          /* gdrls.c:34-77: refcount, free at zero */
{
    pthread_mutex_lock(&g_mtx);
    if (!grid_ok(gd)) { pthread_mutex_unlock(&g_mtx); return -1; }
    ezh_grid *g = &G[gd];
    if (g->refcount > 0) g->refcount--;
    if (g->refcount == 0) {
        for (ezh_set *s = g->sets; s;) { ezh_set *n = s->next; free_set(s); s = n; }
        for (int k = 0; k < nG; k++) {                     /* sets of other targets that use this grid as source */
            if (!G[k].used || k == gd) continue;
            ezh_set **pp = &G[k].sets;
            while (*pp) { if ((*pp)->gdin == gd) { ezh_set *d = *pp; *pp = d->next; free_set(d); } else pp = &(*pp)->next; }
        }
        free(g->ax); free(g->ay); free(g->ncx); free(g->ncy);
        free(g->lat1d); free(g->lon1d); free(g->lat2d); free(g->lon2d); free(g->mask);
        ezhip_free(g->d_ax); ezhip_free(g->d_ay); ezhip_free(g->d_ncx); ezhip_free(g->d_ncy); ezhip_free(g->d_ncx8); ezhip_free(g->d_ncy8); ezhip_free(g->d_xrec10); ezhip_free(g->d_yrec10); ezhip_free(g->d_xrec8); ezhip_free(g->d_yrec8);
        ezhip_free(g->d_lat); ezhip_free(g->d_lon); ezhip_free(g->d_plon2); g->d_plon2 = NULL; ezhip_free(g->d_windtrig); g->d_windtrig = NULL;
        memset(g, 0, sizeof(*g));
        if (cur_gdin == gd) cur_gdin = -1;
        if (cur_gdout == gd) cur_gdout = -1;
    }
    pthread_mutex_unlock(&g_mtx);
    return 0;
}

static ezh_set *find_set(int gdout, int gdin, int create)
{
    ezh_grid *go = &G[gdout];
    for (ezh_set *s = go->sets; s; s = s->next) if (s->gdin == gdin) return s;
    if (!create) return NULL;
    ezh_set *s = (ezh_set *)calloc(1, sizeof(ezh_set));
    s->gdin = gdin; s->gdout = gdout;
    s->next = go->sets; go->sets = s;
    return s;
}

int32_t c_ezdefset(int32_t gdout, int32_t gdin)     /* ezdefset.c:38-171 */
{
    if (!grid_ok(gdout) || !grid_ok(gdin)) { fprintf(stderr, "<c_ezdefset> invalid grid id\n"); return -1; }
    if (G[gdin].grtyp == 'Y') {       /* the reference interpolates FROM a cloud with its cloud_interp_alg (gdinterp.c:207-230): not on this path */
        fprintf(stderr, "<c_ezdefset> a 'Y' cloud as SOURCE grid is outside the MI355X hot-path scope\n");
        return -1;
    }
    pthread_mutex_lock(&g_mtx);
    find_set(gdout, gdin, 1);
    pthread_mutex_unlock(&g_mtx);
    cur_gdin = gdin; cur_gdout = gdout;
    return 1;
}
int32_t c_ezgetgdin(void) { return cur_gdin; }
int32_t c_ezgetgdout(void) { return cur_gdout; }

int32_t c_ezgprm(int32_t gd, char *grtyp, int32_t *ni, int32_t *nj, int32_t *ig1, int32_t *ig2, int32_t *ig3, int32_t *ig4)
{
    if (!grid_ok(gd)) return -1;
    ezh_grid *g = &G[gd];
    grtyp[0] = g->shown ? g->shown : g->grtyp; *ni = g->ni; *nj = g->nj; *ig1 = g->ig[0]; *ig2 = g->ig[1]; *ig3 = g->ig[2]; *ig4 = g->ig[3];
    return 0;
}
int32_t c_ezgxprm(int32_t gd, int32_t *ni, int32_t *nj, char *grtyp, int32_t *ig1, int32_t *ig2, int32_t *ig3, int32_t *ig4,
                  char *grref, int32_t *ig1ref, int32_t *ig2ref, int32_t *ig3ref, int32_t *ig4ref)
{
    if (!grid_ok(gd)) return -1;
    ezh_grid *g = &G[gd];
    grtyp[0] = g->shown ? g->shown : g->grtyp; grtyp[1] = 0; *ni = g->ni; *nj = g->nj;
    *ig1 = g->ig[0]; *ig2 = g->ig[1]; *ig3 = g->ig[2]; *ig4 = g->ig[3];
    if (g->grtyp == 'Z' || g->grtyp == 'G') {
        grref[0] = g->grref; grref[1] = 0;
        *ig1ref = g->igref[0]; *ig2ref = g->igref[1]; *ig3ref = g->igref[2]; *ig4ref = g->igref[3];
    } else { grref[0] = ' '; grref[1] = 0; *ig1ref = *ig2ref = *ig3ref = *ig4ref = -1; }
    return 0;
}
int32_t c_gdgaxes(int32_t gd, float *ax, float *ay)
{
    if (!grid_ok(gd) || !G[gd].ax) return -1;
    size_t nx = G[gd].ni, ny = G[gd].nj;
    if (G[gd].grtyp == 'Y') nx = ny = (size_t)G[gd].ni * G[gd].nj;        /* gdgaxes.c:40-45 */
    memcpy(ax, G[gd].ax, sizeof(float) * nx); memcpy(ay, G[gd].ay, sizeof(float) * ny);
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* options: ezsetopt.c:59-215, ezgetopt.c, ezsetval.c:86-166                                    */
/* ------------------------------------------------------------------------------------------ */
static void lower32(char *dst, const char *src)
{
    memset(dst, 0, 32);
    strncpy(dst, src, 31);
    for (char *p = dst; *p; p++) *p = (char)tolower((int)*p);
}
static int deg_of(const char *v)
{
    if (!strcmp(v, "nearest")) return 0;
    if (!strcmp(v, "linear")) return 1;
    if (!strcmp(v, "cubic")) return 3;
    return -1;
}
int32_t c_ezsetopt(char *option, char *value)
{
    char o[32], v[32];
    lower32(o, option); lower32(v, value);
    static const char *oa[][2] = {{"correction_polaire", "polar_correction"}, {"degre_interp", "interp_degree"},
                                  {"degre_extrap", "extrap_degree"}, {"use_1sousgrille", "use_1subgrid"}};
    static const char *va[][2] = {{"oui", "yes"}, {"ouiouioui", "yesyesyes"}, {"non", "no"}, {"voisin", "nearest"}, {"lineair", "linear"},
                                  {"lineaire", "linear"}, {"cubique", "cubic"}, {"neutre", "neutral"}, {"valeur", "value"}};
    for (unsigned k = 0; k < sizeof(oa) / sizeof(oa[0]); k++) if (!strcmp(o, oa[k][0])) strcpy(o, oa[k][1]);
    for (unsigned k = 0; k < sizeof(va) / sizeof(va[0]); k++) if (!strcmp(v, va[k][0])) strcpy(v, va[k][1]);
    int ook = 0, vok = 0;
    if (!strcmp(o, "use_1subgrid")) { ook = 1; vok = 1; if (!strcmp(v, "yes")) O.use_1subgrid = 1; else if (!strcmp(v, "no")) O.use_1subgrid = 0; else vok = 0; }
    if (!strcmp(o, "verbose")) { ook = 1; vok = 1; if (!strcmp(v, "yes")) O.verbose = 1; else if (!strcmp(v, "yesyesyes")) O.verbose = 2; else if (!strcmp(v, "no")) O.verbose = 0; else vok = 0; }
    if (!strcmp(o, "polar_correction")) { ook = 1; vok = 1; if (!strcmp(v, "yes")) O.polar_correction = 1; else if (!strcmp(v, "no")) O.polar_correction = 0; else vok = 0; }
    if (!strcmp(o, "interp_degree")) {
        ook = 1; vok = 1;
        int d = deg_of(v);
        if (d >= 0) O.degre_interp = d;
        else if (!strcmp(v, "average")) O.degre_interp = 4;
        else if (!strcmp(v, "sph_average")) O.degre_interp = 5;
        else vok = 0;
    }
    if (!strcmp(o, "extrap_degree")) {
        ook = 1; vok = 1;
        int d = deg_of(v);
        if (!strcmp(v, "neutral")) O.degre_extrap = O.degre_interp;
        else if (d >= 0) O.degre_extrap = d;
        else if (!strcmp(v, "maximum")) O.degre_extrap = XT_MAXIMUM;
        else if (!strcmp(v, "minimum")) O.degre_extrap = XT_MINIMUM;
        else if (!strcmp(v, "value")) O.degre_extrap = XT_VALUE;
        else if (!strcmp(v, "abort")) O.degre_extrap = XT_ABORT;
        else vok = 0;
    }
    if (!strcmp(o, "cloud_interp_alg")) {
        if (!strcmp(v, "linear")) { ook = vok = 1; O.cld_interp_alg = 1; }
        if (!strcmp(v, "distance")) { ook = vok = 1; O.cld_interp_alg = 4; }
    }
    if (ook + vok != 2) {
        if (!ook) fprintf(stderr, "ezsetopt : option not recognized : %s\n", option);
        if (!vok) fprintf(stderr, "ezsetopt : value not recognized : %s\n", value);
        return -1;
    }
    return 0;
}
int32_t c_ezgetopt(char *option, char *value)
{
    char o[32];
    lower32(o, option);
    static const char *dn[] = {"nearest", "linear", "?", "cubic"};
    if (!strcmp(o, "verbose")) strcpy(value, O.verbose == 1 ? "yes" : O.verbose == 2 ? "yesyesyes" : "no");
    if (!strcmp(o, "use_1subgrid")) strcpy(value, O.use_1subgrid == 1 ? "yes" : "no");
    if (!strcmp(o, "polar_correction")) strcpy(value, O.polar_correction == 1 ? "yes" : "no");
    if (!strcmp(o, "interp_degree")) strcpy(value, (O.degre_interp == 0 || O.degre_interp == 1 || O.degre_interp == 3) ? dn[O.degre_interp] : "error");
    if (!strcmp(o, "degre_interp")) strcpy(value, O.degre_interp == 0 ? "voisin" : O.degre_interp == 1 ? "lineaire" : O.degre_interp == 3 ? "cubique" : "erreur");
    if (!strcmp(o, "extrap_degree")) {
        switch (O.degre_extrap) {
        case 0: case 1: case 3: strcpy(value, dn[O.degre_extrap]); break;
        case XT_MAXIMUM: strcpy(value, "maximum"); break;
        case XT_MINIMUM: strcpy(value, "minimum"); break;
        case XT_VALUE: strcpy(value, "value"); break;
        case XT_ABORT: strcpy(value, "abort"); break;
        default: strcpy(value, "error");
        }
    }
    if (!strcmp(o, "cloud_interp_alg")) strcpy(value, O.cld_interp_alg == 1 ? "linear" : "distance");
    return 0;
}
int32_t c_ezsetval(char *option, float fvalue)
{
    char o[32];
    lower32(o, option);
    if (!strcmp(o, "extrap_value")) O.valeur_extrap = fvalue;
    if (!strcmp(o, "missing_gridpt_distance")) O.msg_gridpt_dist = fvalue;
    if (!strcmp(o, "missing_distance_threshold")) O.msg_dist_thresh = fvalue;
    return 0;
}
int32_t c_ezsetival(char *option, int32_t ivalue)
{
    char o[32];
    lower32(o, option);
    if (!strcmp(o, "weight_number")) O.wgt_num = ivalue;
    if (!strcmp(o, "missing_points_tolerance")) O.msg_pt_tol = ivalue;
    if (!strcmp(o, "subgridid")) O.valeur_1subgrid = ivalue;
    return 0;
}
int32_t c_ezgetval(char *option, float *fvalue)
{
    char o[32];
    lower32(o, option);
    if (!strcmp(o, "extrap_value")) *fvalue = O.valeur_extrap;
    if (!strcmp(o, "missing_gridpt_distance")) *fvalue = O.msg_gridpt_dist;
    if (!strcmp(o, "missing_distance_threshold")) *fvalue = O.msg_dist_thresh;
    return 0;
}
int32_t c_ezgetival(char *option, int32_t *ivalue)
{
    char o[32];
    lower32(o, option);
    if (!strcmp(o, "weight_number")) *ivalue = O.wgt_num;
    if (!strcmp(o, "missing_points_tolerance")) *ivalue = O.msg_pt_tol;
    if (!strcmp(o, "subgridid")) *ivalue = O.valeur_1subgrid;
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* target coordinates: ez_calclatlon.c:30-284 (GRLL grll.f:57-64, c_llfgr ez_llfgr.c:25-33)       */
/* ------------------------------------------------------------------------------------------ */
static int grid_yinv(const ezh_grid *g);
static void ensure_coords(ezh_grid *g)
{
    if (g->coords_ready) return;
    int ni = g->ni, nj = g->nj;
    g->separable = !(g->grtyp == 'E' || (g->grtyp == 'Z' && g->grref == 'E') || g->grtyp == 'N' || g->grtyp == 'S' || g->grtyp == 'Y' || g->grtyp == '!');
    g->lat1d = (float *)calloc((size_t)nj, sizeof(float));
    g->lon1d = (float *)calloc((size_t)ni, sizeof(float));
    if (g->grtyp == 'Y') {                                  /* ez_calclatlon.c:127-143 */
        size_t n = (size_t)ni * nj;
        g->lat2d = (float *)malloc(sizeof(float) * n); g->lon2d = (float *)malloc(sizeof(float) * n);
        memcpy(g->lon2d, g->ax, sizeof(float) * n); memcpy(g->lat2d, g->ay, sizeof(float) * n);
        g->coords_ready = 1;
        return;
    }
    if (g->grtyp == '!') {                                  /* ez_calclatlon.c:229-250 */
        size_t n = (size_t)ni * nj;
        g->lat2d = (float *)malloc(sizeof(float) * n); g->lon2d = (float *)malloc(sizeof(float) * n);
        float *x = (float *)malloc(sizeof(float) * n), *y = (float *)malloc(sizeof(float) * n);
        for (int j = 0; j < nj; j++) for (int i = 0; i < ni; i++) { x[(size_t)j * ni + i] = (float)(i + 1.0); y[(size_t)j * ni + i] = (float)(j + 1.0); }
        h_llflamb(g, g->lat2d, g->lon2d, x, y, (int)n);
        for (size_t k = 0; k < n; k++) if (g->lon2d[k] < 0.0f) g->lon2d[k] = g->lon2d[k] + 360.0f;
        free(x); free(y);
        /* igaxg95 hands back ig3 = 256 * sub-centre + projection code and ig4 = 0 THROUGH its arguments, and ez_llflamb passes the grid's own
         * descriptors (:245-247): from its first coordinates on, the grid's ig3 / ig4 are these (c_ezgprm shows them) */
        g->ig[2] = 256 * (int)lroundf(g->lamb[12]) + (int)lroundf(g->lamb[13]); g->ig[3] = 0;
        g->coords_ready = 1;
        return;
    }
    if (g->grtyp == 'N' || g->grtyp == 'S') {              /* ez_calclatlon.c:83-97 */
        size_t n = (size_t)ni * nj;
        g->lat2d = (float *)malloc(sizeof(float) * n); g->lon2d = (float *)malloc(sizeof(float) * n);
        h_grps(g->lat2d, g->lon2d, ni, nj, g->xg[0], g->xg[1], g->xg[2], g->xg[3], g->grtyp == 'N' ? 1 : 2);
        g->coords_ready = 1;
        return;
    }
    if (g->grtyp == 'L' || g->grtyp == 'A' || g->grtyp == 'B') {
        for (int j = 0; j < nj; j++) g->lat1d[j] = g->xg[0] + (float)j * g->xg[2];
        for (int i = 0; i < ni; i++) g->lon1d[i] = fmodf(g->xg[1] + (float)i * g->xg[3], 360.0f);
    } else if (g->grtyp == 'E') {
        float dlon = (float)(360. / (ni - 1)), dlat = (float)(180. / nj);
        float lat00 = (float)(-90. + 0.5 * (double)dlat);
        for (int j = 0; j < nj; j++) g->lat1d[j] = lat00 + (float)j * dlat;
        for (int i = 0; i < ni; i++) g->lon1d[i] = fmodf(0.0f + (float)i * dlon, 360.0f);
    } else {   /* G, Z */
        for (int j = 0; j < nj; j++) g->lat1d[j] = g->ay[(g->grtyp == 'G' && g->ig[0] == 1) ? j + nj : j];      /* ez_calclatlon.c:167-176 */
        for (int i = 0; i < ni; i++) g->lon1d[i] = g->ax[i];
        if (g->grref == 'L') {
            for (int j = 0; j < nj; j++) {
                float y = (float)((double)g->lat1d[j] + 1.0);
                g->lat1d[j] = (float)((double)g->xgref[0] + (double)g->xgref[2] * ((double)y - 1.0));
            }
            for (int i = 0; i < ni; i++) {
                float x = (float)((double)g->lon1d[i] + 1.0);
                float lo = (float)((double)g->xgref[1] + (double)g->xgref[3] * ((double)x - 1.0));
                g->lon1d[i] = (float)fmod(fmod((double)lo, 360.0) + 360.0, 360.0);
            }
        }
    }
    if (grid_yinv(g))                                      /* ez_calclatlon.c:268-279: PERMUT of the latitudes */
        for (int j = 0; j < nj / 2; j++) { float t = g->lat1d[j]; g->lat1d[j] = g->lat1d[nj - 1 - j]; g->lat1d[nj - 1 - j] = t; }
    if (!g->separable) {   /* rotated grids: full 2-D true lat/lon through ez_gfllfxy */
        size_t n = (size_t)ni * nj;
        float *rlat = (float *)malloc(sizeof(float) * n), *rlon = (float *)malloc(sizeof(float) * n);
        for (int j = 0; j < nj; j++) for (int i = 0; i < ni; i++) { rlat[(size_t)j * ni + i] = g->lat1d[j]; rlon[(size_t)j * ni + i] = g->lon1d[i]; }
        g->lat2d = (float *)malloc(sizeof(float) * n); g->lon2d = (float *)malloc(sizeof(float) * n);
        h_rotate(g->lon2d, g->lat2d, rlon, rlat, (int)n, g->grtyp == 'E' ? g->xg : g->xgref, 0);
        free(rlat); free(rlon);
    }
    g->coords_ready = 1;
}

int32_t c_gdll(int32_t gd, float *lat, float *lon)
{
    if (!grid_ok(gd)) return -1;
    ezh_grid *g = &G[gd];
    if (g->nsub > 0) {                                     /* gdll.c:46-70: Yin then Yang */
        size_t nsub = (size_t)G[g->sub[0]].ni * G[g->sub[0]].nj;
        if (c_gdll(g->sub[0], lat, lon)) return -1;
        return c_gdll(g->sub[1], lat + nsub, lon + nsub);
    }
    ensure_coords(g);
    size_t n = (size_t)g->ni * g->nj;
    if (g->separable) {
        for (int j = 0; j < g->nj; j++) for (int i = 0; i < g->ni; i++) { lat[(size_t)j * g->ni + i] = g->lat1d[j]; lon[(size_t)j * g->ni + i] = g->lon1d[i]; }
    } else { memcpy(lat, g->lat2d, sizeof(float) * n); memcpy(lon, g->lon2d, sizeof(float) * n); }
    return 0;
}

/* c_gdxyfll / c_gdxyfll_orig, gdxyfll.c:90-139: regular sources work on a copy of lon, G/Z sources
 * modify the caller's lon in place. */
/* c_gdxyfll_orig (gdxyfll.c:107-159): what the library itself locates with (pole positions, c_gdllsval, c_gdllvval, c_gdxyzfll, Yin-Yang) */
int32_t c_gdxyfll_orig(int32_t gd, float *x, float *y, float *lat, float *lon, int32_t n)
{
    if (!grid_ok(gd)) return -1;
    ezh_grid *g = &G[gd];
    if (g->grtyp == 'G' && g->ig[0] != 0) { t_locate_j2 = 1; int rc = h_locate(g, x, y, lat, lon, n); t_locate_j2 = 0; return rc; }
    if (g->grtyp == 'G' || g->grtyp == 'Z') return h_locate_mt(g, x, y, lat, lon, n);
    float *tmp = (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    memcpy(tmp, lon, sizeof(float) * (size_t)n);
    int rc = h_locate_mt(g, x, y, lat, tmp, n);
    free(tmp);
    return rc;
}
/* the public c_gdxyfll (gdxyfll.c:29-104 for every grid but 'U'): the same, and on a y-inverted Gaussian grid the row coordinate counted
 * from the other end, y := nj_in + 1 - y with nj_in = j2 (:88-94) -- the stored order of such a grid's rows */
int32_t c_gdxyfll(int32_t gd, float *x, float *y, float *lat, float *lon, int32_t n)
{
    int rc = c_gdxyfll_orig(gd, x, y, lat, lon, n);
    if (rc >= 0 && G[gd].grtyp == 'G' && G[gd].ig[1] == 1) {
        const int nj_in = G[gd].j2;
        for (int k = 0; k < n; k++) y[k] = (float)(((double)nj_in + 1.0) - (double)y[k]);
    }
    return rc;
}

/* ------------------------------------------------------------------------------------------ */
/* device mirrors                                                                               */
/* ------------------------------------------------------------------------------------------ */
static void *upload(const void *h, size_t nbytes)
{
    void *d = ezhip_malloc(nbytes);
    if (d && ezhip_h2d(d, h, nbytes)) { ezhip_free(d); return NULL; }
    return d;
}
/* a kernel of an earlier (asynchronous) call gave up -- today only k_sepx's bounded wait for the in-launch pole sums: the outputs
 * of that call carry NaN polar rows.  Sticky until reported: entry points report it on entry, the synchronising host-pointer
 * calls also before they return. */
static int device_error(const char *who)
{
    if (!ezhip_device_error()) return 0;
    fprintf(stderr, "<%s> a device kernel of this or an earlier call gave up waiting for its pole values: polar rows are NaN\n", who);
    return -1;
}
/* a device-side step failed: say why (the shim keeps the HIP error text) instead of a bare -1 */
static int dev_fail(const char *what)
{
    fprintf(stderr, "<ezscint-hip> %s failed: %s\n", what, ezhip_last_error());
    return -1;
}
static int need_device(const char *who)
{
    if (ezhip_runtime_ok()) return ezhip_bound_device_ok(who);
    fprintf(stderr, "<%s> no usable HIP device: the MI355X interpolation path has no CPU fallback\n", who);
    return -1;
}
static int ensure_grid_dev(ezh_grid *g)
{
    if (g->ax && !g->d_ax) {
        g->d_ax = (float *)upload(g->ax, sizeof(float) * g->ni);
        const int nr = ay_rows(g);
        g->d_ay = (float *)upload(g->ay, sizeof(float) * nr);
        g->d_ncx = (float *)upload(g->ncx, sizeof(float) * 6 * g->ni);
        g->d_ncy = (float *)upload(g->ncy, sizeof(float) * 6 * nr);
        {
            float *t8 = (float *)calloc((size_t)8 * (g->ni > nr ? g->ni : nr), sizeof(float));
            for (int i = 0; i < g->ni; i++) for (int k = 0; k < 6; k++) t8[8 * i + k] = g->ncx[k * g->ni + i];
            g->d_ncx8 = (float *)upload(t8, sizeof(float) * 8 * g->ni);
            ezhip_sync();
            for (int j = 0; j < nr; j++) for (int k = 0; k < 6; k++) t8[8 * j + k] = g->ncy[k * nr + j];
            g->d_ncy8 = (float *)upload(t8, sizeof(float) * 8 * nr);
            ezhip_sync();
            free(t8);
        }
        {   /* the 48-byte records k_uvt stages per tile: axis entries i-1 .. i+2 next to the six coefficients of stencil i (interior indices only: the
             * tiles that touch the seam or the first / last rows' clamps never read the others) */
            const int m = g->ni > nr ? g->ni : nr;
            double *t10 = (double *)calloc((size_t)10 * m, sizeof(double));
            if (!t10) return -1;
            for (int i = 1; i + 2 < g->ni; i++) {
                for (int k = 0; k < 3; k++) t10[10 * i + k] = (double)g->ax[i - 1 + k];
                for (int k = 0; k < 6; k++) t10[10 * i + 3 + k] = (double)g->ncx[k * g->ni + i];
                t10[10 * i + 9] = (double)g->ncx[4 * g->ni + i] + (double)g->ncx[1 * g->ni + i];              /* c5 + c2, as the kernels form it */
            }
            g->d_xrec10 = (double *)upload(t10, sizeof(double) * 10 * g->ni);
            ezhip_sync();
            memset(t10, 0, sizeof(double) * 10 * (size_t)m);
            for (int j = 1; j + 2 < nr; j++) {
                for (int k = 0; k < 3; k++) t10[10 * j + k] = (double)g->ay[j - 1 + k];
                for (int k = 0; k < 6; k++) t10[10 * j + 3 + k] = (double)g->ncy[k * nr + j];
                t10[10 * j + 9] = (double)g->ncy[4 * nr + j] + (double)g->ncy[1 * nr + j];
            }
            g->d_yrec10 = (double *)upload(t10, sizeof(double) * 10 * nr);
            ezhip_sync();
            free(t10);
        }
        {   /* the 32-byte records of the wind pair kernels (interior stencils only, as above): the axis entries of stencil i and 1 / prod_{m != k} (x_k - x_m),
             * formed in REAL*8 from the REAL axis and rounded once */
            const int m = g->ni > nr ? g->ni : nr;
            float *t8 = (float *)calloc((size_t)8 * m, sizeof(float));
            if (!t8) return -1;
            for (int pass = 0; pass < 2; pass++) {
                const float *a = pass ? g->ay : g->ax;
                const int n = pass ? nr : g->ni;
                memset(t8, 0, sizeof(float) * 8 * (size_t)m);
                for (int i = 1; i + 2 < n; i++) {
                    double x[4];
                    for (int k = 0; k < 4; k++) { x[k] = (double)a[i - 1 + k]; t8[8 * i + k] = a[i - 1 + k]; }
                    for (int k = 0; k < 4; k++) {
                        double den = 1.0;
                        for (int q = 0; q < 4; q++) if (q != k) den *= x[k] - x[q];
                        t8[8 * i + 4 + k] = (float)(1.0 / den);
                    }
                }
                float *d = (float *)upload(t8, sizeof(float) * 8 * n);
                ezhip_sync();
                if (pass) g->d_yrec8 = d; else g->d_xrec8 = d;
            }
            free(t8);
        }
        ezhip_sync();
        if (!g->d_ax || !g->d_ay || !g->d_ncx || !g->d_ncy || !g->d_ncx8 || !g->d_ncy8 || !g->d_xrec10 || !g->d_yrec10 || !g->d_xrec8 || !g->d_yrec8) return -1;
    }
    return 0;
}
static int ensure_coords_dev(ezh_grid *g)
{
    ensure_coords(g);
    if (g->d_coords_valid) return 0;
    ezhip_free(g->d_lat); ezhip_free(g->d_lon);
    if (g->separable) {
        g->d_lat = (float *)upload(g->lat1d, sizeof(float) * g->nj);
        g->d_lon = (float *)upload(g->lon1d, sizeof(float) * g->ni);
    } else {
        size_t n = (size_t)g->ni * g->nj;
        g->d_lat = (float *)upload(g->lat2d, sizeof(float) * n);
        g->d_lon = (float *)upload(g->lon2d, sizeof(float) * n);
    }
    ezhip_sync();
    g->d_coords_valid = (g->d_lat && g->d_lon);
    return g->d_coords_valid ? 0 : -1;
}

/* ------------------------------------------------------------------------------------------ */
/* set analysis: zones (ez_defzones.c:25-113) and kernel-family choice                           */
/* ------------------------------------------------------------------------------------------ */
static int src_is_separable(const ezh_grid *g)
{   /* x depends on lon only and y on lat only: everything but the rotated frames */
    return !(g->grtyp == 'E' || (g->grtyp == 'Z' && g->grref == 'E') || g->grtyp == 'N' || g->grtyp == 'S' || g->grtyp == '!');
}
static int src_irregular(const ezh_grid *g) { return g->grtyp == 'G' || g->grtyp == 'Z'; }
static int src_hemi(const ezh_grid *g) { return (g->grtyp == 'A' || g->grtyp == 'B' || g->grtyp == 'G') && g->ig[0] != 0; }
static int grid_yinv(const ezh_grid *g) { return (g->grtyp == 'A' || g->grtyp == 'B' || g->grtyp == 'G') && g->ig[1] == 1; }      /* axe_y_inverse, ez_defxg.c:68-72 */

static int set_extrap(const ezh_grid *gi)
{
    if (gi->grtyp == 'N' || gi->grtyp == 'S' || gi->grtyp == '!') return 1;       /* ez_defzones.c:46-50 */
    if (gi->grtyp == 'L') return gi->extension == 0;
    if (gi->grtyp == 'Z' && (gi->grref == 'E' || gi->grref == 'L'))
        return 358.0 > (gi->ax[gi->ni - 1] - gi->ax[0]);
    return 0;
}

static void pole_y(const ezh_grid *gi, float *yn, float *ys)
{   /* ez_defzone_polenord.c:46-53, ez_defzone_polesud.c:46-54 */
    if (gi->grtyp == 'Z' && gi->grref == 'E') { *yn = (float)(gi->nj + 0.5); *ys = 0.5f; return; }
    float x, lat, lon;
    lat = 90.0f; lon = 0.0f; c_gdxyfll_orig((int)(gi - G), &x, yn, &lat, &lon, 1);
    lat = -90.0f; lon = 0.0f; c_gdxyfll_orig((int)(gi - G), &x, ys, &lat, &lon, 1);
}

static void analyse_set(ezh_set *s)
{
    ezh_grid *gi = &G[s->gdin], *go = &G[s->gdout];
    ensure_coords(go);
    s->extrap = set_extrap(gi);
    pole_y(gi, &s->ypole_n, &s->ypole_s);
    s->sep_capable = go->separable && src_is_separable(gi);
    if (s->sep_capable && !s->x1d) {
        /* 1-D locate with the exact host code; the reference's in-place edits of the target's cached
         * longitudes land in lon1d, exactly as they would in its 2-D cache */
        int ni = go->ni, nj = go->nj, n = imax(ni, nj);
        float *lat = (float *)malloc(sizeof(float) * n), *lon = (float *)malloc(sizeof(float) * n);
        float *x = (float *)malloc(sizeof(float) * n), *y = (float *)malloc(sizeof(float) * n);
        for (int k = 0; k < n; k++) { lat[k] = go->lat1d[k < nj ? k : nj - 1]; lon[k] = go->lon1d[k < ni ? k : ni - 1]; }
        h_locate(gi, x, y, lat, lon, n);
        s->x1d = (float *)malloc(sizeof(float) * ni); memcpy(s->x1d, x, sizeof(float) * ni);
        s->y1d = (float *)malloc(sizeof(float) * nj); memcpy(s->y1d, y, sizeof(float) * nj);
        if (memcmp(go->lon1d, lon, sizeof(float) * ni)) { memcpy(go->lon1d, lon, sizeof(float) * ni); go->d_coords_valid = 0; }
        free(lat); free(lon); free(x); free(y);
        s->have_dehors = 0;
        if (s->extrap) {
            for (int i = 0; i < ni && !s->have_dehors; i++) { int ix = (int)((double)s->x1d[i] + 0.5); if (ix < 1 || ix > gi->ni) s->have_dehors = 1; }
            for (int j = 0; j < nj && !s->have_dehors; j++) { int iy = (int)((double)s->y1d[j] + 0.5); if (iy < 1 || iy > gi->nj) s->have_dehors = 1; }
        }
    }
}

/* 1 = separable kernel, 2 = per-point kernel */
static int choose_mode(const ezh_set *s, int degree, int polar)
{
    const ezh_grid *gi = &G[s->gdin];
    if (!s->sep_capable) return 2;
    if (getenv("EZHIP_FORCE_PTS")) return 2;
    if (src_hemi(gi)) return 2;                            /* rows j1..j2 of the expanded field: the per-point kernel indexes them natively */
    if (grid_yinv(&G[s->gdout])) return 2;                 /* target rows run north to south */
    /* ez_irgdint_3_nw rounds every intermediate to REAL (ez_irgdint_3_nw.inc:32): not a linear
     * functional of the stencil, keep it on the point-exact kernel */
    if (src_irregular(gi) && gi->extension == 0 && degree == DEG_CUBIC) return 2;
    /* out-of-grid points re-interpolated with another degree (ez_corrval.c:98-110): point kernel */
    if (polar && s->extrap && s->have_dehors && O.degre_extrap < XT_MAXIMUM) return 2;
    return 1;
}

/* ------------------------------------------------------------------------------------------ */
/* separable plan construction                                                                  */
/* ------------------------------------------------------------------------------------------ */
static double fa_(double a1, double a2, double a3, double a4, double x, double x1, double x2, double x3)
{ return a1 + (x - x1) * (a2 + (x - x2) * (a3 + a4 * (x - x3))); }
static double fa2_(double c1, double a1, double a2) { return c1 * (a2 - a1); }
static double fa3_(double c1, double c2, double c3, double a1, double a2, double a3) { return c2 * (c3 * (a3 - a2) - c1 * (a2 - a1)); }
static double fa4_(double c1, double c2, double c3, double c4, double c5, double c6, double a1, double a2, double a3, double a4)
{ return c4 * (c5 * (c6 * (a4 - a3) - c3 * (a3 - a2)) - c2 * (c3 * (a3 - a2) - c1 * (a2 - a1))); }

/* weights of the Newton-form 1-D cubic (ez_irgdint_3_w.inc:86-89): the functional applied to unit vectors */
static void newton_weights(double w[4], const double c[6], double x, double x1, double x2, double x3)
{
    for (int k = 0; k < 4; k++) {
        double z[4] = {0, 0, 0, 0};
        z[k] = 1.0;
        double a2 = fa2_(c[0], z[0], z[1]);
        double a3 = fa3_(c[0], c[1], c[2], z[0], z[1], z[2]);
        double a4 = fa4_(c[0], c[1], c[2], c[3], c[4], c[5], z[0], z[1], z[2], z[3]);
        w[k] = fa_(z[0], a2, a3, a4, x, x1, x2, x3);
    }
}
/* weights of cubic8.cdk:1-3 (uniform spacing; REAL literals) */
static double cubic8(double z1, double z2, double z3, double z4, double dx)
{
    const double c6 = (double)0.1666666666666f, c3 = (double)0.3333333333333f;
    return ((((z4 - z1) * c6 + 0.5 * (z2 - z3)) * dx + 0.5 * (z1 + z3) - z2) * dx + z3 - c6 * z4 - 0.5 * z2 - c3 * z1) * dx + z2;
}
static void cubic_weights(double w[4], double dx)
{
    for (int k = 0; k < 4; k++) { double z[4] = {0, 0, 0, 0}; z[k] = 1.0; w[k] = cubic8(z[0], z[1], z[2], z[3], dx); }
}

/* seam columns of ez_irgdint_3_w.inc:100-156 / _wnnc: 1-based indices and float abscissae */
static void irr_cols(const float *ax, int ni, int wrap, int i, int ip2_wrap1, int idx[4], float xs[4])
{
#define AX(k) ax[(k) - 1]
    idx[0] = i - 1; idx[1] = i; idx[2] = i + 1; idx[3] = i + 2;
    if (wrap == 1 && (i <= 1 || i >= (ni - wrap))) {
        if (i == 1) { idx[0] = ni - 1; idx[2] = 2; idx[3] = 3; xs[0] = AX(ni - 1) - 360.0f; xs[1] = AX(1); xs[2] = AX(2); xs[3] = AX(3); }
        if (i == (ni - 1)) { idx[0] = ni - 2; idx[2] = ni; idx[3] = ip2_wrap1; xs[0] = AX(ni - 2); xs[1] = AX(ni - 1); xs[2] = AX(ni); xs[3] = AX(2) + 360.0f; }
    } else if (wrap == 2 && (i <= 1 || i > (ni - wrap))) {
        if (i == 1) { idx[0] = ni; idx[2] = 2; idx[3] = 3; xs[0] = AX(ni) - 360.0f; xs[1] = AX(1); xs[2] = AX(2); xs[3] = AX(3); }
        if (i == (ni - 1)) { idx[0] = ni - 2; idx[2] = ni; idx[3] = 1; xs[0] = AX(ni - 2); xs[1] = AX(ni - 1); xs[2] = AX(ni); xs[3] = AX(1) + 360.0f; }
        if (i == ni) { idx[0] = ni - 1; idx[2] = 1; idx[3] = 2; xs[0] = AX(ni - 1); xs[1] = AX(ni); xs[2] = AX(1) + 360.0f; xs[3] = AX(2) + 360.0f; }
    } else {
        xs[0] = AX(idx[0]); xs[1] = AX(i); xs[2] = AX(idx[2]); xs[3] = AX(idx[3]);
    }
#undef AX
}
/* seam columns of ez_rgdint_3_w.inc:72-90 */
static void reg_cols(int ni, int wrap, int *i, int idx[4])
{
    int limite = ni + 2 - wrap;
    int im1 = (limite + *i - 1) % limite, ip1 = (limite + *i + 1) % limite, ip2 = (limite + *i + 2) % limite;
    if (im1 == 0) im1 = ni;
    if (*i == 0) *i = ni;
    if (ip1 == 0) ip1 = ni;
    if (ip2 == 0) ip2 = ni;
    if (wrap == 1) { if (ip2 == ni) ip2 = 2; if (im1 == ni) im1 = ni - 1; }
    idx[0] = im1; idx[1] = *i; idx[2] = ip1; idx[3] = ip2;
}

/* column taps (1-based idx -> stored 0-based) for the MAIN kernel of the dispatch table gdinterp.c:133-309 */
static void col_taps_main(const ezh_grid *g, int degree, float px, int idx[4], double w[4])
{
    int ni = g->ni, wrap = g->extension, irr = src_irregular(g);
    idx[0] = idx[1] = idx[2] = idx[3] = 1; w[0] = w[1] = w[2] = w[3] = 0.0;
    if (degree == DEG_NEAREST) { idx[0] = imin(ni, imax(1, nintf_(px))); return; }           /* ez_rgdint_0.inc */
    if (degree == DEG_LINEAR) {
        if (!irr) {
            if (wrap == 2) {                                                                  /* ez_rgdint_1_w.inc */
                int limite = ni + 2 - wrap, i = imin(ni - 2 + wrap, imax(1, (int)px)), ip1 = i + 1;
                if (i == (ni - 2 + wrap)) ip1 = (limite + i + 1) % limite;
                idx[0] = i; idx[1] = ip1; w[0] = (double)(px - (float)i);
            } else {                                                                          /* ez_rgdint_1_nw.inc */
                int i = imin(ni - 1, imax(1, (int)px));
                idx[0] = i; idx[1] = i + 1; w[0] = (double)(px - (float)i);
            }
        } else if (wrap == 0) {                                                               /* ez_irgdint_1_nw.inc */
            int i = imin(ni - 1, imax(1, (int)px));
            double x1 = g->ax[i - 1], x2 = g->ax[i];
            double x = (double)g->ax[i - 1] + (x2 - x1) * (double)(px - (float)i);
            idx[0] = i; idx[1] = i + 1; w[0] = (x - x1) / (x2 - x1);
        } else {                                                                              /* ez_irgdint_1_w.inc */
            int limite = ni + 2 - wrap, i = imin(ni - 2 + wrap, imax(1, (int)px)), ip1 = i + 1;
            double x1 = g->ax[i - 1], x2 = 0.0;
            if (ip1 <= ni) x2 = g->ax[ip1 - 1];
            if (i == (ni - 2 + wrap)) { ip1 = (limite + i + 1) % limite; x2 = (double)(g->ax[1] + g->ax[ni - 1]); }
            double x = x1 + (x2 - x1) * (double)(px - (float)i);
            idx[0] = i; idx[1] = ip1; w[0] = (x - x1) / (x2 - x1);
        }
        return;
    }
    if (!irr) {
        if (wrap == 0) {                                                                      /* ez_rgdint_3_nw.inc */
            int i = imin(ni - 2, imax(2, (int)px));
            idx[0] = i - 1; idx[1] = i; idx[2] = i + 1; idx[3] = i + 2;
            cubic_weights(w, (double)(px - (float)i));
        } else {                                                                              /* ez_rgdint_3_w.inc */
            int i = imin(ni - 2 + wrap, imax(1, imax(2 - wrap, (int)px)));
            reg_cols(ni, wrap, &i, idx);
            cubic_weights(w, (double)(px - (float)i));
        }
        return;
    }
    /* ez_irgdint_3_w.inc (wrap > 0; wrap == 0 never reaches the separable kernel) */
    int i = imin(ni - 2 + wrap, imax(1, imax(2 - wrap, (int)px)));
    float xs[4];
    irr_cols(g->ax, ni, wrap, i, 2, idx, xs);
    double x = (double)(xs[1] + (xs[2] - xs[1]) * (px - (float)i));
    double c[6];
    for (int k = 0; k < 6; k++) c[k] = (double)g->ncx[k * ni + (i - 1)];
    newton_weights(w, c, x, (double)xs[0], (double)xs[1], (double)xs[2]);
}

/* column taps for the polar STRIP kernels (ez_corrval_aunord.c:60-104) */
static void col_taps_strip(const ezh_grid *g, int degree, float px, int idx[4], double w[4])
{
    int ni = g->ni, wrap = g->extension;
    idx[0] = idx[1] = idx[2] = idx[3] = 1; w[0] = w[1] = w[2] = w[3] = 0.0;
    if (degree == DEG_NEAREST) { idx[0] = imin(ni, imax(1, nintf_(px))); return; }
    if (degree == DEG_LINEAR) {                                                               /* ez_rgdint_1_w on the strip */
        int limite = ni + 2 - wrap, i = imin(ni - 2 + wrap, imax(1, (int)px)), ip1 = i + 1;
        if (wrap > 0 && i == (ni - 2 + wrap)) ip1 = (limite + i + 1) % limite;
        idx[0] = i; idx[1] = ip1; w[0] = (double)(px - (float)i);
        return;
    }
    int i = imin(ni - 2 + wrap, imax(1, imax(2 - wrap, (int)px)));
    if (g->grtyp == 'Z' || g->grtyp == 'E' || g->grtyp == 'G') {                             /* ez_irgdint_3_wnnc.inc */
        float xs[4];
        irr_cols(g->ax, ni, wrap, i, 1, idx, xs);
        double x1 = xs[0], x2 = xs[1], x3 = xs[2], x4 = xs[3];
        double x = x2 + (x3 - x2) * (double)(px - (float)i);
        double c[6] = {1.0 / (x2 - x1), 1.0 / (x3 - x1), 1.0 / (x3 - x2), 1.0 / (x4 - x1), 1.0 / (x4 - x2), 1.0 / (x4 - x3)};
        newton_weights(w, c, x, x1, x2, x3);
    } else {                                                                                  /* ez_rgdint_3_wnnc.inc */
        if ((wrap > 0 && i <= 1) || i >= (ni - 1)) reg_cols(ni, wrap, &i, idx);
        else { idx[0] = i - 1; idx[1] = i; idx[2] = i + 1; idx[3] = i + 2; }
        cubic_weights(w, (double)(px - (float)i));
    }
}

/* row taps of the MAIN kernel: base row (1-based j of the first tap) and weights */
static void row_taps_main(const ezh_grid *g, int degree, float py, int *jfirst, double w[4])
{
    int nj = g->nj, j1 = g->j1, j2 = g->j2, wrap = g->extension, irr = src_irregular(g);
    w[0] = w[1] = w[2] = w[3] = 0.0;
    if (degree == DEG_NEAREST) { *jfirst = imin(j2, imax(j1, nintf_(py))); return; }
    if (degree == DEG_LINEAR) {
        if (!irr) { int j = imin(j2 - 1, imax(j1, (int)py)); *jfirst = j; w[0] = (double)(py - (float)j); return; }
        int j;
        if (wrap == 0) j = imin(nj - 1, imax(1, (int)py));
        else { j = imin(j2 - 1, imax(j1 + 1, (int)py)); if (j < 0) j = j - 1; }
        float a = g->ay[j - 1], b = g->ay[j];
        double y = (double)(a + (b - a) * (py - (float)j));
        *jfirst = j; w[0] = (y - (double)a) / (double)(b - a);
        return;
    }
    int j = imin(j2 - 2, imax(j1 + 1, (int)py));
    *jfirst = j - 1;
    if (!irr) { cubic_weights(w, (double)(py - (float)j)); return; }
    const float *ay = g->ay;
    double y = (double)(ay[j - 1] + (ay[j] - ay[j - 1]) * (py - (float)j));
    double c[6];
    for (int k = 0; k < 6; k++) c[k] = (double)g->ncy[k * nj + (j - 1)];
    newton_weights(w, c, y, (double)ay[j - 2], (double)ay[j - 1], (double)ay[j]);
}

/* one polar-strip row: ez_corrval_aunord.c:41-111 (north = 1), ez_corrval_ausud.c:55-126 (north = 0) */
static void strip_row(const ezh_grid *g, int degree, float py, int north, ezhip_special_row *sr)
{
    int j1 = g->j1, j2 = g->j2;
    int j1s = north ? j2 - 2 : j1 - 1, j2s = j1s + 3;
    int rows[4] = {0, 0, 0, 0}, ntap;
    sr->kind = 0; sr->w[0] = sr->w[1] = sr->w[2] = sr->w[3] = 0.0;
    if (degree == DEG_CUBIC) {
        int j = imin(j2s - 2, imax(j1s + 1, (int)py));
        ntap = 4;
        for (int k = 0; k < 4; k++) rows[k] = j - 1 + k;
        if (g->grtyp == 'Z' || g->grtyp == 'E' || g->grtyp == 'G') {
            float ay4[4];
            if (north) {
                if (g->ay[j2 - 1] == 90.0) { ay4[0] = g->ay[j2 - 4]; ay4[1] = g->ay[j2 - 3]; ay4[2] = g->ay[j2 - 2]; ay4[3] = g->ay[j2 - 1]; }
                else { ay4[0] = g->ay[j2 - 3]; ay4[1] = g->ay[j2 - 2]; ay4[2] = g->ay[j2 - 1]; ay4[3] = 90.0f; }
            } else {
                if (g->ay[j1 - 1] == -90.0) { ay4[0] = g->ay[0]; ay4[1] = g->ay[1]; ay4[2] = g->ay[2]; ay4[3] = g->ay[3]; }
                else { ay4[0] = -90.0f; ay4[1] = g->ay[0]; ay4[2] = g->ay[1]; ay4[3] = g->ay[2]; }
            }
            const float *b = ay4 - j1s;
            double y = (double)(b[j] + (b[j + 1] - b[j]) * (py - (float)j));
            double y1 = b[j - 1], y2 = b[j], y3 = b[j + 1], y4 = b[j + 2];
            double c[6] = {1.0 / (y2 - y1), 1.0 / (y3 - y1), 1.0 / (y3 - y2), 1.0 / (y4 - y1), 1.0 / (y4 - y2), 1.0 / (y4 - y3)};
            newton_weights(sr->w, c, y, y1, y2, y3);
        } else cubic_weights(sr->w, (double)(py - (float)j));
    } else if (north) {
        float ty = (float)((double)py - (1.0 * (j2 - 3)));
        int off = j2 - 3;
        if (degree == DEG_LINEAR) { int j = imin(3, imax(1, (int)ty)); ntap = 2; rows[0] = j + off; rows[1] = j + 1 + off; sr->w[0] = (double)(ty - (float)j); }
        else { int j = imin(4, imax(1, nintf_(ty))); ntap = 1; rows[0] = j + off; }
    } else {
        if (degree == DEG_LINEAR) { int j = imin(j2s - 1, imax(j1s, (int)py)); ntap = 2; rows[0] = j; rows[1] = j + 1; sr->w[0] = (double)(py - (float)j); }
        else { int j = imin(j2s, imax(j1s, nintf_(py))); ntap = 1; rows[0] = j; }
    }
    for (int k = 0; k < 4; k++) {
        int r = k < ntap ? rows[k] : rows[0];
        sr->tap[k] = r > j2 ? EZ_ROW_POLE_N : r < j1 ? EZ_ROW_POLE_S : r - j1;
    }
}

/* k_sepx geometry: valid row-blocks, their staging steps (first-of-segment / continuing), ring slots.
 * Leaves p->x_nseg == 0 when the plan does not qualify (then k_sep runs). */
static void build_sepx_geometry(ezh_sepplan *sp, ezhip_sep_plan *p, int degree, int ntap, int nbx, int nic, int njr,
                                const int *blk_base, const int *rbase, const double *rw, const unsigned char *rflag)
{
    p->x_nseg = 0;
    p->debug_flags = EZH_DEVINT("EZHIP_DEBUG");
    for (int bx = 0; bx < nbx; bx++) if (blk_base[bx] < 0) return;
    if ((size_t)njr * (size_t)nic >= ((size_t)1 << 31)) return;
    const int xr_rows = EZHIP_SEP_ROWS;          /* target rows per step (row-block of k_sepx) */
    int nby = (njr + xr_rows - 1) / xr_rows;
    int *vb = (int *)malloc(sizeof(int) * (nby + 1)), *wa = (int *)malloc(sizeof(int) * (nby + 1)), *wn = (int *)malloc(sizeof(int) * (nby + 1));
    int nvb = 0, tr = 4, ok = 1;
    for (int by = 0; by < nby && ok; by++) {
        int r0 = by * xr_rows, r1 = imin(r0 + xr_rows, njr), lo = 1 << 30, hi = -1;
        for (int r = r0; r < r1; r++) {
            if (rflag[r]) continue;
            if (rbase[r] < lo) lo = rbase[r];
            if (rbase[r] + ntap > hi) hi = rbase[r] + ntap;
        }
        if (hi < 0) continue;                                /* no main row */
        if (hi - lo > EZHIP_SEP_RMAX) { ok = 0; break; }     /* source window taller than the patch: k_sep's gather path */
        if (hi - lo > tr) tr = hi - lo;
        wa[nvb] = lo; wn[nvb] = hi - lo; vb[nvb++] = by;
    }
    if (!ok || nvb == 0) { free(vb); free(wa); free(wn); return; }
    ezhip_xstep *first = (ezhip_xstep *)calloc(nvb, sizeof(ezhip_xstep)), *cont = (ezhip_xstep *)calloc(nvb, sizeof(ezhip_xstep));
    ezhip_xrow *xr = (ezhip_xrow *)calloc((size_t)nvb * xr_rows, sizeof(ezhip_xrow));
    int prows = 1;
    for (int i = 0; i < nvb; i++) {
        int by = vb[i], a = wa[i], b = a + wn[i];
        first[i].s0 = a; first[i].n = b - a; first[i].slot0 = a % tr; first[i].by = by;
        cont[i] = first[i];
        if (i > 0) {       /* rows already in the ring: the previous window [pa, pb) */
            int pa = wa[i - 1], pb = pa + wn[i - 1];
            if (a >= pa && b >= pb && a <= pb) { cont[i].s0 = pb; cont[i].n = b - pb; }
            else if (a <= pa && b <= pb && b >= pa) { cont[i].s0 = a; cont[i].n = pa - a; }
            cont[i].slot0 = cont[i].s0 % tr;
        }
        if (first[i].n > prows) prows = first[i].n;
        /* target rows: a row that is not a main row re-stores the nearest main row of the same row-block */
        int r0 = by * xr_rows, lastmain = -1;
        for (int k = 0; k < xr_rows; k++) { int r = r0 + k; if (r < njr && !rflag[r]) { lastmain = r; break; } }
        for (int k = 0; k < xr_rows; k++) {
            int r = r0 + k;
            if (r < njr && !rflag[r]) lastmain = r;
            ezhip_xrow *q = &xr[(size_t)i * xr_rows + k];
            for (int w = 0; w < 4; w++) {
                q->w[w] = rw[w * njr + lastmain];
                q->t_off[w] = ((rbase[lastmain] + w) % tr) * (int)(EZHIP_SEP_COLS * sizeof(double));
            }
            q->o_off = (unsigned)((size_t)lastmain * (size_t)nic);
        }
    }
    size_t lds = ezhip_sepx_lds_bytes(tr, xr_rows, prows, p->wstride);
    int cap = ezhip_sepx_capacity(degree, xr_rows, lds);
    if (cap <= 0) cap = 768;
    int nseg = cap / nbx;
    if (nseg < 1) nseg = 1;
    int rb = (nvb + nseg - 1) / nseg;
    if (getenv("EZHIP_SEPX_RB")) rb = atoi(getenv("EZHIP_SEPX_RB"));
    if (rb < 1) rb = 1;
    p->x_rb = rb; p->x_nseg = (nvb + rb - 1) / rb; p->x_nvb = nvb;
    p->x_tr = tr; p->x_prows = prows; p->x_rows_per_step = xr_rows;
    if (getenv("EZHIP_VERBOSE"))
        fprintf(stderr, "k_sepx plan: degree %d, %d column blocks x %d segments of %d steps of %d rows (%d valid), ring %d rows, patch %d rows x %d, LDS %zu B, capacity %d blocks\n",
                degree, nbx, p->x_nseg, rb, xr_rows, nvb, tr, prows, p->wstride, lds, cap);
    p->x_first = (const ezhip_xstep *)(sp->dev[21] = upload(first, sizeof(ezhip_xstep) * nvb));
    p->x_cont = (const ezhip_xstep *)(sp->dev[22] = upload(cont, sizeof(ezhip_xstep) * nvb));
    p->x_rows = (const ezhip_xrow *)(sp->dev[23] = upload(xr, sizeof(ezhip_xrow) * (size_t)nvb * xr_rows));
    if (!sp->dev[21] || !sp->dev[22] || !sp->dev[23]) p->x_nseg = 0;
    sp->h_nvb = nvb;
    sp->h_vb_by = (int *)malloc(sizeof(int) * nvb); sp->h_vb_send = (int *)malloc(sizeof(int) * nvb);
    sp->h_rflag = (unsigned char *)malloc((size_t)njr);
    if (sp->h_vb_by && sp->h_vb_send && sp->h_rflag) {
        for (int i = 0; i < nvb; i++) { sp->h_vb_by[i] = vb[i]; sp->h_vb_send[i] = wa[i] + wn[i]; }
        memcpy(sp->h_rflag, rflag, (size_t)njr);
    } else sp->h_nvb = 0;
    free(first); free(cont); free(xr); free(vb); free(wa); free(wn);
}

/* Tables of the exact-extrema pass (ezhip_sep_plan.bb_*, k_bb_* in ez_kernels.hip): which source windows hold target points, and which.
 * A value of the separable kernels is sum_j wy_j (sum_k wx_k z_jk) over ntap x ntap source points -- the WINDOW named by its first source column
 * (modulo ni: the longitude seam) and first source row.  Not built (bb_ok = 0) for vector plans, DEHORS columns / fill rows, taps that are not
 * consecutive, or windows that leave the source rows. */
static void build_bb_tables(ezh_sepplan *sp, ezhip_sep_plan *p, const ezh_grid *gi, int degree, int ntap, int nic, int njr, const int *cidx, const double *cw,
                            const unsigned char *cflag, const int *rbase, const double *rw, const unsigned char *rflag,
                            const ezhip_special_row *special, int nspecial, int vector_mode)
{
    p->bb_ok = 0; p->bb_ntap = ntap;
    if (vector_mode || getenv("EZHIP_NO_BB")) return;
    const int nis = gi->ni, njs = gi->nj;
    if (nis < ntap + 3 || njs < ntap) return;
    for (int c = 0; c < nic; c++) {
        if (cflag[c]) return;
        if (cidx[c] < 0 || cidx[c] >= nis) return;
        for (int k = 1; k < ntap; k++) if (cidx[k * nic + c] != (cidx[c] + k) % nis) return;
    }
    for (int k = 0; k < nspecial; k++) if (special[k].kind == 3) return;
    int nmain = 0;
    for (int r = 0; r < njr; r++) {
        if (rflag[r]) continue;
        if (rbase[r] < 0 || rbase[r] + ntap > njs) return;
        nmain++;
    }
    /* K = max sum |wx| * max sum |wy|; sdev bounds |sum wx * sum wy - 1| (cubic: the weights are the reference's Newton form evaluated on
     * unit vectors; their sum is 1 up to REAL*8 rounding) */
    double kx = 1.0, ky = 1.0, dx = 0.0, dy = 0.0;
    float *colk = (float *)malloc(sizeof(float) * (size_t)nis), *rowk = (float *)malloc(sizeof(float) * (size_t)njs);
    if (!colk || !rowk) { free(colk); free(rowk); return; }
    for (int i = 0; i < nis; i++) colk[i] = 1.0f;
    for (int j = 0; j < njs; j++) rowk[j] = 1.0f;
    for (int c = 0; c < nic; c++) {
        double sa = 0.0, sm = 0.0;
        if (degree == DEG_CUBIC) for (int k = 0; k < 4; k++) { sa += fabs(cw[k * nic + c]); sm += cw[k * nic + c]; }
        else if (degree == DEG_LINEAR) { sa = fabs(1.0 - cw[c]) + fabs(cw[c]); sm = 1.0; }
        else { sa = 1.0; sm = 1.0; }
        if (!(sa <= 1.0e6)) { free(colk); free(rowk); return; }                           /* NaN / absurd weights: no bound */
        if (sa > kx) kx = sa;
        { float k = nextafterf((float)sa, INFINITY); if (k > colk[cidx[c]]) colk[cidx[c]] = k; }
        if (fabs(sm - 1.0) > dx) dx = fabs(sm - 1.0);
    }
    for (int r = 0; r < njr; r++) {
        if (rflag[r]) continue;
        double sa = 0.0, sm = 0.0;
        if (degree == DEG_CUBIC) for (int k = 0; k < 4; k++) { sa += fabs(rw[k * njr + r]); sm += rw[k * njr + r]; }
        else if (degree == DEG_LINEAR) { sa = fabs(1.0 - rw[r]) + fabs(rw[r]); sm = 1.0; }
        else { sa = 1.0; sm = 1.0; }
        if (!(sa <= 1.0e6)) { free(colk); free(rowk); return; }
        if (sa > ky) ky = sa;
        { float k = nextafterf((float)sa, INFINITY); if (k > rowk[rbase[r]]) rowk[rbase[r]] = k; }
        if (fabs(sm - 1.0) > dy) dy = fabs(sm - 1.0);
    }
    const double K = kx * ky;
    p->bb_a = 0.5 * (K - 1.0) * (1.0 + 1.0e-12) + 1.0e-12;
    p->bb_s = (dx + dy + dx * dy) * kx * ky + K * 64.0 * 2.220446049250313e-16 + 1.0e-13;
    unsigned char *colhas = (unsigned char *)calloc((size_t)nis, 1), *rowhas = (unsigned char *)calloc((size_t)njs, 1);
    int *colstart = (int *)calloc((size_t)nis + 1, sizeof(int)), *rowstart = (int *)calloc((size_t)njs + 1, sizeof(int));
    int *collist = (int *)malloc(sizeof(int) * (size_t)(nic > 0 ? nic : 1)), *rowlist = (int *)malloc(sizeof(int) * (size_t)(nmain > 0 ? nmain : 1));
    if (colhas && rowhas && colstart && rowstart && collist && rowlist) {
        for (int c = 0; c < nic; c++) { colhas[cidx[c]] = 1; colstart[cidx[c] + 1]++; }
        for (int i = 0; i < nis; i++) colstart[i + 1] += colstart[i];
        int *fill = (int *)calloc((size_t)(nis > njs ? nis : njs), sizeof(int));
        if (fill) {
            for (int c = 0; c < nic; c++) collist[colstart[cidx[c]] + fill[cidx[c]]++] = c;
            memset(fill, 0, sizeof(int) * (size_t)(nis > njs ? nis : njs));
            for (int r = 0; r < njr; r++) if (!rflag[r]) { rowhas[rbase[r]] = 1; rowstart[rbase[r] + 1]++; }
            for (int j = 0; j < njs; j++) rowstart[j + 1] += rowstart[j];
            for (int r = 0; r < njr; r++) if (!rflag[r]) rowlist[rowstart[rbase[r]] + fill[rbase[r]]++] = r;
            free(fill);
            p->bb_colhas = (const unsigned char *)(sp->dev[24] = upload(colhas, (size_t)nis));
            p->bb_rowhas = (const unsigned char *)(sp->dev[25] = upload(rowhas, (size_t)njs));
            p->bb_colstart = (const int *)(sp->dev[26] = upload(colstart, sizeof(int) * ((size_t)nis + 1)));
            p->bb_collist = (const int *)(sp->dev[27] = upload(collist, sizeof(int) * (size_t)(nic > 0 ? nic : 1)));
            p->bb_rowstart = (const int *)(sp->dev[28] = upload(rowstart, sizeof(int) * ((size_t)njs + 1)));
            p->bb_rowlist = (const int *)(sp->dev[29] = upload(rowlist, sizeof(int) * (size_t)(nmain > 0 ? nmain : 1)));
            p->bb_colk = (const float *)(sp->dev[30] = upload(colk, sizeof(float) * (size_t)nis));
            p->bb_rowk = (const float *)(sp->dev[31] = upload(rowk, sizeof(float) * (size_t)njs));
            p->bb_ok = sp->dev[24] && sp->dev[25] && sp->dev[26] && sp->dev[27] && sp->dev[28] && sp->dev[29] && sp->dev[30] && sp->dev[31];
            ezhip_sync();
        }
    }
    free(colhas); free(rowhas); free(colstart); free(rowstart); free(collist); free(rowlist); free(colk); free(rowk);
}


/* geometry of k_sepx_enc (interpolate + armn_compress in one launch): strips of 256 target columns every 255, row groups of 16 target rows every
 * 15 (a context column / row + 85 x 5 tiles of 3 x 3); leaves p->e_ok == 0 when a strip's taps are not consecutive patch columns, a row group's
 * source window is taller than the patch, or the plan has fill columns / rows (then the two-kernel pipeline runs) */
static void build_enc_geometry(ezh_sepplan *sp, ezhip_sep_plan *p, const ezh_grid *gi, int ntap, int nic, int njr, const int *cidx,
                               const unsigned char *cflag, const int *rbase, const double *rw, const unsigned char *rflag,
                               const ezhip_special_row *special, int nspecial)
{
    p->e_ok = 0;
    if (p->x_nseg <= 0 || nic < 256 || njr < 16 || getenv("EZHIP_NO_SEPX_ENC")) return;
    if ((size_t)njr * (size_t)nic >= ((size_t)1 << 31)) return;
    for (int c = 0; c < nic; c++) if (cflag[c]) return;
    for (int k = 0; k < nspecial; k++) if (special[k].kind == 3) return;
    const int CS = 255, RS = 15;
    const int ntx = (nic - 1 + 2) / 3, nty = (njr - 1 + 2) / 3;
    const int nstrips = (ntx + 84) / 85, nrg = (nty + 4) / 5;
    int *bbase = (int *)malloc(sizeof(int) * nstrips), *bw = (int *)malloc(sizeof(int) * nstrips);
    ezhip_xstep *step = (ezhip_xstep *)calloc(nrg, sizeof(ezhip_xstep));
    ezhip_xrow *xr = (ezhip_xrow *)calloc((size_t)nrg * 16, sizeof(ezhip_xrow));
    int *esp = (int *)malloc(sizeof(int) * (size_t)nrg * 16), *row2sp = (int *)malloc(sizeof(int) * njr);
    if (!bbase || !bw || !step || !xr || !esp || !row2sp) goto done;
    int wmax = 1, ok = 1;
    for (int s = 0; s < nstrips && ok; s++) {
        const int c0 = s * CS, c1 = imin(c0 + 256, nic);
        const int base = cidx[c0];
        int maxoff = 0;
        for (int c = c0; c < c1 && ok; c++) {
            const int off0 = (cidx[c] - base + gi->ni) % gi->ni;
            for (int k = 0; k < ntap; k++) {
                const int off = (cidx[k * nic + c] - base + gi->ni) % gi->ni;
                if (off != off0 + k) { ok = 0; break; }
                if (off > maxoff) maxoff = off;
            }
        }
        if (maxoff + 1 > EZHIP_SEP_WMAX || maxoff + 1 > gi->ni) ok = 0;
        bbase[s] = base; bw[s] = maxoff + 1;
        if (bw[s] > wmax) wmax = bw[s];
    }
    if (!ok) goto done;
    for (int r = 0; r < njr; r++) row2sp[r] = -1;
    for (int k = 0; k < nspecial; k++) if (special[k].row >= 0 && special[k].row < njr) row2sp[special[k].row] = k;
    int tr = 4, prows = 1;
    for (int i = 0; i < nrg; i++) {           /* first pass: the ring must hold the tallest window */
        int lo = 1 << 30, hi = -1;
        for (int k = 0; k < 16; k++) { const int r = i * RS + k; if (r >= njr || rflag[r]) continue; if (rbase[r] < lo) lo = rbase[r]; if (rbase[r] + ntap > hi) hi = rbase[r] + ntap; }
        if (hi < 0) { step[i].s0 = 0; step[i].n = 0; continue; }
        if (hi - lo > EZHIP_SEP_RMAX) { ok = 0; break; }
        step[i].s0 = lo; step[i].n = hi - lo; step[i].by = i;
        if (hi - lo > tr) tr = hi - lo;
        if (hi - lo > prows) prows = hi - lo;
    }
    if (!ok) goto done;
    for (int i = 0; i < nrg; i++) {
        step[i].slot0 = step[i].s0 % tr;
        int lastmain = -1;
        for (int k = 0; k < 16; k++) { const int r = i * RS + k; if (r < njr && !rflag[r]) { lastmain = r; break; } }
        for (int k = 0; k < 16; k++) {
            const int r = i * RS + k;
            esp[i * 16 + k] = (r < njr && rflag[r]) ? row2sp[r] : -1;
            if (r < njr && rflag[r] && row2sp[r] < 0) ok = 0;                       /* a flagged row without a special record: not expected */
            if (r < njr && !rflag[r]) lastmain = r;
            ezhip_xrow *q = &xr[(size_t)i * 16 + k];
            if (lastmain < 0) continue;                                              /* a row group without a main row: no y-pass */
            for (int w = 0; w < 4; w++) {
                q->w[w] = rw[w * njr + lastmain];
                q->t_off[w] = ((rbase[lastmain] + w) % tr) * (int)(EZHIP_SEP_COLS * sizeof(double));
            }
        }
    }
    if (!ok) goto done;
    p->e_nstrips = nstrips; p->e_nrg = nrg; p->e_tr = tr; p->e_prows = prows; p->e_wstride = 64 * ((wmax + 63) / 64);
    p->e_blk_base = (const int *)(sp->dev[32] = upload(bbase, sizeof(int) * nstrips));
    p->e_blk_w = (const int *)(sp->dev[33] = upload(bw, sizeof(int) * nstrips));
    p->e_step = (const ezhip_xstep *)(sp->dev[34] = upload(step, sizeof(ezhip_xstep) * nrg));
    p->e_rows = (const ezhip_xrow *)(sp->dev[35] = upload(xr, sizeof(ezhip_xrow) * (size_t)nrg * 16));
    p->e_special = (const int *)(sp->dev[36] = upload(esp, sizeof(int) * (size_t)nrg * 16));
    p->e_ok = sp->dev[32] && sp->dev[33] && sp->dev[34] && sp->dev[35] && sp->dev[36];
    if (p->e_ok && ezhip_sepenc_lds_bytes(p) > 150 * 1024) p->e_ok = 0;
    if (getenv("EZHIP_VERBOSE"))
        fprintf(stderr, "k_sepx_enc plan: %d strips x %d row groups, ring %d rows, patch %d rows x %d, LDS %zu B, ok %d\n", nstrips, nrg, tr, prows, p->e_wstride, ezhip_sepenc_lds_bytes(p), p->e_ok);
    ezhip_sync();
done:
    free(bbase); free(bw); free(step); free(xr); free(esp); free(row2sp);
}

static double wall_ms(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return 1e3 * (double)t.tv_sec + 1e-6 * (double)t.tv_nsec; }
/* EZHIP_TRACE_FIRST=1: a set's first-call work by part, on stderr */
#define TRACE_PART(label) do { if (trace) { const double t1_ = wall_ms(); fprintf(stderr, "%s: %s %.2f ms\n", __func__, label, t1_ - t0_); t0_ = t1_; } } while (0)
static int build_sep_plan(ezh_set *s, int degree, int vector_mode, int polar)
{
    const int trace = getenv("EZHIP_TRACE_FIRST") != NULL;
    double t0_ = trace ? wall_ms() : 0.0;
    if (degree != DEG_NEAREST && degree != DEG_LINEAR && degree != DEG_CUBIC) return -1;
    int di = degree == DEG_CUBIC ? 2 : degree;
    ezh_sepplan *sp = &s->sep[di][vector_mode];
    /* plans depend on polar_correction: bit 1 of built records the setting they were built for */
    if (sp->built == 1 + polar) return 0;
    free_sepplan(sp);
    ezh_grid *gi = &G[s->gdin], *go = &G[s->gdout];
    int nic = go->ni, njr = go->nj;
    int *cidx = (int *)malloc(sizeof(int) * 4 * nic), *cidx_s = (int *)malloc(sizeof(int) * 4 * nic);
    double *cw = (double *)malloc(sizeof(double) * 4 * nic), *cw_s = (double *)malloc(sizeof(double) * 4 * nic);
    unsigned char *cflag = (unsigned char *)calloc(nic, 1), *rflag = (unsigned char *)calloc(njr, 1);
    int *rbase = (int *)malloc(sizeof(int) * njr);
    double *rw = (double *)malloc(sizeof(double) * 4 * njr);
    ezhip_special_row *special = (ezhip_special_row *)calloc((size_t)njr + 1, sizeof(ezhip_special_row));
    int nspecial = 0;
    for (int c = 0; c < nic; c++) {
        int idx[4]; double w[4];
        col_taps_main(gi, degree, s->x1d[c], idx, w);
        for (int k = 0; k < 4; k++) { cidx[k * nic + c] = idx[k] - 1; cw[k * nic + c] = w[k]; }
        col_taps_strip(gi, degree, s->x1d[c], idx, w);
        for (int k = 0; k < 4; k++) { cidx_s[k * nic + c] = idx[k] - 1; cw_s[k * nic + c] = w[k]; }
        if (polar && s->extrap) { int ix = (int)((double)s->x1d[c] + 0.5); if (ix < 1 || ix > gi->ni) cflag[c] = EZF_DEHORS; }
    }
    for (int r = 0; r < njr; r++) {
        float py = s->y1d[r];
        int jf; double w[4];
        row_taps_main(gi, degree, py, &jf, w);
        rbase[r] = jf - gi->j1;
        for (int k = 0; k < 4; k++) rw[k * njr + r] = w[k];
        if (!polar) continue;
        ezhip_special_row *sr = &special[nspecial];
        sr->row = r;
        if (s->extrap) {                                          /* ez_defzone_dehors.c:63-74 */
            int iy = (int)((double)py + 0.5);
            if (iy < 1 || iy > gi->nj) { sr->kind = 3; rflag[r] = 1; nspecial++; }
            continue;
        }
        int au_n = (int)py > (gi->j2 - 2), au_s = (int)py < (gi->j1 + 1);
        int po_n = fabs((double)(py - s->ypole_n)) < 1.0e-3, po_s = fabs((double)(py - s->ypole_s)) < 1.0e-3;
        if (vector_mode) { po_n = po_s = 0; }
        if (po_s) sr->kind = 2;
        else if (po_n) sr->kind = 1;
        else if (au_s) strip_row(gi, degree, py, 0, sr);
        else if (au_n) strip_row(gi, degree, py, 1, sr);
        else continue;
        rflag[r] = 1; nspecial++;
    }
    /* LDS staging geometry of k_sep: per column-block the contiguous (modulo ni) source column window,
     * per row-block the source row window.  A block whose taps do not fit falls back to gathers. */
    const int ntap = degree == DEG_CUBIC ? 4 : (degree == DEG_LINEAR ? 2 : 1);
    int nbx = (nic + EZHIP_SEP_COLS - 1) / EZHIP_SEP_COLS, nby = (njr + EZHIP_SEP_ROWS - 1) / EZHIP_SEP_ROWS;
    int *coff = (int *)calloc((size_t)4 * nic, sizeof(int));
    int *blk_base = (int *)malloc(sizeof(int) * nbx), *blk_w = (int *)malloc(sizeof(int) * nbx);
    int *brow_s0 = (int *)malloc(sizeof(int) * nby), *brow_n = (int *)malloc(sizeof(int) * nby);
    for (int bx = 0; bx < nbx; bx++) {
        int c0 = bx * EZHIP_SEP_COLS, c1 = imin(c0 + EZHIP_SEP_COLS, nic);
        int base = cidx[c0], maxoff = 0;
        for (int c = c0; c < c1; c++)
            for (int k = 0; k < ntap; k++) {
                int off = (cidx[k * nic + c] - base + gi->ni) % gi->ni;
                coff[k * nic + c] = off;
                if (off > maxoff) maxoff = off;
            }
        int consecutive = 1;      /* the staged kernel reads taps at patch columns off0, off0+1, .. */
        for (int c = c0; c < c1 && consecutive; c++)
            for (int k = 1; k < ntap; k++) if (coff[k * nic + c] != coff[c] + k) { consecutive = 0; break; }
        if (!consecutive || maxoff + 1 > EZHIP_SEP_WMAX || maxoff + 1 > gi->ni) { blk_base[bx] = -1; blk_w[bx] = 0; }
        else { blk_base[bx] = base; blk_w[bx] = maxoff + 1; }
    }
    for (int by = 0; by < nby; by++) {
        int r0 = by * EZHIP_SEP_ROWS, r1 = imin(r0 + EZHIP_SEP_ROWS, njr), lo = 1 << 30, hi = -1;
        for (int r = r0; r < r1; r++) {
            if (rflag[r]) continue;
            if (rbase[r] < lo) lo = rbase[r];
            if (rbase[r] + ntap > hi) hi = rbase[r] + ntap;
        }
        if (hi < 0 || hi - lo > EZHIP_SEP_RMAX) { brow_s0[by] = 0; brow_n[by] = 0; }
        else { brow_s0[by] = lo; brow_n[by] = hi - lo; }
    }
    ezhip_rowinfo *rinfo = (ezhip_rowinfo *)calloc((size_t)nby * EZHIP_SEP_ROWS, sizeof(ezhip_rowinfo));
    for (int by = 0; by < nby; by++)
        for (int k = 0; k < EZHIP_SEP_ROWS; k++) {
            int r = by * EZHIP_SEP_ROWS + k;
            ezhip_rowinfo *q = &rinfo[r];
            if (r >= njr || rflag[r]) { q->flag = 1; continue; }
            q->jb = rbase[r] - brow_s0[by];
            for (int w = 0; w < 4; w++) q->w[w] = rw[w * njr + r];
        }
    /* pad0 = 1 on every row of a row-block that the straight-line kernel body can take: 16 main rows, each
     * advancing the source window by 0 or 1, and the whole window inside the staged patch */
    for (int by = 0; by < nby; by++) {
        int ok = brow_n[by] > 0 && !getenv("EZHIP_SEP_NOSIMPLE");
        for (int k = 0; k < EZHIP_SEP_ROWS && ok; k++) {
            int r = by * EZHIP_SEP_ROWS + k;
            if (r >= njr || rflag[r]) { ok = 0; break; }
            if (k > 0) { int d = rbase[r] - rbase[r - 1]; if (d < 0 || d > 1) ok = 0; }
        }
        for (int k = 0; k < EZHIP_SEP_ROWS; k++) rinfo[by * EZHIP_SEP_ROWS + k].pad0 = ok;
    }
    TRACE_PART("host tables (columns, rows, blocks)");
    ezhip_sep_plan *p = &sp->p;
    memset(p, 0, sizeof(*p));
    p->degree = degree; p->ni_src = gi->ni; p->nj_src = gi->nj; p->ni_dst = nic; p->nj_dst = njr;
    {   /* LDS geometry: row stride = widest staged block, rows = tallest staged row-block; the buffer also
         * serves the sequential pole sums of the special blocks (1025 floats) */
        int wmax = 1, nmax = 1;
        for (int bx = 0; bx < nbx; bx++) if (blk_w[bx] > wmax) wmax = blk_w[bx];
        for (int by = 0; by < nby; by++) if (brow_n[by] > nmax) nmax = brow_n[by];
        /* k_sepx stages rows with LDS-DMA: 16 bytes per lane (rows of 4-float pieces) or, for a window that wraps at the seam, 64 dwords per instruction with the
         * lanes past the row's end masked.  EZHIP_SEPX_WSTRIDE64: the round-2 layout (rows padded to whole 64-float chunks) */
        wmax = getenv("EZHIP_SEPX_WSTRIDE64") ? 64 * ((wmax + 63) / 64) : 4 * ((wmax + 3) / 4);
        p->wstride = wmax; p->patch_elems = wmax * nmax;
        p->nblk_y = nby;
    }
    p->rowinfo = (const ezhip_rowinfo *)(sp->dev[14] = upload(rinfo, sizeof(ezhip_rowinfo) * (size_t)nby * EZHIP_SEP_ROWS));
    free(rinfo);
    p->cidx = (const int *)(sp->dev[0] = upload(cidx, sizeof(int) * 4 * nic));
    p->cw = (const double *)(sp->dev[1] = upload(cw, sizeof(double) * 4 * nic));
    p->cidx_s = (const int *)(sp->dev[2] = upload(cidx_s, sizeof(int) * 4 * nic));
    p->cw_s = (const double *)(sp->dev[3] = upload(cw_s, sizeof(double) * 4 * nic));
    p->cflag = (const unsigned char *)(sp->dev[4] = upload(cflag, nic));
    p->rbase = (const int *)(sp->dev[5] = upload(rbase, sizeof(int) * njr));
    p->rw = (const double *)(sp->dev[6] = upload(rw, sizeof(double) * 4 * njr));
    p->rflag = (const unsigned char *)(sp->dev[7] = upload(rflag, njr));
    p->special = (const ezhip_special_row *)(sp->dev[8] = upload(special, sizeof(ezhip_special_row) * (nspecial + 1)));
    p->coff = (const int *)(sp->dev[9] = upload(coff, sizeof(int) * 4 * nic));
    p->blk_base = (const int *)(sp->dev[10] = upload(blk_base, sizeof(int) * nbx));
    p->blk_w = (const int *)(sp->dev[11] = upload(blk_w, sizeof(int) * nbx));
    p->brow_s0 = (const int *)(sp->dev[12] = upload(brow_s0, sizeof(int) * nby));
    p->brow_n = (const int *)(sp->dev[13] = upload(brow_n, sizeof(int) * nby));
    p->n_special = nspecial;
    for (int k = 0; k < nspecial && !vector_mode; k++) {
        if (special[k].kind == 1 || special[k].kind == 2) p->need_poles = 1;
        if (special[k].kind == 0) for (int q = 0; q < 4; q++) if (special[k].tap[q] < 0) p->need_poles = 1;
    }
    TRACE_PART("15 uploads");
    build_sepx_geometry(sp, p, degree, ntap, nbx, nic, njr, blk_base, rbase, rw, rflag);
    TRACE_PART("k_sepx geometry");
    p->pole_weighted = (gi->grtyp == 'Z' && gi->grref == 'E');
    p->vector_mode = vector_mode;
    build_bb_tables(sp, p, gi, degree, ntap, nic, njr, cidx, cw, cflag, rbase, rw, rflag, special, nspecial, vector_mode);
    TRACE_PART("bounds tables");
    if (!vector_mode) build_enc_geometry(sp, p, gi, ntap, nic, njr, cidx, cflag, rbase, rw, rflag, special, nspecial);
    TRACE_PART("encoder geometry");
    ezhip_sync();
    TRACE_PART("sync");
    int ok = 1;
    for (int k = 0; k < 15; k++) ok &= (sp->dev[k] != NULL);
    free(coff); free(blk_base); free(blk_w); free(brow_s0); free(brow_n);
    free(cidx); free(cidx_s); free(cw); free(cw_s); free(cflag); free(rflag); free(rbase); free(rw); free(special);
    if (!ok) { free_sepplan(sp); return -1; }
    sp->built = 1 + polar;
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* per-point plan                                                                               */
/* ------------------------------------------------------------------------------------------ */
static void fill_locate_plan(const ezh_grid *gi, ezhip_locate_plan *lp)
{
    memset(lp, 0, sizeof(*lp));
    lp->ni = gi->ni; lp->nj = gi->nj;
    float r[9], ri[9];
    switch (gi->grtyp) {
    case 'A': lp->kind = 0; lp->dlon = 360.0f / (float)gi->ni; lp->lon0 = 0.0f; lp->lon_fix = 2;
        if (gi->ig[0] == 1) { lp->dlat = 90.0f / (float)gi->nj; lp->lat0 = lp->dlat * 0.5f; }
        else if (gi->ig[0] == 2) { lp->dlat = 90.0f / (float)gi->nj; lp->lat0 = -90.0f + lp->dlat * 0.5f; }
        else { lp->dlat = 180.0f / (float)gi->nj; lp->lat0 = -90.0f + lp->dlat * 0.5f; }
        break;
    case 'B': lp->kind = 0; lp->dlon = 360.0f / (float)(gi->ni - 1); lp->lon0 = 0.0f; lp->lon_fix = 2;
        if (gi->ig[0] == 1) { lp->dlat = 90.0f / (float)(gi->nj - 1); lp->lat0 = 0.0f; }
        else if (gi->ig[0] == 2) { lp->dlat = 90.0f / (float)(gi->nj - 1); lp->lat0 = -90.0f; }
        else { lp->dlat = 180.0f / (float)(gi->nj - 1); lp->lat0 = -90.0f; }
        break;
    case 'L': lp->kind = 0; h_cigaxg('L', &lp->lat0, &lp->lon0, &lp->dlat, &lp->dlon, gi->ig[0], gi->ig[1], gi->ig[2], gi->ig[3]); lp->lon_fix = 1; break;
    case 'N': case 'S':                                     /* kind 4: ez_vxyfll with (pi, pj, d60, dgrw) in (lat0, lon0, dlat, dlon), hemisphere in lon_fix */
        lp->kind = 4; lp->lat0 = gi->xg[0]; lp->lon0 = gi->xg[1]; lp->dlat = gi->xg[2]; lp->dlon = gi->xg[3]; lp->lon_fix = gi->grtyp == 'N' ? 1 : 2;
        break;
    case 'E':
        lp->kind = 3; lp->dlon = 360.0f / (float)(gi->ni - 1); lp->lon0 = 0.0f; lp->dlat = 180.0f / (float)gi->nj; lp->lat0 = -90.f + 0.5f * lp->dlat;
        h_crot(r, ri, gi->xg[1], gi->xg[0], gi->xg[3], gi->xg[2]); memcpy(lp->r, r, sizeof(r));
        break;
    default:
        lp->ax = gi->d_ax; lp->ay = gi->d_ay;
        if (gi->grref == 'L') {
            lp->kind = 1;
            h_cigaxg('L', &lp->lat0, &lp->lon0, &lp->dlat, &lp->dlon, gi->igref[0], gi->igref[1], gi->igref[2], gi->igref[3]);
            lp->lonref = (gi->ax[0] < 0.0f) ? -180.0f : 0.0f;
        } else {
            lp->kind = 2;
            h_crot(r, ri, gi->xgref[1], gi->xgref[0], gi->xgref[3], gi->xgref[2]); memcpy(lp->r, r, sizeof(r));
        }
    }
}

/* x,y of every target point of a set on the per-point path (the gridset cache of ez_calcxy.c:28-137), device resident.
 * Rotated sources ('E', Z-on-'E': ez_gfxyfll.c:38-57 -- REAL trig through libm) are located on the device as well: k_locate
 * kinds 2 / 3 evaluate sinf / cosf / asinf / atan2f as GNU libc 2.35 does, operation by operation (libm_exact.h, equal to the C
 * library over every REAL argument: tools/check_libm_exact.c), so x, y have c_gdxyfll's bits
 * (test_rotated_source_locate_is_bit_exact).  EZHIP_HOST_LOCATE=1 keeps the host threads (h_locate_mt) that did this until
 * round 5 -- for a C library other than the one libm_exact.h restates.
 * The same first-call pass classifies the DEHORS zone (ez_defzone_dehors.c:63-74): have_dehors decides rc = 2 and
 * extrap_degree = abort on every entry point. */
/* Is the C library behind this process the one libm_exact.h restates?  Checked once, on the host compilation of the header: 4 x 16 384 arguments in the ranges the
 * rotated-frame locate uses plus arbitrary bit patterns (another implementation of sinf differs on a large share of arguments; tools/check_libm_exact.c is the
 * exhaustive form).  0: rotated sources are located by host threads through that library instead (what EZHIP_HOST_LOCATE=1 asks for) */
static int libm_exact_matches_host(void)
{
    static int state = -1;
    if (state >= 0) return state;
    unsigned long long r = 0x9E3779B97F4A7C15ULL;
    int ok = 1;
    for (int k = 0; k < 4096 && ok; k++) {
        r ^= r << 13; r ^= r >> 7; r ^= r << 17;
        const float any = glx_float((uint32_t)r), ang = (float)((double)(r >> 40) * (14.0 / 16777216.0) - 7.0), unit = (float)((double)((r >> 16) & 0xFFFFFF) * (2.0 / 16777216.0) - 1.0);
        const float other = (float)((double)(r & 0xFFFF) * (2.0 / 65536.0) - 1.0);
        const float a[4] = {any, ang, unit, other};
        for (int q = 0; q < 4 && ok; q++) {
            const float x = a[q], y = a[(q + 1) & 3];
            const float g0 = glx_sinf(x), h0 = sinf(x), g1 = glx_cosf(x), h1 = cosf(x), g2 = glx_asinf(x), h2 = asinf(x), g3 = glx_atan2f(x, y), h3 = atan2f(x, y);
            ok = (glx_bits(g0) == glx_bits(h0) || (g0 != g0 && h0 != h0)) && (glx_bits(g1) == glx_bits(h1) || (g1 != g1 && h1 != h1))
              && (glx_bits(g2) == glx_bits(h2) || (g2 != g2 && h2 != h2)) && (glx_bits(g3) == glx_bits(h3) || (g3 != g3 && h3 != h3));
        }
    }
    if (!ok) fprintf(stderr, "<ezscint> the C library's sinf / cosf / asinf / atan2f are not the ones libm_exact.h restates (GNU libc 2.35, x86-64 FMA variants): "
                             "rotated sources are located on host threads through the library itself\n");
    state = ok;
    return ok;
}
int32_t ezhip_libm_exact_matches_host(void) { return libm_exact_matches_host(); }
/* The in-place edits the reference's locate makes to the longitudes it is handed -- for the x, y of a grid set those are the TARGET grid's cached longitudes
 * (ez_calcxy.c:41-57: Grille[gdout].lon itself), which c_ezuvint's wind chain reads afterwards (and c_gdll returns): ez_ll2rgd.inc:137-145 for an 'L'
 * source, ez_llll2gd.inc:40-45 behind every regular and Z-on-L source.  h_locate makes them as it goes; this is the same without the locate, for targets whose 2-D longitudes
 * were located on the device or on copies (round 6: with them the exact-winds mode equals the reference bit for bit on regional 'L' sources too).  Returns 1 when something changed. */
static int h_lon_edits(const ezh_grid *g, float *lon, size_t n)
{
    float lat0, lon0, dlat, dlon, lonref = 0.0f;
    int changed = 0;
    switch (g->grtyp) {
    case 'A': case 'B':
        break;                                              /* (lon < 0 -> + 360: the rule of ez_llll2gd below) */
    case 'L':
        h_cigaxg('L', &lat0, &lon0, &dlat, &dlon, g->ig[0], g->ig[1], g->ig[2], g->ig[3]);
        for (size_t i = 0; i < n; i++) {
            const float v = lon[i];
            if (lon[i] < lon0) lon[i] = lon[i] + 360.0f;
            if (lon[i] > (lon0 + (float)g->ni * dlon)) lon[i] = lon[i] - 360.0f;
            if (lon[i] != v) changed = 1;
        }
        break;
    case 'G': case 'Z':
        if (g->grref != 'L') return 0;                      /* a rotated frame: the locate works on the rotated copies */
        lonref = (g->ax[0] < 0.0f) ? -180.0f : 0.0f;
        break;
    default:
        return 0;                                           /* E, N, S, !: nothing is edited */
    }
    if (lonref == -180.0f) { for (size_t i = 0; i < n; i++) if (lon[i] > 180.0f) { lon[i] = lon[i] - 360.0f; changed = 1; } }
    else { for (size_t i = 0; i < n; i++) if (lon[i] < 0.0f) { lon[i] = lon[i] + 360.0f; changed = 1; } }
    return changed;
}
static int ensure_points(ezh_set *s)
{
    if (s->d_x) return 0;
    const int trace = getenv("EZHIP_TRACE_FIRST") != NULL;      /* first-call work by part, on stderr */
    double t0 = trace ? wall_ms() : 0.0, t1;
    ezh_grid *gi = &G[s->gdin], *go = &G[s->gdout];
    size_t n = (size_t)go->ni * go->nj;
    int rotated = gi->grtyp == 'E' || (gi->grtyp == 'Z' && gi->grref == 'E');
    int host_locate = (rotated && (getenv("EZHIP_HOST_LOCATE") || !libm_exact_matches_host())) || (gi->grtyp == 'G' && gi->ig[0] != 0) || gi->grtyp == '!';      /* '!': the REAL libm chain of ez_lambxyfll99 lives in the host code */      /* hemispheric 'G': the reference's search quirk lives in the host code only */
    float *hx = NULL, *hy = NULL;
    if (ensure_grid_dev(gi)) return -1;
    if (trace) { ezhip_sync(); t1 = wall_ms(); fprintf(stderr, "ensure_points: source tables on the device %.2f ms\n", t1 - t0); t0 = t1; }
    float *dx = (float *)ezhip_malloc(sizeof(float) * n), *dy = (float *)ezhip_malloc(sizeof(float) * n);
    if (!dx || !dy) { ezhip_free(dx); ezhip_free(dy); return -1; }
    if (trace) { t1 = wall_ms(); fprintf(stderr, "ensure_points: 2 x %zu bytes allocated %.2f ms\n", sizeof(float) * n, t1 - t0); t0 = t1; }
    int rc = 0;
    if (host_locate) {
        ensure_coords(go);
        hx = (float *)malloc(sizeof(float) * n); hy = (float *)malloc(sizeof(float) * n);
        float *lat = (float *)malloc(sizeof(float) * n), *lon = (float *)malloc(sizeof(float) * n);
        if (!hx || !hy || !lat || !lon) rc = -1;
        else {
            if (go->separable) rc = h_locate_mt2(gi, hx, hy, lat, lon, (int)n, go->lat1d, go->lon1d, go->ni);
            else { memcpy(lat, go->lat2d, sizeof(float) * n); memcpy(lon, go->lon2d, sizeof(float) * n); rc = h_locate_mt(gi, hx, hy, lat, lon, (int)n); }
            if (!rc) rc = ezhip_h2d(dx, hx, sizeof(float) * n) || ezhip_h2d(dy, hy, sizeof(float) * n) || ezhip_sync();
        }
        free(lat); free(lon);
    } else {
        if (ensure_coords_dev(go)) rc = -1;
        else {
            if (trace) { ezhip_sync(); t1 = wall_ms(); fprintf(stderr, "ensure_points: target coordinates on the device %.2f ms\n", t1 - t0); t0 = t1; }
            ezhip_locate_plan lp;
            fill_locate_plan(gi, &lp);
            rc = ezhip_locate(&lp, dx, dy, go->d_lat, go->d_lon, go->ni, go->nj, go->separable);
        }
    }
    if (trace) { ezhip_sync(); t1 = wall_ms(); fprintf(stderr, "ensure_points: located (%s) %.2f ms\n", host_locate ? "host" : "device", t1 - t0); t0 = t1; }
    if (!rc && s->extrap && !s->sep_capable) {
        s->have_dehors = 0;
        if (hx) {
            for (size_t k = 0; k < n; k++) {
                int ix = (int)((double)hx[k] + 0.5), iy = (int)((double)hy[k] + 0.5);
                if (ix < 1 || iy < 1 || ix > gi->ni || iy > gi->nj) { s->have_dehors = 1; break; }
            }
        } else {                                            /* the points never leave the device: one flag comes back */
            int *d_flag = (int *)ezhip_malloc(sizeof(int)), flag = 0;
            if (!d_flag || ezhip_any_dehors(dx, dy, n, gi->ni, gi->nj, d_flag) || ezhip_d2h(&flag, d_flag, sizeof(int)) || ezhip_sync()) rc = -1;
            ezhip_free(d_flag);
            s->have_dehors = flag;
        }
    }
    free(hx); free(hy);
    if (trace) { t1 = wall_ms(); fprintf(stderr, "ensure_points: outside-the-source test %.2f ms\n", t1 - t0); }
    if (rc) { ezhip_free(dx); ezhip_free(dy); return -1; }
    /* a target with 2-D coordinates: the reference's locate has edited its cached longitudes as it went (a separable target's 1-D ones: analyse_set) */
    if (!go->separable && go->lon2d && h_lon_edits(gi, go->lon2d, n)) go->d_coords_valid = 0;
    s->d_y = dy; s->d_x = dx;
    return 0;
}

static void fill_pts_plan(const ezh_set *s, const ezh_grid *gi, ezhip_pts_plan *pp, int degree, int zones, int vector_mode)
{
    memset(pp, 0, sizeof(*pp));
    pp->degree = degree; pp->irregular = src_irregular(gi);
    pp->ni = gi->ni; pp->nj = gi->nj; pp->i1 = gi->i1; pp->i2 = gi->i2; pp->j1 = gi->j1; pp->j2 = gi->j2; pp->wrap = gi->extension;
    pp->ax = gi->d_ax; pp->ay = gi->d_ay; pp->ncx = gi->d_ncx; pp->ncy = gi->d_ncy;
    pp->ncx8 = gi->d_ncx8; pp->ncy8 = gi->d_ncy8; pp->xrec10 = gi->d_xrec10; pp->yrec10 = gi->d_yrec10; pp->xrec8 = gi->d_xrec8; pp->yrec8 = gi->d_yrec8;
    pp->zones = zones; pp->degre_extrap = O.degre_extrap; pp->vector_mode = vector_mode;
    pp->pole_weighted = (gi->grtyp == 'Z' && gi->grref == 'E');
    if (s) { pp->ypole_n = s->ypole_n; pp->ypole_s = s->ypole_s; }
    if (gi->ay && gi->nj >= 4) {
        int j1 = gi->j1, j2 = gi->j2;
        const float *ay = gi->ay;
        if (ay[j2 - 1] == 90.0) { pp->ay4_n[0] = ay[j2 - 4]; pp->ay4_n[1] = ay[j2 - 3]; pp->ay4_n[2] = ay[j2 - 2]; pp->ay4_n[3] = ay[j2 - 1]; }
        else { pp->ay4_n[0] = ay[j2 - 3]; pp->ay4_n[1] = ay[j2 - 2]; pp->ay4_n[2] = ay[j2 - 1]; pp->ay4_n[3] = 90.0f; }
        /* northern hemispheric 'G' (j1 = 1 - nj): the reference reads ay[j1 - 1] in FRONT of the table (ez_corrval_ausud.c); entry 0 here */
        if (ay[j1 - 1 < 0 ? 0 : j1 - 1] == -90.0) { pp->ay4_s[0] = ay[0]; pp->ay4_s[1] = ay[1]; pp->ay4_s[2] = ay[2]; pp->ay4_s[3] = ay[3]; }
        else { pp->ay4_s[0] = -90.0f; pp->ay4_s[1] = ay[0]; pp->ay4_s[2] = ay[1]; pp->ay4_s[3] = ay[2]; }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* scalar interpolation on device pointers                                                       */
/* ------------------------------------------------------------------------------------------ */
static ezh_set *current_set(const char *who)
{
    if (cur_gdin < 0 || cur_gdout < 0 || !grid_ok(cur_gdin) || !grid_ok(cur_gdout)) {
        fprintf(stderr, "<%s> Source or target grid undefined! Aborting...\n", who);
        return NULL;
    }
    pthread_mutex_lock(&g_mtx);
    ezh_set *s = find_set(cur_gdout, cur_gdin, 1);
    if (!s->x1d && !s->d_x && G[cur_gdin].grtyp != 'U' && G[cur_gdout].grtyp != 'U') analyse_set(s);
    pthread_mutex_unlock(&g_mtx);
    return s;
}

/* fill value + pole values of the field being interpolated: per host THREAD (two threads may interpolate different
 * fields on the same grid set at the same time, each on its own stream) */
static __thread float *t_scratch8 = NULL;
static int ensure_scratch(ezh_set *s)
{
    (void)s;
    if (!t_scratch8) t_scratch8 = (float *)ezhip_malloc(sizeof(float) * 8);
    return t_scratch8 ? 0 : -1;
}
/* lazily built per-set state (plans, located points, device mirrors) is built under this lock */

/* in-kernel pole values of k_sepx: per host thread a ring of 4 {values, flags} buffers (consecutive launches of one
 * thread are stream-ordered; the ring only guards a thread that switches streams between calls) and a launch epoch */
static __thread struct { unsigned long long *gran; int cap; unsigned epoch; } t_pole;
static int pole_ring(int nfields, ezhip_sep_plan *p)
{
    int nb = ((2 * nfields + 7) / 8) * 8;
    if (t_pole.cap < nb) {
        ezhip_sync();
        ezhip_free(t_pole.gran);
        size_t bytes = (size_t)4 * nb * sizeof(unsigned long long);
        t_pole.gran = (unsigned long long *)ezhip_malloc(bytes);
        if (!t_pole.gran) { t_pole.cap = 0; return -1; }
        if (ezhip_memset(t_pole.gran, 0, bytes) || ezhip_sync()) return -1;
        t_pole.cap = nb; t_pole.epoch = 0;
    }
    t_pole.epoch++;
    if (t_pole.epoch == 0) t_pole.epoch = 1;
    int slot = (int)(t_pole.epoch & 3);
    p->pole_blocks = nb; p->pole_epoch = t_pole.epoch;
    p->pole_gran = t_pole.gran + (size_t)slot * t_pole.cap;
    return 0;
}

/* one field, device pointers; vector_mode: 0 scalar, 1 = u/v component with synthetic pole rows
 * (prow_n / prow_s device pointers, may be NULL when polar correction is off) */
/* per host thread: the expansion of a hemispheric source field (ez_xpnsrcgd) */
static __thread struct { float *p; size_t cap; } t_xpn;
static __thread int t_symmetrie = 1;                     /* groptions.symmetrie: 1 SYM, 0 ANTISYM (the v component of c_gdxyvval: ez_xpngdag2.inc:37-41 mirrors it with its sign changed) */
static const float *hemi_expand(const ezh_grid *gi, const float *d_zin)
{
    size_t n = (size_t)gi->ni * (size_t)(gi->j2 - gi->j1 + 1);
    if (t_xpn.cap < n) { if (t_xpn.p) { ezhip_sync(); ezhip_free(t_xpn.p); } t_xpn.p = (float *)ezhip_malloc(sizeof(float) * n); t_xpn.cap = t_xpn.p ? n : 0; }
    if (!t_xpn.p) return NULL;
    if (ezhip_hemi_expand(t_xpn.p, d_zin, gi->ni, gi->nj, gi->j1, gi->j2, src_hemi(gi) ? gi->ig[0] : 0, gi->grtyp == 'B', t_symmetrie, grid_yinv(gi))) return NULL;
    return t_xpn.p;
}
/* Arrays the caller has page-locked for the library (ezhip_register_host_buffer): copies to / from them are asynchronous, which is what
 * lets c_ezsint overlap the upload of the source with the download of the result.  From pageable memory hipMemcpyAsync returns only when
 * the copy is done (measured: 38.7 MB up 0.70 ms, 103.7 MB down 1.87 ms, both at the PCIe rate, strictly one after the other). */
#define EZH_MAX_PINNED 64
static struct { const char *p; size_t n; } g_pinned[EZH_MAX_PINNED];
static pthread_mutex_t g_pin_mtx = PTHREAD_MUTEX_INITIALIZER;
int32_t ezhip_register_host_buffer(void *p, size_t nbytes)
{
    if (!p || !nbytes) return -1;
    if (need_device("ezhip_register_host_buffer")) return -1;
    pthread_mutex_lock(&g_pin_mtx);
    int k = 0;
    while (k < EZH_MAX_PINNED && g_pinned[k].p) k++;
    int rc = -1;
    if (k < EZH_MAX_PINNED && !ezhip_host_pin(p, nbytes)) { g_pinned[k].p = (const char *)p; g_pinned[k].n = nbytes; rc = 0; }
    pthread_mutex_unlock(&g_pin_mtx);
    if (rc) fprintf(stderr, "<ezhip_register_host_buffer> cannot page-lock %zu bytes: %s\n", nbytes, k < EZH_MAX_PINNED ? ezhip_last_error() : "table full");
    return rc;
}
int32_t ezhip_unregister_host_buffer(void *p)
{
    int rc = -1;
    pthread_mutex_lock(&g_pin_mtx);
    for (int k = 0; k < EZH_MAX_PINNED; k++)
        if (g_pinned[k].p == (const char *)p) { ezhip_sync(); rc = ezhip_host_unpin(p) ? -1 : 0; g_pinned[k].p = NULL; g_pinned[k].n = 0; break; }
    pthread_mutex_unlock(&g_pin_mtx);
    return rc;
}
int ezh_host_is_pinned(const void *p, size_t n);
static int host_pinned(const void *p, size_t nbytes)
{
    int yes = 0;
    pthread_mutex_lock(&g_pin_mtx);
    for (int k = 0; k < EZH_MAX_PINNED && !yes; k++)
        if (g_pinned[k].p && (const char *)p >= g_pinned[k].p && (const char *)p + nbytes <= g_pinned[k].p + g_pinned[k].n) yes = 1;
    pthread_mutex_unlock(&g_pin_mtx);
    return yes;
}
int ezh_host_is_pinned(const void *p, size_t n) { return host_pinned(p, n); }

/* Host-pointer ABI (c_ezsint on host arrays): the caller's arrays of the field in flight.  state 0: none; 1: the source is still on the
 * host only (whoever needs all of it on the device first calls hio_full); 2: source uploaded, result still to be fetched by the caller;
 * 3: result already in the caller's array (sepx_host_chunks copied it out row range by row range). */
static __thread struct { const float *zin; float *zout; size_t nin; int state, pinned; } t_hio;
static int hio_full(float *d_zin)
{
    if (t_hio.state != 1) return 0;
    t_hio.state = 2;
    return ezhip_h2d(d_zin, t_hio.zin, sizeof(float) * t_hio.nin);
}

/* One field through k_sepx in row ranges: the source rows a range needs go up, the range is launched, and its result rows come back on
 * the side stream while the next range uploads and computes -- PCIe carries both directions at once, so the call costs about the larger
 * copy (D2H, 104 MB at cfg2) instead of the sum.  The last range carries the special (polar / outside) rows and the pole sums, which may
 * need any source row. */
/* ordinary (pageable) arrays: an upload from them blocks the calling thread for its whole duration, and the download goes through the page-locked
 * bounce buffers on the calling thread too -- so a second thread brings the source rows up, range by range, while the caller fetches the finished
 * rows: the call costs about the download (1.9 ms at cfg2) instead of upload + download (2.6) */
typedef struct { const float *src; float *d_dst; size_t row_floats; const int *need; int nchunk; int rows_up, fail, cancel; pthread_mutex_t m; pthread_cond_t cv; } hio_uploader;
static void *hio_uploader_main(void *arg)
{
    hio_uploader *u = (hio_uploader *)arg;
    int done = 0;
    for (int k = 0; k < u->nchunk; k++) {
        pthread_mutex_lock(&u->m); const int cancel = u->cancel; pthread_mutex_unlock(&u->m);
        if (cancel) break;
        int bad = 0;
        if (u->need[k] > done) {
            bad = ezhip_h2d_blocking_own_stream(u->d_dst + (size_t)done * u->row_floats, u->src + (size_t)done * u->row_floats, sizeof(float) * (size_t)(u->need[k] - done) * u->row_floats);
            done = u->need[k];
        }
        pthread_mutex_lock(&u->m);
        if (bad) u->fail = 1; else u->rows_up = done;
        pthread_cond_broadcast(&u->cv);
        pthread_mutex_unlock(&u->m);
        if (bad) break;
    }
    ezhip_own_stream_release();
    return NULL;
}
static int sepx_host_chunks(const ezh_sepplan *sp, ezhip_sep_plan p, float *d_zout, float *d_zin)
{
    const int nchunk_env = getenv("EZHIP_HOST_CHUNKS") ? atoi(getenv("EZHIP_HOST_CHUNKS")) : 6;
    const int nseg = p.x_nseg, nid = p.ni_dst, njd = p.nj_dst, nis = p.ni_src, njs = p.nj_src, rows_per_vb = p.x_rows_per_step;
    int nchunk = nchunk_env < 1 ? 1 : (nchunk_env > nseg ? nseg : nchunk_env);
    unsigned char *copied = (unsigned char *)calloc((size_t)njd, 1);
    int *need_of = (int *)malloc(sizeof(int) * (size_t)nchunk);
    if (!copied || !need_of) { free(copied); free(need_of); return -1; }
    int src_done = 0, rc = -1;
    const int pole_blocks = p.pole_blocks;
    for (int k = 0; k < nchunk; k++) {       /* source rows the first k + 1 ranges need on the device */
        const int s_lo = (int)((long long)nseg * k / nchunk), s_hi = (int)((long long)nseg * (k + 1) / nchunk);
        const int v_lo = s_lo * p.x_rb, v_hi = imin(s_hi * p.x_rb, sp->h_nvb);
        int need = k ? need_of[k - 1] : 0;
        for (int i = v_lo; i < v_hi; i++) if (sp->h_vb_send[i] > need) need = sp->h_vb_send[i];
        if (k == nchunk - 1 || need > njs) need = njs;
        need_of[k] = need;
    }
    hio_uploader up;
    pthread_t up_th;
    int up_on = 0;
    if (!t_hio.pinned) {
        memset(&up, 0, sizeof(up));
        up.src = t_hio.zin; up.d_dst = d_zin; up.row_floats = (size_t)nis; up.need = need_of; up.nchunk = nchunk;
        pthread_mutex_init(&up.m, NULL); pthread_cond_init(&up.cv, NULL);
        if (pthread_create(&up_th, NULL, hio_uploader_main, &up) != 0) { pthread_mutex_destroy(&up.m); pthread_cond_destroy(&up.cv); goto out; }
        up_on = 1;
    }
    for (int k = 0; k < nchunk; k++) {
        const int s_lo = (int)((long long)nseg * k / nchunk), s_hi = (int)((long long)nseg * (k + 1) / nchunk), last = k == nchunk - 1;
        const int v_lo = s_lo * p.x_rb, v_hi = imin(s_hi * p.x_rb, sp->h_nvb);
        const int need = need_of[k];
        if (up_on) {
            pthread_mutex_lock(&up.m);
            while (up.rows_up < need && !up.fail) pthread_cond_wait(&up.cv, &up.m);
            const int failed = up.fail;
            pthread_mutex_unlock(&up.m);
            if (failed) goto out;
        } else if (need > src_done) {
            if (ezhip_h2d(d_zin + (size_t)src_done * nis, t_hio.zin + (size_t)src_done * nis, sizeof(float) * (size_t)(need - src_done) * nis)) goto out;
            src_done = need;
        }
        ezhip_sep_plan q = p;
        q.special_last = 1; q.by_lo = s_lo; q.by_cnt = (s_hi - s_lo) + (last ? p.n_special : 0);
        q.pole_blocks = last ? pole_blocks : 0;
        if (ezhip_interp_sep(&q, d_zout, d_zin)) { dev_fail("the separable interpolation kernel (row range)"); goto out; }
        /* result rows of this range: the main rows of its row-blocks; after the last range everything not yet fetched */
        if (!last) for (int i = v_lo; i < v_hi; i++) {
            const int r0 = sp->h_vb_by[i] * rows_per_vb, r1 = imin(r0 + rows_per_vb, njd);
            for (int r = r0; r < r1; r++) if (!sp->h_rflag[r]) copied[r] = 2;         /* 2: fetch now */
        }
        if (ezhip_side_begin()) goto out;
        int bad = 0;
        for (int r = 0; r < njd && !bad; ) {
            const int want = last ? copied[r] == 0 : copied[r] == 2;
            if (!want) { r++; continue; }
            int e = r;
            while (e < njd && (last ? copied[e] == 0 : copied[e] == 2)) copied[e++] = 1;
            bad = ezhip_d2h(t_hio.zout + (size_t)r * nid, d_zout + (size_t)r * nid, sizeof(float) * (size_t)(e - r) * nid);
            r = e;
        }
        if (ezhip_side_end() || bad) goto out;
    }
    if (ezhip_side_join()) goto out;
    t_hio.state = 3;
    rc = 0;
out:
    if (up_on) {
        pthread_mutex_lock(&up.m); up.cancel = 1; pthread_mutex_unlock(&up.m);
        pthread_join(up_th, NULL);
        if (up.fail) rc = -1;
        pthread_mutex_destroy(&up.m); pthread_cond_destroy(&up.cv);
    }
    free(copied); free(need_of);
    return rc;
}

static int run_field_p(ezh_set *s, float *d_zout, const float *d_zin, int vector_mode, const float *prow_n, const float *prow_s, const float *d_poles_pre);
static int run_field(ezh_set *s, float *d_zout, const float *d_zin, int vector_mode, const float *prow_n, const float *prow_s)
{
    return run_field_p(s, d_zout, d_zin, vector_mode, prow_n, prow_s, NULL);
}

/* interp_degree = "average" (degre_interp 4; gdinterp.c:181-187 -> ez_avg.inc): area-weighted means of the source cells under every target
 * cell, then the parts of the polar correction that are defined for it: the points outside a regional source (fill value or re-interpolation
 * with extrap_degree) and the pole values.  The strip routines have no case for this degree (ez_corrval_aunord.c:52-110) and scatter an
 * uninitialised array over their zones: a set that has strip points which are not pole points is refused (set polar_correction = no). */
static int gdllfxy_one(const ezh_grid *g, float *lat, float *lon, const float *x, const float *y, int n, int newform);
static int run_average(ezh_set *s, float *d_zout, const float *d_zin, int sph)
{
    ezh_grid *gi = &G[s->gdin], *go = &G[s->gdout];
    const int polar = O.polar_correction == 1;
    if (src_hemi(gi) || grid_yinv(gi) || grid_yinv(go)) { fprintf(stderr, "<ezsint> interp_degree = average: hemispheric / y-inverted grids are outside the MI355X hot-path scope\n"); return -1; }
    if (!s->sep_capable || !s->x1d || go->ni < 2 || go->nj < 2) {
        fprintf(stderr, "<ezsint> interp_degree = average needs a target whose rows / columns map to rows / columns of the source (ez_avg reads the first row and column of x, y only)\n");
        return -1;
    }
    const int nid = go->ni, njd = go->nj, nis = gi->ni, njs = gi->nj;
    if (sph) {
        ensure_coords(go);
        for (int j = 0; j < njd; j++) if (go->separable && fabsf(go->lat1d[j]) > 89.0f) {
            fprintf(stderr, "<ezsint> interp_degree = sph_average: target row %d lies at a pole, where the cells are widened by 1 / cos(latitude) without bound (ez_avg_sph.inc:63-93 then indexes far outside the source)\n", j + 1);
            return -1;
        }
    }
    if (polar && !s->extrap)
        for (int j = 0; j < njd; j++) {
            const float py = s->y1d[j];
            const int strip = (int)py > (gi->j2 - 2) || (int)py < (gi->j1 + 1);
            const int pole = fabs((double)(py - s->ypole_n)) < 1.0e-3 || fabs((double)(py - s->ypole_s)) < 1.0e-3;
            if (strip && !pole) {
                fprintf(stderr, "<ezsint> interp_degree = average with polar_correction = yes: target row %d lies in a polar strip, where the reference writes an uninitialised array (ez_corrval_aunord.c:52-114); set polar_correction to no\n", j + 1);
                return -1;
            }
        }
    pthread_mutex_lock(&g_plan_mtx);
    int brc = 0;
    if (!s->d_avg[sph]) {
        float *b = (float *)malloc(sizeof(float) * ((size_t)nid + 3 * (size_t)njd));
        float *x = b, *amp = b + nid, *yl = b + nid + njd, *yh = b + nid + 2 * njd;
        const float *y = s->y1d;
        memcpy(x, s->x1d, sizeof(float) * nid);
        if (x[0] > (float)(nis - 1)) x[0] = 1.0f;                                   /* ez_avg.inc:57 */
        yl[0] = fmaxf(1.0f, y[0] - 0.5f * (y[1] - y[0]));
        for (int j = 1; j < njd; j++) yl[j] = y[j] - 0.5f * (y[j] - y[j - 1]);
        yh[njd - 1] = fminf(1.0f * (float)njs, y[njd - 1] + 0.5f * (y[njd - 1] - y[njd - 2]));
        for (int j = 0; j < njd - 1; j++) yh[j] = y[j] + 0.5f * (y[j + 1] - y[j]);
        for (int j = 0; j < njd; j++) amp[j] = 1.0f;
        if (sph) {
            /* gdinterp.c:189-198: the latitude of every target row at its first column (c_gdllfxy_orig), then 1 / cos of it in REAL with
             * the host's libm (ez_avg_sph.inc:63-65) -- the one transcendental of the method, kept off the device so that the cell bounds
             * are the reference's bits */
            const float degre_a_radian = 0.017453295199f;
            for (int j = 0; j < njd && !brc; j++) {
                float la = 0.0f, lo = 0.0f, xo = 1.0f, yo = 1.0f * (float)(j + 1);
                if (gdllfxy_one(go, &la, &lo, &xo, &yo, 1, 0) < 0) brc = -1;
                amp[j] = 1.0f / cosf(la * degre_a_radian);
            }
        }
        if (!brc) { s->d_avg[sph] = (float *)upload(b, sizeof(float) * ((size_t)nid + 3 * (size_t)njd)); if (!s->d_avg[sph] || ezhip_sync()) brc = -1; }
        free(b);
    }
    pthread_mutex_unlock(&g_plan_mtx);
    if (brc) return -1;
    if (ezhip_average(d_zout, d_zin, s->d_avg[sph], nid, njd, nis, njs, gi->extension, s->y1d[njd - 1])) return dev_fail("the averaging kernel");
    if (!polar) return 0;
    /* what is left of ez_corrval: outside points / pole values through the per-point kernel, whose normal points keep the averages */
    int ierc = 0;
    if (ensure_scratch(s)) return -1;
    float *d_fill = t_scratch8, *d_poles = t_scratch8 + 4;
    const int zones = s->extrap ? 2 : 1;
    if (s->extrap) {
        if (!s->have_dehors) return 0;
        ierc = 2;
        if (O.degre_extrap == XT_ABORT) {
            fprintf(stderr, "<ez_corrval> There are points on the destination grid that lie outside the source grid\n");
            fprintf(stderr, "<ez_corrval> aborting at your request!\n\n\n");
            return -1;
        }
        if (O.degre_extrap >= XT_MAXIMUM && ezhip_fill_value(d_fill, d_zin, (size_t)nis * njs, O.degre_extrap, O.valeur_extrap, 0)) return -1;
    }
    ezhip_pts_plan pp;
    pthread_mutex_lock(&g_plan_mtx);
    int erc = ensure_points(s);
    if (!erc) fill_pts_plan(s, gi, &pp, DEG_LINEAR, zones, 0);
    pthread_mutex_unlock(&g_plan_mtx);
    if (erc) return -1;
    pp.only_special = 1; pp.fill = d_fill; pp.polevals = d_poles;
    if (zones == 1 && ezhip_polevals(d_poles, d_zin, nis, njs, pp.pole_weighted, gi->d_ax)) return -1;
    if (ezhip_interp_pts(&pp, d_zout, d_zin, s->d_x, s->d_y, nid * njd)) return dev_fail("the per-point kernel (outside / pole points of an averaged field)");
    return ierc;
}

/* d_poles_pre: device float[2] pole values of this field already computed (batch entry point), or NULL */
/* ez_corrval.c:146-148: a Z- or #-on-E source and a 'B' target end the SCALAR polar correction with ez_corrbgd (the target's pole rows -> their mean);
 * in vector mode ez_corrval returns before it (:117-118), so the components of a wind never see it */
static int corrbgd_applies(const ezh_grid *gi, const ezh_grid *go)
{
    return O.polar_correction == 1 && (gi->grtyp == 'Z' || gi->grtyp == '#') && gi->grref == 'E' && go->grtyp == 'B';
}
static int run_field_q(ezh_set *s, float *d_zout, const float *d_zin, int vector_mode, const float *prow_n, const float *prow_s, const float *d_poles_pre);
static int run_field_p(ezh_set *s, float *d_zout, const float *d_zin, int vector_mode, const float *prow_n, const float *prow_s, const float *d_poles_pre)
{
    int rc = run_field_q(s, d_zout, d_zin, vector_mode, prow_n, prow_s, d_poles_pre);
    const ezh_grid *gi = &G[s->gdin], *go = &G[s->gdout];
    if (rc >= 0 && !vector_mode && corrbgd_applies(gi, go) && ezhip_corrbgd(d_zout, go->ni, go->nj, go->ig[0])) return -1;
    return rc;
}
static int run_field_q(ezh_set *s, float *d_zout, const float *d_zin, int vector_mode, const float *prow_n, const float *prow_s, const float *d_poles_pre)
{
    ezh_grid *gi = &G[s->gdin], *go = &G[s->gdout];
    int degree = O.degre_interp, polar = O.polar_correction == 1;
    if ((degree == 4 || degree == 5) && !vector_mode) { if (hio_full((float *)d_zin)) return -1; return run_average(s, d_zout, d_zin, degree == 5); }
    if (degree != DEG_NEAREST && degree != DEG_LINEAR && degree != DEG_CUBIC) {
        fprintf(stderr, "<ezsint> interp_degree %d (%s) is outside the MI355X hot-path scope\n", degree, degree == 4 || degree == 5 ? "an averaging degree, for winds" : "?");
        return -1;
    }
    int ierc = 0;
    if (ensure_scratch(s)) return -1;
    float *d_fill = t_scratch8, *d_poles = t_scratch8 + 4;
    if ((grid_yinv(gi) || src_hemi(gi) || vector_mode || d_poles_pre) && hio_full((float *)d_zin)) return -1;
    if (grid_yinv(gi)) {
        if (vector_mode) { fprintf(stderr, "<c_ezuvint> y-inverted source grids are outside the MI355X hot-path scope for winds\n"); return -1; }
        if (!src_hemi(gi)) {                               /* ezsint.c:100-106: the rows in reverse order (PERMUT); a hemisphere is flipped while it is expanded */
            d_zin = hemi_expand(gi, d_zin);
            if (!d_zin) return -1;
            d_poles_pre = NULL;
        }
    }
    pthread_mutex_lock(&g_plan_mtx);
    int mode = choose_mode(s, degree, polar);          /* may analyse the set (first use) */
    pthread_mutex_unlock(&g_plan_mtx);
    int need_fill = 0;
    if (mode != 1) {                                    /* per-point path: x,y (and with them the DEHORS classification) first */
        pthread_mutex_lock(&g_plan_mtx);
        int erc0 = ensure_points(s);
        pthread_mutex_unlock(&g_plan_mtx);
        if (erc0) return -1;
    }
    if (polar && s->extrap && s->have_dehors) {         /* ez_corrval.c:54-60 */
        ierc = 2;
        if (O.degre_extrap == XT_ABORT) {
            fprintf(stderr, "<ez_corrval> There are points on the destination grid that lie outside the source grid\n");
            fprintf(stderr, "<ez_corrval> aborting at your request!\n\n\n");
            return -1;
        }
        need_fill = (O.degre_extrap >= XT_MAXIMUM) || vector_mode;
    }
    if ((need_fill || mode != 1) && hio_full((float *)d_zin)) return -1;
    if (need_fill && ezhip_fill_value(d_fill, d_zin, (size_t)gi->ni * gi->nj, O.degre_extrap, O.valeur_extrap, vector_mode)) return dev_fail("the extrapolation-value reduction");
    if (mode == 1) {
        pthread_mutex_lock(&g_plan_mtx);
        int brc = build_sep_plan(s, degree, vector_mode, polar);
        int di = degree == DEG_CUBIC ? 2 : degree;
        ezhip_sep_plan p = s->sep[di][vector_mode].p;
        if (!brc && p.pole_weighted) brc = ensure_grid_dev(gi);
        pthread_mutex_unlock(&g_plan_mtx);
        if (brc) return -1;
        p.fill = d_fill;
        if (p.pole_weighted) p.ax = gi->d_ax;
        p.pole_row_n = prow_n; p.pole_row_s = prow_s;
        p.polevals = d_poles_pre ? d_poles_pre : d_poles;
        if (getenv("EZHIP_NO_SEPX")) p.x_nseg = 0;       /* tests: force the fallback tile kernel k_sep */
        /* row ranges pay only when the copies are asynchronous, i.e. between page-locked arrays (EZHIP_HOST_CHUNKS: their number), and are
         * never used on ordinary memory: many short device writes into one pageable array make the runtime lock / unlock overlapping page ranges
         * in quick succession, and the gpu test suite died there once in ten runs ("Write access to a read-only page") */
        const int chunked = t_hio.state == 1 && p.x_nseg > 1 && s->sep[di][vector_mode].h_nvb > 0 && !getenv("EZHIP_HOST_NO_CHUNKS") &&
                            (t_hio.pinned || getenv("EZHIP_HOST_UPLOADER"));
        if (!chunked && hio_full((float *)d_zin)) return -1;
        if (p.x_nseg > 0) {
            /* a lone field: the special rows go last in the work order (mid-order they hold slots while the pole
             * producers of the same launch run) */
            p.special_last = 1;
            /* the plan's row-blocks per thread block make the main blocks of ONE field just fill the device (strips x segments <= co-resident thread
             * blocks): kept for a lone field.  (With three thread blocks per CU two thirds of it were faster: 44.3 -> 39.7 us at cfg2; with four -- 39.7 KB
             * of LDS -- one full round wins: 29.3 - 30.7 us against 30.7 - 32.8 with five row-blocks and 32.2 - 32.4 with six or eight, tools/sweep_lone.py
             * interleaved on one device.  EZHIP_SINGLE_RB overrides) */
            if (getenv("EZHIP_SINGLE_RB")) { int rb = atoi(getenv("EZHIP_SINGLE_RB")); if (rb >= 1) { p.x_rb = rb; p.x_nseg = (p.x_nvb + rb - 1) / rb; } }
            if (getenv("EZHIP_SPECIAL_PCT")) { int pct = atoi(getenv("EZHIP_SPECIAL_PCT")); if (pct >= 0 && pct <= 100) p.special_last = 2 + p.x_nseg * pct / 100; }     /* development: special rows at pct % of the work order */
        }
        if (p.need_poles && !d_poles_pre) {
            /* k_sepx sums the pole rows itself (producer blocks at the head of the launch): 43 us per lone cfg2 field
             * with the halved row-blocks per thread block above, against 57 us with a k_polevals launch (29 us, serial)
             * in front; the fallback kernel needs the separate launch */
            if (p.x_nseg > 0 && (chunked || !getenv("EZHIP_POLES_PRELAUNCH"))) { if (pole_ring(1, &p)) return -1; }
            else if (ezhip_polevals(d_poles, d_zin, gi->ni, gi->nj, p.pole_weighted, p.ax)) return -1;
        }
        if (chunked) {
            pthread_mutex_lock(&g_plan_mtx);          /* the host geometry belongs to the set's plan: hold it while it is read */
            int crc = (s->sep[di][vector_mode].built == 1 + polar) ? sepx_host_chunks(&s->sep[di][vector_mode], p, d_zout, (float *)d_zin) : -1;
            pthread_mutex_unlock(&g_plan_mtx);
            return crc ? -1 : ierc;
        }
        if (vector_mode && ezhip_side_join()) return -1;      /* the special rows of this launch read the polar wind rows */
        if (ezhip_interp_sep(&p, d_zout, d_zin)) return dev_fail("the separable interpolation kernel");
        return ierc;
    }
    ezhip_pts_plan pp;
    int zones = !polar ? 0 : (s->extrap ? 2 : 1);
    pthread_mutex_lock(&g_plan_mtx);
    int erc = ensure_points(s);
    if (!erc) fill_pts_plan(s, gi, &pp, degree, zones, vector_mode);
    pthread_mutex_unlock(&g_plan_mtx);
    if (erc) return -1;
    pp.fill = d_fill; pp.polevals = d_poles_pre ? d_poles_pre : d_poles;
    pp.pole_row_n = prow_n; pp.pole_row_s = prow_s;
    int nj_rows = gi->nj;
    if (src_hemi(gi)) {                                     /* ezsint.c:108-113: the field mirrored into the other hemisphere, rows j1..j2 */
        if (vector_mode) { fprintf(stderr, "<c_ezuvint> hemispheric source grids are outside the MI355X hot-path scope for winds\n"); return -1; }
        d_zin = hemi_expand(gi, d_zin);
        if (!d_zin) return -1;
        nj_rows = gi->j2 - gi->j1 + 1;
        d_poles_pre = NULL; pp.polevals = d_poles;          /* pole values of the EXPANDED field */
    }
    /* the field's pole values: two producer blocks of the k_pts launch (EZHIP_PTS_POLEVALS_FIRST=1: the separate launch in front, as before) */
    if (zones == 1 && !vector_mode && !d_poles_pre) {
        if (getenv("EZHIP_PTS_POLEVALS_FIRST")) { if (ezhip_polevals(d_poles, d_zin, gi->ni, nj_rows, pp.pole_weighted, gi->d_ax)) return -1; }
        else { pp.pv_out = d_poles; pp.pv_nj = nj_rows; }
    }
    /* bicubic from an irregular source with wrap (a rotated global grid), the whole target grid in row order: from the second call of a set on the field runs from
     * LDS-staged stencil windows (k_st: bit-identical to k_pts; the tile table and the tile-ordered x, y are built once per set and zone options, behind the
     * first call).  EZHIP_NO_ST=1: k_pts every time */
    const int st_ok = ((degree == DEG_CUBIC && (pp.wrap != 0 || (pp.i1 == 1 && pp.i2 == pp.ni)) && pp.xrec10) || degree == DEG_LINEAR) && pp.irregular && !pp.out_idx && (!vector_mode || !getenv("EZHIP_ST_NO_VECTOR")) && !src_hemi(gi) &&
                      (size_t)go->ni * go->nj >= (getenv("EZHIP_ST_MIN_POINTS") ? (size_t)atol(getenv("EZHIP_ST_MIN_POINTS")) : (size_t)262144) && !getenv("EZHIP_NO_ST");
    const int st_key = 1 | zones << 1 | (O.degre_extrap & 0xFF) << 4 | degree << 12 | (vector_mode ? 1 << 20 : 0), sl = degree == DEG_CUBIC;      /* (one table per set and degree: the first zone options that ask; vector mode, round 6: the component passes of the exact-winds mode -- their zone rules differ) */
    int st_build = 0;
    if (st_ok) {
        pp.tile_ni = go->ni; pp.tile_nj = go->nj;
        pthread_mutex_lock(&g_plan_mtx);
        if (s->st_key[sl] == st_key && s->d_st_tiles[sl]) {
            pp.uvt_tiles = s->d_st_tiles[sl]; pp.uvt_shape = 3232; pp.uvt_cap = s->st_cap[sl]; pp.uvt_streams = s->d_st_streams[sl];
            pp.uvt_nhb = (sl == 1 && !getenv("EZHIP_UVT_TABLE_ORDER")) ? s->st_stats[sl][1] + s->st_stats[sl][2] : 0;      /* (bicubic: k_st; the bilinear k_st1 keeps table order) */
            if (sl == 1 && s->sspec_key == st_key && s->sspec_count > 0) {      /* the set's special points under these options: the kernel's producer blocks take them along */
                pp.cspec_valid = 1; pp.cspec_count = s->sspec_count; pp.cspec_list = s->d_sspec_list; pp.cspec_x = s->d_sspec_xy; pp.cspec_y = s->d_sspec_xy + s->sspec_count;
            }
        }
        else if (s->st_key[sl] == 0) st_build = 1;
        pthread_mutex_unlock(&g_plan_mtx);
    }
    if (ezhip_interp_pts(&pp, d_zout, d_zin, s->d_x, s->d_y, go->ni * go->nj)) return dev_fail("the per-point interpolation kernel");
    if (st_build) {
        int cap = getenv("EZHIP_ST_CAP") ? atoi(getenv("EZHIP_ST_CAP")) : 2560;
        if (cap < 256) cap = 256;
        if (cap > 8192) cap = 8192;
        ezhip_pts_plan pt = pp;
        pt.pv_out = NULL; pt.uvt_cap = cap;
        const int nt = ezhip_uvt_ntiles(&pt, 3232);
        const size_t st_bytes = nt > 0 ? 20 * (size_t)nt + (size_t)8 * 1024 * (size_t)nt : 0;      /* (16 per table entry + the gathering path's tile list) */
        pthread_mutex_lock(&g_plan_mtx);
        const int fits = nt > 0 && cache_reserve(s, st_bytes);
        pthread_mutex_unlock(&g_plan_mtx);
        void *dt = fits ? ezhip_malloc(20 * (size_t)nt) : NULL, *ds = fits ? ezhip_malloc((size_t)8 * 1024 * (size_t)nt) : NULL;
        int st[4] = {0, 0, 0, 0}, good = 0;
        if (dt && ds && (degree == DEG_CUBIC ? ezhip_uvt_build(&pt, s->d_x, s->d_y, dt, 3232, st, 2) : ezhip_st1_build(&pt, s->d_x, s->d_y, dt, st)) == 0 && st[0] >= 4 * st[1] && ezhip_st_pack_streams(&pt, s->d_x, s->d_y, ds) == 0) good = 1;
        pthread_mutex_lock(&g_plan_mtx);
        if (s->st_key[sl] == 0) {
            if (good) { s->d_st_tiles[sl] = dt; s->d_st_streams[sl] = ds; s->st_cap[sl] = cap; memcpy(s->st_stats[sl], st, sizeof(st)); s->st_key[sl] = st_key; dt = ds = NULL; }
            else s->st_key[sl] = -1;
        }
        if ((dt || ds || !good) && fits) cache_unreserve(s, st_bytes);      /* not kept: the reservation goes back */
        pthread_mutex_unlock(&g_plan_mtx);
        /* ... and the special points this call's gathering kernel has just listed (pole points, polar strips; a set with re-interpolated points keeps the launch of its own) */
        if (good && sl == 1 && zones == 1 && pp.pv_out && !getenv("EZHIP_NO_SPEC_CACHE")) {      /* (pv_out: this call listed its pole points too -- their values came out of its own launch) */
            const int cnt = ezhip_pts2_special_snapshot(NULL, NULL, NULL, 0, s->d_x, s->d_y);
            if (cnt > 0 && cnt <= 4096) {
                int *dl = (int *)ezhip_malloc(sizeof(int) * (size_t)cnt);
                float *dxy = (float *)ezhip_malloc(sizeof(float) * 2 * (size_t)cnt);
                if (dl && dxy && ezhip_pts2_special_snapshot(dl, dxy, dxy + cnt, cnt, s->d_x, s->d_y) == cnt) {
                    pthread_mutex_lock(&g_plan_mtx);
                    if (s->sspec_key == 0 && s->st_key[sl] == st_key) { s->d_sspec_list = dl; s->d_sspec_xy = dxy; s->sspec_count = cnt; s->sspec_key = st_key; dl = NULL; dxy = NULL; }
                    pthread_mutex_unlock(&g_plan_mtx);
                }
                ezhip_free(dl); ezhip_free(dxy);
            }
        }
        if (good && getenv("EZHIP_VERBOSE")) fprintf(stderr, "<ezhip> k_st tile table of set (%d, %d): %d tiles staged, %d gathered, %d without a normal point, largest window %d cells\n", s->gdin, s->gdout, st[0], st[1], st[2], st[3]);
        ezhip_free(dt); ezhip_free(ds);
    }
    return ierc;
}

/* c_ezuvint_batch_dev in flight on this host thread: n pairs (0: none), pair f's components at d_in + f * in_stride, its results at d_out + f * out_stride;
 * refused := 1 when the launch layer has no batch form for the plan after all (nothing was launched: the caller goes pair by pair) */
static __thread struct { int n, refused; size_t in_stride, out_stride; } t_pairs;
/* the polar wind rows of the pair being interpolated, handed from uvint_impl to the per-point pair launch (per host thread) */
static __thread struct { float *out; const float *plon2, *ax; float xg4_n, xg4_s; int weighted, active; } t_pwjob;
/* both components of a wind pair on the per-point path in ONE pass (k_pts2); returns -2 when the set is not on that path */
/* additive (round 6): EXACT WINDS.  By default c_ezuvint applies the wind chain of a grid pair as a per-point 2 x 2 matrix made once per set from the chain itself, and
 * evaluates bicubic pairs from rotated sources in REAL with a REAL*8 second pass: within 2e-6 |V| of the reference (the chain's own noise: its direction passes through REAL
 * degrees) at 58 - 70 us per cfg3 pair.  With ezhip_set_wind_exact(1) every call runs the reference's chain as written (k_wind_rotate: speed / direction through REAL degrees, the C
 * library's REAL trig from libm_exact.h, REAL*8 where the reference has it) on components interpolated by the scalar kernels, pole rows included: at cfg3 the 16 M values of
 * nearest, bilinear and bicubic winds equal the reference build's BIT FOR BIT -- at ~290 us per pair.  Process-wide; 0 switches back. */
static int g_wind_exact = 0;
void ezhip_set_wind_exact(int32_t on) { g_wind_exact = on ? 1 : 0; }
int32_t ezhip_get_wind_exact(void) { return g_wind_exact; }
#define WIND_EXACT() (g_wind_exact != 0)

static int run_pair_pts(ezh_set *s, float *d_uo, float *d_vo, const float *d_ui, const float *d_vi,
                        const float *pun, const float *pus, const float *pvn, const float *pvs, const void *d_M, int m_half, int dst_rot)
{
    ezh_grid *gi = &G[s->gdin], *go = &G[s->gdout];
    int degree = O.degre_interp, polar = O.polar_correction == 1;
    if (degree != DEG_NEAREST && degree != DEG_LINEAR && degree != DEG_CUBIC) return -2;
    if (getenv("EZHIP_NO_PTS2") || WIND_EXACT() || src_hemi(gi) || grid_yinv(gi)) return -2;
    pthread_mutex_lock(&g_plan_mtx);
    int mode = choose_mode(s, degree, polar);
    pthread_mutex_unlock(&g_plan_mtx);
    if (mode != 2) return -2;
    if (ensure_scratch(s)) return -1;
    float *d_fill = t_scratch8, *d_poles = t_scratch8 + 4;
    int ierc = 0;
    ezhip_pts_plan pu, pv;
    int zones = !polar ? 0 : (s->extrap ? 2 : 1);
    pthread_mutex_lock(&g_plan_mtx);
    int erc = ensure_points(s);
    if (!erc) fill_pts_plan(s, gi, &pu, degree, zones, 1);
    pthread_mutex_unlock(&g_plan_mtx);
    if (erc) return -1;
    if (polar && s->extrap && s->have_dehors) {
        if (O.degre_extrap == XT_ABORT) return -2;          /* let the per-field path report it (ez_corrval.c:56-60) */
        if (ezhip_fill_value(d_fill, d_ui, (size_t)gi->ni * gi->nj, O.degre_extrap, O.valeur_extrap, 1)) return -1;
    }
    pu.fill = d_fill; pu.polevals = d_poles;
    pu.wind_M = d_M; pu.wind_M_half = m_half; pu.wind_dst_rot = dst_rot;
    pv = pu;
    pu.pole_row_n = pun; pu.pole_row_s = pus; pv.pole_row_n = pvn; pv.pole_row_s = pvs;
    if (zones == 2 && s->have_dehors) ierc = 2;
    pu.tile_ni = go->ni; pu.tile_nj = go->nj;                  /* the points are the whole target grid in row order: 2-D tile order */
    if (t_pwjob.active) { pu.pw_out = t_pwjob.out; pu.pw_plon2 = t_pwjob.plon2; pu.pw_xg4_n = t_pwjob.xg4_n; pu.pw_xg4_s = t_pwjob.xg4_s; pu.pw_weighted = t_pwjob.weighted; pu.pw_ax = t_pwjob.ax; t_pwjob.active = 0; }
    /* the special points of the set under these zone options: listed by the first launch, kept with the set, handed to the later ones */
    const int key = 1 | zones << 1 | (O.degre_extrap & 0xFF) << 4 | degree << 12;
    const int use_cache = !getenv("EZHIP_NO_SPEC_CACHE");
    if (use_cache) {
        pthread_mutex_lock(&g_plan_mtx);
        if (s->cspec_key == key) {
            pu.cspec_valid = 1; pu.cspec_count = s->cspec_count; pu.cspec_list = s->d_cspec_list;
            pu.cspec_x = s->d_cspec_xy; pu.cspec_y = s->d_cspec_xy ? s->d_cspec_xy + s->cspec_count : NULL;
            if (s->d_uvt_tiles && !getenv("EZHIP_NO_UVT")) { pu.uvt_tiles = s->d_uvt_tiles; pu.uvt_shape = s->uvt_shape; pu.uvt_cap = s->uvt_cap; pu.uvt_nhb = getenv("EZHIP_UVT_TABLE_ORDER") ? 0 : s->uvt_stats[1];
                if (s->d_uvt_streams && s->uvt_streams_M == d_M && (!d_M || m_half) && !getenv("EZHIP_UVT_NO_STREAMS")) pu.uvt_streams = s->d_uvt_streams; }      /* EZHIP_NO_UVT: the gathering kernel (same results) */
        }
        pthread_mutex_unlock(&g_plan_mtx);
    }
    if (t_pairs.n > 1) {
        if (!pu.cspec_valid || !pu.uvt_streams) { t_pairs.refused = 1; return -1; }
        pu.npairs = t_pairs.n; pu.pair_in_stride = t_pairs.in_stride; pu.pair_out_stride = t_pairs.out_stride; pu.pair_rows_stride = 4 * gi->ni;
    }
    {
        const int prc = ezhip_interp_pts2(&pu, &pv, d_uo, d_vo, d_ui, d_vi, s->d_x, s->d_y, go->ni * go->nj);
        if (prc == -2 && t_pairs.n > 1) { t_pairs.refused = 1; return -1; }
        if (prc) return -1;
    }
    if (use_cache && !pu.cspec_valid && s->cspec_key == 0) {              /* (the first option set seen on the set is the one that is kept: the arrays are never replaced while launches may read them) */
        const int cnt = ezhip_pts2_special_snapshot(NULL, NULL, NULL, 0, s->d_x, s->d_y);        /* synchronises: once per set and option set */
        if (cnt >= 0) {
            int *dl = cnt ? (int *)ezhip_malloc(sizeof(int) * (size_t)cnt) : NULL;
            float *dxy = cnt ? (float *)ezhip_malloc(sizeof(float) * 2 * (size_t)cnt) : NULL;
            if (cnt == 0 || (dl && dxy && ezhip_pts2_special_snapshot(dl, dxy, dxy + cnt, cnt, s->d_x, s->d_y) == cnt)) {
                pthread_mutex_lock(&g_plan_mtx);
                int first = 0;
                if (s->cspec_key == 0) { s->d_cspec_list = dl; s->d_cspec_xy = dxy; s->cspec_count = cnt; s->cspec_key = key; dl = NULL; dxy = NULL; first = 1; }
                pthread_mutex_unlock(&g_plan_mtx);
                /* with the special points known, the irregular bicubic pair can run from LDS-staged stencil windows: the tile table over the set's x, y, once */
                if (first && degree == DEG_CUBIC && pu.irregular && (pu.wrap != 0 || (pu.i1 == 1 && pu.i2 == pu.ni)) && pu.tile_ni > 0 && pu.xrec10) {
                    const int th = pu.wrap == 0 ? 3232 : getenv("EZHIP_UVT_SHAPE") ? atoi(getenv("EZHIP_UVT_SHAPE")) : 3232;      /* 100 TW + TH: 3232, 3216, 6416, 6408 (same results) */
                    int cap = getenv("EZHIP_UVT_CAP") ? atoi(getenv("EZHIP_UVT_CAP")) : 2560;      /* 8 bytes a cell + 9 KB of records: five blocks per CU */
                    if (cap < 256) cap = 256;
                    if (cap > 4096) cap = 4096;
                    pu.uvt_cap = cap;
                    const int nt = ezhip_uvt_ntiles(&pu, th);
                    const size_t tb_bytes = nt > 0 ? 20 * (size_t)nt : 0, sb_bytes = nt > 0 ? ezhip_uvt_stream_bytes(&pu, th) : 0;      /* (16 per entry + the handed-back tiles' list) */
                    pthread_mutex_lock(&g_plan_mtx);
                    const int fits_t = nt > 0 && cache_reserve(s, tb_bytes);
                    pthread_mutex_unlock(&g_plan_mtx);
                    void *dt = fits_t ? ezhip_malloc(tb_bytes) : NULL;
                    int st[4] = {0, 0, 0, 0}, kept_t = 0;
                    if (dt && ezhip_uvt_build(&pu, s->d_x, s->d_y, dt, th, st, 1) == 0 && st[0] >= 4 * st[1]) {      /* (a set whose tiles mostly do not qualify keeps the gathering kernel) */
                        pthread_mutex_lock(&g_plan_mtx);
                        s->uvt_shape = th; s->uvt_cap = cap; memcpy(s->uvt_stats, st, sizeof(st)); s->d_uvt_tiles = dt; dt = NULL; kept_t = 1;
                        const int fits_s = (!d_M || m_half) && cache_reserve(s, sb_bytes);
                        pthread_mutex_unlock(&g_plan_mtx);
                        if (fits_s) {      /* x, y and the pair's packed rotation once more in tile order (12 bytes per target point) */
                            void *ds = ezhip_malloc(sb_bytes);
                            int kept_s = 0;
                            if (ds && ezhip_uvt_pack_streams(&pu, s->d_x, s->d_y, ds, th) == 0) {
                                pthread_mutex_lock(&g_plan_mtx);
                                s->d_uvt_streams = ds; s->uvt_streams_M = d_M; ds = NULL; kept_s = 1;
                                pthread_mutex_unlock(&g_plan_mtx);
                            }
                            ezhip_free(ds);
                            if (!kept_s) { pthread_mutex_lock(&g_plan_mtx); cache_unreserve(s, sb_bytes); pthread_mutex_unlock(&g_plan_mtx); }
                        }
                        if (getenv("EZHIP_VERBOSE")) fprintf(stderr, "<ezhip> k_uvt tile table of set (%d, %d): %d tiles staged, %d gathered, %d empty, largest window %d cells\n", s->gdin, s->gdout, st[0], st[1], st[2], st[3]);
                    }
                    if (fits_t && !kept_t) { pthread_mutex_lock(&g_plan_mtx); cache_unreserve(s, tb_bytes); pthread_mutex_unlock(&g_plan_mtx); }
                    ezhip_free(dt);
                }
            }
            ezhip_free(dl); ezhip_free(dxy);                        /* (another thread was first, or the copy failed) */
        }
    }
    return ierc;
}

void ezhip_use_stream(void *hip_stream) { ezhip_set_stream(hip_stream); }
int32_t ezhip_develop_build(void) { return EZH_DEVELOP_BUILD; }      /* 1: built with -DEZHIP_DEVELOP (kernel knock-outs compiled in; never the shipped library) */
int32_t ezhip_available(void) { return ezhip_runtime_ok(); }

static int yy_plan(ezh_set *s);
static int32_t yy_sint(ezh_set *s, float *d_zout, const float *d_zin);
static int32_t yy_sint_to_u(ezh_set *s, float *d_zout, const float *d_zin);
static int32_t yy_uvint(ezh_set *s, float *d_uuout, float *d_vvout, const float *d_uuin, const float *d_vvin, int wd_only);

int32_t c_ezsint_dev(float *d_zout, const float *d_zin)
{
    ezh_set *s = current_set("c_ezsint");
    if (!s) return -1;
    if (need_device("c_ezsint") || device_error("c_ezsint")) return -1;
    if (G[s->gdout].grtyp == 'U') return yy_sint_to_u(s, d_zout, d_zin);
    if (G[s->gdin].grtyp == 'U') return yy_sint(s, d_zout, d_zin);
    if (s->gdin == s->gdout) { ezhip_d2d(d_zout, d_zin, sizeof(float) * (size_t)G[s->gdin].ni * G[s->gdin].nj); return 1; }
    return run_field(s, d_zout, d_zin, 0, NULL, NULL);
}

static int ensure_batch_poles(ezh_set *s, int nfields)
{
    if (nfields > s->poles_cap) {
        ezhip_sync(); ezhip_free(s->d_poles_batch);
        s->d_poles_batch = (float *)ezhip_malloc(sizeof(float) * 2 * (size_t)nfields);
        s->poles_cap = s->d_poles_batch ? nfields : 0;
    }
    return s->d_poles_batch ? 0 : -1;
}

/* stat_partials != NULL: the launch also leaves, per field, {min key, max key, 0} triples of every thread block's
 * output at stat_partials[f * stat_stride + 3 k], k < *partials_per_field (compact_float's min/max pass fused into the
 * interpolation); only the single-launch k_sepx path can do that: returns -2 when it does not apply */
typedef struct { int mode; const void *qparams; size_t qstride, out_stride_words; } batch_out;      /* ezhip_sep_plan.out_mode and friends */
/* c_ezsint_batch_dev on the per-point path with the set's staged-tile table in place: one k_st launch + one k_pts_special launch for all fields.  -2: not this case
 * (no table yet or built under other zone options, extrapolation zones -- their fill value is a field's own --, a source that is expanded or flipped first, the
 * background correction of 'B' grids behind the interpolation) */
static int run_batch_st(ezh_set *s, float *d_zout, const float *d_zin, int nfields, size_t nin, size_t nout, const float *d_poles_all)
{
    ezh_grid *gi = &G[s->gdin], *go = &G[s->gdout];
    const int degree = O.degre_interp, polar = O.polar_correction == 1;
    if ((degree != DEG_CUBIC && degree != DEG_LINEAR) || grid_yinv(gi) || src_hemi(gi) || corrbgd_applies(gi, go) || (polar && s->extrap)) return -2;
    if (polar && !d_poles_all) return -2;
    pthread_mutex_lock(&g_plan_mtx);
    const int mode = choose_mode(s, degree, polar);
    const int zones = !polar ? 0 : 1;
    const int st_key = 1 | zones << 1 | (O.degre_extrap & 0xFF) << 4 | degree << 12, sl = degree == DEG_CUBIC;
    const int have = mode != 1 && s->st_key[sl] == st_key && s->d_st_tiles[sl] != NULL;
    pthread_mutex_unlock(&g_plan_mtx);
    if (!have) return -2;
    if (hio_full((float *)d_zin) || ensure_scratch(s)) return -1;
    ezhip_pts_plan pp;
    pthread_mutex_lock(&g_plan_mtx);
    int erc = ensure_points(s);
    if (!erc) fill_pts_plan(s, gi, &pp, degree, zones, 0);
    pthread_mutex_unlock(&g_plan_mtx);
    if (erc) return -1;
    if (!pp.irregular || (degree == DEG_CUBIC && !pp.xrec10) || pp.out_idx) return -2;
    pp.fill = t_scratch8; pp.polevals = d_poles_all ? d_poles_all : t_scratch8 + 4;
    pp.tile_ni = go->ni; pp.tile_nj = go->nj;
    pp.uvt_tiles = s->d_st_tiles[sl]; pp.uvt_shape = 3232; pp.uvt_cap = s->st_cap[sl]; pp.uvt_streams = s->d_st_streams[sl];
            pp.uvt_nhb = (sl == 1 && !getenv("EZHIP_UVT_TABLE_ORDER")) ? s->st_stats[sl][1] + s->st_stats[sl][2] : 0;      /* (bicubic: k_st; the bilinear k_st1 keeps table order) */
    const int rc = ezhip_interp_pts_batch(&pp, d_zout, d_zin, s->d_x, s->d_y, go->ni * go->nj, nfields, nin, nout);
    if (rc == -2) return -2;
    return rc ? dev_fail("the per-point interpolation kernel (batch)") : 0;
}

static int32_t batch_impl_o(float *d_zout, const float *d_zin, int32_t nfields, unsigned *stat_partials, size_t stat_stride, int32_t *partials_per_field, const batch_out *bo);
static int32_t batch_impl(float *d_zout, const float *d_zin, int32_t nfields, unsigned *stat_partials, size_t stat_stride, int32_t *partials_per_field)
{
    return batch_impl_o(d_zout, d_zin, nfields, stat_partials, stat_stride, partials_per_field, NULL);
}
/* bo != NULL: the launch leaves min/max partials only (mode 2) or 16-bit tokens (mode 3) instead of float fields; only the
 * single-launch k_sepx path can do that: returns -2 when it does not apply */
static int32_t batch_impl_o(float *d_zout, const float *d_zin, int32_t nfields, unsigned *stat_partials, size_t stat_stride, int32_t *partials_per_field, const batch_out *bo)
{
    ezh_set *s = current_set("c_ezsint_batch");
    if (!s) return -1;
    if (need_device("c_ezsint_batch") || device_error("c_ezsint_batch")) return -1;
    size_t nin = (size_t)G[s->gdin].ni * G[s->gdin].nj, nout = (size_t)G[s->gdout].ni * G[s->gdout].nj;
    int rc = 0;
    if (G[s->gdin].grtyp == 'U' || G[s->gdout].grtyp == 'U') {  /* Yin-Yang source or target: field by field */
        if (bo) return -2;
        if (stat_partials) return -1;
        for (int f = 0; f < nfields; f++) {
            int r = G[s->gdout].grtyp == 'U' ? yy_sint_to_u(s, d_zout + f * nout, d_zin + f * nin) : yy_sint(s, d_zout + f * nout, d_zin + f * nin);
            if (r < 0) return r;
            if (r > rc) rc = r;
        }
        return rc;
    }
    /* separable plan without extrapolation fill: ALL fields in one k_sepx launch (no ramp-up / drain gap between
     * fields; the pole rows are summed by producer blocks of the same launch) */
    if (s->gdin != s->gdout && (nfields > 1 || stat_partials || bo) && !(O.polar_correction == 1 && s->extrap) && !grid_yinv(&G[s->gdin]) &&
        (O.degre_interp == DEG_NEAREST || O.degre_interp == DEG_LINEAR || O.degre_interp == DEG_CUBIC) &&
        choose_mode(s, O.degre_interp, O.polar_correction == 1) == 1 && !getenv("EZHIP_NO_BATCH_LAUNCH") && !getenv("EZHIP_NO_SEPX")) {
        int degree = O.degre_interp, polar = O.polar_correction == 1;
        if (ensure_scratch(s)) return -1;
        pthread_mutex_lock(&g_plan_mtx);
        int brc = build_sep_plan(s, degree, 0, polar);
        ezhip_sep_plan p = s->sep[degree == DEG_CUBIC ? 2 : degree][0].p;
        pthread_mutex_unlock(&g_plan_mtx);
        if (brc) return -1;
        if (p.x_nseg > 0) {
            ezh_grid *gi = &G[s->gdin];
            p.fill = t_scratch8;
            if (p.pole_weighted) { if (ensure_grid_dev(gi)) return -1; p.ax = gi->d_ax; }
            if (p.need_poles) {
                if (nfields >= 4) { if (pole_ring(nfields, &p)) return -1; }      /* producer blocks of the same launch */
                else {                                                              /* tiny batch: k_polevals first (see run_field_p) */
                    if (ensure_batch_poles(s, nfields)) return -1;
                    if (ezhip_polevals_batch(s->d_poles_batch, d_zin, nin, nfields, gi->ni, gi->nj, p.pole_weighted, p.ax)) return -1;
                    p.polevals = s->d_poles_batch;
                }
            }
            p.batch_fields = nfields; p.batch_in_stride = nin; p.batch_out_stride = nout;
            if (EZH_DEVENV("EZHIP_DEBUG")) p.debug_flags = EZH_DEVINT("EZHIP_DEBUG");      /* development: the kernel's knock-outs / variants at call time too (the plan keeps what the environment held when it was built) */
            if (getenv("EZHIP_BATCH_RB")) { int rb = atoi(getenv("EZHIP_BATCH_RB")); if (rb >= 1) { p.x_rb = rb; p.x_nseg = (p.x_nvb + rb - 1) / rb; } }      /* development: row-blocks per thread block of a batch launch */
            if (getenv("EZHIP_BATCH_SPECIAL_PCT")) { int pct = atoi(getenv("EZHIP_BATCH_SPECIAL_PCT")); if (pct >= 0 && pct <= 100) p.special_last = 2 + p.x_nseg * pct / 100; }      /* development: the special rows at pct % of a field's work order (default: the middle) */
            if (bo) {
                if (bo->mode == 3 && (p.ni_dst & 1)) return -2;               /* a token word holds two columns of one row */
                p.out_mode = bo->mode; p.quant_params = bo->qparams; p.quant_stride = bo->qstride;
                if (bo->mode == 3) {
                    p.batch_out_stride = bo->out_stride_words;
                    /* the token pass stores half the bytes of the float pass: the source rows a thread block stages again where the one above it stopped weigh
                     * more, and longer walks down a strip pay (tools/sweep_cfg5.py, interleaved: 79.2 us per field of the cfg5 pipeline with the plan's 7
                     * row-blocks per thread block, 78.2 with 10, 77.9 with 14, 77.6 with 19, 77.9 with 28; the float pass is flat from 7 to 14) */
                    if (!getenv("EZHIP_BATCH_RB")) { int rb = (8 * p.x_rb + 2) / 3; if (rb > p.x_nvb) rb = p.x_nvb; if (rb >= 1) { p.x_rb = rb; p.x_nseg = (p.x_nvb + rb - 1) / rb; } }
                }
            }
            if (stat_partials) {
                int nbx = (p.ni_dst + EZHIP_SEP_COLS - 1) / EZHIP_SEP_COLS, npf = nbx * (p.x_nseg + p.n_special);
                if ((size_t)3 * npf > stat_stride) return -2;
                p.stat_partials = stat_partials; p.stat_stride = stat_stride;
                if (partials_per_field) *partials_per_field = npf;
            }
            return ezhip_interp_sep(&p, d_zout, d_zin) ? -1 : 0;
        }
    }
    if (stat_partials || bo) return -2;
    /* field by field; the pole values (a sequential REAL sum per pole row) of ALL fields from one k_polevals launch */
    float *d_poles_all = NULL;
    if (s->gdin != s->gdout && O.polar_correction == 1 && !s->extrap && nfields > 1) {
        ezh_grid *gi = &G[s->gdin];
        if (ensure_batch_poles(s, nfields)) return -1;
        int weighted = (gi->grtyp == 'Z' && gi->grref == 'E');
        if (weighted && ensure_grid_dev(gi)) return -1;
        if (s->d_poles_batch && ezhip_polevals_batch(s->d_poles_batch, d_zin, nin, nfields, gi->ni, gi->nj, weighted, gi->d_ax) == 0)
            d_poles_all = s->d_poles_batch;
    }
    /* a rotated / regional source whose set has its staged-tile table (k_st, from the second call on): the whole batch in one launch -- x, y, the zones and the
     * list of special points once per batch instead of once per field */
    if (s->gdin != s->gdout && nfields > 1 && !getenv("EZHIP_NO_ST_BATCH")) {
        const int brc = run_batch_st(s, d_zout, d_zin, nfields, nin, nout, d_poles_all);
        if (brc != -2) return brc;
    }
    for (int f = 0; f < nfields; f++) {
        int r = (s->gdin == s->gdout) ? (ezhip_d2d(d_zout + f * nout, d_zin + f * nin, sizeof(float) * nin), 1)
                                      : run_field_p(s, d_zout + f * nout, d_zin + f * nin, 0, NULL, NULL, d_poles_all ? d_poles_all + 2 * f : NULL);
        if (r < 0) return r;
        if (r > rc) rc = r;
    }
    return rc;
}

int32_t c_ezsint_batch_dev(float *d_zout, const float *d_zin, int32_t nfields)
{
    return batch_impl(d_zout, d_zin, nfields, NULL, 0, NULL);
}

int32_t ezhip_ezsint_batch_minmax_dev(float *d_zout, const float *d_zin, int32_t nfields, uint32_t *d_partials, int64_t stride_words, int32_t *partials_per_field)
{
    if (!d_partials || stride_words <= 0) return -1;
    return batch_impl(d_zout, d_zin, nfields, d_partials, (size_t)stride_words, partials_per_field);
}

/* cfg5 pipeline, pass A: interpolate nfields fields WITHOUT storing them; only the {min key, max key, 0} triples of every
 * thread block (compact_float's min/max pass, compact.tmplc:173-204) are left in d_partials.  -2: plan not on the k_sepx path */
int32_t ezhip_ezsint_batch_minmax_only_dev(const float *d_zin, int32_t nfields, uint32_t *d_partials, int64_t stride_words, int32_t *partials_per_field)
{
    if (!d_partials || stride_words <= 0) return -1;
    batch_out bo = {2, NULL, 0, 0};
    return batch_impl_o(NULL, d_zin, nfields, d_partials, (size_t)stride_words, partials_per_field, &bo);
}
/* cfg5 pipeline, pass A without interpolating (k_bb_*, ez_kernels.hip): the exact {min key, max key, 0} of the values c_ezsint would produce for
 * each of nfields source fields, left in d_partials[f * stride_words + 0..2] (*partials_per_field = 1).  d_flags[f] (device) := 1 for a field in
 * which too many source windows can hold the extremum: its triple is NOT valid and the caller runs ezhip_ezsint_batch_minmax_only_dev for it.
 * Asynchronous.  -2: the plan of the current grid set has no bound tables (not on the k_sepx path, vector plan, extrapolation, ...). */
static __thread struct { void *p; size_t cap; } t_bbws;
int32_t ezhip_ezsint_batch_minmax_bb_dev(const float *d_zin, int32_t nfields, uint32_t *d_partials, int64_t stride_words, int32_t *partials_per_field, int32_t *d_flags)
{
    if (!d_partials || stride_words < 3 || !d_flags || nfields < 1) return -1;
    ezh_set *s = current_set("ezhip_ezsint_batch_minmax_bb_dev");
    if (!s) return -1;
    if (need_device("ezhip_ezsint_batch_minmax_bb_dev") || device_error("ezhip_ezsint_batch_minmax_bb_dev")) return -1;
    if (G[s->gdin].grtyp == 'U' || G[s->gdout].grtyp == 'U' || s->gdin == s->gdout) return -2;
    if ((O.polar_correction == 1 && s->extrap) || grid_yinv(&G[s->gdin])) return -2;
    if (!(O.degre_interp == DEG_NEAREST || O.degre_interp == DEG_LINEAR || O.degre_interp == DEG_CUBIC)) return -2;
    if (choose_mode(s, O.degre_interp, O.polar_correction == 1) != 1 || getenv("EZHIP_NO_BATCH_LAUNCH") || getenv("EZHIP_NO_SEPX")) return -2;
    const int degree = O.degre_interp, polar = O.polar_correction == 1;
    if (ensure_scratch(s)) return -1;
    pthread_mutex_lock(&g_plan_mtx);
    int brc = build_sep_plan(s, degree, 0, polar);
    ezhip_sep_plan p = s->sep[degree == DEG_CUBIC ? 2 : degree][0].p;
    pthread_mutex_unlock(&g_plan_mtx);
    if (brc) return -1;
    if (p.x_nseg <= 0 || !p.bb_ok) return -2;
    ezh_grid *gi = &G[s->gdin];
    const size_t nin = (size_t)gi->ni * gi->nj;
    p.fill = t_scratch8;
    if (p.pole_weighted) { if (ensure_grid_dev(gi)) return -1; p.ax = gi->d_ax; }
    const size_t wb = ezhip_bb_work_bytes(&p, nfields);
    if (t_bbws.cap < wb) {
        if (t_bbws.p) { ezhip_sync(); ezhip_free(t_bbws.p); }
        t_bbws.p = ezhip_malloc(wb + wb / 4); t_bbws.cap = t_bbws.p ? wb + wb / 4 : 0;
        if (!t_bbws.p) return -1;
    }
    const float *d_poles = NULL;
    int side = 0;
    if (p.need_poles && p.n_special > 0) {
        /* the pole values (a sequential REAL sum per pole row: ~25 us on one lane) are only read by the special rows at the end: side stream */
        if (ensure_batch_poles(s, nfields)) return -1;
        if (ezhip_side_begin()) return -1;
        side = 1;
        int bad = ezhip_polevals_batch(s->d_poles_batch, d_zin, nin, nfields, gi->ni, gi->nj, p.pole_weighted, p.ax);
        if (ezhip_side_end() || bad) { ezhip_side_join(); return -1; }
        d_poles = s->d_poles_batch;
    }
    /* (round 6) the special rows need the bounds pass's reduction (it initialises the fields' keys and flags) and the pole values, nothing of the second sweep: they run on
     * the side stream -- behind the pole sums, which are already there -- beside k_bb_select / k_bb_eval (39 of the pass's 420 us per batch of 32) */
    int rc = ezhip_minmax_bb_stage(&p, d_zin, nin, nfields, d_partials, (size_t)stride_words, d_flags, d_poles, t_bbws.p, getenv("EZHIP_BB_ONE_STREAM") ? 0 : 1);
    if (rc) { if (side) ezhip_side_join(); return rc == -2 ? -2 : -1; }
    if (getenv("EZHIP_BB_ONE_STREAM")) {
        if (side && ezhip_side_join()) return -1;
        if (ezhip_minmax_bb_special(&p, d_zin, nin, nfields, d_partials, (size_t)stride_words, d_flags, d_poles, t_bbws.p)) return -1;
    } else {
        if (ezhip_side_begin()) return -1;              /* (the side stream now waits for the reduction as well; it is in order behind its own pole sums) */
        int bad = ezhip_minmax_bb_special(&p, d_zin, nin, nfields, d_partials, (size_t)stride_words, d_flags, d_poles, t_bbws.p);
        if (ezhip_side_end() || bad) { ezhip_side_join(); return -1; }
        rc = ezhip_minmax_bb_stage(&p, d_zin, nin, nfields, d_partials, (size_t)stride_words, d_flags, d_poles, t_bbws.p, 2);
        if (ezhip_side_join() || rc) return -1;
    }
    if (partials_per_field) *partials_per_field = 1;
    return 0;
}

/* cfg5 pipeline, pass B: interpolate again and leave compact_float's 16-bit tokens of every value (two per word, first in the
 * high half: the layout armn_compress consumes), quantised with the {minF, mulFactor} at d_params + f * param_stride_bytes */
int32_t ezhip_ezsint_batch_tokens_dev(uint32_t *d_tokens, int64_t token_stride_words, const float *d_zin, int32_t nfields,
                                      const void *d_params, int64_t param_stride_bytes)
{
    if (!d_tokens || !d_params) return -1;
    batch_out bo = {3, d_params, (size_t)param_stride_bytes, (size_t)token_stride_words};
    return batch_impl_o((float *)d_tokens, d_zin, nfields, NULL, 0, NULL, &bo);
}

/* cfg5 pipeline, passes B + E in ONE launch (k_sepx_enc, ez_kernels.hip): interpolate nfields source fields, quantise with the {minF, mulFactor} at
 * d_params + f * param_stride_bytes (compact.tmplc:285-300) and armn_compress (c_zfstlib.c:646-789, container 4) straight into d_z + f * z_stride_words;
 * the tokens never reach HBM.  d_work: ezhip_sepenc_layout_of(ni_out, nj_out, nfields).total bytes of device scratch; the stream's prefix words are left
 * to packhip_sepenc_prefix (same scratch).  d_zlng[f] := byte count, -1 / -2 (not compressible / needs the 5-bit width field: the caller redoes the field
 * with the two-kernel path) or untouched when the launch gave up.  poles_ready: the pole values of exactly these fields are in the set's batch array
 * (ezhip_ezsint_batch_minmax_bb_dev ran on them).  Asynchronous.  -2: the plan of the current grid set has no fused geometry */
int32_t ezhip_ezsint_batch_encode_dev(uint32_t *d_z, int64_t z_stride_words, int64_t z_cap_words, const float *d_zin, int32_t nfields,
                                      const void *d_params, int64_t param_stride_bytes, int32_t nbits, void *d_work, int32_t *d_zlng, int32_t poles_ready)
{
    if (!d_z || !d_zin || !d_params || !d_work || !d_zlng || nfields < 1 || nbits < 5 || nbits > 16) return -1;
    ezh_set *s = current_set("ezhip_ezsint_batch_encode_dev");
    if (!s) return -1;
    if (need_device("ezhip_ezsint_batch_encode_dev") || device_error("ezhip_ezsint_batch_encode_dev")) return -1;
    if (G[s->gdin].grtyp == 'U' || G[s->gdout].grtyp == 'U' || s->gdin == s->gdout) return -2;
    if ((O.polar_correction == 1 && s->extrap) || grid_yinv(&G[s->gdin])) return -2;
    if (!(O.degre_interp == DEG_NEAREST || O.degre_interp == DEG_LINEAR || O.degre_interp == DEG_CUBIC)) return -2;
    if (choose_mode(s, O.degre_interp, O.polar_correction == 1) != 1 || getenv("EZHIP_NO_BATCH_LAUNCH") || getenv("EZHIP_NO_SEPX")) return -2;
    const int degree = O.degre_interp, polar = O.polar_correction == 1;
    if (ensure_scratch(s)) return -1;
    pthread_mutex_lock(&g_plan_mtx);
    int brc = build_sep_plan(s, degree, 0, polar);
    ezhip_sep_plan p = s->sep[degree == DEG_CUBIC ? 2 : degree][0].p;
    pthread_mutex_unlock(&g_plan_mtx);
    if (brc) return -1;
    if (!p.e_ok) return -2;
    ezh_grid *gi = &G[s->gdin];
    const size_t nin = (size_t)gi->ni * gi->nj;
    p.fill = t_scratch8;
    if (p.pole_weighted) { if (ensure_grid_dev(gi)) return -1; p.ax = gi->d_ax; }
    const float *d_poles = NULL;
    if (p.need_poles && p.n_special > 0) {
        if (ensure_batch_poles(s, nfields)) return -1;
        if (!poles_ready && ezhip_polevals_batch(s->d_poles_batch, d_zin, nin, nfields, gi->ni, gi->nj, p.pole_weighted, p.ax)) return -1;
        d_poles = s->d_poles_batch;
    }
    const ezhip_sepenc_layout L = ezhip_sepenc_layout_of(p.ni_dst, p.nj_dst, nfields);
    if (L.nstrips != p.e_nstrips) return -1;
    if (ezhip_memset(d_work, 0, L.zero_bytes)) return -1;
    ezhip_sepenc_args a;
    memset(&a, 0, sizeof(a));
    a.z = d_z; a.z_stride = (size_t)z_stride_words; a.z_cap = (size_t)z_cap_words;
    a.zin = d_zin; a.in_stride = nin; a.nfields = nfields; a.poles = d_poles;
    a.quant_params = d_params; a.quant_stride = (size_t)param_stride_bytes;
    a.nbits = nbits; a.container = 4; a.ntx = L.ntx; a.nty = L.nty; a.nchunks = L.nchunks;
    a.status = (unsigned long long *)((char *)d_work + L.off_status); a.tail = (unsigned long long *)((char *)d_work + L.off_tail);
    a.ctl = (unsigned *)((char *)d_work + L.off_ctl); a.head = (unsigned *)((char *)d_work + L.off_head);
    a.ptok = (unsigned short *)((char *)d_work + L.off_ptok); a.ptok_stride = L.ptok_stride;
    a.zlng = d_zlng;
    a.debug = EZH_DEVINT("EZHIP_ENC_DEBUG");
    return ezhip_interp_sep_enc(&p, &a) ? -1 : 0;
}

/* dimensions of the current grid set (c_ezdefset): what the batch launches read and write per field.  -1 without a set */
int32_t ezhip_current_set_dims(int32_t *ni_in, int32_t *nj_in, int32_t *ni_out, int32_t *nj_out)
{
    if (cur_gdin < 0 || cur_gdout < 0 || !grid_ok(cur_gdin) || !grid_ok(cur_gdout)) return -1;
    if (ni_in) *ni_in = G[cur_gdin].ni;
    if (nj_in) *nj_in = G[cur_gdin].nj;
    if (ni_out) *ni_out = G[cur_gdout].ni;
    if (nj_out) *nj_out = G[cur_gdout].nj;
    return 0;
}

int32_t ezhip_prepare_set(void)
{
    ezh_set *s = current_set("ezhip_prepare_set");
    if (!s) return -1;
    if (need_device("ezhip_prepare_set")) return -1;
    if (G[s->gdout].grtyp == 'U') return 0;
    if (G[s->gdin].grtyp == 'U') return yy_plan(s);
    int degree = O.degre_interp, polar = O.polar_correction == 1;
    if (degree != DEG_NEAREST && degree != DEG_LINEAR && degree != DEG_CUBIC) return 0;       /* the averaging degrees build nothing ahead of the call */
    const int trace = getenv("EZHIP_TRACE_FIRST") != NULL;
    double t0_ = trace ? wall_ms() : 0.0;
    const int mode = choose_mode(s, degree, polar);
    TRACE_PART("choose_mode");
    if (mode == 1) return build_sep_plan(s, degree, 0, polar);
    pthread_mutex_lock(&g_plan_mtx);
    int erc = ensure_points(s);                     /* locates and classifies the DEHORS zone (first-call work, like ez_defzones) */
    pthread_mutex_unlock(&g_plan_mtx);
    return erc ? -1 : 0;
}

/* the located x,y of the current set (the reference's gridset cache gset.x / gset.y, ez_calcxy.c:56-134), copied to
 * caller-owned device arrays of ni_out*nj_out floats: lets a test compare the locate on its own */
int32_t ezhip_set_xy_dev(float *d_x, float *d_y)
{
    ezh_set *s = current_set("ezhip_set_xy_dev");
    if (!s) return -1;
    if (need_device("ezhip_set_xy_dev")) return -1;
    if (G[s->gdin].grtyp == 'U' || G[s->gdout].grtyp == 'U') return -1;
    pthread_mutex_lock(&g_plan_mtx);
    int erc = ensure_points(s);
    pthread_mutex_unlock(&g_plan_mtx);
    if (erc) return -1;
    size_t n = (size_t)G[s->gdout].ni * G[s->gdout].nj;
    if (ezhip_d2d(d_x, s->d_x, sizeof(float) * n) || ezhip_d2d(d_y, s->d_y, sizeof(float) * n)) return -1;
    return 0;
}

int32_t ezhip_set_mode(void)
{
    ezh_set *s = current_set("ezhip_set_mode");
    if (!s) return -1;
    if (G[s->gdin].grtyp == 'U' || G[s->gdout].grtyp == 'U') return 2;
    return choose_mode(s, O.degre_interp, O.polar_correction == 1);
}

/* host-pointer ABI: stage through device buffers owned by the calling THREAD (grow-only, like t_scratch8 and the packers'
 * workspaces): two host threads may run c_ezsint / c_ezuvint on the same grid pair at the same time, each on its own stream
 * (buffers owned by the set were overwritten by the other thread, and allocated without a lock) */
static __thread struct { float *p; size_t cap; } t_stage[4];
static float *stage(int slot, size_t n)
{
    if (t_stage[slot].cap < n) {
        if (t_stage[slot].p) { ezhip_sync(); ezhip_free(t_stage[slot].p); }
        t_stage[slot].p = (float *)ezhip_malloc(sizeof(float) * n);
        t_stage[slot].cap = t_stage[slot].p ? n : 0;
    }
    return t_stage[slot].p;
}
/* everything the calling thread owns on the device: workspaces, staging buffers, pole granules (ezhip_thread_release / thread exit) */
void ezh_ez_thread_release(void)
{
    for (int k = 0; k < 4; k++) { ezhip_free(t_stage[k].p); t_stage[k].p = NULL; t_stage[k].cap = 0; }
    ezhip_free(t_scratch8); t_scratch8 = NULL;
    ezhip_free(t_pole.gran); t_pole.gran = NULL; t_pole.cap = 0;
    ezhip_free(t_xpn.p); t_xpn.p = NULL; t_xpn.cap = 0;
    ezhip_free(t_bbws.p); t_bbws.p = NULL; t_bbws.cap = 0;
}

int32_t c_ezsint(float *zout, float *zin)               /* ezsint.c:38-135 */
{
    ezh_set *s = current_set("c_ezsint");
    if (!s) return -1;
    size_t nin = (size_t)G[s->gdin].ni * G[s->gdin].nj, nout = (size_t)G[s->gdout].ni * G[s->gdout].nj;
    if (s->gdin == s->gdout) { memcpy(zout, zin, sizeof(float) * nin); return 1; }
    if (need_device("c_ezsint")) return -1;
    float *d_in = stage(0, nin), *d_out = stage(1, nout);
    if (!d_in || !d_out) return -1;
    int yy = G[s->gdin].grtyp == 'U' || G[s->gdout].grtyp == 'U';
    if (!yy && ezhip_prepare_set()) return -1;
    int rc;
    if (yy) {
        if (ezhip_h2d(d_in, zin, sizeof(float) * nin)) return -1;
        rc = G[s->gdout].grtyp == 'U' ? yy_sint_to_u(s, d_out, d_in) : yy_sint(s, d_out, d_in);
    } else {
        /* the source goes up where run_field_p first needs it: whole (hio_full), or row range by row range overlapped with the
         * download of the finished rows (sepx_host_chunks) */
        t_hio.zin = zin; t_hio.zout = zout; t_hio.nin = nin; t_hio.state = 1;
        t_hio.pinned = host_pinned(zin, sizeof(float) * nin) && host_pinned(zout, sizeof(float) * nout);
        rc = run_field(s, d_out, d_in, 0, NULL, NULL);
        const int st = t_hio.state;
        t_hio.state = 0;
        if (rc < 0) { ezhip_side_join(); ezhip_sync(); return rc; }
        if (st == 1) { fprintf(stderr, "<c_ezsint> internal error: the source field never reached the device\n"); return -1; }
        if (st == 3) { if (ezhip_sync()) return -1; if (device_error("c_ezsint")) return -1; return rc; }
    }
    if (rc < 0) return rc;
    if (ezhip_d2h(zout, d_out, sizeof(float) * nout) || ezhip_sync()) return -1;
    if (device_error("c_ezsint")) return -1;
    return rc;
}

/* c_gdxysint / c_gdxysval (gdxysint.c:30-54, gdxysval.c:50-123): caller-supplied x,y, no polar correction */
int32_t c_gdxysint_dev(float *d_zout, const float *d_zin, int32_t gdin, const float *d_x, const float *d_y, int32_t npts)
{
    if (!grid_ok(gdin)) return -1;
    if (need_device("c_gdxysint")) return -1;
    ezh_grid *gi = &G[gdin];
    int degree = O.degre_interp;
    if (degree != DEG_NEAREST && degree != DEG_LINEAR && degree != DEG_CUBIC) return -1;
    if (ensure_grid_dev(gi)) return -1;
    ezhip_pts_plan pp;
    fill_pts_plan(NULL, gi, &pp, degree, 0, 0);
    if (src_hemi(gi) || grid_yinv(gi)) { d_zin = hemi_expand(gi, d_zin); if (!d_zin) return -1; }      /* gdxysint.c:35-47 */
    return ezhip_interp_pts(&pp, d_zout, d_zin, d_x, d_y, npts) ? -1 : 0;
}
static float *stage(int slot, size_t n);
int32_t c_gdxysint(float *zout, float *zin, int32_t gdin, float *x, float *y, int32_t npts)
{
    if (!grid_ok(gdin)) return -1;
    if (need_device("c_gdxysint")) return -1;
    if (npts <= 0) return 0;
    /* the calling thread's grow-only staging buffers (four hipMalloc / hipFree pairs per call cost more than the copies) */
    size_t nin = (size_t)G[gdin].ni * G[gdin].nj, np = (size_t)npts;
    float *d_in = stage(0, nin), *d_out = stage(1, np), *d_xy = stage(2, 2 * np);
    if (!d_in || !d_out || !d_xy) return -1;
    if (ezhip_h2d(d_in, zin, sizeof(float) * nin) || ezhip_h2d(d_xy, x, sizeof(float) * np) || ezhip_h2d(d_xy + np, y, sizeof(float) * np)) return -1;
    int rc = c_gdxysint_dev(d_out, d_in, gdin, d_xy, d_xy + np, npts);
    if (rc == 0 && ezhip_d2h(zout, d_out, sizeof(float) * np)) rc = -1;
    if (ezhip_sync()) rc = -1;
    return rc;
}
int32_t c_gdxysval(int32_t gdin, float *zout, float *zin, float *x, float *y, int32_t n) { return c_gdxysint(zout, zin, gdin, x, y, n); }

/* interpolation at caller-supplied lat/lon points: locate (c_gdxyfll_orig) then c_gdxysval; gdllsval.c:33-55 returns 0 */
int32_t c_gdllsval(int32_t gdid, float *zout, float *zin, float *lat, float *lon, int32_t n)
{
    if (!grid_ok(gdid) || n < 0) return -1;
    float *x = (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1)), *y = (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    int rc = c_gdxyfll_orig(gdid, x, y, lat, lon, n);
    if (rc >= 0) rc = c_gdxysval(gdid, zout, zin, x, y, n);
    free(x); free(y);
    return rc < 0 ? rc : 0;
}
/* vector values at x,y: two scalar interpolations, no rotation (gdxyvval.c:89-102) */
int32_t c_gdxyvval(int32_t gdin, float *uuout, float *vvout, float *uuin, float *vvin, float *x, float *y, int32_t n)
{
    /* gdxyvval.c:103-113: u with symmetrie = SYM, v with ANTISYM -- the mirror image of a hemispheric source carries v with its sign changed */
    int rc = c_gdxysint(uuout, uuin, gdin, x, y, n);
    if (rc < 0) return rc;
    t_symmetrie = 0;
    rc = c_gdxysint(vvout, vvin, gdin, x, y, n);
    t_symmetrie = 1;
    return rc < 0 ? rc : 0;
}
/* gdllvval.c:34-57: locate, then c_gdxyvval */
int32_t c_gdllvval(int32_t gdid, float *uuout, float *vvout, float *uuin, float *vvin, float *lat, float *lon, int32_t n)
{
    if (!grid_ok(gdid) || n < 0) return -1;
    float *x = (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1)), *y = (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    int rc = c_gdxyfll_orig(gdid, x, y, lat, lon, n);
    if (rc >= 0) rc = c_gdxyvval(gdid, uuout, vvout, uuin, vvin, x, y, n);
    free(x); free(y);
    return rc < 0 ? rc : 0;
}

int32_t c_gdxyfll_dev(int32_t gd, float *d_x, float *d_y, const float *d_lat, const float *d_lon, int32_t n)
{
    if (!grid_ok(gd)) return -1;
    if (need_device("c_gdxyfll")) return -1;
    ezh_grid *g = &G[gd];
    if (ensure_grid_dev(g)) return -1;
    ezhip_locate_plan lp;
    fill_locate_plan(g, &lp);
    return ezhip_locate(&lp, d_x, d_y, d_lat, d_lon, n, 1, 0) ? -1 : 0;
}

/* ------------------------------------------------------------------------------------------ */
/* vector interpolation: c_ezuvint_orig, ezuvint.c:51-94                                         */
/* ------------------------------------------------------------------------------------------ */
static const float RDTODG = 57.295779513082f, DGTORD = 1.7453292519943e-2f;

static void h_llwfgdw(float *z1, float *z2, const float *xlon, int n, char t, float xg4)
{   /* ez_llwfgdw.inc:22-170 (N, S, and the lat-lon family) */
    for (int i = 0; i < n; i++) {
        float uu = z1[i], vv = z2[i], spd = sqrtf(uu * uu + vv * vv), dir;
        if (spd == 0.0f) dir = 0.0f;
        else if (t == 'N') dir = (uu == 0.0f) ? ((vv >= 0.0f) ? xlon[i] + xg4 - 90.0f : xlon[i] + xg4 + 90.0f) : xlon[i] + xg4 - RDTODG * atan2f(vv, uu);
        else if (t == 'S') dir = (uu == 0.0f) ? ((vv >= 0.0f) ? 90.0f - xlon[i] + xg4 : 270.0f - xlon[i] + xg4) : 180.0f - xlon[i] + xg4 - RDTODG * atan2f(vv, uu);
        else dir = (uu == 0.0f) ? ((vv >= 0.0f) ? 180.0f : 0.0f) : 270.0f - RDTODG * atan2f(vv, uu);
        dir = fmodf(fmodf(dir, 360.0f) + 360.0f, 360.0f);
        z1[i] = spd; z2[i] = dir;
    }
}
static void h_gdwfllw(float *z1, float *z2, const float *xlon, int n, char t, float xg4)
{   /* ez_gdwfllw.inc:23-137 */
    for (int i = 0; i < n; i++) {
        float psi = t == 'N' ? xlon[i] + xg4 - z2[i] : t == 'S' ? 180.0f - xlon[i] + xg4 - z2[i] : 270.0f - z2[i];
        float u = cosf(psi * DGTORD) * z1[i], v = sinf(psi * DGTORD) * z1[i];
        z1[i] = u; z2[i] = v;
    }
}
static float h_poleval(const float *z, int ni, const ezh_grid *g)
{   /* ez_calcpoleval.inc:21-48 (host: used on O(ni) synthetic wind rows only) */
    float p = 0.0f;
    if (g->grtyp == 'Z' && g->grref == 'E') {
        for (int i = 0; i + 1 < ni; i++) p = p + z[i] * (g->ax[i + 1] - g->ax[i]);
        if ((g->ax[ni - 1] - g->ax[0]) != 0.0f) p = p / (g->ax[ni - 1] - g->ax[0]);
        return p;
    }
    for (int i = 0; i < ni; i++) p = p + z[i];
    return p / (1.0f * (float)ni);
}
/* lat/lon of one source row: c_gdllfxy_orig, gdllfxy.c:245-390 */
static void h_row_latlon(const ezh_grid *g, float *lat, float *lon, int row1 /* 1-based */)
{
    int ni = g->ni;
    float y = (float)(1.0 * row1);
    for (int i = 0; i < ni; i++) {
        float x = (float)(1.0 * (i + 1));
        if (g->grtyp == 'A' || g->grtyp == 'B' || g->grtyp == 'L') {
            lat[i] = (float)(((double)y - 1.0) * (double)g->xg[2] + (double)g->xg[0]);
            lon[i] = (float)(((double)x - 1.0) * (double)g->xg[3] + (double)g->xg[1]);
            lon[i] = (float)fmod((double)(lon[i] + 360.0), 360.0);
        } else if (g->grtyp == 'E') {
            float dlat = (float)(180.0 / g->nj), dlon = (float)(360.0 / (g->ni - 1));
            float swlat = (float)(-90.0 + 0.5 * (double)dlat);
            lon[i] = (float)(((double)x - 1.0) * (double)dlon + 0.0);
            lat[i] = (float)(((double)y - 1.0) * (double)dlat + (double)swlat);
        } else {
            int ix = imin(imax((int)x - 1, 0), g->ni - 2), iy = imin(imax((int)y - 1, 0), g->j2 - 2);
            float dx = g->ax[ix + 1] - g->ax[ix], dy = g->ay[iy + 1] - g->ay[iy];
            float tx = (float)((double)g->ax[ix] + (((double)x - 1.0 - ix) * (double)dx));
            float ty = (float)((double)g->ay[iy] + (((double)y - 1.0 - iy) * (double)dy));
            if (g->grref == 'L') {
                lat[i] = ty * g->xgref[2] + g->xgref[0];
                lon[i] = tx * g->xgref[3] + g->xgref[1];
                lon[i] = (float)fmod((double)(lon[i] + 360.0), 360.0);
            } else { lat[i] = ty; lon[i] = tx; }   /* rotated-frame coordinates; converted below */
        }
    }
    if (g->grtyp == 'E') { float *a = (float *)malloc(sizeof(float) * ni), *b = (float *)malloc(sizeof(float) * ni); memcpy(a, lon, sizeof(float) * ni); memcpy(b, lat, sizeof(float) * ni); h_rotate(lon, lat, a, b, ni, g->xg, 0); free(a); free(b); }
    if (g->grtyp == 'Z' && g->grref == 'E') { float *a = (float *)malloc(sizeof(float) * ni), *b = (float *)malloc(sizeof(float) * ni); memcpy(a, lon, sizeof(float) * ni); memcpy(b, lat, sizeof(float) * ni); h_rotate(lon, lat, a, b, ni, g->xgref, 0); free(a); free(b); }
}

/* longitudes of the last (north) / first source row as ez_calcnpolarwind sees them (for Z-on-E: rotated frame) */
static void polar_row_lon(const ezh_grid *g, float *plon, float *tmp_lat, int north)
{
    int ni = g->ni;
    h_row_latlon(g, tmp_lat, plon, north ? g->nj : 1);
    if (g->grtyp == 'Z' && g->grref == 'E') {
        float *a = (float *)malloc(sizeof(float) * ni), *b = (float *)malloc(sizeof(float) * ni);
        memcpy(a, plon, sizeof(float) * ni); memcpy(b, tmp_lat, sizeof(float) * ni);
        h_rotate(plon, tmp_lat, a, b, ni, g->xgref, 1);
        free(a); free(b);
    }
}

/* ez_calcnpolarwind.c:28-138 / ez_calcspolarwind.c: synthetic pole row of (u,v) from the last/first source row */
static void h_polar_wind(const ezh_grid *g, float *pu, float *pv, const float *urow, const float *vrow, int north)
{
    int ni = g->ni;
    float *plat = (float *)malloc(sizeof(float) * ni), *plon = (float *)malloc(sizeof(float) * ni);
    float *spd = (float *)malloc(sizeof(float) * ni), *wd = (float *)malloc(sizeof(float) * ni);
    h_row_latlon(g, plat, plon, north ? g->nj : 1);
    if (g->grtyp == 'Z' && g->grref == 'E') {
        float *a = (float *)malloc(sizeof(float) * ni), *b = (float *)malloc(sizeof(float) * ni);
        memcpy(a, plon, sizeof(float) * ni); memcpy(b, plat, sizeof(float) * ni);
        h_rotate(plon, plat, a, b, ni, g->xgref, 1);
        free(a); free(b);
    }
    memcpy(spd, urow, sizeof(float) * ni); memcpy(wd, vrow, sizeof(float) * ni);
    h_llwfgdw(spd, wd, plon, ni, 'A', 0.f);
    char hs = north ? 'N' : 'S';
    int i1, i2, i3, i4; float x1, x2, x3, x4;
    h_cxgaig(hs, &i1, &i2, &i3, &i4, 0.0f, 0.0f, 1000.0f, 0.0f);
    h_cigaxg(hs, &x1, &x2, &x3, &x4, i1, i2, i3, i4);
    memcpy(pu, spd, sizeof(float) * ni); memcpy(pv, wd, sizeof(float) * ni);
    h_gdwfllw(pu, pv, plon, ni, hs, x4);
    float s0 = h_poleval(pu, ni, g), w0 = h_poleval(pv, ni, g), zero = 0.0f;
    h_llwfgdw(&s0, &w0, &zero, 1, hs, x4);
    spd[0] = s0; wd[0] = w0;
    for (int i = 1; i < ni; i++) { wd[i] = north ? wd[0] + plon[i] : wd[0] - plon[i]; spd[i] = spd[0]; }
    wd[0] = wd[0] + plon[0];
    memcpy(pu, spd, sizeof(float) * ni); memcpy(pv, wd, sizeof(float) * ni);
    h_gdwfllw(pu, pv, plon, ni, 'A', 0.f);
    free(plat); free(plon); free(spd); free(wd);
}

/* wd_only: c_ezwdint_orig (ezwdint.c:62-113) = the same path stopped after c_gdwdfuv */
static int32_t uvint_body(float *d_uuout, float *d_vvout, const float *d_uuin, const float *d_vvin, int wd_only);
/* extrap_degree = abort with target points outside the source: each of the two c_ezsint calls inside c_ezuvint_orig prints the abort message and
 * returns -1 from ez_corrval BEFORE anything is corrected (ez_corrval.c:54-60), c_ezuvint_orig looks for the value 2 only (ezuvint.c:68-74) and
 * carries on: ez_corrvec finds no strip zones (a set with outside points has none: ez_defzones.c:46-113), the wind chain runs, 0 is returned.
 * I.e. the answer of the same call with the polar correction off -- found by tools/fuzz_vs_ref2.py. */
static int32_t uvint_impl(float *d_uuout, float *d_vvout, const float *d_uuin, const float *d_vvin, int wd_only)
{
    ezh_set *s0 = current_set(wd_only ? "c_ezwdint" : "c_ezuvint");
    if (s0 && O.polar_correction == 1 && O.degre_extrap == XT_ABORT && s0->extrap && s0->gdin != s0->gdout &&
        G[s0->gdin].grtyp != 'U' && G[s0->gdout].grtyp != 'U' && !need_device("c_ezuvint")) {
        if (!s0->sep_capable) {
            pthread_mutex_lock(&g_plan_mtx);
            int e = ensure_points(s0);
            pthread_mutex_unlock(&g_plan_mtx);
            if (e) return -1;
        }
        if (s0->have_dehors) {
            for (int k = 0; k < 2; k++) {
                fprintf(stderr, "<ez_corrval> There are points on the destination grid that lie outside the source grid\n");
                fprintf(stderr, "<ez_corrval> aborting at your request!\n\n\n");
            }
            O.polar_correction = 0;
            int32_t rc = uvint_body(d_uuout, d_vvout, d_uuin, d_vvin, wd_only);
            O.polar_correction = 1;
            return rc < 0 ? rc : 0;
        }
    }
    return uvint_body(d_uuout, d_vvout, d_uuin, d_vvin, wd_only);
}
static int32_t uvint_body(float *d_uuout, float *d_vvout, const float *d_uuin, const float *d_vvin, int wd_only)
{
    ezh_set *s = current_set(wd_only ? "c_ezwdint" : "c_ezuvint");
    if (!s) return -1;
    if (need_device("c_ezuvint") || device_error("c_ezuvint")) return -1;
    if (G[s->gdout].grtyp == 'U') {                        /* ezyyuvint.c:88-106 / :216-300: the two Z-on-E subgrids one after the other */
        const ezh_grid *gu = &G[s->gdout];
        const size_t nsub = (size_t)G[gu->sub[0]].ni * G[gu->sub[0]].nj;
        const int save_in = cur_gdin, save_out = cur_gdout, gdin_ = s->gdin, sub0 = gu->sub[0], sub1 = gu->sub[1];
        int rc = 0;
        for (int k = 0; k < 2; k++) {
            cur_gdin = gdin_; cur_gdout = k ? sub1 : sub0;
            int r = uvint_impl(d_uuout + k * nsub, d_vvout + k * nsub, d_uuin, d_vvin, wd_only);
            if (r < 0) { rc = r; break; }
            if (r == 2) rc = 2;
        }
        cur_gdin = save_in; cur_gdout = save_out;
        return rc;
    }
    if (G[s->gdin].grtyp == 'U') return yy_uvint(s, d_uuout, d_vvout, d_uuin, d_vvin, wd_only);
    ezh_grid *gi = &G[s->gdin], *go = &G[s->gdout];
    if (src_hemi(gi) || grid_yinv(gi)) { fprintf(stderr, "<c_ezuvint> hemispheric / y-inverted source grids are outside the MI355X hot-path scope for winds\n"); return -1; }
    if (go->grtyp == 'E') {                                 /* regular 'E' targets: the reference reads uninitialised memory */
        fprintf(stderr, "<c_ezuvint> '%c' target grids are outside the MI355X hot-path scope for winds\n", go->grtyp);
        return -1;
    }
    int ni = gi->ni, nj = gi->nj, polar = O.polar_correction == 1;
    /* one grid on both sides: the two c_ezsint calls of c_ezuvint_orig (ezuvint.c:68-71) copy their fields (ezsint.c: identical grids, return 1) and
     * never define the set's zones, so ez_corrvec finds none; the wind chain still runs on the copies (ierc stays 0) */
    const int same = s->gdin == s->gdout;
    const float *pun = NULL, *pus = NULL, *pvn = NULL, *pvs = NULL;
    if (polar && !s->extrap && !same) {
        /* synthetic polar wind rows (ez_calcnpolarwind.c): one small kernel, no host round trip.  The row longitudes
         * and the polar-stereographic xg4 depend on the grid only: computed once on the host. */
        {
            const int np = t_pairs.n > 1 ? t_pairs.n : 1;
            if (!s->d_prow || s->prow_pairs < np) {             /* (ezhip_free waits for the device: no launch still reads the rows that go) */
                ezhip_free(s->d_prow);
                s->d_prow = (float *)ezhip_malloc(sizeof(float) * 4 * (size_t)ni * (size_t)np);
                s->prow_pairs = s->d_prow ? np : 0;
            }
        }
        if (!s->d_prow) return -1;
        if (!gi->d_plon2) {
            float *pl = (float *)malloc(sizeof(float) * 3 * (size_t)ni);
            polar_row_lon(gi, pl, pl + 2 * ni, 1); polar_row_lon(gi, pl + ni, pl + 2 * ni, 0);
            gi->d_plon2 = (float *)upload(pl, sizeof(float) * 2 * (size_t)ni);
            free(pl);
            for (int k = 0; k < 2; k++) {
                int i1, i2, i3, i4; float x1, x2, x3, x4;
                h_cxgaig(k == 0 ? 'N' : 'S', &i1, &i2, &i3, &i4, 0.0f, 0.0f, 1000.0f, 0.0f);
                h_cigaxg(k == 0 ? 'N' : 'S', &x1, &x2, &x3, &x4, i1, i2, i3, i4);
                gi->xg4_pole[k] = x4;
            }
            if (!gi->d_plon2) return -1;
        }
        int weighted = (gi->grtyp == 'Z' && gi->grref == 'E');
        if (weighted && ensure_grid_dev(gi)) return -1;
        if (getenv("EZHIP_POLAR_WIND_HOST")) {       /* development / tests: the host restatement with the host libm */
            float *rows = (float *)malloc(sizeof(float) * 8 * (size_t)ni);
            float *un = rows, *vn = rows + ni, *us = rows + 2 * ni, *vs = rows + 3 * ni, *out = rows + 4 * ni;
            ezhip_d2h(un, d_uuin + (size_t)(nj - 1) * ni, sizeof(float) * ni); ezhip_d2h(vn, d_vvin + (size_t)(nj - 1) * ni, sizeof(float) * ni);
            ezhip_d2h(us, d_uuin, sizeof(float) * ni); ezhip_d2h(vs, d_vvin, sizeof(float) * ni);
            if (ezhip_sync()) { free(rows); return -1; }
            h_polar_wind(gi, out, out + 2 * ni, un, vn, 1);          /* out: [u_n, u_s, v_n, v_s] */
            h_polar_wind(gi, out + ni, out + 3 * ni, us, vs, 0);
            ezhip_h2d(s->d_prow, out, sizeof(float) * 4 * (size_t)ni);
            if (ezhip_sync()) { free(rows); return -1; }
            free(rows);
        } else {
            /* on the side stream: only the special points (k_pts_special, or the special rows of the separable kernel) read
             * these rows, the main kernels of both components overlap the 44 us of sequential pole sums */
            int per_point = 0;
            if (!same && (O.degre_interp == DEG_NEAREST || O.degre_interp == DEG_LINEAR || O.degre_interp == DEG_CUBIC) && !getenv("EZHIP_NO_PTS2") && !WIND_EXACT()) {
                pthread_mutex_lock(&g_plan_mtx);
                per_point = choose_mode(s, O.degre_interp, polar) == 2;
                pthread_mutex_unlock(&g_plan_mtx);
            }
            if (per_point) {       /* the per-point pair kernel takes the job along (two producer blocks of its own launch, or its own side-stream launch) */
                t_pwjob.out = s->d_prow; t_pwjob.plon2 = gi->d_plon2; t_pwjob.xg4_n = gi->xg4_pole[0]; t_pwjob.xg4_s = gi->xg4_pole[1];
                t_pwjob.weighted = weighted; t_pwjob.ax = gi->d_ax; t_pwjob.active = 1;
            } else {
                if (ezhip_side_begin()) return -1;
                /* (round 6) the standalone launch -- on the side stream, beside the main kernels -- always uses the C library's trig (libm_exact.h): winds between unrotated grids then equal
                 * the reference bit for bit in the default mode too (219 of 219 random pairs; 211 with the device library's functions in here) */
                int prc = ezhip_polar_wind(s->d_prow, d_uuin, d_vvin, gi->d_plon2, ni, nj, gi->xg4_pole[0], gi->xg4_pole[1], weighted, gi->d_ax, getenv("EZHIP_POLAR_WIND_OCML") ? WIND_EXACT() : 1);
                if (ezhip_side_end() || prc) return -1;
            }
        }
        pun = s->d_prow; pus = s->d_prow + ni; pvn = s->d_prow + 2 * ni; pvs = s->d_prow + 3 * ni;
    }
    /* the wind chain of the pair first: where it is a per-point matrix (below) the per-point kernel applies it as it stores */
    {
        int degree_ = O.degre_interp;
        pthread_mutex_lock(&g_plan_mtx);
        int erc_ = 0;
        if (degree_ == DEG_NEAREST || degree_ == DEG_LINEAR || degree_ == DEG_CUBIC)
            if (choose_mode(s, degree_, polar) == 2) erc_ = ensure_points(s);          /* the locate may still edit the target's cached longitudes */
        pthread_mutex_unlock(&g_plan_mtx);
        if (erc_) return -1;
    }
    if (ensure_coords_dev(go)) return -1;
    ezhip_wind_plan wp;
    memset(&wp, 0, sizeof(wp));
    wp.separable = go->separable;
    wp.wd_only = wd_only;
    wp.src_rotated = (gi->grtyp == 'E' || (gi->grtyp == 'Z' && gi->grref == 'E'));
    wp.fast_trig = getenv("EZHIP_WIND_FAST_TRIG") != NULL;
    if (gi->grtyp == 'N' || gi->grtyp == 'S') { wp.src_ps = gi->grtyp == 'N' ? 1 : 2; wp.src_xg4 = gi->xg[3]; }
    if (go->grtyp == 'N' || go->grtyp == 'S') { wp.dst_ps = go->grtyp == 'N' ? 1 : 2; wp.dst_xg4 = go->xg[3]; }
    if (go->grtyp == 'Z' && go->grref == 'E') { float ri_[9]; wp.dst_rotated = 1; h_crot(wp.r_dst, ri_, go->xgref[1], go->xgref[0], go->xgref[3], go->xgref[2]); }
    if (wp.src_rotated) { const float *xg = gi->grtyp == 'E' ? gi->xg : gi->xgref; h_crot(wp.r, wp.ri, xg[1], xg[0], xg[3], xg[2]); }
    else if (gi->grtyp == 'Z' && gi->grref != 'L') return -1;
    if (gi->grtyp == '!' || go->grtyp == '!') {
        /* Lambert legs (gdwdfuv.c default branch -> ez_llwfgdw '!' -> ez_lamb_llwfgdw; gduvfwd.c -> ez_gdwfllw '!' -> ez_lamb_gdwfllw): a rotation per point whose
         * angle depends on the point's longitude only; {cos, sin} from the host's REAL chain, once per set */
        const float *lon_h = go->separable ? go->lon1d : go->lon2d;
        const int nl = go->separable ? go->ni : go->ni * go->nj;
        pthread_mutex_lock(&g_plan_mtx);
        if (gi->grtyp == '!' && !s->d_lamb_cs[0]) s->d_lamb_cs[0] = lamb_wind_cs_dev(gi, lon_h, nl, 1);
        if (go->grtyp == '!' && !s->d_lamb_cs[1]) s->d_lamb_cs[1] = lamb_wind_cs_dev(go, lon_h, nl, 0);
        const int bad = (gi->grtyp == '!' && !s->d_lamb_cs[0]) || (go->grtyp == '!' && !s->d_lamb_cs[1]);
        pthread_mutex_unlock(&g_plan_mtx);
        if (bad) return -1;
        if (gi->grtyp == '!') wp.src_lamb_cs = s->d_lamb_cs[0];
        if (go->grtyp == '!') wp.dst_lamb_cs = s->d_lamb_cs[1];
    }
    if (wp.src_rotated && wp.separable) {
        if (!go->d_windtrig) {
            size_t nt = 2 * ((size_t)go->ni + go->nj);               /* doubles, then the same count of floats */
            go->d_windtrig = (double *)ezhip_malloc(sizeof(double) * nt + sizeof(float) * nt);
            if (!go->d_windtrig) return -1;
            float *tf = (float *)(go->d_windtrig + nt);
            if (ezhip_wind_trig_tables(go->d_windtrig, go->d_windtrig + 2 * (size_t)go->ni, tf, tf + 2 * (size_t)go->ni,
                                       go->d_lat, go->d_lon, go->ni, go->nj)) return -1;
        }
        wp.lon_trig = go->d_windtrig; wp.lat_trig = go->d_windtrig + 2 * (size_t)go->ni;
        {
            const float *tf = (const float *)(go->d_windtrig + 2 * ((size_t)go->ni + go->nj));
            wp.lon_trigf = tf; wp.lat_trigf = tf + 2 * (size_t)go->ni;
        }
    }
    /* components out through a rotated frame (E / Z-on-E source or target): the chain is a rotation per point that depends on the grid
     * pair only -- built once per set from the chain itself (ezhip_wind_matrix), then two multiply-adds per component and call: inside
     * k_pts2 as it stores (per-point sets), or k_wind_apply after the separable launches (k_wind_rotate: 112 us per cfg3 pair) */
    const void *d_M = NULL;
    int m_half = 0;
    if (!wd_only && (wp.src_rotated || wp.dst_rotated) && !getenv("EZHIP_WIND_NO_MATRIX") && !WIND_EXACT() &&
        20 * (size_t)go->ni * go->nj <= ((size_t)3 << 30)) {           /* 16 + 4 bytes per target point, kept with the set: up to 3 GiB, beyond that the chain runs per call */
        pthread_mutex_lock(&g_plan_mtx);
        int mrc = 0;
        if (!s->d_windM) {
            s->d_windM = ezhip_malloc(20 * (size_t)go->ni * go->nj + 16);
            if (!s->d_windM || ezhip_wind_matrix(&wp, s->d_windM, go->d_lat, go->d_lon, go->ni, go->nj, &s->windM_dev)) { ezhip_free(s->d_windM); s->d_windM = NULL; mrc = -1; }
            /* the chain is a rotation by construction (speed kept, direction turned); its REAL roundings leave a - d and b + c at a few 1e-7.  Where that holds
             * at every point the rotation is read as ONE packed word per point (rot_pack in the kernels): a quarter of the matrix bytes per call.  Anything else (a non-finite coefficient, a frame that is no rotation)
             * keeps the four coefficients */
            s->windM_half = !mrc && s->windM_dev <= 4.0e-6f;
            if (getenv("EZHIP_WIND_MATRIX_REPORT")) fprintf(stderr, "<ezhip> wind matrix of set (%d, %d): max distance from a rotation %.3g -> %s\n", s->gdin, s->gdout, (double)s->windM_dev, s->windM_half ? "packed rotation" : "four coefficients");
        }
        m_half = s->windM_half && !getenv("EZHIP_WIND_FULL_MATRIX");
        d_M = m_half ? (const void *)((const char *)s->d_windM + 16 * (size_t)go->ni * go->nj) : s->d_windM;
        pthread_mutex_unlock(&g_plan_mtx);
        if (mrc) return dev_fail("the wind rotation matrix");
    }
    int r1 = 0, r2 = 0;
    if (same) {
        const size_t nb = sizeof(float) * (size_t)ni * nj;
        if ((d_uuout != d_uuin && ezhip_d2d(d_uuout, d_uuin, nb)) || (d_vvout != d_vvin && ezhip_d2d(d_vvout, d_vvin, nb))) return -1;
    }
    if (!same) r1 = r2 = run_pair_pts(s, d_uuout, d_vvout, d_uuin, d_vvin, pun, pus, pvn, pvs, getenv("EZHIP_WIND_NO_FUSE") ? NULL : d_M, m_half, wp.dst_rotated);
    if (t_pwjob.active) {                                     /* the pair kernel did not take the polar wind rows along (set not on its path after all) */
        t_pwjob.active = 0;
        if (r1 != -1) {
            if (ezhip_side_begin()) return -1;
            int prc = ezhip_polar_wind(t_pwjob.out, d_uuin, d_vvin, t_pwjob.plon2, ni, nj, t_pwjob.xg4_n, t_pwjob.xg4_s, t_pwjob.weighted, t_pwjob.ax, getenv("EZHIP_POLAR_WIND_OCML") ? 0 : 1);
            if (ezhip_side_end() || prc) return -1;
        }
    }
    if (r1 == -1) return -1;
    const int fused = !same && r1 != -2 && d_M && !getenv("EZHIP_WIND_NO_FUSE");
    if (r1 == -2) {                                          /* separable set: one launch per component */
        r1 = run_field(s, d_uuout, d_uuin, 1, pun, pus);
        if (r1 < 0) return r1;
        r2 = run_field(s, d_vvout, d_vvin, 1, pvn, pvs);
        if (r2 < 0) return r2;
    }
    if (ezhip_side_join()) return -1;                        /* nothing may outlive the call on the side stream */
    if (d_M) { if (!fused && ezhip_wind_apply(d_M, m_half, d_uuout, d_vvout, (size_t)go->ni * go->nj, wp.dst_rotated)) return -1; }
    else if (ezhip_wind_rotate(&wp, d_uuout, d_vvout, go->d_lat, go->d_lon, go->ni, go->nj)) return -1;
    return (r1 == 2 || r2 == 2) ? 2 : 0;
}

int32_t c_ezuvint_dev(float *d_uuout, float *d_vvout, const float *d_uuin, const float *d_vvin) { return uvint_impl(d_uuout, d_vvout, d_uuin, d_vvin, 0); }
int32_t c_ezwdint_dev(float *d_spdout, float *d_dirout, const float *d_uuin, const float *d_vvin) { return uvint_impl(d_spdout, d_dirout, d_uuin, d_vvin, 1); }

/* npairs wind pairs of the current grid set, device resident: pair f's components at d_uuin / d_vvin + f * ni_in * nj_in, its results at d_uuout / d_vvout +
 * f * ni_out * nj_out.  The results are those of npairs c_ezuvint_dev calls, bit for bit (ezuvint.c:51-94 per pair).  Where the set runs its pairs from LDS-staged
 * stencil windows (bicubic from a rotated source with wrap, from the set's second call on: k_uvt) ONE launch takes all pairs: x, y and the rotation of a target
 * point -- 12 of the 34 bytes a pair moves per point -- are read once per batch, the tile table and the axis records too; any other set, option or state: pair
 * by pair.  Returns 0 / 2 like c_ezuvint_dev (2: points outside the source), -1 on error. */
static int pair_batch_ready(ezh_set *s)
{
    if (G[s->gdin].grtyp == 'U' || G[s->gdout].grtyp == 'U' || s->gdin == s->gdout) return 0;
    const int degree = O.degre_interp, polar = O.polar_correction == 1;
    if (degree != DEG_CUBIC) return 0;
    if (WIND_EXACT() || getenv("EZHIP_POLAR_WIND_HOST") || getenv("EZHIP_NO_PTS2") || getenv("EZHIP_NO_UVT") || getenv("EZHIP_WIND_NO_FUSE") || getenv("EZHIP_WIND_NO_MATRIX") ||
        getenv("EZHIP_WIND_FULL_MATRIX") || getenv("EZHIP_UVT_NO_STREAMS") || getenv("EZHIP_NO_SPEC_CACHE") || getenv("EZHIP_NO_PAIR_BATCH")) return 0;
    if (polar && s->extrap && s->have_dehors) return 0;          /* (a fill value per pair: pair by pair) */
    const int zones = !polar ? 0 : (s->extrap ? 2 : 1);
    const int key = 1 | zones << 1 | (O.degre_extrap & 0xFF) << 4 | degree << 12;
    pthread_mutex_lock(&g_plan_mtx);
    const int ok = choose_mode(s, degree, polar) == 2 && s->cspec_key == key && s->d_uvt_tiles && s->d_uvt_streams && s->uvt_shape == 3232 && s->d_windM && s->windM_half &&
                   s->uvt_streams_M == (const void *)((const char *)s->d_windM + 16 * (size_t)G[s->gdout].ni * G[s->gdout].nj);
    pthread_mutex_unlock(&g_plan_mtx);
    return ok;
}
int32_t c_ezuvint_batch_dev(float *d_uuout, float *d_vvout, const float *d_uuin, const float *d_vvin, int32_t npairs)
{
    if (npairs < 1 || !d_uuout || !d_vvout || !d_uuin || !d_vvin) return -1;
    ezh_set *s = current_set("c_ezuvint_batch_dev");
    if (!s) return -1;
    if (need_device("c_ezuvint_batch_dev")) return -1;
    const size_t nin = (size_t)G[s->gdin].ni * G[s->gdin].nj, nout = (size_t)G[s->gdout].ni * G[s->gdout].nj;
    int first = 0, rc = 0;
    /* the set's first pair builds its caches (special points, tile table, tile-ordered streams); the batch form needs them */
    if (npairs >= 2 && !pair_batch_ready(s)) {
        const int r = c_ezuvint_dev(d_uuout, d_vvout, d_uuin, d_vvin);
        if (r < 0) return r;
        if (r == 2) rc = 2;
        first = 1;
    }
    if (npairs - first >= 2 && pair_batch_ready(s)) {
        t_pairs.n = npairs - first; t_pairs.refused = 0; t_pairs.in_stride = nin; t_pairs.out_stride = nout;
        const int r = uvint_impl(d_uuout + first * nout, d_vvout + first * nout, d_uuin + first * nin, d_vvin + first * nin, 0);
        const int refused = t_pairs.refused;
        t_pairs.n = 0; t_pairs.refused = 0;
        if (!refused) return r < 0 ? r : (r == 2 || rc == 2) ? 2 : 0;
    }
    for (int f = first; f < npairs; f++) {
        const int r = c_ezuvint_dev(d_uuout + f * nout, d_vvout + f * nout, d_uuin + f * nin, d_vvin + f * nin);
        if (r < 0) return r;
        if (r == 2) rc = 2;
    }
    return rc;
}

static int32_t uvint_host(float *uuout, float *vvout, float *uuin, float *vvin, int wd_only)
{
    ezh_set *s = current_set(wd_only ? "c_ezwdint" : "c_ezuvint");
    if (!s) return -1;
    if (need_device(wd_only ? "c_ezwdint" : "c_ezuvint")) return -1;
    size_t nin = (size_t)G[s->gdin].ni * G[s->gdin].nj, nout = (size_t)G[s->gdout].ni * G[s->gdout].nj;
    float *d_in = stage(0, nin), *d_out = stage(1, nout), *d_in2 = stage(2, nin), *d_out2 = stage(3, nout);
    if (!d_in || !d_out || !d_in2 || !d_out2) return -1;
    if (ezhip_h2d(d_in, uuin, sizeof(float) * nin) || ezhip_h2d(d_in2, vvin, sizeof(float) * nin)) return -1;
    int rc = uvint_impl(d_out, d_out2, d_in, d_in2, wd_only);
    if (rc < 0) return rc;
    if (ezhip_d2h(uuout, d_out, sizeof(float) * nout) || ezhip_d2h(vvout, d_out2, sizeof(float) * nout) || ezhip_sync()) return -1;
    return rc;
}
int32_t c_ezuvint(float *uuout, float *vvout, float *uuin, float *vvin) { return uvint_host(uuout, vvout, uuin, vvin, 0); }
int32_t c_ezwdint(float *spdout, float *dirout, float *uuin, float *vvin) { return uvint_host(spdout, dirout, uuin, vvin, 1); }

/* ------------------------------------------------------------------------------------------ */
/* Fortran twins (rpnmacros.h:21 f77name; scalars by reference; hidden string lengths last)       */
/* ------------------------------------------------------------------------------------------ */
/* small public entry points around the same kernels                                            */
/* ------------------------------------------------------------------------------------------ */
/* c_ezgdef (ezgdef.c:42-95): the memory form is c_ezgdef_fmem; grids described by an FST file are out of scope */
int32_t c_ezgdef(int32_t ni, int32_t nj, char *grtyp, char *grref, int32_t ig1, int32_t ig2, int32_t ig3, int32_t ig4, float *ax, float *ay)
{
    if (grtyp[0] == '#') { fprintf(stderr, "The '#' grid type is not supported with ezgdef.\nPlease use ezgdef_ffile or ezgdef_fmem\n"); return -1; }
    if ((grtyp[0] == 'Y' || grtyp[0] == 'Z') && grref && (!strcmp(grref, "FILE") || !strcmp(grref, "file"))) {
        fprintf(stderr, "<c_ezgdef> grid descriptors read from an FST file are outside the MI355X hot-path scope; use c_ezgdef_fmem\n");
        return -1;
    }
    return c_ezgdef_fmem(ni, nj, grtyp, grref, ig1, ig2, ig3, ig4, ax, ay);
}
int32_t c_gdxpncf(int32_t gdin, int32_t *i1, int32_t *i2, int32_t *j1, int32_t *j2)                                           /* gdxpncf.c:33-50 */
{
    if (!grid_ok(gdin)) return -1;
    if (G[gdin].nsub > 0) { fprintf(stderr, "<gdxpncf> This operation is not supported for 'U' grids.\n"); return -1; }
    *i1 = G[gdin].i1; *i2 = G[gdin].i2; *j1 = G[gdin].j1; *j2 = G[gdin].j2;
    return 0;
}
int32_t c_ezget_nsubgrids(int32_t gdid) { if (!grid_ok(gdid)) return -1; return G[gdid].nsub ? G[gdid].nsub : 1; }           /* ezget_nsubgrids.c:25-33 */
int32_t c_ezget_subgridids(int32_t gdid, int32_t *subgrid)                                                                    /* ezget_subgridids.c:25-37 */
{
    if (!grid_ok(gdid)) return -1;
    if (G[gdid].nsub == 0) { *subgrid = gdid; return 1; }
    for (int k = 0; k < G[gdid].nsub; k++) subgrid[k] = G[gdid].sub[k];
    return G[gdid].nsub;
}
int32_t c_ezgdef_fll(int32_t ni, int32_t nj, float *lat, float *lon)                                                          /* ezgdef_fll.c:36-51 */
{
    int ig1, ig2, ig3, ig4;
    h_cxgaig('L', &ig1, &ig2, &ig3, &ig4, 0.0f, 0.0f, 1.0f, 1.0f);
    char t[2] = "Y", r[2] = "L";
    return c_ezgdef_fmem(ni, nj, t, r, ig1, ig2, ig3, ig4, lon, lat);
}
/* c_gdxyzfll (gdxyzfll.c:33-80): like c_gdxyfll for the regular types; 'Z' grids: coordinates on the reference grid
 * (ABSOLU: the search in the axes is skipped) */
int32_t c_gdxyzfll(int32_t gdid, float *x, float *y, float *lat, float *lon, int32_t n)
{
    if (!grid_ok(gdid)) return -1;
    ezh_grid *g = &G[gdid];
    switch (g->grtyp) {
    case 'A': case 'B': case 'E': case 'G': case 'L': case 'N': case 'S': case '!': c_gdxyfll_orig(gdid, x, y, lat, lon, n); break;
    case 'Y': fprintf(stderr, "<gdxyzfll>: This operation is not supported for 'Y' grids\n"); break;
    case 'Z':
        if (g->grref == 'L') {
            float lat0, lon0, dlat, dlon;
            h_cigaxg('L', &lat0, &lon0, &dlat, &dlon, g->igref[0], g->igref[1], g->igref[2], g->igref[3]);
            h_llll2gd(x, y, lat, lon, n, lat0, lon0, dlat, dlon, (g->ax[0] < 0.0f) ? -180.0f : 0.0f);
            for (int i = 0; i < n; i++) { x[i] = x[i] - 1.0f; y[i] = y[i] - 1.0f; }
        } else h_rotate(x, y, lon, lat, n, g->xgref, 1);
        break;
    default: break;
    }
    return 0;
}

int32_t c_gdwdfuv(int32_t gdid, float *spd_out, float *wd_out, float *uuin, float *vvin, float *latin, float *lonin, int32_t npts);
/* ez_vllfxy (ez_vllfxy.inc:24-98): polar-stereographic x, y -> lat, lon, in double */
static void h_vllfxy(float *dlat, float *dlon, const float *x, const float *y, int n, float d60, float dgrw, float pi, float pj, int nhem)
{
    const float rdtodg = 57.295779513082f;
    double re = 1.866025 * 6.371e+6 / (double)d60, re2 = re * re;
    double rlat = 0.0, rlon = 0.0;                       /* the Fortran keeps them across iterations (x1 = y1 = 0 sets both) */
    for (int i = 0; i < n; i++) {
        double x1 = (double)(float)(x[i] - pi), y1 = (double)(float)(y[i] - pj);
        if (x1 == 0. && y1 == 0.) { rlat = 90.0; rlon = 0.0; }
        if (x1 == 0.0) rlon = copysign(90.0, y1);
        if (x1 != 0.0) rlon = atan(y1 / x1) * (double)rdtodg;
        if (x1 < 0.0) rlon = rlon + copysign(180.0, y1);
        rlon = rlon - (double)dgrw;
        if (rlon < 0.0) rlon = rlon + 3.6e2;
        double r2 = x1 * x1 + y1 * y1;
        rlat = (re2 - r2) / (re2 + r2);
        rlat = fmax(-1.0, fmin(rlat, 1.0));
        rlat = asin(rlat) * (double)rdtodg;
        if (nhem == 2) { rlat = -rlat; rlon = -rlon; if (rlon < 0.0) rlon = rlon + 360.0; }
        dlat[i] = (float)rlat; dlon[i] = (float)rlon;
    }
}
/* c_gdllfxy_new (gdllfxy.c:92-260), host only: grid coordinates -> lat, lon */
/* newform: c_gdllfxy_new (gdllfxy.c:103-250, what the public c_gdllfxy calls): rows of a 'G' / 'Z' / '#' grid whose ig2 is 1 count from the north
 * (:190-195); 0: c_gdllfxy_orig (:252-), the form the library calls internally */
static int gdllfxy_one(const ezh_grid *g, float *lat, float *lon, const float *x, const float *y, int n, int newform)
{
    switch (g->grtyp) {
    case 'A': case 'B': case 'L':
        for (int i = 0; i < n; i++) {
            lat[i] = (float)(((double)y[i] - 1.0) * (double)g->xg[2] + (double)g->xg[0]);
            lon[i] = (float)(((double)x[i] - 1.0) * (double)g->xg[3] + (double)g->xg[1]);
            lon[i] = (float)fmod((double)lon[i] + 360.0, 360.0);
        }
        return 0;
    case 'E': {
        float *tx = (float *)malloc(sizeof(float) * (size_t)(n + 1)), *ty = (float *)malloc(sizeof(float) * (size_t)(n + 1));
        float dlat = (float)(180.0 / g->nj), dlon = (float)(360.0 / (g->ni - 1));
        float swlat = (float)(-90.0 + 0.5 * (double)dlat), swlon = 0.0f;
        for (int i = 0; i < n; i++) { tx[i] = (float)(((double)x[i] - 1.0) * (double)dlon + (double)swlon); ty[i] = (float)(((double)y[i] - 1.0) * (double)dlat + (double)swlat); }
        h_rotate(lon, lat, tx, ty, n, g->xg, 0);
        free(tx); free(ty);
        return 0;
    }
    case 'N': case 'S':
        h_vllfxy(lat, lon, x, y, n, g->xg[2], g->xg[3], g->xg[0], g->xg[1], g->grtyp == 'N' ? 1 : 2);
        for (int i = 0; i < n; i++) lon[i] = (float)fmod((double)lon[i] + 360.0, 360.0);
        return 0;
    case '!': {                                            /* gdllfxy.c:172-174 (the caller's x, y stay as they are here: the reference overwrites them with projection metres) */
        float *tx = (float *)malloc(sizeof(float) * (size_t)(n + 1)), *ty = (float *)malloc(sizeof(float) * (size_t)(n + 1));
        memcpy(tx, x, sizeof(float) * (size_t)n); memcpy(ty, y, sizeof(float) * (size_t)n);
        h_llflamb(g, lat, lon, tx, ty, n);
        free(tx); free(ty);
        return 0;
    }
    case 'Y': fprintf(stderr, "<gdllfxy>: This operation is not supported for 'Y' grids\n"); return 0;
    case 'Z': case 'G': {
        float *tx = (float *)malloc(sizeof(float) * (size_t)(n + 1)), *ty = (float *)malloc(sizeof(float) * (size_t)(n + 1));
        for (int i = 0; i < n; i++) {
            const float yy = (newform && g->ig[1] == 1) ? (float)((double)g->nj + 1.0 - (double)y[i]) : y[i];
            int indx = (int)x[i] - 1, indy = (int)yy - 1;
            indx = indx < 0 ? 0 : indx; indy = indy < 0 ? 0 : indy;
            indx = indx > g->ni - 2 ? g->ni - 2 : indx; indy = indy > g->j2 - 2 ? g->j2 - 2 : indy;
            float delxx = g->ax[indx + 1] - g->ax[indx];
            tx[i] = (float)((double)g->ax[indx] + (((double)x[i] - 1.0 - indx) * (double)delxx));
            float delyy = g->ay[indy + 1] - g->ay[indy];
            ty[i] = (float)((double)g->ay[indy] + (((double)yy - 1.0 - indy) * (double)delyy));
        }
        if (g->grref == 'E') h_rotate(lon, lat, tx, ty, n, g->xgref, 0);
        else for (int i = 0; i < n; i++) {
            lat[i] = ty[i] * g->xgref[2] + g->xgref[0];                                       /* float arithmetic in the C source (:244-245) */
            lon[i] = tx[i] * g->xgref[3] + g->xgref[1];
            lon[i] = (float)fmod((double)lon[i] + 360.0, 360.0);
        }
        free(tx); free(ty);
        return 0;
    }
    }
    return -1;
}
int32_t c_gdllfxy(int32_t gdid, float *lat, float *lon, float *x, float *y, int32_t n)
{
    if (!grid_ok(gdid)) return -1;
    ezh_grid *g = &G[gdid];
    if (g->nsub > 0) { fprintf(stderr, "<gdllfxy> 'U' grids are outside the MI355X hot-path scope here\n"); return -1; }
    return gdllfxy_one(g, lat, lon, x, y, n, 1);
}
/* c_gdxywdval (gdxywdval.c:38-110) / c_gdllwdval (gdllwdval.c:36-100): interpolated winds at points as speed / direction */
int32_t c_gdxywdval(int32_t gdin, float *uuout, float *vvout, float *uuin, float *vvin, float *x, float *y, int32_t n)
{
    if (!grid_ok(gdin) || n < 0) return -1;
    if (G[gdin].nsub > 0) { fprintf(stderr, "<gdxywdval> 'U' grids are outside the MI355X hot-path scope here\n"); return -1; }
    size_t nb = sizeof(float) * (size_t)(n > 0 ? n : 1);
    float *tlat = (float *)malloc(nb), *tlon = (float *)malloc(nb), *tu = (float *)malloc(nb), *tv = (float *)malloc(nb);
    int rc = c_gdxyvval(gdin, tu, tv, uuin, vvin, x, y, n);
    if (rc >= 0) rc = gdllfxy_one(&G[gdin], tlat, tlon, x, y, n, 0);       /* c_gdllfxy_orig (gdxywdval.c:94): rows as they are counted in the array */
    if (rc >= 0) rc = c_gdwdfuv(gdin, uuout, vvout, tu, tv, tlat, tlon, n);
    free(tlat); free(tlon); free(tu); free(tv);
    return rc < 0 ? -1 : 0;
}
int32_t c_gdllwdval(int32_t gdid, float *uuout, float *vvout, float *uuin, float *vvin, float *lat, float *lon, int32_t n)
{
    if (!grid_ok(gdid) || n < 0) return -1;
    if (G[gdid].nsub > 0) { fprintf(stderr, "<gdllwdval> 'U' grids are outside the MI355X hot-path scope here\n"); return -1; }
    if (c_gdllvval(gdid, uuout, vvout, uuin, vvin, lat, lon, n) < 0) return -1;           /* gdllwdval.c:79-80 */
    size_t nb = sizeof(float) * (size_t)(n > 0 ? n : 1);
    float *s = (float *)malloc(nb), *d = (float *)malloc(nb);
    int rc = c_gdwdfuv(gdid, s, d, uuout, vvout, lat, lon, n);
    if (rc >= 0) { memcpy(uuout, s, sizeof(float) * (size_t)n); memcpy(vvout, d, sizeof(float) * (size_t)n); }
    free(s); free(d);
    return rc < 0 ? -1 : 0;
}

/* c_gdwdfuv (gdwdfuv.c:29-110): grid components at (lat, lon) -> speed, direction; c_gduvfwd (gduvfwd.c:29-96): back */
static int wind_conv_dev(int32_t gdid, float *d_a, float *d_b, const float *d_lat, const float *d_lon, int32_t npts, int to_wd)
{
    if (!grid_ok(gdid)) return -1;
    if (need_device(to_wd ? "c_gdwdfuv" : "c_gduvfwd")) return -1;
    ezh_grid *g = &G[gdid];
    if (g->nsub > 0 || g->grtyp == 'Y') { fprintf(stderr, "<%s> '%c' grids are outside the MI355X hot-path scope here\n", to_wd ? "c_gdwdfuv" : "c_gduvfwd", g->grtyp); return -1; }
    int rotated = g->grtyp == 'E' || (g->grtyp == 'Z' && g->grref == 'E');
    ezhip_wind_plan wp;
    memset(&wp, 0, sizeof(wp));
    if (to_wd) {
        wp.wd_only = 1; wp.src_rotated = rotated;
        if (rotated) { const float *xg = g->grtyp == 'E' ? g->xg : g->xgref; h_crot(wp.r, wp.ri, xg[1], xg[0], xg[3], xg[2]); }
        if (g->grtyp == 'N' || g->grtyp == 'S') { wp.src_ps = g->grtyp == 'N' ? 1 : 2; wp.src_xg4 = g->xg[3]; }
    } else {
        if (g->grtyp == 'E') { fprintf(stderr, "<c_gduvfwd> '%c' target grids are outside the MI355X hot-path scope for winds\n", g->grtyp); return -1; }
        wp.wd_in = 1;
        if (rotated) { float ri_[9]; wp.dst_rotated = 1; h_crot(wp.r_dst, ri_, g->xgref[1], g->xgref[0], g->xgref[3], g->xgref[2]); }
        if (g->grtyp == 'N' || g->grtyp == 'S') { wp.dst_ps = g->grtyp == 'N' ? 1 : 2; wp.dst_xg4 = g->xg[3]; }
    }
    if (npts <= 0) return 0;
    float *d_cs = NULL;
    if (g->grtyp == '!') {                                  /* the rotation angles of the caller's longitudes: host (REAL libm chain of ez_lambxyfll99), then back */
        float *lon_h = (float *)malloc(sizeof(float) * (size_t)npts);
        if (!lon_h) return -1;
        if (ezhip_d2h(lon_h, d_lon, sizeof(float) * (size_t)npts) || ezhip_sync()) { free(lon_h); return -1; }
        d_cs = lamb_wind_cs_dev(g, lon_h, npts, to_wd);
        free(lon_h);
        if (!d_cs) return -1;
        if (to_wd) wp.src_lamb_cs = d_cs; else wp.dst_lamb_cs = d_cs;
    }
    int rc = ezhip_wind_rotate(&wp, d_a, d_b, d_lat, d_lon, npts, 1) ? -1 : 0;
    if (d_cs) { ezhip_sync(); ezhip_free(d_cs); }
    return rc;
}
int32_t c_gdwdfuv_dev(int32_t gdid, float *d_spd, float *d_wd, const float *d_uu, const float *d_vv, const float *d_lat, const float *d_lon, int32_t npts)
{
    if (npts > 0 && (d_spd != d_uu) && ezhip_d2d(d_spd, d_uu, sizeof(float) * (size_t)npts)) return -1;
    if (npts > 0 && (d_wd != d_vv) && ezhip_d2d(d_wd, d_vv, sizeof(float) * (size_t)npts)) return -1;
    return wind_conv_dev(gdid, d_spd, d_wd, d_lat, d_lon, npts, 1);
}
int32_t c_gduvfwd_dev(int32_t gdid, float *d_uu, float *d_vv, const float *d_spd, const float *d_wd, const float *d_lat, const float *d_lon, int32_t npts)
{
    if (npts > 0 && (d_uu != d_spd) && ezhip_d2d(d_uu, d_spd, sizeof(float) * (size_t)npts)) return -1;
    if (npts > 0 && (d_vv != d_wd) && ezhip_d2d(d_vv, d_wd, sizeof(float) * (size_t)npts)) return -1;
    return wind_conv_dev(gdid, d_uu, d_vv, d_lat, d_lon, npts, 0);
}
static int32_t wind_conv_host(int32_t gdid, float *o1, float *o2, float *i1, float *i2, float *lat, float *lon, int32_t npts, int to_wd)
{
    if (!grid_ok(gdid)) return -1;
    if (need_device(to_wd ? "c_gdwdfuv" : "c_gduvfwd")) return -1;
    size_t nb = sizeof(float) * (size_t)(npts > 0 ? npts : 1);
    float *d1 = (float *)upload(i1, nb), *d2 = (float *)upload(i2, nb), *dla = (float *)upload(lat, nb), *dlo = (float *)upload(lon, nb);
    int rc = (d1 && d2 && dla && dlo) ? wind_conv_dev(gdid, d1, d2, dla, dlo, npts, to_wd) : -1;
    if (rc == 0 && (ezhip_d2h(o1, d1, sizeof(float) * (size_t)npts) || ezhip_d2h(o2, d2, sizeof(float) * (size_t)npts) || ezhip_sync())) rc = -1;
    ezhip_sync();
    ezhip_free(d1); ezhip_free(d2); ezhip_free(dla); ezhip_free(dlo);
    return rc;
}
int32_t c_gdwdfuv(int32_t gdid, float *spd_out, float *wd_out, float *uuin, float *vvin, float *latin, float *lonin, int32_t npts)
{ return wind_conv_host(gdid, spd_out, wd_out, uuin, vvin, latin, lonin, npts, 1); }
int32_t c_gduvfwd(int32_t gdid, float *uugdout, float *vvgdout, float *uullin, float *vvllin, float *latin, float *lonin, int32_t npts)
{ return wind_conv_host(gdid, uugdout, vvgdout, uullin, vvllin, latin, lonin, npts, 0); }

/* ------------------------------------------------------------------------------------------ */
/* masks: src/interp/ez_mask.c                                                                  */
/* ------------------------------------------------------------------------------------------ */
int c_gdsetmask(int gdid, int *mask)                       /* :67-87 */
{
    if (!grid_ok(gdid)) return -1;
    ezh_grid *g = &G[gdid];
    if (g->nsub > 0) { fprintf(stderr, "<gdsetmask> This operation is not supported for 'U' grids.\n"); return -1; }
    size_t n = (size_t)g->ni * g->nj;
    free(g->mask);
    g->mask = (int *)malloc(sizeof(int) * n);
    memcpy(g->mask, mask, sizeof(int) * n);
    return 0;
}
int c_gdgetmask(int gdid, int *mask)                       /* :89-113 */
{
    if (!grid_ok(gdid)) return -1;
    ezh_grid *g = &G[gdid];
    if (g->nsub > 0) { fprintf(stderr, "<gdgetmask> This operation is not supported for 'U' grids.\n"); return -1; }
    if (!g->mask) return -1;
    memcpy(mask, g->mask, sizeof(int) * (size_t)g->ni * g->nj);
    return 0;
}
static ezh_set *mask_set(const char *who)
{
    ezh_set *s = current_set(who);
    if (!s) return NULL;
    if (G[s->gdin].nsub > 0 || G[s->gdout].nsub > 0) { fprintf(stderr, "<%s> This operation is not supported for 'U' grids.\n", who); return NULL; }
    if (need_device(who)) return NULL;
    pthread_mutex_lock(&g_plan_mtx);
    int rc = ensure_points(s);                              /* the per-point x, y of the set (the reference reads its cached gset x, y) */
    pthread_mutex_unlock(&g_plan_mtx);
    return rc ? NULL : s;
}
static int mask_dev(const char *who, int *d_mask_out, const int *d_mask_in, int mode)
{
    ezh_set *s = mask_set(who);
    if (!s) return -1;
    ezh_grid *gi = &G[s->gdin], *go = &G[s->gdout];
    return ezhip_mask(d_mask_out, s->d_x, s->d_y, d_mask_in, gi->ni, gi->nj, go->ni, go->nj, mode, mode == 0 && O.cld_interp_alg == 1) ? -1 : 0;
}
int c_ezsint_mask_dev(int *d_mask_out, const int *d_mask_in) { return mask_dev("ezsint_mask", d_mask_out, d_mask_in, 0); }           /* :184-227 */
int c_ezget_mask_zones_dev(int *d_mask_out, const int *d_mask_in) { return mask_dev("ezget_mask_zones", d_mask_out, d_mask_in, 1); } /* :231-264 */
static int mask_host(const char *who, int *mask_out, int *mask_in, int mode)
{
    ezh_set *s = mask_set(who);
    if (!s) return -1;
    size_t nin = (size_t)G[s->gdin].ni * G[s->gdin].nj, nout = (size_t)G[s->gdout].ni * G[s->gdout].nj;
    int *d_in = (int *)upload(mask_in, sizeof(int) * nin), *d_out = (int *)ezhip_malloc(sizeof(int) * nout);
    int rc = (d_in && d_out) ? mask_dev(who, d_out, d_in, mode) : -1;
    if (rc == 0 && (ezhip_d2h(mask_out, d_out, sizeof(int) * nout) || ezhip_sync())) rc = -1;
    ezhip_sync();
    ezhip_free(d_in); ezhip_free(d_out);
    return rc;
}
int c_ezsint_mask(int *mask_out, int *mask_in) { return mask_host("ezsint_mask", mask_out, mask_in, 0); }
int c_ezget_mask_zones(int *mask_out, int *mask_in) { return mask_host("ezget_mask_zones", mask_out, mask_in, 1); }
int c_ezsint_m(float *zout, float *zin) { (void)zout; (void)zin; fprintf(stderr, "<ezsint_m> This operation is currently not implemented.\n"); return 0; }   /* :115-118 */
int c_ezuvint_m(float *uuout, float *vvout, float *uuin, float *vvin) { (void)uuout; (void)vvout; (void)uuin; (void)vvin; fprintf(stderr, "<ezuvint_m> This operation is currently not implemented.\n"); return 0; }

/* c_ezsint_mdm (:127-153): c_ezsint, c_ezsint_mask, then lorenzo_mask_fill method 2 (masked points <- the field's minimum) */
int c_ezsint_mdm_dev(float *d_zout, int *d_mask_out, const float *d_zin, const int *d_mask_in)
{
    ezh_set *s = mask_set("ezsint_mdm");
    if (!s) return -1;
    size_t nout = (size_t)G[s->gdout].ni * G[s->gdout].nj;
    if (c_ezsint_dev(d_zout, d_zin) < 0) return -1;
    if (c_ezsint_mask_dev(d_mask_out, d_mask_in)) return -1;
    if (ensure_scratch(s)) return -1;
    return ezhip_mask_fill_min(d_zout, d_mask_out, nout, (unsigned *)(t_scratch8 + 6)) ? -1 : 0;
}
int c_ezuvint_mdm_dev(float *d_uuout, float *d_vvout, int *d_mask_out, const float *d_uuin, const float *d_vvin, const int *d_mask_in)   /* :155-181 */
{
    ezh_set *s = mask_set("ezuvint_mdm");
    if (!s) return -1;
    size_t nout = (size_t)G[s->gdout].ni * G[s->gdout].nj;
    if (c_ezsint_mask_dev(d_mask_out, d_mask_in)) return -1;
    if (c_ezuvint_dev(d_uuout, d_vvout, d_uuin, d_vvin) < 0) return -1;
    if (ensure_scratch(s)) return -1;
    if (ezhip_mask_fill_min(d_uuout, d_mask_out, nout, (unsigned *)(t_scratch8 + 6))) return -1;
    return ezhip_mask_fill_min(d_vvout, d_mask_out, nout, (unsigned *)(t_scratch8 + 6)) ? -1 : 0;
}
static int mdm_host(float *o1, float *o2, int *mask_out, float *i1, float *i2, int *mask_in)
{
    ezh_set *s = mask_set(o2 ? "ezuvint_mdm" : "ezsint_mdm");
    if (!s) return -1;
    size_t nin = (size_t)G[s->gdin].ni * G[s->gdin].nj, nout = (size_t)G[s->gdout].ni * G[s->gdout].nj;
    float *d_i1 = (float *)upload(i1, sizeof(float) * nin), *d_i2 = o2 ? (float *)upload(i2, sizeof(float) * nin) : NULL;
    float *d_o1 = (float *)ezhip_malloc(sizeof(float) * nout), *d_o2 = o2 ? (float *)ezhip_malloc(sizeof(float) * nout) : NULL;
    int *d_mi = (int *)upload(mask_in, sizeof(int) * nin), *d_mo = (int *)ezhip_malloc(sizeof(int) * nout);
    int rc = -1;
    if (d_i1 && d_o1 && d_mi && d_mo && (!o2 || (d_i2 && d_o2))) {
        rc = o2 ? c_ezuvint_mdm_dev(d_o1, d_o2, d_mo, d_i1, d_i2, d_mi) : c_ezsint_mdm_dev(d_o1, d_mo, d_i1, d_mi);
        if (rc == 0 && (ezhip_d2h(o1, d_o1, sizeof(float) * nout) || (o2 && ezhip_d2h(o2, d_o2, sizeof(float) * nout)) ||
                        ezhip_d2h(mask_out, d_mo, sizeof(int) * nout) || ezhip_sync())) rc = -1;
    }
    ezhip_sync();
    ezhip_free(d_i1); ezhip_free(d_i2); ezhip_free(d_o1); ezhip_free(d_o2); ezhip_free(d_mi); ezhip_free(d_mo);
    return rc;
}
int c_ezsint_mdm(float *zout, int *mask_out, float *zin, int *mask_in) { return mdm_host(zout, NULL, mask_out, zin, NULL, mask_in); }
int c_ezuvint_mdm(float *uuout, float *vvout, int *mask_out, float *uuin, float *vvin, int *mask_in) { return mdm_host(uuout, vvout, mask_out, uuin, vvin, mask_in); }

/* ------------------------------------------------------------------------------------------ */
/* Yin-Yang 'U' source grids: c_ezgdef_supergrid (ezgdef_supergrid.c:40-150), c_ezgdef_yymask       */
/* (ezgdef_yymask.c:25-52), c_ezyymint (ezyymint.c:27-95), c_ezyy_calcxy (ezyy_calcxy.c:25-179),     */
/* c_ezyysint (ezyysint.c:27-170), c_ezyyuvint (ezyyuvint.c:29-215) -- towards ONE ordinary target   */
/* ------------------------------------------------------------------------------------------ */
static int yy_maskgrid(const ezh_grid *sub)
{   /* the part of a subgrid with 45 <= ax <= 315 and -45 <= ay <= 45, as a grid of its own */
    int k = 0, i0 = 0, j0 = 0;
    for (int i = 0; i < sub->ni; i++) if (sub->ax[i] >= 45.0 && sub->ax[i] <= 315.0) { k++; if (k == 1) i0 = i; }
    int yni = k;
    k = 0;
    for (int j = 0; j < sub->nj; j++) if (sub->ay[j] >= -45.0 && sub->ay[j] <= 45.0) { k++; if (k == 1) j0 = j; }
    int ynj = k;
    char t[2] = {sub->grtyp, 0}, r[2] = {sub->grref, 0};
    float *ax = (float *)malloc(sizeof(float) * (size_t)(yni > 0 ? yni : 1)), *ay = (float *)malloc(sizeof(float) * (size_t)(ynj > 0 ? ynj : 1));
    memcpy(ax, &sub->ax[i0], sizeof(float) * (size_t)yni); memcpy(ay, &sub->ay[j0], sizeof(float) * (size_t)ynj);
    int gd = c_ezgdef_fmem(yni, ynj, t, r, sub->igref[0], sub->igref[1], sub->igref[2], sub->igref[3], ax, ay);
    free(ax); free(ay);
    return gd;
}

int32_t c_ezgdef_supergrid(int32_t ni, int32_t nj, char *grtyp, char *grref, int32_t vercode, int32_t nsubgrids, int32_t *subgrid)
{
    (void)grref;
    if (nsubgrids <= 1) { fprintf(stderr, "<c_ezgdef_supergrid> nsubgrids given is less than 2! Aborting...\n"); return -1; }
    if (vercode != 1) { fprintf(stderr, "<c_ezgdef_supergrid> invalid vercode! Aborting...\n"); return -1; }
    if (nsubgrids != 2 || grtyp[0] != 'U' || !grid_ok(subgrid[0]) || !grid_ok(subgrid[1])) {
        fprintf(stderr, "<c_ezgdef_supergrid> only 'U' grids made of two defined subgrids are in the MI355X hot-path scope\n"); return -1;
    }
    for (int k = 0; k < 2; k++) {
        const ezh_grid *sg = &G[subgrid[k]];
        if (sg->grtyp != 'Z' || sg->grref != 'E' || sg->ni != ni || 2 * sg->nj != nj) {
            fprintf(stderr, "<c_ezgdef_supergrid> the subgrids of a 'U' grid must be Z-on-E grids of ni x nj/2 points\n"); return -1;
        }
    }
    int m0 = yy_maskgrid(&G[subgrid[0]]), m1 = yy_maskgrid(&G[subgrid[1]]);
    if (m0 < 0 || m1 < 0) return -1;
    pthread_mutex_lock(&g_mtx);
    int gd;
    for (gd = 0; gd < nG; gd++) if (G[gd].used && G[gd].grtyp == 'U' && G[gd].sub[0] == subgrid[0] && G[gd].sub[1] == subgrid[1]) { G[gd].refcount++; pthread_mutex_unlock(&g_mtx); return gd; }
    gd = grid_slot();
    if (gd < 0) { pthread_mutex_unlock(&g_mtx); return -1; }
    ezh_grid *g = &G[gd];
    memset(g, 0, sizeof(*g));
    g->used = 1; g->refcount = 1; g->grtyp = 'U'; g->grref = 'F';
    g->ni = ni; g->nj = nj; memcpy(g->ig, G[subgrid[0]].ig, sizeof(g->ig));
    g->igref[0] = vercode;
    g->nsub = 2; g->sub[0] = subgrid[0]; g->sub[1] = subgrid[1]; g->maskgrid[0] = m0; g->maskgrid[1] = m1;
    pthread_mutex_unlock(&g_mtx);
    return gd;
}

/* c_ezyymint + c_ezyy_calcxy for a single target grid: host, once per set (first-call work like ez_calcxy) */
static int yy_plan_locked(ezh_set *s);
static int yy_plan(ezh_set *s)
{
    if (s->yy_ready) return 0;
    pthread_mutex_lock(&g_plan_mtx);                        /* lazily built per-set state: one builder at a time */
    int rc = s->yy_ready ? 0 : yy_plan_locked(s);
    pthread_mutex_unlock(&g_plan_mtx);
    return rc;
}
static int yy_plan_locked(ezh_set *s)
{
    ezh_grid *gu = &G[s->gdin], *go = &G[s->gdout];
    const int n = go->ni * go->nj;
    float *lat = (float *)malloc(sizeof(float) * (size_t)n), *lon = (float *)malloc(sizeof(float) * (size_t)n);
    float *x = (float *)malloc(sizeof(float) * (size_t)n), *y = (float *)malloc(sizeof(float) * (size_t)n);
    float *tl = (float *)malloc(sizeof(float) * (size_t)n);
    int *idx = (int *)malloc(sizeof(int) * (size_t)n);
    unsigned char *yang = (unsigned char *)malloc((size_t)n);
    int rc = -1;
    if (c_gdll(s->gdout, lat, lon)) goto done;
    {   /* the mask: a point outside the Yin core grid (zone DEHORS of the nearest-neighbour interpolation with extrapolation
         * value 1 that ezyymint.c:44-66 runs) is served by Yang */
        const ezh_grid *mg = &G[gu->maskgrid[0]];
        /* that interpolation is a c_ezsint_orig, and c_ezsint_orig extrapolates (ez_corrval) only with polar_correction = yes (ezsint.c): with the
         * option off when the set is first used the mask stays 0 everywhere -- every point is served by Yin, far outside it -- and the set keeps
         * that mask (yyflags, ezyy_calcxy.c:40-42).  Reproduced: a drop-in may not know better than the reference here. */
        const int extrapolates = O.polar_correction == 1;
        memcpy(tl, lon, sizeof(float) * (size_t)n);
        if (h_locate_mt(mg, x, y, lat, tl, n)) goto done;
        for (int k = 0; k < n; k++) {
            int ix = (int)((double)x[k] + 0.5), iy = (int)((double)y[k] + 0.5);
            yang[k] = extrapolates && (ix < 1 || iy < 1 || ix > mg->ni || iy > mg->nj);
        }
    }
    for (int sub = 0; sub < 2; sub++) {
        int c = 0;
        for (int k = 0; k < n; k++) if (yang[k] == sub) { idx[c] = k; x[c] = lat[k]; y[c] = lon[k]; c++; }      /* x, y: the list's lat, lon */
        s->yy_count[sub] = c;
        float *lx = (float *)malloc(sizeof(float) * (size_t)(c + 1)), *ly = (float *)malloc(sizeof(float) * (size_t)(c + 1));
        /* c_gdxyfll_orig on the list (ezyy_calcxy.c:112-121) -- NOT the locate of a grid set (ez_calcxy): the two routines do the same job with
         * different arithmetic and differ in the last bit of x or y at about one point in a hundred (found by tools/fuzz_vs_ref4.py: bilinear values
         * one ulp off on 37 of 1568 points of a Gaussian target) */
        if (c_gdxyfll_orig(gu->sub[sub], lx, ly, x, y, c) < 0) { free(lx); free(ly); goto done; }
        s->d_yy_idx[sub] = (int *)upload(idx, sizeof(int) * (size_t)(c + 1));
        s->d_yy_lat[sub] = (float *)upload(x, sizeof(float) * (size_t)(c + 1));
        s->d_yy_lon[sub] = (float *)upload(y, sizeof(float) * (size_t)(c + 1));
        s->d_yy_x[sub] = (float *)upload(lx, sizeof(float) * (size_t)(c + 1));
        s->d_yy_y[sub] = (float *)upload(ly, sizeof(float) * (size_t)(c + 1));
        ezhip_sync();
        free(lx); free(ly);
        if (!s->d_yy_idx[sub] || !s->d_yy_lat[sub] || !s->d_yy_lon[sub] || !s->d_yy_x[sub] || !s->d_yy_y[sub]) goto done;
    }
    for (int k = 0; k < 4; k++) { s->d_yy_tmp[k] = (float *)ezhip_malloc(sizeof(float) * (size_t)(n + 1)); if (!s->d_yy_tmp[k]) goto done; }
    s->yy_ready = 1;
    rc = 0;
done:
    free(lat); free(lon); free(x); free(y); free(tl); free(idx); free(yang);
    return rc;
}

/* the degenerate cases of ezyysint.c:88-97: the target IS one of the subgrids */
static int yy_same_subgrid(const ezh_set *s) { const ezh_grid *gu = &G[s->gdin]; return s->gdout == gu->sub[0] ? 0 : s->gdout == gu->sub[1] ? 1 : -1; }

static int32_t yy_sint(ezh_set *s, float *d_zout, const float *d_zin)
{
    ezh_grid *gu = &G[s->gdin], *go = &G[s->gdout];
    const size_t nsub = (size_t)G[gu->sub[0]].ni * G[gu->sub[0]].nj;
    int same = yy_same_subgrid(s);
    if (same >= 0) { ezhip_d2d(d_zout, d_zin + same * nsub, sizeof(float) * nsub); return 1; }
    if (O.use_1subgrid == 1) {                               /* ezyysint.c:99-123: the caller picks ONE subgrid as the source */
        if (O.valeur_1subgrid != gu->sub[0] && O.valeur_1subgrid != gu->sub[1]) {
            fprintf(stderr, "<c_ezyysint> define src subgridid in ezsetival(subgridid)! Aborting...\n");
            return -1;
        }
        const int k = O.valeur_1subgrid == gu->sub[1];
        const int save_in = cur_gdin, save_out = cur_gdout, gdout_ = s->gdout, sub_ = gu->sub[k];
        cur_gdin = sub_; cur_gdout = gdout_;
        int r = ezhip_prepare_set() ? -1 : c_ezsint_dev(d_zout, d_zin + k * nsub);     /* prepare: the set's out-of-grid classification (return code 2) */
        cur_gdin = save_in; cur_gdout = save_out;
        return r;
    }
    int degree = O.degre_interp;
    if (degree != DEG_NEAREST && degree != DEG_LINEAR && degree != DEG_CUBIC) return -1;
    if (yy_plan(s)) return -1;
    (void)go;
    for (int sub = 0; sub < 2; sub++) {                      /* c_gdxysval on each subgrid (no polar correction), then the merge */
        ezh_grid *gs = &G[gu->sub[sub]];
        if (ensure_grid_dev(gs)) return -1;
        ezhip_pts_plan pp;
        fill_pts_plan(NULL, gs, &pp, degree, 0, 0);
        pp.out_idx = s->d_yy_idx[sub];                       /* each point writes its own target position: the merge */
        if (ezhip_interp_pts(&pp, d_zout, d_zin + sub * nsub, s->d_yy_x[sub], s->d_yy_y[sub], s->yy_count[sub])) return -1;
    }
    return 0;
}

/* target = a 'U' grid (ezyysint.c:79-86 for an ordinary source, :162-230 for a 'U' source): its two subgrids are two
 * ordinary Z-on-E targets, each with its own grid set (and, for a 'U' source, its own mask and point lists) */
static int32_t yy_sint_to_u(ezh_set *s, float *d_zout, const float *d_zin)
{
    const ezh_grid *gu = &G[s->gdout];
    const size_t nsub = (size_t)G[gu->sub[0]].ni * G[gu->sub[0]].nj;
    const int save_in = cur_gdin, save_out = cur_gdout, gdin = s->gdin, sub0 = gu->sub[0], sub1 = gu->sub[1];
    int rc = 0;
    for (int k = 0; k < 2; k++) {
        cur_gdin = gdin; cur_gdout = k ? sub1 : sub0;
        int r = c_ezsint_dev(d_zout + k * nsub, d_zin);
        if (r < 0) { rc = r; break; }
        if (r == 2) rc = 2;
    }
    cur_gdin = save_in; cur_gdout = save_out;
    return rc;
}

static int32_t yy_uvint(ezh_set *s, float *d_uuout, float *d_vvout, const float *d_uuin, const float *d_vvin, int wd_only)
{
    ezh_grid *gu = &G[s->gdin], *go = &G[s->gdout];
    const size_t nsub = (size_t)G[gu->sub[0]].ni * G[gu->sub[0]].nj;
    if (yy_same_subgrid(s) >= 0) {
        fprintf(stderr, "<c_ezuvint> this Yin-Yang case (target = one of the subgrids) is outside the MI355X hot-path scope\n");
        return -1;
    }
    if (go->grtyp == 'E' || go->grtyp == '!') { fprintf(stderr, "<c_ezuvint> '%c' target grids are outside the MI355X hot-path scope for winds\n", go->grtyp); return -1; }
    int degree = O.degre_interp;
    if (degree != DEG_NEAREST && degree != DEG_LINEAR && degree != DEG_CUBIC) return -1;
    if (yy_plan(s) || ensure_coords_dev(go)) return -1;
    for (int sub = 0; sub < 2; sub++) {
        /* c_gdxyvval + c_gdwdfuv of the subgrid on its list (ezyyuvint.c:177-180), speed / direction merged into the outputs */
        ezh_grid *gs = &G[gu->sub[sub]];
        if (ensure_grid_dev(gs)) return -1;
        ezhip_pts_plan pp;
        fill_pts_plan(NULL, gs, &pp, degree, 0, 0);
        const int c = s->yy_count[sub];
        if (ezhip_interp_pts2(&pp, &pp, s->d_yy_tmp[0], s->d_yy_tmp[1], d_uuin + sub * nsub, d_vvin + sub * nsub, s->d_yy_x[sub], s->d_yy_y[sub], c)) return -1;
        ezhip_wind_plan wp;
        memset(&wp, 0, sizeof(wp));
        wp.wd_only = 1; wp.src_rotated = 1;
        h_crot(wp.r, wp.ri, gs->xgref[1], gs->xgref[0], gs->xgref[3], gs->xgref[2]);
        if (c && ezhip_wind_rotate(&wp, s->d_yy_tmp[0], s->d_yy_tmp[1], s->d_yy_lat[sub], s->d_yy_lon[sub], c, 1)) return -1;
        if (ezhip_scatter(d_uuout, s->d_yy_tmp[0], s->d_yy_idx[sub], c) || ezhip_scatter(d_vvout, s->d_yy_tmp[1], s->d_yy_idx[sub], c)) return -1;
    }
    if (wd_only) return 0;                                   /* c_ezyywdint (ezyywdint.c:157-205): the merged speed / direction are the result */
    /* ONE c_gduvfwd on the target with its own lat/lon (ezyyuvint.c:201) */
    ezhip_wind_plan wp;
    memset(&wp, 0, sizeof(wp));
    wp.separable = go->separable; wp.wd_in = 1;
    if (go->grtyp == 'N' || go->grtyp == 'S') { wp.dst_ps = go->grtyp == 'N' ? 1 : 2; wp.dst_xg4 = go->xg[3]; }
    if (go->grtyp == 'Z' && go->grref == 'E') { float ri_[9]; wp.dst_rotated = 1; h_crot(wp.r_dst, ri_, go->xgref[1], go->xgref[0], go->xgref[3], go->xgref[2]); }
    if (ezhip_wind_rotate(&wp, d_uuout, d_vvout, go->d_lat, go->d_lon, go->ni, go->nj)) return -1;
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
int32_t ezqkdef_(int32_t *ni, int32_t *nj, char *grtyp, int32_t *ig1, int32_t *ig2, int32_t *ig3, int32_t *ig4, int32_t *iunit, int32_t l)
{ (void)l; char t[2] = {grtyp[0], 0}; return c_ezqkdef(*ni, *nj, t, *ig1, *ig2, *ig3, *ig4, *iunit); }
int32_t ezgdef_fmem_(int32_t *ni, int32_t *nj, char *grtyp, char *grref, int32_t *ig1, int32_t *ig2, int32_t *ig3, int32_t *ig4, float *ax, float *ay, int32_t l1, int32_t l2)
{ (void)l1; (void)l2; char t[2] = {grtyp[0], 0}, r[2] = {grref[0], 0}; return c_ezgdef_fmem(*ni, *nj, t, r, *ig1, *ig2, *ig3, *ig4, ax, ay); }
int32_t ezdefset_(int32_t *gdout, int32_t *gdin) { return c_ezdefset(*gdout, *gdin); }
int gdsetmask_(int *gdid, int *mask) { return c_gdsetmask(*gdid, mask); }
int gdgetmask_(int *gdid, int *mask) { return c_gdgetmask(*gdid, mask); }
int ezsint_mdm_(float *zout, int *mask_out, float *zin, int *mask_in) { return c_ezsint_mdm(zout, mask_out, zin, mask_in); }
int ezuvint_mdm_(float *uuout, float *vvout, int *mask_out, float *uuin, float *vvin, int *mask_in) { return c_ezuvint_mdm(uuout, vvout, mask_out, uuin, vvin, mask_in); }
int ezsint_mask_(int *mask_out, int *mask_in) { return c_ezsint_mask(mask_out, mask_in); }
int ezget_mask_zones_(int *mask_out, int *mask_in) { return c_ezget_mask_zones(mask_out, mask_in); }
int32_t ezgdef_supergrid_(int32_t *ni, int32_t *nj, char *grtyp, char *grref, int32_t *vercode, int32_t *nsubgrids, int32_t *subgrid, int32_t l1, int32_t l2)
{ (void)l1; (void)l2; char t[2] = {grtyp[0], 0}, r[2] = {grref[0], 0}; return c_ezgdef_supergrid(*ni, *nj, t, r, *vercode, *nsubgrids, subgrid); }
static void ftn2c(char *dst, const char *src, int len)
{
    int n = len < 31 ? len : 31;
    memset(dst, 0, 32); memcpy(dst, src, (size_t)n);
    while (n > 0 && (dst[n - 1] == ' ' || dst[n - 1] == 0)) dst[--n] = 0;
}
int32_t ezsetopt_(char *option, char *value, int32_t lo, int32_t lv)
{ char o[32], v[32]; ftn2c(o, option, lo); ftn2c(v, value, lv); return c_ezsetopt(o, v); }
int32_t ezsint_(float *zout, float *zin) { return c_ezsint(zout, zin); }
int32_t ezuvint_(float *uo, float *vo, float *ui, float *vi) { return c_ezuvint(uo, vo, ui, vi); }
int32_t ezwdint_(float *so, float *dout, float *ui, float *vi) { return c_ezwdint(so, dout, ui, vi); }
int32_t gdxysint_(float *zout, float *zin, int32_t *gdin, float *x, float *y, int32_t *npts) { return c_gdxysint(zout, zin, *gdin, x, y, *npts); }
int32_t gdxysval_(int32_t *gdin, float *zout, float *zin, float *x, float *y, int32_t *n) { return c_gdxysval(*gdin, zout, zin, x, y, *n); }
int32_t gdxyfll_(int32_t *gd, float *x, float *y, float *lat, float *lon, int32_t *n) { return c_gdxyfll(*gd, x, y, lat, lon, *n); }
int32_t gdllsval_(int32_t *gd, float *zout, float *zin, float *lat, float *lon, int32_t *n) { return c_gdllsval(*gd, zout, zin, lat, lon, *n); }
int32_t gdxyvval_(int32_t *gd, float *uo, float *vo, float *ui, float *vi, float *x, float *y, int32_t *n) { return c_gdxyvval(*gd, uo, vo, ui, vi, x, y, *n); }
int32_t gdllvval_(int32_t *gd, float *uo, float *vo, float *ui, float *vi, float *lat, float *lon, int32_t *n) { return c_gdllvval(*gd, uo, vo, ui, vi, lat, lon, *n); }
int32_t gdll_(int32_t *gd, float *lat, float *lon) { return c_gdll(*gd, lat, lon); }
int32_t gdrls_(int32_t *gd) { return c_gdrls(*gd); }
