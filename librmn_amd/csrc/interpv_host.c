/*
 * interpv_host.c -- host front-end of the vertical interpolation: argument checks and messages of the
 * reference's routines, the device-pointer entry points, and the reference's Fortran-callable symbols, which stage
 * host arrays through HBM.  Kernels: interpv_kernels.hip.  No CPU fallback: without a HIP device every entry fails loudly.
 *
 * Reference interface: src/interpv/Interp1D_Arguments.inc:21-58 (+ Interp1D_Arguments_X.inc:3-7 for the _X forms),
 * Interp1D_FindPos_Body.inc:33-52; wrappers Interp1D_Linear.F90:22-99 and its siblings; the old-interface shell
 * Interp1D_PreX_Shell.inc:36-62 (null extension arrays).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "../../include/interpv_hip.h"
#include "ezhip_shim.h"
#include "interpv_shim.h"

static int iv_need_device(const char *who)
{
    if (ezhip_runtime_ok()) return ezhip_bound_device_ok(who);
    fprintf(stderr, "<%s> no usable HIP device: the MI355X vertical interpolation has no CPU fallback\n", who);
    return -1;
}

static const char *const algo_name[] = { "", "Interp1D_NearestNeighbour", "Interp1D_Linear", "Interp1D_CubicLagrange", "Interp1D_CubicWithDerivs" };
static const char *const extrap_name[] = { "", "Extrap1D_Fixed", "Extrap1D_LapseRate", "Extrap1D_Abort" };

/* the reference's own refusals (Interp1D_Linear_Body.inc:67-70, Interp1D_CubicLagrange_Body.inc:88-91,
 * Interp1D_CubicWithDerivs_Body.inc:131-134): an error line, nothing computed */
static int enough_levels(int algo, int ns)
{
    if (algo == IV_LINEAR && ns < 2) { fprintf(stderr, "Interp1D_Linear: The linear algorithm requires 2 source points\n"); return 0; }
    if (algo == IV_CUBIC_LAGRANGE && ns < 4) { fprintf(stderr, "Interp1D_CubicLagrange: The cubic Lagrange algorithm requires 4 source points\n"); return 0; }
    if (algo == IV_CUBIC_DERIVS && ns < 2) { fprintf(stderr, "Interp1D_CubicWithDerivs: The cubic algorithm with derivatives requires 2 source points\n"); return 0; }
    if (ns < 2) { fprintf(stderr, "interpv: srcNumLevels = %d: the bracket of a destination level needs 2 source levels\n", ns); return 0; }
    return 1;
}

static int shape_ok(const char *who, int n, int ns, int nd, int sij, int dij)
{
    if (n < 0 || nd < 0 || ns < 0 || sij < n || dij < n) {
        fprintf(stderr, "<%s> inconsistent dimensions: numInterpSets %d, srcNumLevels %d, destNumLevels %d, src_ijDim %d, dst_ijDim %d\n", who, n, ns, nd, sij, dij);
        return 0;
    }
    return 1;
}

/* per host thread: the word the abort kernel reduces into */
static __thread unsigned long long *t_abort_key;

static int run(const char *who, int prec8, int algo, int extrap, int find, int write_posn, int n, int ns, int nd, int sij, int dij,
               const void *vls, const void *ss, const void *sds, int32_t *posn, const void *vld, void *sd, void *sdd,
               int xdown, int xup, double gdown, double gup)
{
    if (iv_need_device(who)) return -1;
    if (!shape_ok(who, n, ns, nd, sij, dij)) return -1;
    if (!enough_levels(algo, ns)) return -1;
    if (n == 0 || nd == 0) return 0;
    if (!vls || !vld || ((!find || write_posn) && !posn) || ((algo || extrap == IVX_LAPSERATE) && !ss) ||
        ((algo || extrap == IVX_FIXED || extrap == IVX_LAPSERATE) && !sd) || (algo == IV_CUBIC_DERIVS && (!sds || !sdd))) {
        fprintf(stderr, "<%s> a required array is NULL\n", who);
        return -1;
    }
    ivhip_args a;
    memset(&a, 0, sizeof(a));
    a.prec8 = prec8; a.algo = algo; a.extrap = extrap; a.find = find; a.write_posn = write_posn;
    a.n = n; a.ns = ns; a.nd = nd; a.sij = sij; a.dij = dij;
    a.vls = vls; a.ss = ss; a.sds = sds; a.posn = posn; a.vld = vld; a.sd = sd; a.sdd = sdd;
    a.xdown = xdown != 0; a.xup = xup != 0; a.gdown = gdown; a.gup = gup;
    if (extrap == IVX_ABORT) {
        if (!t_abort_key) t_abort_key = (unsigned long long *)ezhip_malloc(sizeof(unsigned long long));
        if (!t_abort_key || ezhip_memset(t_abort_key, 0xFF, sizeof(unsigned long long))) return -1;
        a.abort_key = t_abort_key;
    }
    if (ivhip_launch(&a)) { fprintf(stderr, "<%s> kernel launch failed: %s\n", who, ezhip_last_error()); return -1; }
    if (extrap == IVX_ABORT) {
        unsigned long long key = 0;
        if (ezhip_d2h(&key, t_abort_key, sizeof(key)) || ezhip_sync()) return -1;
        if (key != ~0ULL) {
            /* Extrap1D_Abort_Body.inc:76-79 / :85-88: the first offender in the reference's loop order (vt outer, i inner) */
            const int above = (int)(key & 1);
            const unsigned long long lin = key >> 1;
            const size_t i = (size_t)(lin % (unsigned long long)n), t = (size_t)(lin / (unsigned long long)n);
            const size_t es = prec8 ? 8 : 4;
            double lv[2] = { 0, 0 };
            unsigned char raw[16], dir[16];
            int asc = 1;
            if (!ezhip_d2h(dir, vls, es) && !ezhip_d2h(dir + 8, (const char *)vls + (size_t)sij * es, es) && !ezhip_sync())
                asc = prec8 ? (*(double *)(dir + 8) > *(double *)dir) : (*(float *)(dir + 8) > *(float *)dir);
            const int edge = (above ? asc : !asc) ? ns : 1;          /* index of the highest (above) / lowest (below) level */
            if (!ezhip_d2h(raw, (const char *)vld + (i + t * (size_t)dij) * es, es) &&
                !ezhip_d2h(raw + 8, (const char *)vls + (i + (size_t)(edge - 1) * (size_t)sij) * es, es) && !ezhip_sync()) {
                lv[0] = prec8 ? *(double *)raw : (double)*(float *)raw;
                lv[1] = prec8 ? *(double *)(raw + 8) : (double)*(float *)(raw + 8);
            }
            fprintf(stderr, " Extrap1D_Abort: Attempting extrapolation to level  %.9g , this is %s level  %.9g\n", lv[0],
                    above ? "above the highest" : "below the lowest", lv[1]);
            return 2;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* device-pointer entry points                                                                  */
/* ------------------------------------------------------------------------------------------ */
int32_t interpv_hip_findpos_dev(int prec8, int n, int ns, int nd, int sij, int dij, const void *d_vls, int32_t *d_posn, const void *d_vld)
{
    return run("Interp1D_FindPos", prec8, IV_NONE, IVX_NONE, 1, 1, n, ns, nd, sij, dij, d_vls, NULL, NULL, d_posn, d_vld, NULL, NULL, 0, 0, 0, 0);
}

int32_t interpv_hip_interp_dev(int prec8, int algo, int n, int ns, int nd, int sij, int dij, const void *d_vls, const void *d_ss, const void *d_sds,
                               const int32_t *d_posn, const void *d_vld, void *d_sd, void *d_sdd, int xdown, int xup)
{
    if (algo < IV_NEAREST || algo > IV_CUBIC_DERIVS) { fprintf(stderr, "<interpv_hip_interp_dev> unknown algorithm %d\n", algo); return -1; }
    return run(algo_name[algo], prec8, algo, IVX_NONE, 0, 0, n, ns, nd, sij, dij, d_vls, d_ss, d_sds, (int32_t *)d_posn, d_vld, d_sd, d_sdd, xdown, xup, 0, 0);
}

int32_t interpv_hip_extrap_dev(int prec8, int extrap, int n, int ns, int nd, int sij, int dij, const void *d_vls, const void *d_ss, const void *d_sds,
                               const int32_t *d_posn, const void *d_vld, void *d_sd, void *d_sdd, int xdown, int xup, double gdown, double gup)
{
    if (extrap < IVX_FIXED || extrap > IVX_ABORT) { fprintf(stderr, "<interpv_hip_extrap_dev> unknown extrapolation %d\n", extrap); return -1; }
    return run(extrap_name[extrap], prec8, IV_NONE, extrap, 0, 0, n, ns, nd, sij, dij, d_vls, d_ss, d_sds, (int32_t *)d_posn, d_vld, d_sd, d_sdd, xdown, xup, gdown, gup);
}

int32_t interpv_hip_column_dev(int prec8, int algo, int extrap, int n, int ns, int nd, int sij, int dij, const void *d_vls, const void *d_ss,
                               const void *d_sds, int32_t *d_posn, const void *d_vld, void *d_sd, void *d_sdd, int xdown, int xup, double gdown, double gup)
{
    if (algo < IV_NONE || algo > IV_CUBIC_DERIVS || extrap < IVX_NONE || extrap > IVX_ABORT) { fprintf(stderr, "<interpv_hip_column_dev> unknown algorithm %d / extrapolation %d\n", algo, extrap); return -1; }
    return run("interpv_hip_column_dev", prec8, algo, extrap, 1, d_posn != NULL, n, ns, nd, sij, dij, d_vls, d_ss, d_sds, d_posn, d_vld, d_sd, d_sdd, xdown, xup, gdown, gup);
}

/* ------------------------------------------------------------------------------------------ */
/* the reference's Fortran-callable symbols: host arrays staged through HBM                      */
/* ------------------------------------------------------------------------------------------ */
enum { SL_VLS, SL_SS, SL_SDS, SL_POSN, SL_VLD, SL_SD, SL_SDD, SL_COUNT };
static __thread struct { void *p; size_t cap; } t_slot[SL_COUNT];

static void *slot(int k, size_t nbytes)
{
    if (t_slot[k].cap < nbytes) {
        if (t_slot[k].p) { ezhip_sync(); ezhip_free(t_slot[k].p); }
        t_slot[k].p = ezhip_malloc(nbytes);
        t_slot[k].cap = t_slot[k].p ? nbytes : 0;
    }
    return t_slot[k].p;
}

/* the calling thread's device slots and page-locked buffers (ezhip_thread_release / thread exit) */
void ezh_interpv_thread_release(void)
{
    for (int k = 0; k < SL_COUNT; k++) { ezhip_free(t_slot[k].p); t_slot[k].p = NULL; t_slot[k].cap = 0; }
}
static void bounce_release(void);

static void die_no_device(const char *who)
{
    fprintf(stderr, "<%s> cannot continue without the HIP device (a Fortran subroutine has no status to return)\n", who);
    abort();
}

/* columns 1..n of a host (ijDim, levels) array <-> a dense (n, levels) device array.  Dense host arrays (ijDim == n) are copied as they are.  Arrays
 * with gaps between their rows go through a page-locked buffer of the library (rows gathered / scattered by the CPU): a pitched copy is one copy per
 * row, and rows read or written one by one in the caller's ordinary memory make the runtime lock and unlock overlapping page ranges of one array in
 * quick succession -- the gpu test suite died there once in eight runs ("Memory access fault ... Write access to a read-only page", the address
 * inside a result array); page-locking the caller's array for the length of the call made it worse (four runs in twelve). */
static __thread struct { void *p; size_t cap; } t_bounce[SL_COUNT];
static void *bounce(int k, size_t nbytes)
{
    if (t_bounce[k].cap < nbytes) {
        if (t_bounce[k].p) { ezhip_sync(); ezhip_host_free(t_bounce[k].p); }
        t_bounce[k].p = ezhip_host_alloc(nbytes);
        t_bounce[k].cap = t_bounce[k].p ? nbytes : 0;
    }
    return t_bounce[k].p;
}
static void bounce_release(void) { for (int k = 0; k < SL_COUNT; k++) { ezhip_host_free(t_bounce[k].p); t_bounce[k].p = NULL; t_bounce[k].cap = 0; } }
void ezh_interpv_thread_release2(void) { bounce_release(); }
static int up(int k, const void *h, size_t es, int n, int ijdim, int rows)
{
    const size_t row = es * (size_t)n, total = row * (size_t)rows;
    void *d = slot(k, total);
    if (!d) return -1;
    if (ijdim == n || rows == 1) return ezhip_h2d(d, h, total);
    char *b = (char *)bounce(k, total);
    if (!b) return -1;
    for (int r = 0; r < rows; r++) memcpy(b + (size_t)r * row, (const char *)h + (size_t)r * es * (size_t)ijdim, row);
    return ezhip_h2d(d, b, total);
}
/* results: queued here, scattered into the caller's arrays by finish() once the device is done */
static __thread struct { void *h; int k, n, ijdim, rows; size_t es; } t_pending[2];
static __thread int t_npending;
static int down(void *h, int k, size_t es, int n, int ijdim, int rows)
{
    const size_t row = es * (size_t)n, total = row * (size_t)rows;
    if (ijdim == n || rows == 1) return ezhip_d2h(h, t_slot[k].p, total);
    char *b = (char *)bounce(k, total);
    if (!b || t_npending >= 2) return -1;
    if (ezhip_d2h_pinned(b, t_slot[k].p, total)) return -1;
    t_pending[t_npending].h = h; t_pending[t_npending].k = k; t_pending[t_npending].n = n; t_pending[t_npending].ijdim = ijdim;
    t_pending[t_npending].rows = rows; t_pending[t_npending].es = es; t_npending++;
    return 0;
}
static int finish(void)
{
    const int rc = ezhip_sync();
    for (int q = 0; q < t_npending && !rc; q++) {
        const size_t row = t_pending[q].es * (size_t)t_pending[q].n;
        const char *b = (const char *)t_bounce[t_pending[q].k].p;
        for (int r = 0; r < t_pending[q].rows; r++) memcpy((char *)t_pending[q].h + (size_t)r * t_pending[q].es * (size_t)t_pending[q].ijdim, b + (size_t)r * row, row);
    }
    t_npending = 0;
    return rc;
}

static void findpos_host(int prec8, int n, int ns, int nd, int sij, int dij, const void *vls, int32_t *posn, const void *vld)
{
    const char *who = "Interp1D_FindPos";
    if (iv_need_device(who)) die_no_device(who);
    if (!shape_ok(who, n, ns, nd, sij, dij) || !enough_levels(IV_NONE, ns) || n == 0 || nd == 0) return;
    const size_t es = prec8 ? 8 : 4;
    if (up(SL_VLS, vls, es, n, sij, ns) || up(SL_VLD, vld, es, n, dij, nd) || !slot(SL_POSN, 4 * (size_t)n * nd)) goto fail;
    if (interpv_hip_findpos_dev(prec8, n, ns, nd, n, n, t_slot[SL_VLS].p, (int32_t *)t_slot[SL_POSN].p, t_slot[SL_VLD].p)) goto fail;
    if (down(posn, SL_POSN, 4, n, dij, nd) || finish()) goto fail;
    return;
fail:
    t_npending = 0;
    fprintf(stderr, "<%s> device error: %s\n", who, ezhip_last_error());
    abort();
}

static void apply_host(int prec8, int algo, int extrap, int n, int ns, int nd, int sij, int dij, const void *vls, const void *ss, const void *sds,
                       const int32_t *posn, const void *vld, void *sd, void *sdd, int xdown, int xup, double gdown, double gup)
{
    const char *who = algo ? algo_name[algo] : extrap_name[extrap];
    if (iv_need_device(who)) die_no_device(who);
    if (!shape_ok(who, n, ns, nd, sij, dij) || !enough_levels(algo, ns) || n == 0 || nd == 0) return;
    const size_t es = prec8 ? 8 : 4;
    const int need_ss = algo != IV_NONE || extrap == IVX_LAPSERATE, need_sds = algo == IV_CUBIC_DERIVS;
    const int inout_sd = algo == IV_NONE && extrap != IVX_ABORT;      /* the extrapolators leave every other value as it was */
    if (up(SL_VLS, vls, es, n, sij, ns) || up(SL_VLD, vld, es, n, dij, nd) || up(SL_POSN, posn, 4, n, dij, nd)) goto fail;
    if (need_ss && up(SL_SS, ss, es, n, sij, ns)) goto fail;
    if (need_sds && up(SL_SDS, sds, es, n, sij, ns)) goto fail;
    if (inout_sd ? up(SL_SD, sd, es, n, dij, nd) : !slot(SL_SD, es * (size_t)n * nd)) goto fail;
    if (need_sds && !slot(SL_SDD, es * (size_t)n * nd)) goto fail;
    int rc;
    if (algo)
        rc = interpv_hip_interp_dev(prec8, algo, n, ns, nd, n, n, t_slot[SL_VLS].p, t_slot[SL_SS].p, t_slot[SL_SDS].p, (const int32_t *)t_slot[SL_POSN].p,
                                    t_slot[SL_VLD].p, t_slot[SL_SD].p, t_slot[SL_SDD].p, xdown, xup);
    else
        rc = interpv_hip_extrap_dev(prec8, extrap, n, ns, nd, n, n, t_slot[SL_VLS].p, t_slot[SL_SS].p, t_slot[SL_SDS].p, (const int32_t *)t_slot[SL_POSN].p,
                                    t_slot[SL_VLD].p, t_slot[SL_SD].p, t_slot[SL_SDD].p, xdown, xup, gdown, gup);
    if (rc == 2) exit(2);                                            /* Extrap1D_Abort_Body.inc:79, :88 */
    if (rc) goto fail;
    if (extrap != IVX_ABORT && down(sd, SL_SD, es, n, dij, nd)) goto fail;
    if (need_sds && down(sdd, SL_SDD, es, n, dij, nd)) goto fail;
    if (finish()) goto fail;
    return;
fail:
    t_npending = 0;
    fprintf(stderr, "<%s> device error: %s\n", who, ezhip_last_error());
    abort();
}

void interp1d_findpos_(const int32_t *n, const int32_t *ns, const int32_t *nd, const int32_t *sij, const int32_t *dij,
                       const float *vls, int32_t *posn, const float *vld) { findpos_host(0, *n, *ns, *nd, *sij, *dij, vls, posn, vld); }
void interp1d_findpos8_(const int32_t *n, const int32_t *ns, const int32_t *nd, const int32_t *sij, const int32_t *dij,
                        const double *vls, int32_t *posn, const double *vld) { findpos_host(1, *n, *ns, *nd, *sij, *dij, vls, posn, vld); }

#define IV_CALL(P8, A, X) apply_host(P8, A, X, *numInterpSets, *srcNumLevels, *destNumLevels, *src_ijDim, *dst_ijDim, vLevelSource, stateSource,   \
    stateDerivSource, posnDestInSrc, vLevelDestn, stateDestn, stateDerivDestn, *extrapEnableDown, *extrapEnableUp, (double)*extrapGuideDown, (double)*extrapGuideUp)
#define IV_TWINS(name, A, X)                                                                                                        \
    void name##_(INTERPV_F_ARGS(float)) { IV_CALL(0, A, X); }                                                                       \
    void name##8_(INTERPV_F_ARGS(double)) { IV_CALL(1, A, X); }                                                                     \
    void name##_x_(INTERPV_F_ARGS(float) INTERPV_F_XARGS(float))                                                                    \
    { (void)flux; (void)numExtArraysIn; (void)numExtArraysOut; (void)ExtArraysIn; (void)ExtArraysOut; IV_CALL(0, A, X); }           \
    void name##_x8_(INTERPV_F_ARGS(double) INTERPV_F_XARGS(double))                                                                 \
    { (void)flux; (void)numExtArraysIn; (void)numExtArraysOut; (void)ExtArraysIn; (void)ExtArraysOut; IV_CALL(1, A, X); }
IV_TWINS(interp1d_nearestneighbour, IV_NEAREST, IVX_NONE)
IV_TWINS(interp1d_linear, IV_LINEAR, IVX_NONE)
IV_TWINS(interp1d_cubiclagrange, IV_CUBIC_LAGRANGE, IVX_NONE)
IV_TWINS(interp1d_cubicwithderivs, IV_CUBIC_DERIVS, IVX_NONE)
IV_TWINS(extrap1d_fixed, IV_NONE, IVX_FIXED)
IV_TWINS(extrap1d_lapserate, IV_NONE, IVX_LAPSERATE)
IV_TWINS(extrap1d_abort, IV_NONE, IVX_ABORT)
