/*
 * interpv_kernels.hip -- gfx950 kernel of the 1-D (vertical) interpolation package, SURVEY.md 8f row 4.
 *
 * Replaces the column loops of src/interpv:
 *   Interp1D_FindPos_Body.inc:80-144           bracket search (REAL index / uncertainty halving)
 *   Interp1D_NearestNeighbour_Body.inc:68-97   Interp1D_Linear_Body.inc:83-112
 *   Interp1D_CubicLagrange_Body.inc:112-157    Interp1D_CubicWithDerivs_Body.inc:151-200
 *   Extrap1D_Fixed_Body.inc:80-99              Extrap1D_LapseRate_Body.inc:92-116        Extrap1D_Abort_Body.inc:70-92
 *
 * One kernel, k_interpv: a block owns COLS adjacent columns (the fast index of every array, so each level row is one
 * contiguous COLS * sizeof(REAL) segment) and all, or a slice of, the destination levels.  The block's source levels
 * are staged once in LDS ([level][column]: bank = column, conflict-free), because the search reads them
 * 2 * log2(srcNumLevels) times per destination value; state values are read where the bracket points (twice to
 * eight times per value, neighbouring columns share rows, served by L2).  Every thread walks its column's destination
 * levels four at a time so that four independent searches overlap their LDS latencies.  Search, interpolation and
 * extrapolation are stages of the same pass: a caller that wants all three pays the level / state traffic once and
 * never materialises posnDestInSrc.
 *
 * Arithmetic: the type and operation order of the reference text, no contraction (-ffp-contract=off in the Makefile):
 * REAL (float) or REAL*8 for NearestNeighbour / Linear / LapseRate, REAL*8 always for the two cubics.
 * HBM-bound: bytes per column = (ns * narrays_in + nd * (1 + narrays_out)) * sizeof(REAL) (+ 4 * nd when the brackets
 * are written or read).
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "ezhip_shim.h"
#include "interpv_shim.h"

#define IV_THREADS 256
#define IV_TU 4                         /* destination levels in flight per thread */
#define IV_SB 16                        /* level rows in flight per lane while staging */
#define IV_LDS_MAX (80 * 1024)          /* staged levels per block, at most: two blocks per CU */
#define IV_LDS_SOFT (20 * 1024)         /* preferred: eight blocks (32 waves) per CU -- REAL*8, 80 levels: 64 columns (40 KB) 2.75 ms, 32 columns (20 KB) 2.33 ms for the search */

template <typename R>
struct iv_args {
    int algo, extrap, find, write_posn;
    int n, ns, nd, sij, dij;
    const R *vls, *ss, *sds;
    int *posn;
    const R *vld;
    R *sd, *sdd;
    int xdown, xup;
    R gdown, gup;
    unsigned long long *abort_key;
    int t_per_block;
    unsigned nbx;                   /* column blocks; the grid's x size is nbx rounded up to a multiple of 8 when xcd_map is set */
    int xcd_map;                    /* 0, or the number of consecutive column blocks given to one XCD at a time */
};

/* COLS = 0: levels are read from memory (srcNumLevels too large for LDS), 64 columns per wave */
/* waves_per_eu(8, 8): 64 VGPRs (a few spilled dwords) for eight waves per SIMD -- the kernel lives on occupancy: REAL search 1.95 -> 1.65 ms,
 * fused linear 2.70 -> 2.36 ms, cubic with derivatives 3.57 -> 2.90 ms against the 76 VGPRs / six waves the compiler picks by itself */
/* ALGO: the interpolation algorithm as a compile-time constant (the other algorithms' code and registers are gone: the body is VALU-bound,
 * 1828 VALU instructions per wave with everything in one body), or -1: read from the arguments (the levels-in-memory form) */
template <typename R, int COLS, int ALGO>
__global__ __launch_bounds__(IV_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_interpv(iv_args<R> a)
{
    const int algo = ALGO >= 0 ? ALGO : a.algo;
    extern __shared__ __attribute__((aligned(16))) unsigned char iv_smem[];
    R *lev = (R *)iv_smem;
    constexpr int C = COLS ? COLS : IV_THREADS;
    constexpr int NTG = IV_THREADS / C;                       /* groups of threads sharing the block's columns */
    const int lc = threadIdx.x % C, tg = threadIdx.x / C;
    /* consecutive column blocks on ONE XCD (workgroups go round the 8 XCDs in turn): the 256-byte row segments of neighbouring blocks share
     * 128-byte lines (rows are aligned to the element only), and the two blocks' fetches of such a line then meet in the same L2 */
    unsigned bx = blockIdx.x;
    if (a.xcd_map) { const unsigned sq = blockIdx.x >> 3, G = (unsigned)a.xcd_map; bx = ((sq / G) * 8u + (blockIdx.x & 7u)) * G + sq % G; if (bx >= a.nbx) return; }
    const long long i = (long long)bx * C + lc;
    const bool live = i < a.n;
    const size_t sij = (size_t)a.sij, dij = (size_t)a.dij;
    const int ns = a.ns;
    if (COLS) {
        /* thread group tg stages a contiguous run of rows of its column (rows tg, tg + NTG, ... measured slower: REAL*8 search 4.1 -> 2.3 ms) */
        const int per = (ns + NTG - 1) / NTG, k0 = tg * per, k1 = min(ns, k0 + per);
        /* IV_SB rows in flight per lane: written as `lev[..] = vls[..]` per row the loop waits for every load before it issues the next
         * (one memory round trip per row: 20 per block at 80 levels) -- REAL fused linear 2.45 -> ... ms */
        const R *vsrc = a.vls + (live ? i : (long long)a.n - 1);           /* dead lanes of the last block read the last column */
        for (int k = k0; k < k1; k += IV_SB) {
            R x[IV_SB];
#pragma unroll
            for (int j = 0; j < IV_SB; j++) x[j] = vsrc[(size_t)min(k + j, k1 - 1) * sij];
#pragma unroll
            for (int j = 0; j < IV_SB; j++) if (k + j < k1) lev[(size_t)(k + j) * C + lc] = x[j];
        }
        __syncthreads();
    }
    if (!live) return;
    const R *vcol = a.vls + i;
#define LEV(k) (COLS ? lev[((k) - 1) * C + lc] : vcol[(size_t)((k) - 1) * sij])
#define SSRC(k) a.ss[i + (size_t)((k) - 1) * sij]
#define DSRC(k) a.sds[i + (size_t)((k) - 1) * sij]
    /* direction of the levels: column 1 speaks for all (Interp1D_FindPos_Body.inc:88, Interp1D_Linear_Body.inc:73) */
    const bool asc = a.vls[sij] > a.vls[0];
    const int smallest = asc ? 1 : ns, largest = asc ? ns : 1;
    const int loposn = asc ? 1 : ns - 1, hiposn = asc ? ns - 1 : 1;
    const int t_begin = blockIdx.y * a.t_per_block, t_end = min(a.nd, t_begin + a.t_per_block);
    const R lev_small = LEV(smallest), lev_large = LEV(largest);     /* the column's extreme levels: the clamp and the extrapolators compare against them */
    const R *levc = lev + lc - C;                                      /* LEV(k) = levc[k * C] */

    for (int tb = t_begin + tg * IV_TU; tb < t_end; tb += NTG * IV_TU) {
        R v[IV_TU];
        int p[IV_TU];
#pragma unroll
        for (int u = 0; u < IV_TU; u++) {
            const int t = min(tb + u, t_end - 1);
            v[u] = a.vld[i + (size_t)t * dij];
        }
        if (a.find) {
            float idx[IV_TU];
#pragma unroll
            for (int u = 0; u < IV_TU; u++) idx[u] = 0.5f * (float)(ns + 1);
            float unc = 0.5f * (float)(ns - 1);
            while (unc > 0.95f) {
                unc = 0.5f * unc;
                const float up = asc ? unc : -unc;
#pragma unroll
                for (int u = 0; u < IV_TU; u++) {
                    const int ii = (int)idx[u];
                    R l0, l1;
                    if (COLS) { const R *q = levc + ii * C; l0 = q[0]; l1 = q[C]; } else { l0 = LEV(ii); l1 = LEV(ii + 1); }
                    if (v[u] >= l0) idx[u] = idx[u] + up;
                    if (v[u] <= l1) idx[u] = idx[u] - up;
                }
            }
#pragma unroll
            for (int u = 0; u < IV_TU; u++) p[u] = (int)idx[u];
        } else {
#pragma unroll
            for (int u = 0; u < IV_TU; u++) {
                const int t = min(tb + u, t_end - 1);
                p[u] = min(max(a.posn[i + (size_t)t * dij], 1), ns - 1);     /* a bracket outside 1..ns-1 would read outside the arrays */
            }
        }
#pragma unroll
        for (int u = 0; u < IV_TU; u++) {
            const int t = tb + u;
            if (t >= t_end) break;
            const size_t o = i + (size_t)t * dij;
            const R x = v[u];
            const int q = p[u];
            if (a.find && a.write_posn) a.posn[o] = q;
            R out = 0, outd = 0;
            bool have = false, have_d = false;
            switch (algo) {
            case IV_NEAREST: {
                const R lb = LEV(q), la = LEV(q + 1);
                out = (fabs(x - lb) < fabs(x - la)) ? SSRC(q) : SSRC(q + 1);
                have = true;
                break;
            }
            case IV_LINEAR: {
                const R lev1 = LEV(q), lev2 = LEV(q + 1), s1 = SSRC(q), s2 = SSRC(q + 1);
                const R slope = (s2 - s1) / (lev2 - lev1);
                out = slope * (x - lev1) + s1;
                have = true;
                break;
            }
            case IV_CUBIC_LAGRANGE: {
                const int f = max(min(q - 1, ns - 3), 1);
                const double x1 = LEV(f), x2 = LEV(f + 1), x3 = LEV(f + 2), x4 = LEV(f + 3);
                const double z1 = SSRC(f), z2 = SSRC(f + 1), z3 = SSRC(f + 2), z4 = SSRC(f + 3);
                const double cx1 = 1.0 / (x2 - x1), cx2 = 1.0 / (x3 - x1), cx3 = 1.0 / (x3 - x2);
                const double cx4 = 1.0 / (x4 - x1), cx5 = 1.0 / (x4 - x2), cx6 = 1.0 / (x4 - x3);
                const double a1 = z1;
                const double a2 = cx1 * (z2 - z1);
                const double a3 = cx2 * (cx3 * (z3 - z2) - cx1 * (z2 - z1));
                const double a4 = cx4 * (cx5 * (cx6 * (z4 - z3) - cx3 * (z3 - z2)) - cx2 * (cx3 * (z3 - z2) - cx1 * (z2 - z1)));
                const double xt = x;
                out = (R)(a1 + (xt - x1) * (a2 + (xt - x2) * (a3 + a4 * (xt - x3))));
                have = true;
                break;
            }
            case IV_CUBIC_DERIVS: {
                const double lb = LEV(q), la = LEV(q + 1);
                const double lc_ = 0.5 * (lb + la);
                const double sb0 = SSRC(q), sa0 = SSRC(q + 1), sb1 = DSRC(q), sa1 = DSRC(q + 1);
                const double dlin2 = la - lb;
                const double o2 = 1 / dlin2;
                const double dltar = (double)x - lc_;
                const double dd = 0.125 * dlin2 * dlin2 - 0.5 * dltar * dltar;
                const double sc0 = 0.5 * (sa0 + sb0);
                const double sc1 = (sa0 - sb0) * o2;
                const double sc2 = (sa1 - sb1) * o2;
                const double sc3 = ((sa1 - sc1) - (sc1 - sb1)) * o2 * o2;
                const double st2 = sc2 + 2. * dltar * sc3;
                const double st1 = sc1 + dltar * st2;
                const double st0 = sc0 + dltar * sc1;
                out = (R)(st0 - dd * st2);
                outd = (R)(st1 - 2. * dd * sc3);
                have = have_d = true;
                break;
            }
            default: break;
            }
            if (algo >= IV_LINEAR) {                          /* the clamp that closes the three polynomial routines */
                if (!a.xdown && x < lev_small) { out = SSRC(smallest); if (have_d) outd = DSRC(smallest); }
                else if (!a.xup && x > lev_large) { out = SSRC(largest); if (have_d) outd = DSRC(largest); }
            }
            if (a.extrap) {
                int side = -1;                                    /* 0 below the lowest level, 1 above the highest */
                if (a.xdown && q == loposn) { if (x < lev_small) side = 0; }
                else if (a.xup && q == hiposn) { if (x > lev_large) side = 1; }
                if (side >= 0) {
                    const int e = side ? largest : smallest;
                    if (a.extrap == IVX_FIXED) { out = side ? a.gup : a.gdown; have = true; }
                    else if (a.extrap == IVX_LAPSERATE) {
                        const R d = x - (side ? lev_large : lev_small);
                        const R pr = (side ? a.gup : a.gdown) * d;
                        out = SSRC(e) + pr;
                        have = true;
                    } else {
                        atomicMin(a.abort_key, (((unsigned long long)t * (unsigned long long)a.n + (unsigned long long)i) << 1) | (unsigned)side);
                    }
                }
            }
            if (have) a.sd[o] = out;
            if (have_d) a.sdd[o] = outd;
        }
    }
#undef LEV
#undef SSRC
#undef DSRC
}

template <typename R, int COLS, int ALGO>
static int launch_cols_algo(const iv_args<R> &a0, hipStream_t st)
{
    iv_args<R> a = a0;
    constexpr int C = COLS ? COLS : IV_THREADS;
    const unsigned bx = (unsigned)(((long long)a.n + C - 1) / C);
    /* few columns: slice the destination levels over blockIdx.y until the chip is covered */
    int slices = 1;
    const int chunk = (IV_THREADS / C) * IV_TU;
    while (slices < 64 && (long long)bx * slices < 2048 && (a.nd + slices - 1) / slices > chunk) slices *= 2;
    a.t_per_block = (((a.nd + slices - 1) / slices) + chunk - 1) / chunk * chunk;
    const unsigned by = (unsigned)((a.nd + a.t_per_block - 1) / a.t_per_block);
    const size_t lds = COLS ? (size_t)COLS * a.ns * sizeof(R) : 0;
    static const bool no_xcd = getenv("INTERPV_HIP_NO_XCD_MAP") != nullptr;     /* development */
    static const char *xg = getenv("INTERPV_HIP_XCD_GROUP");                     /* development: consecutive blocks per XCD (0: an eighth of the grid) */
    unsigned G = xg ? (unsigned)atoi(xg) : 16u;          /* 4 / 16 / 64 / 512 / an eighth: 1.67 / 1.60 / 1.59 / 1.58 / 1.59 ms (REAL fused linear), unmapped 1.71 */
    if (G == 0) G = (bx + 7u) / 8u;
    a.nbx = bx; a.xcd_map = (bx >= 64 && !no_xcd) ? (int)G : 0;
    const unsigned gx = a.xcd_map ? (bx + 8u * G - 1u) / (8u * G) * (8u * G) : bx;
    if (lds > 64 * 1024) {
        static bool raised;                                         /* one per instantiation */
        if (!raised) { if (hipFuncSetAttribute((const void *)k_interpv<R, COLS, ALGO>, hipFuncAttributeMaxDynamicSharedMemorySize, IV_LDS_MAX) != hipSuccess) return -1; raised = true; }
    }
    hipLaunchKernelGGL((k_interpv<R, COLS, ALGO>), dim3(gx, by), dim3(IV_THREADS), lds, st, a);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
template <typename R, int COLS>
static int launch_cols(const iv_args<R> &a, hipStream_t st)
{
    if constexpr (COLS == 0) return launch_cols_algo<R, COLS, -1>(a, st);
    else switch (a.algo) {
    case IV_NONE: return launch_cols_algo<R, COLS, IV_NONE>(a, st);
    case IV_NEAREST: return launch_cols_algo<R, COLS, IV_NEAREST>(a, st);
    case IV_LINEAR: return launch_cols_algo<R, COLS, IV_LINEAR>(a, st);
    case IV_CUBIC_LAGRANGE: return launch_cols_algo<R, COLS, IV_CUBIC_LAGRANGE>(a, st);
    case IV_CUBIC_DERIVS: return launch_cols_algo<R, COLS, IV_CUBIC_DERIVS>(a, st);
    default: return -1;
    }
}

template <typename R>
static int launch_typed(const ivhip_args *h)
{
    iv_args<R> a;
    a.algo = h->algo; a.extrap = h->extrap; a.find = h->find; a.write_posn = h->write_posn;
    a.n = h->n; a.ns = h->ns; a.nd = h->nd; a.sij = h->sij; a.dij = h->dij;
    a.vls = (const R *)h->vls; a.ss = (const R *)h->ss; a.sds = (const R *)h->sds;
    a.posn = h->posn; a.vld = (const R *)h->vld; a.sd = (R *)h->sd; a.sdd = (R *)h->sdd;
    a.xdown = h->xdown; a.xup = h->xup; a.gdown = (R)h->gdown; a.gup = (R)h->gup;
    a.abort_key = h->abort_key; a.t_per_block = 0; a.nbx = 0; a.xcd_map = 0;
    hipStream_t st = (hipStream_t)ezhip_get_stream();
    const size_t per_col = (size_t)a.ns * sizeof(R);
    /* occupancy decides (measured, 80 -> 60 levels, REAL: 256 columns per block 3.9 ms, 128: 2.3 ms, 64: 1.8 ms for the search):
     * the most columns whose level tile still leaves six blocks per CU, down to 16 columns (one 64-byte segment per row) */
    const char *force = getenv("INTERPV_HIP_COLS");               /* development: 64 / 32 / 16 / 0 */
    int cols = per_col * 64 <= IV_LDS_SOFT ? 64 : per_col * 32 <= IV_LDS_SOFT ? 32 : per_col * 16 <= IV_LDS_SOFT ? 16 :
               per_col * 16 <= IV_LDS_MAX ? 16 : 0;
    if (force) { const int fc = atoi(force); if (fc == 0 || (fc <= cols && (fc == 64 || fc == 32 || fc == 16))) cols = fc; }
    switch (cols) {
    case 64: return launch_cols<R, 64>(a, st);
    case 32: return launch_cols<R, 32>(a, st);
    case 16: return launch_cols<R, 16>(a, st);
    default: return launch_cols<R, 0>(a, st);
    }
}

extern "C" int ivhip_launch(const ivhip_args *h)
{
    if (h->n <= 0 || h->nd <= 0) return 0;
    return h->prec8 ? launch_typed<double>(h) : launch_typed<float>(h);
}

extern "C" int ivhip_copy2d(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width_bytes, size_t rows, int to_device)
{
    if (!width_bytes || !rows) return 0;
    if (!to_device) for (size_t r = 0; r < rows; r++) ezhip_touch_writable((char *)dst + r * dpitch, width_bytes);      /* see ezhip_d2h */
    return hipMemcpy2DAsync(dst, dpitch, src, spitch, width_bytes, rows, to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost,
                            (hipStream_t)ezhip_get_stream()) == hipSuccess ? 0 : -1;
}
