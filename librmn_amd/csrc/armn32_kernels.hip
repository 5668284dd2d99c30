/*
 * armn32_kernels.hip -- HIP kernels of the IEEE-32 compressor c_armn_compress32 / c_armn_uncompress32 (datyp 133;
 * reference src/compresseur/armn_compress_32.c:59-275, :285-437).  gfx950 only.
 *
 * A field of floats is split into three planes -- sign bits, 8-bit exponents (minus their minimum), mantissas cut to
 * znbits - 9 bits -- and the exponent and mantissa planes go through the same 3 x 3-tile Lorenzo ("parallelogram") coder as
 * the 16-bit compressor, with other constants (packTokensParallelogram_8 :642-750: 4-bit width field, bit-length widths;
 * packTokensParallelogram32 :456-565: 5-bit width field, widths from the exponent of (float)max, a capacity check).
 *   encode: k_pg_tiles (per tile: differences -> width -> bit count) -> block sums + scan -> k_pg_emit (a thread writes the
 *           contiguous bit range of its 4 tiles; only its first / last word is shared: atomicOr) + k_pg_prefix;
 *   decode: the chain of tile headers is walked on the host (a tile's position is known only once every earlier width field
 *           has been read; armn32_host.c), then k_pg_dec_tiles / k_pg_dec_prefix rebuild the differences and a 2-D prefix sum
 *           (row scans, banded column scans) undoes the predictor, exactly like the 16-bit decoder does.
 * The sign plane's run-length coder (pack1bitRLE :827-901) is a state machine over 1 bit per point with seven states: a prefix scan of state maps (k_re_*, below).
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "packhip_shim.h"

extern "C" void *ezhip_get_stream(void);
#define STREAM ((hipStream_t)ezhip_get_stream())
static int chk32(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) return 0;
    fprintf(stderr, "<armn_compress32> %s: %s\n", what, hipGetErrorString(e));
    return -1;
}

/* ---- planes ------------------------------------------------------------------------------------------------ */
/* st[0] = OR of the words, st[1] = AND, st[2] = min exponent, st[3] = max exponent (initialised 0, ~0, 255, 0) */
__global__ __launch_bounds__(256) void k_a32_stats(unsigned *st, const unsigned *f, size_t n)
{
    unsigned o = 0, a = 0xFFFFFFFFu, lo = 255, hi = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const unsigned t = f[i], e = (t >> 23) & 0xFF;
        o |= t; a &= t; lo = min(lo, e); hi = max(hi, e);
    }
    for (int off = 32; off > 0; off >>= 1) {
        o |= __shfl_down(o, off, 64); a &= __shfl_down(a, off, 64);
        lo = min(lo, (unsigned)__shfl_down(lo, off, 64)); hi = max(hi, (unsigned)__shfl_down(hi, off, 64));
    }
    if ((threadIdx.x & 63) == 0) { atomicOr(&st[0], o); atomicAnd(&st[1], a); atomicMin(&st[2], lo); atomicMax(&st[3], hi); }
}
/* exponent - exp_base (bytes), mantissa >> (23 - nbits), sign mask (bit i & 31 of word i >> 5: compact_mask_char's layout, :958-975) */
__global__ __launch_bounds__(256) void k_a32_split(unsigned char *expo, unsigned *mant, unsigned *smask, const unsigned *f, size_t n, int nbits, unsigned exp_base)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const unsigned t = i < n ? f[i] : 0u;
    if (i < n) {
        expo[i] = (unsigned char)(((t >> 23) & 0xFF) - exp_base);
        unsigned m = t & 0x7FFFFFu;
        mant[i] = nbits < 23 ? m >> (23 - nbits) : m;
    }
    const unsigned long long b = __ballot(i < n && (t >> 31));
    const int lane = threadIdx.x & 63;
    if (lane == 0 && i < n) smask[i >> 5] = (unsigned)b;
    if (lane == 32 && i < n) smask[i >> 5] = (unsigned)(b >> 32);
}
/* the inverse: sign (code 0x00 all +, 0x10 all -, else the mask), exponent plane + exp_min, mantissa plane << (23 - nbits) (:398-424) */
__global__ __launch_bounds__(256) void k_a32_combine(unsigned *f, const int *expo, const int *mant, const unsigned *smask, size_t n, int nbits, unsigned exp_min, int code_signe, int have_expo)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned s = code_signe == 0x00 ? 0u : code_signe == 0x10 ? 1u : (smask[i >> 5] >> (i & 31)) & 1u;
    const unsigned e = have_expo ? ((unsigned)expo[i] + exp_min) & 0xFFu : exp_min & 0xFFu;
    f[i] = s << 31 | e << 23 | (unsigned)mant[i] << (23 - nbits);
}

/* ---- the parallelogram coder on a plane of T ---------------------------------------------------------------- */
__device__ __forceinline__ int pg_bitlen(unsigned v) { return v ? 32 - __clz((int)v) : 0; }
template <bool WIDE> __device__ __forceinline__ int pg_need(int mx)
{
    if (mx == 0) return 0;
    if (WIDE) return (__float_as_int((float)mx) >> 23) - 126;          /* :524-525: the exponent of (float)local_max */
    return pg_bitlen((unsigned)mx);                                     /* the fastlog[] table, :709-714 */
}
struct PgTile { int i0, j0, tm, tn; };
__device__ __forceinline__ PgTile pg_tile(unsigned t, int ni, int nj, int ntx)
{
    PgTile T; const int ty = (int)(t / (unsigned)ntx), tx = (int)(t - (unsigned)ty * (unsigned)ntx);
    T.i0 = 1 + 3 * tx; T.j0 = 1 + 3 * ty; T.tm = min(3, ni - T.i0); T.tn = min(3, nj - T.j0);
    return T;
}
template <class T> __device__ __forceinline__ int pg_diff(const T *u, int ni, int i, int j)
{
    const size_t k = (size_t)j * ni + i;
    return (int)((unsigned)u[k] - ((unsigned)u[k - ni] + (unsigned)u[k - 1] - (unsigned)u[k - 1 - ni]));
}
/* tinfo[t] = bit count of the tile | width << 24 */
template <class T, bool WIDE>
__global__ __launch_bounds__(256) void k_pg_tiles(unsigned *tinfo, const T *u, int ni, int nj, int ntx, unsigned ntiles, int container)
{
    const unsigned t = blockIdx.x * 256 + threadIdx.x;
    if (t >= ntiles) return;
    const PgTile P = pg_tile(t, ni, nj, ntx);
    int mx = 0;
    for (int n = 0; n < P.tn; n++) for (int m = 0; m < P.tm; m++) mx = max(mx, abs(pg_diff(u, ni, P.i0 + m, P.j0 + n)));
    const int need = pg_need<WIDE>(mx);
    tinfo[t] = (unsigned)(container + (need ? P.tm * P.tn * (need + 1) : 0)) | (unsigned)need << 24;
}
#define PG_ITEMS 4
__global__ __launch_bounds__(256) void k_pg_blocksum(unsigned long long *bsum, const unsigned *tinfo, unsigned ntiles)
{
    __shared__ unsigned long long sh[4];
    unsigned long long s = 0;
    const unsigned base = (blockIdx.x * 256 + threadIdx.x) * PG_ITEMS;
    for (int q = 0; q < PG_ITEMS; q++) if (base + q < ntiles) s += tinfo[base + q] & 0xFFFFFFu;
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) bsum[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}
/* exclusive scan of the block sums by one block (their number is small); total in *total */
__global__ __launch_bounds__(1024) void k_pg_scan(unsigned long long *bsum, int nblocks, unsigned long long *total)
{
    __shared__ unsigned long long wsum[16];
    const int per = (nblocks + 1023) / 1024;
    const int i0 = threadIdx.x * per, i1 = min(i0 + per, nblocks);
    unsigned long long run = 0;
    for (int i = i0; i < i1; i++) run += bsum[i];
    unsigned long long incl = run;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int off = 1; off < 64; off <<= 1) { const unsigned long long o = __shfl_up(incl, off, 64); if (lane >= off) incl += o; }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    unsigned long long base = 0;
    for (int k = 0; k < wv; k++) base += wsum[k];
    unsigned long long excl = base + incl - run;
    for (int i = i0; i < i1; i++) { const unsigned long long v = bsum[i]; bsum[i] = excl; excl += v; }
    if (threadIdx.x == 1023) *total = base + incl;
}
/* MSB-first writer over a ZEROED stream: a thread owns a contiguous bit range; only its first and last words can be shared */
struct PgOut {
    unsigned *z; size_t w; unsigned cw; int used; bool first_shared;
    __device__ __forceinline__ void init(unsigned *zz, unsigned long long pos) { z = zz; w = (size_t)(pos >> 5); used = (int)(pos & 31); cw = 0; first_shared = used != 0; }
    __device__ __forceinline__ void put(unsigned tok, int nbits)
    {
        while (nbits > 0) {
            const int take = min(nbits, 32 - used);
            const unsigned piece = take == 32 ? tok : ((tok >> (nbits - take)) & ((1u << take) - 1));
            cw |= take == 32 ? piece : piece << (32 - used - take);
            used += take; nbits -= take;
            if (used == 32) { if (first_shared) { atomicOr(&z[w], cw); first_shared = false; } else z[w] = cw; w++; cw = 0; used = 0; }
        }
    }
    __device__ __forceinline__ void finish() { if (used > 0 && cw) atomicOr(&z[w], cw); }
};
/* words flushed by the reference's `stuff` once P bits have been written: a full word leaves only when the next token arrives */
__device__ __forceinline__ long long pg_flushed(unsigned long long P) { return P ? (long long)((P - 1) >> 5) : 0; }
template <class T, bool WIDE>
__global__ __launch_bounds__(256) void k_pg_emit(unsigned *z, const unsigned long long *bsum, const unsigned *tinfo, const T *u, int ni, int nj, int ntx,
                                                 unsigned ntiles, int nbits, int container, unsigned long long body_start, long long remaining_space, int *fail)
{
    __shared__ unsigned long long sh[256];
    const unsigned base = (blockIdx.x * 256 + threadIdx.x) * PG_ITEMS;
    unsigned long long mine = 0;
    for (int q = 0; q < PG_ITEMS; q++) if (base + q < ntiles) mine += tinfo[base + q] & 0xFFFFFFu;
    sh[threadIdx.x] = mine;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const unsigned long long add = threadIdx.x >= off ? sh[threadIdx.x - off] : 0;
        __syncthreads();
        sh[threadIdx.x] += add;
        __syncthreads();
    }
    if (base >= ntiles) return;
    unsigned long long pos = body_start + bsum[blockIdx.x] + sh[threadIdx.x] - mine;
    PgOut bo; bo.init(z, pos);
    for (int q = 0; q < PG_ITEMS && base + q < ntiles; q++) {
        const unsigned ti = tinfo[base + q];
        const int need = (int)(ti >> 24);
        const PgTile P = pg_tile(base + q, ni, nj, ntx);
        bo.put((unsigned)need, container);
        if (need) {
            const int w = need + 1;
            const unsigned mask = w == 32 ? 0xFFFFFFFFu : ((1u << w) - 1);
            for (int n = 0; n < P.tn; n++) for (int m = 0; m < P.tm; m++) bo.put((unsigned)pg_diff(u, ni, P.i0 + m, P.j0 + n) & mask, w);
        }
        pos += ti & 0xFFFFFFu;
        if (WIDE && need && remaining_space < pg_flushed(pos) + 1 + ((need + 9 * nbits) >> 5)) *fail = 1;      /* :546-549 */
    }
    bo.finish();
}
template <class T>
__global__ __launch_bounds__(256) void k_pg_prefix(unsigned *z, const T *u, int ni, int nj, int nbits, int container)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    auto put = [&](unsigned long long pos, unsigned tok, int w) {
        const size_t wi = (size_t)(pos >> 5); const int sh = (int)(pos & 31);
        const unsigned long long v = (unsigned long long)(tok & (w == 32 ? 0xFFFFFFFFu : ((1u << w) - 1))) << (64 - w - sh);
        if ((unsigned)(v >> 32)) atomicOr(&z[wi], (unsigned)(v >> 32));
        if ((unsigned)v) atomicOr(&z[wi + 1], (unsigned)v);
    };
    if (t == 0) put(0, (unsigned)container, 3);                       /* the width-field size, written in `istep` = 3 bits (:500, :691) */
    if (t < ni) put(3 + (unsigned long long)t * nbits, (unsigned)u[t], nbits);
    else if (t < ni + nj - 1) put(3 + (unsigned long long)t * nbits, (unsigned)u[(size_t)ni * (t - ni + 1)], nbits);
}

/* encode one plane into the zeroed scratch stream d_z (z_words words); work: tinfo[ntiles] | bsum[nb + 1] | fail.  Asynchronous: the
 * total bit count of the tiles lands in work (packhip_pg_result reads it back) */
static size_t pg_ntiles(int ni, int nj, int *ntx) { *ntx = (ni - 1 + 2) / 3; return (size_t)*ntx * (size_t)((nj - 1 + 2) / 3); }
extern "C" size_t packhip_pg_work_bytes(int ni, int nj)
{
    int ntx; const size_t nt = pg_ntiles(ni, nj, &ntx);
    return 4 * nt + 8 * (nt / (256 * PG_ITEMS) + 4) + 64;
}
extern "C" int packhip_pg_encode(unsigned *d_z, size_t z_words, const void *d_u, int elem_bytes, int ni, int nj, int nbits, long long remaining_space, void *d_work)
{
    int ntx; const size_t nt = pg_ntiles(ni, nj, &ntx);
    if (nt >= (1ull << 31) || (size_t)ni * nj >= (1ull << 31)) return -1;
    const unsigned ntiles = (unsigned)nt;
    const int nb = (int)((nt + 256 * PG_ITEMS - 1) / (256 * PG_ITEMS));
    unsigned *tinfo = (unsigned *)d_work;
    unsigned long long *bsum = (unsigned long long *)((char *)d_work + ((4 * nt + 15) & ~(size_t)15));
    unsigned long long *total = bsum + nb;
    int *fail = (int *)(total + 1);
    const bool wide = elem_bytes == 4;
    const int container = wide ? 5 : 4;
    hipStream_t st = STREAM;
    if (hipMemsetAsync(d_z, 0, 4 * z_words, st) != hipSuccess || hipMemsetAsync(fail, 0, 4, st) != hipSuccess) return -1;
    const unsigned long long body_start = 3ull + (unsigned long long)(ni + nj - 1) * (unsigned long long)nbits;
    const unsigned tb = (ntiles + 255) / 256;
    if (wide) hipLaunchKernelGGL((k_pg_tiles<unsigned, true>), dim3(tb), dim3(256), 0, st, tinfo, (const unsigned *)d_u, ni, nj, ntx, ntiles, container);
    else hipLaunchKernelGGL((k_pg_tiles<unsigned char, false>), dim3(tb), dim3(256), 0, st, tinfo, (const unsigned char *)d_u, ni, nj, ntx, ntiles, container);
    hipLaunchKernelGGL(k_pg_blocksum, dim3(nb), dim3(256), 0, st, bsum, tinfo, ntiles);
    hipLaunchKernelGGL(k_pg_scan, dim3(1), dim3(1024), 0, st, bsum, nb, total);
    if (wide) {
        hipLaunchKernelGGL((k_pg_emit<unsigned, true>), dim3(nb), dim3(256), 0, st, d_z, bsum, tinfo, (const unsigned *)d_u, ni, nj, ntx, ntiles, nbits, container, body_start, remaining_space, fail);
        hipLaunchKernelGGL(k_pg_prefix<unsigned>, dim3((ni + nj + 255) / 256), dim3(256), 0, st, d_z, (const unsigned *)d_u, ni, nj, nbits, container);
    } else {
        hipLaunchKernelGGL((k_pg_emit<unsigned char, false>), dim3(nb), dim3(256), 0, st, d_z, bsum, tinfo, (const unsigned char *)d_u, ni, nj, ntx, ntiles, nbits, container, body_start, remaining_space, fail);
        hipLaunchKernelGGL(k_pg_prefix<unsigned char>, dim3((ni + nj + 255) / 256), dim3(256), 0, st, d_z, (const unsigned char *)d_u, ni, nj, nbits, container);
    }
    return chk32("pg_encode");
}
/* after packhip_pg_encode: synchronises; *bits = stream bits without the terminator, *failed = the capacity check of the 32-bit form tripped */
extern "C" int packhip_pg_result(void *d_work, int ni, int nj, int nbits, unsigned long long *bits, int *failed)
{
    int ntx; const size_t nt = pg_ntiles(ni, nj, &ntx);
    const int nb = (int)((nt + 256 * PG_ITEMS - 1) / (256 * PG_ITEMS));
    unsigned long long *bsum = (unsigned long long *)((char *)d_work + ((4 * nt + 15) & ~(size_t)15));
    unsigned long long h[2] = {0, 0};
    if (hipMemcpyAsync(h, bsum + nb, 12, hipMemcpyDeviceToHost, STREAM) != hipSuccess || hipStreamSynchronize(STREAM) != hipSuccess) return -1;
    *bits = 3ull + (unsigned long long)(ni + nj - 1) * (unsigned long long)nbits + h[0];
    *failed = (int)(h[1] & 0xFFFFFFFFu);
    return 0;
}
extern "C" int packhip_a32_split(unsigned char *d_expo, unsigned *d_mant, unsigned *d_smask, unsigned *d_stats, const float *d_f, size_t n, int nbits, unsigned *h_stats4)
{
    hipStream_t st = STREAM;
    const unsigned init[4] = {0u, 0xFFFFFFFFu, 255u, 0u};
    if (hipMemcpyAsync(d_stats, init, 16, hipMemcpyHostToDevice, st) != hipSuccess) return -1;
    int nb = (int)((n + 4095) / 4096); if (nb > 2048) nb = 2048; if (nb < 1) nb = 1;
    hipLaunchKernelGGL(k_a32_stats, dim3(nb), dim3(256), 0, st, d_stats, (const unsigned *)d_f, n);
    if (hipMemcpyAsync(h_stats4, d_stats, 16, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return -1;
    hipLaunchKernelGGL(k_a32_split, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_expo, d_mant, d_smask, (const unsigned *)d_f, n, nbits, h_stats4[2]);
    return chk32("a32_split");
}

/* ---- decode ------------------------------------------------------------------------------------------------------ */
__device__ __forceinline__ unsigned pg_get(const unsigned *z, unsigned long long pos, int w)
{   /* w <= 32 bits at bit position pos of the MSB-first stream */
    const size_t wi = (size_t)(pos >> 5); const int sh = (int)(pos & 31);
    const unsigned long long v = (unsigned long long)z[wi] << 32 | z[wi + 1];
    return (unsigned)((v << sh) >> (64 - w));
}
/* D = the 2-D differences whose inclusive prefix sum is the plane: tile interiors = the coded differences */
__global__ __launch_bounds__(256) void k_pg_dec_tiles(int *D, const unsigned *z, const unsigned long long *tpos, const unsigned *tpos32, int ni, int nj, int ntx, unsigned ntiles, int container)
{
    const unsigned t = blockIdx.x * 256 + threadIdx.x;
    if (t >= ntiles) return;
    const PgTile P = pg_tile(t, ni, nj, ntx);
    unsigned long long pos = tpos32 ? (unsigned long long)tpos32[t] : tpos[t];     /* (the device walk of unpack_kernels.hip leaves 32-bit positions) */
    const int need = (int)pg_get(z, pos, container);
    pos += container;
    for (int n = 0; n < P.tn; n++) for (int m = 0; m < P.tm; m++) {
        int v = 0;
        if (need) { const int w = need + 1; v = (int)(pg_get(z, pos, w) << (32 - w)) >> (32 - w); pos += w; }
        D[(size_t)(P.j0 + n) * ni + P.i0 + m] = v;
    }
}
/* first row / column: D(0,0) = u(0,0), D(i,0) = u(i,0) - u(i-1,0), D(0,j) = u(0,j) - u(0,j-1) */
__global__ __launch_bounds__(256) void k_pg_dec_prefix(int *D, const unsigned *z, int ni, int nj, int nbits)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= ni + nj - 1) return;
    auto tok = [&](int q) { return (int)pg_get(z, 3ull + (unsigned long long)q * nbits, nbits); };
    if (t == 0) D[0] = tok(0);
    else if (t < ni) D[t] = tok(t) - tok(t - 1);
    else { const int j = t - ni + 1; D[(size_t)j * ni] = tok(t) - (j == 1 ? tok(0) : tok(t - 1)); }
}
/* inclusive prefix sum along i, one block per row: 1024 elements per step as four coalesced sub-chunks of 256 (wave scans by shuffles, the sixteen wave totals
 * through LDS, one barrier per step).  (The first version gave every thread a contiguous 29-element piece -- two passes of strided accesses: 561 us per
 * 7200 x 3601 plane against ~50 here.) */
__global__ __launch_bounds__(256) void k_pg_rowscan(int *D, int ni)
{
    __shared__ int wsum[2][16];
    int *row = D + (size_t)blockIdx.x * ni;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    int carry = 0;
    for (int base = 0, it = 0; base < ni; base += 1024, it++) {
        int v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { const int i = base + k * 256 + t; v[k] = i < ni ? row[i] : 0; }
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
#pragma unroll
            for (int k = 0; k < 4; k++) { const int u = __shfl_up(v[k], o, 64); if (lane >= o) v[k] += u; }
        }
        int *ws = wsum[it & 1];
        if (lane == 63) {
#pragma unroll
            for (int k = 0; k < 4; k++) ws[k * 4 + wv] = v[k];
        }
        __syncthreads();
        int run = carry;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int off = run;
#pragma unroll
            for (int w = 0; w < 4; w++) { const int s = ws[k * 4 + w]; if (w < wv) off += s; run += s; }
            const int i = base + k * 256 + t;
            if (i < ni) row[i] = v[k] + off;
        }
        carry = run;
    }
}
/* column sums in bands of PG_BAND rows: band totals, their exclusive scan per column, then the running sums */
#define PG_BAND 32
__global__ __launch_bounds__(256) void k_pg_bandsum(int *bs, const int *D, int ni, int nj)
{
    const int i = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (i >= ni) return;
    int s = 0;
    for (int j = b * PG_BAND; j < min((b + 1) * PG_BAND, nj); j++) s += D[(size_t)j * ni + i];
    bs[(size_t)b * ni + i] = s;
}
__global__ __launch_bounds__(256) void k_pg_bandscan(int *bs, int ni, int nbands)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= ni) return;
    int run = 0;
    for (int b = 0; b < nbands; b++) { const int v = bs[(size_t)b * ni + i]; bs[(size_t)b * ni + i] = run; run += v; }
}
__global__ __launch_bounds__(256) void k_pg_colscan(int *D, const int *bs, int ni, int nj, unsigned mask)
{
    const int i = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (i >= ni) return;
    int run = bs[(size_t)b * ni + i];
    for (int j = b * PG_BAND; j < min((b + 1) * PG_BAND, nj); j++) { run += D[(size_t)j * ni + i]; D[(size_t)j * ni + i] = (int)((unsigned)run & mask); }
}
/* d_z: the plane's stream (device), d_tpos / d_tpos32: bit position of every tile header (the host walk's 64-bit ones, or the device walk's 32-bit ones: the
 * other pointer is NULL), d_D: ni*nj ints out (the plane), d_bs: ceil(nj/32)*ni ints */
extern "C" int packhip_pg_decode2(int *d_D, int *d_bs, const unsigned *d_z, const unsigned long long *d_tpos, const unsigned *d_tpos32, int ni, int nj, int nbits, int wide);
extern "C" int packhip_pg_decode(int *d_D, int *d_bs, const unsigned *d_z, const unsigned long long *d_tpos, int ni, int nj, int nbits, int wide)
{
    return packhip_pg_decode2(d_D, d_bs, d_z, d_tpos, NULL, ni, nj, nbits, wide);
}
extern "C" int packhip_pg_decode2(int *d_D, int *d_bs, const unsigned *d_z, const unsigned long long *d_tpos, const unsigned *d_tpos32, int ni, int nj, int nbits, int wide)
{
    int ntx; const unsigned ntiles = (unsigned)pg_ntiles(ni, nj, &ntx);
    hipStream_t st = STREAM;
    hipLaunchKernelGGL(k_pg_dec_tiles, dim3((ntiles + 255) / 256), dim3(256), 0, st, d_D, d_z, d_tpos, d_tpos32, ni, nj, ntx, ntiles, wide ? 5 : 4);
    hipLaunchKernelGGL(k_pg_dec_prefix, dim3((ni + nj + 255) / 256), dim3(256), 0, st, d_D, d_z, ni, nj, nbits);
    hipLaunchKernelGGL(k_pg_rowscan, dim3(nj), dim3(256), 0, st, d_D, ni);
    const int nbands = (nj + PG_BAND - 1) / PG_BAND;
    hipLaunchKernelGGL(k_pg_bandsum, dim3((ni + 255) / 256, nbands), dim3(256), 0, st, d_bs, d_D, ni, nj);
    hipLaunchKernelGGL(k_pg_bandscan, dim3((ni + 255) / 256), dim3(256), 0, st, d_bs, ni, nbands);
    hipLaunchKernelGGL(k_pg_colscan, dim3((ni + 255) / 256, nbands), dim3(256), 0, st, d_D, d_bs, ni, nj, wide ? 0xFFFFFFFFu : 0xFFu);
    return chk32("pg_decode");
}
/* ---- the sign run lengths (unpack1bitRLE, armn_compress_32.c:904-955) on the device ----------------------------------------------------------------------
 * The stream is a sequence of tokens of EIGHT bits each (only the very last one may be shorter): flag 0 + seven raw sign bits, or flag 1 + value bit + a 6-bit
 * count (63: a run of 255 with the value of the last token that carried one).  So token k is byte k of the MSB-first stream; where its points start is a prefix sum
 * of the run lengths, and "the value of the last token that carried one" is a prefix operation too: both ride one scan over {length, last defined value}
 * (k_rle_blocks: 4096 tokens per block; k_rle_scan: the blocks' aggregates on one block; k_rle_emit: every thread re-reads its 16 tokens and ORs their bits into the
 * zeroed mask).  The host did this on one thread: ~8 ms for the 26 M points of a 7200 x 3601 field with mixed signs. */
#define RLE_TPT 16
__device__ __forceinline__ unsigned rle_byte(const unsigned *z, size_t k, size_t nbytes) { return k < nbytes ? (z[k >> 2] >> (24 - 8 * (unsigned)(k & 3))) & 0xFFu : 0u; }
__device__ __forceinline__ uint2 rle_tok(unsigned b)           /* {points covered, 2 | value when the token defines the running value} */
{
    if (!(b & 0x80u)) return make_uint2(7u, 0u);
    const unsigned cnt = b & 63u;
    return cnt == 63u ? make_uint2(255u, 0u) : make_uint2(cnt, 2u | ((b >> 6) & 1u));
}
__device__ __forceinline__ uint2 rle_comb(uint2 a, uint2 b) { return make_uint2(a.x + b.x, b.y ? b.y : a.y); }
/* aggregate of this thread's tokens, inclusive scan over the block in sh[] (returns the thread's EXCLUSIVE prefix within the block; sh[255] = block aggregate) */
__device__ __forceinline__ uint2 rle_block_scan(uint2 mine, uint2 *sh)
{
    const unsigned t = threadIdx.x;
    sh[t] = mine;
    __syncthreads();
    for (unsigned o = 1; o < 256; o <<= 1) {
        uint2 v = sh[t];
        if (t >= o) v = rle_comb(sh[t - o], v);
        __syncthreads();
        sh[t] = v;
        __syncthreads();
    }
    return t ? sh[t - 1] : make_uint2(0u, 0u);
}
__global__ __launch_bounds__(256) void k_rle_blocks(uint2 *bagg, const unsigned *z, size_t nbytes)
{
    __shared__ uint2 sh[256];
    const size_t k0 = ((size_t)blockIdx.x * 256 + threadIdx.x) * RLE_TPT;
    uint2 a = make_uint2(0u, 0u);
#pragma unroll
    for (int q = 0; q < RLE_TPT; q++) if (k0 + q < nbytes) a = rle_comb(a, rle_tok(rle_byte(z, k0 + q, nbytes)));
    rle_block_scan(a, sh);
    if (threadIdx.x == 255) bagg[blockIdx.x] = sh[255];
}
/* exclusive prefix of the blocks' aggregates (one block); *bad = 1 when the stream's tokens cover fewer than npts points */
__global__ __launch_bounds__(1024) void k_rle_scan(uint2 *bpre, const uint2 *bagg, unsigned nblk, unsigned npts, int *bad)
{
    __shared__ uint2 sh[1024];
    __shared__ uint2 carry;
    if (threadIdx.x == 0) carry = make_uint2(0u, 0u);
    __syncthreads();
    for (unsigned base = 0; base < nblk; base += 1024) {
        const unsigned k = base + threadIdx.x;
        uint2 v = k < nblk ? bagg[k] : make_uint2(0u, 0u);
        sh[threadIdx.x] = v;
        __syncthreads();
        for (unsigned o = 1; o < 1024; o <<= 1) {
            uint2 w = sh[threadIdx.x];
            if (threadIdx.x >= o) w = rle_comb(sh[threadIdx.x - o], w);
            __syncthreads();
            sh[threadIdx.x] = w;
            __syncthreads();
        }
        const uint2 c = carry;
        if (k < nblk) bpre[k] = rle_comb(c, threadIdx.x ? sh[threadIdx.x - 1] : make_uint2(0u, 0u));
        __syncthreads();
        if (threadIdx.x == 1023) carry = rle_comb(c, sh[1023]);
        __syncthreads();
    }
    if (threadIdx.x == 0) *bad = carry.x < npts ? 1 : 0;
}
__device__ __forceinline__ void rle_or(unsigned *mask, unsigned i, unsigned pattern, unsigned nbits_)   /* `nbits_` <= 32 bits of `pattern` (LSB = point i) into the mask */
{
    if (!pattern) return;
    const unsigned sh = i & 31u;
    atomicOr(&mask[i >> 5], pattern << sh);
    if (sh && sh + nbits_ > 32u) atomicOr(&mask[(i >> 5) + 1], pattern >> (32u - sh));
}
__global__ __launch_bounds__(256) void k_rle_emit(unsigned *mask, const uint2 *bpre, const unsigned *z, size_t nbytes, unsigned npts)
{
    __shared__ uint2 sh[256];
    const size_t k0 = ((size_t)blockIdx.x * 256 + threadIdx.x) * RLE_TPT;
    unsigned bytes[RLE_TPT];
    uint2 a = make_uint2(0u, 0u);
#pragma unroll
    for (int q = 0; q < RLE_TPT; q++) { bytes[q] = rle_byte(z, k0 + q, nbytes); if (k0 + q < nbytes) a = rle_comb(a, rle_tok(bytes[q])); }
    const uint2 pre = rle_comb(bpre[blockIdx.x], rle_block_scan(a, sh));
    unsigned i = pre.x, last = pre.y ? (pre.y & 1u) : 1u;          /* (the reference's running value starts as 1) */
#pragma unroll
    for (int q = 0; q < RLE_TPT; q++) {
        if (k0 + q >= nbytes || i >= npts) break;
        const unsigned b = bytes[q];
        if (!(b & 0x80u)) {
            const unsigned lim = min(7u, npts - i);
            unsigned pat = 0;                                       /* raw bit j (MSB first behind the flag) is point i + j */
            for (unsigned j = 0; j < lim; j++) pat |= ((b >> (6u - j)) & 1u) << j;
            rle_or(mask, i, pat, lim);
            i += 7u;
        } else {
            const unsigned cnt = b & 63u, run = cnt == 63u ? 255u : cnt, bit = cnt == 63u ? last : (b >> 6) & 1u;
            if (cnt != 63u) last = (b >> 6) & 1u;
            if (bit) {
                unsigned p = i;
                const unsigned end = min(i + run, npts);
                while (p < end) { const unsigned m = min(32u - (p & 31u), end - p); rle_or(mask, p, m == 32u ? 0xFFFFFFFFu : (1u << m) - 1u, m); p += m; }
            }
            i += run;
        }
    }
}
/* d_mask: npts / 32 + 2 words, zeroed here; d_z: the sub-stream's words on the device, nbytes of them meaningful; d_work: 16 bytes per 4096 tokens + 16;
 * *h_bad (host): 1 when the tokens cover fewer than npts points (a damaged stream).  Synchronises once (the flag). */
extern "C" size_t packhip_a32_rle_work_bytes(size_t nbytes) { return 16 * ((nbytes + 256 * RLE_TPT - 1) / (256 * RLE_TPT) + 1) + 64; }
extern "C" int packhip_a32_rle_decode(unsigned *d_mask, const unsigned *d_z, size_t nbytes, size_t npts, void *d_work, int *h_bad)
{
    const unsigned nblk = (unsigned)((nbytes + 256 * RLE_TPT - 1) / (256 * RLE_TPT));
    if (nblk == 0 || npts == 0 || npts >= (1ull << 31)) return -1;
    uint2 *bagg = (uint2 *)d_work, *bpre = bagg + nblk;
    int *d_bad = (int *)(bpre + nblk);
    if (hipMemsetAsync(d_mask, 0, 4 * (npts / 32 + 2), STREAM) != hipSuccess) return -1;
    hipLaunchKernelGGL(k_rle_blocks, dim3(nblk), dim3(256), 0, STREAM, bagg, d_z, nbytes);
    hipLaunchKernelGGL(k_rle_scan, dim3(1), dim3(1024), 0, STREAM, bpre, bagg, nblk, (unsigned)npts, d_bad);
    hipLaunchKernelGGL(k_rle_emit, dim3(nblk), dim3(256), 0, STREAM, d_mask, bpre, d_z, nbytes, (unsigned)npts);
    if (chk32("k_rle")) return -1;
    if (hipMemcpyAsync(h_bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, STREAM) != hipSuccess || hipStreamSynchronize(STREAM) != hipSuccess) return -1;
    return 0;
}

/* ---- the sign run lengths the other way: pack1bitRLE (armn_compress_32.c:827-901) on the device (round 5) -------------------------------------------------------
 * The coder is a state machine over positions, but its state is small.  At a position f it looks at the run of equal signs that STARTS at f (length L up to the
 * next sign change): L < 8 -> one RAW token (flag 0 + the next seven signs, whatever runs they belong to), next position f + 7; otherwise count tokens that end
 * exactly at the run's end E -- {1, sign, 62}, then 0xFF (255 points) while more than 256 remain, then {1, sign, 62} while 63 or more remain, then {1, sign, rem} --
 * except that a last remainder of 1 .. 7 points goes out as a RAW token at E - rem, which reaches 7 - rem points into the following runs.  So the only thing a run
 * needs to know from everything in front of it is HOW MANY OF ITS FIRST POINTS a RAW token has already taken: d in 0 .. 6 (a run shorter than d is swallowed and
 * hands d - length on).  Every run is a map {0..6} -> {0..6}; the entry state of every run is a prefix "sum" of those maps under composition (21 bits per map), and
 * with it the run's tokens -- all 8 bits wide, only the field's very last one may be a shorter RAW -- follow in closed form: a second prefix sum gives their byte
 * positions.  Kernels: change bits per mask word -> run starts (scan 1) -> maps -> entry states (scan 2, composition) -> token counts -> byte offsets (scan 3) ->
 * one thread per stream word writes its four tokens.  The host did this walk on a downloaded mask: 1 - 3.4 ms per 7200 x 3601 field of both signs. */
#define RE_ITEMS 8
__device__ __forceinline__ unsigned re_block_excl_scan(unsigned v, unsigned *sh, unsigned &total)      /* 256 threads; returns the exclusive prefix of v */
{
    const unsigned t = threadIdx.x;
    sh[t] = v;
    __syncthreads();
    for (unsigned o = 1; o < 256; o <<= 1) {
        unsigned w = sh[t];
        if (t >= o) w += sh[t - o];
        __syncthreads();
        sh[t] = w;
        __syncthreads();
    }
    total = sh[255];
    const unsigned r = t ? sh[t - 1] : 0u;
    __syncthreads();
    return r;
}
/* exclusive scan of nblk block sums on one block (in place: agg[k] := sum of agg[0 .. k - 1]); total[0] := the sum of all */
__global__ __launch_bounds__(1024) void k_re_scan_sums(unsigned *agg, unsigned nblk, unsigned *total)
{
    __shared__ unsigned sh[1024];
    __shared__ unsigned carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (unsigned base = 0; base < nblk; base += 1024) {
        const unsigned k = base + threadIdx.x;
        const unsigned v = k < nblk ? agg[k] : 0u;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (unsigned o = 1; o < 1024; o <<= 1) {
            unsigned w = sh[threadIdx.x];
            if (threadIdx.x >= o) w += sh[threadIdx.x - o];
            __syncthreads();
            sh[threadIdx.x] = w;
            __syncthreads();
        }
        const unsigned c = carry;
        if (k < nblk) agg[k] = c + sh[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = c + sh[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) total[0] = carry;
}
/* bit i of the result: point 32 w + i starts a new run (its sign differs from the point in front of it); point 0 never does */
__device__ __forceinline__ unsigned re_change_bits(const unsigned *mask, size_t w, size_t nwords, unsigned npts)
{
    if (w >= nwords) return 0u;
    const unsigned m = mask[w], prev = w ? mask[w - 1] >> 31 : (m & 1u);
    unsigned cb = m ^ ((m << 1) | prev);
    const size_t base = 32 * w;
    if (base + 32 > npts) cb &= (npts > base) ? (0xFFFFFFFFu >> (32u - (unsigned)(npts - base))) : 0u;
    return cb;
}
__global__ __launch_bounds__(256) void k_re_count(unsigned *bagg, const unsigned *mask, size_t nwords, unsigned npts)
{
    __shared__ unsigned sh[256];
    const size_t w0 = ((size_t)blockIdx.x * 256 + threadIdx.x) * RE_ITEMS;
    unsigned c = 0;
#pragma unroll
    for (int q = 0; q < RE_ITEMS; q++) c += __popc(re_change_bits(mask, w0 + q, nwords, npts));
    unsigned tot;
    re_block_excl_scan(c, sh, tot);
    if (threadIdx.x == 0) bagg[blockIdx.x] = tot;
}
/* starts[k] = first point of run k (starts[0] = 0, starts[nruns] = npts) */
__global__ __launch_bounds__(256) void k_re_starts(unsigned *starts, const unsigned *bpre, const unsigned *mask, size_t nwords, unsigned npts, const unsigned *nchanges)
{
    __shared__ unsigned sh[256];
    const size_t w0 = ((size_t)blockIdx.x * 256 + threadIdx.x) * RE_ITEMS;
    unsigned cb[RE_ITEMS], c = 0;
#pragma unroll
    for (int q = 0; q < RE_ITEMS; q++) { cb[q] = re_change_bits(mask, w0 + q, nwords, npts); c += __popc(cb[q]); }
    unsigned tot;
    unsigned k = 1u + bpre[blockIdx.x] + re_block_excl_scan(c, sh, tot);
#pragma unroll
    for (int q = 0; q < RE_ITEMS; q++) {
        unsigned b = cb[q];
        while (b) { const unsigned i = __ffs(b) - 1u; b &= b - 1u; starts[k++] = (unsigned)(32 * (w0 + q)) + i; }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { starts[0] = 0u; starts[nchanges[0] + 1u] = npts; }
}
/* what a run of `len` points does when its first d points are already taken: returns the state it hands on; ntok: tokens it emits */
__device__ __forceinline__ unsigned re_run(unsigned len, unsigned d, unsigned &ntok)
{
    if (d >= len) { ntok = 0; return d - len; }
    const unsigned L = len - d;
    if (L < 8u) { ntok = 1; return 7u - L; }
    if (L <= 62u) { ntok = 1; return 0u; }
    const unsigned rem0 = L - 62u, n255 = rem0 > 256u ? (rem0 - 256u + 254u) / 255u : 0u;
    unsigned rem = rem0 - 255u * n255, n62 = 0;
    if (rem >= 63u) { n62 = (rem - 63u) / 62u + 1u; rem -= 62u * n62; }
    ntok = 1u + n255 + n62 + (rem ? 1u : 0u);
    return (rem >= 1u && rem <= 7u) ? 7u - rem : 0u;
}
__device__ __forceinline__ unsigned re_map_of(unsigned len)      /* seven 3-bit states: entry d -> exit */
{
    unsigned m = 0, nt;
#pragma unroll
    for (unsigned d = 0; d < 7u; d++) m |= re_run(len, d, nt) << (3u * d);
    return m;
}
#define RE_ID_MAP 0x1AC688u                               /* d -> d: 0 | 1 << 3 | 2 << 6 | ... | 6 << 18 */
__device__ __forceinline__ unsigned re_compose(unsigned first, unsigned then)      /* d -> then[first[d]] */
{
    unsigned m = 0;
#pragma unroll
    for (unsigned d = 0; d < 7u; d++) m |= ((then >> (3u * ((first >> (3u * d)) & 7u))) & 7u) << (3u * d);
    return m;
}
__device__ __forceinline__ unsigned re_block_excl_scan_maps(unsigned v, unsigned *sh, unsigned &total)
{
    const unsigned t = threadIdx.x;
    sh[t] = v;
    __syncthreads();
    for (unsigned o = 1; o < 256; o <<= 1) {
        unsigned w = sh[t];
        if (t >= o) w = re_compose(sh[t - o], w);
        __syncthreads();
        sh[t] = w;
        __syncthreads();
    }
    total = sh[255];
    const unsigned r = t ? sh[t - 1] : RE_ID_MAP;
    __syncthreads();
    return r;
}
#define RE_RUNS 4                                          /* runs per thread */
__global__ __launch_bounds__(256) void k_re_maps(unsigned *bmap, const unsigned *starts, unsigned nruns)
{
    __shared__ unsigned sh[256];
    const unsigned k0 = (blockIdx.x * 256u + threadIdx.x) * RE_RUNS;
    unsigned m = RE_ID_MAP;
#pragma unroll
    for (unsigned q = 0; q < RE_RUNS; q++) if (k0 + q < nruns) m = re_compose(m, re_map_of(starts[k0 + q + 1] - starts[k0 + q]));
    unsigned tot;
    re_block_excl_scan_maps(m, sh, tot);
    if (threadIdx.x == 0) bmap[blockIdx.x] = tot;
}
__global__ __launch_bounds__(1024) void k_re_scan_maps(unsigned *bmap, unsigned nblk)      /* in place: bmap[k] := composition of the blocks in front of k */
{
    __shared__ unsigned sh[1024];
    __shared__ unsigned carry;
    if (threadIdx.x == 0) carry = RE_ID_MAP;
    __syncthreads();
    for (unsigned base = 0; base < nblk; base += 1024) {
        const unsigned k = base + threadIdx.x;
        const unsigned v = k < nblk ? bmap[k] : RE_ID_MAP;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (unsigned o = 1; o < 1024; o <<= 1) {
            unsigned w = sh[threadIdx.x];
            if (threadIdx.x >= o) w = re_compose(sh[threadIdx.x - o], w);
            __syncthreads();
            sh[threadIdx.x] = w;
            __syncthreads();
        }
        const unsigned c = carry;
        if (k < nblk) bmap[k] = re_compose(c, threadIdx.x ? sh[threadIdx.x - 1] : RE_ID_MAP);
        __syncthreads();
        if (threadIdx.x == 1023) carry = re_compose(c, sh[1023]);
        __syncthreads();
    }
}
/* entry state of every run (packed with its token count: state << 28 | ntok) and the blocks' token sums */
__global__ __launch_bounds__(256) void k_re_states(unsigned *runinfo, unsigned *bsum, const unsigned *bmap, const unsigned *starts, unsigned nruns)
{
    __shared__ unsigned sh[256];
    const unsigned k0 = (blockIdx.x * 256u + threadIdx.x) * RE_RUNS;
    unsigned len[RE_RUNS], m = RE_ID_MAP;
#pragma unroll
    for (unsigned q = 0; q < RE_RUNS; q++) { len[q] = k0 + q < nruns ? starts[k0 + q + 1] - starts[k0 + q] : 0u; if (k0 + q < nruns) m = re_compose(m, re_map_of(len[q])); }
    unsigned tot;
    const unsigned pre = re_compose(bmap[blockIdx.x], re_block_excl_scan_maps(m, sh, tot));
    unsigned d = pre & 7u, nsum = 0;                       /* the coder starts in state 0 */
#pragma unroll
    for (unsigned q = 0; q < RE_RUNS; q++) {
        if (k0 + q >= nruns) break;
        unsigned nt;
        const unsigned dn = re_run(len[q], d, nt);
        runinfo[k0 + q] = d << 28 | nt;
        nsum += nt; d = dn;
    }
    unsigned tt;
    re_block_excl_scan(nsum, sh, tt);
    if (threadIdx.x == 0) bsum[blockIdx.x] = tt;
}
/* first token (byte) index of every run */
__global__ __launch_bounds__(256) void k_re_offsets(unsigned *tokoff, const unsigned *bpre, const unsigned *runinfo, unsigned nruns)
{
    __shared__ unsigned sh[256];
    const unsigned k0 = (blockIdx.x * 256u + threadIdx.x) * RE_RUNS;
    unsigned nt[RE_RUNS], nsum = 0;
#pragma unroll
    for (unsigned q = 0; q < RE_RUNS; q++) { nt[q] = k0 + q < nruns ? runinfo[k0 + q] & 0x0FFFFFFFu : 0u; nsum += nt[q]; }
    unsigned tt;
    unsigned o = bpre[blockIdx.x] + re_block_excl_scan(nsum, sh, tt);
#pragma unroll
    for (unsigned q = 0; q < RE_RUNS; q++) { if (k0 + q < nruns) tokoff[k0 + q] = o; o += nt[q]; }
}
__device__ __forceinline__ unsigned re_sign(const unsigned *mask, unsigned i) { return (mask[i >> 5] >> (i & 31u)) & 1u; }
__device__ __forceinline__ unsigned re_raw(const unsigned *mask, unsigned f, unsigned npts)      /* 0 flag + the seven signs from f on, first point on top; points beyond the field: 0 */
{
    unsigned b = 0;
    for (unsigned j = 0; j < 7u; j++) b = b << 1 | (f + j < npts ? re_sign(mask, f + j) : 0u);
    return b;
}
/* token j of a run that starts at `start` with `len` points, entered in state d */
__device__ __forceinline__ unsigned re_token(const unsigned *mask, unsigned start, unsigned len, unsigned d, unsigned j, unsigned npts)
{
    const unsigned f = start + d, L = len - d, s = re_sign(mask, f);
    if (L < 8u) return re_raw(mask, f, npts);
    if (L <= 62u) return 0x80u | s << 6 | L;
    const unsigned rem0 = L - 62u, n255 = rem0 > 256u ? (rem0 - 256u + 254u) / 255u : 0u;
    unsigned rem = rem0 - 255u * n255, n62 = 0;
    if (rem >= 63u) { n62 = (rem - 63u) / 62u + 1u; rem -= 62u * n62; }
    if (j == 0u) return 0x80u | s << 6 | 62u;
    if (j <= n255) return 0xFFu;
    if (j <= n255 + n62) return 0x80u | s << 6 | 62u;
    return rem >= 8u ? (0x80u | s << 6 | rem) : re_raw(mask, start + len - rem, npts);
}
/* one thread per stream word: tokens 4 w .. 4 w + 3; res[0] = tokens, res[1] = stream bits (the last token may be a RAW of fewer than seven points) */
__global__ __launch_bounds__(256) void k_re_emit(unsigned *z, unsigned *res, const unsigned *tokoff, const unsigned *runinfo, const unsigned *starts,
                                                 const unsigned *mask, unsigned nruns, const unsigned *ntok_total, unsigned npts)
{
    const unsigned T = ntok_total[0];
    const unsigned w = blockIdx.x * 256u + threadIdx.x;
    if (w == 0u) {
        /* the last token: the last run that emits one */
        unsigned bits = 8u * T;
        unsigned lo = 0, hi = nruns;                       /* last run k with tokoff[k] < T and ntok > 0: search the largest k with tokoff[k] <= T - 1 */
        if (T) {
            while (lo + 1 < hi) { const unsigned mid = (lo + hi) >> 1; if (tokoff[mid] <= T - 1u) lo = mid; else hi = mid; }
            const unsigned info = runinfo[lo], d = info >> 28, st = starts[lo], len = starts[lo + 1] - st, j = T - 1u - tokoff[lo];
            const unsigned L = len - d;
            unsigned rawpos = 0xFFFFFFFFu;                 /* where the last token reads its seven signs, if it is a RAW */
            if (L < 8u) rawpos = st + d;
            else if (L > 62u) {
                const unsigned rem0 = L - 62u, n255 = rem0 > 256u ? (rem0 - 256u + 254u) / 255u : 0u;
                unsigned rem = rem0 - 255u * n255, n62 = 0;
                if (rem >= 63u) { n62 = (rem - 63u) / 62u + 1u; rem -= 62u * n62; }
                if (j > n255 + n62 && rem >= 1u && rem <= 7u) rawpos = st + len - rem;
            }
            if (rawpos != 0xFFFFFFFFu && rawpos + 7u > npts) bits = 8u * (T - 1u) + 1u + (npts - rawpos);
        }
        res[0] = T; res[1] = bits;
    }
    if (4u * w >= T) return;
    unsigned t = 4u * w;
    unsigned lo = 0, hi = nruns;
    while (lo + 1 < hi) { const unsigned mid = (lo + hi) >> 1; if (tokoff[mid] <= t) lo = mid; else hi = mid; }
    unsigned k = lo;
    while ((runinfo[k] & 0x0FFFFFFFu) == 0u || tokoff[k] + (runinfo[k] & 0x0FFFFFFFu) <= t) k++;      /* (runs without tokens share their neighbour's offset) */
    unsigned word = 0;
    for (unsigned q = 0; q < 4u; q++, t++) {
        unsigned byte = 0;
        if (t < T) {
            while ((runinfo[k] & 0x0FFFFFFFu) == 0u || tokoff[k] + (runinfo[k] & 0x0FFFFFFFu) <= t) k++;
            const unsigned st = starts[k];
            byte = re_token(mask, st, starts[k + 1] - st, runinfo[k] >> 28, t - tokoff[k], npts);
        }
        word |= byte << (24u - 8u * q);
    }
    z[w] = word;
}
/* d_z: the stream (ceil(tokens / 4) words are written: at most npts / 7 / 4 + 2); d_mask: npts / 32 + 1 words (bits beyond npts in the last word may hold anything);
 * d_work: packhip_a32_rle_enc_work_bytes(npts); h_bits (host): the stream's bit count.  Synchronises twice (the number of runs, the result). */
extern "C" size_t packhip_a32_rle_enc_work_bytes(size_t npts)
{
    const size_t nwords = npts / 32 + 1, nb1 = (nwords + 256 * RE_ITEMS - 1) / (256 * RE_ITEMS);
    return 4 * (nb1 + 16) + 64;                            /* the run arrays are sized once the number of runs is known: packhip_a32_rle_encode takes them separately */
}
extern "C" int packhip_a32_rle_encode(unsigned *d_z, const unsigned *d_mask, size_t npts, void *d_work, void *(*grow)(int slot, size_t bytes), unsigned long long *h_bits)
{
    if (npts == 0 || npts >= (1ull << 28)) return -1;      /* (token counts ride in 28 bits) */
    const size_t nwords = (npts + 31) / 32;
    const unsigned nb1 = (unsigned)((nwords + 256 * RE_ITEMS - 1) / (256 * RE_ITEMS));
    unsigned *bagg = (unsigned *)d_work, *tot = bagg + nb1 + 1;          /* tot[0] changes, tot[1] tokens, tot[2..3] result */
    hipStream_t st = STREAM;
    hipLaunchKernelGGL(k_re_count, dim3(nb1), dim3(256), 0, st, bagg, d_mask, nwords, (unsigned)npts);
    hipLaunchKernelGGL(k_re_scan_sums, dim3(1), dim3(1024), 0, st, bagg, nb1, tot);
    unsigned nch = 0;
    if (chk32("k_re_count") || hipMemcpyAsync(&nch, tot, 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return -1;
    const unsigned nruns = nch + 1u, nb2 = (nruns + 256 * RE_RUNS - 1) / (256 * RE_RUNS);
    unsigned *starts = (unsigned *)grow(0, 4 * ((size_t)nruns + 2)), *runinfo = (unsigned *)grow(1, 4 * ((size_t)nruns + 1)), *tokoff = (unsigned *)grow(2, 4 * ((size_t)nruns + 1)),
             *bmap = (unsigned *)grow(3, 8 * ((size_t)nb2 + 1));
    if (!starts || !runinfo || !tokoff || !bmap) return -1;
    unsigned *bsum = bmap + nb2 + 1;
    hipLaunchKernelGGL(k_re_starts, dim3(nb1), dim3(256), 0, st, starts, bagg, d_mask, nwords, (unsigned)npts, tot);
    hipLaunchKernelGGL(k_re_maps, dim3(nb2), dim3(256), 0, st, bmap, starts, nruns);
    hipLaunchKernelGGL(k_re_scan_maps, dim3(1), dim3(1024), 0, st, bmap, nb2);
    hipLaunchKernelGGL(k_re_states, dim3(nb2), dim3(256), 0, st, runinfo, bsum, bmap, starts, nruns);
    hipLaunchKernelGGL(k_re_scan_sums, dim3(1), dim3(1024), 0, st, bsum, nb2, tot + 1);
    hipLaunchKernelGGL(k_re_offsets, dim3(nb2), dim3(256), 0, st, tokoff, bsum, runinfo, nruns);
    const unsigned maxw = (unsigned)(npts / 7 / 4 + 2);
    hipLaunchKernelGGL(k_re_emit, dim3((maxw + 255) / 256), dim3(256), 0, st, d_z, tot + 2, tokoff, runinfo, starts, d_mask, nruns, tot + 1, (unsigned)npts);
    unsigned res[2] = {0, 0};
    if (chk32("k_re_emit") || hipMemcpyAsync(res, tot + 2, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return -1;
    *h_bits = res[1];
    return 0;
}

extern "C" int packhip_a32_combine(float *d_f, const int *d_expo, const int *d_mant, const unsigned *d_smask, size_t n, int nbits, unsigned exp_min, int code_signe, int have_expo)
{
    hipLaunchKernelGGL(k_a32_combine, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, STREAM, (unsigned *)d_f, d_expo, d_mant, d_smask, n, nbits, exp_min, code_signe, have_expo);
    return chk32("a32_combine");
}
