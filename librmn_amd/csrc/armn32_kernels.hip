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
 * The sign plane's run-length coder (pack1bitRLE :827-901) is a sequential state machine over 1 bit per point: host code.
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

extern "C" int packhip_a32_combine(float *d_f, const int *d_expo, const int *d_mant, const unsigned *d_smask, size_t n, int nbits, unsigned exp_min, int code_signe, int have_expo)
{
    hipLaunchKernelGGL(k_a32_combine, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, STREAM, (unsigned *)d_f, d_expo, d_mant, d_smask, n, nbits, exp_min, code_signe, have_expo);
    return chk32("a32_combine");
}
