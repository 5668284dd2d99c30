/*
 * pack_kernels.hip -- HIP kernels (gfx950) for librmn's field packers: compact_float, compact_integer,
 * c_float_packer and armn_compress (COMPRESS), plus their C-ABI launchers (packhip_*).
 *
 * All streams are MSB-first token streams in host-endian uint32 words (include/bitPacking.h `stuff`).
 * Mapping to the GPU: a token stream with a FIXED token width is embarrassingly parallel once every
 * thread owns whole 32-bit output words (no atomics); the variable-width armn_compress stream is
 * sized by an exclusive scan of per-tile bit lengths and then emitted independently per tile into a
 * zeroed buffer with atomicOr (tiles share boundary words).  Integer / bit work only: HBM bound.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "packhip_shim.h"

#pragma clang fp contract(off)

extern "C" void *ezhip_get_stream(void);
#define STREAM ((hipStream_t)ezhip_get_stream())
static int chk(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", what, hipGetErrorString(e)); return -1; }
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* reductions                                                                                   */
/* ------------------------------------------------------------------------------------------ */
__device__ __forceinline__ unsigned f2key(float f) { unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float key2f(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

/* stats[0] = min key, stats[1] = max key, stats[2] = OR of raw words (compact_integer auto width) */
template <int KIND>   /* 0: float keys, 1: uint32, 2: int32 (biased) */
__global__ __launch_bounds__(256) void k_stats(unsigned *stats, const unsigned *a, size_t n, int stride, int has_missing, float tag)
{
    unsigned kmin = 0xffffffffu, kmax = 0u, kor = 0u;
#define STAT1(raw_) do { unsigned raw = (raw_), k; bool skip = false;                                              \
        if (KIND == 0) { float f = __uint_as_float(raw); skip = has_missing && f == tag; k = f2key(f); }            \
        else if (KIND == 1) k = raw; else k = raw ^ 0x80000000u;                                                   \
        if (!skip) { kmin = min(kmin, k); kmax = max(kmax, k); kor |= raw; } } while (0)
    size_t i0 = 0;
    /* 16 B per lane, four loads in flight; normally ONE pass per block (grid = n / 4096: the many-short-blocks shape
     * of k_cf_pack16 streams at 6.9 TB/s where a 2048-block grid-stride loop reached 3.9) */
    if (stride == 1 && ((uintptr_t)a & 15) == 0) {
        const uint4 *a4 = (const uint4 *)a;
        size_t n4 = n / 4, step = (size_t)gridDim.x * 256;
        size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
        for (; i + 3 * step < n4; i += 4 * step) {
            uint4 u = a4[i], v = a4[i + step], x = a4[i + 2 * step], y = a4[i + 3 * step];
            STAT1(u.x); STAT1(u.y); STAT1(u.z); STAT1(u.w); STAT1(v.x); STAT1(v.y); STAT1(v.z); STAT1(v.w);
            STAT1(x.x); STAT1(x.y); STAT1(x.z); STAT1(x.w); STAT1(y.x); STAT1(y.y); STAT1(y.z); STAT1(y.w);
        }
        for (; i < n4; i += step) { uint4 u = a4[i]; STAT1(u.x); STAT1(u.y); STAT1(u.z); STAT1(u.w); }
        i0 = n4 * 4;
    }
    for (size_t i = i0 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) STAT1(a[i * stride]);
#undef STAT1
    for (int off = 32; off > 0; off >>= 1) {
        kmin = min(kmin, (unsigned)__shfl_down((int)kmin, off, 64));
        kmax = max(kmax, (unsigned)__shfl_down((int)kmax, off, 64));
        kor |= (unsigned)__shfl_down((int)kor, off, 64);
    }
    /* per-block partial triple, reduced by k_stats_final: same-address device atomics cost ~23 ns EACH on
     * MI355X (they resolve on the memory side of the 8 L2s): 570 us with one triple per wave, 60 us with
     * one per block, nothing this way */
    __shared__ unsigned sh[3][4];
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = kmin; sh[1][threadIdx.x >> 6] = kmax; sh[2][threadIdx.x >> 6] = kor; }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned *p = stats + PACKHIP_STATS_PARTIALS + 3 * blockIdx.x;
        p[0] = min(min(sh[0][0], sh[0][1]), min(sh[0][2], sh[0][3]));
        p[1] = max(max(sh[1][0], sh[1][1]), max(sh[1][2], sh[1][3]));
        p[2] = sh[2][0] | sh[2][1] | sh[2][2] | sh[2][3];
    }
}

/* reduction of the per-block partial triples by one block of 256 threads; result in stats[0..2] */
__device__ __forceinline__ void stats_reduce_partials(unsigned *stats, int nb)
{
    unsigned kmin = 0xffffffffu, kmax = 0u, kor = 0u;
    const unsigned *p = stats + PACKHIP_STATS_PARTIALS;
    for (int b = threadIdx.x; b < nb; b += 256) { kmin = min(kmin, p[3 * b]); kmax = max(kmax, p[3 * b + 1]); kor |= p[3 * b + 2]; }
    for (int off = 32; off > 0; off >>= 1) {
        kmin = min(kmin, (unsigned)__shfl_down((int)kmin, off, 64));
        kmax = max(kmax, (unsigned)__shfl_down((int)kmax, off, 64));
        kor |= (unsigned)__shfl_down((int)kor, off, 64);
    }
    __shared__ unsigned sh[3][4];
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = kmin; sh[1][threadIdx.x >> 6] = kmax; sh[2][threadIdx.x >> 6] = kor; }
    __syncthreads();
    if (threadIdx.x == 0) {
        stats[0] = min(min(sh[0][0], sh[0][1]), min(sh[0][2], sh[0][3]));
        stats[1] = max(max(sh[1][0], sh[1][1]), max(sh[1][2], sh[1][3]));
        stats[2] = sh[2][0] | sh[2][1] | sh[2][2] | sh[2][3];
        stats[3] = 0;
    }
    __syncthreads();
}
__global__ __launch_bounds__(256) void k_stats_final(unsigned *stats, int nb) { stats_reduce_partials(stats, nb); }

/* d_stats: PACKHIP_STATS_WORDS u32 of device scratch; result in d_stats[0..2] */
static int stats_blocks(size_t n) { int nb = (int)((n + 4095) / 4096); if (nb > PACKHIP_STATS_MAXBLK) nb = PACKHIP_STATS_MAXBLK; return nb < 1 ? 1 : nb; }
/* per-block partials only; returns the number of partials (the consumer reduces them: k_cf_header) or -1 */
extern "C" int packhip_stats_partials(unsigned *d_stats, const void *d_a, size_t n, int stride, int kind, int has_missing, float tag)
{
    int nb = stats_blocks(n);
    if (kind == 0) hipLaunchKernelGGL(k_stats<0>, dim3(nb), dim3(256), 0, STREAM, d_stats, (const unsigned *)d_a, n, stride, has_missing, tag);
    else if (kind == 1) hipLaunchKernelGGL(k_stats<1>, dim3(nb), dim3(256), 0, STREAM, d_stats, (const unsigned *)d_a, n, stride, 0, 0.f);
    else hipLaunchKernelGGL(k_stats<2>, dim3(nb), dim3(256), 0, STREAM, d_stats, (const unsigned *)d_a, n, stride, 0, 0.f);
    return chk("k_stats") ? -1 : nb;
}
extern "C" int packhip_stats(unsigned *d_stats, const void *d_a, size_t n, int stride, int kind, int has_missing, float tag)
{
    int nb = packhip_stats_partials(d_stats, d_a, n, stride, kind, has_missing, tag);
    if (nb < 0) return -1;
    hipLaunchKernelGGL(k_stats_final, dim3(1), dim3(256), 0, STREAM, d_stats, nb);
    return chk("k_stats_final");
}

/* ------------------------------------------------------------------------------------------ */
/* compact_float: header (compact.tmplc:210-290) on one lane, then the token stream               */
/* ------------------------------------------------------------------------------------------ */
/* one thread block per field: blockIdx.x selects {params, header, stats} through the strides (all 0 for a single field) */
__global__ __launch_bounds__(256) void k_cf_header(packhip_cf_params *pp, unsigned *hdr, unsigned *stats, int npartials, unsigned n, int bs, int style, int has_missing,
                                                   size_t hdr_stride_words, size_t stats_stride_words)
{
    pp += blockIdx.x; hdr += blockIdx.x * hdr_stride_words; stats += blockIdx.x * stats_stride_words;
    if (npartials > 0) stats_reduce_partials(stats, npartials);     /* fused final reduction of k_stats (one launch less) */
    if (threadIdx.x != 0) return;
    double minF = (double)key2f(stats[0]), maxF = (double)key2f(stats[1]);
    unsigned missingToken = (bs != 32) ? ~(0xFFFFFFFFu << bs) : ~0u;
    int too_large = (maxF > 1.0e+38) || (minF < -1.0e+38);
    unsigned long long range = (unsigned long long)__double_as_longlong((maxF - minF) * 2);
    unsigned long long minbits = (unsigned long long)__double_as_longlong(minF);
    range &= 0xFFF0000000000000ull;
    double ranged = __longlong_as_double((long long)range);
    unsigned tempInt = (ranged == 0) ? 0 : (unsigned)(long long)((maxF - minF) * ldexp(1.0, bs) / ranged);
    if (tempInt == missingToken && has_missing) range += 0x0010000000000000ull;
    int rexpo = (int)((range >> 52) & 0x7FF);
    int tempExpo = (ranged == 0) ? 0 : (rexpo - 1023);
    unsigned scaledExpOfMinFloat = (unsigned)((int)((minbits >> 52) & 0x7FF) - 1023 + 1024 - 48);
    unsigned scaledExpOfRange = (unsigned)(tempExpo - bs);
    unsigned signOfMinFloat = (minF < 0) ? 1 : 0;
    if (minF == 0.0) scaledExpOfMinFloat &= 0x00000111;
    unsigned headerType = (style == 1) ? (has_missing == 1 ? 0x7ef : 0x7ff) : (has_missing == 1 ? 0xfef : 0xfff);
    hdr[0] = headerType << 20 | ((n << 12) >> 12);
    hdr[1] = ((scaledExpOfRange + 4096) << 16) | ((scaledExpOfMinFloat << 4) | signOfMinFloat);
    if (minF == 0.0) hdr[2] = 0;
    else hdr[2] = 0x80000000u | ((unsigned)((minbits >> 32) & 0xFFFFF) << 11) | ((unsigned)((minbits >> 29) & 0x7) << 8);
    unsigned w3 = (unsigned)bs << 8 | ((n << 4) >> 24);
    hdr[3] = w3;       /* style 1 (offset 24): the stream then overwrites the low byte, as in the reference */
    pp->hdr3 = w3;
    pp->minF = minF;
    pp->mulFactor = ldexp(1.0, bs) / ldexp(1.0, tempExpo);
    pp->missingToken = missingToken;
    pp->too_large = too_large;
}

__device__ __forceinline__ unsigned cf_token(const float *a, size_t i, int stride, const packhip_cf_params &p, int has_missing, float tag)
{
    float x = a[i * stride];
    if (has_missing == 1 && x == tag) return p.missingToken;
    return (unsigned)(long long)(((double)x - p.minF) * p.mulFactor);
}

/* generic fixed-width stream: one thread per output word; TOK(t) yields token t (already masked to E bits
 * when it has to be).  Bits before `offset` in the first word and after the last token in the last word
 * are preserved (compact.tmplc:292-300, :329-332). */
template <class TokFn>
__device__ __forceinline__ void emit_word(unsigned *out, size_t w, size_t n, int E, int offset, TokFn TOK)
{
    const long long first_bit = (long long)w * 32, end_bit = (long long)offset + (long long)n * E;
    long long t0 = (first_bit - offset) / E;
    if (first_bit < offset) t0 = 0;
    unsigned word = 0, written = 0;           /* written: mask of bits this stream owns in the word */
    for (long long t = t0; t < (long long)n; t++) {
        long long tb = (long long)offset + t * E;            /* first stream bit of token t */
        if (tb >= first_bit + 32) break;
        unsigned long long tok = (unsigned long long)TOK((size_t)t) & (E == 32 ? 0xFFFFFFFFull : ((1ull << E) - 1));
        long long sh = (first_bit + 32) - (tb + E);          /* left shift that aligns the token's LSB in this word */
        unsigned piece, mask;
        unsigned long long full = (E == 32) ? 0xFFFFFFFFull : ((1ull << E) - 1);
        if (sh >= 0) { piece = (unsigned)(tok << sh); mask = (unsigned)(full << sh); }
        else { piece = (unsigned)(tok >> (-sh)); mask = (unsigned)(full >> (-sh)); }
        word |= piece; written |= mask;
    }
    (void)end_bit;
    if (written != 0xFFFFFFFFu) word |= out[w] & ~written;   /* partial first / last word: keep the neighbours */
    out[w] = word;
}

__global__ __launch_bounds__(256) void k_cf_pack(unsigned *out, const float *a, size_t n, int E, int offset, int stride,
                                                 const packhip_cf_params *pp, int has_missing, float tag, size_t w0, size_t nwords)
{
    size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= nwords) return;
    const packhip_cf_params p = *pp;
    emit_word(out, w0 + k, n, E, offset, [&](size_t t) { return cf_token(a, t, stride, p, has_missing, tag); });
}

/* fast path: 16-bit slots, offset 0, stride 1: 8 floats -> 4 words per thread (16 B in / 8 B out per token pair) */
__global__ __launch_bounds__(256) void k_cf_pack16(uint4 *out, const float4 *a, size_t n8, const packhip_cf_params *pp)
{
    size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n8) return;
    const double minF = pp->minF, mul = pp->mulFactor;
    float4 x = a[2 * k], y = a[2 * k + 1];
#define TK(v) ((unsigned)(long long)(((double)(v) - minF) * mul) & 0xFFFFu)
    uint4 o;
    o.x = TK(x.x) << 16 | TK(x.y); o.y = TK(x.z) << 16 | TK(x.w);
    o.z = TK(y.x) << 16 | TK(y.y); o.w = TK(y.z) << 16 | TK(y.w);
#undef TK
    out[k] = o;
}

extern "C" int packhip_cf_header(packhip_cf_params *d_pp, unsigned *d_hdr, unsigned *d_stats, int npartials, unsigned n, int bs, int style, int has_missing)
{
    hipLaunchKernelGGL(k_cf_header, dim3(1), dim3(256), 0, STREAM, d_pp, d_hdr, d_stats, npartials, n, bs, style, has_missing, (size_t)0, (size_t)0);
    return chk("k_cf_header");
}
/* nfields headers in one launch: field f uses d_pp[f], d_hdr + f * hdr_stride_words, d_stats + f * stats_stride_words */
extern "C" int packhip_cf_header_batch(packhip_cf_params *d_pp, unsigned *d_hdr, size_t hdr_stride_words, unsigned *d_stats, size_t stats_stride_words,
                                       int npartials, unsigned n, int bs, int style, int has_missing, int nfields)
{
    hipLaunchKernelGGL(k_cf_header, dim3(nfields), dim3(256), 0, STREAM, d_pp, d_hdr, d_stats, npartials, n, bs, style, has_missing, hdr_stride_words, stats_stride_words);
    return chk("k_cf_header");
}

extern "C" int packhip_cf_pack(unsigned *d_out, const float *d_a, size_t n, int E, int offset, int stride,
                               const packhip_cf_params *d_pp, int has_missing, float tag)
{
    if (n == 0) return 0;
    size_t w0 = (size_t)offset / 32, w1 = ((size_t)offset + n * (size_t)E + 31) / 32;
    if (E == 16 && offset == 0 && stride == 1 && !has_missing && ((uintptr_t)d_a % 16 == 0) && ((uintptr_t)d_out % 16 == 0)) {
        size_t n8 = n / 8;
        if (n8) hipLaunchKernelGGL(k_cf_pack16, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, STREAM, (uint4 *)d_out, (const float4 *)d_a, n8, d_pp);
        size_t done_words = n8 * 4;
        if (done_words < w1)    /* tail words through the generic kernel */
            hipLaunchKernelGGL(k_cf_pack, dim3((unsigned)((w1 - done_words + 255) / 256)), dim3(256), 0, STREAM, d_out, d_a, n, E, offset, stride, d_pp, 0, tag, done_words, w1 - done_words);
        return chk("k_cf_pack16");
    }
    hipLaunchKernelGGL(k_cf_pack, dim3((unsigned)((w1 - w0 + 255) / 256)), dim3(256), 0, STREAM, d_out, d_a, n, E, offset, stride, d_pp, has_missing, tag, w0, w1 - w0);
    return chk("k_cf_pack");
}

/* token t of a fixed-width stream */
__device__ __forceinline__ unsigned read_token(const unsigned *in, size_t t, int E, int offset)
{
    unsigned long long bit = (unsigned long long)offset + (unsigned long long)t * E;
    size_t w = (size_t)(bit >> 5);
    int sh = (int)(bit & 31);
    unsigned long long two = ((unsigned long long)in[w] << 32) | (sh + E > 32 ? in[w + 1] : 0u);
    return (unsigned)((two << sh) >> (64 - E));
}

/* FLOAT_UNPACK, compact.tmplc:336-425 */
__global__ __launch_bounds__(256) void k_cf_unpack(float *a, const unsigned *in, size_t n, int tokenSize, int offset, int stride,
                                                   double minF, double mulFactor, unsigned missingToken, int has_missing, float tag)
{
    size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    int sig = tokenSize > 32 ? 32 : tokenSize;
    unsigned tok = read_token(in, t, sig, offset);       /* tokenSize > 32 is not produced by the packer */
    float v;
    if (has_missing == 1 && tok == missingToken) v = tag;
    else if (tok == 0) v = (float)minF;
    else v = (float)((tok * mulFactor) * 1.0000000000001 + minF);
    a[t * stride] = v;
}
extern "C" int packhip_cf_unpack(float *d_a, const unsigned *d_in, size_t n, int tokenSize, int offset, int stride,
                                 double minF, double mulFactor, unsigned missingToken, int has_missing, float tag)
{
    if (!n) return 0;
    hipLaunchKernelGGL(k_cf_unpack, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, STREAM, d_a, d_in, n, tokenSize, offset, stride, minF, mulFactor, missingToken, has_missing, tag);
    return chk("k_cf_unpack");
}

/* ------------------------------------------------------------------------------------------ */
/* compact_integer (compact_integer.c:69-170, 245-303)                                           */
/* ------------------------------------------------------------------------------------------ */
__global__ __launch_bounds__(256) void k_ci_pack(unsigned *out, const unsigned *a, size_t n, int E, int offset, int stride,
                                                 int with_header, int is_signed, unsigned minv, int shift, unsigned positiveMask, size_t w0, size_t nwords)
{
    size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= nwords) return;
    emit_word(out, w0 + k, n, E, offset, [&](size_t t) {
        unsigned x = a[t * stride];
        if (!with_header) return x + positiveMask;
        if (is_signed) return (unsigned)(((int)x - (int)minv) >> shift);
        return (x - minv) >> shift;
    });
}
extern "C" int packhip_ci_pack(unsigned *d_out, const void *d_a, size_t n, int E, int offset, int stride,
                               int with_header, int is_signed, unsigned minv, int shift, unsigned positiveMask)
{
    if (!n) return 0;
    size_t w0 = (size_t)offset / 32, w1 = ((size_t)offset + n * (size_t)E + 31) / 32;
    hipLaunchKernelGGL(k_ci_pack, dim3((unsigned)((w1 - w0 + 255) / 256)), dim3(256), 0, STREAM, d_out, (const unsigned *)d_a, n, E, offset, stride, with_header, is_signed, minv, shift, positiveMask, w0, w1 - w0);
    return chk("k_ci_pack");
}
__global__ __launch_bounds__(256) void k_ci_unpack(unsigned *a, const unsigned *in, size_t n, int E, int offset, int stride,
                                                   int with_header, unsigned minv, int shift, unsigned negMask)
{
    size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    unsigned tok = read_token(in, t, E > 32 ? 32 : E, offset);
    a[t * stride] = with_header ? (tok << shift) + minv : tok + negMask;
}
extern "C" int packhip_ci_unpack(void *d_a, const unsigned *d_in, size_t n, int E, int offset, int stride, int with_header, unsigned minv, int shift, unsigned negMask)
{
    if (!n) return 0;
    hipLaunchKernelGGL(k_ci_unpack, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, STREAM, (unsigned *)d_a, d_in, n, E, offset, stride, with_header, minv, shift, negMask);
    return chk("k_ci_unpack");
}

/* ------------------------------------------------------------------------------------------ */
/* c_float_packer (float_packer.c:128-202): two 16-bit tokens per word                           */
/* ------------------------------------------------------------------------------------------ */
__device__ __forceinline__ int aligned_mantissa(int src, int maxExp)
{
    int mantis = (1 << 23) | (0x7FFFFF & src);
    int e = (src >> 23) & 0xFF;
    int shift = maxExp - e;
    if (shift > 31) shift = 31;
    mantis = mantis >> shift;
    if (src >> 31) mantis = -mantis;
    return mantis;
}
__global__ __launch_bounds__(256) void k_fp_pack(unsigned *stream, const int *src, int npts, int MaxExp, int Minimum, int Round, int Shift2, int Mask)
{
    int w = blockIdx.x * 256 + threadIdx.x;
    if (2 * w >= npts) return;
    unsigned word = 0;
    for (int h = 0; h < 2; h++) {
        int k = 2 * w + h;
        unsigned tok = 0;
        if (k < npts) {
            int m = aligned_mantissa(src[k], MaxExp);
            m = m - Minimum; m = m + Round; m = m >> Shift2;
            if (m > Mask) m = Mask;
            tok = (unsigned)m & 0xFFFFu;
        }
        word = (word << 16) | tok;
    }
    stream[w] = word;
}
extern "C" int packhip_fp_pack(int *d_stream, const float *d_src, int npts, int MaxExp, int Minimum, int Round, int Shift2, int Mask)
{
    if (npts <= 0) return 0;
    int nw = (npts + 1) / 2;
    hipLaunchKernelGGL(k_fp_pack, dim3((nw + 255) / 256), dim3(256), 0, STREAM, (unsigned *)d_stream, (const int *)d_src, npts, MaxExp, Minimum, Round, Shift2, Mask);
    return chk("k_fp_pack");
}
/* float_unpacker_1 (float_packer.c:40-113) */
__global__ __launch_bounds__(256) void k_fp_unpack(float *dest, const unsigned *stream, int npts, int maxExp, int minimum, int shift2)
{
    int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= npts) return;
    if (maxExp == 0) { dest[k] = 0.0f; return; }
    unsigned word = stream[k >> 1];
    int mantis = (int)((k & 1) ? (word & 0xFFFF) : (word >> 16));
    mantis = mantis << shift2;
    mantis = mantis + minimum;
    int sgn = (mantis >> 31) & 1;
    if (sgn) mantis = -mantis;
    if (mantis > 0xFFFFFF) mantis = 0xFFFFFF;
    int ti = (mantis & 0x7FFFFF) | (maxExp << 23);
    ti = ti | (sgn << 31);
    if (mantis & (1 << 23)) dest[k] = __int_as_float(ti);
    else {
        int t2 = maxExp << 23;
        t2 = t2 | (sgn << 31);
        t2 = t2 & (~((mantis << 8) >> 31));
        dest[k] = __int_as_float(ti) - __int_as_float(t2);
    }
}
extern "C" int packhip_fp_unpack(float *d_dest, const int *d_stream, int npts, int maxExp, int minimum, int shift2)
{
    if (npts <= 0) return 0;
    hipLaunchKernelGGL(k_fp_unpack, dim3((npts + 255) / 256), dim3(256), 0, STREAM, d_dest, (const unsigned *)d_stream, npts, maxExp, minimum, shift2);
    return chk("k_fp_unpack");
}

/* ------------------------------------------------------------------------------------------ */
/* armn_compress COMPRESS (c_zfstlib.c:469-588 MINIMUM, :646-789 PARALLELOGRAM)                   */
/* ------------------------------------------------------------------------------------------ */
/* token k of the packed-word layout: two 16-bit tokens per word, first in the high half (the reference
 * gets the same ushort array by swapping halves on little-endian hosts, c_zfstlib.c:119-126) */
__device__ __forceinline__ int tokat(const unsigned *w, size_t k) { unsigned x = w[k >> 1]; return (int)((k & 1) ? (x & 0xFFFFu) : (x >> 16)); }

__device__ __forceinline__ int bitlen(unsigned v) { return v ? 32 - __clz((int)v) : 0; }

/* OR `nbits` bits of `tok` into a zeroed MSB-first stream at absolute bit position `pos` */
__device__ __forceinline__ void put_bits(unsigned *z, unsigned long long pos, unsigned tok, int nbits)
{
    size_t w = (size_t)(pos >> 5);
    int sh = (int)(pos & 31);
    unsigned long long v = (unsigned long long)(tok & (nbits == 32 ? 0xFFFFFFFFu : ((1u << nbits) - 1))) << (64 - nbits - sh);
    unsigned hi = (unsigned)(v >> 32), lo = (unsigned)v;
    if (hi) atomicOr(&z[w], hi);
    if (lo) atomicOr(&z[w + 1], lo);
}

struct Tile { int i0, j0, m, n; };   /* 0-based origin and extent of a tile */
__device__ __forceinline__ Tile tile_of(long long t, int ni, int nj, int istep, int origin, int ntx)
{
    Tile T;
    int ty = (int)(t / ntx), tx = (int)(t % ntx);
    T.i0 = origin + tx * istep; T.j0 = origin + ty * istep;
    T.m = min(istep, ni - T.i0); T.n = min(istep, nj - T.j0);
    return T;
}
__device__ __forceinline__ int lorenzo(const unsigned *w, int ni, int i, int j)
{   /* d(i,j) = u(i,j) - (u(i,j-1) + u(i-1,j) - u(i-1,j-1)), from ORIGINAL values (:691-696) */
    size_t k = (size_t)ni * j + i;
    return tokat(w, k) - (tokat(w, k - ni) + tokat(w, k - 1) - tokat(w, k - 1 - ni));
}

/* pass A: per-tile nbits_needed (uint8) + global flag "some |d| > 65535" (parallelogram only) */
template <int PARA>
__global__ __launch_bounds__(256) void k_armn_tiles(unsigned char *nb, unsigned *gt16, const unsigned *w, int ni, int nj, int istep, int ntx, long long ntiles)
{
    long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= ntiles) return;
    Tile T = tile_of(t, ni, nj, istep, PARA ? 1 : 0, ntx);
    unsigned need;
    if (PARA) {
        /* the 3 x 3 differences need a 4 x 4 patch of tokens (one row above, one column left): all 16 loads are issued
         * before any is used (with run-time loop bounds the 36 loads of the 9 Lorenzo stencils chained one memory round
         * trip per token: 51 -> 24 us per cfg5 field); clipped tiles repeat an edge token and mask the result */
        int u[4][4];
#pragma unroll
        for (int n = 0; n < 4; n++)
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const int jj = T.j0 - 1 + min(n, T.n), ii = T.i0 - 1 + min(m, T.m);
                u[n][m] = tokat(w, (size_t)ni * jj + ii);
            }
        int mx = 0;
#pragma unroll
        for (int n = 1; n < 4; n++)
#pragma unroll
            for (int m = 1; m < 4; m++) {
                const int d = abs(u[n][m] - (u[n - 1][m] + u[n][m - 1] - u[n - 1][m - 1]));
                if (n <= T.n && m <= T.m) mx = max(mx, d);
            }
        if (mx > 65535) atomicOr(gt16, 1u);
        need = (unsigned)bitlen((unsigned)mx);
    } else {
        int lo = 65536, hi = -1;
        for (int n = 0; n < T.n; n++) for (int m = 0; m < T.m; m++) { int v = tokat(w, (size_t)ni * (T.j0 + n) + T.i0 + m); lo = min(lo, v); hi = max(hi, v); }
        need = (unsigned)bitlen((unsigned)(hi - lo));
    }
    if (need == 16) need = 15;
    nb[t] = (unsigned char)need;
}

__device__ __forceinline__ unsigned tile_bits(int PARA, unsigned need, int cnt, int container, int nbits)
{
    if (PARA) return (unsigned)container + (need == 0 ? 0u : (unsigned)cnt * (need == 15 ? 17u : need + 1u));
    if (need == 0) return 4u + (unsigned)nbits;
    if (need == 15) return 4u + 16u * (unsigned)cnt;
    return 4u + (unsigned)nbits + need * (unsigned)cnt;
}

#define SCAN_TPB 256
#define SCAN_ITEMS 4      /* tiles per thread (1024 tiles per block: the emit kernel's LDS image stays at 20 KB, 5 blocks per CU) */
/* pass B: per-block total of tile bit lengths */
template <int PARA>
__global__ __launch_bounds__(SCAN_TPB) void k_armn_blocksum(unsigned long long *bsum, const unsigned char *nb, const unsigned *gt16,
                                                            int ni, int nj, int istep, int ntx, long long ntiles, int nbits)
{
    __shared__ unsigned long long sh[SCAN_TPB / 64];
    const int container = (PARA && nbits >= 15 && *gt16) ? 5 : 4;
    long long base = ((long long)blockIdx.x * SCAN_TPB + threadIdx.x) * SCAN_ITEMS;
    unsigned long long s = 0;
    for (int q = 0; q < SCAN_ITEMS; q++) {
        long long t = base + q;
        if (t < ntiles) { Tile T = tile_of(t, ni, nj, istep, PARA ? 1 : 0, ntx); s += tile_bits(PARA, nb[t], T.m * T.n, container, nbits); }
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) bsum[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}
/* pass S: exclusive scan of the block totals (single block, sequential chunks; nblocks is small) */
__global__ __launch_bounds__(1024) void k_armn_scan(unsigned long long *bsum, int nblocks, unsigned long long *total)
{
    /* one block of 1024 threads: each thread owns a contiguous run of block sums (serial), one shuffle + LDS scan over
     * the 1024 run totals, then the runs are rewritten as exclusive prefixes (a 256-thread chunked scan with 16 barriers
     * per chunk took 14 us for the 2813 sums of a cfg5 field) */
    __shared__ unsigned long long wsum[16];
    const int per = (nblocks + 1023) / 1024;
    const int i0 = threadIdx.x * per, i1 = min(i0 + per, nblocks);
    unsigned long long run = 0;
    for (int i = i0; i < i1; i++) run += bsum[i];
    unsigned long long incl = run;                               /* inclusive scan of `run` over the block */
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int off = 1; off < 64; off <<= 1) {
        unsigned long long o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
    }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    unsigned long long base = 0;
    for (int k = 0; k < wv; k++) base += wsum[k];
    unsigned long long excl = base + incl - run;
    for (int i = i0; i < i1; i++) { unsigned long long v = bsum[i]; bsum[i] = excl; excl += v; }
    if (threadIdx.x == 1023) *total = base + incl;
}
/* MSB-first bit writer over a zeroed stream: a thread owns a CONTIGUOUS bit range, so only its first and
 * last (partial) words can be shared with neighbours and need an atomic OR; interior words are plain stores.
 * (One atomicOr per token cost 1.2 ms per 26 M-token field.)  The stream is either global memory or the
 * thread block's LDS staging buffer. */
template <class P, int SCOPE>     /* P: `unsigned *` (global stream) or an LDS pointer (staging buffer) */
struct BitOutT {
    P z; size_t w; unsigned cw; int used; bool first_shared;
    __device__ __forceinline__ void init(P zz, unsigned long long pos) { z = zz; w = (size_t)(pos >> 5); used = (int)(pos & 31); cw = 0; first_shared = used != 0; }
    __device__ __forceinline__ void flush_full()
    {
        if (first_shared) { (void)__hip_atomic_fetch_or(&z[w], cw, __ATOMIC_RELAXED, SCOPE); first_shared = false; } else z[w] = cw;
        w++; cw = 0; used = 0;
    }
    __device__ __forceinline__ void put(unsigned tok, int nbits)
    {
        tok &= (nbits == 32) ? 0xFFFFFFFFu : ((1u << nbits) - 1);
        while (nbits > 0) {
            int take = min(nbits, 32 - used);
            unsigned piece = (take == 32) ? tok : ((tok >> (nbits - take)) & ((1u << take) - 1));
            cw |= piece << (32 - used - take);
            used += take; nbits -= take;
            if (used == 32) flush_full();
        }
    }
    __device__ __forceinline__ void finish() { if (used > 0 && cw) (void)__hip_atomic_fetch_or(&z[w], cw, __ATOMIC_RELAXED, SCOPE); }
};
typedef __attribute__((address_space(3))) unsigned *lds_uptr;

/* pass C: emission.  Each thread rescans its SCAN_ITEMS consecutive tiles from the block base and writes them.
 * A thread's bit range is contiguous but the 64 lanes of a wave write ~80 B apart: straight to global memory that is
 * ~40 cache lines per store instruction (255 us per cfg5 field).  So the block's whole bit range (<= EMIT_LDS_WORDS
 * words) is assembled in LDS, word-aligned with the global stream, and copied out with coalesced stores; only the
 * block's first and last words are shared with the neighbouring blocks (atomicOr into the zeroed stream).  Blocks whose
 * range does not fit (MINIMUM method with wide tokens) write directly. */
#define EMIT_LDS_WORDS 6144
template <int PARA>
__global__ __launch_bounds__(SCAN_TPB) void k_armn_emit(unsigned *z, const unsigned long long *bsum, const unsigned char *nb, const unsigned *gt16,
                                                        const unsigned *w, int ni, int nj, int istep, int ntx, long long ntiles, int nbits,
                                                        unsigned long long body_start)
{
    __shared__ unsigned long long sh[SCAN_TPB];
    __shared__ unsigned stage[EMIT_LDS_WORDS];
    const int container = (PARA && nbits >= 15 && *gt16) ? 5 : 4;
    long long base = ((long long)blockIdx.x * SCAN_TPB + threadIdx.x) * SCAN_ITEMS;
    unsigned long long mine = 0;
    for (int q = 0; q < SCAN_ITEMS; q++) {
        long long t = base + q;
        if (t < ntiles) { Tile T = tile_of(t, ni, nj, istep, PARA ? 1 : 0, ntx); mine += tile_bits(PARA, nb[t], T.m * T.n, container, nbits); }
    }
    sh[threadIdx.x] = mine;
    __syncthreads();
    for (int off = 1; off < SCAN_TPB; off <<= 1) {
        unsigned long long add = threadIdx.x >= off ? sh[threadIdx.x - off] : 0;
        __syncthreads();
        sh[threadIdx.x] += add;
        __syncthreads();
    }
    const unsigned long long blk_pos = body_start + bsum[blockIdx.x], blk_bits = sh[SCAN_TPB - 1];
    const size_t w0 = (size_t)(blk_pos >> 5);                                   /* first global word of the block */
    const size_t nw = (size_t)((blk_pos + blk_bits + 31) >> 5) - w0;           /* words the block touches */
    const bool staged = nw <= EMIT_LDS_WORDS;                                   /* block-uniform */
    if (staged) {
        for (size_t k = threadIdx.x; k < nw; k += SCAN_TPB) stage[k] = 0;
        __syncthreads();
    }
    if (base < ntiles) {
        const unsigned long long mypos = blk_pos + sh[threadIdx.x] - mine;
        auto emit_tiles = [&](auto &bo) {
            for (int q = 0; q < SCAN_ITEMS; q++) {
                long long t = base + q;
                if (t >= ntiles) break;
                Tile T = tile_of(t, ni, nj, istep, PARA ? 1 : 0, ntx);
                unsigned need = nb[t];
                if (PARA) {
                    bo.put(need, container);
                    if (need) {
                        int width = need == 15 ? 17 : (int)need + 1;
                        for (int n = 0; n < T.n; n++) for (int m = 0; m < T.m; m++) bo.put((unsigned)lorenzo(w, ni, T.i0 + m, T.j0 + n), width);
                    }
                } else {
                    int lo = 65536;
                    for (int n = 0; n < T.n; n++) for (int m = 0; m < T.m; m++) lo = min(lo, tokat(w, (size_t)ni * (T.j0 + n) + T.i0 + m));
                    bo.put(need, 4);
                    if (need != 15) bo.put((unsigned)lo, nbits);
                    if (need) {
                        int width = need == 15 ? 16 : (int)need;
                        for (int n = 0; n < T.n; n++) for (int m = 0; m < T.m; m++) {
                            int v = tokat(w, (size_t)ni * (T.j0 + n) + T.i0 + m);
                            bo.put((unsigned)(need == 15 ? v : v - lo), width);
                        }
                    }
                }
            }
            bo.finish();
        };
        if (staged) {
            BitOutT<lds_uptr, __HIP_MEMORY_SCOPE_WORKGROUP> bo;
            bo.init((lds_uptr)stage, mypos - ((unsigned long long)w0 << 5));
            emit_tiles(bo);
        } else {
            BitOutT<unsigned *, __HIP_MEMORY_SCOPE_AGENT> bo;
            bo.init(z, mypos);
            emit_tiles(bo);
        }
    }
    if (staged) {
        __syncthreads();
        for (size_t k = threadIdx.x; k < nw; k += SCAN_TPB) {
            const unsigned v = stage[k];
            if (k == 0 || k == nw - 1) { if (v) atomicOr(&z[w0 + k], v); }     /* shared with the neighbouring blocks */
            else z[w0 + k] = v;
        }
    }
}
/* pass C for the PARALLELOGRAM method, one LANE PER TOKEN.  The per-thread sequential writer above spends ~80
 * instructions per token (Lorenzo difference, variable-width put loop) with every lane walking its own 72 tokens.
 * Here phase 0 turns the per-thread scan into per-tile bit offsets (LDS), then every lane takes one token slot
 * (tile, row, column), computes difference, width and absolute bit position in closed form and ORs its bits into the
 * block's LDS image of the stream (ds_or_b32, at most two words); the image is copied out with coalesced stores. */
#define TOK_LDS_WORDS (SCAN_TPB * SCAN_ITEMS * 5 + 64)        /* >= tiles per block x (5 + 9 x 17) bits / 32 + alignment */
__global__ __launch_bounds__(SCAN_TPB) void k_armn_emit_tok(unsigned *z, const unsigned long long *bsum, const unsigned char *nb, const unsigned *gt16,
                                                            const unsigned *w, int ni, int nj, int ntx, long long ntiles, int nbits,
                                                            unsigned long long body_start)
{
    __shared__ unsigned long long sh[SCAN_TPB];
    __shared__ unsigned stage[TOK_LDS_WORDS];
    __shared__ unsigned tmeta[SCAN_TPB * SCAN_ITEMS];      /* bit offset of the tile in the block (28 bits) | need << 28 */
    __shared__ unsigned tcoord[SCAN_TPB * SCAN_ITEMS];     /* tile column | tile row << 16 (the launcher checks ntx, nty < 65536) */
    const int container = (nbits >= 15 && *gt16) ? 5 : 4;
    const long long tbase = (long long)blockIdx.x * SCAN_TPB * SCAN_ITEMS;
    const long long base = tbase + (long long)threadIdx.x * SCAN_ITEMS;
    unsigned long long mine = 0;
    unsigned lbits[SCAN_ITEMS], tcs[SCAN_ITEMS], needs[SCAN_ITEMS];
    {   /* one 32-bit division per thread (ntiles < 2^31, checked by the launcher), then (tile column, tile row) by increments */
        unsigned ty = (unsigned)base / (unsigned)ntx, tx = (unsigned)base - ty * (unsigned)ntx;
        for (int q = 0; q < SCAN_ITEMS; q++) {
            const long long t = base + q;
            lbits[q] = 0; needs[q] = 0; tcs[q] = tx | (ty << 16);
            if (t < ntiles) {
                const int tm = min(3, ni - 1 - (int)tx * 3), tn = min(3, nj - 1 - (int)ty * 3);
                needs[q] = nb[t];
                lbits[q] = tile_bits(1, needs[q], tm * tn, container, nbits);
            }
            mine += lbits[q];
            if (++tx == (unsigned)ntx) { tx = 0; ty++; }
        }
    }
    sh[threadIdx.x] = mine;
    __syncthreads();
    for (int off = 1; off < SCAN_TPB; off <<= 1) {
        unsigned long long add = threadIdx.x >= off ? sh[threadIdx.x - off] : 0;
        __syncthreads();
        sh[threadIdx.x] += add;
        __syncthreads();
    }
    const unsigned long long blk_pos = body_start + bsum[blockIdx.x], blk_bits = sh[SCAN_TPB - 1];
    const size_t w0 = (size_t)(blk_pos >> 5);
    const size_t nw = (size_t)((blk_pos + blk_bits + 31) >> 5) - w0;           /* <= TOK_LDS_WORDS by construction */
    {
        unsigned off = (unsigned)(blk_pos & 31) + (unsigned)(sh[threadIdx.x] - mine);
        for (int q = 0; q < SCAN_ITEMS; q++) {
            tmeta[threadIdx.x * SCAN_ITEMS + q] = off | (needs[q] << 28);
            tcoord[threadIdx.x * SCAN_ITEMS + q] = tcs[q];
            off += lbits[q];
        }
    }
    for (size_t k = threadIdx.x; k < nw; k += SCAN_TPB) stage[k] = 0;
    __syncthreads();
    const int ntl = (int)min((long long)(SCAN_TPB * SCAN_ITEMS), ntiles - tbase);
    /* Four token slots per lane and iteration, 32-bit token indices (the launcher checks ni * nj < 2^31) and 16-bit
     * loads: token k of the two-per-word layout (first token in the HIGH half) is the little-endian halfword k ^ 1. */
    auto or_bits = [&](unsigned bitpos, unsigned val, int width) {
        const unsigned wi = bitpos >> 5; const int sh_ = (int)(bitpos & 31);
        const unsigned long long v = (unsigned long long)val << (64 - width - sh_);
        if ((unsigned)(v >> 32)) atomicOr(&stage[wi], (unsigned)(v >> 32));
        if ((unsigned)v) atomicOr(&stage[wi + 1], (unsigned)v);
    };
    const unsigned short *w16 = (const unsigned short *)w;
    /* Token order of the loop: tile-internal row n outermost, then the block's tiles left to right, then the column m
     * inside the tile.  Consecutive lanes are then consecutive tokens of ONE field row, so each of the four stencil
     * loads of a wave is a contiguous ~128-byte segment (1-2 cache lines).  In tile-major order (9 tokens of a tile on
     * 9 lanes) a load touched 3 rows x ~6 lines and the loads alone cost 48 of the kernel's 100 us.  Four columns per
     * lane and iteration keep 16 loads in flight. */
    const int ncols = 3 * ntl;                      /* token columns of the block (clipped tiles have dead columns) */
    for (int n_ = 0; n_ < 3; n_++) {
        for (int c0 = threadIdx.x; c0 < ncols; c0 += 4 * SCAN_TPB) {
            unsigned meta[4]; int idx[4]; bool live[4]; unsigned kk[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int c = c0 + q * SCAN_TPB;
                const bool have = c < ncols;
                const unsigned cc = (unsigned)(have ? c : 0);
                const unsigned tl = (cc * 21846u) >> 16;             /* cc / 3 for cc < 2^14 (at most 3072 columns per block) */
                const int m_ = (int)(cc - 3u * tl);
                meta[q] = tmeta[tl];
                const unsigned tc = tcoord[tl];
                const int i0 = 1 + (int)(tc & 0xFFFFu) * 3, j0 = 1 + (int)(tc >> 16) * 3;
                const int tm = min(3, ni - i0), tn = min(3, nj - j0);
                live[q] = have && (meta[q] >> 28) != 0 && m_ < tm && n_ < tn;
                idx[q] = n_ * tm + m_;
                kk[q] = (unsigned)ni * (unsigned)(j0 + min(n_, tn - 1)) + (unsigned)(i0 + min(m_, tm - 1));   /* clamped: always a valid token */
                if (have && n_ == 0 && m_ == 0) or_bits(meta[q] & 0x0FFFFFFFu, meta[q] >> 28, container);      /* the tile's container field */
            }
            int u00[4], u01[4], u10[4], u11[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const unsigned k = kk[q], kn = k - (unsigned)ni;
                u11[q] = w16[k ^ 1u]; u10[q] = w16[(k - 1u) ^ 1u]; u01[q] = w16[kn ^ 1u]; u00[q] = w16[(kn - 1u) ^ 1u];
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if (!live[q]) continue;
                const unsigned need = meta[q] >> 28;
                const int width = need == 15 ? 17 : (int)need + 1;
                const int d = u11[q] - (u01[q] + u10[q] - u00[q]);
                or_bits((meta[q] & 0x0FFFFFFFu) + (unsigned)container + (unsigned)idx[q] * (unsigned)width,
                        (unsigned)d & ((1u << width) - 1), width);
            }
        }
    }
    __syncthreads();
    for (size_t k = threadIdx.x; k < nw; k += SCAN_TPB) {
        const unsigned v = stage[k];
        if (k == 0 || k == nw - 1) { if (v) atomicOr(&z[w0 + k], v); }     /* shared with the neighbouring blocks */
        else z[w0 + k] = v;
    }
}

/* the words that two emit blocks (or the last block and the terminator) share start at zero; every other word of the stream
 * is stored whole by exactly one block */
__global__ __launch_bounds__(256) void k_armn_zero_bounds(unsigned *z, size_t z_words, const unsigned long long *bsum, int nblocks,
                                                          const unsigned long long *total, unsigned long long body_start)
{
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= nblocks) return;
    const unsigned long long pos = body_start + bsum[b], end = body_start + (b + 1 < nblocks ? bsum[b + 1] : *total);
    if (end > pos) {
        const size_t w0 = (size_t)(pos >> 5), w1 = (size_t)((end + 31) >> 5) - 1;
        if (w0 < z_words) z[w0] = 0u;
        if (w1 < z_words) z[w1] = 0u;
    }
    if (b == nblocks - 1)                                     /* the two 16-bit terminator tokens and the byte after zlng */
        for (size_t w = (size_t)(end >> 5); w < (size_t)(end >> 5) + 3 && w < z_words; w++) z[w] = 0u;
}

/* parallelogram prefix: 3-bit container, row 1, column 1 (:712-721) */
__global__ __launch_bounds__(256) void k_armn_prefix(unsigned *z, const unsigned *w, const unsigned *gt16, int ni, int nj, int nbits, unsigned header)
{
    int t = blockIdx.x * 256 + threadIdx.x;
    if (t == 0) { z[0] = header; put_bits(z, 32, (nbits >= 15 && *gt16) ? 5u : 4u, 3); }
    if (t < ni) put_bits(z, 32 + 3 + (unsigned long long)t * nbits, (unsigned)tokat(w, t), nbits);
    else if (t < ni + nj - 1) { int j = t - ni + 1; put_bits(z, 32 + 3 + (unsigned long long)t * nbits, (unsigned)tokat(w, (size_t)ni * j), nbits); }
}

__global__ void k_set_word(unsigned *z, unsigned v) { z[0] = v; }

/* d_work: nb[ntiles] bytes, then (8-aligned) bsum[nblocks+1] u64, gt16 u32.  Returns total stream bits
 * (from the start of word 1, terminator excluded) through *h_bits after synchronising. */
extern "C" int packhip_armn_encode(unsigned *d_z, size_t z_words, const unsigned *d_words, int ni, int nj, int nbits, int minimum_method,
                                   void *d_work, unsigned long long *h_bits)
{
    const int istep = minimum_method ? 5 : 3, origin = minimum_method ? 0 : 1;
    const int ntx = (ni - origin + istep - 1) / istep, nty = (nj - origin + istep - 1) / istep;
    const long long ntiles = (long long)ntx * nty;
    const int per_block = SCAN_TPB * SCAN_ITEMS;
    const int nblocks = (int)((ntiles + per_block - 1) / per_block);
    unsigned char *nb = (unsigned char *)d_work;
    unsigned long long *bsum = (unsigned long long *)((char *)d_work + (((size_t)ntiles + 15) & ~(size_t)15));
    unsigned long long *total = bsum + nblocks;
    unsigned *gt16 = (unsigned *)(total + 1);
    hipStream_t st = STREAM;
    /* the stream is assembled with plain stores except for the words that blocks share: only those have to start at zero.
     * The lane-per-token emitter of the PARALLELOGRAM method gets them zeroed by k_armn_zero_bounds once the scan is known
     * (52 MB of memset per cfg5 field otherwise); the other emitters OR every token into a fully zeroed buffer. */
    const bool tok_emit = !minimum_method && !(getenv("EZHIP_ARMN_SEQ_EMIT") || ntx >= 65536 || nty >= 65536 || (long long)ni * nj >= (1ll << 31));
    if (!tok_emit && hipMemsetAsync(d_z, 0, z_words * 4, st) != hipSuccess) return -1;
    if (hipMemsetAsync(gt16, 0, 4, st) != hipSuccess) return -1;
    const unsigned tb = (unsigned)((ntiles + 255) / 256);
    unsigned long long body_start;
    if (minimum_method) {
        unsigned header = 3u | 0u << 4 | 5u << 7 | ((unsigned)nbits & 31u) << 10 | 1u << 15;
        hipLaunchKernelGGL(k_armn_tiles<0>, dim3(tb), dim3(256), 0, st, nb, gt16, d_words, ni, nj, istep, ntx, ntiles);
        hipLaunchKernelGGL(k_armn_blocksum<0>, dim3(nblocks), dim3(SCAN_TPB), 0, st, bsum, nb, gt16, ni, nj, istep, ntx, ntiles, nbits);
        hipLaunchKernelGGL(k_armn_scan, dim3(1), dim3(1024), 0, st, bsum, nblocks, total);
        hipLaunchKernelGGL(k_set_word, dim3(1), dim3(1), 0, st, d_z, header);
        body_start = 32;
        hipLaunchKernelGGL(k_armn_emit<0>, dim3(nblocks), dim3(SCAN_TPB), 0, st, d_z, bsum, nb, gt16, d_words, ni, nj, istep, ntx, ntiles, nbits, body_start);
    } else {
        unsigned header = 4u | 1u << 4 | 3u << 7 | ((unsigned)nbits & 31u) << 10 | 1u << 15;
        hipLaunchKernelGGL(k_armn_tiles<1>, dim3(tb), dim3(256), 0, st, nb, gt16, d_words, ni, nj, istep, ntx, ntiles);
        hipLaunchKernelGGL(k_armn_blocksum<1>, dim3(nblocks), dim3(SCAN_TPB), 0, st, bsum, nb, gt16, ni, nj, istep, ntx, ntiles, nbits);
        hipLaunchKernelGGL(k_armn_scan, dim3(1), dim3(1024), 0, st, bsum, nblocks, total);
        body_start = 32 + 3 + (unsigned long long)(ni + nj - 1) * nbits;
        if (tok_emit) {
            const size_t head_words = (size_t)(body_start >> 5) + 2;
            if (hipMemsetAsync(d_z, 0, 4 * (head_words < z_words ? head_words : z_words), st) != hipSuccess) return -1;
            hipLaunchKernelGGL(k_armn_zero_bounds, dim3((nblocks + 255) / 256), dim3(256), 0, st, d_z, z_words, bsum, nblocks, total, body_start);
        }
        hipLaunchKernelGGL(k_armn_prefix, dim3((ni + nj + 255) / 256), dim3(256), 0, st, d_z, d_words, gt16, ni, nj, nbits, header);
        if (getenv("EZHIP_ARMN_SEQ_EMIT") || ntx >= 65536 || nty >= 65536 || (long long)ni * nj >= (1ll << 31))     /* development / huge fields: the per-thread sequential writer */
            hipLaunchKernelGGL(k_armn_emit<1>, dim3(nblocks), dim3(SCAN_TPB), 0, st, d_z, bsum, nb, gt16, d_words, ni, nj, istep, ntx, ntiles, nbits, body_start);
        else
            hipLaunchKernelGGL(k_armn_emit_tok, dim3(nblocks), dim3(SCAN_TPB), 0, st, d_z, bsum, nb, gt16, d_words, ni, nj, ntx, ntiles, nbits, body_start);
    }
    if (chk("armn_encode")) return -1;
    if (!h_bits) return 0;                       /* asynchronous use: packhip_armn_commit reads the total on the device */
    unsigned long long tot = 0;
    if (hipMemcpyAsync(&tot, total, 8, hipMemcpyDeviceToHost, st) != hipSuccess) return -1;
    if (hipStreamSynchronize(st) != hipSuccess) return -1;
    *h_bits = body_start - 32 + tot;
    return 0;
}

/* Device-side epilogue of an asynchronous encode: zlng = 1 + 4 (1 + ceil(bits / 32)) from the scan total still in
 * d_work (c_zfstlib.c:160-179); when the stream is shorter than the plain 16-bit record (zlng < 1 + 2 n) its zlng bytes
 * replace the tokens at `dst` and *d_zlng = zlng, else dst keeps the plain pack and *d_zlng = -1. */
__global__ __launch_bounds__(256) void k_armn_commit(unsigned *dst, const unsigned *z, const unsigned long long *total, unsigned long long body_bits,
                                                     unsigned long long n, int *d_zlng)
{
    const unsigned long long bits = body_bits + *total;
    const long long zlng = 1 + 4 * (1 + (long long)((bits + 31) / 32));
    const bool ok = zlng < (long long)(1 + 2 * n);
    if (blockIdx.x == 0 && threadIdx.x == 0) *d_zlng = ok ? (int)zlng : -1;
    if (!ok) return;
    const size_t nwords = (size_t)((zlng + 3) / 4);
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < nwords; k += (size_t)gridDim.x * 256) dst[k] = z[k];
}
extern "C" int packhip_armn_commit(unsigned *d_dst, const unsigned *d_z, void *d_work, int ni, int nj, int nbits, int minimum_method, int *d_zlng)
{
    const int istep = minimum_method ? 5 : 3, origin = minimum_method ? 0 : 1;
    const int ntx = (ni - origin + istep - 1) / istep, nty = (nj - origin + istep - 1) / istep;
    const long long ntiles = (long long)ntx * nty;
    const int nblocks = (int)((ntiles + SCAN_TPB * SCAN_ITEMS - 1) / (SCAN_TPB * SCAN_ITEMS));
    unsigned long long *bsum = (unsigned long long *)((char *)d_work + (((size_t)ntiles + 15) & ~(size_t)15));
    const unsigned long long *total = bsum + nblocks;
    const unsigned long long body_bits = minimum_method ? 0ull : 3ull + (unsigned long long)(ni + nj - 1) * nbits;
    hipLaunchKernelGGL(k_armn_commit, dim3(2048), dim3(256), 0, STREAM, d_dst, d_z, total, body_bits, (unsigned long long)ni * nj, d_zlng);
    return chk("k_armn_commit");
}
extern "C" size_t packhip_armn_work_bytes(int ni, int nj)
{
    long long ntiles = ((long long)(ni + 2) / 3 + 1) * ((long long)(nj + 2) / 3 + 1);
    return (size_t)ntiles + 64 + 8 * ((size_t)(ntiles / (SCAN_TPB * SCAN_ITEMS)) + 8) + 64;
}
