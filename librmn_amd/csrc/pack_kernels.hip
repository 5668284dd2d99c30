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
#include <string.h>
#include "packhip_shim.h"
#include "armn_dev.h"

#pragma clang fp contract(off)

extern "C" void *ezhip_get_stream(void);
#define STREAM ((hipStream_t)ezhip_get_stream())
extern "C" void ezhip_note_error(void);
static int chk(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { ezhip_note_error(); fprintf(stderr, "%s: %s\n", what, hipGetErrorString(e)); return -1; }
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
    /* 16 B per lane, four loads in flight, a block sweeps 16 KB in ascending order; normally ONE pass per block (grid = n / 4096: many short
     * blocks stream at 5.5 TB/s from HBM where a 2048-block grid-stride loop reached 3.9).  Plain loads in array order on purpose: the
     * pass that follows (k_cf_pack16) reads the same field again and finds it in the Infinity Cache -- with `nt` loads here this pass
     * alone is 1.4 us shorter and the pair 4.4 us longer; with the four loads a quarter of the array apart the pair is 3.9 us longer
     * (tools/ubench_pack.hip) */
    if (stride == 1 && ((uintptr_t)a & 15) == 0) {
        const uint4 *a4 = (const uint4 *)a;
        const size_t n4 = n / 4;
        for (size_t base = (size_t)blockIdx.x * 1024; base < n4; base += (size_t)gridDim.x * 1024) {
            if (base + 1024 <= n4) {
                const uint4 *q = a4 + base + threadIdx.x;
                uint4 u = q[0], v = q[256], x = q[512], y = q[768];
                STAT1(u.x); STAT1(u.y); STAT1(u.z); STAT1(u.w); STAT1(v.x); STAT1(v.y); STAT1(v.z); STAT1(v.w);
                STAT1(x.x); STAT1(x.y); STAT1(x.z); STAT1(x.w); STAT1(y.x); STAT1(y.y); STAT1(y.z); STAT1(y.w);
            } else
                for (size_t i = base + threadIdx.x; i < n4; i += 256) { uint4 u = a4[i]; STAT1(u.x); STAT1(u.y); STAT1(u.z); STAT1(u.w); }
        }
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
    int b = threadIdx.x;
    for (; b + 768 < nb; b += 1024) {           /* four triples in flight per lane: the loop is a chain of memory round trips otherwise */
        const unsigned *q = p + 3 * b;
        unsigned a0 = q[0], a1 = q[1], a2 = q[2], b0 = q[768], b1 = q[769], b2 = q[770];
        unsigned c0 = q[1536], c1 = q[1537], c2 = q[1538], d0 = q[2304], d1 = q[2305], d2 = q[2306];
        kmin = min(min(kmin, a0), min(min(b0, c0), d0)); kmax = max(max(kmax, a1), max(max(b1, c1), d1)); kor |= a2 | b2 | c2 | d2;
    }
    for (; b < nb; b += 256) { kmin = min(kmin, p[3 * b]); kmax = max(kmax, p[3 * b + 1]); kor |= p[3 * b + 2]; }
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
/* the header words and quantisation parameters from the extrema (one lane): compact.tmplc:206-290 */
__device__ __forceinline__ void cf_header_write(packhip_cf_params *pp, unsigned *hdr, double minF, double maxF, unsigned n, int bs, int style, int has_missing)
{
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
/* one thread block per field: blockIdx.x selects {params, header, stats} through the strides (all 0 for a single field) */
__global__ __launch_bounds__(256) void k_cf_header(packhip_cf_params *pp, unsigned *hdr, unsigned *stats, int npartials, unsigned n, int bs, int style, int has_missing,
                                                   size_t hdr_stride_words, size_t stats_stride_words)
{
    pp += blockIdx.x; hdr += blockIdx.x * hdr_stride_words; stats += blockIdx.x * stats_stride_words;
    if (npartials > 0) stats_reduce_partials(stats, npartials);     /* fused final reduction of k_stats (one launch less) */
    if (threadIdx.x != 0) return;
    cf_header_write(pp, hdr, (double)key2f(stats[0]), (double)key2f(stats[1]), n, bs, style, has_missing);
}

/* ---- compact_double (compact.c:28-32: the same template on REAL*8 arrays) -------------------------------------------- */
/* min / max of a double array: per-block pairs, then one block; the missing-value tag is a FLOAT in the reference too
 * (compact.tmplc:106: `float missingValueTag = *((FLOAT_TYPE *)missingTag)`) */
__global__ __launch_bounds__(256) void k_stats_f64(double *partials, const double *a, size_t n, int stride, int has_missing, float tag)
{
    double lo = INFINITY, hi = -INFINITY;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const double v = a[i * stride];
        if (has_missing && v == (double)tag) continue;
        lo = fmin(lo, v); hi = fmax(hi, v);
    }
    for (int off = 32; off > 0; off >>= 1) { lo = fmin(lo, __shfl_down(lo, off, 64)); hi = fmax(hi, __shfl_down(hi, off, 64)); }
    __shared__ double sh[2][4];
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = lo; sh[1][threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partials[2 * blockIdx.x] = fmin(fmin(sh[0][0], sh[0][1]), fmin(sh[0][2], sh[0][3]));
        partials[2 * blockIdx.x + 1] = fmax(fmax(sh[1][0], sh[1][1]), fmax(sh[1][2], sh[1][3]));
    }
}
__global__ __launch_bounds__(256) void k_cf_header_f64(packhip_cf_params *pp, unsigned *hdr, const double *partials, int nb, unsigned n, int bs, int style, int has_missing)
{
    double lo = INFINITY, hi = -INFINITY;
    for (int b = threadIdx.x; b < nb; b += 256) { lo = fmin(lo, partials[2 * b]); hi = fmax(hi, partials[2 * b + 1]); }
    for (int off = 32; off > 0; off >>= 1) { lo = fmin(lo, __shfl_down(lo, off, 64)); hi = fmax(hi, __shfl_down(hi, off, 64)); }
    __shared__ double sh[2][4];
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = lo; sh[1][threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x != 0) return;
    lo = fmin(fmin(sh[0][0], sh[0][1]), fmin(sh[0][2], sh[0][3])); hi = fmax(fmax(sh[1][0], sh[1][1]), fmax(sh[1][2], sh[1][3]));
    cf_header_write(pp, hdr, lo, hi, n, bs, style, has_missing);
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

/* fast path: 16-bit slots, offset 0, stride 1.  A block turns 4096 floats into 2048 words: 4 x (16 B in, 8 B out) per lane, every wave
 * instruction contiguous, all four loads in flight.  Both streams are `nt`: the field is read for the last time (its first read, by
 * k_stats, left it in the Infinity Cache: plain loads there) and the tokens are not read again by this kernel, so neither should displace
 * the part of the field that has yet to be read a second time.  tools/ubench_pack.hip on one cfg2 field, 8 fields in rotation:
 * k_stats + header + this pass 43.5 us, against 51.5 us with plain accesses in any shape (47.0 with nt stores only, 53.6 with nt loads only). */
typedef float cf_v4f __attribute__((ext_vector_type(4)));
typedef unsigned cf_v2u __attribute__((ext_vector_type(2)));
typedef unsigned cf_v4u __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_cf_pack16(cf_v2u *out, const cf_v4f *a, size_t n4, const packhip_cf_params *pp)
{
    const double minF = pp->minF, mul = pp->mulFactor;
    const size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
#define TK(v) ((unsigned)(long long)(((double)(v) - minF) * mul) & 0xFFFFu)
#define TK4(x_, o_) do { (o_).x = TK((x_).x) << 16 | TK((x_).y); (o_).y = TK((x_).z) << 16 | TK((x_).w); } while (0)
    if ((size_t)(blockIdx.x + 1) * 1024 <= n4) {
        cf_v4f x[4];
#pragma unroll
        for (int k = 0; k < 4; k++) x[k] = __builtin_nontemporal_load(&a[i + 256 * k]);
#pragma unroll
        for (int k = 0; k < 4; k++) { cf_v2u o; TK4(x[k], o); __builtin_nontemporal_store(o, &out[i + 256 * k]); }
    } else                                      /* the last, partial block */
        for (size_t j = i; j < n4; j += 256) { cf_v4f x = a[j]; cf_v2u o; TK4(x, o); out[j] = o; }
#undef TK4
#undef TK
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
        size_t n4 = n / 4;              /* groups of four points = two stream words; the rest (< 4 points) goes through the generic kernel */
        if (n4) hipLaunchKernelGGL(k_cf_pack16, dim3((unsigned)((n4 + 1023) / 1024)), dim3(256), 0, STREAM, (cf_v2u *)d_out, (const cf_v4f *)d_a, n4, d_pp);
        size_t done_words = n4 * 2;
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
__global__ __launch_bounds__(256) void k_cf_pack_f64(unsigned *out, const double *a, size_t n, int E, int offset, int stride,
                                                     const packhip_cf_params *pp, int has_missing, float tag, size_t w0, size_t nwords)
{
    size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= nwords) return;
    const packhip_cf_params p = *pp;
    emit_word(out, w0 + k, n, E, offset, [&](size_t t) {
        const double x = a[t * stride];
        if (has_missing == 1 && x == (double)tag) return p.missingToken;
        return (unsigned)(long long)((x - p.minF) * p.mulFactor);
    });
}
__global__ __launch_bounds__(256) void k_cf_unpack_f64(double *a, const unsigned *in, size_t n, int tokenSize, int offset, int stride,
                                                       double minF, double mulFactor, unsigned missingToken, int has_missing, float tag)
{
    size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    int sig = tokenSize > 32 ? 32 : tokenSize;
    unsigned tok = read_token(in, t, sig, offset);
    double v;
    if (has_missing == 1 && tok == missingToken) v = (double)tag;
    else if (tok == 0) v = minF;
    else v = (tok * mulFactor) * 1.0000000000001 + minF;
    a[t * stride] = v;
}
/* compact_double, pack side: extrema -> header + parameters -> stream.  d_partials: 2 * 1024 doubles of scratch */
extern "C" int packhip_cf_pack_f64(unsigned *d_out, unsigned *d_hdr, packhip_cf_params *d_pp, double *d_partials, const double *d_a, size_t n,
                                   int bs, int E, int offset, int stride, int style, int has_missing, float tag)
{
    if (n == 0) return 0;
    int nb = (int)((n + 4095) / 4096); if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(k_stats_f64, dim3(nb), dim3(256), 0, STREAM, d_partials, d_a, n, stride, has_missing, tag);
    hipLaunchKernelGGL(k_cf_header_f64, dim3(1), dim3(256), 0, STREAM, d_pp, d_hdr, d_partials, nb, (unsigned)n, bs, style, has_missing);
    size_t w0 = (size_t)offset / 32, w1 = ((size_t)offset + n * (size_t)E + 31) / 32;
    hipLaunchKernelGGL(k_cf_pack_f64, dim3((unsigned)((w1 - w0 + 255) / 256)), dim3(256), 0, STREAM, d_out, d_a, n, E, offset, stride, d_pp, has_missing, tag, w0, w1 - w0);
    return chk("k_cf_pack_f64");
}
extern "C" int packhip_cf_unpack_f64(double *d_a, const unsigned *d_in, size_t n, int tokenSize, int offset, int stride,
                                     double minF, double mulFactor, unsigned missingToken, int has_missing, float tag)
{
    if (!n) return 0;
    hipLaunchKernelGGL(k_cf_unpack_f64, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, STREAM, d_a, d_in, n, tokenSize, offset, stride, minF, mulFactor, missingToken, has_missing, tag);
    return chk("k_cf_unpack_f64");
}
/* compact_short / compact_char (compact_integer.c:592, :830): the unsigned short / byte elements i * stride <-> a dense uint32 array */
template <class T> __global__ __launch_bounds__(256) void k_widen(unsigned *dst, const T *src, size_t n, int stride)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = (unsigned)src[i * stride];
}
template <class T> __global__ __launch_bounds__(256) void k_narrow(T *dst, const unsigned *src, size_t n, int stride)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i * stride] = (T)src[i];
}
extern "C" int packhip_widen(unsigned *d_dst, const void *d_src, size_t n, int stride, int elem_bytes)
{
    if (!n) return 0;
    const unsigned nb = (unsigned)((n + 255) / 256);
    if (elem_bytes == 2) hipLaunchKernelGGL(k_widen<unsigned short>, dim3(nb), dim3(256), 0, STREAM, d_dst, (const unsigned short *)d_src, n, stride);
    else hipLaunchKernelGGL(k_widen<unsigned char>, dim3(nb), dim3(256), 0, STREAM, d_dst, (const unsigned char *)d_src, n, stride);
    return chk("k_widen");
}
extern "C" int packhip_narrow(void *d_dst, const unsigned *d_src, size_t n, int stride, int elem_bytes)
{
    if (!n) return 0;
    const unsigned nb = (unsigned)((n + 255) / 256);
    if (elem_bytes == 2) hipLaunchKernelGGL(k_narrow<unsigned short>, dim3(nb), dim3(256), 0, STREAM, (unsigned short *)d_dst, d_src, n, stride);
    else hipLaunchKernelGGL(k_narrow<unsigned char>, dim3(nb), dim3(256), 0, STREAM, (unsigned char *)d_dst, d_src, n, stride);
    return chk("k_narrow");
}

/* fast path of FLOAT_UNPACK: 16-bit slots, offset 0, stride 1, no missing values: a block turns 2048 words into 4096 floats, 4 x (8 B in, 16 B out)
 * per lane, both streams `nt` (each byte is touched once) */
__global__ __launch_bounds__(256) void k_cf_unpack16(cf_v4f *a, const cf_v2u *in, size_t n4, double minF, double mulFactor)
{
    const size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
    const float fmin = (float)minF;
#define UV(t) ((t) == 0 ? fmin : (float)(((t) * mulFactor) * 1.0000000000001 + minF))
#define UV4(w_, v_) do { unsigned t0 = (w_).x >> 16, t1 = (w_).x & 0xFFFFu, t2 = (w_).y >> 16, t3 = (w_).y & 0xFFFFu; \
                         (v_).x = UV(t0); (v_).y = UV(t1); (v_).z = UV(t2); (v_).w = UV(t3); } while (0)
    if ((size_t)(blockIdx.x + 1) * 1024 <= n4) {
        cf_v2u w[4];
#pragma unroll
        for (int k = 0; k < 4; k++) w[k] = __builtin_nontemporal_load(&in[i + 256 * k]);
#pragma unroll
        for (int k = 0; k < 4; k++) { cf_v4f v; UV4(w[k], v); __builtin_nontemporal_store(v, &a[i + 256 * k]); }
    } else
        for (size_t j = i; j < n4; j += 256) { cf_v2u w = in[j]; cf_v4f v; UV4(w, v); a[j] = v; }
#undef UV4
#undef UV
}
extern "C" int packhip_cf_unpack(float *d_a, const unsigned *d_in, size_t n, int tokenSize, int offset, int stride,
                                 double minF, double mulFactor, unsigned missingToken, int has_missing, float tag)
{
    if (!n) return 0;
    if (tokenSize == 16 && offset == 0 && stride == 1 && has_missing != 1 && ((uintptr_t)d_a % 16 == 0) && ((uintptr_t)d_in % 8 == 0) && n >= 4) {
        size_t n4 = n / 4, done = n4 * 4;
        hipLaunchKernelGGL(k_cf_unpack16, dim3((unsigned)((n4 + 1023) / 1024)), dim3(256), 0, STREAM, (cf_v4f *)d_a, (const cf_v2u *)d_in, n4, minF, mulFactor);
        if (done < n)       /* the rest: token `done` starts at word done / 2 */
            hipLaunchKernelGGL(k_cf_unpack, dim3((unsigned)((n - done + 255) / 256)), dim3(256), 0, STREAM, d_a + done, d_in + done / 2, n - done, tokenSize, 0, 1,
                               minF, mulFactor, missingToken, has_missing, tag);
        return chk("k_cf_unpack16");
    }
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

/* ------------------------------------------------------------------------------------------ */
/* armn_compress COMPRESS, PARALLELOGRAM method, ONE PASS (round 2)                              */
/* ------------------------------------------------------------------------------------------ */
/* The multi-kernel encoder above reads the 16-bit tokens three times (tile widths, block sums, emission), writes the
 * per-tile widths, scans them in two more launches and assembles the stream in a scratch buffer that a last kernel
 * copies to its place: six dependent launches, ~7.5 x the algorithmic bytes of a cfg5 field.  Here one launch does it:
 *   - a thread block owns a CHUNK of consecutive tiles of the stream order (a segment of one row of 3 x 3 tiles, or a few
 *     whole tile rows of a narrow field); chunks are handed out through a ticket counter, so chunk c only ever waits
 *     for chunks that are already running (no assumption about the dispatch order);
 *   - the 4 x (3 T + 1) tokens the chunk needs are staged in LDS once (coalesced word loads); a thread computes the nine
 *     Lorenzo differences of each of its tiles (kept in registers), the tile's width and bit count;
 *   - block scan -> tile bit offsets inside the chunk and the chunk's bit count, published as the chunk's AGGREGATE;
 *     the chunk's absolute position comes from a decoupled look-back over the status words of the earlier chunks
 *     (8-byte {state, value} granules, agent-scope relaxed stores / loads);
 *   - every thread writes its tiles into the chunk's LDS image of the stream (bit 0 = chunk start; only a tile's first
 *     and last words are shared with its neighbours: ds_or, the words in between are plain stores) WHILE wave 0's
 *     look-back result is still on its way; the image is then shifted to its absolute bit position and copied out with
 *     coalesced stores, straight into the caller's record;
 *   - the one word two consecutive chunks share is written by the LATER chunk: a chunk publishes the bits of its
 *     incomplete last word (`tail` granule) and ORs its predecessor's into its own first word;
 *   - the chunk that finishes last computes zlng (c_zfstlib.c:160-179).  The width of the per-tile width field (4 bits, 5
 *     when some |difference| > 65535, :701-711) is a whole-field property: the pass assumes 4, raises a flag otherwise,
 *     and the host re-runs such a field with 5 (zlng = -2 marks it; never seen on smooth fields).
 * Bit-exact against the restated reference encoder (tests/test_gpu_packers.py, tools/fuzz_armn.py). */
#define ENC_TPB 256
#define ENC_TPT 4                       /* tiles per thread: up to 1024 tiles per chunk */
#define ENC_MAX_PATCH_WORDS 6400        /* LDS words of the token patch / stream image (25 KB: 6 blocks per CU) */
struct enc1_args {
    unsigned *z; size_t z_stride, z_cap;                /* stream: per-field stride and capacity, in words */
    const unsigned *tok; size_t tok_stride;             /* token words (two 16-bit tokens per word) */
    int ni, nj, nbits, ntx, nty;
    int nseg, seg_tiles, rows_per_chunk;                /* nseg > 1: chunk = segment of one tile row; else rows_per_chunk tile rows */
    int nchunks, npre, row_words, container, nfields;
    unsigned long long *status, *tail;                  /* [nfields][nchunks] look-back / shared-word granules (zeroed) */
    unsigned *ctl;                                      /* [0] abort flag (zeroed) */
    size_t tok_words;                                   /* words of one field's token array ((ni nj + 1) / 2): bound of the 16-byte staging loads */
    int vec;                                            /* 1: token arrays are 16-byte aligned (base and stride): stage with 16-byte loads */
    unsigned long long *prof;                           /* development (EZHIP_ENC_DEBUG & 16): per-phase clock sums of thread 0 of every chunk block */
    int debug;                                          /* development knock-outs (EZHIP_ENC_DEBUG): 1 no emission, 2 no copy-out, 4 no staging, 8 no look-back wait */
    int *zlng;                                          /* [nfields] result: byte count, -1 not compressible, -2 re-run with container 5 */
};

/* word w (>= 1) of the stream prefix: 3-bit container, row 1, column 1 (c_zfstlib.c:712-721); bits at or beyond body_start are zero */
template <bool SWAPPED>
__device__ __forceinline__ unsigned enc1_prefix_word(const unsigned *tok, int ni, int nj, int nbits, int container, unsigned long long w)
{
    const unsigned short *t16 = (const unsigned short *)tok;
    const long long P0 = (long long)(w - 1) * 32;            /* first prefix bit of the word (prefix bit 0 = stream bit 32) */
    const long long ntok = (long long)ni + nj - 1;
    unsigned word = 0;
    if (P0 < 3) word |= (unsigned)container << 29;            /* only w == 1 */
    long long q = P0 <= 3 ? 0 : (P0 - 3) / nbits;
    for (; q < ntok; q++) {
        const long long tb = 3 + q * nbits;                   /* first bit of token q */
        if (tb >= P0 + 32) break;
        const size_t k = q < ni ? (size_t)q : (size_t)ni * (size_t)(q - ni + 1);
        const unsigned long long v = (unsigned long long)t16[k ^ (SWAPPED ? 1u : 0u)] & ((1ull << nbits) - 1);
        const long long sh = (P0 + 32) - (tb + nbits);
        word |= sh >= 0 ? (unsigned)(v << sh) : (unsigned)(v >> (-sh));
    }
    return word;
}

struct EncChunk { int f, c, ty0, nrow_t, tx0, nt_x, ntl, col_lo, prow_n; };
__device__ __forceinline__ EncChunk enc1_chunk(const enc1_args &a, unsigned g)
{
    EncChunk k;
    const unsigned cc_ = g / (unsigned)a.nfields;              /* chunks INTERLEAVED over the fields: the blocks in flight cover few chunks of each chain */
    k.f = (int)(g - cc_ * (unsigned)a.nfields); k.c = (int)cc_;
    if (a.nseg > 1) { k.ty0 = k.c / a.nseg; k.nrow_t = 1; k.tx0 = (k.c - k.ty0 * a.nseg) * a.seg_tiles; k.nt_x = min(a.seg_tiles, a.ntx - k.tx0); }
    else { k.ty0 = k.c * a.rows_per_chunk; k.nrow_t = min(a.rows_per_chunk, a.nty - k.ty0); k.tx0 = 0; k.nt_x = a.ntx; }
    k.ntl = k.nt_x * k.nrow_t;
    k.col_lo = 3 * k.tx0;
    k.prow_n = min(a.nj, 3 * (k.ty0 + k.nrow_t) + 1) - 3 * k.ty0;
    return k;
}

#define ENC_WK 1                        /* look-back: granules per lane and window */
#define ENC_NLD 7                       /* 16-byte pieces a lane stages up front: rows of up to 448 pieces (chunks of up to 1024 tiles) */
/* seven waves per SIMD (65 VGPRs, nothing spilled).  Eight (64 VGPRs, four of them spilled to scratch; the 19.3 KB of LDS of a cfg5 chunk allow it) was measured
 * slower (profiles/r03_experiments.txt); so was this body behind a __forceinline__ function taking the arguments by reference (71 VGPRs, 10 spilled: + 14 %) */
template <bool SWAPPED>
__global__ __launch_bounds__(ENC_TPB) __attribute__((amdgpu_waves_per_eu(7, 8))) void k_armn_enc1(enc1_args a)
{
    extern __shared__ unsigned lds[];                      /* token patch, then (aliased) the chunk's stream image */
    __shared__ unsigned s_wsum[ENC_TPT][ENC_TPB / 64], s_gt, s_abort, s_gt_all;
    __shared__ unsigned long long s_start;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ni = a.ni, nj = a.nj, nbits = a.nbits, container = a.container, rw = a.row_words;
    const unsigned long long body_start = 32ull + 3ull + (unsigned long long)(ni + nj - 1) * (unsigned long long)nbits;
    /* The chain order is the LAUNCH order of the thread blocks: block npre_all + g works on chunk g of the field-interleaved
     * chunk order and waits only for chunks of lower g.  What was measured instead (32 full-size cfg5 fields, 115 200 chunks):
     *   - a ticket counter (order = arrival order, independent of the dispatch order): one returning atomic per block on ONE
     *     word saturates at ~88 per us (MI355X_MICROARCH.md, `dequeue`): 2.16 ms;
     *   - persistent blocks with static ownership (block b takes chunks b, b + G, ...), tokens of the next chunk prefetched:
     *     the resident blocks march in lock-step generations, each as slow as its slowest member: 2.6 ms;
     *   - persistent blocks pulling from 64 work queues: the queues drift apart and chunks wait for predecessors that
     *     nobody has started: 2.75 ms;
     *   - this form: 1.3 ms.  Thread blocks are dispatched in index order on every XCD, so the lowest unfinished chunk is always
     *     resident or next in line.  Should that ever not hold, every wait below is BOUNDED: a block that gives up raises the
     *     abort flag, all others leave, and the host falls back to the multi-kernel encoder. */
    const unsigned npre_all = (unsigned)a.npre * (unsigned)a.nfields;
    if (blockIdx.x < npre_all) {
        /* prefix block: the stream words that lie wholly inside the prefix (c_zfstlib.c:712-721), one owner thread per word; word 0 = _fstzip header */
        const unsigned fz = blockIdx.x / (unsigned)a.npre, r = blockIdx.x - fz * (unsigned)a.npre;
        unsigned *zf = a.z + (size_t)fz * a.z_stride;
        const unsigned long long wend = body_start >> 5;      /* first word chunk 0 takes part in */
        for (unsigned long long w = (unsigned long long)r * (ENC_TPB * 4) + tid, k = 0; k < 4; k++, w += ENC_TPB) {
            if (w == 0) { if (a.z_cap > 0) zf[0] = 4u | 1u << 4 | 3u << 7 | ((unsigned)nbits & 31u) << 10 | 1u << 15; }
            else if (w < wend && w < a.z_cap) zf[w] = enc1_prefix_word<SWAPPED>(a.tok + (size_t)fz * a.tok_stride, ni, nj, nbits, container, w);
        }
        return;
    }
    const EncChunk ck = enc1_chunk(a, blockIdx.x - npre_all);
    const int f = ck.f, c = ck.c, ty0 = ck.ty0, nrow_t = ck.nrow_t, tx0 = ck.tx0, nt_x = ck.nt_x, ntl = ck.ntl, col_lo = ck.col_lo;
    unsigned *z = a.z + (size_t)f * a.z_stride;
    const unsigned *tok = a.tok + (size_t)f * a.tok_stride;
    unsigned long long tstamp[7]; int nst = 0;
#define STAMP() do { if (a.prof && tid == 0) tstamp[nst] = __builtin_readcyclecounter(); nst++; } while (0)
    STAMP();
    if (tid == 0) { s_gt = 0; s_abort = 0; }
    /* ---- stage the tokens: LDS row pr holds the words [wbase(pr), wbase(pr) + rw) of the field, wbase = first word of the row's
     * tokens rounded down to 4 words; every thread issues ALL its 16-byte loads before the first wait (a dword per lane and a
     * wait per loop iteration kept 8 KB in flight per CU: 1 TB/s); rotated to token order when the first token of a word sits in
     * the high half (c_zfstlib.c:119-126) ---- */
    {
        const int rv = rw >> 2, nitems = ck.prow_n * rv;
        uint4 *lds4 = (uint4 *)lds;
        const int wv_u = __builtin_amdgcn_readfirstlane(wave);
        const size_t wb_u = (size_t)((((unsigned)(3 * ty0 + wv_u) * (unsigned)ni + (unsigned)col_lo) >> 1) & ~3u);      /* first staged word of patch row `wave` */
        const size_t wb_last = (size_t)((((unsigned)(3 * ty0 + ck.prow_n - 1) * (unsigned)ni + (unsigned)col_lo) >> 1) & ~3u);   /* the chunk's last row: the one nearest the array's end */
        if (nrow_t == 1 && rv <= 64 * ENC_NLD && wb_last + 4u * (unsigned)rv <= a.tok_words) {        /* block-uniform */
            /* the common shape (one tile row per chunk): wave w stages patch row w; the row base is wave-uniform, a lane's pieces are 1 KB apart.  By LDS-DMA
             * (round 5): no registers, no ds_write -- the seven ds_write_b128 per wave were 316 LDS-array cycles per wave (44 each), a quarter of what the LDS
             * array did in this kernel.  The words arrive as they lie in memory; which half of a word holds the earlier token (SWAPPED) is the readers' business */
            if (wv_u < ck.prow_n) {
                const uint4 *src = (const uint4 *)(tok + wb_u) + lane;
                const unsigned dst0 = (unsigned)(size_t)(__attribute__((address_space(3))) const void *)(lds4 + wv_u * rv);
#pragma unroll
                for (int u = 0; u < ENC_NLD; u++)
                    if (lane + 64 * u < rv)
                        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(src + 64 * u), "s"(__builtin_amdgcn_readfirstlane((int)(dst0 + 1024u * (unsigned)u))) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            /* chunks of several tile rows (narrow fields) and the last rows of the token array: piece by piece, bounds checked */
            for (int idx = tid; idx < nitems; idx += ENC_TPB) {
                const int pr = idx / rv, v = idx - pr * rv;
                const size_t gw = (size_t)((((unsigned)(3 * ty0 + pr) * (unsigned)ni + (unsigned)col_lo) >> 1) & ~3u) + 4u * (unsigned)v;
                uint4 x = make_uint4(0u, 0u, 0u, 0u);
                if (gw + 3 < a.tok_words) x = *(const uint4 *)(tok + gw);
                else { if (gw < a.tok_words) x.x = tok[gw]; if (gw + 1 < a.tok_words) x.y = tok[gw + 1]; if (gw + 2 < a.tok_words) x.z = tok[gw + 2]; }
                lds4[idx] = x;
            }
        }
    }
    __syncthreads();
    STAMP();
    if (EZH_DBG(a.debug) & 32) { if (tid == 0 && (c == a.nchunks - 1 || lds[0] == 0x12345u)) a.zlng[f] = 1000; return; }      /* development: staging only */
    /* ---- per tile: differences (kept in registers), width, bit count ------------------------------------------- */
    unsigned long long dpk[ENC_TPT][3];                     /* the tokens of a tile row packed at the tile's width, first token on top (36 plain registers for the differences cost two waves per SIMD) */
    unsigned bits[ENC_TPT], meta[ENC_TPT];                  /* meta: need | tm << 8 | tn << 12 */
    bool gt = false;
    /* halfword offset of patch row n of the chunk's FIRST tile row (block-uniform): row base + token offset inside the row's first 4 staged words */
    unsigned rowoff[4];
#pragma unroll
    for (int n = 0; n < 4; n++) rowoff[n] = (unsigned)n * (unsigned)rw * 2u + (((unsigned)(3 * ty0 + n) * (unsigned)ni + (unsigned)col_lo) & 7u);
#pragma unroll
    for (int q = 0; q < ENC_TPT; q++) {
        const int tl = tid + ENC_TPB * q;
        bits[q] = 0; meta[q] = 0;
        dpk[q][0] = dpk[q][1] = dpk[q][2] = 0ull;
        if (tl < ntl) {
            int trow = 0, tcx = tl;
            if (nrow_t > 1) { trow = tl / nt_x; tcx = tl - trow * nt_x; }
            const int tm = min(3, ni - (1 + 3 * (tx0 + tcx))), tn = min(3, nj - (1 + 3 * (ty0 + trow)));
            /* the 4 x 4 token patch, unclamped: a clipped tile reads tokens that are not its own (inside the LDS allocation) and
             * masks them below; four row bases, the columns are immediate offsets.  Horizontal differences first: 12 + 9 subtractions */
            int h[4][3];
#pragma unroll
            for (int n = 0; n < 4; n++) {
                /* the row's four tokens as THREE aligned dwords and two funnel shifts (round 5).  As four ds_read_u16 they were 19 LDS-array cycles each (lanes 6 bytes
                 * apart, sub-dword accesses): 908 of the kernel's 1319 LDS cycles per wave, and the LDS array was 75 % busy -- the encoder's real bound
                 * (SQ_LDS_IDX_ACTIVE, profiles/r05_experiments.txt); a dword read is 2 cycles.  (The third dword may lie one word past the row: inside the allocation.) */
                const unsigned hidx = 3u * (unsigned)tcx + (nrow_t > 1 ? (unsigned)(3 * trow + n) * (unsigned)rw * 2u + (((unsigned)(3 * (ty0 + trow) + n) * (unsigned)ni + (unsigned)col_lo) & 7u) : rowoff[n]);
                const unsigned *pw = lds + (hidx >> 1);
                const unsigned d0 = pw[0], d1 = pw[1], d2 = pw[2];
                /* u0 | u1 << 16 and u2 | u3 << 16 by byte permutes of (d1 : d0) and (d2 : d1): the selector knows the row's parity and which half of a word holds the
                 * earlier token (c_zfstlib.c:119-126: the high half when SWAPPED) */
                const unsigned sel = (hidx & 1u) ? (SWAPPED ? 0x07060100u : 0x05040302u) : (SWAPPED ? 0x01000302u : 0x03020100u);
                const unsigned ta = __builtin_amdgcn_perm(d1, d0, sel), tb = __builtin_amdgcn_perm(d2, d1, sel);
                const int u0 = (int)(ta & 0xFFFFu), u1 = (int)(ta >> 16), u2 = (int)(tb & 0xFFFFu), u3 = (int)(tb >> 16);
                h[n][0] = u1 - u0; h[n][1] = u2 - u1; h[n][2] = u3 - u2;
            }
            int d[3][3];
#pragma unroll
            for (int n = 0; n < 3; n++)
#pragma unroll
                for (int m = 0; m < 3; m++) d[n][m] = h[n + 1][m] - h[n][m];        /* u11 - (u01 + u10 - u00), c_zfstlib.c:691-696 */
            if (tm < 3 || tn < 3) {                          /* the last tile column / row of the field */
#pragma unroll
                for (int n = 0; n < 3; n++)
#pragma unroll
                    for (int m = 0; m < 3; m++) if (n >= tn || m >= tm) d[n][m] = 0;
            }
            int hi = max(max(d[0][0], d[0][1]), d[0][2]), lo = min(min(d[0][0], d[0][1]), d[0][2]);
#pragma unroll
            for (int n = 1; n < 3; n++) { hi = max(max(hi, d[n][0]), max(d[n][1], d[n][2])); lo = min(min(lo, d[n][0]), min(d[n][1], d[n][2])); }
            const int mx = max(hi, -lo);
            if (mx > 65535) gt = true;
            unsigned need = (unsigned)bitlen((unsigned)mx);
            if (need == 16) need = 15;
            {   /* the tile's rows as the emission wants them: a row's tokens side by side at the tile's own width, first token on top (round 5; 18-bit fields that the
                 * emission unpacked and packed again before: 24 of its 93 VALU instructions per tile were that) */
                const int width = need == 0 ? 0 : (need == 15 ? 17 : (int)need + 1);
                const unsigned mask = (1u << width) - 1u;
#pragma unroll
                for (int n = 0; n < 3; n++) {
                    const unsigned a0 = (unsigned)d[n][0] & mask, a1 = (unsigned)d[n][1] & mask, a2 = (unsigned)d[n][2] & mask;
                    /* (three tokens, then the missing ones of a clipped tile shifted out again: no branch on tm) */
                    const unsigned long long v = ((unsigned long long)a0 << (2 * width)) | ((unsigned long long)a1 << width) | (unsigned long long)a2;
                    dpk[q][n] = v >> ((3 - tm) * width);
                }
            }
            bits[q] = tile_bits(1, need, tm * tn, container, nbits);
            meta[q] = need | (unsigned)tm << 8 | (unsigned)tn << 12;
        }
    }
    if (gt) s_gt = 1;                                       /* benign race: every writer stores 1 */
    /* ---- block scan, layer by layer (layer q = tiles [256 q, 256 q + 255]) ------------------------------------ */
    /* ONE wave scan for the four layers: their bit counts (<= 171 per tile, <= 10 944 per wave) ride in the 16-bit fields of a 64-bit value */
    unsigned incl[ENC_TPT];
    {
        static_assert(ENC_TPT == 4, "the packed scan holds four layers");
        unsigned long long v = (unsigned long long)bits[0] | (unsigned long long)bits[1] << 16 | (unsigned long long)bits[2] << 32 | (unsigned long long)bits[3] << 48;
        for (int off = 1; off < 64; off <<= 1) { const unsigned long long o = __shfl_up(v, off, 64); if (lane >= off) v += o; }
#pragma unroll
        for (int q = 0; q < ENC_TPT; q++) { incl[q] = (unsigned)(v >> (16 * q)) & 0xFFFFu; if (lane == 63) s_wsum[q][wave] = incl[q]; }
    }
    __syncthreads();                                        /* also: every thread is done reading the token patch */
    unsigned excl[ENC_TPT], agg = 0;
#pragma unroll
    for (int q = 0; q < ENC_TPT; q++) {
        unsigned before = agg;
        for (int w = 0; w < ENC_TPB / 64; w++) { if (w < wave) before += s_wsum[q][w]; agg += s_wsum[q][w]; }
        excl[q] = before + incl[q] - bits[q];
    }
    unsigned long long *status = a.status + (size_t)f * a.nchunks;
    if (tid == 0) st_granule(&status[c], (c == 0 ? ST_PFX : ST_AGG) | (s_gt ? ST_GT : 0ull) | (unsigned long long)agg);      /* chunk 0: its aggregate IS its inclusive prefix */
    STAMP();
    /* ---- stream image of the chunk in LDS (bit 0 = first bit of the chunk), over the token patch ---------------- */
    unsigned *img = lds;
    const unsigned img_words = ((agg + 31) >> 5) + 4;
    for (unsigned w = tid; w < img_words; w += ENC_TPB) img[w] = 0;
    __syncthreads();
    /* a thread writes its tiles MSB first through a 64-bit accumulator; only a tile's first and last words are shared with
     * the neighbouring tiles (ds_or), the words in between are plain stores.  (Three unconditional ds_or per tile row, branch
     * free, were slower: the same-word and same-bank conflicts of adjacent tiles cost more than the branches.) */
#pragma unroll
    for (int q = 0; q < ENC_TPT; q++) {
        if (tid + ENC_TPB * q >= ntl || (EZH_DBG(a.debug) & 1)) continue;
        const unsigned need = meta[q] & 0xFF; const int tm = (int)(meta[q] >> 8) & 0xF, tn = (int)(meta[q] >> 12) & 0xF;
        unsigned wi = excl[q] >> 5; int fill = (int)(excl[q] & 31);
        unsigned long long acc = 0; bool first = true;
        /* the (up to) three tokens of a tile row are ONE value of at most 3 x 18 = 54 bits (the first row carries the tile's width field
         * in front: <= 59); it goes through a 96-bit window (acc + the row): at most two words are completed per row */
        const int width = need == 0 ? 0 : (need == 15 ? 17 : (int)need + 1);
        const int rowlen = tm * width;
        auto flush = [&]() {
            const unsigned word = (unsigned)(acc >> 32);
            if (first) { atomicOr(&img[wi], word); first = false; } else img[wi] = word;
            wi++; acc <<= 32; fill -= 32;
        };
#pragma unroll
        for (int n = 0; n < 3; n++) {
            if (n >= tn || (n > 0 && need == 0)) break;
            unsigned long long v = dpk[q][n];
            int L = rowlen;
            if (n == 0) { v |= (unsigned long long)need << rowlen; L += container; }
            /* fill < 32 pending bits in acc's top; append L <= 59 bits: first the part that fits the 64-bit accumulator */
            const int room = 64 - fill;
            if (L <= room) { acc |= L == 64 ? v : v << (room - L); fill += L; }
            else { acc |= v >> (L - room); fill = 64; flush(); flush(); acc = v << (64 - (L - room)); fill = L - room; continue; }
            if (fill >= 32) flush();
            if (fill >= 32) flush();
        }
        if (fill > 0) atomicOr(&img[wi], (unsigned)(acc >> 32));
    }
    __syncthreads();                                        /* the image is complete */
    STAMP();
    /* ---- the last 32 bits of the stream up to the END of this chunk (position independent): the next chunk takes the bits of
     * the word it shares with this one from here, long before this chunk knows where it starts ------------------------- */
    unsigned long long *tail = a.tail + (size_t)f * a.nchunks;
    auto prefix_tail = [&]() -> unsigned {
        const unsigned long long P = body_start - 32; const unsigned s_ = (unsigned)(P & 31);
        const unsigned wa = enc1_prefix_word<SWAPPED>(tok, ni, nj, nbits, container, P >> 5);
        return s_ ? (wa << s_) | (enc1_prefix_word<SWAPPED>(tok, ni, nj, nbits, container, (P >> 5) + 1) >> (32 - s_)) : wa;
    };
    auto wait_tail = [&](int idx, unsigned long long t) -> unsigned {       /* cumulative tail of chunk idx (t: an earlier load of it) */
        int spins = 0;
        while ((t >> 62) == 0) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > (1 << 19) || ((spins & 63) == 0 && __hip_atomic_load(&a.ctl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                __hip_atomic_store(&a.ctl[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); s_abort = 1; break;
            }
            t = ld_granule(&tail[idx]);
        }
        return (unsigned)t;
    };
    /* ---- wave 0: publish the tail, then look back for the exclusive prefix of the chunk aggregates ------------------ */
    if (wave == 0) {
        unsigned long long tprev = (lane == 0 && c > 0) ? ld_granule(&tail[c - 1]) : 0ull;      /* needed for the shared word below: ask early */
        if (lane == 0) {
            unsigned tl32;
            if (agg >= 32) { const unsigned e = agg & 31, w = agg >> 5; tl32 = e ? (img[w - 1] << e) | (img[w] >> (32 - e)) : img[w - 1]; }
            else {                                               /* a chunk shorter than a word (tiny fields): chained */
                const unsigned tp = c > 0 ? wait_tail(c - 1, tprev) : prefix_tail();
                tl32 = (tp << agg) | (img[0] >> (32 - agg));
                tprev = ST_PFX | tp;
            }
            st_granule(&tail[c], ST_PFX | (unsigned long long)tl32);
        }
        unsigned long long excl_chunks = 0, gt_before = 0;
        bool gave_up = false;
        if (c > 0 && !(EZH_DBG(a.debug) & 8)) {
            /* windows of 256 predecessors (4 per lane, nearest first) */
            int basei = c - 1, spins = 0;
            for (;;) {
                if (a.prof && lane == 0) atomicAdd(&a.prof[6], 1ull);            /* development: look-back windows read */
                unsigned long long st[ENC_WK];
#pragma unroll
                for (int k = 0; k < ENC_WK; k++) { const int idx = basei - (lane * ENC_WK + k); st[k] = idx >= 0 ? ld_granule(&status[idx]) : ST_PFX; }
                int kp = ENC_WK;
#pragma unroll
                for (int k = ENC_WK - 1; k >= 0; k--) if ((st[k] >> 62) == 2) kp = k;
                bool lane_ok = true;
#pragma unroll
                for (int k = 0; k < ENC_WK; k++) if (k <= kp && (st[k] >> 62) == 0) lane_ok = false;
                const unsigned long long pm = __ballot(kp < ENC_WK), okm = __ballot(lane_ok);
                const int firstp = pm ? __builtin_ctzll(pm) : 63;
                const unsigned long long needm = firstp == 63 ? ~0ull : ((2ull << firstp) - 1);
                if ((okm & needm) != needm) {
                    /* not there yet: ONE lane polls the ONE nearest granule that is missing, with a pause, then the window is read again */
                    const int bad_lane = __builtin_ctzll(~okm & needm);
                    int kbad = 0;
#pragma unroll
                    for (int k = ENC_WK - 1; k >= 0; k--) if ((st[k] >> 62) == 0) kbad = k;
                    const int bad_idx = __shfl(basei - (lane * ENC_WK + kbad), bad_lane, 64);
                    if (lane == 0) {
                        while ((ld_granule(&status[bad_idx]) >> 62) == 0) {
                            if (a.prof) atomicAdd(&a.prof[7], 1ull);               /* development: polls of a missing granule */
                            __builtin_amdgcn_s_sleep(8);
                            if (++spins > (1 << 19) || ((spins & 63) == 0 && __hip_atomic_load(&a.ctl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) { gave_up = true; break; }
                        }
                    }
                    gave_up = __shfl((int)gave_up, 0, 64) != 0;
                    if (gave_up) break;
                    continue;
                }
                unsigned long long v = 0; bool gg = false;
                if (!pm || lane <= firstp) {
#pragma unroll
                    for (int k = 0; k < ENC_WK; k++) if (k <= kp) { v += ST_VAL(st[k]); gg = gg || (st[k] & ST_GT) != 0; }
                }
                for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
                excl_chunks += v;
                gt_before |= __ballot(gg);
                if (pm) break;
                basei -= 64 * ENC_WK;
            }
            if (lane == 0 && !gave_up) st_granule(&status[c], ST_PFX | ((gt_before || s_gt) ? ST_GT : 0ull) | (excl_chunks + agg));
        }
        if (lane == 0) {
            const unsigned long long S = body_start + excl_chunks;
            s_start = S;
            s_gt_all = (gt_before || s_gt) ? 1u : 0u;
            if (gave_up) { s_abort = 1; __hip_atomic_store(&a.ctl[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            else {
                /* the stream word this chunk shares with the earlier ones: the bits in front of S come from their tail (the prefix's for chunk 0) */
                const unsigned sh = (unsigned)(S & 31);
                if (sh && (((sh + agg) >> 5) >= 1 || c == a.nchunks - 1)) {         /* the word is completed here (else a later chunk stores it) */
                    const unsigned tp = c > 0 ? wait_tail(c - 1, tprev) : prefix_tail();
                    const unsigned v0 = (img[0] >> sh) | ((tp & ((1u << sh) - 1u)) << (32 - sh));
                    if ((S >> 5) < a.z_cap) z[S >> 5] = v0;
                }
            }
        }
    }
    __syncthreads();
    STAMP();
    if (s_abort) return;
    /* ---- copy out: image shifted to its absolute bit position -------------------------------------------------- */
    const unsigned long long S = s_start;
    const unsigned sh = (unsigned)(S & 31);
    const unsigned long long gw0 = S >> 5;
    const unsigned nwout = (sh + agg + 31) >> 5;             /* stream words the chunk touches */
    const bool last_chunk = c == a.nchunks - 1;
    const bool tail_open = ((sh + agg) & 31) != 0 && !last_chunk;      /* the last word is completed (and stored) by a later chunk */
    const unsigned nstore = nwout - (tail_open ? 1u : 0u);   /* words [0, nstore) are completed by this chunk */
    if (!(EZH_DBG(a.debug) & 2))
        for (unsigned k = (sh ? 1u : 0u) + tid; k < nstore; k += ENC_TPB) {        /* word 0 with bits of earlier chunks: stored above */
            const unsigned lo = img[k];
            const unsigned v = sh == 0 ? lo : ((k ? img[k - 1] : 0u) << (32 - sh)) | (lo >> sh);
            if (gw0 + k < a.z_cap) z[gw0 + k] = v;
        }
    if (tid == 0 && last_chunk) {
        if (gw0 + nwout < a.z_cap) z[gw0 + nwout] = 0u;       /* the word after the last one: the byte zlng points at (undefined in the reference) */
        const unsigned long long bits_total = S + agg - 32;  /* stream bits from the start of word 1, terminator excluded (c_zfstlib.c:160-179) */
        const long long zl = 1 + 4 * (1 + (long long)((bits_total + 31) / 32));
        const long long lng_origin = 1 + 2 * (long long)ni * nj;
        a.zlng[f] = (s_gt_all && container == 4 && nbits >= 15) ? -2 : (zl >= lng_origin ? -1 : (int)zl);
    }
    STAMP();
    if (a.prof && tid == 0) {       /* [0] chunks, [1..5] cycles of: staging, tiles + scan, emission, look-back wait, copy-out */
        atomicAdd(&a.prof[0], 1ull);
        for (int k = 1; k < 6; k++) atomicAdd(&a.prof[k], tstamp[k] - tstamp[k - 1]);
    }
#undef STAMP
}

static unsigned long long *g_enc1_prof = nullptr;
/* development: phase clock sums of the last profiled launch (EZHIP_ENC_DEBUG & 16) -> out[8] */
extern "C" int packhip_armn_enc1_profile(unsigned long long *out8)
{
    if (!g_enc1_prof) return -1;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    return hipMemcpy(out8, g_enc1_prof, 64, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
/* geometry of the one-pass encoder for a field shape; returns 0 when the shape is outside its limits (-> multi-kernel path) */
static int enc1_geometry(enc1_args *a, int ni, int nj, int nbits)
{
    if (ni < 16 || nj < 16 || nbits <= 4 || nbits > 16) return 0;
    if ((long long)ni * nj >= (1ll << 31)) return 0;
    a->ni = ni; a->nj = nj; a->nbits = nbits;
    a->ntx = (ni - 1 + 2) / 3; a->nty = (nj - 1 + 2) / 3;
    const int maxt = ENC_TPB * ENC_TPT;
    if (a->ntx > maxt) {
        a->nseg = (a->ntx + maxt - 1) / maxt; a->seg_tiles = (a->ntx + a->nseg - 1) / a->nseg; a->rows_per_chunk = 1;
        a->nchunks = a->nseg * a->nty;
        a->row_words = (((3 * a->seg_tiles + 1 + 2) / 2 + 1 + 3) + 3) & ~3;      /* + up to 3 words in front (16-byte aligned start), multiple of 4 */
    } else {
        a->nseg = 1; a->seg_tiles = a->ntx;
        a->row_words = (((ni + 2) / 2 + 1 + 3) + 3) & ~3;
        int R = maxt / a->ntx;
        const int rmax = (ENC_MAX_PATCH_WORDS / a->row_words - 1) / 3;
        if (R > rmax) R = rmax;
        if (R < 1) return 0;
        a->rows_per_chunk = R;
        a->nchunks = (a->nty + R - 1) / R;
    }
    const unsigned long long body_start = 32ull + 3ull + (unsigned long long)(ni + nj - 1) * (unsigned long long)nbits;
    a->npre = (int)(((body_start >> 5) + ENC_TPB * 4 - 1) / (ENC_TPB * 4));
    if (a->npre < 1) a->npre = 1;
    return 1;
}
static size_t enc1_patch_words(const enc1_args *a)
{
    const size_t prow = (size_t)(a->nseg > 1 ? 4 : 3 * a->rows_per_chunk + 1);
    return (prow * (size_t)a->row_words + 3) & ~(size_t)3;
}
static size_t enc1_lds_bytes(const enc1_args *a)
{
    /* the larger of the token patch and the image (tiles x (5 + 9 x 18) bits worst case): the image overwrites the patch */
    size_t tiles = (size_t)(a->nseg > 1 ? a->seg_tiles : a->ntx * a->rows_per_chunk);
    size_t img = (tiles * 167 + 31) / 32 + 8;
    size_t patch = enc1_patch_words(a);
    return 4 * (patch > img ? patch : img) + 16;      /* (+ 16: the tile phase reads a row's tokens as three aligned dwords, the third may lie behind the last row) */
}
/* control / granule storage of one launch over nfields fields (device scratch, zeroed by the launcher) */
extern "C" size_t packhip_armn_enc1_work_bytes(int ni, int nj, int nbits, int nfields)
{
    enc1_args a;
    if (!enc1_geometry(&a, ni, nj, nbits)) return 0;
    return 16 * (size_t)a.nchunks * nfields + 16 + 64 + 64;
}
/* One-pass PARALLELOGRAM encode of nfields token fields (asynchronous).  d_zlng[f] receives the byte count, -1 (not
 * compressible: the stream may be truncated at z_cap words) or -2 (re-run the field with container = 5).
 * Returns 1 when the shape is outside the one-pass limits (nothing launched), -1 on error. */
extern "C" int packhip_armn_encode1(unsigned *d_z, size_t z_stride_words, size_t z_cap_words, const unsigned *d_tok, size_t tok_stride_words,
                                    int ni, int nj, int nbits, int nfields, int swapped, int container, void *d_work, int *d_zlng)
{
    enc1_args a;
    memset(&a, 0, sizeof(a));
    if (!enc1_geometry(&a, ni, nj, nbits)) return 1;
    const size_t lds = enc1_lds_bytes(&a);
    if (lds > 64 * 1024) return 1;
    a.z = d_z; a.z_stride = z_stride_words; a.z_cap = z_cap_words; a.tok = d_tok; a.tok_stride = tok_stride_words;
    a.container = container; a.nfields = nfields; a.zlng = d_zlng;
    a.debug = EZH_DEVINT("EZHIP_ENC_DEBUG");
    a.tok_words = ((size_t)ni * nj + 1) / 2;
    if (((uintptr_t)d_tok % 16 != 0) || (tok_stride_words % 4 != 0 && nfields > 1)) return 1;      /* the LDS-DMA staging moves aligned 16-byte pieces */
    a.vec = 1;
    char *w = (char *)d_work;
    a.status = (unsigned long long *)w; w += 8 * (size_t)a.nchunks * nfields;
    a.tail = (unsigned long long *)w; w += 8 * (size_t)a.nchunks * nfields;
    a.ctl = (unsigned *)w; w += 16;
    a.prof = (EZH_DBG(a.debug) & 16) ? (unsigned long long *)w : nullptr; w += 64;
    g_enc1_prof = (EZH_DBG(a.debug) & 16) ? (unsigned long long *)(w - 64) : nullptr;
    hipStream_t st = STREAM;
    if (hipMemsetAsync(d_work, 0, (size_t)(w - (char *)d_work), st) != hipSuccess) return -1;
    const size_t nblocks = (size_t)(a.npre + a.nchunks) * (size_t)nfields;
    if (nblocks >= (1ull << 31)) return 1;
    if (swapped) hipLaunchKernelGGL(k_armn_enc1<true>, dim3((unsigned)nblocks), dim3(ENC_TPB), lds, st, a);
    else hipLaunchKernelGGL(k_armn_enc1<false>, dim3((unsigned)nblocks), dim3(ENC_TPB), lds, st, a);
    return chk("k_armn_enc1") ? -1 : 0;
}

/* the prefix words of the streams k_sepx_enc (ez_kernels.hip: interpolation and encoder in one launch) left without them: word 0 (the _fstzip header),
 * the 3-bit container, row 0 and column 0 verbatim (c_zfstlib.c:712-721) from the side array of ni + nj - 1 tokens per field, and the word the prefix
 * shares with the first chunk (`head` = that chunk's first image word) */
__global__ __launch_bounds__(256) void k_sepenc_prefix(unsigned *z, size_t z_stride, size_t z_cap, const unsigned short *ptok, size_t ptok_stride,
                                                       const unsigned *head, const int *zlng, int ni, int nj, int nbits, int container)
{
    const int f = blockIdx.y;
    if (zlng[f] <= 0) return;                                 /* the host redoes such a field with the two-kernel path */
    unsigned *zf = z + (size_t)f * z_stride;
    const unsigned short *pt = ptok + (size_t)f * ptok_stride;
    const unsigned long long body_start = 32ull + 3ull + (unsigned long long)(ni + nj - 1) * (unsigned long long)nbits;
    const unsigned long long wend = body_start >> 5;
    const unsigned sh = (unsigned)(body_start & 31);
    const long long ntok = (long long)ni + nj - 1;
    for (unsigned long long w = (unsigned long long)blockIdx.x * 256 + threadIdx.x; w <= wend; w += (unsigned long long)gridDim.x * 256) {
        if (w >= z_cap) continue;
        if (w == 0) { zf[0] = 4u | 1u << 4 | 3u << 7 | ((unsigned)nbits & 31u) << 10 | 1u << 15; continue; }
        if (w == wend && !sh) continue;                       /* the body starts on a word: chunk 0 stored it */
        const long long P0 = (long long)(w - 1) * 32;        /* first prefix bit of the word (prefix bit 0 = stream bit 32) */
        unsigned word = 0;
        if (P0 < 3) word |= (unsigned)container << 29;
        long long q = P0 <= 3 ? 0 : (P0 - 3) / nbits;
        for (; q < ntok; q++) {
            const long long tb = 3 + q * nbits;
            if (tb >= P0 + 32) break;
            const unsigned long long v = (unsigned long long)pt[q] & ((1ull << nbits) - 1);
            const long long s_ = (P0 + 32) - (tb + nbits);
            word |= s_ >= 0 ? (unsigned)(v << s_) : (unsigned)(v >> (-s_));
        }
        if (w == wend) word |= head[f] >> sh;
        zf[w] = word;
    }
}
extern "C" int packhip_sepenc_prefix(unsigned *d_z, size_t z_stride_words, size_t z_cap_words, const unsigned short *d_ptok, size_t ptok_stride,
                                     const unsigned *d_head, const int *d_zlng, int ni, int nj, int nbits, int container, int nfields)
{
    const unsigned long long wend = (32ull + 3ull + (unsigned long long)(ni + nj - 1) * (unsigned long long)nbits) >> 5;
    hipLaunchKernelGGL(k_sepenc_prefix, dim3((unsigned)((wend + 256) / 256), (unsigned)nfields), dim3(256), 0, STREAM,
                       d_z, z_stride_words, z_cap_words, d_ptok, ptok_stride, d_head, d_zlng, ni, nj, nbits, container);
    return chk("k_sepenc_prefix");
}

/* conditional copies behind an asynchronous encode, all fields in one launch (blockIdx.y = field):
 * mode 0: zlng[f] > 0  -> the stream's ceil(zlng / 4) words (commit of a stream assembled in scratch);
 * mode 1: zlng[f] == -1 -> nwords token words (a field that is not compressible keeps the plain 16-bit pack) */
__global__ __launch_bounds__(256) void k_cond_copy(unsigned *dst, size_t dst_stride, const unsigned *src, size_t src_stride, size_t nwords, const int *zlng, int mode)
{
    const int f = blockIdx.y, zl = zlng[f];
    size_t n;
    if (mode == 0) { if (zl <= 0) return; n = (size_t)(zl + 3) / 4; } else { if (zl != -1) return; n = nwords; }
    dst += (size_t)f * dst_stride; src += (size_t)f * src_stride;
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (size_t)gridDim.x * 256) dst[k] = src[k];
}
extern "C" int packhip_cond_copy(unsigned *d_dst, size_t dst_stride, const unsigned *d_src, size_t src_stride, size_t nwords, const int *d_zlng, int nfields, int mode)
{
    hipLaunchKernelGGL(k_cond_copy, dim3(1024, nfields), dim3(256), 0, STREAM, d_dst, dst_stride, d_src, src_stride, nwords, d_zlng, mode);
    return chk("k_cond_copy");
}
/* ---- FST framing of a device-resident cfg5 record (fst_record_host.c: ezhip_fst_frame_record_dev) ------------------------------------ */
/* datyp 129 (fstd98.c:1164-1190): data[0] = nw, data[1..4] = the four compact_float header words, data[5..] = the armn_compress stream, zero up to nw.
 * In place (data + 1 == record) only word 0 and the padding are written. */
__global__ __launch_bounds__(256) void k_fst_frame129(unsigned *data, const unsigned *rec, unsigned nw, unsigned ncopy)
{
    const unsigned k = blockIdx.x * 256 + threadIdx.x;
    if (k >= nw) return;
    if (k == 0) data[0] = nw;
    else if (k <= ncopy) { if (data + 1 != rec) data[k] = rec[k - 1]; }
    else data[k] = 0u;
}
/* the 16-bit-slot tokens of a record (two per word, the first in the high half) as one token per 32-bit word */
__global__ __launch_bounds__(256) void k_fst_tok16_widen(unsigned *tok32, const unsigned *slots, size_t n)
{
    const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const unsigned w = slots[k >> 1];
    tok32[k] = (k & 1) ? (w & 0xFFFFu) : (w >> 16);
}
/* datyp 1 header (style 1, compact.tmplc:303-318) from the style-2 words of the record: the type code changes, word 3 keeps its upper 24 bits (the stream
 * starts in its last byte) */
__global__ void k_fst_hdr_style1(unsigned *data, const unsigned *rec)
{
    data[0] = (rec[0] & 0x000FFFFFu) | (((rec[0] >> 20) == 0xFEFu ? 0x7EFu : 0x7FFu) << 20);
    data[1] = rec[1]; data[2] = rec[2]; data[3] = rec[3] & 0xFFFFFF00u;
}
extern "C" int packhip_fst_frame129(unsigned *d_data, const unsigned *d_rec, unsigned nw, unsigned ncopy)
{
    hipLaunchKernelGGL(k_fst_frame129, dim3((nw + 255) / 256), dim3(256), 0, STREAM, d_data, d_rec, nw, ncopy);
    return chk("k_fst_frame129");
}
extern "C" int packhip_fst_plain_from_slots(unsigned *d_data, unsigned *d_tok32, const unsigned *d_rec, size_t n)
{
    hipLaunchKernelGGL(k_fst_tok16_widen, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, STREAM, d_tok32, d_rec + 4, n);
    hipLaunchKernelGGL(k_fst_hdr_style1, dim3(1), dim3(1), 0, STREAM, d_data, d_rec);
    return chk("k_fst_plain_from_slots");
}

/* exchange the 16-bit halves of every word (c_armn_compress_setswap(0) on the multi-kernel path) */
__global__ __launch_bounds__(256) void k_swap_halves(unsigned *dst, const unsigned *src, size_t n)
{
    size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (k < n) { unsigned v = src[k]; dst[k] = v << 16 | v >> 16; }
}
extern "C" int packhip_swap_halves(unsigned *d_dst, const unsigned *d_src, size_t nwords)
{
    if (!nwords) return 0;
    hipLaunchKernelGGL(k_swap_halves, dim3((unsigned)((nwords + 255) / 256)), dim3(256), 0, STREAM, d_dst, d_src, nwords);
    return chk("k_swap_halves");
}

extern "C" size_t packhip_armn_work_bytes(int ni, int nj)
{
    long long ntiles = ((long long)(ni + 2) / 3 + 1) * ((long long)(nj + 2) / 3 + 1);
    return (size_t)ntiles + 64 + 8 * ((size_t)(ntiles / (SCAN_TPB * SCAN_ITEMS)) + 8) + 64;
}
