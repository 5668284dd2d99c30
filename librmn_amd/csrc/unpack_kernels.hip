/* unpack_kernels.hip -- armn_compress UNCOMPRESS on gfx950.
 *
 * Reference: c_armn_compress op UNCOMPRESS (src/compresseur/c_zfstlib.c:181-200), c_fstunzip (:261-292),
 * unpackTokensMinimum (:592-643), unpackTokensParallelogram (:792-872).
 *
 * The stream is a chain: tile t+1's header sits C + n_t x width(header_t) bits after tile t's, so a tile's
 * position is known only once every earlier header has been read (~100 clk per tile for one lane: 0.25 s for the
 * 2.9 M tiles of a cfg2 field, the speed of the CPU decoder).  The chain is resolved in three kernels:
 *
 *   k_armn_dec_spec   speculation, all CUs: the stream is cut into windows of DW = 2048 bits.  A tile is at most `ext`
 *                     bits long, so the chain enters a window at one of its first `ext` bit positions; for EVERY such
 *                     entry the walk through the window (as if a tile header started there, interior point count) is
 *                     recorded: where it leaves the window and how many tiles it saw: tab[window][entry].  One wave per
 *                     window: the successor of every bit position once, then the entries chase through LDS (round 3).
 *   k_armn_dec_hop    one workgroup per FIELD: follows the true chain with one LDS lookup per window -- or per EIGHT windows
 *                     (k_armn_dec_compose_lds builds the composed table by doubling in LDS) -- on one wave; tables and stream are staged in LDS
 *                     in batches of ~100 windows by LDS-DMA (every row of a batch in flight at once).  A window in which a
 *                     row of tiles ends (its last tile holds fewer points, everything behind it shifts) is resolved by the
 *                     WHOLE workgroup when it holds many tiles: successor and tile count of every bit position, pointer
 *                     doubling, the walk to the row's last tile read off its binary digits (round 3); windows with few
 *                     tiles and the last row of tiles are walked tile by tile.  Writes each window's true entry and first
 *                     tile index.
 *   k_armn_dec_emit   all CUs, one lane per window: walks its window from the true entry -> bit position of every tile.
 *
 * Since the end of round 3 the serial kernel is the FALLBACK.  Streams with long rows of tiles (>= 768 tiles: the fields of the cfg2 / cfg5 path) take
 *   k_dsc_*           the chain between row ends resolved in parallel: eight composed windows leave one exit whatever the entry (measured: tools/probe_merge.py),
 *                     so the chain that never meets a row end is known for all eight-window blocks at once; the odd tile that ends a row shifts the tile index
 *                     for a few windows, and the shifts of the ~1200 rows are a recurrence on a table computed for every tile in parallel (see the block in
 *                     front of k_dsc_init).  A stream that form gives up on (a flag per field, set on the device) goes through k_armn_dec_hop as before.
 *
 * then
 *   k_armn_dec_tiles  one thread per tile: reads its tokens.  MINIMUM: value = minimum + token, final.
 *                     PARALLELOGRAM: the sign-extended Lorenzo difference mod 2^16.
 *   k_armn_dec_rows / k_armn_dec_colsum / k_armn_dec_cols   the inverse Lorenzo recurrence
 *                     u(i,j) = d(i,j) + u(i-1,j) + u(i,j-1) - u(i-1,j-1) is a 2-D prefix sum of d once the first row
 *                     and column are written as differences; the reference stores 16-bit results, and so
 *                     the whole sum is carried mod 2^16 (a ring homomorphism: same bits).
 *
 * Fields of a batch are decoded concurrently (grid dimension), the serial hop kernel one field per CU.
 * Output: two 16-bit tokens per word, first in the high half (the reference's half-word swap on little-endian
 * hosts, :191-198), or natural ushort order when c_armn_compress_setswap(0).                               */
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "packhip_shim.h"

extern "C" void *ezhip_get_stream(void);
#define STREAM ((hipStream_t)ezhip_get_stream())

#define DW        2048                  /* bits per window (round 1, before the 8-window table and with row-end windows walked tile by tile: 512 -> 40 ms, 1024 -> 24 ms, 2048 -> 27 ms; round 3, see profiles/r03_experiments.txt) */
#define DWW       (DW / 32)
#define DEXT      416                   /* table entries per window: >= the longest tile (MINIMUM: 4 + 25 x 16 bits) */
#define DTAIL     16                    /* stream words staged past a window (>= DEXT + 32 bits) */
#define HOP_TPB   1024
#define HOP_NP    (DW + DEXT)           /* bit positions a walk inside one window can reach (the last tile may end DEXT bits past it) */
#define HOP_DBL_WORDS (2 * HOP_NP)      /* [2][HOP_NP] words: {successor | tiles << 16} */
#define HOP_DENSE 20                    /* tiles in a row-end window from which the workgroup resolves it (measured: below, wave 0 walking it tile by tile is faster) */
#define HOP_LDS   (156 * 1024)          /* of the CU's 160 KB: 112 windows per staged batch at PARALLELOGRAM's 157 entries per window (96 KB: 69) */

/* LDS-DMA of 16 bytes per lane (gfx950): lane l brings base + voff(l) to lds_byte_addr + 16 l, no VGPRs in between -- every row of a batch is in flight at
 * once (through registers a thread holds 16 loads: two rounds of memory latency per batch of 69 windows were most of the kernel's time).  Through inline
 * asm: completion is ordered by the s_waitcnt vmcnt(0) + barrier that follow (ez_kernels.hip, k_sepx, says why the builtin is avoided) */
__device__ __forceinline__ void dec_dma16(const unsigned *base_uniform, unsigned voff_bytes, unsigned lds_byte_addr)
{
    const char *addr = (const char *)base_uniform + voff_bytes;
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                 :: "v"(addr), "s"(__builtin_amdgcn_readfirstlane((int)lds_byte_addr)) : "memory");
}
__device__ __forceinline__ unsigned dec_lds_addr(const void *p) { return (unsigned)(size_t)(__attribute__((address_space(3))) const void *)p; }

struct DecGeom {
    int method, istep, origin, nbits, C;    /* C: width of a tile header */
    int plain;                               /* a plane of c_armn_compress32: a tile's tokens are hdr + 1 bits wide, no 17-bit escape (armn_compress_32.c:753-824) */
    int ntx, nty, mlast, hlast;
    int n_int, ext;                          /* points of an interior tile; bits of the longest tile */
    long long ntiles;
    unsigned body;                           /* bit position of the first tile header (bit 0 = MSB of z[1]) */
};

__device__ __forceinline__ bool dec_geom(DecGeom &g, const unsigned *z, int ni, int nj, int plain = 0)
{
    const unsigned h = z[0];
    g.plain = plain;
    g.method = h & 15; g.istep = (h >> 7) & 7; g.nbits = (h >> 10) & 31;
    /* the encoder names PARALLELOGRAM with step 3 and MINIMUM with step 5, nothing else (:132, :157) */
    if (g.method == 4 && g.istep == 3) {
        g.origin = 1; g.C = (int)(z[1] >> 29); g.body = 3u + (unsigned)(ni + nj - 1) * (unsigned)g.nbits;
        if (g.C != 4 && g.C != 5) return false;
        g.n_int = 9; g.ext = g.C + 9 * (g.C == 5 ? 32 : 17);
    } else if (g.method == 3 && g.istep == 5) {
        g.origin = 0; g.C = 4; g.body = 0;
        g.n_int = 25; g.ext = 4 + 25 * 16;
    } else return false;
    g.ntx = (ni - g.origin + g.istep - 1) / g.istep; g.nty = (nj - g.origin + g.istep - 1) / g.istep;
    g.mlast = ni - g.origin - (g.ntx - 1) * g.istep; g.hlast = nj - g.origin - (g.nty - 1) * g.istep;
    g.ntiles = (long long)g.ntx * g.nty;
    return true;
}

/* bits of tile `hdr` with n points: the distance to the next header */
__device__ __forceinline__ int dec_step(const DecGeom &g, unsigned hdr, int n)
{
    if (g.method == 4) return g.C + (hdr == 0 ? 0 : n * ((!g.plain && (hdr == 15 || hdr == 16)) ? 17 : (int)hdr + 1));     /* :826-858 */
    return hdr == 0 ? 4 + g.nbits : (hdr >= 15 ? 4 + n * 16 : 4 + g.nbits + n * (int)hdr);                    /* :610-638 */
}

/* nb <= 32 bits at bit position p of an MSB-first word stream */
template <class P>
__device__ __forceinline__ unsigned getbits(P s, unsigned p, int nb)
{
    if (nb == 0) return 0;
    const unsigned long long v = (unsigned long long)s[p >> 5] << 32 | s[(p >> 5) + 1];
    return (unsigned)(v >> (64 - nb - (int)(p & 31))) & (nb == 32 ? 0xFFFFFFFFu : (1u << nb) - 1u);
}
/* the 64 - (p & 31) stream bits from position p on, left-aligned (the rest zero) */
template <class P>
__device__ __forceinline__ unsigned long long bits64(P s, unsigned p) { return ((unsigned long long)s[p >> 5] << 32 | s[(p >> 5) + 1]) << (p & 31u); }
/* PARALLELOGRAM streams and planes: an empty tile is its header alone (C zero bits), so a run of zero bits is a run of empty tiles whatever their point counts.
 * How many whole empty tiles start at p, given v = bits64(s, p), of those that start before `lim`: 0 when the header at p is not empty.  (A constant region -- a
 * mask, dry precipitation -- is thousands of them: taken one by one, 512 per window, they were the long pole of every lane-per-window walk) */
__device__ __forceinline__ unsigned dec_empty_run(const DecGeom &g, unsigned long long v, unsigned p, unsigned lim)
{
    if (g.method != 4) return 0u;
    const unsigned avail = 64u - (p & 31u), z = v ? (unsigned)__clzll((long long)v) : 64u;
    const unsigned k = (z < avail ? z : avail) / (unsigned)g.C, kmax = (lim - p + (unsigned)g.C - 1u) / (unsigned)g.C;
    return k < kmax ? k : kmax;
}
/* word w of the stream, 0 behind its end.  No branch around the load (index clamped, result masked): behind `w < nwords ? z1[w] : 0` the compiler put every load in a
 * block of its own and waited for it before the next one -- the two words of a getbits_g one after the other, 80 staged words per lane as 80 round trips */
__device__ __forceinline__ unsigned gword(const unsigned *z1, size_t w, size_t nwords)
{
    const bool in = w < nwords;
    const unsigned v = *(in ? z1 + w : z1 - 1);          /* (z1 = the stream + 1 in every caller: the header word is always there) */
    return in ? v : 0u;
}
__device__ __forceinline__ unsigned getbits_g(const unsigned *z1, size_t nwords, unsigned p, int nb)
{
    if (nb == 0) return 0;
    const unsigned long long v = (unsigned long long)gword(z1, p >> 5, nwords) << 32 | gword(z1, (p >> 5) + 1, nwords);
    return (unsigned)(v >> (64 - nb - (int)(p & 31))) & (nb == 32 ? 0xFFFFFFFFu : (1u << nb) - 1u);
}

/* points of tile t and the number of consecutive tiles from t (t included) that have the same point count */
__device__ __forceinline__ void dec_run(const DecGeom &g, long long t, int &n, long long &run)
{
    const int y = (int)(t / g.ntx), x = (int)(t % g.ntx);
    const int h = (y == g.nty - 1) ? g.hlast : g.istep, m = (x == g.ntx - 1) ? g.mlast : g.istep;
    n = m * h;
    if (g.mlast != g.istep) run = (x < g.ntx - 1) ? (g.ntx - 1 - x) : 1;
    else if (y == g.nty - 1 || g.hlast == g.istep) run = g.ntiles - t;
    else run = (long long)(g.nty - 1 - y) * g.ntx - x;
}

/* tab[window][entry] = (bits past the window's end where the walk from `entry` lands) | tiles seen << 16.
 * One WAVE per window, four windows per thread block, no block-wide barrier (round 3: a thread block per window was bound by the dispatch of 6.3 M tiny
 * thread blocks per 32 cfg5 streams, and walked every entry with a header decode per tile): the successor of EVERY bit position once (16 per lane), then
 * the entries chase through LDS -- one 16-bit read per tile; the ext walks of a window merge after a few tiles (157 x 15 - 38 decodes per window before, 1024 now) */
#define SPEC_WPB 4
__global__ __launch_bounds__(64 * SPEC_WPB) void k_armn_dec_spec(const unsigned *z_all, size_t z_stride, size_t z_words, int ni, int nj,
                                                                unsigned *tab_all, size_t tab_stride, int nwin, int plain)
{
    __shared__ unsigned s_all[SPEC_WPB][DWW + DTAIL];
    __shared__ __attribute__((aligned(16))) unsigned short nx_all[SPEC_WPB][DW];
    const int f = blockIdx.y, wv = threadIdx.x >> 6, lane = threadIdx.x & 63, win = blockIdx.x * SPEC_WPB + wv;
    if (win >= nwin) return;                                      /* (a whole wave leaves: nothing below synchronises across waves) */
    const unsigned *z = z_all + (size_t)f * z_stride;
    DecGeom g;
    if (!dec_geom(g, z, ni, nj, plain)) return;
    const size_t nwords = z_words - 1;
    unsigned *s = s_all[wv];
    unsigned short *nx = nx_all[wv];
    unsigned mine = 0;
    for (int k = lane; k < DWW + DTAIL; k += 64) { const unsigned v = gword(z + 1, (size_t)win * DWW + k, nwords); s[k] = v; mine |= v; }
    const bool nonzero = __ballot(mine != 0) != 0;
    unsigned *tab = tab_all + (size_t)f * tab_stride + (size_t)win * DEXT;
    const unsigned step0 = (unsigned)dec_step(g, 0, g.n_int);
    if (!nonzero) {                                               /* zero words (past the end of the stream): every header reads 0 */
        for (int e = lane; e < g.ext; e += 64) { const unsigned cnt = (DW - (unsigned)e + step0 - 1) / step0; tab[e] = ((unsigned)e + cnt * step0 - DW) | cnt << 16; }
        return;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    /* the successor of every bit position: a lane takes the positions 2 lane, 2 lane + 1 of every group of 128 -- one 64-bit read of the stream serves both
     * headers (C <= 5 bits each), the two 16-bit successors go out as one word, lane after lane.  (Round 5; a position per lane and iteration read the stream
     * twice as often and wrote half-words.  The kernel's time did not move -- 2.40 ms per 16 cfg5 streams before and after: it is the chase below, ~36 dependent
     * LDS reads per lane at scattered addresses with the LDS array 63 % busy, that bounds it: profiles/r05_experiments.txt) */
    {
        unsigned *nx32 = (unsigned *)nx;
#pragma unroll 4
        for (int k = 0; k < DW / 128; k++) {
            const unsigned p0 = 128u * (unsigned)k + 2u * (unsigned)lane, wi = p0 >> 5, sh = p0 & 31u;
            const unsigned long long v = ((unsigned long long)s[wi] << 32 | s[wi + 1]) << sh;
            const unsigned t = (unsigned)(v >> 32);                                   /* the 32 bits from position p0 on */
            const unsigned h0 = t >> (32 - g.C), h1 = (t << 1) >> (32 - g.C);
            nx32[64 * k + lane] = ((p0 + (unsigned)dec_step(g, h0, g.n_int)) & 0xFFFFu) | (p0 + 1u + (unsigned)dec_step(g, h1, g.n_int)) << 16;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int e = lane; e < g.ext; e += 64) {
        unsigned p = (unsigned)e, cnt = 0;
        while (p < DW) { p = nx[p]; cnt++; }
        tab[e] = (p - DW) | cnt << 16;
    }
}

/* tab8[w][e]: the same pair after EIGHT windows starting at window w (exit offset into window w + 8, tiles seen);
 * tiles = 0xFFFF marks "not usable" (the table would run past the last window or the count does not fit) */
__global__ __launch_bounds__(256) void k_armn_dec_compose(const unsigned *z_all, size_t z_stride, int ni, int nj,
                                                          const unsigned *tab_all, unsigned *tab8_all, size_t tab_stride, int nwin)
{
    const int f = blockIdx.y, w = blockIdx.x, tid = threadIdx.x;
    DecGeom g;
    if (!dec_geom(g, z_all + (size_t)f * z_stride, ni, nj)) return;
    const unsigned *tab = tab_all + (size_t)f * tab_stride;
    unsigned *tab8 = tab8_all + (size_t)f * tab_stride + (size_t)w * DEXT;
    for (int e0 = tid; e0 < g.ext; e0 += 256) {
        unsigned e = (unsigned)e0, cnt = 0;
        bool ok = w + 8 <= nwin;
        for (int k = 0; k < 8 && ok; k++) {
            const unsigned v = tab[(size_t)(w + k) * DEXT + e];
            cnt += v >> 16; e = v & 0xFFFFu;
            if (e >= (unsigned)g.ext || cnt >= 0xFFFFu) ok = false;
        }
        tab8[e0] = ok ? (e | cnt << 16) : 0xFFFF0000u;
    }
}

/* the same table by DOUBLING in LDS (round 3): a thread block takes G consecutive windows, stages the G + 7 rows they need once and composes
 * 1 -> 2 -> 4 -> 8 windows: three LDS lookups per entry instead of eight dependent global loads (10.8 ms per 32 cfg5 streams at ratio 0.5: latency-bound).
 * An element that cannot be chained (exit beyond the table, count at the sentinel, window past the last one) is the sentinel 0xFFFF0000; counts only grow, so
 * checking the sums after 2, 4 and 8 windows marks exactly the entries the step-by-step form marked */
#define CMP_INV 0xFFFF0000u
#define CMP_TPB 1024                    /* two thread blocks of 64 KB per CU: 32 waves, ~6 independent loads per thread while a group is staged */
__device__ __forceinline__ unsigned dec_compose2(unsigned a, const unsigned *next_row)
{
    if (a == CMP_INV) return CMP_INV;
    const unsigned v = next_row[a & 0xFFFFu];
    if (v == CMP_INV) return CMP_INV;
    const unsigned cnt = (a >> 16) + (v >> 16);
    return cnt >= 0xFFFFu ? CMP_INV : ((v & 0xFFFFu) | cnt << 16);
}
__global__ __launch_bounds__(CMP_TPB) void k_armn_dec_compose_lds(const unsigned *z_all, size_t z_stride, int ni, int nj,
                                                              const unsigned *tab_all, unsigned *tab8_all, size_t tab_stride, int nwin, int G, int lds_words)
{
    extern __shared__ unsigned cl[];
    const int f = blockIdx.y, tid = threadIdx.x;
    DecGeom g;
    if (!dec_geom(g, z_all + (size_t)f * z_stride, ni, nj)) return;
    const int ext = g.ext, ext4 = (ext + 3) & ~3;
    const unsigned *tab = tab_all + (size_t)f * tab_stride;
    unsigned *tab8 = tab8_all + (size_t)f * tab_stride;
    /* the launch sizes the LDS for PARALLELOGRAM's 157 entries per window; a MINIMUM stream (404) takes its G windows in smaller groups */
    const int Gc = min(G, (lds_words / ext4 - 17) / 3);
    const int Wend = min(blockIdx.x * G + G, nwin);
    for (int W = blockIdx.x * G; W < Wend; W += Gc) {
        const int Gn = min(Gc, Wend - W);
        unsigned *t1 = cl, *t2 = t1 + (size_t)(Gn + 7) * ext4, *t4 = t2 + (size_t)(Gn + 6) * ext4;
        __syncthreads();                                              /* the previous group is done with the buffers */
        const float rcp4 = 1.0f / (float)ext4;                        /* k / ext4 for k < 2^16: (int)((k + 0.5) / ext4) is exact */
        auto row_of = [&](int k) { return (int)(((float)k + 0.5f) * rcp4); };
        {   /* staging: every thread's loads issued before its first store (clamped addresses: unconditional loads stay in flight together) */
            constexpr int U = 6;
            const int total = (Gn + 7) * ext4, wlast = nwin - 1;
            for (int k0 = tid; k0 < total; k0 += CMP_TPB * U) {
                unsigned v[U];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const int k = min(k0 + u * CMP_TPB, total - 1), r = row_of(k), e = k - r * ext4;
                    v[u] = tab[(size_t)min(W + r, wlast) * DEXT + min(e, ext - 1)];
                }
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const int k = k0 + u * CMP_TPB;
                    if (k < total) {
                        const int r = row_of(k), e = k - r * ext4;
                        const bool ok = W + r < nwin && e < ext && (v[u] & 0xFFFFu) < (unsigned)ext && (v[u] >> 16) < 0xFFFFu;
                        t1[k] = ok ? v[u] : CMP_INV;
                    }
                }
            }
        }
        __syncthreads();
        for (int k = tid; k < (Gn + 6) * ext4; k += CMP_TPB) { const int r = row_of(k); t2[k] = dec_compose2(t1[k], t1 + (size_t)(r + 1) * ext4); }
        __syncthreads();
        for (int k = tid; k < (Gn + 4) * ext4; k += CMP_TPB) { const int r = row_of(k); t4[k] = dec_compose2(t2[k], t2 + (size_t)(r + 2) * ext4); }
        __syncthreads();
        for (int k = tid; k < Gn * ext4; k += CMP_TPB) {
            const int r = row_of(k), e = k - r * ext4, w = W + r;
            if (e < ext) tab8[(size_t)w * DEXT + e] = (w + 8 <= nwin) ? dec_compose2(t4[k], t4 + (size_t)(r + 4) * ext4) : CMP_INV;
        }
    }
}

__device__ __forceinline__ unsigned uni(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }

/* 32-bit twin of dec_run for the serial walker (tile counts are < 2^31, checked by the host) */
__device__ __forceinline__ void dec_run32(const DecGeom &g, unsigned t, int &n, unsigned &run)
{
    const unsigned y = t / (unsigned)g.ntx, x = t - y * (unsigned)g.ntx;
    const int h = ((int)y == g.nty - 1) ? g.hlast : g.istep, m = ((int)x == g.ntx - 1) ? g.mlast : g.istep;
    n = m * h;
    if (g.mlast != g.istep) run = ((int)x < g.ntx - 1) ? (unsigned)(g.ntx - 1) - x : 1u;
    else if ((int)y == g.nty - 1 || g.hlast == g.istep) run = (unsigned)g.ntiles - t;
    else run = (unsigned)(g.nty - 1 - (int)y) * (unsigned)g.ntx - x;
}

/* went[2 w] = entry bit of window w (0xFFFFFFFF: the chain has no tile header in it), went[2 w + 1] = its first tile.
 * The walk runs on wave 0 with every value forced wave-uniform (readfirstlane), i.e. on the scalar unit: one lane of
 * vector code costs ~10 clk per dependent instruction, scalar code about half of that and its branches are free. */
__global__ __launch_bounds__(HOP_TPB) void k_armn_dec_hop(const unsigned *z_all, size_t z_stride, size_t z_words, int ni, int nj,
                                                          const unsigned *tab_all, const unsigned *tab8_all, size_t tab_stride,
                                                          unsigned *went_all, unsigned *went8_all, size_t went_stride,
                                                          int nwin, int *status, int dbg, int plain, const unsigned *skip, size_t skip_stride)
{
    extern __shared__ unsigned lds[];
    if (skip && skip[(size_t)blockIdx.x * skip_stride]) return;          /* the parallel form (k_dsc_*) has the field's tile positions already */
    __shared__ unsigned s_pos, s_done, s_t, s_mid, s_state, s_rw, s_re, s_rcnt, s_dense;
    const int f = blockIdx.x, tid = threadIdx.x;
    const unsigned *z = z_all + (size_t)f * z_stride;
    DecGeom g;
    if (!dec_geom(g, z, ni, nj, plain)) { if (tid == 0) status[f] = -1; return; }
    const unsigned *z1 = z + 1;
    const size_t nwords = z_words - 1;
    const unsigned *tab = tab_all + (size_t)f * tab_stride, *tab8 = tab8_all + (size_t)f * tab_stride;
    unsigned *went = went_all + (size_t)f * went_stride, *went8 = went8_all + (size_t)f * went_stride;
    const int ext = (int)uni((unsigned)g.ext), C = (int)uni((unsigned)g.C), n_int = g.n_int;
    const unsigned ntiles = uni((unsigned)g.ntiles);
    const int ext4 = (ext + 3) & ~3, cpr = ext4 >> 2;                  /* LDS row stride (words), 16-byte chunks per row */
    const int nb = (HOP_LDS / 4 - DTAIL - HOP_DBL_WORDS) / (2 * ext4 + DWW + 4);       /* windows per staged batch */
    unsigned *tabs = lds, *tabs8 = lds + (size_t)nb * ext4, *strm = tabs8 + (size_t)nb * ext4, *wl = strm + (size_t)nb * DWW + DTAIL, *wl8 = wl + 2 * nb;
    unsigned *dbl = wl8 + 2 * nb;                                     /* [2][HOP_NP]: successor and tile count of every bit position of ONE window, double-buffered */
    if (tid == 0) { s_pos = g.body; s_t = 0; s_done = 0; s_mid = 0; s_dense = 0; }
    __syncthreads();
    for (;;) {
        const unsigned pos0 = s_pos;
        const int b0 = (int)(pos0 / DW);
        if (s_done || b0 >= nwin) break;
        const int nbw = min(nb, nwin - b0);
        /* staging: a lone workgroup is latency-bound (bytes in flight / latency), so every thread issues up to 16
         * independent 16-byte loads before the first store: lane = 16-byte chunk of a row, 8 rows per pass, 8 passes */
        if (dbg & 2) {                                            /* development: the round-1 staging through registers (two tables x 64 rows per round trip) */
        for (int row0 = 0; row0 < nbw; row0 += 8 * 8) {
            uint4 r[8], r8[8];
            const int ch = tid & 127, rl = tid >> 7;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int row = min(row0 + 8 * k + rl, nbw - 1), chc = min(ch, cpr - 1);      /* unconditional (clamped): the loads must all be in flight together */
                r[k] = *(const uint4 *)(tab + (size_t)(b0 + row) * DEXT + 4 * chc);
                r8[k] = *(const uint4 *)(tab8 + (size_t)(b0 + row) * DEXT + 4 * chc);
            }
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int row = row0 + 8 * k + rl;
                if (ch < cpr && row < nbw) { *(uint4 *)(tabs + (size_t)row * ext4 + 4 * ch) = r[k]; *(uint4 *)(tabs8 + (size_t)row * ext4 + 4 * ch) = r8[k]; }
            }
        }
        } else {
            /* wave v brings rows v, v + 16, ... of both tables: a row is cpr 16-byte pieces, one per lane (two instructions for MINIMUM's 101 pieces) */
            const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), ln = tid & 63;
            for (int row = wv; row < nbw; row += HOP_TPB / 64) {
                const unsigned *g1 = tab + (size_t)(b0 + row) * DEXT, *g8 = tab8 + (size_t)(b0 + row) * DEXT;
                for (int c0 = 0; c0 < cpr; c0 += 64) {
                    if (c0 + ln < cpr) {
                        dec_dma16(g1, (unsigned)(c0 + ln) * 16u, dec_lds_addr(tabs + (size_t)row * ext4 + 4 * c0));
                        dec_dma16(g8, (unsigned)(c0 + ln) * 16u, dec_lds_addr(tabs8 + (size_t)row * ext4 + 4 * c0));
                    }
                }
            }
        }
        {   /* stream words of the batch (+ tail): clamped unconditional loads, zero past the end of the stream */
            const int total = nbw * DWW + DTAIL;
            for (int k0 = 0; k0 < total; k0 += 4 * HOP_TPB) {
                unsigned r[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const size_t w = (size_t)b0 * DWW + (size_t)min(k0 + k * HOP_TPB + tid, total - 1);
                    const unsigned v = z1[w < nwords ? w : nwords - 1];
                    r[k] = w < nwords ? v : 0u;
                }
#pragma unroll
                for (int k = 0; k < 4; k++) if (k0 + k * HOP_TPB + tid < total) strm[k0 + k * HOP_TPB + tid] = r[k];
            }
        }
        for (int k = tid; k < 2 * nbw; k += HOP_TPB) { wl[k] = 0xFFFFFFFFu; wl8[k] = 0xFFFFFFFFu; }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          /* the DMA'd rows of this wave have landed; the barrier makes that true for all */
        __syncthreads();
        if (dbg & 1) { if (tid == 0) { s_pos = (unsigned)(b0 + nbw) * DW; } __syncthreads(); continue; }
      for (;;) {                                                  /* segments of the batch: wave 0 walks; a window that holds a row end is resolved by the whole workgroup */
        if (tid < 64) {
            unsigned pos = uni(s_pos), t = uni(s_t), run;
            int n;
            bool mid = uni(s_mid) != 0;                               /* pos lies INSIDE a window whose entry is already recorded (the workgroup left it there) */
            bool dense = uni(s_dense) != 0;                           /* the last row-end window held many tiles: such windows go to the workgroup (few: one wave is faster) */
            unsigned state = 0;
            dec_run32(g, t, n, run);
            n = (int)uni((unsigned)n); run = uni(run);
            const unsigned nbw_u = (unsigned)nbw, ext_u = (unsigned)ext, ext4_u = (unsigned)ext4, base_pos = (unsigned)b0 * DW;
            auto step_tile = [&](unsigned &pp, int nn) {            /* one tile header at pp: its length with nn points */
                const unsigned q = pp - base_pos;
                const unsigned long long v = (unsigned long long)uni(strm[q >> 5]) << 32 | uni(strm[(q >> 5) + 1]);
                const unsigned hdr = (unsigned)(v >> (64 - C - (int)(q & 31))) & ((1u << C) - 1u);
                pp += (unsigned)dec_step(g, hdr, nn);
            };
            while (t < ntiles) {
                unsigned wr = pos / DW - (unsigned)b0;
                if (wr >= nbw_u) break;
                unsigned e = pos & (DW - 1);
                if (!mid) { wl[2 * wr] = e; wl[2 * wr + 1] = t; }
                if (n == n_int && !mid && e < ext_u) {
                    /* the table path as two tight loops (the one loop with every case in it compiled into a state machine of ~100 scalar instructions per
                     * step: 0.4 us per step for the one wave of the CU) */
                    bool settled = false;                             /* (wr, e, t) is a window entry not recorded yet: back to the top */
                    unsigned cnt = 0;
                    while (wr + 8 <= nbw_u) {                         /* eight windows at once while no point-count change lies within */
                        const unsigned v8 = uni(tabs8[wr * ext4_u + e]), cnt8 = v8 >> 16;
                        if (cnt8 > run) break;                        /* (0xFFFF = no composed entry: larger than any run) */
                        wl8[2 * wr] = e; wl8[2 * wr + 1] = t;
                        t += cnt8; run -= cnt8; wr += 8; e = v8 & 0xFFFFu;
                        if (run == 0 || e >= ext_u || wr >= nbw_u) { settled = true; break; }
                        wl[2 * wr] = e; wl[2 * wr + 1] = t;
                    }
                    if (!settled) {
                        for (;;) {                                    /* window by window up to the one that holds the row's end */
                            const unsigned v = uni(tabs[wr * ext4_u + e]);
                            cnt = v >> 16;
                            if (cnt > run) break;
                            t += cnt; run -= cnt; wr += 1; e = v & 0xFFFFu;
                            if (run == 0 || e >= ext_u || wr >= nbw_u) { settled = true; break; }
                            wl[2 * wr] = e; wl[2 * wr + 1] = t;
                        }
                    }
                    pos = base_pos + wr * DW + e;
                    if (settled) {
                        if (run == 0 && t < ntiles) { dec_run32(g, t, n, run); n = (int)uni((unsigned)n); run = uni(run); }
                        continue;
                    }
                    dense = cnt >= ((dbg >> 8) ? (unsigned)(dbg >> 8) : (unsigned)HOP_DENSE);      /* the row ends inside window wr (its entry is recorded); development: threshold in EZHIP_DEC_DEBUG's bits 8.. */
                    if (dense && !(dbg & 4)) { s_rw = wr; s_re = e; s_rcnt = cnt; state = 1; break; }
                } else if (n == n_int && dense && !(dbg & 4)) {       /* no table entry for this position (behind a row's last tile, or beyond the table): the workgroup */
                    s_rw = wr; s_re = e; s_rcnt = (DW - e) / (unsigned)C + 1u; state = 1; break;
                } else if (n != n_int && run <= 2 && dense && !(dbg & 4)) {        /* the odd tile(s) at a row's end: stepped over here, the rest of the window is the workgroup's */
                    const unsigned w_abs = pos / DW;
                    while (run > 0 && t < ntiles) { step_tile(pos, n); t++; run--; }
                    if (t < ntiles) { dec_run32(g, t, n, run); n = (int)uni((unsigned)n); run = uni(run); }
                    mid = pos / DW == w_abs;
                    continue;
                }
                /* tile by tile to the window's end: row-end windows that hold few tiles, the last row of tiles (every tile of it is short) */
                const unsigned end = base_pos + (wr + 1) * DW;
                while (pos < end && t < ntiles) {
                    while (pos < end && run > 0) { step_tile(pos, n); t++; run--; }
                    if (run == 0 && t < ntiles) { dec_run32(g, t, n, run); n = (int)uni((unsigned)n); run = uni(run); }
                }
                mid = false;
            }
            if (tid == 0) { s_pos = pos; s_t = t; s_state = state; s_mid = 0; s_dense = dense ? 1u : 0u; if (t >= ntiles) s_done = 1; }
        }
        __syncthreads();
        if (s_state == 0) break;
        /* ---- a window with a row end, all threads: successor and tile count of EVERY bit position of the window under the interior point count, then
         * pointer doubling (up to ten rounds: a window holds at most 512 tiles): after it nxt[p] = where the walk from p leaves the window, cnt[p] = its
         * tiles; on the way lane 0 follows the binary digits of `run` from the entry to the row's last tile (one wave walking a window tile by tile
         * cost ~600 clk per tile: 38 tiles per row-end window at a compression ratio of 0.2, 1200 such windows per cfg5 record) ---- */
        {
            const unsigned wr = s_rw, e = s_re, t0 = s_t;
            int n0; unsigned run0;
            dec_run32(g, t0, n0, run0);                                /* n0 == n_int here */
            /* {successor | tiles << 16} per bit position, two buffers; positions past the window's end are their own successor with no tile (both buffers) */
            unsigned *sc[2] = {(unsigned *)dbl, (unsigned *)dbl + HOP_NP};
            for (int p = tid; p < HOP_NP; p += HOP_TPB) {
                unsigned v_ = (unsigned)p;
                if (p < DW) {
                    const unsigned q = wr * DW + (unsigned)p;
                    const unsigned long long v = (unsigned long long)strm[q >> 5] << 32 | strm[(q >> 5) + 1];
                    const unsigned hdr = (unsigned)(v >> (64 - C - (int)(q & 31))) & ((1u << C) - 1u);
                    v_ = (unsigned)(p + dec_step(g, hdr, n_int)) | 1u << 16;
                } else sc[1][p] = v_;
                sc[0][p] = v_;
            }
            /* rounds: the walk from e holds s_rcnt tiles (its table entry; without one at most (DW - e) / C + 1); the walk behind the row's last tile is
             * checked for having left the window below */
            int rounds = 1;
            while (rounds < 10 && (1u << rounds) <= s_rcnt) rounds++;                  /* (a window holds at most DW / 4 = 512 tiles) */
            __syncthreads();
            unsigned pc = e;
            int cur = 0;
            for (int j = 0; j < rounds; j++) {
                for (int p = tid; p < DW; p += HOP_TPB) {                                /* (DW / HOP_TPB positions per thread) */
                    const unsigned w_ = sc[cur][p], w2 = sc[cur][w_ & 0xFFFFu];
                    sc[cur ^ 1][p] = (w2 & 0xFFFFu) | ((w_ >> 16) + (w2 >> 16)) << 16;
                }
                if (tid == 0 && ((run0 >> j) & 1u)) pc = sc[cur][pc] & 0xFFFFu;
                __syncthreads();
                cur ^= 1;
            }
            const unsigned *nxcn = sc[cur];
            if (tid == 0) {
                const unsigned wbase = ((unsigned)b0 + wr) * DW;
                unsigned pos, t, mid = 0;
                const unsigned cexit = nxcn[e] >> 16, xexit = nxcn[e] & 0xFFFFu;
                if (xexit >= DW && cexit <= run0) { pos = wbase + xexit; t = t0 + cexit; }        /* no row end in the window after all (an entry beyond the table) */
                else {
                    t = t0 + run0;                                     /* the row's last tile sits at pc */
                    unsigned p2 = pc;
                    int n1 = n_int; unsigned run1 = 0;
                    if (t < ntiles) {
                        dec_run32(g, t, n1, run1);
                        for (unsigned k = 0; k < run1 && p2 < DW && t < ntiles && n1 != n_int; k++) {        /* (one tile; two in no geometry) */
                            const unsigned q = wr * DW + p2;
                            const unsigned long long v = (unsigned long long)strm[q >> 5] << 32 | strm[(q >> 5) + 1];
                            const unsigned hdr = (unsigned)(v >> (64 - C - (int)(q & 31))) & ((1u << C) - 1u);
                            p2 += (unsigned)dec_step(g, hdr, n1); t++;
                        }
                    }
                    pos = wbase + p2;
                    if (p2 < DW && t < ntiles) {
                        int n2; unsigned run2;
                        dec_run32(g, t, n2, run2);
                        if (n2 == n_int && (nxcn[p2] & 0xFFFFu) >= DW && (nxcn[p2] >> 16) <= run2) { pos = wbase + (nxcn[p2] & 0xFFFFu); t += nxcn[p2] >> 16; }      /* the next row's tiles up to the window's end */
                        else mid = 1;                                   /* (rows shorter than a window, the last row): wave 0 goes on from inside the window */
                    }
                }
                s_pos = pos; s_t = t; s_mid = mid; if (t >= ntiles) s_done = 1;
            }
            __syncthreads();
        }
      }
        for (int k = tid; k < 2 * nbw; k += HOP_TPB) { went[2 * (size_t)b0 + k] = wl[k]; went8[2 * (size_t)b0 + k] = wl8[k]; }
        __syncthreads();                                          /* the next batch re-arms wl / wl8 right away (its staging no longer waits in between) */
    }
    if (tid == 0) status[f] = s_done ? ((plain && (unsigned long long)s_pos > 32ull * nwords) ? -2 : 0) : -2;      /* (a plane: its last tile ends inside the stream) */
}

/* the seven windows an eight-window hop jumped over: their entries follow from the one-window table */
__global__ __launch_bounds__(256) void k_armn_dec_expand8(const unsigned *z_all, size_t z_stride, int ni, int nj, const unsigned *tab_all, size_t tab_stride,
                                                          unsigned *went_all, const unsigned *went8_all, size_t went_stride, int nwin, const unsigned *skip, size_t skip_stride)
{
    const int f = blockIdx.y;
    if (skip && skip[(size_t)f * skip_stride]) return;
    const int w = blockIdx.x * 256 + threadIdx.x;
    if (w >= nwin) return;
    DecGeom g;
    if (!dec_geom(g, z_all + (size_t)f * z_stride, ni, nj)) return;
    const unsigned *went8 = went8_all + (size_t)f * went_stride, *tab = tab_all + (size_t)f * tab_stride;
    unsigned *went = went_all + (size_t)f * went_stride;
    unsigned e = went8[2 * (size_t)w];
    if (e == 0xFFFFFFFFu) return;
    unsigned t = went8[2 * (size_t)w + 1];
    for (int k = 0; k < 8 && w + k < nwin; k++) {
        went[2 * (size_t)(w + k)] = e; went[2 * (size_t)(w + k) + 1] = t;
        const unsigned v = tab[(size_t)(w + k) * DEXT + e];
        t += v >> 16; e = v & 0xFFFFu;
    }
}

__global__ __launch_bounds__(256) void k_armn_dec_emit(const unsigned *z_all, size_t z_stride, size_t z_words, int ni, int nj,
                                                       const unsigned *went_all, size_t went_stride, unsigned *tilepos_all, size_t tp_stride, int nwin, int plain, const unsigned *skip, size_t skip_stride)
{
    const int f = blockIdx.y;
    if (skip && skip[(size_t)f * skip_stride]) return;
    const int w = blockIdx.x * 256 + threadIdx.x;
    if (w >= nwin) return;
    const unsigned *z = z_all + (size_t)f * z_stride;
    DecGeom g;
    if (!dec_geom(g, z, ni, nj, plain)) return;
    const unsigned e = went_all[(size_t)f * went_stride + 2 * (size_t)w];
    if (e == 0xFFFFFFFFu) return;
    long long t = went_all[(size_t)f * went_stride + 2 * (size_t)w + 1];
    unsigned *tilepos = tilepos_all + (size_t)f * tp_stride;
    const size_t nwords = z_words - 1;
    unsigned pos = (unsigned)w * DW + e;
    const unsigned end = (unsigned)(w + 1) * DW;
    int x = (int)(t % g.ntx), y = (int)(t / g.ntx);
    while (pos < end && t < g.ntiles) {
        tilepos[t] = pos;
        const int n = ((x == g.ntx - 1) ? g.mlast : g.istep) * ((y == g.nty - 1) ? g.hlast : g.istep);
        pos += (unsigned)dec_step(g, getbits_g(z + 1, nwords, pos, g.C), n);
        t++;
        if (++x == g.ntx) { x = 0; y++; }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* token k of the output goes to half-word k^1 when the stream is swapped (high half first)      */
__device__ __forceinline__ size_t out_slot(size_t k, int swap) { return swap ? (k ^ 1) : k; }


/* ============================================================================================================================================
 * The chain between row ends resolved in parallel (round 3, opt-in: EZHIP_DEC_SCAN=1; DESIGN_LOG.md 9 item 4, tools/probe_merge.py).
 * The walks of eight composed windows leave ONE exit for (nearly) all entries: the entry of an eight-window block follows from the block before it alone.  So
 * the CANONICAL chain -- the one from the first tile that never meets a row end -- is written down for all blocks at once (k_dsc_m8, k_dsc_blocks, k_dsc_scan,
 * k_armn_dec_expand8, k_dsc_emit: the position of every canonical tile).  The real chain leaves it at the odd tile that ends a row of tiles and rejoins it
 * within a few windows, its tile index shifted.  k_dsc_dtab computes, for EVERY canonical tile c, the change of that shift if the row ended at c; the shifts of
 * the 1200 rows are then a recurrence of 1200 table lookups (k_dsc_rows); k_dsc_stretch re-walks the stretches between an odd tile and the rejoining point,
 * k_dsc_final shifts the rest, k_dsc_lastrow walks the last row of tiles (another height) tile by tile.  Anything unexpected (short rows, a block that does not
 * merge for long, a stretch that does not rejoin before the next row end) leaves st[DSC_OK] at 0 and the serial chain kernel runs as before.
 * ============================================================================================================================================ */
enum { DSC_OK = 0, DSC_WB, DSC_EB, DSC_TB, DSC_K, DSC_NCANON, DSC_FAIL, DSC_RINT, DSC_WORDS = 16 };
#define DSC_D_INVALID 0x7FFF
#define DM_MAXLEV 9                     /* levels of composed window maps (k_dmin_up): blocks of 8^l windows */
/* canonical tiles kept beyond the field's own tile count: the canonical chain reads the bits of a row's narrow last tile as interior tiles and runs ahead of the
 * real tile index by some tens of tiles per row (mantissa plane of a 7200 x 3601 field: 67 596 at the last row) */
static inline size_t dsc_margin(size_t max_tiles) { return max_tiles / 4 > 65536 ? max_tiles / 4 : 65536; }
#define DSC_LIST 512                     /* windows a stretch may take to rejoin the canonical chain (mantissa planes of c_armn_compress32: 16 on average, 130 seen) */

__device__ __forceinline__ bool dsc_eligible(const DecGeom &g, int nwin, int min_ntx) { return g.ntx >= min_ntx && g.nty >= 3 && nwin >= 64; }       /* (both methods: the geometry and dec_step carry the difference) */

/* one wave per field: the first (partial) window tile by tile, the block origin */
__global__ __launch_bounds__(64) void k_dsc_init(const unsigned *z_all, size_t z_stride, size_t z_words, int ni, int nj, unsigned *st_all, size_t st_stride,
                                                 unsigned *wentc_all, size_t went_stride, int nwin, int plain, int min_ntx,
                                                 const unsigned *done_all = nullptr, size_t done_stride = 0)     /* second use (the composed ragged form): the first form's state */
{
    const int f = blockIdx.x;
    if (threadIdx.x) return;
    unsigned *st = st_all + (size_t)f * st_stride;
    for (int k = 0; k < DSC_WORDS; k++) st[k] = 0;
    const unsigned *z = z_all + (size_t)f * z_stride;
    DecGeom g;
    if (!dec_geom(g, z, ni, nj, plain) || !dsc_eligible(g, nwin, min_ntx)) { st[DSC_FAIL] = 1; return; }
    if (done_all && (done_all[(size_t)f * done_stride + DSC_OK] || g.mlast == g.istep)) { st[DSC_FAIL] = 100; return; }      /* resolved already; whole rows of tiles (k_dmin_* has those) */
    const size_t nwords = z_words - 1;
    unsigned *wentc = wentc_all + (size_t)f * went_stride;
    unsigned pos = g.body, t = 0;
    const unsigned w0 = pos / DW, end = (w0 + 1) * DW;
    if ((int)w0 + 17 >= nwin) { st[DSC_FAIL] = 2; return; }
    wentc[2 * w0] = pos - w0 * DW; wentc[2 * w0 + 1] = 0;
    while (pos < end) { pos += (unsigned)dec_step(g, getbits_g(z + 1, nwords, pos, g.C), g.n_int); t++; }
    if (t + 1 >= (unsigned)g.ntx) { st[DSC_FAIL] = 12; return; }          /* a row of tiles ends inside the first window */
    st[DSC_WB] = w0 + 1; st[DSC_EB] = pos - end; st[DSC_TB] = t; st[DSC_K] = ((unsigned)nwin - (w0 + 1)) / 8u;
    st[DSC_RINT] = (unsigned)((g.hlast != g.istep) ? g.nty - 1 : g.nty);
}

/* m8[w] = the exit all entries of tab8[w] share, 0xFFFF when they do not (or one cannot be composed); one wave per window */
__global__ __launch_bounds__(256) void k_dsc_m8(const unsigned *z_all, size_t z_stride, int ni, int nj, const unsigned *st_all, size_t st_stride,
                                                const unsigned *tab8_all, size_t tab_stride, unsigned short *m8_all, size_t m8_stride, int nwin, int plain)
{
    const int f = blockIdx.y, w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (w >= nwin || st_all[(size_t)f * st_stride + DSC_FAIL]) return;
    DecGeom g;
    if (!dec_geom(g, z_all + (size_t)f * z_stride, ni, nj, plain)) return;
    const unsigned *row = tab8_all + (size_t)f * tab_stride + (size_t)w * DEXT;
    const unsigned x0 = row[0];
    bool same = x0 != CMP_INV;
    for (int e = lane; e < g.ext; e += 64) { const unsigned v = row[e]; if (v == CMP_INV || (v & 0xFFFFu) != (x0 & 0xFFFFu)) same = false; }
    const bool all = __ballot(!same) == 0ull;
    if (lane == 0) m8_all[(size_t)f * m8_stride + w] = all ? (unsigned short)(x0 & 0xFFFFu) : (unsigned short)0xFFFFu;
}

/* block k = windows wB + 8k ..: its canonical entry (from the nearest merged block behind it) and tile count */
__global__ __launch_bounds__(256) void k_dsc_blocks(unsigned *st_all, size_t st_stride, const unsigned *tab8_all, size_t tab_stride,
                                                    const unsigned short *m8_all, size_t m8_stride, unsigned *eblk_all, unsigned *cblk_all, size_t blk_stride, int ext)
{
    const int f = blockIdx.y;
    unsigned *st = st_all + (size_t)f * st_stride;
    if (st[DSC_FAIL]) return;
    const unsigned K = st[DSC_K], wB = st[DSC_WB], k = blockIdx.x * 256 + threadIdx.x;
    if (k > K) return;
    const unsigned *tab8 = tab8_all + (size_t)f * tab_stride;
    const unsigned short *m8 = m8_all + (size_t)f * m8_stride;
    unsigned j = k;
    while (j > 0 && m8[wB + 8 * (j - 1)] == 0xFFFFu && k - j < 256) j--;
    if (j > 0 && m8[wB + 8 * (j - 1)] == 0xFFFFu) { st[DSC_FAIL] = 3; return; }       /* a long run of blocks that do not merge */
    unsigned e = j > 0 ? (unsigned)m8[wB + 8 * (j - 1)] : st[DSC_EB];
    for (unsigned i = j; i < k; i++) {
        const unsigned v = e < (unsigned)ext ? tab8[(size_t)(wB + 8 * i) * DEXT + e] : CMP_INV;
        if (v == CMP_INV) { st[DSC_FAIL] = 4; return; }
        e = v & 0xFFFFu;
    }
    unsigned cnt = 0;
    if (k < K) {
        const unsigned v = e < (unsigned)ext ? tab8[(size_t)(wB + 8 * k) * DEXT + e] : CMP_INV;
        if (v == CMP_INV) { st[DSC_FAIL] = 5; return; }
        cnt = v >> 16;
    } else if (e >= (unsigned)ext) { st[DSC_FAIL] = 6; return; }
    eblk_all[(size_t)f * blk_stride + k] = e; cblk_all[(size_t)f * blk_stride + k] = cnt;
}

/* exclusive prefix sum of the blocks' tile counts -> the eight-window entries of the canonical chain; the windows behind the last block one by one */
__global__ __launch_bounds__(1024) void k_dsc_scan(unsigned *st_all, size_t st_stride, const unsigned *tab_all, size_t tab_stride, const unsigned *eblk_all,
                                                   const unsigned *cblk_all, size_t blk_stride, unsigned *wentc_all, unsigned *wentc8_all, size_t went_stride, int nwin, int ext)
{
    __shared__ unsigned part[1024];
    const int f = blockIdx.x, tid = threadIdx.x;
    unsigned *st = st_all + (size_t)f * st_stride;
    if (st[DSC_FAIL]) return;
    const unsigned K = st[DSC_K], wB = st[DSC_WB];
    const unsigned *eblk = eblk_all + (size_t)f * blk_stride, *cblk = cblk_all + (size_t)f * blk_stride;
    unsigned *wentc = wentc_all + (size_t)f * went_stride, *wentc8 = wentc8_all + (size_t)f * went_stride;
    const unsigned per = (K + 1 + 1023) / 1024, k0 = tid * per, k1 = min(k0 + per, K + 1);
    unsigned s = 0;
    for (unsigned k = k0; k < k1; k++) s += cblk[k];
    part[tid] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) { const unsigned add = tid >= off ? part[tid - off] : 0u; __syncthreads(); part[tid] += add; __syncthreads(); }
    unsigned run = st[DSC_TB] + part[tid] - s;
    unsigned tK = 0;
    for (unsigned k = k0; k < k1; k++) {
        if (k < K) { wentc8[2 * (size_t)(wB + 8 * k)] = eblk[k]; wentc8[2 * (size_t)(wB + 8 * k) + 1] = run; }
        else tK = run;
        run += cblk[k];
    }
    if (k0 <= K && K < k1) {                                   /* the thread that holds block K: the tail */
        unsigned w = wB + 8 * K, e = eblk[K], t = tK;
        const unsigned *tab = tab_all + (size_t)f * tab_stride;
        while ((int)w < nwin) {
            wentc[2 * (size_t)w] = e; wentc[2 * (size_t)w + 1] = t;
            if (e >= (unsigned)ext) { st[DSC_FAIL] = 7; break; }
            const unsigned v = tab[(size_t)w * DEXT + e];
            t += v >> 16; e = v & 0xFFFFu; w++;
        }
        st[DSC_NCANON] = t;
    }
}

/* position of every canonical tile (interior point count everywhere).  A lane per window as before, but the 64 windows of a wave are STAGED IN LDS first (round 5:
 * coalesced reads, every word of the stream once; 81-word rows so that the lanes' reads fall on different banks) and walked from there -- the walk's ~ten dependent
 * reads per window were global ones, 132 us per 7200 x 3601 stream (a wave per window walking in scalar registers: 236 us, the CU's one scalar unit serialises them) */
#define EMIT_STRIDE (DWW + DTAIL + 1)
__global__ __launch_bounds__(64) void k_dsc_emit(const unsigned *z_all, size_t z_stride, size_t z_words, int ni, int nj, const unsigned *st_all, size_t st_stride,
                                                 const unsigned *wentc_all, size_t went_stride, unsigned *tpc_all, size_t tpc_stride, unsigned capc, int nwin, int plain)
{
    __shared__ unsigned sw[64 * EMIT_STRIDE];
    const int f = blockIdx.y, wbase = blockIdx.x * 64, lane = (int)threadIdx.x, w = wbase + lane;
    if (st_all[(size_t)f * st_stride + DSC_FAIL]) return;
    const unsigned *z = z_all + (size_t)f * z_stride;
    DecGeom g;
    if (!dec_geom(g, z, ni, nj, plain)) return;
    const size_t nwords = z_words - 1;
    /* (the window's entry and first tile index are asked for before the staging: in flight with it) */
    const unsigned e = w < nwin ? wentc_all[(size_t)f * went_stride + 2 * (size_t)w] : 0xFFFFFFFFu;
    unsigned t = w < nwin ? wentc_all[(size_t)f * went_stride + 2 * (size_t)w + 1] : 0u;
#pragma unroll 16
    for (int k = 0; k < 64; k++) sw[k * EMIT_STRIDE + lane] = gword(z + 1, (size_t)(wbase + k) * DWW + (size_t)lane, nwords);
#pragma unroll
    for (int q = 0; q < 64 * DTAIL / 64; q++) {
        const int k = q * (64 / DTAIL) + (lane / DTAIL), i = lane % DTAIL;
        sw[k * EMIT_STRIDE + DWW + i] = gword(z + 1, (size_t)(wbase + k + 1) * DWW + (size_t)i, nwords);
    }
    __syncthreads();
    if (e == 0xFFFFFFFFu) return;
    unsigned *tpc = tpc_all + (size_t)f * tpc_stride;
    const unsigned *my = sw + lane * EMIT_STRIDE;
    const unsigned base = (unsigned)w * DW;
    unsigned pos = e;
    while (pos < DW) {
        const unsigned long long v = bits64(my, pos);
        const unsigned k = dec_empty_run(g, v, pos, DW);
        if (k) {                                                     /* k empty tiles in a row: their positions without another look at the stream */
            for (unsigned q = 0; q < k; q++) if (t + q < capc) tpc[t + q] = base + pos + q * (unsigned)g.C;
            t += k; pos += k * (unsigned)g.C;
            continue;
        }
        if (t < capc) tpc[t] = base + pos;
        pos += (unsigned)dec_step(g, (unsigned)(v >> (64 - g.C)), g.n_int);
        t++;
    }
}

/* dtab[c]: if the odd tile that ends a row sat at canonical tile c, the tile-index shift behind it minus the shift in front of it.
 * A lane per canonical tile; the stream under the wave's 64 consecutive tiles -- from the first one's word to the end of the last one's window, at most
 * 64 x 416 bits + a window + the tail: DTAB_SPAN words -- is staged in LDS first (round 5): the header of the tile and the walk to the end of its window read it
 * there instead of through ~10 dependent global reads per lane (the waves of this kernel spent 84 % of their cycles waiting: SQ counters, r05_experiments.txt) */
#define DTAB_SPAN 1024
__global__ __launch_bounds__(256) void k_dsc_dtab(const unsigned *z_all, size_t z_stride, size_t z_words, int ni, int nj, const unsigned *st_all, size_t st_stride,
                                                  const unsigned *tab_all, size_t tab_stride, const unsigned *wentc_all, size_t went_stride,
                                                  const unsigned *tpc_all, size_t tpc_stride, short *dtab_all, size_t dtab_stride, unsigned capc, int nwin, int plain)
{
    __shared__ unsigned sp_all[4][DTAB_SPAN];
    const int f = blockIdx.y;
    const unsigned *st = st_all + (size_t)f * st_stride;
    if (st[DSC_FAIL]) return;
    const unsigned lim = min(st[DSC_NCANON], capc), lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const unsigned c = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x * 256 + wv * 64 >= lim) return;                   /* (the whole wave) */
    const unsigned *z = z_all + (size_t)f * z_stride;
    DecGeom g;
    if (!dec_geom(g, z, ni, nj, plain)) return;
    short *dtab = dtab_all + (size_t)f * dtab_stride;
    const bool act = c < lim;
    if (g.mlast == g.istep) { if (act) dtab[c] = 0; return; }
    const size_t nwords = z_words - 1;
    const unsigned *tab = tab_all + (size_t)f * tab_stride, *wentc = wentc_all + (size_t)f * went_stride;
    const unsigned pos = tpc_all[(size_t)f * tpc_stride + (act ? c : lim - 1)];      /* (canonical positions grow with the index: lane 0 holds the wave's first, lane 63 its last) */
    const unsigned pfirst = (unsigned)__builtin_amdgcn_readfirstlane((int)pos), plast = (unsigned)__builtin_amdgcn_readlane((int)pos, 63);
    const unsigned ws = pfirst >> 5, nsp = (plast / DW + 1) * DWW + DTAIL - ws;
    const bool staged = nsp <= DTAB_SPAN;                            /* (always, by the bound above; a stream that breaks it reads through the global path) */
    unsigned *sp = sp_all[wv];
    if (staged) {
#pragma unroll 8
        for (unsigned i = lane; i < nsp; i += 64) sp[i] = gword(z + 1, (size_t)ws + i, nwords);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (!act) return;
    const unsigned o = ws << 5;
#define DTAB_HDR(P) (staged ? getbits(sp, (P) - o, g.C) : getbits_g(z + 1, nwords, (P), g.C))
    unsigned w = pos / DW;
    unsigned p = pos + (unsigned)dec_step(g, DTAB_HDR(pos), g.mlast * g.istep);
    unsigned j = 0;
    const unsigned end = (w + 1) * DW;
    while (p < end) {
        if (staged) {
            const unsigned long long v = bits64(sp, p - o);
            const unsigned k = dec_empty_run(g, v, p, end);
            if (k) { p += k * (unsigned)g.C; j += k; continue; }
            p += (unsigned)dec_step(g, (unsigned)(v >> (64 - g.C)), g.n_int); j++;
        } else { p += (unsigned)dec_step(g, DTAB_HDR(p), g.n_int); j++; }
    }
#undef DTAB_HDR
    unsigned e = p - end;
    w++;
    int d = DSC_D_INVALID;
    /* the canonical entries of the next windows do not depend on the walk (only its offset e does): the first four asked for at once, the table lookups between
     * them are the dependent chain */
    const uint2 *wc2 = (const uint2 *)wentc;
    uint2 pre[4];
#pragma unroll
    for (int k = 0; k < 4; k++) pre[k] = (int)(w + k) < nwin ? wc2[(size_t)w + k] : make_uint2(0xFFFFFFFFu, 0u);
    bool done = false;
#pragma unroll
    for (int h = 0; h < 4; h++) {
        if (done || (int)w >= nwin) { done = true; continue; }
        if (pre[h].x == e) {
            const long long dd = (long long)(1u + j) - ((long long)pre[h].y - (long long)c);
            if (dd > -16000 && dd < 16000 && 1u + j + 2u < (unsigned)g.ntx) d = (int)dd;
            done = true; continue;
        }
        if (e >= (unsigned)g.ext) { done = true; continue; }
        const unsigned v = tab[(size_t)w * DEXT + e];
        j += v >> 16; e = v & 0xFFFFu; w++;
    }
    for (int h = 4; !done && h < DSC_LIST && (int)w < nwin; h++) {
        const uint2 q = wc2[(size_t)w];
        if (q.x == e) {
            const long long dd = (long long)(1u + j) - ((long long)q.y - (long long)c);
            if (dd > -16000 && dd < 16000 && 1u + j + 2u < (unsigned)g.ntx) d = (int)dd;
            break;
        }
        if (e >= (unsigned)g.ext) break;
        const unsigned v = tab[(size_t)w * DEXT + e];
        j += v >> 16; e = v & 0xFFFFu; w++;
    }
    dtab[c] = (short)d;
}

/* the recurrence over the rows of tiles: rowc[r] = canonical index of the odd tile that ends row r; c(r + 1) = c(r) + ntx - dtab[c(r)].  A dependent global
 * load per row costs ~0.4 us (1200 rows: 0.5 ms, measured; prefetching the neighbourhoods of the next rows through LDS did not beat it: the loads of a chunk
 * cost what they save).  Instead the table is composed with itself three times, for ALL indices in parallel (k_dsc_jump: J1[c] = the shift change over two
 * rows from c, J2 over four, J3 over eight; one gather pass each), the serial walk takes eight rows per step (k_dsc_rows: 150 dependent loads) and the
 * rows in between follow from the one-row table, every group of eight on its own lane (k_dsc_rowfill) */
#define DSC_J_INVALID 0x7FFFFFFF
template <class TIN>
__global__ __launch_bounds__(256) void k_dsc_jump(const unsigned *z_all, size_t z_stride, int ni, int nj, const unsigned *st_all, size_t st_stride,
                                                  const TIN *in_all, size_t in_stride, int *out_all, size_t out_stride, unsigned rows, unsigned capc, int plain)
{
    const int f = blockIdx.y;
    const unsigned *st = st_all + (size_t)f * st_stride;
    if (st[DSC_FAIL]) return;
    const unsigned lim = min(st[DSC_NCANON], capc), c = blockIdx.x * 256 + threadIdx.x;
    if (c >= lim) return;
    DecGeom g;
    if (!dec_geom(g, z_all + (size_t)f * z_stride, ni, nj, plain)) return;
    const TIN *in = in_all + (size_t)f * in_stride;
    const int inval = sizeof(TIN) == 2 ? DSC_D_INVALID : DSC_J_INVALID;
    int out = DSC_J_INVALID;
    const int a = (int)in[c];
    if (a != inval) {
        const long long c2 = (long long)c + (long long)rows * g.ntx - a;
        if (c2 >= 0 && c2 < (long long)lim) { const int b = (int)in[c2]; if (b != inval) out = a + b; }
    }
    out_all[(size_t)f * out_stride + c] = out;
}
/* A row whose stretch does NOT rejoin the canonical chain before the row's own odd tile (MINIMUM streams of 16-bit tokens: every tile length is a multiple of 5
 * but the raw tile's 404 bits, a misread header of 15 throws the canonical chain into another residue class and it takes several of them to come back: 1 - 6 % of
 * the rows of a 7201-point-wide field) is walked EXPLICITLY by the row recurrence: from the bit position behind the odd tile of the row before, `need` interior tiles
 * ahead through the composed maps of 8^l windows (k_dmin_up's tables: the largest aligned block whose tile count still fits, ~20 dependent lookups instead of the
 * ~150 windows of a row), then tile by tile inside the window where the row's odd tile starts.  pn = that tile's bit position; cn = its canonical index when the
 * chain is the canonical one again by that window (the recurrence goes back to its tables), 0xFFFFFFFF when not (the next row is walked the same way). */
/* (the whole wave runs this, every lane alike: the window under the walk is staged in LDS by all lanes -- one coalesced read instead of a dependent global read per tile
 * step -- and read back by all of them; sw_w = the window sw holds, -1 none) */
__device__ __forceinline__ void dsc_stage_window(unsigned *sw, int &sw_w, const unsigned *z1, size_t nwords, unsigned w)
{
    if (sw_w == (int)w) return;
    const unsigned lane = threadIdx.x & 63u;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();          /* (reads of the window before are done) */
    sw[lane] = gword(z1, (size_t)w * DWW + lane, nwords);
    if (lane < DTAIL) sw[DWW + lane] = gword(z1, (size_t)(w + 1) * DWW + lane, nwords);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    sw_w = (int)w;
}
__device__ __forceinline__ unsigned dsc_hdr(const DecGeom &g, const unsigned *sw, int sw_w, const unsigned *z1, size_t nwords, unsigned p)
{
    return (int)(p / DW) == sw_w ? getbits(sw, p - (unsigned)sw_w * DW, g.C) : getbits_g(z1, nwords, p, g.C);
}
__device__ bool dsc_walk_row(const DecGeom &g, const unsigned *z1, size_t nwords, const unsigned *tab, const unsigned *wentc, const uint2 *lev,
                             const unsigned *lev_nblk, const unsigned long long *lev_off, int nlev, int nwin, unsigned p, unsigned need, unsigned &pn, unsigned &cn,
                             unsigned *sw, int &sw_w)
{
    unsigned w = p / DW, tacc = 0;
    const unsigned end = (w + 1) * DW;
    dsc_stage_window(sw, sw_w, z1, nwords, w);
    while (p < end && tacc < need) {
        const unsigned long long v = bits64(sw, p - w * DW);
        unsigned k = dec_empty_run(g, v, p, end);                    /* (a run of empty tiles at once) */
        if (k) { k = k < need - tacc ? k : need - tacc; p += k * (unsigned)g.C; tacc += k; continue; }
        p += (unsigned)dec_step(g, (unsigned)(v >> (64 - g.C)), g.n_int); tacc++;
    }
    if (p < end) return false;                                       /* (a row inside one window: not a geometry this form takes) */
    unsigned e = p - end;
    w++;
    for (;;) {
        if ((int)w >= nwin || e >= (unsigned)g.ext) return false;
        /* candidates of every level aligned at w, all loaded at once (their addresses depend on (w, e) only); the largest that fits is taken */
        unsigned ex[DM_MAXLEV + 1], cnt[DM_MAXLEV + 1];
        bool ok[DM_MAXLEV + 1];
        { const unsigned v = tab[(size_t)w * DEXT + e]; ex[0] = v & 0xFFFFu; cnt[0] = v >> 16; ok[0] = true; }
#pragma unroll
        for (int l = 1; l <= DM_MAXLEV; l++) {
            const unsigned b = w >> (3 * l);
            ok[l] = l <= nlev && !(w & ((1u << (3 * l)) - 1u)) && b < lev_nblk[l] && ((unsigned long long)(b + 1) << (3 * l)) <= (unsigned long long)nwin;
            ex[l] = 0; cnt[l] = 0;
            if (ok[l]) { const uint2 v = lev[lev_off[l] + (size_t)b * DEXT + e]; ex[l] = v.x; cnt[l] = v.y; }
        }
        int take = -1;
        unsigned tex = 0, tcnt = 0;
#pragma unroll
        for (int l = 0; l <= DM_MAXLEV; l++)
            if (ok[l] && tacc + cnt[l] <= need) { take = l; tex = ex[l]; tcnt = cnt[l]; }
        if (take < 0) break;                                         /* the odd tile starts in window w */
        tacc += tcnt; e = tex; w += 1u << (3 * take);
    }
    dsc_stage_window(sw, sw_w, z1, nwords, w);
    unsigned q = e;                                                   /* (relative to the window) */
    for (unsigned k = tacc; k < need; ) {
        const unsigned long long v = bits64(sw, q);
        unsigned kk = dec_empty_run(g, v, q, DW);
        if (kk) { kk = kk < need - k ? kk : need - k; q += kk * (unsigned)g.C; k += kk; continue; }
        q += (unsigned)dec_step(g, (unsigned)(v >> (64 - g.C)), g.n_int); k++;
    }
    pn = w * DW + q;
    cn = wentc[2 * (size_t)w] == e ? wentc[2 * (size_t)w + 1] + (need - tacc) : 0xFFFFFFFFu;
    return true;
}
struct DscLev { const uint2 *lev; size_t lev_stride; int nlev; unsigned nblk[DM_MAXLEV + 1]; unsigned long long off[DM_MAXLEV + 1]; };      /* lev == nullptr: no composed maps (the merged-exit form) */
__global__ __launch_bounds__(64) void k_dsc_rows(const unsigned *z_all, size_t z_stride, int ni, int nj, unsigned *st_all, size_t st_stride,
                                                 const short *dtab_all, size_t dtab_stride, const int *j3_all, size_t j_stride,
                                                 unsigned *rowc_all, unsigned *head_all, size_t row_stride, unsigned capc, int plain,
                                                 size_t z_words = 0, const unsigned *tab_all = nullptr, const unsigned *wentc_all = nullptr, const unsigned *tpc_all = nullptr,
                                                 size_t w_stride = 0 /* of the three, in words */, unsigned *rowpn_all = nullptr, DscLev lv = DscLev(), int nwin = 0)
{
    __shared__ unsigned sw[DWW + DTAIL];                              /* the stream window under an explicit walk (dsc_walk_row) */
    int sw_w = -1;
    const int f = blockIdx.x;
    const bool l0 = threadIdx.x == 0;                                /* every lane runs the recurrence (the same loads, the same values); lane 0 stores */
    unsigned *st = st_all + (size_t)f * st_stride;
    if (st[DSC_FAIL]) return;
    const unsigned *z = z_all + (size_t)f * z_stride;
    DecGeom g;
    if (!dec_geom(g, z, ni, nj, plain)) return;
    const short *dtab = dtab_all + (size_t)f * dtab_stride;
    const int *j3 = j3_all + (size_t)f * j_stride;
    unsigned *rowc = rowc_all + (size_t)f * row_stride, *head = head_all + (size_t)f * row_stride;
    unsigned *rowpn = rowpn_all ? rowpn_all + (size_t)f * row_stride : nullptr;
    const unsigned rint = st[DSC_RINT], lim = min(st[DSC_NCANON], capc), ntx = (unsigned)g.ntx;
    const size_t nwords = z_words ? z_words - 1 : 0;
    unsigned c = ntx - 1, r = 0, pn = 0xFFFFFFFFu, nslow = 0, njump = 0, nstep = 0;        /* the odd tile of row r: canonical tile c, or (c == 0xFFFFFFFF) the tile at bit pn */
    /* Lane 0 alone walks the tables (a dependent load per step: with all 64 lanes in it a step took 0.7 us instead of 0.47); when a row has to be walked explicitly the
     * whole wave joins in -- it stages the stream windows of that walk -- and lane 0 goes on alone behind it.  what: 0 go on, 1 done, 2 an explicit walk is wanted, 3 failed */
    for (;;) {
        int what = 0;
        if (l0) {
            while (r < rint) {
                if (c != 0xFFFFFFFFu && c >= lim) { st[DSC_FAIL] = 8; what = 3; break; }
                rowc[r] = c; head[r] = 0;
                if (rowpn) rowpn[r] = pn;
                if (r + 1 >= rint) { what = 1; break; }
                if (c == 0xFFFFFFFFu) { what = 2; break; }
                const int j = r + 8 < rint ? j3[c] : DSC_J_INVALID;
                if (j != DSC_J_INVALID) { head[r] = 1; c = c + 8 * ntx - (unsigned)j; r += 8; njump++; continue; }       /* rows r + 1 .. r + 7: k_dsc_rowfill */
                const int d = dtab[c];
                if (d != DSC_D_INVALID) { c = c + ntx - (unsigned)d; r++; nstep++; continue; }
                if (!rowpn || !lv.lev) { st[DSC_FAIL] = 9; st[8] = r; st[9] = c; what = 3; break; }      /* (st[8], st[9]: where, for EZHIP_DEC_SCAN=2) */
                pn = tpc_all[(size_t)f * w_stride + c];
                what = 2; break;
            }
            if (what == 0) what = 1;                                 /* (r reached rint) */
        }
        what = __builtin_amdgcn_readfirstlane(what);
        if (what != 2) break;
        pn = (unsigned)__builtin_amdgcn_readfirstlane((int)pn); r = (unsigned)__builtin_amdgcn_readfirstlane((int)r);
        /* the explicit walk of row r + 1 from behind the odd tile at pn, the whole wave */
        const unsigned start = pn + (unsigned)dec_step(g, dsc_hdr(g, sw, sw_w, z + 1, nwords, pn), g.mlast * g.istep);
        unsigned cn;
        if (!dsc_walk_row(g, z + 1, nwords, tab_all + (size_t)f * w_stride, wentc_all + (size_t)f * w_stride, lv.lev + (size_t)f * lv.lev_stride, lv.nblk, lv.off, lv.nlev, nwin,
                          start, ntx - 1, pn, cn, sw, sw_w)) { if (l0) { st[DSC_FAIL] = 15; st[8] = r; } return; }
        c = cn;
        if (c != 0xFFFFFFFFu) pn = 0xFFFFFFFFu;
        r++; nslow++;
    }
    if (l0) { st[10] = nslow; st[11] = njump; st[12] = nstep; }      /* (EZHIP_DEC_SCAN=2: rows walked explicitly, eight-row jumps, single-row steps) */
}
__global__ __launch_bounds__(64) void k_dsc_rowfill(const unsigned *z_all, size_t z_stride, int ni, int nj, unsigned *st_all, size_t st_stride,
                                                    const short *dtab_all, size_t dtab_stride, unsigned *rowc_all, const unsigned *head_all, size_t row_stride, unsigned capc, int plain)
{
    const int f = blockIdx.y;
    unsigned *st = st_all + (size_t)f * st_stride;
    if (st[DSC_FAIL]) return;
    const unsigned r = blockIdx.x * 64 + threadIdx.x;                     /* (one lane per row; the rows at which the walk took an eight-row step have work) */
    if (r + 8 >= st[DSC_RINT]) return;
    DecGeom g;
    if (!dec_geom(g, z_all + (size_t)f * z_stride, ni, nj, plain)) return;
    if (!head_all[(size_t)f * row_stride + r]) return;
    const short *dtab = dtab_all + (size_t)f * dtab_stride;
    unsigned *rowc = rowc_all + (size_t)f * row_stride;
    unsigned c = rowc[r];
    const unsigned lim = min(st[DSC_NCANON], capc);
    for (int i = 1; i < 8; i++) {                                         /* (J3[c] valid: every step on the way is) */
        if (c >= lim) { st[DSC_FAIL] = 14; return; }
        c = c + (unsigned)g.ntx - (unsigned)(int)dtab[c]; rowc[r + i] = c;
    }
}

/* the stretch between the odd tile that ends row r and the window in which the real chain is canonical again.  One lane per row: tile by tile to the end of
 * the odd tile's window, then window by window through the speculation table until the entry is the canonical one -- the windows on the way are listed (window,
 * entry, first tile) and k_dsc_stretch2 writes their tiles, one lane per listed window.  rowmerge[r + 1] = the first tile that is canonical again */
__global__ __launch_bounds__(64) void k_dsc_stretch(const unsigned *z_all, size_t z_stride, size_t z_words, int ni, int nj, unsigned *st_all, size_t st_stride,
                                                    const unsigned *tab_all, size_t tab_stride, const unsigned *wentc_all, size_t went_stride,
                                                    const unsigned *tpc_all, size_t tpc_stride, const unsigned *rowc_all, unsigned *rowmerge_all, size_t row_stride,
                                                    unsigned *slist_all, unsigned *nlist_all, size_t list_stride,
                                                    unsigned *tilepos_all, size_t tp_stride, int nwin, int plain, const unsigned *rowpn_all = nullptr)
{
    const int f = blockIdx.y;
    unsigned *st = st_all + (size_t)f * st_stride;
    if (st[DSC_FAIL]) return;
    const unsigned r = blockIdx.x * 64 + threadIdx.x, rint = st[DSC_RINT];
    if (r >= rint) return;
    unsigned *rowmerge = rowmerge_all + (size_t)f * row_stride;
    unsigned *nlist = nlist_all + (size_t)f * row_stride;
    nlist[r] = 0;
    if (r == 0) rowmerge[0] = 0;
    if (r + 1 >= rint) return;
    const unsigned *z = z_all + (size_t)f * z_stride;
    DecGeom g;
    if (!dec_geom(g, z, ni, nj, plain)) return;
    const size_t nwords = z_words - 1;
    const unsigned *tpc = tpc_all + (size_t)f * tpc_stride, *tab = tab_all + (size_t)f * tab_stride, *wentc = wentc_all + (size_t)f * went_stride;
    unsigned *tilepos = tilepos_all + (size_t)f * tp_stride;
    unsigned *slist = slist_all + (size_t)f * list_stride + (size_t)r * DSC_LIST * 3;
    const unsigned ntx = (unsigned)g.ntx;
    const unsigned c = rowc_all[(size_t)f * row_stride + r], cn = rowc_all[(size_t)f * row_stride + r + 1];
    unsigned t = (r + 1) * ntx;                               /* first tile of row r + 1 */
    const unsigned tend = t + ntx - 1;                        /* the next odd tile: the chains have met before it, or the recurrence walked this row explicitly (cn == 0xFFFFFFFF) */
    const unsigned pos = c != 0xFFFFFFFFu ? tpc[c] : rowpn_all[(size_t)f * row_stride + r];
    unsigned p = pos + (unsigned)dec_step(g, getbits_g(z + 1, nwords, pos, g.C), g.mlast * g.istep);
    unsigned w = pos / DW;
    const unsigned end = (w + 1) * DW;
    while (p < end && t < tend) {
        const unsigned long long v = (unsigned long long)gword(z + 1, p >> 5, nwords) << 32 | gword(z + 1, (p >> 5) + 1, nwords);
        const unsigned long long vs = v << (p & 31u);
        unsigned k = dec_empty_run(g, vs, p, end);                   /* (a run of empty tiles: their positions without another read) */
        if (k) {
            k = k < tend - t ? k : tend - t;
            for (unsigned q = 0; q < k; q++) tilepos[t + q] = p + q * (unsigned)g.C;
            t += k; p += k * (unsigned)g.C;
            continue;
        }
        tilepos[t] = p; p += (unsigned)dec_step(g, (unsigned)(vs >> (64 - g.C)), g.n_int); t++;
    }
    if (p < end) { st[DSC_FAIL] = 10; return; }
    unsigned e = p - end, n = 0;
    w++;
    bool ok = false;
    for (int h = 0; h <= DSC_LIST && (int)w < nwin && t <= tend; h++) {
        if (wentc[2 * (size_t)w] == e) { ok = cn != 0xFFFFFFFFu && t - wentc[2 * (size_t)w + 1] == (r + 2) * ntx - 1 - cn; break; }      /* (the shift behind the stretch is the one the recurrence used) */
        if (n >= DSC_LIST || e >= (unsigned)g.ext) break;
        slist[3 * n] = w; slist[3 * n + 1] = e; slist[3 * n + 2] = t; n++;
        const unsigned v = tab[(size_t)w * DEXT + e];
        t += v >> 16; e = v & 0xFFFFu; w++;
    }
    if (cn == 0xFFFFFFFFu) {
        /* a row the recurrence walked explicitly: never canonical up to its odd tile -- every tile of it comes from the listed windows (k_dsc_stretch2 stops at tend) */
        if (ok || t <= tend) { st[DSC_FAIL] = 16; return; }          /* (met the canonical chain after all, or the list ran out before the row did) */
        nlist[r] = n;
        rowmerge[r + 1] = tend + 1;
        return;
    }
    if (!ok || t > tend) { st[DSC_FAIL] = 13; return; }
    nlist[r] = n;
    rowmerge[r + 1] = t;
}
__global__ __launch_bounds__(256) void k_dsc_stretch2(const unsigned *z_all, size_t z_stride, size_t z_words, int ni, int nj, const unsigned *st_all, size_t st_stride,
                                                      const unsigned *slist_all, const unsigned *nlist_all, size_t list_stride, size_t row_stride,
                                                      unsigned *tilepos_all, size_t tp_stride, int plain)
{
    const int f = blockIdx.y;
    const unsigned *st = st_all + (size_t)f * st_stride;
    if (st[DSC_FAIL]) return;
    const unsigned idx = blockIdx.x * 256 + threadIdx.x, r = idx / DSC_LIST, i = idx % DSC_LIST;
    if (r >= st[DSC_RINT] || i >= nlist_all[(size_t)f * row_stride + r]) return;
    const unsigned *z = z_all + (size_t)f * z_stride;
    DecGeom g;
    if (!dec_geom(g, z, ni, nj, plain)) return;
    const size_t nwords = z_words - 1;
    const unsigned *sl = slist_all + (size_t)f * list_stride + ((size_t)r * DSC_LIST + i) * 3;
    unsigned *tilepos = tilepos_all + (size_t)f * tp_stride;
    unsigned pos = sl[0] * DW + sl[1], t = sl[2];
    const unsigned end = (sl[0] + 1) * DW, tend = (r + 2) * (unsigned)g.ntx - 1;          /* (the odd tile of row r + 1: the last tile a stretch behind row r may hold) */
    while (pos < end && t <= tend) {
        const unsigned long long v = ((unsigned long long)gword(z + 1, pos >> 5, nwords) << 32 | gword(z + 1, (pos >> 5) + 1, nwords)) << (pos & 31u);
        unsigned k = dec_empty_run(g, v, pos, end);
        if (k) {
            k = k < tend + 1u - t ? k : tend + 1u - t;
            for (unsigned q = 0; q < k; q++) tilepos[t + q] = pos + q * (unsigned)g.C;
            t += k; pos += k * (unsigned)g.C;
            continue;
        }
        tilepos[t++] = pos; pos += (unsigned)dec_step(g, (unsigned)(v >> (64 - g.C)), g.n_int);
    }
}

/* every tile outside the stretches: its canonical twin, shifted */
__global__ __launch_bounds__(256) void k_dsc_final(const unsigned *z_all, size_t z_stride, int ni, int nj, const unsigned *st_all, size_t st_stride,
                                                   const unsigned *tpc_all, size_t tpc_stride, const unsigned *rowc_all, const unsigned *rowmerge_all, size_t row_stride,
                                                   unsigned *tilepos_all, size_t tp_stride, int plain)
{
    const int f = blockIdx.y;
    const unsigned *st = st_all + (size_t)f * st_stride;
    if (st[DSC_FAIL]) return;
    DecGeom g;
    if (!dec_geom(g, z_all + (size_t)f * z_stride, ni, nj, plain)) return;
    const unsigned ntx = (unsigned)g.ntx, t = blockIdx.x * 256 + threadIdx.x;
    if (t >= st[DSC_RINT] * ntx) return;
    const unsigned r = t / ntx;
    if (t < rowmerge_all[(size_t)f * row_stride + r]) return;
    const unsigned shift = (r + 1) * ntx - 1 - rowc_all[(size_t)f * row_stride + r];           /* true index minus canonical index in row r (mod 2^32) */
    tilepos_all[(size_t)f * tp_stride + t] = tpc_all[(size_t)f * tpc_stride + (t - shift)];       /* (t - shift < the canonical tiles kept: the stretch kernel met that index) */
}

/* the last row of tiles when its height differs: tile by tile behind the odd tile of the row before it; then the verdict.  The whole wave: the window under the walk
 * sits in LDS (one coalesced read per window instead of a dependent global read per tile: 1441 tiles of a 7201-point row took 0.41 ms), the next window's words are
 * on their way into registers while this one is walked; every lane walks alike, lane 0 stores */
__global__ __launch_bounds__(64) void k_dsc_lastrow(const unsigned *z_all, size_t z_stride, size_t z_words, int ni, int nj, unsigned *st_all, size_t st_stride,
                                                    const unsigned *tpc_all, size_t tpc_stride, const unsigned *rowc_all, size_t row_stride,
                                                    unsigned *tilepos_all, size_t tp_stride, int *status, int plain, const unsigned *rowpn_all = nullptr)
{
    __shared__ unsigned sw[DWW + DTAIL];
    const int f = blockIdx.x;
    const unsigned lane = threadIdx.x;
    const bool l0 = lane == 0;
    unsigned *st = st_all + (size_t)f * st_stride;
    if (st[DSC_FAIL]) return;
    const unsigned *z = z_all + (size_t)f * z_stride;
    DecGeom g;
    if (!dec_geom(g, z, ni, nj, plain)) return;
    const size_t nwords = z_words - 1;
    if (g.hlast != g.istep) {
        const unsigned rint = st[DSC_RINT], ntx = (unsigned)g.ntx;
        unsigned *tilepos = tilepos_all + (size_t)f * tp_stride;
        const unsigned cl = rowc_all[(size_t)f * row_stride + rint - 1];
        const unsigned pos = cl != 0xFFFFFFFFu ? tpc_all[(size_t)f * tpc_stride + cl] : rowpn_all[(size_t)f * row_stride + rint - 1];
        unsigned p = pos + (unsigned)dec_step(g, getbits_g(z + 1, nwords, pos, g.C), g.mlast * g.istep);
        unsigned t = rint * ntx, x = 0;
        unsigned w = p / DW;
        unsigned n0 = gword(z + 1, (size_t)w * DWW + lane, nwords), n1 = lane < DTAIL ? gword(z + 1, (size_t)(w + 1) * DWW + lane, nwords) : 0u;
        while (x < ntx) {
            /* window w into LDS (its words are in n0 / n1), window w + 1 asked for */
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
            sw[lane] = n0;
            if (lane < DTAIL) sw[DWW + lane] = n1;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            n0 = gword(z + 1, (size_t)(w + 1) * DWW + lane, nwords); n1 = lane < DTAIL ? gword(z + 1, (size_t)(w + 2) * DWW + lane, nwords) : 0u;
            const unsigned base = w * DW, end = base + DW;
            while (x < ntx && p < end) {
                if (l0) tilepos[t] = p;
                p += (unsigned)dec_step(g, getbits(sw, p - base, g.C), (x == ntx - 1 ? g.mlast : g.istep) * g.hlast);
                x++; t++;
            }
            if (x < ntx && p >= end + DW) {                       /* (cannot happen: a tile is shorter than a window) */
                w = p / DW;
                n0 = gword(z + 1, (size_t)w * DWW + lane, nwords); n1 = lane < DTAIL ? gword(z + 1, (size_t)(w + 1) * DWW + lane, nwords) : 0u;
            } else w++;
        }
        if ((unsigned long long)p > 32ull * nwords + 64) { if (l0) st[DSC_FAIL] = 11; return; }
    }
    if (l0) { st[DSC_OK] = 1; status[f] = 0; }
}

/* ============================================================================================================================================
 * Streams whose rows hold WHOLE tiles: the chain of tile headers by COMPOSITION (round 4).
 * When every tile of a row holds the same number of points -- MINIMUM (packTokensMinimum, c_zfstlib.c:469-588) with ni a multiple of 5, PARALLELOGRAM (:646-789)
 * and the planes of c_armn_compress32 (armn_compress_32.c:456-565, :642-750) with ni - 1 a multiple of 3 -- the chain up to the last row of tiles is a PURE
 * function of the bit position, and functions compose: the speculation table of a window IS its map entry -> (exit, tiles); the maps of 8, 64, 512 ... windows
 * follow level by level (k_dmin_up: one thread per (block, entry), eight lookups), the true entry of every block from the top down (k_dmin_top seeds the blocks
 * next to the start, k_dmin_down: one thread per block, eight lookups), down to the windows: went[w] = (entry, first tile) for ALL windows after ~6 + 6 short
 * launches instead of 10^5 dependent steps on one CU.  The last row of tiles (another height: another tile length) is walked tile by tile from LDS by one
 * workgroup (k_dmin_lastrow); k_armn_dec_emit then writes the position of every tile as before.
 * Who needs it: MINIMUM streams (a tile is 4 + nbits (+ 25 x nb) bits: walks from different entries never merge -- five residue classes mod 5 -- so the parallel
 * form above, one exit per eight-window block, gives up on them: 16 ms per 7200 x 3601 stream through the serial chain kernel, 1.1 ms here); the mantissa planes
 * of c_armn_compress32 (tiles of ~185 bits: no merged exit within eight windows either); PARALLELOGRAM streams with short rows or noisy data the parallel form
 * gives up on.  The form applies to what k_dsc_* left unresolved (dst[DSC_OK] == 0); ragged rows (a narrower last tile per row shifts everything behind it)
 * go to the composed ragged form (k_drg_*), streams of a few windows keep the serial chain kernel; dm[DM_NOTMINE] says which.
 * ============================================================================================================================================ */
enum { DM_NOTMINE = 0, DM_W0, DM_RAGGED, DM_WORDS = 8 };      /* DM_RAGGED: the canonical chain of the composed ragged form (k_drg_*) is wanted */
struct DmPlan { int nlev; unsigned nblk[DM_MAXLEV + 1]; unsigned long long off[DM_MAXLEV + 1]; unsigned long long eoff[DM_MAXLEV + 1]; };      /* level l: blocks of 8^l windows; level 0 = windows */
/* both methods and the planes of c_armn_compress32: what counts is that every tile of a row holds the same number of points */
__device__ __forceinline__ bool dmin_eligible(const DecGeom &g, int nwin) { return g.mlast == g.istep && g.nty >= 2 && g.ntx >= 2 && nwin >= 64; }
__global__ void k_dmin_init(const unsigned *z_all, size_t z_stride, int ni, int nj, unsigned *dm_all, size_t dm_stride, const unsigned *dst_all, size_t dst_stride, int nwin, int plain,
                            int ragged_min_ntx)
{
    const int f = blockIdx.x;
    if (threadIdx.x) return;
    unsigned *dm = dm_all + (size_t)f * dm_stride;
    DecGeom g;
    const bool geom = dec_geom(g, z_all + (size_t)f * z_stride, ni, nj, plain), todo = !(dst_all && dst_all[(size_t)f * dst_stride + DSC_OK]);
    const bool ok = geom && dmin_eligible(g, nwin) && (int)(g.body / DW) + 17 < nwin && todo;
    /* ragged rows of a PARALLELOGRAM stream or plane that the merged-exit form did not resolve: the same composition gives the CANONICAL chain (k_drg_*) */
    const bool rag = geom && todo && g.mlast != g.istep && dsc_eligible(g, nwin, ragged_min_ntx) && (int)(g.body / DW) + 17 < nwin;
    dm[DM_NOTMINE] = ok ? 0u : 1u; dm[DM_W0] = 0; dm[DM_RAGGED] = rag ? 1u : 0u;
}
/* level l + 1 from level l: T[l + 1][b][e] = the walk from entry e through blocks 8 b .. 8 b + 7 of level l: (exit, tiles).  Level 0 is the speculation table
 * (exit | tiles << 16 in one word), the levels above hold uint2 */
__global__ __launch_bounds__(256) void k_dmin_up(const unsigned *z_all, size_t z_stride, int ni, int nj, const unsigned *dm_all, size_t dm_stride,
                                                 const unsigned *tab_all, size_t tab_stride, uint2 *lev_all, size_t lev_stride, DmPlan pl, int l, int ext /* row stride of the tables */, int plain)
{
    const int f = blockIdx.y;
    if (dm_all[(size_t)f * dm_stride + DM_NOTMINE] && !dm_all[(size_t)f * dm_stride + DM_RAGGED]) return;
    const unsigned long long idx = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    const unsigned b = (unsigned)(idx / (unsigned)ext), e0 = (unsigned)(idx % (unsigned)ext);
    if (b >= pl.nblk[l + 1]) return;
    DecGeom g;
    if (!dec_geom(g, z_all + (size_t)f * z_stride, ni, nj, plain) || e0 >= (unsigned)g.ext) return;      /* (entries beyond the longest tile do not occur) */
    uint2 *lev = lev_all + (size_t)f * lev_stride;
    unsigned e = e0, cnt = 0;
    if (l == 0) {
        const unsigned *tab = tab_all + (size_t)f * tab_stride;
        for (unsigned c = 8 * b; c < 8 * b + 8 && c < pl.nblk[0]; c++) { const unsigned v = tab[(size_t)c * DEXT + e]; cnt += v >> 16; e = v & 0xFFFFu; }
    } else {
        const uint2 *in = lev + pl.off[l];
        for (unsigned c = 8 * b; c < 8 * b + 8 && c < pl.nblk[l]; c++) { const uint2 v = in[(size_t)c * (unsigned)ext + e]; cnt += v.y; e = v.x; }
    }
    lev[pl.off[l + 1] + (size_t)b * (unsigned)ext + e0] = make_uint2(e, cnt);
}
/* the seeds, one lane: the window in which the first tile starts is walked tile by tile (the chain does not enter it at one of its first `ext` bits: the
 * prefix of a PARALLELOGRAM stream ends anywhere); from the next window on, the entries of the windows up to the next multiple of 8, then of the level-1 blocks up
 * to the next multiple of 8 ... and of every block of the top level.  Everything to the right of a seed follows from its parent in k_dmin_down; blocks to the left
 * of the start stay invalid */
__global__ void k_dmin_top(const unsigned *z_all, size_t z_stride, size_t z_words, int ni, int nj, unsigned *dm_all, size_t dm_stride, const unsigned *tab_all, size_t tab_stride,
                           uint2 *lev_all, size_t lev_stride, unsigned *went_all, size_t went_stride, DmPlan pl, int ext, int plain, int ragged = 0 /* 1: the canonical chain of the ragged form, went_all = its array */)
{
    const int f = blockIdx.x;
    unsigned *dm = dm_all + (size_t)f * dm_stride;
    if (threadIdx.x || (ragged ? !dm[DM_RAGGED] : dm[DM_NOTMINE])) return;
    const unsigned *z = z_all + (size_t)f * z_stride;
    DecGeom g;
    if (!dec_geom(g, z, ni, nj, plain)) return;
    const size_t nwords = z_words - 1;
    const unsigned *tab = tab_all + (size_t)f * tab_stride;
    uint2 *lev = lev_all + (size_t)f * lev_stride;
    unsigned *went = went_all + (size_t)f * went_stride;
    unsigned pos = g.body, t = 0;
    const unsigned w0 = pos / DW, end0 = (w0 + 1) * DW;
    went[2 * (size_t)w0] = pos - w0 * DW; went[2 * (size_t)w0 + 1] = 0;
    dm[DM_W0] = w0;
    while (pos < end0) { pos += (unsigned)dec_step(g, getbits_g(z + 1, nwords, pos, g.C), g.n_int); t++; }      /* (a row of >= 2 tiles and >= 64 windows: the first run does not end here... */
    if (ragged ? (long long)t + 1 >= (long long)g.ntx : (long long)t >= (long long)g.ntx * (g.nty - 1)) { dm[ragged ? DM_RAGGED : DM_NOTMINE] = ragged ? 0u : 1u; return; }      /* ... unless the field is tiny: leave it to the serial kernel) */
    unsigned e = pos - end0, idx = w0 + 1;
    const int L = pl.nlev;
    for (int l = 0; l < L; l++) {
        while ((idx & 7u) && idx < pl.nblk[l]) {
            if (l == 0) { went[2 * (size_t)idx] = e; went[2 * (size_t)idx + 1] = t; const unsigned v = tab[(size_t)idx * DEXT + e]; t += v >> 16; e = v & 0xFFFFu; }
            else { lev[pl.eoff[l] + idx] = make_uint2(e, t); const uint2 v = lev[pl.off[l] + (size_t)idx * (unsigned)ext + e]; t += v.y; e = v.x; }
            idx++;
        }
        if (idx >= pl.nblk[l]) return;                               /* the stream's windows ended at this level */
        idx >>= 3;
    }
    for (; idx < pl.nblk[L]; idx++) {
        lev[pl.eoff[L] + idx] = make_uint2(e, t);
        const uint2 v = lev[pl.off[L] + (size_t)idx * (unsigned)ext + e];
        e = v.x; t += v.y;
    }
}
/* entries of level l from those of level l + 1; at level 0 they are the windows' (went) */
__global__ __launch_bounds__(256) void k_dmin_down(const unsigned *dm_all, size_t dm_stride, const unsigned *tab_all, size_t tab_stride, uint2 *lev_all, size_t lev_stride,
                                                   unsigned *went_all, size_t went_stride, DmPlan pl, int l, int ext, int ragged = 0)
{
    const int f = blockIdx.y;
    if (ragged ? !dm_all[(size_t)f * dm_stride + DM_RAGGED] : dm_all[(size_t)f * dm_stride + DM_NOTMINE]) return;
    const unsigned b = blockIdx.x * 256 + threadIdx.x;
    if (b >= pl.nblk[l + 1]) return;
    uint2 *lev = lev_all + (size_t)f * lev_stride;
    const uint2 en = lev[pl.eoff[l + 1] + b];
    if (en.x == 0xFFFFFFFFu) return;                                 /* left of the start, or seeded child by child */
    unsigned e = en.x, t = en.y;
    if (l == 0) {
        const unsigned *tab = tab_all + (size_t)f * tab_stride;
        unsigned *went = went_all + (size_t)f * went_stride;
        for (unsigned c = 8 * b; c < 8 * b + 8 && c < pl.nblk[0]; c++) {
            went[2 * (size_t)c] = e; went[2 * (size_t)c + 1] = t;
            const unsigned v = tab[(size_t)c * DEXT + e]; t += v >> 16; e = v & 0xFFFFu;
        }
    } else {
        const uint2 *in = lev + pl.off[l];
        for (unsigned c = 8 * b; c < 8 * b + 8 && c < pl.nblk[l]; c++) {
            lev[pl.eoff[l] + c] = make_uint2(e, t);
            const uint2 v = in[(size_t)c * (unsigned)ext + e]; t += v.y; e = v.x;
        }
    }
}
/* the last row of tiles (hlast != istep: another point count, another tile length) and the end of the chain.  One workgroup per field: finds the window in which
 * the regular chain reaches tile T = (nty - 1) ntx, walks to it, then walks the ntx tiles of the last row from LDS (chunks of DM_LR_WORDS words); every window
 * behind that one is taken out of went (the regular chain means nothing there).  status: -2 when the chain leaves the stream. */
#define DM_LR_WORDS 12288
__global__ __launch_bounds__(256) void k_dmin_lastrow(const unsigned *z_all, size_t z_stride, size_t z_words, int ni, int nj, const unsigned *dm_all, size_t dm_stride,
                                                      unsigned *went_all, size_t went_stride, unsigned *tilepos_all, size_t tp_stride, int nwin, int *status, unsigned *dst_all, size_t dst_stride, int plain)
{
    __shared__ unsigned lr[DM_LR_WORDS + 2];
    __shared__ unsigned s_pos, s_t, s_wstar, s_bad;
    const int f = blockIdx.x, tid = threadIdx.x;
    if (dm_all[(size_t)f * dm_stride + DM_NOTMINE]) return;
    const unsigned *z = z_all + (size_t)f * z_stride;
    DecGeom g;
    if (!dec_geom(g, z, ni, nj, plain)) return;
    const size_t nwords = z_words - 1;
    unsigned *went = went_all + (size_t)f * went_stride, *tilepos = tilepos_all + (size_t)f * tp_stride;
    const unsigned ntiles = (unsigned)g.ntiles;
    const unsigned Treg = g.hlast != g.istep ? (unsigned)(g.nty - 1) * (unsigned)g.ntx : ntiles;       /* tiles of the regular chain */
    if (tid == 0) {
        /* the last window whose first tile is <= Treg (first tiles are non-decreasing) */
        int lo = (int)dm_all[(size_t)f * dm_stride + DM_W0], hi = nwin - 1;      /* (the windows in front of the first tile hold no entry) */
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (went[2 * (size_t)mid + 1] <= Treg) lo = mid; else hi = mid - 1; }
        unsigned pos = (unsigned)lo * DW + went[2 * (size_t)lo], t = went[2 * (size_t)lo + 1];
        unsigned bad = 0;
        while (t < Treg) {                                            /* <= the tiles of one window */
            if ((size_t)(pos >> 5) >= nwords) { bad = 1; break; }
            pos += (unsigned)dec_step(g, getbits_g(z + 1, nwords, pos, g.C), g.n_int); t++;
        }
        s_wstar = (unsigned)lo; s_pos = pos; s_t = t; s_bad = bad;
    }
    __syncthreads();
    for (int w = (int)s_wstar + 1 + tid; w < nwin; w += 256) went[2 * (size_t)w] = 0xFFFFFFFFu;
    if (g.hlast != g.istep && !s_bad) {
        const int nlast = g.istep * g.hlast;
        while (s_t < ntiles && !s_bad) {
            const unsigned base_w = s_pos >> 5;                       /* chunk of the stream in LDS */
            __syncthreads();
            for (int k = tid; k < DM_LR_WORDS + 2; k += 256) lr[k] = gword(z + 1, (size_t)base_w + k, nwords);
            __syncthreads();
            if (tid == 0) {
                unsigned pos = s_pos, t = s_t;
                const unsigned lim = (base_w + DM_LR_WORDS) * 32u - (unsigned)g.ext - 32u;      /* a whole tile (and its header's two words) inside the chunk */
                while (t < ntiles && pos < lim) {
                    tilepos[t] = pos;
                    pos += (unsigned)dec_step(g, getbits(lr, pos - base_w * 32u, g.C), nlast); t++;
                }
                if (pos == s_pos) s_bad = 1;                          /* (cannot happen: ext + 32 bits < the chunk) */
                s_pos = pos; s_t = t;
            }
            __syncthreads();
        }
    }
    if (tid == 0) {
        /* the chain must end inside the stream: the terminator (two 16-bit zero tokens) follows the last tile */
        const bool broken = s_bad || (size_t)(s_pos >> 5) > nwords;
        if (status) status[f] = broken ? -2 : 0;
        if (!broken && dst_all) dst_all[(size_t)f * dst_stride + DSC_OK] = 1u;      /* the serial chain kernel and its followers skip this field */
    }
}

/* ============================================================================================================================================
 * RAGGED rows by composition (round 4, k_drg_*): PARALLELOGRAM streams and planes of c_armn_compress32 whose rows end on a narrower tile ((ni - 1) % 3 != 0), since
 * round 5 MINIMUM streams as well (ni % 5 != 0; rows whose stretch outlasts them are walked explicitly: dsc_walk_row above k_dsc_rows),
 * that the merged-exit form gave up on (mantissa planes: tiles of ~185 bits do not merge within eight windows; short rows; noisy data).  That form consists of two
 * halves: the CANONICAL chain (the walk from the first tile that never meets a row end), which it gets from merged exits of eight-window blocks, and the row
 * recurrence on top of it (k_dsc_emit .. k_dsc_lastrow: the shift of the tile index per row end, the stretches between a row end and the rejoining point).  Only
 * the first half needs merging -- and the composition of the windows' maps (k_dmin_up / top / down with the interior tile rule) yields the canonical entries of
 * ALL windows without it.  So: canonical entries by composition into wentc, the count of canonical tiles (k_drg_ncanon), then the second half unchanged, on a
 * state of its own; k_drg_done hands a resolved field to the kernels behind (dst[DSC_OK]).  A stretch rejoins the canonical chain after 16 windows on average in a
 * mantissa plane (130 seen: DSC_LIST).
 * ============================================================================================================================================ */
__global__ void k_drg_ncanon(unsigned *st_all, size_t st_stride, const unsigned *dm_all, size_t dm_stride, const unsigned *tab_all, size_t tab_stride,
                             const unsigned *wentc_all, size_t went_stride, int nwin, int ext)
{
    const int f = blockIdx.x;
    if (threadIdx.x) return;
    unsigned *st = st_all + (size_t)f * st_stride;
    if (st[DSC_FAIL]) return;
    if (!dm_all[(size_t)f * dm_stride + DM_RAGGED]) { st[DSC_FAIL] = 101; return; }
    const unsigned *wentc = wentc_all + (size_t)f * went_stride;
    const unsigned e = wentc[2 * (size_t)(nwin - 1)], t = wentc[2 * (size_t)(nwin - 1) + 1];
    if (e >= (unsigned)ext) { st[DSC_FAIL] = 102; return; }
    st[DSC_NCANON] = t + (tab_all[(size_t)f * tab_stride + (size_t)(nwin - 1) * DEXT + e] >> 16);
}
__global__ void k_drg_done(unsigned *dst_all, const unsigned *st2_all, size_t st_stride)
{
    const int f = blockIdx.x;
    if (threadIdx.x == 0 && st2_all[(size_t)f * st_stride + DSC_OK]) dst_all[(size_t)f * st_stride + DSC_OK] = 1u;
}
/* the tile walk of a plane without the serial chain kernel behind it (the caller has a faster fallback: the host's walk): 0 stays 0, -2 stays, anything else 1 */
__global__ void k_drg_verdict(const unsigned *dst_all, size_t st_stride, int *status)
{
    const int f = blockIdx.x;
    if (threadIdx.x == 0 && !dst_all[(size_t)f * st_stride + DSC_OK] && status[f] != -2) status[f] = 1;
}

__global__ __launch_bounds__(256) void k_armn_dec_tiles(const unsigned *z_all, size_t z_stride, size_t z_words, int ni, int nj,
                                                        const unsigned *tilepos_all, size_t tp_stride,
                                                        unsigned short *d16_all, size_t d_stride,           /* PARALLELOGRAM differences */
                                                        unsigned short *out_all, size_t out_stride, int swap, /* MINIMUM: final tokens (half-words) */
                                                        long long max_tiles)
{
    const int f = blockIdx.y;
    const unsigned *z = z_all + (size_t)f * z_stride;
    DecGeom g;
    if (!dec_geom(g, z, ni, nj)) return;
    const unsigned *z1 = z + 1;
    const size_t nwords = z_words - 1;
    unsigned short *d16 = d16_all + (size_t)f * d_stride, *out = out_all + (size_t)f * out_stride;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (g.method == 4 && t < ni + nj - 1) {
        /* row 1 and column 1 as differences (:808-816): the 2-D prefix sum then reproduces them */
        const int k = (int)t;
        const unsigned u = getbits_g(z1, nwords, 3u + (unsigned)k * g.nbits, g.nbits);
        if (k == 0) d16[0] = (unsigned short)u;
        else if (k < ni) d16[k] = (unsigned short)(u - getbits_g(z1, nwords, 3u + (unsigned)(k - 1) * g.nbits, g.nbits));
        else {
            const int j = k - ni + 1;
            const unsigned prev = getbits_g(z1, nwords, 3u + (unsigned)(j == 1 ? 0 : k - 1) * g.nbits, g.nbits);
            d16[(size_t)ni * j] = (unsigned short)(u - prev);
        }
    }
    if (t >= g.ntiles || t >= max_tiles) return;
    const int ty = (int)(t / g.ntx), tx = (int)(t % g.ntx);
    const int i0 = g.origin + tx * g.istep, j0 = g.origin + ty * g.istep;
    const int m = min(g.istep, ni - i0), nn = min(g.istep, nj - j0);
    unsigned pos = tilepos_all[(size_t)f * tp_stride + t];
    const unsigned hdr = getbits_g(z1, nwords, pos, g.C);
    pos += g.C;
    if (g.method == 4) {
        const int w = hdr == 0 ? 0 : ((hdr == 15 || hdr == 16) ? 17 : (int)hdr + 1);
        for (int b = 0; b < nn; b++)
            for (int a = 0; a < m; a++) {
                int v = 0;
                if (w) { const unsigned tok = getbits_g(z1, nwords, pos, w); pos += w; v = (int)(tok << (32 - w)) >> (32 - w); }
                d16[(size_t)ni * (j0 + b) + i0 + a] = (unsigned short)v;
            }
    } else {
        unsigned lo = 0;
        if (hdr < 15) { lo = getbits_g(z1, nwords, pos, g.nbits); pos += g.nbits; }
        const int w = hdr == 0 ? 0 : (hdr >= 15 ? 16 : (int)hdr);
        for (int b = 0; b < nn; b++)
            for (int a = 0; a < m; a++) {
                unsigned v = lo;
                if (w) { v += getbits_g(z1, nwords, pos, w); pos += w; }
                out[out_slot((size_t)ni * (j0 + b) + i0 + a, swap)] = (unsigned short)v;
            }
    }
}

/* inclusive scan of every row, in place, mod 2^16.  One block per row, 2048-element segments through LDS. */
__global__ __launch_bounds__(256) void k_armn_dec_rows(const unsigned *z_all, size_t z_stride, unsigned short *d16_all, size_t d_stride, int ni, int nj)
{
    const int f = blockIdx.y;
    if ((z_all[(size_t)f * z_stride] & 15) != 4) return;
    unsigned short *row = d16_all + (size_t)f * d_stride + (size_t)ni * blockIdx.x;
    __shared__ unsigned short seg[2048];
    __shared__ unsigned wtot[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    unsigned carry = 0;
    for (int s0 = 0; s0 < ni; s0 += 2048) {
        const int len = min(2048, ni - s0);
        for (int k = tid; k < len; k += 256) seg[k] = row[s0 + k];
        __syncthreads();
        unsigned v[8], sum = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) { const int idx = tid * 8 + k; sum += (idx < len) ? seg[idx] : 0u; v[k] = sum; }
        unsigned incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const unsigned o = __shfl_up(incl, d); if (lane >= d) incl += o; }
        if (lane == 63) wtot[wv] = incl;
        __syncthreads();
        unsigned off = carry + incl - sum, tot = 0;
        for (int k = 0; k < 4; k++) { if (k < wv) off += wtot[k]; tot += wtot[k]; }
#pragma unroll
        for (int k = 0; k < 8; k++) { const int idx = tid * 8 + k; if (idx < len) seg[idx] = (unsigned short)(off + v[k]); }
        carry += tot;
        __syncthreads();
        for (int k = tid; k < len; k += 256) row[s0 + k] = seg[k];
        __syncthreads();
    }
}

/* column totals of each band of `band` rows -> part[band index][ni] */
__global__ __launch_bounds__(256) void k_armn_dec_colsum(const unsigned *z_all, size_t z_stride, const unsigned short *d16_all, size_t d_stride,
                                                         unsigned short *part_all, size_t part_stride, int ni, int nj, int band)
{
    const int f = blockIdx.z;
    if ((z_all[(size_t)f * z_stride] & 15) != 4) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= ni) return;
    const unsigned short *d = d16_all + (size_t)f * d_stride;
    const int j0 = blockIdx.y * band, j1 = min(nj, j0 + band);
    unsigned s = 0;
    for (int j = j0; j < j1; j++) s += d[(size_t)ni * j + i];
    part_all[(size_t)f * part_stride + (size_t)blockIdx.y * ni + i] = (unsigned short)s;
}

/* running column sums inside each band, offset by the bands above: the final tokens */
__global__ __launch_bounds__(256) void k_armn_dec_cols(const unsigned *z_all, size_t z_stride, const unsigned short *d16_all, size_t d_stride,
                                                       const unsigned short *part_all, size_t part_stride,
                                                       unsigned short *out_all, size_t out_stride, int swap, int ni, int nj, int band)
{
    const int f = blockIdx.z;
    if ((z_all[(size_t)f * z_stride] & 15) != 4) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= ni) return;
    const unsigned short *d = d16_all + (size_t)f * d_stride, *part = part_all + (size_t)f * part_stride;
    unsigned short *out = out_all + (size_t)f * out_stride;
    unsigned s = 0;
    for (int b = 0; b < (int)blockIdx.y; b++) s += part[(size_t)b * ni + i];
    const int j0 = blockIdx.y * band, j1 = min(nj, j0 + band);
    for (int j = j0; j < j1; j++) {
        const size_t k = (size_t)ni * j + i;
        s += d[k];
        out[out_slot(k, swap)] = (unsigned short)s;
    }
}

#define DEC_BANDS 32
static size_t dec_max_tiles(int ni, int nj) { return ((size_t)(ni + 2) / 3 + 1) * ((size_t)(nj + 2) / 3 + 1); }   /* tile step >= 3 */
static size_t dec_nwin(size_t z_words) { return (z_words * 32 + DW - 1) / DW + 1; }
static size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
/* levels of the composition: nblk[0] = windows, nblk[l + 1] = ceil(nblk[l] / 8) down to <= 8 blocks; tables (uint2 [nblk][DEXT]) from level 1, entry arrays beside them */
static void dmin_plan(DmPlan *pl, size_t nwin)
{
    memset(pl, 0, sizeof(*pl));
    pl->nblk[0] = (unsigned)nwin;
    int l = 0;
    while (pl->nblk[l] > 8 && l < DM_MAXLEV) { pl->nblk[l + 1] = (pl->nblk[l] + 7) / 8; l++; }
    pl->nlev = l;
    unsigned long long o = 0;
    for (int k = 1; k <= l; k++) { pl->off[k] = o; o += (unsigned long long)pl->nblk[k] * DEXT; }
    for (int k = 1; k <= l; k++) { pl->eoff[k] = o; o += ((unsigned long long)pl->nblk[k] + 31) & ~31ull; }
    pl->off[0] = o;                                             /* (total, in uint2) */
}
static size_t dmin_bytes(size_t nwin) { DmPlan pl; dmin_plan(&pl, nwin); return al256(8 * (size_t)pl.off[0]) + 256; }
extern "C" size_t packhip_armn_dec_work_bytes(int ni, int nj, size_t z_words)
{
    const size_t n = (size_t)ni * nj, nwin = dec_nwin(z_words);
    const size_t capc = dec_max_tiles(ni, nj) + dsc_margin(dec_max_tiles(ni, nj)), nty = (size_t)(nj + 2) / 3 + 2;
    return al256(4 * dec_max_tiles(ni, nj)) + al256(2 * (n + 8)) + al256(2 * (size_t)DEC_BANDS * ni) + 2 * al256(4 * nwin * DEXT) + 2 * al256(8 * nwin) + 256
           /* the parallel form: canonical tile positions, shift changes, merged exits, block entries / counts, canonical window entries, rows, state */
           + al256(4 * capc) + al256(2 * capc) + al256(2 * nwin) + 2 * al256(4 * (nwin / 8 + 2)) + 2 * al256(8 * nwin) + 5 * al256(4 * nty) + al256(12 * DSC_LIST * nty) + 2 * al256(4 * capc) + 256
           /* MINIMUM streams by composition: the tables of 8, 64, ... windows and the blocks' entries (k_dmin_*) */
           + dmin_bytes(nwin);
}

/* d_out: (1 + ni*nj/2) words per field, zero-filled first (the odd trailing half-word).  d_status: one int per field
 * (0 ok, -1 unknown header, -2 broken chain).  Everything is asynchronous on the current stream. */
extern "C" int packhip_armn_decode_phase(unsigned *d_out, size_t out_stride_words, const unsigned *d_z, size_t z_stride_words, size_t z_words,
                                         int ni, int nj, int nfields, int swap, void *d_work, size_t work_stride_bytes, int *d_status, int phase);
extern "C" int packhip_armn_decode(unsigned *d_out, size_t out_stride_words, const unsigned *d_z, size_t z_stride_words, size_t z_words,
                                   int ni, int nj, int nfields, int swap, void *d_work, size_t work_stride_bytes, int *d_status)
{
    return packhip_armn_decode_phase(d_out, out_stride_words, d_z, z_stride_words, z_words, ni, nj, nfields, swap, d_work, work_stride_bytes, d_status, 0);
}
/* phase 0: everything; 1: the kernels that run on all CUs in front of the chain kernel (speculation, composed table); 2: the chain kernel and what follows.
 * A caller with a batch runs phase 1 of one half, then ALL of the other half on a second stream, then phase 2 of the first half: the chain kernels (one CU
 * per stream) of the first half run beside the all-CU kernels of the second.
 * walk_only: stop behind k_armn_dec_emit (the bit position of every tile header, 32-bit, at the start of the work area); plain: the tile rule of a plane of
 * c_armn_compress32 */
static int dec_launch(unsigned *d_out, size_t out_stride_words, const unsigned *d_z, size_t z_stride_words, size_t z_words,
                      int ni, int nj, int nfields, int swap, void *d_work, size_t work_stride_bytes, int *d_status, int phase, int plain, int walk_only, int no_serial = 0);
extern "C" int packhip_armn_decode_phase(unsigned *d_out, size_t out_stride_words, const unsigned *d_z, size_t z_stride_words, size_t z_words,
                                         int ni, int nj, int nfields, int swap, void *d_work, size_t work_stride_bytes, int *d_status, int phase)
{
    return dec_launch(d_out, out_stride_words, d_z, z_stride_words, z_words, ni, nj, nfields, swap, d_work, work_stride_bytes, d_status, phase, 0, 0);
}
/* The chain of tile headers of ONE parallelogram plane of c_armn_compress32 (armn_compress_32.c:568-639, :753-824: the decoders read it tile after tile) on
 * the device.  d_z[0]: a header word in armn_compress's layout (method 4, step 3, the plane's nbits: packhip_armn_plane_header), d_z[1..]: the plane's
 * stream, z_words in all, followed by >= DTAIL + 2 readable words.  The positions (bits from the plane's first word, 32-bit) of the (ni+1)/3 x (nj+1)/3 tiles
 * land at the START of d_work (packhip_armn_dec_work_bytes(ni, nj, z_words) bytes); *d_status: 0, -1 (header), -2 (the chain leaves the stream). */
extern "C" unsigned packhip_armn_plane_header(int nbits) { return 4u | 1u << 4 | 3u << 7 | ((unsigned)nbits & 31u) << 10; }
extern "C" int packhip_armn_tile_walk(const unsigned *d_z, size_t z_words, int ni, int nj, void *d_work, int *d_status)
{
    return dec_launch(NULL, 0, d_z, 0, z_words, ni, nj, 1, 0, d_work, 0, d_status, 0, 1, 1);
}
/* the same without the serial chain kernel behind the parallel forms: *d_status = 1 when none of them resolved the chain (the caller walks it on the host) */
extern "C" int packhip_armn_tile_walk_parallel(const unsigned *d_z, size_t z_words, int ni, int nj, void *d_work, int *d_status)
{
    return dec_launch(NULL, 0, d_z, 0, z_words, ni, nj, 1, 0, d_work, 0, d_status, 0, 1, 1, 1);
}
/* one region per field set to a byte value, all fields in ONE launch (a batch of 32 streams issued ~400 hipMemsetAsync calls per decode: ~10 us of host time each, more
 * than the fills themselves) */
__global__ __launch_bounds__(256) void k_dec_fill(unsigned *base, size_t stride_words, size_t nwords, unsigned v)
{
    unsigned *p = base + (size_t)blockIdx.y * stride_words;
    const size_t head = (4u - (unsigned)(((size_t)p >> 2) & 3u)) & 3u;            /* words in front of the first 16-byte boundary */
    const size_t h = head < nwords ? head : nwords, nq = (nwords - h) / 4, tail = h + 4 * nq;
    uint4 *q = (uint4 *)(p + h);
    const uint4 vv = make_uint4(v, v, v, v);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nq; i += (size_t)gridDim.x * 256) q[i] = vv;
    if (blockIdx.x == 0) {
        if (threadIdx.x < h) p[threadIdx.x] = v;
        if (tail + threadIdx.x < nwords) p[tail + threadIdx.x] = v;              /* (at most 3 words) */
    }
}
static int dec_fill(hipStream_t st, void *base, size_t stride_bytes, int nfields, size_t nbytes, int byte)
{
    if (!nbytes || nfields <= 0) return 0;
    if (nfields == 1 || (nbytes & 3) || (stride_bytes & 3) || ((size_t)base & 3))         /* (one field: the runtime's fill is one call as well) */
    {
        for (int f = 0; f < nfields; f++)
            if (hipMemsetAsync((char *)base + (size_t)f * stride_bytes, byte, nbytes, st) != hipSuccess) return -1;
        return 0;
    }
    const unsigned v = 0x01010101u * (unsigned)(byte & 0xFF);
    const size_t nq = nbytes / 16;
    unsigned bx = (unsigned)((nq + 256 * 8 - 1) / (256 * 8));
    if (bx < 1) bx = 1;
    if (bx > 2048) bx = 2048;
    hipLaunchKernelGGL(k_dec_fill, dim3(bx, (unsigned)nfields), dim3(256), 0, st, (unsigned *)base, stride_bytes / 4, nbytes / 4, v);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
static int dec_launch(unsigned *d_out, size_t out_stride_words, const unsigned *d_z, size_t z_stride_words, size_t z_words,
                      int ni, int nj, int nfields, int swap, void *d_work, size_t work_stride_bytes, int *d_status, int phase, int plain, int walk_only, int no_serial)
{
    static int attr_done = 0;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void *)k_armn_dec_hop, hipFuncAttributeMaxDynamicSharedMemorySize, HOP_LDS) != hipSuccess) return -1;
        attr_done = 1;
    }
    hipStream_t st = STREAM;
    const size_t n = (size_t)ni * nj;
    if ((unsigned long long)z_words * 32 + 4 * DW >= (1ull << 32)) { fprintf(stderr, "<armn_compress> UNCOMPRESS: stream too large for 32-bit bit positions\n"); return -1; }
    const long long max_tiles = (long long)dec_max_tiles(ni, nj);
    if (max_tiles >= (1ll << 31)) { fprintf(stderr, "<armn_compress> UNCOMPRESS: field too large\n"); return -1; }
    const int nwin = (int)dec_nwin(z_words);
    char *wk = (char *)d_work;
    unsigned *tilepos = (unsigned *)wk;                     wk += al256(4 * (size_t)max_tiles);
    unsigned short *d16 = (unsigned short *)wk;             wk += al256(2 * (n + 8));
    unsigned short *part = (unsigned short *)wk;            wk += al256(2 * (size_t)DEC_BANDS * ni);
    unsigned *tab = (unsigned *)wk;                         wk += al256(4 * (size_t)nwin * DEXT);
    unsigned *tab8 = (unsigned *)wk;                        wk += al256(4 * (size_t)nwin * DEXT);
    unsigned *went = (unsigned *)wk;                        wk += al256(8 * (size_t)nwin);
    unsigned *went8 = (unsigned *)wk;                       wk += al256(8 * (size_t)nwin) + 256;
    const size_t capc = (size_t)max_tiles + dsc_margin((size_t)max_tiles), ntyc = (size_t)(nj + 2) / 3 + 2, nblk = (size_t)nwin / 8 + 2;
    unsigned *tpc = (unsigned *)wk;                         wk += al256(4 * capc);
    short *dtab = (short *)wk;                              wk += al256(2 * capc);
    unsigned short *m8 = (unsigned short *)wk;              wk += al256(2 * (size_t)nwin);
    unsigned *eblk = (unsigned *)wk;                        wk += al256(4 * nblk);
    unsigned *cblk = (unsigned *)wk;                        wk += al256(4 * nblk);
    unsigned *wentc = (unsigned *)wk;                       wk += al256(8 * (size_t)nwin);
    unsigned *wentc8 = (unsigned *)wk;                      wk += al256(8 * (size_t)nwin);
    unsigned *rowc = (unsigned *)wk;                        wk += al256(4 * ntyc);
    unsigned *rowmerge = (unsigned *)wk;                    wk += al256(4 * ntyc);
    unsigned *nlist = (unsigned *)wk;                       wk += al256(4 * ntyc);
    unsigned *slist = (unsigned *)wk;                       wk += al256(12 * DSC_LIST * ntyc);
    unsigned *rowhead = (unsigned *)wk;                     wk += al256(4 * ntyc);
    unsigned *rowpn = (unsigned *)wk;                       wk += al256(4 * ntyc);      /* bit position of a row's odd tile where it is not a canonical tile (k_dsc_rows' explicit walk) */
    int *jA = (int *)wk;                                    wk += al256(4 * capc);
    int *jB = (int *)wk;                                    wk += al256(4 * capc);
    unsigned *dst = (unsigned *)wk;                         wk += 128;
    unsigned *dmst = (unsigned *)wk;                        wk += 64;
    unsigned *dst2 = (unsigned *)wk;                        wk += 64;       /* (DSC_WORDS = 16 words; dmst: DM_WORDS = 8) */
    uint2 *dmlev = (uint2 *)wk;
    const char *scan_env = getenv("EZHIP_DEC_SCAN");
    const int scan = scan_env ? atoi(scan_env) : 1;          /* the chain between row ends in parallel (k_dsc_*, the default since the end of round 3); the serial chain
                                                              * kernel runs only for the fields that form gives up on (small fields, streams whose windows do not merge);
                                                              * EZHIP_DEC_SCAN=0: the serial kernel for everything, 2: print every field's state words */
    const size_t ws4 = work_stride_bytes / 4, ws2 = work_stride_bytes / 2;
    /* MINIMUM streams only?  Their walks do not merge within eight windows, so the merged-exit form (k_dsc_m8 .. k_dsc_scan) and the eight-window tables it and the
     * serial chain kernel read (k_armn_dec_compose_lds) only cost time: 0.45 ms of a 1.2 ms decode.  The headers are read first (one small copy, one wait); such a batch
     * goes straight to the composed forms, and only if those leave a field open are the tables made for the serial kernel.  EZHIP_DEC_MIN_FIRST_FORM=1: as before */
    int all_min = 0;
    if (phase == 0 && !plain && !walk_only && scan && nfields <= 256 && !getenv("EZHIP_DEC_ASYNC") && !getenv("EZHIP_DEC_NO_DMIN") && !getenv("EZHIP_DEC_MIN_FIRST_FORM")) {
        unsigned hm[256];
        hipError_t ec = nfields == 1 ? hipMemcpyAsync(hm, d_z, 4, hipMemcpyDeviceToHost, st)
                                     : hipMemcpy2DAsync(hm, 4, d_z, 4 * z_stride_words, 4, (size_t)nfields, hipMemcpyDeviceToHost, st);
        if (ec != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { fprintf(stderr, "<armn_compress> UNCOMPRESS: reading the stream headers failed\n"); return -1; }
        all_min = 1;
        for (int f = 0; f < nfields; f++) if ((hm[f] & 15u) != 3u) all_min = 0;
    }
    auto launch_compose = [&]() {
        const int G = 28, lds_words = (3 * G + 17) * 160;          /* 64 640 B: G windows of 160-entry rows (t1: G + 7 rows, t2: G + 6, t4: G + 4) */
        hipLaunchKernelGGL(k_armn_dec_compose_lds, dim3((nwin + G - 1) / G, nfields), dim3(CMP_TPB), (size_t)4 * lds_words, st, d_z, z_stride_words, ni, nj, tab, tab8, ws4, nwin, G, lds_words);
    };
    if (phase != 2) {
    if (d_out && dec_fill(st, d_out, 4 * out_stride_words, nfields, 4 * (1 + n / 2), 0)) return -1;
    if (dec_fill(st, went, work_stride_bytes, nfields, 8 * (size_t)nwin, 0xFF) || dec_fill(st, went8, work_stride_bytes, nfields, 8 * (size_t)nwin, 0xFF)) return -1;
    hipLaunchKernelGGL(k_armn_dec_spec, dim3((nwin + SPEC_WPB - 1) / SPEC_WPB, nfields), dim3(64 * SPEC_WPB), 0, st, d_z, z_stride_words, z_words, ni, nj, tab, ws4, nwin, plain);
    if ((plain && no_serial) || all_min) { }        /* (the eight-window tables serve the merged-exit form and the serial chain kernel: neither runs for a plane walked this way; MINIMUM: see above) */
    else if (getenv("EZHIP_DEC_COMPOSE_GLOBAL"))        /* development: eight dependent global loads per entry (rounds 1 - 2) */
        hipLaunchKernelGGL(k_armn_dec_compose, dim3(nwin, nfields), dim3(256), 0, st, d_z, z_stride_words, ni, nj, tab, tab8, ws4, nwin);
    else launch_compose();
    }
    if (phase == 1) { hipError_t e1 = hipGetLastError(); if (e1 != hipSuccess) { fprintf(stderr, "<armn_compress> UNCOMPRESS launch failed: %s\n", hipGetErrorString(e1)); return -1; } return 0; }
    const unsigned *skip = NULL;
    int all_ok_known = 0;                                    /* every field's chain resolved by a parallel form (known when the verdicts were read back) */
    if (no_serial && hipMemsetAsync(d_status, 0, 4 * (size_t)nfields, st) != hipSuccess) return -1;
    if (scan) {
        const int ext_max = DEXT;                            /* (the kernels read the geometry from the stream; here ext only bounds a table row) */
        if (dec_fill(st, wentc, work_stride_bytes, nfields, 8 * (size_t)nwin, 0xFF) || dec_fill(st, wentc8, work_stride_bytes, nfields, 8 * (size_t)nwin, 0xFF)
            || dec_fill(st, rowhead, work_stride_bytes, nfields, 4 * ntyc, 0)) return -1;      /* (the row walk marks the rows it steps from) */
        const size_t ws2b = work_stride_bytes / 2;
        if (all_min) { if (dec_fill(st, dst, work_stride_bytes, nfields, 4 * DSC_WORDS, 0)) return -1; }      /* (no merged-exit form: nothing resolved yet) */
        else
        hipLaunchKernelGGL(k_dsc_init, dim3(nfields), dim3(64), 0, st, d_z, z_stride_words, z_words, ni, nj, dst, ws4, wentc, ws4, nwin, plain,
                           /* a plane of c_armn_compress32 goes straight to the composed forms below (the parallel form's kernels then see FAIL and
                            * leave at once): its mantissa tiles (~185 bits) do not merge within eight windows, the attempt cost ~1 ms per plane */
                           (plain && !getenv("EZHIP_DEC_NO_DMIN")) ? 0x7fffffff : getenv("EZHIP_DEC_SCAN_MIN_NTX") ? atoi(getenv("EZHIP_DEC_SCAN_MIN_NTX")) : 768);       /* (rows of 853 tiles: 0.65 against 0.65 - 1.1 ms; of 480: the form gives up and costs 0.2 ms more than it saves) */
        if (!(plain && !getenv("EZHIP_DEC_NO_DMIN")) && !all_min) {        /* (a plane: k_dsc_init has just said FAIL, the composed forms below take it) */
        hipLaunchKernelGGL(k_dsc_m8, dim3((nwin + 3) / 4, nfields), dim3(256), 0, st, d_z, z_stride_words, ni, nj, dst, ws4, tab8, ws4, m8, ws2b, nwin, plain);
        hipLaunchKernelGGL(k_dsc_blocks, dim3((unsigned)((nblk + 255) / 256), nfields), dim3(256), 0, st, dst, ws4, tab8, ws4, m8, ws2b, eblk, cblk, ws4, ext_max);
        hipLaunchKernelGGL(k_dsc_scan, dim3(nfields), dim3(1024), 0, st, dst, ws4, tab, ws4, eblk, cblk, ws4, wentc, wentc8, ws4, nwin, ext_max);
        hipLaunchKernelGGL(k_armn_dec_expand8, dim3((nwin + 255) / 256, nfields), dim3(256), 0, st, d_z, z_stride_words, ni, nj, tab, ws4, wentc, wentc8, ws4, nwin, (const unsigned *)NULL, (size_t)0);
        hipLaunchKernelGGL(k_dsc_emit, dim3((nwin + 63) / 64, nfields), dim3(64), 0, st, d_z, z_stride_words, z_words, ni, nj, dst, ws4, wentc, ws4, tpc, ws4, (unsigned)capc, nwin, plain);
        hipLaunchKernelGGL(k_dsc_dtab, dim3((unsigned)((capc + 255) / 256), nfields), dim3(256), 0, st, d_z, z_stride_words, z_words, ni, nj, dst, ws4, tab, ws4, wentc, ws4, tpc, ws4, dtab, ws2b, (unsigned)capc, nwin, plain);
        {
            const dim3 gj((unsigned)((capc + 255) / 256), nfields);
            hipLaunchKernelGGL(k_dsc_jump<short>, gj, dim3(256), 0, st, d_z, z_stride_words, ni, nj, dst, ws4, dtab, ws2b, jA, ws4, 1u, (unsigned)capc, plain);
            hipLaunchKernelGGL(k_dsc_jump<int>, gj, dim3(256), 0, st, d_z, z_stride_words, ni, nj, dst, ws4, jA, ws4, jB, ws4, 2u, (unsigned)capc, plain);
            hipLaunchKernelGGL(k_dsc_jump<int>, gj, dim3(256), 0, st, d_z, z_stride_words, ni, nj, dst, ws4, jB, ws4, jA, ws4, 4u, (unsigned)capc, plain);
        }
        hipLaunchKernelGGL(k_dsc_rows, dim3(nfields), dim3(64), 0, st, d_z, z_stride_words, ni, nj, dst, ws4, dtab, ws2b, jA, ws4, rowc, rowhead, ws4, (unsigned)capc, plain);
        hipLaunchKernelGGL(k_dsc_rowfill, dim3((unsigned)((ntyc + 63) / 64), nfields), dim3(64), 0, st, d_z, z_stride_words, ni, nj, dst, ws4, dtab, ws2b, rowc, rowhead, ws4, (unsigned)capc, plain);
        hipLaunchKernelGGL(k_dsc_stretch, dim3((unsigned)((ntyc + 63) / 64), nfields), dim3(64), 0, st, d_z, z_stride_words, z_words, ni, nj, dst, ws4, tab, ws4, wentc, ws4, tpc, ws4, rowc, rowmerge, ws4, slist, nlist, ws4, tilepos, ws4, nwin, plain);
        hipLaunchKernelGGL(k_dsc_stretch2, dim3((unsigned)((ntyc * DSC_LIST + 255) / 256), nfields), dim3(256), 0, st, d_z, z_stride_words, z_words, ni, nj, dst, ws4, slist, nlist, ws4, ws4, tilepos, ws4, plain);
        hipLaunchKernelGGL(k_dsc_final, dim3((unsigned)((max_tiles + 255) / 256), nfields), dim3(256), 0, st, d_z, z_stride_words, ni, nj, dst, ws4, tpc, ws4, rowc, rowmerge, ws4, tilepos, ws4, plain);
        hipLaunchKernelGGL(k_dsc_lastrow, dim3(nfields), dim3(64), 0, st, d_z, z_stride_words, z_words, ni, nj, dst, ws4, tpc, ws4, rowc, ws4, tilepos, ws4, d_status, plain);
        }
        /* What is still to do?  The two composed forms below are second chances: ~35 launches that leave at once when the first form resolved every field --
         * ~125 us of dispatches behind a 1 ms decode.  For a plane the geometry says which of them applies; for armn_compress streams the fields' verdicts and
         * methods are read back (one small copy, one wait: ~20 us) and the forms nobody needs are not launched.  EZHIP_DEC_ASYNC=1: no read-back, everything queued */
        int want_whole = 1, want_ragged = 1, all_ok = 0;
        if (plain) { want_whole = (ni - 1) % 3 == 0; want_ragged = !want_whole; }
        else if (all_min) { want_whole = ni % 5 == 0; want_ragged = !want_whole; }
        else if (!getenv("EZHIP_DEC_ASYNC") && nfields <= 256) {
            unsigned hv[2 * 256];
            hipError_t ec = nfields == 1 ? hipMemcpyAsync(hv, dst, 4, hipMemcpyDeviceToHost, st)
                                         : hipMemcpy2DAsync(hv, 4, dst, work_stride_bytes, 4, (size_t)nfields, hipMemcpyDeviceToHost, st);
            if (ec == hipSuccess) ec = nfields == 1 ? hipMemcpyAsync(hv + 256, d_z, 4, hipMemcpyDeviceToHost, st)
                                                    : hipMemcpy2DAsync(hv + 256, 4, d_z, 4 * z_stride_words, 4, (size_t)nfields, hipMemcpyDeviceToHost, st);
            if (ec != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { fprintf(stderr, "<armn_compress> UNCOMPRESS: reading the verdicts back failed\n"); return -1; }
            want_whole = want_ragged = 0; all_ok = 1;
            for (int f = 0; f < nfields; f++) {
                if (hv[f]) continue;                              /* dst[DSC_OK] */
                all_ok = 0;
                const unsigned method = hv[256 + f] & 15u;
                if (method == 4) { if ((ni - 1) % 3 == 0) want_whole = 1; else want_ragged = 1; }
                else if (method == 3) { if (ni % 5 == 0) want_whole = 1; else want_ragged = 1; }
            }
        }
        if (!getenv("EZHIP_DEC_NO_DMIN") && (want_whole || want_ragged)) {
            /* streams whose tiles all hold the same number of points up to the last row (MINIMUM with ni a multiple of 5, PARALLELOGRAM and the planes of
             * c_armn_compress32 with ni - 1 a multiple of 3) that the form above did not resolve: the chain by composition of the windows' maps (k_dmin_*) */
            DmPlan pl;
            dmin_plan(&pl, (size_t)nwin);
            const size_t ws8 = work_stride_bytes / 8;
            const int rag_min_ntx = getenv("EZHIP_DEC_NO_RAGGED") ? 0x7fffffff : getenv("EZHIP_DEC_RAGGED_MIN_NTX") ? atoi(getenv("EZHIP_DEC_RAGGED_MIN_NTX")) : plain ? 256 : 768;
            /* (rows the stretches fit in: a mantissa plane rejoins after ~150 tiles on average, 1000 seen; for armn_compress streams the threshold of the first form: the
             * composed form is its second chance on long rows whose eight-window blocks do not merge.  tools: EZHIP_DEC_SCAN=2 prints both verdicts) */
            hipLaunchKernelGGL(k_dmin_init, dim3(nfields), dim3(64), 0, st, d_z, z_stride_words, ni, nj, dmst, ws4, dst, ws4, nwin, plain, rag_min_ntx);
            for (int l = 0; l < pl.nlev; l++)
                hipLaunchKernelGGL(k_dmin_up, dim3((unsigned)(((unsigned long long)pl.nblk[l + 1] * DEXT + 255) / 256), nfields), dim3(256), 0, st, d_z, z_stride_words, ni, nj, dmst, ws4, tab, ws4, dmlev, ws8, pl, l, DEXT, plain);
            if (pl.nlev >= 1) {
                /* the blocks' entries start out invalid (small: one uint2 per block of every level) */
                if (dec_fill(st, dmlev + pl.eoff[1], work_stride_bytes, nfields, 8 * (size_t)(pl.off[0] - pl.eoff[1]), 0xFF)) return -1;
                if (want_whole) {
                hipLaunchKernelGGL(k_dmin_top, dim3(nfields), dim3(64), 0, st, d_z, z_stride_words, z_words, ni, nj, dmst, ws4, tab, ws4, dmlev, ws8, went, ws4, pl, DEXT, plain);
                for (int l = pl.nlev - 1; l >= 0; l--)
                    hipLaunchKernelGGL(k_dmin_down, dim3((pl.nblk[l + 1] + 255) / 256, nfields), dim3(256), 0, st, dmst, ws4, tab, ws4, dmlev, ws8, went, ws4, pl, l, DEXT);
                hipLaunchKernelGGL(k_dmin_lastrow, dim3(nfields), dim3(256), 0, st, d_z, z_stride_words, z_words, ni, nj, dmst, ws4, went, ws4, tilepos, ws4, nwin, d_status, dst, ws4, plain);
                hipLaunchKernelGGL(k_armn_dec_emit, dim3((nwin + 255) / 256, nfields), dim3(256), 0, st, d_z, z_stride_words, z_words, ni, nj, went, ws4, tilepos, ws4, nwin, plain, dmst, ws4);
                }
                if (want_ragged && rag_min_ntx != 0x7fffffff) {
                    /* ragged rows the first form left unresolved: the canonical chain by composition (into wentc), then the row recurrence of the first form on
                     * a state of its own (k_drg_* above) */
                    if (dec_fill(st, wentc, work_stride_bytes, nfields, 8 * (size_t)nwin, 0xFF)) return -1;
                    hipLaunchKernelGGL(k_dsc_init, dim3(nfields), dim3(64), 0, st, d_z, z_stride_words, z_words, ni, nj, dst2, ws4, wentc, ws4, nwin, plain, rag_min_ntx, (const unsigned *)dst, ws4);
                    hipLaunchKernelGGL(k_dmin_top, dim3(nfields), dim3(64), 0, st, d_z, z_stride_words, z_words, ni, nj, dmst, ws4, tab, ws4, dmlev, ws8, wentc, ws4, pl, DEXT, plain, 1);
                    for (int l = pl.nlev - 1; l >= 0; l--)
                        hipLaunchKernelGGL(k_dmin_down, dim3((pl.nblk[l + 1] + 255) / 256, nfields), dim3(256), 0, st, dmst, ws4, tab, ws4, dmlev, ws8, wentc, ws4, pl, l, DEXT, 1);
                    hipLaunchKernelGGL(k_drg_ncanon, dim3(nfields), dim3(64), 0, st, dst2, ws4, dmst, ws4, tab, ws4, wentc, ws4, nwin, ext_max);
                    hipLaunchKernelGGL(k_dsc_emit, dim3((nwin + 63) / 64, nfields), dim3(64), 0, st, d_z, z_stride_words, z_words, ni, nj, dst2, ws4, wentc, ws4, tpc, ws4, (unsigned)capc, nwin, plain);
                    hipLaunchKernelGGL(k_dsc_dtab, dim3((unsigned)((capc + 255) / 256), nfields), dim3(256), 0, st, d_z, z_stride_words, z_words, ni, nj, dst2, ws4, tab, ws4, wentc, ws4, tpc, ws4, dtab, ws2b, (unsigned)capc, nwin, plain);
                    const dim3 gj((unsigned)((capc + 255) / 256), nfields);
                    hipLaunchKernelGGL(k_dsc_jump<short>, gj, dim3(256), 0, st, d_z, z_stride_words, ni, nj, dst2, ws4, dtab, ws2b, jA, ws4, 1u, (unsigned)capc, plain);
                    hipLaunchKernelGGL(k_dsc_jump<int>, gj, dim3(256), 0, st, d_z, z_stride_words, ni, nj, dst2, ws4, jA, ws4, jB, ws4, 2u, (unsigned)capc, plain);
                    hipLaunchKernelGGL(k_dsc_jump<int>, gj, dim3(256), 0, st, d_z, z_stride_words, ni, nj, dst2, ws4, jB, ws4, jA, ws4, 4u, (unsigned)capc, plain);
                    if (dec_fill(st, rowhead, work_stride_bytes, nfields, 4 * ntyc, 0)) return -1;
                    DscLev lv;
                    lv.lev = dmlev; lv.lev_stride = ws8; lv.nlev = pl.nlev;
                    for (int l = 0; l <= DM_MAXLEV; l++) { lv.nblk[l] = pl.nblk[l]; lv.off[l] = pl.off[l]; }
                    if (getenv("EZHIP_DEC_NO_SLOW_ROWS")) lv.lev = NULL;      /* development: a row whose stretch does not rejoin in time fails the form, as before round 5 */
                    hipLaunchKernelGGL(k_dsc_rows, dim3(nfields), dim3(64), 0, st, d_z, z_stride_words, ni, nj, dst2, ws4, dtab, ws2b, jA, ws4, rowc, rowhead, ws4, (unsigned)capc, plain,
                                       z_words, (const unsigned *)tab, (const unsigned *)wentc, (const unsigned *)tpc, ws4, rowpn, lv, nwin);
                    hipLaunchKernelGGL(k_dsc_rowfill, dim3((unsigned)((ntyc + 63) / 64), nfields), dim3(64), 0, st, d_z, z_stride_words, ni, nj, dst2, ws4, dtab, ws2b, rowc, rowhead, ws4, (unsigned)capc, plain);
                    hipLaunchKernelGGL(k_dsc_stretch, dim3((unsigned)((ntyc + 63) / 64), nfields), dim3(64), 0, st, d_z, z_stride_words, z_words, ni, nj, dst2, ws4, tab, ws4, wentc, ws4, tpc, ws4, rowc, rowmerge, ws4, slist, nlist, ws4, tilepos, ws4, nwin, plain, (const unsigned *)rowpn);
                    hipLaunchKernelGGL(k_dsc_stretch2, dim3((unsigned)((ntyc * DSC_LIST + 255) / 256), nfields), dim3(256), 0, st, d_z, z_stride_words, z_words, ni, nj, dst2, ws4, slist, nlist, ws4, ws4, tilepos, ws4, plain);
                    hipLaunchKernelGGL(k_dsc_final, dim3((unsigned)((max_tiles + 255) / 256), nfields), dim3(256), 0, st, d_z, z_stride_words, ni, nj, dst2, ws4, tpc, ws4, rowc, rowmerge, ws4, tilepos, ws4, plain);
                    hipLaunchKernelGGL(k_dsc_lastrow, dim3(nfields), dim3(64), 0, st, d_z, z_stride_words, z_words, ni, nj, dst2, ws4, tpc, ws4, rowc, ws4, tilepos, ws4, d_status, plain, (const unsigned *)rowpn);
                    hipLaunchKernelGGL(k_drg_done, dim3(nfields), dim3(64), 0, st, dst, dst2, ws4);
                }
            }
        }
        skip = dst;                                          /* dst[DSC_OK] of a field: 1 when its tile positions are complete */
        all_ok_known = all_ok;
        if (all_min) {                                       /* did the composed form resolve every field?  (one small copy, one wait) else: the tables of the serial kernel, now */
            unsigned hv[256];
            hipError_t ec = nfields == 1 ? hipMemcpyAsync(hv, dst, 4, hipMemcpyDeviceToHost, st)
                                         : hipMemcpy2DAsync(hv, 4, dst, work_stride_bytes, 4, (size_t)nfields, hipMemcpyDeviceToHost, st);
            if (ec != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { fprintf(stderr, "<armn_compress> UNCOMPRESS: reading the verdicts back failed\n"); return -1; }
            all_ok_known = 1;
            for (int f = 0; f < nfields; f++) if (!hv[f]) all_ok_known = 0;
            if (!all_ok_known) launch_compose();
        }
        if (scan == 2) {                                     /* development: the state words of every field */
            unsigned h[DSC_WORDS];
            for (int f = 0; f < nfields; f++) {
                if (hipMemcpyAsync(h, dst + (size_t)f * ws4, sizeof h, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return -1;
                fprintf(stderr, "<armn_compress> scan form, field %d: ok %u fail %u  first block window %u entry %u tiles %u  blocks %u  canonical tiles %u  interior rows %u\n",
                        f, h[DSC_OK], h[DSC_FAIL], h[DSC_WB], h[DSC_EB], h[DSC_TB], h[DSC_K], h[DSC_NCANON], h[DSC_RINT]);
                if (hipMemcpyAsync(h, dst2 + (size_t)f * ws4, sizeof h, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return -1;
                fprintf(stderr, "<armn_compress> composed ragged form, field %d: ok %u fail %u  canonical tiles %u  interior rows %u  (row %u, canonical tile %u)  rows walked explicitly %u, eight-row jumps %u, single steps %u\n", f, h[DSC_OK], h[DSC_FAIL], h[DSC_NCANON], h[DSC_RINT], h[8], h[9], h[10], h[11], h[12]);
            }
        }
    }
    if (no_serial) {
        if (!scan) return -1;
        hipLaunchKernelGGL(k_drg_verdict, dim3(nfields), dim3(64), 0, st, dst, ws4, d_status);
        hipError_t e3 = hipGetLastError(); if (e3 != hipSuccess) { fprintf(stderr, "<armn_compress> tile walk launch failed: %s\n", hipGetErrorString(e3)); return -1; }
        return 0;
    }
    if (!all_ok_known) {
    hipLaunchKernelGGL(k_armn_dec_hop, dim3(nfields), dim3(HOP_TPB), HOP_LDS, st, d_z, z_stride_words, z_words, ni, nj, tab, tab8, ws4, went, went8, ws4, nwin, d_status, getenv("EZHIP_DEC_DEBUG") ? atoi(getenv("EZHIP_DEC_DEBUG")) : 0, plain, skip, ws4);
    hipLaunchKernelGGL(k_armn_dec_expand8, dim3((nwin + 255) / 256, nfields), dim3(256), 0, st, d_z, z_stride_words, ni, nj, tab, ws4, went, went8, ws4, nwin, skip, ws4);
    hipLaunchKernelGGL(k_armn_dec_emit, dim3((nwin + 255) / 256, nfields), dim3(256), 0, st, d_z, z_stride_words, z_words, ni, nj, went, ws4, tilepos, ws4, nwin, plain, skip, ws4);
    }
    if (walk_only) { hipError_t e2 = hipGetLastError(); if (e2 != hipSuccess) { fprintf(stderr, "<armn_compress> tile walk launch failed: %s\n", hipGetErrorString(e2)); return -1; } return 0; }
    const long long grid_tiles = max_tiles > (long long)(ni + nj) ? max_tiles : (long long)(ni + nj);
    hipLaunchKernelGGL(k_armn_dec_tiles, dim3((unsigned)((grid_tiles + 255) / 256), nfields), dim3(256), 0, st, d_z, z_stride_words, z_words, ni, nj,
                       tilepos, ws4, d16, ws2, (unsigned short *)d_out, out_stride_words * 2, swap, max_tiles);
    hipLaunchKernelGGL(k_armn_dec_rows, dim3(nj, nfields), dim3(256), 0, st, d_z, z_stride_words, d16, ws2, ni, nj);
    const int band = (nj + DEC_BANDS - 1) / DEC_BANDS;
    const int nbands = (nj + band - 1) / band;
    hipLaunchKernelGGL(k_armn_dec_colsum, dim3((ni + 255) / 256, nbands, nfields), dim3(256), 0, st, d_z, z_stride_words, d16, ws2, part, ws2, ni, nj, band);
    hipLaunchKernelGGL(k_armn_dec_cols, dim3((ni + 255) / 256, nbands, nfields), dim3(256), 0, st, d_z, z_stride_words, d16, ws2, part, ws2,
                       (unsigned short *)d_out, out_stride_words * 2, swap, ni, nj, band);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { fprintf(stderr, "<armn_compress> UNCOMPRESS launch failed: %s\n", hipGetErrorString(e)); return -1; }
    return 0;
}
