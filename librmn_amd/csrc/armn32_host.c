/*
 * armn32_host.c -- C host front-end of the IEEE-32 compressor (datyp 133): c_armn_compress32 / c_armn_uncompress32 of librmn
 * (src/compresseur/armn_compress_32.c:59-275, :285-437; called by c_fstecr / c_fstluk, fstd98.c:1309, :2436) over the HIP kernels of
 * armn32_kernels.hip.  There is NO CPU fallback: without a HIP device every entry point fails loudly.
 *
 * What runs where: the three planes (sign, exponent, mantissa), the two parallelogram encoders, and on the way back the token extraction,
 * the inverse predictor (2-D prefix sums) and the re-assembly of the floats are device kernels.  Two strictly sequential steps over small
 * data stay on the host: the run-length coder of the sign plane (1 bit per point, a state machine: pack1bitRLE :827-901) and, when
 * decoding, the walk along the chain of tile headers (a tile's position is the sum of all earlier tiles' lengths, each read from the stream).
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

#include "ezhip_shim.h"
#include "packhip_shim.h"
#include "../../include/packers_hip.h"

static int need_dev32(const char *who)
{
    if (ezhip_runtime_ok()) return ezhip_bound_device_ok(who);
    fprintf(stderr, "<%s> no usable HIP device: the MI355X packer path has no CPU fallback\n", who);
    return -1;
}
/* per-thread grow-only device workspaces */
static __thread struct { void *p; size_t cap; } t_w32[14];
static void *w32(int slot, size_t bytes)
{
    if (t_w32[slot].cap < bytes) {
        if (t_w32[slot].p) { ezhip_sync(); ezhip_free(t_w32[slot].p); }
        t_w32[slot].p = ezhip_malloc(bytes + 256);
        t_w32[slot].cap = t_w32[slot].p ? bytes : 0;
    }
    return t_w32[slot].p;
}

void ezh_armn32_thread_release(void) { for (int k = 0; k < 14; k++) { ezhip_free(t_w32[k].p); t_w32[k].p = NULL; t_w32[k].cap = 0; } }

/* ---- MSB-first bit stream on host words: the `stuff` / `extract` rules (include/bitPacking.h:59-139) ---- */
typedef struct { uint32_t *z; uint64_t pos; } bitw32;              /* over a zeroed buffer */
static inline void bw32_put(bitw32 *w, uint32_t tok, int bits)              /* bits <= 32, tok < 2^bits, buffer zeroed: at most two words touched */
{
    const uint64_t p = w->pos;
    const int sh = 64 - bits - (int)(p & 31);
    const uint64_t v = (uint64_t)tok << sh;
    w->z[p >> 5] |= (uint32_t)(v >> 32);
    if ((uint32_t)v) w->z[(p >> 5) + 1] |= (uint32_t)v;
    w->pos = p + (uint64_t)bits;
}
static inline uint32_t br32_get(const uint32_t *z, uint64_t *pos, int bits)  /* bits <= 32; the second word is read only when the field reaches it */
{
    if (!bits) return 0;
    const uint64_t p = *pos;
    const int o = (int)(p & 31);
    uint64_t v = (uint64_t)z[p >> 5] << 32;
    if (o + bits > 32) v |= z[(p >> 5) + 1];
    *pos = p + (uint64_t)bits;
    return (uint32_t)(v >> (64 - bits - o)) & (bits == 32 ? 0xFFFFFFFFu : (1u << bits) - 1u);
}
/* a stream of P data bits followed by the 32 terminator bits occupies ceil(P / 32) flushed words: zlng = 1 + 4 * that (:559-561) */
static uint32_t zlng_of_bits(uint64_t P) { return 1u + 4u * (uint32_t)((P + 31) >> 5); }

/* sign plane -> run-length stream (pack1bitRLE): runs of 8 .. 62 equal signs are one {1, sign, count} token, a run longer than 256 repeats
 * its 62-token once and then 0xFF tokens of 255 points, anything shorter is 7 raw bits behind a 0.  s(i) = bit i & 31 of mask word i >> 5. */
static uint32_t rle_encode(uint32_t *z, const uint32_t *mask, int npts)
{
#define SGN(i) ((mask[(i) >> 5] >> ((i) & 31)) & 1u)
    bitw32 w = {z, 0};
    int last = 0, idx = 1;
    while (idx <= npts) {
        {   /* the end of the run of equal signs that starts at `last`, 32 points per step */
            const uint32_t flip = SGN(last) ? 0xFFFFFFFFu : 0u;
            while (idx < npts) {
                uint32_t dif = (mask[idx >> 5] ^ flip) & (0xFFFFFFFFu << (idx & 31));       /* points idx .. of this word that differ from the run's sign */
                if (dif) { const int nxt = (idx & ~31) + __builtin_ctz(dif); idx = nxt < npts ? nxt : npts; break; }
                idx = (idx & ~31) + 32;
                if (idx > npts) idx = npts;
            }
        }
        const int count = idx - last;
        int i = 0, repeat = 0;
        do {
            int c = count < 8 ? count : ((count - i) >= 63 ? 62 : count - i);
            if (c < 8) {
                const int lim = last + 7 > npts ? npts - last : 7;
                uint32_t raw = 0;
                for (int j = 0; j < lim; j++) raw = raw << 1 | SGN(last + j);
                bw32_put(&w, raw, 1 + lim);                      /* the 0 flag, then the signs, first point first */
                last += 7;
            } else if (c == 62 && (count - i) > 256 && repeat) {
                c = 255;
                bw32_put(&w, 0xFF, 8);
                last += c;
            } else {
                bw32_put(&w, 1, 1); bw32_put(&w, SGN(last), 1); bw32_put(&w, (uint32_t)c, 6);
                last += c;
                if (c == 62) repeat = 1;
            }
            idx = last + 1;
            i += c;
        } while (count >= 8 && i < count);
    }
#undef SGN
    return zlng_of_bits(w.pos);
}
/* run-length stream -> sign mask (unpack1bitRLE :904-955) */
/* max_bits: the bits of the sub-stream (its length word, read from the record): a damaged stream stops there instead of running off the caller's buffer.  0 / -1 */
static int rle_decode(uint32_t *mask, const uint32_t *z, int npts, uint64_t max_bits)
{
    uint64_t pos = 0;
    uint32_t last_val = 1;                   /* (unsigned char)0xFFFFFFFF of the reference is never used before a count token sets it */
    int i = 0;
#define SETS(k, v) do { if ((k) < npts && (v)) mask[(k) >> 5] |= 1u << ((k) & 31); } while (0)
    while (i < npts) {
        if (pos + 1 > max_bits) return -1;
        if (br32_get(z, &pos, 1) == 0) {
            const int lim = i + 7 > npts ? npts - i : 7;
            if (pos + (uint64_t)lim > max_bits) return -1;
            const uint32_t raw = br32_get(z, &pos, lim);
            for (int j = 0; j < lim; j++) SETS(i + j, (raw >> (lim - 1 - j)) & 1u);
            i += lim;
        } else {
            if (pos + 7 > max_bits) return -1;
            const uint32_t tok = br32_get(z, &pos, 7), val = tok >> 6; const int count = (int)(tok & 63u);
            const int run = count == 63 ? 255 : count;
            const uint32_t bit = count == 63 ? (last_val & 1u) : val;
            if (bit) for (int j = 0; j < run; j++) SETS(i + j, 1u);
            i += run;
            if (count != 63) last_val = val;
        }
    }
#undef SETS
    return 0;
}

static int float_width(uint32_t v) { union { float f; int32_t i; } r; r.f = (float)v; return v ? (r.i >> 23) - 126 : 0; }

static void *rle_grow(int slot, size_t bytes) { return w32(10 + slot, bytes); }      /* the run arrays of the device run-length coder: slots 10 .. 13 */
/* c_armn_compress32 on device data.  d_z: device buffer of at least ni*nj*znbits/8 + 64 bytes; returns the stream's byte count or -1 */
int c_armn_compress32_dev(void *d_z, const float *d_fld, int ni, int nj, int nk, int znbits)
{
    (void)nk;
    if (ni < 16 || nj < 16) { fprintf(stderr, "<c_armn_compress32> The dimensions of NI and NJ have to be > 16\n"); return -1; }
    if (need_dev32("c_armn_compress32")) return -1;
    const int nbits = znbits - 9;
    if (nbits < 1 || nbits > 23) { fprintf(stderr, "<c_armn_compress32> nbits %d outside 10 .. 32\n", znbits); return -1; }
    const size_t n = (size_t)ni * nj;
    const size_t plane_words = n + n / 4 + 64;                  /* capacity of one plane's scratch stream */
    unsigned char *d_expo = (unsigned char *)w32(0, n + 16);
    unsigned *d_mant = (unsigned *)w32(1, 4 * n + 16), *d_smask = (unsigned *)w32(2, 4 * (n / 32 + 2) + 16);
    unsigned *d_ze = (unsigned *)w32(3, 4 * plane_words), *d_zm = (unsigned *)w32(4, 4 * plane_words);
    char *d_work = (char *)w32(5, 2 * packhip_pg_work_bytes(ni, nj) + 256);
    if (!d_expo || !d_mant || !d_smask || !d_ze || !d_zm || !d_work) return -1;
    unsigned st[4];
    if (packhip_a32_split(d_expo, d_mant, d_smask, (unsigned *)d_work, d_fld, n, nbits, st)) return -1;
    const int meme_signe = (st[0] >> 31) == (st[1] >> 31);
    const unsigned exp_base = st[2], exp_span = st[3] - st[2];
    const int need_e = float_width(exp_span);                    /* :143-145 */
    unsigned char *z8 = (unsigned char *)d_z;
    size_t off = 8;                                              /* bytes written so far */
    uint32_t lng_signe = 0, code_signe, code_expo = 0;
    /* sign stream (:148-180) */
    if (meme_signe) code_signe = (st[0] >> 31) ? 0x10 : 0x00;
    else if (!getenv("EZHIP_A32_RLE_ENC_HOST")) {
        /* the run-length coder on the device (round 5: a prefix scan of seven-state maps, armn32_kernels.hip k_re_*): the mask stays in HBM, 8 bytes come down */
        const size_t zs_bytes = 4 * (n / 28 + 8);
        unsigned *d_zs = (unsigned *)w32(8, zs_bytes);
        void *d_wk = w32(9, packhip_a32_rle_enc_work_bytes(n));
        unsigned long long bits_s = 0;
        if (!d_zs || !d_wk || ezhip_memset(d_zs, 0, zs_bytes) || packhip_a32_rle_encode(d_zs, d_smask, n, d_wk, rle_grow, &bits_s)) return -1;
        lng_signe = zlng_of_bits(bits_s);
        code_signe = 0x20;
        if (lng_signe % 4) lng_signe += 4 - lng_signe % 4;
        if ((size_t)lng_signe > zs_bytes || ezhip_h2d(z8 + off, &lng_signe, 4) || ezhip_d2d(z8 + off + 4, d_zs, lng_signe) || ezhip_sync()) return -1;
        off += 4 + lng_signe;
    } else {
        const size_t mw = n / 32 + 2;
        uint32_t *mask = (uint32_t *)calloc(mw, 4), *zs = (uint32_t *)calloc(n / 16 + 64, 4);      /* the RLE needs at most 8 bits per 7 points */
        if (!mask || !zs || ezhip_d2h(mask, d_smask, 4 * (n / 32 + 1)) || ezhip_sync()) { free(mask); free(zs); return -1; }
        lng_signe = rle_encode(zs, mask, (int)n);
        code_signe = 0x20;
        if (lng_signe % 4) lng_signe += 4 - lng_signe % 4;
        int bad = ezhip_h2d(z8 + off, &lng_signe, 4) || ezhip_h2d(z8 + off + 4, zs, lng_signe) || ezhip_sync();
        free(mask); free(zs);
        if (bad) return -1;
        off += 4 + lng_signe;
    }
    /* the two planes are encoded back to back; their lengths come back with one synchronisation each */
    if (need_e && packhip_pg_encode(d_ze, plane_words, d_expo, 1, ni, nj, need_e, 0, d_work)) return -1;
    unsigned long long bits_e = 0; int failed = 0;
    if (need_e) {
        if (packhip_pg_result(d_work, ni, nj, need_e, &bits_e, &failed)) return -1;
        uint32_t lng_expo = zlng_of_bits(bits_e);
        if (lng_expo > n) { fprintf(stderr, "<c_armn_compress32> Exponent range too large, original field left uncompressed\n"); return -1; }
        if (lng_expo % 4) lng_expo += 4 - lng_expo % 4;
        code_expo = 0x08;
        if (ezhip_h2d(z8 + off, &lng_expo, 4) || ezhip_d2d(z8 + off + 4, d_ze, lng_expo) || ezhip_sync()) return -1;
        off += 4 + lng_expo;
    }
    const size_t pos_lng_mant = off;
    off += 4;
    const long long remaining = (long long)(((long long)ni * nj * znbits) / 32) - (long long)(off / 4);
    if (packhip_pg_encode(d_zm, plane_words, d_mant, 4, ni, nj, nbits, remaining, d_work)) return -1;
    unsigned long long bits_m = 0;
    if (packhip_pg_result(d_work, ni, nj, nbits, &bits_m, &failed)) return -1;
    if (failed) { fprintf(stderr, "<c_armn_compress32> IEEE compressed field is larger than original, keeping original\n"); return -1; }
    uint32_t lng_mant = zlng_of_bits(bits_m);
    if (lng_mant % 4) lng_mant += 4 - lng_mant % 4;
    if ((size_t)off + lng_mant > (size_t)n * (size_t)znbits / 8 + 64) return -1;
    const uint32_t head[2] = {5u | 1u << 4 | 3u << 7 | ((uint32_t)nbits & 31u) << 10 | 1u << 15 | 2u << 18,          /* _fstzip: PARALLELOGRAM32, degree 1, step 3, nbits, levels 1, version 2 */
                              (exp_base & 0xFF) << 16 | ((uint32_t)need_e & 0xFF) << 8 | code_signe | code_expo};
    if (ezhip_h2d(z8, head, 8) || ezhip_h2d(z8 + pos_lng_mant, &lng_signe, 4)          /* the mantissa length slot receives lng_signe (sic, :237) */
        || ezhip_d2d(z8 + off, d_zm, lng_mant) || ezhip_sync()) return -1;
    return (int)(off + lng_mant);
}

int c_armn_compress32(unsigned char *zstream, float *fld, int ni, int nj, int nk, int znbits)
{
    if (ni < 16 || nj < 16) { fprintf(stderr, "<c_armn_compress32> The dimensions of NI and NJ have to be > 16\n"); return -1; }
    if (need_dev32("c_armn_compress32")) return -1;
    const size_t n = (size_t)ni * nj, cap = n * (size_t)(znbits > 0 ? znbits : 32) / 8 + 256;
    float *d_f = (float *)w32(6, 4 * n);
    unsigned char *d_z = (unsigned char *)w32(7, cap);
    if (!d_f || !d_z || ezhip_h2d(d_f, fld, 4 * n)) return -1;
    int zlng = c_armn_compress32_dev(d_z, d_f, ni, nj, nk, znbits);
    if (zlng > 0 && (ezhip_d2h(zstream, d_z, (size_t)zlng) || ezhip_sync())) zlng = -1;
    ezhip_sync();
    return zlng;
}

/* bit position of every tile header of a parallelogram stream (host words): 3-bit width-field size, row 1, column 1, then the chain.
 * Runs of zero bits are taken at once: a header inside one says "no tokens" whatever the tile's point count, so every tile in it is `container` bits long (the
 * exponent plane of a smooth field is mostly such runs: 10 ms of dependent steps become ~1 ms of stores) */
static uint64_t *walk_tiles(const uint32_t *z, int ni, int nj, int nbits, size_t *ntiles_out, uint64_t max_bits)
{
    uint64_t pos = 0;
    const int container = (int)br32_get(z, &pos, 3);
    pos += (uint64_t)(ni + nj - 1) * (uint64_t)nbits;
    const int ntx = (ni - 1 + 2) / 3, nty = (nj - 1 + 2) / 3;
    const size_t ntiles = (size_t)ntx * nty;
    uint64_t *tp = (uint64_t *)malloc(sizeof(uint64_t) * (ntiles + 1));
    if (!tp) return NULL;
    const uint64_t max_word = max_bits >> 5;                      /* whole words the stream is known to hold */
    size_t t = 0;
    int tx = 0, ty = 0;
    while (t < ntiles) {
        if (pos + (uint64_t)container > max_bits) { free(tp); return NULL; }      /* a truncated or corrupt record: the chain left the stream */
        const uint64_t wi = pos >> 5;
        /* (no word is looked at that the remaining tiles, were they all empty, would not cover: the record's end is not always known) */
        const uint64_t cover = ((pos + (uint64_t)(ntiles - t) * (uint64_t)container) >> 5) + 1, lim = cover < max_word ? cover : max_word;
        if (container > 0 && wi + 1 < lim && (z[wi] & (0xFFFFFFFFu >> (pos & 31))) == 0 && z[wi + 1] == 0) {
            uint64_t wj = wi + 2;
            while (wj < lim && z[wj] == 0) wj++;
            const uint64_t zend = 32 * wj + (wj < lim ? (uint64_t)__builtin_clz(z[wj]) : 0);      /* zero bits from pos to here */
            size_t k = (size_t)((zend - pos) / (uint64_t)container);
            if (k > ntiles - t) k = ntiles - t;
            for (size_t i = 0; i < k; i++) tp[t + i] = pos + (uint64_t)i * (uint64_t)container;
            t += k; pos += (uint64_t)k * (uint64_t)container;
            const size_t x = (size_t)tx + k;
            ty += (int)(x / (size_t)ntx); tx = (int)(x % (size_t)ntx);
            continue;
        }
        const int tn = nj - (1 + 3 * ty) < 3 ? nj - (1 + 3 * ty) : 3, tm = ni - (1 + 3 * tx) < 3 ? ni - (1 + 3 * tx) : 3;
        tp[t++] = pos;
        uint64_t p2 = pos;
        const int need = (int)br32_get(z, &p2, container);
        pos += (uint64_t)container + (need ? (uint64_t)(tm * tn) * (uint64_t)(need + 1) : 0);
        if (++tx == ntx) { tx = 0; ty++; }
    }
    tp[t] = pos;
    if (pos > max_bits) { free(tp); return NULL; }
    *ntiles_out = t;
    return tp;
}

/* one plane back: host stream words + the tile positions walked on the host -> device plane of ints (slot: which workspaces) */
static int decode_plane(int *d_plane, const uint32_t *z, const uint64_t *tp, size_t ntiles, int ni, int nj, int nbits, int wide, int slot)
{
    const size_t zwords = (size_t)((tp[ntiles] + 63) >> 5) + 2;
    unsigned *d_zs = (unsigned *)w32(slot, 4 * zwords);
    unsigned long long *d_tp = (unsigned long long *)w32(slot + 1, 8 * (ntiles + 1));
    int *d_bs = (int *)w32(5, 4 * (size_t)ni * ((size_t)(nj + 31) / 32 + 1));
    int rc = -1;
    if (d_zs && d_tp && d_bs && !ezhip_h2d(d_zs, z, 4 * zwords) && !ezhip_h2d(d_tp, tp, 8 * (ntiles + 1)))
        rc = packhip_pg_decode(d_plane, d_bs, d_zs, d_tp, ni, nj, nbits, wide);
    if (ezhip_sync()) rc = -1;
    return rc;
}

/* the three sequential host walks of a stream (sign run lengths, exponent tile chain, mantissa tile chain) are independent: one thread each */
typedef struct { const uint32_t *z; int ni, nj, nbits; uint64_t *tp; size_t ntiles; uint64_t max_bits; } walk_job;
static void *walk_thread(void *a) { walk_job *j = (walk_job *)a; j->tp = walk_tiles(j->z, j->ni, j->nj, j->nbits, &j->ntiles, j->max_bits); return NULL; }
typedef struct { const uint32_t *z; uint32_t *mask; int npts; uint64_t max_bits; int rc; } rle_job;
static void *rle_thread(void *a) { rle_job *j = (rle_job *)a; j->rc = rle_decode(j->mask, j->z, j->npts, j->max_bits); return NULL; }

/* the sign run lengths on the device (packhip_a32_rle_decode): the sub-stream goes up (at most n / 7 bytes), three short kernels write the mask.  The launches
 * are queued at once -- they run while the host walks the tile chains -- and *bad is read back behind them.  EZHIP_A32_RLE_HOST=1: the host thread (rle_decode) */
static int sign_mask_on_device(unsigned *d_smask, const uint32_t *z_s, int z_on_device, uint64_t bits_s, size_t n, int *bad)
{
    const size_t nbytes = (size_t)(bits_s / 8);
    unsigned *d_zs = (unsigned *)w32(8, nbytes + 64);
    void *d_wk = w32(9, packhip_a32_rle_work_bytes(nbytes));
    if (!d_zs || !d_wk) return -1;
    if (z_on_device ? ezhip_d2d(d_zs, z_s, nbytes) : ezhip_h2d(d_zs, z_s, nbytes)) return -1;
    return packhip_a32_rle_decode(d_smask, d_zs, nbytes, n, d_wk, bad);
}

/* c_armn_uncompress32 with the result on the device; zstream: HOST memory (the chain of tile headers is walked on the host).  word_limit: words the
 * caller's buffer is known to hold (0: unknown -- no stream of c_armn_compress32 is longer than the field it replaces) */
static int uncompress32_host_walk(float *d_fld, const unsigned char *zstream, size_t word_limit, int ni, int nj, int nk, int znbits);
static int uncompress32_device_walk(float *d_fld, const uint32_t *z0, int z_on_device, size_t zwords, int ni, int nj);
static int device_walk_wanted(int ni, int nj, size_t zwords);
static __thread int t_quiet32;                /* a first attempt on a GUESSED length must not report a broken stream */
/* words of [p, p + 4 want) that this process may READ, from p on: the readable mappings of /proc/self/maps that follow one another without a gap (mincore would
 * also count PROT_NONE reservations -- glibc's arenas end in one).  c_armn_uncompress32 has no length argument; reading what is readable cannot fault, and
 * nothing read behind the record's true end is ever used (the chains stop after the field's tiles).  ~0.1 ms per call */
static size_t readable_words(const void *p, size_t want_words)
{
    FILE *f = fopen("/proc/self/maps", "r");
    if (!f) return 0;
    const uintptr_t a = (uintptr_t)p, want_end = a + 4 * want_words;
    uintptr_t end = 0;                            /* end of the readable run that holds p (0: p not found yet) */
    char line[512];
    while (fgets(line, sizeof(line), f)) {
        unsigned long lo, hi; char perms[8];
        if (sscanf(line, "%lx-%lx %7s", &lo, &hi, perms) != 3) continue;
        if (!end) { if (a >= lo && a < hi) { if (perms[0] != 'r') break; end = hi; } }
        else if (lo == end && perms[0] == 'r') end = hi;
        else break;                               /* (the file is sorted by address) */
        if (end >= want_end) break;
    }
    fclose(f);
    if (end <= a) return 0;
    const size_t w = (size_t)(((end < want_end ? end : want_end) - a) / 4);
    return w < want_words ? w : want_words;
}
/* words in front of the mantissa plane (header, sign runs, exponent plane, the length slot) and the mantissa width of a record in host memory; 0: not a record */
static size_t record_prefix_words(const uint32_t *z, size_t bound, int *nbits_out)
{
    if (bound < 4 || (z[0] & 15u) != 5u) return 0;
    const uint32_t codes = z[1] & 0xFF;
    size_t cur = 2;
    if ((codes & 0x30) == 0x20 || (codes & 0x30) == 0x30) { if (cur >= bound) return 0; cur += 1 + (size_t)(z[cur] >> 2); }
    if ((codes & 0xC) == 0x08 || (codes & 0xC) == 0x0C) { if (cur >= bound) return 0; cur += 1 + (size_t)(z[cur] >> 2); }
    *nbits_out = (int)((z[0] >> 10) & 31);
    return cur + 1 < bound ? cur + 1 : 0;
}
/* bits the first `maxtiles` tiles of a parallelogram plane take (host words, at most max_bits of them readable); 0: the walk left the readable part */
static uint64_t plane_head_bits(const uint32_t *z, int ni, int nj, int nbits, size_t maxtiles, uint64_t max_bits, size_t *tiles_out)
{
    uint64_t pos = 0;
    if (max_bits < 64) return 0;
    const int container = (int)br32_get(z, &pos, 3);
    pos += (uint64_t)(ni + nj - 1) * (uint64_t)nbits;
    const int ntx = (ni - 1 + 2) / 3, nty = (nj - 1 + 2) / 3;
    size_t ntiles = (size_t)ntx * nty, t = 0;
    if (ntiles > maxtiles) ntiles = maxtiles;
    int tx = 0, ty = 0;
    while (t < ntiles) {
        if (pos + (uint64_t)container + 64 > max_bits) return 0;
        const int tn = nj - (1 + 3 * ty) < 3 ? nj - (1 + 3 * ty) : 3, tm = ni - (1 + 3 * tx) < 3 ? ni - (1 + 3 * tx) : 3;
        uint64_t p2 = pos;
        const int need = (int)br32_get(z, &p2, container);
        pos += (uint64_t)container + (need ? (uint64_t)(tm * tn) * (uint64_t)(need + 1) : 0);
        t++;
        if (++tx == ntx) { tx = 0; ty++; }
    }
    *tiles_out = t;
    return pos;
}
int c_armn_uncompress32_dev(float *d_fld, const unsigned char *zstream, int ni, int nj, int nk, int znbits)
{
    /* (round 5) the record's end is not an argument and cannot be read from the record (its mantissa length slot holds the sign stream's length, armn_compress_32.c:237).
     * Rounds 3 - 4 found it by walking both tile chains on host threads (10 - 12 ms of a 7200 x 3601 field's 18).  Now: whatever of the largest possible record
     * (no record is longer than its field) is READABLE behind the pointer may be read; a first attempt uploads the sub-streams whose lengths the record states and
     * a mantissa plane of ESTIMATED length (its first tiles walked on the host, extrapolated, + 12 %) and resolves the chains on the device (the forms of c_armn_uncompress32_lng); a chain that leaves that piece
     * sends the rest up for a second attempt; planes the device forms leave open and small fields take the host walk as before */
    if (!getenv("EZHIP_A32_HOST_END") && need_dev32("c_armn_uncompress32") == 0 && ni >= 16 && nj >= 16) {
        const size_t n = (size_t)ni * nj, bound = readable_words(zstream, n + 64);
        int nbits = 0;
        const size_t pre = bound >= 4 ? record_prefix_words((const uint32_t *)zstream, bound, &nbits) : 0;
        if (pre && nbits >= 1 && nbits <= 23 && device_walk_wanted(ni, nj, bound)) {
            /* the mantissa plane's length from its first 98 304 tiles (~0.3 ms on the host: a field's tiles are much alike), + 12 % */
            size_t seen = 0;
            const size_t ntiles = (size_t)((ni - 1 + 2) / 3) * (size_t)((nj - 1 + 2) / 3);
            const uint64_t hb = plane_head_bits((const uint32_t *)zstream + pre, ni, nj, nbits, 98304, 32ull * (bound - pre), &seen);
            size_t guess = bound;
            if (hb && seen) {
                const double est = (double)hb / (double)seen * (double)ntiles * 1.12 / 32.0;
                if (est < (double)(bound - pre)) guess = pre + (size_t)est + 16384;
            }
            if (guess > bound) guess = bound;
            t_quiet32 = guess < bound;
            int rc = uncompress32_device_walk(d_fld, (const uint32_t *)zstream, 0, guess, ni, nj);
            t_quiet32 = 0;
            if (rc == -1 && guess < bound) rc = uncompress32_device_walk(d_fld, (const uint32_t *)zstream, 0, bound, ni, nj);
            if (rc != 1) return rc;
        }
    }
    return uncompress32_host_walk(d_fld, zstream, 0, ni, nj, nk, znbits);
}
static int uncompress32_host_walk(float *d_fld, const unsigned char *zstream, size_t word_limit, int ni, int nj, int nk, int znbits)
{
    (void)nk; (void)znbits;
    if (need_dev32("c_armn_uncompress32")) return -1;
    const size_t n = (size_t)ni * nj;
    const uint32_t *cur = (const uint32_t *)zstream;
    const uint32_t w0 = cur[0], info = cur[1];
    cur += 2;
    if ((w0 & 15u) != 5u) { fprintf(stderr, "<c_armn_uncompress32> not a PARALLELOGRAM32 stream\n"); return -1; }
    const int nbits = (int)((w0 >> 10) & 31);
    const uint32_t exp_min = info >> 16, need_e = (info >> 8) & 0xFF, codes = info & 0xFF;
    const int code_signe = (int)(codes & 0x30), code_expo = (int)(codes & 0xC), code_mant = (int)(codes & 0x3);
    if (code_mant != 0) { fprintf(stderr, "<c_armn_uncompress32> plain mantissa streams are not produced by c_armn_compress32\n"); return -1; }
    /* header fields a record of c_armn_compress32 can hold (armn_compress_32.c:143-145, :96-99): anything else is a corrupt record */
    if (need_e > 8 || nbits < 1 || nbits > 23) { fprintf(stderr, "<c_armn_uncompress32> broken stream (exponent width %u, mantissa width %d)\n", need_e, nbits); return -1; }
    /* no stream of c_armn_compress32 is longer than the field it replaces (:240-247): every length read from the record is held to that */
    const uint64_t max_words = (word_limit && word_limit < n + 64) ? (uint64_t)word_limit : (uint64_t)n + 64;
    unsigned *d_smask = (unsigned *)w32(2, 4 * (n / 32 + 2) + 16);
    int *d_expo = (int *)w32(0, 4 * n + 16), *d_mant = (int *)w32(1, 4 * n + 16);
    if (!d_smask || !d_expo || !d_mant) return -1;
    /* the three sub-streams: [lng, sign run lengths] [lng, exponent plane] [slot, mantissa plane] */
    const int have_s = code_signe == 0x20 || code_signe == 0x30, have_e = code_expo == 0x08 || code_expo == 0x0C;
    const uint32_t *z_s = NULL, *z_e = NULL, *z_m;
    uint64_t bits_s = 0;
    if (have_s) { const uint32_t lng = *cur++; if ((uint64_t)(lng >> 2) > max_words) goto broken; z_s = cur; bits_s = 32ull * (lng >> 2); cur += lng >> 2; }
    if (have_e) { const uint32_t lng = *cur++; if ((uint64_t)(cur - (const uint32_t *)zstream) + (lng >> 2) > max_words) goto broken; z_e = cur; cur += lng >> 2; }
    cur++;                                                       /* the mantissa length slot */
    z_m = cur;
    rle_job rj = { z_s, NULL, (int)n, bits_s, 0 };
    const uint64_t used_words = (uint64_t)(cur - (const uint32_t *)zstream);
    walk_job we = { z_e, ni, nj, (int)need_e, NULL, 0, have_e ? 32ull * (uint64_t)(z_m - 1 - z_e) : 0 };
    walk_job wm = { z_m, ni, nj, nbits, NULL, 0, 32ull * (max_words > used_words ? max_words - used_words : 0) };
    pthread_t th_s, th_e;
    int run_s = 0, run_e = 0, rc = -1;
    const int rle_host = getenv("EZHIP_A32_RLE_HOST") != NULL;
    if (have_s && rle_host) {
        rj.mask = (uint32_t *)calloc(n / 32 + 2 + 16, 4);
        if (!rj.mask) return -1;
        run_s = pthread_create(&th_s, NULL, rle_thread, &rj) == 0;
        if (!run_s) rle_thread(&rj);
    }
    if (have_e) { run_e = pthread_create(&th_e, NULL, walk_thread, &we) == 0; if (!run_e) walk_thread(&we); }
    walk_thread(&wm);                                            /* this thread walks the longest chain */
    if (run_e) pthread_join(th_e, NULL);
    if (run_s) pthread_join(th_s, NULL);
    if (have_s && !rle_host) { int bad = 0; if (sign_mask_on_device(d_smask, z_s, 0, bits_s, n, &bad)) goto out; rj.rc = bad ? -1 : 0; }
    if (have_s && rj.rc) { fprintf(stderr, "<c_armn_uncompress32> broken stream (the sign runs leave their sub-stream)\n"); goto out; }
    if ((have_e && !we.tp) || !wm.tp) { fprintf(stderr, "<c_armn_uncompress32> broken stream (a tile chain leaves the record)\n"); goto out; }
    if (have_s && rle_host && (ezhip_h2d(d_smask, rj.mask, 4 * (n / 32 + 1)) || ezhip_sync())) goto out;
    if (have_e && decode_plane(d_expo, z_e, we.tp, we.ntiles, ni, nj, (int)need_e, 0, 3)) goto out;
    if (decode_plane(d_mant, z_m, wm.tp, wm.ntiles, ni, nj, nbits, 1, 3)) goto out;
    if (packhip_a32_combine(d_fld, d_expo, d_mant, d_smask, n, nbits, exp_min, code_signe, code_expo != 0)) goto out;
    rc = (int)n;
out:
    free(rj.mask); free(we.tp); free(wm.tp);
    return rc;
broken:
    fprintf(stderr, "<c_armn_uncompress32> broken stream (a sub-stream length exceeds the field)\n");
    return -1;
}

/* ---- the same with the stream's LENGTH known (the data part of an FST record, fstd98.c:2436: its word count is data[0]).  c_armn_uncompress32 itself has
 * no length argument and the mantissa length slot of a record holds the sign stream's length (:237, reproduced), so the plain entry point above can only FIND
 * the end of the record by walking it -- on the host, where the caller's buffer is.  With the length in hand the walk is held to it, and (EZHIP_A32_DEVICE_WALK=1)
 * the planes can go up once and both chains of tile headers be followed on the device (packhip_armn_tile_walk: the speculation tables / composed tables /
 * chain kernel of armn_compress UNCOMPRESS with the plane's tile rule; only the run-length decoder of the sign plane stays on the host, in its own thread).
 * That form is NOT the default: measured on 7200 x 3601 fields (tools/probe_a32.py, profiles/r03_experiments.txt) it takes 43 - 55 ms against 22 - 25 with
 * the host walks -- a mantissa plane is a stream of 300 - 550 Mbit, two to three times a cfg5 record, the chain kernel is one CU following it at ~0.25 us
 * per dependent step, and a host core walks the same chain in ~10 ms while the other planes are walked on other threads. ---- */
static __thread uint32_t t_plane_hdr[2], t_plane_status[2];
static int decode_plane_walked_on_device(int *d_plane, const uint32_t *z, int z_on_device, size_t words, int ni, int nj, int nbits, int wide, int which)
{
    const size_t zw = words + 1;                                  /* [header word in armn_compress's layout][the plane's stream] */
    unsigned *d_zs = (unsigned *)w32(3, 4 * (zw + 64));
    void *d_work = w32(4, packhip_armn_dec_work_bytes(ni, nj, zw));
    int *d_bs = (int *)w32(5, 4 * (size_t)ni * ((size_t)(nj + 31) / 32 + 1));
    int *d_status = (int *)w32(7, 256);
    if (!d_zs || !d_work || !d_bs || !d_status) return -1;
    t_plane_hdr[which] = packhip_armn_plane_header(nbits);
    if (ezhip_h2d(d_zs, &t_plane_hdr[which], 4) || (z_on_device ? ezhip_d2d(d_zs + 1, z, 4 * words) : ezhip_h2d(d_zs + 1, z, 4 * words)) || ezhip_memset(d_zs + zw, 0, 4 * 64)) return -1;
    if (packhip_armn_tile_walk_parallel(d_zs, zw, ni, nj, d_work, d_status + which)) return -1;
    if (ezhip_d2h(&t_plane_status[which], d_status + which, 4) || ezhip_sync()) return -1;      /* (the verdict first: positions of an unresolved chain are not positions) */
    if (t_plane_status[which] != 0) return t_plane_status[which] == 1 ? 1 : -1;      /* 1: the parallel forms did not resolve the chain (the caller walks it on the host) */
    if (packhip_pg_decode2(d_plane, d_bs, d_zs + 1, NULL, (const unsigned *)d_work, ni, nj, nbits, wide) || ezhip_sync()) return -1;
    return 0;
}

static int device_walk_wanted(int ni, int nj, size_t zwords)
{
    const char *dw = getenv("EZHIP_A32_DEVICE_WALK");
    /* whole rows of tiles: composition (k_dmin_*); ragged rows: composition + the row recurrence (k_drg_*), which wants rows of a few hundred tiles */
    const int device_walk = dw ? atoi(dw) != 0 : ((size_t)ni * nj >= 65536 && ((ni - 1) % 3 == 0 || ni >= 768));
    return device_walk && (uint64_t)zwords * 32 + 16384 < (1ull << 32);         /* (the device walk holds bit positions in 32 bits) */
}
/* one word of the record: from the host's copy, or -- the record in HBM -- read back (four dependent words per record: w0, info and the two length words) */
static int rec_word(const uint32_t *z0, int z_on_device, size_t idx, uint32_t *out)
{
    if (!z_on_device) { *out = z0[idx]; return 0; }
    return (ezhip_d2h(out, z0 + idx, 4) || ezhip_sync()) ? -1 : 0;
}
/* a plane in HBM: chain and tiles on the device; when the parallel forms leave the chain unresolved -- typically the exponent plane of a smooth field, whose long
 * runs of empty tiles keep walks of different phase apart for good -- THIS plane's chain is walked on the host (zero runs at once) and only its positions go up */
static int plane_on_device(int *d_plane, const uint32_t *z, int z_on_device, size_t words, int ni, int nj, int nbits, int wide, int which)
{
    const int r = decode_plane_walked_on_device(d_plane, z, z_on_device, words, ni, nj, nbits, wide, which);
    if (r != 1) return r;
    const uint32_t *zh = z;
    uint32_t *tmp = NULL;
    if (z_on_device) {
        tmp = (uint32_t *)malloc(4 * (words + 4));
        if (!tmp) return -1;
        if (ezhip_d2h(tmp, z, 4 * words) || ezhip_sync()) { free(tmp); return -1; }
        tmp[words] = tmp[words + 1] = 0;
        zh = tmp;
    }
    size_t ntiles = 0;
    uint64_t *tp = walk_tiles(zh, ni, nj, nbits, &ntiles, 32ull * words);
    int rc = -1;
    if (tp) {
        unsigned *d_zs = (unsigned *)w32(3, 4 * (words + 1 + 64));                    /* (still [header word][the plane] from the attempt above) */
        unsigned long long *d_tp = (unsigned long long *)w32(4, 8 * (ntiles + 1));
        int *d_bs = (int *)w32(5, 4 * (size_t)ni * ((size_t)(nj + 31) / 32 + 1));
        if (d_zs && d_tp && d_bs && !ezhip_h2d(d_tp, tp, 8 * (ntiles + 1))) rc = packhip_pg_decode(d_plane, d_bs, d_zs + 1, d_tp, ni, nj, nbits, wide);
        if (ezhip_sync()) rc = -1;
    }
    free(tp); free(tmp);
    return rc;
}

int c_armn_uncompress32_lng_dev(float *d_fld, const unsigned char *zstream, size_t zbytes, int ni, int nj, int nk, int znbits)
{
    if (need_dev32("c_armn_uncompress32")) return -1;
    const size_t zwords = zbytes / 4;
    if (ni < 16 || nj < 16 || zwords < 4) return -1;
    /* the chains of tile headers on the device when every tile of a row holds nine points (ni - 1 a multiple of 3): the chain is then a pure function of the bit
     * position and resolves by composition of the windows' maps (unpack_kernels.hip, k_dmin_*) in a fraction of a millisecond per plane.  With a ragged last tile
     * per row the device can only follow the chain step by step on one CU (43 - 55 ms per 7200 x 3601 field against 21 - 24 with the host threads): host walk.
     * EZHIP_A32_DEVICE_WALK=0 / 1 forces either */
    if (device_walk_wanted(ni, nj, zwords)) {
        const int rc = uncompress32_device_walk(d_fld, (const uint32_t *)zstream, 0, zwords, ni, nj);
        if (rc != 1) return rc;                                  /* 1: a chain the device forms did not resolve (rows that rejoin late, tiny planes) */
    }
    return uncompress32_host_walk(d_fld, zstream, zwords, ni, nj, nk, znbits);
}

/* the record AND the field in HBM (a record the device compressor wrote, or one read into device memory): whole-tile rows decode without the host seeing more
 * than four words of the record; a ragged last tile per row sends the record down to the host's walk (the chain is sequential there: see above) */
int c_armn_uncompress32_zdev(float *d_fld, const void *d_zstream, size_t zbytes, int ni, int nj, int nk, int znbits)
{
    if (need_dev32("c_armn_uncompress32")) return -1;
    const size_t zwords = zbytes / 4;
    if (ni < 16 || nj < 16 || zwords < 4 || !d_zstream || !d_fld) return -1;
    if (device_walk_wanted(ni, nj, zwords)) {
        const int rc1 = uncompress32_device_walk(d_fld, (const uint32_t *)d_zstream, 1, zwords, ni, nj);
        if (rc1 != 1) return rc1;
    }
    unsigned char *h = (unsigned char *)malloc(4 * zwords + 64);
    if (!h) return -1;
    int rc = -1;
    if (!ezhip_d2h(h, d_zstream, 4 * zwords) && !ezhip_sync()) rc = uncompress32_host_walk(d_fld, h, zwords, ni, nj, nk, znbits);
    if (ezhip_sync()) rc = -1;                                    /* (the host copy is read by queued uploads until here) */
    free(h);
    return rc;
}

static int uncompress32_device_walk(float *d_fld, const uint32_t *z0, int z_on_device, size_t zwords, int ni, int nj)
{
    const size_t n = (size_t)ni * nj;
    const uint32_t *cur = z0 + 2, *zend = z0 + zwords;
    uint32_t w0, info;
    if (z_on_device) { uint32_t h2[2]; if (ezhip_d2h(h2, z0, 8) || ezhip_sync()) return -1; w0 = h2[0]; info = h2[1]; }
    else { w0 = z0[0]; info = z0[1]; }
    if ((w0 & 15u) != 5u) { fprintf(stderr, "<c_armn_uncompress32> not a PARALLELOGRAM32 stream\n"); return -1; }
    const int nbits = (int)((w0 >> 10) & 31);
    const uint32_t exp_min = info >> 16, need_e = (info >> 8) & 0xFF, codes = info & 0xFF;
    const int code_signe = (int)(codes & 0x30), code_expo = (int)(codes & 0xC), code_mant = (int)(codes & 0x3);
    if (code_mant != 0) { fprintf(stderr, "<c_armn_uncompress32> plain mantissa streams are not produced by c_armn_compress32\n"); return -1; }
    if (need_e > 8 || nbits < 1 || nbits > 23) { fprintf(stderr, "<c_armn_uncompress32> broken stream (exponent width %u, mantissa width %d)\n", need_e, nbits); return -1; }
    unsigned *d_smask = (unsigned *)w32(2, 4 * (n / 32 + 2) + 16);
    int *d_expo = (int *)w32(0, 4 * n + 16), *d_mant = (int *)w32(1, 4 * n + 16);
    if (!d_smask || !d_expo || !d_mant) return -1;
    const int have_s = code_signe == 0x20 || code_signe == 0x30, have_e = code_expo == 0x08 || code_expo == 0x0C;
    const uint32_t *z_s = NULL, *z_e = NULL, *z_m;
    size_t words_e = 0;
    uint64_t bits_s = 0;
    if (have_s) { uint32_t lng; if (cur >= zend || rec_word(z0, z_on_device, (size_t)(cur - z0), &lng)) goto broken; cur++; if ((size_t)(lng >> 2) > (size_t)(zend - cur)) goto broken; z_s = cur; bits_s = 32ull * (lng >> 2); cur += lng >> 2; }
    if (have_e) { uint32_t lng; if (cur >= zend || rec_word(z0, z_on_device, (size_t)(cur - z0), &lng)) goto broken; cur++; if ((size_t)(lng >> 2) > (size_t)(zend - cur)) goto broken; z_e = cur; words_e = lng >> 2; cur += lng >> 2; }
    if (cur + 1 >= zend) goto broken;
    cur++;                                                       /* the mantissa length slot */
    z_m = cur;
    rle_job rj = { z_s, NULL, (int)n, bits_s, 0 };
    pthread_t th_s;
    int run_s = 0, rc = -1;
    const int rle_host = getenv("EZHIP_A32_RLE_HOST") != NULL && !z_on_device;
    if (have_s && rle_host) {
        rj.mask = (uint32_t *)calloc(n / 32 + 2 + 16, 4);
        if (!rj.mask) return -1;
        run_s = pthread_create(&th_s, NULL, rle_thread, &rj) == 0;
        if (!run_s) rle_thread(&rj);
    }
    int bad = 0, unresolved = 0;
    if (have_s && !rle_host) { int sb = 0; if (sign_mask_on_device(d_smask, z_s, z_on_device, bits_s, n, &sb)) bad = 1; rj.rc = sb ? -1 : 0; }
    if (have_e) { const int r = plane_on_device(d_expo, z_e, z_on_device, words_e, ni, nj, (int)need_e, 0, 0); if (r == 1) unresolved = 1; else if (r) bad = 1; }
    if (!bad && !unresolved) { const int r = plane_on_device(d_mant, z_m, z_on_device, (size_t)(zend - z_m), ni, nj, nbits, 1, 1); if (r == 1) unresolved = 1; else if (r) bad = 1; }
    if (run_s) pthread_join(th_s, NULL);
    if (have_s && rj.rc) bad = 1;
    if (!bad && unresolved) { rc = 1; goto out; }
    if (bad) { if (!t_quiet32) fprintf(stderr, "<c_armn_uncompress32> broken stream (a tile chain leaves the record)\n"); goto out; }
    if (have_s && rle_host && (ezhip_h2d(d_smask, rj.mask, 4 * (n / 32 + 1)) || ezhip_sync())) goto out;
    if (packhip_a32_combine(d_fld, d_expo, d_mant, d_smask, n, nbits, exp_min, code_signe, code_expo != 0)) goto out;
    if (ezhip_sync()) goto out;
    rc = (int)n;
out:
    free(rj.mask);
    return rc;
broken:
    if (!t_quiet32) fprintf(stderr, "<c_armn_uncompress32> broken stream (a sub-stream length exceeds the record)\n");
    return -1;
}

int c_armn_uncompress32_lng(float *fld, const unsigned char *zstream, size_t zbytes, int ni, int nj, int nk, int znbits)
{
    if (need_dev32("c_armn_uncompress32")) return -1;
    const size_t n = (size_t)ni * nj;
    float *d_f = (float *)w32(6, 4 * n);
    if (!d_f) return -1;
    int rc = c_armn_uncompress32_lng_dev(d_f, zstream, zbytes, ni, nj, nk, znbits);
    if (rc > 0 && (ezhip_d2h(fld, d_f, 4 * n) || ezhip_sync())) rc = -1;
    ezhip_sync();
    return rc;
}

int c_armn_uncompress32(float *fld, unsigned char *zstream, int ni, int nj, int nk, int znbits)
{
    if (need_dev32("c_armn_uncompress32")) return -1;
    const size_t n = (size_t)ni * nj;
    float *d_f = (float *)w32(6, 4 * n);
    if (!d_f) return -1;
    int rc = c_armn_uncompress32_dev(d_f, zstream, ni, nj, nk, znbits);
    if (rc > 0 && (ezhip_d2h(fld, d_f, 4 * n) || ezhip_sync())) rc = -1;
    ezhip_sync();
    return rc;
}
/* Fortran twins (armn_compress_32.c:53-56, :280-283) */
int armn_compress32_(unsigned char *zstream, float *fld, int *ni, int *nj, int *nk, int *nbits) { return c_armn_compress32(zstream, fld, *ni, *nj, *nk, *nbits); }
int armn_uncompress32_(float *fld, unsigned char *zstream, int *ni, int *nj, int *nk, int *nbits) { return c_armn_uncompress32(fld, zstream, *ni, *nj, *nk, *nbits); }
