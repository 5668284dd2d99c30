/*
 * oracle/orc_armn32.c -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * PARITY UNPINNED: src/compresseur/armn_compress_32.c #includes <App.h> (the un-vendored App submodule), so the reference file
 * cannot be compiled here; the reference holds no vectors for it.  What pins this restatement: compress -> uncompress round trips
 * (the decoders are restated from the reference's decoders, an independent code path), hand-checked headers, and the shared
 * stuff / extract semantics already pinned for the 16-bit compressor (tests/test_oracle_armn32.py).
 *
 * Restatement of the IEEE-32 compressor c_armn_compress32 / c_armn_uncompress32 (datyp 133):
 *   stream = [_fstzip word][zieee_info word]{[lng][sign stream]}{[lng][exponent stream]}[lng][mantissa stream]
 *   sign:     pack1bitRLE / unpack1bitRLE                      armn_compress_32.c:827-901, :904-955
 *   exponent: packTokensParallelogram_8 / unpack..._8          :642-750, :753-824  (container 4, 8-bit tokens)
 *   mantissa: packTokensParallelogram32 / unpack...32          :456-565, :568-639  (container 5, tokens of znbits - 9 bits)
 *   driver:   c_armn_compress32 :59-275, c_armn_uncompress32 :285-437, c_fstzip32 :440-462
 * Quirks reproduced on purpose: the mantissa length slot receives lng_signe (:237); the 3-bit container field is written with
 * `istep` as its width; nbits_needed of the mantissa tiles comes from the exponent of (float)local_max (:524-525).
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* MSB-first writer / reader with the flush rule of the stuff / extract macros (include/bitPacking.h:59-139) */
typedef struct { uint32_t *cur, last; int space; } a32_w;
static void w_init(a32_w *w, uint32_t *z) { w->cur = z; w->last = 0; w->space = 32; *z = 0; }
static void w_put(a32_w *w, uint32_t tok, int bits)
{
    if (w->space >= bits) { w->last = (bits == 32 ? 0 : (w->last << bits)) | tok; w->space -= bits; }
    else {
        *w->cur = (w->space ? (w->last << w->space) : w->last) | (tok >> (bits - w->space));
        int rest = bits - w->space;
        w->last = tok & (rest == 32 ? 0xFFFFFFFFu : ((1u << rest) - 1u));
        w->space = 32 - rest;
        w->cur++;
    }
}
typedef struct { const uint32_t *cur; uint32_t word; int left; } a32_r;
static void r_init(a32_r *r, const uint32_t *z) { r->cur = z; r->word = *z; r->left = 32; }
static uint32_t r_get(a32_r *r, int bits)
{
    uint32_t tok;
    if (r->left >= bits) { tok = r->word >> (32 - bits); r->word = bits == 32 ? 0 : r->word << bits; r->left -= bits; }
    else {
        tok = r->left ? r->word >> (32 - bits) : 0;
        r->cur++; r->word = *r->cur;
        int rest = bits - r->left;
        tok |= r->word >> (32 - rest);
        r->word = rest == 32 ? 0 : r->word << rest;
        r->left = 32 - rest;
    }
    if (r->left == 0) { r->cur++; r->word = *r->cur; r->left = 32; }
    return tok;
}

static int float_bitlen(int v)      /* :524-525: exponent of (float)v, minus 126 */
{
    union { float f; int32_t i; } r; r.f = (float)v;
    return (r.i >> 23) - 126;
}

/* Lorenzo differences from the ORIGINAL values, first row / column zero (:479-496, :670-687) */
static int *lorenzo_u32(const uint32_t *u, int ni, int nj)
{
    int *d = (int *)calloc((size_t)ni * nj, sizeof(int));
    for (int j = 1; j < nj; j++) for (int i = 1; i < ni; i++) {
        size_t k = (size_t)j * ni + i;
        d[k] = (int)(u[k] - (u[k - ni] + u[k - 1] - u[k - 1 - ni]));
    }
    return d;
}

/* the common body of packTokensParallelogram_8 (container 4, bitlen widths) and packTokensParallelogram32 (container 5, float
 * widths, capacity check :546-549).  Returns zlng (0: the 32-bit form ran out of space) */
static int pack_parallelogram(uint32_t *z, const uint32_t *u, int ni, int nj, int nbits, int container, int wide, int remaining_space)
{
    int *d = lorenzo_u32(u, ni, nj);
    a32_w w; w_init(&w, z);
    w_put(&w, (uint32_t)container, 3);
    for (int i = 0; i < ni; i++) w_put(&w, u[i], nbits);
    for (int j = 1; j < nj; j++) w_put(&w, u[(size_t)j * ni], nbits);
    for (int j0 = 1; j0 < nj; j0 += 3) {
        const int tn = nj - j0 < 3 ? nj - j0 : 3;
        for (int i0 = 1; i0 < ni; i0 += 3) {
            const int tm = ni - i0 < 3 ? ni - i0 : 3;
            int mx = d[(size_t)j0 * ni + i0];
            for (int n = 0; n < tn; n++) for (int m = 0; m < tm; m++) { int a = abs(d[(size_t)(j0 + n) * ni + i0 + m]); if (mx < a) mx = a; }
            int need;
            if (mx == 0) need = 0;
            else if (wide) need = float_bitlen(mx);
            else { need = 0; for (int v = mx; v; v >>= 1) need++; }       /* fastlog[] table = bit length, :709-714 */
            w_put(&w, (uint32_t)need, container);
            if (need) {
                const int nb2 = need + 1;
                for (int n = 0; n < tn; n++) for (int m = 0; m < tm; m++)
                    w_put(&w, (uint32_t)d[(size_t)(j0 + n) * ni + i0 + m] & (nb2 == 32 ? 0xFFFFFFFFu : ((1u << nb2) - 1u)), nb2);
                if (wide && remaining_space < (int)(w.cur - z) + (1 + ((need + 9 * nbits) >> 5))) { free(d); return 0; }
            }
        }
    }
    w_put(&w, 0, 16); w_put(&w, 0, 16);
    free(d);
    return 1 + (int)(w.cur - z) * 4;
}

static void unpack_parallelogram(uint32_t *u, const uint32_t *z, int ni, int nj, int nbits, int elem_bits)
{
    int *t = (int *)calloc((size_t)ni * nj, sizeof(int));
    a32_r r; r_init(&r, z);
    const int container = (int)r_get(&r, 3);
    for (int i = 0; i < ni; i++) u[i] = r_get(&r, nbits);
    for (int j = 1; j < nj; j++) u[(size_t)j * ni] = r_get(&r, nbits);
    for (int j0 = 1; j0 < nj; j0 += 3) {
        const int tn = nj - j0 < 3 ? nj - j0 : 3;
        for (int i0 = 1; i0 < ni; i0 += 3) {
            const int tm = ni - i0 < 3 ? ni - i0 : 3;
            const int need = (int)r_get(&r, container);
            for (int n = 0; n < tn; n++) for (int m = 0; m < tm; m++) {
                int v = 0;
                if (need) { const int nb2 = need + 1; v = (int)r_get(&r, nb2); v = (int)((uint32_t)v << (32 - nb2)) >> (32 - nb2); }
                t[(size_t)(j0 + n) * ni + i0 + m] = v;
            }
        }
    }
    const uint32_t mask = elem_bits == 32 ? 0xFFFFFFFFu : ((1u << elem_bits) - 1u);      /* the _8 form stores unsigned chars */
    for (int j = 1; j < nj; j++) for (int i = 1; i < ni; i++) {
        size_t k = (size_t)j * ni + i;
        u[k] = ((uint32_t)t[k] + (u[k - ni] + u[k - 1] - u[k - 1 - ni])) & mask;
    }
    free(t);
}

/* pack1bitRLE, :827-901 (ufld[npts] is read but never decides anything) */
static int pack_rle(uint32_t *z, const unsigned char *s, int npts)
{
    a32_w w; w_init(&w, z);
    int last = 0, idx = 1;
    while (idx <= npts) {
        while (idx < npts && s[idx] == s[last]) idx++;
        int count = idx - last;
        if (count < 8) {
            w_put(&w, 0, 1);
            int lim = last + 7 > npts ? npts - last : 7;
            for (int i = 0; i < lim; i++) w_put(&w, s[last + i], 1);
            last += 7; idx = last + 1;
        } else {
            int i = 0, repeat = 0;
            while (i < count) {
                int c = (count - i) >= 63 ? 62 : (count - i);
                if (c < 8) {
                    w_put(&w, 0, 1);
                    int lim = last + 7 > npts ? npts - last : 7;
                    for (int j = 0; j < lim; j++) w_put(&w, s[last + j], 1);
                    last += 7; idx = last + 1;
                } else if (c == 62 && (count - i) > 256 && repeat == 1) {
                    c = 0xFF;
                    w_put(&w, 0xFF, 8);
                    last += c; idx = last + 1;
                } else {
                    w_put(&w, 1, 1); w_put(&w, s[last], 1); w_put(&w, (uint32_t)c, 6);
                    last += c; idx = last + 1;
                    if (c == 62) repeat = 1;
                }
                i += c;
            }
        }
    }
    w_put(&w, 0, 16); w_put(&w, 0, 16);
    return 1 + (int)(w.cur - z) * 4;
}
/* unpack1bitRLE, :904-955 */
static void unpack_rle(unsigned char *s, const uint32_t *z, int npts)
{
    a32_r r; r_init(&r, z);
    uint32_t last_val = 0xFFFFFFFFu;
    int i = 0;
    while (i < npts) {
        if (r_get(&r, 1) == 0) {
            int lim = i + 7 > npts ? npts - i : 7;
            for (int j = 0; j < lim; j++) s[i + j] = (unsigned char)r_get(&r, 1);
            i += lim;
        } else {
            uint32_t val = r_get(&r, 1); int count = (int)r_get(&r, 6);
            if (count == 63) { for (int j = 0; j < 255 && i + j < npts; j++) s[i + j] = (unsigned char)last_val; i += 255; }
            else { for (int j = 0; j < count && i + j < npts; j++) s[i + j] = (unsigned char)val; i += count; last_val = val; }
        }
    }
}

/* c_armn_compress32, :59-275.  zstream must hold ni*nj*znbits/8 bytes (+ slack).  Returns the byte count or -1. */
int orc_armn_compress32(unsigned char *zstream, const float *fld, int ni, int nj, int nk, int znbits)
{
    (void)nk;
    if (ni < 16 || nj < 16) return -1;
    const int nbits = znbits - 9;
    const unsigned npts = (unsigned)ni * nj;
    unsigned char *signe = (unsigned char *)calloc(2 * (size_t)npts, 1);
    uint32_t *expo = (uint32_t *)malloc(sizeof(uint32_t) * npts), *mant = (uint32_t *)malloc(sizeof(uint32_t) * npts);
    uint32_t s_or = 0, s_and = 0xFFFFFFFFu;
    for (unsigned i = 0; i < npts; i++) {
        uint32_t t; memcpy(&t, &fld[i], 4);
        signe[i] = (unsigned char)(t >> 31); s_or |= t; s_and &= t;
        expo[i] = (t >> 23) & 0xFF;
        mant[i] = t & 0x7FFFFF;
        if (nbits < 23) mant[i] >>= (23 - nbits);
    }
    const int meme_signe = (s_or >> 31) == (s_and >> 31);
    uint32_t emin = expo[0], emax = expo[0];
    for (unsigned i = 0; i < npts; i++) { if (expo[i] < emin) emin = expo[i]; if (expo[i] > emax) emax = expo[i]; }
    const uint32_t exp_base = emin;
    emax -= emin;
    for (unsigned i = 0; i < npts; i++) expo[i] -= exp_base;
    const int need_e = emax == 0 ? 0 : float_bitlen((int)emax);
    unsigned char *p = zstream + 8, *pos_lng_signe = NULL;
    uint32_t lng_signe = 0, lng_expo = 0, lng_mant;
    unsigned char code_signe, code_expo;
    if (meme_signe) code_signe = (s_or >> 31) ? 0x10 : 0x00;
    else {
        pos_lng_signe = p; p += 4;
        lng_signe = (uint32_t)pack_rle((uint32_t *)p, signe, (int)npts);
        code_signe = 0x20;
    }
    /* (:164-168: a plain bit mask would replace an RLE stream longer than npts / 4 bytes; the RLE never is: at most 8 bits per 7 points) */
    if (lng_signe % 4) lng_signe += 4 - lng_signe % 4;
    p += lng_signe;
    if (pos_lng_signe) memcpy(pos_lng_signe, &lng_signe, 4);
    unsigned char *pos_lng_expo = p;
    if (need_e == 0) code_expo = 0x00;
    else {
        code_expo = 0x08;
        p += 4;
        lng_expo = (uint32_t)pack_parallelogram((uint32_t *)p, expo, ni, nj, need_e, 4, 0, 0);
        if (lng_expo > npts) { free(signe); free(expo); free(mant); return -1; }
        if (lng_expo % 4) lng_expo += 4 - lng_expo % 4;
        memcpy(pos_lng_expo, &lng_expo, 4);
        p += lng_expo;
    }
    unsigned char *pos_lng_mant = p;
    p += 4;
    int remaining = (int)(((long long)ni * nj * znbits) / 32) - (int)((p - zstream) / 4);
    lng_mant = (uint32_t)pack_parallelogram((uint32_t *)p, mant, ni, nj, nbits, 5, 1, remaining);
    free(signe); free(expo); free(mant);
    if (lng_mant == 0) return -1;
    if (lng_mant % 4) lng_mant += 4 - lng_mant % 4;
    memcpy(pos_lng_mant, &lng_signe, 4);                                        /* sic, :237 */
    const uint32_t w0 = 5u | 1u << 4 | 3u << 7 | ((uint32_t)nbits & 31u) << 10 | 1u << 15 | 2u << 18;      /* _fstzip: PARALLELOGRAM32, degree 1, step 3, nbits, levels 1, version 2 */
    memcpy(zstream, &w0, 4);
    const uint32_t info = (exp_base & 0xFF) << 16 | ((uint32_t)need_e & 0xFF) << 8 | (uint32_t)(code_signe | code_expo);
    memcpy(zstream + 4, &info, 4);
    p += lng_mant;
    return (int)(p - zstream);
}

/* c_armn_uncompress32, :285-437.  Returns ni * nj. */
int orc_armn_uncompress32(float *fld, const unsigned char *zstream, int ni, int nj, int nk, int znbits)
{
    (void)nk; (void)znbits;
    const unsigned npts = (unsigned)ni * nj;
    const uint32_t *cur = (const uint32_t *)zstream;
    const uint32_t w0 = cur[0], info = cur[1];
    cur += 2;
    const int nbits = (int)((w0 >> 10) & 31);
    const uint32_t exp_min = info >> 16, need_e = (info >> 8) & 0xFF, codes = info & 0xFF;
    const uint32_t code_signe = codes & 0x30, code_expo = codes & 0xC, code_mant = codes & 0x3;
    unsigned char *signe = (unsigned char *)calloc(2 * (size_t)npts + 256, 1);
    uint32_t *expo = (uint32_t *)malloc(sizeof(uint32_t) * npts), *mant = (uint32_t *)malloc(sizeof(uint32_t) * npts);
    if (code_signe == 0x20 || code_signe == 0x30) {
        uint32_t lng = *cur++;
        unpack_rle(signe, cur, (int)npts);
        cur += lng >> 2;
    } else memset(signe, code_signe == 0x00 ? 0 : 1, npts);
    if (code_expo == 0x08 || code_expo == 0x0C) {
        uint32_t lng = *cur++;
        unpack_parallelogram(expo, cur, ni, nj, (int)need_e, 8);
        cur += lng >> 2;
        for (unsigned i = 0; i < npts; i++) expo[i] = (expo[i] + exp_min) & 0xFF;
    } else for (unsigned i = 0; i < npts; i++) expo[i] = exp_min;
    cur++;                                                                       /* the mantissa length slot (holds lng_signe) */
    if (code_mant == 0) unpack_parallelogram(mant, cur, ni, nj, nbits, 32);
    else { a32_r r; r_init(&r, cur); for (unsigned i = 0; i < npts; i++) mant[i] = r_get(&r, nbits); }
    for (unsigned i = 0; i < npts; i++) {
        uint32_t t = (uint32_t)signe[i] << 31 | expo[i] << 23 | mant[i] << (23 - nbits);
        memcpy(&fld[i], &t, 4);
    }
    free(signe); free(expo); free(mant);
    return (int)npts;
}
