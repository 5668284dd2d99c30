/*
 * oracle/orc_pack.c -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * PARITY UNPINNED: the reference's packers (src/packers/compact.c + compact.tmplc,
 * compact_integer.c, float_packer.c) and compressor (src/compresseur/c_zfstlib.c) #include <App.h>
 * from the un-vendored App submodule, so they cannot be compiled here without writing stand-ins,
 * and the reference holds no golden vectors for them.  This file restates the
 * algorithms from the reference sources line by line; the pins available are (a) the survey's probe
 * header for a cfg5-like field, (b) encode->decode round trips through the restated reference
 * DECODERS (an independent code path of the reference), (c) hand-computed known answers
 * (tests/test_oracle_packers.py).
 *
 * Streams are arrays of host-endian uint32 words filled MSB-first by the `stuff` macro
 * (include/bitPacking.h:126-139).
 */
#include "orc_pack.h"
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------------------------------------
 * MSB-first bit writer / reader: include/bitPacking.h stuff (:126-139), extract (:59-82), discard (:95-112)
 * ------------------------------------------------------------------------------------------- */
typedef struct { uint32_t *ptr; uint32_t last; int space; } bitw;

static void bw_init(bitw *w, uint32_t *arr, int offset_bits)
{   /* compact.tmplc:292-300 : keep the bits that precede the offset in the first word */
    w->space = 32 - (offset_bits % 32);
    w->ptr = arr + (offset_bits / 32);
    w->last = (w->space == 32) ? 0 : (*w->ptr >> w->space);
}
static void bw_put(bitw *w, uint32_t token, int nbits)
{
    if (w->space >= nbits) {
        w->last = (nbits == 32) ? token : ((w->last << nbits) | token);
        w->space -= nbits;
    } else {
        int rem = nbits - w->space;
        *w->ptr = (w->last << w->space) | (rem >= 32 ? 0 : (token >> rem));   /* space <= 31 here */
        w->last = token;                    /* `token & (-1 >> k)` keeps every bit: -1 >> k is still -1 */
        w->space = 32 - rem;
        w->ptr++;
    }
}
static void bw_flush(bitw *w)
{   /* compact.tmplc:329-332 : read-modify-write of the last partial word */
    if (w->space < 32) {
        uint32_t keep = (w->space == 0) ? 0 : (*w->ptr & ~(0xFFFFFFFFu << w->space));
        *w->ptr = (w->space == 0 ? w->last : (w->last << w->space)) | keep;
    }
}

typedef struct { const uint32_t *ptr; uint32_t word; int left; } bitr;
static void br_init(bitr *r, const uint32_t *arr, int offset_bits)
{
    r->left = 32 - (offset_bits % 32);
    r->ptr = arr + (offset_bits / 32);
    r->word = (r->left == 32) ? *r->ptr : (*r->ptr << (32 - r->left));
}
static uint32_t br_get(bitr *r, int nbits)
{
    uint32_t tok;
    if (r->left >= nbits) {
        tok = r->word >> (32 - nbits);
        r->word = (nbits == 32) ? 0 : (r->word << nbits);
        r->left -= nbits;
    } else {
        tok = r->word >> (32 - nbits);
        r->ptr++;
        r->word = *r->ptr;
        int need = nbits - r->left;
        tok |= r->word >> (32 - need);
        r->word <<= need;
        r->left = 32 - need;
    }
    if (r->left == 0) { r->ptr++; r->word = *r->ptr; r->left = 32; }
    return tok;
}

/* ---------------------------------------------------------------------------------------------
 * compact_float, src/packers/compact.tmplc:37-431 (FLOAT_TYPE = float)
 * ------------------------------------------------------------------------------------------- */
typedef union { double d; uint64_t u; } dbits;

#define ORC_FT float
#define ORC_COMPACT_NAME orc_compact_float
#include "orc_compact_tmpl.h"
#undef ORC_COMPACT_NAME
#undef ORC_FT
#define ORC_FT double
#define ORC_COMPACT_NAME orc_compact_double
#include "orc_compact_tmpl.h"
#undef ORC_COMPACT_NAME
#undef ORC_FT

/* ---------------------------------------------------------------------------------------------
 * compact_integer, src/packers/compact_integer.c:325-570
 * ------------------------------------------------------------------------------------------- */
int orc_compact_integer(void *unpacked, void *packedHeader, void *packed, int elementCount,
                        int bitSizeOfPackedToken, int off_set, int stride, int opCode)
{
    uint32_t *au = (uint32_t *)unpacked;
    int32_t *as = (int32_t *)unpacked;
    uint32_t *hdr = (uint32_t *)packedHeader, *out = (uint32_t *)packed;
    int n = elementCount;
    if (bitSizeOfPackedToken == 0) return 0;
    uint32_t minU = 0, maxU = 0; int32_t minS = 0, maxS = 0;
    int shiftRequired = 0;
    if (opCode == 1 || opCode == 3) {
        if (hdr != NULL) {                                                           /* constructHeader :186-231 */
            uint32_t maxRange;
            if (opCode == 1) {
                minU = maxU = au[0];
                for (size_t i = (size_t)stride; i < (size_t)n * stride; i += stride) { if (au[i] < minU) minU = au[i]; else if (au[i] > maxU) maxU = au[i]; }
                maxRange = maxU - minU;
            } else {
                minS = maxS = as[0];
                for (size_t i = (size_t)stride; i < (size_t)n * stride; i += stride) { if (as[i] < minS) minS = as[i]; else if (as[i] > maxS) maxS = as[i]; }
                if (maxS > 0 && minS < 0) maxRange = (uint32_t)maxS + (uint32_t)(-minS); else maxRange = (uint32_t)(maxS - minS);
            }
            int bitRequired = 0;
            while (maxRange != 0) { maxRange >>= 1; bitRequired++; }
            if (bitSizeOfPackedToken == -1) bitSizeOfPackedToken = bitRequired;
            else while ((bitRequired - bitSizeOfPackedToken) > 0) { shiftRequired++; bitRequired--; }
            hdr[0] = 0xFD000000u | ((uint32_t)shiftRequired << 6) | (uint32_t)bitSizeOfPackedToken;
            hdr[1] = (uint32_t)n;
            hdr[2] = opCode == 1 ? minU : (uint32_t)minS;
            hdr[3] = opCode == 1 ? maxU : (uint32_t)maxS;
        } else if (bitSizeOfPackedToken == -1) {                                     /* :446-489 */
            uint32_t maxSpan;
            if (opCode == 3) {
                minS = maxS = as[0];
                for (size_t i = (size_t)stride; i < (size_t)n * stride; i += stride) { if (as[i] < minS) minS = as[i]; else if (as[i] > maxS) maxS = as[i]; }
                maxSpan = (uint32_t)((abs(minS) > maxS) ? abs(minS) : maxS);
            } else {
                maxSpan = au[0];
                for (size_t i = (size_t)stride; i < (size_t)n * stride; i += stride) maxSpan |= au[i];
            }
            bitSizeOfPackedToken = 0;
            while (maxSpan != 0) { maxSpan >>= 1; bitSizeOfPackedToken++; }
            if (opCode == 3) bitSizeOfPackedToken++;
        }
        uint32_t cleanupMask = 0xFFFFFFFFu >> (32 - bitSizeOfPackedToken);
        int32_t positiveMask = (opCode < 3) ? 0 : (int32_t)(1u << (bitSizeOfPackedToken - 1));
        bitw w;
        bw_init(&w, out, off_set);
        if (w.space == 32 && bitSizeOfPackedToken == 32) {                           /* pack32Bit :69-90 */
            uint32_t *p = w.ptr;
            for (size_t i = 0; i < (size_t)n * stride; i += stride)
                *p++ = hdr ? (opCode == 1 ? (au[i] - minU) >> shiftRequired : (uint32_t)((as[i] - minS) >> shiftRequired))
                           : au[i] + (uint32_t)positiveMask;
            return bitSizeOfPackedToken;
        }
        for (size_t i = 0; i < (size_t)n * stride; i += stride) {                    /* packBit :101-126 */
            uint32_t t;
            if (hdr) t = opCode == 1 ? (au[i] - minU) >> shiftRequired : (uint32_t)((as[i] - minS) >> shiftRequired);
            else t = au[i] + (uint32_t)positiveMask;
            bw_put(&w, t & cleanupMask, bitSizeOfPackedToken);
        }
        bw_flush(&w);
        return bitSizeOfPackedToken;
    }
    if (opCode == 2 || opCode == 4) {                                                /* Unpack :245-303 */
        int tokenSize, shift; uint32_t minv;
        if (hdr != NULL) { tokenSize = hdr[0] & 0x3F; shift = (hdr[0] >> 6) & 0x3F; n = (int)hdr[1]; minv = hdr[2]; }
        else { tokenSize = bitSizeOfPackedToken; shift = 0; minv = 0; }
        int32_t positiveMask = (opCode < 3) ? 0 : (int32_t)(1u << (bitSizeOfPackedToken - 1));
        positiveMask = -positiveMask;
        int sig = tokenSize > 32 ? 32 : tokenSize, insig = tokenSize > 32 ? tokenSize - 32 : 0;
        bitr r;
        br_init(&r, out, off_set);
        for (size_t i = 0; i < (size_t)n * stride; i += stride) {
            uint32_t t = br_get(&r, sig);
            if (insig > 0) (void)br_get(&r, insig);
            au[i] = hdr ? (t << shift) + minv : t + (uint32_t)positiveMask;
        }
        return bitSizeOfPackedToken;
    }
    return 0;
}

/* ---------------------------------------------------------------------------------------------
 * c_float_packer / c_float_unpacker, src/packers/float_packer.c:128-202, 258-283, 40-113
 * ------------------------------------------------------------------------------------------- */
typedef union { float f; int32_t i; } fint;

static int32_t aligned_mantissa(int32_t src, int32_t maxExp)
{
    int32_t mantis = (1 << 23) | (0x7FFFFF & src);
    int32_t e = (src >> 23) & 0xFF;
    int32_t shift = maxExp - e;
    if (shift > 31) shift = 31;
    mantis = mantis >> shift;
    if (src >> 31) mantis = -mantis;
    return mantis;
}

int32_t orc_float_packer(float *source, int32_t nbits, int32_t *header, int32_t *stream, int32_t npts)
{
    if (nbits > 16 || nbits < 1) return -1;
    header[2] = npts;
    header[0] = (int32_t)(0xEFFu << 20);
    header[0] = header[0] | ((nbits - 1) << 16);
    fint fmin, fmax;
    fmin.f = fmax.f = source[0];
    for (int32_t k = 0; k < npts; k++) {
        fmin.f = fmin.f > source[k] ? source[k] : fmin.f;
        fmax.f = fmax.f < source[k] ? source[k] : fmax.f;
    }
    int32_t MaxExp = (fmax.i >> 23) & 0xFF, Exp = (fmin.i >> 23) & 0xFF;
    MaxExp = MaxExp > Exp ? MaxExp : Exp;
    int32_t Maximum = aligned_mantissa(fmax.i, MaxExp);
    if (((fmax.i >> 23) & 0xFF) < 1) Maximum = 0;
    int32_t Minimum = aligned_mantissa(fmin.i, MaxExp);
    if (((fmin.i >> 23) & 0xFF) < 1) Minimum = 0;
    Maximum = Maximum - Minimum;
    int32_t Shift2 = 0, Round = 1, Mask = ~(-1 << nbits);
    while (Maximum > Mask) { Maximum = Maximum >> 1; Round = Round << 1; Shift2++; }
    Round = Round >> 1;
    header[1] = Minimum;
    header[0] = header[0] | ((MaxExp & 0xFF) << 8) | (Shift2 & 0xFF);
    uint32_t Accu = 0;      /* the reference leaves Accu uninitialised; its first value is shifted out (:196) */
    int32_t Store = 0;
    const int32_t *src = (const int32_t *)source;
    for (int32_t k = 0; k < npts; k++) {
        int32_t Mantis = aligned_mantissa(src[k], MaxExp);
        Mantis = Mantis - Minimum;
        Mantis = Mantis + Round;
        Mantis = Mantis >> Shift2;
        if (Mantis > Mask) Mantis = Mask;
        Accu = (Accu << 16) | (uint32_t)Mantis;
        if (Store) *stream++ = (int32_t)Accu;
        Store ^= 1;
    }
    if (Store) *stream++ = (int32_t)(Accu << 16);
    return 0;
}

/* float_unpacker_1, src/packers/float_packer.c:40-113 */
int32_t orc_float_unpacker(float *dest, int32_t *header, int32_t *stream, int32_t npts, int32_t *nbits)
{
    *nbits = ((header[0] >> 16) & 0xF) + 1;
    if (0xEFF != ((header[0] >> 20) & 0xFFF)) return -1;
    if (npts != header[2]) return -1;
    int32_t MaxExp = (header[0] >> 8) & 0xFF, Shift2 = header[0] & 0xFF, Minimum = header[1];
    if (MaxExp == 0) { for (int32_t k = 0; k < npts; k++) dest[k] = 0.0f; return 0; }
    for (int32_t k = 0; k < npts; k++) {
        uint32_t word = (uint32_t)stream[k >> 1];
        int32_t tok = (k & 1) ? (word & 0xFFFF) : (word >> 16);
        int32_t mantis = (tok << Shift2) + Minimum;
        int32_t sgn = 0;
        if (mantis < 0) { sgn = 1; mantis = -mantis; }
        if (mantis > 0xFFFFFF) mantis = 0xFFFFFF;
        fint t, t2;
        t.i = (mantis & (~(-1 << 23))) | (MaxExp << 23);
        t.i = t.i | (sgn << 31);
        if (mantis & (1 << 23)) dest[k] = t.f;                       /* hidden 1 is genuine */
        else {                                                        /* subtract the bogus hidden 1 (:93-104) */
            t2.i = MaxExp << 23;
            t2.i = t2.i | (sgn << 31);
            t2.i = t2.i & (~((mantis << 8) >> 31));
            dest[k] = t.f - t2.f;
        }
    }
    return 0;
}

/* ---------------------------------------------------------------------------------------------
 * armn_compress, src/compresseur/c_zfstlib.c:67-203 ; packTokensMinimum :469-588 ;
 * packTokensParallelogram :646-789 ; decoders :590-643, :792-872
 * ------------------------------------------------------------------------------------------- */
static int bitlen_u(uint32_t v) { int n = 0; while (v) { v >>= 1; n++; } return n; }   /* == the fastlog table / log() forms */

static int g_level = -1;
void orc_armn_compress_setlevel(int level) { g_level = level; }

/* tokens are given as an explicit ushort array (the reference obtains it by swapping 16-bit halves of
 * the packed words on little-endian hosts, :119-126) */
static int pack_parallelogram(uint32_t *z, const uint16_t *u, int ni, int nj, int nbits, int istep, uint32_t header)
{
    int32_t *d = (int32_t *)calloc((size_t)ni * nj, sizeof(int32_t));
    for (int j = 2; j <= nj; j++)
        for (int i = 2; i <= ni; i++) {
            size_t k = (size_t)ni * (j - 1) + (i - 1);
            d[k] = (int)u[k] - ((int)u[k - ni] + (int)u[k - 1] - (int)u[k - 1 - ni]);
        }
    uint32_t container = 4;
    if (nbits >= 15)
        for (size_t k = 0; k < (size_t)ni * nj; k++) if (65535 < abs(d[k])) { container = 5; break; }
    bitw w;
    z[0] = header; z[1] = 0;
    w.ptr = z + 1; w.last = 0; w.space = 32;
    bw_put(&w, container, 3);
    for (int i = 1; i <= ni; i++) bw_put(&w, u[i - 1], nbits);
    for (int j = 2; j <= nj; j++) bw_put(&w, u[(size_t)ni * (j - 1)], nbits);
    for (int j = 2; j <= nj; j += istep) {
        int lcl_n = ((j + istep - 1) >= nj ? nj - j : istep - 1);
        for (int i = 2; i <= ni; i += istep) {
            int lcl_m = ((i + istep - 1) >= ni ? ni - i : istep - 1);
            int local_max = d[(size_t)ni * (j - 1) + (i - 1)];
            for (int n = 0; n <= lcl_n; n++)
                for (int m = 0; m <= lcl_m; m++) {
                    int v = abs(d[(size_t)ni * (j + n - 1) + (i + m - 1)]);
                    if (local_max < v) local_max = v;
                }
            uint32_t nb = local_max == 0 ? 0 : (uint32_t)bitlen_u((uint32_t)local_max);
            if (nb == 16) nb = 15;
            bw_put(&w, nb, (int)container);
            if (nb == 0) continue;
            int width = (nb == 15) ? 17 : (int)nb + 1;
            for (int n = 0; n <= lcl_n; n++)
                for (int m = 0; m <= lcl_m; m++) {
                    uint32_t tok = (uint32_t)d[(size_t)ni * (j + n - 1) + (i + m - 1)] & ~(0xFFFFFFFFu << width);
                    bw_put(&w, tok, width);
                }
        }
    }
    bw_put(&w, 0, 16); bw_put(&w, 0, 16);
    free(d);
    return 1 + (int)(w.ptr - z) * 4;
}

static int pack_minimum(uint32_t *z, const uint16_t *u, int ni, int nj, int nbits, int istep, uint32_t header)
{
    bitw w;
    z[0] = header; z[1] = 0;
    w.ptr = z + 1; w.last = 0; w.space = 32;
    for (int j = 1; j <= nj; j += istep) {
        int lcl_n = ((j + istep - 1) >= nj ? nj - j : istep - 1);
        for (int i = 1; i <= ni; i += istep) {
            int lcl_m = ((i + istep - 1) >= ni ? ni - i : istep - 1);
            uint32_t lo = u[(size_t)ni * (j - 1) + (i - 1)], hi = lo;
            for (int n = 0; n <= lcl_n; n++)
                for (int m = 0; m <= lcl_m; m++) {
                    uint32_t v = u[(size_t)ni * (j + n - 1) + (i + m - 1)];
                    if (lo > v) lo = v;
                    if (hi < v) hi = v;
                }
            uint32_t nb = (hi - lo) == 0 ? 0 : (uint32_t)bitlen_u(hi - lo);
            if (nb == 16) nb = 15;
            bw_put(&w, nb, 4);
            if (nb == 0) { bw_put(&w, lo, nbits); continue; }
            if (nb != 15) bw_put(&w, lo, nbits);
            for (int n = 0; n <= lcl_n; n++)
                for (int m = 0; m <= lcl_m; m++) {
                    uint32_t v = u[(size_t)ni * (j + n - 1) + (i + m - 1)];
                    if (nb == 15) bw_put(&w, v, 16); else bw_put(&w, v - lo, (int)nb);
                }
        }
    }
    bw_put(&w, 0, 16); bw_put(&w, 0, 16);
    return 1 + (int)(w.ptr - z) * 4;
}

static uint32_t zip_header(int method, int degree, int step, int nbits)
{   /* _fstzip, zfstlib.h:47-53, little-endian bit-field layout */
    return (uint32_t)method | (uint32_t)degree << 4 | (uint32_t)step << 7 | ((uint32_t)nbits & 31u) << 10 | 1u << 15;
}

/* out-of-place core used by the tests: tokens[ni*nj] -> z words; returns zlng (bytes) */
int orc_armn_encode(uint32_t *z, const uint16_t *tokens, int ni, int nj, int nbits)
{
    int level = g_level == -1 ? 1 : g_level;
    if (level == 0 || ni < 16 || nj < 16 || nbits <= 4) return pack_minimum(z, tokens, ni, nj, nbits, 5, zip_header(3, 0, 5, nbits));
    return pack_parallelogram(z, tokens, ni, nj, nbits, 3, zip_header(4, 1, 3, nbits));
}

int orc_armn_decode(uint16_t *tokens, const uint32_t *z, int ni, int nj)
{
    uint32_t h = z[0];
    int method = h & 15, istep = (h >> 7) & 7, nbits = (h >> 10) & 31;
    bitr r;
    r.ptr = z + 1; r.word = z[1]; r.left = 32;
    if (method == 3) {                                                               /* unpackTokensMinimum :590-643 */
        for (int j = 1; j <= nj; j += istep) {
            int lcl_n = ((j + istep - 1) >= nj ? nj - j : istep - 1);
            for (int i = 1; i <= ni; i += istep) {
                int lcl_m = ((i + istep - 1) >= ni ? ni - i : istep - 1);
                uint32_t nb = br_get(&r, 4), lo = 0;
                if (nb != 15 && nb != 16) lo = br_get(&r, nbits);
                for (int n = 0; n <= lcl_n; n++)
                    for (int m = 0; m <= lcl_m; m++) {
                        size_t k = (size_t)ni * (j + n - 1) + (i + m - 1);
                        if (nb == 0) tokens[k] = (uint16_t)lo;
                        else if (nb == 15 || nb == 16) tokens[k] = (uint16_t)br_get(&r, 16);
                        else tokens[k] = (uint16_t)(br_get(&r, (int)nb) + lo);
                    }
            }
        }
        return 0;
    }
    if (method != 4) return -1;
    int32_t *d = (int32_t *)calloc((size_t)ni * nj, sizeof(int32_t));               /* unpackTokensParallelogram :792-872 */
    uint32_t container = br_get(&r, 3);
    for (int i = 1; i <= ni; i++) tokens[i - 1] = (uint16_t)br_get(&r, nbits);
    for (int j = 2; j <= nj; j++) tokens[(size_t)ni * (j - 1)] = (uint16_t)br_get(&r, nbits);
    for (int j = 2; j <= nj; j += istep) {
        int lcl_n = ((j + istep - 1) >= nj ? nj - j : istep - 1);
        for (int i = 2; i <= ni; i += istep) {
            int lcl_m = ((i + istep - 1) >= ni ? ni - i : istep - 1);
            uint32_t nb = br_get(&r, (int)container);
            int width = nb == 0 ? 0 : ((nb == 15 || nb == 16) ? 17 : (int)nb + 1);
            for (int n = 0; n <= lcl_n; n++)
                for (int m = 0; m <= lcl_m; m++) {
                    size_t k = (size_t)ni * (j + n - 1) + (i + m - 1);
                    if (!width) { d[k] = 0; continue; }
                    int32_t t = (int32_t)br_get(&r, width);
                    d[k] = (int32_t)((uint32_t)t << (32 - width)) >> (32 - width);
                }
        }
    }
    for (int j = 2; j <= nj; j++)
        for (int i = 2; i <= ni; i++) {
            size_t k = (size_t)ni * (j - 1) + (i - 1);
            tokens[k] = (uint16_t)(d[k] + ((int)tokens[k - ni] + (int)tokens[k - 1] - (int)tokens[k - 1 - ni]));
        }
    free(d);
    return 0;
}

/* armn_compress COMPRESS / UNCOMPRESS on the packed-word layout (two 16-bit tokens per word, first in the
 * high half), in place, src/compresseur/c_zfstlib.c:108-200 */
int orc_armn_compress(unsigned char *fld, int ni, int nj, int nk, int nbits, int op_code)
{
    uint32_t *w = (uint32_t *)fld;
    size_t n = (size_t)ni * nj;
    if (op_code == 1) {
        if (nbits > 16 || ni == 1 || nj == 1) return -1;
        uint16_t *tok = (uint16_t *)malloc(sizeof(uint16_t) * (n + 2));
        for (size_t k = 0; k < n; k++) tok[k] = (uint16_t)((k & 1) ? (w[k >> 1] & 0xFFFF) : (w[k >> 1] >> 16));
        uint32_t *z = (uint32_t *)calloc(n * nk + 16, sizeof(uint32_t));
        int zlng = orc_armn_encode(z, tok, ni, nj, nbits);
        int lng_origin = 1 + ni * nj * nk * 16 / 8;
        int rc = -1;
        if (zlng < lng_origin) { memcpy(fld, z, (size_t)zlng); rc = zlng; }
        free(z); free(tok);
        return rc;
    }
    if (op_code == 2) {
        if (nbits > 16 || ni == 1 || nj == 1) return 1 + ni * nj * nk * nbits / 8;
        uint16_t *tok = (uint16_t *)calloc(n + 2, sizeof(uint16_t));
        if (orc_armn_decode(tok, w, ni, nj)) { free(tok); return -1; }
        for (size_t k = 0; k < n; k += 2) w[k >> 1] = (uint32_t)tok[k] << 16 | (k + 1 < n ? tok[k + 1] : 0);
        free(tok);
        return (int)(n * sizeof(short));
    }
    return 0;
}

/* ---------------------------------------------------------------------------------------------
 * compact_short (compact_integer.c:592-826, opCode 5 / 6) and compact_char (:830-1060, opCode 9 / 10): the unsigned forms of
 * compact_integer restated on arrays of unsigned short / unsigned char (the reference's code is the same text with the array
 * type changed; its store truncates to the element width)
 * ------------------------------------------------------------------------------------------- */
static int orc_compact_narrow(void *unpacked, void *packedHeader, void *packed, int elementCount, int bitSize, int off_set, int stride,
                              int opCode, int op_pack, int op_unpack, int elem_bytes)
{
    if (bitSize == 0) return 0;
    size_t n = (size_t)elementCount;
    if (opCode == op_unpack && packedHeader) n = ((uint32_t *)packedHeader)[1];
    uint32_t *tmp = (uint32_t *)calloc(n ? n : 1, sizeof(uint32_t));
    int rc;
    if (opCode == op_pack) {
        for (size_t i = 0; i < n; i++) tmp[i] = elem_bytes == 2 ? ((unsigned short *)unpacked)[i * stride] : ((unsigned char *)unpacked)[i * stride];
        rc = orc_compact_integer(tmp, packedHeader, packed, elementCount, bitSize, off_set, 1, 1);
    } else if (opCode == op_unpack) {
        rc = orc_compact_integer(tmp, packedHeader, packed, elementCount, bitSize, off_set, 1, 2);
        for (size_t i = 0; i < n; i++) {
            if (elem_bytes == 2) ((unsigned short *)unpacked)[i * stride] = (unsigned short)tmp[i];
            else ((unsigned char *)unpacked)[i * stride] = (unsigned char)tmp[i];
        }
    } else rc = 0;
    free(tmp);
    return rc;
}
int orc_compact_short(void *unpacked, void *packedHeader, void *packed, int elementCount, int bitSize, int off_set, int stride, int opCode)
{
    return orc_compact_narrow(unpacked, packedHeader, packed, elementCount, bitSize, off_set, stride, opCode, 5, 6, 2);
}
int orc_compact_char(void *unpacked, void *packedHeader, void *packed, int elementCount, int bitSize, int off_set, int stride, int opCode)
{
    return orc_compact_narrow(unpacked, packedHeader, packed, elementCount, bitSize, off_set, stride, opCode, 9, 10, 1);
}
