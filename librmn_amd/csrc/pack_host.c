/*
 * pack_host.c -- C host front-end of the MI355X-native field packers (librmn_ez_hip.so): librmn's
 * compact_float / compact_integer / c_float_packer / armn_compress entry points (include/packers_hip.h)
 * over the HIP kernels of pack_kernels.hip.  Host-pointer calls stage through device memory; there is
 * NO CPU fallback (calls fail loudly without a HIP device).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ezhip_shim.h"
#include "packhip_shim.h"
#include "../../include/packers_hip.h"

static int need_device(const char *who)
{
    if (ezhip_runtime_ok()) return ezhip_bound_device_ok(who);
    fprintf(stderr, "<%s> no usable HIP device: the MI355X packer path has no CPU fallback\n", who);
    return -1;
}
static unsigned key_of(float f) { unsigned u; memcpy(&u, &f, 4); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
static float float_of(unsigned k) { unsigned u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k; float f; memcpy(&f, &u, 4); return f; }

/* small per-thread device scratch: stats (4 u32) + cf params at +64 B + the reduction partials */
static __thread void *t_scratch = NULL;
/* per-thread grow-only device workspaces (slot 0/1: staged operands of the host-pointer entry points and the
 * compressed stream, slot 2: armn_compress scan storage): hipMalloc + hipFree per call cost more than the kernels
 * (a 52 MB hipMalloc/hipFree pair is ~0.4 ms and hipFree synchronises the device) */
static __thread struct { void *p; size_t cap; } t_ws[8];
static void *ws(int slot, size_t bytes)
{
    if (t_ws[slot].cap < bytes) {
        if (t_ws[slot].p) { ezhip_sync(); ezhip_free(t_ws[slot].p); }
        size_t cap = bytes + bytes / 4 + 256;
        t_ws[slot].p = ezhip_malloc(cap);
        t_ws[slot].cap = t_ws[slot].p ? cap : 0;
    }
    return t_ws[slot].p;
}
void ezhip_pack_release(void)
{
    ezhip_sync();
    for (int k = 0; k < 8; k++) { ezhip_free(t_ws[k].p); t_ws[k].p = NULL; t_ws[k].cap = 0; }
    ezhip_free(t_scratch); t_scratch = NULL;
}

static void *scratch(void)
{
    if (!t_scratch) t_scratch = ezhip_malloc(4 * PACKHIP_STATS_WORDS);
    return t_scratch;
}

/* ------------------------------------------------------------------------------------------ */
/* compact_float                                                                                */
/* ------------------------------------------------------------------------------------------ */
static void split_token_size(int arg, int *bs, int *eff)
{   /* compact.tmplc:121-129 */
    if (arg > 64) { *eff = arg >> 6; *bs = arg & 0x3F; } else { *bs = arg; *eff = arg; }
}

void *compact_float_dev(void *d_unpacked, void *d_header, void *d_packed, int elementCount, int packedTokenBitSize,
                        int offset, int stride, int opCode, int hasMissing, const void *missingTag, int header_style)
{
    if (need_device("compact_float")) return NULL;
    if (packedTokenBitSize == 0) return NULL;
    if (packedTokenBitSize == 1 && hasMissing) return NULL;
    int bs, eff;
    split_token_size(packedTokenBitSize, &bs, &eff);
    if (eff > 32 || bs > 32) { fprintf(stderr, "<compact_float> tokens wider than 32 bits are not supported on the MI355X path\n"); return NULL; }
    float tag = *(const float *)missingTag;
    unsigned *d_stats = (unsigned *)scratch();
    if (!d_stats) return NULL;
    packhip_cf_params *d_pp = (packhip_cf_params *)((char *)d_stats + 64);
    if (opCode == 1) {
        unsigned n = (unsigned)elementCount;
        if (header_style == 2 && n > 268435455u) { fprintf(stderr, "<compact_float> element count overflow in xxpack header\n"); return NULL; }
        int npart = packhip_stats_partials(d_stats, d_unpacked, n, stride, 0, hasMissing, tag);
        if (npart < 0 || packhip_cf_header(d_pp, (unsigned *)d_header, d_stats, npart, n, bs, header_style, hasMissing)) return NULL;
        if (packhip_cf_pack((unsigned *)d_packed, (const float *)d_unpacked, n, eff, offset, stride, d_pp, hasMissing, tag)) return NULL;
        return d_packed;
    }
    if (opCode == 2) {
        unsigned hdr[4];
        if (ezhip_d2h(hdr, d_header, sizeof(hdr)) || ezhip_sync()) return NULL;
        unsigned marker = hdr[0] >> 20, counter = hdr[0] & 0xFFFFF;
        unsigned rangeExpo = hdr[1] >> 16, minExpo = (hdr[1] >> 4) & 0xFFF, minSign = hdr[1] & 0xF;
        unsigned minMantisa32 = hdr[2], bitSize = (hdr[3] >> 8) & 0xFF, emptySpace = hdr[3] & 0xFF;
        unsigned intCount = (marker == 0x7ff || marker == 0x7ef) ? (unsigned)elementCount : (emptySpace << 20 | counter);
        int tokenSize = (int)bitSize;
        if (tokenSize > 32 || tokenSize < 1) return NULL;
        unsigned missingToken = (tokenSize != 32) ? ~(0xFFFFFFFFu << tokenSize) : ~0u;
        unsigned rangeExponent = rangeExpo - 4096 + 127 + tokenSize;
        double mulFactor = ldexp(1.0, (int)(rangeExponent - 127 - tokenSize)), minF = 0;
        if (!(minMantisa32 == 0 || minExpo < 849)) {
            unsigned u = (minSign & 1) << 31 | ((minExpo + 127 - 1024 + 48) & 0xFF) << 23 | ((minMantisa32 >> 8) & 0x7fffff);
            float f; memcpy(&f, &u, 4); minF = f;
        }
        if (packhip_cf_unpack((float *)d_unpacked, (const unsigned *)d_packed, intCount, tokenSize, offset, stride, minF, mulFactor, missingToken, hasMissing, tag)) return NULL;
        return d_unpacked;
    }
    fprintf(stderr, "<compact_float> opCode (%d) is not defined\n", opCode);
    return NULL;
}

void *compact_float(void *unpackedArrayOfFloat, void *packedHeader, void *packedArrayOfInt, const int elementCount,
                    const int packedTokenBitSize, const int offset, const int stride, const int opCode,
                    const int hasMissing, const void *const missingTag)
{
    if (need_device("compact_float")) return NULL;
    if (packedTokenBitSize == 0) return NULL;
    int bs, eff;
    split_token_size(packedTokenBitSize, &bs, &eff);
    if (eff > 32) { fprintf(stderr, "<compact_float> tokens wider than 32 bits are not supported on the MI355X path\n"); return NULL; }
    unsigned *hdr = (unsigned *)packedHeader, *out = (unsigned *)packedArrayOfInt;
    size_t n = (size_t)elementCount, nfl = n ? (n - 1) * (size_t)stride + 1 : 0;
    size_t span_words = ((size_t)offset + n * (size_t)eff + 31) / 32;      /* words the stream touches, from the stream base */
    /* header style (compact.tmplc:159-168): the stream may START INSIDE the header (style 1); mirror that
     * aliasing on the device by staging one contiguous image [header | stream] */
    int style = ((&hdr[3] == out && offset == 24) || (&hdr[0] == out && offset == 120)) ? 1 : 2;
    size_t lead = (style == 1) ? (size_t)(out - hdr) : 4;       /* words between header start and stream base */
    float *d_a = (float *)ws(0, sizeof(float) * (nfl ? nfl : 1));
    unsigned *d_img = (unsigned *)ws(1, 4 * (lead + span_words + 8));
    void *ret = NULL;
    if (!d_a || !d_img) goto done;
    unsigned *d_hdr = d_img, *d_out = d_img + lead;
    if (opCode == 1) {
        if (ezhip_h2d(d_a, unpackedArrayOfFloat, sizeof(float) * nfl)) goto done;
        if (ezhip_h2d(d_out, out, 4 * span_words)) goto done;      /* neighbouring bits of the first / last word survive */
        if (style == 1 && lead) { if (ezhip_h2d(d_hdr, hdr, 4 * lead)) goto done; }
        if (!compact_float_dev(d_a, d_hdr, d_out, elementCount, packedTokenBitSize, offset, stride, 1, hasMissing, missingTag, style)) goto done;
        if (style == 1) { size_t tot = lead + span_words; if (tot < 4) tot = 4; if (ezhip_d2h(hdr, d_hdr, 4 * tot)) goto done; }
        else { if (ezhip_d2h(hdr, d_hdr, 16) || ezhip_d2h(out, d_out, 4 * span_words)) goto done; }
        if (ezhip_sync()) goto done;
        ret = packedArrayOfInt;
    } else if (opCode == 2) {
        if (ezhip_h2d(d_hdr, hdr, 16) || ezhip_h2d(d_out, out, 4 * span_words)) goto done;
        if (nfl && stride != 1 && ezhip_h2d(d_a, unpackedArrayOfFloat, sizeof(float) * nfl)) goto done;   /* keep the gaps */
        if (!compact_float_dev(d_a, d_hdr, d_out, elementCount, packedTokenBitSize, offset, stride, 2, hasMissing, missingTag, style)) goto done;
        if (ezhip_d2h(unpackedArrayOfFloat, d_a, sizeof(float) * nfl) || ezhip_sync()) goto done;
        ret = unpackedArrayOfFloat;
    } else fprintf(stderr, "<compact_float> opCode (%d) is not defined\n", opCode);
done:
    ezhip_sync();
    return ret;
}

/* compact_double (compact.c:28-32: compact.tmplc on REAL*8 arrays; c_fstecr's packer for 64-bit IEEE fields, fstd98.c:826).
 * Same header and stream as compact_float; the extrema, the quantisation and the unpacked values are doubles.  The missing-value
 * tag is read through a double pointer and narrowed to float, as the reference does (compact.tmplc:106). */
void *compact_double(void *unpackedArrayOfFloat, void *packedHeader, void *packedArrayOfInt, const int elementCount,
                     const int packedTokenBitSize, const int offset, const int stride, const int opCode,
                     const int hasMissing, const void *const missingTag)
{
    if (need_device("compact_double")) return NULL;
    if (packedTokenBitSize == 0) return NULL;
    if (packedTokenBitSize == 1 && hasMissing) return NULL;
    int bs, eff;
    split_token_size(packedTokenBitSize, &bs, &eff);
    if (eff > 32 || bs > 32) { fprintf(stderr, "<compact_double> tokens wider than 32 bits are not supported on the MI355X path\n"); return NULL; }
    const float tag = (float)*(const double *)missingTag;
    unsigned *hdr = (unsigned *)packedHeader, *out = (unsigned *)packedArrayOfInt;
    size_t n = (size_t)elementCount, nfl = n ? (n - 1) * (size_t)stride + 1 : 0;
    size_t span_words = ((size_t)offset + n * (size_t)eff + 31) / 32;
    int style = ((&hdr[3] == out && offset == 24) || (&hdr[0] == out && offset == 120)) ? 1 : 2;
    size_t lead = (style == 1) ? (size_t)(out - hdr) : 4;
    double *d_a = (double *)ws(0, sizeof(double) * (nfl ? nfl : 1));
    unsigned *d_img = (unsigned *)ws(1, 4 * (lead + span_words + 8));
    char *d_sc = (char *)ws(4, sizeof(packhip_cf_params) + 64 + sizeof(double) * 2048);
    void *ret = NULL;
    if (!d_a || !d_img || !d_sc) goto done;
    packhip_cf_params *d_pp = (packhip_cf_params *)d_sc;
    double *d_partials = (double *)(d_sc + ((sizeof(packhip_cf_params) + 63) & ~(size_t)63));
    unsigned *d_hdr = d_img, *d_out = d_img + lead;
    if (opCode == 1) {
        if (style == 2 && n > 268435455u) { fprintf(stderr, "<compact_double> element count overflow in xxpack header\n"); goto done; }
        if (ezhip_h2d(d_a, unpackedArrayOfFloat, sizeof(double) * nfl)) goto done;
        if (ezhip_h2d(d_out, out, 4 * span_words)) goto done;
        if (style == 1 && lead) { if (ezhip_h2d(d_hdr, hdr, 4 * lead)) goto done; }
        if (packhip_cf_pack_f64(d_out, d_hdr, d_pp, d_partials, d_a, n, bs, eff, offset, stride, style, hasMissing, tag)) goto done;
        if (style == 1) { size_t tot = lead + span_words; if (tot < 4) tot = 4; if (ezhip_d2h(hdr, d_hdr, 4 * tot)) goto done; }
        else { if (ezhip_d2h(hdr, d_hdr, 16) || ezhip_d2h(out, d_out, 4 * span_words)) goto done; }
        if (ezhip_sync()) goto done;
        ret = packedArrayOfInt;
    } else if (opCode == 2) {
        unsigned marker = hdr[0] >> 20, counter = hdr[0] & 0xFFFFF;
        unsigned rangeExpo = hdr[1] >> 16, minExpo = (hdr[1] >> 4) & 0xFFF, minSign = hdr[1] & 0xF;
        unsigned minMantisa32 = hdr[2], bitSize = (hdr[3] >> 8) & 0xFF, emptySpace = hdr[3] & 0xFF;
        unsigned intCount = (marker == 0x7ff || marker == 0x7ef) ? (unsigned)elementCount : (emptySpace << 20 | counter);
        int tokenSize = (int)bitSize;
        if (tokenSize > 32 || tokenSize < 1) goto done;
        unsigned missingToken = (tokenSize != 32) ? ~(0xFFFFFFFFu << tokenSize) : ~0u;
        unsigned rangeExponent = rangeExpo - 4096 + 127 + tokenSize;
        double mulFactor = ldexp(1.0, (int)(rangeExponent - 127 - tokenSize)), minF = 0;
        if (!(minMantisa32 == 0 || minExpo < 849)) {         /* compact.tmplc:374-381: the minimum comes back as a REAL*4 */
            unsigned u = (minSign & 1) << 31 | ((minExpo + 127 - 1024 + 48) & 0xFF) << 23 | ((minMantisa32 >> 8) & 0x7fffff);
            float f; memcpy(&f, &u, 4); minF = f;
        }
        size_t nfl2 = intCount ? ((size_t)intCount - 1) * (size_t)stride + 1 : 0, span2 = ((size_t)offset + (size_t)intCount * (size_t)tokenSize + 31) / 32;
        d_a = (double *)ws(0, sizeof(double) * (nfl2 ? nfl2 : 1)); d_img = (unsigned *)ws(1, 4 * (span2 + 8));
        if (!d_a || !d_img || ezhip_h2d(d_img, out, 4 * span2)) goto done;
        if (nfl2 && stride != 1 && ezhip_h2d(d_a, unpackedArrayOfFloat, sizeof(double) * nfl2)) goto done;   /* keep the gaps */
        if (packhip_cf_unpack_f64(d_a, d_img, intCount, tokenSize, offset, stride, minF, mulFactor, missingToken, hasMissing, tag)) goto done;
        if (ezhip_d2h(unpackedArrayOfFloat, d_a, sizeof(double) * nfl2) || ezhip_sync()) goto done;
        ret = unpackedArrayOfFloat;
    } else fprintf(stderr, "<compact_double> opCode (%d) is not defined\n", opCode);
done:
    ezhip_sync();
    return ret;
}

/* ------------------------------------------------------------------------------------------ */
/* compact_integer                                                                              */
/* ------------------------------------------------------------------------------------------ */
static int bitlen32(unsigned v) { int n = 0; while (v) { v >>= 1; n++; } return n; }

typedef struct { int E, shift, with_header, is_signed; unsigned minv, positiveMask, hdr[4]; } ci_plan;

/* everything compact_integer decides before the token loop (compact_integer.c:396-530) */
static int ci_prepare_pack(ci_plan *pl, const void *d_unpacked, int has_header, size_t n, int bitSize, int stride, int opCode)
{
    unsigned *d_stats = (unsigned *)scratch();
    if (!d_stats) return -1;
    unsigned st[4] = {0, 0, 0, 0};
    memset(pl, 0, sizeof(*pl));
    pl->with_header = has_header; pl->is_signed = (opCode == 3);
    if (has_header || bitSize == -1) {
        if (packhip_stats(d_stats, d_unpacked, n, stride, opCode == 1 ? 1 : 2, 0, 0.f)) return -1;
        if (ezhip_d2h(st, d_stats, sizeof(st)) || ezhip_sync()) return -1;
    }
    if (has_header) {                                                     /* constructHeader :186-231 */
        unsigned maxRange;
        if (opCode == 1) { maxRange = st[1] - st[0]; pl->minv = st[0]; pl->hdr[2] = st[0]; pl->hdr[3] = st[1]; }
        else {
            int mn = (int)(st[0] ^ 0x80000000u), mx = (int)(st[1] ^ 0x80000000u);
            maxRange = (mx > 0 && mn < 0) ? (unsigned)mx + (unsigned)(-mn) : (unsigned)(mx - mn);
            pl->minv = (unsigned)mn; pl->hdr[2] = (unsigned)mn; pl->hdr[3] = (unsigned)mx;
        }
        int bitRequired = bitlen32(maxRange);
        if (bitSize == -1) bitSize = bitRequired;
        else while ((bitRequired - bitSize) > 0) { pl->shift++; bitRequired--; }
        pl->hdr[0] = 0xFD000000u | ((unsigned)pl->shift << 6) | (unsigned)bitSize;
        pl->hdr[1] = (unsigned)n;
    } else if (bitSize == -1) {                                           /* :446-489 */
        unsigned maxSpan;
        if (opCode == 3) { int mn = (int)(st[0] ^ 0x80000000u), mx = (int)(st[1] ^ 0x80000000u); maxSpan = (unsigned)((abs(mn) > mx) ? abs(mn) : mx); }
        else maxSpan = st[2];
        bitSize = bitlen32(maxSpan) + (opCode == 3 ? 1 : 0);
    }
    pl->E = bitSize;
    if (bitSize >= 1 && bitSize <= 32) pl->positiveMask = (opCode < 3) ? 0 : (1u << (bitSize - 1));
    return 0;
}

int compact_integer_dev(void *d_unpacked, void *d_header, void *d_packed, int elementCount,
                        int bitSizeOfPackedToken, int off_set, int stride, int opCode)
{
    if (need_device("compact_integer")) return 0;
    if (bitSizeOfPackedToken == 0) return 0;
    size_t n = (size_t)elementCount;
    if (opCode == 1 || opCode == 3) {
        ci_plan pl;
        if (ci_prepare_pack(&pl, d_unpacked, d_header != NULL, n, bitSizeOfPackedToken, stride, opCode)) return 0;
        if (d_header && (ezhip_h2d(d_header, pl.hdr, 16) || ezhip_sync())) return 0;
        if (pl.E < 1 || pl.E > 32) return pl.E;
        if (packhip_ci_pack((unsigned *)d_packed, d_unpacked, n, pl.E, off_set, stride, pl.with_header, pl.is_signed, pl.minv, pl.shift, pl.positiveMask)) return 0;
        return pl.E;
    }
    if (opCode == 2 || opCode == 4) {
        int tokenSize = bitSizeOfPackedToken, shift = 0; unsigned minv = 0;
        if (d_header != NULL) {
            unsigned hdr[4];
            if (ezhip_d2h(hdr, d_header, sizeof(hdr)) || ezhip_sync()) return 0;
            tokenSize = hdr[0] & 0x3F; shift = (hdr[0] >> 6) & 0x3F; n = hdr[1]; minv = hdr[2];
        }
        if (tokenSize < 1 || tokenSize > 32) { fprintf(stderr, "<compact_integer> token size %d not supported on the MI355X path\n", tokenSize); return 0; }
        unsigned negMask = (opCode < 3) ? 0 : (unsigned)(-(int)(1u << (bitSizeOfPackedToken - 1)));
        if (packhip_ci_unpack(d_unpacked, (const unsigned *)d_packed, n, tokenSize, off_set, stride, d_header != NULL, minv, shift, negMask)) return 0;
        return bitSizeOfPackedToken;
    }
    fprintf(stderr, "<compact_integer> opCode (%d) is not defined\n", opCode);
    return 0;
}

int compact_integer(void *unpackedArrayOfInt, void *packedHeader, void *packedArrayOfInt, int elementCount,
                    int bitSizeOfPackedToken, int off_set, int stride, int opCode)
{
    if (need_device("compact_integer")) return 0;
    if (bitSizeOfPackedToken == 0) return 0;
    size_t n = (size_t)elementCount;
    int rc = 0;
    unsigned *d_a = NULL, *d_out = NULL;
    if (opCode == 1 || opCode == 3) {
        size_t nin = n ? (n - 1) * (size_t)stride + 1 : 0;
        d_a = (unsigned *)ws(0, 4 * (nin ? nin : 1));
        if (!d_a || ezhip_h2d(d_a, unpackedArrayOfInt, 4 * nin)) goto done;
        ci_plan pl;
        if (ci_prepare_pack(&pl, d_a, packedHeader != NULL, n, bitSizeOfPackedToken, stride, opCode)) goto done;
        if (packedHeader) memcpy(packedHeader, pl.hdr, 16);
        rc = pl.E;
        if (pl.E < 1 || pl.E > 32) goto done;
        size_t words = ((size_t)off_set + n * (size_t)pl.E + 31) / 32;
        d_out = (unsigned *)ws(1, 4 * (words + 4));
        rc = 0;
        if (!d_out || ezhip_h2d(d_out, packedArrayOfInt, 4 * words)) goto done;     /* first/last word neighbours survive */
        if (packhip_ci_pack(d_out, d_a, n, pl.E, off_set, stride, pl.with_header, pl.is_signed, pl.minv, pl.shift, pl.positiveMask)) goto done;
        if (ezhip_d2h(packedArrayOfInt, d_out, 4 * words) || ezhip_sync()) goto done;
        rc = pl.E;
    } else if (opCode == 2 || opCode == 4) {
        int tokenSize = bitSizeOfPackedToken, shift = 0; unsigned minv = 0;
        if (packedHeader) { unsigned *h = (unsigned *)packedHeader; tokenSize = h[0] & 0x3F; shift = (h[0] >> 6) & 0x3F; n = h[1]; minv = h[2]; }
        if (tokenSize < 1 || tokenSize > 32) { fprintf(stderr, "<compact_integer> token size %d not supported on the MI355X path\n", tokenSize); goto done; }
        size_t nin = n ? (n - 1) * (size_t)stride + 1 : 0, words = ((size_t)off_set + n * (size_t)tokenSize + 31) / 32;
        d_a = (unsigned *)ws(0, 4 * (nin ? nin : 1)); d_out = (unsigned *)ws(1, 4 * (words + 4));
        if (!d_a || !d_out || ezhip_h2d(d_out, packedArrayOfInt, 4 * words)) goto done;
        if (stride != 1 && ezhip_h2d(d_a, unpackedArrayOfInt, 4 * nin)) goto done;
        unsigned negMask = (opCode < 3) ? 0 : (unsigned)(-(int)(1u << (bitSizeOfPackedToken - 1)));
        if (packhip_ci_unpack(d_a, d_out, n, tokenSize, off_set, stride, packedHeader != NULL, minv, shift, negMask)) goto done;
        if (ezhip_d2h(unpackedArrayOfInt, d_a, 4 * nin) || ezhip_sync()) goto done;
        rc = bitSizeOfPackedToken;
    } else fprintf(stderr, "<compact_integer> opCode (%d) is not defined\n", opCode);
done:
    ezhip_sync();
    return rc;
}

/* compact_short (compact_integer.c:592-826, opCode 5 pack / 6 unpack) and compact_char (:830-1060, opCode 9 / 10): compact_integer's
 * unsigned forms on arrays of unsigned short / unsigned char (c_fstecr / c_fstluk use them for datyp 2 fields held in 16- / 8-bit
 * arrays, fstd98.c:1214-1231, :2312-2322).  The elements are widened to a dense uint32 array on the device and go through the
 * compact_integer kernels; unpacked values are truncated to the element width like the reference's store does.
 * The SIGNED opcodes 7 / 8 and 11 / 12 that the `#else` branch of fstd98.c:1283-1294 names do not exist in compact_integer.c (:797-816,
 * :1034-1053 take 5 / 6 and 9 / 10 only); the reference is compiled with use_old_signed_pack_unpack_code (fstd98.c:22) and packs datyp 4
 * fields of short / byte elements by sign-extending them and calling compact_integer(opCode 3) (:1267-1279).  Here as there those opcodes
 * print "opCode (%d) is not defined" and return 0; the datyp 4 framing (fst_record_host.c) follows the branch that is compiled. */
static int compact_narrow(const char *who, int elem_bytes, int op_pack, int op_unpack, void *unpacked, void *packedHeader, void *packedArrayOfInt,
                          int elementCount, int bitSizeOfPackedToken, int off_set, int stride, int opCode)
{
    if (need_device(who)) return 0;
    if (bitSizeOfPackedToken == 0) return 0;
    size_t n = (size_t)elementCount;
    int rc = 0;
    if (opCode == op_pack) {
        size_t nin = n ? (n - 1) * (size_t)stride + 1 : 0;
        void *d_src = ws(4, (size_t)elem_bytes * (nin ? nin : 1) + 16);
        unsigned *d_a = (unsigned *)ws(0, 4 * (n ? n : 1));
        if (!d_src || !d_a || ezhip_h2d(d_src, unpacked, (size_t)elem_bytes * nin) || packhip_widen(d_a, d_src, n, stride, elem_bytes)) goto done;
        ci_plan pl;
        if (ci_prepare_pack(&pl, d_a, packedHeader != NULL, n, bitSizeOfPackedToken, 1, 1)) goto done;
        if (packedHeader) memcpy(packedHeader, pl.hdr, 16);
        rc = pl.E;
        if (pl.E < 1 || pl.E > 32) goto done;
        size_t words = ((size_t)off_set + n * (size_t)pl.E + 31) / 32;
        unsigned *d_out = (unsigned *)ws(1, 4 * (words + 4));
        rc = 0;
        if (!d_out || ezhip_h2d(d_out, packedArrayOfInt, 4 * words)) goto done;     /* first/last word neighbours survive */
        if (packhip_ci_pack(d_out, d_a, n, pl.E, off_set, 1, pl.with_header, 0, pl.minv, pl.shift, 0)) goto done;
        if (ezhip_d2h(packedArrayOfInt, d_out, 4 * words) || ezhip_sync()) goto done;
        rc = pl.E;
    } else if (opCode == op_unpack) {
        int tokenSize = bitSizeOfPackedToken, shift = 0; unsigned minv = 0;
        if (packedHeader) { unsigned *h = (unsigned *)packedHeader; tokenSize = h[0] & 0x3F; shift = (h[0] >> 6) & 0x3F; n = h[1]; minv = h[2]; }
        if (tokenSize < 1 || tokenSize > 32) { fprintf(stderr, "<%s> token size %d not supported on the MI355X path\n", who, tokenSize); goto done; }
        size_t nin = n ? (n - 1) * (size_t)stride + 1 : 0, words = ((size_t)off_set + n * (size_t)tokenSize + 31) / 32;
        void *d_dst = ws(4, (size_t)elem_bytes * (nin ? nin : 1) + 16);
        unsigned *d_a = (unsigned *)ws(0, 4 * (n ? n : 1)), *d_in = (unsigned *)ws(1, 4 * (words + 4));
        if (!d_dst || !d_a || !d_in || ezhip_h2d(d_in, packedArrayOfInt, 4 * words)) goto done;
        if (stride != 1 && ezhip_h2d(d_dst, unpacked, (size_t)elem_bytes * nin)) goto done;        /* keep the gaps */
        if (packhip_ci_unpack(d_a, d_in, n, tokenSize, off_set, 1, packedHeader != NULL, minv, shift, 0)) goto done;
        if (packhip_narrow(d_dst, d_a, n, stride, elem_bytes)) goto done;
        if (ezhip_d2h(unpacked, d_dst, (size_t)elem_bytes * nin) || ezhip_sync()) goto done;
        rc = bitSizeOfPackedToken;
    } else fprintf(stderr, "<%s> opCode (%d) is not defined\n", who, opCode);
done:
    ezhip_sync();
    return rc;
}
int compact_short(void *unpackedArrayOfShort, void *packedHeader, void *packedArrayOfInt, int elementCount,
                  int bitSizeOfPackedToken, int off_set, int stride, int opCode)
{
    return compact_narrow("compact_short", 2, 5, 6, unpackedArrayOfShort, packedHeader, packedArrayOfInt, elementCount, bitSizeOfPackedToken, off_set, stride, opCode);
}
int compact_char(void *unpackedArrayOfBytes, void *packedHeader, void *packedArrayOfInt, int elementCount,
                 int bitSizeOfPackedToken, int off_set, int stride, int opCode)
{
    return compact_narrow("compact_char", 1, 9, 10, unpackedArrayOfBytes, packedHeader, packedArrayOfInt, elementCount, bitSizeOfPackedToken, off_set, stride, opCode);
}

/* ------------------------------------------------------------------------------------------ */
/* c_float_packer / c_float_unpacker                                                            */
/* ------------------------------------------------------------------------------------------ */
static int aligned_mantissa(int src, int maxExp)
{
    int mantis = (1 << 23) | (0x7FFFFF & src);
    int e = (src >> 23) & 0xFF, shift = maxExp - e;
    if (shift > 31) shift = 31;
    mantis = mantis >> shift;
    if (src >> 31) mantis = -mantis;
    return mantis;
}

int32_t c_float_packer(float *source, int32_t nbits, int32_t *header, int32_t *stream, int32_t npts)
{
    if (nbits > 16 || nbits < 1) { fprintf(stderr, "<c_float_packer> nbits must be > 0 and <= 16 ,nbits = %d\n", nbits); return -1; }
    if (need_device("c_float_packer")) return -1;
    header[2] = npts;
    header[0] = (int32_t)(0xEFFu << 20);
    header[0] = header[0] | ((nbits - 1) << 16);
    float *d_src = (float *)ws(0, 4 * (size_t)(npts > 0 ? npts : 1));
    int *d_stream = (int *)ws(1, 4 * (size_t)((npts + 1) / 2 + 1));
    unsigned *d_stats = (unsigned *)scratch();
    int rc = -1;
    if (!d_src || !d_stream || !d_stats) goto done;
    unsigned st[4];
    if (ezhip_h2d(d_src, source, 4 * (size_t)npts)) goto done;
    if (packhip_stats(d_stats, d_src, (size_t)npts, 1, 0, 0, 0.f)) goto done;
    if (ezhip_d2h(st, d_stats, sizeof(st)) || ezhip_sync()) goto done;
    {   /* float_packer_1, float_packer.c:128-202: header arithmetic on the min / max values */
        float fminf_ = float_of(st[0]), fmaxf_ = float_of(st[1]);
        int fmin_i, fmax_i;
        memcpy(&fmin_i, &fminf_, 4); memcpy(&fmax_i, &fmaxf_, 4);
        int MaxExp = (fmax_i >> 23) & 0xFF, Exp = (fmin_i >> 23) & 0xFF;
        MaxExp = MaxExp > Exp ? MaxExp : Exp;
        int Maximum = aligned_mantissa(fmax_i, MaxExp);
        if (((fmax_i >> 23) & 0xFF) < 1) Maximum = 0;
        int Minimum = aligned_mantissa(fmin_i, MaxExp);
        if (((fmin_i >> 23) & 0xFF) < 1) Minimum = 0;
        Maximum = Maximum - Minimum;
        int Shift2 = 0, Round = 1, Mask = (int)~(~0u << nbits);
        while (Maximum > Mask) { Maximum = Maximum >> 1; Round = Round << 1; Shift2++; }
        Round = Round >> 1;
        header[1] = Minimum;
        header[0] = header[0] | ((MaxExp & 0xFF) << 8) | (Shift2 & 0xFF);
        if (packhip_fp_pack(d_stream, d_src, npts, MaxExp, Minimum, Round, Shift2, Mask)) goto done;
    }
    if (ezhip_d2h(stream, d_stream, 4 * (size_t)((npts + 1) / 2)) || ezhip_sync()) goto done;
    rc = 0;
done:
    ezhip_sync();
    (void)key_of;
    return rc;
}

int32_t c_float_unpacker(float *dest, int32_t *header, int32_t *stream, int32_t npts, int32_t *nbits)
{
    *nbits = ((header[0] >> 16) & 0xF) + 1;
    if (0xEFF != ((header[0] >> 20) & 0xFFF)) { fprintf(stderr, "<c_float_unpacker> Invalid header\n"); return -1; }
    if (npts != header[2]) { fprintf(stderr, "<c_float_unpacker> Inconsistent number of points (header/request mismatch)\n"); return -1; }
    if (need_device("c_float_unpacker")) return -1;
    float *d_dest = (float *)ws(0, 4 * (size_t)(npts > 0 ? npts : 1));
    int *d_stream = (int *)ws(1, 4 * (size_t)((npts + 1) / 2 + 1));
    int rc = -1;
    if (!d_dest || !d_stream) goto done;
    if (ezhip_h2d(d_stream, stream, 4 * (size_t)((npts + 1) / 2))) goto done;
    if (packhip_fp_unpack(d_dest, d_stream, npts, (header[0] >> 8) & 0xFF, header[1], header[0] & 0xFF)) goto done;
    if (ezhip_d2h(dest, d_dest, 4 * (size_t)npts) || ezhip_sync()) goto done;
    rc = 0;
done:
    ezhip_sync();
    return rc;
}

void c_float_packer_params(int32_t *header_size, int32_t *stream_size, int32_t *p1, int32_t *p2, int32_t npts)
{   /* float_packer.c:289-319 */
    *header_size = 3 * (int32_t)sizeof(int32_t);
    *stream_size = ((npts + 1) / 2) * (int32_t)sizeof(int32_t);
    *p1 = 0; *p2 = 0;
}

/* ------------------------------------------------------------------------------------------ */
/* armn_compress                                                                                */
/* ------------------------------------------------------------------------------------------ */
static int g_level = -1, g_swap = 1;
void c_armn_compress_setlevel(int level) { g_level = level; }
int  c_armn_compress_getlevel(void) { return g_level; }
void c_armn_compress_setswap(int swapState) { g_swap = swapState; }

/* one field through the one-pass encoder (k_armn_enc1); the rare field with a |difference| > 65535 runs twice (zlng -2 ->
 * container 5, c_zfstlib.c:701-711).  Returns zlng, -1 (not compressible / error) or -3 (shape outside the encoder's limits). */
static int armn_encode1_sync(unsigned *d_z, size_t z_cap_words, const unsigned *d_words, int ni, int nj, int nbits)
{
    size_t wb = packhip_armn_enc1_work_bytes(ni, nj, nbits, 1);
    if (!wb || getenv("EZHIP_ARMN_MULTIPASS")) return -3;
    char *d_work = (char *)ws(2, wb + 64);
    if (!d_work) return -1;
    int *d_zl = (int *)(d_work + ((wb + 15) & ~(size_t)15));
    int zl = -1;
    for (int container = 4; container <= 5; container++) {
        if (ezhip_memset(d_zl, 0xFD, sizeof(int))) return -1;          /* "unfinished": the kernel's bounded waits gave up (see k_armn_enc1) */
        int rc = packhip_armn_encode1(d_z, 0, z_cap_words, d_words, 0, ni, nj, nbits, 1, g_swap == 1, container, d_work, d_zl);
        if (rc == 1) return -3;
        if (rc || ezhip_d2h(&zl, d_zl, sizeof(int)) || ezhip_sync()) return -1;
        if (zl != -2) break;
    }
    if (zl < -2) { fprintf(stderr, "<armn_compress> one-pass encoder gave up waiting on the launch order; using the multi-kernel encoder\n"); return -3; }
    return zl;
}

static int armn_compress_dev_ex(void *d_z, const void *d_words, int ni, int nj, int nbits, int allow_onepass);
int armn_compress_dev(void *d_z, const void *d_words, int ni, int nj, int nbits) { return armn_compress_dev_ex(d_z, d_words, ni, nj, nbits, 1); }
static int armn_compress_dev_ex(void *d_z, const void *d_words, int ni, int nj, int nbits, int allow_onepass)
{
    if (need_device("armn_compress")) return -1;
    if (nbits > 16 || ni == 1 || nj == 1) { fprintf(stderr, "<armn_compress> Cannot compress if nbits>16 or ni=1 or nj=1. Returning original field\n"); return -1; }
    if (g_level == -1) g_level = 1;                                   /* BEST, c_zfstlib.c:92-97 */
    int minimum = (g_level == 0) || ni < 16 || nj < 16 || nbits <= 4; /* :132 */
    size_t n = (size_t)ni * nj;
    /* the raw stream can exceed the source size for incompressible fields: the caller's buffer holds n/2 + 16 words; the
     * one-pass encoder drops every store at or beyond that capacity (the byte count then says "not compressible") */
    size_t z_words = n / 2 + 16;
    if (!minimum && allow_onepass) {
        int zl = armn_encode1_sync((unsigned *)d_z, z_words, (const unsigned *)d_words, ni, nj, nbits);
        if (zl != -3) return zl < 0 ? -1 : zl;
    }
    /* MINIMUM method (level FAST, small fields, nbits <= 4) and shapes outside the one-pass limits: the multi-kernel encoder.
     * Its emitters do not bound their stores: it works in a scratch stream with 25 % slack and only a compressible result
     * (which fits the caller's buffer by definition) is copied out. */
    void *d_work = ws(2, packhip_armn_work_bytes(ni, nj));
    size_t zs_words = n / 2 + n / 8 + 64;
    unsigned *d_zs = (unsigned *)ws(3, 4 * zs_words + 4 * ((n + 1) / 2 + 4));
    if (!d_work || !d_zs) return -1;
    const unsigned *d_tok = (const unsigned *)d_words;
    if (g_swap != 1) {                                                /* c_zfstlib.c:119-126: without the swap the ushort array is the raw memory order */
        unsigned *d_sw = d_zs + zs_words;
        if (packhip_swap_halves(d_sw, d_tok, (n + 1) / 2)) return -1;
        d_tok = d_sw;
    }
    unsigned long long bits = 0;
    int rc = packhip_armn_encode(d_zs, zs_words, d_tok, ni, nj, nbits, minimum, d_work, &bits);
    if (rc) return -1;
    /* a word is flushed only when a later token crosses its end (stuff macro), the two 16-bit terminator
     * tokens flush every word that holds stream bits: words = ceil(bits / 32) */
    long long zlng = 1 + 4 * (1 + (long long)((bits + 31) / 32));
    long long lng_origin = 1 + (long long)n * 2;
    if (zlng >= lng_origin) return -1;
    if (ezhip_d2d(d_z, d_zs, (size_t)((zlng + 3) / 4) * 4) || ezhip_sync()) return -1;
    return (int)zlng;
}

/* UNCOMPRESS on device data (c_zfstlib.c:181-200): nfields streams, z_stride_words apart, each with room for z_words
 * words -> (1 + ni*nj/2) words of 16-bit tokens each, out_stride_words apart.  Returns ni*nj*2 like the reference. */
int32_t ezhip_armn_uncompress_batch_dev(void *d_words, int64_t out_stride_words, const void *d_z, int64_t z_stride_words, int64_t z_words,
                                        int ni, int nj, int nbits, int32_t nfields)
{
    if (nbits > 16 || ni == 1 || nj == 1) return 1 + ni * nj * nbits / 8;          /* :182-184: nothing is touched */
    if (need_device("armn_compress")) return -1;
    if (nfields < 1) return ni * nj * 2;
    size_t per = (packhip_armn_dec_work_bytes(ni, nj, (size_t)z_words) + 255) & ~(size_t)255;
    char *d_work = (char *)ws(2, per * (size_t)nfields + 4 * (size_t)nfields + 256);
    if (!d_work) return -1;
    int *d_status = (int *)(d_work + per * (size_t)nfields);
    if (nfields >= 8 && out_stride_words > 0 && z_stride_words > 0 && !getenv("EZHIP_DEC_ONE_STREAM")) {
        /* a batch in two halves: the chain kernels of the first half (one CU per stream, ~half of a batch's time) run beside the speculation and composition
         * kernels of the second half, which fill the device: first half up to its composed table; second half whole, on the side stream, from there on;
         * the first half's chain kernel and what follows on the caller's stream meanwhile */
        const int h = nfields / 2;
        unsigned *o = (unsigned *)d_words; const unsigned *z = (const unsigned *)d_z;
        if (packhip_armn_decode_phase(o, (size_t)out_stride_words, z, (size_t)z_stride_words, (size_t)z_words, ni, nj, h, g_swap == 1, d_work, per, d_status, 1)) return -1;
        if (ezhip_side_begin()) return -1;
        int bad = packhip_armn_decode_phase(o + (size_t)h * (size_t)out_stride_words, (size_t)out_stride_words, z + (size_t)h * (size_t)z_stride_words, (size_t)z_stride_words,
                                            (size_t)z_words, ni, nj, nfields - h, g_swap == 1, d_work + per * (size_t)h, per, d_status + h, 0);
        if (ezhip_side_end()) bad = 1;
        if (!bad) bad = packhip_armn_decode_phase(o, (size_t)out_stride_words, z, (size_t)z_stride_words, (size_t)z_words, ni, nj, h, g_swap == 1, d_work, per, d_status, 2);
        if (ezhip_side_join() || bad) { ezhip_sync(); return -1; }
    } else if (packhip_armn_decode((unsigned *)d_words, (size_t)out_stride_words, (const unsigned *)d_z, (size_t)z_stride_words, (size_t)z_words,
                            ni, nj, nfields, g_swap == 1, d_work, per, d_status)) return -1;
    int st[64], rc = ni * nj * 2;
    for (int f0 = 0; f0 < nfields; f0 += 64) {
        int c = nfields - f0 < 64 ? nfields - f0 : 64;
        if (ezhip_d2h(st, d_status + f0, sizeof(int) * (size_t)c) || ezhip_sync()) return -1;
        for (int k = 0; k < c; k++)
            if (st[k]) { fprintf(stderr, "<armn_compress> UNCOMPRESS: field %d: %s\n", f0 + k, st[k] == -1 ? "Unknown compression algorithm" : "broken stream"); rc = -1; }
    }
    return rc;
}
int armn_uncompress_dev(void *d_words, const void *d_z, int64_t z_words, int ni, int nj, int nbits)
{
    return ezhip_armn_uncompress_batch_dev(d_words, 0, d_z, 0, z_words, ni, nj, nbits, 1);
}

int armn_compress(unsigned char *fld, int ni, int nj, int nk, int nbits, int op_code)
{
    (void)nk;
    if (op_code == 2) {
        if (nbits > 16 || ni == 1 || nj == 1) return 1 + ni * nj * nk * nbits / 8;
        if (need_device("armn_compress")) return -1;
        /* the stream is shorter than the plain record it replaced; the reference writes (1 + ni*nj/2) words back (:189) */
        size_t words = 1 + (size_t)ni * nj / 2;
        unsigned *d_zc = (unsigned *)ws(0, 4 * (words + 2)), *d_tok = (unsigned *)ws(1, 4 * (words + 2));
        int rc = -1;
        if (d_zc && d_tok && !ezhip_h2d(d_zc, fld, 4 * words)) {
            rc = armn_uncompress_dev(d_tok, d_zc, (int64_t)words, ni, nj, nbits);
            if (rc > 0 && (ezhip_d2h(fld, d_tok, 4 * words) || ezhip_sync())) rc = -1;
        }
        ezhip_sync();
        return rc;
    }
    if (op_code != 1) return 0;
    if (need_device("armn_compress")) return -1;
    if (nbits > 16 || ni == 1 || nj == 1) { fprintf(stderr, "<armn_compress> Cannot compress if nbits>16 or ni=1 or nj=1. Returning original field\n"); return -1; }
    size_t n = (size_t)ni * nj, words = (n + 1) / 2;
    unsigned *d_in = (unsigned *)ws(0, 4 * (words + 2)), *d_z = (unsigned *)ws(1, 4 * (n / 2 + 32));
    int zlng = -1;
    if (!d_in || !d_z) goto done;
    if (ezhip_h2d(d_in, fld, 4 * words)) goto done;
    zlng = armn_compress_dev(d_z, d_in, ni, nj, nbits);
    if (zlng > 0) {
        /* the reference copies zlng bytes; its last byte comes from an unwritten malloc'ed word (undefined):
         * here it is the zero byte of the zero-initialised stream */
        if (ezhip_d2h(fld, d_z, (size_t)zlng) || ezhip_sync()) zlng = -1;
    }
done:
    ezhip_sync();
    return zlng;
}

static int pack16_compress_dev_ex(void *d_record, const float *d_field, int ni, int nj, int nbits, int32_t *failed);
/* fused cfg5 step on device data (fstd98.c:1170-1172): compact_float(16-bit slots, style 2) + armn_compress */
int ezhip_pack16_compress_dev(void *d_record, const float *d_field, int ni, int nj, int nbits)
{
    float tag = 0.f;
    unsigned *rec = (unsigned *)d_record;
    size_t n = (size_t)ni * nj;
    if (!compact_float_dev((void *)d_field, rec, rec + 4, (int)n, nbits + 64 * (nbits > 16 ? nbits : 16), 0, 1, 1, 0, &tag, 2)) return -1;
    unsigned *d_z = (unsigned *)ws(1, 4 * (n / 2 + 32));
    if (!d_z) return -1;
    int zlng = armn_compress_dev(d_z, rec + 4, ni, nj, nbits);
    if (zlng > 0 && ezhip_d2d(rec + 4, d_z, (size_t)zlng)) zlng = -1;
    ezhip_sync();
    return zlng;
}

/* the same, telling "not compressible" (-1, *failed = 0) from an error (-1, *failed = 1: a runtime call or launch failed underneath) */
static int pack16_compress_dev_ex(void *d_record, const float *d_field, int ni, int nj, int nbits, int32_t *failed)
{
    const unsigned before = ezhip_error_count();
    const int zl = ezhip_pack16_compress_dev(d_record, d_field, ni, nj, nbits);
    *failed = zl < 0 && ezhip_error_count() != before;
    return zl;
}

/* read side of ezhip_pack16_compress_dev (fstd98.c:2285-2290): a record of 4 header words + armn stream (compressed != 0) or
 * plain 16-bit slots -> floats.  The record is left untouched. */
int ezhip_uncompress_unpack16_dev(float *d_field, const void *d_record, int ni, int nj, int nbits, int compressed)
{
    float tag = 0.f;
    const unsigned *rec = (const unsigned *)d_record;
    size_t n = (size_t)ni * nj, words = 1 + n / 2;
    const unsigned *tok = rec + 4;
    if (compressed) {
        unsigned *d_tok = (unsigned *)ws(1, 4 * (words + 2));
        if (!d_tok) return -1;
        if (armn_uncompress_dev(d_tok, rec + 4, (int64_t)words, ni, nj, nbits) < 0) return -1;
        tok = d_tok;
    }
    if (!compact_float_dev(d_field, (void *)rec, (void *)tok, (int)n, nbits + 64 * (nbits > 16 ? nbits : 16), 0, 1, 2, 0, &tag, 2)) return -1;
    return 0;
}

/* The same two steps between HOST arrays with ONE trip each way (what c_fstecr / c_fstluk do for datyp 129, nbits <= 16: fstd98.c:1170-1172, :2285-2290):
 * through compact_float + armn_compress on host arrays a field crossed PCIe four times (up, tokens down, tokens up, stream down).
 * ezhip_pack16_compress_host: the field goes up, `out_words` words [4 header words][stream][what the token array held behind it, as the in-place
 * compressor leaves it] come down; returns zlng; -1 when compression does not pay (armn_compress's own answer; nothing written: the caller packs the
 * plain form); -2 on an ERROR (no device, workspace, copy or kernel failure): the caller must fail, not change the record's type.
 * ezhip_uncompress_unpack16_host: the record goes up, the field comes down. */
int32_t ezhip_pack16_compress_host(uint32_t *record, int64_t (*out_words_of)(int32_t zlng), const float *field, int32_t ni, int32_t nj, int32_t nbits)
{
    if (need_device("ezhip_pack16_compress_host")) return -2;
    const size_t n = (size_t)ni * nj;
    if (!record || !field || !out_words_of || n == 0 || nbits < 1 || nbits > 16) return -2;
    float *d_f = (float *)ws(6, 4 * n + 64);
    unsigned *d_rec = (unsigned *)ws(7, 4 * (4 + n / 2 + 64));
    if (!d_f || !d_rec) return -2;
    if (ezhip_h2d(d_f, field, 4 * n)) return -2;
    int32_t failed = 0;
    const int zl = pack16_compress_dev_ex(d_rec, d_f, ni, nj, nbits, &failed);
    if (failed) return -2;
    if (zl <= 0) return -1;                                         /* not compressible */
    int64_t nw = out_words_of(zl);
    if (nw > (int64_t)(4 + (n + 1) / 2)) nw = (int64_t)(4 + (n + 1) / 2);
    if (ezhip_d2h(record, d_rec, 4 * (size_t)nw) || ezhip_sync()) return -2;
    return zl;
}
int32_t ezhip_uncompress_unpack16_host(float *field, const uint32_t *record, int64_t record_words, int32_t ni, int32_t nj, int32_t nbits)
{
    if (need_device("ezhip_uncompress_unpack16_host")) return -1;
    const size_t n = (size_t)ni * nj, cap = 4 + n / 2 + 64;
    if (!record || !field || n == 0 || nbits < 1 || nbits > 16 || record_words < 5) return -1;
    if ((size_t)record_words > cap) record_words = (int64_t)cap;
    float *d_f = (float *)ws(6, 4 * n + 64);
    unsigned *d_rec = (unsigned *)ws(7, 4 * cap);
    if (!d_f || !d_rec) return -1;
    if (ezhip_h2d(d_rec, record, 4 * (size_t)record_words)) return -1;
    if (ezhip_uncompress_unpack16_dev(d_f, d_rec, ni, nj, nbits, 1) < 0) { ezhip_sync(); return -1; }
    if (ezhip_d2h(field, d_f, 4 * n) || ezhip_sync()) return -1;
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* fused cfg5 front half: interpolate a batch and pack every output field to 16-bit slots         */
/* ------------------------------------------------------------------------------------------ */
int32_t c_ezsint_batch_dev(float *d_zout, const float *d_zin, int32_t nfields);
int32_t ezhip_ezsint_batch_minmax_dev(float *d_zout, const float *d_zin, int32_t nfields, uint32_t *d_partials, int64_t stride_words, int32_t *partials_per_field);

/* c_ezsint_batch_dev + compact_float(nbits + 64*max(16,nbits), style 2) of every output field (fstd98.c:1170), all on
 * the calling thread's stream, no synchronisation.  The min/max pass of compact_float comes out of the interpolation
 * kernel itself (every k_sepx thread block leaves the extrema of what it stored; the header kernel reduces them), so
 * packing a field costs the token kernel only.  d_records: nfields records of record_stride_words words,
 * [4 header words][stream].  Returns the c_ezsint_batch_dev code (0 / 2) or -1. */
int32_t ezhip_ezsint_pack16_batch_dev(void *d_records, int64_t record_stride_words, float *d_zout, const float *d_zin,
                                      int32_t nfields, int32_t npts_out, int32_t nbits)
{
    if (need_device("ezhip_ezsint_pack16_batch_dev")) return -1;
    if (nfields < 1 || npts_out < 1 || nbits < 1 || nbits > 32) return -1;
    const int bs = nbits, eff = nbits > 16 ? nbits : 16;
    const size_t stride = PACKHIP_STATS_PARTIALS + 3 * (size_t)8192;          /* words per field: result area + up to 8192 triples */
    unsigned *d_stats_all = (unsigned *)ws(3, (4 * stride + sizeof(packhip_cf_params)) * (size_t)nfields + 64);
    if (!d_stats_all) return -1;
    packhip_cf_params *d_pp = (packhip_cf_params *)(d_stats_all + stride * (size_t)nfields);      /* one per field */
    int32_t npf = 0;
    int32_t rc = ezhip_ezsint_batch_minmax_dev(d_zout, d_zin, nfields, d_stats_all + PACKHIP_STATS_PARTIALS, (int64_t)stride, &npf);
    int fused = rc >= 0;
    if (rc == -2) rc = c_ezsint_batch_dev(d_zout, d_zin, nfields);             /* plan without the single-launch path: plain pipeline */
    if (rc < 0) return -1;
    float tag = 0.f;
    if (fused && packhip_cf_header_batch(d_pp, (unsigned *)d_records, (size_t)record_stride_words, d_stats_all, stride,
                                         npf, (unsigned)npts_out, bs, 2, 0, nfields)) return -1;       /* all headers in one launch */
    for (int f = 0; f < nfields; f++) {
        unsigned *rec = (unsigned *)d_records + (size_t)f * (size_t)record_stride_words;
        float *field = d_zout + (size_t)f * (size_t)npts_out;
        if (fused) {
            if (packhip_cf_pack(rec + 4, field, (size_t)npts_out, eff, 0, 1, d_pp + f, 0, tag)) return -1;
        } else if (!compact_float_dev(field, rec, rec + 4, npts_out, nbits + 64 * eff, 0, 1, 1, 0, &tag, 2)) return -1;
    }
    return rc;
}

/* cfg5 back half for a batch: compact_float(16-bit slots) + armn_compress of nfields device-resident fields with ONE
 * synchronisation at the end (ezhip_pack16_compress_dev returns zlng per call: one sync per field, ~25 % of its time).
 * prepacked != 0: the records already hold the 16-bit-slot pack (ezhip_ezsint_pack16_batch_dev).  zlng_out[f] = byte
 * count of record f's compressed stream, or -1 (not compressible: the record keeps the plain pack). */
/* fields whose one-pass encode asked for the 5-bit container (zlng -2) or did not finish (< -2): re-run them one by one, synchronously */
static int redo_wide_fields(unsigned *d_records, size_t record_stride_words, const unsigned *d_tok, size_t tok_stride_words, int in_place,
                            int32_t nfields, int ni, int nj, int nbits, int32_t *zlng_out)
{
    const size_t n = (size_t)ni * nj;
    for (int f = 0; f < nfields; f++) {
        if (zlng_out[f] >= -1) continue;                              /* -2: 5-bit container needed; < -2: the one-pass launch gave up */
        const int onepass = zlng_out[f] == -2;
        unsigned *rec = d_records + (size_t)f * record_stride_words;
        const unsigned *tok = in_place ? rec + 4 : d_tok + (size_t)f * tok_stride_words;
        unsigned *d_z = (unsigned *)ws(4, 4 * (n / 2 + 32));          /* its own slot: slot 1 may hold the batch's tokens */
        if (!d_z) return -1;
        int zl = armn_compress_dev_ex(d_z, tok, ni, nj, nbits, onepass);
        if (zl > 0) { if (ezhip_d2d(rec + 4, d_z, (size_t)zl) || ezhip_sync()) return -1; }
        else if (!in_place && (ezhip_d2d(rec + 4, tok, 4 * ((n + 1) / 2)) || ezhip_sync())) return -1;
        zlng_out[f] = zl;
    }
    return 0;
}

int32_t ezhip_pack16_compress_batch_dev(void *d_records, int64_t record_stride_words, const float *d_fields, int64_t field_stride,
                                        int32_t nfields, int32_t ni, int32_t nj, int32_t nbits, int32_t prepacked, int32_t *zlng_out)
{
    if (need_device("ezhip_pack16_compress_batch_dev")) return -1;
    if (nfields < 1 || nbits > 16 || ni == 1 || nj == 1) return -1;
    if (g_level == -1) g_level = 1;
    const int minimum = (g_level == 0) || ni < 16 || nj < 16 || nbits <= 4;
    const size_t n = (size_t)ni * nj;
    unsigned *d_z = (unsigned *)ws(1, 4 * (n / 2 + n / 8 + 64));
    const size_t wb1 = minimum || getenv("EZHIP_ARMN_MULTIPASS") ? 0 : packhip_armn_enc1_work_bytes(ni, nj, nbits, 1);
    void *d_work = ws(2, wb1 ? wb1 : packhip_armn_work_bytes(ni, nj));
    int *d_zlng = (int *)ws(0, sizeof(int) * (size_t)nfields + 64);
    if (!d_z || !d_work || !d_zlng) return -1;
    float tag = 0.f;
    if (ezhip_memset(d_zlng, 0xFD, sizeof(int) * (size_t)nfields)) return -1;       /* "unfinished" until a kernel says otherwise */
    for (int f = 0; f < nfields; f++) {
        unsigned *rec = (unsigned *)d_records + (size_t)f * (size_t)record_stride_words;
        if (!prepacked && !compact_float_dev((void *)(d_fields + (size_t)f * (size_t)field_stride), rec, rec + 4, (int)n,
                                             nbits + 64 * 16, 0, 1, 1, 0, &tag, 2)) return -1;
        if (wb1) {
            /* the tokens sit where the stream has to go: encode into the scratch stream, then the stream's words (zlng > 0 only) back */
            if (packhip_armn_encode1(d_z, 0, n / 2 + 16, rec + 4, 0, ni, nj, nbits, 1, g_swap == 1, 4, d_work, d_zlng + f)) return -1;
            if (packhip_cond_copy(rec + 4, 0, d_z, 0, 0, d_zlng + f, 1, 0)) return -1;
        } else {
            if (g_swap != 1) { fprintf(stderr, "<armn_compress> swap state 0 is not supported by the batched MINIMUM path\n"); return -1; }
            if (packhip_armn_encode(d_z, n / 2 + n / 8 + 64, rec + 4, ni, nj, nbits, minimum, d_work, NULL)) return -1;
            if (packhip_armn_commit(rec + 4, d_z, d_work, ni, nj, nbits, minimum, d_zlng + f)) return -1;
        }
    }
    if (ezhip_d2h(zlng_out, d_zlng, sizeof(int) * (size_t)nfields) || ezhip_sync()) return -1;
    return redo_wide_fields((unsigned *)d_records, (size_t)record_stride_words, NULL, 0, 1, nfields, ni, nj, nbits, zlng_out);
}

/* ------------------------------------------------------------------------------------------ */
/* the whole cfg5 pipeline on a batch, fused (round 2)                                            */
/* ------------------------------------------------------------------------------------------ */
int32_t ezhip_current_set_dims(int32_t *ni_in, int32_t *nj_in, int32_t *ni_out, int32_t *nj_out);
int32_t ezhip_ezsint_batch_minmax_bb_dev(const float *d_zin, int32_t nfields, uint32_t *d_partials, int64_t stride_words, int32_t *partials_per_field, int32_t *d_flags);
int32_t ezhip_ezsint_batch_minmax_only_dev(const float *d_zin, int32_t nfields, uint32_t *d_partials, int64_t stride_words, int32_t *partials_per_field);
int32_t ezhip_ezsint_batch_tokens_dev(uint32_t *d_tokens, int64_t token_stride_words, const float *d_zin, int32_t nfields,
                                      const void *d_params, int64_t param_stride_bytes);

/* c_ezsint (current grid set) -> compact_float(nbits + 64*16, header style 2) -> armn_compress of nfields device-resident source
 * fields (fstd98.c:1170-1172), without ever storing the interpolated floats:
 *   A  k_sepx<.., min/max only>: the values are computed and reduced, nothing is stored (reads the sources)
 *   H  one header launch: min/max -> compact_float header words of every record + quantisation parameters
 *   B  k_sepx<.., tokens>: the same values again -> 16-bit tokens (reads the sources, writes 2 bytes per point)
 *   E  k_armn_enc1: tokens -> streams, written in place in the records (one launch for the batch)
 *   F  fields that are not compressible get the plain 16-bit pack (conditional copy, one launch)
 * One synchronisation at the end (zlng_out).  Returns the c_ezsint code (0), -1 on error, or -2 when the grid set is not on the
 * single-launch k_sepx path / the shape is outside the one-pass encoder (use the unfused entry points then). */
/* ------------------------------------------------------------------------------------------ */
/* FST framing of a record the cfg5 pipeline left in HBM (fstd98.c:1164-1190)   */
/* ------------------------------------------------------------------------------------------ */
/* d_record = [4 compact_float header words (16-bit slots, style 2)][armn_compress stream of zlng bytes | the 16-bit-slot tokens when zlng < 0]: what
 * ezhip_ezsint_pack16_compress_batch_dev / ezhip_pack16_compress_dev leave per field.  d_data (device, cap_words) receives the DATA PART of the FST
 * record c_fstecr(datyp 129, nbits) writes for that field:
 *   zlng > 0   [nw][4 header words][stream], zero padded to nw = 2 ceil((16 + zlng) 8 / 64) words; *datyp_out = 129.  d_data + 1 == d_record frames in
 *              place: one word in front of the record is all it takes; any OTHER overlap of the two ranges is refused (the shift by one word would race).
 *              The up to two padding words behind the stream are ZERO here; c_fstecr (and the host path ezhip_fst_pack_data) leave there what the
 *              in-place compressor left of its token array -- bytes no reader looks at (c_fstluk hands armn_compress the stream, whose header holds its
 *              own extent), so a device-framed data part equals the reference's on its first 5 + ceil(zlng / 4) words, not necessarily on the padding;
 *   zlng < 0   compression did not pay: the reference packs again as datyp 1 -- header style 1 (3 words + 24 bits) and the nbits-wide tokens from bit
 *              120 on; the tokens are the same numbers, re-packed on the device; *datyp_out = 1.
 * Asynchronous on the current stream; returns the number of words of the data part, -1 on error. */
int32_t ezhip_fst_frame_record_dev(uint32_t *d_data, int64_t cap_words, const uint32_t *d_record, int32_t zlng, int32_t ni, int32_t nj, int32_t nk,
                                   int32_t nbits, int32_t *datyp_out)
{
    if (need_device("ezhip_fst_frame_record_dev")) return -1;
    const int64_t n = (int64_t)ni * nj * nk;
    if (n <= 0 || nbits < 1 || nbits > 16 || !d_data || !d_record) return -1;
    if (zlng > 0) {
        const int64_t nbytes = 16 + (int64_t)zlng, nw = 2 * ((nbytes * 8 + 63) / 64);
        if (nw > cap_words) { fprintf(stderr, "<ezhip_fst_frame_record_dev> the data part needs %lld words\n", (long long)nw); return -1; }
        int64_t ncopy = 4 + ((int64_t)zlng + 3) / 4;
        if (ncopy > nw - 1) ncopy = nw - 1;
        if (d_data + 1 != d_record) {                              /* not the in-place form: the ranges must be disjoint */
            const uintptr_t a0 = (uintptr_t)d_data, a1 = a0 + 4 * (uintptr_t)nw, b0 = (uintptr_t)d_record, b1 = b0 + 4 * (uintptr_t)ncopy;
            if (a0 < b1 && b0 < a1) { fprintf(stderr, "<ezhip_fst_frame_record_dev> d_data and d_record overlap (only d_data + 1 == d_record frames in place)\n"); return -1; }
        }
        if (packhip_fst_frame129(d_data, d_record, (unsigned)nw, (unsigned)ncopy)) return -1;
        if (datyp_out) *datyp_out = 129;
        return (int32_t)nw;
    }
    if (d_data + 1 == d_record || d_data == d_record) { fprintf(stderr, "<ezhip_fst_frame_record_dev> the datyp 1 form cannot be built in place\n"); return -1; }
    const int64_t nw = 2 * ((n * (nbits > 16 ? nbits : 16) + 128 + 32 + 63) / 64);      /* the record keeps the length computed for datyp 129 (fstd98.c:971-973: nw is not recomputed) */
    if (nw > cap_words) { fprintf(stderr, "<ezhip_fst_frame_record_dev> the data part needs %lld words\n", (long long)nw); return -1; }
    unsigned *d_tok32 = (unsigned *)ws(5, 4 * (size_t)n + 64);
    if (!d_tok32) return -1;
    if (ezhip_memset(d_data, 0, 4 * (size_t)nw)) return -1;
    if (packhip_fst_plain_from_slots(d_data, d_tok32, d_record, (size_t)n)) return -1;
    if (packhip_ci_pack(d_data + 3, d_tok32, (size_t)n, nbits, 24, 1, 0, 0, 0, 0, 0)) return -1;
    if (datyp_out) *datyp_out = 1;
    return (int32_t)nw;
}

int32_t ezhip_ezsint_batch_encode_dev(uint32_t *d_z, int64_t z_stride_words, int64_t z_cap_words, const float *d_zin, int32_t nfields,
                                      const void *d_params, int64_t param_stride_bytes, int32_t nbits, void *d_work, int32_t *d_zlng, int32_t poles_ready);
/* use_bb: pass A from bounds of the source windows (else the interpolating pass); use_fused: passes B + E in one launch (k_sepx_enc: the tokens never
 * reach HBM), else tokens to HBM and the one-pass encoder.  Both fall back by themselves when the grid set's plan does not qualify; fields a first
 * choice hands back (bound pass: no extremum stands out; fused launch: not compressible / wide differences / gave up) are redone without it. */
static int32_t cfg5_batch(void *d_records, int64_t record_stride_words, const float *d_zin, int32_t nfields,
                          int32_t ni_out, int32_t nj_out, int32_t nbits, int32_t *zlng_out, int use_bb, int use_fused)
{
    int32_t ni_in = 0, nj_in = 0, ni_set = 0, nj_set = 0;
    if (ezhip_current_set_dims(&ni_in, &nj_in, &ni_set, &nj_set)) return -1;
    const size_t n = (size_t)ni_out * nj_out;
    /* sub-batches (EZHIP_CFG5_SUB, development): the encoder of sub-batch k on the side stream beside passes A and B of sub-batch k + 1.
     * Measured SLOWER on MI355X (90.7 us per field for the whole batch of 32; 92.6 / 101.0 / 112.2 for sub-batches of 16 / 8 / 4: co-running
     * stretches k_sepx 2.5 x and the encoder 1.4 x -- both are bound by VALU issue, there is nothing to overlap): the default is one batch */
    int sub = nfields;
    { const char *e = getenv("EZHIP_CFG5_SUB"); const int v = e ? atoi(e) : 0; if (v >= 4 && v < nfields) sub = v; }
    const int nsub = (nfields + sub - 1) / sub;
    const size_t wb1 = packhip_armn_enc1_work_bytes(ni_out, nj_out, nbits, sub);
    if (!wb1) return -2;
    const size_t wb = (wb1 + 64 + 255) & ~(size_t)255;
    const size_t tok_stride = ((n + 1) / 2 + 4 + 3) & ~(size_t)3;               /* words per field, a multiple of 4: the one-pass encoder stages every field's tokens in aligned 16-byte pieces */
    const size_t stride = PACKHIP_STATS_PARTIALS + 3 * (size_t)8192;
    unsigned *d_stats_all = (unsigned *)ws(3, (4 * stride + sizeof(packhip_cf_params)) * (size_t)nfields + 64);
    /* MEASURED (MI355X, 32 cfg5 fields, profiles/r03_experiments.txt): the one-launch form moves half the bytes (tokens never reach HBM) and takes TWICE the
     * time of the two kernels (135 against 68 us per field: 49 KB of LDS per thread block = 3 waves per SIMD for an encoder that wants 6, a row group brings
     * all its source rows itself, the look-back of five chunks sits on the critical path of every thread block).  Kept as a switch: EZHIP_CFG5_FUSED=1 */
    if (nsub > 1 || !getenv("EZHIP_CFG5_FUSED")) use_fused = 0;
    const ezhip_sepenc_layout L = ezhip_sepenc_layout_of(ni_out, nj_out, nfields);
    unsigned *d_tok = use_fused ? NULL : (unsigned *)ws(1, 4 * tok_stride * (size_t)nfields);      /* the fused launch needs no token array (1.66 GB for 32 cfg5 fields) */
    char *d_work = (char *)ws(2, use_fused && L.total > wb * (size_t)nsub ? L.total : wb * (size_t)nsub);
    int *d_zlng = (int *)ws(0, 2 * sizeof(int) * (size_t)nfields + 64);           /* zlng[nfields] | flags[nfields] */
    if (!d_stats_all || (!d_tok && !use_fused) || !d_work || !d_zlng) return -1;
    int *d_flags = d_zlng + nfields;
    int fused_used = 0;
    packhip_cf_params *d_pp = (packhip_cf_params *)(d_stats_all + stride * (size_t)nfields);
    const size_t nin = (size_t)ni_in * nj_in;
    const size_t z_cap = (size_t)record_stride_words - 4 < n / 2 + 16 ? (size_t)record_stride_words - 4 : n / 2 + 16;
    if (ezhip_memset(d_zlng, 0xFD, sizeof(int) * (size_t)nfields)) return -1;       /* "unfinished" until the encoder says otherwise */
    if (ezhip_memset(d_flags, 0, sizeof(int) * (size_t)nfields)) return -1;
    int32_t rc = 0;
    int bb_used = 0;
    for (int k = 0; k < nsub; k++) {
        const int f0 = k * sub, nf = nfields - f0 < sub ? nfields - f0 : sub;
        unsigned *rec0 = (unsigned *)d_records + (size_t)f0 * (size_t)record_stride_words;
        unsigned *stats0 = d_stats_all + stride * (size_t)f0;
        unsigned *tok0 = d_tok ? d_tok + tok_stride * (size_t)f0 : NULL;
        int32_t npf = 0, r = -2;
        /* pass A: the field's extrema.  First choice: from bounds of the SOURCE windows + exact evaluation of the few that matter (no interpolation) */
        if (use_bb) { r = ezhip_ezsint_batch_minmax_bb_dev(d_zin + nin * (size_t)f0, nf, stats0 + PACKHIP_STATS_PARTIALS, (int64_t)stride, &npf, d_flags + f0); if (r == 0) bb_used = 1; }
        if (r == -2) r = ezhip_ezsint_batch_minmax_only_dev(d_zin + nin * (size_t)f0, nf, stats0 + PACKHIP_STATS_PARTIALS, (int64_t)stride, &npf);
        if (r < 0) { if (k) { ezhip_side_join(); ezhip_sync(); } return k ? -1 : r; }          /* -2: not on the single-launch path (nothing launched yet) */
        if (r > rc) rc = r;
        if (packhip_cf_header_batch(d_pp + f0, rec0, (size_t)record_stride_words, stats0, stride, npf, (unsigned)n, nbits, 2, 0, nf)) goto fail;
        if (use_fused) {
            /* passes B + E in one launch; -2: the plan has no fused geometry -> the two kernels below (nothing launched) */
            int32_t fr = ezhip_ezsint_batch_encode_dev(rec0 + 4, record_stride_words, (int64_t)z_cap, d_zin + nin * (size_t)f0, nf, d_pp + f0,
                                                       (int64_t)sizeof(packhip_cf_params), nbits, d_work, d_zlng + f0, bb_used);
            if (fr == 0) {
                if (packhip_sepenc_prefix(rec0 + 4, (size_t)record_stride_words, z_cap, (const unsigned short *)(d_work + L.off_ptok), L.ptok_stride,
                                          (const unsigned *)(d_work + L.off_head), d_zlng + f0, ni_out, nj_out, nbits, 4, nf)) goto fail;
                fused_used = 1;
                continue;
            }
            if (fr != -2) goto fail;
            use_fused = 0;
            d_tok = (unsigned *)ws(1, 4 * tok_stride * (size_t)nfields);
            if (!d_tok) goto fail;
            tok0 = d_tok + tok_stride * (size_t)f0;
        }
        if (ezhip_ezsint_batch_tokens_dev(tok0, (int64_t)tok_stride, d_zin + nin * (size_t)f0, nf, d_pp + f0, (int64_t)sizeof(packhip_cf_params)) < 0) goto fail;
        const int side = nsub > 1;
        if (side && ezhip_side_begin()) goto fail;
        int bad = packhip_armn_encode1(rec0 + 4, (size_t)record_stride_words, z_cap, tok0, tok_stride, ni_out, nj_out, nbits, nf, 1, 4, d_work + wb * (size_t)k, d_zlng + f0) != 0;
        bad = bad || packhip_cond_copy(rec0 + 4, (size_t)record_stride_words, tok0, tok_stride, (n + 1) / 2, d_zlng + f0, nf, 1);
        if (side && ezhip_side_end()) goto fail;
        if (bad) goto fail;
    }
    if (ezhip_side_join()) goto fail;
    {
        int32_t *h = (int32_t *)malloc(2 * sizeof(int32_t) * (size_t)nfields);
        if (!h) goto fail;
        if (ezhip_d2h(h, d_zlng, 2 * sizeof(int) * (size_t)nfields) || ezhip_sync()) { free(h); return -1; }
        memcpy(zlng_out, h, sizeof(int32_t) * (size_t)nfields);
        if (fused_used && EZH_DEVENV("EZHIP_CFG5_ONLY_FUSED")) { free(h); return rc; }      /* development: knock-out timings, results unchecked */
        if (fused_used) {
            /* the fused launch has no token array to fall back on: a field it could not finish (not compressible, differences beyond 16 bits, launch
             * gave up: zlng still "unfinished") goes through the two-kernel path, extrema pass included (runs of such fields in one call) */
            for (int f = 0; f < nfields; f++) {
                if (h[f] > 0 && !h[nfields + f]) continue;
                int f1 = f + 1;
                while (f1 < nfields && !(h[f1] > 0 && !h[nfields + f1])) f1++;
                if (getenv("EZHIP_VERBOSE")) fprintf(stderr, "cfg5: fields %d..%d redone with the two-kernel path (zlng %d, bound-pass flag %d)\n", f, f1 - 1, h[f], h[nfields + f]);
                int32_t r = cfg5_batch((unsigned *)d_records + (size_t)f * (size_t)record_stride_words, record_stride_words, d_zin + nin * (size_t)f, f1 - f,
                                       ni_out, nj_out, nbits, zlng_out + f, use_bb && !h[nfields + f], 0);
                if (r < 0) { free(h); return -1; }
                f = f1 - 1;
            }
            free(h);
            return rc;
        }
        if (redo_wide_fields((unsigned *)d_records, (size_t)record_stride_words, d_tok, tok_stride, 0, nfields, ni_out, nj_out, nbits, zlng_out)) { free(h); return -1; }
        /* fields the bound pass gave up on (flags: too many source windows could hold the extremum): the interpolating pass A for those */
        for (int f = 0; f < nfields && bb_used; f++) {
            if (!h[nfields + f]) continue;
            int f1 = f + 1;
            while (f1 < nfields && h[nfields + f1]) f1++;
            int32_t r = cfg5_batch((unsigned *)d_records + (size_t)f * (size_t)record_stride_words, record_stride_words, d_zin + nin * (size_t)f, f1 - f,
                                   ni_out, nj_out, nbits, zlng_out + f, 0, use_fused);
            if (r < 0) { free(h); return -1; }
            f = f1 - 1;
        }
        free(h);
    }
    return rc;
fail:
    ezhip_side_join(); ezhip_sync();
    return -1;
}

int32_t ezhip_ezsint_pack16_compress_batch_dev(void *d_records, int64_t record_stride_words, const float *d_zin, int32_t nfields,
                                               int32_t ni_out, int32_t nj_out, int32_t nbits, int32_t *zlng_out)
{
    if (need_device("ezhip_ezsint_pack16_compress_batch_dev")) return -1;
    /* the two k_sepx passes write by the CURRENT grid set's target dimensions: the caller's must be those */
    int32_t ni_in = 0, nj_in = 0, ni_set = 0, nj_set = 0;
    if (ezhip_current_set_dims(&ni_in, &nj_in, &ni_set, &nj_set)) { fprintf(stderr, "<ezhip_ezsint_pack16_compress_batch_dev> no grid set defined (c_ezdefset)\n"); return -1; }
    if (ni_out != ni_set || nj_out != nj_set) {
        fprintf(stderr, "<ezhip_ezsint_pack16_compress_batch_dev> ni_out x nj_out = %d x %d, but the current grid set writes %d x %d\n", ni_out, nj_out, ni_set, nj_set);
        return -1;
    }
    if (nfields < 1 || nbits < 5 || nbits > 16 || ni_out < 16 || nj_out < 16 || (ni_out & 1)) return -2;
    if (g_level == -1) g_level = 1;
    if (g_level == 0 || g_swap != 1) return -2;
    const size_t n = (size_t)ni_out * nj_out;
    if (record_stride_words < 4 || (size_t)record_stride_words < 4 + (n + 1) / 2) {
        fprintf(stderr, "<ezhip_ezsint_pack16_compress_batch_dev> record_stride_words = %lld, a record needs at least %zu words\n", (long long)record_stride_words, 4 + (n + 1) / 2);
        return -1;
    }
    return cfg5_batch(d_records, record_stride_words, d_zin, nfields, ni_out, nj_out, nbits, zlng_out, getenv("EZHIP_CFG5_NO_BB") ? 0 : 1, 1);
}
