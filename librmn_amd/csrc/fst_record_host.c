/*
 * fst_record_host.c -- the DATA PART of an FST record, as c_fstecr builds it and c_fstluk takes it apart
 * (reference src/fstd98/fstd98.c:935-1000 sizes, :1140-1370 the packing switch, :2270-2440 the unpacking switch), so that a field packed on
 * the MI355X can be handed to the file layer as a ready record and a record read from a file can be unpacked on the device.
 * Everything here is composition: the packers are the library's own entry points (compact_float, compact_integer, c_float_packer,
 * armn_compress, c_armn_compress32: HIP kernels behind them, no CPU fallback).  What is NOT here: the XDF file layer (directory,
 * keys, c_xdfput) -- SURVEY.md section 8f row 2 stops at the record.
 *
 * data[0 ..) is what c_fstecr stores from buffer->data[keys_len] on: for the "turbo" types (datyp > 128) the first word is the length of
 * the data part in 32-bit words, then the packer header(s), then the armn_compress stream.
 * Supported: datyp 0, 1, 129, 2, 130, 4, 5 (32 bits), 6, 134, 133 on 32-bit elements (REAL*4 / INTEGER*4 arrays).
 */
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/packers_hip.h"

#define W64TOWD(n) ((n) << 1)
static int imax(int a, int b) { return a > b ? a : b; }

/* words of the data part before packing (fstd98.c:955-995); *hs / *ss: c_float_packer header / stream words */
static int64_t plain_words(int datyp, int64_t n, int nbits, int *hs, int *ss)
{
    int32_t header_size = 0, stream_size = 0, p1, p2;
    int64_t nw;
    switch (datyp) {
    case 6:   c_float_packer_params(&header_size, &stream_size, &p1, &p2, (int32_t)n); nw = ((int64_t)(header_size + stream_size) * 8 + 63) / 64; break;
    case 134: c_float_packer_params(&header_size, &stream_size, &p1, &p2, (int32_t)n); nw = ((int64_t)(header_size + stream_size) * 8 + 32 + 63) / 64; break;
    case 129: nw = (n * imax(nbits, 16) + 128 + 32 + 63) / 64; break;
    case 130: nw = (n * imax(nbits, 16) + 32 + 63) / 64; break;
    default:  nw = (n * nbits + 120 + 63) / 64; break;
    }
    *hs = header_size / 4; *ss = stream_size / 4;
    return W64TOWD(nw);
}

int32_t ezhip_fst_pack_data(uint32_t *data, int64_t cap_words, void *field, int32_t ni, int32_t nj, int32_t nk, int32_t datyp, int32_t nbits,
                            int32_t *datyp_out, int32_t *nbits_out)
{
    const int64_t n = (int64_t)ni * nj * nk;
    if (n <= 0 || nbits < 1 || nbits > 64) return -1;
    /* fstd98.c:934: no extra compression beyond 16 bits (IEEE 133 aside): the turbo flag is dropped BEFORE the sizes and the switch */
    if (nbits > 16 && datyp != 133) datyp &= 0x7F;
    /* fstd98.c:936-951: float_packer beyond its range becomes IEEE or the classic packer */
    if (datyp == 6 && nbits > 24) { datyp = 5; nbits = 32; }
    if (datyp == 6 && nbits > 16) datyp = 1;
    int hs, ss;
    int64_t words = plain_words(datyp, n, nbits, &hs, &ss);
    if (words + 8 > cap_words) { fprintf(stderr, "<ezhip_fst_pack_data> the data part needs %lld words\n", (long long)(words + 8)); return -1; }
    memset(data, 0, (size_t)words * 4);
    int out_datyp = datyp;
    double tempfloat = 99999.0;
    switch (datyp) {
    case 0: {                                                                      /* transparent (:1142-1153) */
        const int64_t lngw = (n * nbits + 31) / 32;
        memcpy(data, field, (size_t)lngw * 4);
        break;
    }
    case 1: case 129:                                                               /* :1155-1190 */
        if (datyp > 128 && nbits <= 16) {
            if (!compact_float(field, &data[1], &data[5], (int)n, nbits + 64 * imax(16, nbits), 0, 1, 1, 0, &tempfloat)) return -1;
            const int zl = armn_compress((unsigned char *)&data[5], ni, nj, nk, nbits, 1);
            if (zl < 0) {
                out_datyp = 1;
                if (!compact_float(field, &data[0], &data[3], (int)n, nbits, 24, 1, 1, 0, &tempfloat)) return -1;
            } else {
                const int64_t nbytes = 16 + zl;
                words = W64TOWD((nbytes * 8 + 63) / 64);
                data[0] = (uint32_t)words;
            }
        } else if (!compact_float(field, &data[0], &data[3], (int)n, nbits, 24, 1, 1, 0, &tempfloat)) return -1;
        break;
    case 2: case 130: {                                                             /* :1192-1243, INTEGER*4 elements */
        const int offset = datyp > 128 ? 1 : 0;
        if (datyp > 128) {
            const int32_t mask = nbits >= 32 ? -1 : ~(-1 << nbits);
            int16_t *p16 = (int16_t *)&data[offset];
            const int32_t *p32 = (const int32_t *)field;
            for (int64_t i = 0; i < n; i++) p16[i] = (int16_t)(p32[i] & mask);      /* memcpy_32_16 */
            c_armn_compress_setswap(0);
            const int zl = armn_compress((unsigned char *)&data[offset], ni, nj, nk, nbits, 1);
            c_armn_compress_setswap(1);
            if (zl < 0) {
                out_datyp = 2;                                                      /* (the 16-bit copy stays behind the packed integers, as in the reference) */
                if (!compact_integer(field, NULL, &data[offset], (int)n, nbits, 0, 1, 1)) return -1;      /* lands one word in (sic: the reader of datyp 2 starts at word 0) */
            } else {
                const int64_t nbytes = 4 + zl;
                words = W64TOWD((nbytes * 8 + 63) / 64);
                data[0] = (uint32_t)words;
            }
        } else if (!compact_integer(field, NULL, &data[0], (int)n, nbits, 0, 1, 1)) return -1;
        break;
    }
    case 4:                                                                         /* :1263-1296 */
        if (!compact_integer(field, NULL, &data[0], (int)n, nbits, 0, 1, 3)) return -1;
        break;
    case 5:                                                                         /* IEEE: only the 32-bit form (ieeepak is a copy there) */
        if (nbits != 32) { fprintf(stderr, "<ezhip_fst_pack_data> datyp 5 is supported with 32 bits only\n"); return -1; }
        memcpy(data, field, (size_t)n * 4);
        break;
    case 133: {                                                                     /* :1307-1321 */
        const int zl = c_armn_compress32((unsigned char *)&data[1], (float *)field, ni, nj, nk, nbits);
        if (zl < 0) {
            if (nbits != 32) { fprintf(stderr, "<ezhip_fst_pack_data> datyp 133 fell back to datyp 5, which is supported with 32 bits only\n"); return -1; }
            out_datyp = 5;
            memcpy(data, field, (size_t)n * 4);
        } else {
            const int64_t nbytes = 16 + zl;
            words = W64TOWD((nbytes * 8 + 63) / 64);
            data[0] = (uint32_t)words;
        }
        break;
    }
    case 6: case 134:                                                               /* :1329-1348 */
        if (datyp > 128 && nbits <= 16) {
            if (c_float_packer((float *)field, nbits, (int32_t *)&data[1], (int32_t *)&data[1 + hs], (int32_t)n)) return -1;
            const int zl = armn_compress((unsigned char *)&data[1 + hs], ni, nj, nk, nbits, 1);
            if (zl < 0) {
                out_datyp = 6;
                if (c_float_packer((float *)field, nbits, (int32_t *)&data[0], (int32_t *)&data[hs], (int32_t)n)) return -1;
            } else {
                const int64_t nbytes = 16 + (int64_t)hs * 4 + zl;
                words = W64TOWD((nbytes * 8 + 63) / 64);
                data[0] = (uint32_t)words;
            }
        } else if (c_float_packer((float *)field, nbits, (int32_t *)&data[0], (int32_t *)&data[hs], (int32_t)n)) return -1;
        break;
    default:
        fprintf(stderr, "<ezhip_fst_pack_data> datyp %d is outside the MI355X hot-path scope\n", datyp);
        return -1;
    }
    if (datyp_out) *datyp_out = out_datyp;
    if (nbits_out) *nbits_out = nbits;
    return (int32_t)words;
}

/* the read side (fstd98.c:2270-2440): data part -> field (32-bit elements).  The armn_compress streams are expanded IN PLACE in `data`, like
 * the reference does in its record buffer: data must be the record buffer sized by ezhip_fst_pack_data's plain word count.  Returns 0 / -1. */
int32_t ezhip_fst_unpack_data(void *field, uint32_t *data, int32_t ni, int32_t nj, int32_t nk, int32_t datyp, int32_t nbits)
{
    const int64_t n = (int64_t)ni * nj * nk;
    if (n <= 0) return -1;
    double tempfloat = 99999.0;
    int hs, ss;
    (void)plain_words(datyp == 134 ? 134 : 6, n, nbits, &hs, &ss);
    switch (datyp) {
    case 0: memcpy(field, data, (size_t)((n * nbits + 31) / 32) * 4); return 0;
    case 1: return compact_float(field, data, data + 3, (int)n, nbits, 24, 1, 2, 0, &tempfloat) ? 0 : -1;
    case 129:
        if (armn_compress((unsigned char *)(data + 5), ni, nj, nk, nbits, 2) < 0) return -1;
        return compact_float(field, data + 1, data + 5, (int)n, nbits + 64 * imax(16, nbits), 0, 1, 2, 0, &tempfloat) ? 0 : -1;
    case 2: return compact_integer(field, NULL, data, (int)n, nbits, 0, 1, 2) ? 0 : -1;
    case 130: {
        c_armn_compress_setswap(0);
        const int nb = armn_compress((unsigned char *)(data + 1), ni, nj, nk, nbits, 2);
        c_armn_compress_setswap(1);
        if (nb < 0) return -1;
        const int16_t mask = (int16_t)(nbits >= 16 ? -1 : ~(-1 << nbits));
        const int16_t *p16 = (const int16_t *)(data + 1);
        int32_t *p32 = (int32_t *)field;
        for (int64_t i = 0; i < n; i++) p32[i] = p16[i] & mask;                  /* memcpy_16_32 */
        return 0;
    }
    case 4: return compact_integer(field, NULL, data, (int)n, nbits, 0, 1, 4) ? 0 : -1;
    case 5: if (nbits != 32) return -1; memcpy(field, data, (size_t)n * 4); return 0;
    case 133: return c_armn_uncompress32((float *)field, (unsigned char *)(data + 1), ni, nj, nk, nbits) < 0 ? -1 : 0;
    case 6: { int32_t nb; return c_float_unpacker((float *)field, (int32_t *)data, (int32_t *)(data + hs), (int32_t)n, &nb); }
    case 134: {
        int32_t nb;
        if (armn_compress((unsigned char *)(data + 1 + hs), ni, nj, nk, nbits, 2) < 0) return -1;
        return c_float_unpacker((float *)field, (int32_t *)(data + 1), (int32_t *)(data + 1 + hs), (int32_t)n, &nb);
    }
    default:
        fprintf(stderr, "<ezhip_fst_unpack_data> datyp %d is outside the MI355X hot-path scope\n", datyp);
        return -1;
    }
}
