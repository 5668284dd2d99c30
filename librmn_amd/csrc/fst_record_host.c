/*
 * fst_record_host.c -- the DATA PART of an FST record, as c_fstecr builds it and c_fstluk takes it apart
 * (reference src/fstd98/fstd98.c:808-826 flags, :935-1000 sizes, :1133-1370 the packing switch, :2169-2183 and :2270-2466 the unpacking switch),
 * so that a field packed on the MI355X can be handed to the file layer as a ready record and a record read from a file can be unpacked on the
 * device.  Everything here is composition: the packers are the library's own entry points (compact_float / compact_double, compact_integer /
 * compact_short / compact_char, c_float_packer, armn_compress, c_armn_compress32: HIP kernels behind them, no CPU fallback).  What is NOT here:
 * the XDF file layer (directory, keys, c_xdfput) -- SURVEY.md section 8f row 2 stops at the record.
 *
 * data[0 ..) is what c_fstecr stores from buffer->data[keys_len] on: for the "turbo" types (datyp > 128) the first word is the length of
 * the data part in 32-bit words, then the packer header(s), then the armn_compress stream.
 *
 * Element sizes: the reference learns them from the Fortran interface that was called (xdf_byte / xdf_short / xdf_double, fstd98.c:1133-1136);
 * here the caller says it: elem_bytes = 1, 2, 4 or 8.  Supported: datyp 0, 1, 129 (REAL and REAL*8 arrays), 2, 130 (32-, 16-, 8-bit elements),
 * 4 (the same), 5 (32 and 64 bits), 6, 134, 133, each with or without the missing-value flag (+64, src/fstd98/fst_missing.c).
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/packers_hip.h"

#define W64TOWD(n) ((n) << 1)
int32_t ezhip_pack16_compress_host(uint32_t *record, int64_t (*out_words_of)(int32_t zlng), const float *field, int32_t ni, int32_t nj, int32_t nbits);
int32_t ezhip_uncompress_unpack16_host(float *field, const uint32_t *record, int64_t record_words, int32_t ni, int32_t nj, int32_t nbits);
/* words of a datyp-129 data part behind its length word: 16 header bytes + zlng stream bytes, padded to 64 bits (fstd98.c:1178-1180), minus the length word */
static int64_t fst129_words_after_length(int32_t zlng) { return ((((int64_t)16 + zlng) * 8 + 63) / 64) * 2 - 1; }
static int imax(int a, int b) { return a > b ? a : b; }
static int imin(int a, int b) { return a < b ? a : b; }

/* ------------------------------------------------------------------------------------------ */
/* missing values (fst_missing.c): "magic" values of the caller's arrays become a value above the field's maximum before packing, and the         */
/* field's maximum becomes the magic value again after unpacking                                                                                 */
/* ------------------------------------------------------------------------------------------ */
static struct {
    int mode;                                /* -1 not initialised, 0 off, 1 on (fst_missing.c:186) */
    float f; int i; unsigned ui; double d; short s; unsigned short us; signed char b; unsigned char ub;
} MV = {-1, -1.0E+38f, (int)0x80000000u, 0xFFFFFFFFu, -1.0E+38, (short)0x8000, 0xFFFF, (signed char)0x80, 0xFF};

/* fst_missing.c:245-305: the mode is on when MISSING_VALUE_FLAGS is set (its eight numbers replace the defaults); plug-in libraries
 * (MISSING_VALUE_PLUGINS) are host callbacks of the reference's file layer and are not loaded here */
int32_t ezhip_fst_missing_value_used(void)
{
    if (MV.mode == -1) {
        const char *t = getenv("MISSING_VALUE_FLAGS");
        if (!t) MV.mode = 0;
        else { MV.mode = 1; sscanf(t, "%g %d %u %lg %hd %hu %hhd %hhu", &MV.f, &MV.i, &MV.ui, &MV.d, &MV.s, &MV.us, &MV.b, &MV.ub); }
        if (getenv("MISSING_VALUE_PLUGINS")) fprintf(stderr, "<ezhip_fst> MISSING_VALUE_PLUGINS is not supported by the MI355X record path: the built-in encoders are used\n");
    }
    return MV.mode;
}
int32_t ezhip_fst_force_missing_value_usage(int32_t enable) { return MV.mode = enable ? 1 : 0; }       /* ForceMissingValueUsage, fst_missing.c:309-315 */
void ezhip_fst_set_missing_value_flags(const float *f, const int32_t *i, const uint32_t *ui, const double *d, const int16_t *s, const uint16_t *us,
                                       const int8_t *b, const uint8_t *ub)
{   /* set_missing_value_flags, fst_missing.c:100-111 */
    if (f) MV.f = *f;
    if (i) MV.i = *i;
    if (ui) MV.ui = *ui;
    if (d) MV.d = *d;
    if (s) MV.s = *s;
    if (us) MV.us = *us;
    if (b) MV.b = *b;
    if (ub) MV.ub = *ub;
}

/* extremes of the values that are not the magic value, count of those that are (fld_*_anal, fst_missing.c:465-732: a leading run of magic values is
 * skipped by re-seeding; a field of nothing but magic values reports the magic value as both extremes) */
#define MV_SCAN(NAME, T, MAGIC)                                                                      \
static int NAME(const T *a, int n, T *vmax, T *vmin)                                                 \
{                                                                                                    \
    int count = a[0] == (MAGIC) ? 1 : 0;                                                             \
    T hi = a[0], lo = a[0];                                                                          \
    for (int k = 1; k < n; k++) {                                                                    \
        if (hi == (MAGIC)) hi = a[k];                                                                \
        if (lo == (MAGIC)) lo = a[k];                                                                \
        if (a[k] == (MAGIC)) count++;                                                                \
        else { if (a[k] < lo) lo = a[k]; if (a[k] > hi) hi = a[k]; }                                 \
    }                                                                                                \
    *vmax = hi; *vmin = lo;                                                                          \
    return count;                                                                                    \
}
MV_SCAN(scan_f, float, MV.f)
MV_SCAN(scan_d, double, MV.d)
MV_SCAN(scan_i, int, MV.i)
MV_SCAN(scan_s, short, MV.s)
MV_SCAN(scan_b, signed char, MV.b)
MV_SCAN(scan_ui, unsigned, MV.ui)
MV_SCAN(scan_us, unsigned short, MV.us)
MV_SCAN(scan_ub, unsigned char, MV.ub)

/* after unpacking: the field's maximum is the magic value again -- from element 1 on (sic: fst_*_decode_missing loop from i = 1, :745, :762 ...) */
#define MV_RESTORE(NAME, T, SCAN, MAGIC)                                                             \
static void NAME(T *a, int n)                                                                        \
{                                                                                                    \
    T hi, lo;                                                                                        \
    (void)SCAN(a, n, &hi, &lo);                                                                      \
    for (int k = 1; k < n; k++) if (a[k] == hi) a[k] = (MAGIC);                                      \
}
MV_RESTORE(restore_f, float, scan_f, MV.f)
MV_RESTORE(restore_d, double, scan_d, MV.d)
MV_RESTORE(restore_i, int, scan_i, MV.i)
MV_RESTORE(restore_s, short, scan_s, MV.s)
MV_RESTORE(restore_b, signed char, scan_b, MV.b)
MV_RESTORE(restore_ui, unsigned, scan_ui, MV.ui)
MV_RESTORE(restore_us, unsigned short, scan_us, MV.us)
MV_RESTORE(restore_ub, unsigned char, scan_ub, MV.ub)

/* the stand-in of a real field: a little above the maximum, by a share of the range that survives quantisation to nbits (fst_missing.c:884-951) */
static const float MV_SHARE[9] = {2.0f, 1.1f, 0.55f, 0.28f, 0.14f, 0.07f, 0.035f, 0.017f, 0.01f};
static int plug_f(float *dst, const float *src, int n, int nbits)
{
    float hi, lo;
    const int count = scan_f(src, n, &hi, &lo);
    if (!count) return 0;
    float standin = nbits > 8 ? (float)(hi + (hi - lo) * 0.01) : hi + (hi - lo) * MV_SHARE[nbits < 0 ? 0 : nbits];
    if (standin == hi) standin = hi == 0.0f ? 1.0f : (float)(hi * 2.0);          /* a constant field */
    for (int k = 0; k < n; k++) dst[k] = src[k] == MV.f ? standin : src[k];
    return count;
}
static int plug_d(double *dst, const double *src, int n, int nbits)
{
    double hi, lo;
    const int count = scan_d(src, n, &hi, &lo);
    if (!count) return 0;
    double standin = nbits > 8 ? hi + (hi - lo) * 0.01 : hi + (hi - lo) * MV_SHARE[nbits < 0 ? 0 : nbits];
    if (standin == hi) standin = hi == 0.0 ? 1.0 : hi * 2.0;
    for (int k = 0; k < n; k++) dst[k] = src[k] == MV.d ? standin : src[k];
    return count;
}
/* integers: maximum + 1 when that still fits nbits, else the largest value of nbits bits stays (with the reference's warning).  The signed short and
 * byte forms start from an all-ones SIGNED pattern whose right shifts stay -1 (fst_missing.c:1037-1050, :1100-1113): the stand-in is -1 unless the
 * maximum is below -1 -- kept, records must equal the reference's */
#define MV_PLUG_INT(NAME, T, SCAN, MAGIC, START, FULL, HALVE)                                             \
static int NAME(T *dst, const T *src, int n, int nbits)                                              \
{                                                                                                    \
    T hi, lo;                                                                                        \
    const int count = SCAN(src, n, &hi, &lo);                                                        \
    if (!count) return 0;                                                                            \
    T standin = START;                                                                               \
    if (nbits < FULL) standin = (T)(standin >> (FULL - nbits));                                      \
    if (HALVE) standin = (T)(standin >> 1);                                                          \
    if (standin > hi) standin = (T)(hi + 1);                                                         \
    else fprintf(stderr, "<ezhip_fst_encode_missing_value> maximum value >= encoded missing value flag\n"); \
    for (int k = 0; k < n; k++) dst[k] = src[k] == (MAGIC) ? standin : src[k];                       \
    return count;                                                                                    \
}
static int plug_i(int *dst, const int *src, int n, int nbits)
{   /* the 32-bit signed form computes its pattern UNSIGNED (fst_missing.c:975-983): 2^(nbits-1) - 1 */
    int hi, lo;
    const int count = scan_i(src, n, &hi, &lo);
    if (!count) return 0;
    unsigned pattern = 0xFFFFFFFFu;
    if (nbits < 32) pattern >>= (32 - nbits);
    pattern >>= 1;
    int standin = (int)pattern;
    if (standin > hi) standin = hi + 1;
    else fprintf(stderr, "<ezhip_fst_encode_missing_value> maximum value >= encoded missing value flag\n");
    for (int k = 0; k < n; k++) dst[k] = src[k] == MV.i ? standin : src[k];
    return count;
}
MV_PLUG_INT(plug_s, short, scan_s, MV.s, (short)0xFFFF, 16, 1)
MV_PLUG_INT(plug_b, signed char, scan_b, MV.b, (signed char)0xFF, 8, 1)
MV_PLUG_INT(plug_ui, unsigned, scan_ui, MV.ui, 0xFFFFFFFFu, 32, 0)
MV_PLUG_INT(plug_us, unsigned short, scan_us, MV.us, (unsigned short)0xFFFF, 16, 0)
MV_PLUG_INT(plug_ub, unsigned char, scan_ub, MV.ub, (unsigned char)0xFF, 8, 0)

/* EncodeMissingValue (fst_missing.c:1472-1538): dst := src with the magic values replaced; returns their count (0: nothing done, dst untouched) */
int32_t ezhip_fst_encode_missing_value(void *dst, const void *src, int32_t n, int32_t datatype, int32_t nbits, int32_t is_byte, int32_t is_short, int32_t is_double)
{
    if (ezhip_fst_missing_value_used() == 0 || n < 1) return 0;
    const int kind = datatype & 0xF;
    if (kind == 1 || kind == 5 || kind == 6) {
        if (is_double || (kind == 5 && nbits == 64)) return plug_d((double *)dst, (const double *)src, n, nbits);
        if (nbits > 32) return 0;
        return plug_f((float *)dst, (const float *)src, n, nbits);
    }
    if (kind == 4) return is_short ? plug_s((short *)dst, (const short *)src, n, nbits) : is_byte ? plug_b((signed char *)dst, (const signed char *)src, n, nbits)
                                                                                                   : plug_i((int *)dst, (const int *)src, n, nbits);
    if (kind == 2) return is_short ? plug_us((unsigned short *)dst, (const unsigned short *)src, n, nbits)
                        : is_byte ? plug_ub((unsigned char *)dst, (const unsigned char *)src, n, nbits) : plug_ui((unsigned *)dst, (const unsigned *)src, n, nbits);
    return 0;                                                                       /* transparent, character, complex types */
}
/* DecodeMissingValue (fst_missing.c:1579-1640) */
void ezhip_fst_decode_missing_value(void *field, int32_t n, int32_t datatype, int32_t is_byte, int32_t is_short, int32_t is_double)
{
    if (ezhip_fst_missing_value_used() == 0 || n < 1) return;
    const int kind = datatype & 0xF;
    if (kind == 1 || kind == 5 || kind == 6) { if (is_double) restore_d((double *)field, n); else restore_f((float *)field, n); }
    else if (kind == 4) { if (is_short) restore_s((short *)field, n); else if (is_byte) restore_b((signed char *)field, n); else restore_i((int *)field, n); }
    else if (kind == 2) { if (is_short) restore_us((unsigned short *)field, n); else if (is_byte) restore_ub((unsigned char *)field, n); else restore_ui((unsigned *)field, n); }
}

/* ------------------------------------------------------------------------------------------ */
/* the packing switch                                                                            */
/* ------------------------------------------------------------------------------------------ */
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

int32_t ezhip_fst_pack_data_ex(uint32_t *data, int64_t cap_words, void *field_in, int32_t ni, int32_t nj, int32_t nk, int32_t datyp_ori, int32_t nbits,
                               int32_t elem_bytes, int32_t *datyp_out, int32_t *nbits_out)
{
    const int64_t n = (int64_t)ni * nj * nk;
    if (n <= 0 || n > 0x7FFFFFFF || nbits < 1 || nbits > 64) return -1;
    if (elem_bytes != 1 && elem_bytes != 2 && elem_bytes != 4 && elem_bytes != 8) { fprintf(stderr, "<ezhip_fst_pack_data> elem_bytes must be 1, 2, 4 or 8\n"); return -1; }
    const int is_byte = elem_bytes == 1, is_short = elem_bytes == 2;
    int is_missing = datyp_ori & 64;                                               /* fstd98.c:808-811 */
    int datyp = datyp_ori & 0xFFBF;
    if ((datyp & 0xF) == 8 || (datyp & 0xF) == 3 || (datyp & 0xF) == 7) { fprintf(stderr, "<ezhip_fst_pack_data> datyp %d is outside the MI355X hot-path scope\n", datyp_ori); return -1; }
    const int is_double = elem_bytes == 8 && (datyp & 0xF) != 5;                  /* REAL*8 arrays through the xdf_double interface; 8-byte IEEE is its own case (IEEE_64) */
    const int in_datyp = datyp;
    if (datyp == 133 && nbits > 32) datyp = 5;                                     /* :868-873 */
    if (in_datyp == 1 && (nbits == 31 || nbits == 32)) { datyp = 5; nbits = 32; } /* R32 -> E32, :875-880 */
    const int ieee64 = (in_datyp & 0xF) == 5 && nbits == 64;
    if (elem_bytes == 8 && !is_double && !ieee64) { fprintf(stderr, "<ezhip_fst_pack_data> 8-byte elements need datyp 1 / 129 (REAL*8) or datyp 5 with 64 bits\n"); return -1; }
    if ((is_byte || is_short) && (datyp & 0xF) != 2 && (datyp & 0xF) != 4) { fprintf(stderr, "<ezhip_fst_pack_data> 1- and 2-byte elements are integer types (datyp 2, 130, 4)\n"); return -1; }
    /* no extra compression beyond 16 bits (IEEE 133 aside): the turbo flag is dropped BEFORE the sizes and the switch (:934) */
    if (nbits > 16 && datyp != 133) datyp &= 0x7F;
    if (datyp == 6 && nbits > 24) { datyp = 5; nbits = 32; }                       /* :936-951 */
    if (datyp == 6 && nbits > 16) datyp = 1;
    int hs, ss;
    int64_t words = plain_words(datyp, n, nbits, &hs, &ss);
    if (ieee64) words = W64TOWD(n);
    if (words + 8 > cap_words) { fprintf(stderr, "<ezhip_fst_pack_data> the data part needs %lld words\n", (long long)(words + 8)); return -1; }
    memset(data, 0, (size_t)words * 4);
    /* the magic values of the caller's array become packable stand-ins in a copy (:1133-1145); no magic value found: the flag is cancelled */
    void *field = field_in, *copy = NULL;
    if (is_missing) {
        copy = malloc((size_t)n * (size_t)elem_bytes);
        if (!copy) return -1;
        if (ezhip_fst_encode_missing_value(copy, field_in, (int32_t)n, in_datyp, nbits, is_byte, is_short, is_double || ieee64) == 0) { free(copy); copy = NULL; is_missing = 0; }
        else field = copy;
    }
    int out_datyp = datyp | is_missing;
    int32_t rc = -1;
    double tempfloat = 99999.0;
    void *(*packfunc)(void *, void *, void *, const int, const int, const int, const int, const int, const int, const void *const) = is_double ? compact_double : compact_float;
    switch (datyp) {
    case 0: {                                                                      /* transparent (:1147-1159) */
        const int64_t lngw = (n * nbits + 31) / 32;
        memcpy(data, field, (size_t)lngw * 4);
        break;
    }
    case 1: case 129:                                                               /* :1161-1190 */
        if (datyp > 128 && nbits <= 16 && !is_double && nk == 1 && ni > 1 && nj > 1 && !getenv("EZHIP_FST_FOUR_TRIPS")) {
            /* REAL field, one level: the field goes up once and the data part comes down once (compact_float + armn_compress on host arrays moved it
             * over PCIe four times); the words behind the stream are what the in-place compressor leaves there (token words), as before */
            const int zl = ezhip_pack16_compress_host(&data[1], fst129_words_after_length, (const float *)field, ni, nj, nbits);
            if (zl == -2) goto done;                                                /* an error underneath, not armn_compress's "does not pay": the record's type is not changed over it */
            if (zl < 0) {
                out_datyp = 1;                                                      /* (sic: the missing-value flag is lost with the turbo flag, :1173) */
                if (!packfunc(field, &data[0], &data[3], (int)n, nbits, 24, 1, 1, 0, &tempfloat)) goto done;
            } else {
                const int64_t nbytes = 16 + zl;
                words = W64TOWD((nbytes * 8 + 63) / 64);
                data[0] = (uint32_t)words;
            }
        } else if (datyp > 128 && nbits <= 16) {
            if (!packfunc(field, &data[1], &data[5], (int)n, nbits + 64 * imax(16, nbits), 0, 1, 1, 0, &tempfloat)) goto done;
            const int zl = armn_compress((unsigned char *)&data[5], ni, nj, nk, nbits, 1);
            if (zl < 0) {
                out_datyp = 1;                                                      /* (sic: the missing-value flag is lost with the turbo flag, :1173) */
                if (!packfunc(field, &data[0], &data[3], (int)n, nbits, 24, 1, 1, 0, &tempfloat)) goto done;
            } else {
                const int64_t nbytes = 16 + zl;
                words = W64TOWD((nbytes * 8 + 63) / 64);
                data[0] = (uint32_t)words;
            }
        } else if (!packfunc(field, &data[0], &data[3], (int)n, nbits, 24, 1, 1, 0, &tempfloat)) goto done;
        break;
    case 2: case 130: {                                                             /* :1192-1243 */
        const int offset = datyp > 128 ? 1 : 0;
        if (is_short) nbits = imin(16, nbits);                                      /* :1199, :1226 */
        if (is_byte) nbits = imin(8, nbits);
        if (datyp > 128) {
            int16_t *p16 = (int16_t *)&data[offset];
            if (is_short) memcpy(p16, field, (size_t)n * 2);                        /* :1201 */
            else if (is_byte) { const int8_t *p8 = (const int8_t *)field; for (int64_t i = 0; i < n; i++) p16[i] = p8[i]; }      /* memcpy_8_16 */
            else {
                const int32_t mask = nbits >= 32 ? -1 : (int32_t)~(~0u << nbits);
                const int32_t *p32 = (const int32_t *)field;
                for (int64_t i = 0; i < n; i++) p16[i] = (int16_t)(p32[i] & mask);  /* memcpy_32_16 */
            }
            c_armn_compress_setswap(0);
            const int zl = armn_compress((unsigned char *)&data[offset], ni, nj, nk, nbits, 1);
            c_armn_compress_setswap(1);
            if (zl < 0) {
                out_datyp = 2;                                                      /* (the 16-bit copy stays behind the packed integers, as in the reference) */
                /* the reference hands the caller's array to compact_integer whatever its element size (:1212-1214: it reads a short or byte array as
                 * INTEGER*4, beyond its end); the packer of the element size is used here, at the same place: one word in (sic: the reader of datyp 2
                 * starts at word 0) */
                const int ok = is_short ? compact_short(field, NULL, &data[offset], (int)n, nbits, 0, 1, 5)
                             : is_byte ? compact_char(field, NULL, &data[offset], (int)n, nbits, 0, 1, 9) : compact_integer(field, NULL, &data[offset], (int)n, nbits, 0, 1, 1);
                if (!ok) goto done;
            } else {
                const int64_t nbytes = 4 + zl;
                words = W64TOWD((nbytes * 8 + 63) / 64);
                data[0] = (uint32_t)words;
            }
        } else {
            const int ok = is_short ? compact_short(field, NULL, &data[0], (int)n, nbits, 0, 1, 5)
                         : is_byte ? compact_char(field, NULL, &data[0], (int)n, nbits, 0, 1, 9) : compact_integer(field, NULL, &data[0], (int)n, nbits, 0, 1, 1);
            if (!ok) goto done;
        }
        break;
    }
    case 4: {                                                                       /* :1263-1281 as compiled (use_old_signed_pack_unpack_code, fstd98.c:22): short / byte elements are
                                                                                       sign-extended into INTEGER*4 and go through compact_integer's signed form */
        out_datyp = is_missing | 4;
        int32_t *wide = NULL;
        if (is_short || is_byte) {
            wide = (int32_t *)malloc((size_t)n * 4);
            if (!wide) goto done;
            if (is_short) { const int16_t *s = (const int16_t *)field; for (int64_t i = 0; i < n; i++) wide[i] = s[i]; }
            else { const int8_t *b = (const int8_t *)field; for (int64_t i = 0; i < n; i++) wide[i] = b[i]; }
        }
        const int ok = compact_integer(wide ? (void *)wide : field, NULL, &data[0], (int)n, nbits, 0, 1, 3);
        free(wide);
        if (!ok) goto done;
        break;
    }
    case 5:                                                                         /* IEEE (:1299-1327): ieeepak moves 32- or 64-bit elements unchanged on this byte order */
        if (nbits == 32) memcpy(data, field, (size_t)n * 4);
        else if (nbits == 64 && elem_bytes == 8) {                                  /* the two words of a REAL*8 change places (the file holds big-endian pairs) */
            const uint32_t *s = (const uint32_t *)field;
            for (int64_t i = 0; i < n; i++) { data[2 * i] = s[2 * i + 1]; data[2 * i + 1] = s[2 * i]; }
        } else if (nbits < 32 && elem_bytes == 4) {                                 /* the top nbits of every REAL, packed MSB first (ieeepak.F90: RSHIFT(IFLD, 32 - NBITS)) */
            uint32_t *tok = (uint32_t *)malloc((size_t)n * 4);
            if (!tok) goto done;
            const uint32_t *s = (const uint32_t *)field;
            for (int64_t i = 0; i < n; i++) tok[i] = s[i] >> (32 - nbits);
            const int ok = compact_integer(tok, NULL, &data[0], (int)n, nbits, 0, 1, 1);
            free(tok);
            if (!ok) goto done;
        } else { fprintf(stderr, "<ezhip_fst_pack_data> datyp 5: REAL elements with up to 32 bits, REAL*8 elements with 64 bits\n"); goto done; }
        break;
    case 133: {                                                                     /* :1307-1321 */
        const int zl = c_armn_compress32((unsigned char *)&data[1], (float *)field, ni, nj, nk, nbits);
        if (zl < 0) {
            if (nbits != 32) { fprintf(stderr, "<ezhip_fst_pack_data> datyp 133 fell back to datyp 5, which is supported with 32 bits only\n"); goto done; }
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
            if (c_float_packer((float *)field, nbits, (int32_t *)&data[1], (int32_t *)&data[1 + hs], (int32_t)n)) goto done;
            const int zl = armn_compress((unsigned char *)&data[1 + hs], ni, nj, nk, nbits, 1);
            if (zl < 0) {
                out_datyp = 6;
                if (c_float_packer((float *)field, nbits, (int32_t *)&data[0], (int32_t *)&data[hs], (int32_t)n)) goto done;
            } else {
                const int64_t nbytes = 16 + (int64_t)hs * 4 + zl;
                words = W64TOWD((nbytes * 8 + 63) / 64);
                data[0] = (uint32_t)words;
            }
        } else if (c_float_packer((float *)field, nbits, (int32_t *)&data[0], (int32_t *)&data[hs], (int32_t)n)) goto done;
        break;
    default:
        fprintf(stderr, "<ezhip_fst_pack_data> datyp %d is outside the MI355X hot-path scope\n", datyp);
        goto done;
    }
    if (datyp_out) *datyp_out = out_datyp;
    if (nbits_out) *nbits_out = nbits;
    rc = (int32_t)words;
done:
    free(copy);
    return rc;
}

int32_t ezhip_fst_pack_data(uint32_t *data, int64_t cap_words, void *field, int32_t ni, int32_t nj, int32_t nk, int32_t datyp, int32_t nbits,
                            int32_t *datyp_out, int32_t *nbits_out)
{
    return ezhip_fst_pack_data_ex(data, cap_words, field, ni, nj, nk, datyp, nbits, 4, datyp_out, nbits_out);
}

/* the read side (fstd98.c:2169-2183, :2270-2466): data part -> field of elem_bytes-sized elements.  The armn_compress streams are expanded IN PLACE in
 * `data`, like the reference does in its record buffer: data must be the record buffer sized by ezhip_fst_pack_data's plain word count.  datyp may carry
 * the missing-value flag (+64): the field's maximum becomes the magic value again.  Returns 0 / -1. */
int32_t ezhip_fst_unpack_data_ex(void *field, uint32_t *data, int32_t ni, int32_t nj, int32_t nk, int32_t datyp_in, int32_t nbits, int32_t elem_bytes)
{
    const int64_t n = (int64_t)ni * nj * nk;
    if (n <= 0 || n > 0x7FFFFFFF) return -1;
    if (elem_bytes != 1 && elem_bytes != 2 && elem_bytes != 4 && elem_bytes != 8) return -1;
    const int is_byte = elem_bytes == 1, is_short = elem_bytes == 2;
    const int has_missing = datyp_in & 64, datyp = datyp_in & 0xBF;
    const int is_double = elem_bytes == 8 && (datyp & 0xF) != 5;
    double tempfloat = 99999.0;
    int hs, ss;
    (void)plain_words(datyp == 134 ? 134 : 6, n, nbits, &hs, &ss);
    void *(*packfunc)(void *, void *, void *, const int, const int, const int, const int, const int, const int, const void *const) = is_double ? compact_double : compact_float;
    int32_t rc = 0;
    switch (datyp) {
    case 0: memcpy(field, data, (size_t)((n * nbits + 31) / 32) * 4); break;
    case 1: rc = packfunc(field, data, data + 3, (int)n, nbits, 24, 1, 2, 0, &tempfloat) ? 0 : -1; break;
    case 129:
        if (!is_double && nk == 1 && nbits <= 16 && ni > 1 && nj > 1 && !getenv("EZHIP_FST_FOUR_TRIPS")) {      /* the record up, the field down (the data part is left as it is) */
            /* data[0] = the words of the data part, itself included (:1186): data + 1 holds data[0] - 1 of them.  The value comes from the file: it is
             * held to the documented size of the buffer (the plain word count) before anything is read */
            const int64_t cap129 = plain_words(129, n, nbits, &hs, &ss);
            if ((int64_t)data[0] < 6 || (int64_t)data[0] > cap129) { fprintf(stderr, "<ezhip_fst_unpack_data> datyp 129: the record's length word (%u) is outside 6 .. %lld\n", data[0], (long long)cap129); return -1; }
            rc = ezhip_uncompress_unpack16_host((float *)field, data + 1, (int64_t)data[0] - 1, ni, nj, nbits);
            break;
        }
        if (armn_compress((unsigned char *)(data + 5), ni, nj, nk, nbits, 2) < 0) return -1;
        rc = packfunc(field, data + 1, data + 5, (int)n, nbits + 64 * imax(16, nbits), 0, 1, 2, 0, &tempfloat) ? 0 : -1;
        break;
    case 2:
        if (is_short) rc = compact_short(field, NULL, data, (int)n, nbits, 0, 1, 6) ? 0 : -1;
        else if (is_byte) rc = compact_char(field, NULL, data, (int)n, 8, 0, 1, 10) ? 0 : -1;      /* (sic: 8 bits whatever the record says, :2321) */
        else rc = compact_integer(field, NULL, data, (int)n, nbits, 0, 1, 2) ? 0 : -1;
        break;
    case 130: {
        c_armn_compress_setswap(0);
        const int nb = armn_compress((unsigned char *)(data + 1), ni, nj, nk, nbits, 2);
        c_armn_compress_setswap(1);
        if (nb < 0) return -1;
        const int16_t *p16 = (const int16_t *)(data + 1);
        if (is_short) memcpy(field, p16, (size_t)n * 2);                             /* :2310 (nbytes = 2 n) */
        else if (is_byte) { int8_t *p8 = (int8_t *)field; for (int64_t i = 0; i < n; i++) p8[i] = (int8_t)p16[i]; }      /* memcpy_16_8 */
        else {
            const int16_t mask = (int16_t)(nbits >= 16 ? -1 : (int)~(~0u << nbits));
            int32_t *p32 = (int32_t *)field;
            for (int64_t i = 0; i < n; i++) p32[i] = p16[i] & mask;                  /* memcpy_16_32 */
        }
        break;
    }
    case 4: {                                                                       /* :2350-2372 as compiled: INTEGER*4 out of compact_integer, then narrowed */
        int32_t *wide = (is_short || is_byte) ? (int32_t *)malloc((size_t)n * 4) : (int32_t *)field;
        if (!wide) return -1;
        rc = compact_integer(wide, NULL, data, (int)n, nbits, 0, 1, 4) ? 0 : -1;
        if (is_short) { int16_t *s = (int16_t *)field; for (int64_t i = 0; i < n; i++) s[i] = (int16_t)wide[i]; }
        if (is_byte) { int8_t *b = (int8_t *)field; for (int64_t i = 0; i < n; i++) b[i] = (int8_t)wide[i]; }
        if (is_short || is_byte) free(wide);
        break;
    }
    case 5:
        if (nbits == 32) memcpy(field, data, (size_t)n * 4);
        else if (nbits == 64 && elem_bytes == 8) { uint32_t *d = (uint32_t *)field; for (int64_t i = 0; i < n; i++) { d[2 * i] = data[2 * i + 1]; d[2 * i + 1] = data[2 * i]; } }
        else if (nbits < 32 && elem_bytes == 4) {
            uint32_t *d = (uint32_t *)field;
            if (!compact_integer(d, NULL, data, (int)n, nbits, 0, 1, 2)) return -1;
            for (int64_t i = 0; i < n; i++) d[i] <<= (32 - nbits);                    /* LSHIFT(IFLD, 32 - NBITS) */
        } else return -1;
        break;
    case 133:                                                                       /* :2436; data[0] = the data part's words (:1318): the stream's length, so its tile chains are followed on the device */
        if ((int64_t)data[0] < 5 || (int64_t)data[0] > plain_words(133, n, nbits, &hs, &ss)) {      /* from the file: held to the buffer's documented size */
            fprintf(stderr, "<ezhip_fst_unpack_data> datyp 133: the record's length word (%u) does not fit the data part\n", data[0]); return -1; }
        rc = c_armn_uncompress32_lng((float *)field, (const unsigned char *)(data + 1), 4 * ((size_t)data[0] - 1), ni, nj, nk, nbits) < 0 ? -1 : 0;
        break;
    case 6: { int32_t nb; rc = c_float_unpacker((float *)field, (int32_t *)data, (int32_t *)(data + hs), (int32_t)n, &nb); break; }
    case 134: {
        int32_t nb;
        if (armn_compress((unsigned char *)(data + 1 + hs), ni, nj, nk, nbits, 2) < 0) return -1;
        rc = c_float_unpacker((float *)field, (int32_t *)(data + 1), (int32_t *)(data + 1 + hs), (int32_t)n, &nb);
        break;
    }
    default:
        fprintf(stderr, "<ezhip_fst_unpack_data> datyp %d is outside the MI355X hot-path scope\n", datyp);
        return -1;
    }
    if (rc == 0 && has_missing) ezhip_fst_decode_missing_value(field, (int32_t)n, datyp & 0x3F, is_byte, is_short, is_double || (datyp == 5 && nbits == 64));
    return rc;
}

int32_t ezhip_fst_unpack_data(void *field, uint32_t *data, int32_t ni, int32_t nj, int32_t nk, int32_t datyp, int32_t nbits)
{
    return ezhip_fst_unpack_data_ex(field, data, ni, nj, nk, datyp, nbits, 4);
}
