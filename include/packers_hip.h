/*
 * packers_hip.h -- C ABI of the MI355X-native field packers (librmn_ez_hip.so).
 *
 * Drop-in for librmn's compact_float / compact_integer / c_float_packer / armn_compress entry points
 * (same names, argument meaning, return values; reference lines cited per function).  Host-pointer
 * calls stage through device memory; the additive *_dev entry points take DEVICE pointers and enqueue
 * on the stream selected with ezhip_use_stream (ezscint_hip.h).
 *
 * Rejected loudly (NULL / -1): tokens wider than 32 bits, streams of the deactivated SAMPLE predictor.
 */
#ifndef PACKERS_HIP_H
#define PACKERS_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* include/armn_compress.h:30-41 ; src/packers/compact.tmplc:37-431.  opCode 1 = FLOAT_PACK, 2 = FLOAT_UNPACK.
 * packedTokenBitSize > 64 means (slot width << 6) | significant bits, as c_fstecr passes it (fstd98.c:1170).
 * missingTag must be non-NULL (the reference dereferences it unconditionally). */
void *compact_float(void *unpackedArrayOfFloat, void *packedHeader, void *packedArrayOfInt, const int elementCount,
                    const int packedTokenBitSize, const int offset, const int stride, const int opCode,
                    const int hasMissing, const void *const missingTag);

/* src/packers/packers.h:6 ; src/packers/compact_integer.c:325-570.  opCode 1/3 = pack unsigned/signed,
 * 2/4 = unpack.  Returns bitSizeOfPackedToken (or 0). */
int compact_integer(void *unpackedArrayOfInt, void *packedHeader, void *packedArrayOfInt, int elementCount,
                    int bitSizeOfPackedToken, int off_set, int stride, int opCode);

/* src/packers/compact.c:28-32 (compact.tmplc with FLOAT_TYPE double): unpackedArrayOfFloat and missingTag point to doubles */
void *compact_double(void *unpackedArrayOfFloat, void *packedHeader, void *packedArrayOfInt, const int elementCount,
                     const int packedTokenBitSize, const int offset, const int stride, const int opCode,
                     const int hasMissing, const void *const missingTag);
/* src/packers/compact_integer.c:592 (opCode 5 = pack, 6 = unpack unsigned short) and :830 (opCode 9 = pack, 10 = unpack unsigned char).
 * Any other opCode -- including the signed 7 / 8 and 11 / 12 named by a branch of fstd98.c that is not compiled (fstd98.c:22) -- prints
 * "opCode (%d) is not defined" and returns 0, as in the reference (:797-816, :1034-1053). */
int compact_short(void *unpackedArrayOfShort, void *packedHeader, void *packedArrayOfInt, int elementCount,
                  int bitSizeOfPackedToken, int off_set, int stride, int opCode);
int compact_char(void *unpackedArrayOfBytes, void *packedHeader, void *packedArrayOfInt, int elementCount,
                 int bitSizeOfPackedToken, int off_set, int stride, int opCode);

/* src/packers/packers.h:11 ; src/packers/float_packer.c:258-283, 212-243, 289-319 */
int32_t c_float_packer(float *source, int32_t nbits, int32_t *header, int32_t *stream, int32_t npts);
int32_t c_float_unpacker(float *dest, int32_t *header, int32_t *stream, int32_t npts, int32_t *nbits);
void    c_float_packer_params(int32_t *header_size, int32_t *stream_size, int32_t *p1, int32_t *p2, int32_t npts);

/* include/armn_compress.h:17 ; src/compresseur/c_zfstlib.c:67-203.  op_code 1 = COMPRESS (in place on the
 * 16-bit-slot token words produced by compact_float), returns the compressed byte count or -1.
 * op_code 2 = UNCOMPRESS (:181-200, in place: (1 + ni*nj/2) words of tokens come back), returns ni*nj*2.
 * Streams of the deactivated SAMPLE predictor (c_fstunzip_sample) are refused (-1). */
int  armn_compress(unsigned char *fld, int ni, int nj, int nk, int nbits, int op_code);
/* src/compresseur/armn_compress_32.h ; armn_compress_32.c:59-275, :285-437 (datyp 133: sign RLE + 8-bit exponent parallelogram + mantissa
 * parallelogram of znbits - 9 bits).  zstream: host memory of ni*nj*znbits/8 bytes (+ 64).  c_armn_compress32 returns the byte count or -1
 * (not compressible / ni, nj < 16); c_armn_uncompress32 returns ni*nj.  Fortran twins armn_compress32_ / armn_uncompress32_. */
int c_armn_compress32(unsigned char *zstream, float *fld, int ni, int nj, int nk, int znbits);
/* NOTE on c_armn_uncompress32 (the reference's own signature: NO stream length).  To upload the record in one piece the library has to bound it: it reads
 * /proc/self/maps (Linux procfs; one parse per call) for the readable extent behind `zstream`, estimates the mantissa plane's length from its first tiles
 * (+ 12 %) and uploads that much -- i.e. it READS caller memory beyond the record's end, up to the end of the mapping at most, never beyond it -- and tries
 * once more with everything readable if a chain leaves the estimate; without procfs the chains are walked on host threads first.  Results are the reference's
 * either way.  A caller that knows the record's byte length (every FST reader does: fstd98.c:2436) should call c_armn_uncompress32_lng below: no procfs, no
 * over-read, no second attempt -- it is the documented default for new call sites, this entry exists for source compatibility. */
int c_armn_uncompress32(float *fld, unsigned char *zstream, int ni, int nj, int nk, int znbits);
int armn_compress32_(unsigned char *zstream, float *fld, int *ni, int *nj, int *nk, int *nbits);
int armn_uncompress32_(float *fld, unsigned char *zstream, int *ni, int *nj, int *nk, int *nbits);
/* additive: the field on the device (the stream of c_armn_uncompress32_dev stays in HOST memory: its chain of tile headers is walked there) */
int c_armn_compress32_dev(void *d_zstream, const float *d_fld, int ni, int nj, int nk, int znbits);
int c_armn_uncompress32_dev(float *d_fld, const unsigned char *zstream_host, int ni, int nj, int nk, int znbits);
/* additive: the same with the stream's byte length given (an upper bound that the caller's buffer holds, e.g. the data part of an FST record, fstd98.c:2436):
 * c_armn_uncompress32 has no length argument and has to find the record's end by walking its tile headers on the host; with the length the planes are
 * uploaded at once and the sign runs and both planes' chains of tile headers are resolved on the device when the field has >= 65536 points and either every
 * row ends on a whole tile ((ni - 1) % 3 == 0: composition of the windows' maps) or ni >= 768 (ragged rows: composition + the row recurrence).  A plane the
 * device forms leave open (long runs of empty tiles, rows that rejoin late), smaller fields and records beyond 2^32 bits are walked on host threads, plane by
 * plane, as in c_armn_uncompress32 */
int c_armn_uncompress32_lng(float *fld, const unsigned char *zstream, size_t zbytes, int ni, int nj, int nk, int znbits);
int c_armn_uncompress32_lng_dev(float *d_fld, const unsigned char *zstream_host, size_t zbytes, int ni, int nj, int nk, int znbits);
/* additive: record AND field in device memory (the counterpart of c_armn_compress32_dev, whose stream stays in HBM).  Under the routing above the host reads
 * four words of the record and everything else is device work; a plane the device forms leave open is copied down for the host's walk.  Returns ni*nj or -1. */
int c_armn_uncompress32_zdev(float *d_fld, const void *d_zstream, size_t zbytes, int ni, int nj, int nk, int znbits);
void c_armn_compress_setlevel(int level);       /* src/compresseur/c_zfstlib.c:1325 ; BEST = 1, FAST = 0 */
int  c_armn_compress_getlevel(void);
void c_armn_compress_setswap(int swapState);

/* ---- additive: device-resident entry points -------------------------------------------------- */
void *compact_float_dev(void *d_unpacked, void *d_header, void *d_packed, int elementCount, int packedTokenBitSize,
                        int offset, int stride, int opCode, int hasMissing, const void *missingTag_host, int header_style);
int   compact_integer_dev(void *d_unpacked, void *d_header, void *d_packed, int elementCount,
                          int bitSizeOfPackedToken, int off_set, int stride, int opCode);
/* out-of-place: d_words = ni*nj 16-bit-slot tokens (two per word), d_z >= (ni*nj/2 + 16) words: no store ever goes past that
 * capacity, also for fields that turn out not to be compressible.  Returns zlng or -1. */
int   armn_compress_dev(void *d_z, const void *d_words, int ni, int nj, int nbits);
/* UNCOMPRESS out of place on device data: d_z holds a stream with room for z_words words (reads never go past it),
 * d_words receives (1 + ni*nj/2) words of 16-bit-slot tokens.  Returns ni*nj*2, or -1 (unknown header / broken stream).
 * The batch form decodes nfields streams (z_stride_words apart) concurrently, one workgroup per stream for the
 * serial part (the chain of tile headers), all CUs for the rest; one synchronisation at the end. */
int   armn_uncompress_dev(void *d_words, const void *d_z, int64_t z_words, int ni, int nj, int nbits);
int32_t ezhip_armn_uncompress_batch_dev(void *d_words, int64_t out_stride_words, const void *d_z, int64_t z_stride_words, int64_t z_words,
                                        int ni, int nj, int nbits, int32_t nfields);
/* read side of ezhip_pack16_compress_dev (fstd98.c:2285-2290): record [4 header words][armn stream if compressed != 0,
 * else plain 16-bit slots] -> ni*nj floats at d_field.  Returns 0 or -1. */
int   ezhip_uncompress_unpack16_dev(float *d_field, const void *d_record, int ni, int nj, int nbits, int compressed);
/* fused cfg5 pipeline step: quantise + pack to 16-bit slots + compress one field that is already on the device
 * (fstd98.c:1170-1172).  d_record receives [4 header words][stream]; returns zlng (>0), or -1 when the
 * field is not compressible (d_record then holds the plain 16-bit-slot pack). */
int   ezhip_pack16_compress_dev(void *d_record, const float *d_field, int ni, int nj, int nbits);
/* fused front half of the cfg5 pipeline: c_ezsint_batch_dev on the current grid set + compact_float(nbits + 64*max(16,nbits),
 * header style 2: fstd98.c:1170) of every output field, on the calling thread's stream, no synchronisation.  The min/max
 * pass of compact_float is fused into the interpolation kernel.  d_records: nfields records of record_stride_words
 * words, [4 header words][stream]; d_zout keeps the interpolated fields.  Returns the c_ezsint code (0 / 2) or -1. */
int32_t ezhip_ezsint_pack16_batch_dev(void *d_records, int64_t record_stride_words, float *d_zout, const float *d_zin,
                                      int32_t nfields, int32_t npts_out, int32_t nbits);
/* cfg5 back half for a batch: compact_float(16-bit slots, unless prepacked) + armn_compress of nfields device-resident
 * fields, asynchronous per field, ONE synchronisation at the end.  zlng_out[f] (host) = compressed byte count of record f
 * or -1 (not compressible: the record keeps the plain 16-bit pack). */
int32_t ezhip_pack16_compress_batch_dev(void *d_records, int64_t record_stride_words, const float *d_fields, int64_t field_stride,
                                        int32_t nfields, int32_t ni, int32_t nj, int32_t nbits, int32_t prepacked, int32_t *zlng_out);
/* The whole cfg5 pipeline on a batch, fused: c_ezsint on the current grid set -> compact_float(nbits + 64*16, style 2) ->
 * armn_compress of nfields device-resident SOURCE fields (fstd98.c:1170-1172) without ever storing the interpolated floats:
 * the interpolation runs twice (min/max only, then straight to 16-bit tokens), the one-pass encoder writes every stream in
 * place.  d_records: nfields records of record_stride_words (>= 4 + ni_out*nj_out/2 + 16) words, [4 header words][stream];
 * zlng_out[f] (host) = byte count of the stream or -1 (not compressible: plain 16-bit pack).  One synchronisation at the end.
 * Returns 0, -1 on error, -2 when the grid set / shape is outside the fused path (use the two calls above instead). */
int32_t ezhip_ezsint_pack16_compress_batch_dev(void *d_records, int64_t record_stride_words, const float *d_zin, int32_t nfields,
                                               int32_t ni_out, int32_t nj_out, int32_t nbits, int32_t *zlng_out);
/* ---- FST record framing around the packers ---------------------------------------------------------------------
 * The DATA PART of an FST record exactly as c_fstecr lays it out from buffer->data[keys_len] on (src/fstd98/fstd98.c:808-826 flags, :935-1000
 * sizes, :1133-1370 the datyp switch): for datyp > 128 the length word, the packer header(s), the armn_compress stream; compression that does
 * not pay falls back to the plain type (*datyp_out); turbo types beyond 16 bits lose the flag (:934).  Returns the number of 32-bit words of the
 * data part (what c_xdfput would be given) or -1.  The read side (:2169-2183, :2270-2466) expands the stream in place in `data` like the reference
 * does in its record buffer.
 * _ex: elem_bytes = size of the caller's array elements (what xdf_byte / xdf_short / xdf_double tell the reference): 4 REAL / INTEGER, 8 REAL*8
 * (datyp 1, 129 through compact_double; datyp 5 with 64 bits), 2 and 1 for integer types (datyp 2, 130, 4 through compact_short / compact_char,
 * the signed type through the sign-extension of the compiled branch, fstd98.c:22).  datyp may carry the missing-value flag (+64): the magic
 * values of the array are replaced by a packable stand-in above the maximum (EncodeMissingValue, fst_missing.c:1472) and restored after
 * unpacking (DecodeMissingValue :1579) when the mode is on (MISSING_VALUE_FLAGS in the environment, or ezhip_fst_force_missing_value_usage).
 * The plain entry points are the 4-byte forms. */
int32_t ezhip_fst_pack_data(uint32_t *data, int64_t cap_words, void *field, int32_t ni, int32_t nj, int32_t nk, int32_t datyp, int32_t nbits,
                            int32_t *datyp_out, int32_t *nbits_out);
int32_t ezhip_fst_unpack_data(void *field, uint32_t *data, int32_t ni, int32_t nj, int32_t nk, int32_t datyp, int32_t nbits);
int32_t ezhip_fst_pack_data_ex(uint32_t *data, int64_t cap_words, void *field, int32_t ni, int32_t nj, int32_t nk, int32_t datyp, int32_t nbits,
                               int32_t elem_bytes, int32_t *datyp_out, int32_t *nbits_out);
int32_t ezhip_fst_unpack_data_ex(void *field, uint32_t *data, int32_t ni, int32_t nj, int32_t nk, int32_t datyp, int32_t nbits, int32_t elem_bytes);
/* A record the cfg5 pipeline left in HBM -> the data part of the FST record c_fstecr(datyp 129) writes for it, without leaving the device: the length word in
 * front of [4 header words][stream] (in place when d_data + 1 == d_record), or -- zlng < 0: compression did not pay -- the datyp 1 form re-packed from
 * the record's 16-bit-slot tokens (fstd98.c:1164-1190).  Asynchronous; returns the words of the data part, *datyp_out = 129 or 1; -1 on error */
int32_t ezhip_fst_frame_record_dev(uint32_t *d_data, int64_t cap_words, const uint32_t *d_record, int32_t zlng, int32_t ni, int32_t nj, int32_t nk,
                                   int32_t nbits, int32_t *datyp_out);
/* src/fstd98/fst_missing.c: missing_value_used :245, ForceMissingValueUsage :309, set_missing_value_flags :100 (NULL = keep), EncodeMissingValue :1472,
 * DecodeMissingValue :1579 */
int32_t ezhip_fst_missing_value_used(void);
int32_t ezhip_fst_force_missing_value_usage(int32_t enable);
void    ezhip_fst_set_missing_value_flags(const float *f, const int32_t *i, const uint32_t *ui, const double *d, const int16_t *s, const uint16_t *us,
                                          const int8_t *b, const uint8_t *ub);
int32_t ezhip_fst_encode_missing_value(void *dst, const void *src, int32_t n, int32_t datatype, int32_t nbits, int32_t is_byte, int32_t is_short, int32_t is_double);
void    ezhip_fst_decode_missing_value(void *field, int32_t n, int32_t datatype, int32_t is_byte, int32_t is_short, int32_t is_double);
/* frees the calling thread's grow-only device workspaces (staged operands of the host-pointer entry points,
 * the compressed stream, the armn_compress scan storage); they are re-created on the next call */
void  ezhip_pack_release(void);

#ifdef __cplusplus
}
#endif
#endif
