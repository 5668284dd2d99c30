/* oracle/orc_pack.h -- TEST INFRASTRUCTURE (CPU oracle).  See orc_pack.c ("parity unpinned"). */
#ifndef ORC_PACK_H
#define ORC_PACK_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* compact_float, src/packers/compact.tmplc:37-431 (opCode 1 = FLOAT_PACK, 2 = FLOAT_UNPACK) */
void *orc_compact_float(void *unpacked, void *packedHeader, void *packed, int elementCount,
                        int packedTokenBitSize, int offset, int stride, int opCode, int hasMissing, const void *missingTag);
/* compact_double, src/packers/compact.c:28-32 (the same template on double arrays) */
void *orc_compact_double(void *unpacked, void *packedHeader, void *packed, int elementCount,
                         int packedTokenBitSize, int offset, int stride, int opCode, int hasMissing, const void *missingTag);
/* compact_short / compact_char, src/packers/compact_integer.c:592, :830 (opCode 5/6, 9/10) */
int orc_compact_short(void *unpacked, void *packedHeader, void *packed, int elementCount, int bitSize, int off_set, int stride, int opCode);
int orc_compact_char(void *unpacked, void *packedHeader, void *packed, int elementCount, int bitSize, int off_set, int stride, int opCode);
/* compact_integer, src/packers/compact_integer.c:325-570 */
int orc_compact_integer(void *unpacked, void *packedHeader, void *packed, int elementCount,
                        int bitSizeOfPackedToken, int off_set, int stride, int opCode);
/* c_float_packer / c_float_unpacker, src/packers/float_packer.c */
int32_t orc_float_packer(float *source, int32_t nbits, int32_t *header, int32_t *stream, int32_t npts);
int32_t orc_float_unpacker(float *dest, int32_t *header, int32_t *stream, int32_t npts, int32_t *nbits);
/* armn_compress, src/compresseur/c_zfstlib.c:67-203 */
int orc_armn_compress(unsigned char *fld, int ni, int nj, int nk, int nbits, int op_code);
void orc_armn_compress_setlevel(int level);
int orc_armn_encode(uint32_t *z, const uint16_t *tokens, int ni, int nj, int nbits);
int orc_armn_decode(uint16_t *tokens, const uint32_t *z, int ni, int nj);
#ifdef __cplusplus
}
#endif
#endif
