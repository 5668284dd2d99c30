/* packhip_shim.h -- internal boundary between pack_host.c (C) and pack_kernels.hip.  Plain C types. */
#ifndef PACKHIP_SHIM_H
#define PACKHIP_SHIM_H
#include <stddef.h>
#include "ezhip_develop.h"
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct {
    double minF, mulFactor;
    unsigned missingToken, hdr3;
    int too_large, pad;
} packhip_cf_params;

/* device scratch layout of the reductions: result words, then one {min,max,or} triple per block */
#define PACKHIP_STATS_PARTIALS 256
#define PACKHIP_STATS_MAXBLK 16384          /* one block per 4096 elements up to 67 M elements, then blocks loop */
#define PACKHIP_STATS_WORDS (PACKHIP_STATS_PARTIALS + 3 * PACKHIP_STATS_MAXBLK)
/* stats[0..2] = min key, max key, OR; kind 0 float keys, 1 uint32, 2 int32 (biased by 0x80000000) */
int packhip_stats(unsigned *d_stats, const void *d_a, size_t n, int stride, int kind, int has_missing, float tag);
/* per-block partials only (returns their count, -1 on error); packhip_cf_header(npartials > 0) reduces them itself */
int packhip_stats_partials(unsigned *d_stats, const void *d_a, size_t n, int stride, int kind, int has_missing, float tag);
int packhip_cf_header(packhip_cf_params *d_pp, unsigned *d_hdr, unsigned *d_stats, int npartials, unsigned n, int bs, int style, int has_missing);
int packhip_cf_header_batch(packhip_cf_params *d_pp, unsigned *d_hdr, size_t hdr_stride_words, unsigned *d_stats, size_t stats_stride_words,
                            int npartials, unsigned n, int bs, int style, int has_missing, int nfields);
int packhip_cf_pack(unsigned *d_out, const float *d_a, size_t n, int E, int offset, int stride,
                    const packhip_cf_params *d_pp, int has_missing, float tag);
int packhip_cf_unpack(float *d_a, const unsigned *d_in, size_t n, int tokenSize, int offset, int stride,
                      double minF, double mulFactor, unsigned missingToken, int has_missing, float tag);
/* compact_double (REAL*8 arrays); d_partials: 2048 doubles of scratch */
int packhip_cf_pack_f64(unsigned *d_out, unsigned *d_hdr, packhip_cf_params *d_pp, double *d_partials, const double *d_a, size_t n,
                        int bs, int E, int offset, int stride, int style, int has_missing, float tag);
int packhip_cf_unpack_f64(double *d_a, const unsigned *d_in, size_t n, int tokenSize, int offset, int stride,
                          double minF, double mulFactor, unsigned missingToken, int has_missing, float tag);
/* compact_short / compact_char: elements i * stride of a 2- / 1-byte array <-> a dense uint32 array */
int packhip_widen(unsigned *d_dst, const void *d_src, size_t n, int stride, int elem_bytes);
int packhip_narrow(void *d_dst, const unsigned *d_src, size_t n, int stride, int elem_bytes);
int packhip_ci_pack(unsigned *d_out, const void *d_a, size_t n, int E, int offset, int stride,
                    int with_header, int is_signed, unsigned minv, int shift, unsigned positiveMask);
int packhip_ci_unpack(void *d_a, const unsigned *d_in, size_t n, int E, int offset, int stride, int with_header, unsigned minv, int shift, unsigned negMask);
int packhip_fp_pack(int *d_stream, const float *d_src, int npts, int MaxExp, int Minimum, int Round, int Shift2, int Mask);
int packhip_fp_unpack(float *d_dest, const int *d_stream, int npts, int maxExp, int minimum, int shift2);
int packhip_armn_encode(unsigned *d_z, size_t z_words, const unsigned *d_words, int ni, int nj, int nbits, int minimum_method,
                        void *d_work, unsigned long long *h_bits);
size_t packhip_armn_work_bytes(int ni, int nj);
/* after packhip_armn_encode(..., h_bits = NULL): zlng on the device, stream committed to d_dst when compressible */
int packhip_armn_commit(unsigned *d_dst, const unsigned *d_z, void *d_work, int ni, int nj, int nbits, int minimum_method, int *d_zlng);
/* one-pass PARALLELOGRAM encoder (k_armn_enc1): scratch bytes for nfields fields (0: shape outside its limits) */
size_t packhip_armn_enc1_work_bytes(int ni, int nj, int nbits, int nfields);
/* asynchronous; d_zlng[f] = byte count, -1 (not compressible) or -2 (re-run that field with container 5).  Returns 0, 1 when the
 * shape is outside the one-pass limits (nothing launched: use packhip_armn_encode), -1 on error */
int packhip_armn_encode1(unsigned *d_z, size_t z_stride_words, size_t z_cap_words, const unsigned *d_tok, size_t tok_stride_words,
                         int ni, int nj, int nbits, int nfields, int swapped, int container, void *d_work, int *d_zlng);
int packhip_cond_copy(unsigned *d_dst, size_t dst_stride, const unsigned *d_src, size_t src_stride, size_t nwords, const int *d_zlng, int nfields, int mode);
int packhip_swap_halves(unsigned *d_dst, const unsigned *d_src, size_t nwords);
/* FST framing of a device-resident cfg5 record: the datyp-129 data part (length word + record + zero padding); the datyp-1 fallback's header and
 * one-token-per-word array (packhip_ci_pack then packs them behind the header) */
int packhip_fst_frame129(unsigned *d_data, const unsigned *d_rec, unsigned nw, unsigned ncopy);
int packhip_fst_plain_from_slots(unsigned *d_data, unsigned *d_tok32, const unsigned *d_rec, size_t n);
/* prefix words of the streams the fused interpolate + encode launch (k_sepx_enc, ez_kernels.hip) produced */
int packhip_sepenc_prefix(unsigned *d_z, size_t z_stride_words, size_t z_cap_words, const unsigned short *d_ptok, size_t ptok_stride,
                          const unsigned *d_head, const int *d_zlng, int ni, int nj, int nbits, int container, int nfields);
/* armn_compress UNCOMPRESS (unpack_kernels.hip): nfields streams of z_words capacity each -> (1 + ni*nj/2) token words each */
size_t packhip_armn_dec_work_bytes(int ni, int nj, size_t z_words);
int packhip_armn_decode_phase(unsigned *d_out, size_t out_stride_words, const unsigned *d_z, size_t z_stride_words, size_t z_words,
                              int ni, int nj, int nfields, int swap, void *d_work, size_t work_stride_bytes, int *d_status, int phase);      /* 0 all, 1 up to the composed table, 2 the chain kernel and what follows */
int packhip_armn_decode(unsigned *d_out, size_t out_stride_words, const unsigned *d_z, size_t z_stride_words, size_t z_words,
                        int ni, int nj, int nfields, int swap, void *d_work, size_t work_stride_bytes, int *d_status);
/* ---- IEEE-32 compressor (armn32_kernels.hip): planes of a float field and the parallelogram coder on a plane ---- */
int packhip_a32_split(unsigned char *d_expo, unsigned *d_mant, unsigned *d_smask, unsigned *d_stats, const float *d_f, size_t n, int nbits, unsigned *h_stats4);
size_t packhip_pg_work_bytes(int ni, int nj);
int packhip_pg_encode(unsigned *d_z, size_t z_words, const void *d_u, int elem_bytes, int ni, int nj, int nbits, long long remaining_space, void *d_work);
int packhip_pg_result(void *d_work, int ni, int nj, int nbits, unsigned long long *bits, int *failed);
int packhip_pg_decode(int *d_D, int *d_bs, const unsigned *d_z, const unsigned long long *d_tpos, int ni, int nj, int nbits, int wide);
int packhip_pg_decode2(int *d_D, int *d_bs, const unsigned *d_z, const unsigned long long *d_tpos, const unsigned *d_tpos32, int ni, int nj, int nbits, int wide);
/* the tile-header chain of one plane of c_armn_compress32 walked on the device (unpack_kernels.hip) */
unsigned packhip_armn_plane_header(int nbits);
int packhip_armn_tile_walk(const unsigned *d_z, size_t z_words, int ni, int nj, void *d_work, int *d_status);
int packhip_armn_tile_walk_parallel(const unsigned *d_z, size_t z_words, int ni, int nj, void *d_work, int *d_status);      /* no serial chain kernel behind the parallel forms: *d_status = 1 when they did not resolve the chain */
/* sign run lengths -> sign mask on the device (unpack1bitRLE); d_z: the sub-stream on the device */
size_t packhip_a32_rle_work_bytes(size_t nbytes);
int packhip_a32_rle_decode(unsigned *d_mask, const unsigned *d_z, size_t nbytes, size_t npts, void *d_work, int *h_bad);
/* pack1bitRLE on the device: d_z receives the token stream (at most npts / 28 + 2 words), *h_bits its bit count; grow(slot, bytes): the caller's grow-only device
 * workspaces (four slots: the run arrays are sized by the number of runs) */
size_t packhip_a32_rle_enc_work_bytes(size_t npts);
int packhip_a32_rle_encode(unsigned *d_z, const unsigned *d_mask, size_t npts, void *d_work, void *(*grow)(int slot, size_t bytes), unsigned long long *h_bits);
int packhip_a32_combine(float *d_f, const int *d_expo, const int *d_mant, const unsigned *d_smask, size_t n, int nbits, unsigned exp_min, int code_signe, int have_expo);
#ifdef __cplusplus
}
#endif
#endif
