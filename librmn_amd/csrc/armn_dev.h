/*
 * armn_dev.h -- device helpers of the armn_compress PARALLELOGRAM encoder shared by its two homes: the one-pass encoder of a token
 * array (k_armn_enc1, pack_kernels.hip) and the kernel that interpolates and encodes in one launch (k_sepx_enc, ez_kernels.hip).
 * HIP translation units only.
 */
#ifndef ARMN_DEV_H
#define ARMN_DEV_H

__device__ __forceinline__ int bitlen(unsigned v) { return v ? 32 - __clz((int)v) : 0; }

/* bits of one tile in the stream: PARALLELOGRAM (c_zfstlib.c:722-768) `container` bits of width field + cnt tokens of need + 1 bits (17 when the width
 * field says 15); MINIMUM (c_zfstlib.c:520-575) */
__device__ __forceinline__ unsigned tile_bits(int PARA, unsigned need, int cnt, int container, int nbits)
{
    if (PARA) return (unsigned)container + (need == 0 ? 0u : (unsigned)cnt * (need == 15 ? 17u : need + 1u));
    if (need == 0) return 4u + (unsigned)nbits;
    if (need == 15) return 4u + 16u * (unsigned)cnt;
    return 4u + (unsigned)nbits + need * (unsigned)cnt;
}

/* look-back granules: {state (2 bits), "a |difference| > 65535 so far" (1 bit), value}; agent-scope relaxed accesses (served by the coherence point) */
#define ST_AGG (1ull << 62)
#define ST_PFX (2ull << 62)
#define ST_GT  (1ull << 61)          /* a |difference| > 65535 in this chunk (AGG) / in this or an earlier chunk (PFX) */
#define ST_VAL(x) ((x) & 0x1FFFFFFFFFFFFFFFull)
__device__ __forceinline__ unsigned long long ld_granule(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_granule(unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

#endif
