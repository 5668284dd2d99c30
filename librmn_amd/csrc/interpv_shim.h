/* interpv_shim.h -- internal boundary between interpv_host.c (C) and interpv_kernels.hip.  Plain C types. */
#ifndef INTERPV_SHIM_H
#define INTERPV_SHIM_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
enum { IV_NONE = 0, IV_NEAREST = 1, IV_LINEAR = 2, IV_CUBIC_LAGRANGE = 3, IV_CUBIC_DERIVS = 4 };      /* algo   */
enum { IVX_NONE = 0, IVX_FIXED = 1, IVX_LAPSERATE = 2, IVX_ABORT = 3 };                               /* extrap */

/* one launch of k_interpv: [find the brackets] -> [interpolate] -> [extrapolate], any subset, all DEVICE pointers.
 * Arrays are the reference's (ijDim, levels), column index fastest (Interp1D_Arguments.inc:31-58). */
typedef struct {
    int prec8;                    /* 0: REAL arrays, 1: REAL*8 arrays */
    int algo, extrap;
    int find;                     /* 1: brackets computed in the kernel (Interp1D_FindPos); 0: read from posn */
    int write_posn;               /* with find: store the brackets to posn */
    int n, ns, nd, sij, dij;      /* numInterpSets, srcNumLevels, destNumLevels, src_ijDim, dst_ijDim */
    const void *vls, *ss, *sds;
    int *posn;
    const void *vld;
    void *sd, *sdd;
    int xdown, xup;
    double gdown, gup;
    unsigned long long *abort_key;/* IVX_ABORT: atomicMin of (vt * n + i) * 2 + (0 below, 1 above) over the offending points */
} ivhip_args;

int ivhip_launch(const ivhip_args *a);                /* asynchronous on the current stream; 0 / -1 */
int ivhip_copy2d(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width_bytes, size_t rows, int to_device);
#ifdef __cplusplus
}
#endif
#endif
