/*
 * interpv_hip.h -- C ABI of the MI355X (gfx950) vertical interpolation, a drop-in for librmn's 1-D interpolation
 * package src/interpv.  Shipped in librmn_ez_hip.so.
 *
 * 1. The reference's own Fortran-callable symbols (host arrays, every argument by reference, LOGICAL = 4 bytes,
 *    arrays (ijDim, levels) with the column index fastest; Interp1D_Arguments.inc:21-58,
 *    Interp1D_Arguments_X.inc:3-7).  Each routine exists, as in the reference, in four forms
 *    (e.g. Interp1D_Linear.F90:22-99):  name_  name8_  name_x_  name_x8_   (REAL / REAL*8, plain / extended interface;
 *    the extended arguments flux, numExtArraysIn/Out, ExtArraysIn/Out are accepted and unused, as in the reference):
 *
 *        interp1d_findpos_  interp1d_findpos8_                                   Interp1D_FindPos.F90:18-44
 *        interp1d_nearestneighbour{_,8_,_x_,_x8_}                                Interp1D_NearestNeighbour.F90
 *        interp1d_linear{...}                                                    Interp1D_Linear.F90
 *        interp1d_cubiclagrange{...}                                             Interp1D_CubicLagrange.F90
 *        interp1d_cubicwithderivs{...}                                           Interp1D_CubicWithDerivs.F90
 *        extrap1d_fixed{...}  extrap1d_lapserate{...}  extrap1d_abort{...}       Extrap1D_*.F90
 *
 *    Extrap1D_Abort ends the program with exit(2) after the reference's message when a destination level needs
 *    extrapolation (Extrap1D_Abort_Body.inc:76-79, :85-88).  Without a usable HIP device every routine prints an
 *    error and aborts the program: there is no CPU fallback.
 *    Not provided: Extrap1D_Surface / Extrap1D_SurfaceWind (they call a host routine of the physics library,
 *    `external flux`, once per level: host control flow, outside the accelerated path).
 *
 * 2. Device-pointer entry points (arrays already in HBM, asynchronous on the stream given to ezhip_use_stream()):
 *    the same three steps, and the three of them fused in one pass that never writes the bracket array.
 */
#ifndef INTERPV_HIP_H
#define INTERPV_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { INTERPV_NONE = 0, INTERPV_NEAREST = 1, INTERPV_LINEAR = 2, INTERPV_CUBIC_LAGRANGE = 3, INTERPV_CUBIC_DERIVS = 4 };
enum { EXTRAPV_NONE = 0, EXTRAPV_FIXED = 1, EXTRAPV_LAPSERATE = 2, EXTRAPV_ABORT = 3 };

/* prec8: 0 REAL arrays, 1 REAL*8 arrays.  Return 0, -1 on error (message on stderr). */
int32_t interpv_hip_findpos_dev(int prec8, int numInterpSets, int srcNumLevels, int destNumLevels, int src_ijDim, int dst_ijDim,
                                const void *d_vLevelSource, int32_t *d_posnDestInSrc, const void *d_vLevelDestn);
int32_t interpv_hip_interp_dev(int prec8, int algo, int numInterpSets, int srcNumLevels, int destNumLevels, int src_ijDim, int dst_ijDim,
                               const void *d_vLevelSource, const void *d_stateSource, const void *d_stateDerivSource,
                               const int32_t *d_posnDestInSrc, const void *d_vLevelDestn, void *d_stateDestn, void *d_stateDerivDestn,
                               int extrapEnableDown, int extrapEnableUp);
/* EXTRAPV_ABORT synchronises the stream and returns 2 (after the reference's message) when a level needs extrapolation */
int32_t interpv_hip_extrap_dev(int prec8, int extrap, int numInterpSets, int srcNumLevels, int destNumLevels, int src_ijDim, int dst_ijDim,
                               const void *d_vLevelSource, const void *d_stateSource, const void *d_stateDerivSource,
                               const int32_t *d_posnDestInSrc, const void *d_vLevelDestn, void *d_stateDestn, void *d_stateDerivDestn,
                               int extrapEnableDown, int extrapEnableUp, double extrapGuideDown, double extrapGuideUp);
/* FindPos -> algo -> extrap in one pass over the columns (d_posnDestInSrc may be NULL: the brackets are not stored) */
int32_t interpv_hip_column_dev(int prec8, int algo, int extrap, int numInterpSets, int srcNumLevels, int destNumLevels, int src_ijDim, int dst_ijDim,
                               const void *d_vLevelSource, const void *d_stateSource, const void *d_stateDerivSource,
                               int32_t *d_posnDestInSrc, const void *d_vLevelDestn, void *d_stateDestn, void *d_stateDerivDestn,
                               int extrapEnableDown, int extrapEnableUp, double extrapGuideDown, double extrapGuideUp);

/* ---- the reference's Fortran-callable symbols ---- */
#define INTERPV_F_ARGS(R) const int32_t *numInterpSets, const int32_t *srcNumLevels, const int32_t *destNumLevels,                   \
    const int32_t *src_ijDim, const int32_t *dst_ijDim, const R *vLevelSource, const R *stateSource, const R *stateDerivSource,      \
    const int32_t *posnDestInSrc, const R *vLevelDestn, R *stateDestn, R *stateDerivDestn,                                            \
    const int32_t *extrapEnableDown, const int32_t *extrapEnableUp, const R *extrapGuideDown, const R *extrapGuideUp
#define INTERPV_F_XARGS(R) , void *flux, const int32_t *numExtArraysIn, const int32_t *numExtArraysOut, const R *ExtArraysIn, R *ExtArraysOut
#define INTERPV_F_DECL(name)                                  \
    void name##_(INTERPV_F_ARGS(float));                      \
    void name##8_(INTERPV_F_ARGS(double));                    \
    void name##_x_(INTERPV_F_ARGS(float) INTERPV_F_XARGS(float));    \
    void name##_x8_(INTERPV_F_ARGS(double) INTERPV_F_XARGS(double));
void interp1d_findpos_(const int32_t *numInterpSets, const int32_t *srcNumLevels, const int32_t *destNumLevels, const int32_t *src_ijDim,
                       const int32_t *dst_ijDim, const float *vLevelSource, int32_t *posnDestInSrc, const float *vLevelDestn);
void interp1d_findpos8_(const int32_t *numInterpSets, const int32_t *srcNumLevels, const int32_t *destNumLevels, const int32_t *src_ijDim,
                        const int32_t *dst_ijDim, const double *vLevelSource, int32_t *posnDestInSrc, const double *vLevelDestn);
INTERPV_F_DECL(interp1d_nearestneighbour)
INTERPV_F_DECL(interp1d_linear)
INTERPV_F_DECL(interp1d_cubiclagrange)
INTERPV_F_DECL(interp1d_cubicwithderivs)
INTERPV_F_DECL(extrap1d_fixed)
INTERPV_F_DECL(extrap1d_lapserate)
INTERPV_F_DECL(extrap1d_abort)

#ifdef __cplusplus
}
#endif
#endif
