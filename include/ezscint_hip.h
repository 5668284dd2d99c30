/*
 * ezscint_hip.h -- C ABI of the MI355X-native EZ interpolator (librmn_ez_hip.so).
 *
 * Drop-in for the EZSCINT entry points of librmn on the c_ezdefset / c_ezsint / c_ezuvint /
 * c_gdxysint path: same names, argument meaning, ownership and return codes as the reference
 * header src/PUBLIC_INCLUDES/rmn/ezscint.h (line numbers cited per function).  Every c_foo has
 * its Fortran twin foo_ (scalars by reference, hidden trailing string lengths), as in the
 * reference (rpnmacros.h:21 f77name).
 *
 * Additive entry points (not in the reference, never change the above): *_dev variants taking
 * DEVICE pointers, batched / fused pipeline calls, and stream selection.  See INTEGRATION.md.
 */
#ifndef EZSCINT_HIP_H
#define EZSCINT_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- grid definition / set selection (host only) --------------------------------------- */
int32_t c_ezqkdef(int32_t ni, int32_t nj, char *grtyp, int32_t ig1, int32_t ig2, int32_t ig3, int32_t ig4, int32_t iunit);          /* ezscint.h:66 ; src/interp/ezqkdef.c:25 */
int32_t c_ezgdef_fmem(int32_t ni, int32_t nj, char *grtyp, char *grref, int32_t ig1, int32_t ig2, int32_t ig3, int32_t ig4,
                      float *ax, float *ay);                                                                                         /* ezscint.h:28 ; ezgdef_fmem.c:37 (ax/ay are copied) */
/* Yin-Yang 'U' grid made of two Z-on-E subgrids (ni x nj/2 each), as a SOURCE grid of c_ezsint / c_ezuvint towards one ordinary
 * target (c_ezyysint / c_ezyyuvint); fields are [Yin; Yang] concatenated.  ezscint.h:32 ; ezgdef_supergrid.c:40 */
int32_t c_ezgdef_supergrid(int32_t ni, int32_t nj, char *grtyp, char *grref, int32_t vercode, int32_t nsubgrids, int32_t *subgrid);
int32_t c_ezdefset(int32_t gdout, int32_t gdin);                                                                                     /* ezscint.h:11 ; ezdefset.c:38 ; returns 1 */
int32_t c_gdrls(int32_t gdin);                                                                                                       /* ezscint.h:75 ; gdrls.c:34 */
int32_t c_ezgetgdin(void);                                                                                                           /* ezscint.h:186 */
int32_t c_ezgetgdout(void);                                                                                                          /* ezscint.h:187 */
int32_t c_ezgprm(int32_t gdid, char *grtyp, int32_t *ni, int32_t *nj, int32_t *ig1, int32_t *ig2, int32_t *ig3, int32_t *ig4);       /* ezscint.h:51 */
int32_t c_ezgxprm(int32_t gdid, int32_t *ni, int32_t *nj, char *grtyp, int32_t *ig1, int32_t *ig2, int32_t *ig3, int32_t *ig4,
                  char *grref, int32_t *ig1ref, int32_t *ig2ref, int32_t *ig3ref, int32_t *ig4ref);                                  /* ezscint.h:57 */
int32_t c_gdgaxes(int32_t gdid, float *ax, float *ay);                                                                               /* ezscint.h:96 */
int32_t c_gdll(int32_t gdid, float *lat, float *lon);                                                                                /* ezscint.h:62 */
int32_t c_gdxyfll(int32_t gdid, float *x, float *y, float *lat, float *lon, int32_t n);                                              /* ezscint.h:138 ; gdxyfll.c:29-104 (c_gdxyfll_new), :162 */
int32_t c_gdxyfll_orig(int32_t gdid, float *x, float *y, float *lat, float *lon, int32_t n);                                         /* ez_funcdef.h:178 ; gdxyfll.c:107-159: without the row inversion of y-inverted 'G' grids */

/* ---- options (thread-local, string keyed; src/interp/ezsetopt.c:59-215) ------------------ */
int32_t c_ezsetopt(char *option, char *value);                                                                                       /* ezscint.h:78 ; 0 ok / -1 */
int32_t c_ezgetopt(char *option, char *value);                                                                                       /* ezscint.h:35 */
int32_t c_ezsetval(char *option, float fvalue);                                                                                      /* ezscint.h:81 */
int32_t c_ezsetival(char *option, int32_t ivalue);                                                                                   /* ezscint.h:84 */
int32_t c_ezgetval(char *option, float *fvalue);                                                                                     /* ezscint.h:38 */
int32_t c_ezgetival(char *option, int32_t *ivalue);                                                                                  /* ezscint.h:41 */

/* ---- interpolation, HOST pointers (reference semantics) ---------------------------------- */
int32_t c_ezsint(float *zout, float *zin);                                                                                           /* ezscint.h:87 ; ezsint.c:38 ; 0 ok, 1 same grid, 2 extrapolated, -1 error */
int32_t c_ezuvint(float *uuout, float *vvout, float *uuin, float *vvin);                                                             /* ezscint.h:90 ; ezuvint.c:32 */
int32_t c_ezwdint(float *spdout, float *dirout, float *uuin, float *vvin);                                                           /* ezscint.h:93 ; ezwdint.c:37,62 : wind speed / direction on the target grid */
int32_t c_gdxysint(float *zout, float *zin, int32_t gdin, float *x, float *y, int32_t npts);                                         /* src/interp/gdxysint.h:4 ; gdxysint.c:30 */
int32_t c_gdxysval(int32_t gdin, float *zout, float *zin, float *x, float *y, int32_t n);                                            /* ezscint.h:120 ; gdxysval.c:50 */
int32_t c_gdllsval(int32_t gdid, float *zout, float *zin, float *lat, float *lon, int32_t n);                                        /* ezscint.h:108 ; gdllsval.c:33 : locate + c_gdxysval */
int32_t c_gdxyvval(int32_t gdin, float *uuout, float *vvout, float *uuin, float *vvin, float *x, float *y, int32_t n);               /* ezscint.h:126 ; gdxyvval.c:89 : two scalar interpolations */
int32_t c_gdllvval(int32_t gdid, float *uuout, float *vvout, float *uuin, float *vvin, float *lat, float *lon, int32_t n);           /* ezscint.h:111 ; gdllvval.c:34 */

/* ---- Fortran twins (f77name(x) = x_) ------------------------------------------------------ */
int32_t ezqkdef_(int32_t *ni, int32_t *nj, char *grtyp, int32_t *ig1, int32_t *ig2, int32_t *ig3, int32_t *ig4, int32_t *iunit, int32_t lengrtyp);
int32_t ezgdef_fmem_(int32_t *ni, int32_t *nj, char *grtyp, char *grref, int32_t *ig1, int32_t *ig2, int32_t *ig3, int32_t *ig4,
                     float *ax, float *ay, int32_t lengrtyp, int32_t lengrref);
int32_t ezdefset_(int32_t *gdout, int32_t *gdin);
int32_t ezgdef_supergrid_(int32_t *ni, int32_t *nj, char *grtyp, char *grref, int32_t *vercode, int32_t *nsubgrids, int32_t *subgrid, int32_t lengrtyp, int32_t lengrref);   /* ezscint.h:31 */
int32_t ezsetopt_(char *option, char *value, int32_t lenoption, int32_t lenvalue);
int32_t ezsint_(float *zout, float *zin);
int32_t ezuvint_(float *uuout, float *vvout, float *uuin, float *vvin);
int32_t ezwdint_(float *spdout, float *dirout, float *uuin, float *vvin);                    /* ezwdint.c:28 */
int32_t gdllsval_(int32_t *gdid, float *zout, float *zin, float *lat, float *lon, int32_t *n);
int32_t gdxyvval_(int32_t *gdin, float *uuout, float *vvout, float *uuin, float *vvin, float *x, float *y, int32_t *n);
int32_t gdllvval_(int32_t *gdid, float *uuout, float *vvout, float *uuin, float *vvin, float *lat, float *lon, int32_t *n);
int32_t gdxysint_(float *zout, float *zin, int32_t *gdin, float *x, float *y, int32_t *npts);
int32_t gdxysval_(int32_t *gdin, float *zout, float *zin, float *x, float *y, int32_t *n);
int32_t gdxyfll_(int32_t *gdid, float *x, float *y, float *lat, float *lon, int32_t *n);
int32_t gdll_(int32_t *gdid, float *lat, float *lon);
int32_t gdrls_(int32_t *gdin);

/* ---- the set-up routines the grid table is filled by, under the reference's own (internal) signatures: host only ---------------- */
/* Newton divided-difference tables of an irregular axis pair, cx(ni, 6), cy(j1:j2, 6): ez_nwtncof.inc:20-178 (f_ezscint.F90), called from ez_calcntncof.c:44 */
void ez_nwtncof_(float *cx, float *cy, const float *ax, const float *ay, const int32_t *ni, const int32_t *nj, const int32_t *i1, const int32_t *i2,
                 const int32_t *j1, const int32_t *j2, const int32_t *extension);
/* row / column bounds of a source and its longitude extension (0, 1, 2) from the grid's descriptors: ez_xpncof.c:48-226, ez_funcdef.h:67 */
void ez_xpncof(int32_t *i1, int32_t *i2, int32_t *j1, int32_t *j2, int32_t *extension, int32_t ni, int32_t nj, char grtyp, char grref,
               int32_t ig1, int32_t ig2, int32_t ig3, int32_t ig4, int32_t sym, float *ax, float *ay);

/* ---- additive: device-resident entry points ----------------------------------------------- */
/* All pointers are DEVICE pointers on the current HIP device; work is enqueued on the stream set by
 * ezhip_use_stream (default: the null stream) and NOT synchronised. */
/* FIRST CALL of a grid set on a per-point route (rotated / irregular sources: c_ezsint_dev, c_ezuvint_dev, c_ezsint_batch_dev): besides the kernels it allocates
 * and builds the set's caches -- located x, y (8 bytes per target point), the wind matrix (20), and for the staged-tile kernels a tile table plus a tile-ordered
 * copy of those streams (8 bytes per target point and degree for scalars, 12 for wind pairs) -- and SYNCHRONISES the stream two or three times while doing so
 * (hipMalloc, hipStreamSynchronize, one blocking copy).  Do not issue a set's first call inside a stream capture; call ezhip_prepare_set() (or any first call)
 * beforehand.  Later calls of the set only enqueue.  The caches live until c_gdrls of either grid.  Their total is bounded by a byte budget (default 4 GiB,
 * EZHIP_CACHE_MB at first use, or the call below); a set that does not fit keeps the gathering kernels: same results, slower. */
void ezhip_set_cache_budget_mb(int32_t mb);                 /* 0: no staged-tile caches at all */
/* additive (round 6): EXACT WINDS for c_ezuvint / c_ezwdint and their _dev forms.  Default (0): the wind chain of a grid pair (c_gdwdfuv + c_gduvfwd, gdwdfuv.c:29-100,
 * gduvfwd.c:29-96) is applied as a per-point 2 x 2 matrix made once per set from the chain itself, bicubic pairs from rotated sources are evaluated in REAL with a REAL*8 second
 * pass: within 2e-6 |V| of the reference (the size of the chain's own noise: its wind direction passes through REAL degrees).  1: every call runs the reference's chain as written
 * (speed / direction through REAL degrees, the C library's REAL trig restated in libm_exact.h, REAL*8 where the reference has it) on components interpolated by the scalar kernels:
 * at BASELINE configs[2] all 16 M values of nearest, bilinear and bicubic winds equal the reference build's BIT FOR BIT -- at ~5 x the time
 * (tests/test_gpu_wind_pin.py).  Process-wide. */
void ezhip_set_wind_exact(int32_t on);
int32_t ezhip_get_wind_exact(void);
long long ezhip_cache_bytes(void);                           /* bytes the staged-tile caches of all sets hold now */
/* Rotated sources ('E', Z-on-'E') are located on the device with the C library's REAL sinf / cosf / asinf / atan2f restated operation by operation
 * (librmn_amd/csrc/libm_exact.h: GNU libc 2.35, x86-64 FMA variants -- what the reference's ez_gfxyfll.c:38-57 reaches through the Fortran intrinsics on such a host).
 * 1: this process's C library is that one (sampled once); 0: it is not, and such sets are located by host threads through the library itself, as with EZHIP_HOST_LOCATE=1 */
int32_t ezhip_libm_exact_matches_host(void);
void    ezhip_use_stream(void *hip_stream);
/* 0 for the shipped library; 1 when it was built with -DEZHIP_DEVELOP (`make develop`: the kernels' development knock-outs, which an
 * environment variable can switch on, exist only in that build). */
int32_t ezhip_develop_build(void);
/* Optional, for callers of the host-array entry points who reuse their arrays: page-lock an array once (hipHostRegister).  c_ezsint between
 * two registered arrays uploads the source in row ranges while the finished rows of the result download (PCIe carries both directions at
 * once); between ordinary arrays the two copies run one after the other.  Unregister before freeing the memory.  0 / -1. */
int32_t ezhip_register_host_buffer(void *p, size_t nbytes);
int32_t ezhip_unregister_host_buffer(void *p);
int32_t c_ezsint_dev(float *d_zout, const float *d_zin);
int32_t c_ezuvint_dev(float *d_uuout, float *d_vvout, const float *d_uuin, const float *d_vvin);
int32_t c_ezwdint_dev(float *d_spdout, float *d_dirout, const float *d_uuin, const float *d_vvin);
int32_t c_gdxysint_dev(float *d_zout, const float *d_zin, int32_t gdin, const float *d_x, const float *d_y, int32_t npts);
int32_t c_gdxyfll_dev(int32_t gdid, float *d_x, float *d_y, const float *d_lat, const float *d_lon, int32_t n);   /* c_gdxyfll_orig on device data (no row inversion on y-inverted 'G' grids; not for hemispheric 'G' grids) */
/* nfields independent fields on the current grid set; field f at d_zin + f*ni_in*nj_in, d_zout + f*ni_out*nj_out */
int32_t c_ezsint_batch_dev(float *d_zout, const float *d_zin, int32_t nfields);
/* npairs wind pairs of the current grid set (ezuvint.c:51-94 per pair), device resident and contiguous: pair f's components at d_uuin / d_vvin + f * ni_in * nj_in,
 * its results at d_uuout / d_vvout + f * ni_out * nj_out -- the results of npairs c_ezuvint_dev calls, bit for bit.  Bicubic from a rotated source with wrap
 * (BASELINE cfg3), from the set's second pair on: ONE staged-tile launch for all pairs (x, y and the rotation of a point read once per batch); anything else:
 * pair by pair.  Returns like c_ezuvint_dev. */
int32_t c_ezuvint_batch_dev(float *d_uuout, float *d_vvout, const float *d_uuin, const float *d_vvin, int32_t npairs);
/* the same launch additionally leaves, per field, {min key, max key, 0} triples (order-preserving uint32 keys of the floats)
 * of every thread block's output at d_partials[f * stride_words + 3 k], k < *partials_per_field -- compact_float's
 * min/max pass fused into the interpolation.  -2: the plan has no single-launch path (use c_ezsint_batch_dev). */
int32_t ezhip_ezsint_batch_minmax_dev(float *d_zout, const float *d_zin, int32_t nfields, uint32_t *d_partials,
                                      int64_t stride_words, int32_t *partials_per_field);
/* the same extrema WITHOUT interpolating: bounds of every source window, exact evaluation of the few windows that can hold the extremum
 * (librmn_amd/csrc/ez_kernels.hip, k_bb_*).  One triple per field; d_flags[f] (device) = 1: no valid triple for field f (too many windows
 * qualify), use ezhip_ezsint_batch_minmax_only_dev for it.  -2: not applicable to the current grid set */
int32_t ezhip_ezsint_batch_minmax_bb_dev(const float *d_zin, int32_t nfields, uint32_t *d_partials, int64_t stride_words, int32_t *partials_per_field, int32_t *d_flags);
/* the two interpolation passes of the fused cfg5 pipeline (packers_hip.h: ezhip_ezsint_pack16_compress_batch_dev): A stores
 * nothing and leaves only the min/max partials; B stores compact_float's 16-bit tokens (two per word, first in the high half)
 * quantised with the {double minF, double mulFactor} found at d_params + f * param_stride_bytes.  -2: not on the k_sepx path */
int32_t ezhip_ezsint_batch_minmax_only_dev(const float *d_zin, int32_t nfields, uint32_t *d_partials, int64_t stride_words, int32_t *partials_per_field);
int32_t ezhip_ezsint_batch_tokens_dev(uint32_t *d_tokens, int64_t token_stride_words, const float *d_zin, int32_t nfields,
                                      const void *d_params, int64_t param_stride_bytes);
/* frees what the CALLING host thread owns in the library: device workspaces, page-locked staging buffers, its side stream.  Called automatically when a
 * thread that used the library ends; grids, sets and plans are process-wide and stay */
void ezhip_thread_release(void);
/* dimensions of the current grid set (the pair of c_ezdefset): -1 when none is defined */
int32_t ezhip_current_set_dims(int32_t *ni_in, int32_t *nj_in, int32_t *ni_out, int32_t *nj_out);
/* forces plan construction for the current set / options (what the reference does lazily in its first call) */
int32_t ezhip_prepare_set(void);
/* which kernel family the current set uses: 1 = separable (k_sepx, or its fallback k_sep), 2 = per-point (k_pts) */
int32_t ezhip_set_mode(void);
/* copies the current set's located x,y (the reference's gridset cache, ez_calcxy.c:56-134) to device arrays of ni_out*nj_out floats */
int32_t ezhip_set_xy_dev(float *d_x, float *d_y);
/* Asynchronous (*_dev) calls cannot report what a kernel finds out while it runs.  The one such condition -- k_sepx's bounded wait for
 * the pole values computed inside the same launch gave up: the polar rows of that call are NaN -- sets a sticky error word that the
 * next entry point (and the synchronising host-pointer calls, before they return) report as -1; this call reads and clears it. */
int32_t ezhip_device_error(void);
/* 1 when a HIP device is usable */
int32_t ezhip_available(void);

/* ---- wind conversions on their own, subgrid queries, small definitions ------------------------ */
int32_t c_gdwdfuv(int32_t gdid, float *spd_out, float *wd_out, float *uuin, float *vvin, float *latin, float *lonin, int32_t npts);          /* ezscint.h:132 ; gdwdfuv.c:29 */
int32_t c_gduvfwd(int32_t gdid, float *uugdout, float *vvgdout, float *uullin, float *vvllin, float *latin, float *lonin, int32_t npts);     /* ezscint.h:129 ; gduvfwd.c:29 ; E / Z targets refused */
int32_t c_gdwdfuv_dev(int32_t gdid, float *d_spd, float *d_wd, const float *d_uu, const float *d_vv, const float *d_lat, const float *d_lon, int32_t npts);
int32_t c_gduvfwd_dev(int32_t gdid, float *d_uu, float *d_vv, const float *d_spd, const float *d_wd, const float *d_lat, const float *d_lon, int32_t npts);
int32_t c_gdllfxy(int32_t gdid, float *lat, float *lon, float *x, float *y, int32_t n);      /* ezscint.h:102 ; gdllfxy.c:92 (host) */
int32_t c_gdxywdval(int32_t gdin, float *uuout, float *vvout, float *uuin, float *vvin, float *x, float *y, int32_t n);       /* gdxywdval.c:38 */
int32_t c_gdllwdval(int32_t gdid, float *uuout, float *vvout, float *uuin, float *vvin, float *lat, float *lon, int32_t n);   /* gdllwdval.c:36 */
int32_t c_gdxyzfll(int32_t gdid, float *x, float *y, float *lat, float *lon, int32_t n);     /* ezscint.h:141 ; gdxyzfll.c:33 (host) */
int32_t c_ezgdef(int32_t ni, int32_t nj, char *grtyp, char *grref, int32_t ig1, int32_t ig2, int32_t ig3, int32_t ig4, float *ax, float *ay);   /* ezscint.h:15 ; ezgdef.c:42 (memory form only) */
int32_t c_gdxpncf(int32_t gdin, int32_t *i1, int32_t *i2, int32_t *j1, int32_t *j2);        /* ezscint.h:117 ; gdxpncf.c:33 */
int32_t c_ezgdef_fll(int32_t ni, int32_t nj, float *lat, float *lon);                       /* ezscint.h:24 ; ezgdef_fll.c:36 ('Y' on 'L') */
int32_t c_ezget_nsubgrids(int32_t gdid);                                                    /* ezscint.h:169 */
int32_t c_ezget_subgridids(int32_t gdid, int32_t *subgrid);                                 /* ezscint.h:172 */

/* ---- masked interpolation (src/interp/ez_mask.c) ------------------------------------------- */
int c_gdsetmask(int gdid, int *mask);                                                                   /* ezscint.h:146 ; ez_mask.c:67 */
int c_gdgetmask(int gdid, int *mask);                                                                   /* ezscint.h:149 ; ez_mask.c:89 */
int c_ezsint_m(float *zout, float *zin);                                                                /* ezscint.h:152 ; "not implemented" in the reference too */
int c_ezuvint_m(float *uuout, float *vvout, float *uuin, float *vvin);                                  /* ezscint.h:155 */
int c_ezsint_mdm(float *zout, int *mask_out, float *zin, int *mask_in);                                 /* ezscint.h:158 ; ez_mask.c:127 */
int c_ezuvint_mdm(float *uuout, float *vvout, int *mask_out, float *uuin, float *vvin, int *mask_in);   /* ezscint.h:161 ; ez_mask.c:155 */
int c_ezsint_mask(int *mask_out, int *mask_in);                                                         /* ezscint.h:164 ; ez_mask.c:184 */
int c_ezget_mask_zones(int *mask_out, int *mask_in);                                                    /* ez_mask.c:231 */
int gdsetmask_(int *gdid, int *mask); int gdgetmask_(int *gdid, int *mask);                             /* Fortran twins, ezscint.h:145-163 */
int ezsint_mdm_(float *zout, int *mask_out, float *zin, int *mask_in);
int ezuvint_mdm_(float *uuout, float *vvout, int *mask_out, float *uuin, float *vvin, int *mask_in);
int ezsint_mask_(int *mask_out, int *mask_in); int ezget_mask_zones_(int *mask_out, int *mask_in);
int c_ezsint_mask_dev(int *d_mask_out, const int *d_mask_in);                                           /* additive: device pointers */
int c_ezget_mask_zones_dev(int *d_mask_out, const int *d_mask_in);
int c_ezsint_mdm_dev(float *d_zout, int *d_mask_out, const float *d_zin, const int *d_mask_in);
int c_ezuvint_mdm_dev(float *d_uuout, float *d_vvout, int *d_mask_out, const float *d_uuin, const float *d_vvin, const int *d_mask_in);

#ifdef __cplusplus
}
#endif
#endif
