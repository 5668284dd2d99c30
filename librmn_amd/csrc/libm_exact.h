/* The REAL elementary functions the reference's rotated-frame locate reaches through the Fortran intrinsics (ez_lac.inc:31-47 COS / SIN,
 * ez_cal.inc:22-47 ASIN / ATAN2), restated operation by operation as GNU libc 2.35 (x86-64, the FMA-selected variants this image's and the
 * GPU boxes' processors resolve sinf / cosf to) computes them, so that the device locates a rotated source with the bits c_gdxyfll has on the host:
 *
 *   sinf, cosf   sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c, sincosf.h, s_sincosf_data.c: REAL*8 polynomials on [-pi/4, pi/4] after a reduction by
 *                pi/2 (a scaled multiply for |x| < 120, a 192-bit window of 2/pi above); in the FMA variant every a + b * c of sinf_poly and
 *                the x - n * hpi of reduce_fast is ONE fused operation (read off libm.so.6's code), written here as fma()
 *   asinf        sysdeps/ieee754/flt-32/e_asinf.c (REAL arithmetic; the base x86-64 build has no fused operations)
 *   atanf        sysdeps/ieee754/flt-32/s_atanf.c, atan2f: e_atan2f.c (the same)
 *
 * Nothing here is taken on trust: tools/check_libm_exact.c compares each function with the C library's over EVERY REAL argument (2^32 of them;
 * atan2f over its quotient's 2^32 and a few 10^9 random pairs) on the host, tests/test_gpu_interp.py compares the device's x, y with c_gdxyfll's.
 * Compiled with -ffp-contract=off on both sides (the Makefile's flags): the only fused operations are the ones spelled fma().
 *
 * PROVENANCE AND LICENCE.  This header is not reference (librmn) code.  It is a restatement of algorithms published in the GNU C Library 2.35
 * (sysdeps/ieee754/flt-32/: s_sinf.c, s_cosf.c, sincosf.h, s_sincosf_data.c -- Copyright (C) 2018-2022 Free Software Foundation, Inc., contributed by
 * Arm Ltd.; e_asinf.c, s_atanf.c, e_atan2f.c -- derived from FreeBSD msun, Copyright (C) 1993 Sun Microsystems, Inc., "Permission to use, copy, modify,
 * and distribute this software is freely granted, provided that this notice is preserved").  The polynomial coefficients, reduction constants and the
 * 2/pi table are those sources' published values (the tables were cross-checked against the .rodata of this image's libm.so.6); the control flow was
 * rewritten for this file.  glibc is distributed under the GNU Lesser General Public License, version 2.1 or later: anyone redistributing a binary of
 * librmn_ez_hip.so should treat this header as LGPL-2.1-or-later material (its source is this file; relinking against a modified version is possible by
 * rebuilding librmn_amd/csrc).  It welds the product's rotated-frame results to glibc 2.35's REAL functions: a process whose C library computes them
 * otherwise is recognised at run time (ez_host.c: 16 384 sampled arguments per function at the first rotated set) and falls back to host locates through
 * its own library; the WIND chain (round 6: d_rotate*, k_wind_rotate) uses this header unconditionally -- the reference platform's answers. */
#ifndef LIBM_EXACT_H
#define LIBM_EXACT_H
#include <stdint.h>
#include <math.h>
#include <string.h>

#ifdef __HIPCC__
#define GLX __device__ __forceinline__
#define GLX_CONST __device__ const
#else
#define GLX static inline
#define GLX_CONST static const
#endif

GLX uint32_t glx_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
GLX float glx_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* s_sincosf_data.c: the second set is the first with cosine coefficients negated (quadrants 2 and 3) */
typedef struct { double sign[4], hpi_inv, hpi, c0, c1, c2, c3, c4, s1, s2, s3; } glx_sincos_t;
GLX_CONST glx_sincos_t glx_sincos_table[2] = {
    {{1.0, -1.0, -1.0, 1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0,
     0x1p0, -0x1.ffffffd0c621cp-2, 0x1.55553e1068f19p-5, -0x1.6c087e89a359dp-10, 0x1.99343027bf8c3p-16,
     -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13},
    {{1.0, -1.0, -1.0, 1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0,
     -0x1p0, 0x1.ffffffd0c621cp-2, -0x1.55553e1068f19p-5, 0x1.6c087e89a359dp-10, -0x1.99343027bf8c3p-16,
     -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13}};
/* 2/pi in 32-bit windows that advance by 8 bits */
GLX_CONST uint32_t glx_inv_pio4[24] = {
    0xa2, 0xa2f9, 0xa2f983, 0xa2f9836e, 0xf9836e4e, 0x836e4e44, 0x6e4e4415, 0x4e441529, 0x441529fc, 0x1529fc27, 0x29fc2757, 0xfc2757d1,
    0x2757d1f5, 0x57d1f534, 0xd1f534dd, 0xf534ddc0, 0x34ddc0db, 0xddc0db62, 0xc0db6295, 0xdb629599, 0x6295993c, 0x95993c43, 0x993c4390, 0x3c439041};

GLX float glx_sinf_poly(double x, double x2, const glx_sincos_t *p, int n)
{
    if ((n & 1) == 0) {
        const double x3 = x * x2, s1 = fma(x2, p->s3, p->s2), x7 = x3 * x2, s = fma(x3, p->s1, x);
        return (float)fma(x7, s1, s);
    }
    const double x4 = x2 * x2, c2 = fma(x2, p->c4, p->c3), c1 = fma(x2, p->c1, p->c0), x6 = x4 * x2, c = fma(x4, p->c2, c1);
    return (float)fma(x6, c2, c);
}
GLX double glx_reduce_fast(double x, const glx_sincos_t *p, int *np)
{
    const double r = x * p->hpi_inv;
    const int n = ((int32_t)r + 0x800000) >> 24;
    *np = n;
    return fma(-(double)n, p->hpi, x);
}
GLX double glx_reduce_large(uint32_t xi, int *np)
{
    const uint32_t *arr = &glx_inv_pio4[(xi >> 26) & 15];
    const int shift = (xi >> 23) & 7;
    uint64_t n, res0, res1, res2;
    xi = (xi & 0xffffff) | 0x800000;
    xi <<= shift;
    res0 = (uint32_t)(xi * arr[0]);
    res1 = (uint64_t)xi * arr[4];
    res2 = (uint64_t)xi * arr[8];
    res0 = (res2 >> 32) | (res0 << 32);
    res0 += res1;
    n = (res0 + (1ULL << 61)) >> 62;
    res0 -= n << 62;
    *np = (int)n;
    return (double)(int64_t)res0 * 0x1.921FB54442D18p-62;
}
GLX uint32_t glx_abstop12(float x) { return (glx_bits(x) >> 20) & 0x7ff; }

/* both functions: quadrant 0 directly, |x| < 120 by reduce_fast, the rest by reduce_large; cosine = sine's polynomial choice with n ^ 1 */
GLX float glx_sincosf_one(float y, int cosine)
{
    double x = (double)y, s;
    int n;
    const glx_sincos_t *p = &glx_sincos_table[0];
    const uint32_t top = glx_abstop12(y);
    if (top < glx_abstop12(0x1.921FB6p-1f)) {
        s = x * x;
        if (top < glx_abstop12(0x1p-12f)) return cosine ? 1.0f : y;
        return glx_sinf_poly(x, s, p, cosine);
    }
    if (top < glx_abstop12(120.0f)) {
        x = glx_reduce_fast(x, p, &n);
        s = p->sign[n & 3];
        if (n & 2) p = &glx_sincos_table[1];
        return glx_sinf_poly(x * s, x * x, p, n ^ cosine);
    }
    if (top < glx_abstop12(INFINITY)) {
        const uint32_t xi = glx_bits(y);
        const int sign = (int)(xi >> 31);
        x = glx_reduce_large(xi, &n);
        s = p->sign[(n + sign) & 3];
        if ((n + sign) & 2) p = &glx_sincos_table[1];
        return glx_sinf_poly(x * s, x * x, p, n ^ cosine);
    }
    return y - y;                                           /* infinities and NaNs: NaN (__math_invalidf) */
}
GLX float glx_sinf(float y) { return glx_sincosf_one(y, 0); }
GLX float glx_cosf(float y) { return glx_sincosf_one(y, 1); }

/* e_asinf.c */
GLX float glx_asinf(float x)
{
    const float one = 1.0f, huge = 1.000e+30f, pio2_hi = 1.57079637050628662109375f, pio2_lo = -4.37113900018624283e-8f, pio4_hi = 0.785398185253143310546875f;
    const float p0 = 1.666675248e-1f, p1 = 7.495297643e-2f, p2 = 4.547037598e-2f, p3 = 2.417951451e-2f, p4 = 4.216630880e-2f;
    float t, w, p, q, c, r, s;
    const int32_t hx = (int32_t)glx_bits(x), ix = hx & 0x7fffffff;
    if (ix == 0x3f800000) return x * pio2_hi + x * pio2_lo;
    if (ix > 0x3f800000) return (x - x) / (x - x);
    if (ix < 0x3f000000) {
        if (ix < 0x32000000) { if (huge + x > one) return x; }
        else {
            t = x * x;
            w = t * (p0 + t * (p1 + t * (p2 + t * (p3 + t * p4))));
            return x + x * w;
        }
    }
    w = one - fabsf(x);
    t = w * 0.5f;
    p = t * (p0 + t * (p1 + t * (p2 + t * (p3 + t * p4))));
    s = sqrtf(t);
    if (ix >= 0x3F79999A) t = pio2_hi - (2.0f * (s + s * p) - pio2_lo);
    else {
        w = glx_float(glx_bits(s) & 0xfffff000u);
        c = (t - w * w) / (s + w);
        r = p;
        p = 2.0f * s * r - (pio2_lo - 2.0f * c);
        q = pio4_hi - 2.0f * w;
        t = pio4_hi - (p - q);
    }
    return hx > 0 ? t : -t;
}

/* s_atanf.c */
GLX float glx_atanf(float x)
{
    const float atanhi[4] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
    const float atanlo[4] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
    const float aT[11] = {3.3333334327e-01f, -2.0000000298e-01f, 1.4285714924e-01f, -1.1111110449e-01f, 9.0908870101e-02f, -7.6918758452e-02f,
                          6.6610731184e-02f, -5.8335702866e-02f, 4.9768779427e-02f, -3.6531571299e-02f, 1.6285819933e-02f};
    const float one = 1.0f, huge = 1.0e30f;
    float w, s1, s2, z;
    const int32_t hx = (int32_t)glx_bits(x), ix = hx & 0x7fffffff;
    int id;
    if (ix >= 0x4c000000) {
        if (ix > 0x7f800000) return x + x;
        return hx > 0 ? atanhi[3] + atanlo[3] : -atanhi[3] - atanlo[3];
    }
    if (ix < 0x3ee00000) {
        if (ix < 0x31000000) { if (huge + x > one) return x; }
        id = -1;
    } else {
        x = fabsf(x);
        if (ix < 0x3f980000) {
            if (ix < 0x3f300000) { id = 0; x = (2.0f * x - one) / (2.0f + x); }
            else { id = 1; x = (x - one) / (x + one); }
        } else {
            if (ix < 0x401c0000) { id = 2; x = (x - 1.5f) / (one + 1.5f * x); }
            else { id = 3; x = -1.0f / x; }
        }
    }
    z = x * x;
    w = z * z;
    s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
    s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
    if (id < 0) return x - x * (s1 + s2);
    z = atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
    return hx < 0 ? -z : z;
}

/* e_atan2f.c */
GLX float glx_atan2f(float y, float x)
{
    const float tiny = 1.0e-30f, zero = 0.0f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    float z;
    const int32_t hx = (int32_t)glx_bits(x), hy = (int32_t)glx_bits(y), ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
    if (hx == 0x3f800000) return glx_atanf(y);
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
    if (iy == 0) {
        switch (m) {
        case 0: case 1: return y;
        case 2: return pi + tiny;
        default: return -pi - tiny;
        }
    }
    if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000) {
            switch (m) {
            case 0: return pi_o_4 + tiny;
            case 1: return -pi_o_4 - tiny;
            case 2: return 3.0f * pi_o_4 + tiny;
            default: return -3.0f * pi_o_4 - tiny;
            }
        } else {
            switch (m) {
            case 0: return zero;
            case 1: return -zero;
            case 2: return pi + tiny;
            default: return -pi - tiny;
            }
        }
    }
    if (iy == 0x7f800000) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    const int k = (iy - ix) >> 23;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
    else if (hx < 0 && k < -60) z = 0.0f;
    else z = glx_atanf(fabsf(y / x));
    switch (m) {
    case 0: return z;
    case 1: return glx_float(glx_bits(z) ^ 0x80000000u);
    case 2: return pi - (z - pi_lo);
    default: return (z - pi_lo) - pi;
    }
}
#endif
