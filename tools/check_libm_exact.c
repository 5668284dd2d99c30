/* librmn_amd/csrc/libm_exact.h against the C library of the machine it runs on, over every REAL argument:
 *   gcc -O2 -ffp-contract=off -mfma -pthread tools/check_libm_exact.c -o /tmp/check_libm_exact -lm && /tmp/check_libm_exact [threads] [pairs-in-millions] [stride]
 * sinf, cosf, asinf, atanf: all 2^32 bit patterns; atan2f: (y, 1) ... covered by atanf, plus random pairs of every magnitude and the rotated frame's
 * (components of unit vectors).  stride > 1: every stride-th bit pattern only (the quick form tests/test_libm_exact.py runs).  Prints the number of arguments whose results differ in any bit (two NaNs count as equal); exit code 1 if any does. */
#include <stdio.h>
#include <stdlib.h>
#include <pthread.h>
#include "../librmn_amd/csrc/libm_exact.h"

typedef struct { int t, nt; unsigned long long bad[5], pairs, stride; } job_t;
static int same(float a, float b) { return glx_bits(a) == glx_bits(b) || (a != a && b != b); }
static unsigned long long rng(unsigned long long *s) { *s ^= *s << 13; *s ^= *s >> 7; *s ^= *s << 17; return *s; }
static void *work(void *arg)
{
    job_t *j = (job_t *)arg;
    const unsigned long long per = (1ULL << 32) / j->nt, lo = per * j->t, hi = j->t == j->nt - 1 ? (1ULL << 32) : lo + per;
    for (unsigned long long u = lo; u < hi; u += j->stride) {
        const float x = glx_float((uint32_t)u);
        if (!same(glx_sinf(x), sinf(x))) { if (j->bad[0]++ < 3) fprintf(stderr, "sinf(%a): %a, library %a\n", x, glx_sinf(x), sinf(x)); }
        if (!same(glx_cosf(x), cosf(x))) { if (j->bad[1]++ < 3) fprintf(stderr, "cosf(%a): %a, library %a\n", x, glx_cosf(x), cosf(x)); }
        if (!same(glx_asinf(x), asinf(x))) { if (j->bad[2]++ < 3) fprintf(stderr, "asinf(%a): %a, library %a\n", x, glx_asinf(x), asinf(x)); }
        if (!same(glx_atanf(x), atanf(x))) { if (j->bad[3]++ < 3) fprintf(stderr, "atanf(%a): %a, library %a\n", x, glx_atanf(x), atanf(x)); }
    }
    unsigned long long s = 0x9E3779B97F4A7C15ULL * (j->t + 1);
    for (unsigned long long k = 0; k < j->pairs; k++) {
        float y, x;
        const unsigned long long r = rng(&s);
        if (k & 1) { y = glx_float((uint32_t)r); x = glx_float((uint32_t)(r >> 32)); }                 /* any two bit patterns */
        else {                                                                                       /* a direction: components of a unit vector, as ez_cal sees them */
            const double a = (double)(r >> 11) * (6.283185307179586 / 9007199254740992.0), c = (double)(rng(&s) >> 11) / 9007199254740992.0;
            y = (float)(sin(a) * c); x = (float)(cos(a) * c);
        }
        if (!same(glx_atan2f(y, x), atan2f(y, x))) { if (j->bad[4]++ < 3) fprintf(stderr, "atan2f(%a, %a): %a, library %a\n", y, x, glx_atan2f(y, x), atan2f(y, x)); }
    }
    return NULL;
}
int main(int argc, char **argv)
{
    const int nt = argc > 1 ? atoi(argv[1]) : 8;
    const unsigned long long pairs = (argc > 2 ? strtoull(argv[2], NULL, 10) : 2000ULL) * 1000000ULL;
    const unsigned long long stride = argc > 3 ? strtoull(argv[3], NULL, 10) : 1ULL;
    pthread_t th[256]; job_t job[256];
    for (int t = 0; t < nt; t++) { job[t] = (job_t){t, nt, {0, 0, 0, 0, 0}, pairs / nt, stride ? stride : 1}; pthread_create(&th[t], NULL, work, &job[t]); }
    unsigned long long bad[5] = {0, 0, 0, 0, 0};
    for (int t = 0; t < nt; t++) { pthread_join(th[t], NULL); for (int k = 0; k < 5; k++) bad[k] += job[t].bad[k]; }
    printf("arguments with a differing result: sinf %llu, cosf %llu, asinf %llu, atanf %llu of %llu each; atan2f %llu of %llu pairs\n",
           bad[0], bad[1], bad[2], bad[3], (1ULL << 32) / (stride ? stride : 1), bad[4], pairs / nt * nt);
    return (bad[0] | bad[1] | bad[2] | bad[3] | bad[4]) ? 1 : 0;
}
