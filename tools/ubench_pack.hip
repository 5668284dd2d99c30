// ubench_pack.hip -- access-pattern study for compact_float's two passes on one cfg2 field (25 927 200 floats):
//   the min / max pass (read only) and the 16-bit token pass (16 B in, 8 B out per 4 points), cold (8 buffers in
//   rotation, 830 MB > the 256 MiB Infinity Cache) and warm (one buffer).
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_pack tools/ubench_pack.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <functional>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

static const size_t N = 25927200;      // 7200 x 3601
static const int NBUF = 8;

__device__ __forceinline__ unsigned f2key(float f) { unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
#define ST1(r) do { unsigned r_ = (r), key_ = f2key(__uint_as_float(r_)); kmin = min(kmin, key_); kmax = max(kmax, key_); kor |= r_; } while (0)
#define ST4(u) do { ST1((u).x); ST1((u).y); ST1((u).z); ST1((u).w); } while (0)

__device__ __forceinline__ void block_out(unsigned *part, unsigned kmin, unsigned kmax, unsigned kor)
{
    for (int off = 32; off > 0; off >>= 1) {
        kmin = min(kmin, (unsigned)__shfl_down((int)kmin, off, 64));
        kmax = max(kmax, (unsigned)__shfl_down((int)kmax, off, 64));
        kor |= (unsigned)__shfl_down((int)kor, off, 64);
    }
    __shared__ unsigned sh[3][16];
    int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if ((threadIdx.x & 63) == 0) { sh[0][w] = kmin; sh[1][w] = kmax; sh[2][w] = kor; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < nw; i++) { kmin = min(kmin, sh[0][i]); kmax = max(kmax, sh[1][i]); kor |= sh[2][i]; }
        part[3 * blockIdx.x] = kmin; part[3 * blockIdx.x + 1] = kmax; part[3 * blockIdx.x + 2] = kor;
    }
}

// S0: the product's shape: one pass per block, 4 loads per lane a quarter of the array apart
__global__ __launch_bounds__(256) void s_far(unsigned *part, const uint4 *a4, size_t n4)
{
    unsigned kmin = ~0u, kmax = 0, kor = 0;
    size_t step = (size_t)gridDim.x * 256, i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * step < n4; i += 4 * step) { uint4 u = a4[i], v = a4[i + step], x = a4[i + 2 * step], y = a4[i + 3 * step]; ST4(u); ST4(v); ST4(x); ST4(y); }
    for (; i < n4; i += step) { uint4 u = a4[i]; ST4(u); }
    block_out(part, kmin, kmax, kor);
}
// S1: a block reads U x 4 KB contiguous (U loads per lane, 4 KB apart)
template <int U, int NT, int TPB>
__global__ __launch_bounds__(TPB) void s_near(unsigned *part, const uint4 *a4, size_t n4)
{
    unsigned kmin = ~0u, kmax = 0, kor = 0;
    size_t i = (size_t)blockIdx.x * (TPB * U) + threadIdx.x;
    uint4 u[U];
#pragma unroll
    for (int k = 0; k < U; k++) {
        size_t j = i + (size_t)k * TPB;
        if (NT) { v4u t = __builtin_nontemporal_load((const v4u *)&a4[j]); u[k] = make_uint4(t.x, t.y, t.z, t.w); } else u[k] = a4[j];
    }
#pragma unroll
    for (int k = 0; k < U; k++) ST4(u[k]);
    block_out(part, kmin, kmax, kor);
}
// S2: persistent grid, grid-stride, U loads in flight
template <int U>
__global__ __launch_bounds__(256) void s_persist(unsigned *part, const uint4 *a4, size_t n4)
{
    unsigned kmin = ~0u, kmax = 0, kor = 0;
    size_t chunk = (size_t)256 * U;
    for (size_t base = (size_t)blockIdx.x * chunk; base < n4; base += (size_t)gridDim.x * chunk) {
        uint4 u[U];
#pragma unroll
        for (int k = 0; k < U; k++) { size_t j = base + threadIdx.x + (size_t)k * 256; u[k] = a4[j]; }
#pragma unroll
        for (int k = 0; k < U; k++) ST4(u[k]);
    }
    block_out(part, kmin, kmax, kor);
}

struct pp_t { double minF, mul; };
#define TK(v) ((unsigned)(long long)(((double)(v) - minF) * mul) & 0xFFFFu)
// P0: the product's shape: 32 B in per lane (two float4 next to each other), 16 B out
__global__ __launch_bounds__(256) void p_pair(uint4 *out, const float4 *a, size_t n8, const pp_t *pp)
{
    size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n8) return;
    const double minF = pp->minF, mul = pp->mul;
    float4 x = a[2 * k], y = a[2 * k + 1];
    uint4 o;
    o.x = TK(x.x) << 16 | TK(x.y); o.y = TK(x.z) << 16 | TK(x.w); o.z = TK(y.x) << 16 | TK(y.y); o.w = TK(y.z) << 16 | TK(y.w);
    out[k] = o;
}
// P1: 16 B in, 8 B out per lane and load, U loads per lane 4 KB apart (every wave instruction contiguous)
template <int U, int NTL, int NTS>
__global__ __launch_bounds__(256) void p_lin(uint2 *out, const float4 *a, size_t n4, const pp_t *pp)
{
    const double minF = pp->minF, mul = pp->mul;
    size_t i = (size_t)blockIdx.x * (256 * U) + threadIdx.x;
    float4 x[U];
#pragma unroll
    for (int k = 0; k < U; k++) { size_t j = i + (size_t)k * 256; if (NTL) { v4f t = __builtin_nontemporal_load((const v4f *)&a[j]); x[k] = make_float4(t.x, t.y, t.z, t.w); } else x[k] = a[j]; }
#pragma unroll
    for (int k = 0; k < U; k++) {
        size_t j = i + (size_t)k * 256;
        uint2 o; o.x = TK(x[k].x) << 16 | TK(x[k].y); o.y = TK(x[k].z) << 16 | TK(x[k].w);
        if (NTS) { v2u t; t.x = o.x; t.y = o.y; __builtin_nontemporal_store(t, (v2u *)&out[j]); } else out[j] = o;
    }
}
// P2: P0 with two pairs per lane
__global__ __launch_bounds__(256) void p_pair2(uint4 *out, const float4 *a, size_t n8, const pp_t *pp)
{
    size_t k = (size_t)blockIdx.x * 512 + threadIdx.x;
    const double minF = pp->minF, mul = pp->mul;
    float4 x0, y0, x1, y1;
    bool v0 = k < n8, v1 = k + 256 < n8;
    if (v0) { x0 = a[2 * k]; y0 = a[2 * k + 1]; }
    if (v1) { x1 = a[2 * (k + 256)]; y1 = a[2 * (k + 256) + 1]; }
    if (v0) { uint4 o; o.x = TK(x0.x) << 16 | TK(x0.y); o.y = TK(x0.z) << 16 | TK(x0.w); o.z = TK(y0.x) << 16 | TK(y0.y); o.w = TK(y0.z) << 16 | TK(y0.w); out[k] = o; }
    if (v1) { uint4 o; o.x = TK(x1.x) << 16 | TK(x1.y); o.y = TK(x1.z) << 16 | TK(x1.w); o.z = TK(y1.x) << 16 | TK(y1.y); o.w = TK(y1.z) << 16 | TK(y1.w); out[k + 256] = o; }
}

// S3: s_near + "last block done": the partials' reduction and the header step without a launch of their own.
// Two counter levels: same-address atomics cost ~23 ns each, 6330 of them on one word would serialise the kernel.
__device__ unsigned g_cnt[65];
template <int U, int NT>
__global__ __launch_bounds__(256) void s_near_last(unsigned *part, const uint4 *a4, size_t n4, unsigned *stats, pp_t *pp)
{
    unsigned kmin = ~0u, kmax = 0, kor = 0;
    size_t i = (size_t)blockIdx.x * (256 * U) + threadIdx.x;
    uint4 u[U];
#pragma unroll
    for (int k = 0; k < U; k++) {
        size_t j = i + (size_t)k * 256;
        if (NT) { v4u t = __builtin_nontemporal_load((const v4u *)&a4[j]); u[k] = make_uint4(t.x, t.y, t.z, t.w); } else u[k] = a4[j];
    }
#pragma unroll
    for (int k = 0; k < U; k++) ST4(u[k]);
    block_out(part, kmin, kmax, kor);
    __shared__ int last;
    if (threadIdx.x == 0) {
        int is_last = 0;
        __threadfence();
        unsigned grp = blockIdx.x & 63u, ngrp = gridDim.x < 64u ? gridDim.x : 64u;
        unsigned gsize = (gridDim.x - grp + 63u) / 64u;
        if (atomicAdd(&g_cnt[grp], 1u) == gsize - 1) {
            g_cnt[grp] = 0;
            __threadfence();
            if (atomicAdd(&g_cnt[64], 1u) == ngrp - 1) { g_cnt[64] = 0; is_last = 1; }
        }
        last = is_last;
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    kmin = ~0u; kmax = 0; kor = 0;
    const unsigned *p = part;
    for (int b = threadIdx.x; b < (int)gridDim.x; b += 256) {
        kmin = min(kmin, __builtin_nontemporal_load(&p[3 * b])); kmax = max(kmax, __builtin_nontemporal_load(&p[3 * b + 1])); kor |= __builtin_nontemporal_load(&p[3 * b + 2]);
    }
    __syncthreads();
    for (int off = 32; off > 0; off >>= 1) {
        kmin = min(kmin, (unsigned)__shfl_down((int)kmin, off, 64));
        kmax = max(kmax, (unsigned)__shfl_down((int)kmax, off, 64));
        kor |= (unsigned)__shfl_down((int)kor, off, 64);
    }
    __shared__ unsigned sh2[3][4];
    if ((threadIdx.x & 63) == 0) { sh2[0][threadIdx.x >> 6] = kmin; sh2[1][threadIdx.x >> 6] = kmax; sh2[2][threadIdx.x >> 6] = kor; }
    __syncthreads();
    if (threadIdx.x == 0) {
        stats[0] = min(min(sh2[0][0], sh2[0][1]), min(sh2[0][2], sh2[0][3]));
        stats[1] = max(max(sh2[1][0], sh2[1][1]), max(sh2[1][2], sh2[1][3]));
        stats[2] = sh2[2][0] | sh2[2][1] | sh2[2][2] | sh2[2][3];
        pp->minF = -1000.0; pp->mul = 16.0;
    }
}
// P3: p_lin, blocks in descending order (the second pass meets the most recently read lines first)
template <int U, int NTL, int NTS>
__global__ __launch_bounds__(256) void p_lin_rev(uint2 *out, const float4 *a, size_t n4, const pp_t *pp)
{
    const double minF = pp->minF, mul = pp->mul;
    size_t i = (size_t)(gridDim.x - 1 - blockIdx.x) * (256 * U) + threadIdx.x;
    float4 x[U];
#pragma unroll
    for (int k = 0; k < U; k++) { size_t j = i + (size_t)k * 256; if (NTL) { v4f t = __builtin_nontemporal_load((const v4f *)&a[j]); x[k] = make_float4(t.x, t.y, t.z, t.w); } else x[k] = a[j]; }
#pragma unroll
    for (int k = 0; k < U; k++) {
        size_t j = i + (size_t)k * 256;
        uint2 o; o.x = TK(x[k].x) << 16 | TK(x[k].y); o.y = TK(x[k].z) << 16 | TK(x[k].w);
        if (NTS) { v2u t; t.x = o.x; t.y = o.y; __builtin_nontemporal_store(t, (v2u *)&out[j]); } else out[j] = o;
    }
}
// the header step: one block reduces nb partial triples (the product's k_cf_header without the header arithmetic)
__global__ __launch_bounds__(256) void k_final(unsigned *stats, const unsigned *p, int nb, pp_t *pp)
{
    unsigned kmin = ~0u, kmax = 0, kor = 0;
    for (int b = threadIdx.x; b < nb; b += 256) { kmin = min(kmin, p[3 * b]); kmax = max(kmax, p[3 * b + 1]); kor |= p[3 * b + 2]; }
    block_out(stats, kmin, kmax, kor);
    if (threadIdx.x == 0) { pp->minF = -1000.0; pp->mul = 16.0; }
}

static float timeit(std::function<void(int)> f, int reps = 24)
{
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 8; i++) f(i);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; i++) f(i);
    CHECK(hipEventRecord(e1, 0)); CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / reps;
}

int main()
{
    setvbuf(stdout, NULL, _IONBF, 0);
    std::vector<float> h(N);
    unsigned s = 12345u;
    for (size_t i = 0; i < N; i++) { s = s * 1664525u + 1013904223u; h[i] = -1000.f + (float)(s >> 8) * (1.f / 16777216.f) * 3000.f; }
    float *in[NBUF]; uint4 *out[NBUF];
    for (int b = 0; b < NBUF; b++) {
        CHECK(hipMalloc(&in[b], N * 4 + (1 << 20))); CHECK(hipMalloc(&out[b], N * 2 + (1 << 20)));   // padded: the near / linear kernels run whole blocks past the end
        CHECK(hipMemset(in[b], 0, N * 4 + (1 << 20)));
        CHECK(hipMemcpy(in[b], h.data(), N * 4, hipMemcpyHostToDevice));
    }
    unsigned *part; CHECK(hipMalloc(&part, 3 * 65536 * 4)); pp_t *pp; CHECK(hipMalloc(&pp, sizeof(pp_t)));
    pp_t hp = {-1000.0, 16.0}; CHECK(hipMemcpy(pp, &hp, sizeof hp, hipMemcpyHostToDevice));
    const size_t n4 = N / 4, n8 = N / 8;
    auto rate = [&](const char *name, double bytes, float cold, float warm) {
        printf("%-44s cold %7.2f us %6.2f TB/s   warm %7.2f us %6.2f TB/s\n", name, cold, bytes / cold / 1e6, warm, bytes / warm / 1e6);
    };
#define BOTH(name, bytes, LAUNCH) do { printf("[%s]\n", name); float c = timeit([&](int it) { int b = it % NBUF; (void)b; LAUNCH; }); \
                                       float w = timeit([&](int it) { int b = 0; (void)it; LAUNCH; }); rate(name, bytes, c, w); } while (0)
    const double RB = N * 4.0, PB = N * 6.0;
    int nb0 = (int)((N + 4095) / 4096);
    BOTH("stats far x4 (product)", RB, hipLaunchKernelGGL(s_far, dim3(nb0), dim3(256), 0, 0, part, (const uint4 *)in[b], n4));
#define SNEAR(U, NT, TPB) BOTH("stats near U=" #U " nt=" #NT " tpb=" #TPB, RB, hipLaunchKernelGGL((s_near<U, NT, TPB>), dim3((unsigned)((n4 + (size_t)TPB * U - 1) / ((size_t)TPB * U))), dim3(TPB), 0, 0, part, (const uint4 *)in[b], n4))
    SNEAR(1, 0, 256); SNEAR(2, 0, 256); SNEAR(4, 0, 256); SNEAR(8, 0, 256); SNEAR(4, 1, 256); SNEAR(8, 1, 256); SNEAR(4, 0, 512); SNEAR(4, 0, 1024); SNEAR(2, 0, 1024);
#define SPERS(U, G) BOTH("stats persistent U=" #U " grid=" #G, RB, hipLaunchKernelGGL(s_persist<U>, dim3(G), dim3(256), 0, 0, part, (const uint4 *)in[b], n4))
    SPERS(4, 2048); SPERS(8, 2048); SPERS(4, 4096); SPERS(8, 1024);
    BOTH("header step: 6330 partials", 0.0, hipLaunchKernelGGL(k_final, dim3(1), dim3(256), 0, 0, part + 3 * 60000, part, nb0, pp));
    BOTH("header step: 1583 partials", 0.0, hipLaunchKernelGGL(k_final, dim3(1), dim3(256), 0, 0, part + 3 * 60000, part, nb0 / 4, pp));
    BOTH("pack pair (product)", PB, hipLaunchKernelGGL(p_pair, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, 0, out[b], (const float4 *)in[b], n8, pp));
    BOTH("pack pair x2", PB, hipLaunchKernelGGL(p_pair2, dim3((unsigned)((n8 + 511) / 512)), dim3(256), 0, 0, out[b], (const float4 *)in[b], n8, pp));
#define PLIN(U, NTL, NTS) BOTH("pack linear U=" #U " ntl=" #NTL " nts=" #NTS, PB, hipLaunchKernelGGL((p_lin<U, NTL, NTS>), dim3((unsigned)((n4 + 256 * U - 1) / (256 * U))), dim3(256), 0, 0, (uint2 *)out[b], (const float4 *)in[b], n4, pp))
    PLIN(1, 0, 0); PLIN(2, 0, 0); PLIN(4, 0, 0); PLIN(8, 0, 0); PLIN(4, 1, 0); PLIN(4, 0, 1); PLIN(4, 1, 1); PLIN(2, 1, 1);
    // the sequence of the product on rotating fields: stats -> header -> pack of the SAME field (the second read may hit the Infinity Cache)
    {
        float c = timeit([&](int it) { int b = it % NBUF;
            hipLaunchKernelGGL(s_far, dim3(nb0), dim3(256), 0, 0, part, (const uint4 *)in[b], n4);
            hipLaunchKernelGGL(k_final, dim3(1), dim3(256), 0, 0, part + 3 * 60000, part, nb0, pp);
            hipLaunchKernelGGL(p_pair, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, 0, out[b], (const float4 *)in[b], n8, pp); });
        printf("sequence product shapes (rotating)          %7.2f us\n", c);
        c = timeit([&](int it) { int b = it % NBUF;
            hipLaunchKernelGGL((s_near<4, 0, 256>), dim3((unsigned)((n4 + 1023) / 1024)), dim3(256), 0, 0, part, (const uint4 *)in[b], n4);
            hipLaunchKernelGGL(k_final, dim3(1), dim3(256), 0, 0, part + 3 * 60000, part, nb0, pp);
            hipLaunchKernelGGL((p_lin<4, 0, 0>), dim3((unsigned)((n4 + 1023) / 1024)), dim3(256), 0, 0, (uint2 *)out[b], (const float4 *)in[b], n4, pp); });
        printf("sequence near U=4 + linear U=4 (rotating)   %7.2f us\n", c);
        c = timeit([&](int it) { int b = it % NBUF;
            hipLaunchKernelGGL((s_near<4, 0, 256>), dim3((unsigned)((n4 + 1023) / 1024)), dim3(256), 0, 0, part, (const uint4 *)in[b], n4);
            hipLaunchKernelGGL(k_final, dim3(1), dim3(256), 0, 0, part + 3 * 60000, part, nb0, pp);
            hipLaunchKernelGGL((p_lin<4, 1, 1>), dim3((unsigned)((n4 + 1023) / 1024)), dim3(256), 0, 0, (uint2 *)out[b], (const float4 *)in[b], n4, pp); });
        printf("sequence near U=4 + linear U=4 nt (rotating) %7.2f us\n", c);

#define G4 dim3((unsigned)((n4 + 1023) / 1024))
#define SEQ(name, S, P) do { float c = timeit([&](int it) { int b = it % NBUF; S; hipLaunchKernelGGL(k_final, dim3(1), dim3(256), 0, 0, part + 3 * 60000, part, nb0, pp); P; }, 48); \
                             printf("sequence %-60s %7.2f us\n", name, c); } while (0)
#define SN(NT) hipLaunchKernelGGL((s_near<4, NT, 256>), G4, dim3(256), 0, 0, part, (const uint4 *)in[b], n4)
#define PL(NTL, NTS) hipLaunchKernelGGL((p_lin<4, NTL, NTS>), G4, dim3(256), 0, 0, (uint2 *)out[b], (const float4 *)in[b], n4, pp)
#define PR(NTL, NTS) hipLaunchKernelGGL((p_lin_rev<4, NTL, NTS>), G4, dim3(256), 0, 0, (uint2 *)out[b], (const float4 *)in[b], n4, pp)
        SEQ("near4 + lin4", SN(0), PL(0, 0));
        SEQ("near4 + lin4 nt-load", SN(0), PL(1, 0));
        SEQ("near4 + lin4 nt-store", SN(0), PL(0, 1));
        SEQ("near4 + lin4 nt-load nt-store", SN(0), PL(1, 1));
        SEQ("near4 nt + lin4 nt-load nt-store", SN(1), PL(1, 1));
        SEQ("near4 nt + lin4 nt-store", SN(1), PL(0, 1));
        SEQ("near4 + lin4 reversed", SN(0), PR(0, 0));
        SEQ("near4 + lin4 reversed nt-store", SN(0), PR(0, 1));
        SEQ("near4 + lin4 reversed nt-load nt-store", SN(0), PR(1, 1));
        SEQ("far x4 (product) + lin4 nt-load nt-store", hipLaunchKernelGGL(s_far, dim3(nb0), dim3(256), 0, 0, part, (const uint4 *)in[b], n4), PL(1, 1));
        SEQ("far x4 (product) + pair (product)", hipLaunchKernelGGL(s_far, dim3(nb0), dim3(256), 0, 0, part, (const uint4 *)in[b], n4),
            hipLaunchKernelGGL(p_pair, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, 0, out[b], (const float4 *)in[b], n8, pp));
        {   // the header step folded into the stats kernel
            float c = timeit([&](int it) { int b = it % NBUF;
                hipLaunchKernelGGL((s_near_last<4, 0>), G4, dim3(256), 0, 0, part, (const uint4 *)in[b], n4, part + 3 * 60000, pp); PL(1, 1); }, 48);
            printf("sequence %-60s %7.2f us\n", "near4 last-block header + lin4 nt-load nt-store", c);
            c = timeit([&](int it) { int b = it % NBUF;
                hipLaunchKernelGGL((s_near_last<4, 0>), G4, dim3(256), 0, 0, part, (const uint4 *)in[b], n4, part + 3 * 60000, pp); PR(1, 1); }, 48);
            printf("sequence %-60s %7.2f us\n", "near4 last-block header + lin4 reversed nt-load nt-store", c);
            c = timeit([&](int it) { int b = it % NBUF;
                hipLaunchKernelGGL((s_near_last<4, 0>), G4, dim3(256), 0, 0, part, (const uint4 *)in[b], n4, part + 3 * 60000, pp); }, 48);
            printf("%-69s %7.2f us\n", "stats near4 with the last-block header alone", c);
            unsigned hs[3], cnt[65]; CHECK(hipMemcpy(hs, part + 3 * 60000, 12, hipMemcpyDeviceToHost)); CHECK(hipMemcpyFromSymbol(cnt, HIP_SYMBOL(g_cnt), sizeof cnt));
            unsigned bad = 0; for (int i = 0; i < 65; i++) bad |= cnt[i];
            printf("last-block result: min key %08x max key %08x or %08x, counters all zero: %s\n", hs[0], hs[1], hs[2], bad ? "NO" : "yes");
            hipLaunchKernelGGL((s_near<4, 0, 256>), G4, dim3(256), 0, 0, part, (const uint4 *)in[0], n4);
            hipLaunchKernelGGL(k_final, dim3(1), dim3(256), 0, 0, part + 3 * 60000, part, nb0, pp);
            CHECK(hipMemcpy(hs, part + 3 * 60000, 12, hipMemcpyDeviceToHost));
            printf("two-kernel result: min key %08x max key %08x or %08x\n", hs[0], hs[1], hs[2]);
        }
    }
    return 0;
}
