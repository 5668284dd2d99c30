// The memory floor of the headline launch, measured: 32 fields per launch, a source of 4400 x 2200 floats read once and a target of 7200 x 3601 floats written once per
// field (k_sepx's HBM traffic: 28 % reads, 72 % writes) with perfectly coalesced float4 streams and no arithmetic -- next to a pure fill and a 1:1 copy of the same
// bytes.  hipcc --offload-arch=gfx950 -O3 tools/ubench_floor.hip -o tools/ubench_floor && tools/ubench_floor      (development tool; results in profiles/r04_experiments.txt)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
static const size_t NOUT = (size_t)7200 * 3601, NIN = (size_t)4400 * 2200;
static const int F = 32;

// every thread writes PER float4 of the target; the source float4 it reads is the one at the same relative position (neighbouring threads share source lines: each
// source byte comes from HBM once, as in k_sepx, whose traffic is 1.012 x algorithmic)
template <int MODE>     // 0: fill, 1: mix (read source, write target), 2: copy 1:1 (target-sized read)
__global__ __launch_bounds__(256) void k_floor(f4 *out, const f4 *in, size_t nout4, size_t nin4, float v)
{
    const double ratio = (double)nin4 / (double)nout4;
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nout4; i += stride) {
        f4 q = {v, v, v, v};
        if (MODE == 1) { size_t s = (size_t)((double)i * ratio); if (s >= nin4) s = nin4 - 1; q = __builtin_nontemporal_load(in + s); }
        if (MODE == 2) q = __builtin_nontemporal_load(in + i);
        __builtin_nontemporal_store(q, out + i);
    }
}
template <class L> static double timeit(L launch, int reps = 20)
{
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int k = 0; k < 5; k++) launch();
    CHECK(hipDeviceSynchronize());
    std::vector<float> t;
    for (int b = 0; b < 5; b++) {
        CHECK(hipEventRecord(e0, 0));
        for (int k = 0; k < reps; k++) launch();
        CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms * 1e3f / reps);
    }
    std::sort(t.begin(), t.end());
    return t[2];
}
int main()
{
    f4 *out, *in, *big;
    const size_t nout4 = NOUT * F / 4, nin4 = NIN * F / 4;
    CHECK(hipMalloc(&out, nout4 * 16)); CHECK(hipMalloc(&in, nin4 * 16)); CHECK(hipMalloc(&big, nout4 * 16));
    CHECK(hipMemset(in, 0, nin4 * 16)); CHECK(hipMemset(big, 0, nout4 * 16));
    const double wb = (double)nout4 * 16, rb = (double)nin4 * 16;
    for (int blocks : {2048, 4096, 8192, 16384, 65536}) {
        const double tf = timeit([&] { hipLaunchKernelGGL(k_floor<0>, dim3(blocks), dim3(256), 0, 0, out, in, nout4, nin4, 1.f); });
        const double tm = timeit([&] { hipLaunchKernelGGL(k_floor<1>, dim3(blocks), dim3(256), 0, 0, out, in, nout4, nin4, 1.f); });
        const double tc = timeit([&] { hipLaunchKernelGGL(k_floor<2>, dim3(blocks), dim3(256), 0, 0, out, big, nout4, nin4, 1.f); });
        printf("%6d blocks: fill %7.1f us/launch = %5.2f us/field = %5.2f TB/s | read source + write target %7.1f us = %5.2f us/field = %5.2f TB/s | copy 1:1 %7.1f us = %5.2f TB/s\n",
               blocks, tf, tf / F, wb / tf / 1e6, tm, tm / F, (wb + rb) / tm / 1e6, tc, 2 * wb / tc / 1e6);
    }
    return 0;
}
