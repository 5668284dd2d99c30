// micro-benchmarks for the k_sep memory patterns (development tool)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define NI 7200
#define NJ 3601
#define NIS 4400
#define NJS 2200
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int ROWS, int VEC>
__global__ __launch_bounds__(256) void k_store(float *out, float v)
{
    int c = (blockIdx.x * 256 + threadIdx.x) * VEC;
    int r0 = blockIdx.y * ROWS;
    if (c >= NI) return;
    for (int k = 0; k < ROWS; k++) {
        int r = r0 + k;
        if (r >= NJ) break;
        float *o = out + (size_t)r * NI + c;
        if (VEC == 1) o[0] = v + k;
        else { float4 q = {v, v + 1, v + 2, v + k}; *(float4 *)o = q; }
    }
}
// patch staging only (loads into LDS, one dummy store per block)
__global__ __launch_bounds__(256) void k_stage(float *out, const float *in)
{
    __shared__ float patch[20 * 272];
    int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int base = (int)(blockIdx.x * 156.4f), s0 = (int)(blockIdx.y * 9.77f);
    for (int row = wv; row < 14; row += 4) {
        const float *zr = in + (size_t)(s0 + row) * NIS;
        for (int q = 0; q < 3; q++) { int u = lane + 64 * q; int col = base + u; if (col >= NIS) col -= NIS; if (u < 164) patch[row * 272 + u] = zr[col]; }
    }
    __syncthreads();
    float s = 0; for (int k = 0; k < 14; k++) s += patch[k * 272 + (threadIdx.x % 160)];
    if (s == 12345.f) out[threadIdx.x] = s;
}
// nearest-like: stage + 16 rows of LDS read + store
template <int ROWS>
__global__ __launch_bounds__(256) void k_near(float *out, const float *in)
{
    __shared__ float patch[(ROWS * 5 / 8 + 5) * 272];
    int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int base = (int)(blockIdx.x * 156.4f), s0 = (int)(blockIdx.y * (ROWS * 0.6107f));
    const int nrows = ROWS * 5 / 8 + 4;
    for (int row = wv; row < nrows; row += 4) {
        if (s0 + row >= NJS) break;
        const float *zr = in + (size_t)(s0 + row) * NIS;
        for (int q = 0; q < 3; q++) { int u = lane + 64 * q; int col = base + u; if (col >= NIS) col -= NIS; if (u < 164) patch[row * 272 + u] = zr[col]; }
    }
    __syncthreads();
    int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= NI) return;
    int off = (int)(threadIdx.x * 0.611f);
    int r0 = blockIdx.y * ROWS;
    for (int k = 0; k < ROWS; k++) {
        int r = r0 + k; if (r >= NJ) break;
        out[(size_t)r * NI + c] = patch[(int)(k * 0.611f) * 272 + off];
    }
}
// V2: even blocks stage only, odd blocks store only (do phases of DIFFERENT blocks overlap?)
__global__ __launch_bounds__(256) void k_mix(float *out, const float *in)
{
    __shared__ float patch[20 * 272];
    int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (blockIdx.y & 1) {
        int by = blockIdx.y >> 1;
        int base = (int)(blockIdx.x * 156.4f), s0 = (int)(by * 2 * 9.77f);
        for (int h = 0; h < 2; h++) {
        for (int row = wv; row < 14; row += 4) {
            const float *zr = in + (size_t)(s0 + h * 9 + row) * NIS;
            for (int q = 0; q < 3; q++) { int u = lane + 64 * q; int col = base + u; if (col >= NIS) col -= NIS; if (u < 164) patch[row * 272 + u] = zr[col]; }
        }
        __syncthreads();
        float s = 0; for (int k = 0; k < 14; k++) s += patch[k * 272 + (threadIdx.x % 160)];
        if (s == 12345.f) out[threadIdx.x] = s;
        __syncthreads();
        }
    } else {
        int c = blockIdx.x * 256 + threadIdx.x; if (c >= NI) return;
        int r0 = (blockIdx.y >> 1) * 32;
        for (int k = 0; k < 32; k++) { int r = r0 + k; if (r >= NJ) break; out[(size_t)r * NI + c] = 1.f + k; }
    }
}
// V3: 512-thread blocks, waves 4-7 load the NEXT row-block's patch while waves 0-3 compute/store the current one
template <int K>
__global__ __launch_bounds__(512) void k_spec(float *out, const float *in)
{
    __shared__ float patch[2][14 * 176];
    int tid = threadIdx.x & 255, lane = threadIdx.x & 63, wv = (threadIdx.x >> 6) & 3;
    bool loader = threadIdx.x >= 256;
    int base = (int)(blockIdx.x * 156.4f);
    int rb0 = blockIdx.y * K;
    auto load = [&](int rb, int buf) {
        int s0 = (int)(rb * 9.77f);
        for (int row = wv; row < 14; row += 4) {
            if (s0 + row >= NJS) break;
            const float *zr = in + (size_t)(s0 + row) * NIS;
            for (int q = 0; q < 3; q++) { int u = lane + 64 * q; int col = base + u; if (col >= NIS) col -= NIS; if (u < 164) patch[buf][row * 176 + u] = zr[col]; }
        }
    };
    if (loader) load(rb0, 0);
    __syncthreads();
    int c = blockIdx.x * 256 + tid;
    int off = (int)(tid * 0.611f);
    for (int i = 0; i < K; i++) {
        int rb = rb0 + i, buf = i & 1;
        if (rb * 16 >= NJ) break;
        if (loader) { if (i + 1 < K) load(rb + 1, buf ^ 1); }
        else if (c < NI) {
            for (int k = 0; k < 16; k++) { int r = rb * 16 + k; if (r >= NJ) break; out[(size_t)r * NI + c] = patch[buf][(int)(k * 0.611f) * 176 + off]; }
        }
        __syncthreads();
    }
}
// loader-only: one wave per block streams `nrows` source rows of a 164-column strip through an LDS ring
template <int MODE, int DEPTH>   // MODE 0: LDS-DMA dword ; 1: global_load dword -> regs -> ds_write ; 2: LDS-DMA but 4 loader waves
__global__ __launch_bounds__(320) void k_loader(float *out, const float *in, int nrows_per_block)
{
    __shared__ float ring[40 * 192];
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int base = (int)(blockIdx.x * 156.4f);
    int s_first = blockIdx.y * nrows_per_block;
    if (MODE != 2 && wave != 4) return;
    unsigned coloff[3];
    for (int q = 0; q < 3; q++) { int col = base + min(lane + 64 * q, 163); if (col >= NIS) col -= NIS; coloff[q] = col; }
    int step = (MODE == 2) ? 5 : 1, start = (MODE == 2) ? wave : 0;
    float acc = 0;
    for (int s = start; s < nrows_per_block; s += step) {
        if (s_first + s >= NJS) break;
        const float *zr = in + (size_t)(s_first + s) * NIS;
        float *slot = ring + (s % 40) * 192;
        if (MODE == 1) {
            float v0 = zr[coloff[0]], v1 = zr[coloff[1]], v2 = zr[coloff[2]];
            slot[lane] = v0; slot[lane + 64] = v1; slot[lane + 128] = v2;
        } else {
            for (int q = 0; q < 3; q++)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(zr + coloff[q]),
                                                 (__attribute__((address_space(3))) void *)(slot + 64 * q), 4, 0, 0);
            if (s / step >= DEPTH) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * DEPTH) : "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    acc = ring[lane];
    if (acc == 12345.f) out[lane] = acc;
}
template <class F> float timeit(F f, int reps = 20)
{
    f(); CHECK(hipDeviceSynchronize());
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a); for (int i = 0; i < reps; i++) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms * 1000.f / reps;
}
int main()
{
    float *out, *in;
    size_t nout = (size_t)NI * NJ;
    CHECK(hipMalloc(&out, nout * 4 * 8)); CHECK(hipMalloc(&in, (size_t)NIS * NJS * 4 * 8));
    CHECK(hipMemset(in, 0, (size_t)NIS * NJS * 4 * 8));
    int it = 0;
    // rotate over 8 buffers so caches are not unrealistically warm
    #define OUTB (out + (size_t)((it++) & 7) * nout)
    #define INB (in + (size_t)((it) & 7) * NIS * NJS)
    printf("store dword 16 rows/blk : %7.2f us\n", timeit([&] { hipLaunchKernelGGL((k_store<16, 1>), dim3(29, 226), dim3(256), 0, 0, OUTB, 1.f); }));
    printf("store dword 64 rows/blk : %7.2f us\n", timeit([&] { hipLaunchKernelGGL((k_store<64, 1>), dim3(29, 57), dim3(256), 0, 0, OUTB, 1.f); }));
    printf("store dword 256 rows/blk: %7.2f us\n", timeit([&] { hipLaunchKernelGGL((k_store<256, 1>), dim3(29, 15), dim3(256), 0, 0, OUTB, 1.f); }));
    printf("store float4 16 rows/blk: %7.2f us\n", timeit([&] { hipLaunchKernelGGL((k_store<16, 4>), dim3(8, 226), dim3(256), 0, 0, OUTB, 1.f); }));
    printf("store float4 4 rows/blk : %7.2f us\n", timeit([&] { hipLaunchKernelGGL((k_store<4, 4>), dim3(8, 901), dim3(256), 0, 0, OUTB, 1.f); }));
    printf("stage only 29x226       : %7.2f us\n", timeit([&] { hipLaunchKernelGGL(k_stage, dim3(29, 226), dim3(256), 0, 0, OUTB, INB); }));
    printf("nearest-like 16 rows    : %7.2f us\n", timeit([&] { hipLaunchKernelGGL(k_near<16>, dim3(29, 226), dim3(256), 0, 0, OUTB, INB); }));
    printf("nearest-like 32 rows    : %7.2f us\n", timeit([&] { hipLaunchKernelGGL(k_near<32>, dim3(29, 113), dim3(256), 0, 0, OUTB, INB); }));
    printf("nearest-like 64 rows    : %7.2f us\n", timeit([&] { hipLaunchKernelGGL(k_near<64>, dim3(29, 57), dim3(256), 0, 0, OUTB, INB); }));
    printf("mix stage/store blocks  : %7.2f us\n", timeit([&] { hipLaunchKernelGGL(k_mix, dim3(29, 226), dim3(256), 0, 0, OUTB, INB); }));
    printf("specialized K=4         : %7.2f us\n", timeit([&] { hipLaunchKernelGGL(k_spec<4>, dim3(29, 57), dim3(512), 0, 0, OUTB, INB); }));
    printf("specialized K=8         : %7.2f us\n", timeit([&] { hipLaunchKernelGGL(k_spec<8>, dim3(29, 29), dim3(512), 0, 0, OUTB, INB); }));
    printf("specialized K=2         : %7.2f us\n", timeit([&] { hipLaunchKernelGGL(k_spec<2>, dim3(29, 113), dim3(512), 0, 0, OUTB, INB); }));
    printf("specialized K=16        : %7.2f us\n", timeit([&] { hipLaunchKernelGGL(k_spec<16>, dim3(29, 15), dim3(512), 0, 0, OUTB, INB); }));
    printf("loader-only DMA depth20 : %7.2f us\n", timeit([&] { hipLaunchKernelGGL((k_loader<0, 20>), dim3(29, 35), dim3(320), 0, 0, OUTB, INB, 63); }));
    printf("loader-only DMA depth8  : %7.2f us\n", timeit([&] { hipLaunchKernelGGL((k_loader<0, 8>), dim3(29, 35), dim3(320), 0, 0, OUTB, INB, 63); }));
    printf("loader-only regs        : %7.2f us\n", timeit([&] { hipLaunchKernelGGL((k_loader<1, 8>), dim3(29, 35), dim3(320), 0, 0, OUTB, INB, 63); }));
    printf("loader 5 waves DMA d12  : %7.2f us\n", timeit([&] { hipLaunchKernelGGL((k_loader<2, 12>), dim3(29, 35), dim3(320), 0, 0, OUTB, INB, 63); }));
    printf("loader-only DMA 29x140  : %7.2f us\n", timeit([&] { hipLaunchKernelGGL((k_loader<0, 20>), dim3(29, 140), dim3(320), 0, 0, OUTB, INB, 16); }));
    return 0;
}
