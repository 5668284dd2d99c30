// irate.hip -- issue-rate micro-benchmark of the instructions k_sepx is made of (development tool).
// One block per CU x WAVES waves per SIMD, each wave runs REP x 32 independent copies of one instruction
// between two s_memtime reads; prints cycles per wave-instruction per SIMD (VALU) or per CU (LDS).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/irate tools/irate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>


#define BODY8(I) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7)

template <int KIND>
__global__ __launch_bounds__(1024) void k_rate(unsigned long long *out, const float *in, int lds_stride, int REP)
{
    extern __shared__ double lds[];
    double a[8], b = in[threadIdx.x], c = in[threadIdx.x + 1];
    float f[8]; unsigned u[8]; unsigned long long q[8];
    for (int i = 0; i < 8; i++) { a[i] = in[i + threadIdx.x]; f[i] = in[i + 8]; u[i] = threadIdx.x * 8 + i * lds_stride; q[i] = (unsigned long long)in + i; }
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = i;
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < REP; r++) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (KIND == 0) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
                if (KIND == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (KIND == 2) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[i]) : "v"(f[i]));
                if (KIND == 3) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(a[i]));
                if (KIND == 4) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
                if (KIND == 5) asm volatile("v_lshl_add_u64 %0, %0, 2, %1" : "+v"(q[i]) : "v"(q[(i + 1) & 7]));
                if (KIND == 6) asm volatile("v_readlane_b32 s20, %0, 3" :: "v"(u[i]) : "s20");
                if (KIND == 7) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
                if (KIND == 8) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
                if (KIND == 9) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (KIND == 10) asm volatile("ds_read_b64 %0, %1" : "=v"(a[i]) : "v"(u[i]));
                if (KIND == 11) { double x, y; asm volatile("ds_read2st64_b64 %0, %1 offset1:4" : "=v"(*(__attribute__((ext_vector_type(2))) double *)&a[i & 6]) : "v"(u[i])); (void)x; (void)y; }
                if (KIND == 12) asm volatile("ds_read_b32 %0, %1" : "=v"(f[i]) : "v"(u[i]));
                if (KIND == 13) asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(a[i]) : "v"(u[i]));
                if (KIND == 14) asm volatile("ds_write_b64 %0, %1" :: "v"(u[i]), "v"(a[i]));
                if (KIND == 15) asm volatile("ds_read_b128 %0, %1" : "=v"(*(__attribute__((ext_vector_type(2))) double *)&a[i & 6]) : "v"(u[i] & ~15u));
                if (KIND == 16) asm volatile("v_fma_f64 %0, %1, s[20:21], %0" : "+v"(a[i]) : "v"(b) : "s20", "s21");
                if (KIND == 17) asm volatile("v_lshl_add_u32 %0, %1, 11, %0" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
                if (KIND == 18) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b));
                if (KIND == 19) asm volatile("v_mov_b64 %0, %1" : "=v"(a[i]) : "v"(b));
                if (KIND == 20) asm volatile("ds_read2_b64 %0, %1 offset1:32" : "=v"(*(__attribute__((ext_vector_type(2))) double *)&a[i & 6]) : "v"(u[i]));
                if (KIND == 21) asm volatile("ds_read2_b64 %0, %1 offset1:33" : "=v"(*(__attribute__((ext_vector_type(2))) double *)&a[i & 6]) : "v"(u[i]));
                if (KIND == 22) asm volatile("ds_read2_b64 %0, %1 offset1:1" : "=v"(*(__attribute__((ext_vector_type(2))) double *)&a[i & 6]) : "v"(u[i] & ~15u));
                if (KIND == 23) asm volatile("ds_read_b32 %0, %1" : "=v"(f[i]) : "v"(u[i] >> 1));            /* lane stride 4 B */
                if (KIND == 24) asm volatile("ds_read_b32 %0, %1" : "=v"(f[i]) : "v"((u[i] * 5 >> 4) & ~3u)); /* ~0.61 dword per lane (x-pass pattern) */
                if (KIND == 25) asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(a[i]) : "v"((u[i] * 5 >> 4) & ~3u));
                if (KIND == 26) asm volatile("ds_write_b32 %0, %1" :: "v"(u[i] >> 1), "v"(f[i]));
                if (KIND == 27) asm volatile("ds_write_b128 %0, %1" :: "v"(u[i] * 2), "v"(*(__attribute__((ext_vector_type(2))) double *)&a[i & 6]));
                if (KIND == 28) asm volatile("global_store_dword %0, %1, off" :: "v"(q[i] ), "v"(f[i]));
            }
        }
        if (KIND >= 10 && KIND != 16 && KIND != 17 && KIND != 18 && KIND != 19 && KIND != 28) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    unsigned long long t1 = __builtin_readcyclecounter();
    double s = 0; float fs = 0; unsigned us = 0;
    for (int i = 0; i < 8; i++) { s += a[i]; fs += f[i]; us += u[i] + (unsigned)q[i]; }
    if (s == 1.2345 && fs == 3.f && us == 7) out[1] = 1;     // keep everything alive
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = t1 - t0;
}

template <int KIND>
static void run(const char *name, unsigned long long *d_out, float *d_in, int stride)
{
    static double ghz = 0;
    if (ghz == 0) { int khz = 0; hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0); ghz = khz * 1e-6; printf("shader clock %.3f GHz\n", ghz); }
    const int REP = 2048;
    printf("%-24s", name);
    for (int waves : {1, 2, 4}) {
        unsigned long long h = 0;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k_rate<KIND>, dim3(256), dim3(256 * waves), 65536, 0, d_out, d_in, stride, 8);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_rate<KIND>, dim3(256), dim3(256 * waves), 65536, 0, d_out, d_in, stride, REP);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(&h, d_out, 8, hipMemcpyDeviceToHost);
        double clk_per_instr_simd = ms * 1e-3 * ghz * 1e9 / (REP * 32.0 * waves);
        printf("  w%d: %6.2f clk/instr/SIMD (%5.2f/CU) tick/clk=%.3f", waves, clk_per_instr_simd, clk_per_instr_simd / 4, (double)h / (ms * 1e-3 * ghz * 1e9));
    }
    printf("\n");
}

int main()
{
    unsigned long long *d_out; float *d_in;
    hipMalloc(&d_out, 64); hipMalloc(&d_in, 1 << 20); hipMemset(d_in, 0, 1 << 20);
    run<0>("v_fma_f64", d_out, d_in, 0);
    run<16>("v_fma_f64 (sgpr src)", d_out, d_in, 0);
    run<1>("v_mul_f64", d_out, d_in, 0);
    run<9>("v_add_f64", d_out, d_in, 0);
    run<2>("v_cvt_f64_f32", d_out, d_in, 0);
    run<3>("v_cvt_f32_f64", d_out, d_in, 0);
    run<4>("v_cndmask_b32", d_out, d_in, 0);
    run<5>("v_lshl_add_u64", d_out, d_in, 0);
    run<17>("v_lshl_add_u32", d_out, d_in, 0);
    run<6>("v_readlane_b32", d_out, d_in, 0);
    run<7>("v_add_f32", d_out, d_in, 0);
    run<8>("v_fma_f32", d_out, d_in, 0);
    run<18>("v_pk_fma_f32", d_out, d_in, 0);
    run<19>("v_mov_b64", d_out, d_in, 0);
    run<10>("ds_read_b64", d_out, d_in, 0);
    run<11>("ds_read2st64_b64", d_out, d_in, 0);
    run<12>("ds_read_b32", d_out, d_in, 0);
    run<13>("ds_read2_b32", d_out, d_in, 0);
    run<14>("ds_write_b64", d_out, d_in, 0);
    run<15>("ds_read_b128", d_out, d_in, 0);
    run<20>("ds_read2_b64 o1:32", d_out, d_in, 0);
    run<21>("ds_read2_b64 o1:33", d_out, d_in, 0);
    run<22>("ds_read2_b64 o1:1", d_out, d_in, 0);
    run<23>("ds_read_b32 stride4", d_out, d_in, 0);
    run<24>("ds_read_b32 xpass", d_out, d_in, 0);
    run<25>("ds_read2_b32 xpass", d_out, d_in, 0);
    run<26>("ds_write_b32 stride4", d_out, d_in, 0);
    run<27>("ds_write_b128", d_out, d_in, 0);
    return 0;
}
