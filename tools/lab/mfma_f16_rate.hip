// Micro-benchmark: cycles per v_mfma_f32_32x32x16_f16 on one SIMD (one wave per SIMD, 256 workgroups of 4 waves), as a function of the
// number of INDEPENDENT accumulators the MFMAs rotate over (1: one dependent chain; 2, 4, 8), with constant operands in registers.
// s_memtime ticks / MFMAs.   hipcc -O3 --offload-arch=gfx950 tools/lab/mfma_f16_rate.hip -o tools/bin/mfma_f16_rate (git-ignored; it travels with gpurun) && ./tools/bin/mfma_f16_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int NACC>
__global__ __launch_bounds__(256, 1) void k(float* out, const float* rnd, int iters, unsigned long long* clk)
{
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    const int lane = threadIdx.x & 63;
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)rnd[lane * 8 + e]; b[e] = (_Float16)rnd[512 + lane * 8 + e]; }
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8 / NACC; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 7) clk[0] = t1 - t0;
}

// 54 MFMAs (two chains, three dependent MFMAs at a time, as a k-step of sr_conv_block2.h's first stage) between workgroup barriers
__global__ __launch_bounds__(256, 1) void kb(float* out, const float* rnd, int iters, unsigned long long* clk, int barrier)
{
    f32x16 za, zb;
    for (int j = 0; j < 16; ++j) { za[j] = 0.f; zb[j] = 0.f; }
    const int lane = threadIdx.x & 63;
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)rnd[lane * 8 + e]; b[e] = (_Float16)rnd[512 + lane * 8 + e]; }
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            za = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, za, 0, 0, 0); za = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, za, 0, 0, 0); za = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, za, 0, 0, 0);
            zb = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, zb, 0, 0, 0); zb = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, zb, 0, 0, 0); zb = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, zb, 0, 0, 0);
        }
        if (barrier) __syncthreads();
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int j = 0; j < 16; ++j) s += za[j] + zb[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 7) clk[0] = t1 - t0;
}

template <int NACC>
void run(float* out, float* rnd, unsigned long long* clk, const char* what)
{
    const int iters = 2000;
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(256), 0, 0, out, rnd, iters, clk);
    hipDeviceSynchronize();
    unsigned long long h = 0;
    hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
    printf("%-28s %6.1f cycles per MFMA\n", what, (double)h / (iters * 8.0));
}

int main()
{
    float *out, *rnd; unsigned long long* clk;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&rnd, 4096 * 4); hipMalloc(&clk, 16);
    float h[4096];
    for (int z = 0; z < 2; ++z) {
        for (int i = 0; i < 4096; ++i) h[i] = z ? 0.0f : (float)((i * 2654435761u >> 8) & 1023) / 512.0f - 1.0f;
        hipMemcpy(rnd, h, sizeof(h), hipMemcpyHostToDevice);
        printf(z ? "zero operands:\n" : "random operands:\n");
        run<1>(out, rnd, clk, "1 accumulator (one chain)");
        run<2>(out, rnd, clk, "2 accumulators");
        run<4>(out, rnd, clk, "4 accumulators");
        run<8>(out, rnd, clk, "8 accumulators");
        for (int bar = 0; bar < 2; ++bar)
            for (int iters = 4; iters <= 400; iters *= 10) {
                for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(kb, dim3(256), dim3(256), 0, 0, out, rnd, iters, clk, bar);
                hipDeviceSynchronize();
                unsigned long long hh = 0;
                hipMemcpy(&hh, clk, 8, hipMemcpyDeviceToHost);
                printf("54 MFMAs x %3d k-steps, %s  %6.1f cycles per MFMA\n", iters, bar ? "barrier per k-step:" : "no barrier:        ", (double)hh / (iters * 54.0));
            }
    }
    return 0;
}
