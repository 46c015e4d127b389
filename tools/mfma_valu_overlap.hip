// Micro-benchmark: do vector instructions hide under v_mfma_f32_32x32x16_f16 -- (a) inside ONE wave's stream when they are interleaved
// NV per MFMA (sched_group_barrier), (b) inside one wave's stream when they come in a clump after each group of 12 MFMAs, (c) when they
// are issued by ANOTHER wave of the same SIMD (role split: waves 0-3 of a 512-thread workgroup multiply, waves 4-7 do the vector work),
// (d) with two waves per SIMD that each do both.  This is the question behind every split-operand convolution kernel of this package
// (profiles/r03_pmc_ups.md: "MFMA busy cycles and vector issue cycles add up to the launch").
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_valu_overlap.hip -o tools/bin/mfma_valu_overlap && ./tools/bin/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define MFMA(acc) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0)

// MODE 0: NV vector instructions after every MFMA (interleaved); 1: 12 MFMAs, then 12 NV vector instructions (clumped);
// 2: role split -- waves < 4 only multiply, waves >= 4 only do the vector work; 3: MFMAs only; 4: vector work only
template <int MODE, int NV>
__global__ __launch_bounds__(512, 2) void k(float* out, const float* rnd, int iters, unsigned long long* clk)
{
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)rnd[lane * 8 + e]; b[e] = (_Float16)rnd[512 + lane * 8 + e]; }
    float x[8];
    for (int e = 0; e < 8; ++e) x[e] = rnd[1024 + lane + e];
    const float c0 = rnd[2000], c1 = rnd[2001];
    const bool mm = MODE == 3 || (MODE == 2 ? wave < 4 : MODE != 4);
    const bool vv = MODE == 4 || (MODE == 2 ? wave >= 4 : MODE != 3);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int m = 0; m < 12; ++m) {
                MFMA(acc[m & 3]);
#pragma unroll
                for (int v = 0; v < NV; ++v) x[(m * NV + v) & 7] = __builtin_fmaf(x[(m * NV + v) & 7], c0, c1);
            }
#pragma unroll
            for (int m = 0; m < 12; ++m) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (NV > 0) __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
            }
        } else {
            if (mm) {
#pragma unroll
                for (int m = 0; m < 12; ++m) MFMA(acc[m & 3]);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (vv) {
#pragma unroll
                for (int v = 0; v < 12 * NV; ++v) x[v & 7] = __builtin_fmaf(x[v & 7], c0, c1);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    for (int e = 0; e < 8; ++e) s += x[e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 7) clk[0] = t1 - t0;
    if (threadIdx.x == 256 && blockIdx.x == 7) clk[1] = t1 - t0;
}

template <int MODE, int NV>
void run(float* out, float* rnd, unsigned long long* clk, int threads, int blocks, const char* what)
{
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<MODE, NV>), dim3(blocks), dim3(threads), 0, 0, out, rnd, iters, clk);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<MODE, NV>), dim3(blocks), dim3(threads), 0, 0, out, rnd, iters, clk);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2] = {0, 0};
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    printf("%-62s NV=%d  %7.1f cycles per 12 MFMAs (wave 0)  %7.1f (wave 4)   %8.1f us\n", what, NV, (double)h[0] / iters, threads > 256 ? (double)h[1] / iters : 0.0, ms * 1e3);
}

template <int NV>
void sweep(float* out, float* rnd, unsigned long long* clk)
{
    run<0, NV>(out, rnd, clk, 256, 256, "1 wave/SIMD, vector work interleaved NV per MFMA");
    run<1, NV>(out, rnd, clk, 256, 256, "1 wave/SIMD, vector work in a clump after 12 MFMAs");
    run<0, NV>(out, rnd, clk, 512, 256, "2 waves/SIMD, each interleaved");
    run<1, NV>(out, rnd, clk, 512, 256, "2 waves/SIMD, each clumped");
    run<2, NV>(out, rnd, clk, 512, 256, "2 waves/SIMD, role split (one multiplies, one does the vector work)");
    run<0, NV>(out, rnd, clk, 512, 512, "4 waves/SIMD (2 workgroups), each interleaved");
    run<1, NV>(out, rnd, clk, 512, 512, "4 waves/SIMD (2 workgroups), each clumped");
    run<2, NV>(out, rnd, clk, 512, 512, "4 waves/SIMD (2 workgroups), role split");
    run<4, NV>(out, rnd, clk, 256, 256, "1 wave/SIMD, the vector work alone");
}

int main()
{
    float *out, *rnd; unsigned long long* clk;
    hipMalloc(&out, 512 * 512 * 4); hipMalloc(&rnd, 4096 * 4); hipMalloc(&clk, 16);
    float h[4096];
    for (int i = 0; i < 4096; ++i) h[i] = (float)((i * 2654435761u >> 8) & 1023) / 512.0f - 1.0f;
    hipMemcpy(rnd, h, sizeof(h), hipMemcpyHostToDevice);
    run<3, 0>(out, rnd, clk, 256, 256, "1 wave/SIMD, MFMAs alone");
    run<3, 0>(out, rnd, clk, 512, 256, "2 waves/SIMD, MFMAs alone");
    sweep<2>(out, rnd, clk);
    sweep<4>(out, rnd, clk);
    sweep<6>(out, rnd, clk);
    sweep<8>(out, rnd, clk);
    return 0;
}
