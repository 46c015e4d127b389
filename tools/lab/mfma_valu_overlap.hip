// Micro-benchmark: do vector instructions hide under v_mfma_f32_32x32x16_f16 -- (a) inside ONE wave's stream when they are interleaved
// NV per MFMA (sched_group_barrier), (b) inside one wave's stream when they come in a clump after each group of 12 MFMAs, (c) when they
// are issued by ANOTHER wave of the same SIMD (role split: waves 0-3 of a 512-thread workgroup multiply, waves 4-7 do the vector work),
// (d) with two waves per SIMD that each do both.  This is the question behind every split-operand convolution kernel of this package
// (profiles/r03_pmc_ups.md: "MFMA busy cycles and vector issue cycles add up to the launch").
//   hipcc -O3 --offload-arch=gfx950 tools/lab/mfma_valu_overlap.hip -o tools/bin/mfma_valu_overlap && ./tools/bin/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MFMA(acc) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0)

// MODE 0: NV vector instructions after every MFMA (interleaved); 1: 12 MFMAs, then 12 NV vector instructions (clumped);
// 2: role split -- waves < 4 only multiply, waves >= 4 only do the vector work; 3: MFMAs only; 4: vector work only;
// 5: role split with the vector waves at s_setprio 3; 6: clumped with s_setprio 3 around the clump;
// 7: MFMAs only, but every MFMA on ANOTHER pair of operand registers (eight random fragments each side, as a convolution's taps
//    are) instead of the same pair every time: what the data's toggling costs (power -> clock), nothing else differs from mode 3;
// 8: as 7 with all sixteen fragments zero;
// 9: as 7 on v_mfma_f32_16x16x32_f16 -- 24 of them per iteration, the same multiply-adds as 12 of the 32x32x16 (MI355X_MICROARCH.md,
//    DVFS give-back (7): the chip holds a higher clock on that shape)
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
    const bool split = MODE == 2 || MODE == 5;
    const bool mm = MODE == 3 || (split ? wave < 4 : MODE != 4);
    const bool vv = MODE == 4 || (split ? wave >= 4 : MODE != 3);
    if (MODE == 5 && vv) __builtin_amdgcn_s_setprio(3);
    f16x8 av[8], bv[8];
    f32x4 acc4[16];
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 4; ++j) acc4[i][j] = 0.f;
    if (MODE == 7 || MODE == 8 || MODE == 9) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                av[q][e] = MODE == 8 ? (_Float16)0.f : (_Float16)rnd[(lane * 8 + e + 97 * q) & 2047];
                bv[q][e] = MODE == 8 ? (_Float16)0.f : (_Float16)rnd[(1024 + lane * 8 + e + 131 * q) & 2047];
            }
            asm volatile("" : "+v"(av[q]), "+v"(bv[q]));                     // in registers before the clock is read
        }
    }
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 9) {
#pragma unroll
            for (int m = 0; m < 24; ++m) acc4[m & 15] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[(m * 5) & 7], bv[(m * 3) & 7], acc4[m & 15], 0, 0, 0);
        } else if (MODE == 7 || MODE == 8) {
#pragma unroll
            for (int m = 0; m < 12; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[(m * 5) & 7], bv[(m * 3) & 7], acc[m & 3], 0, 0, 0);
        } else if (MODE == 0) {
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
                if (MODE == 6) __builtin_amdgcn_s_setprio(3);
#pragma unroll
                for (int v = 0; v < 12 * NV; ++v) x[v & 7] = __builtin_fmaf(x[v & 7], c0, c1);
                if (MODE == 6) __builtin_amdgcn_s_setprio(0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    for (int e = 0; e < 8; ++e) s += x[e];
    if (MODE == 9) for (int i = 0; i < 16; ++i) for (int j = 0; j < 4; ++j) s += acc4[i][j];
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
    run<5, NV>(out, rnd, clk, 512, 256, "2 waves/SIMD, role split, vector waves at s_setprio 3");
    run<6, NV>(out, rnd, clk, 512, 256, "2 waves/SIMD, each clumped, s_setprio 3 around the clump");
    run<0, NV>(out, rnd, clk, 512, 512, "4 waves/SIMD (2 workgroups), each interleaved");
    run<1, NV>(out, rnd, clk, 512, 512, "4 waves/SIMD (2 workgroups), each clumped");
    run<2, NV>(out, rnd, clk, 512, 512, "4 waves/SIMD (2 workgroups), role split");
    run<5, NV>(out, rnd, clk, 512, 512, "4 waves/SIMD (2 workgroups), role split, vector waves at s_setprio 3");
    run<6, NV>(out, rnd, clk, 512, 512, "4 waves/SIMD (2 workgroups), each clumped, s_setprio 3 around the clump");
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
    run<7, 0>(out, rnd, clk, 256, 256, "1 wave/SIMD, MFMAs alone, operands rotating over 8 + 8 random fragments");
    run<7, 0>(out, rnd, clk, 512, 256, "2 waves/SIMD, MFMAs alone, operands rotating over 8 + 8 random fragments");
    run<8, 0>(out, rnd, clk, 512, 256, "2 waves/SIMD, MFMAs alone, operands rotating over 8 + 8 ZERO fragments");
    run<9, 0>(out, rnd, clk, 256, 256, "1 wave/SIMD, 24 x 16x16x32 (= 12 x 32x32x16 of work), rotating random fragments");
    run<9, 0>(out, rnd, clk, 512, 256, "2 waves/SIMD, 24 x 16x16x32 (= 12 x 32x32x16 of work), rotating random fragments");
    run<7, 0>(out, rnd, clk, 512, 256, "2 waves/SIMD, MFMAs alone, operands rotating over 8 + 8 random fragments (again)");
    run<9, 0>(out, rnd, clk, 512, 256, "2 waves/SIMD, 24 x 16x16x32, rotating random fragments (again)");
    sweep<2>(out, rnd, clk);
    sweep<4>(out, rnd, clk);
    sweep<6>(out, rnd, clk);
    sweep<8>(out, rnd, clk);
    return 0;
}
