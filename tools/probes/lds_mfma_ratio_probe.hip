// Probe: how many ds_read_b128 fit beside v_mfma_f32_32x32x16_f16 on this chip -- is the LDS array 128 or
// 256 B/clk/CU (MI355X_MICROARCH.md, LDS table: 256), and at what reads-per-MFMA ratio does the MFMA *gap* (not a wait) grow?
//
// One workgroup = 256 threads = one wave per SIMD; WPS workgroups per CU (occupancy set by the dynamic LDS size) give WPS waves per
// SIMD.  Per loop iteration a wave issues 12 MFMAs (4 accumulators x 3 dependent products, as the split kernels do) and R
// conflict-free ds_read_b128 (lane l reads 16 bytes at base + 16 l: 1 KiB contiguous per wave instruction).
//   MODE 0 "prefetch": the reads of iteration i fill the fragment set iteration i + 1 multiplies with -- no wait is exposed, only the
//                      issue cost / the LDS array's occupancy can lengthen the gap;
//   MODE 1 "just in time": each MFMA group multiplies with the fragments read immediately in front of it (a wait per group: what the
//                      default upsampling kernel's single-buffered weight fragments do);
//   MODE 2: reads only (no MFMAs): the array's rate by itself.
// Output: cycles per 12 MFMAs of wave 0 (s_memtime, median over workgroups), wall time, and the LDS bytes per clock per CU the wall
// implies.  hipcc -O3 --offload-arch=gfx950 tools/probes/lds_mfma_ratio_probe.hip -o tools/bin/lds_mfma_ratio_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// Every instruction of the timed loop is its own `asm volatile` (these keep their program order): the loop is exactly what is written here.
#define DS_READ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define MFMA(acc, a, b) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// one half-iteration of MODE 0: 12 MFMAs on `use`, R reads into `fill`, the reads spread over the first 8 MFMA gaps (the last reads have
// four MFMAs = 128 cycles to land before the closing wait)
template <int R, int H>
__device__ __forceinline__ void half_prefetch(f32x16 (&acc)[4], f16x8* use, f16x8* fill, unsigned addr)
{
    constexpr int NF = R > 0 ? R : 1;
#pragma unroll
    for (int m = 0; m < 12; ++m) {
        MFMA(acc[m & 3], use[(2 * m) % NF], use[(2 * m + 1) % NF]);
        if (m < 8) {
#pragma unroll
            for (int q = 0; q < 24; ++q)
                if (q < R && q >= (m * R) / 8 && q < ((m + 1) * R) / 8) {
                    switch ((q + 3 * H) & 7) {
                    case 0: DS_READ(fill[q], addr, 0); break;    case 1: DS_READ(fill[q], addr, 1024); break;
                    case 2: DS_READ(fill[q], addr, 2048); break; case 3: DS_READ(fill[q], addr, 3072); break;
                    case 4: DS_READ(fill[q], addr, 4096); break; case 5: DS_READ(fill[q], addr, 5120); break;
                    case 6: DS_READ(fill[q], addr, 6144); break; default: DS_READ(fill[q], addr, 7168); break;
                    }
                }
        }
    }
    LGKM0();
}

template <int MODE, int R>
__global__ __launch_bounds__(256, 1) void probe(float* out, const unsigned* rnd, int iters, unsigned long long* clk)
{
    extern __shared__ u32x4 lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // 32 KiB of random fp16 bit patterns with small exponents (finite, |v| < 2)
    for (int i = tid; i < 2048; i += 256) {
        u32x4 v;
        for (int e = 0; e < 4; ++e) v[e] = (rnd[(i * 4 + e) & 4095] & 0x3bff3bffu);
        lds[i] = v;
    }
    __syncthreads();
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    constexpr int NF = R > 0 ? R : 1;
    f16x8 cur[NF], nxt[NF];
    // each wave walks its own 8 KiB window, one KiB per read: conflict-free
    const unsigned addr = (unsigned)(wave * 512 + lane) * 16u;
    const u32x4* base = lds + wave * 512 + lane;
#pragma unroll
    for (int q = 0; q < NF; ++q) { cur[q] = __builtin_bit_cast(f16x8, base[(q & 7) * 64]); nxt[q] = cur[q]; }
#pragma unroll
    for (int q = 0; q < NF; ++q) asm volatile("" : "+v"(cur[q]), "+v"(nxt[q]));
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(acc[i]));
    LGKM0();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
            half_prefetch<R, 0>(acc, cur, nxt, addr);
            half_prefetch<R, 1>(acc, nxt, cur, addr);
        } else if (MODE == 1) {
            // groups of 3 dependent MFMAs, each group behind its own R / 4 reads and a wait
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                constexpr int RG = R / 4 > 0 ? R / 4 : 1;
#pragma unroll
                for (int q = 0; q < RG; ++q) {
                    if (q & 1) DS_READ(cur[q], addr, 1024 * ((q >> 1) & 3) + 4096); else DS_READ(cur[q], addr, 1024 * ((q >> 1) & 3));
                }
                LGKM0();
#pragma unroll
                for (int m = 0; m < 3; ++m) MFMA(acc[g], cur[(2 * m) % RG], cur[(2 * m + 1) % RG]);
            }
        } else {
#pragma unroll
            for (int q = 0; q < R; ++q) {
                if (q & 1) DS_READ(nxt[q], addr, 1024 * ((q >> 1) & 3) + 4096); else DS_READ(nxt[q], addr, 1024 * ((q >> 1) & 3));
            }
            LGKM0();
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    for (int q = 0; q < NF; ++q) s += (float)nxt[q][0] + (float)cur[q][1];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) clk[blockIdx.x] = t1 - t0;
}

template <int MODE, int R>
static void run(const char* what, int wps, float* out, const unsigned* rnd, unsigned long long* clk, int iters)
{
    const int lds = wps == 1 ? 160 * 1024 : wps == 2 ? 80 * 1024 : 53 * 1024;
    (void)hipFuncSetAttribute((const void*)probe<MODE, R>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int grid = 256 * wps;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(clk, 0, grid * 8);
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<MODE, R>), dim3(grid), dim3(256), lds, 0, out, rnd, iters, clk);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(grid);
    hipMemcpy(h.data(), clk, grid * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double cyc = (double)h[grid / 2] / iters / (MODE == 0 ? 2 : 1);
    // LDS bytes per clock per CU: 4 waves x wps x R KiB per iteration, per the median wave's cycles
    const double bpc = MODE == 1 ? 4.0 * wps * (R / 4 * 4) * 1024.0 / cyc : 4.0 * wps * R * 1024.0 / cyc;
    printf("%-14s R=%2d (%.2f per MFMA)  %d wave(s)/SIMD  %8.1f cycles per iteration (12 MFMAs) = %5.1f per MFMA   wall %8.1f us   LDS %6.1f B/clk/CU   clock %.2f GHz\n",
           what, R, R / 12.0, wps, cyc, cyc / 12.0, ms * 1000.0, bpc, cyc * iters * (MODE == 0 ? 2 : 1) / (ms * 1e6));
    hipEventDestroy(e0); hipEventDestroy(e1);
}

int main()
{
    float* out; unsigned* rnd; unsigned long long* clk;
    hipMalloc(&out, 256 * 3 * 256 * 4);
    hipMalloc(&rnd, 4096 * 4);
    hipMalloc(&clk, 256 * 3 * 8);
    std::vector<unsigned> h(4096);
    unsigned s = 12345u;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = s; }
    hipMemcpy(rnd, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    const int iters = 20000;
    for (int wps = 1; wps <= 3; ++wps) {
        run<0, 0>("mfma alone", wps, out, rnd, clk, iters);
        run<0, 6>("prefetch", wps, out, rnd, clk, iters);
        run<0, 8>("prefetch", wps, out, rnd, clk, iters);
        run<0, 12>("prefetch", wps, out, rnd, clk, iters);
        run<0, 24>("prefetch", wps, out, rnd, clk, iters);
        run<1, 8>("just in time", wps, out, rnd, clk, iters);
        run<1, 12>("just in time", wps, out, rnd, clk, iters);
        run<1, 24>("just in time", wps, out, rnd, clk, iters);
        run<2, 8>("reads alone", wps, out, rnd, clk, iters);
        run<2, 24>("reads alone", wps, out, rnd, clk, iters);
    }
    return 0;
}
