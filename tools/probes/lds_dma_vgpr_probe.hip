// Probe: does LDS-DMA (global_load_lds_dwordx4 from inline assembly) disturb VECTOR REGISTERS of the issuing wave or of the wave that
// shares its SIMD when two workgroups are resident per CU?  Every thread holds 96 sentinel registers (kept live through an opaque asm
// barrier) while its wave streams DMA pieces into LDS for many rounds, then checks the sentinels.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/lds_dma_vgpr_probe.hip -o /tmp/probe3 && /tmp/probe3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char lchar_t;
constexpr int UNITS = 5040;
constexpr int NREG = 96;
__device__ __forceinline__ void dma16(const void* base, unsigned voff, unsigned ldsaddr)
{
    ldsaddr = __builtin_amdgcn_readfirstlane(ldsaddr);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(ldsaddr), "v"(voff), "s"(base) : "memory");
}
template <int MODE>
__global__ __launch_bounds__(256, 2) void k(const u32x4* src, unsigned* bad, unsigned* first, int rounds)
{
    extern __shared__ u32x4 lds[];
    const unsigned ldsBase = (unsigned)(uintptr_t)(lchar_t*)lds;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned r[NREG];
#pragma unroll
    for (int i = 0; i < NREG; ++i) r[i] = 0x5A000000u + (unsigned)i * 0x10000u + (unsigned)tid;
#pragma unroll
    for (int i = 0; i < NREG; ++i) asm volatile("" : "+v"(r[i]));          // materialised in registers
    for (int it = 0; it < rounds; ++it) {
        if (MODE == 0) {
            for (int pc = wave; pc < 78; pc += 4)
                dma16(src, (unsigned)(((blockIdx.x * 131 + it * 17) % 1000) * UNITS + pc * 64 + lane) * 16u, ldsBase + (unsigned)pc * 1024u);
        } else {                                                             // the same traffic as ordinary loads + ds_write (control)
            for (int pc = wave; pc < 78; pc += 4) lds[pc * 64 + lane] = src[((blockIdx.x * 131 + it * 17) % 1000) * UNITS + pc * 64 + lane];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NREG; ++i) asm volatile("" : "+v"(r[i]));
    }
    unsigned nb = 0;
#pragma unroll
    for (int i = 0; i < NREG; ++i) {
        const unsigned want = 0x5A000000u + (unsigned)i * 0x10000u + (unsigned)tid;
        if (r[i] != want) { ++nb; if (atomicAdd(bad, 1u) == 0) { first[0] = blockIdx.x; first[1] = tid; first[2] = i; first[3] = r[i]; first[4] = want; } }
    }
    if (nb == 12345u) bad[1] = lds[tid].x;
}
int main()
{
    const size_t n = (size_t)1000 * UNITS;
    std::vector<unsigned> h(n * 4);
    for (size_t i = 0; i < n; ++i) for (int e = 0; e < 4; ++e) h[i * 4 + e] = 0xC0000000u + (unsigned)i + e * 0x1000000u;
    u32x4* src; unsigned *bad, *first;
    (void)hipMalloc(&src, n * 16); (void)hipMalloc(&bad, 8); (void)hipMalloc(&first, 32);
    (void)hipMemcpy(src, h.data(), n * 16, hipMemcpyHostToDevice);
    (void)hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, UNITS * 16 + 20480);
    (void)hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, UNITS * 16 + 20480);
    for (int mode = 0; mode < 2; ++mode)
        for (int extra : {0, 20480})
            for (int grid : {512, 2048}) {
                (void)hipMemset(bad, 0, 8); (void)hipMemset(first, 0, 32);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), UNITS * 16 + extra, 0, src, bad, first, 50);
                else hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), UNITS * 16 + extra, 0, src, bad, first, 50);
                unsigned b, f[8];
                (void)hipMemcpy(&b, bad, 4, hipMemcpyDeviceToHost); (void)hipMemcpy(f, first, 32, hipMemcpyDeviceToHost);
                printf("%s, lds %d B (%s per CU), grid %4d: corrupted sentinel registers %u", mode == 0 ? "LDS-DMA" : "load + ds_write", UNITS * 16 + extra,
                       extra ? "one workgroup" : "two workgroups", grid, b);
                if (b) printf("  first: block %u thread %u (lane %u) register #%u got %#x want %#x", f[0], f[1], f[1] & 63, f[2], f[3], f[4]);
                printf("\n");
            }
    return 0;
}
