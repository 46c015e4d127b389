// Probe: `global_load_lds_dwordx4` issued from inline assembly (M0 = LDS address of the wave's destination) with TWO workgroups per CU.
// Every workgroup fills its LDS with a marker, DMAs a pattern that encodes (workgroup, destination unit) into units spread over its whole
// 80 KB allocation, and reads them back with ds_read.  Question: does the DMA land in the issuing workgroup's own allocation at every
// offset (also above 64 KB), whatever the allocation's base is?
//   hipcc -O3 --offload-arch=gfx950 tools/probes/lds_dma_two_wg_probe.hip -o /tmp/probe2 && /tmp/probe2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char lchar_t;
constexpr int UNITS = 5040;                    // 80 640 bytes
__device__ __forceinline__ void dma16(const void* base, unsigned voff, unsigned ldsaddr)
{
    ldsaddr = __builtin_amdgcn_readfirstlane(ldsaddr);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(ldsaddr), "v"(voff), "s"(base) : "memory");
}
__global__ __launch_bounds__(256, 2) void k(const u32x4* src, unsigned* bad, unsigned* first, int rounds)
{
    extern __shared__ u32x4 lds[];
    const unsigned ldsBase = (unsigned)(uintptr_t)(lchar_t*)lds;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned nbad = 0;
    for (int it = 0; it < rounds; ++it) {
        for (int i = tid; i < UNITS; i += 256) { u32x4 f = {0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu}; lds[i] = f; }
        __syncthreads();
        // pieces of 64 units: piece pc -> units 64 pc .. ; waves take pieces round robin; 78 pieces cover 4992 units
        for (int pc = wave; pc < 78; pc += 4)
            dma16(src, (unsigned)(((blockIdx.x * 131 + it * 17) % 1000) * UNITS + pc * 64 + lane) * 16u, ldsBase + (unsigned)pc * 1024u);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int i = tid; i < 78 * 64; i += 256) {
            const u32x4 v = lds[i];
            const unsigned want = (unsigned)(((blockIdx.x * 131 + it * 17) % 1000) * UNITS + i);
            if (v.x != want || v.w != want + 3u * 0x10000000u) { ++nbad; if (atomicAdd(bad, 1u) == 0) { first[0] = blockIdx.x; first[1] = i; first[2] = v.x; first[3] = want; first[4] = ldsBase; } }
        }
        __syncthreads();
    }
}
int main()
{
    const size_t n = (size_t)1000 * UNITS;
    std::vector<unsigned> h(n * 4);
    for (size_t i = 0; i < n; ++i) for (int e = 0; e < 4; ++e) h[i * 4 + e] = (unsigned)i + e * 0x10000000u;
    u32x4* src; unsigned *bad, *first;
    hipMalloc(&src, n * 16); hipMalloc(&bad, 4); hipMalloc(&first, 32);
    hipMemcpy(src, h.data(), n * 16, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, UNITS * 16 + 20480);
    for (int extra : {0, 20480}) {
        for (int grid : {64, 512, 2048}) {
            hipMemset(bad, 0, 4); hipMemset(first, 0, 32);
            hipLaunchKernelGGL(k, dim3(grid), dim3(256), UNITS * 16 + extra, 0, src, bad, first, 20);
            unsigned b, f[8];
            hipMemcpy(&b, bad, 4, hipMemcpyDeviceToHost); hipMemcpy(f, first, 32, hipMemcpyDeviceToHost);
            printf("lds %d B (%s per CU), grid %4d: wrong units %u", UNITS * 16 + extra, extra ? "one workgroup" : "two workgroups", grid, b);
            if (b) printf("  first: block %u unit %u got %#x want %#x ldsBase %u", f[0], f[1], f[2], f[3], f[4]);
            printf("\n");
        }
    }
    return 0;
}
