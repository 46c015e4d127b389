// Probe: does an LDS return (ds_read_b128) into the DATA registers of a 16-byte buffer store issued directly in front
// of it corrupt the stored data on gfx950?  Round 5 found the first dword of a stored unit replaced by LDS bits in 8 lanes of the
// phase-decomposed upsampling kernel (opt-in) at two workgroups per CU and attributed it to exactly this sequence
//     v_permlane32_swap ... ; buffer_store_dwordx4 v[10:13], ... ; ds_read_b128 v[10:13], ...
// This probe runs that sequence in isolation, every instruction spelled out in one asm block (nothing for the compiler to reorder):
//   variant 0: data made by v_mov_b32,          store, ds_read into the same registers at once
//   variant 1: data made by v_permlane32_swap,  store, ds_read into the same registers at once     <- the round-5 sequence
//   variant 2: as 1 with `s_nop 7` between the store and the ds_read                                <- the round-5 fix's equivalent
//   variant 3: as 1, the ds_read into OTHER registers (control: must be clean)
// with 1, 2 or 3 workgroups per CU (dynamic LDS sets the occupancy) and the memory pipeline kept busy (no vmcnt wait inside the loop:
// up to 64 stores in flight per wave).  Stored words are < 2^30; the LDS holds words >= 0xC0000000: a stored word with its top bits set
// is an LDS return that overtook the store's data read.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/store_lds_return_probe.hip -o tools/bin/store_lds_return_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

template <int VARIANT>
__global__ __launch_bounds__(256, 3) void probe(unsigned* out, int iters, unsigned bytes)
{
    extern __shared__ u32x4 lds[];
    const int tid = threadIdx.x;
    for (int i = tid; i < 1024; i += 256) lds[i] = u32x4{0xC0000000u + i, 0xD0000000u + i, 0xE0000000u + i, 0xF0000000u + i};
    __syncthreads();
    const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)bytes, 0x00020000);
    const unsigned gid = blockIdx.x * 256 + tid;
    const unsigned laddr = (unsigned)(tid * 16);
    unsigned keep = 0;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        const unsigned voff = ((unsigned)it * gridDim.x * 256u + gid) * 16u;
        const unsigned a = (gid * 4u + 0u + (unsigned)it * 977u) & 0x3fffffffu, b = (gid * 4u + 1u + (unsigned)it * 977u) & 0x3fffffffu;
        const unsigned c = (gid * 4u + 2u + (unsigned)it * 977u) & 0x3fffffffu, d = (gid * 4u + 3u + (unsigned)it * 977u) & 0x3fffffffu;
        unsigned r;
        if (VARIANT == 0)
            asm volatile("v_mov_b32 v10, %1\n\tv_mov_b32 v11, %2\n\tv_mov_b32 v12, %3\n\tv_mov_b32 v13, %4\n\t"
                         "s_nop 1\n\t"
                         "buffer_store_dwordx4 v[10:13], %5, %6, 0 offen\n\t"
                         "ds_read_b128 v[10:13], %7\n\t"
                         "s_waitcnt lgkmcnt(0)\n\tv_mov_b32 %0, v10"
                         : "=v"(r) : "v"(a), "v"(b), "v"(c), "v"(d), "v"(voff), "s"(rs), "v"(laddr) : "v10", "v11", "v12", "v13", "memory");
        else if (VARIANT == 1)
            asm volatile("v_mov_b32 v10, %1\n\tv_mov_b32 v11, %2\n\tv_mov_b32 v12, %3\n\tv_mov_b32 v13, %4\n\t"
                         "s_nop 1\n\t"
                         "v_permlane32_swap_b32_e32 v10, v11\n\tv_permlane32_swap_b32_e32 v12, v13\n\t"
                         "buffer_store_dwordx4 v[10:13], %5, %6, 0 offen\n\t"
                         "ds_read_b128 v[10:13], %7\n\t"
                         "s_waitcnt lgkmcnt(0)\n\tv_mov_b32 %0, v10"
                         : "=v"(r) : "v"(a), "v"(b), "v"(c), "v"(d), "v"(voff), "s"(rs), "v"(laddr) : "v10", "v11", "v12", "v13", "memory");
        else if (VARIANT == 2)
            asm volatile("v_mov_b32 v10, %1\n\tv_mov_b32 v11, %2\n\tv_mov_b32 v12, %3\n\tv_mov_b32 v13, %4\n\t"
                         "s_nop 1\n\t"
                         "v_permlane32_swap_b32_e32 v10, v11\n\tv_permlane32_swap_b32_e32 v12, v13\n\t"
                         "buffer_store_dwordx4 v[10:13], %5, %6, 0 offen\n\t"
                         "s_nop 7\n\t"
                         "ds_read_b128 v[10:13], %7\n\t"
                         "s_waitcnt lgkmcnt(0)\n\tv_mov_b32 %0, v10"
                         : "=v"(r) : "v"(a), "v"(b), "v"(c), "v"(d), "v"(voff), "s"(rs), "v"(laddr) : "v10", "v11", "v12", "v13", "memory");
        else
            asm volatile("v_mov_b32 v10, %1\n\tv_mov_b32 v11, %2\n\tv_mov_b32 v12, %3\n\tv_mov_b32 v13, %4\n\t"
                         "s_nop 1\n\t"
                         "v_permlane32_swap_b32_e32 v10, v11\n\tv_permlane32_swap_b32_e32 v12, v13\n\t"
                         "buffer_store_dwordx4 v[10:13], %5, %6, 0 offen\n\t"
                         "ds_read_b128 v[14:17], %7\n\t"
                         "s_waitcnt lgkmcnt(0)\n\tv_mov_b32 %0, v14"
                         : "=v"(r) : "v"(a), "v"(b), "v"(c), "v"(d), "v"(voff), "s"(rs), "v"(laddr) : "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "memory");
        keep ^= r;
    }
    if (keep == 0x12345u) out[0] = keep;                                     // (keeps the LDS returns alive)
}

template <int VARIANT>
static long run(int wgPerCu, unsigned* out, size_t words, int iters)
{
    const int lds = wgPerCu == 1 ? 160 * 1024 : wgPerCu == 2 ? 80 * 1024 : 53 * 1024;
    (void)hipFuncSetAttribute((const void*)probe<VARIANT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int grid = 256 * wgPerCu * 4;                                       // four rounds of resident workgroups
    hipMemset(out, 0, words * 4);
    hipLaunchKernelGGL((probe<VARIANT>), dim3(grid), dim3(256), lds, 0, out, iters, (unsigned)(words * 4));
    hipDeviceSynchronize();
    std::vector<unsigned> h((size_t)grid * 256 * 4 * iters);
    hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
    long bad = 0, wrong = 0;
    const size_t threads = (size_t)grid * 256;
    auto val = [](size_t g, int e, int it) { return (unsigned)((g * 4u + (unsigned)e + (unsigned)it * 977u) & 0x3fffffffu); };
    for (int it = 0; it < iters; ++it)
        for (size_t g = 0; g < threads; ++g) {
            const unsigned* w = &h[((size_t)it * threads + g) * 4];
            const bool lo = (g & 63) < 32;
            unsigned e[4];
            if (VARIANT == 0) { for (int k = 0; k < 4; ++k) e[k] = val(g, k, it); }
            else {   // v_permlane32_swap a, b: a's upper 32 lanes trade places with b's lower 32
                e[0] = lo ? val(g, 0, it) : val(g - 32, 1, it); e[1] = lo ? val(g + 32, 0, it) : val(g, 1, it);
                e[2] = lo ? val(g, 2, it) : val(g - 32, 3, it); e[3] = lo ? val(g + 32, 2, it) : val(g, 3, it);
            }
            for (int k = 0; k < 4; ++k) { bad += w[k] >= 0xC0000000u; wrong += w[k] != e[k]; }
        }
    printf("variant %d, %d workgroup(s) per CU: %zu words stored, %ld carry LDS bits, %ld differ from the expected value\n", VARIANT, wgPerCu, h.size(), bad, wrong);
    return bad + wrong;
}

int main()
{
    const int iters = 64;
    const size_t words = (size_t)256 * 3 * 4 * 256 * 4 * iters;
    unsigned* out; hipMalloc(&out, words * 4);
    long bad[4] = {0, 0, 0, 0};
    for (int rep = 0; rep < 3; ++rep)
        for (int w = 1; w <= 3; ++w) {
            bad[0] += run<0>(w, out, words, iters);
            bad[1] += run<1>(w, out, words, iters);
            bad[2] += run<2>(w, out, words, iters);
            bad[3] += run<3>(w, out, words, iters);
        }
    printf("TOTAL words carrying LDS bits or otherwise wrong: plain %ld, permlane32_swap %ld, permlane32_swap + s_nop %ld, other registers (control) %ld\n", bad[0], bad[1], bad[2], bad[3]);
    return 0;
}
