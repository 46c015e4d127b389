// Probe (round 6): may a VALU instruction overwrite the DATA registers of a 16-byte buffer store in the very next issue slot?
// LLVM's hazard recogniser (GCNHazardRecognizer::createsVALUHazard) inserts wait states behind a > 64-bit MUBUF store only when the
// store's soffset is NOT an SGPR; with an SGPR soffset it assumes the data has been read.  The activation-once form of the packed
// epilogue (sr_split_common.h) put `v_pk_fma_f32 v[106:107], ...` directly behind `buffer_store_dwordx4 v[106:109], v110, s[56:59],
// s23 offen` and lanes 12-15 / 28-31 of the second stored dword came out as the NEXT group's fp32 values in five kernels.
// Variants (every instruction spelled out in one asm block):
//   0: store `0 offen`,        next instruction v_mov_b32 into data register 1      (the documented hazard: the compiler would pad)
//   1: store `sN offen` (SGPR), next instruction v_mov_b32 into data register 1
//   2: store `sN offen` (SGPR), next instruction v_pk_mul_f32 into data registers 0..1
//   3: as 2 with `s_nop 1` between (two wait states)
//   4: as 2 with `s_nop 0` between (one wait state)
//   5: as 2, the VALU writes OTHER registers (control)
//   6: store `0 offen`, ONE wait state (s_nop 0), v_mov_b32 into data register 1
//   7: store `0 offen`, TWO wait states (s_nop 1: what the compiler pads on gfx940+), v_mov_b32 into data register 1
//   8: buffer_store_dwordx2 `sN offen`, next instruction v_mov_b32 into data register 1   (<= 64 bits: the compiler never pads)
//   9: buffer_store_dwordx2 `0 offen`,  next instruction v_mov_b32 into data register 1
//  10: buffer_store_dword   `sN offen`, next instruction v_mov_b32 into the data register
//  11: global_store_dwordx4 (saddr off), next instruction v_mov_b32 into data register 1   (the compiler pads these)
//  12: buffer_store_dwordx4 `sN offen`, next instruction v_mov_b32 into data register 3
//  13: buffer_store_dwordx4 `sN offen`, next instruction v_mov_b32 into data register 0
//  14: buffer_store_dwordx4 `0 offen`,  next instruction v_mov_b32 into the ADDRESS (voffset) register
//  15: buffer_store_dwordx4 `sN offen`, next instruction v_mov_b32 into the ADDRESS (voffset) register
// Stored words are < 2^30, the overwriting values have the top bit set.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/store_valu_overwrite_probe.hip -o tools/bin/store_valu_overwrite_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define PRE "v_mov_b32 v10, %1\n\tv_mov_b32 v11, %2\n\tv_mov_b32 v12, %3\n\tv_mov_b32 v13, %4\n\ts_nop 4\n\t"
#define OPS : "=v"(r) : "v"(a), "v"(b), "v"(c), "v"(d), "v"(voff), "s"(rs), "s"(soff), "v"(g0), "v"(gp) : "v10", "v11", "v12", "v13", "v14", "v15", "memory"

template <int VARIANT>
__global__ __launch_bounds__(256) void probe(unsigned* out, int iters, unsigned bytes, unsigned soff_in)
{
    const int tid = threadIdx.x;
    const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)bytes, 0x00020000);
    const unsigned gid = blockIdx.x * 256 + tid;
    const unsigned soff = __builtin_amdgcn_readfirstlane(soff_in);
    unsigned keep = 0;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        const unsigned voff = ((unsigned)it * gridDim.x * 256u + gid) * 16u;
        const unsigned a = (gid * 4u + 0u + (unsigned)it * 977u) & 0x3fffffffu, b = (gid * 4u + 1u + (unsigned)it * 977u) & 0x3fffffffu;
        const unsigned c = (gid * 4u + 2u + (unsigned)it * 977u) & 0x3fffffffu, d = (gid * 4u + 3u + (unsigned)it * 977u) & 0x3fffffffu;
        const float g0 = -1.5f - (float)tid; const f32x2 gp = {-1.5f - (float)tid, -2.5f - (float)tid};   // (negative floats: top bit set; squares of them below: still |x| > 2, top bits 0x4...: >= 2^30)
        unsigned r;
        if (VARIANT == 0)
            asm volatile(PRE "buffer_store_dwordx4 v[10:13], %5, %6, 0 offen\n\t"
                         "v_mov_b32 v11, %8\n\tv_mov_b32 %0, v11" OPS);
        else if (VARIANT == 1)
            asm volatile(PRE "buffer_store_dwordx4 v[10:13], %5, %6, %7 offen\n\t"
                         "v_mov_b32 v11, %8\n\tv_mov_b32 %0, v11" OPS);
        else if (VARIANT == 2)
            asm volatile(PRE "buffer_store_dwordx4 v[10:13], %5, %6, %7 offen\n\t"
                         "v_pk_mul_f32 v[10:11], %9, %9\n\tv_mov_b32 %0, v11" OPS);
        else if (VARIANT == 3)
            asm volatile(PRE "buffer_store_dwordx4 v[10:13], %5, %6, %7 offen\n\ts_nop 1\n\t"
                         "v_pk_mul_f32 v[10:11], %9, %9\n\tv_mov_b32 %0, v11" OPS);
        else if (VARIANT == 4)
            asm volatile(PRE "buffer_store_dwordx4 v[10:13], %5, %6, %7 offen\n\ts_nop 0\n\t"
                         "v_pk_mul_f32 v[10:11], %9, %9\n\tv_mov_b32 %0, v11" OPS);
        else if (VARIANT == 5)
            asm volatile(PRE "buffer_store_dwordx4 v[10:13], %5, %6, %7 offen\n\t"
                         "v_pk_mul_f32 v[14:15], %9, %9\n\tv_mov_b32 %0, v15" OPS);
        else if (VARIANT == 6)
            asm volatile(PRE "buffer_store_dwordx4 v[10:13], %5, %6, 0 offen\n\ts_nop 0\n\t"
                         "v_mov_b32 v11, %8\n\tv_mov_b32 %0, v11" OPS);
        else if (VARIANT == 7)
            asm volatile(PRE "buffer_store_dwordx4 v[10:13], %5, %6, 0 offen\n\ts_nop 1\n\t"
                         "v_mov_b32 v11, %8\n\tv_mov_b32 %0, v11" OPS);
        else if (VARIANT == 8)
            asm volatile(PRE "buffer_store_dwordx2 v[10:11], %5, %6, %7 offen\n\t"
                         "v_mov_b32 v11, %8\n\tv_mov_b32 %0, v11" OPS);
        else if (VARIANT == 9)
            asm volatile(PRE "buffer_store_dwordx2 v[10:11], %5, %6, 0 offen\n\t"
                         "v_mov_b32 v11, %8\n\tv_mov_b32 %0, v11" OPS);
        else if (VARIANT == 10)
            asm volatile(PRE "buffer_store_dword v10, %5, %6, %7 offen\n\t"
                         "v_mov_b32 v10, %8\n\tv_mov_b32 %0, v10" OPS);
        else if (VARIANT == 11) {
            unsigned* dst = out + voff / 4;
            asm volatile(PRE "global_store_dwordx4 %10, v[10:13], off\n\t"
                         "v_mov_b32 v11, %8\n\tv_mov_b32 %0, v11"
                         : "=v"(r) : "v"(a), "v"(b), "v"(c), "v"(d), "v"(voff), "s"(rs), "s"(soff), "v"(g0), "v"(gp), "v"(dst) : "v10", "v11", "v12", "v13", "v14", "v15", "memory");
        } else if (VARIANT == 12)
            asm volatile(PRE "buffer_store_dwordx4 v[10:13], %5, %6, %7 offen\n\t"
                         "v_mov_b32 v13, %8\n\tv_mov_b32 %0, v13" OPS);
        else if (VARIANT == 13)
            asm volatile(PRE "buffer_store_dwordx4 v[10:13], %5, %6, %7 offen\n\t"
                         "v_mov_b32 v10, %8\n\tv_mov_b32 %0, v10" OPS);
        else if (VARIANT == 14)
            asm volatile(PRE "v_mov_b32 v14, %5\n\ts_nop 4\n\tbuffer_store_dwordx4 v[10:13], v14, %6, 0 offen\n\t"
                         "v_mov_b32 v14, 0\n\tv_mov_b32 %0, v14" OPS);
        else
            asm volatile(PRE "v_mov_b32 v14, %5\n\ts_nop 4\n\tbuffer_store_dwordx4 v[10:13], v14, %6, %7 offen\n\t"
                         "v_mov_b32 v14, 0\n\tv_mov_b32 %0, v14" OPS);
        keep ^= r;
    }
    if (keep == 0x12345u) out[0] = keep;
}

template <int VARIANT>
static long run(unsigned* out, size_t words, int iters)
{
    const int grid = 256 * 8;
    hipMemset(out, 0, words * 4);
    hipLaunchKernelGGL((probe<VARIANT>), dim3(grid), dim3(256), 0, 0, out, iters, (unsigned)(words * 4), 0u);
    hipDeviceSynchronize();
    std::vector<unsigned> h((size_t)grid * 256 * 4 * iters);
    hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
    long wrong = 0, perDword[4] = {0, 0, 0, 0}, perLane[64] = {0};
    const size_t threads = (size_t)grid * 256;
    for (int it = 0; it < iters; ++it)
        for (size_t g = 0; g < threads; ++g)
            for (int k = 0; k < (VARIANT == 8 || VARIANT == 9 ? 2 : VARIANT == 10 ? 1 : 4); ++k) {
                const unsigned e = (unsigned)((g * 4u + (unsigned)k + (unsigned)it * 977u) & 0x3fffffffu);
                if (h[((size_t)it * threads + g) * 4 + k] != e) { ++wrong; ++perDword[k]; ++perLane[g & 63]; }
            }
    printf("variant %d: %zu words stored, %ld wrong (dword 0..3: %ld %ld %ld %ld)", VARIANT, h.size(), wrong, perDword[0], perDword[1], perDword[2], perDword[3]);
    if (wrong) { printf("  lanes:"); for (int l = 0; l < 64; ++l) if (perLane[l]) printf(" %d", l); }
    printf("\n");
    return wrong;
}

int main()
{
    const int iters = 32;
    const size_t words = (size_t)256 * 8 * 256 * 4 * iters;
    unsigned* out; hipMalloc(&out, words * 4);
    for (int rep = 0; rep < 2; ++rep) {
        run<0>(out, words, iters); run<1>(out, words, iters); run<2>(out, words, iters);
        run<3>(out, words, iters); run<4>(out, words, iters); run<5>(out, words, iters);
        run<6>(out, words, iters); run<7>(out, words, iters); run<8>(out, words, iters); run<9>(out, words, iters);
        run<10>(out, words, iters); run<11>(out, words, iters); run<12>(out, words, iters); run<13>(out, words, iters); run<14>(out, words, iters); run<15>(out, words, iters);
    }
    return 0;
}
