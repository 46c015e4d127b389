// Micro-benchmark of the conv main loop shape: 8 MFMA 32x32x2 per k-step, operands double buffered
// from LDS one k-step ahead.  Variants isolate what costs cycles.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MF(acc, a, b) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0)

template <int VAR>
__global__ __launch_bounds__(256, 1) void k(float* out, const float* rnd, int iters, unsigned long long* clk)
{
    extern __shared__ float sm[];
    for (int i = threadIdx.x; i < 36864; i += 256) sm[i] = rnd[i & 16383];
    __syncthreads();
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    const int lane = threadIdx.x & 63;
    const int j = lane & 31, kh = lane >> 5, wave = threadIdx.x >> 6;
    const float* pb = sm + kh * 612 + wave * 4 * 34 + j;          // patch-like
    const float* wb = sm + 20000 + kh * 64 + j;                    // weight-like
    float a0[2], b0[4], a1[2], b1[4];
    auto ld = [&](float (&a)[2], float (&b)[4], int kk) {
        a[0] = wb[(2 * kk) * 64]; a[1] = wb[(2 * kk) * 64 + 32];
        b[0] = pb[(2 * kk) * 612]; b[1] = pb[(2 * kk) * 612 + 34]; b[2] = pb[(2 * kk) * 612 + 68]; b[3] = pb[(2 * kk) * 612 + 102];
    };
    auto mm = [&](float (&a)[2], float (&b)[4]) {
        MF(acc[0], a[0], b[0]); MF(acc[1], a[0], b[1]); MF(acc[2], a[0], b[2]); MF(acc[3], a[0], b[3]);
        MF(acc[4], a[1], b[0]); MF(acc[5], a[1], b[1]); MF(acc[6], a[1], b[2]); MF(acc[7], a[1], b[3]);
    };
    ld(a0, b0, 0);
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kk = 0; kk < 8; kk += 2) {
            if (VAR == 0) {           // pipelined + fenced (what the conv kernel does)
                ld(a1, b1, kk + 1); __builtin_amdgcn_sched_barrier(0);
                mm(a0, b0); __builtin_amdgcn_sched_barrier(0);
                ld(a0, b0, (kk + 2) & 7); __builtin_amdgcn_sched_barrier(0);
                mm(a1, b1); __builtin_amdgcn_sched_barrier(0);
            } else if (VAR == 1) {    // no LDS reads at all (operands stay)
                mm(a0, b0); mm(a0, b0);
            } else if (VAR == 2) {    // pipelined, compiler-scheduled (no fences)
                ld(a1, b1, kk + 1); mm(a0, b0); ld(a0, b0, (kk + 2) & 7); mm(a1, b1);
            } else if (VAR == 3) {    // fenced, only B reloaded (A fixed)
                b1[0] = pb[(2 * kk + 2) * 612]; b1[1] = pb[(2 * kk + 2) * 612 + 34]; b1[2] = pb[(2 * kk + 2) * 612 + 68]; b1[3] = pb[(2 * kk + 2) * 612 + 102];
                __builtin_amdgcn_sched_barrier(0);
                mm(a0, b0); __builtin_amdgcn_sched_barrier(0);
                b0[0] = pb[(2 * kk + 4) * 612]; b0[1] = pb[(2 * kk + 4) * 612 + 34]; b0[2] = pb[(2 * kk + 4) * 612 + 68]; b0[3] = pb[(2 * kk + 4) * 612 + 102];
                __builtin_amdgcn_sched_barrier(0);
                mm(a0, b1); __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) for (int jj = 0; jj < 16; ++jj) s += acc[i][jj];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + a1[0] + b1[0];
    if (threadIdx.x == 0 && blockIdx.x == 7) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}
template <int VAR>
void run(int blocks, int iters, const float* rnd, const char* name)
{
    float* d; hipMalloc(&d, blocks * 256 * 4);
    unsigned long long* c; hipMalloc(&c, 16);
    hipFuncSetAttribute((const void*)k<VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, 150000);
    for (int w = 0; w < 2; ++w) k<VAR><<<blocks, 256, 150000>>>(d, rnd, iters, c);
    hipDeviceSynchronize();
    unsigned long long h[2]; hipMemcpy(h, c, 16, hipMemcpyDeviceToHost);
    printf("%-44s clock %.0f MHz  cycles/MFMA %.1f\n", name, (double)h[0] / (double)h[1] * 100.0, (double)h[0] / (64.0 * iters));
    hipFree(d); hipFree(c);
}
int main()
{
    std::vector<float> h(16384);
    srand(1);
    for (auto& v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    float* rnd; hipMalloc(&rnd, h.size() * 4); hipMemcpy(rnd, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    run<0>(256, 3000, rnd, "pipelined + fenced (conv loop)");
    run<1>(256, 3000, rnd, "no LDS reads");
    run<2>(256, 3000, rnd, "pipelined, compiler scheduled");
    run<3>(256, 3000, rnd, "fenced, B only");
    return 0;
}
