// Micro-benchmark: achievable fp32 MFMA rate on this device, with operands (a) constant in
// registers, (b) re-read from LDS holding random data every k-step (what a real conv/GEMM does).
// Also reports the in-kernel shader clock (s_memtime / s_memrealtime, 100 MHz reference).
// Build: hipcc -O3 --offload-arch=gfx950 tools/lab/mfma_peak.hip -o /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <bool LDS>
__global__ __launch_bounds__(256, 1) void k(float* out, const float* rnd, int iters, unsigned long long* clk)
{
    __shared__ float sm[16384];
    for (int i = threadIdx.x; i < 16384; i += 256) sm[i] = rnd[i];
    __syncthreads();
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    const int lane = threadIdx.x & 63;
    float a0 = rnd[lane], a1 = rnd[64 + lane], b0 = rnd[128 + lane], b1 = rnd[192 + lane], b2 = rnd[256 + lane], b3 = rnd[320 + lane];
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (LDS) {
            const float* q = sm + ((it * 384) & 8191) + lane;
            a0 = q[0]; a1 = q[64]; b0 = q[128]; b1 = q[192]; b2 = q[256]; b3 = q[320];
        }
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b2, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b3, acc[3], 0, 0, 0);
        acc[4] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[4], 0, 0, 0);
        acc[5] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[5], 0, 0, 0);
        acc[6] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b2, acc[6], 0, 0, 0);
        acc[7] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b3, acc[7], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 7) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}
template <bool LDS>
void run(int blocks, int iters, const float* rnd, const char* name)
{
    float* d; hipMalloc(&d, blocks * 256 * 4);
    unsigned long long* c; hipMalloc(&c, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) k<LDS><<<blocks, 256>>>(d, rnd, iters, c);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<LDS><<<blocks, 256>>>(d, rnd, iters, c);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, c, 16, hipMemcpyDeviceToHost);
    double flops = 2.0 * 32 * 32 * 2 * 8.0 * iters * 4 * blocks;
    printf("%-34s blocks=%d: %.3f ms  %.1f TFLOP/s  clock %.0f MHz  cycles/MFMA %.1f\n", name, blocks, ms, flops / ms / 1e9,
           (double)h[0] / (double)h[1] * 100.0, (double)h[0] / (8.0 * iters));
    hipFree(d); hipFree(c);
}
// "mfma_peak <variant 0..3> <seconds>": keep one variant running so that rocm-smi can sample the package power.
template <bool LDS>
void soak(const float* src, double seconds)
{
    float* d; hipMalloc(&d, 256 * 256 * 4);
    unsigned long long* c; hipMalloc(&c, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    double total = 0;
    while (total < seconds * 1e3) {
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) k<LDS><<<256, 256>>>(d, src, 20000, c);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        total += ms;
    }
}
int main(int argc, char** argv)
{
    std::vector<float> h(16384);
    srand(1);
    for (auto& v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    float* rnd; hipMalloc(&rnd, h.size() * 4); hipMemcpy(rnd, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    float* zer; hipMalloc(&zer, h.size() * 4); hipMemset(zer, 0, h.size() * 4);
    if (argc > 2) {
        int v = atoi(argv[1]);
        double sec = atof(argv[2]);
        if (v & 2) soak<true>((v & 1) ? zer : rnd, sec); else soak<false>((v & 1) ? zer : rnd, sec);
        return 0;
    }
    run<false>(256, 20000, rnd, "registers, random data");
    run<false>(256, 20000, zer, "registers, zeros");
    run<true>(256, 20000, rnd, "LDS operands, random data");
    run<true>(256, 20000, zer, "LDS operands, zeros");
    run<true>(512, 10000, rnd, "LDS operands, random, 2 WG/CU?");
    return 0;
}
