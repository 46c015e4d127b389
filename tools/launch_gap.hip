// Dependent-launch gap on one stream: N back-to-back launches of a kernel that runs ~d microseconds on every CU.
// usage: launch_gap [N] ; prints total time / N for a few kernel durations, as stream launches and as a captured graph.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
// mode 0: spin only; 1: spin, then every thread writes 64 floats (33.5 MB per launch); 2: reads them instead;
// 3: reads the previous launch's buffer and writes its own (a dependent layer)
__global__ void spin(long long ticks, float* sink, float* buf, const float* prev, int mode)
{
    extern __shared__ float lds[];
    if (ticks < 0) lds[threadIdx.x] = 1.0f;
    const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
    float a = threadIdx.x;
    const size_t base = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const size_t stride = (size_t)gridDim.x * blockDim.x * 4;
    if (mode >= 2) for (int i = 0; i < 16; ++i) { const float4 v = *reinterpret_cast<const float4*>(prev + base + i * stride); a += v.x + v.w; }
    while ((long long)__builtin_amdgcn_s_memrealtime() - t0 < ticks) a = a * 1.0001f + 1.0f;
    if (mode == 1 || mode == 3) for (int i = 0; i < 16; ++i) *reinterpret_cast<float4*>(buf + base + i * stride) = make_float4(a, a, a, a);
    if (a == 12345.678f) *sink = a;
}
int main(int argc, char** argv)
{
    const int N = argc > 1 ? atoi(argv[1]) : 200;
    float* sink; hipMalloc(&sink, 4);
    float* bufs[2]; hipMalloc(&bufs[0], 512 * 256 * 64 * 4); hipMalloc(&bufs[1], 512 * 256 * 64 * 4);
    hipMemset(bufs[0], 0, 512 * 256 * 64 * 4); hipMemset(bufs[1], 0, 512 * 256 * 64 * 4);
    hipStream_t s; hipStreamCreate(&s);
    const int LDS = argc > 2 ? atoi(argv[2]) : 0;
    const int EXT = argc > 3 ? atoi(argv[3]) : 0;
    if (LDS > 65536) hipFuncSetAttribute((const void*)spin, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    printf("dynamic LDS %d bytes, %s\n", LDS, EXT ? "hipExtLaunchKernelGGL" : "hipLaunchKernelGGL");
  for (int kmode = 0; kmode < 2; ++kmode)
    for (int us : {80}) {
        for (int mode = 0; mode < 2; ++mode) {
            hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
            if (mode == 1) {
                hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
                for (int i = 0; i < N; ++i) { if (EXT) hipExtLaunchKernelGGL(spin, dim3(512), dim3(256), LDS, s, nullptr, nullptr, 0, (long long)us * 100, sink, bufs[i & 1], bufs[(i & 1) ^ 1], kmode); else hipLaunchKernelGGL(spin, dim3(512), dim3(256), LDS, s, (long long)us * 100, sink, bufs[i & 1], bufs[(i & 1) ^ 1], kmode); }
                hipStreamEndCapture(s, &g);
                hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
                hipGraphLaunch(ge, s); hipStreamSynchronize(s);
            } else {
                for (int i = 0; i < 10; ++i) { if (EXT) hipExtLaunchKernelGGL(spin, dim3(512), dim3(256), LDS, s, nullptr, nullptr, 0, (long long)us * 100, sink, bufs[i & 1], bufs[(i & 1) ^ 1], kmode); else hipLaunchKernelGGL(spin, dim3(512), dim3(256), LDS, s, (long long)us * 100, sink, bufs[i & 1], bufs[(i & 1) ^ 1], kmode); }
                hipStreamSynchronize(s);
            }
            auto t0 = std::chrono::high_resolution_clock::now();
            if (mode == 1) hipGraphLaunch(ge, s);
            else for (int i = 0; i < N; ++i) { if (EXT) hipExtLaunchKernelGGL(spin, dim3(512), dim3(256), LDS, s, nullptr, nullptr, 0, (long long)us * 100, sink, bufs[i & 1], bufs[(i & 1) ^ 1], kmode); else hipLaunchKernelGGL(spin, dim3(512), dim3(256), LDS, s, (long long)us * 100, sink, bufs[i & 1], bufs[(i & 1) ^ 1], kmode); }
            hipStreamSynchronize(s);
            const double dt = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
            printf("kernel mode %d, %s, kernel %3d us: %.2f us per launch -> overhead %.2f us\n", kmode, mode ? "graph " : "stream", us, dt / N * 1e6, dt / N * 1e6 - us);
            if (ge) hipGraphExecDestroy(ge);
            if (g) hipGraphDestroy(g);
        }
    }
    return 0;
}
