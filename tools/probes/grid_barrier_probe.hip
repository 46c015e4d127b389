// Probe: what does a grid-wide barrier between "layers" cost on MI355X when every workgroup has just written 64 KB?
// 512 workgroups x 256 threads (2 per CU, 80 KB LDS each as the split conv kernel), L rounds of
//   { write 64 KB per workgroup (plain stores), s_waitcnt, barrier, lane 0: release fence + atomic arrive + bounded spin,
//     barrier, acquire fence, read a neighbour's 64 KB }.
// Every spin is bounded by a timeout: the kernel always terminates.  Build + run: see tools/probes/run_grid_barrier_probe.sh
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256, 2) void probe(float* buf, unsigned* counter, unsigned* abortFlag, int rounds, int nwg, int mode, float* sink)
{
    extern __shared__ float lds[];
    const int tid = threadIdx.x, wg = blockIdx.x;
    float acc = 0.f;
    for (int r = 0; r < rounds; ++r) {
        float* mine = buf + ((size_t)(r & 1) * nwg + wg) * 16384;            // 64 KB per workgroup, ping-pong
        for (int i = tid; i < 16384 / 4; i += 256) reinterpret_cast<float4*>(mine)[i] = make_float4(r + acc, wg, i, 1.f);
        if (mode == 0) continue;                                             // mode 0: no synchronisation at all (lower bound)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)(r + 1) * (unsigned)nwg;
            const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
            for (;;) {
                if (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) break;
                if (__hip_atomic_load(abortFlag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                if ((long long)__builtin_amdgcn_s_memrealtime() - t0 > 5000000LL) {          // 50 ms at 100 MHz
                    __hip_atomic_store(abortFlag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if (mode == 2) {                                                     // mode 2: every wave's own acquire as well (L1 of the CU is shared: not needed)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        const float* other = buf + ((size_t)(r & 1) * nwg + (wg + 37) % nwg) * 16384;   // a tile another (often other-XCD) workgroup wrote
        for (int i = tid; i < 16384 / 4; i += 256) { const float4 v = reinterpret_cast<const float4*>(other)[i]; acc += v.x * 1e-9f + v.w * 1e-9f; }
    }
    if (acc == 12345.f) sink[0] = acc;
}

int main()
{
    const int nwg = 512, rounds = 20;
    float* buf; unsigned* counter; float* sink;
    hipMalloc(&buf, (size_t)2 * nwg * 65536); hipMalloc(&counter, 8); hipMalloc(&sink, 4);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 80384);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 3; ++mode) {
        float best = 1e9f; unsigned aborted = 0;
        for (int rep = 0; rep < 5; ++rep) {
            hipMemset(counter, 0, 8);
            hipEventRecord(e0);
            hipLaunchKernelGGL(probe, dim3(nwg), dim3(256), 80384, 0, buf, counter, counter + 1, rounds, nwg, mode, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
            unsigned h[2]; hipMemcpy(h, counter, 8, hipMemcpyDeviceToHost); aborted |= h[1];
        }
        printf("mode %d: %.1f us per round (%d rounds, %d workgroups)%s\n", mode, best * 1000.f / rounds, rounds, nwg, aborted ? "  ABORTED (timeout)" : "");
    }
    // the same 20 rounds as 20 dependent launches of a kernel that does one round without synchronisation
    hipEventRecord(e0);
    for (int r = 0; r < rounds; ++r) hipLaunchKernelGGL(probe, dim3(nwg), dim3(256), 80384, 0, buf, counter, counter + 1, 1, nwg, 0, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("20 dependent launches of one unsynchronised round each: %.1f us per launch\n", ms * 1000.f / rounds);
    return 0;
}
