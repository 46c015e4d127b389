// EXPERIMENT (not on the product path): a chain of L plain 64 -> 64 convolutions (+ bias, ReLU) of one 480 x 270-class image as ONE
// persistent launch with tile-level dataflow instead of L dependent launches.
//
// A trunk layer of this network is one round of 510 workgroups on 512 slots: every workgroup stages, multiplies and stores at the
// same time as every other, so the memory system idles while the matrix pipes work and vice versa, and every layer pays a launch
// boundary and a tail (37-42 us per layer for 11.5 us of matrix issue).  Here workgroup w owns tile w through ALL layers; layer l
// of a tile may start as soon as its 3 x 3 neighbourhood has finished layer l - 1 (a per-tile progress counter), so neighbouring
// tiles stay within one layer of each other but the chip as a whole de-phases: some workgroups store while others multiply.
// Two ping-pong tensors suffice: a tile overwrites its region of the buffer it read two layers ago only after all its
// neighbours -- the only other readers of that region -- have finished the layer in between.
//
// Visibility across CUs / XCDs (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility"):
//   producer: every output store is `sc1` (write-through), every wave drains its stores (s_waitcnt vmcnt(0)), the workgroup meets
//             at a barrier, ONE lane publishes the tile's progress with an agent-scope (sc1) store;
//   consumer: lanes 0..8 of wave 0 poll the neighbours' counters with relaxed agent-scope loads (s_sleep between polls, a
//             deadline on the chip's 100 MHz clock: a stuck neighbour ends the launch with an error word, never a hang), then ONE
//             agent-scope acquire (buffer_inv sc1), s_waitcnt vmcnt(0), a barrier, plain loads.
// Every workgroup must be resident at once (ntiles <= 2 x #CUs: checked on the host).
#include "sr_split_common.h"

namespace {

constexpr int CH_MAX_LAYERS = 32;
constexpr int CH_QPR = (ST_W + 8) / 4;
constexpr int CH_QUNITS = S_GROUPS * SP_H * CH_QPR;                           // 400 staging units per 32-channel chunk
constexpr int CH_LDS_BYTES = S_LDS_BYTES + 64;

struct ChainParams {
    const float* x0;                 // input of layer 0
    float* buf[2];                   // layer l writes buf[l & 1] and (l > 0) reads buf[(l - 1) & 1]
    const u32x4* wq[CH_MAX_LAYERS];
    const float* bias[CH_MAX_LAYERS];
    unsigned* done;                  // [tiles] layers finished by each tile (zeroed before the launch)
    unsigned* error;                 // set to 1 + layer if a wait timed out
    int H, W, plane, tilesX, tilesY, layers;
    unsigned long long timeoutTicks; // 100 MHz ticks
    int startDelayTicks;             // the second half of the grid starts this much later (forced de-phasing), or 0
};

__global__ __launch_bounds__(S_THREADS, 2) void conv_chain_kernel(const ChainParams p)
{
    extern __shared__ u32x4 patch[];
    u32x4* wbuf = patch + S_PUNITS;
    int* flags = reinterpret_cast<int*>(wbuf + S_WUNITS);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int ntiles = p.tilesX * p.tilesY;
    int tile;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    }
    if (tile >= ntiles) return;
    const int tx = tile % p.tilesX, ty = tile / p.tilesX;
    const int oy0 = ty * ST_H, ox0 = tx * ST_W;
    const unsigned planeBytes = (unsigned)p.plane * 4u;
    const unsigned long long deadline = __builtin_amdgcn_s_memrealtime() + p.timeoutTicks;
    if (p.startDelayTicks > 0 && (int)blockIdx.x >= ((int)gridDim.x >> 1)) {
        const unsigned long long until = __builtin_amdgcn_s_memrealtime() + (unsigned long long)p.startDelayTicks;
        while (__builtin_amdgcn_s_memrealtime() < until) __builtin_amdgcn_s_sleep(8);
    }
    u32x4* wdst = wbuf + (tid >> 7) * S_WPART + (tid & 127);
    u32x4 wreg[9];

    for (int l = 0; l < p.layers; ++l) {
        // ---- wait for the 3 x 3 neighbourhood to have finished layer l - 1 ------------------------------------------------
        if (l > 0) {
            if (tid == 0) flags[0] = 0;
            __syncthreads();
            if (tid < 9 && tid != 4) {
                const int ny = ty + tid / 3 - 1, nx = tx + tid % 3 - 1;
                if ((unsigned)ny < (unsigned)p.tilesY && (unsigned)nx < (unsigned)p.tilesX) {
                    const unsigned* f = p.done + ny * p.tilesX + nx;
                    while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)l) {
                        __builtin_amdgcn_s_sleep(4);
                        if (__builtin_amdgcn_s_memrealtime() > deadline) { flags[0] = 1; break; }
                    }
                }
            }
            if (wave == 0) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();
            if (flags[0]) {                                                  // a neighbour never arrived: give up loudly
                if (tid == 0) atomicMax(p.error, (unsigned)(1 + l));
                return;
            }
        }
        const float* xin = l == 0 ? p.x0 : p.buf[(l - 1) & 1];
        const rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xin), 0, (int)((size_t)64 * p.plane * 4), 0x00020000);
        const rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(p.wq[l] + 1), 0, 9 * 4 * 4096, 0x00020000);
        auto wfetch = [&](int ks) {
            if (ks >= 4) return;
#pragma unroll
            for (int i = 0; i < 9; ++i) wreg[i] = __builtin_amdgcn_raw_buffer_load_b128(wrs, tid * 16, (i * 4 + ks) * 4096, 0);
        };
        auto wpark = [&]() {
#pragma unroll
            for (int i = 0; i < 9; ++i) wdst[i * 128] = wreg[i];
        };
        f32x16 acc[2][2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[cb][r][i] = 0.0f;
        wfetch(0);
        for (int cin0 = 0; cin0 < 64; cin0 += S_CHUNK) {
            for (int u0 = tid; u0 < CH_QUNITS; u0 += 2 * S_THREADS) {
                u32x4 v[2][8];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int u = u0 + k * S_THREADS;
                    const int g = u / (SP_H * CH_QPR), rem = u - g * (SP_H * CH_QPR);
                    const int r = rem / CH_QPR, q = rem - r * CH_QPR;
                    const int iy = oy0 + r - 1, ix = ox0 - 4 + 4 * q;
                    const bool ok = u < CH_QUNITS && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                    const unsigned base = (unsigned)(cin0 + g * 8) * planeBytes + (unsigned)(iy * p.W + ix) * 4u;
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        v[k][e] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)(ok ? base : BAD_OFFSET), (int)((unsigned)e * planeBytes), 0);
                }
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int u = u0 + k * S_THREADS;
                    if (u >= CH_QUNITS) continue;
                    const int g = u / (SP_H * CH_QPR), rem = u - g * (SP_H * CH_QPR);
                    const int r = rem / CH_QPR, q = rem - r * CH_QPR;
                    f16x8 h0, h1, h2, h3, l0, l1, l2, l3;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float4 f = __builtin_bit_cast(float4, v[k][e]);
                        _Float16 a, b;
                        split16x(f.x, a, b); h0[e] = a; l0[e] = b;
                        split16x(f.y, a, b); h1[e] = a; l1[e] = b;
                        split16x(f.z, a, b); h2[e] = a; l2[e] = b;
                        split16x(f.w, a, b); h3[e] = a; l3[e] = b;
                    }
                    u32x4* dst = patch + g * SP_PIX + r * SP_W + 4 * q - 3;
                    if (q > 0) { dst[0] = __builtin_bit_cast(u32x4, h0); dst[S_PART] = __builtin_bit_cast(u32x4, l0); }
                    if (q > 0 && q < CH_QPR - 1) {
                        dst[1] = __builtin_bit_cast(u32x4, h1); dst[S_PART + 1] = __builtin_bit_cast(u32x4, l1);
                        dst[2] = __builtin_bit_cast(u32x4, h2); dst[S_PART + 2] = __builtin_bit_cast(u32x4, l2);
                    }
                    if (q < CH_QPR - 1) { dst[3] = __builtin_bit_cast(u32x4, h3); dst[S_PART + 3] = __builtin_bit_cast(u32x4, l3); }
                }
            }
            wpark();
            __syncthreads();
#pragma unroll
            for (int S = 0; S < 2; ++S) {
                wfetch((cin0 >> 4) + S + 1);
                split_kstep(acc, wbuf + h * 64 + j, patch + (2 * S + h) * SP_PIX + (wave * 2) * SP_W + j, true);
                __syncthreads();
                if (S == 0) { wpark(); __syncthreads(); }
            }
        }
        SplitConvParams q;
        q.x = nullptr; q.wq = p.wq[l]; q.bias = p.bias[l]; q.residual = nullptr; q.y = p.buf[l & 1];
        q.N = 1; q.Cin = 64; q.H = p.H; q.W = p.W; q.Cout = 64;
        q.xPlane = p.plane; q.yPlane = p.plane; q.rPlane = p.plane; q.xImage = 0; q.yImage = 0; q.rImage = 0;
        q.ksteps = 4; q.coutPad = 64; q.cgroups = 1; q.tilesX = p.tilesX; q.tilesY = p.tilesY;
        q.act = ISR_ACT_RELU; q.slope = 0.f; q.Hin = p.H; q.Win = p.W; q.quads = 1; q.dbg = 0; q.stamps = nullptr;
        q.ps = nullptr; q.psPlane = 0; q.xps = nullptr; q.xpsPlane = 0; q.zero = nullptr; q.absmax = nullptr;
        split_epilogue<true, 16>(q, acc, patch, 0, oy0, ox0, 0, true, lane, wave, j, h);      // sc1: write-through stores
        // ---- publish: every wave's stores drained, then one agent-scope store of the tile's progress -------------------------
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(p.done + tile, (unsigned)(l + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

} // namespace

extern "C" {

// ws: [tiles + 1] unsigned words (progress counters, then the error word).  Returns 0, -1 bad arguments, -2 launch failure,
// -3 the image has more tiles than the GPU holds workgroups.
int isrDebugConvChain(const float* x0, float* bufA, float* bufB, const void* const* wq, const float* const* bias, int layers,
                      int H, int W, long long plane, void* ws, int startDelayTicks, void* stream)
{
    if (!x0 || !bufA || !bufB || !wq || !bias || !ws || layers <= 0 || layers > CH_MAX_LAYERS || H <= 0 || W <= 0 || (W & 3) || (plane & 3)) return -1;
    ChainParams p;
    p.x0 = x0; p.buf[0] = bufA; p.buf[1] = bufB;
    for (int l = 0; l < CH_MAX_LAYERS; ++l) { p.wq[l] = l < layers ? (const u32x4*)wq[l] : nullptr; p.bias[l] = l < layers ? bias[l] : nullptr; }
    p.H = H; p.W = W; p.plane = (int)plane; p.layers = layers;
    p.tilesX = (W + ST_W - 1) / ST_W; p.tilesY = (H + ST_H - 1) / ST_H;
    const int ntiles = p.tilesX * p.tilesY;
    int dev = 0, cus = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    if (ntiles > 2 * cus) return -3;
    p.done = (unsigned*)ws; p.error = p.done + ntiles;
    p.timeoutTicks = 2000000ull;                  // 20 ms
    p.startDelayTicks = startDelayTicks;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(ws, 0, (size_t)(ntiles + 1) * sizeof(unsigned), s) != hipSuccess) return -2;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)conv_chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, CH_LDS_BYTES); attr = true; }
    hipLaunchKernelGGL(conv_chain_kernel, dim3((unsigned)(((ntiles + 7) / 8) * 8)), dim3(S_THREADS), CH_LDS_BYTES, s, p);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

} // extern "C"
