// The low-resolution TRUNK of EnhanceNet -- preblock (conv3x3 101 -> 64 + ReLU) and the ten residual blocks
// (SuperresolutionNetwork/models/enhancenet.py:92-112,136-141: f = relu(pre(x)); f = f + conv2(relu(conv1(f))) x 10) -- as ONE persistent
// launch with tile-level DATAFLOW instead of 21 dependent launches.  Same split-operand products in the same order as the
// per-layer kernels of sr_conv_split.hip: the output is bit-identical to them.
//
// Why.  At 480 x 270 a trunk layer is ONE round of 510 workgroups on 512 slots.  Every workgroup stages, multiplies and stores in
// step with every other one, so the memory system idles while the matrix pipes work and vice versa; every layer pays a launch
// boundary, an exposed first staging, a chip-wide store burst and a tail: 37-50 us per layer for 11.5 us of matrix issue, 40 % of
// the frame.  Round 2 ruled out a persistent trunk with GRID barriers (62 us per barrier: 512 concurrent L2 write-back fences
// serialise).  Here there is no global barrier at all: workgroup w owns tile w through all 21 layers, and layer l of a tile starts
// as soon as its 3 x 3 neighbourhood has finished layer l - 1 (one progress counter per tile).  Neighbours stay within one layer
// of each other, the chip as a whole de-phases -- some workgroups store while others multiply -- and nothing is launched in between.
// Measured on a chain of 20 plain layers (tools/bench_chain.py): 34 us per layer against 51 for dependent launches on the same box.
//
// Storage: the carried feature tensor F is updated IN PLACE and one intermediate tensor T is reused by every block.  A tile
// overwrites its region of a tensor only after all its neighbours -- the only other readers of that region -- have finished the
// layer that read it: conv2 of tile t (writes F[t], reads T with halo) needs its neighbours' conv1 of the same block done, which
// was their last read of F[t]'s halo before the next block; conv1 of the next block (writes T[t]) needs their conv2 done.
//
// Visibility across CUs / XCDs (MI355X_MICROARCH.md, "inter-workgroup visibility": producer with drained write-through stores,
// consumer with one agent-scope acquire):
//   producer: every output store is `sc1` (write-through), every wave drains its stores (s_waitcnt vmcnt(0)), the workgroup meets
//             at a barrier, ONE lane publishes the tile's progress with an agent-scope store;
//   consumer: lanes 0..8 of wave 0 poll the neighbours' counters (relaxed agent-scope loads, s_sleep between polls, a deadline
//             on the chip's 100 MHz clock: a neighbour that never arrives ends the launch with an error word -- never a hang),
//             then ONE agent-scope acquire (buffer_inv sc1) + s_waitcnt vmcnt(0) + barrier, then plain loads.
// Every workgroup must be resident at once: the host refuses images of more than 2 x #CUs tiles (the per-layer kernels take those).
#include <cstring>

#include "sr_split_common.h"

namespace {

constexpr int TK_MAX_LAYERS = 24;
constexpr int TK_QPR = (ST_W + 8) / 4;
constexpr int TK_QUNITS = S_GROUPS * SP_H * TK_QPR;                           // 400 staging units per 32-channel chunk
constexpr int TK_LDS_BYTES = S_LDS_BYTES + 64;

struct TrunkLayer {
    const float* in; const float* residual; float* out;
    const u32x4* wq; const float* bias;
    int cin, inPlane, act;
};

struct TrunkParams {
    TrunkLayer layer[TK_MAX_LAYERS];
    unsigned* done;                  // [tiles] layers finished by each tile (zeroed before the launch)
    unsigned* error;                 // 1 + the layer at which a wait timed out (0: none)
    unsigned* absmax;                // range guard over every value stored (may be NULL)
    int H, W, plane, tilesX, tilesY, layers;
    unsigned long long timeoutTicks; // of the chip's 100 MHz clock
};

__global__ __launch_bounds__(S_THREADS, 2) void trunk_dataflow_kernel(const TrunkParams p)
{
    extern __shared__ u32x4 patch[];
    u32x4* wbuf = patch + S_PUNITS;
    int* flags = reinterpret_cast<int*>(wbuf + S_WUNITS);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int ntiles = p.tilesX * p.tilesY;
    int tile;
    {   // an XCD (= an L2) gets a contiguous range of tiles: neighbours share halo lines
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    }
    if (tile >= ntiles) return;
    const int tx = tile % p.tilesX, ty = tile / p.tilesX;
    const int oy0 = ty * ST_H, ox0 = tx * ST_W;
    const unsigned long long deadline = __builtin_amdgcn_s_memrealtime() + p.timeoutTicks;
    u32x4* wdst = wbuf + (tid >> 7) * S_WPART + (tid & 127);
    u32x4 wreg[9];

#pragma unroll 1
    for (int l = 0; l < p.layers; ++l) {
        const TrunkLayer& L = p.layer[l];
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
            if (flags[0]) {                                                  // a neighbour never arrived: give up, loudly
                if (tid == 0) atomicMax(p.error, (unsigned)(1 + l));
                return;
            }
        }
        const int ksteps = (L.cin + 15) >> 4;
        const unsigned planeBytes = (unsigned)L.inPlane * 4u;
        const rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(L.in), 0, (int)((size_t)L.cin * L.inPlane * 4), 0x00020000);
        const rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(L.wq + 1), 0, 9 * ksteps * 4096, 0x00020000);
        auto wfetch = [&](int ks) {
            if (ks >= ksteps) return;
#pragma unroll
            for (int i = 0; i < 9; ++i) wreg[i] = __builtin_amdgcn_raw_buffer_load_b128(wrs, tid * 16, (i * ksteps + ks) * 4096, 0);
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
#pragma unroll 1
        for (int cin0 = 0; cin0 < L.cin; cin0 += S_CHUNK) {
            const int ks0 = cin0 >> 4;
            const int nks = min(2, ksteps - ks0);
            // the 32-channel chunk of the patch, split into (hi, lo'): unit = (channel group, patch row, aligned quad of 4 pixels);
            // channels beyond cin fall behind the descriptor's range and read as zero
            for (int u0 = tid; u0 < TK_QUNITS; u0 += 2 * S_THREADS) {
                u32x4 v[2][8];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int u = u0 + k * S_THREADS;
                    const int g = u / (SP_H * TK_QPR), rem = u - g * (SP_H * TK_QPR);
                    const int r = rem / TK_QPR, q = rem - r * TK_QPR;
                    const int iy = oy0 + r - 1, ix = ox0 - 4 + 4 * q;
                    const bool ok = u < TK_QUNITS && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                    const unsigned base = (unsigned)(cin0 + g * 8) * planeBytes + (unsigned)(iy * p.W + ix) * 4u;
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        v[k][e] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)(ok ? base + (unsigned)e * planeBytes : BAD_OFFSET), 0, 0);
                }
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int u = u0 + k * S_THREADS;
                    if (u >= TK_QUNITS) continue;
                    const int g = u / (SP_H * TK_QPR), rem = u - g * (SP_H * TK_QPR);
                    const int r = rem / TK_QPR, q = rem - r * TK_QPR;
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
                    if (q > 0 && q < TK_QPR - 1) {
                        dst[1] = __builtin_bit_cast(u32x4, h1); dst[S_PART + 1] = __builtin_bit_cast(u32x4, l1);
                        dst[2] = __builtin_bit_cast(u32x4, h2); dst[S_PART + 2] = __builtin_bit_cast(u32x4, l2);
                    }
                    if (q < TK_QPR - 1) { dst[3] = __builtin_bit_cast(u32x4, h3); dst[S_PART + 3] = __builtin_bit_cast(u32x4, l3); }
                }
            }
            wpark();
            __syncthreads();
#pragma unroll
            for (int S = 0; S < 2; ++S) {
                if (S < nks) {
                    wfetch(ks0 + S + 1);
                    split_kstep(acc, wbuf + h * 64 + j, patch + (2 * S + h) * SP_PIX + (wave * 2) * SP_W + j, true);
                    __syncthreads();
                    if (S + 1 < nks) { wpark(); __syncthreads(); }
                }
            }
        }
        SplitConvParams q;
        q.x = nullptr; q.wq = L.wq; q.bias = L.bias; q.residual = L.residual; q.y = L.out;
        q.N = 1; q.Cin = L.cin; q.H = p.H; q.W = p.W; q.Cout = 64;
        q.xPlane = L.inPlane; q.yPlane = p.plane; q.rPlane = p.plane; q.xImage = 0; q.yImage = 0; q.rImage = 0;
        q.ksteps = ksteps; q.coutPad = 64; q.cgroups = 1; q.tilesX = p.tilesX; q.tilesY = p.tilesY;
        q.act = L.act; q.slope = 0.f; q.Hin = p.H; q.Win = p.W; q.quads = 1; q.dbg = 0; q.stamps = nullptr;
        q.ps = nullptr; q.psPlane = 0; q.xps = nullptr; q.xpsPlane = 0; q.zero = nullptr; q.absmax = p.absmax;
        split_epilogue<true, 16>(q, acc, patch, 0, oy0, ox0, 0, true, lane, wave, j, h);      // sc1: write-through stores
        // ---- publish: every wave's stores drained, then one agent-scope store of the tile's progress -------------------------
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(p.done + tile, (unsigned)(l + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

} // namespace

extern "C" {

int isrTrunkDataflowMaxTiles(void)
{
    int dev = 0, cus = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    return 2 * cus;
}

long long isrTrunkDataflowWorkspaceBytes(int H, int W)
{
    if (H <= 0 || W <= 0) return -1;
    const long long tiles = (long long)((W + ST_W - 1) / ST_W) * ((H + ST_H - 1) / ST_H);
    return (tiles + 2) * 4;
}

int isrTrunkDataflowSupported(const float* x, int H, int W, long long xPlane, long long plane)
{
    if (!x || H <= 0 || W <= 0 || (W & 3) || (xPlane & 3) || (plane & 3) || ((uintptr_t)x & 15)) return 0;
    if (xPlane < (long long)H * W || plane < (long long)H * W || xPlane * 101 * 4 > 0x7fffffffLL || plane * 64 * 4 > 0x7fffffffLL) return 0;
    return ((W + ST_W - 1) / ST_W) * ((H + ST_H - 1) / ST_H) <= isrTrunkDataflowMaxTiles() ? 1 : 0;
}

/* x [cin0][H][W] -> F = relu(conv(x, w[0]) + b[0]); then nblocks times F += conv(relu(conv(F, w[2k+1]) + b[2k+1]), w[2k+2]) + b[2k+2]
 * (F updated in place, T the intermediate; both [64][H][W] with `plane` floats per channel).  wq[l]: isrConvSplitPrepare images. */
int isrTrunkDataflow(const float* x, int cin0, long long xPlane, float* F, float* T, long long plane, const void* const* wq,
                     const float* const* bias, int nblocks, int H, int W, void* workspace, void* stream)
{
    if (!x || !F || !T || !wq || !bias || !workspace || nblocks < 0 || 1 + 2 * nblocks > TK_MAX_LAYERS || cin0 <= 0) return -1;
    if (!isrTrunkDataflowSupported(x, H, W, xPlane, plane) || ((uintptr_t)F & 15) || ((uintptr_t)T & 15)) return -3;
    TrunkParams p;
    std::memset(&p, 0, sizeof(p));
    const int layers = 1 + 2 * nblocks;
    for (int l = 0; l < layers; ++l) {
        TrunkLayer& L = p.layer[l];
        if (!wq[l]) return -1;
        L.wq = (const u32x4*)wq[l]; L.bias = bias[l];
        if (l == 0) { L.in = x; L.cin = cin0; L.inPlane = (int)xPlane; L.out = F; L.residual = nullptr; L.act = ISR_ACT_RELU; }
        else if (l & 1) { L.in = F; L.cin = 64; L.inPlane = (int)plane; L.out = T; L.residual = nullptr; L.act = ISR_ACT_RELU; }
        else { L.in = T; L.cin = 64; L.inPlane = (int)plane; L.out = F; L.residual = F; L.act = ISR_ACT_NONE; }
    }
    p.H = H; p.W = W; p.plane = (int)plane; p.layers = layers;
    p.tilesX = (W + ST_W - 1) / ST_W; p.tilesY = (H + ST_H - 1) / ST_H;
    const int ntiles = p.tilesX * p.tilesY;
    p.done = (unsigned*)workspace; p.error = p.done + ntiles;
    p.absmax = isr_take_range_flag();
    p.timeoutTicks = 5000000ull;                                             // 50 ms: a frame is 2 ms
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(workspace, 0, (size_t)(ntiles + 1) * sizeof(unsigned), s) != hipSuccess) return -2;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)trunk_dataflow_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, TK_LDS_BYTES); attr = true; }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    isr_profile_record(ISR_VARIANT_SPLIT_TRUNK, 2.0 * 9 * 64 * ((double)cin0 + 2.0 * nblocks * 64) * (double)H * W, &e0, &e1);
    const dim3 grid((unsigned)(((ntiles + 7) / 8) * 8)), block(S_THREADS);
    if (e0 || e1) hipExtLaunchKernelGGL(trunk_dataflow_kernel, grid, block, TK_LDS_BYTES, s, e0, e1, 0, p);
    else hipLaunchKernelGGL(trunk_dataflow_kernel, grid, block, TK_LDS_BYTES, s, p);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

} // extern "C"
