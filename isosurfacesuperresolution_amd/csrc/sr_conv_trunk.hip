// The low-resolution TRUNK of EnhanceNet -- preblock (conv3x3 101 -> 64 + ReLU) and the ten residual blocks
// (SuperresolutionNetwork/models/enhancenet.py:92-112,136-141: f = relu(pre(x)); f = f + conv2(relu(conv1(f))) x 10) -- as ONE persistent
// launch with tile-level DATAFLOW: workgroup w owns tile w through all 21 layers, and layer l of a tile starts as soon as its
// 3 x 3 neighbourhood has finished layer l - 1 (one progress counter per tile, no grid barrier).  Same split-operand products in
// the same order as the per-layer kernels of sr_conv_split.hip: the output is bit-identical to them.
//
// Second form of the idea (the first, in the history of this file, kept the per-layer kernel's shape: 8 x 32 tiles, two workgroups
// per CU, staging through registers).  Its phase stamps showed a workgroup's four phases -- wait 8 us, staging 8, MFMAs 12,
// epilogue 6..12 per layer -- running strictly one after the other, AND the two workgroups of a CU doing so in step (they are near
// neighbours in the image, the dataflow keeps them within a layer of each other): nothing overlapped, 40 us per layer for 11.5 us
// of matrix issue.  This form overlaps inside the workgroup instead:
//
//   * tile 16 x 32, 512 threads, ONE workgroup per CU (480 x 270 = 255 tiles on 256 CUs), 152 KB of LDS:
//     two patch buffers (one k-step = 16 channels of the 18 x 34 patch, hi and lo') and two weight buffers (one k-step);
//   * every byte that enters LDS from memory comes by LDS-DMA (global_load_lds_dwordx4, scalar base + 32-bit lane offset, issued
//     from inline assembly): activations travel between layers in the PACKED-SPLIT format (sr_split_common.h: already split into
//     fp16 pairs, 8 channels of a pixel per 16-byte unit; every plane ends in a zero unit that padding pixels read), so staging is
//     a copy -- no registers, no conversion.  While k-step s multiplies, k-step s + 1's patch and weights land in the other
//     buffers, one request after each tap's MFMAs, waves 0..3 fetching weights and waves 4..7 the patch (Trunk16Lane): ONE barrier
//     per k-step;
//   * the epilogue works straight from the MFMA result layout (no transposition through LDS).  Channels 0 .. 31 of the tile -- the
//     next layer's first two k-steps -- are written INTO the two patch buffers (the centre never leaves the CU) and to memory only
//     on the tile's outermost ring (what the neighbours read); channels 32 .. 63 go to memory (8 bytes per lane, a wave instruction
//     = 512 contiguous bytes) and return by DMA under those two k-steps.  Per tile and layer 77 KB are stored and 91 KB fetched
//     instead of 131 + 157: an XCD's 32 tiles fit its 4 MB L2;
//   * the residual stream F never leaves the registers: each lane keeps its 64 values of the tile across the ten blocks
//     (F += conv2(...) is an add between two register arrays); memory only ever sees the packed-split copy the next conv1 reads;
//   * the next layer's first weights and its bias travel under the last k-step;
//   * the range guard (SplitConvParams::absmax) is ONE atomic per wave per launch: one per layer (2040 atomics on one address,
//     each waited for by the publish) cost the first form 20 us per layer when nothing else did.
//
// Visibility across CUs / XCDs (MI355X_MICROARCH.md, "inter-workgroup visibility"): producer stores are `sc1` (write-through), every
// wave drains its stores (s_waitcnt vmcnt(0)), barrier, ONE lane publishes the tile's progress with an agent-scope store; the
// consumer polls the eight neighbours' counters (relaxed agent-scope loads, s_sleep, a deadline on the chip's 100 MHz clock: a
// neighbour that never arrives ends the launch with an error word -- never a hang), the other waves follow behind a workgroup
// barrier, and every activation byte is then loaded `sc1` (past the CU's L1).  That is the guide's hand-off "one lane of each storing
// workgroup signals with an sc1 flag store for all its stores / sc1 poll / sc1 stores / sc1 loads, one workgroup per CU" with its
// four conditions met; the one difference is that the loads are LDS-DMA (global_load_lds_dwordx4 sc1) instead of loads to registers,
// which the guide's table does not list -- the bit-exactness tests (tests/test_trunk_gpu.py, also under a second stream's load and
// forty launches back to back) are what covers it.  No cache is ever invalidated: an acquire per layer and CU cost more than it
// saved (the weights stay in L2).
// Every workgroup must be resident at once: the host refuses images of more than #CUs tiles (the per-layer kernels take those).
//
// Measured (tools/lab/bench_trunk.py, tools/lab/trunk_timeline.py; 480 x 270, 21 layers): 0.58-0.65 ms against 0.84 for the first form and
// 0.83-1.15 for 21 launches.  Per layer ~29 us: MFMA phase 22 (the 108 x 4 MFMAs of a wave take 17 at the 1.6 GHz the chip holds
// under this load -- 11 with operands read once, i.e. at full clock), epilogue 2.2, drain 1.4, wait 2.5, halo fetch 1.8.
#include "sr_diag.h"
#include <cstdlib>
#include <cstring>

#include "sr_split_common.h"

namespace {

#ifndef T16_EARLY_PATCH
#define T16_EARLY_PATCH 5
#endif
constexpr int T16_H = 16, T16_W = 32, T16_THREADS = 512, T16_WAVES = 8;
constexpr int P16_H = T16_H + 2, P16_W = T16_W + 2, P16_PIX = P16_H * P16_W;          // 612 patch pixels
constexpr int P16_PART = 2 * P16_PIX;                                                 // one k-step: 2 channel groups; hi, then lo at + P16_PART
constexpr int P16_UNITS = 2 * P16_PART;                                               // 2448 units = 39168 bytes
constexpr int P16_SUBS = (P16_PIX + 63) / 64;                                         // 10 wave-wide DMA pieces per (part, group)
constexpr int T16_LDS_UNITS = 2 * P16_UNITS + 2 * S_WUNITS;                           // 2 patch + 2 weight buffers
constexpr int T16_LDS_BYTES = T16_LDS_UNITS * 16 + 2 * 64 * 4 + 64;                   // + two bias buffers + flags = 152640
constexpr int T16_MAX_LAYERS = 24;

// The layers are regular: layer 0 is the preblock (x -> F, ReLU), odd layers are a block's first convolution (F -> T, ReLU), even
// layers its second (F += conv(T)); only the weights and biases differ.
struct Trunk16Params {
    const u32x4* wq[T16_MAX_LAYERS]; const float* bias[T16_MAX_LAYERS];
    const char* ws;                  // the workspace; its first 16 bytes are zero: source of the padding units
    unsigned xpsOff, fpsOff, tpsOff; // byte offsets in the workspace of the packed-split tensors [hi | lo][groups][plane units]: input, F, T
    int groups0;                     // channel groups of 8 of the input (a multiple of 2: 14 for 101 channels); F and T have 8
    float* y; int yPlane;            // the trunk's result, fp32 planes [64][yPlane]
    unsigned* done;                  // [tiles] layers finished by each tile (zeroed before the launch)
    unsigned* error;                 // 1 + the layer at which a wait timed out (0: none)
    unsigned* absmax;                // range guard over every value stored (may be NULL)
    int H, W, plane, tilesX, tilesY, layers;
    ISR_DIAG_MEMBER(int, dbg, 0);                         // diagnostics (isrDebugSetTrunkAblation): 1 no MFMAs, 2 no activation DMA, 4 no stores, 8 no waits, 16 no weight DMA,
                                     // 32 the MFMAs on operands read once per k-step, 64 (with 32) ... but every tap's fragments read from LDS all the same
    ISR_DIAG_MEMBER(int, faultTile, -1);                   // diagnostics (isrDebugSetTrunkFault): the tile that never publishes, or -1
    int lastPacked;                  // isrSetTrunkPackedResult: the last layer ALSO writes its result packed-split (for sr_conv_upsp.h)
    unsigned long long timeoutTicks; // of the chip's 100 MHz clock
    ISR_DIAG_MEMBER(unsigned long long*, stamps, nullptr);      // diagnostics: [tile][layer][8] = ticks at: layer start | neighbours there | first k-step staged | MFMAs done | epilogue done | stores drained
};

// a wave-uniform pointer, held in scalar registers
template <typename T>
__device__ __forceinline__ T* trunk16_uniform(T* q)
{
    const unsigned long long v = (unsigned long long)q;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (T*)(((unsigned long long)hi << 32) | lo);
}

[[maybe_unused]] int g_trunk_dbg = 0;
[[maybe_unused]] unsigned long long* g_trunk_stamps = nullptr;
unsigned* g_trunk_error_word = nullptr;          // isrSetTrunkErrorWord: where launches report a timed-out wait (NULL: the workspace's own word)
[[maybe_unused]] int g_trunk_fault_tile = -1;                     // isrDebugSetTrunkFault: this tile never publishes its progress (tests of the timeout path)
unsigned long long g_trunk_timeout_ticks = 5000000ull;     // 50 ms of the chip's 100 MHz clock: a frame is 2 ms

typedef __attribute__((address_space(3))) char t16_lds_char;

// (No "memory" clobber on these: the requests land in LDS buffers nobody reads before the next barrier, which is preceded by an
// explicit s_waitcnt vmcnt(0) that does carry one; with it every request would pin the k-step's operand reads in place and the
// compiler could no longer fetch a tap's operands under the MFMAs of the tap before.)
// LDS-DMA with a SCALAR base and a 32-bit lane offset (the builtin only produces the 64-bit-lane-address form, which costs a 64-bit
// vector add per request): lane l's 16 bytes at base + voff land at LDS address ldsaddr + 16 l.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
// COHERENT: an agent-scope load (sc1): past this CU's L1 (which no other CU's store ever refreshes); the producers' `sc1` stores have
// dropped the line from their XCD's L2, so the bytes come from memory / the Infinity Cache.
template <bool COHERENT = false>
__device__ __forceinline__ void trunk16_dma16(const void* base, unsigned voff, unsigned ldsaddr)
{
    // (readfirstlane: a uniform address the compiler chose to compute on the vector side -- it multiplies the buffer index with a
    // constant it holds in a vector register for the fragment addresses -- otherwise reaches the "s" operand as a VGPR)
    ldsaddr = __builtin_amdgcn_readfirstlane(ldsaddr);
    if (COHERENT)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 sc1" :: "s"(ldsaddr), "v"(voff), "s"(base) : "m0");
    else
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(ldsaddr), "v"(voff), "s"(base) : "m0");
}
// ... for the lanes of `mask` only, WITHOUT a branch (the 4-row form: a branch per request makes every tap a basic block of its own
// and nothing covers the bookkeeping between them): EXEC is narrowed and put back inside the statement
template <bool COHERENT = true>
__device__ __forceinline__ void trunk16_dma16_masked(const void* base, unsigned voff, unsigned ldsaddr, unsigned long long mask)
{
    ldsaddr = __builtin_amdgcn_readfirstlane(ldsaddr);
    unsigned long long saved;
    if (COHERENT)
        asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %4\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3 sc1\n\ts_mov_b64 exec, %0"
                     : "=&s"(saved) : "s"(ldsaddr), "v"(voff), "s"(base), "s"(mask) : "m0");
    else
        asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %4\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b64 exec, %0"
                     : "=&s"(saved) : "s"(ldsaddr), "v"(voff), "s"(base), "s"(mask) : "m0");
}
// ... and 64 floats (lane l's dword to ldsaddr + 4 l)
__device__ __forceinline__ void trunk16_dma4(const void* base, unsigned voff, unsigned ldsaddr)
{
    ldsaddr = __builtin_amdgcn_readfirstlane(ldsaddr);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" :: "s"(ldsaddr), "v"(voff), "s"(base) : "m0");
}
#pragma clang diagnostic pop

// One k-step of MFMAs on this form's LDS geometry: the arithmetic of split_kstep (sr_split_common.h), tap for tap, product for
// product.  `between(tap)` runs after each tap's MFMAs are issued: the caller spreads its DMA requests over the k-step there, so
// that a request that has to queue for the memory pipeline does so behind 12 MFMAs of cover instead of in front of the k-step.
// ROWS: image rows of the tile per wave -- 2 (eight waves, two per SIMD) or 4 (four waves, one per SIMD: the same four weight
// fragments serve twice the MFMAs, i.e. 0.5 instead of 0.67 LDS fragment reads per MFMA, and a wave has the SIMD's 512 registers).
template <int ROWS>
struct Trunk16Operands { f16x8 a0h, a0l, a1h, a1l, bh[ROWS], bo[ROWS]; };

template <int ROWS>
__device__ __forceinline__ Trunk16Operands<ROWS> trunk16_operands(const u32x4* wl, const u32x4* bl, int tap)
{
    const int dy = tap / 3, dx = tap - dy * 3;
    Trunk16Operands<ROWS> o;
    o.a0h = __builtin_bit_cast(f16x8, wl[tap * 128]);
    o.a0l = __builtin_bit_cast(f16x8, wl[S_WPART + tap * 128]);
    o.a1h = __builtin_bit_cast(f16x8, wl[tap * 128 + 32]);
    o.a1l = __builtin_bit_cast(f16x8, wl[S_WPART + tap * 128 + 32]);
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        o.bh[r] = __builtin_bit_cast(f16x8, bl[(r + dy) * P16_W + dx]);
        o.bo[r] = __builtin_bit_cast(f16x8, bl[P16_PART + (r + dy) * P16_W + dx]);
    }
    return o;
}

struct Trunk16NoMid { __device__ __forceinline__ void operator()(int, int) const {} };

// `mid(tap, q)` (4-row form only) runs after the tap's product groups q = 0 and 1: straight-line DMA requests placed UNDER the tap's MFMAs
template <int ROWS, typename Between, typename Mid = Trunk16NoMid>
__device__ __forceinline__ void trunk16_kstep(f32x16 (&acc)[2][ROWS], const u32x4* wl, const u32x4* bl, Between between, Mid mid = Mid())
{
    // Software-pipelined over the taps: tap t + 1's eight operand fragments are requested BEFORE tap t's MFMAs are issued and are
    // consumed one basic block later (`between` branches, so every tap is a block of its own and the compiler's scheduler cannot
    // do this by itself: it issued a tap's reads at the top of the tap's block and waited for them with lgkmcnt(0), eight LDS
    // latencies per k-step and wave with only the other wave of the SIMD to cover them).  The fence keeps the requests in front.
    Trunk16Operands<ROWS> cur = trunk16_operands<ROWS>(wl, bl, 0);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        Trunk16Operands<ROWS> nxt = cur;
        if (tap < 8) {
            nxt = trunk16_operands<ROWS>(wl, bl, tap + 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        const f16x8 a0s = cur.a0h * (_Float16)0.00048828125f;               // w_hi 2^-11: partner of the scaled x_lo'
        const f16x8 a1s = cur.a1h * (_Float16)0.00048828125f;
        if (ROWS == 2) {
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                acc[0][r] = mfma16(cur.a0l, cur.bh[r], acc[0][r]);
                acc[0][r] = mfma16(a0s, cur.bo[r], acc[0][r]);
                acc[0][r] = mfma16(cur.a0h, cur.bh[r], acc[0][r]);
                mid(tap, r);
                acc[1][r] = mfma16(cur.a1l, cur.bh[r], acc[1][r]);
                acc[1][r] = mfma16(a1s, cur.bo[r], acc[1][r]);
                acc[1][r] = mfma16(cur.a1h, cur.bh[r], acc[1][r]);
            }
        } else {
            // one wave per SIMD: nobody else fills the pipe behind a dependent MFMA, so the three products of an accumulator are
            // issued eight MFMAs apart (product-major) -- per accumulator the same three products in the same order: same bits
#pragma unroll
            for (int r = 0; r < ROWS; ++r) { acc[0][r] = mfma16(cur.a0l, cur.bh[r], acc[0][r]); acc[1][r] = mfma16(cur.a1l, cur.bh[r], acc[1][r]); }
            mid(tap, 0);
#pragma unroll
            for (int r = 0; r < ROWS; ++r) { acc[0][r] = mfma16(a0s, cur.bo[r], acc[0][r]); acc[1][r] = mfma16(a1s, cur.bo[r], acc[1][r]); }
            mid(tap, 1);
#pragma unroll
            for (int r = 0; r < ROWS; ++r) { acc[0][r] = mfma16(cur.a0h, cur.bh[r], acc[0][r]); acc[1][r] = mfma16(cur.a1h, cur.bh[r], acc[1][r]); }
        }
        between(tap);
        cur = nxt;
    }
}

// DMA ROLES.  The 76 wave-wide requests of a k-step (40 patch pieces: 2 parts x 2 channel groups x 10 runs of 64 patch pixels; 36
// weight pieces: 2 parts x 9 taps x 2 halves) are dealt by kind: waves 0..3 fetch weights -- wave w the nine taps of (part w / 2,
// half w % 2): one source pointer and one LDS address that advance by a constant per tap -- and waves 4..7 fetch the patch -- wave
// 4 + q the ten runs of (part q / 2, group q % 2): one plane pointer per k-step and this table of ten lane offsets, the same for
// every k-step of every layer.  A SIMD holds waves w and w + 4, i.e. one of each kind.  (The first form dealt both kinds to every
// wave: twice the scalar bookkeeping per request and ~250 scalar registers spilled.)
struct Trunk16Lane {
    unsigned poff[P16_SUBS];         // byte offset of the pixel's unit inside a plane (padding: the plane's zero unit), per run
    unsigned live, centre;           // bit s: the lane takes part in run s / its pixel belongs to the tile's own 16 x 32 centre
};

__device__ __forceinline__ Trunk16Lane trunk16_lane_setup(const Trunk16Params& p, int oy0, int ox0, int lane)
{
    Trunk16Lane t;
    t.live = 0u; t.centre = 0u;
#pragma unroll
    for (int sub = 0; sub < P16_SUBS; ++sub) {
        const int off = sub * 64 + lane;
        const int r = off / P16_W, c = off - r * P16_W;
        const int iy = oy0 + r - 1, ix = ox0 + c - 1;
        t.poff[sub] = (unsigned)(((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) ? iy * p.W + ix : p.H * p.W) * 16u;
        if (off < P16_PIX) t.live |= 1u << sub;
        if (r >= 1 && r <= T16_H && c >= 1 && c <= T16_W) t.centre |= 1u << sub;
    }
    return t;
}

// Run `sub` (0 .. 9) of the patch wave's (part, group) of k-step ks into the patch buffer at LDS address pbuf.  HALO_ONLY: only the
// one-pixel halo (the centre is in LDS already).  Every activation byte is loaded `sc1` (agent scope: past this CU's L1) -- the
// halo because other CUs wrote it, the centre because this CU's L1 may still hold the line from two layers ago, when the same
// buffer held the previous block's tensor (MI355X_MICROARCH.md: every load of handed-off bytes must be such a load; the
// producer's `sc1` stores drop the line from L2 anyway, so nothing is lost).
template <bool HALO_ONLY>
__device__ __forceinline__ void trunk16_patch_run(int sub, const char* plane, unsigned pbufRole, const Trunk16Lane& t)
{
    const unsigned take = HALO_ONLY ? (t.live & ~t.centre) : t.live;
    if (take & (1u << sub)) trunk16_dma16<true>(plane, t.poff[sub], pbufRole + (unsigned)sub * 1024u);
}

// The epilogue of one layer, straight from the D layout: lane (j, h) holds pixel j, channels 32 cb + 8 gi + 4 h + e.
// KIND 0: F = relu(conv + b); 1: T = relu(conv + b); 2: F += conv + b.  LAST: the result goes to y (fp32 planes), else to the
// packed-split tensor `out` (write-through stores: other CUs read it after the publish).
// Not LAST: channels 0 .. 31 -- the next layer's first two k-steps -- go straight into the two patch buffers (pk0, pk1: LDS, this
// tile's centre; pixels outside the image as zeros) and to memory only where a neighbour will read them (the tile's outermost
// ring); channels 32 .. 63 go to memory whole and come back by DMA under those two k-steps.
template <int KIND, bool LAST, bool DIAG, int ROWS>
__device__ __forceinline__ void trunk16_epilogue(const Trunk16Params& p, f32x16 (&acc)[2][ROWS], f32x16 (&F)[2][ROWS], unsigned& mag, float unscale,
                                                 const float* biasl, const char* out, unsigned planeBytes, int oy0, int ox0, int wave, int j, int h,
                                                 u32x4* pk0, u32x4* pk1)
{
    const rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(LAST ? reinterpret_cast<const char*>(p.y) : out), 0,
                                                         LAST ? (int)((size_t)64 * p.yPlane * 4) : (int)(16u * planeBytes), 0x00020000);
    // LAST: the result ALSO goes out packed-split (into `out`, the F tensor of the workspace; plain stores: the reader is the next
    // kernel) -- the phase-decomposed upsampling layer stages it by LDS-DMA (sr_conv_upsp.h)
    const rsrc_t qrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(out), 0, (int)(16u * planeBytes), 0x00020000);
    const int ox = ox0 + j;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const int oy = oy0 + wave * ROWS + r;
        const bool inside = oy < p.H && ox < p.W && !(DIAG && (p.dbg & 4));
        const unsigned voff = !inside ? BAD_OFFSET : LAST ? (unsigned)(oy * p.W + ox + 4 * h * p.yPlane) * 4u      // the lane half's 4 channels: in the lane offset
                                                          : (unsigned)(oy * p.W + ox) * 16u + 8u * (unsigned)h;
        const int row = wave * ROWS + r;
        const bool ring = row == 0 || row == T16_H - 1 || j == 0 || j == T16_W - 1;
        const unsigned voffRing = ring ? voff : BAD_OFFSET;
        const bool live = oy < p.H && ox < p.W;                              // (the stores' ablation switch does not touch the LDS hand-over)
        u32x2* const c0 = reinterpret_cast<u32x2*>(pk0 + (row + 1) * P16_W + j + 1) + h;     // this lane's 8 bytes of the pixel's unit
        u32x2* const c1 = reinterpret_cast<u32x2*>(pk1 + (row + 1) * P16_W + j + 1) + h;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
#pragma unroll
            for (int gi = 0; gi < 4; ++gi) {
                const float4 b4 = *reinterpret_cast<const float4*>(biasl + cb * 32 + 8 * gi + 4 * h);
                const float bv[4] = {b4.x, b4.y, b4.z, b4.w};
                f16x4 th, tl;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = acc[cb][r][4 * gi + e] * unscale + bv[e];
                    if (KIND != 2) v = v > 0.f ? v : 0.f;
                    else v += F[cb][r][4 * gi + e];
                    if (KIND != 1) F[cb][r][4 * gi + e] = v;
                    if (inside) mag = isr_umax(mag, isr_mag(v));
                    if (LAST) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), prs, (int)voff, (cb * 32 + 8 * gi + e) * p.yPlane * 4, 0);
                    _Float16 a, b;
                    split16x(v, a, b);
                    th[e] = a; tl[e] = b;
                }
                if (LAST) {
                    const int g = cb * 4 + gi;
                    const unsigned vq = (inside && p.lastPacked) ? (unsigned)(oy * p.W + ox) * 16u + 8u * (unsigned)h : BAD_OFFSET;
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, th), qrs, (int)vq, g * planeBytes, 0);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, tl), qrs, (int)vq, (8 + g) * planeBytes, 0);
                } else {
                    const int g = cb * 4 + gi;
                    const unsigned vo = cb == 0 ? voffRing : voff;
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, th), prs, (int)vo, g * planeBytes, 16);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, tl), prs, (int)vo, (8 + g) * planeBytes, 16);
                    if (cb == 0) {
                        const u32x2 zero2 = {0u, 0u};
                        u32x2* const c = (gi < 2 ? c0 : c1) + (gi & 1) * (P16_PIX * 2);          // channel group of the k-step; lo' at + P16_PART units
                        c[0] = live ? __builtin_bit_cast(u32x2, th) : zero2;
                        c[P16_PART * 2] = live ? __builtin_bit_cast(u32x2, tl) : zero2;
                    }
                }
            }
        }
    }
}

// DIAG: the instantiation with the diagnostics (ablation switches, phase stamps) compiled in; the product launch is the one without:
// the 108 unrolled MFMAs of the "operands read once" path and its nine copies of the DMA bookkeeping made the kernel's code large
// enough to slow the real path down (measured when a second such path was added: 636 -> 755 us).
template <bool DIAG, int ROWS, bool LINE = (ROWS == 4)>
__global__ __launch_bounds__(64 * T16_H / ROWS) void trunk_dataflow_kernel(const Trunk16Params p)
{
    constexpr int WAVES = T16_H / ROWS;                                      // 8 or 4
    const int dbg = DIAG ? p.dbg : 0;
    extern __shared__ u32x4 lds[];
    u32x4* const pbuf0 = lds;                                                // patch buffers at + P16_UNITS
    u32x4* const wbuf0 = lds + 2 * P16_UNITS;                                // weight buffers at + S_WUNITS
    float* const biasl0 = reinterpret_cast<float*>(lds + T16_LDS_UNITS);     // two bias buffers of 64 floats
    int* const flags = reinterpret_cast<int*>(biasl0 + 128);
    const unsigned ldsBase = (unsigned)(uintptr_t)(t16_lds_char*)lds;        // LDS byte addresses for the DMA
    const unsigned pAddr = ldsBase, wAddr = ldsBase + 2 * P16_UNITS * 16, bAddr = ldsBase + T16_LDS_UNITS * 16;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // in a scalar register: everything derived from it is scalar work
    const int j = lane & 31, h = lane >> 5;
    const int ntiles = p.tilesX * p.tilesY;
    int tile;
    {   // an XCD (= an L2) gets a contiguous range of tiles: neighbours share halo lines
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    }
    if (tile >= ntiles) return;
    const int tx = tile % p.tilesX, ty = tile / p.tilesX;
    const int oy0 = ty * T16_H, ox0 = tx * T16_W;
    const unsigned long long deadline = __builtin_amdgcn_s_memrealtime() + p.timeoutTicks;
    const Trunk16Lane lanes = trunk16_lane_setup(p, oy0, ox0, lane);

    f32x16 F[2][ROWS];                                                       // the residual stream of this lane's ROWS rows x 64 channels
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int r = 0; r < ROWS; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) F[cb][r][i] = 0.0f;
    unsigned mag = 0u;
    int gk = 0;                                                              // k-steps done so far, all layers: k-step gk uses buffers gk & 1

    const char* const ws = trunk16_uniform(p.ws);
    const unsigned planeBytes = (unsigned)p.plane * 16u;                     // H W units + the plane's zero unit, rounded up to whole cache lines
    const bool dmaX = !(dbg & 2), dmaW = !(dbg & 16);
    const u32x4* wq = trunk16_uniform(p.wq[0]);
    // bias of layer l into bias buffer l & 1: one dword-wide DMA (64 floats) by the last wave; a layer without bias reads zeros
    auto stage_bias = [&](int l, const float* b) {
        if (wave != WAVES - 1) return;
        if (b) trunk16_dma4(b, (unsigned)lane * 4u, bAddr + (unsigned)(l & 1) * 256u);
        else biasl0[(l & 1) * 64 + lane] = 0.0f;
    };
    // this wave's role (see Trunk16Lane): weights (waves 0..3: part, half) or patch (waves 4..7: part, group); with four waves
    // every wave has one role of each kind
    const bool wrole = WAVES == 4 || wave < 4, prole = WAVES == 4 || wave >= 4;
    const int rpart = (wave & 3) >> 1, rsel = wave & 1;
    const unsigned wdstRole = (unsigned)(rpart * S_WPART + rsel * 64) * 16u;             // + the weight buffer + tap * 2048
    const unsigned pdstRole = (unsigned)((rpart * 2 + rsel) * P16_PIX) * 16u;            // + the patch buffer + run * 1024
    const unsigned wlane = (unsigned)lane * 16u;
    // tap `tap` of this wave's (part, half) of k-step ks of a weight image, into the weight buffer at LDS address wbuf
    auto weight_tap = [&](int tap, const u32x4* image, int ksteps_, int ks, unsigned wbuf) {
        trunk16_dma16(image + 1 + (size_t)(tap * ksteps_ + ks) * 256 + rpart * 128 + rsel * 64, wlane, wbuf + wdstRole + (unsigned)tap * 2048u);
    };
    auto patch_plane = [&](const char* tensor, int groups_, int ks) {
        return tensor + (size_t)(rpart * groups_ + 2 * ks + rsel) * planeBytes;
    };
    // the first layer's first k-step: nothing to wait for
    stage_bias(0, trunk16_uniform(p.bias[0]));
    if (wrole && dmaW) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) weight_tap(tap, wq, p.groups0 >> 1, 0, wAddr);
    }
    if (prole && dmaX) {
        const char* const plane = patch_plane(ws + p.xpsOff, p.groups0, 0);
#pragma unroll
        for (int sub = 0; sub < P16_SUBS; ++sub) trunk16_patch_run<false>(sub, plane, pAddr + pdstRole, lanes);
    }

#pragma unroll 1
    for (int l = 0; l < p.layers; ++l) {
        const int kind = l == 0 ? 0 : (l & 1) ? 1 : 2;
        const int groups = l == 0 ? p.groups0 : 8, ksteps = groups >> 1;
        const char* const tin = ws + (l == 0 ? p.xpsOff : (l & 1) ? p.fpsOff : p.tpsOff);
        const char* const tout = ws + ((l & 1) ? p.tpsOff : p.fpsOff);       // this layer's output = the next layer's input
        const bool last = l + 1 == p.layers;
        const u32x4* const wqNext = last ? nullptr : trunk16_uniform(p.wq[l + 1]);
        // (fetched here, not where it is used: a scalar load inside the k-loop makes every LDS wait behind it an lgkmcnt(0))
        const float* const biasNext = last ? nullptr : trunk16_uniform(p.bias[l + 1]);
        // diagnostics: raw ticks at the phase boundaries, straight to memory (no registers held for it)
        auto lap = [&](int slot) {
            if (DIAG && p.stamps && tid == 0) p.stamps[((size_t)tile * p.layers + l) * 8 + slot] = __builtin_amdgcn_s_memrealtime();
        };
        lap(0);
        // ---- wait for the 3 x 3 neighbourhood to have finished layer l - 1, then fetch the halo of the first k-step ---------------
        if (l > 0) {
            if (!(dbg & 8)) {
                if (tid == 0) flags[0] = 0;
                __syncthreads();
                if (tid < 9 && tid != 4) {
                    const int ny = ty + tid / 3 - 1, nx = tx + tid % 3 - 1;
                    if ((unsigned)ny < (unsigned)p.tilesY && (unsigned)nx < (unsigned)p.tilesX) {
                        const unsigned* f = p.done + ny * p.tilesX + nx;
                        while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)l) {
                            __builtin_amdgcn_s_sleep(2);
                            if (__builtin_amdgcn_s_memrealtime() > deadline) { flags[0] = 1; break; }
                        }
                    }
                }
                __syncthreads();
                if (flags[0]) {                                              // a neighbour never arrived: give up, loudly
                    if (tid == 0) atomicMax(p.error, (unsigned)(1 + l));
                    return;
                }
            }
            lap(1);
            if (dmaX && prole) {
                const char* const plane = patch_plane(tin, groups, 0);
#pragma unroll
                for (int sub = 0; sub < P16_SUBS; ++sub) trunk16_patch_run<true>(sub, plane, pAddr + (unsigned)(gk & 1) * (P16_UNITS * 16) + pdstRole, lanes);
            }
        }
        f32x16 acc[2][ROWS];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < ROWS; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[cb][r][i] = 0.0f;
#pragma unroll 1
        for (int ks = 0; ks < ksteps; ++ks, ++gk) {
            // k-step gk's patch and weights have landed (this wave's DMA: the explicit vmcnt(0) -- the compiler does not count the
            // requests issued from inline assembly; everyone's: the barrier), and every wave is done with k-step gk - 1: the other
            // buffers are free
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (ks == 0) lap(2);
            const int cur = gk & 1, nxt = cur ^ 1;
            const u32x4* const pcur = pbuf0 + cur * P16_UNITS;
            const u32x4* const wcur = wbuf0 + cur * S_WUNITS;
            const unsigned pnxt = pAddr + (unsigned)nxt * (P16_UNITS * 16), wnxt = wAddr + (unsigned)nxt * (S_WUNITS * 16);
            const bool more = ks + 1 < ksteps;
            // the next k-step's weights (weight waves: one tap's piece after each tap's MFMAs) and patch (patch waves: T16_EARLY_PATCH
            // runs after each of the first taps) -- or, under the layer's last k-step, the next layer's first weights and its bias
            const u32x4* const wimg = more ? wq : wqNext;
            const int wks = more ? ks + 1 : 0, wksteps = more ? ksteps : 4;
            const bool wany = dmaW && (more || !last), pany = dmaX && more;
            const bool haloOnly = l > 0 && ks == 0;                          // k-step 1 of a layer fed by this kernel: its centre is in LDS
            const char* const pplane = patch_plane(tin, groups, ks + 1);
            auto between = [&](int tap) {
                if (wrole) {
                    if (wany) weight_tap(tap, wimg, wksteps, wks, wnxt);
                }
                if (prole) {
                    if (pany) {
                        // T16_EARLY_PATCH runs after each of the first taps (5: all ten behind taps 0 and 1) instead of one per tap and two
                        // after the last: the activation bytes come over the fabric (microseconds), and what is requested behind the last
                        // tap has ~0.5 us until the k-step's barrier waits for it.  In the frame, trunk: 0.59-0.61 ms with one run per tap,
                        // 0.58-0.60 with 2 or 3, 0.57-0.59 with 5, 0.57-0.60 with all ten behind tap 0 (profiles/r05_trunk_rows.md)
                        if (T16_EARLY_PATCH) {                          // (= runs per tap)
#pragma unroll
                            for (int q = 0; q < T16_EARLY_PATCH; ++q) {
                                const int run = T16_EARLY_PATCH * tap + q;
                                if (run < P16_SUBS) {
                                    if (haloOnly) trunk16_patch_run<true>(run, pplane, pnxt + pdstRole, lanes);
                                    else trunk16_patch_run<false>(run, pplane, pnxt + pdstRole, lanes);
                                }
                            }
                        } else if (haloOnly) {
                            trunk16_patch_run<true>(tap, pplane, pnxt + pdstRole, lanes);
                            if (tap == 8) trunk16_patch_run<true>(9, pplane, pnxt + pdstRole, lanes);
                        } else {
                            trunk16_patch_run<false>(tap, pplane, pnxt + pdstRole, lanes);
                            if (tap == 8) trunk16_patch_run<false>(9, pplane, pnxt + pdstRole, lanes);
                        }
                    } else if (!more && !last && tap == 0) stage_bias(l + 1, biasNext);
                }
            };
            if (ROWS == 2 && (dbg & 32)) {                                  // diagnostics: the k-step's 108 MFMAs on operands read once (eight-wave form only: registers)
                const f16x8 a = __builtin_bit_cast(f16x8, wcur[h * 64 + j]), b = __builtin_bit_cast(f16x8, pcur[h * P16_PIX + j]);
                if (dbg & 64) {                                            // ... with every tap's eight fragments read from LDS all the same (and dropped)
#pragma unroll
                    for (int t = 0; t < 9; ++t) {
                        const Trunk16Operands<ROWS> o = trunk16_operands<ROWS>(wcur + h * 64 + j, pcur + h * P16_PIX + (wave * ROWS) * P16_W + j, t);
                        asm volatile("" :: "v"(o.a0h), "v"(o.a0l), "v"(o.a1h), "v"(o.a1l));
#pragma unroll
                        for (int r = 0; r < ROWS; ++r) asm volatile("" :: "v"(o.bh[r]), "v"(o.bo[r]));
#pragma unroll
                        for (int q = 0; q < 3; ++q)
#pragma unroll
                            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                                for (int r = 0; r < ROWS; ++r) acc[cb][r] = mfma16(a, b, acc[cb][r]);
                        between(t);
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < 27; ++t)
#pragma unroll
                        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                            for (int r = 0; r < ROWS; ++r) acc[cb][r] = mfma16(a, b, acc[cb][r]);
#pragma unroll
                    for (int t = 0; t < 9; ++t) between(t);
                }
            } else if (LINE && !(dbg & 1)) {
                // Four waves, one per SIMD: no other wave covers a tap's DMA bookkeeping, so it is straight-line code under the tap's MFMAs --
                // the weight piece after the first product group, the patch run after the second, each switched off by an empty lane mask
                // instead of a branch: the k-step is ONE basic block.  (The same code serves the 2-row form -- LINE = true there measured
                // within 1 % of its branching form: two waves per SIMD cover each other's bookkeeping anyway.)
                if (!more && !last) stage_bias(l + 1, biasNext);
                const u32x4* const wsrc = (wany ? wimg : wq) + 1 + (size_t)wks * 256 + rpart * 128 + rsel * 64;
                const size_t wtap = (size_t)(wany ? wksteps : ksteps) * 256;
                const unsigned wdst = wnxt + wdstRole;
                const unsigned takeBits = (l > 0 && ks == 0) ? (lanes.live & ~lanes.centre) : lanes.live;
                const unsigned long long pmask = __builtin_amdgcn_ballot_w64(pany && prole), wmask = __builtin_amdgcn_ballot_w64(wany && wrole);   // all lanes or none
                auto mid = [&](int tap, int q) {
                    if (q == 0) trunk16_dma16_masked<false>(wsrc + tap * wtap, wlane, wdst + (unsigned)tap * 2048u, wmask);
                    else {
#pragma unroll
                        for (int qq = 0; qq < T16_EARLY_PATCH; ++qq) {          // the patch runs early in the k-step, as in the eight-wave form
                            const int run = T16_EARLY_PATCH * tap + qq;
                            if (run < P16_SUBS)
                                trunk16_dma16_masked(pplane, lanes.poff[run], pnxt + pdstRole + (unsigned)run * 1024u, __builtin_amdgcn_ballot_w64((takeBits >> run) & 1u) & pmask);
                        }
                    }
                };
                trunk16_kstep<ROWS>(acc, wcur + h * 64 + j, pcur + h * P16_PIX + (wave * ROWS) * P16_W + j, [](int) {}, mid);
            } else if (!(dbg & 1)) {
                trunk16_kstep<ROWS>(acc, wcur + h * 64 + j, pcur + h * P16_PIX + (wave * ROWS) * P16_W + j, between);
            } else {
#pragma unroll
                for (int t = 0; t < 9; ++t) between(t);
            }
        }
        __syncthreads();                                                     // both patch buffers are free: the epilogue fills them
        lap(3);
        {
            const float unscale = reinterpret_cast<const float*>(wq)[1];
            const float* const biasl = biasl0 + (l & 1) * 64;
            u32x4* const pk0 = pbuf0 + (gk & 1) * P16_UNITS;                 // the next layer's k-steps 0 and 1
            u32x4* const pk1 = pbuf0 + ((gk + 1) & 1) * P16_UNITS;
            if (last) {
                if (kind == 0) trunk16_epilogue<0, true, DIAG, ROWS>(p, acc, F, mag, unscale, biasl, tout, planeBytes, oy0, ox0, wave, j, h, pk0, pk1);
                else trunk16_epilogue<2, true, DIAG, ROWS>(p, acc, F, mag, unscale, biasl, tout, planeBytes, oy0, ox0, wave, j, h, pk0, pk1);
            } else if (kind == 0) trunk16_epilogue<0, false, DIAG, ROWS>(p, acc, F, mag, unscale, biasl, tout, planeBytes, oy0, ox0, wave, j, h, pk0, pk1);
            else if (kind == 1) trunk16_epilogue<1, false, DIAG, ROWS>(p, acc, F, mag, unscale, biasl, tout, planeBytes, oy0, ox0, wave, j, h, pk0, pk1);
            else trunk16_epilogue<2, false, DIAG, ROWS>(p, acc, F, mag, unscale, biasl, tout, planeBytes, oy0, ox0, wave, j, h, pk0, pk1);
        }
        lap(4);
        // ---- publish: every wave's stores drained, then one agent-scope store of the tile's progress -------------------------
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0 && tile != p.faultTile) __hip_atomic_store(p.done + tile, (unsigned)(l + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        wq = wqNext;
        lap(5);
    }
    isr_range_note(p.absmax, mag);
}

// ---- the same trunk for images of MORE tiles than CUs: trunk_mt_kernel ---------------------------------------------------------
// One persistent workgroup per CU again, but workgroup g now owns tiles g, g + G, g + 2 G, ... (G workgroups) and walks
// (layer 0: its tiles in order), (layer 1: its tiles), ...: at any time the G tiles in flight form a band of consecutive tile rows, the
// neighbours a tile waits for are tiles of the SAME round of other workgroups one layer back, and the launch cannot lock up: a task
// (layer l, tile t) waits only for tasks of layer l - 1, every workgroup runs its tasks in (layer, tile) order, so by induction over
// l every task is reached.  Nothing of a tile stays in the CU between its layers, so
//   * every k-step's whole 18 x 34 patch comes by LDS-DMA (no centre hand-over through LDS),
//   * all 64 output channels go to memory packed-split (not just the ring and channels 32 .. 63),
//   * the residual stream F lives in the RESULT tensor y (fp32): a block's second convolution reads y, adds and writes it back --
//     the same fp32 values the one-tile form keeps in registers -- with agent-scope loads (the line may sit in this CU's L1 from
//     the block before).
// Same products, same order, same epilogue arithmetic as trunk_dataflow_kernel and the per-layer kernels: bit-identical output.
// Per tile and layer ~29 us (MFMA phase 22 as the one-tile form, + the first k-step's staging exposed) against ~39 for the
// per-layer launches of sr_conv_split.hip at 960 x 540.
template <int KIND, bool LAST>
__device__ __forceinline__ void trunk_mt_epilogue(const Trunk16Params& p, f32x16 (&acc)[2][2], unsigned& mag, float unscale, const float* biasl,
                                                  const char* out, unsigned planeBytes, int oy0, int ox0, int wave, int j, int h)
{
    const rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(out), 0, (int)(16u * planeBytes), 0x00020000);
    const rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(p.y), 0, (int)((size_t)64 * p.yPlane * 4), 0x00020000);
    const int ox = ox0 + j;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int oy = oy0 + wave * 2 + r;
        const bool inside = oy < p.H && ox < p.W;
        const unsigned voff = !inside ? BAD_OFFSET : (unsigned)(oy * p.W + ox) * 16u + 8u * (unsigned)h;
        const unsigned yoff = !inside ? BAD_OFFSET : (unsigned)(oy * p.W + ox + 4 * h * p.yPlane) * 4u;
        float fold[2][16];
        if (KIND == 2) {                                                     // F of this lane's pixel: all 32 loads first, one latency
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    fold[cb][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(yrs, (int)yoff, (cb * 32 + 8 * (i >> 2) + (i & 3)) * p.yPlane * 4, 16));
        }
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
#pragma unroll
            for (int gi = 0; gi < 4; ++gi) {
                const float4 b4 = *reinterpret_cast<const float4*>(biasl + cb * 32 + 8 * gi + 4 * h);
                const float bv[4] = {b4.x, b4.y, b4.z, b4.w};
                f16x4 th, tl;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = acc[cb][r][4 * gi + e] * unscale + bv[e];
                    if (KIND != 2) v = v > 0.f ? v : 0.f;
                    else v += fold[cb][4 * gi + e];
                    if (inside) mag = isr_umax(mag, isr_mag(v));
                    if (KIND != 1) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), yrs, (int)yoff, (cb * 32 + 8 * gi + e) * p.yPlane * 4, LAST ? 0 : 16);
                    _Float16 a, b;
                    split16x(v, a, b);
                    th[e] = a; tl[e] = b;
                }
                {   // (LAST: the packed-split copy of the result is the phase-decomposed upsampling layer's input -- on request; plain stores)
                    const int g = cb * 4 + gi;
                    const unsigned vq = (LAST && !p.lastPacked) ? BAD_OFFSET : voff;
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, th), prs, (int)vq, g * planeBytes, LAST ? 0 : 16);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, tl), prs, (int)vq, (8 + g) * planeBytes, LAST ? 0 : 16);
                }
            }
        }
    }
}

__global__ __launch_bounds__(T16_THREADS) void trunk_mt_kernel(const Trunk16Params p)
{
    extern __shared__ u32x4 lds[];
    u32x4* const pbuf0 = lds;
    u32x4* const wbuf0 = lds + 2 * P16_UNITS;
    float* const biasl0 = reinterpret_cast<float*>(lds + T16_LDS_UNITS);
    int* const flags = reinterpret_cast<int*>(biasl0 + 128);
    const unsigned ldsBase = (unsigned)(uintptr_t)(t16_lds_char*)lds;
    const unsigned pAddr = ldsBase, wAddr = ldsBase + 2 * P16_UNITS * 16, bAddr = ldsBase + T16_LDS_UNITS * 16;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int ntiles = p.tilesX * p.tilesY;
    const int nwg = gridDim.x;
    int slot;
    {   // an XCD (= an L2) gets a contiguous range of every round's tiles: neighbours share halo lines
        const int q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7;
        slot = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    }
    if (slot >= ntiles) return;
    // The deadline is PER WAIT and grows with the rounds a layer takes: a healthy launch of many rounds (65 535 tiles are admitted:
    // ~170 ms) outlives any fixed limit counted from its start, while a single wait never lasts longer than the slowest workgroup's
    // pass over ONE layer (rounds x ~31 us).  timeoutTicks x rounds per wait keeps the 1 600-fold margin of the one-round form.
    const unsigned long long waitTicks = p.timeoutTicks * (unsigned long long)((ntiles + nwg - 1) / nwg);
    unsigned mag = 0u;
    int gk = 0;
    const char* const ws = trunk16_uniform(p.ws);
    const unsigned planeBytes = (unsigned)p.plane * 16u;
    const bool wrole = wave < 4;
    const int rpart = (wave & 3) >> 1, rsel = wave & 1;
    const unsigned wdstRole = (unsigned)(rpart * S_WPART + rsel * 64) * 16u;
    const unsigned pdstRole = (unsigned)((rpart * 2 + rsel) * P16_PIX) * 16u;
    const unsigned wlane = (unsigned)lane * 16u;
    auto weight_tap = [&](int tap, const u32x4* image, int ksteps_, int ks, unsigned wbuf) {
        trunk16_dma16(image + 1 + (size_t)(tap * ksteps_ + ks) * 256 + rpart * 128 + rsel * 64, wlane, wbuf + wdstRole + (unsigned)tap * 2048u);
    };
    auto patch_plane = [&](const char* tensor, int groups_, int ks) {
        return tensor + (size_t)(rpart * groups_ + 2 * ks + rsel) * planeBytes;
    };
#pragma unroll 1
    for (int l = 0; l < p.layers; ++l) {
        const int kind = l == 0 ? 0 : (l & 1) ? 1 : 2;
        const int groups = l == 0 ? p.groups0 : 8, ksteps = groups >> 1;
        const char* const tin = ws + (l == 0 ? p.xpsOff : (l & 1) ? p.fpsOff : p.tpsOff);
        const char* const tout = ws + ((l & 1) ? p.tpsOff : p.fpsOff);
        const bool last = l + 1 == p.layers;
        const u32x4* const wq = trunk16_uniform(p.wq[l]);
        const float* const bias = trunk16_uniform(p.bias[l]);
        const float unscale = reinterpret_cast<const float*>(wq)[1];
        bool staged = false;                                                 // this tile's first k-step is on its way already (fetched under the previous tile's last)
        int owed = -1;                                                       // a tile whose stores are out but whose progress is not published yet
        Trunk16Lane lanes = trunk16_lane_setup(p, (slot / p.tilesX) * T16_H, (slot % p.tilesX) * T16_W, lane);
#pragma unroll 1
        for (int tile = slot; tile < ntiles; tile += nwg) {
            const int tx = tile % p.tilesX, ty = tile / p.tilesX;
            const int oy0 = ty * T16_H, ox0 = tx * T16_W;
            const int tnext = tile + nwg;                                    // this workgroup's next tile of the layer
            const int txn = tnext % p.tilesX, tyn = tnext / p.tilesX;
            // ---- wait for the 3 x 3 neighbourhood (itself included: its own layer l - 1 wrote what this layer reads) to be there;
            //      and look (without waiting) whether the NEXT tile's neighbourhood is there as well: then its first k-step travels
            //      under this tile's last one instead of in front of its own ----------------------------------------------------
            bool ahead = tnext < ntiles;
            if (l > 0) {
                if (tid == 0) { flags[0] = 0; flags[1] = 0; }
                __syncthreads();
                if (tid < 9) {
                    const int ny = ty + tid / 3 - 1, nx = tx + tid % 3 - 1;
                    if ((unsigned)ny < (unsigned)p.tilesY && (unsigned)nx < (unsigned)p.tilesX) {
                        const unsigned* f = p.done + ny * p.tilesX + nx;
                        if (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)l) {
                            const unsigned long long deadline = __builtin_amdgcn_s_memrealtime() + waitTicks;       // armed when THIS wait begins
                            do {
                                __builtin_amdgcn_s_sleep(2);
                                if (__builtin_amdgcn_s_memrealtime() > deadline) { flags[0] = 1; break; }
                            } while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)l);
                        }
                    }
                } else if (tid >= 64 && tid < 73 && ahead) {
                    const int q = tid - 64, ny = tyn + q / 3 - 1, nx = txn + q % 3 - 1;
                    if ((unsigned)ny < (unsigned)p.tilesY && (unsigned)nx < (unsigned)p.tilesX &&
                        __hip_atomic_load(p.done + ny * p.tilesX + nx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)l) flags[1] = 1;
                }
                __syncthreads();
                if (flags[0]) {
                    if (tid == 0) atomicMax(p.error, (unsigned)(1 + l));
                    return;
                }
                ahead = ahead && !__builtin_amdgcn_readfirstlane(flags[1]);  // (provably uniform: the DMA addresses chosen under it stay scalar)
            }
            const Trunk16Lane lanesNext = trunk16_lane_setup(p, tyn * T16_H, txn * T16_W, lane);
            // ---- the tile's first k-step: weights, bias (the layer's first tile), whole patch ------------------------------------
            if (!staged) {
                if (tile == slot && wave == T16_WAVES - 1) {
                    if (bias) trunk16_dma4(bias, (unsigned)lane * 4u, bAddr);
                    else biasl0[lane] = 0.0f;
                }
                if (wrole) {
#pragma unroll
                    for (int tap = 0; tap < 9; ++tap) weight_tap(tap, wq, ksteps, 0, wAddr + (unsigned)(gk & 1) * (S_WUNITS * 16));
                } else {
                    const char* const plane = patch_plane(tin, groups, 0);
#pragma unroll
                    for (int sub = 0; sub < P16_SUBS; ++sub) trunk16_patch_run<false>(sub, plane, pAddr + (unsigned)(gk & 1) * (P16_UNITS * 16) + pdstRole, lanes);
                }
            }
            f32x16 acc[2][2];
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[cb][r][i] = 0.0f;
#pragma unroll 1
            for (int ks = 0; ks < ksteps; ++ks, ++gk) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (owed >= 0) {                                             // the previous tile's stores have drained with this k-step's operands: publish it now
                    if (tid == 0 && owed != p.faultTile) __hip_atomic_store(p.done + owed, (unsigned)(l + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    owed = -1;
                }
                const int cur = gk & 1, nxt = cur ^ 1;
                const u32x4* const pcur = pbuf0 + cur * P16_UNITS;
                const u32x4* const wcur = wbuf0 + cur * S_WUNITS;
                const unsigned pnxt = pAddr + (unsigned)nxt * (P16_UNITS * 16), wnxt = wAddr + (unsigned)nxt * (S_WUNITS * 16);
                const bool more = ks + 1 < ksteps;
                const char* const pplane = patch_plane(tin, groups, more ? ks + 1 : 0);
                auto between = [&](int tap) {
                    if (more) {
                        if (wrole) weight_tap(tap, wq, ksteps, ks + 1, wnxt);
                        else {
#pragma unroll
                            for (int q = 0; q < T16_EARLY_PATCH; ++q)
                                if (T16_EARLY_PATCH * tap + q < P16_SUBS) trunk16_patch_run<false>(T16_EARLY_PATCH * tap + q, pplane, pnxt + pdstRole, lanes);
                        }
                    } else if (ahead) {                                        // the next tile's first k-step (same layer: same weights image, same bias)
                        if (wrole) weight_tap(tap, wq, ksteps, 0, wnxt);
                        else {
#pragma unroll
                            for (int q = 0; q < T16_EARLY_PATCH; ++q)
                                if (T16_EARLY_PATCH * tap + q < P16_SUBS) trunk16_patch_run<false>(T16_EARLY_PATCH * tap + q, pplane, pnxt + pdstRole, lanesNext);
                        }
                    }
                };
                trunk16_kstep<2>(acc, wcur + h * 64 + j, pcur + h * P16_PIX + (wave * 2) * P16_W + j, between);
            }
            __syncthreads();                                                 // every wave is done with the last k-step's buffers (and the bias row is read below)
            if (last) {
                if (kind == 0) trunk_mt_epilogue<0, true>(p, acc, mag, unscale, biasl0, tout, planeBytes, oy0, ox0, wave, j, h);
                else trunk_mt_epilogue<2, true>(p, acc, mag, unscale, biasl0, tout, planeBytes, oy0, ox0, wave, j, h);
            } else if (kind == 0) trunk_mt_epilogue<0, false>(p, acc, mag, unscale, biasl0, tout, planeBytes, oy0, ox0, wave, j, h);
            else if (kind == 1) trunk_mt_epilogue<1, false>(p, acc, mag, unscale, biasl0, tout, planeBytes, oy0, ox0, wave, j, h);
            else trunk_mt_epilogue<2, false>(p, acc, mag, unscale, biasl0, tout, planeBytes, oy0, ox0, wave, j, h);
            // ---- publish: every wave's stores drained, then one agent-scope store of the tile's progress.  Where this workgroup has
            //      another tile of the layer to do, the drain is the next tile's first k-step barrier (the stores land under its
            //      operands' flight) and the publish follows it: its consumers are a whole round behind.  The layer's LAST tile of the
            //      workgroup is published at once: the next layer's first tiles everywhere are waiting for exactly these. ----------
            if (tnext < ntiles) owed = tile;
            else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0 && tile != p.faultTile) __hip_atomic_store(p.done + tile, (unsigned)(l + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            staged = ahead;
            lanes = lanesNext;
        }
    }
    isr_range_note(p.absmax, mag);
}

// x fp32 planes [cin][xPlane] -> packed-split [hi | lo][groups][npix + 1 units] (channels beyond cin are zero), and the zero unit
// that ends every plane of the three packed-split tensors of the launch (the padding pixels of the LDS-DMA staging read it)
__global__ __launch_bounds__(256) void trunk_pack_input_kernel(const float* __restrict__ x, int cin, long long xPlane, int npix, u32x4* __restrict__ ps,
                                                               int groups, u32x4* __restrict__ fps, u32x4* __restrict__ tps,
                                                               unsigned* __restrict__ done, int ntiles)
{
    const int pix = blockIdx.x * 256 + threadIdx.x, g = blockIdx.y;
    // ... and the tiles' progress counters back to zero for the launch that follows (a hipMemsetAsync of 1 036 bytes was two fill
    // kernels and two more kernel boundaries per frame)
    if (blockIdx.x == 0 && g == 0)
        for (int i = threadIdx.x; i < ntiles; i += 256) done[i] = 0u;
    const size_t plane = ((size_t)npix + 8) & ~(size_t)7;
    const u32x4 zero = {0u, 0u, 0u, 0u};
    if (blockIdx.x == 0 && threadIdx.x < 2) {
        if (g < groups) ps[(size_t)(threadIdx.x * groups + g) * plane + npix] = zero;
        if (g < 8) {
            fps[(size_t)(threadIdx.x * 8 + g) * plane + npix] = zero;
            tps[(size_t)(threadIdx.x * 8 + g) * plane + npix] = zero;
        }
    }
    if (pix >= npix || g >= groups) return;
    f16x8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = g * 8 + e;
        const float v = c < cin ? x[(size_t)c * xPlane + pix] : 0.0f;
        _Float16 a, b;
        split16x(v, a, b);
        hi[e] = a; lo[e] = b;
    }
    ps[(size_t)g * plane + pix] = __builtin_bit_cast(u32x4, hi);
    ps[(size_t)(groups + g) * plane + pix] = __builtin_bit_cast(u32x4, lo);
}

inline long long align256(long long v) { return (v + 255) & ~255LL; }

struct Trunk16Layout { long long header, xps, fps, tps, total; int tiles, groups0; };

Trunk16Layout trunk16_layout(int cin0, int H, int W)
{
    Trunk16Layout o;
    o.tiles = ((W + T16_W - 1) / T16_W) * ((H + T16_H - 1) / T16_H);
    o.groups0 = 2 * ((cin0 + 15) / 16);
    const long long plane = ((long long)H * W + 8) & ~7LL;                   // + the zero unit that ends every plane, whole 128-byte lines
    o.header = 0;                                                            // [16 zero bytes][done: tiles][error]
    o.xps = align256(16 + (long long)(o.tiles + 1) * 4);
    o.fps = o.xps + align256(2LL * o.groups0 * plane * 16);
    o.tps = o.fps + align256(2LL * 8 * plane * 16);
    o.total = o.tps + align256(2LL * 8 * plane * 16);
    return o;
}

} // namespace

extern "C" {

/* Diagnostics only (tools/lab/bench_trunk.py): bit 0 skip the MFMAs, 1 skip the activation DMA, 2 skip the stores, 3 skip the waits on
 * the neighbours, 4 skip the weight DMA, 5 the MFMAs on operands read once per k-step (no LDS traffic).  Results are wrong with any bit set; bench.py refuses to report with a non-zero
 * isrDebugTrunkState(). */
#ifdef ISR_DIAG
void isrDebugSetTrunkAblation(int mask) { g_trunk_dbg = mask & 127; }
/* [tiles][layers][8] unsigned long long of ZEROED device memory (or NULL): per tile and layer the tick (100 MHz, one clock for the
 * chip) at the layer's start | the neighbours' arrival | the first k-step staged | the MFMAs done | the epilogue done | the stores
 * drained (slot 1 stays 0 for the first layer). */
void isrDebugSetTrunkStampBuffer(unsigned long long* stamps) { g_trunk_stamps = stamps; }
int isrDebugTrunkState(void);
/* Tests of the timeout path: tile `tile` (>= 0) never publishes its progress, so its neighbours' waits run into the deadline, which
 * is `timeoutTicks` of the 100 MHz clock after the kernel's start (0: the default 50 ms); tile < 0 switches the fault off.  The launch
 * then ends with the error word set BY THE KERNEL and an incomplete output. */
void isrDebugSetTrunkFault(int tile, unsigned long long timeoutTicks)
{
    g_trunk_fault_tile = tile;
    g_trunk_timeout_ticks = timeoutTicks ? timeoutTicks : 5000000ull;
}
#endif
/* Where every later isrTrunkDataflow launch reports a timed-out wait (atomic maximum of 1 + layer; sticky until the caller clears
 * it): a device word the caller mirrors to the host once per frame (ops.guards_publish), or NULL for the workspace's own word. */
void isrSetTrunkErrorWord(unsigned* word) { g_trunk_error_word = word; }

// ISR_TRUNK_MT: 1 (default) images of more tiles than CUs take trunk_mt_kernel; 0 they are refused (per-layer kernels); 2 every image
// takes it (A/B against the one-tile form)
static int g_trunk_mt = isr_diag_env_int("ISR_TRUNK_MT", 1);
// ISR_TRUNK_ROWS: image rows per wave of the one-tile form -- 2 (eight waves per workgroup) or 4 (four waves, one per SIMD)
static int g_trunk_rows = (getenv("ISR_TRUNK_ROWS") && atoi(getenv("ISR_TRUNK_ROWS")) == 4) ? 4 : 2;
void isrSetTrunkRows(int rows) { g_trunk_rows = rows == 4 ? 4 : 2; }
#ifdef ISR_DIAG
void isrDebugSetTrunkMultiTile(int mode) { g_trunk_mt = mode; }
int isrDebugTrunkState(void)
{
    return g_trunk_dbg | ((g_trunk_fault_tile >= 0 || g_trunk_timeout_ticks != 5000000ull) ? 128 : 0) | (g_trunk_stamps ? 256 : 0) | (g_trunk_mt != 1 ? 512 : 0);
}
#endif

int isrTrunkDataflowMaxTiles(void)
{
    int dev = 0, cus = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    return cus;
}

long long isrTrunkDataflowWorkspaceBytes(int cin0, int H, int W)
{
    if (H <= 0 || W <= 0 || cin0 <= 0) return -1;
    return trunk16_layout(cin0, H, W).total;
}

/* isrSetTrunkPackedResult(1): later isrTrunkDataflow launches leave their result PACKED-SPLIT as well (besides y), at the byte offset
 * into the workspace and with the plane stride in 16-byte units that isrTrunkDataflowPackedResult reports -- [2 parts][8 groups][plane]
 * -- valid until the next launch on the same workspace. */
static int g_trunk_last_packed = 0;
void isrSetTrunkPackedResult(int on) { g_trunk_last_packed = on ? 1 : 0; }

int isrTrunkDataflowPackedResult(int cin0, int H, int W, long long* offsetBytes, long long* planeUnits)
{
    if (H <= 0 || W <= 0 || cin0 <= 0 || !offsetBytes || !planeUnits) return -1;
    *offsetBytes = trunk16_layout(cin0, H, W).fps;
    *planeUnits = ((long long)H * W + 8) & ~7LL;
    return 0;
}

int isrTrunkDataflowSupported(const float* x, int cin0, int H, int W, long long xPlane, long long plane)
{
    if (!x || H <= 0 || W <= 0 || cin0 <= 0 || cin0 > 1024) return 0;
    if (xPlane < (long long)H * W || plane < (long long)H * W || plane * 64 * 4 > 0x7fffffffLL) return 0;
    if (trunk16_layout(cin0, H, W).total > 0xffffffffLL || ((long long)H * W + 8) * 16 * 16 > 0x7fffffffLL) return 0;   // 32-bit offsets into the workspace
    const long long tiles = (long long)((W + T16_W - 1) / T16_W) * ((H + T16_H - 1) / T16_H);
    return (tiles <= isrTrunkDataflowMaxTiles() || (g_trunk_mt && tiles <= 65535)) ? 1 : 0;
}

/* Where in the workspace an isrTrunkDataflow launch of this size keeps its packed-split tensors (byte offsets: input, F, T), the input's channel
 * groups of 8 and the number of tiles (progress counters at byte 16): for a producer that writes the input there itself (isrAssembleInputPacked). */
int isrTrunkDataflowInputLayout(int cin0, int H, int W, long long* offsets3, int* groups0, int* tiles)
{
    if (H <= 0 || W <= 0 || cin0 <= 0 || !offsets3 || !groups0 || !tiles) return -1;
    const Trunk16Layout lay = trunk16_layout(cin0, H, W);
    if (lay.total > 0xffffffffLL) return -3;
    offsets3[0] = lay.xps; offsets3[1] = lay.fps; offsets3[2] = lay.tps;
    *groups0 = lay.groups0; *tiles = lay.tiles;
    return 0;
}

static int trunk_dataflow_launch(const float* x, bool prepacked, int cin0, long long xPlane, float* y, long long plane, const void* const* wq,
                                 const float* const* bias, int nblocks, int H, int W, void* workspace, void* stream);

/* x [cin0][H][W] -> y = F after: F = relu(conv(x, w[0]) + b[0]); nblocks times F += conv(relu(conv(F, w[2k+1]) + b[2k+1]), w[2k+2]) + b[2k+2]
 * (y [64][H][W] with `plane` floats per channel).  wq[l]: isrConvSplitPrepare images. */
int isrTrunkDataflow(const float* x, int cin0, long long xPlane, float* y, long long plane, const void* const* wq,
                     const float* const* bias, int nblocks, int H, int W, void* workspace, void* stream)
{
    return trunk_dataflow_launch(x, false, cin0, xPlane, y, plane, wq, bias, nblocks, H, W, workspace, stream);
}

/* ... with the input ALREADY packed-split in the workspace (isrAssembleInputPacked on the same stream, which also did the packing pass's
 * housekeeping: zero units, progress counters): no packing pass. */
int isrTrunkDataflowPrepacked(int cin0, float* y, long long plane, const void* const* wq, const float* const* bias, int nblocks, int H, int W,
                              void* workspace, void* stream)
{
    return trunk_dataflow_launch(nullptr, true, cin0, (long long)H * W, y, plane, wq, bias, nblocks, H, W, workspace, stream);
}

static int trunk_dataflow_launch(const float* x, bool prepacked, int cin0, long long xPlane, float* y, long long plane, const void* const* wq,
                                 const float* const* bias, int nblocks, int H, int W, void* workspace, void* stream)
{
    unsigned* const rangeFlag = isr_take_range_flag();       // taken first: an error return must not leave it armed
    if ((!x && !prepacked) || !y || !wq || !bias || !workspace || nblocks < 0 || 1 + 2 * nblocks > T16_MAX_LAYERS || cin0 <= 0) return -1;
    if (!isrTrunkDataflowSupported(prepacked ? y : x, cin0, H, W, xPlane, plane) || ((uintptr_t)workspace & 255)) return -3;
    const Trunk16Layout lay = trunk16_layout(cin0, H, W);
    char* ws = (char*)workspace;
    u32x4* xps = (u32x4*)(ws + lay.xps);
    Trunk16Params p;
    std::memset(&p, 0, sizeof(p));
    const int layers = 1 + 2 * nblocks;
    for (int l = 0; l < layers; ++l) {
        if (!wq[l]) return -1;
        p.wq[l] = (const u32x4*)wq[l]; p.bias[l] = bias[l];
    }
    p.xpsOff = (unsigned)lay.xps; p.fpsOff = (unsigned)lay.fps; p.tpsOff = (unsigned)lay.tps; p.groups0 = lay.groups0;
    p.y = y; p.yPlane = (int)plane;
    p.H = H; p.W = W; p.plane = (int)(((long long)H * W + 8) & ~7LL); p.layers = layers;
    p.tilesX = (W + T16_W - 1) / T16_W; p.tilesY = (H + T16_H - 1) / T16_H;
    const int ntiles = p.tilesX * p.tilesY;
    p.ws = ws; p.done = (unsigned*)(ws + 16); p.error = g_trunk_error_word ? g_trunk_error_word : p.done + ntiles;
    p.absmax = rangeFlag;
    ISR_DIAG_SET(p.dbg, g_trunk_dbg); ISR_DIAG_SET(p.stamps, g_trunk_stamps);
    p.timeoutTicks = g_trunk_timeout_ticks;                                  // 50 ms unless a test shortened it: a frame is 2 ms
    ISR_DIAG_SET(p.faultTile, g_trunk_fault_tile);
    p.lastPacked = g_trunk_last_packed;
    hipStream_t s = (hipStream_t)stream;
    // the progress counters start at zero every launch; the error word behind them is STICKY (the caller zero-fills the workspace
    // once, reads the word when it likes and resets it then): an error of any launch since the last look stays visible
    // (trunk_pack_input_kernel zeroes the counters; the 16-byte zero unit in front of them is never written)
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)trunk_dataflow_kernel<false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, T16_LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)trunk_dataflow_kernel<false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, T16_LDS_BYTES);
#ifdef ISR_DIAG                                                              // (the diagnostics instantiations exist in the diagnostics build only)
        (void)hipFuncSetAttribute((const void*)trunk_dataflow_kernel<true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, T16_LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)trunk_dataflow_kernel<true, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, T16_LDS_BYTES);
#endif
        (void)hipFuncSetAttribute((const void*)trunk_mt_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, T16_LDS_BYTES);
        attr = true;
    }
#ifdef ISR_DIAG
    const bool diag = p.dbg != 0 || p.stamps != nullptr;
#endif
    const int npix = H * W;
    if (!prepacked)
        ISR_LAUNCH_PROFILED(ISR_VARIANT_TRUNK_PACK, trunk_pack_input_kernel, dim3((unsigned)((npix + 255) / 256), (unsigned)(lay.groups0 > 8 ? lay.groups0 : 8)), dim3(256), 0, s,
                            x, cin0, xPlane, npix, xps, lay.groups0, (u32x4*)(ws + lay.fps), (u32x4*)(ws + lay.tps), p.done, ntiles);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const int cus = isrTrunkDataflowMaxTiles();
    const bool mt = g_trunk_mt == 2 || ntiles > cus;
    isr_profile_record(mt ? ISR_VARIANT_SPLIT_TRUNK_MT : ISR_VARIANT_SPLIT_TRUNK, 2.0 * 9 * 64 * ((double)cin0 + 2.0 * nblocks * 64) * (double)H * W, &e0, &e1);
    const dim3 grid((unsigned)(((ntiles + 7) / 8) * 8)), block(T16_THREADS);
    if (mt) {
        // one workgroup per CU, each walking every cus-th tile (a whole number of XCD groups; never more workgroups than CUs: all must be resident)
        const int round8 = ((ntiles + 7) / 8) * 8;
        const dim3 mgrid((unsigned)(round8 < cus ? round8 : cus));
        if (e0 || e1) hipExtLaunchKernelGGL(trunk_mt_kernel, mgrid, block, T16_LDS_BYTES, s, e0, e1, 0, p);
        else hipLaunchKernelGGL(trunk_mt_kernel, mgrid, block, T16_LDS_BYTES, s, p);
    } else if (g_trunk_rows == 4) {
        const dim3 block4(256);
#ifdef ISR_DIAG
        if (diag) {
            if (e0 || e1) hipExtLaunchKernelGGL((trunk_dataflow_kernel<true, 4>), grid, block4, T16_LDS_BYTES, s, e0, e1, 0, p);
            else hipLaunchKernelGGL((trunk_dataflow_kernel<true, 4>), grid, block4, T16_LDS_BYTES, s, p);
        } else
#endif
        if (e0 || e1) hipExtLaunchKernelGGL((trunk_dataflow_kernel<false, 4>), grid, block4, T16_LDS_BYTES, s, e0, e1, 0, p);
        else hipLaunchKernelGGL((trunk_dataflow_kernel<false, 4>), grid, block4, T16_LDS_BYTES, s, p);
    } else {
#ifdef ISR_DIAG
        if (diag) {
            if (e0 || e1) hipExtLaunchKernelGGL((trunk_dataflow_kernel<true, 2>), grid, block, T16_LDS_BYTES, s, e0, e1, 0, p);
            else hipLaunchKernelGGL((trunk_dataflow_kernel<true, 2>), grid, block, T16_LDS_BYTES, s, p);
        } else
#endif
        if (e0 || e1) hipExtLaunchKernelGGL((trunk_dataflow_kernel<false, 2>), grid, block, T16_LDS_BYTES, s, e0, e1, 0, p);
        else hipLaunchKernelGGL((trunk_dataflow_kernel<false, 2>), grid, block, T16_LDS_BYTES, s, p);
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

} // extern "C"
