// Two chained 64 -> 64 convolutions of a batch of SMALL images in ONE launch: a residual block of the training trunk,
//     z = relu(conv(x, wa) + ba),   y = conv(z, wb) + bb + x          (models/enhancenet.py:18-33,139-141)
// or its data gradient (backward of the same block, weights flipped / transposed by the caller),
//     z = gate > 0 ? conv(x, wa) : 0,   y = conv(z, wb) + x           (x = the gradient of the block's output, gate = the saved relu output).
// Included by sr_conv_split.hip (same translation unit as conv3x3_split_rows2_kernel, whose geometry and arithmetic this is).
//
// Why.  A B = 16 batch of 32 x 32 training crops runs its 20 trunk convolutions forward and 20 backward per frame as launches of
// ~13.5 us each for ~2 us of MFMA work (conv3x3_split_rows2_kernel: 256 workgroups of 2 rows x 32 pixels): a launch is a chain of
// latencies -- dispatch, first fetch, store drain, the kernel boundary's cache write-back.  A dataflow chain with flags between the
// tiles (sr_conv_trunk.hip) does not pay here: with 2-row tiles EVERY row is a halo row, and drain + wait + fetch through memory cost
// what the launch boundary costs.  Halo recomputation does: the workgroup stages SIX rows of x, computes the FOUR rows of z its two
// output rows need (its own two go to memory -- they are saved for the backward pass -- and all four into LDS as the (hi, lo')
// operand image of the second convolution, zero outside the image exactly like the zero padding the second launch would have read)
// and then its two rows of y.  1.5 x the MFMAs of two launches, one chain of latencies instead of two.  z's halo rows are the same
// products in the same order as the neighbour workgroup's own rows: results are bit-identical to two conv3x3_split_rows2_kernel
// launches (tests/test_train_kernels_gpu.py).
//
// A wave = (row r in {0, 1}, 32-channel block cb): stage 1 gives it z rows r and r + 2 of the four (two accumulators, the weight
// fragments shared: 6 MFMAs per tap), stage 2 its output row r (one accumulator).  The eight k-steps of weights (four per stage)
// arrive by LDS-DMA (global_load_lds_dwordx4: no registers, so the compiler can fetch a tap's operands under the MFMAs of the tap
// before) through a ring of THREE k-step buffers, each requested two k-steps ahead; a k-step ends in one barrier, in front of it a
// COUNTED wait (the nine requests just issued may stay in flight).  x patch + three buffers + biases = 163 328 of the 163 840 bytes
// of LDS.  The z image overwrites the x patch.
#pragma once
#include "sr_diag.h"
#include "sr_split_common.h"

namespace {

constexpr int B2_XROWS = R2_H + 4;                                           // 6 patch rows of x
constexpr int B2_XPIX = B2_XROWS * SP_W;                                     // 204 patch pixels per channel group
constexpr int B2_XPART = 8 * B2_XPIX;                                        // units of the hi (or lo) patch of the 64 channels: 1632
constexpr int B2_XUNITS = 2 * B2_XPART;                                      // 3264 units = 52 224 B
constexpr int B2_STAGE = 8 * B2_XROWS * SQ_QPR;                              // 480 staging units (channel group, patch row, quad)
constexpr int B2_WRING = 3;                                                  // k-step buffers of weights
constexpr int B2_LDS_BYTES = (B2_XUNITS + B2_WRING * S_WUNITS) * 16 + 128 * 4;   // 52 224 + 110 592 + 512 = 163 328
static_assert(B2_LDS_BYTES <= 160 * 1024, "LDS");

typedef __attribute__((address_space(3))) char b2_lds_char;
// LDS-DMA with a scalar base and a 32-bit lane offset: lane l's 16 bytes at base + voff land at LDS address ldsaddr + 16 l.  Issued
// from inline assembly: the compiler does not count these requests, every wait for them below is explicit.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void b2_dma16(const void* base, unsigned voff, unsigned ldsaddr)
{
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(ldsaddr), "v"(voff), "s"(base) : "m0");
}
#pragma clang diagnostic pop

struct Block2Params {
    const float* x;                  // [N][64][H][W]: input of stage 1, residual of stage 2
    const u32x4* wa; const float* ba; const float* gate;   // stage 1 (gate == NULL: bias + ReLU; else: no bias, gated by gate > 0)
    const u32x4* wb; const float* bb;                      // stage 2 (bb may be NULL)
    float* z; float* y;              // [N][64][H][W] each
    int N, H, W, tilesY;
    unsigned* absmax;                // range guard over z and y (may be NULL)
    unsigned* zmax; unsigned* ymax;  // [4 x workgroups] bit patterns of max |z| / max |y| per WAVE (plain stores, may be NULL): the
                                     // weight-gradient kernels scale their gz operand by the maximum (isrConv3x3WeightGradSegmentsSplitMax).
                                     // (One word and atomics: 1024 waves on one address per launch cost 4.5 us per word.)
    ISR_DIAG_MEMBER(int, dbg, 0);                         // diagnostics: 1 skip the MFMAs, 4 skip the stores, 8 stage 1 on operands read once per k-step, 16 no weight DMA, 32 stage 1 eight times
    ISR_DIAG_MEMBER(unsigned long long*, stamps, nullptr);      // diagnostics: [workgroup][8] s_memrealtime ticks (100 MHz): entry | x patch parked | stage 1 done |
                                     // z image written | stage 2 done | stores issued | stores drained
};

// the fragments of one tap: stage 1 (two z rows share the weights) / stage 2
struct B2Taps1 { f16x8 ah, al, bh0, bo0, bh1, bo1; };
struct B2Taps2 { f16x8 ah, al, bh, bo; };
__device__ __forceinline__ B2Taps1 b2_taps1(const u32x4* wl, const u32x4* bl, int tap)
{
    const int dy = tap / 3, dx = tap - dy * 3;
    B2Taps1 o;
    o.ah = __builtin_bit_cast(f16x8, wl[tap * 128]);
    o.al = __builtin_bit_cast(f16x8, wl[S_WPART + tap * 128]);
    o.bh0 = __builtin_bit_cast(f16x8, bl[dy * SP_W + dx]);
    o.bo0 = __builtin_bit_cast(f16x8, bl[B2_XPART + dy * SP_W + dx]);
    o.bh1 = __builtin_bit_cast(f16x8, bl[(dy + 2) * SP_W + dx]);
    o.bo1 = __builtin_bit_cast(f16x8, bl[B2_XPART + (dy + 2) * SP_W + dx]);
    return o;
}
__device__ __forceinline__ B2Taps2 b2_taps2(const u32x4* wl, const u32x4* bl, int tap)
{
    const int dy = tap / 3, dx = tap - dy * 3;
    B2Taps2 o;
    o.ah = __builtin_bit_cast(f16x8, wl[tap * 128]);
    o.al = __builtin_bit_cast(f16x8, wl[S_WPART + tap * 128]);
    o.bh = __builtin_bit_cast(f16x8, bl[dy * SP_W + dx]);
    o.bo = __builtin_bit_cast(f16x8, bl[R2_PART + dy * SP_W + dx]);
    return o;
}

__global__ __launch_bounds__(S_THREADS, 1) void conv3x3_split_block2_kernel(const Block2Params p)
{
    extern __shared__ u32x4 patch[];
    u32x4* const wbuf = patch + B2_XUNITS;
    float* const biases = reinterpret_cast<float*>(wbuf + B2_WRING * S_WUNITS);   // ba[64], bb[64] (zeros where there is none)
    const unsigned wAddr = (unsigned)(uintptr_t)(b2_lds_char*)wbuf;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int row = wave >> 1, cb = wave & 1;
    int bid;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    }
    auto lap = [&](int slot) { if (p.stamps && tid == 0) p.stamps[(size_t)blockIdx.x * 8 + slot] = __builtin_amdgcn_s_memrealtime(); };
    unsigned long long clk1 = 0;       // shader-clock ticks over stage 1 (slot 7): the clock the kernel runs at = ticks / (10 ns x real-time ticks)
    lap(0);
    // the biases travel with the first staging pass (a global load per channel in the epilogues would queue behind their stores)
    float bval = 0.0f;
    if (tid < 128) {
        const float* b = tid < 64 ? p.ba : p.bb;
        if (b) bval = b[tid & 63];
    }
    const int ty = bid % p.tilesY, n = bid / p.tilesY;
    const int oy0 = ty * R2_H;
    const unsigned plane = (unsigned)(p.H * p.W), planeBytes = plane * 4u;
    const size_t image = (size_t)64 * plane;
    const rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x + (size_t)n * image), 0, (int)(64u * planeBytes), 0x00020000);

    // ---- the x patch: rows oy0 - 2 .. oy0 + 3, columns -1 .. 32, all 64 channels -----------------------------------------------
    u32x4 v[2][8];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int u = tid + k * S_THREADS;
        const int g = u / (B2_XROWS * SQ_QPR), rem = u - g * (B2_XROWS * SQ_QPR);
        const int r = rem / SQ_QPR, q = rem - r * SQ_QPR;
        const int iy = oy0 + r - 2, ix = 4 * q - 4;
        const bool ok = u < B2_STAGE && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        const unsigned base = (unsigned)(g * 8) * planeBytes + (unsigned)(iy * p.W + ix) * 4u;
#pragma unroll
        for (int e = 0; e < 8; ++e)
            v[k][e] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)(ok ? base + (unsigned)e * planeBytes : BAD_OFFSET), 0, 0);
    }
    // k-step gk (0..3: wa, 4..7: wb) into ring buffer gk % 3: 36 pieces of 64 units (part, tap, lane half; the 64 couts contiguous in
    // the prepared image), nine per wave
    auto wdma = [&](int gk) {
        if (p.dbg & 16) return;                                              // diagnostics: no weight traffic
        const char* img = reinterpret_cast<const char*>(gk < 4 ? p.wa : p.wb) + 16;
        const unsigned dst = wAddr + (unsigned)(gk % B2_WRING) * (S_WUNITS * 16);
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int ch = wave * 9 + i;
            const int part = ch / 18, t2 = ch - part * 18, tap = t2 >> 1, hh = t2 & 1;
            b2_dma16(img + (size_t)((((tap * 4 + (gk & 3)) * 2 + part) * 2 + hh) * 64) * 16, (unsigned)lane * 16u, dst + (unsigned)ch * 1024u);
        }
    };
    wdma(0);
    wdma(1);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int u = tid + k * S_THREADS;
        if (u >= B2_STAGE) continue;
        const int g = u / (B2_XROWS * SQ_QPR), rem = u - g * (B2_XROWS * SQ_QPR);
        const int r = rem / SQ_QPR, q = rem - r * SQ_QPR;
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
        u32x4* dst = patch + g * B2_XPIX + r * SP_W + 4 * q - 3;
        if (q > 0) { dst[0] = __builtin_bit_cast(u32x4, h0); dst[B2_XPART] = __builtin_bit_cast(u32x4, l0); }
        if (q > 0 && q < SQ_QPR - 1) {
            dst[1] = __builtin_bit_cast(u32x4, h1); dst[B2_XPART + 1] = __builtin_bit_cast(u32x4, l1);
            dst[2] = __builtin_bit_cast(u32x4, h2); dst[B2_XPART + 2] = __builtin_bit_cast(u32x4, l2);
        }
        if (q < SQ_QPR - 1) { dst[3] = __builtin_bit_cast(u32x4, h3); dst[B2_XPART + 3] = __builtin_bit_cast(u32x4, l3); }
    }
    if (tid < 128) biases[tid] = bval;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    lap(1);
    if (p.stamps) clk1 = __builtin_amdgcn_s_memtime();

    // ---- stage 1: z rows `row` and `row + 2` of the four (image rows oy0 - 1 + tr), 32 channels each --------------------------------
    const rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.gate ? p.gate + (size_t)n * image : p.x), 0,
                                                         p.gate ? (int)(64u * planeBytes) : 0, 0x00020000);
    float gv[2][16];
    f32x16 za, zb;
#pragma unroll
    for (int i = 0; i < 16; ++i) { za[i] = 0.0f; zb[i] = 0.0f; }
    const int reps1 = (p.dbg & 32) ? 32 : 4;                                // diagnostics: stage 1 eight times over (with 16: steady-state rate)
#pragma unroll 1
    for (int gq = 0; gq < reps1; ++gq) {
        const int gk = gq & 3;
        wdma(gk + 2);                                                        // two k-steps ahead (gk = 2, 3: the second stage's first two)
        if (gk == 2 && p.gate) {                                             // the gate operand of the epilogue: in flight under two k-steps
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int iy = oy0 - 1 + row + 2 * s;
                const bool ok = (unsigned)iy < (unsigned)p.H && j < p.W;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int c = cb * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                    gv[s][i] = buf_load(grs, ok ? (unsigned)c * planeBytes + (unsigned)(iy * p.W + j) * 4u : BAD_OFFSET);
                }
            }
        }
        if (!(p.dbg & 1)) {
            const u32x4* wl = wbuf + (gk % B2_WRING) * S_WUNITS + h * 64 + cb * 32 + j;
            const u32x4* bl = patch + (2 * gk + h) * B2_XPIX + row * SP_W + j;
            // the operands of tap t + 1 are requested BEFORE the six MFMAs of tap t are issued (left to itself the compiler reads
            // each fragment right in front of its MFMA and waits for it: one wave per SIMD, nothing else to run meanwhile)
            B2Taps1 cur = b2_taps1(wl, bl, 0);
            if (p.dbg & 8) {                                                 // diagnostics: the k-step's 54 MFMAs on operands read once
                const f16x8 as = cur.ah * (_Float16)0.00048828125f;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    za = mfma16(cur.al, cur.bh0, za); za = mfma16(as, cur.bo0, za); za = mfma16(cur.ah, cur.bh0, za);
                    zb = mfma16(cur.al, cur.bh1, zb); zb = mfma16(as, cur.bo1, zb); zb = mfma16(cur.ah, cur.bh1, zb);
                }
            } else
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                B2Taps1 nxt = cur;
                if (tap < 8) nxt = b2_taps1(wl, bl, tap + 1);
                __builtin_amdgcn_sched_barrier(0);
                const f16x8 as = cur.ah * (_Float16)0.00048828125f;         // w_hi 2^-11: partner of the scaled x_lo'
                za = mfma16(cur.al, cur.bh0, za);
                za = mfma16(as, cur.bo0, za);
                za = mfma16(cur.ah, cur.bh0, za);
                zb = mfma16(cur.al, cur.bh1, zb);
                zb = mfma16(as, cur.bo1, zb);
                zb = mfma16(cur.ah, cur.bh1, zb);
                __builtin_amdgcn_sched_barrier(0);
                cur = nxt;
            }
        }
        // k-step gk + 1 has landed (everything but this iteration's requests: 9 DMA pieces, at gk = 2 the 32 gate loads behind them --
        // loads return in order), everyone is done with buffer gk % 3
        if (gk == 2 && p.gate) asm volatile("s_waitcnt vmcnt(41)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
        __syncthreads();
    }
    lap(2);
    if (p.stamps && tid == 0) p.stamps[(size_t)blockIdx.x * 8 + 7] = __builtin_amdgcn_s_memtime() - clk1;
    // ---- z: own rows to memory, all four rows into LDS as the second convolution's operand image (rows2 geometry) ------------------
    unsigned mag = 0u, ymag = 0u;
    {
        const float unscale = reinterpret_cast<const float*>(p.wa)[1];
        const rsrc_t zrs = __builtin_amdgcn_make_buffer_rsrc(p.z + (size_t)n * image, 0, (int)(64u * planeBytes), 0x00020000);
        if (tid < 128) {                                                     // the zero columns left and right of the image
            const u32x4 zero = {0u, 0u, 0u, 0u};
            const int part = tid >> 6, g = (tid >> 3) & 7, r4 = (tid >> 1) & 3;
            patch[part * R2_PART + g * R2_PIX + r4 * SP_W + ((tid & 1) ? SP_W - 1 : 0)] = zero;
        }
        // this lane's sixteen bias values (the same channels for both rows) in one batch: a ds_read per value between the stores cost
        // the forward launch a microsecond
        float bz[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) bz[i] = p.gate ? 0.0f : biases[cb * 32 + (i & 3) + 8 * (i >> 2) + 4 * h];
        const bool track = p.absmax != nullptr || p.zmax != nullptr;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int tr = row + 2 * s, iy = oy0 - 1 + tr;
            const bool in = (unsigned)iy < (unsigned)p.H && j < p.W;
            const bool ownRow = (tr == 1 || tr == 2) && !(p.dbg & 4);         // wave uniform: the halo rows are not stored at all
            const unsigned pix = (unsigned)(iy * p.W + j) * 4u;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                h4 vh, vl;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = g4 * 4 + e;
                    const int c = cb * 32 + e + 8 * g4 + 4 * h;
                    float val = (s ? zb[i] : za[i]) * unscale;
                    if (p.gate) val = gv[s][i] > 0.f ? val : 0.f;
                    else { val += bz[i]; val = val > 0.f ? val : 0.f; }
                    if (!in) val = 0.0f;
                    if (track) mag = isr_umax(mag, isr_mag(val));
                    if (ownRow) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), zrs, in ? (int)((unsigned)c * planeBytes + pix) : (int)BAD_OFFSET, 0, 0);
                    _Float16 a, b;
                    split16x(val, a, b);
                    vh[e] = a; vl[e] = b;
                }
                char* unit = reinterpret_cast<char*>(patch + (cb * 4 + g4) * R2_PIX + tr * SP_W + j + 1) + h * 8;
                *reinterpret_cast<uint2*>(unit) = __builtin_bit_cast(uint2, vh);
                *reinterpret_cast<uint2*>(unit + R2_PART * 16) = __builtin_bit_cast(uint2, vl);
            }
        }
    }
    // the residual of stage 2 (x, own rows; 4 quads per thread, the index map of the stores): in flight under stage 2
    const int oy = oy0 + row;
    u32x4 rq[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int q = lane + 64 * k;
        const int co = cb * 32 + (q >> 3), px = (q & 7) * 4;
        const bool ok = oy < p.H && px < p.W;
        rq[k] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)(ok ? (unsigned)co * planeBytes + (unsigned)(oy * p.W + px) * 4u : BAD_OFFSET), 0, 0);
    }
    __syncthreads();
    lap(3);

    // ---- stage 2: output row `row`, 32 channels (conv3x3_split_rows2_kernel's loop on the z image) ----------------------------------
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
#pragma unroll 1
    for (int gk = 4; gk < 8; ++gk) {
        if (gk + 2 < 8) wdma(gk + 2);
        if (!(p.dbg & 1)) {
            const u32x4* wl = wbuf + (gk % B2_WRING) * S_WUNITS + h * 64 + cb * 32 + j;
            const u32x4* bl = patch + (2 * (gk - 4) + h) * R2_PIX + row * SP_W + j;
            B2Taps2 cur = b2_taps2(wl, bl, 0);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                B2Taps2 nxt = cur;
                if (tap < 8) nxt = b2_taps2(wl, bl, tap + 1);
                __builtin_amdgcn_sched_barrier(0);
                const f16x8 as = cur.ah * (_Float16)0.00048828125f;
                acc = mfma16(cur.al, cur.bh, acc);
                acc = mfma16(as, cur.bo, acc);
                acc = mfma16(cur.ah, cur.bh, acc);
                __builtin_amdgcn_sched_barrier(0);
                cur = nxt;
            }
        }
        if (gk + 2 < 8) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    lap(4);
    // ---- y = conv + bias + x: one output row x 32 channels per wave, transposed through 4 KB of the idle patch -----------------------
    if (!(p.dbg & 4)) {
        const float unscale = reinterpret_cast<const float*>(p.wb)[1];
        const rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(p.y + (size_t)n * image, 0, (int)(64u * planeBytes), 0x00020000);
        float* tr = reinterpret_cast<float*>(patch) + wave * (32 * 32);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int c = (i & 3) + 8 * (i >> 2) + 4 * h;
            tr[c * 32 + j] = acc[i] * unscale + biases[64 + cb * 32 + c];
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);                                  // lgkmcnt(0): same-wave hand-off through LDS
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int q = lane + 64 * k;                                     // float4 index: cout = q / 8, pixel group = q % 8
            const int co = cb * 32 + (q >> 3), px = (q & 7) * 4;
            const bool ok = oy < p.H && px < p.W;
            float4 val = reinterpret_cast<const float4*>(tr)[q];
            const float4 rf = __builtin_bit_cast(float4, rq[k]);
            val.x += rf.x; val.y += rf.y; val.z += rf.z; val.w += rf.w;
            if (ok) ymag = isr_umax(isr_umax(ymag, isr_umax(isr_mag(val.x), isr_mag(val.y))), isr_umax(isr_mag(val.z), isr_mag(val.w)));
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, val), yrs,
                                                   (int)(ok ? (unsigned)co * planeBytes + (unsigned)(oy * p.W + px) * 4u : BAD_OFFSET), 0, 0);
        }
    }
    lap(5);
    if (p.stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); lap(6); }
    if (p.zmax) {
        unsigned a = mag, b = ymag;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            a = isr_umax(a, (unsigned)__shfl_xor((int)a, o, 64));
            b = isr_umax(b, (unsigned)__shfl_xor((int)b, o, 64));
        }
        if (lane == 0) { p.zmax[blockIdx.x * 4 + wave] = a; p.ymax[blockIdx.x * 4 + wave] = b; }
    }
    isr_range_note(p.absmax, isr_umax(mag, ymag));
}

} // namespace
