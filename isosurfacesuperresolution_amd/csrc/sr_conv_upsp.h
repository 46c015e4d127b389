// conv3x3(U(x)) for EnhanceNet's two upsampling layers (SuperresolutionNetwork/models/enhancenet.py:113-124), PHASE-DECOMPOSED:
// no interpolation at run time.  Included by sr_conv_split.hip (same translation unit as the other split-operand kernels).
//
// U = bilinear x2 (align_corners=False) is linear, so conv3x3(U(x)) at the high-resolution pixel (2y + py, 2x + px) is a 3 x 3
// convolution of the LOW-resolution image around (y, x) with weights that depend on the output's parity only:
//     W_eff[py][px][r][s] = sum_{dy, dx} w[dy][dx] A[py][dy][r] A[px][dx][s],          r, s in {-1, 0, +1}
//     A[0] = [[3/4, 1/4, 0], [1/4, 3/4, 0], [0, 3/4, 1/4]]       (rows: high-res tap d = -1, 0, +1; columns: low-res offset)
//     A[1] = [[1/4, 3/4, 0], [0, 3/4, 1/4], [0, 1/4, 3/4]]
// on the REPLICATE-padded low-resolution image (index clamping is what the bilinear resize does at the image border).  The four
// parities are four plain split-operand convolutions that share one staged low-resolution patch; their outputs interleave into the
// high-resolution tensor.  What the workgroup no longer does: 340 interpolation units of ~230 vector instructions per k-step
// (61 % of the vector work of conv3x3_split_ups3_kernel, whose matrix pipe was busy 54 % of the launch with the vector ALUs
// saturated beside it, profiles/r04_pmc_ups.md) -- the B operand is the producer's packed-split tensor, copied into LDS as it is.
// The same multiply-accumulates (9 taps x Cin per output value), 4 x the weight images (prepared once per weight version).
//
// The one place where this is NOT the same function: the convolution's ZERO padding at the high resolution.  For the outermost
// one-pixel frame of the output (Y = 0, Y = H - 1, X = 0, X = W - 1) the taps that fall outside the image must be dropped, which
// is a different effective weight set per edge (and per lane for the columns).  The main kernel does not store those pixels;
// ups_frame_kernel computes them directly -- interpolation and fp32 FMA chain per tap, 0.3 % of the output.
//
// Numbers: W_eff is formed in fp64 and rounded once to fp32, then split like every weight (hi, lo, one power-of-two scale for the
// four images); x is the producer's (hi, lo') pair.  Each product carries 22 + 22 bits as in the other split kernels; the result is
// NOT bit-identical to interpolate-then-convolve (different roundings: U(x) is never rounded to fp32 here) and is tested at the
// same distance from an fp64 convolution as the kernels it replaces (tests/test_upsp_gpu.py).
#pragma once
#ifndef UP_BIAS_IN_LOOP
#define UP_BIAS_IN_LOOP 0
#endif
#include "sr_split_common.h"

namespace {

constexpr int UP_PART = 2 * SP_PIX;                                          // one k-step of the low-res patch: 2 channel groups; hi, lo' at + UP_PART
constexpr int UP_PUNITS = 2 * UP_PART;                                       // 1360 units = 21 760 B
constexpr int UP_WROW = 3 * 128;                                             // one tap row of one plane: 3 taps x [lane half][64 couts]
constexpr int UP_WUNITS = 2 * UP_WROW;                                       // planes hi, lo (hi 2^-11 is made in registers): 768 units = 12 288 B
constexpr int UP_WBUFS = 3;                                                  // weight rows in flight: the row being multiplied + two on their way
constexpr int UP_LDS_UNITS = 2 * UP_PUNITS + UP_WBUFS * UP_WUNITS;           // 80 384 B
constexpr int UP_LDS_BYTES = UP_LDS_UNITS * 16 + 256;                        // + the bias row: two workgroups per CU
constexpr int UP_PPIECES = (UP_PUNITS + 63) / 64;                            // 22 wave-wide pieces of a patch slice (the last one 16 units)
constexpr int UP_WPIECES = UP_WUNITS / 64;                                   // 12 pieces of a tap row: three per wave

typedef __attribute__((address_space(3))) char up_lds_char;

// LDS-DMA from inline assembly (as sr_conv_trunk.hip): 64 lanes x 16 bytes, lane l's bytes land at ldsaddr + 16 l; the compiler does
// not count these requests -- the kernel waits for them with its own s_waitcnt vmcnt
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void up_dma16(const void* base, unsigned voff, unsigned ldsaddr)
{
    ldsaddr = __builtin_amdgcn_readfirstlane(ldsaddr);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(ldsaddr), "v"(voff), "s"(base) : "m0", "memory");
}
#pragma clang diagnostic pop

// p.H, p.W: OUTPUT (high-resolution) size; p.Hin, p.Win: input size; p.xps / p.ps: packed-split input / output; p.wq: the prepared
// image of the four stacked effective weight sets (Cout = 256: image m = 2 py + px at output channels 64 m ..); tiles of 8 x 32
// LOW-resolution pixels (p.tilesX, p.tilesY).
//
// Schedule.  A tile is 48 steps (image m, k-step ks, tap row dy), each 36 MFMAs per wave (~0.6 us) on one tap row of weights (12 KB:
// hi and lo planes; the partner of the scaled x_lo', w_hi 2^-11, is an exponent shift of the hi fragment in registers) and one k-step
// slice of the patch (21 KB, shared by the three rows of a k-step).  Operands arrive by LDS-DMA straight from L2 -- no registers, no
// conversion.  A request takes 1-2 us under load, longer than a step: the weights of step G + 2 are requested when step G opens (THREE
// rotating row buffers), the next patch slice when its predecessor's first row opens (two buffers), and the wait that opens a step
// leaves everything younger than its own operands in flight (`s_waitcnt vmcnt(N)`: requests complete in issue order; N = what this
// wave issued after the operands it needs now).  ONE barrier per step.  The only vector work left is the epilogue of each image
// (64 values per lane), which runs beside the other workgroup's MFMAs.
__global__ __launch_bounds__(S_THREADS, 2) void conv3x3_split_upsp_kernel(const SplitConvParams p)
{
    extern __shared__ u32x4 lds[];
    u32x4* const pbuf0 = lds;                                                // [2][UP_PUNITS]
    u32x4* const wbuf0 = lds + 2 * UP_PUNITS;                                // [UP_WBUFS][UP_WUNITS]
    float* const biasl = reinterpret_cast<float*>(lds + UP_LDS_UNITS);
    const unsigned ldsBase = (unsigned)(uintptr_t)(up_lds_char*)lds;
    const unsigned pAddr = ldsBase, wAddr = ldsBase + 2 * UP_PUNITS * 16;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    int bid;
    {   // an XCD (= an L2) gets a contiguous range of tiles
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    }
    const int tx = bid % p.tilesX, ty = bid / p.tilesX;
    const int oy0 = ty * ST_H, ox0 = tx * ST_W;                              // low-resolution origin of the tile
    const int groups = p.Cin >> 3;
    constexpr int K = 4;                                                     // k-steps (64 input channels): the launcher admits nothing else
    constexpr int CP = 256;                                                  // coutPad of the stacked image
    constexpr int STEPS = 4 * K * 3;
    if (tid < 64) biasl[tid] = p.bias ? p.bias[tid] : 0.0f;
    // diagnostics (isrDebugSetSplitStampBuffer): 10 ticks of the 100 MHz clock per workgroup -- start, first operands landed, and per image
    // "MFMAs issued" / "epilogue issued"; straight to memory, no registers held
    auto lap = [&](int slot) {
        if (p.stamps && tid == 0) p.stamps[(size_t)blockIdx.x * 10 + slot] = __builtin_amdgcn_s_memrealtime();
    };
    lap(0);

    // ---- patch slice of k-step ks: 2 parts x 2 channel groups x 10 x 34 pixels.  Wave w moves pieces w, w + 4, ..; lane l of piece pc moves
    //      unit u = 64 pc + l.  Source pixels are CLAMPED into the image (replicate padding); the lane offsets depend on the tile only.
    //      Waves 0, 1 issue six requests per slice, waves 2, 3 five (the wait counts below know).
    unsigned poff[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int u = (wave + 4 * k) * 64 + lane;
        const int part = u / UP_PART, rem = u - part * UP_PART;
        const int g = rem / SP_PIX, pix = rem - g * SP_PIX;
        const int r = pix / SP_W, c = pix - r * SP_W;
        const int iy = min(max(oy0 + r - 1, 0), p.Hin - 1), ix = min(max(ox0 + c - 1, 0), p.Win - 1);
        poff[k] = ((unsigned)(part * groups + g) * (unsigned)p.xpsPlane + (unsigned)(iy * p.Win + ix)) * 16u;
    }
    auto patch_dma = [&](int ks, int buf) {
        if (p.dbg & 2) return;                                               // (diagnostics: no patch requests)
        const unsigned so = (unsigned)(2 * ks) * (unsigned)p.xpsPlane * 16u;
        const unsigned dst = pAddr + (unsigned)buf * (UP_PUNITS * 16);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int pc = wave + 4 * k;
            if (pc < UP_PPIECES - 1) up_dma16(p.xps, poff[k] + so, dst + (unsigned)pc * 1024u);
            else if (pc == UP_PPIECES - 1 && lane < UP_PUNITS - 64 * (UP_PPIECES - 1)) up_dma16(p.xps, poff[k] + so, dst + (unsigned)pc * 1024u);
        }
    };
    // ---- weights of step g (image m, k-step ks, row dy): 12 pieces = (plane: hi | lo) x (tap dx) x (lane half); lane = output channel;
    //      three requests per wave
    auto weight_dma = [&](int g) {
        if (p.dbg & 4) return;                                               // (diagnostics: no weight requests)
        const int m = g / (3 * K), rem = g - m * (3 * K);
        const int ks = rem / 3, dy = rem - 3 * ks;
        const unsigned dst = wAddr + (unsigned)(g % UP_WBUFS) * (UP_WUNITS * 16);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int pc = wave + 4 * k;
            const int plane = pc / 6, dx = (pc % 6) >> 1, hh = pc & 1;
            const int tk = (3 * dy + dx) * K + ks;
            const unsigned unit = (unsigned)(((tk * 2 + plane) * 2 + hh) * CP + m * 64);
            up_dma16(p.wq + 1, unit * 16u + (unsigned)lane * 16u, dst + (unsigned)pc * 1024u);
        }
    };

    const float unscale = reinterpret_cast<const float*>(p.wq)[1];
    const int ogroups = p.Cout >> 3;
    const rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(p.ps, 0, (int)((size_t)2 * ogroups * p.psPlane * 16), 0x00020000);
    unsigned mag = 0u;
    const unsigned lopart = (unsigned)h * (unsigned)(ogroups * p.psPlane) * 16u;
    const bool six = wave < 2;                                               // this wave's patch requests per slice: 6, else 5

    // requests in issue order: W(0), P(0) | step 0: W(2)?? -- no: W(0), P(0), W(1) before the loop; step G issues W(G + 2), then (dy = 0) P(next)
    weight_dma(0);
    patch_dma(0, 0);
    weight_dma(1);
    int G = 0;                                                               // step counter of the tile
#pragma unroll 1
    for (int m = 0; m < 4; ++m) {
        f32x16 acc[2][2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[cb][r][i] = 0.0f;
#pragma unroll 1
        for (int ks = 0; ks < K; ++ks) {
            const int q = m * K + ks;                                        // slice counter: patch buffer q & 1
#pragma unroll 1
            for (int dy = 0; dy < 3; ++dy, ++G) {
                // Step G needs W(G) (requested when step G - 2 opened) and its patch slice (requested earlier still).  Younger, and
                // allowed to stay in flight: W(G + 1) (3 requests), a patch slice requested behind W(G) or W(G + 1) (dy = 2 / dy = 1:
                // 6 or 5), and the 16 stores of an epilogue that ran between step G - 1 and this one (dy = 0, ks = 0, m > 0).
                //   (the tile's last slice requests no further slice: dy = 1 leaves W(G + 1) alone in flight, dy = 2 nothing)
                if ((p.dbg & 32) || G + 1 >= STEPS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else if (dy == 0) {
                    if (ks == 0 && m > 0 && !(p.dbg & 8)) asm volatile("s_waitcnt vmcnt(19)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                } else if (q + 1 >= 4 * K) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                else if (six) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                if (!(p.dbg & 64)) __syncthreads();                          // everyone's requests for this step have landed; everyone is done with step G - 1 (64: diagnostics, no barrier)
                if (G == 0) lap(1);
                if (G + 2 < STEPS) weight_dma(G + 2);                        // its buffer was step G - 1's
                if (dy == 0 && q + 1 < 4 * K) patch_dma(ks + 1 < K ? ks + 1 : 0, (q + 1) & 1);      // its buffer was slice q - 1's
                if (!(p.dbg & 1)) {
                    const u32x4* wl = wbuf0 + (G % UP_WBUFS) * UP_WUNITS + h * 64 + j;
                    const u32x4* bl = pbuf0 + (q & 1) * UP_PUNITS + h * SP_PIX + (wave * 2 + dy) * SP_W + j;
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const f16x8 a0h = __builtin_bit_cast(f16x8, wl[dx * 128]);
                        const f16x8 a0l = __builtin_bit_cast(f16x8, wl[UP_WROW + dx * 128]);
                        const f16x8 a1h = __builtin_bit_cast(f16x8, wl[dx * 128 + 32]);
                        const f16x8 a1l = __builtin_bit_cast(f16x8, wl[UP_WROW + dx * 128 + 32]);
                        const f16x8 a0s = a0h * (_Float16)0.00048828125f;   // w_hi 2^-11: partner of the scaled x_lo'
                        const f16x8 a1s = a1h * (_Float16)0.00048828125f;
#pragma unroll
                        for (int r = 0; r < 2; ++r) {
                            const f16x8 bh = __builtin_bit_cast(f16x8, bl[r * SP_W + dx]);
                            const f16x8 bo = __builtin_bit_cast(f16x8, bl[UP_PART + r * SP_W + dx]);
                            acc[0][r] = mfma16(a0l, bh, acc[0][r]);
                            acc[0][r] = mfma16(a0s, bo, acc[0][r]);
                            acc[0][r] = mfma16(a0h, bh, acc[0][r]);
                            acc[1][r] = mfma16(a1l, bh, acc[1][r]);
                            acc[1][r] = mfma16(a1s, bo, acc[1][r]);
                            acc[1][r] = mfma16(a1h, bh, acc[1][r]);
                        }
                    }
                }
            }
        }
        lap(2 + 2 * m);
        // ---- epilogue of parity (py, px): act(acc 2^-S + bias) as (hi, lo') units at the high-resolution pixel (2 y + py, 2 x + px);
        //      the frame pixels are ups_frame_kernel's.  Lane pairs trade halves as in split_epilogue_ps: one 16-byte store per lane.
        //      (The next image's first operands are already on their way.)
        if (p.dbg & 8) {
            if (acc[0][0][0] == 123.456f) p.ps[0] = u32x4{1u, 2u, 3u, 4u};
            continue;
        }
        const int py = m >> 1, px = m & 1;
        const int lx = ox0 + j, X = 2 * lx + px;
        // All bias values into registers BEFORE the first store, behind a scheduling fence.  Found the hard way (round 5): with the bias
        // read from LDS inside the loop the compiler placed `ds_read_b128 v[10:13]` directly behind `buffer_store_dwordx4 v[10:13]`; a
        // 16-byte store reads its data registers a little AFTER it issues, the LDS return is not ordered against that, and with two
        // workgroups per CU (a busier memory pipeline) the first dword of the unit -- 8 lanes of one channel -- was overwritten by bias
        // bits before the store had taken it.  The hazard recogniser guards vector-ALU writes behind wide stores, not LDS returns.
        // (Round 6: the LDS-return reading never reproduced in isolation; what does reproduce is a VECTOR write in the issue slot behind a
        //  16-byte store whose soffset is an SGPR -- the one case the recogniser does not pad: tools/probes/store_valu_overwrite_probe.hip.
        //  The stores below carry their plane offset in the vector offset now.)
        float bvv[2][4][4];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int gi = 0; gi < 4; ++gi) {
                const float4 bq = *reinterpret_cast<const float4*>(biasl + cb * 32 + 8 * gi + 4 * h);
                bvv[cb][gi][0] = bq.x; bvv[cb][gi][1] = bq.y; bvv[cb][gi][2] = bq.z; bvv[cb][gi][3] = bq.w;
            }
        if (!UP_BIAS_IN_LOOP) {
            __builtin_amdgcn_s_waitcnt(0xC07F);                              // lgkmcnt(0): the LDS returns have landed
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int ly = oy0 + wave * 2 + r, Y = 2 * ly + py;
            const bool inside = ly < p.Hin && lx < p.Win && Y > 0 && Y < p.H - 1 && X > 0 && X < p.W - 1;
            const unsigned voff = inside ? (unsigned)(Y * p.W + X) * 16u + lopart : BAD_OFFSET;
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
#pragma unroll
                for (int gi = 0; gi < 4; ++gi) {
                    f16x4 th, tl;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = acc[cb][r][4 * gi + e] * unscale + (UP_BIAS_IN_LOOP ? biasl[cb * 32 + 8 * gi + 4 * h + e] : bvv[cb][gi][e]);   // (UP_BIAS_IN_LOOP: round-6 investigation, the first version's bias read)
                        if (p.act == ISR_ACT_RELU) v = v > 0.f ? v : 0.f;
                        else if (p.act == ISR_ACT_LEAKY) v = v > 0.f ? v : v * p.slope;
                        _Float16 a, b;
                        split16x(v, a, b);
                        th[e] = a; tl[e] = b;
                        if (inside) mag = isr_umax(mag, isr_mag(v));
                    }
                    const int g = cb * 4 + gi;
                    const u32x2 uh = __builtin_bit_cast(u32x2, th), ul = __builtin_bit_cast(u32x2, tl);
                    const u32x2 s0 = __builtin_amdgcn_permlane32_swap(uh.x, ul.x, false, false);
                    const u32x2 s1 = __builtin_amdgcn_permlane32_swap(uh.y, ul.y, false, false);
                    const u32x4 unit = {s0.x, s1.x, s0.y, s1.y};            // h = 0: channels 8 g .. + 7 hi; h = 1: the same channels' lo'
                    __builtin_amdgcn_raw_buffer_store_b128(unit, prs, (int)(((p.dbg & 16) || voff == BAD_OFFSET) ? BAD_OFFSET : voff + (unsigned)(g * p.psPlane * 16)), 0, 0);   // (soffset 0: sr_split_common.h)
                }
            }
        }
        lap(3 + 2 * m);
    }
    isr_range_note(p.absmax, mag);
}

// ---- form Q: a larger register tile, activations straight from L1 -----------------------------------------------------------------
// What bounds the kernel above is operand traffic through LDS: 0.67 ds_read_b128 per MFMA (2 rows x 64 channels per wave: 4 weight +
// 4 activation fragments per 12 MFMAs) + the DMA's LDS writes (profiles/r05_upsp_ablation.md).  Here a wave owns FOUR rows of 32 pixels
// (tile 16 x 32 low-resolution pixels, 128 accumulator registers) and walks the taps column by column (dx outer, dy inner): the six
// activation rows a column needs stay in registers across its three vertical taps, and they come from the packed-split tensor by
// ordinary 16-byte loads (a lane's B fragment IS one unit of the tensor; neighbouring lanes read neighbouring units; the three columns
// and the neighbouring waves' rows re-read the same lines: L1 hits) -- each row is reloaded for the next column right behind its last
// use.  LDS holds weights only: 12 ds_read_b128 per 72 MFMAs (0.17), three rotating 12 KB rows by DMA, one barrier per 72 MFMAs.
// Tap order is (k-step, dx, dy): not the other kernels' order -- rounding-level differences, as the phase decomposition has anyway.
constexpr int UQ_ROWS = 4;                                                   // output rows per wave
constexpr int UQ_TILE_H = 4 * UQ_ROWS;                                       // 16 low-resolution rows per workgroup
constexpr int UQ_LDS_BYTES = UP_WBUFS * UP_WUNITS * 16 + 256;                // weights + the bias row: 37 120 B

__global__ __launch_bounds__(S_THREADS, 2) void conv3x3_split_upsq_kernel(const SplitConvParams p)
{
    extern __shared__ u32x4 lds[];
    u32x4* const wbuf0 = lds;                                                // [UP_WBUFS][UP_WUNITS]
    float* const biasl = reinterpret_cast<float*>(lds + UP_WBUFS * UP_WUNITS);
    const unsigned wAddr = (unsigned)(uintptr_t)(up_lds_char*)lds;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    int bid;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    }
    const int tx = bid % p.tilesX, ty = bid / p.tilesX;
    const int oy0 = ty * UQ_TILE_H + wave * UQ_ROWS, ox0 = tx * ST_W;       // this WAVE's first low-resolution row, the tile's first column
    const int groups = p.Cin >> 3;
    constexpr int K = 4, CP = 256, STEPS = 4 * K * 3;
    if (tid < 64) biasl[tid] = p.bias ? p.bias[tid] : 0.0f;
    auto lap = [&](int slot) {
        if (p.stamps && tid == 0) p.stamps[(size_t)blockIdx.x * 10 + slot] = __builtin_amdgcn_s_memrealtime();
    };
    lap(0);

    // weights of step g = (image m, k-step ks, column dx): the three taps (dy, dx) as (plane: hi | lo) x (dy) x (lane half): 12 pieces
    auto weight_dma = [&](int g) {
        if (p.dbg & 4) return;
        const int m = g / (3 * K), rem = g - m * (3 * K);
        const int ks = rem / 3, dx = rem - 3 * ks;
        const unsigned dst = wAddr + (unsigned)(g % UP_WBUFS) * (UP_WUNITS * 16);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int pc = wave + 4 * k;
            const int plane = pc / 6, dy = (pc % 6) >> 1, hh = pc & 1;
            const int tk = (3 * dy + dx) * K + ks;
            const unsigned unit = (unsigned)(((tk * 2 + plane) * 2 + hh) * CP + m * 64);
            up_dma16(p.wq + 1, unit * 16u + (unsigned)lane * 16u, dst + (unsigned)pc * 1024u);
        }
    };
    // activations: row i (0..5 = image rows oy0 - 1 + i, clamped) of column dx, group 2 ks + h, parts hi / lo' -- one 16-byte load per lane
    const rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(p.xps), 0, (int)((size_t)2 * groups * p.xpsPlane * 16), 0x00020000);
    unsigned rowoff[UQ_ROWS + 2], coloff[3];
#pragma unroll
    for (int i = 0; i < UQ_ROWS + 2; ++i) rowoff[i] = (unsigned)(min(max(oy0 - 1 + i, 0), p.Hin - 1) * p.Win) * 16u + (unsigned)h * (unsigned)p.xpsPlane * 16u;
#pragma unroll
    for (int d = 0; d < 3; ++d) coloff[d] = (unsigned)min(max(ox0 + j + d - 1, 0), p.Win - 1) * 16u;
    const unsigned lostep = (unsigned)groups * (unsigned)p.xpsPlane * 16u;   // hi -> lo' part
    u32x4 bh[UQ_ROWS + 2], bo[UQ_ROWS + 2];
    auto bload = [&](int i, int g) {                                         // row i for step g
        const int rem = g % (3 * K), ks = rem / 3, dx = rem - 3 * ks;
        const unsigned vo = rowoff[i] + (dx == 0 ? coloff[0] : dx == 1 ? coloff[1] : coloff[2]);
        const unsigned so = (unsigned)(2 * ks) * (unsigned)p.xpsPlane * 16u;
        if (p.dbg & 2) return;
        bh[i] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)vo, (int)so, 0);
        bo[i] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)vo, (int)(so + lostep), 0);
    };

    const float unscale = reinterpret_cast<const float*>(p.wq)[1];
    const int ogroups = p.Cout >> 3;
    const rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(p.ps, 0, (int)((size_t)2 * ogroups * p.psPlane * 16), 0x00020000);
    unsigned mag = 0u;
    const unsigned lopart = (unsigned)h * (unsigned)(ogroups * p.psPlane) * 16u;

    weight_dma(0);
    weight_dma(1);
#pragma unroll
    for (int i = 0; i < UQ_ROWS + 2; ++i) { bh[i] = u32x4{0u, 0u, 0u, 0u}; bo[i] = bh[i]; bload(i, 0); }
    int G = 0;
#pragma unroll 1
    for (int m = 0; m < 4; ++m) {
        f32x16 acc[2][UQ_ROWS];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < UQ_ROWS; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[cb][r][i] = 0.0f;
#pragma unroll 1
        for (int s = 0; s < 3 * K; ++s, ++G) {
            // W(G) has landed (requested two steps ago; everything this wave requested since may stay in flight -- simply: all of it
            // has had a step's time), everyone is done with step G - 1
            // Requests complete in issue order; behind W(G) this wave has issued: the 12 row reloads of step G - 2, W(G + 1) (3), the
            // 12 reloads of step G - 1 -- and the 32 stores of an epilogue that ran in between.  All of that may stay in flight.
            if (p.dbg) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (G == 0) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
            else if (s == 0) asm volatile("s_waitcnt vmcnt(59)" ::: "memory");
            else if (G + 1 >= STEPS) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(27)" ::: "memory");
            if (!(p.dbg & 64)) __syncthreads();
            if (G == 0) lap(1);
            if (G + 2 < STEPS) weight_dma(G + 2);
            const u32x4* wl = wbuf0 + (G % UP_WBUFS) * UP_WUNITS + h * 64 + j;
            const int gn = G + 1 < STEPS ? G + 1 : G;                        // the step the rows are reloaded for
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const f16x8 a0h = __builtin_bit_cast(f16x8, wl[dy * 128]);
                const f16x8 a0l = __builtin_bit_cast(f16x8, wl[UP_WROW + dy * 128]);
                const f16x8 a1h = __builtin_bit_cast(f16x8, wl[dy * 128 + 32]);
                const f16x8 a1l = __builtin_bit_cast(f16x8, wl[UP_WROW + dy * 128 + 32]);
                const f16x8 a0s = a0h * (_Float16)0.00048828125f;
                const f16x8 a1s = a1h * (_Float16)0.00048828125f;
#pragma unroll
                for (int r = 0; r < UQ_ROWS; ++r) {
                    if (!(p.dbg & 1)) {
                        const f16x8 vh = __builtin_bit_cast(f16x8, bh[r + dy]);
                        const f16x8 vo = __builtin_bit_cast(f16x8, bo[r + dy]);
                        acc[0][r] = mfma16(a0l, vh, acc[0][r]);
                        acc[0][r] = mfma16(a0s, vo, acc[0][r]);
                        acc[0][r] = mfma16(a0h, vh, acc[0][r]);
                        acc[1][r] = mfma16(a1l, vh, acc[1][r]);
                        acc[1][r] = mfma16(a1s, vo, acc[1][r]);
                        acc[1][r] = mfma16(a1h, vh, acc[1][r]);
                    }
                    // a row is reloaded for the next step right behind its last use: rows 0, 1, 2 after (dy, r = 0), rows 3, 4, 5 after (dy = 2, r)
                    if (r == 0) bload(dy, gn);
                    else if (dy == 2) bload(r + 2, gn);
                }
            }
        }
        lap(2 + 2 * m);
        if (p.dbg & 8) {
            if (acc[0][0][0] == 123.456f) p.ps[0] = u32x4{1u, 2u, 3u, 4u};
            lap(3 + 2 * m);
            continue;
        }
        // ---- epilogue of parity (py, px), as conv3x3_split_upsp_kernel's (bias in registers before the first store)
        const int py = m >> 1, px = m & 1;
        const int lx = ox0 + j, X = 2 * lx + px;
        float bvv[2][4][4];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int gi = 0; gi < 4; ++gi) {
                const float4 bq = *reinterpret_cast<const float4*>(biasl + cb * 32 + 8 * gi + 4 * h);
                bvv[cb][gi][0] = bq.x; bvv[cb][gi][1] = bq.y; bvv[cb][gi][2] = bq.z; bvv[cb][gi][3] = bq.w;
            }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < UQ_ROWS; ++r) {
            const int ly = oy0 + r, Y = 2 * ly + py;
            const bool inside = ly < p.Hin && lx < p.Win && Y > 0 && Y < p.H - 1 && X > 0 && X < p.W - 1;
            const unsigned voff = inside ? (unsigned)(Y * p.W + X) * 16u + lopart : BAD_OFFSET;
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
#pragma unroll
                for (int gi = 0; gi < 4; ++gi) {
                    f16x4 th, tl;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = acc[cb][r][4 * gi + e] * unscale + bvv[cb][gi][e];
                        if (p.act == ISR_ACT_RELU) v = v > 0.f ? v : 0.f;
                        else if (p.act == ISR_ACT_LEAKY) v = v > 0.f ? v : v * p.slope;
                        _Float16 a, b;
                        split16x(v, a, b);
                        th[e] = a; tl[e] = b;
                        if (inside) mag = isr_umax(mag, isr_mag(v));
                    }
                    const int g = cb * 4 + gi;
                    const u32x2 uh = __builtin_bit_cast(u32x2, th), ul = __builtin_bit_cast(u32x2, tl);
                    const u32x2 s0 = __builtin_amdgcn_permlane32_swap(uh.x, ul.x, false, false);
                    const u32x2 s1 = __builtin_amdgcn_permlane32_swap(uh.y, ul.y, false, false);
                    const u32x4 unit = {s0.x, s1.x, s0.y, s1.y};
                    __builtin_amdgcn_raw_buffer_store_b128(unit, prs, (int)(((p.dbg & 16) || voff == BAD_OFFSET) ? BAD_OFFSET : voff + (unsigned)(g * p.psPlane * 16)), 0, 0);   // (soffset 0: sr_split_common.h)
                }
            }
        }
        lap(3 + 2 * m);
    }
    isr_range_note(p.absmax, mag);
}

// ---- the output's one-pixel frame: the convolution as it is defined (interpolate, zero-pad, nine taps), per pixel and group of eight
//      output channels in fp32 -- a k-ordered FMA chain over (tap, input channel) on U(x), x = hi + lo' 2^-11 of the packed input.
struct UpsFrameParams {
    const u32x4* xps; int xpsPlane;          // [2][8][xpsPlane] low-resolution input
    const float* w;                          // [64][64][3][3] the layer's weights (fp32, as the module holds them)
    const float* bias;
    u32x4* ps; int psPlane;                  // [2][8][psPlane] high-resolution output
    int Hin, Win, H, W;
    int act; float slope;
    unsigned* absmax;
};

__global__ __launch_bounds__(64) void ups_frame_kernel(const UpsFrameParams p)
{
    const int f = blockIdx.x * 64 + threadIdx.x;
    const int g = blockIdx.y;                                                // output channels 8 g .. 8 g + 7 (uniform: weights come through scalar loads)
    const int nf = 2 * p.W + 2 * (p.H - 2);
    unsigned mag = 0u;
    if (f < nf) {
        int Y, X;
        if (f < p.W) { Y = 0; X = f; }
        else if (f < 2 * p.W) { Y = p.H - 1; X = f - p.W; }
        else { const int q = f - 2 * p.W; Y = 1 + (q >> 1); X = (q & 1) ? p.W - 1 : 0; }
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = p.bias ? p.bias[8 * g + e] : 0.0f;
        const _Float16* const xh = reinterpret_cast<const _Float16*>(p.xps);
        const size_t lopart = (size_t)8 * p.xpsPlane * 8;                    // in halves: the lo' planes behind the 8 hi planes
        for (int dy = -1; dy <= 1; ++dy) {
            const int Yt = Y + dy;
            if ((unsigned)Yt >= (unsigned)p.H) continue;                     // zero padding of the convolution
            int y0, y1; float ly;
            isr_src_index(Yt, 0.5f, p.Hin, y0, y1, ly);
            for (int dx = -1; dx <= 1; ++dx) {
                const int Xt = X + dx;
                if ((unsigned)Xt >= (unsigned)p.W) continue;
                int x0, x1; float lx;
                isr_src_index(Xt, 0.5f, p.Win, x0, x1, lx);
                const float hy = 1.f - ly, hx = 1.f - lx;
                const int tap = (dy + 1) * 3 + dx + 1;
                for (int gi = 0; gi < 8; ++gi) {                             // input channel group: one 16-byte unit per source pixel and part
                    const size_t base = (size_t)gi * p.xpsPlane * 8;
                    const size_t o00 = base + (size_t)(y0 * p.Win + x0) * 8, o01 = base + (size_t)(y0 * p.Win + x1) * 8;
                    const size_t o10 = base + (size_t)(y1 * p.Win + x0) * 8, o11 = base + (size_t)(y1 * p.Win + x1) * 8;
                    const f16x8 h00 = *reinterpret_cast<const f16x8*>(xh + o00), l00 = *reinterpret_cast<const f16x8*>(xh + lopart + o00);
                    const f16x8 h01 = *reinterpret_cast<const f16x8*>(xh + o01), l01 = *reinterpret_cast<const f16x8*>(xh + lopart + o01);
                    const f16x8 h10 = *reinterpret_cast<const f16x8*>(xh + o10), l10 = *reinterpret_cast<const f16x8*>(xh + lopart + o10);
                    const f16x8 h11 = *reinterpret_cast<const f16x8*>(xh + o11), l11 = *reinterpret_cast<const f16x8*>(xh + lopart + o11);
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        const float a = (float)h00[c] + (float)l00[c] * 0.00048828125f, b = (float)h01[c] + (float)l01[c] * 0.00048828125f;
                        const float cc = (float)h10[c] + (float)l10[c] * 0.00048828125f, d = (float)h11[c] + (float)l11[c] * 0.00048828125f;
                        const float u = hy * (hx * a + lx * b) + ly * (hx * cc + lx * d);
                        const float* wr = p.w + ((size_t)(8 * g) * 64 + (8 * gi + c)) * 9 + tap;      // w[8 g + e][8 gi + c][tap]
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[e] = __builtin_fmaf(u, wr[(size_t)e * 64 * 9], acc[e]);
                    }
                }
            }
        }
        f16x8 qh, ql;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = acc[e];
            if (p.act == ISR_ACT_RELU) v = v > 0.f ? v : 0.f;
            else if (p.act == ISR_ACT_LEAKY) v = v > 0.f ? v : v * p.slope;
            _Float16 a, b;
            split16x(v, a, b);
            qh[e] = a; ql[e] = b;
            mag = isr_umax(mag, isr_mag(v));
        }
        const size_t pix = (size_t)Y * p.W + X;
        p.ps[(size_t)g * p.psPlane + pix] = __builtin_bit_cast(u32x4, qh);
        p.ps[(size_t)(8 + g) * p.psPlane + pix] = __builtin_bit_cast(u32x4, ql);
    }
    isr_range_note(p.absmax, mag);
}

// W_eff of the four parities, stacked as a [256][64][3][3] weight tensor (image m = 2 py + px at output channels 64 m ..): fp64 sums
// of at most four products with the coefficients 9/16, 3/16, 1/16 (exact in binary), one rounding to fp32
__global__ __launch_bounds__(256) void ups_phase_weights_kernel(const float* __restrict__ w, float* __restrict__ weff, int Cout, int Cin)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;                          // (m, co, ci)
    if (idx >= 4 * Cout * Cin) return;
    const int ci = idx % Cin, co = (idx / Cin) % Cout, m = idx / (Cin * Cout);
    const int py = m >> 1, px = m & 1;
    const double A[2][3][3] = { { {0.75, 0.25, 0.0}, {0.25, 0.75, 0.0}, {0.0, 0.75, 0.25} },
                                { {0.25, 0.75, 0.0}, {0.0, 0.75, 0.25}, {0.0, 0.25, 0.75} } };
    const float* src = w + ((size_t)co * Cin + ci) * 9;
    float* dst = weff + (((size_t)m * Cout + co) * Cin + ci) * 9;
    for (int r = 0; r < 3; ++r)
        for (int s = 0; s < 3; ++s) {
            double sum = 0.0;
            for (int dy = 0; dy < 3; ++dy)
                for (int dx = 0; dx < 3; ++dx) sum += (double)src[dy * 3 + dx] * A[py][dy][r] * A[px][dx][s];
            dst[r * 3 + s] = (float)sum;
        }
}

} // namespace
