// conv3x3(U(x)) for EnhanceNet's two upsampling layers (SuperresolutionNetwork/models/enhancenet.py:113-124) on a FOUR-ROW register tile:
// 16 x 32 output tiles, a wave = 4 rows x 32 pixels x 64 output channels (128 accumulator registers), two workgroups per CU (VERDICT r05
// item 1b).  Included by sr_conv_split.hip; the arithmetic -- interpolation, split, products, their order per accumulator -- is
// conv3x3_split_kernel<true>'s and conv3x3_split_ups3_kernel's: bit-identical (tests/test_ups_gpu.py).
//
// Why (profiles/r06_lds_mfma_ratio.txt, tools/probes/lds_mfma_ratio_probe.hip).  The LDS array delivers 256 B/clk/CU: ds_read_b128 up to two per
// MFMA cost nothing when they are PREFETCHED (32.3 cycles per MFMA at 2.0 reads per MFMA), but a fragment set that is read and waited for directly
// in front of its MFMAs costs 49-61 cycles per MFMA with one wave per SIMD and 38 with two.  The three-per-CU kernel (sr_conv_ups3.h) has no
// registers left to prefetch with (168 of 168) and re-reads its weight fragments into the registers the previous tap just released.  Here:
//   * a tap's twelve operand fragments (4 weight, 8 patch) are read while the tap BEFORE it multiplies (two register sets, 96 registers), so the
//     only exposed LDS round trip is the first tap of a tap row (one per 72 MFMAs instead of six per 36);
//   * a weight fragment serves 12 MFMAs instead of 6 (0.5 instead of 0.67 - 0.83 fragment reads per MFMA);
//   * the tile's halo is 18 x 34 / (16 x 32) = 1.20 instead of 1.33: 10 % less interpolation + split work per output pixel;
//   * 4 080 tiles on 512 slots at 1080p = 7.97 rounds (the 8 x 32 tiling: 10.55 rounds on 768 slots, the last one half empty).
// One k-step (16 channels) of the patch is 18 x 34 x 2 groups x (hi, lo') = 39 168 B, two weight slots of a tap row each 24 576 B: 63 744 B.
#pragma once
#include "sr_split_common.h"

#ifndef U4R_DIAG
#define U4R_DIAG 1
#endif

namespace {

constexpr int U4R_TH = 16, U4R_TW = 32;                                      // output tile
constexpr int U4R_PH = U4R_TH + 2, U4R_PW = U4R_TW + 2, U4R_PIX = U4R_PH * U4R_PW;   // 18 x 34 = 612 patch pixels
constexpr int U4R_PART = 2 * U4R_PIX;                                        // one k-step of the patch: 2 channel groups; hi, then lo' at + U4R_PART
constexpr int U4R_PUNITS = 2 * U4R_PART;                                     // 2448 units = 39 168 B
constexpr int U4R_WROW = 3 * 128;                                            // one tap row of one part: 3 taps x [lane half][64 couts]
constexpr int U4R_WUNITS = 2 * U4R_WROW;                                     // hi, lo: 12 288 B
constexpr int U4R_LDS_BYTES = (U4R_PUNITS + 2 * U4R_WUNITS) * 16;            // 63 744: two workgroups per CU
constexpr int U4R_LR_H = U4R_TH / 2 + 2, U4R_LR_W = U4R_TW / 2 + 2;          // 10 x 18 low-resolution pixels
constexpr int U4R_LQ = (U4R_TW / 2 + 8) / 4;                                 // 6 aligned quads per low-resolution row
constexpr int U4R_LUNITS = 16 * U4R_LR_H * U4R_LQ;                           // 960 quads: 3.75 per thread
constexpr int U4R_LR_CS = 186;                                               // channel stride of the fp32 copy (>= 180; = 2 mod 8: the four channel groups of a half-wave hit 32 banks)
constexpr int U4R_QR = U4R_PH / 2, U4R_QC = U4R_PW / 2, U4R_UQ = U4R_QR * U4R_QC;    // 9 x 17 quads of 2 x 2 patch pixels
static_assert(16 * U4R_LR_CS * 4 <= U4R_WUNITS * 16, "the fp32 copy of the low-resolution region lives in the idle weight slot");

__global__ __launch_bounds__(S_THREADS, 2) void conv3x3_split_ups4r_kernel(const SplitConvParams p)
{
    extern __shared__ u32x4 patch[];                                         // the k-step's patch, then two weight slots
    u32x4* const wbuf = patch + U4R_PUNITS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    int bid;
    {   // an XCD (= an L2) gets a contiguous range of tiles
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    }
    const int tx = bid % p.tilesX; bid /= p.tilesX;
    const int ty = bid % p.tilesY, n = bid / p.tilesY;
    const int oy0 = ty * U4R_TH, ox0 = tx * U4R_TW;
    const rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x + (size_t)n * p.xImage), 0,
                                                         (int)((size_t)p.Cin * p.xPlane * 4), 0x00020000);
    const unsigned planeBytes = (unsigned)p.xPlane * 4u;
    const rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(p.wq + 1), 0, 9 * p.ksteps * 4096, 0x00020000);

    // tap row `step` = 3 ks + dy into weight slot step & 1 by LDS-DMA: 12 wave-wide pieces (tap i, part, half), three per wave
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    auto wdma = [&](int step) {
        if (step >= 3 * p.ksteps) return;
        const int ks = step / 3, dy = step - 3 * ks;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int idx = wv + 4 * k, i = idx >> 2, part = (idx >> 1) & 1, half = idx & 1;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (isr_lvoid_t*)(wbuf + (step & 1) * U4R_WUNITS + part * U4R_WROW + i * 128 + half * 64), 16,
                                                     lane * 16, (((3 * dy + i) * p.ksteps + ks) * 256 + part * 128 + half * 64) * 16, 0, 0);
        }
    };

    // ---- staging of one k-step (16 channels): low-resolution region -> fp32 copy on the idle weight slot -> interpolate, split
    float* tmp = reinterpret_cast<float*>(wbuf);
    const int ly0 = oy0 / 2 - 1, lx0 = ox0 / 2 - 1;
    u32x4 v[4];
    // (thread-derived indices are re-derived at every use from a laundered copy of the thread id: hoisted out of the k-loop they would sit in
    // ~40 registers across the MFMA rows, which have none to spare)
    auto fresh_tid = [&]() { int t = tid; asm volatile("" : "+v"(t)); return t; };
    auto lfetch = [&](int cin0) {                                            // requests only: the values are parked after a barrier
        const int t = fresh_tid();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int u = t + k * S_THREADS;
            const int c = u / (U4R_LR_H * U4R_LQ), rem = u - c * (U4R_LR_H * U4R_LQ);
            const int r = rem / U4R_LQ, q = rem - r * U4R_LQ;
            const int iy = ly0 + r, ix = ox0 / 2 - 4 + 4 * q;
            const bool ok = u < U4R_LUNITS && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
            v[k] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)(ok ? (unsigned)(cin0 + c) * planeBytes + (unsigned)(iy * p.Win + ix) * 4u
                                                                       : BAD_OFFSET), 0, 0);
        }
    };
    auto lpark = [&]() {
        const int t = fresh_tid();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int u = t + k * S_THREADS;
            if (u >= U4R_LUNITS) continue;
            const int c = u / (U4R_LR_H * U4R_LQ), rem = u - c * (U4R_LR_H * U4R_LQ);
            const int r = rem / U4R_LQ, q = rem - r * U4R_LQ;
            const float4 f = __builtin_bit_cast(float4, v[k]);
            float* dst = tmp + c * U4R_LR_CS + r * U4R_LR_W + 4 * q - 3;     // quad q holds low-res patch columns 4q - 3 .. 4q
            if (q > 0) dst[0] = f.x;
            if (q > 0 && q < U4R_LQ - 1) { dst[1] = f.y; dst[2] = f.z; }
            if (q < U4R_LQ - 1) dst[3] = f.w;
        }
    };
    // INTERIOR tiles: every patch pixel lies in the image and no source index is clamped -- the blend weights are the constants 3/4, 1/4
    // (sr_conv_ups3.h); the same operations on the same values in the same order as isr_src_index gives there: bit-identical.
    const bool interior = oy0 >= 2 && oy0 + U4R_TH + 2 <= p.H && ox0 >= 2 && ox0 + U4R_TW + 2 <= p.W;
    auto interpolate = [&]() {
        _Float16* const patch16 = reinterpret_cast<_Float16*>(patch);
        for (int u = fresh_tid(); u < 4 * U4R_UQ; u += S_THREADS) {
            const int g4 = u & 3, q = u >> 2;                                // neighbouring lanes: the 4 four-channel groups of one quad
            const int kr = q / U4R_QC, kc = q - kr * U4R_QC;
            f16x4 h00, h01, h10, h11, l00, l01, l10, l11;
            if (interior) {
                const float* ta = tmp + (g4 * 4) * U4R_LR_CS + kr * U4R_LR_W + kc;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float a0 = ta[e * U4R_LR_CS], a1 = ta[e * U4R_LR_CS + 1];
                    const float b0 = ta[e * U4R_LR_CS + U4R_LR_W], b1 = ta[e * U4R_LR_CS + U4R_LR_W + 1];
                    const float al = isr_blend(0.75f, a0, 0.25f, a1), ar = isr_blend(0.25f, a0, 0.75f, a1);
                    const float bl = isr_blend(0.75f, b0, 0.25f, b1), br = isr_blend(0.25f, b0, 0.75f, b1);
                    _Float16 vh, vl;
                    split16x(isr_blend(0.75f, al, 0.25f, bl), vh, vl); h00[e] = vh; l00[e] = vl;
                    split16x(isr_blend(0.75f, ar, 0.25f, br), vh, vl); h01[e] = vh; l01[e] = vl;
                    split16x(isr_blend(0.25f, al, 0.75f, bl), vh, vl); h10[e] = vh; l10[e] = vl;
                    split16x(isr_blend(0.25f, ar, 0.75f, br), vh, vl); h11[e] = vh; l11[e] = vl;
                }
            } else {
                const int Yu = oy0 - 1 + 2 * kr, Xl = ox0 - 1 + 2 * kc;
                const bool oku = (unsigned)Yu < (unsigned)p.H, okd = (unsigned)(Yu + 1) < (unsigned)p.H;
                const bool okl = (unsigned)Xl < (unsigned)p.W, okr = (unsigned)(Xl + 1) < (unsigned)p.W;
                int y0, y1, x0, x1, t0, t1; float lyu, lyd, lxl, lxr, t;
                isr_src_index(oku ? Yu : Yu + 1, 0.5f, p.Hin, y0, y1, t);     // both rows of the pair blend these two source rows
                isr_src_index(okl ? Xl : Xl + 1, 0.5f, p.Win, x0, x1, t);
                isr_src_index(Yu, 0.5f, p.Hin, t0, t1, lyu);
                isr_src_index(Yu + 1, 0.5f, p.Hin, t0, t1, lyd);
                isr_src_index(Xl, 0.5f, p.Win, t0, t1, lxl);
                isr_src_index(Xl + 1, 0.5f, p.Win, t0, t1, lxr);
                const float hyu = 1.f - lyu, hyd = 1.f - lyd, hxl = 1.f - lxl, hxr = 1.f - lxr;
                // rows / columns wholly outside the image (tile overhang) keep their indices inside the staged region
                y0 = min(max(y0 - ly0, 0), U4R_LR_H - 1); y1 = min(max(y1 - ly0, 0), U4R_LR_H - 1);
                x0 = min(max(x0 - lx0, 0), U4R_LR_W - 1); x1 = min(max(x1 - lx0, 0), U4R_LR_W - 1);
                const float* ta = tmp + (g4 * 4) * U4R_LR_CS + y0 * U4R_LR_W;
                const float* tb = tmp + (g4 * 4) * U4R_LR_CS + y1 * U4R_LR_W;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float a0 = ta[e * U4R_LR_CS + x0], a1 = ta[e * U4R_LR_CS + x1];
                    const float b0 = tb[e * U4R_LR_CS + x0], b1 = tb[e * U4R_LR_CS + x1];
                    const float al = isr_blend(hxl, a0, lxl, a1), ar = isr_blend(hxr, a0, lxr, a1);
                    const float bl = isr_blend(hxl, b0, lxl, b1), br = isr_blend(hxr, b0, lxr, b1);
                    _Float16 vh, vl;
                    split16x(isr_blend(hyu, al, lyu, bl), vh, vl); h00[e] = vh; l00[e] = vl;
                    split16x(isr_blend(hyu, ar, lyu, br), vh, vl); h01[e] = vh; l01[e] = vl;
                    split16x(isr_blend(hyd, al, lyd, bl), vh, vl); h10[e] = vh; l10[e] = vl;
                    split16x(isr_blend(hyd, ar, lyd, br), vh, vl); h11[e] = vh; l11[e] = vl;
                }
                const f16x4 z = {(_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f};
                if (!(oku && okl)) { h00 = z; l00 = z; }
                if (!(oku && okr)) { h01 = z; l01 = z; }
                if (!(okd && okl)) { h10 = z; l10 = z; }
                if (!(okd && okr)) { h11 = z; l11 = z; }
            }
            // 16-byte unit (8-channel group g4 / 2, pixel) holds 8 halves: this 4-channel group is its half (g4 & 1)
            _Float16* d = patch16 + ((size_t)((g4 >> 1) * U4R_PIX + (2 * kr) * U4R_PW + 2 * kc)) * 8 + (g4 & 1) * 4;
            *reinterpret_cast<f16x4*>(d) = h00;
            *reinterpret_cast<f16x4*>(d + 8) = h01;
            *reinterpret_cast<f16x4*>(d + U4R_PW * 8) = h10;
            *reinterpret_cast<f16x4*>(d + U4R_PW * 8 + 8) = h11;
            *reinterpret_cast<f16x4*>(d + U4R_PART * 8) = l00;
            *reinterpret_cast<f16x4*>(d + U4R_PART * 8 + 8) = l01;
            *reinterpret_cast<f16x4*>(d + (U4R_PART + U4R_PW) * 8) = l10;
            *reinterpret_cast<f16x4*>(d + (U4R_PART + U4R_PW) * 8 + 8) = l11;
        }
    };

    // acc[half][cb][r]: output row wave * 4 + 2 half + r, channel block cb -- the layout the shared epilogues take two rows at a time
    f32x16 acc[2][2][2];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[hf][cb][r][i] = 0.0f;

    // diagnostics (U4R_DIAG builds): ablation bits of isrDebugSetSplitAblation (1 no MFMAs, 2 no interpolation, 8 no epilogue) and, with a stamp
    // buffer, 8 words per workgroup: start | sum of the staging phases | sum of the MFMA phases | loop end | end (ticks of the 100 MHz clock)
    const int dbg = U4R_DIAG ? p.dbg : 0;
    unsigned long long* const stamps = U4R_DIAG ? p.stamps : nullptr;
    unsigned long long st0 = 0, stStage = 0, stMfma = 0, ta = 0, tb = 0;
    if (stamps) st0 = __builtin_amdgcn_s_memrealtime();
    lfetch(0);
    wdma(0);
#pragma unroll 1
    for (int ks = 0; ks < p.ksteps; ++ks) {
        if (stamps) ta = __builtin_amdgcn_s_memrealtime();
        // row 3 ks multiplies on slot (3 ks) & 1 (in flight or landed); the other slot is idle: the fp32 copy lives there
        tmp = reinterpret_cast<float*>(wbuf + ((3 * ks + 1) & 1) * U4R_WUNITS);
        lpark();
        __syncthreads();
        if (!(dbg & 2)) interpolate();
        if (stamps) { tb = __builtin_amdgcn_s_memrealtime(); stStage += tb - ta; }
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            // this row's weights have landed (every wave's pieces: the wait, then the barrier), the patch is complete (dy = 0) and
            // everybody is done with the slot the next row's weights go to (the row before last read it; dy = 0: the fp32 copy)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            wdma(3 * ks + dy + 1);
            if (dy == 2 && ks + 1 < p.ksteps) lfetch(16 * (ks + 1));
            if (dbg & 1) continue;
            const u32x4* wl = wbuf + ((3 * ks + dy) & 1) * U4R_WUNITS + h * 64 + j;
            const u32x4* bl = patch + h * U4R_PIX + (wave * 4 + dy) * U4R_PW + j;
            // Software-pipelined over the three taps of the row, fragment group by fragment group, so that a tap's operands are requested 16 MFMAs
            // before their first use and land in registers their predecessors have just released (peak: 128 accumulators + 80 operand registers):
            //   G1 = a_lo b_hi (8 MFMAs) | request a_lo, a_hi, b_hi of tap dx + 1 | G2 = a_hi' b_lo' | request b_lo' of tap dx + 1 | G3 = a_hi b_hi
            // -- per accumulator the same three products in the same order as everywhere else (a_lo b_hi, a_hi' b_lo', a_hi b_hi): the same bits.
            f16x8 aH[2], aL[2], bh[4], bo[4];
            aH[0] = __builtin_bit_cast(f16x8, wl[0]); aH[1] = __builtin_bit_cast(f16x8, wl[32]);
            aL[0] = __builtin_bit_cast(f16x8, wl[U4R_WROW]); aL[1] = __builtin_bit_cast(f16x8, wl[U4R_WROW + 32]);
#pragma unroll
            for (int r = 0; r < 4; ++r) { bh[r] = __builtin_bit_cast(f16x8, bl[r * U4R_PW]); bo[r] = __builtin_bit_cast(f16x8, bl[U4R_PART + r * U4R_PW]); }
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                f16x8 aHn[2], aLn[2], bhn[4], bon[4];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 4; ++r) { acc[r >> 1][0][r & 1] = mfma16(aL[0], bh[r], acc[r >> 1][0][r & 1]); acc[r >> 1][1][r & 1] = mfma16(aL[1], bh[r], acc[r >> 1][1][r & 1]); }
                __builtin_amdgcn_sched_barrier(0);
                if (dx < 2) {
                    aLn[0] = __builtin_bit_cast(f16x8, wl[U4R_WROW + (dx + 1) * 128]); aLn[1] = __builtin_bit_cast(f16x8, wl[U4R_WROW + (dx + 1) * 128 + 32]);
                    aHn[0] = __builtin_bit_cast(f16x8, wl[(dx + 1) * 128]); aHn[1] = __builtin_bit_cast(f16x8, wl[(dx + 1) * 128 + 32]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) bhn[r] = __builtin_bit_cast(f16x8, bl[r * U4R_PW + dx + 1]);
                }
                const f16x8 a0s = aH[0] * (_Float16)0.00048828125f;         // w_hi 2^-11: partner of the scaled x_lo'
                const f16x8 a1s = aH[1] * (_Float16)0.00048828125f;
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 4; ++r) { acc[r >> 1][0][r & 1] = mfma16(a0s, bo[r], acc[r >> 1][0][r & 1]); acc[r >> 1][1][r & 1] = mfma16(a1s, bo[r], acc[r >> 1][1][r & 1]); }
                __builtin_amdgcn_sched_barrier(0);
                if (dx < 2) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) bon[r] = __builtin_bit_cast(f16x8, bl[U4R_PART + r * U4R_PW + dx + 1]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 4; ++r) { acc[r >> 1][0][r & 1] = mfma16(aH[0], bh[r], acc[r >> 1][0][r & 1]); acc[r >> 1][1][r & 1] = mfma16(aH[1], bh[r], acc[r >> 1][1][r & 1]); }
                __builtin_amdgcn_sched_barrier(0);
                if (dx < 2) {
                    aH[0] = aHn[0]; aH[1] = aHn[1]; aL[0] = aLn[0]; aL[1] = aLn[1];
#pragma unroll
                    for (int r = 0; r < 4; ++r) { bh[r] = bhn[r]; bo[r] = bon[r]; }
                }
            }
        }
        __syncthreads();                                                     // the patch and the slot the next fp32 copy goes to are free
        if (stamps) stMfma += __builtin_amdgcn_s_memrealtime() - tb;
    }
    unsigned long long st2 = 0;
    if (stamps) st2 = __builtin_amdgcn_s_memrealtime();

    // the shared epilogues take a wave's rows two at a time: rows oy0 + 4 wave + 2 half + r = (oy0 + 2 wave + 2 half) + 2 wave + r
    if (dbg & 8) {
        if (acc[0][0][0][0] == 123.456f) p.ps[0] = u32x4{1u, 2u, 3u, 4u};    // (keeps the accumulators alive)
    } else
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        if (p.ps) split_epilogue_ps(p, acc[hf], oy0 + 2 * wave + 2 * hf, ox0, 0, true, wave, j, h);
        else split_epilogue<true>(p, acc[hf], patch, n, oy0 + 2 * wave + 2 * hf, ox0, 0, true, lane, wave, j, h);
    }
    if (stamps && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long* o = stamps + (size_t)blockIdx.x * 8;
        o[0] = st0; o[1] = stStage; o[2] = stMfma; o[3] = st2; o[4] = __builtin_amdgcn_s_memrealtime();
    }
}

} // namespace

// Launch hook for isrConv3x3ForwardSplit: -1 if this form does not take the layer (64 -> 64 channels, quads, as both of EnhanceNet's are).
static int isr_launch_split_ups4r(const SplitConvParams& p0, hipStream_t s, hipEvent_t e0, hipEvent_t e1)
{
    if (p0.Cin <= 0 || (p0.Cin & 15) || p0.coutPad != 64 || p0.Cout != 64 || p0.cgroups != 1 || p0.xps) return -1;
    if (!p0.ps && ((p0.W | p0.yPlane | p0.rPlane) & 3)) return -1;        // the fp32 epilogue is compiled for quads only
    SplitConvParams p = p0;
    p.tilesY = (p.H + U4R_TH - 1) / U4R_TH;
    const long long nwg = (long long)p.N * p.tilesX * p.tilesY;
    if (nwg > 0x7fffffffLL) return -1;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)conv3x3_split_ups4r_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, U4R_LDS_BYTES); attr = true; }
    if (e0 || e1) hipExtLaunchKernelGGL(conv3x3_split_ups4r_kernel, dim3((unsigned)nwg), dim3(S_THREADS), U4R_LDS_BYTES, s, e0, e1, 0, p);
    else hipLaunchKernelGGL(conv3x3_split_ups4r_kernel, dim3((unsigned)nwg), dim3(S_THREADS), U4R_LDS_BYTES, s, p);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
