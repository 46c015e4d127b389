// conv3x3(U(x)) for EnhanceNet's two upsampling layers (SuperresolutionNetwork/models/enhancenet.py:113-124) with THREE workgroups
// per CU instead of two.  Included by sr_conv_split.hip (same translation unit: the parameter block lives in its anonymous
// namespace); the arithmetic -- interpolation, split, products, their order -- is conv3x3_split_kernel<true>'s: bit-identical.
//
// Why.  The tile kernel's workgroup spends 70 % of its life outside the MFMA loop (staging 9 + 8 us, epilogue 7 of 35 us at 1080p,
// tools/lab/ups_timeline.py) and a CU holds two of them: half the time neither is multiplying (matrix pipe busy 44 %,
// profiles/r03_pmc_ups.md).  What keeps a third workgroup out is LDS (80 KB each: a 32-channel patch + a k-step of weights) and
// registers (184).  Here a workgroup holds ONE k-step of the patch (16 channels, 21.8 KB) and ONE tap row of a k-step's weights
// (3 taps, 12.3 KB): 34 KB and <= 168 registers (3 weight units per thread in flight instead of 9, 3 staging quads), so three
// workgroups share a CU -- at the price of a barrier pair per tap row instead of per k-step.
#pragma once
#include "sr_split_common.h"

// tap row of a k-step under whose MFMAs the NEXT k-step's low-resolution region is requested (0: three rows of cover for bytes that
// come from memory; 2: one row, the registers that hold them live a third as long).  Measured equal (0.617-0.629 ms for the two launches with
// 0, 1 or 2: the other two workgroups of the CU cover the latency): the short lifetime stays.
#ifndef U3_LFETCH_ROW
#define U3_LFETCH_ROW 2
#endif
// U3_SPLANE 1: the partner of the scaled x_lo' (w_hi 2^-11 as fp16) is read from a THIRD weight plane in LDS (round 4: 288 v_pk_mul_f16 per
// wave and tile less); 0: made with v_pk_mul_f16 from the hi fragment, as every other kernel does.  The loop is co-limited by LDS traffic
// (0.83 against 0.67 ds_read_b128 per MFMA, 18 against 12 KB parked per tap row): 0 measured 3 % faster (0.595 against 0.613 ms for the two
// launches, interleaved), the same bits.
#ifndef U3_SPLANE
#define U3_SPLANE 0
#endif
// Channel stride (floats) of the fp32 copy of the low-resolution region.  The interpolation reads it with ds_read_b32 (banks = dword mod 32,
// 32 lanes per LDS cycle): a half-wave is 8 consecutive quads x the 4 four-channel groups, i.e. addresses quad + 4 g LR_CS -- with 113
// (4 x 113 = 4 mod 32) the groups land 4 banks apart and overlap two-way, with 114 (= 8 mod 32) the 32 lanes hit 32 banks.
// U3_WDMA 1: a tap row's weights arrive by LDS-DMA (buffer_load ... lds: no registers, no ds_write_b128 -- 22 % of the loop's LDS cycles
// were those stores) into the OTHER of two 12 KB weight slots while the row before multiplies: ONE barrier per tap row instead of two.  The
// fp32 copy of the low-resolution region moves with the idle slot.  46 KB of LDS: still three workgroups per CU.
#ifndef U3_WDMA
#define U3_WDMA 1
#endif
#ifndef U3_CARRY
#define U3_CARRY 2
#endif
#ifndef U3_APREF
#define U3_APREF 0
#endif
#ifndef U3_LR_CS
#define U3_LR_CS 114
#endif

namespace {

constexpr int U3_PART = 2 * SP_PIX;                                          // one k-step of the patch: 2 channel groups; hi, then lo' at + U3_PART
constexpr int U3_PUNITS = 2 * U3_PART;                                       // 1360 units = 21 760 B
constexpr int U3_WROW = 3 * 128;                                             // one tap row of one part: 3 taps x [lane half][64 couts]
constexpr int U3_WUNITS = (U3_SPLANE ? 3 : 2) * U3_WROW;                     // hi, lo (12 288 B, 3 units per thread) [, then hi 2^-11: 18 432 B]
constexpr int U3_LDS_BYTES = (U3_PUNITS + (U3_WDMA ? 2 : 1) * U3_WUNITS) * 16;   // (with U3_WDMA: 46 336)                   // 34 048 (40 192 with the third plane): three workgroups per CU (registers); the epilogue's 32 KB scratch fits too

__global__ __launch_bounds__(S_THREADS, 3) void conv3x3_split_ups3_kernel(const SplitConvParams p)
{
    extern __shared__ u32x4 patch[];                                         // the k-step's patch, then the tap row's weights
    u32x4* const wbuf = patch + U3_PUNITS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    int bid;
    {   // an XCD (= an L2) gets a contiguous range of tiles
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    }
    const int tx = bid % p.tilesX; bid /= p.tilesX;
    const int ty = bid % p.tilesY, n = bid / p.tilesY;
    const int oy0 = ty * ST_H, ox0 = tx * ST_W;
    const rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x + (size_t)n * p.xImage), 0,
                                                         (int)((size_t)p.Cin * p.xPlane * 4), 0x00020000);
    const unsigned planeBytes = (unsigned)p.xPlane * 4u;
    unsigned long long st0 = 0, st1 = 0, st2 = 0;
    if (p.stamps) st0 = __builtin_amdgcn_s_memrealtime();

    // weights of tap row (ks, dy): thread t moves unit (tap 3 dy + i, part t / 128, t % 128), i = 0..2
    const rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(p.wq + 1), 0, 9 * p.ksteps * 4096, 0x00020000);
    u32x4* const wdst = wbuf + (tid >> 7) * U3_WROW + (tid & 127);
    u32x4 wreg[3];
    // the THIRD plane of the image (isrConvSplitPrepare): w_hi 2^-11 as fp16, [tap][k-step][128 units] behind the (hi, lo) planes -- the A
    // operand of the x_lo' product, which the other kernels make with four v_pk_mul_f16 per fragment (288 vector instructions per wave
    // and tile here).  384 units per tap row: thread t moves unit t, threads < 128 also unit 256 + t.
    const rsrc_t w3rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(p.wq + 1 + 9 * p.ksteps * 256), 0, 9 * p.ksteps * 2048, 0x00020000);
    u32x4 wreg3[2];
    auto wfetch = [&](int step) {                                            // step = 3 ks + dy
        if (step >= 3 * p.ksteps) return;
        const int ks = step / 3, dy = step - 3 * ks;
#pragma unroll
        for (int i = 0; i < 3; ++i) wreg[i] = __builtin_amdgcn_raw_buffer_load_b128(wrs, tid * 16, ((3 * dy + i) * p.ksteps + ks) * 4096, 0);
        if (U3_SPLANE) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int e = tid + 256 * k;                                 // (tap e / 128, unit e % 128) of the tap row
                wreg3[k] = __builtin_amdgcn_raw_buffer_load_b128(w3rs, (int)(e < 384 ? (unsigned)(e & 127) * 16u : BAD_OFFSET), ((3 * dy + (e >> 7)) * p.ksteps + ks) * 2048, 0);
            }
        }
    };
    auto wpark = [&]() {
#pragma unroll
        for (int i = 0; i < 3; ++i) wdst[i * 128] = wreg[i];
        if (U3_SPLANE) {
            wbuf[2 * U3_WROW + tid] = wreg3[0];
            if (tid < 128) wbuf[2 * U3_WROW + 256 + tid] = wreg3[1];
        }
    };

    // ---- staging of one k-step (16 channels): low-resolution region -> fp32 copy on the (idle) weight buffer -> interpolate, split
    constexpr int LR_H = ST_H / 2 + 2, LR_W = ST_W / 2 + 2;                 // 6 x 18 low-res pixels: rows oy0/2 - 1 .., cols ox0/2 - 1 ..
    constexpr int LQ = (ST_W / 2 + 8) / 4;                                   // 6 aligned quads per row
    constexpr int LUNITS = 16 * LR_H * LQ;                                   // (channel, row, quad) = 576: 2.25 per thread
    constexpr int LR_CS = U3_LR_CS;                                          // channel stride of the fp32 copy
    constexpr int QR = SP_H / 2, QC = SP_W / 2, UQ = QR * QC;               // 5 x 17 quads of 2 x 2 patch pixels
    float* tmp = reinterpret_cast<float*>(wbuf);                             // [16][114] fp32 = 7.3 KB of the (idle) 12.3 KB weight buffer / slot
    // U3_WDMA: tap row `step` = 3 ks + dy into weight slot step & 1: 12 wave-wide pieces (tap i, part, half), three per wave
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    auto wdma = [&](int step) {
        if (step >= 3 * p.ksteps) return;
        const int ks = step / 3, dy = step - 3 * ks;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int idx = wv + 4 * k, i = idx >> 2, part = (idx >> 1) & 1, half = idx & 1;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (isr_lvoid_t*)(wbuf + (step & 1) * U3_WUNITS + part * U3_WROW + i * 128 + half * 64), 16,
                                                     lane * 16, (((3 * dy + i) * p.ksteps + ks) * 256 + part * 128 + half * 64) * 16, 0, 0);
        }
    };
    const int ly0 = oy0 / 2 - 1, lx0 = ox0 / 2 - 1;
    u32x4 v[3];
    auto lfetch = [&](int cin0) {                                            // requests only: the values are parked after a barrier
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int u = tid + k * S_THREADS;
            const int c = u / (LR_H * LQ), rem = u - c * (LR_H * LQ);
            const int r = rem / LQ, q = rem - r * LQ;
            const int iy = ly0 + r, ix = ox0 / 2 - 4 + 4 * q;
            const bool ok = u < LUNITS && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
            v[k] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)(ok ? (unsigned)(cin0 + c) * planeBytes + (unsigned)(iy * p.Win + ix) * 4u
                                                                       : BAD_OFFSET), 0, 0);
        }
    };
    auto lpark = [&]() {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int u = tid + k * S_THREADS;
            if (u >= LUNITS) continue;
            const int c = u / (LR_H * LQ), rem = u - c * (LR_H * LQ);
            const int r = rem / LQ, q = rem - r * LQ;
            const float4 f = __builtin_bit_cast(float4, v[k]);
            float* dst = tmp + c * LR_CS + r * LR_W + 4 * q - 3;              // quad q holds low-res patch columns 4q - 3 .. 4q
            if (q > 0) dst[0] = f.x;
            if (q > 0 && q < LQ - 1) { dst[1] = f.y; dst[2] = f.z; }
            if (q < LQ - 1) dst[3] = f.w;
        }
    };
    // INTERIOR tiles (all but the image's outermost ring of tiles): every patch pixel lies in the image and no source index is clamped,
    // so quad (kr, kc) blends the copy's rows kr, kr + 1 and columns kc, kc + 1 with the weights 3/4, 1/4 (upper / left pixel of the
    // quad) and 1/4, 3/4 -- exactly what isr_src_index evaluates to there, as compile-time constants: no index arithmetic, no clamps,
    // no validity selects (~100 of the ~230 vector instructions a unit costs; vector instructions are paid at full price beside the
    // MFMAs, tools/lab/mfma_valu_overlap.hip).  Same operations on the same values in the same order: bit-identical.
    const bool interior = oy0 >= 2 && oy0 + ST_H + 2 <= p.H && ox0 >= 2 && ox0 + ST_W + 2 <= p.W;
    auto interpolate = [&]() {
        _Float16* const patch16 = reinterpret_cast<_Float16*>(patch);
        for (int u = tid; u < 4 * UQ; u += S_THREADS) {
            const int g4 = u & 3, q = u >> 2;                                // neighbouring lanes: the 4 four-channel groups of one quad
            const int kr = q / QC, kc = q - kr * QC;
            f16x4 h00, h01, h10, h11, l00, l01, l10, l11;
            if (interior) {
                const float* ta = tmp + (g4 * 4) * LR_CS + kr * LR_W + kc;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float a0 = ta[e * LR_CS], a1 = ta[e * LR_CS + 1];
                    const float b0 = ta[e * LR_CS + LR_W], b1 = ta[e * LR_CS + LR_W + 1];
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
                y0 = min(max(y0 - ly0, 0), LR_H - 1); y1 = min(max(y1 - ly0, 0), LR_H - 1);
                x0 = min(max(x0 - lx0, 0), LR_W - 1); x1 = min(max(x1 - lx0, 0), LR_W - 1);
                const float* ta = tmp + (g4 * 4) * LR_CS + y0 * LR_W;
                const float* tb = tmp + (g4 * 4) * LR_CS + y1 * LR_W;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float a0 = ta[e * LR_CS + x0], a1 = ta[e * LR_CS + x1];
                    const float b0 = tb[e * LR_CS + x0], b1 = tb[e * LR_CS + x1];
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
            _Float16* d = patch16 + ((size_t)((g4 >> 1) * SP_PIX + (2 * kr) * SP_W + 2 * kc)) * 8 + (g4 & 1) * 4;
            *reinterpret_cast<f16x4*>(d) = h00;
            *reinterpret_cast<f16x4*>(d + 8) = h01;
            *reinterpret_cast<f16x4*>(d + SP_W * 8) = h10;
            *reinterpret_cast<f16x4*>(d + SP_W * 8 + 8) = h11;
            *reinterpret_cast<f16x4*>(d + U3_PART * 8) = l00;
            *reinterpret_cast<f16x4*>(d + U3_PART * 8 + 8) = l01;
            *reinterpret_cast<f16x4*>(d + (U3_PART + SP_W) * 8) = l10;
            *reinterpret_cast<f16x4*>(d + (U3_PART + SP_W) * 8 + 8) = l11;
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[cb][r][i] = 0.0f;

    lfetch(0);
    if (U3_WDMA) {
        wdma(0);
#pragma unroll 1
        for (int ks = 0; ks < p.ksteps; ++ks) {
            // row 3 ks multiplies on slot (3 ks) & 1 (in flight or landed); the other slot is idle: the fp32 copy lives there
            tmp = reinterpret_cast<float*>(wbuf + ((3 * ks + 1) & 1) * U3_WUNITS);
            lpark();
            __syncthreads();
            if (!(p.dbg & 2)) interpolate();
            f16x8 carry[3], carryo[3];                                       // U3_CARRY: the hi (2: and lo') fragments of the patch row two consecutive tap rows share
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                // this row's weights have landed (every wave's pieces: the wait, then the barrier), the patch is complete (dy = 0) and
                // everybody is done with the slot the next row's weights go to (the row before last read it; dy = 0: the fp32 copy)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                wdma(3 * ks + dy + 1);
                if (dy == U3_LFETCH_ROW && ks + 1 < p.ksteps) lfetch(16 * (ks + 1));
                if (!(p.dbg & 1)) {
                    const u32x4* wl = wbuf + ((3 * ks + dy) & 1) * U3_WUNITS + h * 64 + j;
                    const u32x4* bl = patch + h * SP_PIX + (wave * 2 + dy) * SP_W + j;
                    // U3_APREF (round 6, VERDICT r5's reading of the disassembly: the four weight fragments of a tap are single-buffered and waited for with
                    // lgkmcnt(0) in front of their MFMAs): tap dx + 1's weight fragments requested before tap dx's MFMAs issue, into a second set
                    f16x8 na0h, na0l, na1h, na1l;
                    if (U3_APREF) {
                        na0h = __builtin_bit_cast(f16x8, wl[0]); na0l = __builtin_bit_cast(f16x8, wl[U3_WROW]);
                        na1h = __builtin_bit_cast(f16x8, wl[32]); na1l = __builtin_bit_cast(f16x8, wl[U3_WROW + 32]);
                    }
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        f16x8 a0h, a0l, a1h, a1l;
                        if (U3_APREF) {
                            a0h = na0h; a0l = na0l; a1h = na1h; a1l = na1l;
                            if (dx < 2) {
                                na0h = __builtin_bit_cast(f16x8, wl[(dx + 1) * 128]); na0l = __builtin_bit_cast(f16x8, wl[U3_WROW + (dx + 1) * 128]);
                                na1h = __builtin_bit_cast(f16x8, wl[(dx + 1) * 128 + 32]); na1l = __builtin_bit_cast(f16x8, wl[U3_WROW + (dx + 1) * 128 + 32]);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        } else {
                            a0h = __builtin_bit_cast(f16x8, wl[dx * 128]);
                            a0l = __builtin_bit_cast(f16x8, wl[U3_WROW + dx * 128]);
                            a1h = __builtin_bit_cast(f16x8, wl[dx * 128 + 32]);
                            a1l = __builtin_bit_cast(f16x8, wl[U3_WROW + dx * 128 + 32]);
                        }
                        const f16x8 a0s = a0h * (_Float16)0.00048828125f;   // w_hi 2^-11: partner of the scaled x_lo'
                        const f16x8 a1s = a1h * (_Float16)0.00048828125f;
#pragma unroll
                        for (int r = 0; r < 2; ++r) {
                            // (patch row wave * 2 + dy + r: this tap row's r = 1 is the next tap row's r = 0)
                            const f16x8 bh = (U3_CARRY && r == 0 && dy > 0) ? carry[dx] : __builtin_bit_cast(f16x8, bl[r * SP_W + dx]);
                            const f16x8 bo = (U3_CARRY == 2 && r == 0 && dy > 0) ? carryo[dx] : __builtin_bit_cast(f16x8, bl[U3_PART + r * SP_W + dx]);
                            if (U3_CARRY && r == 1) carry[dx] = bh;
                            if (U3_CARRY == 2 && r == 1) carryo[dx] = bo;
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
            __syncthreads();                                                 // the patch and the slot the next fp32 copy goes to are free
        }
    } else {
    wfetch(0);
#pragma unroll 1
    for (int ks = 0; ks < p.ksteps; ++ks) {
        // the patch and the weight buffer are free (barrier at the end of the previous tap row)
        lpark();
        __syncthreads();
        if (!(p.dbg & 2)) interpolate();                                     // (diagnostics, isrDebugSetSplitAblation: 1 no MFMAs, 2 no interpolation, 8 no epilogue, 16 no epilogue stores)
        __syncthreads();                                                     // patch complete; the fp32 copy (on the weight buffer) is done with
        if (p.stamps && ks == 0) st1 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
        for (int dy = 0; dy < 3; ++dy) {
            wpark();
            __syncthreads();
            wfetch(3 * ks + dy + 1);                                         // the next tap row's weights travel under these MFMAs
            if (dy == U3_LFETCH_ROW && ks + 1 < p.ksteps) lfetch(16 * (ks + 1));   // ... and so does the next k-step's low-resolution region
            if (!(p.dbg & 1)) {
                const u32x4* wl = wbuf + h * 64 + j;
                const u32x4* bl = patch + h * SP_PIX + (wave * 2 + dy) * SP_W + j;
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const f16x8 a0h = __builtin_bit_cast(f16x8, wl[dx * 128]);
                    const f16x8 a0l = __builtin_bit_cast(f16x8, wl[U3_WROW + dx * 128]);
                    const f16x8 a1h = __builtin_bit_cast(f16x8, wl[dx * 128 + 32]);
                    const f16x8 a1l = __builtin_bit_cast(f16x8, wl[U3_WROW + dx * 128 + 32]);
                    // w_hi 2^-11, the partner of the scaled x_lo': read from the third plane (U3_SPLANE) or made here (the same fp16 product)
                    const f16x8 a0s = U3_SPLANE ? __builtin_bit_cast(f16x8, wl[2 * U3_WROW + dx * 128]) : a0h * (_Float16)0.00048828125f;
                    const f16x8 a1s = U3_SPLANE ? __builtin_bit_cast(f16x8, wl[2 * U3_WROW + dx * 128 + 32]) : a1h * (_Float16)0.00048828125f;
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        const f16x8 bh = __builtin_bit_cast(f16x8, bl[r * SP_W + dx]);
                        const f16x8 bo = __builtin_bit_cast(f16x8, bl[U3_PART + r * SP_W + dx]);
                        acc[0][r] = mfma16(a0l, bh, acc[0][r]);
                        acc[0][r] = mfma16(a0s, bo, acc[0][r]);
                        acc[0][r] = mfma16(a0h, bh, acc[0][r]);
                        acc[1][r] = mfma16(a1l, bh, acc[1][r]);
                        acc[1][r] = mfma16(a1s, bo, acc[1][r]);
                        acc[1][r] = mfma16(a1h, bh, acc[1][r]);
                    }
                }
            }
            __syncthreads();                                                 // weight buffer (after the third row: the patch too) free
        }
    }
    }

    if (p.stamps) st2 = __builtin_amdgcn_s_memrealtime();
    if (p.dbg & 8) {
        if (acc[0][0][0] == 123.456f) p.ps[0] = u32x4{1u, 2u, 3u, 4u};       // (keeps the accumulators alive)
    } else if (p.ps) split_epilogue_ps(p, acc, oy0, ox0, 0, true, wave, j, h);
    else split_epilogue<true>(p, acc, patch, n, oy0, ox0, 0, true, lane, wave, j, h);       // (the host sends other shapes to the two-per-CU kernel)
    if (p.stamps && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long st3 = __builtin_amdgcn_s_memrealtime();
        unsigned long long* o = p.stamps + (size_t)blockIdx.x * 4;
        o[0] = st0; o[1] = st1; o[2] = st2; o[3] = st3;
    }
}

} // namespace

// Launch hook for isrConv3x3ForwardSplit: -1 if this form does not take the layer (it is for 64-channel layers, as both of
// EnhanceNet's are).
static int isr_launch_split_ups3(const SplitConvParams& p, unsigned nwg, hipStream_t s, hipEvent_t e0, hipEvent_t e1)
{
    if (p.Cin <= 0 || (p.Cin & 15) || p.coutPad != 64 || p.Cout != 64 || p.cgroups != 1 || p.xps) return -1;
    // the fp32 epilogue is compiled for quads only (split_epilogue<WIDE_ONLY>: the per-element path, 20 KB of compare-and-branch code
    // per value, is not in this kernel)
    if (!p.ps && ((p.W | p.yPlane | p.rPlane) & 3)) return -1;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)conv3x3_split_ups3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, U3_LDS_BYTES); attr = true; }
    if (e0 || e1) hipExtLaunchKernelGGL(conv3x3_split_ups3_kernel, dim3(nwg), dim3(S_THREADS), U3_LDS_BYTES, s, e0, e1, 0, p);
    else hipLaunchKernelGGL(conv3x3_split_ups3_kernel, dim3(nwg), dim3(S_THREADS), U3_LDS_BYTES, s, p);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
