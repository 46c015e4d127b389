// conv3x3(U(x)) for EnhanceNet's two upsampling layers (SuperresolutionNetwork/models/enhancenet.py:113-124) with the STAGING AND THE
// MATRIX WORK ON DIFFERENT WAVES.  Included by sr_conv_split.hip (same translation unit as sr_conv_ups3.h); the arithmetic --
// interpolation, split, products, their order -- is conv3x3_split_kernel<true>'s: bit-identical (tests/test_ups_gpu.py).
//
// Why (profiles/r03_pmc_ups.md).  In the tile kernel and in the three-per-CU kernel every wave does everything in
// turn: fetch the low-resolution region, blend + split + pack it into the patch (~1 200 vector instructions per wave and tile), multiply
// (432 MFMAs), epilogue (~770 vector instructions).  The counters said what that costs: matrix pipe busy 44 % of the launch, vector
// issue 39 %, and the two ADD UP -- a wave that stages is not multiplying, and the workgroups of a CU drift into the same phase.  An
// MFMA occupies a SIMD's vector issue for 8 of its 32 cycles (MI355X_MICROARCH.md, cycle constants), so a SECOND wave's vector
// instructions fit beside a wave that only multiplies.  Here a workgroup is eight waves with fixed roles:
//   * waves 0-3, the CONSUMERS (one per SIMD): fragment reads + MFMAs, nothing else between a tile's first and last product; the
//     epilogue (scale, bias, ReLU, split, pack, store) is theirs because the accumulators are;
//   * waves 4-7, the PRODUCERS (one per SIMD): everything that feeds them, one k-step ahead -- the low-resolution region of k-step
//     g + 2 travels from memory while k-step g + 1 is interpolated, split and packed into the OTHER patch slot and the consumers
//     multiply k-step g; the weights arrive by LDS-DMA, one tap row (3 taps x 16 channels x hi / lo = 12 KB) per barrier interval into
//     the other of two weight slots -- no staging registers, no vector instructions.
// ONE barrier per tap row (the three-per-CU kernel needs two: it has one weight buffer), consumers and producers run the same number.
// Workgroups are persistent over an XCD-contiguous range of tiles, so the k-step pipeline runs across tile boundaries: the next
// tile's first patch is being built while this tile's last k-step multiplies and its epilogue runs.
// LDS: 2 patch slots (2 x 21.8 KB) + 2 weight slots (2 x 12.3 KB) + the fp32 copy of one low-resolution region (7.2 KB) + bias =
// 75.6 KB -> two workgroups per CU; registers <= 128 (four waves per SIMD: two consumers whose epilogues and fragment-read latencies
// cover each other, two producers).
#pragma once
#include "sr_split_common.h"

namespace {

constexpr int U4_THREADS = 512;
constexpr int U4_PART = 2 * SP_PIX;                                          // one k-step of the patch: 2 channel groups; hi, then lo' at + U4_PART
constexpr int U4_PUNITS = 2 * U4_PART;                                       // 1360 units = 21 760 B per slot
constexpr int U4_WROW = 3 * 128;                                             // one tap row of one part: 3 taps x [lane half][64 couts]
constexpr int U4_WUNITS = 2 * U4_WROW;                                       // hi then lo: 768 units = 12 288 B per slot
constexpr int U4_LR_CS = 113;                                                // channel stride of the fp32 copy (as sr_conv_ups3.h)
constexpr int U4_TMP_UNITS = (16 * U4_LR_CS * 4 + 15) / 16;                  // 452 units = 7 232 B
constexpr int U4_LDS_BYTES = (2 * U4_PUNITS + 2 * U4_WUNITS + U4_TMP_UNITS) * 16 + 256;      // 75 584

// all LDS traffic of a wave done (consumers: their fragment reads are consumed by then anyway), then the workgroup barrier
__device__ __forceinline__ void u4_barrier_consumer() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// ... and every LDS-DMA / global load of this wave landed: the data it staged is visible to whoever passes the barrier
__device__ __forceinline__ void u4_barrier_producer() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__global__ __launch_bounds__(U4_THREADS, 4) void conv3x3_split_ups4_kernel(const SplitConvParams p)
{
    extern __shared__ u32x4 lds[];
    u32x4* const patch = lds;                                                // [2 slots][U4_PUNITS]
    u32x4* const wbuf = lds + 2 * U4_PUNITS;                                 // [2 slots][U4_WUNITS]
    float* const tmp = reinterpret_cast<float*>(lds + 2 * U4_PUNITS + 2 * U4_WUNITS);       // [16][113] fp32: the low-resolution region being interpolated
    float* const bias_lds = tmp + U4_TMP_UNITS * 4;                          // [64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;

    // this workgroup's tiles: XCD x gets a contiguous range of the tile list, its workgroups take every njw-th tile of it
    const int ntiles = p.N * p.tilesY * p.tilesX;
    const int xcd = blockIdx.x & 7, jw = blockIdx.x >> 3;
    const int njw = ((int)gridDim.x >> 3) + (xcd < ((int)gridDim.x & 7) ? 1 : 0);
    const int tq = ntiles >> 3, trm = ntiles & 7;
    const int tstart = xcd < trm ? xcd * (tq + 1) : trm * (tq + 1) + (xcd - trm) * tq;
    const int tcount = tq + (xcd < trm ? 1 : 0);
    if (jw >= tcount) return;                                                // (the whole workgroup)
    const int mytiles = (tcount - jw + njw - 1) / njw;
    const int total = mytiles * p.ksteps;                                    // k-steps of this workgroup, numbered g = 0 .. total - 1 across its tiles

    struct Tile { int n, oy0, ox0; };
    auto decode = [&](int ti) {                                              // ti-th tile of this workgroup
        int b = tstart + jw + ti * njw;
        Tile r;
        r.ox0 = (b % p.tilesX) * ST_W; b /= p.tilesX;
        r.oy0 = (b % p.tilesY) * ST_H;
        r.n = b / p.tilesY;
        return r;
    };

    // experiment (isrDebugSetSplitAblation bits 8 .. 15): the workgroups dispatched second to a CU (the upper half of the grid) start
    // n x ~4 us late, so that the two consumers of a SIMD are not in their epilogues at the same time
    if (blockIdx.x >= gridDim.x / 2)
        for (int d = (p.dbg >> 8) & 255; d > 0; --d) __builtin_amdgcn_s_sleep(127);

    if (wave >= 4) {
        // =================================================== PRODUCERS ===================================================
        const int ptid = tid - 256, pw = wave - 4;
        constexpr int LR_H = ST_H / 2 + 2, LR_W = ST_W / 2 + 2;             // 6 x 18 low-res pixels: rows oy0/2 - 1 .., cols ox0/2 - 1 ..
        constexpr int LQ = (ST_W / 2 + 8) / 4;                               // 6 aligned quads per row
        constexpr int LUNITS = 16 * LR_H * LQ;                               // (channel, row, quad) = 576: 2.25 per producer thread
        constexpr int QR = SP_H / 2, QC = SP_W / 2, UQ = QR * QC;           // 5 x 17 quads of 2 x 2 patch pixels
        const unsigned planeBytes = (unsigned)p.xPlane * 4u;
        u32x4 v[3];
        // requests only: the values are parked after a barrier
        auto lfetch = [&](int g) {
            if (g >= total || (p.dbg & 2)) return;
            const int ti = g / p.ksteps, cin0 = 16 * (g - ti * p.ksteps);
            const Tile t = decode(ti);
            const rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x + (size_t)t.n * p.xImage), 0,
                                                                 (int)((size_t)p.Cin * p.xPlane * 4), 0x00020000);
            const int ly0 = t.oy0 / 2 - 1;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int u = ptid + k * 256;
                const int c = u / (LR_H * LQ), rem = u - c * (LR_H * LQ);
                const int r = rem / LQ, q = rem - r * LQ;
                const int iy = ly0 + r, ix = t.ox0 / 2 - 4 + 4 * q;
                const bool ok = u < LUNITS && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
                v[k] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)(ok ? (unsigned)(cin0 + c) * planeBytes + (unsigned)(iy * p.Win + ix) * 4u
                                                                           : BAD_OFFSET), 0, 0);
            }
        };
        auto lpark = [&](int g) {
            if (g >= total || (p.dbg & 2)) return;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int u = ptid + k * 256;
                if (u >= LUNITS) continue;
                const int c = u / (LR_H * LQ), rem = u - c * (LR_H * LQ);
                const int r = rem / LQ, q = rem - r * LQ;
                const float4 f = __builtin_bit_cast(float4, v[k]);
                float* dst = tmp + c * U4_LR_CS + r * LR_W + 4 * q - 3;       // quad q holds low-res patch columns 4q - 3 .. 4q
                if (q > 0) dst[0] = f.x;
                if (q > 0 && q < LQ - 1) { dst[1] = f.y; dst[2] = f.z; }
                if (q < LQ - 1) dst[3] = f.w;
            }
        };
        // half `part` (0: units 0 .. 255, 1: units 256 .. 339) of k-step g's patch: interpolate, split, pack into slot g & 1
        auto interpolate = [&](int g, int part) {
            if (g >= total || (p.dbg & 2)) return;
            const int u = ptid + part * 256;
            if (u >= 4 * UQ) return;
            const Tile t = decode(g / p.ksteps);
            const int oy0 = t.oy0, ox0 = t.ox0, ly0 = oy0 / 2 - 1, lx0 = ox0 / 2 - 1;
            _Float16* const patch16 = reinterpret_cast<_Float16*>(patch + (g & 1) * U4_PUNITS);
            const int g4 = u & 3, q = u >> 2;                                // neighbouring lanes: the 4 four-channel groups of one quad
            const int kr = q / QC, kc = q - kr * QC;
            const int Yu = oy0 - 1 + 2 * kr, Xl = ox0 - 1 + 2 * kc;
            const bool oku = (unsigned)Yu < (unsigned)p.H, okd = (unsigned)(Yu + 1) < (unsigned)p.H;
            const bool okl = (unsigned)Xl < (unsigned)p.W, okr = (unsigned)(Xl + 1) < (unsigned)p.W;
            int y0, y1, x0, x1, t0, t1; float lyu, lyd, lxl, lxr, tt;
            isr_src_index(oku ? Yu : Yu + 1, 0.5f, p.Hin, y0, y1, tt);       // both rows of the pair blend these two source rows
            isr_src_index(okl ? Xl : Xl + 1, 0.5f, p.Win, x0, x1, tt);
            isr_src_index(Yu, 0.5f, p.Hin, t0, t1, lyu);
            isr_src_index(Yu + 1, 0.5f, p.Hin, t0, t1, lyd);
            isr_src_index(Xl, 0.5f, p.Win, t0, t1, lxl);
            isr_src_index(Xl + 1, 0.5f, p.Win, t0, t1, lxr);
            const float hyu = 1.f - lyu, hyd = 1.f - lyd, hxl = 1.f - lxl, hxr = 1.f - lxr;
            // rows / columns wholly outside the image (tile overhang) keep their indices inside the staged region
            y0 = min(max(y0 - ly0, 0), LR_H - 1); y1 = min(max(y1 - ly0, 0), LR_H - 1);
            x0 = min(max(x0 - lx0, 0), LR_W - 1); x1 = min(max(x1 - lx0, 0), LR_W - 1);
            const float* ta = tmp + (g4 * 4) * U4_LR_CS + y0 * LR_W;
            const float* tb = tmp + (g4 * 4) * U4_LR_CS + y1 * LR_W;
            f16x4 h00, h01, h10, h11, l00, l01, l10, l11;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a0 = ta[e * U4_LR_CS + x0], a1 = ta[e * U4_LR_CS + x1];
                const float b0 = tb[e * U4_LR_CS + x0], b1 = tb[e * U4_LR_CS + x1];
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
            // 16-byte unit (8-channel group g4 / 2, pixel) holds 8 halves: this 4-channel group is its half (g4 & 1)
            _Float16* d = patch16 + ((size_t)((g4 >> 1) * SP_PIX + (2 * kr) * SP_W + 2 * kc)) * 8 + (g4 & 1) * 4;
            *reinterpret_cast<f16x4*>(d) = h00;
            *reinterpret_cast<f16x4*>(d + 8) = h01;
            *reinterpret_cast<f16x4*>(d + SP_W * 8) = h10;
            *reinterpret_cast<f16x4*>(d + SP_W * 8 + 8) = h11;
            *reinterpret_cast<f16x4*>(d + U4_PART * 8) = l00;
            *reinterpret_cast<f16x4*>(d + U4_PART * 8 + 8) = l01;
            *reinterpret_cast<f16x4*>(d + (U4_PART + SP_W) * 8) = l10;
            *reinterpret_cast<f16x4*>(d + (U4_PART + SP_W) * 8 + 8) = l11;
        };
        // weights of tap row kr (= 3 g + dy: k-step g's channels, taps 3 dy .. 3 dy + 2) into weight slot kr & 1 by LDS-DMA: 12 wave-wide
        // pieces of 64 units [part][tap][half], three per producer wave.  Image: unit (tap ksteps + ks) 256 + part 128 + c (isrConvSplitPrepare)
        auto wdma = [&](int kr) {
            if (kr >= 3 * total || (p.dbg & 8)) return;
            const int g = kr / 3, dy = kr - 3 * g;
            const int ks = g % p.ksteps;
            u32x4* const slot = wbuf + (kr & 1) * U4_WUNITS;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int piece = pw + 4 * i;
                const int part = piece / 6, rem = piece - part * 6, tap = rem >> 1, half = rem & 1;
                isr_dma16(p.wq + 1 + ((size_t)((3 * dy + tap) * p.ksteps + ks)) * 256 + part * 128 + half * 64 + lane,
                          slot + part * U4_WROW + tap * 128 + half * 64);
            }
        };

        lfetch(0);
        wdma(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lpark(0);
        u4_barrier_producer();                                               // A: the fp32 copy of k-step 0's region is complete
        lfetch(1);
        interpolate(0, 0);
        interpolate(0, 1);
        u4_barrier_producer();                                               // B: patch slot 0 and weight slot 0 are ready
#pragma unroll 1
        for (int g = 0; g < total; ++g) {
            // the consumers multiply k-step g; k-step g + 1 is built, k-step g + 2 requested
            u4_barrier_producer();                                           // tap row 0
            wdma(3 * g + 1);
            lpark(g + 1);                                                    // (its loads were requested one k-step ago; the copy was last read in the previous tap row)
            u4_barrier_producer();                                           // tap row 1
            wdma(3 * g + 2);
            lfetch(g + 2);
            interpolate(g + 1, 0);
            u4_barrier_producer();                                           // tap row 2
            wdma(3 * g + 3);
            interpolate(g + 1, 1);
        }
        return;
    }

    // ======================================================= CONSUMERS =======================================================
    if (tid < 64) bias_lds[tid] = p.bias ? p.bias[tid] : 0.0f;
    const float unscale = reinterpret_cast<const float*>(p.wq)[1];          // 2^-S (header of the prepared weights)
    unsigned mag = 0u;
    f32x16 acc[2][2];
    u4_barrier_consumer();                                                   // A
    u4_barrier_consumer();                                                   // B
#pragma unroll 1
    for (int g = 0; g < total; ++g) {
        const int ti = g / p.ksteps, ks = g - ti * p.ksteps;
        if (ks == 0) {
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[cb][r][i] = 0.0f;
        }
#pragma unroll 1
        for (int dy = 0; dy < 3; ++dy) {
            u4_barrier_consumer();
            if (!(p.dbg & 1)) {
                const u32x4* wl = wbuf + ((3 * g + dy) & 1) * U4_WUNITS + h * 64 + j;
                const u32x4* bl = patch + (g & 1) * U4_PUNITS + h * SP_PIX + (wave * 2 + dy) * SP_W + j;
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const f16x8 a0h = __builtin_bit_cast(f16x8, wl[dx * 128]);
                    const f16x8 a0l = __builtin_bit_cast(f16x8, wl[U4_WROW + dx * 128]);
                    const f16x8 a1h = __builtin_bit_cast(f16x8, wl[dx * 128 + 32]);
                    const f16x8 a1l = __builtin_bit_cast(f16x8, wl[U4_WROW + dx * 128 + 32]);
                    const f16x8 a0s = a0h * (_Float16)0.00048828125f;       // w_hi 2^-11: partner of the scaled x_lo'
                    const f16x8 a1s = a1h * (_Float16)0.00048828125f;
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        const f16x8 bh = __builtin_bit_cast(f16x8, bl[r * SP_W + dx]);
                        const f16x8 bo = __builtin_bit_cast(f16x8, bl[U4_PART + r * SP_W + dx]);
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
        if (ks + 1 < p.ksteps) continue;
        // ---- the tile's epilogue: act(acc 2^-S + bias), D row (cout) = (i & 3) + 8 (i >> 2) + 4 h, column (pixel) = j -------------------
        const Tile t = decode(ti);
        const int ox = t.ox0 + j;
        if (p.dbg & 4) continue;
        if (p.ps) {
            // packed-split output: lane (j, h) holds channels 8 g + 4 h .. + 3 of pixel j for the four groups of a 32-channel block, 8 bytes
            // of the hi and 8 of the lo' unit.  The lane pair (j, 0) / (j, 1) trades halves (v_permlane32_swap: the upper 32 lanes of one
            // register against the lower 32 of another), after which lane (j, 0) holds the whole hi unit and lane (j, 1) the whole lo'
            // unit: ONE 16-byte store per lane and unit pair instead of two 8-byte ones -- the epilogue is store-ISSUE bound
            // (tools/lab/bench_ups4.py: 88 of 536 us with 32 b64 stores per wave and tile)
            const rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(p.ps, 0, (int)((size_t)2 * 8 * p.psPlane * 16), 0x00020000);
            const unsigned lopart = (unsigned)h * (unsigned)(8 * p.psPlane) * 16u;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int oy = t.oy0 + wave * 2 + r;
                const unsigned voff = (oy < p.H && ox < p.W) ? (unsigned)(oy * p.W + ox) * 16u + lopart : BAD_OFFSET;
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int gi = 0; gi < 4; ++gi) {
                        const float4 b4 = *reinterpret_cast<const float4*>(bias_lds + cb * 32 + 8 * gi + 4 * h);
                        const float bq[4] = {b4.x, b4.y, b4.z, b4.w};
                        f16x4 th, tl;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float val = acc[cb][r][4 * gi + e] * unscale + bq[e];
                            if (p.act == ISR_ACT_RELU) val = val > 0.f ? val : 0.f;
                            else if (p.act == ISR_ACT_LEAKY) val = val > 0.f ? val : val * p.slope;
                            _Float16 a, b;
                            split16x(val, a, b);
                            th[e] = a; tl[e] = b;
                            mag = isr_umax(mag, isr_mag(val));
                        }
                        const u32x2 uh = __builtin_bit_cast(u32x2, th), ul = __builtin_bit_cast(u32x2, tl);
                        const u32x2 s0 = __builtin_amdgcn_permlane32_swap(uh.x, ul.x, false, false);
                        const u32x2 s1 = __builtin_amdgcn_permlane32_swap(uh.y, ul.y, false, false);
                        const u32x4 unit = {s0.x, s1.x, s0.y, s1.y};        // h = 0: channels 8 g .. + 7 hi; h = 1: the same channels' lo'
                        __builtin_amdgcn_raw_buffer_store_b128(unit, prs, (int)(voff == BAD_OFFSET ? BAD_OFFSET : voff + (unsigned)((cb * 4 + gi) * p.psPlane * 16)), 0, 0);   // (soffset 0: sr_split_common.h)
                    }
            }
        } else {
            // fp32 planes, straight from the D layout: a wave instruction stores 32 consecutive pixels of two channels (2 x 128 bytes)
            const rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(p.y + (size_t)t.n * p.yImage, 0, (int)((size_t)64 * p.yPlane * 4), 0x00020000);
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int oy = t.oy0 + wave * 2 + r;
                const unsigned voff = (oy < p.H && ox < p.W) ? (unsigned)(oy * p.W + ox) * 4u : BAD_OFFSET;
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int gi = 0; gi < 4; ++gi) {
                        const float4 b4 = *reinterpret_cast<const float4*>(bias_lds + cb * 32 + 8 * gi + 4 * h);
                        const float bq[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float val = acc[cb][r][4 * gi + e] * unscale + bq[e];
                            if (p.act == ISR_ACT_RELU) val = val > 0.f ? val : 0.f;
                            else if (p.act == ISR_ACT_LEAKY) val = val > 0.f ? val : val * p.slope;
                            if (voff != BAD_OFFSET) mag = isr_umax(mag, isr_mag(val));
                            const int co = cb * 32 + 8 * gi + 4 * h + e;
                            // (the channel is lane dependent -- 4 h -- so it goes into the vector offset)
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), yrs,
                                                                  (int)(voff != BAD_OFFSET ? voff + (unsigned)co * (unsigned)p.yPlane * 4u : BAD_OFFSET), 0, 0);
                        }
                    }
            }
        }
    }
    isr_range_note(p.absmax, mag);
}

} // namespace

// Launch hook for isrConv3x3ForwardSplit: -1 if this form does not take the layer (64 -> 64-channel layers without residual, as both
// of EnhanceNet's upsampling layers are; the plane stride times 64 channels must fit the buffer descriptor's 32-bit range).
static bool isr_split_ups4_takes(const SplitConvParams& p)
{
    return p.Cin > 0 && !(p.Cin & 15) && p.coutPad == 64 && p.Cout == 64 && p.cgroups == 1 && !p.xps && !p.residual && p.act != ISR_ACT_GATE && !p.slotmax;
}

static int isr_launch_split_ups4(const SplitConvParams& p, hipStream_t s, hipEvent_t e0, hipEvent_t e1)
{
    if (!isr_split_ups4_takes(p)) return -1;
    static int slots = 0;
    if (!slots) {
        int dev = 0, cus = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        slots = 2 * cus;
        (void)hipFuncSetAttribute((const void*)conv3x3_split_ups4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, U4_LDS_BYTES);
    }
    const long long ntiles = (long long)p.N * p.tilesX * p.tilesY;
    if (ntiles <= 0 || ntiles > 0x3fffffffLL) return -1;
    const long long want = ntiles < slots ? ((ntiles + 7) / 8) * 8 : slots;
    const dim3 grid((unsigned)want), block(U4_THREADS);
    if (e0 || e1) hipExtLaunchKernelGGL(conv3x3_split_ups4_kernel, grid, block, U4_LDS_BYTES, s, e0, e1, 0, p);
    else hipLaunchKernelGGL(conv3x3_split_ups4_kernel, grid, block, U4_LDS_BYTES, s, p);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
