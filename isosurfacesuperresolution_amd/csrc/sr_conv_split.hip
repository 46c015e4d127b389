// SPLIT-OPERAND convolution: the fused 3x3 convolution at fp32-equivalent accuracy on the fp16 matrix pipe.
//
// fp32 MFMA (v_mfma_f32_32x32x2_f32) runs at 1/16 of the fp16 rate, and the exact-fmaf kernels of sr_conv3x3.hip sit at
// the board's power cap at ~77 % of that peak: the inference frame is 4.8 ms of fp32 MFMA.  Here every fp32 operand is
// split into two fp16 numbers,  v = hi + lo,  hi = RN16(v),  lo = RN16(v - hi)   (v - hi is exact in fp32), which
// carries 22 significand bits, and a product is three v_mfma_f32_32x32x16_f16 instructions with fp32 accumulation
//        x*w  ~=  x_hi*w_hi + x_hi*w_lo + x_lo*w_hi                      (the dropped x_lo*w_lo is <= 2^-22 |x*w|):
// 16 input channels per instruction at 32 cycles instead of 2 at 64, three instructions instead of one -> 5.3x fewer
// matrix cycles at a per-product relative error of ~3e-7, accumulated in fp32 with 16x fewer roundings than the fmaf
// chain.  Against an fp64 convolution the result is as close as the exact fp32 kernel's (tests/test_conv_gpu.py),
// so this IS the parity path for inference (ops.SPLIT_F16, default on); the exact kernels remain one switch away.
//
// Range: fp16 normals end at 2^-14 and subnormals have an absolute spacing of 2^-24, so a plain `lo` of an O(1) value
// (|lo| <= 2^-11 |v|) would sit among the subnormals and lose its bits.  Both operands are therefore kept in range:
//   * weights (|w| ~ 0.06 for this network) are pre-scaled per layer by 2^S so that max |w| 2^S is in [2^13, 2^14);
//     the epilogue multiplies the accumulator by 2^-S (exact).  w_lo is then a normal number for every weight within
//     2^-17 of the largest one (smaller ones are represented to an absolute 2^-25 -- 2^-38 of the largest);
//   * activations cannot be scaled without knowing their range, so their low part is stored scaled,
//     x_lo' = RN16((x - x_hi) 2^11)  (|x_lo'| <= |x|: a normal number whenever x_hi is), and the product that uses it
//     multiplies by w_hi 2^-11 instead -- an exponent shift of the A fragment in registers (four v_pk_mul_f16 per
//     fragment, issued in the shadow of the MFMAs).  x is carried to 2^-22 relative for 2^-14 <= |x| < 65520;
//     |x| >= 65520 overflows fp16 and shows up as inf/NaN (loud, not silent).
//
// Structure (from the streaming fp16 kernel of sr_conv_f16.hip): workgroup = 8 x 32 output pixels x 64 output
// channels, 4 waves x (2 rows x 2 channel blocks) = 4 accumulators; input staged in chunks of 32 channels as
// LDS[hi|lo][group of 8 channels][patch pixel][8 x fp16] (a lane's B fragment = one conflict-free ds_read_b128, a tap
// shift = +16 bytes); the weights [hi|lo][tap][lane half][cout][8 x fp16] of one 16-channel k-step pass through LDS,
// fetched from L2 into registers under the previous k-step's MFMAs; 80 KB of LDS -> two workgroups per CU, one's
// staging and barriers under the other's MFMAs.  Per tap and k-step: 8 ds_read_b128 feed 12 MFMAs.
#include "sr_diag.h"
#include <cstdlib>
#include <cstring>

#include "sr_split_common.h"

namespace {

template <bool UPS>
__global__ __launch_bounds__(S_THREADS, 2) void conv3x3_split_kernel(const SplitConvParams p)
{
    extern __shared__ u32x4 patch[];                                         // S_PUNITS patch units, then the weight buffer
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    // workgroups are dealt to the 8 XCDs round robin: give every XCD (= every L2) a contiguous range of tiles, so that
    // the halo lines a tile shares with its neighbours are fetched into one L2 once instead of into several
    int bid;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    }
    const int cg = bid % p.cgroups; bid /= p.cgroups;
    const int tx = bid % p.tilesX; bid /= p.tilesX;
    const int ty = bid % p.tilesY, n = bid / p.tilesY;
    const int oy0 = ty * ST_H, ox0 = tx * ST_W, co0 = cg * 64;

    const rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x + (size_t)n * p.xImage), 0,
                                                         (int)((size_t)p.Cin * p.xPlane * 4), 0x00020000);
    const unsigned planeBytes = (unsigned)p.xPlane * 4u;

    u32x4* wbuf = patch + S_PUNITS;
    f32x16 acc[2][2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[cb][r][i] = 0.0f;

    const bool second = co0 + 32 < p.coutPad;                                // the second 32-channel block exists
    const int couts = min(64, p.coutPad - co0);
    unsigned long long st0 = 0, st1 = 0, st2 = 0;
    if (p.stamps) st0 = __builtin_amdgcn_s_memrealtime();

    // weights of k-step ks: 2304 units [part][tap][lane half][64 couts], 9 per thread, L2 -> registers -> LDS
    // With 64 output channels (every layer of this network but the last) thread t moves unit (tap i, part t / 128, t % 128) of the
    // k-step, i = 0..8: byte 16 t + 4096 ksteps i + 4096 ks of the prepared image -- ONE address register and scalar offsets
    // instead of nine 64-bit pointers (which the register allocator spilled)
    const bool w64 = p.coutPad == 64;
    const rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(p.wq + 1), 0, w64 ? 9 * p.ksteps * 4096 : 0, 0x00020000);
    u32x4* wdst = wbuf + (tid >> 7) * S_WPART + (tid & 127);
    u32x4 wreg[9];
    auto wfetch = [&](int ks) {
        if (ks >= p.ksteps) return;
        if (w64) {
#pragma unroll
            for (int i = 0; i < 9; ++i) wreg[i] = __builtin_amdgcn_raw_buffer_load_b128(wrs, tid * 16, (i * p.ksteps + ks) * 4096, 0);
            return;
        }
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int q = tid + i * S_THREADS;
            const int part = q / S_WPART, rem = q - part * S_WPART;
            const int tap = rem >> 7, hh = (rem >> 6) & 1, c = rem & 63;
            if (c < couts) wreg[i] = p.wq[1 + (size_t)(((tap * p.ksteps + ks) * 2 + part) * 2 + hh) * p.coutPad + co0 + c];
        }
    };
    auto wpark = [&]() {
        if (w64) {
#pragma unroll
            for (int i = 0; i < 9; ++i) wdst[i * 128] = wreg[i];
            return;
        }
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int q = tid + i * S_THREADS;
            if ((q & 63) < couts) wbuf[q] = wreg[i];
        }
    };
    wfetch(0);                                                               // in flight under the first staging

    for (int cin0 = 0; cin0 < p.Cin; cin0 += S_CHUNK) {
        const int ks0 = cin0 >> 4;
        const int nks = min(2, p.ksteps - ks0);
        // ---- stage the 32-channel patch, split into hi and lo halves: unit = (channel group g, patch pixel) --------
        if (UPS) {
            // the convolution reads U(x), the x2 bilinear upsampling (align_corners=False) of x [Cin][H/2][W/2]: the
            // low-res region under the tile's patch (6 x 18 pixels x 32 channels, fp32) is staged into the (still idle)
            // weight buffer and the patch is interpolated from it -- the upsampled tensor never exists in memory
            constexpr int LR_H = ST_H / 2 + 2, LR_W = ST_W / 2 + 2;         // 6 x 18 low-res pixels: rows oy0/2 - 1 .., cols ox0/2 - 1 ..
            constexpr int LQ = (ST_W / 2 + 8) / 4;                           // 6 aligned quads per row: columns ox0/2 - 4 .. ox0/2 + 19
            constexpr int LUNITS = S_CHUNK * LR_H * LQ;                      // (channel, row, quad) = 1152
            constexpr int LR_CS = 113;                                       // channel stride (108 used): ds_read_b32 banks = dword mod 32, a half-wave = 8 four-channel groups x 4 quads -> 4 * 113 = 4 (mod 32) apart
            float* tmp = reinterpret_cast<float*>(wbuf);                     // [32][113] fp32 = 14.5 KB of the 36.9 KB weight buffer
            const int ly0 = oy0 / 2 - 1, lx0 = ox0 / 2 - 1;
            for (int u0 = tid; u0 < LUNITS; u0 += 3 * S_THREADS) {
                u32x4 v[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int u = u0 + k * S_THREADS;
                    const int c = u / (LR_H * LQ), rem = u - c * (LR_H * LQ);
                    const int r = rem / LQ, q = rem - r * LQ;
                    const int iy = ly0 + r, ix = ox0 / 2 - 4 + 4 * q;
                    const bool ok = u < LUNITS && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
                    v[k] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)(ok ? (unsigned)(cin0 + c) * planeBytes + (unsigned)(iy * p.Win + ix) * 4u
                                                                               : BAD_OFFSET), 0, 0);
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int u = u0 + k * S_THREADS;
                    if (u >= LUNITS) continue;
                    const int c = u / (LR_H * LQ), rem = u - c * (LR_H * LQ);
                    const int r = rem / LQ, q = rem - r * LQ;
                    const float4 f = __builtin_bit_cast(float4, v[k]);
                    float* dst = tmp + c * LR_CS + r * LR_W + 4 * q - 3;      // quad q holds low-res patch columns 4q - 3 .. 4q
                    if (q > 0) dst[0] = f.x;
                    if (q > 0 && q < LQ - 1) { dst[1] = f.y; dst[2] = f.z; }
                    if (q < LQ - 1) dst[3] = f.w;
                }
            }
            __syncthreads();
            // unit = (4-channel group, 2x2 quad of patch pixels): patch rows 2kr, 2kr+1 are image rows oy0-1+2kr (odd) and the
            // next (even), which blend the same two low-res rows (likewise the columns), so the four source texels of a
            // channel are read once for four outputs and the horizontal blends are shared by the two rows
            constexpr int QR = SP_H / 2, QC = SP_W / 2, UQ = QR * QC;       // 5 x 17 quads
            _Float16* patch16 = reinterpret_cast<_Float16*>(patch);
            for (int u = tid; u < (S_CHUNK / 4) * UQ; u += S_THREADS) {
                const int g4 = u & 7, q = u >> 3;                            // neighbouring lanes: the 8 groups of one quad
                const int kr = q / QC, kc = q - kr * QC;
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
                f16x4 h00, h01, h10, h11, l00, l01, l10, l11;
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
                // 16-byte unit (8-channel group g4 / 2, pixel) holds 8 halves: this 4-channel group is its half (g4 & 1)
                _Float16* d = patch16 + ((size_t)((g4 >> 1) * SP_PIX + (2 * kr) * SP_W + 2 * kc)) * 8 + (g4 & 1) * 4;
                *reinterpret_cast<f16x4*>(d) = h00;
                *reinterpret_cast<f16x4*>(d + 8) = h01;
                *reinterpret_cast<f16x4*>(d + SP_W * 8) = h10;
                *reinterpret_cast<f16x4*>(d + SP_W * 8 + 8) = h11;
                *reinterpret_cast<f16x4*>(d + S_PART * 8) = l00;
                *reinterpret_cast<f16x4*>(d + S_PART * 8 + 8) = l01;
                *reinterpret_cast<f16x4*>(d + (S_PART + SP_W) * 8) = l10;
                *reinterpret_cast<f16x4*>(d + (S_PART + SP_W) * 8 + 8) = l11;
            }
            __syncthreads();                                                 // tmp is free: the weights may land on it
        } else if (p.dbg & 2) {
        } else if (p.xps) {
            // packed-split input: the chunk's units are already what the slot holds -- 2 parts x 1360 units = 44 wave-wide 1 KB
            // pieces, 11 per wave, by LDS-DMA (no registers, no conversion; they land before the barrier below).  Through a buffer
            // descriptor that starts one image row + one pixel in front of the tensor (halo offsets never negative): a 32-bit lane
            // offset, the chunk's / tile's position in the scalar offset, and lanes outside the image (or beyond the last channel
            // group) get the out-of-range offset, for which the hardware writes zeros into LDS (tools/probes/buffer_lds_oob_probe.hip)
            const int groups = p.Cin >> 3;
            const rsrc_t xprs = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(p.xps) - (p.W + 1), 0,
                                                                  (int)(((size_t)2 * groups * p.xpsPlane + p.W + 1) * 16), 0x00020000);
            const int wv = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
            for (int d = 0; d < 11; ++d) {
                const int piece = wv + 4 * d;                                // 0 .. 43: 22 pieces per part
                const int part = piece / 22, pc = piece - part * 22, off = pc * 64 + lane;
                if (off < S_PART) {
                    const int g = off / SP_PIX, pix = off - g * SP_PIX;
                    const int r = pix / SP_W, c = pix - r * SP_W;
                    const int iy = oy0 + r - 1, ix = ox0 + c - 1, gg = (cin0 >> 3) + g;
                    const bool ok = gg < groups && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                    const unsigned vo = ok ? ((unsigned)g * (unsigned)p.xpsPlane + (unsigned)(r * p.W + c)) * 16u : BAD_OFFSET;
                    const unsigned so = ((unsigned)(part * groups + (cin0 >> 3)) * (unsigned)p.xpsPlane + (unsigned)(oy0 * p.W + ox0)) * 16u;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(xprs, (isr_lvoid_t*)(patch + part * S_PART + pc * 64), 16, (int)vo, (int)so, 0, 0);
                }
            }
        } else if (p.quads) {
            // rows of 4-pixel groups aligned to 16 bytes (W, plane stride and tile origin are multiples of 4): one
            // dwordx4 per channel covers 4 pixels.  Unit = (channel group g, patch row r, quad q); quad q holds
            // columns ox0 - 4 + 4q .. +3, i.e. patch columns 4q - 3 .. 4q.
            constexpr int QPR = (ST_W + 8) / 4;                              // 10 quads per patch row
            constexpr int QUNITS = S_GROUPS * SP_H * QPR;                    // 400
            constexpr int QB = 2;                                            // units (8 dwordx4 loads each) in flight per thread
            for (int u0 = tid; u0 < QUNITS; u0 += QB * S_THREADS) {
                u32x4 v[QB][8];
#pragma unroll
                for (int k = 0; k < QB; ++k) {
                    const int u = u0 + k * S_THREADS;
                    const int g = u / (SP_H * QPR), rem = u - g * (SP_H * QPR);
                    const int r = rem / QPR, q = rem - r * QPR;
                    const int iy = oy0 + r - 1, ix = ox0 - 4 + 4 * q;
                    const bool ok = u < QUNITS && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                    const unsigned base = (unsigned)(cin0 + g * 8) * planeBytes + (unsigned)(iy * p.W + ix) * 4u;
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        v[k][e] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)(ok ? base + (unsigned)e * planeBytes : BAD_OFFSET), 0, 0);
                }
#pragma unroll
                for (int k = 0; k < QB; ++k) {
                    const int u = u0 + k * S_THREADS;
                    if (u >= QUNITS) continue;
                    const int g = u / (SP_H * QPR), rem = u - g * (SP_H * QPR);
                    const int r = rem / QPR, q = rem - r * QPR;
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
                    // quad 0 contributes only its last pixel (patch column 0), quad 9 only its first (column 33)
                    if (q > 0) { dst[0] = __builtin_bit_cast(u32x4, h0); dst[S_PART] = __builtin_bit_cast(u32x4, l0); }
                    if (q > 0 && q < QPR - 1) {
                        dst[1] = __builtin_bit_cast(u32x4, h1); dst[S_PART + 1] = __builtin_bit_cast(u32x4, l1);
                        dst[2] = __builtin_bit_cast(u32x4, h2); dst[S_PART + 2] = __builtin_bit_cast(u32x4, l2);
                    }
                    if (q < QPR - 1) { dst[3] = __builtin_bit_cast(u32x4, h3); dst[S_PART + 3] = __builtin_bit_cast(u32x4, l3); }
                }
            }
        } else {
            for (int u0 = tid; u0 < S_PART; u0 += 3 * S_THREADS) {
                float v[3][8];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int u = u0 + k * S_THREADS;
                    const int g = u / SP_PIX, pix = u - g * SP_PIX;
                    const int r = pix / SP_W, c = pix - r * SP_W;
                    const int iy = oy0 + r - 1, ix = ox0 + c - 1;
                    const bool ok = u < S_PART && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                    const unsigned base = (unsigned)(cin0 + g * 8) * planeBytes + (unsigned)(iy * p.W + ix) * 4u;
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[k][e] = buf_load(xrs, ok ? base + (unsigned)e * planeBytes : BAD_OFFSET);
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int u = u0 + k * S_THREADS;
                    f16x8 qh, ql;
#pragma unroll
                    for (int e = 0; e < 8; ++e) { _Float16 a, b; split16x(v[k][e], a, b); qh[e] = a; ql[e] = b; }
                    if (u < S_PART) { patch[u] = __builtin_bit_cast(u32x4, qh); patch[S_PART + u] = __builtin_bit_cast(u32x4, ql); }
                }
            }
        }
        wpark();                                                             // k-step ks0 (its readers passed the barrier that ended the previous chunk)
        __syncthreads();
        if (p.stamps && cin0 == 0) st1 = __builtin_amdgcn_s_memrealtime();
        // ---- MFMAs: k-steps of 16 channels x 9 taps x (2 channel blocks x 2 rows) x 3 products --------------------
#pragma unroll
        for (int S = 0; S < 2; ++S) {
            if (S < nks) {
                wfetch(ks0 + S + 1);                                         // the next k-step's weights travel under these MFMAs
                if (!(p.dbg & 1)) {
                    const u32x4* wl = wbuf + h * 64 + j;
                    const u32x4* bl = patch + (2 * S + h) * SP_PIX + (wave * 2) * SP_W + j;
#pragma unroll
                    for (int tap = 0; tap < 9; ++tap) {
                        const int dy = tap / 3, dx = tap - dy * 3;
                        const f16x8 a0h = __builtin_bit_cast(f16x8, wl[tap * 128]);
                        const f16x8 a0l = __builtin_bit_cast(f16x8, wl[S_WPART + tap * 128]);
                        const f16x8 a1h = __builtin_bit_cast(f16x8, wl[tap * 128 + (second ? 32 : 0)]);
                        const f16x8 a1l = __builtin_bit_cast(f16x8, wl[S_WPART + tap * 128 + (second ? 32 : 0)]);
                        const f16x8 a0s = a0h * (_Float16)0.00048828125f;   // w_hi 2^-11: partner of the scaled x_lo'
                        const f16x8 a1s = a1h * (_Float16)0.00048828125f;
#pragma unroll
                        for (int r = 0; r < 2; ++r) {
                            const f16x8 bh = __builtin_bit_cast(f16x8, bl[(r + dy) * SP_W + dx]);
                            const f16x8 bo = __builtin_bit_cast(f16x8, bl[S_PART + (r + dy) * SP_W + dx]);
                            // the two small cross terms first, then the leading term
                            acc[0][r] = mfma16(a0l, bh, acc[0][r]);
                            acc[0][r] = mfma16(a0s, bo, acc[0][r]);
                            acc[0][r] = mfma16(a0h, bh, acc[0][r]);
                            if (second) {
                                acc[1][r] = mfma16(a1l, bh, acc[1][r]);
                                acc[1][r] = mfma16(a1s, bo, acc[1][r]);
                                acc[1][r] = mfma16(a1h, bh, acc[1][r]);
                            }
                        }
                        if (UPS) __builtin_amdgcn_sched_barrier(0);
                    }
                }
                __syncthreads();                                             // weight buffer (and, after the last k-step, the patch) free
                if (S + 1 < nks) { wpark(); __syncthreads(); }
            }
        }
    }

    if (p.stamps) st2 = __builtin_amdgcn_s_memrealtime();
    if (p.ps) split_epilogue_ps(p, acc, oy0, ox0, co0, second, wave, j, h);
    else if (UPS) split_epilogue(p, acc, patch, n, oy0, ox0, co0, second, lane, wave, j, h);
    else split_epilogue<false, 0, true>(p, acc, patch, n, oy0, ox0, co0, second, lane, wave, j, h);
    if (p.stamps && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long st3 = __builtin_amdgcn_s_memrealtime();
        unsigned long long* o = p.stamps + (size_t)blockIdx.x * 4;
        o[0] = st0; o[1] = st1; o[2] = st2; o[3] = st3;
    }
}

// ---- the streaming form: persistent workgroups, the next k-step's operands in flight under the MFMAs -----------------
// In conv3x3_split_kernel a workgroup's life is: stage (global loads -> convert -> LDS), barrier, MFMAs, ..., epilogue --
// only 20 % of it issues MFMAs and the staging latency is covered by nothing but the CU's other workgroup.  Here a
// workgroup walks a list of tiles as ONE pipeline of 16-channel k-steps: the patch buffer is two k-step slots; while the
// MFMAs of k-step s read slot s % 2, the activations (one 8-channel x 4-pixel unit per thread) and the weights of k-step
// s + 1 -- of this tile or the first of the next one -- are in flight from HBM / L2 into registers, and are split and
// parked into the other slot after the barrier that ends k-step s.  Only a workgroup's very first k-step waits for memory.
// Plain (non-upsampling) layers whose rows allow aligned dwordx4 staging (p.quads).
constexpr int SQ_QPR = (ST_W + 8) / 4;                                       // 10 quads per patch row
constexpr int SQ_UNITS = 2 * SP_H * SQ_QPR;                                  // 200 (channel group, patch row, quad) units per k-step
constexpr int SQ_SLOT = 2 * SP_PIX;                                          // 16-byte units of one k-step slot of the hi (or lo) patch

__global__ __launch_bounds__(S_THREADS, 2) void conv3x3_split_stream_kernel(const SplitConvParams p)
{
    extern __shared__ u32x4 patch[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    u32x4* wbuf = patch + S_PUNITS;
    // the tile list: every XCD (= every L2) owns a contiguous range, its workgroups take the tiles of the range round
    // robin, so that workgroups running at the same time on one XCD work on neighbouring tiles (shared halo lines)
    const int ntiles = p.N * p.tilesY * p.tilesX * p.cgroups;
    const int xcd = blockIdx.x & 7, jw = blockIdx.x >> 3;
    const int njw = ((int)gridDim.x >> 3) + (xcd < ((int)gridDim.x & 7) ? 1 : 0);
    const int tq = ntiles >> 3, trm = ntiles & 7;
    const int tstart = xcd < trm ? xcd * (tq + 1) : trm * (tq + 1) + (xcd - trm) * tq;
    const int tcount = tq + (xcd < trm ? 1 : 0);
    if (jw >= tcount) return;
    const unsigned planeBytes = (unsigned)p.xPlane * 4u;

    struct Tile { int n, oy0, ox0, co0; };
    auto decode = [&](int t) {
        int b = tstart + t;
        Tile r;
        r.co0 = (b % p.cgroups) * 64; b /= p.cgroups;
        r.ox0 = (b % p.tilesX) * ST_W; b /= p.tilesX;
        r.oy0 = (b % p.tilesY) * ST_H; r.n = b / p.tilesY;
        return r;
    };
    // this thread's staging unit (the same for every tile and k-step): channel group ug of the k-step, patch row ur, quad uq
    const bool staging = tid < SQ_UNITS;
    const int ug = tid / (SP_H * SQ_QPR), urem = tid - ug * (SP_H * SQ_QPR);
    const int ur = urem / SQ_QPR, uq = urem - ur * SQ_QPR;
    u32x4 v[8];
    auto issue_loads = [&](const Tile& t, int ks) {
        const rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x + (size_t)t.n * p.xImage), 0,
                                                             (int)((size_t)p.Cin * p.xPlane * 4), 0x00020000);
        const int iy = t.oy0 + ur - 1, ix = t.ox0 - 4 + 4 * uq;
        const bool ok = staging && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W && !(p.dbg & 2);
        const unsigned base = (unsigned)(ks * 16 + ug * 8) * planeBytes + (unsigned)(iy * p.W + ix) * 4u;
#pragma unroll
        for (int e = 0; e < 8; ++e)
            v[e] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)(ok ? base + (unsigned)e * planeBytes : BAD_OFFSET), 0, 0);
    };
    auto park_loads = [&](int slot) {
        if (!staging) return;
        f16x8 h0, h1, h2, h3, l0, l1, l2, l3;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float4 f = __builtin_bit_cast(float4, v[e]);
            _Float16 a, b;
            split16x(f.x, a, b); h0[e] = a; l0[e] = b;
            split16x(f.y, a, b); h1[e] = a; l1[e] = b;
            split16x(f.z, a, b); h2[e] = a; l2[e] = b;
            split16x(f.w, a, b); h3[e] = a; l3[e] = b;
        }
        u32x4* dst = patch + slot * SQ_SLOT + ug * SP_PIX + ur * SP_W + 4 * uq - 3;
        // quad 0 contributes only its last pixel (patch column 0), quad 9 only its first (column 33)
        if (uq > 0) { dst[0] = __builtin_bit_cast(u32x4, h0); dst[S_PART] = __builtin_bit_cast(u32x4, l0); }
        if (uq > 0 && uq < SQ_QPR - 1) {
            dst[1] = __builtin_bit_cast(u32x4, h1); dst[S_PART + 1] = __builtin_bit_cast(u32x4, l1);
            dst[2] = __builtin_bit_cast(u32x4, h2); dst[S_PART + 2] = __builtin_bit_cast(u32x4, l2);
        }
        if (uq < SQ_QPR - 1) { dst[3] = __builtin_bit_cast(u32x4, h3); dst[S_PART + 3] = __builtin_bit_cast(u32x4, l3); }
    };
    const bool w64 = p.coutPad == 64;                                        // see conv3x3_split_kernel: one address register for the weights
    const rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(p.wq + 1), 0, w64 ? 9 * p.ksteps * 4096 : 0, 0x00020000);
    u32x4* wdst = wbuf + (tid >> 7) * S_WPART + (tid & 127);
    u32x4 wreg[9];
    auto wfetch = [&](int ks, int co0) {
        if (w64) {
#pragma unroll
            for (int i = 0; i < 9; ++i) wreg[i] = __builtin_amdgcn_raw_buffer_load_b128(wrs, tid * 16, (i * p.ksteps + ks) * 4096, 0);
            return;
        }
        const int couts = min(64, p.coutPad - co0);
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int q = tid + i * S_THREADS;
            const int part = q / S_WPART, rem = q - part * S_WPART;
            const int tap = rem >> 7, hh = (rem >> 6) & 1, c = rem & 63;
            if (c < couts) wreg[i] = p.wq[1 + (size_t)(((tap * p.ksteps + ks) * 2 + part) * 2 + hh) * p.coutPad + co0 + c];
        }
    };
    auto wpark = [&](int co0) {
        if (w64) {
#pragma unroll
            for (int i = 0; i < 9; ++i) wdst[i * 128] = wreg[i];
            return;
        }
        const int couts = min(64, p.coutPad - co0);
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int q = tid + i * S_THREADS;
            if ((q & 63) < couts) wbuf[q] = wreg[i];
        }
    };

    Tile cur = decode(jw);
    issue_loads(cur, 0);
    wfetch(0, cur.co0);
    park_loads(0);
    wpark(cur.co0);
    __syncthreads();
    int slot = 0;                                                            // patch slot of the k-step about to be multiplied
    for (int t = jw; t < tcount; t += njw) {
        const bool more = t + njw < tcount;
        Tile nxt = cur;
        if (more) nxt = decode(t + njw);
        const bool second = cur.co0 + 32 < p.coutPad;
        f32x16 acc[2][2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[cb][r][i] = 0.0f;
#pragma unroll 1
        for (int ks = 0; ks + 1 < p.ksteps; ++ks) {
            // operands of the next k-step travel under these MFMAs
            issue_loads(cur, ks + 1);
            wfetch(ks + 1, cur.co0);
            if (!(p.dbg & 1)) split_kstep(acc, wbuf + h * 64 + j, patch + slot * SQ_SLOT + h * SP_PIX + (wave * 2) * SP_W + j, second);
            __syncthreads();                                                 // slot ^ 1 (read by k-step ks - 1) and the weight buffer are free
            park_loads(slot ^ 1);
            wpark(cur.co0);
            __syncthreads();
            slot ^= 1;
        }
        // last k-step of the tile: the first k-step of the next tile travels under it
        if (more) {
            issue_loads(nxt, 0);
            wfetch(0, nxt.co0);
        }
        if (!(p.dbg & 1)) split_kstep(acc, wbuf + h * 64 + j, patch + slot * SQ_SLOT + h * SP_PIX + (wave * 2) * SP_W + j, second);
        // the epilogue transposes through the whole (now idle) patch buffer -- 4 x 8 KB, more than one k-step slot -- so the
        // next tile's first k-step stays in registers until it is done
        __syncthreads();
        split_epilogue(p, acc, patch, cur.n, cur.oy0, cur.ox0, cur.co0, second, lane, wave, j, h);
        __syncthreads();
        if (more) {
            park_loads(slot ^ 1);
            wpark(nxt.co0);
            __syncthreads();
        }
        slot ^= 1;
        cur = nxt;
    }
}

// ---- the wide form: one 512-thread workgroup per CU, hand-pipelined operand reads ------------------------------------
// (isrDebugSetSplitAlgo(2); not the default: measured 0.59 vs 0.545 ms on the 1080p layer.  It exists because it answered a
// question: with the fragment reads pipelined by hand and one barrier per k-step the layer does not get faster, because at
// 1080p the layer is POWER bound -- rocm-smi during the streaming kernel: 1333 W package power, shader clock down from 2.39
// to 2.05 GHz; MFMAs alone: 1010 W at 2.39 GHz.  Dynamic energy per launch 0.60 J, of which the MFMAs and their operand
// reads are 0.26 J: at the board's 1.4 kW limit that energy cannot be spent in less than 0.52 ms, and the kernel takes 0.55.)
// What the two kernels above leave on the table is inside the MFMA loop: hipcc sinks every ds_read to just in front of
// the MFMAs that use it, so each tap exposes the LDS latency two or three times and the loop runs at ~55 % of the matrix
// pipe (MFMAs-only ablation of a 1080p layer: 0.346 ms for 0.182 ms of MFMA issue).  Double-buffering the fragments by
// hand needs registers the 256-thread kernels do not have (weights + activations of the next k-step in flight, 9 + 8
// x 4 registers per thread).  With 512 threads per workgroup -- tile 16 x 32 pixels x 64 channels, wave = one 32-channel
// block x 4 rows -- the per-thread staging halves, and one workgroup per CU has the LDS for TWO weight buffers:
//   * LDS: activations 2 k-step slots x (hi | lo) x 2 groups x 18 x 34 pixels x 16 B = 78 336 B, weights 2 x 36 864 B;
//   * per k-step ONE barrier: operands of k-step s + 1 are loaded at the start of k-step s, split / parked into the other
//     slot and the other weight buffer two thirds through its MFMAs (their last readers passed the previous barrier);
//   * per (tap, row) step: the B fragments of the next step and, once per tap, the A fragments of the next tap are read
//     before the step's three MFMAs (sched_barrier fences keep hipcc from sinking them).
constexpr int WT_H = 16, WP_H = WT_H + 2, WP_PIX = WP_H * SP_W;              // 18 x 34 = 612 patch pixels
constexpr int W_SLOT = 2 * WP_PIX;                                           // units of one k-step slot, one part (2 channel groups)
constexpr int W_PART = 2 * W_SLOT;                                           // hi (or lo) part: two slots
constexpr int W_PUNITS = 2 * W_PART;                                         // 4896 units = 78 336 B
constexpr int W_THREADS = 512;
constexpr int W_UNITS = 2 * WP_H * SQ_QPR;                                   // 360 staging units per k-step
constexpr int W_LDS_BYTES = (W_PUNITS + 2 * S_WUNITS) * 16;                  // 78 336 + 73 728 = 152 064

__global__ __launch_bounds__(W_THREADS, 1) void conv3x3_split_wide_kernel(const SplitConvParams p)
{
    extern __shared__ u32x4 patch[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int cb = wave & 1, rg = wave >> 1;                                 // this wave: channel block cb, rows 4 rg .. 4 rg + 3
    u32x4* wbuf = patch + W_PUNITS;
    const int ntiles = p.N * p.tilesY * p.tilesX * p.cgroups;
    const int xcd = blockIdx.x & 7, jw = blockIdx.x >> 3;
    const int njw = ((int)gridDim.x >> 3) + (xcd < ((int)gridDim.x & 7) ? 1 : 0);
    const int tq = ntiles >> 3, trm = ntiles & 7;
    const int tstart = xcd < trm ? xcd * (tq + 1) : trm * (tq + 1) + (xcd - trm) * tq;
    const int tcount = tq + (xcd < trm ? 1 : 0);
    if (jw >= tcount) return;
    const unsigned planeBytes = (unsigned)p.xPlane * 4u;
    const float unscale = reinterpret_cast<const float*>(p.wq)[1];          // 2^-S (header of the prepared weights)

    struct Tile { int n, oy0, ox0, co0; };
    auto decode = [&](int t) {
        int b = tstart + t;
        Tile r;
        r.co0 = (b % p.cgroups) * 64; b /= p.cgroups;
        r.ox0 = (b % p.tilesX) * ST_W; b /= p.tilesX;
        r.oy0 = (b % p.tilesY) * WT_H; r.n = b / p.tilesY;
        return r;
    };
    const bool staging = tid < W_UNITS;
    const int ug = tid / (WP_H * SQ_QPR), urem = tid - ug * (WP_H * SQ_QPR);
    const int ur = urem / SQ_QPR, uq = urem - ur * SQ_QPR;
    u32x4 v[8];
    auto issue_loads = [&](const Tile& t, int ks) {
        const rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x + (size_t)t.n * p.xImage), 0,
                                                             (int)((size_t)p.Cin * p.xPlane * 4), 0x00020000);
        const int iy = t.oy0 + ur - 1, ix = t.ox0 - 4 + 4 * uq;
        const bool ok = staging && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W && !(p.dbg & 2);
        const unsigned base = (unsigned)(ks * 16 + ug * 8) * planeBytes + (unsigned)(iy * p.W + ix) * 4u;
#pragma unroll
        for (int e = 0; e < 8; ++e)
            v[e] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)(ok ? base + (unsigned)e * planeBytes : BAD_OFFSET), 0, 0);
    };
    auto park_loads = [&](int slot) {
        if (!staging) return;
        f16x8 h0, h1, h2, h3, l0, l1, l2, l3;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float4 f = __builtin_bit_cast(float4, v[e]);
            _Float16 a, b;
            split16x(f.x, a, b); h0[e] = a; l0[e] = b;
            split16x(f.y, a, b); h1[e] = a; l1[e] = b;
            split16x(f.z, a, b); h2[e] = a; l2[e] = b;
            split16x(f.w, a, b); h3[e] = a; l3[e] = b;
        }
        u32x4* dst = patch + slot * W_SLOT + ug * WP_PIX + ur * SP_W + 4 * uq - 3;
        if (uq > 0) { dst[0] = __builtin_bit_cast(u32x4, h0); dst[W_PART] = __builtin_bit_cast(u32x4, l0); }
        if (uq > 0 && uq < SQ_QPR - 1) {
            dst[1] = __builtin_bit_cast(u32x4, h1); dst[W_PART + 1] = __builtin_bit_cast(u32x4, l1);
            dst[2] = __builtin_bit_cast(u32x4, h2); dst[W_PART + 2] = __builtin_bit_cast(u32x4, l2);
        }
        if (uq < SQ_QPR - 1) { dst[3] = __builtin_bit_cast(u32x4, h3); dst[W_PART + 3] = __builtin_bit_cast(u32x4, l3); }
    };
    u32x4 wreg[5];
    auto wfetch = [&](int ks, int co0) {
        const int couts = min(64, p.coutPad - co0);
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int q = tid + i * W_THREADS;
            const int part = q / S_WPART, rem = q - part * S_WPART;
            const int tap = rem >> 7, hh = (rem >> 6) & 1, c = rem & 63;
            if (q < S_WUNITS && c < couts) wreg[i] = p.wq[1 + (size_t)(((tap * p.ksteps + ks) * 2 + part) * 2 + hh) * p.coutPad + co0 + c];
        }
    };
    auto wpark = [&](int buf, int co0) {
        const int couts = min(64, p.coutPad - co0);
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int q = tid + i * W_THREADS;
            if (q < S_WUNITS && (q & 63) < couts) wbuf[buf * S_WUNITS + q] = wreg[i];
        }
    };

    f32x16 acc[4];
    unsigned wmag = 0u;
    // one k-step of this wave: 9 taps x 4 rows x 3 products, fragments double-buffered by hand; `mid` runs after two
    // thirds of the steps (the parking of the next k-step's operands)
    auto kstep = [&](int slot, int buf, bool active, auto&& mid) {
        const u32x4* wl = wbuf + buf * S_WUNITS + h * 64 + cb * 32 + j;
        const u32x4* bl = patch + slot * W_SLOT + h * WP_PIX + (4 * rg) * SP_W + j;
        f16x8 ah[2], al[2], bh[2], bo[2];
        if (active) {
            ah[0] = __builtin_bit_cast(f16x8, wl[0]);
            al[0] = __builtin_bit_cast(f16x8, wl[S_WPART]);
            bh[0] = __builtin_bit_cast(f16x8, bl[0]);
            bo[0] = __builtin_bit_cast(f16x8, bl[W_PART]);
        }
        f16x8 as;
#pragma unroll
        for (int i = 0; i < 36; ++i) {
            const int tap = i >> 2, r = i & 3;
            if (active) {
                if (i + 1 < 36) {                                            // B fragments of the next (tap, row) step
                    const int t1 = (i + 1) >> 2, r1 = (i + 1) & 3;
                    const int off = (r1 + t1 / 3) * SP_W + (t1 % 3);
                    bh[(i + 1) & 1] = __builtin_bit_cast(f16x8, bl[off]);
                    bo[(i + 1) & 1] = __builtin_bit_cast(f16x8, bl[W_PART + off]);
                }
                if (r == 0 && tap + 1 < 9) {                                 // A fragments of the next tap
                    ah[(tap + 1) & 1] = __builtin_bit_cast(f16x8, wl[(tap + 1) * 128]);
                    al[(tap + 1) & 1] = __builtin_bit_cast(f16x8, wl[S_WPART + (tap + 1) * 128]);
                }
                if (r == 0) as = ah[tap & 1] * (_Float16)0.00048828125f;     // w_hi 2^-11: partner of the scaled x_lo'
                __builtin_amdgcn_sched_barrier(0);
                acc[r] = mfma16(al[tap & 1], bh[i & 1], acc[r]);
                acc[r] = mfma16(as, bo[i & 1], acc[r]);
                acc[r] = mfma16(ah[tap & 1], bh[i & 1], acc[r]);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (i == 23) mid();
        }
    };

    Tile cur = decode(jw);
    issue_loads(cur, 0);
    wfetch(0, cur.co0);
    park_loads(0);
    wpark(0, cur.co0);
    __syncthreads();
    int slot = 0;                                                            // patch slot / weight buffer of the k-step about to be multiplied
    for (int t = jw; t < tcount; t += njw) {
        const bool more = t + njw < tcount;
        Tile nxt = cur;
        if (more) nxt = decode(t + njw);
        const bool active = cur.co0 + cb * 32 < p.coutPad && !(p.dbg & 1);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[r][i] = 0.0f;
#pragma unroll 1
        for (int ks = 0; ks + 1 < p.ksteps; ++ks) {
            issue_loads(cur, ks + 1);                                        // operands of the next k-step travel under these MFMAs
            wfetch(ks + 1, cur.co0);
            kstep(slot, slot, active, [&]() { park_loads(slot ^ 1); wpark(slot ^ 1, cur.co0); });
            __syncthreads();
            slot ^= 1;
        }
        if (more) {                                                          // last k-step: the next tile's first one travels under it
            issue_loads(nxt, 0);
            wfetch(0, nxt.co0);
        }
        kstep(slot, slot, active, [&]() { if (more) { park_loads(slot ^ 1); wpark(slot ^ 1, nxt.co0); } });
        __syncthreads();
        // ---- epilogue: D row (cout) = (reg & 3) + 8 * (reg >> 2) + 4 * h, column (pixel) = j; each wave transposes one
        // output row (32 couts x 32 pixels) at a time through 4 KB of the patch slot that was just multiplied
        if (cur.co0 + cb * 32 < p.coutPad && !(p.dbg & 4)) {
            const rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(p.y + (size_t)cur.n * p.yImage, 0, (int)((size_t)p.Cout * p.yPlane * 4), 0x00020000);
            const rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.residual ? p.residual + (size_t)cur.n * p.rImage : p.y), 0,
                                                                 p.residual ? (int)((size_t)p.Cout * p.rPlane * 4) : 0, 0x00020000);
            float bv[16];
#pragma unroll
            for (int i = 0; i < 16; ++i)
                bv[i] = p.bias ? p.bias[min(cur.co0 + cb * 32 + (i & 3) + 8 * (i >> 2) + 4 * h, p.Cout - 1)] : 0.0f;
            float* tr = reinterpret_cast<float*>(patch + (wave >> 2) * W_PART + slot * W_SLOT) + (wave & 3) * (32 * 32);
            const bool wide = ((p.W | p.yPlane | p.rPlane) & 3) == 0;
            isr_with_act(p.act, [&](auto A) {                               // (one switch, not one per value: sr_split_common.h)
            constexpr int ACT = decltype(A)::value;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int oy = cur.oy0 + 4 * rg + r;
                if (wide) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        float val = isr_activate<ACT>(acc[r][i] * unscale + bv[i], p.slope);
                        tr[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + j] = val;
                    }
                    __builtin_amdgcn_s_waitcnt(0xC07F);                      // lgkmcnt(0): same-wave hand-off through LDS
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int q = lane + 64 * k;                         // float4 index: cout = q / 8, pixel group = q % 8
                        const int co = cur.co0 + cb * 32 + (q >> 3), px = cur.ox0 + (q & 7) * 4;
                        const bool ok = oy < p.H && px < p.W && co < p.Cout;
                        float4 val = reinterpret_cast<const float4*>(tr)[q];
                        const unsigned pixoff = (unsigned)(oy * p.W + px) * 4u;
                        if (p.residual) {
                            const u32x4 rr = __builtin_amdgcn_raw_buffer_load_b128(rrs, (int)(ok ? pixoff + (unsigned)co * (unsigned)p.rPlane * 4u : BAD_OFFSET), 0, 0);
                            const float4 rf = __builtin_bit_cast(float4, rr);
                            if (ACT == ISR_ACT_GATE) {
                                val.x = rf.x > 0.f ? val.x : 0.f; val.y = rf.y > 0.f ? val.y : 0.f;
                                val.z = rf.z > 0.f ? val.z : 0.f; val.w = rf.w > 0.f ? val.w : 0.f;
                            } else {
                                val.x += rf.x; val.y += rf.y; val.z += rf.z; val.w += rf.w;
                            }
                        }
                        if (ok) wmag = isr_umax(isr_umax(wmag, isr_umax(isr_mag(val.x), isr_mag(val.y))), isr_umax(isr_mag(val.z), isr_mag(val.w)));
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, val), yrs,
                                                               (int)(ok ? pixoff + (unsigned)co * (unsigned)p.yPlane * 4u : BAD_OFFSET), 0, 0);
                    }
                    __builtin_amdgcn_s_waitcnt(0xC07F);                      // reads done before the next row overwrites the slab
                } else {
                    const int ox = cur.ox0 + j;
                    const unsigned pix = (oy < p.H && ox < p.W) ? (unsigned)(oy * p.W + ox) * 4u : BAD_OFFSET;
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int co = cur.co0 + cb * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                        float val = isr_activate<ACT>(acc[r][i] * unscale + bv[i], p.slope);
                        const bool ok = pix != BAD_OFFSET && co < p.Cout;
                        if (p.residual) {
                            const float rv = buf_load(rrs, ok ? pix + (unsigned)co * (unsigned)p.rPlane * 4u : BAD_OFFSET);
                            if (ACT == ISR_ACT_GATE) val = rv > 0.f ? val : 0.f; else val += rv;
                        }
                        if (ok) wmag = isr_umax(wmag, isr_mag(val));
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), yrs,
                                                              ok ? (int)(pix + (unsigned)co * (unsigned)p.yPlane * 4u) : (int)BAD_OFFSET, 0, 0);
                    }
                }
            }
                    });
        }
        isr_range_note(p.absmax, wmag);
        wmag = 0u;
        __syncthreads();                                                     // the scratch is a patch slot: done before the next k-step parks into it
        slot ^= 1;
        cur = nxt;
    }
}

// ---- the small-image form: 2-row tiles, weights two k-steps at a time ------------------------------------------------
// A batch of 32 x 32 training crops is 64 tiles of 8 x 32 pixels: a quarter of the CUs, each running its 432 MFMAs per
// wave alone -- and on the fp32 kernel 7.7 us of matrix work in a 17.8 us launch, 419 times per training step.  Here a
// tile is 2 rows x 32 pixels x 64 channels (256 workgroups for 16 crops), a wave = one row x one 32-channel block = ONE
// accumulator and 108 MFMAs for 64 input channels; the whole 64-channel patch (4 x 34 pixels) is staged at once and the
// weights pass through LDS two k-steps at a time (73.7 KB), the next pair in flight under the MFMAs of the current one.
// One workgroup per CU (108 KB of LDS); the launch is a chain of latencies, not of arithmetic.
constexpr int R2_H = 2, R2_PH = R2_H + 2, R2_PIX = R2_PH * SP_W;             // 4 x 34 = 136 patch pixels
constexpr int R2_PART = 8 * R2_PIX;                                          // units of the hi (or lo) patch of a 64-channel chunk: 1088
constexpr int R2_PUNITS = 2 * R2_PART;                                       // 2176 units = 34 816 B
constexpr int R2_WUNITS = 2 * S_WUNITS;                                      // two k-steps of weights: 4608 units = 73 728 B
constexpr int R2_UNITS = 8 * R2_PH * SQ_QPR;                                 // 320 staging units (channel group, patch row, quad) per chunk
constexpr int R2_LDS_BYTES = (R2_PUNITS + R2_WUNITS) * 16;                   // 108 544

__global__ __launch_bounds__(S_THREADS, 1) void conv3x3_split_rows2_kernel(const SplitConvParams p)
{
    extern __shared__ u32x4 patch[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int row = wave >> 1, cb = wave & 1;
    u32x4* wbuf = patch + R2_PUNITS;
    int bid;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    }
    const int cg = bid % p.cgroups; bid /= p.cgroups;
    const int tx = bid % p.tilesX; bid /= p.tilesX;
    const int ty = bid % p.tilesY, n = bid / p.tilesY;
    const int oy0 = ty * R2_H, ox0 = tx * ST_W, co0 = cg * 64;
    const int couts = min(64, p.coutPad - co0);
    const bool active = co0 + cb * 32 < p.coutPad;
    const rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x + (size_t)n * p.xImage), 0,
                                                         (int)((size_t)p.Cin * p.xPlane * 4), 0x00020000);
    const unsigned planeBytes = (unsigned)p.xPlane * 4u;

    // staging units of this thread: u = tid and (for the first 64 threads) tid + 256
    u32x4 v[2][8];
    auto issue_loads = [&](int cin0) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int u = tid + k * S_THREADS;
            const int g = u / (R2_PH * SQ_QPR), rem = u - g * (R2_PH * SQ_QPR);
            const int r = rem / SQ_QPR, q = rem - r * SQ_QPR;
            const int iy = oy0 + r - 1, ix = ox0 - 4 + 4 * q;
            const bool ok = u < R2_UNITS && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const unsigned base = (unsigned)(cin0 + g * 8) * planeBytes + (unsigned)(iy * p.W + ix) * 4u;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                v[k][e] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)(ok ? base + (unsigned)e * planeBytes : BAD_OFFSET), 0, 0);
        }
    };
    auto park_loads = [&]() {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int u = tid + k * S_THREADS;
            if (u >= R2_UNITS) continue;
            const int g = u / (R2_PH * SQ_QPR), rem = u - g * (R2_PH * SQ_QPR);
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
            u32x4* dst = patch + g * R2_PIX + r * SP_W + 4 * q - 3;
            if (q > 0) { dst[0] = __builtin_bit_cast(u32x4, h0); dst[R2_PART] = __builtin_bit_cast(u32x4, l0); }
            if (q > 0 && q < SQ_QPR - 1) {
                dst[1] = __builtin_bit_cast(u32x4, h1); dst[R2_PART + 1] = __builtin_bit_cast(u32x4, l1);
                dst[2] = __builtin_bit_cast(u32x4, h2); dst[R2_PART + 2] = __builtin_bit_cast(u32x4, l2);
            }
            if (q < SQ_QPR - 1) { dst[3] = __builtin_bit_cast(u32x4, h3); dst[R2_PART + 3] = __builtin_bit_cast(u32x4, l3); }
        }
    };
    // weights of the k-step pair (2 pr, 2 pr + 1): 4608 units, 18 per thread
    u32x4 wreg[18];
    auto wfetch = [&](int pr) {
#pragma unroll
        for (int i = 0; i < 18; ++i) {
            const int q = tid + i * S_THREADS;
            const int kk = q / S_WUNITS, q1 = q - kk * S_WUNITS;
            const int part = q1 / S_WPART, rem = q1 - part * S_WPART;
            const int tap = rem >> 7, hh = (rem >> 6) & 1, c = rem & 63;
            const int ks = 2 * pr + kk;
            if (c < couts && ks < p.ksteps) wreg[i] = p.wq[1 + (size_t)(((tap * p.ksteps + ks) * 2 + part) * 2 + hh) * p.coutPad + co0 + c];
        }
    };
    auto wpark = [&](int pr) {
#pragma unroll
        for (int i = 0; i < 18; ++i) {
            const int q = tid + i * S_THREADS;
            const int kk = q / S_WUNITS;
            if ((q & 63) < couts && 2 * pr + kk < p.ksteps) wbuf[q] = wreg[i];
        }
    };

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    const int npairs = (p.ksteps + 1) >> 1;
    issue_loads(0);
    wfetch(0);
    park_loads();
    wpark(0);
    __syncthreads();
    for (int pr = 0; pr < npairs; ++pr) {
        const bool more = pr + 1 < npairs;
        const bool restage = more && ((pr + 1) & 1) == 0;                    // the next pair starts a new 64-channel chunk
        if (more) wfetch(pr + 1);                                            // in flight under this pair's MFMAs
        if (restage) issue_loads((pr + 1) * 32);
        if (active) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                if (2 * pr + kk < p.ksteps) {
                    const u32x4* wl = wbuf + kk * S_WUNITS + h * 64 + cb * 32 + j;
                    const u32x4* bl = patch + (2 * (2 * (pr & 1) + kk) + h) * R2_PIX + row * SP_W + j;
#pragma unroll
                    for (int tap = 0; tap < 9; ++tap) {
                        const int dy = tap / 3, dx = tap - dy * 3;
                        const f16x8 ah = __builtin_bit_cast(f16x8, wl[tap * 128]);
                        const f16x8 al = __builtin_bit_cast(f16x8, wl[S_WPART + tap * 128]);
                        const f16x8 as = ah * (_Float16)0.00048828125f;     // w_hi 2^-11: partner of the scaled x_lo'
                        const f16x8 bh = __builtin_bit_cast(f16x8, bl[dy * SP_W + dx]);
                        const f16x8 bo = __builtin_bit_cast(f16x8, bl[R2_PART + dy * SP_W + dx]);
                        acc = mfma16(al, bh, acc);
                        acc = mfma16(as, bo, acc);
                        acc = mfma16(ah, bh, acc);
                    }
                }
            }
        }
        __syncthreads();                                                     // weight buffer (and, before a new chunk, the patch) free
        if (more) {
            if (restage) park_loads();
            wpark(pr + 1);
            __syncthreads();
        }
    }
    // ---- epilogue: one output row x 32 channels per wave, transposed through 4 KB of the idle patch buffer ---------
    if (!active || (p.dbg & 4)) return;
    const float unscale = reinterpret_cast<const float*>(p.wq)[1];
    const rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(p.y + (size_t)n * p.yImage, 0, (int)((size_t)p.Cout * p.yPlane * 4), 0x00020000);
    const rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.residual ? p.residual + (size_t)n * p.rImage : p.y), 0,
                                                         p.residual ? (int)((size_t)p.Cout * p.rPlane * 4) : 0, 0x00020000);
    const int oy = oy0 + row;
    float* tr = reinterpret_cast<float*>(patch) + wave * (32 * 32);
    const bool wide = ((p.W | p.yPlane | p.rPlane) & 3) == 0;
    unsigned rmag = 0u;
    isr_with_act(p.act, [&](auto A) {                                       // (one switch, not one per value: sr_split_common.h)
    constexpr int ACT = decltype(A)::value;
    if (wide) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int c = (i & 3) + 8 * (i >> 2) + 4 * h;
            const float val = isr_activate<ACT>(acc[i] * unscale + (p.bias ? p.bias[min(co0 + cb * 32 + c, p.Cout - 1)] : 0.0f), p.slope);
            tr[c * 32 + j] = val;
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);                                  // lgkmcnt(0): same-wave hand-off through LDS
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int q = lane + 64 * k;                                     // float4 index: cout = q / 8, pixel group = q % 8
            const int co = co0 + cb * 32 + (q >> 3), px = ox0 + (q & 7) * 4;
            const bool ok = oy < p.H && px < p.W && co < p.Cout;
            float4 val = reinterpret_cast<const float4*>(tr)[q];
            const unsigned pixoff = (unsigned)(oy * p.W + px) * 4u;
            if (p.residual) {
                const u32x4 rr = __builtin_amdgcn_raw_buffer_load_b128(rrs, (int)(ok ? pixoff + (unsigned)co * (unsigned)p.rPlane * 4u : BAD_OFFSET), 0, 0);
                const float4 rf = __builtin_bit_cast(float4, rr);
                if (ACT == ISR_ACT_GATE) {
                    val.x = rf.x > 0.f ? val.x : 0.f; val.y = rf.y > 0.f ? val.y : 0.f;
                    val.z = rf.z > 0.f ? val.z : 0.f; val.w = rf.w > 0.f ? val.w : 0.f;
                } else {
                    val.x += rf.x; val.y += rf.y; val.z += rf.z; val.w += rf.w;
                }
            }
            if (ok) rmag = isr_umax(isr_umax(rmag, isr_umax(isr_mag(val.x), isr_mag(val.y))), isr_umax(isr_mag(val.z), isr_mag(val.w)));
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, val), yrs,
                                                   (int)(ok ? pixoff + (unsigned)co * (unsigned)p.yPlane * 4u : BAD_OFFSET), 0, 0);
        }
    } else {
        const int ox = ox0 + j;
        const unsigned pix = (oy < p.H && ox < p.W) ? (unsigned)(oy * p.W + ox) * 4u : BAD_OFFSET;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int co = co0 + cb * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
            float val = isr_activate<ACT>(acc[i] * unscale + (p.bias ? p.bias[min(co, p.Cout - 1)] : 0.0f), p.slope);
            const bool ok = pix != BAD_OFFSET && co < p.Cout;
            if (p.residual) {
                const float rv = buf_load(rrs, ok ? pix + (unsigned)co * (unsigned)p.rPlane * 4u : BAD_OFFSET);
                if (ACT == ISR_ACT_GATE) val = rv > 0.f ? val : 0.f; else val += rv;
            }
            if (ok) rmag = isr_umax(rmag, isr_mag(val));
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), yrs,
                                                  ok ? (int)(pix + (unsigned)co * (unsigned)p.yPlane * 4u) : (int)BAD_OFFSET, 0, 0);
        }
    }
    });
    isr_range_note(p.absmax, rmag);
}

// hi 2^-11, element-wise in fp16 (IEEE, subnormals kept): the value `a0h * (_Float16)0.00048828125f` has in the kernels
__device__ __forceinline__ f16x8 split_scaled_hi(f16x8 qh) { return qh * (_Float16)0.00048828125f; }

// header unit of the prepared weights: { 2^S, 2^-S, S (int), 0 } with max |w| 2^S in [2^13, 2^14)
__global__ __launch_bounds__(1024) void split_scale_kernel(const float* __restrict__ w, int count, u32x4* __restrict__ wq)
{
    // one workgroup, but wide and with independent loads in flight: training re-prepares every layer's weights (and their
    // flipped / transposed twin) after every optimizer step, 48 times per step
    __shared__ float red[1024];
    float m0 = 0.0f, m1 = 0.0f, m2 = 0.0f, m3 = 0.0f;
    int i = threadIdx.x;
    for (; i + 3072 < count; i += 4096) {
        m0 = fmaxf(m0, fabsf(w[i])); m1 = fmaxf(m1, fabsf(w[i + 1024]));
        m2 = fmaxf(m2, fabsf(w[i + 2048])); m3 = fmaxf(m3, fabsf(w[i + 3072]));
    }
    for (; i < count; i += 1024) m0 = fmaxf(m0, fabsf(w[i]));
    red[threadIdx.x] = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        int S = 0;
        const float mx = red[0];
        if (mx > 0.0f && mx < 3.0e38f) {
            S = 13 - ilogbf(mx);
            S = S < -100 ? -100 : (S > 100 ? 100 : S);
        }
        u32x4 hdr;
        hdr.x = __builtin_bit_cast(unsigned, ldexpf(1.0f, S));
        hdr.y = __builtin_bit_cast(unsigned, ldexpf(1.0f, -S));
        hdr.z = (unsigned)S; hdr.w = 0u;
        wq[0] = hdr;
    }
}

// w[Cout][Cin][3][3] fp32 -> wq[1 + ((((tap * ksteps + k-step) * 2 + part) * 2 + lane half h) * coutPad + cout)][8 x fp16];
// element e of (k-step s, half h) is input channel 16 s + 8 h + e (zero beyond Cin / Cout); part 0 = hi, 1 = lo of w 2^S
__global__ void prepare_weights_split_kernel(const float* __restrict__ w, u32x4* __restrict__ wq, int Cout, int Cin, int ksteps, int coutPad)
{
    const float scale = reinterpret_cast<const float*>(wq)[0];
    const int total = 9 * ksteps * 2 * coutPad;                              // (tap, k-step, half, cout)
    for (int u = blockIdx.x * blockDim.x + threadIdx.x; u < total; u += gridDim.x * blockDim.x) {
        const int co = u % coutPad;
        const int hh = (u / coutPad) & 1;
        const int s = (u / (coutPad * 2)) % ksteps;
        const int tap = u / (coutPad * 2 * ksteps);
        f16x8 qh, ql;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ci = 16 * s + 8 * hh + e;
            _Float16 a, b;
            split16((co < Cout && ci < Cin) ? w[((size_t)co * Cin + ci) * 9 + tap] * scale : 0.0f, a, b);
            qh[e] = a; ql[e] = b;
        }
        const size_t base = 1 + (size_t)(((tap * ksteps + s) * 2 + 0) * 2 + hh) * coutPad + co;
        wq[base] = __builtin_bit_cast(u32x4, qh);
        wq[base + (size_t)2 * coutPad] = __builtin_bit_cast(u32x4, ql);
        // third plane, behind the (hi, lo) image: hi 2^-11 in fp16 -- what the kernels otherwise compute per fragment (v_pk_mul_f16)
        wq[1 + (size_t)9 * ksteps * 4 * coutPad + (size_t)((tap * ksteps + s) * 2 + hh) * coutPad + co] = __builtin_bit_cast(u32x4, split_scaled_hi(qh));
    }
}

// ---- all layers of a network prepared in two launches (training re-prepares every weight after every optimizer step) -------
// Per layer: the forward weights and their data-gradient twin w'[ci][co][ky][kx] = w[co][ci][2 - ky][2 - kx] (same scale: the
// same numbers), which isrConvSplitPrepare gets from a flipped / transposed copy made by the caller.
constexpr int PM_MAX = 32;
struct PrepManyParams {
    const float* w[PM_MAX];
    u32x4* fwd[PM_MAX];          // may be NULL
    u32x4* bwd[PM_MAX];          // may be NULL
    int cout[PM_MAX], cin[PM_MAX];
    int n;
};

__global__ __launch_bounds__(1024) void split_scale_many_kernel(const PrepManyParams p)
{
    __shared__ float red[1024];
    const int l = blockIdx.x;
    const float* __restrict__ w = p.w[l];
    const int count = p.cout[l] * p.cin[l] * 9;
    float m0 = 0.0f, m1 = 0.0f, m2 = 0.0f, m3 = 0.0f;
    int i = threadIdx.x;
    for (; i + 3072 < count; i += 4096) {
        m0 = fmaxf(m0, fabsf(w[i])); m1 = fmaxf(m1, fabsf(w[i + 1024]));
        m2 = fmaxf(m2, fabsf(w[i + 2048])); m3 = fmaxf(m3, fabsf(w[i + 3072]));
    }
    for (; i < count; i += 1024) m0 = fmaxf(m0, fabsf(w[i]));
    red[threadIdx.x] = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        int S = 0;
        const float mx = red[0];
        if (mx > 0.0f && mx < 3.0e38f) {
            S = 13 - ilogbf(mx);
            S = S < -100 ? -100 : (S > 100 ? 100 : S);
        }
        u32x4 hdr;
        hdr.x = __builtin_bit_cast(unsigned, ldexpf(1.0f, S));
        hdr.y = __builtin_bit_cast(unsigned, ldexpf(1.0f, -S));
        hdr.z = (unsigned)S; hdr.w = 0u;
        if (p.fwd[l]) p.fwd[l][0] = hdr;
        if (p.bwd[l]) p.bwd[l][0] = hdr;
    }
}

// grid: (units, 2 n): blockIdx.y = 2 layer + orientation (0 forward, 1 data gradient)
__global__ __launch_bounds__(256) void prepare_weights_split_many_kernel(const PrepManyParams p)
{
    const int l = blockIdx.y >> 1, tr = blockIdx.y & 1;
    u32x4* __restrict__ wq = tr ? p.bwd[l] : p.fwd[l];
    if (!wq) return;
    const float* __restrict__ w = p.w[l];
    const int Co = p.cout[l], Ci = p.cin[l];                 // of the stored tensor w[Co][Ci][3][3]
    const int Cout = tr ? Ci : Co, Cin = tr ? Co : Ci;       // of the convolution this image serves
    const int ksteps = (Cin + 15) / 16, coutPad = ((Cout + 31) / 32) * 32;
    const float scale = reinterpret_cast<const float*>(wq)[0];
    const int total = 9 * ksteps * 2 * coutPad;
    for (int u = blockIdx.x * blockDim.x + threadIdx.x; u < total; u += gridDim.x * blockDim.x) {
        const int co = u % coutPad;
        const int hh = (u / coutPad) & 1;
        const int s = (u / (coutPad * 2)) % ksteps;
        const int tap = u / (coutPad * 2 * ksteps);
        f16x8 qh, ql;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ci = 16 * s + 8 * hh + e;
            float v = 0.0f;
            if (co < Cout && ci < Cin)
                v = tr ? w[((size_t)ci * Ci + co) * 9 + (8 - tap)] : w[((size_t)co * Ci + ci) * 9 + tap];
            _Float16 a, b;
            split16(v * scale, a, b);
            qh[e] = a; ql[e] = b;
        }
        const size_t base = 1 + (size_t)(((tap * ksteps + s) * 2 + 0) * 2 + hh) * coutPad + co;
        wq[base] = __builtin_bit_cast(u32x4, qh);
        wq[base + (size_t)2 * coutPad] = __builtin_bit_cast(u32x4, ql);
        wq[1 + (size_t)9 * ksteps * 4 * coutPad + (size_t)((tap * ksteps + s) * 2 + hh) * coutPad + co] = __builtin_bit_cast(u32x4, split_scaled_hi(qh));
    }
}

} // namespace

namespace {

// fp32 [C][H][W] (planes xPlane floats apart) -> packed-split [2][C / 8][psPlane] (hi, lo' units): what a producer's packed epilogue writes
__global__ __launch_bounds__(256) void pack_split_kernel(const float* __restrict__ x, u32x4* __restrict__ ps, int groups, int npix, long long xPlane, int psPlane)
{
    const int pix = blockIdx.x * 256 + threadIdx.x, g = blockIdx.y;
    if (pix >= npix) return;
    f16x8 qh, ql;
#pragma unroll
    for (int e = 0; e < 8; ++e) { _Float16 a, b; split16x(x[(size_t)(8 * g + e) * xPlane + pix], a, b); qh[e] = a; ql[e] = b; }
    ps[(size_t)g * psPlane + pix] = __builtin_bit_cast(u32x4, qh);
    ps[(size_t)(groups + g) * psPlane + pix] = __builtin_bit_cast(u32x4, ql);
}

} // namespace

#include "sr_conv_ups3.h"       // the three-workgroups-per-CU upsampling kernel: same translation unit, same parameter block
#ifdef ISR_DIAG                 // forms of the upsampling layer that were built, parity-tested and measured slower (profiles/r0N_ups*.md): diagnostics build only
#include "sr_conv_ups4r.h"      // the four-rows-per-wave upsampling kernel (16 x 32 tiles, two workgroups per CU, prefetched fragments)
#include "sr_conv_upsw.h"       // the one-stream upsampling kernel (persistent workgroup per CU, staging sliced into the MFMA gaps)
#include "sr_conv_ups4.h"       // the role-split upsampling kernel (producer / consumer waves)
#include "sr_conv_ups5.h"       // the software-pipelined upsampling kernel (staging of k-step g + 1 between the MFMAs of k-step g)
#include "sr_conv_upsp.h"       // the phase-decomposed upsampling kernel (no interpolation at run time; packed-split in and out)
#endif
#include "sr_conv_block2.h"     // two chained convolutions of a batch of small images in one launch (training trunk)

__device__ u32x4 g_split_zero_unit[4];      // zero initialised: source of the zero-padding units of the LDS-DMA staging
static unsigned* g_range_flag = nullptr;
static unsigned* g_max_slots = nullptr;   // isrSetMaxSlots: per-wave maxima of the NEXT isrConv3x3ForwardSplit launch
static int g_max_slot_cap = 0, g_max_slot_used = 0;
unsigned* isr_take_range_flag() { unsigned* f = g_range_flag; g_range_flag = nullptr; return f; }
static bool g_ps_in = false;       // set around the launch by isrConv3x3ForwardSplitFromPacked
static bool g_ps_out = false;      // set around the launch by isrConv3x3ForwardSplitPacked (the library is single threaded by contract)
[[maybe_unused]] static unsigned long long* g_split_stamps = nullptr;
[[maybe_unused]] static int g_split_dbg = 0;
static int g_split_small = 1;     // 2-row-tile kernel for small images (isrDebugSetSplitSmall)
static int g_split_slots = 0;     // tests: cap on the persistent kernels' grid (0 = two / one workgroup per CU)
// upsampling layers: 3 = sr_conv_ups3.h (three workgroups per CU; default), 0 = the tile kernel (two); ISR_UPS_FORM overrides
static int g_split_ups_form = isr_diag_env_int("ISR_UPS_FORM", 3);
static int g_split_algo = isr_diag_env_int("ISR_SPLIT_ALGO", 1);      // (ISR_SPLIT_ALGO overrides) plain layers: 1 persistent streaming kernel (default), 2 wide 512-thread kernel, 0 one workgroup per tile

extern "C" {

void isrSetRangeFlag(unsigned* flag) { g_range_flag = flag; }
void isrSetMaxSlots(void* words, int capacity) { g_max_slots = (unsigned*)words; g_max_slot_cap = capacity; g_max_slot_used = 0; }
int isrTakeMaxSlotWords(void) { const int n = g_max_slot_used; g_max_slot_used = 0; return n; }
// bit mask of the process-global diagnostic switches of this translation unit that are NOT in their default position
// (bench.py refuses to report a number measured with any of them set): 1 ablation, 2 kernel form, 4 grid cap, 8 small-image
// form off, 16 stamp buffer
#ifdef ISR_DIAG
int isrDebugSplitState(void)
{
    return (g_split_dbg ? 1 : 0) | (g_split_algo != 1 ? 2 : 0) | (g_split_slots ? 4 : 0) | (g_split_small != 1 ? 8 : 0) | (g_split_stamps ? 16 : 0) | (g_split_ups_form != 3 ? 2 : 0);
}
void isrDebugSetSplitStampBuffer(unsigned long long* buf) { g_split_stamps = buf; }   // not part of the public header
void isrDebugSetSplitAblation(int bits) { g_split_dbg = bits; }
void isrDebugSetSplitAlgo(int a) { g_split_algo = a; }
void isrDebugSetSplitUpsForm(int f) { g_split_ups_form = f; }       // not part of the public header
int isrDebugSplitUpsForm(void) { return g_split_ups_form; }
void isrDebugSetSplitSlots(int n) { g_split_slots = n; }
void isrDebugSetSplitSmall(int on) { g_split_small = on; }
#endif

long long isrConvSplitWeightBytes(int Cin, int Cout)
{
    if (Cin <= 0 || Cout <= 0) return -1;
    // header + [tap][k-step][hi | lo][lane half][coutPad] units + the third plane [tap][k-step][lane half][coutPad] (hi 2^-11)
    return 16 + (long long)9 * ((Cin + 15) / 16) * (2 * 2 + 2) * (((Cout + 31) / 32) * 32) * 16;
}

int isrConvSplitPrepare(const float* w, void* wq, int Cout, int Cin, void* stream)
{
    if (!w || !wq || Cout <= 0 || Cin <= 0) return -1;
    const int ksteps = (Cin + 15) / 16, coutPad = ((Cout + 31) / 32) * 32;
    const int total = 9 * ksteps * 2 * coutPad;
    hipLaunchKernelGGL(split_scale_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, w, Cout * Cin * 9, (u32x4*)wq);
    hipLaunchKernelGGL(prepare_weights_split_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       w, (u32x4*)wq, Cout, Cin, ksteps, coutPad);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int isrConv3x3ForwardSplit(const float* x, const void* wq, const float* bias, const float* residual, float* y,
                           int N, int Cin, int H, int W, int Cout, int act, float slope, int upsample2x,
                           long long xPlane, long long xImage, long long yPlane, long long yImage,
                           long long rPlane, long long rImage, void* stream)
{
    unsigned* const rangeFlag = isr_take_range_flag();       // taken FIRST: an early error return must not leave it armed for the next launch
    if (!x || !wq || !y || N <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return -1;
    if (act < ISR_ACT_NONE || act > ISR_ACT_GATE) return -1;
    if (act == ISR_ACT_GATE && !residual) return -1;
    const int Hin = upsample2x ? H / 2 : H, Win = upsample2x ? W / 2 : W;
    if (upsample2x && ((H & 1) || (W & 1))) return -1;
    if (xPlane < (long long)Hin * Win || yPlane < (long long)H * W || (residual && rPlane < (long long)H * W)) return -1;
    if (xPlane * Cin * 4 > 0x7fffffffLL || yPlane * Cout * 4 > 0x7fffffffLL || (residual && rPlane * Cout * 4 > 0x7fffffffLL)) return -1;
    const bool aligned = (xPlane & 3) == 0 && (xImage & 3) == 0 && ((uintptr_t)x & 15) == 0;
    // the upsampling variant stages aligned groups of four low-res pixels: isrConvF16SupportsUpsample() tells callers
    if (upsample2x && !((Win & 3) == 0 && aligned)) return -3;
    SplitConvParams p;
    p.x = x; p.wq = (const u32x4*)wq; p.bias = bias; p.residual = residual; p.y = y;
    p.N = N; p.Cin = Cin; p.H = H; p.W = W; p.Cout = Cout;
    p.Hin = Hin; p.Win = Win;
    p.xPlane = (int)xPlane; p.yPlane = (int)yPlane; p.rPlane = (int)(residual ? rPlane : yPlane);
    p.xImage = xImage; p.yImage = yImage; p.rImage = rImage;
    p.ksteps = (Cin + 15) / 16; p.coutPad = ((Cout + 31) / 32) * 32;
    p.cgroups = (Cout + 63) / 64;
    p.tilesX = (W + ST_W - 1) / ST_W; p.tilesY = (H + ST_H - 1) / ST_H;
    p.act = act; p.slope = slope;
    ISR_DIAG_SET(p.stamps, g_split_stamps);
    ISR_DIAG_SET(p.dbg, g_split_dbg);
    p.quads = ((W & 3) == 0 && aligned) ? 1 : 0;
    p.ps = nullptr; p.psPlane = 0;
    p.xps = nullptr; p.xpsPlane = 0; p.zero = nullptr;
    p.absmax = rangeFlag;
    p.slotmax = nullptr;
    unsigned* const maxSlots = g_max_slots;
    const int maxCap = g_max_slot_cap;
    g_max_slots = nullptr; g_max_slot_cap = 0; g_max_slot_used = 0;
    if (g_ps_in) {             // isrConv3x3ForwardSplitFromPacked: `x` is a packed-split tensor, xPlane its plane stride in units
        static u32x4* zero = nullptr;
        if (!zero && hipGetSymbolAddress((void**)&zero, HIP_SYMBOL(g_split_zero_unit)) != hipSuccess) return -2;
        if (N != 1 || upsample2x || (Cin & 7) || xPlane * 16 * 2 * (Cin / 8) > 0x7fffffffLL || ((uintptr_t)x & 15)) return -1;
        p.xps = (const u32x4*)x; p.xpsPlane = (int)xPlane; p.zero = zero; p.x = nullptr;
    }
    if (g_ps_out) {            // isrConv3x3ForwardSplitPacked: `y` is the packed-split tensor, yPlane its plane stride in units
        if (N != 1 || residual || (Cout & 7) || act == ISR_ACT_GATE || yPlane * 16 * 2 * (Cout / 8) > 0x7fffffffLL) return -1;
        p.ps = (u32x4*)y; p.psPlane = (int)yPlane; p.y = nullptr;
    }
    const long long nwg = (long long)N * p.tilesX * p.tilesY * p.cgroups;
    if (nwg > 0x7fffffffLL) return -1;
    static bool attr_done = false;
    if (!attr_done) {   // > 64 KiB of LDS needs an explicit opt-in
        (void)hipFuncSetAttribute((const void*)conv3x3_split_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, S_LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)conv3x3_split_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, S_LDS_BYTES);
        attr_done = true;
    }
    const dim3 grid((unsigned)nwg), block(S_THREADS);
    hipStream_t s = (hipStream_t)stream;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    // small images (a batch of training crops): 2-row tiles when the 8-row tiling would leave most CUs idle
    const long long tiles2 = (long long)N * p.tilesX * ((H + R2_H - 1) / R2_H) * p.cgroups;
    if (!upsample2x && p.quads && !g_split_stamps && g_split_small && nwg < 256 && tiles2 >= 64 && tiles2 <= 0x7fffffffLL) {
        static bool attr2 = false;
        if (!attr2) {
            (void)hipFuncSetAttribute((const void*)conv3x3_split_rows2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, R2_LDS_BYTES);
            attr2 = true;
        }
        p.tilesY = (H + R2_H - 1) / R2_H;
        isr_profile_record(ISR_VARIANT_SPLIT_ROWS2, 2.0 * 9 * Cin * Cout * (double)N * H * W, &e0, &e1);
        const dim3 g2((unsigned)tiles2);
        if (e0 || e1) hipExtLaunchKernelGGL(conv3x3_split_rows2_kernel, g2, block, R2_LDS_BYTES, s, e0, e1, 0, p);
        else hipLaunchKernelGGL(conv3x3_split_rows2_kernel, g2, block, R2_LDS_BYTES, s, p);
        return hipGetLastError() == hipSuccess ? 0 : -2;
    }
    if (!upsample2x && p.quads && !g_split_stamps && g_split_algo == 2) {
        // wide form: one 512-thread workgroup per CU, 16 x 32 tiles, hand-pipelined fragment reads
        static int cus = 0;
        if (!cus) {
            int dev = 0;
            (void)hipGetDevice(&dev);
            if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
            (void)hipFuncSetAttribute((const void*)conv3x3_split_wide_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, W_LDS_BYTES);
        }
        p.tilesY = (H + WT_H - 1) / WT_H;
        const long long ntiles = (long long)N * p.tilesX * p.tilesY * p.cgroups;
        const int cap = g_split_slots > 0 ? g_split_slots : cus;
        const long long want = ntiles < cap ? ((ntiles + 7) / 8) * 8 : cap;
        isr_profile_record(ISR_VARIANT_SPLIT_WIDE, 2.0 * 9 * Cin * Cout * (double)N * H * W, &e0, &e1);
        const dim3 pgrid((unsigned)want), wblock(W_THREADS);
        if (e0 || e1) hipExtLaunchKernelGGL(conv3x3_split_wide_kernel, pgrid, wblock, W_LDS_BYTES, s, e0, e1, 0, p);
        else hipLaunchKernelGGL(conv3x3_split_wide_kernel, pgrid, wblock, W_LDS_BYTES, s, p);
        return hipGetLastError() == hipSuccess ? 0 : -2;
    }
    static int slots = 0;
    if (!slots) {
        int dev = 0, cus = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        slots = 2 * cus;
        (void)hipFuncSetAttribute((const void*)conv3x3_split_stream_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, S_LDS_BYTES);
    }
    // persistent form: two workgroups per CU walk the tile list with the next k-step's loads in flight under the MFMAs.  It pays
    // when a workgroup has SEVERAL tiles to walk; a launch that fits in one round (the 480 x 270 trunk: 510 tiles on 512 slots)
    // has nothing to stream into and runs faster on the one-workgroup-per-tile form with its 32-channel staging passes (half the
    // barriers): 37.9 vs 42.5 us per layer in a chain of twenty (tools/lab/bench_trunk_algos.py).  g_split_algo = 3 forces the
    // persistent form for every size.
    // (A few rounds do not pay either: the 1024 tiles of a 16 x 64 x 128 x 128 training layer take 66-95 us on the persistent form and
    // the step is 0.35 ms shorter with them on the one-workgroup-per-tile form; the 2040 tiles of a 960 x 540 layer of the tiled 4K
    // mode 124 against 129 us.  Beyond four rounds the persistent form stays.)
    const bool one_round = nwg <= (g_split_slots > 0 ? g_split_slots : 4 * slots);
    if (!upsample2x && p.quads && !g_split_stamps && (g_split_algo == 3 || (g_split_algo == 1 && !one_round))) {
        const int cap = g_split_slots > 0 ? g_split_slots : slots;
        const long long want = nwg < cap ? ((nwg + 7) / 8) * 8 : cap;
        isr_profile_record(ISR_VARIANT_SPLIT_STREAM, 2.0 * 9 * Cin * Cout * (double)N * H * W, &e0, &e1);
        const dim3 pgrid((unsigned)want);
        if (maxSlots && 4 * want <= maxCap) { p.slotmax = maxSlots; g_max_slot_used = (int)(4 * want); }
        if (e0 || e1) hipExtLaunchKernelGGL(conv3x3_split_stream_kernel, pgrid, block, S_LDS_BYTES, s, e0, e1, 0, p);
        else hipLaunchKernelGGL(conv3x3_split_stream_kernel, pgrid, block, S_LDS_BYTES, s, p);
        return hipGetLastError() == hipSuccess ? 0 : -2;
    }
    // algorithmic flops of the convolution (2 * 9 * Cin * Cout per output pixel), not the 3x matrix flops spent on it
#ifdef ISR_DIAG
    if (upsample2x && g_split_ups_form == 4 && isr_split_ups4_takes(p)) {
        isr_profile_record(ISR_VARIANT_SPLIT_UPS4, 2.0 * 9 * Cin * Cout * (double)N * H * W, &e0, &e1);
        return isr_launch_split_ups4(p, s, e0, e1);
    }
    if (upsample2x && g_split_ups_form == 5 && isr_split_ups5_takes(p)) {
        // (recorded under the three-per-CU kernel's variant: the same layer, the same column of bench.py's kernel table)
        isr_profile_record(ISR_VARIANT_SPLIT_UPS3, 2.0 * 9 * Cin * Cout * (double)N * H * W, &e0, &e1);
        return isr_launch_split_ups5(p, (unsigned)nwg, s, e0, e1);
    }
#endif
    const bool ups3 = upsample2x && (g_split_ups_form == 3 || g_split_ups_form == 4 || g_split_ups_form == 7 || g_split_ups_form == 8) && Cin > 0 && !(Cin & 15) && p.coutPad == 64 && Cout == 64 && p.cgroups == 1 && !p.xps;
    isr_profile_record(ups3 ? ISR_VARIANT_SPLIT_UPS3 : upsample2x ? ISR_VARIANT_SPLIT_UPS : ISR_VARIANT_SPLIT, 2.0 * 9 * Cin * Cout * (double)N * H * W, &e0, &e1);
#ifdef ISR_DIAG
    if (ups3 && g_split_ups_form == 8) {
        const int rc = isr_launch_split_upsw(p, s, e0, e1);
        if (rc != -1) return rc;
    }
    if (ups3 && g_split_ups_form == 7) {
        const int rc = isr_launch_split_ups4r(p, s, e0, e1);
        if (rc != -1) return rc;
    }
#endif
    if (ups3) {
        const int rc = isr_launch_split_ups3(p, (unsigned)nwg, s, e0, e1);
        if (rc != -1) return rc;
    }
    if (upsample2x) {
        if (e0 || e1) hipExtLaunchKernelGGL((conv3x3_split_kernel<true>), grid, block, S_LDS_BYTES, s, e0, e1, 0, p);
        else hipLaunchKernelGGL((conv3x3_split_kernel<true>), grid, block, S_LDS_BYTES, s, p);
    } else {
        if (maxSlots && !p.ps && 4 * nwg <= maxCap) { p.slotmax = maxSlots; g_max_slot_used = (int)(4 * nwg); }
        if (e0 || e1) hipExtLaunchKernelGGL((conv3x3_split_kernel<false>), grid, block, S_LDS_BYTES, s, e0, e1, 0, p);
        else hipLaunchKernelGGL((conv3x3_split_kernel<false>), grid, block, S_LDS_BYTES, s, p);
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int isrConv3x3ForwardSplitPacked(const float* x, const void* wq, const float* bias, void* ps, int Cin, int H, int W, int Cout,
                                 int act, float slope, int upsample2x, long long xPlane, long long psPlane, void* stream)
{
    // only the forms whose epilogue knows the packed-split layout: the one-workgroup-per-tile kernels (every upsampling layer;
    // plain layers are forced onto it for this launch)
    const int algo = g_split_algo, small = g_split_small;
    g_ps_out = true; g_split_algo = 0; g_split_small = 0;
    const int rc = isrConv3x3ForwardSplit(x, wq, bias, nullptr, (float*)ps, 1, Cin, H, W, Cout, act, slope, upsample2x,
                                          xPlane, xPlane * Cin, psPlane, 0, 0, 0, stream);
    g_ps_out = false; g_split_algo = algo; g_split_small = small;
    return rc;
}

int isrConv3x3ForwardSplitFromPacked(const void* xps, const void* wq, const float* bias, const float* residual, void* y, int packed_out,
                                     int Cin, int H, int W, int Cout, int act, float slope, long long xpsPlane, long long yPlane, long long rPlane,
                                     void* stream)
{
    // the one-workgroup-per-tile kernel knows the packed-split layouts; `packed_out`: y is a packed-split tensor as well
    const int algo = g_split_algo, small = g_split_small;
    g_ps_in = true; g_ps_out = packed_out != 0; g_split_algo = 0; g_split_small = 0;
    const int rc = isrConv3x3ForwardSplit((const float*)xps, wq, bias, residual, (float*)y, 1, Cin, H, W, Cout, act, slope, 0,
                                          xpsPlane, xpsPlane * Cin, yPlane, yPlane * Cout, rPlane, rPlane * Cout, stream);
    g_ps_in = false; g_ps_out = false; g_split_algo = algo; g_split_small = small;
    return rc;
}

// ---- phase-decomposed upsampling convolution (sr_conv_upsp.h): an experiment of round 5 (parity-green, slower than the default: profiles/r05_upsp_ablation.md).
// The entry points stay in the ABI; in the product build the layer is "not supported" and callers take the default kernels.
#ifndef ISR_DIAG
long long isrConvUpsPhaseWeightBytes(void) { return isrConvSplitWeightBytes(64, 256); }
long long isrConvUpsPhaseScratchBytes(void) { return 4LL * 64 * 64 * 9 * (long long)sizeof(float); }
int isrConvUpsPhasePrepare(const float*, void*, void*, void*) { return -3; }
int isrConvUpsPhaseSupported(int, int, int, int, long long, long long) { return 0; }
int isrConvUpsPhase(const void*, const void*, const float*, const float*, void*, int, int, int, float, long long, long long, void*) { (void)isr_take_range_flag(); return -3; }
#else
long long isrConvUpsPhaseWeightBytes(void) { return isrConvSplitWeightBytes(64, 256); }
long long isrConvUpsPhaseScratchBytes(void) { return 4LL * 64 * 64 * 9 * (long long)sizeof(float); }

int isrConvUpsPhasePrepare(const float* w, void* wq, void* scratch, void* stream)
{
    if (!w || !wq || !scratch) return -1;
    hipLaunchKernelGGL(ups_phase_weights_kernel, dim3((4 * 64 * 64 + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, (float*)scratch, 64, 64);
    return isrConvSplitPrepare((const float*)scratch, wq, 256, 64, stream);
}

int isrConvUpsPhaseSupported(int Cin, int Cout, int h, int w, long long xpsPlane, long long psPlane)
{
    if (Cin != 64 || Cout != 64 || h < 1 || w < 1) return 0;
    if (xpsPlane < (long long)h * w || psPlane < 4LL * h * w) return 0;
    if (xpsPlane * 16 * 16 > 0x7fffffffLL || psPlane * 16 * 16 > 0x7fffffffLL) return 0;          // 32-bit buffer offsets
    return 1;
}

int isrConvUpsPhase(const void* xps, const void* wq, const float* w, const float* bias, void* ps, int h, int wd, int act, float slope,
                    long long xpsPlane, long long psPlane, void* stream)
{
    unsigned* const rangeFlag = isr_take_range_flag();
    if (!xps || !wq || !w || !ps || (act != ISR_ACT_NONE && act != ISR_ACT_RELU && act != ISR_ACT_LEAKY)) return -1;
    if (!isrConvUpsPhaseSupported(64, 64, h, wd, xpsPlane, psPlane) || ((uintptr_t)xps & 15) || ((uintptr_t)ps & 15)) return -3;
    SplitConvParams p;
    std::memset(&p, 0, sizeof(p));
    p.wq = (const u32x4*)wq; p.bias = bias;
    p.N = 1; p.Cin = 64; p.Cout = 64; p.H = 2 * h; p.W = 2 * wd; p.Hin = h; p.Win = wd;
    p.ksteps = 4; p.coutPad = 256; p.cgroups = 1;
    p.tilesX = (wd + ST_W - 1) / ST_W; p.tilesY = (h + ST_H - 1) / ST_H;
    p.act = act; p.slope = slope;
    ISR_DIAG_SET(p.dbg, g_split_dbg); ISR_DIAG_SET(p.stamps, g_split_stamps);
    p.xps = (const u32x4*)xps; p.xpsPlane = (int)xpsPlane;
    p.ps = (u32x4*)ps; p.psPlane = (int)psPlane;
    p.absmax = rangeFlag;
    hipStream_t s = (hipStream_t)stream;
    static bool attr = false;
    static int ldsExtra = isr_diag_env_int("ISR_UPSP_LDS_EXTRA", 0);      // experiment: pad the allocation (one workgroup per CU)
    const int ldsBytes = UP_LDS_BYTES + ldsExtra;
    if (!attr) { (void)hipFuncSetAttribute((const void*)conv3x3_split_upsp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, ldsBytes); attr = true; }
    // the one-pixel frame first (a few dozen waves), then the body: neither reads what the other writes
    UpsFrameParams fp;
    fp.xps = p.xps; fp.xpsPlane = p.xpsPlane; fp.w = w; fp.bias = bias; fp.ps = p.ps; fp.psPlane = p.psPlane;
    fp.Hin = h; fp.Win = wd; fp.H = p.H; fp.W = p.W; fp.act = act; fp.slope = slope; fp.absmax = rangeFlag;
    const int nf = 2 * p.W + 2 * (p.H - 2);
    ISR_LAUNCH_PROFILED(ISR_VARIANT_UPS_FRAME, ups_frame_kernel, dim3((unsigned)((nf + 63) / 64), 8), dim3(64), 0, s, fp);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    isr_profile_record(ISR_VARIANT_SPLIT_UPSP, 2.0 * 9 * 64 * 64 * (double)p.H * p.W, &e0, &e1);
    static int form = isr_diag_env_int("ISR_UPSP_FORM", 0);      // 0: the LDS-DMA form (default), 1: form Q (4 rows per wave, activations from L1; measured slower)
    const dim3 block(S_THREADS);
    if (form == 1) {
        p.tilesY = (h + UQ_TILE_H - 1) / UQ_TILE_H;
        const dim3 qgrid((unsigned)(p.tilesX * p.tilesY));
        const int qlds = UQ_LDS_BYTES + ldsExtra;
        if (e0 || e1) hipExtLaunchKernelGGL(conv3x3_split_upsq_kernel, qgrid, block, qlds, s, e0, e1, 0, p);
        else hipLaunchKernelGGL(conv3x3_split_upsq_kernel, qgrid, block, qlds, s, p);
        return hipGetLastError() == hipSuccess ? 0 : -2;
    }
    const dim3 grid((unsigned)(p.tilesX * p.tilesY));
    if (e0 || e1) hipExtLaunchKernelGGL(conv3x3_split_upsp_kernel, grid, block, ldsBytes, s, e0, e1, 0, p);
    else hipLaunchKernelGGL(conv3x3_split_upsp_kernel, grid, block, ldsBytes, s, p);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

#endif

int isrPackSplit(const float* x, void* ps, int C, int H, int W, long long xPlane, long long psPlane, void* stream)
{
    if (!x || !ps || C <= 0 || (C & 7) || H <= 0 || W <= 0 || xPlane < (long long)H * W || psPlane < (long long)H * W) return -1;
    if (psPlane * 16 * 2 * (C / 8) > 0x7fffffffLL || ((uintptr_t)ps & 15)) return -1;
    const int npix = H * W;
    hipLaunchKernelGGL(pack_split_kernel, dim3((unsigned)((npix + 255) / 256), (unsigned)(C / 8)), dim3(256), 0, (hipStream_t)stream,
                       x, (u32x4*)ps, C / 8, npix, xPlane, (int)psPlane);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int isrConvSplitPrepareManyMax(void) { return PM_MAX; }

int isrConvSplitPrepareMany(int n, const float* const* w, void* const* wqForward, void* const* wqBackward, const int* cout, const int* cin, void* stream)
{
    if (n <= 0 || n > PM_MAX || !w || !cout || !cin || (!wqForward && !wqBackward)) return -1;
    PrepManyParams p;
    int most = 0;
    for (int l = 0; l < PM_MAX; ++l) {
        const bool on = l < n;
        p.w[l] = on ? w[l] : nullptr;
        p.fwd[l] = on && wqForward ? (u32x4*)wqForward[l] : nullptr;
        p.bwd[l] = on && wqBackward ? (u32x4*)wqBackward[l] : nullptr;
        p.cout[l] = on ? cout[l] : 0; p.cin[l] = on ? cin[l] : 0;
        if (on) {
            if (!w[l] || cout[l] <= 0 || cin[l] <= 0 || (!p.fwd[l] && !p.bwd[l])) return -1;
            const int a = 9 * ((cin[l] + 15) / 16) * 2 * (((cout[l] + 31) / 32) * 32);
            const int b = 9 * ((cout[l] + 15) / 16) * 2 * (((cin[l] + 31) / 32) * 32);
            most = a > most ? a : most; most = b > most ? b : most;
        }
    }
    p.n = n;
    hipLaunchKernelGGL(split_scale_many_kernel, dim3(n), dim3(1024), 0, (hipStream_t)stream, p);
    hipLaunchKernelGGL(prepare_weights_split_many_kernel, dim3((most + 255) / 256, 2 * n), dim3(256), 0, (hipStream_t)stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int isrResBlockSmallSupported(int N, int H, int W)
{
    if (N <= 0 || H <= 0 || W <= 0 || (W & 3) || W > ST_W) return 0;
    const long long tiles = (long long)N * ((H + R2_H - 1) / R2_H);
    return tiles >= 64 && tiles <= 0x7fffffffLL && (long long)64 * H * W * 4 <= 0x7fffffffLL;
}

int isrResBlockSmall(const float* x, const void* wa, const float* ba, const float* gate, const void* wb, const float* bb, float* z, float* y,
                     int N, int H, int W, void* zmax, void* ymax, void* stream)
{
    unsigned* const rangeFlag = isr_take_range_flag();       // taken first (see isrConv3x3ForwardSplit)
    if (!x || !wa || !wb || !z || !y || (!zmax) != (!ymax)) return -1;
    if (!isrResBlockSmallSupported(N, H, W)) return -3;
    if (((uintptr_t)x & 15) || ((uintptr_t)y & 15) || ((uintptr_t)wa & 15) || ((uintptr_t)wb & 15)) return -1;
    Block2Params p;
    p.x = x; p.wa = (const u32x4*)wa; p.ba = ba; p.gate = gate; p.wb = (const u32x4*)wb; p.bb = bb; p.z = z; p.y = y;
    p.N = N; p.H = H; p.W = W; p.tilesY = (H + R2_H - 1) / R2_H;
    p.absmax = rangeFlag;
    p.zmax = (unsigned*)zmax; p.ymax = (unsigned*)ymax;
    ISR_DIAG_SET(p.dbg, g_split_dbg); ISR_DIAG_SET(p.stamps, g_split_stamps);
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)conv3x3_split_block2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, B2_LDS_BYTES); attr = true; }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    isr_profile_record(ISR_VARIANT_SPLIT_BLOCK2, 2.0 * 2.0 * 9 * 64 * 64 * (double)N * H * W, &e0, &e1);
    const dim3 grid((unsigned)(N * p.tilesY)), block(S_THREADS);
    hipStream_t s = (hipStream_t)stream;
    if (e0 || e1) hipExtLaunchKernelGGL(conv3x3_split_block2_kernel, grid, block, B2_LDS_BYTES, s, e0, e1, 0, p);
    else hipLaunchKernelGGL(conv3x3_split_block2_kernel, grid, block, B2_LDS_BYTES, s, p);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

} // extern "C"
