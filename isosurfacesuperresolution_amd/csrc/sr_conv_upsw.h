// conv3x3(U(x)) for EnhanceNet's two upsampling layers (SuperresolutionNetwork/models/enhancenet.py:113-124) as ONE instruction stream per
// SIMD: a persistent workgroup per CU (four waves, one per SIMD, the SIMD's 512 registers and the CU's 160 KB of LDS to itself) walks an
// XCD-contiguous range of 16 x 32 output tiles, and everything that is not an MFMA -- the low-resolution fetch, its fp32 copy, the bilinear
// blend + (hi, lo') split of the NEXT k-step's patch, the weight DMA -- is cut into slices of <= 5 vector instructions that sit in the gaps
// between the MFMAs of the CURRENT k-step, in the multiplying wave's own stream.  Included by sr_conv_split.hip; the arithmetic (interpolation,
// split, products, their order per accumulator) is conv3x3_split_kernel<true>'s: bit-identical (tests/test_ups_gpu.py).
//
// Why.  Vector instructions of ANOTHER wave do not hide under a wave's MFMAs on this chip (profiles/r04_mfma_valu_overlap.txt: role-split waves =
// the sum of both times), a wave's own do, up to ~5 per 32-cycle MFMA (MI355X_MICROARCH.md, cycle constants).  The kernels that stage and
// multiply in separate phases therefore add their vector time to their matrix time whatever the occupancy: the three-per-CU kernel keeps the
// matrix pipe 47 % busy, the four-row one (sr_conv_ups4r.h) 45 % -- per tile 16 us staging + 27 us MFMA rows (13 us of matrix cycles) + 14 us
// epilogue (profiles/r06_ups4r_timeline.txt).  Here a k-step's 216 MFMAs per wave carry ~150 slices (3.3 vector / LDS instructions per gap).
//
// LDS (163 584 of 163 840 B): two patch slices (a k-step = 16 channels of the 18 x 34 patch, hi + lo': 39 168 B each), two weight slots (a
// k-step's nine taps, hi + lo: 36 864 B each), the fp32 copy of the low-resolution region (16 channels x 10 x 18: 11 520 B).
// ONE barrier per k-step (216 MFMAs per wave).  A wave stages the four channels 4 w .. 4 w + 3 of the next k-step end to end (fetch, copy,
// blend, split), so the copy needs no barrier of its own.
#pragma once
#include "sr_split_common.h"

#ifndef UW_SPREAD
#define UW_SPREAD 1
#endif
#ifndef UW_MEMFENCE
#define UW_MEMFENCE 1
#endif
#ifndef UW_LAUNDER
#define UW_LAUNDER 1
#endif

namespace {

constexpr int UW_TH = 16, UW_TW = 32;                                        // output tile
constexpr int UW_PH = UW_TH + 2, UW_PW = UW_TW + 2, UW_PIX = UW_PH * UW_PW;  // 18 x 34 = 612 patch pixels
constexpr int UW_PART = 2 * UW_PIX;                                          // one k-step of the patch: 2 channel groups; hi, then lo' at + UW_PART
constexpr int UW_PUNITS = 2 * UW_PART;                                       // 2448 units = 39 168 B
constexpr int UW_WPART = 9 * 128;                                            // weights of a k-step, one part: [tap][lane half][64 couts]
constexpr int UW_WUNITS = 2 * UW_WPART;                                      // hi then lo: 36 864 B
constexpr int UW_LR_H = UW_TH / 2 + 2, UW_LR_W = UW_TW / 2 + 2;              // 10 x 18 low-resolution pixels
constexpr int UW_LQ = (UW_TW / 2 + 8) / 4;                                   // 6 aligned quads per low-resolution row
constexpr int UW_LR_CS = UW_LR_H * UW_LR_W;                                  // 180 floats per channel
constexpr int UW_WAVE_QUADS = 4 * UW_LR_H * UW_LQ;                           // a wave's share of a k-step's fetch: 4 channels = 240 quads (3.75 per lane)
constexpr int UW_QR = UW_PH / 2, UW_QC = UW_PW / 2, UW_UQ = UW_QR * UW_QC;   // 9 x 17 = 153 quads of 2 x 2 patch pixels: 2.4 per lane
constexpr int UW_LDS_BYTES = (2 * UW_PUNITS + 2 * UW_WUNITS) * 16 + 16 * UW_LR_CS * 4 + 256;     // (+ the parking sink: 163 840 = all of it)
static_assert(UW_LDS_BYTES <= 160 * 1024, "one workgroup per CU");

template <int N> struct UwInt { static constexpr int value = N; };
template <int N, int I = 0, typename F>
__device__ __forceinline__ void uw_static_for(F&& f)
{
    if constexpr (I < N) { f(UwInt<I>{}); uw_static_for<N, I + 1>(f); }
}

struct UwTile { int n, oy0, ox0; bool interior; };

__device__ __forceinline__ UwTile uw_tile(const SplitConvParams& p, int t)
{
    UwTile d;
    const int tx = t % p.tilesX; t /= p.tilesX;
    const int ty = t % p.tilesY; d.n = t / p.tilesY;
    d.oy0 = ty * UW_TH; d.ox0 = tx * UW_TW;
    d.interior = d.oy0 >= 2 && d.oy0 + UW_TH + 2 <= p.H && d.ox0 >= 2 && d.ox0 + UW_TW + 2 <= p.W;
    return d;
}

// ---- a wave's staging work for one k-step of one tile, as pieces ------------------------------------------------------------------------
// What depends on the lane only is computed once per launch and kept PACKED (two registers per fetch quad, one per parked quad): hoisted out of
// the step loop piecemeal by the optimiser the same values took ~50 registers and spilled.
struct UwLane {
    unsigned foff[4];      // fetch quad k: byte offset of (channel c, row r, quad q) relative to the wave's first channel / the region's first row and quad
    unsigned frq[4];       // r | q << 8 | (the lane takes part) << 16
    unsigned park[4];      // parked quad k: float index of its first value in the wave's copy, + 3 | (store value 0, 1 / 2, 3) << 16 .. 18
    unsigned unit[3];      // blended quad of iteration it: float index of its first source value in the wave's copy | index of its first fp16 in the patch slice << 12
};

__device__ __forceinline__ UwLane uw_lane_setup(const SplitConvParams& p, int lane)
{
    UwLane L;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int u = lane + 64 * k;
        const int c = u / (UW_LR_H * UW_LQ), rem = u - c * (UW_LR_H * UW_LQ);
        const int r = rem / UW_LQ, q = rem - r * UW_LQ;
        const bool live = u < UW_WAVE_QUADS;
        L.foff[k] = (unsigned)c * ((unsigned)p.xPlane * 4u) + (unsigned)(r * p.Win + 4 * q) * 4u;
        L.frq[k] = (unsigned)r | ((unsigned)q << 8) | (live ? 1u << 16 : 0u);
        L.park[k] = (unsigned)(c * UW_LR_CS + r * UW_LR_W + 4 * q) | ((live && q > 0) ? 1u << 16 : 0u) | ((live && q > 0 && q < UW_LQ - 1) ? 1u << 17 : 0u)
                    | ((live && q < UW_LQ - 1) ? 1u << 18 : 0u);
    }
#pragma unroll
    for (int it = 0; it < 3; ++it) {
        int q = lane + 64 * it;
        q = q < UW_UQ ? q : UW_UQ - 1;                                       // (a repeated quad rewrites the same bytes)
        const int kr = q / UW_QC, kc = q - kr * UW_QC;
        L.unit[it] = (unsigned)(kr * UW_LR_W + kc) | ((unsigned)((2 * kr) * UW_PW + 2 * kc) << 12);
    }
    return L;
}

// request k (0 .. 3) of the wave's 240 quads of the low-resolution region: channels cin0 + 4 wave .. + 3, rows oy0 / 2 - 1 .., aligned quads
struct UwFetch { rsrc_t xrs; unsigned base; int ly0, lxq; bool live; };      // per step (wave uniform)

__device__ __forceinline__ UwFetch uw_fetch_setup(const SplitConvParams& p, const UwTile& d, int cin0, int wave, bool live)
{
    UwFetch f;
    f.ly0 = d.oy0 / 2 - 1; f.lxq = d.ox0 / 2 - 4; f.live = live;
    f.base = (unsigned)(cin0 + 4 * wave) * ((unsigned)p.xPlane * 4u) + (unsigned)(f.ly0 * p.Win + f.lxq) * 4u;      // (wraps for rows / quads outside: those lanes are not ok)
    f.xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x + (size_t)d.n * p.xImage), 0, (int)((size_t)p.Cin * p.xPlane * 4), 0x00020000);
    return f;
}

__device__ __forceinline__ unsigned uw_fetch_offset(const SplitConvParams& p, const UwLane& L, const UwFetch& f, int k)
{
    const int iy = f.ly0 + (int)(L.frq[k] & 255u), ix = f.lxq + 4 * (int)((L.frq[k] >> 8) & 255u);
    const int ok = (int)f.live & (int)(L.frq[k] >> 16) & (int)((unsigned)iy < (unsigned)p.Hin) & (int)((unsigned)ix < (unsigned)p.Win);      // (no short circuit: no branches between MFMAs)
    return ok ? f.base + L.foff[k] : BAD_OFFSET;
}

__device__ __forceinline__ u32x4 uw_fetch_load(const UwFetch& f, unsigned off) { return __builtin_amdgcn_raw_buffer_load_b128(f.xrs, (int)off, 0, 0); }

// value i (0 .. 3) of quad k of the wave's fetch into the wave's four channels of the fp32 copy (quad q holds low-resolution patch columns
// 4 q - 3 .. 4 q).  Branch-free -- the slices sit between MFMAs: values that belong to no column of the region go to the lane's own word of a
// 256-byte sink.
__device__ __forceinline__ void uw_park(float* tmpw, float* mine, const UwLane& L, u32x4 v, int k, int i)
{
    float* dst = tmpw + (int)(L.park[k] & 0xffffu) - 3 + i;
    const unsigned bit = i == 0 ? 1u << 16 : i == 3 ? 1u << 18 : 1u << 17;
    const float4 f = __builtin_bit_cast(float4, v);
    *((L.park[k] & bit) ? dst : mine) = i == 0 ? f.x : i == 1 ? f.y : i == 2 ? f.z : f.w;
}

// piece i (0 .. 8) of the wave's share of k-step ks's weights (36 wave-wide pieces (tap, part, half), nine per wave: wave w moves (tap i, part w / 2,
// half w % 2)) into a weight slot
struct UwDma { rsrc_t wrs; u32x4* dst0; int soff0, sstride; };               // per step (wave uniform)

__device__ __forceinline__ UwDma uw_dma_setup(const SplitConvParams& p, u32x4* wslot, int ks, int wv)
{
    UwDma d;
    d.wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(p.wq + 1), 0, 9 * p.ksteps * 4096, 0x00020000);
    const int part = (wv >> 1) & 1, half = wv & 1;
    d.dst0 = wslot + part * UW_WPART + half * 64;
    d.soff0 = (ks * 256 + part * 128 + half * 64) * 16; d.sstride = p.ksteps * 4096;
    return d;
}

__device__ __forceinline__ void uw_wdma(const UwDma& d, int lane, int i)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(d.wrs, (isr_lvoid_t*)(d.dst0 + i * 128), 16, lane * 16, d.soff0 + i * d.sstride, 0, 0);
}

// The general blend + split of a wave's four channels (any tile: clamped source indices, zero padding outside the image), not overlapped with
// anything: the first k-step of a workgroup and the k-steps of the image's outermost ring of tiles (6 % of them at 1080p).
__device__ __forceinline__ void uw_interpolate_general(const SplitConvParams& p, const UwTile& d, const float* tmpw, u32x4* pslice, int wave, int lane)
{
    _Float16* const patch16 = reinterpret_cast<_Float16*>(pslice);
    const int ly0 = d.oy0 / 2 - 1, lx0 = d.ox0 / 2 - 1;
    for (int q = lane; q < UW_UQ; q += 64) {
        const int kr = q / UW_QC, kc = q - kr * UW_QC;
        f16x4 h00, h01, h10, h11, l00, l01, l10, l11;
        const int Yu = d.oy0 - 1 + 2 * kr, Xl = d.ox0 - 1 + 2 * kc;
        const bool oku = (unsigned)Yu < (unsigned)p.H, okd = (unsigned)(Yu + 1) < (unsigned)p.H;
        const bool okl = (unsigned)Xl < (unsigned)p.W, okr = (unsigned)(Xl + 1) < (unsigned)p.W;
        int y0, y1, x0, x1, t0, t1; float lyu, lyd, lxl, lxr, t;
        isr_src_index(oku ? Yu : Yu + 1, 0.5f, p.Hin, y0, y1, t);           // both rows of the pair blend these two source rows
        isr_src_index(okl ? Xl : Xl + 1, 0.5f, p.Win, x0, x1, t);
        isr_src_index(Yu, 0.5f, p.Hin, t0, t1, lyu);
        isr_src_index(Yu + 1, 0.5f, p.Hin, t0, t1, lyd);
        isr_src_index(Xl, 0.5f, p.Win, t0, t1, lxl);
        isr_src_index(Xl + 1, 0.5f, p.Win, t0, t1, lxr);
        const float hyu = 1.f - lyu, hyd = 1.f - lyd, hxl = 1.f - lxl, hxr = 1.f - lxr;
        // rows / columns wholly outside the image (tile overhang) keep their indices inside the staged region
        y0 = min(max(y0 - ly0, 0), UW_LR_H - 1); y1 = min(max(y1 - ly0, 0), UW_LR_H - 1);
        x0 = min(max(x0 - lx0, 0), UW_LR_W - 1); x1 = min(max(x1 - lx0, 0), UW_LR_W - 1);
        const float* ta = tmpw + y0 * UW_LR_W;
        const float* tb = tmpw + y1 * UW_LR_W;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a0 = ta[e * UW_LR_CS + x0], a1 = ta[e * UW_LR_CS + x1];
            const float b0 = tb[e * UW_LR_CS + x0], b1 = tb[e * UW_LR_CS + x1];
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
        // 16-byte unit (8-channel group wave / 2, pixel) holds 8 halves: this wave's 4-channel group is its half (wave & 1)
        _Float16* dd = patch16 + ((size_t)((wave >> 1) * UW_PIX + (2 * kr) * UW_PW + 2 * kc)) * 8 + (wave & 1) * 4;
        *reinterpret_cast<f16x4*>(dd) = h00;
        *reinterpret_cast<f16x4*>(dd + 8) = h01;
        *reinterpret_cast<f16x4*>(dd + UW_PW * 8) = h10;
        *reinterpret_cast<f16x4*>(dd + UW_PW * 8 + 8) = h11;
        *reinterpret_cast<f16x4*>(dd + UW_PART * 8) = l00;
        *reinterpret_cast<f16x4*>(dd + UW_PART * 8 + 8) = l01;
        *reinterpret_cast<f16x4*>(dd + (UW_PART + UW_PW) * 8) = l10;
        *reinterpret_cast<f16x4*>(dd + (UW_PART + UW_PW) * 8 + 8) = l11;
    }
}

// ---- the slices of an INTERIOR tile's staging (every patch pixel inside the image, no clamped index: the blend weights are the constants
// 3/4, 1/4 -- the same operations on the same values in the same order as isr_src_index gives there, sr_conv_ups3.h) -----------------------
// Slice numbering (one per MFMA gap, UW_SLICES of a k-step's 216):
//   0 .. 15   park quad k = Q / 4, value Q % 4                     16 .. 23  fetch quad k of the k-step after next: offset, then the load
//   24 .. 32  weight DMA piece i
//   then per unit iteration it (0 .. 2: quad lane + 64 it), 43 slices:
//     +0              addresses, request channel 0's four values
//     +1 + 10 e ..    channel e: [al, ar | request channel e + 1] [bl, br] then per pixel (00, 01, 10, 11): [blend, hi] [lo', insert]
//     +41, +42        the unit's eight 8-byte stores
constexpr int UW_S_PARK = 0, UW_S_FETCH = 16, UW_S_DMA = 24, UW_S_UNIT = 33, UW_S_PER_UNIT = 43, UW_SLICES = UW_S_UNIT + 3 * UW_S_PER_UNIT;   // 162
static_assert(UW_SLICES == 216 * 3 / 4, "three gaps of four carry a slice");

struct UwStage {
    const SplitConvParams* p;
    UwFetch fetch;                     // step s + 2
    UwDma dma;                         // step s + 1
    float* tmpw; float* mine; u32x4* pslice;
    int wave, lane;
    UwLane LN;
    // state
    u32x4 v[4];
    unsigned fo;
    const float* ta; _Float16* dd;
    float cur[4], nxt[4];              // a0, a1, b0, b1 of the channel being blended / of the next one
    float al, ar, bl, br, val;
    _Float16 vh;
    f16x4 H[4], L[4];                  // pixels 00, 01, 10, 11

    template <int Q> __device__ __forceinline__ void slice()
    {
        if constexpr (Q < UW_S_FETCH) uw_park(tmpw, mine, LN, v[Q >> 2], Q >> 2, Q & 3);
        else if constexpr (Q < UW_S_DMA) {
            constexpr int k = (Q - UW_S_FETCH) >> 1;
            if constexpr (((Q - UW_S_FETCH) & 1) == 0) fo = uw_fetch_offset(*p, LN, fetch, k);
            else v[k] = uw_fetch_load(fetch, fo);
        } else if constexpr (Q < UW_S_UNIT) uw_wdma(dma, lane, Q - UW_S_DMA);
        else if constexpr (Q < UW_SLICES) {
            constexpr int it = (Q - UW_S_UNIT) / UW_S_PER_UNIT, s = (Q - UW_S_UNIT) % UW_S_PER_UNIT;
            if constexpr (s == 0) {
                ta = tmpw + (int)(LN.unit[it] & 0xfffu);
                dd = reinterpret_cast<_Float16*>(pslice) + ((size_t)((wave >> 1) * UW_PIX) + (LN.unit[it] >> 12)) * 8 + (wave & 1) * 4;
                nxt[0] = ta[0]; nxt[1] = ta[1]; nxt[2] = ta[UW_LR_W]; nxt[3] = ta[UW_LR_W + 1];
            } else if constexpr (s <= 40) {
                constexpr int e = (s - 1) / 10, j = (s - 1) % 10;
                if constexpr (j == 0) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) cur[i] = nxt[i];
                    if constexpr (e < 3) {
                        nxt[0] = ta[(e + 1) * UW_LR_CS]; nxt[1] = ta[(e + 1) * UW_LR_CS + 1];
                        nxt[2] = ta[(e + 1) * UW_LR_CS + UW_LR_W]; nxt[3] = ta[(e + 1) * UW_LR_CS + UW_LR_W + 1];
                    }
                    al = isr_blend(0.75f, cur[0], 0.25f, cur[1]); ar = isr_blend(0.25f, cur[0], 0.75f, cur[1]);
                    if (UW_LAUNDER) asm volatile("" : "+v"(al), "+v"(ar));                   // (pinned: the optimiser gathers what it may into packed fp32 clumps of 200 cycles)
                } else if constexpr (j == 1) {
                    bl = isr_blend(0.75f, cur[2], 0.25f, cur[3]); br = isr_blend(0.25f, cur[2], 0.75f, cur[3]);
                    if (UW_LAUNDER) asm volatile("" : "+v"(bl), "+v"(br));
                } else {
                    constexpr int px = (j - 2) / 2;                          // 0: 00, 1: 01, 2: 10, 3: 11
                    if constexpr (((j - 2) & 1) == 0) {
                        const float up = (px & 1) ? ar : al, dn = (px & 1) ? br : bl;           // the column's value in the upper / lower source row
                        val = (px & 2) ? isr_blend(0.25f, up, 0.75f, dn) : isr_blend(0.75f, up, 0.25f, dn);
                        // (the blend rounds to fp32 FIRST, as in every other kernel: left alone the optimiser fuses blend + conversion into one
                        // v_fma_mixlo_f16 -- a single rounding, another hi in the rare double-rounding cases, 1-ulp differences in 0.1 % of the outputs)
                        asm volatile("" : "+v"(val));
                        vh = (_Float16)val;
                        if (UW_LAUNDER) asm volatile("" : "+v"(vh));
                    } else {
                        const _Float16 vl = (_Float16)((val - (float)vh) * 2048.0f);       // split16x
                        H[px][e] = vh; L[px][e] = vl;
                        if (UW_LAUNDER) asm volatile("" : "+v"(H[px]), "+v"(L[px]));
                    }
                }
            } else if constexpr (s == 41) {
                *reinterpret_cast<f16x4*>(dd) = H[0];
                *reinterpret_cast<f16x4*>(dd + 8) = H[1];
                *reinterpret_cast<f16x4*>(dd + UW_PW * 8) = H[2];
                *reinterpret_cast<f16x4*>(dd + UW_PW * 8 + 8) = H[3];
            } else {
                *reinterpret_cast<f16x4*>(dd + UW_PART * 8) = L[0];
                *reinterpret_cast<f16x4*>(dd + UW_PART * 8 + 8) = L[1];
                *reinterpret_cast<f16x4*>(dd + (UW_PART + UW_PW) * 8) = L[2];
                *reinterpret_cast<f16x4*>(dd + (UW_PART + UW_PW) * 8 + 8) = L[3];
            }
        }
    }
};

// The MFMA as a volatile statement with its accumulator pinned to the AGPR half of the register file: statements of this kind keep their order,
// and the 128 accumulators never travel (as builtins the register allocator moved all of them between the two halves at every k-step boundary:
// 256 extra vector instructions per 216 MFMAs).  Hazards the compiler no longer sees: none inside a k-step (an accumulator's next MFMA is eight
// MFMAs away, operand registers are written >= 12 MFMAs after their last reader issued); behind the last k-step of a tile uw_drain() pads the
// matrix pipe's write-back before vector instructions read the accumulators.
#ifndef UW_ASM_MFMA
#define UW_ASM_MFMA 0
#endif
__device__ __forceinline__ void uw_mfma(f32x16& c, f16x8 a, f16x8 b)
{
    if (UW_ASM_MFMA) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    else c = mfma16(a, b, c);
}
// (zeroed by the matrix pipe itself -- 0 x 0 + 0, eight MFMAs per tile: a value defined by a vector instruction would make the allocator keep the
// accumulators in the VGPR half across the loop and copy all 128 into the AGPR half in every k-step)
__device__ __forceinline__ void uw_zero(f32x16 (&acc)[2][2][2])
{
    const f16x8 z = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                if (UW_ASM_MFMA) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %1, 0" : "=a"(acc[hf][cb][r]) : "v"(z));
                else {
                    // (an MFMA of its own per accumulator -- the operand laundered so that they are not merged: ONE zeroing MFMA whose result was
                    // copied into the other seven accumulators by v_accvgpr_mov left the last register of each copy unzeroed on the second and
                    // later tiles of a workgroup: the copies read the matrix pipe's result a pass too early)
                    f32x16 zero;
#pragma unroll
                    for (int i = 0; i < 16; ++i) zero[i] = 0.0f;
                    f16x8 zz = z;
                    asm volatile("" : "+v"(zz));
                    acc[hf][cb][r] = mfma16(zz, zz, zero);
                }
            }
}
__device__ __forceinline__ void uw_drain() { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); }

// One k-step of MFMAs of a wave (4 rows x 32 pixels x 64 channels: 9 taps x 24), operands prefetched a tap ahead (two register sets), with
// a slice of the staging in three of every four gaps.  Per accumulator the three products of a tap in the order a_lo b_hi, a_hi' b_lo',
// a_hi b_hi, taps in the order dy, dx: the order of every split-operand kernel -- the same bits.
__device__ __forceinline__ void uw_kstep(f32x16 (&acc)[2][2][2], const u32x4* wl, const u32x4* bl, UwStage& st)
{
    f16x8 aH[2][2], aL[2][2], aS[2][2], bh[2][4], bo[2][4];
    auto request = [&](auto tc, auto kc) {                                   // fragment k (0 .. 11) of tap t into register set t & 1
        constexpr int t = decltype(tc)::value, k = decltype(kc)::value, set = t & 1, dy = t / 3, dx = t % 3;
        if constexpr (k == 0) aL[set][0] = __builtin_bit_cast(f16x8, wl[UW_WPART + t * 128]);
        else if constexpr (k == 1) aL[set][1] = __builtin_bit_cast(f16x8, wl[UW_WPART + t * 128 + 32]);
        else if constexpr (k == 2) aH[set][0] = __builtin_bit_cast(f16x8, wl[t * 128]);
        else if constexpr (k == 3) aH[set][1] = __builtin_bit_cast(f16x8, wl[t * 128 + 32]);
        else if constexpr (k < 8) bh[set][k - 4] = __builtin_bit_cast(f16x8, bl[(k - 4 + dy) * UW_PW + dx]);
        else bo[set][k - 8] = __builtin_bit_cast(f16x8, bl[UW_PART + (k - 8 + dy) * UW_PW + dx]);
    };
    uw_static_for<12>([&](auto kc) { request(UwInt<0>{}, kc); });
    aS[0][0] = aH[0][0] * (_Float16)0.00048828125f;                          // w_hi 2^-11: partner of the scaled x_lo'
    aS[0][1] = aH[0][1] * (_Float16)0.00048828125f;
    uw_static_for<216>([&](auto mc) {
        constexpr int m = decltype(mc)::value, t = m / 24, g = (m % 24) / 8, r = (m % 8) >> 1, cb = m & 1, set = t & 1;
        if (UW_MEMFENCE) asm volatile("" ::: "memory");                      // (no LDS / memory operation changes gaps in the optimiser either)
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (g == 0) uw_mfma(acc[r >> 1][cb][r & 1], aL[set][cb], bh[set][r]);
        else if constexpr (g == 1) uw_mfma(acc[r >> 1][cb][r & 1], aS[set][cb], bo[set][r]);
        else uw_mfma(acc[r >> 1][cb][r & 1], aH[set][cb], bh[set][r]);
        // the next tap's twelve fragments: one request per gap behind the tap's first twelve MFMAs; its scaled weights behind the next eight
        if constexpr (t < 8 && m % 24 < 12) request(UwInt<t + 1>{}, UwInt<m % 24>{});
        if constexpr (t < 8 && m % 24 >= 12 && m % 24 < 20) {
            constexpr int i = m % 24 - 12, c2 = i >> 2, w = i & 3;            // one v_pk_mul_f16 per gap
            typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
            const f16x2 src = {aH[set ^ 1][c2][2 * w], aH[set ^ 1][c2][2 * w + 1]};
            const f16x2 dst = src * (_Float16)0.00048828125f;
            aS[set ^ 1][c2][2 * w] = dst[0]; aS[set ^ 1][c2][2 * w + 1] = dst[1];
        }
        if constexpr (!UW_SPREAD && m < UW_SLICES) st.template slice<m>();
        if constexpr (UW_SPREAD && m % 4 != 3) st.template slice<m - m / 4>();              // (a slice costs up to ~24 issue cycles: three gaps of four keep the average under the MFMA's shadow)
    });
    __builtin_amdgcn_sched_barrier(0);
}

__global__ __launch_bounds__(S_THREADS, 1) void conv3x3_split_upsw_kernel(const SplitConvParams p)
{
    extern __shared__ u32x4 lds[];
    u32x4* const pbuf = lds;                                                 // two patch slices
    u32x4* const wbuf = lds + 2 * UW_PUNITS;                                 // two weight slots
    float* const tmp = reinterpret_cast<float*>(lds + 2 * UW_PUNITS + 2 * UW_WUNITS);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    float* const tmpw = tmp + 4 * wv * UW_LR_CS;
    // this workgroup's tiles: an XCD (= an L2) gets a contiguous range of the tile list, its workgroups contiguous pieces of that
    int lid;
    {
        const int G = gridDim.x, q = G >> 3, r = G & 7, xcd = blockIdx.x & 7;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    }
    const int T = p.N * p.tilesX * p.tilesY;
    const int t0 = (int)((long long)lid * T / (int)gridDim.x), t1 = (int)((long long)(lid + 1) * T / (int)gridDim.x);
    const int nsteps = (t1 - t0) * p.ksteps;
    if (nsteps <= 0) return;

    UwStage st;
    st.p = &p; st.tmpw = tmpw; st.mine = tmp + 16 * UW_LR_CS + lane; st.wave = wave; st.lane = lane;
    st.LN = uw_lane_setup(p, lane);
    st.ta = tmpw; st.dd = reinterpret_cast<_Float16*>(pbuf); st.fo = BAD_OFFSET;
    // ---- prologue: step 0 staged in the open, step 1's fetch on its way
    UwTile cur = uw_tile(p, t0);
    {
        const UwFetch f0 = uw_fetch_setup(p, cur, 0, wv, true);
#pragma unroll
        for (int k = 0; k < 4; ++k) st.v[k] = uw_fetch_load(f0, uw_fetch_offset(p, st.LN, f0, k));
        const UwDma d0 = uw_dma_setup(p, wbuf, 0, wv);
#pragma unroll
        for (int i = 0; i < 9; ++i) uw_wdma(d0, lane, i);
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i) uw_park(tmpw, st.mine, st.LN, st.v[k], k, i);
        uw_interpolate_general(p, cur, tmpw, pbuf, wave, lane);
        const UwFetch f1 = uw_fetch_setup(p, uw_tile(p, t0 + 1 / p.ksteps), 16 * (1 % p.ksteps), wv, nsteps > 1);
#pragma unroll
        for (int k = 0; k < 4; ++k) st.v[k] = uw_fetch_load(f1, uw_fetch_offset(p, st.LN, f1, k));
    }
    f32x16 acc[2][2][2];
    uw_zero(acc);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                         // the weights of step 0 (the four fetches behind them may still fly)
    __syncthreads();

    // diagnostics (U4R_DIAG builds, stamp buffer set): 8 words per workgroup -- start | sum over the steps of: the k-step with its slices, the
    // border staging, the wait + barrier | sum of the epilogues | end | number of border steps (ticks of the 100 MHz clock)
    unsigned long long* const stamps = U4R_DIAG ? p.stamps : nullptr;
    unsigned long long tStart = 0, tStep = 0, tBorder = 0, tWait = 0, tEpi = 0, nBorder = 0, ta = 0, tb = 0, ca = 0, cStep = 0;
    if (stamps) tStart = __builtin_amdgcn_s_memrealtime();
    int tile = t0, ks = 0;
#pragma unroll 1
    for (int s = 0; s < nsteps; ++s) {
        const int slot = s & 1;
        const bool hasNext = s + 1 < nsteps;
#pragma unroll
        for (int k = 0; k < 4; ++k) asm volatile("" : "+v"(st.LN.foff[k]), "+v"(st.LN.frq[k]), "+v"(st.LN.park[k]));      // (nothing derived from them leaves the loop)
#pragma unroll
        for (int k = 0; k < 3; ++k) asm volatile("" : "+v"(st.LN.unit[k]));
        // steps s + 1 (blend target, weights) and s + 2 (fetch).  The slices run in EVERY step: behind the last step of the workgroup they
        // stage a step that does not exist (fetch switched off, everything else lands in buffers nobody reads any more).
        int ks1 = ks + 1, tile1 = tile;
        if (ks1 == p.ksteps) { ks1 = 0; ++tile1; }
        int ks2 = ks1 + 1, tile2 = tile1;
        if (ks2 == p.ksteps) { ks2 = 0; ++tile2; }
        const bool live2 = s + 2 < nsteps;
        const UwTile next = uw_tile(p, hasNext ? tile1 : tile);
        st.fetch = uw_fetch_setup(p, uw_tile(p, live2 ? tile2 : tile), 16 * ks2, wv, live2);
        st.dma = uw_dma_setup(p, wbuf + (slot ^ 1) * UW_WUNITS, ks1, wv);
        st.pslice = pbuf + (slot ^ 1) * UW_PUNITS;
        const u32x4* wl = wbuf + slot * UW_WUNITS + h * 64 + j;
        const u32x4* bl = pbuf + slot * UW_PUNITS + h * UW_PIX + (wave * 4) * UW_PW + j;
        if (stamps) { ta = __builtin_amdgcn_s_memrealtime(); ca = __builtin_amdgcn_s_memtime(); }
        uw_kstep(acc, wl, bl, st);
        if (stamps) { cStep += __builtin_amdgcn_s_memtime() - ca; tb = __builtin_amdgcn_s_memrealtime(); tStep += tb - ta; }
        if (hasNext && !next.interior) {
            // a tile of the image's outermost ring (6 % of them at 1080p): the slices blended with the interior's constants -- done again in the
            // open with clamped indices and zero padding, from the same fp32 copy, over the same bytes of the slice
            uw_interpolate_general(p, next, tmpw, st.pslice, wave, lane);
            if (stamps) { ta = __builtin_amdgcn_s_memrealtime(); tBorder += ta - tb; tb = ta; ++nBorder; }
        }
        // step s + 1's weights have landed, its patch slice is written; everybody is done with this step's slice and slot
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (stamps) { ta = __builtin_amdgcn_s_memrealtime(); tWait += ta - tb; }
        if (ks == p.ksteps - 1) {
            uw_drain();
            const UwTile d = uw_tile(p, tile);
            // the shared epilogues take a wave's rows two at a time: rows oy0 + 4 wave + 2 half + r = (oy0 + 2 wave + 2 half) + 2 wave + r
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                if (p.ps) split_epilogue_ps(p, acc[hf], d.oy0 + 2 * wave + 2 * hf, d.ox0, 0, true, wave, j, h);
                else split_epilogue<true>(p, acc[hf], pbuf + slot * UW_PUNITS, d.n, d.oy0 + 2 * wave + 2 * hf, d.ox0, 0, true, lane, wave, j, h);
            }
            if (!p.ps) __syncthreads();                                      // the fp32 epilogue transposes through the slice the next step's staging writes
            uw_zero(acc);
            if (stamps) tEpi += __builtin_amdgcn_s_memrealtime() - ta;
        }
        ks = ks1; tile = tile1;
    }
    if (stamps && tid == 0) {
        unsigned long long* o = stamps + (size_t)blockIdx.x * 8;
        o[0] = tStart; o[1] = tStep; o[2] = cStep; o[3] = tBorder; o[4] = tWait; o[5] = tEpi; o[6] = __builtin_amdgcn_s_memrealtime(); o[7] = nBorder;
    }
}

} // namespace

// Launch hook for isrConv3x3ForwardSplit: -1 if this form does not take the layer (64 -> 64 channels, quads, as both of EnhanceNet's are).
static int isr_launch_split_upsw(const SplitConvParams& p0, hipStream_t s, hipEvent_t e0, hipEvent_t e1)
{
    if (p0.Cin <= 0 || (p0.Cin & 15) || p0.coutPad != 64 || p0.Cout != 64 || p0.cgroups != 1 || p0.xps) return -1;
    if (!p0.ps && ((p0.W | p0.yPlane | p0.rPlane) & 3)) return -1;        // the fp32 epilogue is compiled for quads only
    SplitConvParams p = p0;
    p.tilesY = (p.H + UW_TH - 1) / UW_TH;
    const long long tiles = (long long)p.N * p.tilesX * p.tilesY;
    if (tiles > 0x7fffffffLL) return -1;
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        (void)hipFuncSetAttribute((const void*)conv3x3_split_upsw_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, UW_LDS_BYTES);
    }
    const unsigned grid = (unsigned)(tiles < cus ? tiles : cus);
    if (e0 || e1) hipExtLaunchKernelGGL(conv3x3_split_upsw_kernel, dim3(grid), dim3(S_THREADS), UW_LDS_BYTES, s, e0, e1, 0, p);
    else hipLaunchKernelGGL(conv3x3_split_upsw_kernel, dim3(grid), dim3(S_THREADS), UW_LDS_BYTES, s, p);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
