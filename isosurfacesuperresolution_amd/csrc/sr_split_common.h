// Shared device code of the split-operand convolution kernels (sr_conv_split.hip, sr_conv_tail.hip): operand types, the
// (hi, lo) split, the parameter block, one k-step of MFMAs and the common epilogue.  Every function is inline / static: each
// translation unit gets its own copy, nothing is called across translation units (no relocatable device code).
#pragma once
#include "sr_diag.h"
#include "sr_act.h"
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "../../include/isr_sr_kernels.h"
#include "sr_finish.h"
#include "sr_profile.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

constexpr unsigned BAD_OFFSET = 0x80000000u;
constexpr int ST_H = 8, ST_W = 32;
constexpr int SP_H = ST_H + 2, SP_W = ST_W + 2, SP_PIX = SP_H * SP_W;       // 340 patch pixels
constexpr int S_CHUNK = 32;                                                  // input channels per staging pass
constexpr int S_GROUPS = S_CHUNK / 8;                                        // 8-channel groups per pass
constexpr int S_PART = S_GROUPS * SP_PIX;                                    // 16-byte units of the hi (or lo) patch: 1360
constexpr int S_PUNITS = 2 * S_PART;                                         // hi then lo
constexpr int S_THREADS = 256;
constexpr int S_WPART = 9 * 2 * 64;                                          // weights of one k-step, one part: [tap][lane half][64 couts]
constexpr int S_WUNITS = 2 * S_WPART;                                        // hi then lo: 2304 units = 9 per thread
constexpr int S_LDS_BYTES = (S_PUNITS + S_WUNITS) * 16;                      // 43520 + 36864 = 80384: two workgroups per CU

struct SplitConvParams {
    const float* x; const u32x4* wq; const float* bias; const float* residual; float* y;
    int N, Cin, H, W, Cout;
    int xPlane, yPlane, rPlane;
    long long xImage, yImage, rImage;
    int ksteps;          // ceil(Cin / 16)
    int coutPad;         // Cout rounded up to 32
    int cgroups;         // 64-channel output groups covered by the grid
    int tilesX, tilesY;
    int act; float slope;
    int Hin, Win;                 // input size: (H, W), or (H / 2, W / 2) for the upsampling variant
    int quads;                    // 1: W, the plane stride of x and its base address allow aligned dwordx4 staging
    ISR_DIAG_MEMBER(int, dbg, 0);                      // diagnostics: 1 skip the MFMAs, 2 skip the staging loads, 4 skip the stores
    ISR_DIAG_MEMBER(unsigned long long*, stamps, nullptr);   // diagnostics: per-workgroup s_memrealtime stamps (100 MHz, one clock for the whole chip), or NULL
    // PACKED-SPLIT output (ps != NULL; y unused): the activations as the NEXT split-operand layer's LDS image, i.e. already
    // split into (hi, lo') fp16 pairs, eight channels of one pixel per 16-byte unit: ps[part: hi | lo][Cout / 8 groups][psPlane
    // units, pixel y W + x].  The consumer stages k-steps with plain 16-byte copies (LDS-DMA); same values as converting the
    // fp32 tensor on the way in, so results do not change.  Single image, Cout a multiple of 8, no residual.
    u32x4* ps; int psPlane;
    // PACKED-SPLIT input (xps != NULL; x unused): the same layout, Cin / 8 groups, xpsPlane units per plane; staged by LDS-DMA,
    // units outside the image (zero padding) and beyond the last group from `zero` (16 zero bytes).  Plain layers, one image.
    const u32x4* xps; int xpsPlane;
    const u32x4* zero;
    // RANGE GUARD (may be NULL): atomic maximum, as the bit pattern of |v|, over every value this launch stores.  The split of an
    // activation overflows fp16 at |x| >= 65520; the host reads the producers' maxima now and then and routes the CONSUMER of a
    // tensor that came close (>= 3e4) to the exact fp32 kernels (ops.RANGE_GUARD) -- no cliff, no host read inside a frame.
    unsigned* absmax;
    // per-WAVE maxima (may be NULL; kernels that end in split_epilogue): word 4 blockIdx.x + wave of a zeroed array receives the bit
    // pattern of the largest |value| the wave stores (an atomic maximum on an address nobody else touches).  Training: the weight-gradient
    // kernels scale their gz operand by the tensor's maximum (isrConv3x3WeightGradSegmentsSplitMax) without a pass over it.
    unsigned* slotmax = nullptr;     // (default: parameter blocks filled field by field elsewhere -- sr_conv_block.hip, sr_conv_tail.hip -- leave it off)
};

// bit pattern of |v|: unsigned order = order of the magnitudes, inf above every finite value, NaN above inf (never lost)
__device__ __forceinline__ unsigned isr_mag(float v) { return __builtin_bit_cast(unsigned, v) & 0x7fffffffu; }
__device__ __forceinline__ unsigned isr_umax(unsigned a, unsigned b) { return a > b ? a : b; }
// At most one atomic per wave: lane maxima combined by butterfly shuffles, and the atomic only if the wave would raise the flag.
// (32 000 waves of a 1080p launch hammering one address are a queue the last workgroups of the launch wait behind; the flag
// reaches its final value within the first few waves, everyone after that reads it -- a cached line -- and moves on.  A stale
// read can only be too LOW, i.e. cost a superfluous atomic, never lose a maximum.)
__device__ __forceinline__ void isr_range_note(unsigned* flag, unsigned m)
{
    if (!flag) return;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = isr_umax(m, (unsigned)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0 && m > __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(flag, m);
}

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) const void isr_gvoid_t;
typedef __attribute__((address_space(3))) void isr_lvoid_t;

// LDS-DMA (global_load_lds_dwordx4): 64 lanes x 16 bytes, lane l's bytes land at dst_wave_base + 16 l (dst is wave uniform,
// src per lane).  The data is ordered for a ds_read by the next __syncthreads() (hipcc waits vmcnt(0) in front of it).
__device__ __forceinline__ void isr_dma16(const u32x4* src, u32x4* dst_wave_base)
{
    __builtin_amdgcn_global_load_lds((isr_gvoid_t*)src, (isr_lvoid_t*)dst_wave_base, 16, 0, 0);
}

// v = hi + lo (+ <= 2^-22 |v|): hi = RN16(v), lo = RN16(v - hi); the subtraction is exact in fp32
__device__ __forceinline__ void split16(float v, _Float16& hi, _Float16& lo)
{
    hi = (_Float16)v;
    lo = (_Float16)(v - (float)hi);
}
// activations: the low part scaled by 2^11 (exact), see the header comment
__device__ __forceinline__ void split16x(float v, _Float16& hi, _Float16& lo)
{
    hi = (_Float16)v;
    lo = (_Float16)((v - (float)hi) * 2048.0f);
}

// h a + l b with the roundings spelled out -- round(l b), then one fused multiply-add -- so that every kernel that interpolates (the
// two upsampling kernels must agree bit for bit) gets the same bits whatever the optimiser would have contracted on its own
__device__ __forceinline__ float isr_blend(float h, float a, float l, float b) { return __builtin_fmaf(h, a, l * b); }

__device__ __forceinline__ float buf_load(rsrc_t r, unsigned voff)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, 0, 0));
}

__device__ __forceinline__ f32x16 mfma16(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

// Epilogue shared by the split kernels: acc * 2^-S + bias, activation, residual / gate, store.  D row (cout) =
// (reg & 3) + 8 * (reg >> 2) + 4 * h, column (pixel) = j.  The wide path transposes through the (idle) patch buffer.
// Packed-split epilogue: act(acc 2^-S + bias) of this wave's two rows as (hi, lo') units straight from the D layout -- lane (j, h)
// holds channels 8 g + 4 h .. + 3 of pixel j for the four groups g of a 32-channel block: 8 bytes of the unit, the lane pair
// (j, 0), (j, 1) writes the 16, a wave instruction 512 contiguous bytes.  No LDS transposition.
template <int ACT>
__device__ __forceinline__ void split_epilogue_ps_act(const SplitConvParams& p, f32x16 (&acc)[2][2], int oy0, int ox0, int co0, bool second,
                                                      int wave, int j, int h)
{
    const float unscale = reinterpret_cast<const float*>(p.wq)[1];
    const int groups = p.Cout >> 3;
    const rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(p.ps, 0, (int)((size_t)2 * groups * p.psPlane * 16), 0x00020000);
    const int ox = ox0 + j;
    float bv[2][16];                                                         // all bias values first: one latency, not 64
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int i = 0; i < 16; ++i)
            bv[cb][i] = p.bias ? p.bias[min(co0 + cb * 32 + 8 * (i >> 2) + 4 * h + (i & 3), p.Cout - 1)] : 0.0f;
    unsigned mag = 0u;
    // The lane pair (j, 0) / (j, 1) holds the two 8-byte halves of the hi unit and of the lo' unit of a pixel's channel group; they trade
    // halves (v_permlane32_swap: the upper 32 lanes of one register against the lower 32 of another), after which lane (j, 0) holds the
    // whole hi unit and lane (j, 1) the whole lo' unit: ONE 16-byte store per lane instead of two 8-byte ones -- the epilogue is
    // store-ISSUE bound (tools/lab/bench_ups4.py)
    const unsigned lopart = (unsigned)h * (unsigned)(groups * p.psPlane) * 16u;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int oy = oy0 + wave * 2 + r;
        const unsigned voff = (oy < p.H && ox < p.W) ? (unsigned)(oy * p.W + ox) * 16u + lopart : BAD_OFFSET;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            if (cb == 1 && !second) break;
#pragma unroll
            for (int gi = 0; gi < 4; ++gi) {
                f16x4 th, tl;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = isr_activate<ACT>(acc[cb][r][4 * gi + e] * unscale + bv[cb][4 * gi + e], p.slope);
                    _Float16 a, b;
                    split16x(v, a, b);
                    th[e] = a; tl[e] = b;
                    mag = isr_umax(mag, isr_mag(v));
                }
                const int g = (co0 >> 3) + cb * 4 + gi;
                const bool live = g < groups;
                const u32x2 uh = __builtin_bit_cast(u32x2, th), ul = __builtin_bit_cast(u32x2, tl);
                const u32x2 s0 = __builtin_amdgcn_permlane32_swap(uh.x, ul.x, false, false);
                const u32x2 s1 = __builtin_amdgcn_permlane32_swap(uh.y, ul.y, false, false);
                const u32x4 unit = {s0.x, s1.x, s0.y, s1.y};                // h = 0: channels 8 g .. + 7 hi; h = 1: the same channels' lo'
                // (the plane offset travels in the VECTOR offset, soffset = 0: behind a 16-byte store with an SGPR soffset the compiler pads
                //  nothing, and a vector instruction that writes one of its data registers in the next issue slot replaces lanes 12-15 of
                //  every 16 of that dword in memory -- tools/probes/store_valu_overwrite_probe.hip; with soffset 0 it pads two wait states)
                __builtin_amdgcn_raw_buffer_store_b128(unit, prs, (int)((live && !(p.dbg & 16)) ? voff + (unsigned)(g * p.psPlane * 16) : BAD_OFFSET), 0, 0);
            }
        }
    }
    isr_range_note(p.absmax, mag);
}

__device__ __forceinline__ void split_epilogue_ps(const SplitConvParams& p, f32x16 (&acc)[2][2], int oy0, int ox0, int co0, bool second,
                                                  int wave, int j, int h)
{
    if (p.act == ISR_ACT_RELU) split_epilogue_ps_act<ISR_ACT_RELU>(p, acc, oy0, ox0, co0, second, wave, j, h);
    else if (p.act == ISR_ACT_LEAKY) split_epilogue_ps_act<ISR_ACT_LEAKY>(p, acc, oy0, ox0, co0, second, wave, j, h);
    else split_epilogue_ps_act<ISR_ACT_NONE>(p, acc, oy0, ox0, co0, second, wave, j, h);
}

// WIDE_ONLY: the caller guarantees W and both plane strides are multiples of 4 (the per-element path is not compiled in).
// AUX: cache policy of the output stores (0 plain; 16 = sc1, write-through to memory: the dataflow kernels' hand-off, see
// sr_conv_trunk.hip).
// (ACT: the activation as a compile-time constant, see isr_activate; split_epilogue below switches once)
template <int ACT, bool WIDE_ONLY, int AUX, bool RES_AHEAD>
__device__ __forceinline__ void split_epilogue_act(const SplitConvParams& p, f32x16 (&acc)[2][2], u32x4* patch, int n, int oy0, int ox0, int co0,
                                               bool second, int lane, int wave, int j, int h)
{
    const float unscale = reinterpret_cast<const float*>(p.wq)[1];          // 2^-S (header of the prepared weights)
    const rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(p.y + (size_t)n * p.yImage, 0, (int)((size_t)p.Cout * p.yPlane * 4), 0x00020000);
    const rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.residual ? p.residual + (size_t)n * p.rImage : p.y), 0,
                                                         p.residual ? (int)((size_t)p.Cout * p.rPlane * 4) : 0, 0x00020000);
    const int ox = ox0 + j;
    unsigned mag = 0u;
    float bv[2][16];                                                         // all bias values first: one latency, not 128
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int i = 0; i < 16; ++i)
            bv[cb][i] = p.bias ? p.bias[min(co0 + cb * 32 + (i & 3) + 8 * (i >> 2) + 4 * h, p.Cout - 1)] : 0.0f;
    if (p.dbg & 4) {
    } else if (WIDE_ONLY || ((p.W | p.yPlane | p.rPlane) & 3) == 0) {
        // wide path: each wave transposes one output row (64 couts x 32 pixels) through 8 KB of the now idle patch, so
        // that a lane owns 4 consecutive pixels of one channel and the stores are dwordx4 (4x fewer instructions)
        float* tr = reinterpret_cast<float*>(patch) + wave * (64 * 32);
        // the residual / gate operand of both rows up front: behind the first store the compiler may not move a load any more (it
        // cannot know that y and the residual do not overlap), and sixteen load -> use -> store round trips in a row cost a gated
        // data gradient a third of its launch
        // (RES_AHEAD: 64 registers -- only where the register budget has them: the one-workgroup-per-tile plain kernel)
        u32x4 rq[2][8];
        if (RES_AHEAD && p.residual) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const int q = lane + 64 * t, oy = oy0 + wave * 2 + r;
                    const int co = co0 + (q >> 3), px = ox0 + (q & 7) * 4;
                    const bool ok = oy < p.H && px < p.W && co < p.Cout;
                    rq[r][t] = __builtin_amdgcn_raw_buffer_load_b128(rrs, (int)(ok ? (unsigned)(oy * p.W + px) * 4u + (unsigned)co * (unsigned)p.rPlane * 4u : BAD_OFFSET), 0, 0);
                }
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int oy = oy0 + wave * 2 + r;
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float v = isr_activate<ACT>(acc[cb][r][i] * unscale + bv[cb][i], p.slope);
                    tr[(cb * 32 + (i & 3) + 8 * (i >> 2) + 4 * h) * 32 + j] = v;
                }
            __builtin_amdgcn_s_waitcnt(0xC07F);                              // lgkmcnt(0): same-wave hand-off through LDS
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int q = lane + 64 * t;                                 // float4 index: cout = q / 8, pixel group = q % 8
                const int co = co0 + (q >> 3), px = ox0 + (q & 7) * 4;
                const bool ok = oy < p.H && px < p.W && co < p.Cout;
                float4 v = reinterpret_cast<const float4*>(tr)[q];
                const unsigned pixoff = (unsigned)(oy * p.W + px) * 4u;
                if (p.residual) {
                    if (!RES_AHEAD) rq[r][t] = __builtin_amdgcn_raw_buffer_load_b128(rrs, (int)(ok ? pixoff + (unsigned)co * (unsigned)p.rPlane * 4u : BAD_OFFSET), 0, 0);
                    const float4 rf = __builtin_bit_cast(float4, rq[r][t]);
                    if (ACT == ISR_ACT_GATE) {
                        v.x = rf.x > 0.f ? v.x : 0.f; v.y = rf.y > 0.f ? v.y : 0.f;
                        v.z = rf.z > 0.f ? v.z : 0.f; v.w = rf.w > 0.f ? v.w : 0.f;
                    } else {
                        v.x += rf.x; v.y += rf.y; v.z += rf.z; v.w += rf.w;
                    }
                }
                if (ok) mag = isr_umax(isr_umax(mag, isr_umax(isr_mag(v.x), isr_mag(v.y))), isr_umax(isr_mag(v.z), isr_mag(v.w)));
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), yrs,
                                                       (int)(ok ? pixoff + (unsigned)co * (unsigned)p.yPlane * 4u : BAD_OFFSET), 0, AUX);
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);                              // reads done before the next row overwrites the slab
        }
    } else {
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        if (cb == 1 && !second) break;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int oy = oy0 + wave * 2 + r;
            const unsigned pix = (oy < p.H && ox < p.W) ? (unsigned)(oy * p.W + ox) * 4u : BAD_OFFSET;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int co = co0 + cb * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                float v = isr_activate<ACT>(acc[cb][r][i] * unscale + bv[cb][i], p.slope);
                const bool ok = pix != BAD_OFFSET && co < p.Cout;
                if (p.residual) {
                    const float rv = buf_load(rrs, ok ? pix + (unsigned)co * (unsigned)p.rPlane * 4u : BAD_OFFSET);
                    if (ACT == ISR_ACT_GATE) v = rv > 0.f ? v : 0.f; else v += rv;
                }
                if (ok) mag = isr_umax(mag, isr_mag(v));
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), yrs,
                                                      ok ? (int)(pix + (unsigned)co * (unsigned)p.yPlane * 4u) : (int)BAD_OFFSET, 0, 0);
            }
        }
    }
    }
    if (p.slotmax) {
        unsigned m = mag;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = isr_umax(m, (unsigned)__shfl_xor((int)m, o, 64));
        if (lane == 0) atomicMax(p.slotmax + blockIdx.x * 4 + wave, m);
    }
    isr_range_note(p.absmax, mag);
}

template <bool WIDE_ONLY = false, int AUX = 0, bool RES_AHEAD = false>
__device__ __forceinline__ void split_epilogue(const SplitConvParams& p, f32x16 (&acc)[2][2], u32x4* patch, int n, int oy0, int ox0, int co0,
                                               bool second, int lane, int wave, int j, int h)
{
    if (p.act == ISR_ACT_RELU) split_epilogue_act<ISR_ACT_RELU, WIDE_ONLY, AUX, RES_AHEAD>(p, acc, patch, n, oy0, ox0, co0, second, lane, wave, j, h);
    else if (p.act == ISR_ACT_LEAKY) split_epilogue_act<ISR_ACT_LEAKY, WIDE_ONLY, AUX, RES_AHEAD>(p, acc, patch, n, oy0, ox0, co0, second, lane, wave, j, h);
    else if (p.act == ISR_ACT_GATE) split_epilogue_act<ISR_ACT_GATE, WIDE_ONLY, AUX, RES_AHEAD>(p, acc, patch, n, oy0, ox0, co0, second, lane, wave, j, h);
    else split_epilogue_act<ISR_ACT_NONE, WIDE_ONLY, AUX, RES_AHEAD>(p, acc, patch, n, oy0, ox0, co0, second, lane, wave, j, h);
}

// One k-step of MFMAs: 16 input channels x 9 taps x (2 channel blocks x 2 rows) x 3 products.  wl: this lane's weight
// units of the k-step in LDS (hi; lo at + S_WPART), bl: this lane's patch units (hi; lo at + S_PART).
__device__ __forceinline__ void split_kstep(f32x16 (&acc)[2][2], const u32x4* wl, const u32x4* bl, bool second)
{
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int dy = tap / 3, dx = tap - dy * 3;
        const f16x8 a0h = __builtin_bit_cast(f16x8, wl[tap * 128]);
        const f16x8 a0l = __builtin_bit_cast(f16x8, wl[S_WPART + tap * 128]);
        const f16x8 a1h = __builtin_bit_cast(f16x8, wl[tap * 128 + (second ? 32 : 0)]);
        const f16x8 a1l = __builtin_bit_cast(f16x8, wl[S_WPART + tap * 128 + (second ? 32 : 0)]);
        const f16x8 a0s = a0h * (_Float16)0.00048828125f;                   // w_hi 2^-11: partner of the scaled x_lo'
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
    }
}

} // namespace
