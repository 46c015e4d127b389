// HALF-PRECISION FAST MODE of the fused 3x3 convolution (inference only; SURVEY.md 8(d): "a bf16 fast mode may be
// reported separately, judged by PSNR only").  NOT the parity path: the 1e-4 comparison with the reference's CPU path
// is defined on the exact fp32 kernels of sr_conv3x3.hip, which stay the default everywhere.
//
// Same operator, same tensors (fp32 NCHW activations in HBM, fp32 bias / residual / output), but the operands of the
// matrix instruction are rounded to fp16 on their way into LDS and multiplied by v_mfma_f32_32x32x16_f16 with fp32
// accumulation: 16 input channels per instruction at 32 cycles instead of 2 at 64.  fp16 rather than bf16: same MFMA
// rate, three more mantissa bits (the sequence PSNR against the fp32 frames was 24.6 dB with bf16 operands), and the
// range is no issue for O(1) activations (conversion saturates).  The arithmetic of a 64 -> 64 layer at 1080p drops
// from 1.2 ms to ~80 us and the layer becomes a streaming kernel bound by its 1.06 GB of activations (roofline: HBM).
//   * workgroup = 8 x 32 output pixels x 64 output channels, 4 waves x (2 rows x 2 channel blocks) = 4 accumulators;
//   * the 64-channel input patch (10 x 34 pixels) is staged once per 64-channel chunk as
//     LDS[channel group of 8][pixel][8 x fp16]: a lane's B fragment (8 consecutive channels of one pixel) is one
//     conflict-free ds_read_b128, a tap shift is +16 bytes; staging = 8 loads (one per channel plane: aligned dwordx4
//     covering 4 pixels when the row length allows, dwords otherwise; out-of-image / out-of-range channels return 0
//     through the buffer descriptor) -> packed converts -> ds_write_b128;
//   * weights are re-laid once as [tap][k-step][lane half][cout][8 x fp16] (74 KB for 64 -> 64, L2 resident) and
//     pass through LDS one 16-channel k-step at a time, double buffered, fetched under the previous k-step's MFMAs
//     (read straight from L2 per MFMA they cost ~1000 cycles each: the loop ran at a quarter of its MFMA time);
//   * two workgroups per CU (80 KB of LDS each) overlap one's staging with the other's MFMAs.
#include "sr_diag.h"
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "../../include/isr_sr_kernels.h"
#include "sr_finish.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

constexpr unsigned BAD_OFFSET = 0x80000000u;
constexpr int BT_H = 8, BT_W = 32;
constexpr int BP_H = BT_H + 2, BP_W = BT_W + 2, BP_PIX = BP_H * BP_W;       // 612 patch pixels
constexpr int B_CHUNK = 64;                                                  // input channels per staging pass
constexpr int B_GROUPS = B_CHUNK / 8;                                        // 8-channel groups per pass
constexpr int B_UNITS = B_GROUPS * BP_PIX;                                   // 16-byte LDS units per pass
constexpr int B_THREADS = 256;
constexpr int B_WUNITS = 9 * 2 * 64;                                         // weights of one k-step: [tap][lane half][64 couts] x 16 bytes
constexpr int B_LDS_BYTES = (B_UNITS + 2 * B_WUNITS) * 16;                   // 43520 + 36864 = 80384: two workgroups per CU

struct F16ConvParams {
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
    ISR_DIAG_MEMBER(unsigned long long*, stamps, nullptr);   // diagnostics (tools/lab/bench_conv_f16.py): per-workgroup s_memtime stamps, or NULL
};

// round to fp16, saturating (activations of this network are O(1) -- depth, normals, ReLU features of unit-gain
// layers -- but an overflow to infinity would turn into NaNs downstream)
__device__ __forceinline__ _Float16 to_half(float v) { return (_Float16)fminf(fmaxf(v, -65504.0f), 65504.0f); }

// operand type of the matrix instruction: fp16 (inference fast mode: three more mantissa bits) or bf16 (the
// mixed-precision training mode: gradients span the fp32 exponent range, fp16 would flush the small ones to zero)
template <bool BF> struct Operand;
template <> struct Operand<false> {
    typedef f16x8 V;
    static __device__ __forceinline__ _Float16 cvt(float v) { return to_half(v); }
    static __device__ __forceinline__ f32x16 mfma(V a, V b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};
template <> struct Operand<true> {
    typedef bf16x8 V;
    static __device__ __forceinline__ __bf16 cvt(float v) { return (__bf16)v; }
    static __device__ __forceinline__ f32x16 mfma(V a, V b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};

__device__ __forceinline__ float buf_load(rsrc_t r, unsigned voff)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, 0, 0));
}

// UPS: the convolution reads U(x), the x2 bilinear upsampling (align_corners=False) of x [Cin][H/2][W/2]: the low-res
// region under the tile's patch (6 x 18 pixels x 64 channels, fp32) is staged into the LDS that will hold the weights,
// and the fp16 patch is built from it with the four-tap blend of isrUpsample2xForward -- the upsampled tensor (530 MB
// at 1080p, written once and read 1.3 times) never exists.
template <bool UPS, bool BF>
__global__ __launch_bounds__(B_THREADS, 2) void conv3x3_f16_kernel(const F16ConvParams p)
{
    using O = Operand<BF>;
    using opx8 = typename O::V;
    extern __shared__ u32x4 patch[];                                         // B_UNITS patch units, then two weight buffers
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
    const int oy0 = ty * BT_H, ox0 = tx * BT_W, co0 = cg * 64;

    const rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x + (size_t)n * p.xImage), 0,
                                                         (int)((size_t)p.Cin * p.xPlane * 4), 0x00020000);
    const unsigned planeBytes = (unsigned)p.xPlane * 4u;

    u32x4* wbuf = patch + B_UNITS;
    f32x16 acc[2][2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[cb][r][i] = 0.0f;

    const bool second = co0 + 32 < p.coutPad;                                // the second 32-channel block exists
    unsigned long long st0 = 0, st1 = 0, st2 = 0;
    if (p.stamps) st0 = __builtin_amdgcn_s_memtime();
    for (int cin0 = 0; cin0 < p.Cin; cin0 += B_CHUNK) {
        // ---- stage the 64-channel patch: unit u = (channel group g, patch pixel) ----------------------------------
        // four units (32 dword loads) in flight per thread: with 8 the kernel sat at 2.1 TB/s, the bandwidth that
        // 16 KB in flight per CU buys at ~2 us of memory latency
        // weights of a k-step: 1152 units [tap][lane half][64 couts], 4.5 per thread, L2 -> registers -> LDS
        const int ks0 = cin0 >> 4;
        const int nks = min(4, p.ksteps - ks0);
        const int couts = min(64, p.coutPad - co0);
        u32x4 wreg[4][5];
        auto wfetch = [&](auto stag) {
            constexpr int S = decltype(stag)::value;
            if (S >= nks) return;
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int q = tid + i * B_THREADS;
                const int tap = q >> 7, hh = (q >> 6) & 1, c = q & 63;
                if (q < B_WUNITS && c < couts) wreg[S][i] = p.wq[(size_t)((tap * p.ksteps + ks0 + S) * 2 + hh) * p.coutPad + co0 + c];
            }
        };
        auto wpark = [&](auto stag) {
            constexpr int S = decltype(stag)::value;
            if constexpr (S < 4) {
                if (S >= nks) return;
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    const int q = tid + i * B_THREADS;
                    if (q < B_WUNITS && (q & 63) < couts) wbuf[(S & 1) * B_WUNITS + q] = wreg[S][i];
                }
            }
        };
        using K0 = std::integral_constant<int, 0>; using K1 = std::integral_constant<int, 1>;
        using K2 = std::integral_constant<int, 2>; using K3 = std::integral_constant<int, 3>;
        wfetch(K0{}); wfetch(K1{});                                      // in flight under the staging
        if (UPS) {
            constexpr int LR_H = BT_H / 2 + 2, LR_W = BT_W / 2 + 2;         // 6 x 18 low-res pixels: rows oy0/2 - 1 .., cols ox0/2 - 1 ..
            constexpr int LQ = (BT_W / 2 + 8) / 4;                           // 6 aligned quads per row: columns ox0/2 - 4 .. ox0/2 + 19
            constexpr int LUNITS = B_CHUNK * LR_H * LQ;                      // (channel, row, quad) = 2304
            float* tmp = reinterpret_cast<float*>(wbuf);                     // [64][6][18] fp32 = 27.6 KB of the 36.9 KB weight area
            const int ly0 = oy0 / 2 - 1, lx0 = ox0 / 2 - 1;
            for (int u0 = tid; u0 < LUNITS; u0 += 3 * B_THREADS) {
                u32x4 v[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int u = u0 + k * B_THREADS;
                    const int c = u / (LR_H * LQ), rem = u - c * (LR_H * LQ);
                    const int r = rem / LQ, q = rem - r * LQ;
                    const int iy = ly0 + r, ix = ox0 / 2 - 4 + 4 * q;
                    const bool ok = u < LUNITS && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
                    v[k] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)(ok ? (unsigned)(cin0 + c) * planeBytes + (unsigned)(iy * p.Win + ix) * 4u
                                                                               : BAD_OFFSET), 0, 0);
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int u = u0 + k * B_THREADS;
                    if (u >= LUNITS) continue;
                    const int c = u / (LR_H * LQ), rem = u - c * (LR_H * LQ);
                    const int r = rem / LQ, q = rem - r * LQ;
                    const float4 f = __builtin_bit_cast(float4, v[k]);
                    float* dst = tmp + (c * LR_H + r) * LR_W + 4 * q - 3;      // quad q holds low-res patch columns 4q - 3 .. 4q
                    if (q > 0) dst[0] = f.x;
                    if (q > 0 && q < LQ - 1) { dst[1] = f.y; dst[2] = f.z; }
                    if (q < LQ - 1) dst[3] = f.w;
                }
            }
            __syncthreads();
            for (int u = tid; u < B_UNITS; u += B_THREADS) {
                const int g = u / BP_PIX, pix = u - g * BP_PIX;
                const int r = pix / BP_W, c = pix - r * BP_W;
                const int Y = oy0 + r - 1, X = ox0 + c - 1;
                opx8 o;
                if ((unsigned)Y < (unsigned)p.H && (unsigned)X < (unsigned)p.W) {
                    int y0, y1, x0, x1; float ly, lx;
                    isr_src_index(Y, 0.5f, p.Hin, y0, y1, ly);
                    isr_src_index(X, 0.5f, p.Win, x0, x1, lx);
                    const float hy = 1.f - ly, hx = 1.f - lx;
                    const float* t0 = tmp + (g * 8) * (LR_H * LR_W) + (y0 - ly0) * LR_W - lx0;
                    const float* t1 = tmp + (g * 8) * (LR_H * LR_W) + (y1 - ly0) * LR_W - lx0;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float* a = t0 + e * (LR_H * LR_W);
                        const float* b = t1 + e * (LR_H * LR_W);
                        o[e] = O::cvt(hy * (hx * a[x0] + lx * a[x1]) + ly * (hx * b[x0] + lx * b[x1]));
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = O::cvt(0.0f);
                }
                patch[u] = __builtin_bit_cast(u32x4, o);
            }
            __syncthreads();                                                 // tmp is free: the weights may land on it
        } else if (p.dbg & 2) {
            __syncthreads();
        } else if (p.quads) {
            // rows of 4-pixel groups aligned to 16 bytes (W, plane stride and tile origin are multiples of 4): one
            // dwordx4 per channel covers 4 pixels -- a quarter of the load instructions (the texture path takes 16
            // cycles per wave instruction whatever its width) and 4x the bytes in flight.  Unit = (channel group g,
            // patch row r, quad q); quad q holds columns ox0 - 4 + 4q .. +3, i.e. patch columns 4q - 3 .. 4q.
            constexpr int QPR = (BT_W + 8) / 4;                              // 10 quads per patch row
            constexpr int QUNITS = B_GROUPS * BP_H * QPR;                    // 800
            constexpr int QB = 2;                                            // units (8 dwordx4 loads each) in flight per thread; 3 or 4 spill (256-register budget)
            for (int u0 = tid; u0 < QUNITS; u0 += QB * B_THREADS) {
                u32x4 v[QB][8];
#pragma unroll
                for (int k = 0; k < QB; ++k) {
                    const int u = u0 + k * B_THREADS;
                    const int g = u / (BP_H * QPR), rem = u - g * (BP_H * QPR);
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
                    const int u = u0 + k * B_THREADS;
                    if (u >= QUNITS) continue;
                    const int g = u / (BP_H * QPR), rem = u - g * (BP_H * QPR);
                    const int r = rem / QPR, q = rem - r * QPR;
                    opx8 o0, o1, o2, o3;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float4 f = __builtin_bit_cast(float4, v[k][e]);
                        o0[e] = O::cvt(f.x); o1[e] = O::cvt(f.y); o2[e] = O::cvt(f.z); o3[e] = O::cvt(f.w);
                    }
                    u32x4* dst = patch + g * BP_PIX + r * BP_W + 4 * q - 3;
                    // quad 0 contributes only its last pixel (patch column 0), quad 9 only its first (column 33)
                    if (q > 0) dst[0] = __builtin_bit_cast(u32x4, o0);
                    if (q > 0 && q < QPR - 1) { dst[1] = __builtin_bit_cast(u32x4, o1); dst[2] = __builtin_bit_cast(u32x4, o2); }
                    if (q < QPR - 1) dst[3] = __builtin_bit_cast(u32x4, o3);
                }
            }
        } else
        for (int u0 = tid; u0 < B_UNITS; u0 += 4 * B_THREADS) {
            float v[4][8];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int u = u0 + k * B_THREADS;
                const int g = u / BP_PIX, pix = u - g * BP_PIX;
                const int r = pix / BP_W, c = pix - r * BP_W;
                const int iy = oy0 + r - 1, ix = ox0 + c - 1;
                const bool ok = u < B_UNITS && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                const unsigned base = (unsigned)(cin0 + g * 8) * planeBytes + (unsigned)(iy * p.W + ix) * 4u;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[k][e] = buf_load(xrs, ok ? base + (unsigned)e * planeBytes : BAD_OFFSET);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int u = u0 + k * B_THREADS;
                opx8 q;
#pragma unroll
                for (int e = 0; e < 8; ++e) q[e] = O::cvt(v[k][e]);
                if (u < B_UNITS) patch[u] = __builtin_bit_cast(u32x4, q);
            }
        }
        wpark(K0{});
        wfetch(K2{}); wfetch(K3{});                                      // two k-steps of MFMAs to arrive
        __syncthreads();
        if (p.stamps) st1 = __builtin_amdgcn_s_memtime();
        // ---- MFMAs: k-steps of 16 channels x 9 taps x (2 channel blocks x 2 rows) ---------------------------------
        auto kstep = [&](auto stag) {
            constexpr int S = decltype(stag)::value;
            if (S >= nks) return;
            // the next k-step's weights go to the other buffer first (its readers passed the last barrier)
            wpark(std::integral_constant<int, S + 1>{});
            const u32x4* wl = wbuf + (S & 1) * B_WUNITS + h * 64 + j;
            const u32x4* bl = patch + (2 * S + h) * BP_PIX + (wave * 2) * BP_W + j;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int dy = tap / 3, dx = tap - dy * 3;
                const opx8 a0 = __builtin_bit_cast(opx8, wl[tap * 128]);
                const opx8 a1 = __builtin_bit_cast(opx8, wl[tap * 128 + (second ? 32 : 0)]);
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const opx8 b = __builtin_bit_cast(opx8, bl[(r + dy) * BP_W + dx]);
                    acc[0][r] = O::mfma(a0, b, acc[0][r]);
                    if (second) acc[1][r] = O::mfma(a1, b, acc[1][r]);
                }
            }
            __syncthreads();
        };
        if (!(p.dbg & 1)) { kstep(K0{}); kstep(K1{}); kstep(K2{}); kstep(K3{}); }
    }

    if (p.stamps) st2 = __builtin_amdgcn_s_memtime();
    // ---- epilogue: D row (cout) = (reg & 3) + 8 * (reg >> 2) + 4 * h, column (pixel) = j ----------------------------
    const rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(p.y + (size_t)n * p.yImage, 0, (int)((size_t)p.Cout * p.yPlane * 4), 0x00020000);
    const rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.residual ? p.residual + (size_t)n * p.rImage : p.y), 0,
                                                         p.residual ? (int)((size_t)p.Cout * p.rPlane * 4) : 0, 0x00020000);
    const int ox = ox0 + j;
    float bv[2][16];                                                         // all bias values first: one latency, not 128
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int i = 0; i < 16; ++i)
            bv[cb][i] = p.bias ? p.bias[min(co0 + cb * 32 + (i & 3) + 8 * (i >> 2) + 4 * h, p.Cout - 1)] : 0.0f;
    if (p.dbg & 4) {
    } else if (((p.W | p.yPlane | p.rPlane) & 3) == 0) {
        // wide path: each wave transposes one output row (64 couts x 32 pixels) through 8 KB of the now idle patch, so
        // that a lane owns 4 consecutive pixels of one channel and the stores are dwordx4 (4x fewer instructions)
        float* tr = reinterpret_cast<float*>(patch) + wave * (64 * 32);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int oy = oy0 + wave * 2 + r;
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    float v = acc[cb][r][i] + bv[cb][i];
                    if (p.act == ISR_ACT_RELU) v = v > 0.f ? v : 0.f;
                    else if (p.act == ISR_ACT_LEAKY) v = v > 0.f ? v : v * p.slope;
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
                    const u32x4 rr = __builtin_amdgcn_raw_buffer_load_b128(rrs, (int)(ok ? pixoff + (unsigned)co * (unsigned)p.rPlane * 4u : BAD_OFFSET), 0, 0);
                    const float4 rf = __builtin_bit_cast(float4, rr);
                    if (p.act == ISR_ACT_GATE) {
                        v.x = rf.x > 0.f ? v.x : 0.f; v.y = rf.y > 0.f ? v.y : 0.f;
                        v.z = rf.z > 0.f ? v.z : 0.f; v.w = rf.w > 0.f ? v.w : 0.f;
                    } else {
                        v.x += rf.x; v.y += rf.y; v.z += rf.z; v.w += rf.w;
                    }
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), yrs,
                                                       (int)(ok ? pixoff + (unsigned)co * (unsigned)p.yPlane * 4u : BAD_OFFSET), 0, 0);
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
                float v = acc[cb][r][i] + bv[cb][i];
                if (p.act == ISR_ACT_RELU) v = v > 0.f ? v : 0.f;
                else if (p.act == ISR_ACT_LEAKY) v = v > 0.f ? v : v * p.slope;
                const bool ok = pix != BAD_OFFSET && co < p.Cout;
                if (p.residual) {
                    const float rv = buf_load(rrs, ok ? pix + (unsigned)co * (unsigned)p.rPlane * 4u : BAD_OFFSET);
                    if (p.act == ISR_ACT_GATE) v = rv > 0.f ? v : 0.f; else v += rv;
                }
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), yrs,
                                                      ok ? (int)(pix + (unsigned)co * (unsigned)p.yPlane * 4u) : (int)BAD_OFFSET, 0, 0);
            }
        }
    }
    }
    if (p.stamps && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long st3 = __builtin_amdgcn_s_memtime();
        unsigned long long* o = p.stamps + (size_t)blockIdx.x * 4;
        o[0] = st0; o[1] = st1; o[2] = st2; o[3] = st3;
    }
}

// w[Cout][Cin][3][3] fp32 -> wq[tap][k-step][lane half h][coutPad][8 x fp16]; element e of (k-step s, half h) is input
// channel 16 s + 8 h + e (zero beyond Cin / Cout)
template <bool BF>
__global__ void prepare_weights_f16_kernel(const float* __restrict__ w, u32x4* __restrict__ wq, int Cout, int Cin, int ksteps, int coutPad)
{
    using O = Operand<BF>;
    const int total = 9 * ksteps * 2 * coutPad;
    for (int u = blockIdx.x * blockDim.x + threadIdx.x; u < total; u += gridDim.x * blockDim.x) {
        const int co = u % coutPad;
        const int hh = (u / coutPad) & 1;
        const int s = (u / (coutPad * 2)) % ksteps;
        const int tap = u / (coutPad * 2 * ksteps);
        typename O::V q;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ci = 16 * s + 8 * hh + e;
            q[e] = O::cvt((co < Cout && ci < Cin) ? w[((size_t)co * Cin + ci) * 9 + tap] : 0.0f);
        }
        wq[u] = __builtin_bit_cast(u32x4, q);
    }
}

} // namespace

[[maybe_unused]] static unsigned long long* g_f16_stamps = nullptr;
[[maybe_unused]] static int g_f16_dbg = 0;

extern "C" {

#ifdef ISR_DIAG
void isrDebugSetF16StampBuffer(unsigned long long* buf) { g_f16_stamps = buf; }   // not part of the public header
void isrDebugSetF16Ablation(int bits) { g_f16_dbg = bits; }
#endif

long long isrConvF16WeightBytes(int Cin, int Cout)
{
    if (Cin <= 0 || Cout <= 0) return -1;
    return (long long)9 * ((Cin + 15) / 16) * 2 * (((Cout + 31) / 32) * 32) * 16;
}

static int prepare_lp(const float* w, void* wq, int Cout, int Cin, void* stream, bool bf)
{
    if (!w || !wq || Cout <= 0 || Cin <= 0) return -1;
    const int ksteps = (Cin + 15) / 16, coutPad = ((Cout + 31) / 32) * 32;
    const int total = 9 * ksteps * 2 * coutPad;
    if (bf) hipLaunchKernelGGL(prepare_weights_f16_kernel<true>, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                               w, (u32x4*)wq, Cout, Cin, ksteps, coutPad);
    else hipLaunchKernelGGL(prepare_weights_f16_kernel<false>, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                            w, (u32x4*)wq, Cout, Cin, ksteps, coutPad);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

static int forward_lp(const float* x, const void* wq, const float* bias, const float* residual, float* y,
                      int N, int Cin, int H, int W, int Cout, int act, float slope, int upsample2x,
                      long long xPlane, long long xImage, long long yPlane, long long yImage,
                      long long rPlane, long long rImage, void* stream, bool bf)
{
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
    F16ConvParams p;
    p.x = x; p.wq = (const u32x4*)wq; p.bias = bias; p.residual = residual; p.y = y;
    p.N = N; p.Cin = Cin; p.H = H; p.W = W; p.Cout = Cout;
    p.Hin = Hin; p.Win = Win;
    p.xPlane = (int)xPlane; p.yPlane = (int)yPlane; p.rPlane = (int)(residual ? rPlane : yPlane);
    p.xImage = xImage; p.yImage = yImage; p.rImage = rImage;
    p.ksteps = (Cin + 15) / 16; p.coutPad = ((Cout + 31) / 32) * 32;
    p.cgroups = (Cout + 63) / 64;
    p.tilesX = (W + BT_W - 1) / BT_W; p.tilesY = (H + BT_H - 1) / BT_H;
    p.act = act; p.slope = slope;
    ISR_DIAG_SET(p.stamps, g_f16_stamps);
    ISR_DIAG_SET(p.dbg, g_f16_dbg);
    p.quads = ((W & 3) == 0 && aligned) ? 1 : 0;
    const long long nwg = (long long)N * p.tilesX * p.tilesY * p.cgroups;
    if (nwg > 0x7fffffffLL) return -1;
    static bool attr_done = false;
    if (!attr_done) {   // > 64 KiB of LDS needs an explicit opt-in
        (void)hipFuncSetAttribute((const void*)conv3x3_f16_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, B_LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)conv3x3_f16_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, B_LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)conv3x3_f16_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, B_LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)conv3x3_f16_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, B_LDS_BYTES);
        attr_done = true;
    }
    const dim3 grid((unsigned)nwg), block(B_THREADS);
    hipStream_t s = (hipStream_t)stream;
    if (upsample2x) {
        if (bf) hipLaunchKernelGGL((conv3x3_f16_kernel<true, true>), grid, block, B_LDS_BYTES, s, p);
        else hipLaunchKernelGGL((conv3x3_f16_kernel<true, false>), grid, block, B_LDS_BYTES, s, p);
    } else {
        if (bf) hipLaunchKernelGGL((conv3x3_f16_kernel<false, true>), grid, block, B_LDS_BYTES, s, p);
        else hipLaunchKernelGGL((conv3x3_f16_kernel<false, false>), grid, block, B_LDS_BYTES, s, p);
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int isrConvF16Prepare(const float* w, void* wq, int Cout, int Cin, void* stream) { return prepare_lp(w, wq, Cout, Cin, stream, false); }
int isrConvBf16Prepare(const float* w, void* wq, int Cout, int Cin, void* stream) { return prepare_lp(w, wq, Cout, Cin, stream, true); }

int isrConv3x3ForwardF16(const float* x, const void* wq, const float* bias, const float* residual, float* y,
                         int N, int Cin, int H, int W, int Cout, int act, float slope, int upsample2x,
                         long long xPlane, long long xImage, long long yPlane, long long yImage,
                         long long rPlane, long long rImage, void* stream)
{
    return forward_lp(x, wq, bias, residual, y, N, Cin, H, W, Cout, act, slope, upsample2x, xPlane, xImage, yPlane, yImage, rPlane, rImage, stream, false);
}

int isrConv3x3ForwardBf16(const float* x, const void* wq, const float* bias, const float* residual, float* y,
                          int N, int Cin, int H, int W, int Cout, int act, float slope, int upsample2x,
                          long long xPlane, long long xImage, long long yPlane, long long yImage,
                          long long rPlane, long long rImage, void* stream)
{
    return forward_lp(x, wq, bias, residual, y, N, Cin, H, W, Cout, act, slope, upsample2x, xPlane, xImage, yPlane, yImage, rPlane, rImage, stream, true);
}

int isrConvF16SupportsUpsample(long long x_address, int Win, long long xPlane, long long xImage)
{
    return ((Win & 3) == 0 && (xPlane & 3) == 0 && (xImage & 3) == 0 && (x_address & 15) == 0) ? 1 : 0;
}

} // extern "C"
