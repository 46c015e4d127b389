// Fused 3x3 convolution (+bias +activation +residual, optional fused x2 bilinear upsample of the
// input) for MI355X / gfx950, fp32 in, fp32 accumulate, on the f32 MFMA pipe.
//
// Replaces the nn.Conv2d / nn.ReLU / residual add / nn.Upsample chain of the reference's
// EnhanceNet (SuperresolutionNetwork/models/enhancenet.py:92-125,136-144), which the reference
// runs as separate cuDNN / ATen calls.
//
// Formulation: implicit GEMM without im2col.  D[cout][pixel] = sum_k W[cout][k] * P[k][pixel],
// k = (tap, cin).  M = output channels (A operand = weights), N = 32 consecutive pixels of one
// image row (B operand = the input patch), so that each accumulator register is a 128-byte
// contiguous run of the NCHW output.  v_mfma_f32_32x32x2_f32 is an exact k-ordered fp32 fmaf chain
// (cdna_hip_programming.md section 3), which is what the 1e-4 parity bar needs.
//
// Workgroup = 4 waves = one 16x32 output tile for all output channels; wave w owns rows 4w..4w+3,
// i.e. MT x 4 accumulator tiles of 32x32.  The haloed 18x34 input patch is staged through LDS in
// chunks of 16 input channels (double buffered, register-staged so the x2 bilinear upsample can be
// fused into the loader); the 9 taps are 9 shifted ds_read_b32 views of the same patch (lanes
// 0-31 read 32 consecutive floats: conflict free).  Weights are read straight from L1/L2 in a
// [tap][cin][cout] layout (256 B per k-step and M tile; fp32 MFMA is so slow -- 64 cycles per
// instruction -- that this is ~4 B/clk/CU).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/isr_sr_kernels.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TH = 16;            // tile rows
constexpr int TW = 32;            // tile cols (= MFMA N)
constexpr int PH = TH + 2;        // patch rows
constexpr int PW = TW + 2;        // patch cols
constexpr int CK = 16;            // input channels per LDS chunk
constexpr int PLANE = PH * PW;    // 612 floats per channel plane
constexpr int CHUNK = CK * PLANE; // 9792 floats per chunk
constexpr int NTHREADS = 256;
constexpr int STAGE_REGS = (CHUNK + NTHREADS - 1) / NTHREADS;   // 39

struct ConvParams {
    const float* x;
    const float* w;          // [9][cinPad][coutPad]
    const float* bias;
    const float* residual;
    float* y;
    int N, Cin, H, W, Cout;  // H, W: output size
    int Hin, Win;            // input size (H/2, W/2 when upsampling)
    int cinPad, coutPad;     // padded channel counts of the weight layout
    int co0;                 // first output channel handled by this launch (multiple of 64)
    int tilesX, tilesY;
    int act;
    float slope;
};

__device__ __forceinline__ int xcd_remap(int bid, int nwg)
{
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// One patch element (hi-res coordinates gy, gx; channel ch) of the conv input, fetched through a
// buffer descriptor of image n: an out-of-range byte offset makes the hardware return 0, which is
// exactly the conv's zero padding (and the channel padding), so there is no branch and no select
// between the load and the ds_write that parks the value in LDS.
template <bool UPS>
__device__ __forceinline__ float load_input(const ConvParams& p, __amdgpu_buffer_rsrc_t rsrc, int ch, int gy, int gx)
{
    const bool ok = ch < p.Cin && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
    const unsigned bad = 0xFFFFFFF0u;
    if (!UPS) {
        const unsigned off = ok ? (unsigned)(((ch * p.Hin + gy) * p.Win + gx) * 4) : bad;
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, off, 0, 0));
    } else {
        // bilinear x2, align_corners=False: src = (dst + .5) * .5 - .5 clamped at 0 (ATen upsample_bilinear2d)
        float sy = ((float)gy + 0.5f) * 0.5f - 0.5f; sy = sy < 0.f ? 0.f : sy;
        float sx = ((float)gx + 0.5f) * 0.5f - 0.5f; sx = sx < 0.f ? 0.f : sx;
        const int y0 = (int)sy, x0 = (int)sx;
        const int y1 = y0 + (y0 < p.Hin - 1 ? 1 : 0), x1 = x0 + (x0 < p.Win - 1 ? 1 : 0);
        const float ly = sy - (float)y0, lx = sx - (float)x0;
        const float hy = 1.0f - ly, hx = 1.0f - lx;
        const int rowbase = ch * p.Hin;
        const unsigned o00 = ok ? (unsigned)(((rowbase + y0) * p.Win + x0) * 4) : bad;
        const unsigned o01 = ok ? (unsigned)(((rowbase + y0) * p.Win + x1) * 4) : bad;
        const unsigned o10 = ok ? (unsigned)(((rowbase + y1) * p.Win + x0) * 4) : bad;
        const unsigned o11 = ok ? (unsigned)(((rowbase + y1) * p.Win + x1) * 4) : bad;
        const float v00 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, o00, 0, 0));
        const float v01 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, o01, 0, 0));
        const float v10 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, o10, 0, 0));
        const float v11 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, o11, 0, 0));
        return hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11);
    }
}

constexpr int STAGE_PER_TAP = (STAGE_REGS + 8) / 9;             // 5 patch elements per thread per tap

template <int MT, bool UPS>
__global__ __launch_bounds__(NTHREADS, 1) void conv3x3_fwd_kernel(const ConvParams p)
{
    constexpr int CP = MT * 32;                 // padded couts
    constexpr int WCHUNK = 9 * CK * CP;         // weight floats per chunk
    constexpr int WSLICE4 = CK * CP / 4;        // float4 per tap slice (256 or 128)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* patch0 = smem;                       // [2][CHUNK]
    float* wlds0 = smem + 2 * CHUNK;            // [2][WCHUNK]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int tilesPerImage = p.tilesX * p.tilesY;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int n = bid / tilesPerImage;
    const int t = bid - n * tilesPerImage;
    const int ty = t / p.tilesX, tx = t - ty * p.tilesX;
    const int oy0 = ty * TH, ox0 = tx * TW;

    f32x16 acc[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][r][i] = 0.0f;

    const int nchunks = p.cinPad / CK;

    // image n of the input as a buffer resource (per image < 4 GiB)
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x + (size_t)n * p.Cin * p.Hin * p.Win), 0, p.Cin * p.Hin * p.Win * 4, 0x00020000);
    // patch element (c, r, col) of chunk `chunk`
    auto patch_value = [&](int chunk, int c, int r, int col) -> float {
        return load_input<UPS>(p, rsrc, c < CK ? chunk * CK + c : p.Cin, oy0 + r - 1, ox0 + col - 1);
    };
    // element index e -> (c, r, col); advancing e by NTHREADS = 7 rows + 18 cols
    static_assert(NTHREADS == 7 * PW + 18, "stride decomposition");
    auto advance = [&](int& c, int& r, int& col) {
        col += 18; r += 7;                      // branch-free carries (selects, not jumps)
        const int wc = col >= PW ? 1 : 0;
        col -= wc * PW; r += wc;
        const int wr = r >= PH ? 1 : 0;
        r -= wr * PH; c += wr;
    };
    const int c_first = tid / PLANE, r_first = (tid - c_first * PLANE) / PW, col_first = tid - c_first * PLANE - r_first * PW;
    // float4 `q` of the [CK][CP] weight slice of (chunk, tap): row = input channel, CP couts from co0
    auto weight_slice = [&](int chunk, int tap, int q) -> float4 {
        const int row = q / (CP / 4), c4 = q - row * (CP / 4);
        return *reinterpret_cast<const float4*>(p.w + ((size_t)tap * p.cinPad + chunk * CK + row) * p.coutPad + p.co0 + c4 * 4);
    };

    // prologue: chunk 0 in full
    {   // batches of 13 loads in flight, then 13 LDS writes (a load->write loop would serialise
        // 39 memory round trips per thread in front of the first MFMA)
        constexpr int PB = 13;
        static_assert(STAGE_REGS == 3 * PB, "prologue batching");
        int c = c_first, r = r_first, col = col_first;
#pragma unroll 1
        for (int b = 0; b < 3; ++b) {
            float v[PB];
#pragma unroll
            for (int i = 0; i < PB; ++i) {
                v[i] = patch_value(0, c, r, col);
                advance(c, r, col);
            }
#pragma unroll
            for (int i = 0; i < PB; ++i) {
                const int e = tid + (b * PB + i) * NTHREADS;
                if (e < CHUNK) patch0[e] = v[i];
            }
        }
    }
    for (int tap = 0; tap < 9; ++tap)
        if (tid < WSLICE4) reinterpret_cast<float4*>(wlds0 + tap * CK * CP)[tid] = weight_slice(0, tap, tid);
    __syncthreads();

    const int j = lane & 31;       // pixel column inside the tile / cout inside the M tile
    const int kh = lane >> 5;      // which of the 2 k values of the 32x32x2 MFMA this lane feeds

    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int buf = chunk & 1;
        const bool more = chunk + 1 < nchunks;
        const float* pb = patch0 + buf * CHUNK + kh * PLANE + (wave * 4) * PW + j;
        const float* wb = wlds0 + buf * WCHUNK + kh * CP + j;
        float* pnext = patch0 + (buf ^ 1) * CHUNK;
        float* wnext = wlds0 + (buf ^ 1) * WCHUNK;
        int sc = c_first, sr = r_first, scol = col_first;      // staging cursor of this thread
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            // 1/9 of the next chunk is fetched now and parked in LDS after this tap's MFMAs; the
            // other buffer is idle (every wave passed the barrier that ended its last use).
            float sv[STAGE_PER_TAP];
            float4 wv;
            if (more) {
#pragma unroll
                for (int i = 0; i < STAGE_PER_TAP; ++i) {
                    sv[i] = patch_value(chunk + 1, sc, sr, scol);
                    advance(sc, sr, scol);
                }
                wv = weight_slice(chunk + 1, tap, min(tid, WSLICE4 - 1));
            }
            const int dy = tap / 3, dx = tap - dy * 3;
            const float* pt = pb + dy * PW + dx;
            const float* wt = wb + tap * CK * CP;
            // Pin the memory order only (ALU may interleave with the MFMAs): the staging loads stay
            // in front of the MFMA block, their ds_writes (and so their vmcnt wait) behind it.
            __builtin_amdgcn_sched_barrier(0x78F);   // everything but VMEM may cross
#pragma unroll
            for (int kk = 0; kk < CK / 2; ++kk) {
                float a[MT], b[4];
#pragma unroll
                for (int m = 0; m < MT; ++m) a[m] = wt[(2 * kk) * CP + m * 32];
#pragma unroll
                for (int r = 0; r < 4; ++r) b[r] = pt[(2 * kk) * PLANE + r * PW];
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        acc[m][r] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], b[r], acc[m][r], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0x57F);   // everything but DS writes may cross
            if (more) {
#pragma unroll
                for (int i = 0; i < STAGE_PER_TAP; ++i) {
                    const int e = tid + (tap * STAGE_PER_TAP + i) * NTHREADS;
                    if (e < CHUNK) pnext[e] = sv[i];
                }
                if (tid < WSLICE4) reinterpret_cast<float4*>(wnext + tap * CK * CP)[tid] = wv;
            }
        }
        __syncthreads();
    }

    // epilogue: D row (cout) = (reg&3) + 8*(reg>>2) + 4*(lane>>5), D col (pixel) = lane&31.
    // Bias and residual are fetched in batches from clamped addresses (no branch per element, one
    // wait per batch); only the stores are predicated.
    const int ox = ox0 + j;
    const int oxc = min(ox, p.W - 1);
    float bv[MT][16];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int i = 0; i < 16; ++i)
            bv[m][i] = p.bias[min(p.co0 + m * 32 + (i & 3) + 8 * (i >> 2) + 4 * kh, p.Cout - 1)];
    const size_t plane = (size_t)p.H * p.W;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int oy = oy0 + wave * 4 + r;
        const int oyc = min(oy, p.H - 1);
        const bool pix_ok = ox < p.W && oy < p.H;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            float rv[16];
            size_t idx[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int co = min(p.co0 + m * 32 + (i & 3) + 8 * (i >> 2) + 4 * kh, p.Cout - 1);
                idx[i] = ((size_t)n * p.Cout + co) * plane + (size_t)oyc * p.W + oxc;
            }
            if (p.residual) {
#pragma unroll
                for (int i = 0; i < 16; ++i) rv[i] = p.residual[idx[i]];
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) rv[i] = 0.0f;
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int co = p.co0 + m * 32 + (i & 3) + 8 * (i >> 2) + 4 * kh;
                float v = acc[m][r][i] + bv[m][i];
                if (p.act == ISR_ACT_RELU) v = v > 0.f ? v : 0.f;
                else if (p.act == ISR_ACT_LEAKY) v = v > 0.f ? v : v * p.slope;
                v += rv[i];
                if (pix_ok && co < p.Cout) p.y[idx[i]] = v;
            }
        }
    }
}

template <int MT>
constexpr size_t conv_fwd_lds_bytes() { return (size_t)(2 * CHUNK + 2 * 9 * CK * MT * 32) * sizeof(float); }

// ---- weight re-layout ------------------------------------------------------------------------
__global__ void prepare_weights_kernel(const float* __restrict__ w, float* __restrict__ wp,
                                       int Cout, int Cin, int cinPad, int coutPad, int transpose_flip)
{
    // wp[tap][ci'][co'] ; forward: ci'=ci, co'=co ; data-grad: ci'=co, co'=ci, taps flipped
    const int total = 9 * cinPad * coutPad;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int co_ = e % coutPad;
        const int ci_ = (e / coutPad) % cinPad;
        const int tap = e / (coutPad * cinPad);
        const int ky = tap / 3, kx = tap % 3;
        float v = 0.0f;
        if (!transpose_flip) {
            if (ci_ < Cin && co_ < Cout) v = w[((co_ * Cin + ci_) * 3 + ky) * 3 + kx];
        } else {
            if (ci_ < Cout && co_ < Cin) v = w[((ci_ * Cin + co_) * 3 + (2 - ky)) * 3 + (2 - kx)];
        }
        wp[e] = v;
    }
}

__global__ void act_backward_kernel(const float* __restrict__ gy, const float* __restrict__ y, float* __restrict__ gz,
                                    long long count, int act, float slope)
{
    const long long stride = (long long)gridDim.x * blockDim.x * 4;
    for (long long e = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; e < count; e += stride) {
        if (e + 3 < count) {
            const float4 g = *reinterpret_cast<const float4*>(gy + e);
            const float4 o = *reinterpret_cast<const float4*>(y + e);
            float4 r;
            const float s = act == ISR_ACT_RELU ? 0.f : slope;
            r.x = o.x > 0.f ? g.x : g.x * s; r.y = o.y > 0.f ? g.y : g.y * s;
            r.z = o.z > 0.f ? g.z : g.z * s; r.w = o.w > 0.f ? g.w : g.w * s;
            if (act == ISR_ACT_NONE) r = g;
            *reinterpret_cast<float4*>(gz + e) = r;
        } else {
            for (long long k = e; k < count; ++k) {
                const float s = act == ISR_ACT_RELU ? 0.f : slope;
                gz[k] = (act == ISR_ACT_NONE || y[k] > 0.f) ? gy[k] : gy[k] * s;
            }
        }
    }
}


// ---- weight gradient -------------------------------------------------------------------------
// dW[co][ci][tap] = sum_pixels gz[co][p] * x[ci][p + tap]:  M = co (A = gz), N = ci (B = x patch),
// K = pixels.  A workgroup walks pixel tiles of 4 rows x 32 cols (grid-stride), keeps its 64x64x9
// partial sums in registers (wave (m,n) owns the 32x32 block (m,n) for all 9 taps = 144 VGPRs) and
// writes one slab; a second kernel reduces the slabs in a fixed order (bitwise reproducible, no
// float atomics).  LDS planes are padded to odd strides: lanes 0-31 index the channel.
constexpr int WG_TH = 4, WG_TW = 32;
constexpr int WG_PX = WG_TH * WG_TW;          // 128 pixels per tile
constexpr int GZ_STRIDE = WG_PX + 1;          // 129
constexpr int XP_W = WG_TW + 2;               // 34
constexpr int XP_PLANE = (WG_TH + 2) * XP_W + 1;   // 205

struct WGradParams {
    const float* x;     // [N][Cin][H][W]
    const float* gz;    // [N][Cout][H][W]
    float* slabs;       // [G][9][64][64]
    int N, Cin, H, W, Cout;
    int ci0, co0;       // channel group handled by this launch
    int tilesX, tilesY, ntiles;
};

__global__ __launch_bounds__(NTHREADS, 1) void conv3x3_wgrad_kernel(const WGradParams p)
{
    __shared__ float gzs[64 * GZ_STRIDE];
    __shared__ float xps[64 * XP_PLANE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = wave >> 1, nn = wave & 1;
    const int j = lane & 31, kh = lane >> 5;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;

    const int tilesPerImage = p.tilesX * p.tilesY;
    for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
        const int n = tile / tilesPerImage;
        const int t2 = tile - n * tilesPerImage;
        const int ty = t2 / p.tilesX, tx = t2 - ty * p.tilesX;
        const int oy0 = ty * WG_TH, ox0 = tx * WG_TW;
        __syncthreads();   // previous tile fully consumed
        for (int e = tid; e < 64 * WG_PX; e += NTHREADS) {
            const int c = e / WG_PX, px = e - c * WG_PX;
            const int ry = px / WG_TW, rx = px - ry * WG_TW;
            const int co = p.co0 + c, gy = oy0 + ry, gx = ox0 + rx;
            float v = 0.0f;
            if (co < p.Cout && gy < p.H && gx < p.W) v = p.gz[(((size_t)n * p.Cout + co) * p.H + gy) * p.W + gx];
            gzs[c * GZ_STRIDE + px] = v;
        }
        for (int e = tid; e < 64 * (WG_TH + 2) * XP_W; e += NTHREADS) {
            const int c = e / ((WG_TH + 2) * XP_W), rem = e - c * ((WG_TH + 2) * XP_W);
            const int r = rem / XP_W, col = rem - r * XP_W;
            const int ci = p.ci0 + c, gy = oy0 + r - 1, gx = ox0 + col - 1;
            float v = 0.0f;
            if (ci < p.Cin && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W)
                v = p.x[(((size_t)n * p.Cin + ci) * p.H + gy) * p.W + gx];
            xps[c * XP_PLANE + r * XP_W + col] = v;
        }
        __syncthreads();
        const float* ga = &gzs[(m * 32 + j) * GZ_STRIDE + kh];
        const float* xb = &xps[(nn * 32 + j) * XP_PLANE + kh];
#pragma unroll
        for (int ry = 0; ry < WG_TH; ++ry) {
#pragma unroll 4
            for (int rx = 0; rx < WG_TW; rx += 2) {
                const float a = ga[ry * WG_TW + rx];
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int dy = tap / 3, dx = tap - dy * 3;
                    const float b = xb[(ry + dy) * XP_W + rx + dx];
                    acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[tap], 0, 0, 0);
                }
            }
        }
    }
    // slab[g][tap][co(64)][ci(64)]
    float* slab = p.slabs + (size_t)blockIdx.x * 9 * 64 * 64;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int co = m * 32 + (i & 3) + 8 * (i >> 2) + 4 * kh;
            slab[((size_t)tap * 64 + co) * 64 + nn * 32 + j] = acc[tap][i];
        }
}

__global__ void wgrad_reduce_kernel(const float* __restrict__ slabs, int G, float* __restrict__ dw,
                                    int Cout, int Cin, int co0, int ci0)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;     // over [9][64][64]
    if (e >= 9 * 64 * 64) return;
    const int ci = e & 63, co = (e >> 6) & 63, tap = e >> 12;
    if (co0 + co >= Cout || ci0 + ci >= Cin) return;
    float s = 0.0f;
    for (int g = 0; g < G; ++g) s += slabs[(size_t)g * 9 * 64 * 64 + e];
    dw[((size_t)(co0 + co) * Cin + (ci0 + ci)) * 9 + tap] = s;
}

// db[c] = sum over n,y,x of gz[n][c][y][x]; one workgroup per channel, fixed reduction order.
__global__ __launch_bounds__(256) void bias_grad_kernel(const float* __restrict__ gz, float* __restrict__ db,
                                                         int N, int C, long long HW)
{
    __shared__ float red[256];
    const int c = blockIdx.x;
    float s = 0.0f;
    for (int n = 0; n < N; ++n) {
        const float* src = gz + ((size_t)n * C + c) * HW;
        for (long long i = threadIdx.x; i < HW; i += 256) s += src[i];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) db[c] = red[0];
}

constexpr int WGRAD_MAX_SLABS = 512;

}  // namespace

extern "C" {

int isrConvCinPad(int Cin) { return ((Cin + CK - 1) / CK) * CK; }
int isrConvCoutPad(int Cout) { return ((Cout + 31) / 32) * 32; }

int isrConvPrepareWeights(const float* w, float* wprep, int Cout, int Cin, int transpose_flip, void* stream)
{
    if (!w || !wprep || Cout <= 0 || Cin <= 0) return -1;
    const int cinPad = transpose_flip ? isrConvCinPad(Cout) : isrConvCinPad(Cin);
    const int coutPad = transpose_flip ? isrConvCoutPad(Cin) : isrConvCoutPad(Cout);
    const int total = 9 * cinPad * coutPad;
    const int blocks = (total + 255) / 256;
    hipLaunchKernelGGL(prepare_weights_kernel, dim3(blocks > 1024 ? 1024 : blocks), dim3(256), 0, (hipStream_t)stream,
                       w, wprep, Cout, Cin, cinPad, coutPad, transpose_flip);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int isrConv3x3Forward(const float* x, const float* wprep, const float* bias, const float* residual, float* y,
                      int N, int Cin, int H, int W, int Cout, int act, float slope, int upsample2x, void* stream)
{
    if (!x || !wprep || !y || N <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return -1;
    if (upsample2x && ((H & 1) || (W & 1))) return -1;
    if (act < ISR_ACT_NONE || act > ISR_ACT_LEAKY) return -1;
    ConvParams p;
    p.x = x; p.w = wprep; p.bias = bias; p.residual = residual; p.y = y;
    p.N = N; p.Cin = Cin; p.H = H; p.W = W; p.Cout = Cout;
    p.Hin = upsample2x ? H / 2 : H; p.Win = upsample2x ? W / 2 : W;
    p.cinPad = isrConvCinPad(Cin); p.coutPad = isrConvCoutPad(Cout);
    p.tilesX = (W + TW - 1) / TW; p.tilesY = (H + TH - 1) / TH;
    p.act = act; p.slope = slope;
    const long long nwg = (long long)N * p.tilesX * p.tilesY;
    if (nwg > 0x7fffffffLL) return -1;
    const dim3 grid((unsigned)nwg), block(NTHREADS);
    hipStream_t s = (hipStream_t)stream;
    static float* zero_bias = nullptr;   // bias == NULL -> a device buffer of zeros (keeps the epilogue branch-free)
    if (!bias) {
        if (!zero_bias) {
            if (hipMalloc(&zero_bias, 4096 * sizeof(float)) != hipSuccess) return -2;
            if (hipMemset(zero_bias, 0, 4096 * sizeof(float)) != hipSuccess) return -2;
        }
        if (Cout > 4096) return -1;
        p.bias = zero_bias;
    }
    static bool attr_done = false;
    if (!attr_done) {   // > 64 KiB of dynamic LDS needs an explicit opt-in
        (void)hipFuncSetAttribute((const void*)conv3x3_fwd_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)conv_fwd_lds_bytes<1>());
        (void)hipFuncSetAttribute((const void*)conv3x3_fwd_kernel<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)conv_fwd_lds_bytes<1>());
        (void)hipFuncSetAttribute((const void*)conv3x3_fwd_kernel<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)conv_fwd_lds_bytes<2>());
        (void)hipFuncSetAttribute((const void*)conv3x3_fwd_kernel<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)conv_fwd_lds_bytes<2>());
        attr_done = true;
    }
    // one launch per group of up to 64 output channels (2 M tiles per wave)
    for (int co0 = 0; co0 < p.coutPad; co0 += 64) {
        p.co0 = co0;
        if (p.coutPad - co0 == 32) {
            if (upsample2x) hipLaunchKernelGGL((conv3x3_fwd_kernel<1, true>), grid, block, conv_fwd_lds_bytes<1>(), s, p);
            else hipLaunchKernelGGL((conv3x3_fwd_kernel<1, false>), grid, block, conv_fwd_lds_bytes<1>(), s, p);
        } else {
            if (upsample2x) hipLaunchKernelGGL((conv3x3_fwd_kernel<2, true>), grid, block, conv_fwd_lds_bytes<2>(), s, p);
            else hipLaunchKernelGGL((conv3x3_fwd_kernel<2, false>), grid, block, conv_fwd_lds_bytes<2>(), s, p);
        }
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int isrActBackward(const float* gy, const float* y, float* gz, long long count, int act, float slope, void* stream)
{
    if (!gy || !y || !gz || count < 0) return -1;
    if (count == 0) return 0;
    long long blocks = (count / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(act_backward_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, gy, y, gz, count, act, slope);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

long long isrConvWeightGradWorkspace(int N, int Cin, int H, int W, int Cout)
{
    (void)N; (void)Cin; (void)H; (void)W; (void)Cout;
    return (long long)WGRAD_MAX_SLABS * 9 * 64 * 64 * sizeof(float);
}

int isrConv3x3WeightGrad(const float* x, const float* gz, float* dw, float* db, void* workspace,
                         int N, int Cin, int H, int W, int Cout, void* stream)
{
    if (!x || !gz || !dw || !workspace || N <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return -1;
    hipStream_t s = (hipStream_t)stream;
    WGradParams p;
    p.x = x; p.gz = gz; p.slabs = (float*)workspace;
    p.N = N; p.Cin = Cin; p.H = H; p.W = W; p.Cout = Cout;
    p.tilesX = (W + WG_TW - 1) / WG_TW; p.tilesY = (H + WG_TH - 1) / WG_TH;
    const long long nt = (long long)N * p.tilesX * p.tilesY;
    if (nt > 0x7fffffffLL) return -1;
    p.ntiles = (int)nt;
    const int G = p.ntiles < WGRAD_MAX_SLABS ? p.ntiles : WGRAD_MAX_SLABS;
    for (int co0 = 0; co0 < Cout; co0 += 64)
        for (int ci0 = 0; ci0 < Cin; ci0 += 64) {
            p.co0 = co0; p.ci0 = ci0;
            hipLaunchKernelGGL(conv3x3_wgrad_kernel, dim3(G), dim3(NTHREADS), 0, s, p);
            hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((9 * 64 * 64 + 255) / 256), dim3(256), 0, s,
                               p.slabs, G, dw, Cout, Cin, co0, ci0);
        }
    if (db) hipLaunchKernelGGL(bias_grad_kernel, dim3(Cout), dim3(256), 0, s, gz, db, N, Cout, (long long)H * W);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // extern "C"
