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
#include "sr_diag.h"
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include <type_traits>
#include <vector>

#include "../../include/isr_sr_kernels.h"
#include "sr_finish.h"
#include "sr_profile.h"
#include "sr_act.h"

// Launch with start/stop events on the dispatch packet when profiling, as a plain launch otherwise (plain launches can be
// captured into a HIP graph -- train.GraphedTrainStep -- the Ext form cannot be relied upon there).
#define ISR_LAUNCH(KERNEL, GRID, BLOCK, LDS, STREAM, E0, E1, ...)                                              \
    do {                                                                                                        \
        if ((E0) || (E1)) hipExtLaunchKernelGGL(KERNEL, GRID, BLOCK, LDS, STREAM, E0, E1, 0, __VA_ARGS__);    \
        else hipLaunchKernelGGL(KERNEL, GRID, BLOCK, LDS, STREAM, __VA_ARGS__);                                \
    } while (0)

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
    int xPlane, yPlane, rPlane;          // channel strides of x / y / residual in floats (>= rows * cols)
    long long xImage, yImage, rImage;    // batch strides in floats
    int cinPad, coutPad;     // padded channel counts of the weight layout
    int co0;                 // first output channel handled by this launch (multiple of 64)
    int cgroups;             // conv3x3_fwd2_kernel: 32-channel groups covered by the grid (co0 + 32 g)
    int tilesX, tilesY;
    int act;
    float slope;
    ISR_DIAG_MEMBER(int, dbg, 0);                 // ablation bits for tools/bench_conv.py (0 in production)
    ISR_DIAG_MEMBER(unsigned long long*, stamps, nullptr);   // dbg & 8: per-workgroup s_memtime stamps (diagnostic builds only)
};

__device__ __forceinline__ int xcd_remap(int bid, int nwg)
{
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

constexpr int STAGE_PER_TAP = (STAGE_REGS + 8) / 9;             // 5 patch elements per thread per tap
constexpr unsigned BAD_OFFSET = 0x80000000u;                    // beyond any buffer (< 2 GiB per image)

typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float buf_load(rsrc_t r, unsigned voff)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, 0, 0));
}

// Staging plan of one thread: element i (i < 45) of a chunk is patch position e = tid + 256*i,
// i.e. channel c = e / 612 of the chunk, patch row/col (r, col).  The byte offset of that element
// inside the chunk's 16 input planes does not depend on the chunk, so it is computed once per
// workgroup and kept in ONE register per element; out-of-image positions get PLAN_BAD (the
// buffer hardware then returns 0 = the conv's zero padding) and channels beyond Cin fall behind
// the descriptor's num_records.
// With the x2-upsampling loader (bilinear, align_corners=False: src = (dst+.5)/2-.5 clamped at 0)
// the word holds the offset of the top-left tap plus 4 flag bits: bits 0/1 = parity of the
// hi-res x/y (odd -> weight .25 on the +1 neighbour, even -> .75), bits 29/30 = "+1 neighbour
// exists" in x/y.  At the image borders (dst 0 and dst max) the neighbour flag is cleared, which
// makes both taps the same texel, so the border cases need no weight of their own.
constexpr unsigned PLAN_BAD = 0x80000000u;
constexpr unsigned PLAN_DX = 1u << 29, PLAN_DY = 1u << 30, PLAN_OFF = 0x1FFFFFFCu;

template <bool UPS, int CHUNK_ = CHUNK, int PLANE_ = PLANE>
__device__ __forceinline__ unsigned plan_element(const ConvParams& p, int e, int oy0, int ox0)
{
    const int c = e / PLANE_;
    const int rem = e - c * PLANE_;
    const int r = rem / PW;
    const int col = rem - r * PW;
    const int gy = oy0 + r - 1, gx = ox0 + col - 1;
    const bool ok = e < CHUNK_ && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
    if (!ok) return PLAN_BAD;
    if (!UPS) return (unsigned)((c * p.xPlane + gy * p.Win + gx) * 4);
    const int x0 = gx > 0 ? (gx - 1) >> 1 : 0, y0 = gy > 0 ? (gy - 1) >> 1 : 0;
    unsigned w = (unsigned)((c * p.xPlane + y0 * p.Win + x0) * 4) | (unsigned)(gx & 1) | ((unsigned)(gy & 1) << 1);
    if (gx > 0 && x0 < p.Win - 1) w |= PLAN_DX;
    if (gy > 0 && y0 < p.Hin - 1) w |= PLAN_DY;
    return w;
}

// phase 1: issue the loads of one element (1 or 4 dwords)
template <bool UPS>
__device__ __forceinline__ void issue_element(rsrc_t rs, unsigned w, unsigned rowBytes, float (&raw)[UPS ? 4 : 1])
{
    if (!UPS) {
        raw[0] = buf_load(rs, w);
    } else {
        const unsigned off = (w & PLAN_BAD) ? PLAN_BAD : (w & PLAN_OFF);
        const unsigned dx = (w & PLAN_DX) ? 4u : 0u, dy = (w & PLAN_DY) ? rowBytes : 0u;
        raw[0] = buf_load(rs, off); raw[1] = buf_load(rs, off + dx);
        raw[2] = buf_load(rs, off + dy); raw[3] = buf_load(rs, off + dy + dx);
    }
}

// phase 2: the value that goes to LDS
template <bool UPS>
__device__ __forceinline__ float finish_element(unsigned w, const float (&raw)[UPS ? 4 : 1])
{
    if (!UPS) return raw[0];
    const float lx = (w & 1u) ? 0.25f : 0.75f, ly = (w & 2u) ? 0.25f : 0.75f;
    const float hx = 1.0f - lx, hy = 1.0f - ly;
    return hy * (hx * raw[0] + lx * raw[1]) + ly * (hx * raw[2] + lx * raw[3]);
}

template <int MT, int R = 4>
__device__ __forceinline__ void conv_epilogue(const ConvParams& p, f32x16 (&acc)[MT][R], float* smem, int n, int oy0, int ox0,
                                              int co0, int wave, int lane);

template <int MT, bool UPS>
__global__ __launch_bounds__(NTHREADS, 1) void conv3x3_fwd_kernel(const ConvParams p)
{
    constexpr int CP = MT * 32;                 // padded couts handled by this launch
    constexpr int WCHUNK = 9 * CK * CP;         // weight floats per chunk
    constexpr int WSLICE4 = CK * CP / 4;        // float4 per tap slice (256 or 128)
    constexpr int NSTAGE = STAGE_PER_TAP * 9;   // 45 planned elements per thread (39 real)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* patch0 = smem;                       // [2][CHUNK]
    float* wlds0 = smem + 2 * CHUNK;            // [2][WCHUNK]
    float* dump = wlds0 + 2 * WCHUNK;           // [4*NTHREADS] write-only sink for masked-off staging lanes

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int tilesPerImage = p.tilesX * p.tilesY;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int n = bid / tilesPerImage;
    const int t = bid - n * tilesPerImage;
    const int ty = t / p.tilesX, tx = t - ty * p.tilesX;
    const int oy0 = ty * TH, ox0 = tx * TW;

    unsigned long long st0 = 0, st1 = 0, st2 = 0;
    if (p.dbg & 8) st0 = __builtin_amdgcn_s_memtime();
    f32x16 acc[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][r][i] = 0.0f;

    const int nchunks = p.cinPad / CK;
    const int planeIn = p.xPlane;
    const float* ximg = p.x + (size_t)n * p.xImage;
    const unsigned rowBytes = (unsigned)p.Win * 4u;
    // descriptor of the 16 input planes of `chunk` (channels >= Cin are out of range -> 0)
    auto chunk_rsrc = [&](int chunk) -> rsrc_t {
        const int left = p.Cin - chunk * CK;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ximg + (size_t)chunk * CK * planeIn), 0,
                                                 left > 0 ? left * planeIn * 4 : 0, 0x00020000);
    };

    unsigned plan[NSTAGE];
#pragma unroll
    for (int i = 0; i < NSTAGE; ++i) plan[i] = plan_element<UPS>(p, tid + i * NTHREADS, oy0, ox0);
    // weights: float4 `tid` of the [CK][CP] slice of (chunk, tap); rows are input channels
    const int wrow = min(tid, WSLICE4 - 1) / (CP / 4), wc4 = min(tid, WSLICE4 - 1) - wrow * (CP / 4);
    const float* wthread = p.w + (size_t)wrow * p.coutPad + p.co0 + wc4 * 4;
    auto weight_slice = [&](int chunk, int tap) -> float4 {
        return *reinterpret_cast<const float4*>(wthread + ((size_t)tap * p.cinPad + chunk * CK) * p.coutPad);
    };

    // prologue: chunk 0 in full, in batches of 13 loads in flight
    {
        const rsrc_t rs = chunk_rsrc(0);
        constexpr int NB = UPS ? 3 : 1;            // batches: 39 (or 3 x 13 x 4 taps) loads in flight
        constexpr int PB = STAGE_REGS / NB;
        static_assert(PB * NB == STAGE_REGS, "prologue batching");
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            float raw[PB][UPS ? 4 : 1];
#pragma unroll
            for (int i = 0; i < PB; ++i) issue_element<UPS>(rs, plan[b * PB + i], rowBytes, raw[i]);
#pragma unroll
            for (int i = 0; i < PB; ++i) {
                const int e = tid + (b * PB + i) * NTHREADS;
                if (e < CHUNK) patch0[e] = finish_element<UPS>(plan[b * PB + i], raw[i]);
            }
        }
        float4 wv[9];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) wv[tap] = weight_slice(0, tap);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
            if (tid < WSLICE4) reinterpret_cast<float4*>(wlds0 + tap * CK * CP)[tid] = wv[tap];
    }
    __syncthreads();
    if (p.dbg & 8) st1 = __builtin_amdgcn_s_memtime();

    const int j = lane & 31;       // pixel column inside the tile / cout inside the M tile
    const int kh = lane >> 5;      // which of the 2 k values of the 32x32x2 MFMA this lane feeds

    auto load_ops = [&](float (&a)[MT], float (&b)[4], const float* w_, const float* p_, int kk) {
#pragma unroll
        for (int m = 0; m < MT; ++m) a[m] = w_[(2 * kk) * CP + m * 32];
#pragma unroll
        for (int r = 0; r < 4; ++r) b[r] = p_[(2 * kk) * PLANE + r * PW];
    };
    auto mfma_step = [&](const float (&a)[MT], const float (&b)[4]) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                acc[m][r] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], b[r], acc[m][r], 0, 0, 0);
    };

    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int buf = chunk & 1;
        const bool more = chunk + 1 < nchunks && !(p.dbg & 1);
        const float* pb = patch0 + buf * CHUNK + kh * PLANE + (wave * 4) * PW + j;
        const float* wb = wlds0 + buf * WCHUNK + kh * CP + j;
        float* pnext = patch0 + (buf ^ 1) * CHUNK + tid;
        float* wnext = wlds0 + (buf ^ 1) * WCHUNK;
        const rsrc_t rsn = chunk_rsrc(chunk + 1);
        // Operand registers are double buffered by hand: the LDS reads of k-step kk+1 are issued
        // before the 8 MFMAs of k-step kk (512 cycles of cover); the taps are fully unrolled so
        // that every LDS address is base + immediate and the staging plan is indexed statically.
        //
        // Staging of the next chunk rides inside the MFMA regions, software pipelined over taps:
        // k-step i (< 6) of tap t issues the loads of element 6t+i (taps 0..6 cover the 39
        // elements) and, first, parks the element issued TWO taps earlier (128 MFMAs, ~3.4 us of
        // cover for an HBM miss) in the idle LDS buffer.  The weight slices go two per tap in
        // k-steps 6 and 7 of taps 0..4 with the same distance.  So every park happens inside the
        // tap loop, and the address / lerp VALU work sits in the shadow of the MFMAs.
        constexpr int EPT = 6;                      // elements issued per tap
        static_assert(EPT * 7 >= STAGE_REGS, "taps 0..6 must cover the chunk");
        float a0[MT], b0[4], a1[MT], b1[4];
        float raw[2][EPT][UPS ? 4 : 1];
        float4 wraw[2][2];
        load_ops(a0, b0, wb, pb, 0);
        // The plan words are loop invariant; without the opaque copy hipcc hoists the decoded
        // offsets / weights of all 39 elements out of the chunk loop (+150 live registers, spills).
        auto plan_word = [&](int q) -> unsigned {
            unsigned w = plan[q];
            asm volatile("" : "+v"(w));
            return w;
        };
        // No branch around any staging access (a branch makes hipcc's waitcnt pass fall back to
        // vmcnt(0) in front of every ds_write): lanes without an element write to the `dump` area.
        float* const plast = (tid < CHUNK - (STAGE_REGS - 1) * NTHREADS) ? pnext + (STAGE_REGS - 1) * NTHREADS : dump + tid;
        auto park = [&](int q, const float (&rw)[UPS ? 4 : 1]) {           // element q (static) -> LDS
            const float v = finish_element<UPS>(plan_word(q), rw);
            if (q < STAGE_REGS - 1) pnext[q * NTHREADS] = v;
            else if (q == STAGE_REGS - 1) *plast = v;
        };
        float4* const wdst = (WSLICE4 >= NTHREADS || tid < WSLICE4) ? reinterpret_cast<float4*>(wnext) + tid
                                                                    : reinterpret_cast<float4*>(dump) + tid;
        const int wstep = (WSLICE4 >= NTHREADS || tid < WSLICE4) ? CK * CP / 4 : 0;   // float4 per tap slice
        auto stage_work = [&](int tap, int kk) {  // called inside the MFMA region of k-step kk
            if (kk < EPT) {
                if (tap >= 2 && (tap - 2) * EPT + kk < STAGE_REGS) park((tap - 2) * EPT + kk, raw[tap & 1][kk]);
                const int q = tap * EPT + kk;
                if (tap <= 6 && q < STAGE_REGS) issue_element<UPS>(rsn, plan_word(q), rowBytes, raw[tap & 1][kk]);
            } else {
                const int h = kk - EPT;                                   // 0 or 1
                const int sp = (tap - 2) * 2 + h, si = tap * 2 + h;       // weight slices parked / issued
                if (tap >= 2 && sp < 9) wdst[sp * wstep] = wraw[tap & 1][h];
                if (si < 9) wraw[tap & 1][h] = weight_slice(chunk + 1, si);
            }
        };
        // `more` is hoisted out of the tap loop (two copies of the loop) for the same reason.
        auto run_taps = [&](auto MORE) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap - dy * 3;
            const int tn = tap + 1, dyn = tn / 3, dxn = tn - dyn * 3;
            const float* pt = pb + dy * PW + dx;
            const float* wt = wb + tap * CK * CP;
            const float* ptn = pb + dyn * PW + dxn;
            const float* wtn = wb + tn * CK * CP;
            // full scheduling fences between the LDS-read groups and the MFMA groups: left alone,
            // hipcc sinks each ds_read to just in front of its consumer (fewer live registers),
            // which re-exposes the LDS latency every k-step
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < CK / 2; kk += 2) {
                load_ops(a1, b1, wt, pt, kk + 1);
                __builtin_amdgcn_sched_barrier(0);
                mfma_step(a0, b0);
                if (decltype(MORE)::value) stage_work(tap, kk);
                __builtin_amdgcn_sched_barrier(0);
                if (kk + 2 < CK / 2) load_ops(a0, b0, wt, pt, kk + 2);
                else if (tap < 8) load_ops(a0, b0, wtn, ptn, 0);
                __builtin_amdgcn_sched_barrier(0);
                mfma_step(a1, b1);
                if (decltype(MORE)::value) stage_work(tap, kk + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        };
        if (more) run_taps(std::true_type{}); else run_taps(std::false_type{});
        __syncthreads();
    }

    if (p.dbg & 8) st2 = __builtin_amdgcn_s_memtime();
    conv_epilogue<MT>(p, acc, smem, n, oy0, ox0, p.co0, wave, lane);
    if ((p.dbg & 8) && tid == 0 && p.stamps) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long st3 = __builtin_amdgcn_s_memtime();
        unsigned long long* o = p.stamps + (size_t)blockIdx.x * 4;
        o[0] = st0; o[1] = st1; o[2] = st2; o[3] = st3;
    }
}


template <int MT, int R>
__device__ __forceinline__ void conv_epilogue(const ConvParams& p, f32x16 (&acc)[MT][R], float* smem, int n, int oy0, int ox0,
                                              int co0, int wave, int lane)
{
    const int j = lane & 31, kh = lane >> 5;
    // epilogue: D row (cout) = (reg&3) + 8*(reg>>2) + 4*(lane>>5), D col (pixel) = lane&31.
    // Output / residual go through buffer descriptors of image n: per-lane byte offset of the pixel
    // (BAD_OFFSET outside the image), per-register scalar offset of the channel plane; channels
    // >= Cout fall behind num_records and are dropped by the hardware.  No per-element branches.
    const int ox = ox0 + j;
    // y and the residual may have different channel strides: the per-lane offset is built from the pixel part
    // and the channel part separately for each of them
    const int outBytes = (int)((size_t)p.Cout * p.yPlane * 4);
    const rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(p.y + (size_t)n * p.yImage, 0, outBytes, 0x00020000);
    const rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.residual ? p.residual + (size_t)n * p.rImage : p.y), 0,
        p.residual ? (int)((size_t)p.Cout * p.rPlane * 4) : 0, 0x00020000);
    float bv[MT][16];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int i = 0; i < 16; ++i)
            bv[m][i] = p.bias[min(co0 + m * 32 + (i & 3) + 8 * (i >> 2) + 4 * kh, p.Cout - 1)];
    const int planeBytes = p.yPlane * 4, rplaneBytes = p.rPlane * 4;
    isr_with_act(p.act, [&](auto A) {                                       // (one switch, not one per value: sr_act.h)
    constexpr int ACT = decltype(A)::value;
    if (((p.W | p.yPlane | p.rPlane) & 3) == 0) {
        // Wide path: each wave transposes one output row (64 couts x 32 pixels) through its own 8 KB
        // of the now idle LDS, so that a lane owns 4 consecutive pixels of one channel and the
        // global traffic is dwordx4 (4x fewer store instructions; the dword epilogue is store-issue
        // bound at ~7 B/clk/CU).  Bias/activation are applied on the way in, the residual on the way out.
        float* tr = smem + wave * (64 * 32);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int oy = oy0 + wave * R + r;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float v = isr_activate<ACT>(acc[m][r][i] + bv[m][i], p.slope);
                    tr[(m * 32 + (i & 3) + 8 * (i >> 2) + 4 * kh) * 32 + j] = v;
                }
            // same-wave hand-off: LDS operations of one wave complete in order
            __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
#pragma unroll
            for (int t = 0; t < MT * 4; ++t) {
                const int q = lane + 64 * t;                     // float4 index: cout = q/8, pixel group = q%8
                const int co = q >> 3, px = ox0 + (q & 7) * 4;
                const bool ok = oy < p.H && px < p.W && !(p.dbg & 2);
                const unsigned off = ok ? (unsigned)((oy * p.W + px) * 4) : BAD_OFFSET;
                const int soff = (co0 + co) * planeBytes;      // per lane -> goes into the vector offset
                float4 v = reinterpret_cast<const float4*>(tr)[q];
                const unsigned voffs = ok ? off + (unsigned)soff : BAD_OFFSET;
                if (p.residual) {
                    const unsigned roffs = ok ? off + (unsigned)((co0 + co) * rplaneBytes) : BAD_OFFSET;
                    const u32x4 rr = __builtin_amdgcn_raw_buffer_load_b128(rrs, (int)roffs, 0, 0);
                    const float4 rf = __builtin_bit_cast(float4, rr);
                    if (ACT == ISR_ACT_GATE) {
                        v.x = rf.x > 0.f ? v.x : 0.f; v.y = rf.y > 0.f ? v.y : 0.f;
                        v.z = rf.z > 0.f ? v.z : 0.f; v.w = rf.w > 0.f ? v.w : 0.f;
                    } else {
                        v.x += rf.x; v.y += rf.y; v.z += rf.z; v.w += rf.w;
                    }
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), yrs, (int)voffs, 0, 0);
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);   // reads done before the next row overwrites the slab
        }
    } else {
    const unsigned khoff = (unsigned)(4 * kh) * (unsigned)planeBytes, rkhoff = (unsigned)(4 * kh) * (unsigned)rplaneBytes;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int oy = oy0 + wave * R + r;
        const bool pix_ok = ox < p.W && oy < p.H && !(p.dbg & 2);
        const unsigned pix = pix_ok ? (unsigned)((oy * p.W + ox) * 4) + khoff : BAD_OFFSET;
        const unsigned rpix = pix_ok ? (unsigned)((oy * p.W + ox) * 4) + rkhoff : BAD_OFFSET;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            float rv[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int soff = (co0 + m * 32 + (i & 3) + 8 * (i >> 2)) * rplaneBytes;
                rv[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrs, (int)rpix, soff, 0));
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int soff = (co0 + m * 32 + (i & 3) + 8 * (i >> 2)) * planeBytes;
                float v = isr_activate<ACT>(acc[m][r][i] + bv[m][i], p.slope);
                if (ACT == ISR_ACT_GATE) v = rv[i] > 0.f ? v : 0.f;
                else v += rv[i];
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), yrs, (int)pix, soff, 0);
            }
        }
    }
    }
    });
}

// ---- forward, two workgroups per CU ------------------------------------------------------------
// Same implicit GEMM, re-budgeted so that TWO workgroups share a CU: one M tile (32 output channels)
// per workgroup and 8-channel chunks -> 66 KB of LDS and < 256 registers per wave.  The point is not
// the MFMA loop (it is the same 8 x 32x32x2 per k-step pair) but everything around it: with one
// workgroup per CU the prologue (first chunk from HBM), the per-chunk barriers and the epilogue
// (bias/act/residual/store) leave the matrix pipe idle ~15 % of a workgroup's life; with two, the
// other workgroup's waves issue MFMAs in those holes.  The price is that a tile's input patch is staged
// once per 32-channel group (both groups of a tile are adjacent workgroups on one XCD, so the second
// read is an L2 hit).
//
// x2-upsampling loader: the 18x34 hi-res patch is 9x17 "quads" of 2x2 pixels (rows 2a-1, 2a; columns
// 2b-1, 2b relative to the tile) that interpolate the SAME four source texels (bilinear,
// align_corners=False: odd pixels weigh the pair .75/.25, even pixels .25/.75; clamping the source
// coordinates to the image reproduces the border rule).  One thread does a whole quad: 4 loads and 16
// FMAs for 4 patch elements instead of 16 loads and 28 FMAs, two ds_write_b64.
//
// R = output rows per wave: 4 (tile 16x32) or 1 (tile 4x32).  The small tile is for small problems (the 32x32
// training crops, 128x128 previews): a 64->64 layer on 16 crops is 64 workgroups of the big tile for 512 slots,
// 256 of the small one; per workgroup it stages the same weights for a quarter of the MFMAs, which only pays when
// the GPU would otherwise idle.
constexpr int CK2 = 8;
constexpr int DUMP2 = 4 * NTHREADS;                                  // write-only sink behind each patch buffer
template <int R> struct Geo2 {
    static constexpr int TH_ = 4 * R;                                    // tile rows
    static constexpr int PH_ = TH_ + 2;                                  // patch rows
    static constexpr int PLANE_ = PH_ * PW;                              // floats per channel plane (612 / 204)
    static constexpr int CHUNK_ = CK2 * PLANE_;                          // 4896 / 1632 floats
    static constexpr int NEL_ = (CHUNK_ + NTHREADS - 1) / NTHREADS;      // 20 / 7 patch elements per thread and chunk
    static constexpr int QUADS_ = (PH_ / 2) * 17;                        // 2x2 quads per channel (x2 loader)
    static constexpr int NQ_ = QUADS_ * CK2;                             // 1224 / 408 quads per chunk
    static constexpr int NQT_ = (NQ_ + NTHREADS - 1) / NTHREADS;         // 5 / 2 per thread
    static constexpr int PSTRIDE_ = CHUNK_ + DUMP2;                      // patch buffer + its sink
};
constexpr int WCH2 = 9 * CK2 * 32;                                   // 2304 weight floats per chunk
constexpr int NW42 = WCH2 / 4;                                       // 576 float4
constexpr int NWI2 = (NW42 + NTHREADS - 1) / NTHREADS;               // 3 per thread
constexpr int KSTEPS2 = CK2 / 2;                                     // 4 k-steps per tap
constexpr int NSLOTS2 = 9 * KSTEPS2;                                 // 36 k-steps per chunk = staging slots
// LDS: [2][WCH2] weights, then 2 x ([CHUNK_] patch, [DUMP2] sink): a masked-off staging lane keeps its
// offset and lands in the sink of whichever buffer is being filled.  (The epilogue transposes through the first
// 4 x 8 KB of it.)
template <int R>
constexpr size_t conv_fwd2_lds_bytes() { return (size_t)(2 * WCH2 + 2 * Geo2<R>::PSTRIDE_) * sizeof(float); }
static_assert(conv_fwd2_lds_bytes<1>() >= 4 * 64 * 32 * sizeof(float), "epilogue slab");

constexpr unsigned Q_DX = 1u, Q_DY = 2u, Q_VX0 = 4u, Q_VX1 = 8u, Q_VY0 = 16u, Q_VY1 = 32u;   // aux bits; LDS float offset << 8

template <bool UPS, int R>
__global__ __launch_bounds__(NTHREADS, 2) void conv3x3_fwd2_kernel(const ConvParams p)
{
    typedef Geo2<R> G;
    constexpr int PLANE_ = G::PLANE_, CHUNK2 = G::CHUNK_, NEL2 = G::NEL_, QUADS2 = G::QUADS_, NQ2 = G::NQ_, NQT2 = G::NQT_, PSTRIDE2 = G::PSTRIDE_;
    constexpr int NEL = UPS ? NQT2 : NEL2;      // patch staging items per thread and chunk
    constexpr int NLD = UPS ? 4 : 1;            // loads per item
    constexpr int NITEMS = NEL + NWI2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* wlds0 = smem;                        // [2][9][CK2][32]
    float* patch0 = smem + 2 * WCH2;            // 2 x (patch, sink)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int tilesPerImage = p.tilesX * p.tilesY;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int grp = lid % p.cgroups;            // the channel groups of one tile are neighbours on one XCD
    const int bid = lid / p.cgroups;
    const int n = bid / tilesPerImage;
    const int t = bid - n * tilesPerImage;
    const int ty = t / p.tilesX, tx = t - ty * p.tilesX;
    const int oy0 = ty * G::TH_, ox0 = tx * TW;
    const int co0 = p.co0 + grp * 32;

    unsigned long long st0 = 0, st1 = 0, st2 = 0;
    if (p.dbg & 8) st0 = __builtin_amdgcn_s_memtime();
    f32x16 acc[1][R];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[0][r][i] = 0.0f;

    const int nchunks = (p.Cin + CK2 - 1) / CK2;     // weight rows >= Cin are zero, input planes >= Cin read as 0
    const int planeIn = p.xPlane;
    const float* ximg = p.x + (size_t)n * p.xImage;
    const unsigned rowBytes = (unsigned)p.Win * 4u;
    auto chunk_rsrc = [&](int chunk) -> rsrc_t {
        const int left = p.Cin - chunk * CK2;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ximg + (size_t)chunk * CK2 * planeIn), 0,
                                                 left > 0 ? left * planeIn * 4 : 0, 0x00020000);
    };

    // Staging plan, one or two words per item, chunk invariant.
    //  plain: plan = byte offset of the element in the chunk's planes (PLAN_BAD outside the image);
    //         item i is patch element tid + 256 i and goes to that float of the patch buffer.
    //  UPS:   plan = byte offset of the quad's top-left source texel (clamped into the image),
    //         aux  = Q_* flags | (float offset of the quad's first element in the patch buffer) << 8.
    unsigned plan[NEL], aux[UPS ? NEL : 1];
    if (UPS) {
        const int m0 = (oy0 >> 1) - 1, q0 = (ox0 >> 1) - 1;
#pragma unroll
        for (int i = 0; i < NEL; ++i) {
            const int qi = tid + i * NTHREADS;
            const int c = qi / QUADS2, rem = qi - c * QUADS2;
            const int a = rem / 17, b = rem - a * 17;
            const int m = m0 + a, q = q0 + b;                          // source row / column of the top-left texel
            const int r0 = min(max(m, 0), p.Hin - 1), r1 = min(max(m + 1, 0), p.Hin - 1);
            const int c0 = min(max(q, 0), p.Win - 1), c1 = min(max(q + 1, 0), p.Win - 1);
            const int gy = oy0 - 1 + 2 * a, gx = ox0 - 1 + 2 * b;      // hi-res coordinates of the quad's first pixel
            unsigned f = 0;
            if (c1 != c0) f |= Q_DX;
            if (r1 != r0) f |= Q_DY;
            if ((unsigned)gx < (unsigned)p.W) f |= Q_VX0;
            if ((unsigned)(gx + 1) < (unsigned)p.W) f |= Q_VX1;
            if ((unsigned)gy < (unsigned)p.H) f |= Q_VY0;
            if ((unsigned)(gy + 1) < (unsigned)p.H) f |= Q_VY1;
            const bool exists = qi < NQ2;
            plan[i] = exists ? (unsigned)((c * p.xPlane + r0 * p.Win + c0) * 4) : PLAN_BAD;
            aux[i] = f | (unsigned)(exists ? c * PLANE_ + (2 * a) * PW + 2 * b : CHUNK2 + 2 * tid) << 8;
        }
    } else {
#pragma unroll
        for (int i = 0; i < NEL; ++i) plan[i] = plan_element<false, CHUNK2, PLANE_>(p, tid + i * NTHREADS, oy0, ox0);
    }
    // weights: float4 f = tid + 256 i of the chunk's [9][CK2][32] block; f = tap*64 + k*8 + c4
    unsigned woff[NWI2];
#pragma unroll
    for (int i = 0; i < NWI2; ++i) {
        const int f = min(tid + i * NTHREADS, NW42 - 1);
        const int tap = f >> 6, k = (f >> 3) & 7, c4 = f & 7;
        woff[i] = (unsigned)((tap * p.cinPad + k) * p.coutPad + co0 + c4 * 4);
    }
    const float* wchunk = p.w;
    const size_t wchunkStep = (size_t)CK2 * p.coutPad;
    auto weight_item = [&](const float* base, int i) -> float4 { return *reinterpret_cast<const float4*>(base + woff[i]); };

    auto issue_patch = [&](rsrc_t rs, unsigned w, unsigned a, float (&raw)[NLD]) {
        if (UPS) {
            const unsigned dx = (a & Q_DX) ? 4u : 0u, dy = (a & Q_DY) ? rowBytes : 0u;
            raw[0] = buf_load(rs, w); raw[UPS ? 1 : 0] = buf_load(rs, w + dx);
            raw[UPS ? 2 : 0] = buf_load(rs, w + dy); raw[UPS ? 3 : 0] = buf_load(rs, w + dy + dx);
        } else {
            raw[0] = buf_load(rs, w);
        }
    };
    // UPS: the four pixels of a quad from its four texels, zeroed outside the image (the conv's padding)
    auto park_quad = [&](float* pbuf, unsigned a, const float (&raw)[NLD]) {
        const float t00 = raw[0], t01 = raw[UPS ? 1 : 0], t10 = raw[UPS ? 2 : 0], t11 = raw[UPS ? 3 : 0];
        const float top0 = __builtin_fmaf(0.25f, t01, 0.75f * t00), top1 = __builtin_fmaf(0.75f, t01, 0.25f * t00);
        const float bot0 = __builtin_fmaf(0.25f, t11, 0.75f * t10), bot1 = __builtin_fmaf(0.75f, t11, 0.25f * t10);
        float v00 = __builtin_fmaf(0.25f, bot0, 0.75f * top0), v01 = __builtin_fmaf(0.25f, bot1, 0.75f * top1);
        float v10 = __builtin_fmaf(0.75f, bot0, 0.25f * top0), v11 = __builtin_fmaf(0.75f, bot1, 0.25f * top1);
        const int ia = (int)a;
        const unsigned mx0 = (unsigned)__builtin_amdgcn_sbfe(ia, 2, 1), mx1 = (unsigned)__builtin_amdgcn_sbfe(ia, 3, 1);
        const unsigned my0 = (unsigned)__builtin_amdgcn_sbfe(ia, 4, 1), my1 = (unsigned)__builtin_amdgcn_sbfe(ia, 5, 1);
        auto masked = [](float v, unsigned m) { return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v) & m); };
        v00 = masked(v00, mx0 & my0); v01 = masked(v01, mx1 & my0);
        v10 = masked(v10, mx0 & my1); v11 = masked(v11, mx1 & my1);
        float* d = pbuf + (a >> 8);
        *reinterpret_cast<float2*>(d) = make_float2(v00, v01);
        *reinterpret_cast<float2*>(d + PW) = make_float2(v10, v11);
    };

    // prologue: chunk 0 in full
    {
        const rsrc_t rs = chunk_rsrc(0);
        float4 wv[NWI2];
#pragma unroll
        for (int i = 0; i < NWI2; ++i) wv[i] = weight_item(wchunk, i);
        float raw[NEL][NLD];
#pragma unroll
        for (int i = 0; i < NEL; ++i) issue_patch(rs, plan[i], UPS ? aux[UPS ? i : 0] : 0u, raw[i]);
#pragma unroll
        for (int i = 0; i < NEL; ++i) {
            if (UPS) park_quad(patch0, aux[UPS ? i : 0], raw[i]);
            else if (tid + i * NTHREADS < CHUNK2) patch0[tid + i * NTHREADS] = raw[i][0];
        }
#pragma unroll
        for (int i = 0; i < NWI2; ++i)
            if (tid + i * NTHREADS < NW42) reinterpret_cast<float4*>(wlds0)[tid + i * NTHREADS] = wv[i];
    }
    __syncthreads();
    if (p.dbg & 8) st1 = __builtin_amdgcn_s_memtime();

    const int j = lane & 31;
    const int kh = lane >> 5;
    auto load_ops = [&](float& a, float (&b)[R], const float* w_, const float* p_, int kk) {
        a = w_[(2 * kk) * 32];
#pragma unroll
        for (int r = 0; r < R; ++r) b[r] = p_[(2 * kk) * PLANE_ + r * PW];
    };
    auto mfma_step = [&](float a, const float (&b)[R]) {
#pragma unroll
        for (int r = 0; r < R; ++r) acc[0][r] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[r], acc[0][r], 0, 0, 0);
    };

    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int buf = chunk & 1;
        const bool more = chunk + 1 < nchunks && !(p.dbg & 1);
        const float* pb = patch0 + buf * PSTRIDE2 + kh * PLANE_ + (wave * R) * PW + j;
        const float* wb = wlds0 + buf * WCH2 + kh * 32 + j;
        float* const pfill = patch0 + (buf ^ 1) * PSTRIDE2;          // buffer being filled (+ its sink)
        float* const pnext = pfill + tid;
        float4* const wnext = reinterpret_cast<float4*>(wlds0 + (buf ^ 1) * WCH2) + tid;
        const rsrc_t rsn = chunk_rsrc(chunk + 1);
        wchunk += wchunkStep;                       // now the next chunk's weight rows (only read when `more`)
        float a0, b0[R], a1, b1[R];
        float raw[NEL][NLD];
        float4 wraw[NWI2];
        load_ops(a0, b0, wb, pb, 0);
        auto opaque = [&](unsigned w) -> unsigned {   // keeps hipcc from hoisting the decode of every item out of the loop
            asm volatile("" : "+v"(w));
            return w;
        };
        float* const plast = (tid < CHUNK2 - (NEL2 - 1) * NTHREADS) ? pnext + (NEL2 - 1) * NTHREADS : pfill + CHUNK2 + tid;
        float4* const wlast = (tid < NW42 - (NWI2 - 1) * NTHREADS) ? wnext + (NWI2 - 1) * NTHREADS
                                                                   : reinterpret_cast<float4*>(pfill + CHUNK2) + tid;
        // Staging of the next chunk rides in the MFMA slots (slot s = k-step s of the chunk, 36 per
        // chunk), all loads early and all LDS writes late so that every load has >= 20 k-steps
        // (>= 5000 cycles) to come back.  No branches around the memory operations: a branch makes
        // hipcc's waitcnt pass fall back to vmcnt(0).
        //   plain: items issued two per slot in slots 0..11, parked two per slot in slots 24..35
        //   UPS:   quad i issued in slot 2i, weights in slots 10..12; quad i parked in slot 20+3i, weights 33..35
        auto park_item = [&](int q) {
            if (q < NEL) {
                if (UPS) park_quad(pfill, opaque(aux[UPS ? q : 0]), raw[q]);
                else if (q < NEL - 1) pnext[q * NTHREADS] = raw[q][0];
                else *plast = raw[q][0];
            } else {
                const int i = q - NEL;
                if (i < NWI2 - 1) wnext[i * NTHREADS] = wraw[i]; else *wlast = wraw[i];
            }
        };
        auto issue_item = [&](int q) {
            if (q < NEL) issue_patch(rsn, opaque(plan[q]), UPS ? opaque(aux[UPS ? q : 0]) : 0u, raw[q]);
            else wraw[q - NEL] = weight_item(wchunk, q - NEL);
        };
        auto slot = [&](int s) {
            if (UPS) {
                if (s >= 20 && s < 20 + 3 * NEL && (s - 20) % 3 == 0) park_item((s - 20) / 3);
                if (s >= NSLOTS2 - NWI2) park_item(NEL + s - (NSLOTS2 - NWI2));
                if (s < 2 * NEL && s % 2 == 0) issue_item(s / 2);
                if (s >= 2 * NEL && s < 2 * NEL + NWI2) issue_item(NEL + s - 2 * NEL);
            } else {
                constexpr int PARK0 = NSLOTS2 - (NITEMS + 1) / 2;     // 24
                static_assert(UPS || PARK0 >= (NITEMS + 1) / 2, "issue phase must end before the park phase starts");
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int qp = (s - PARK0) * 2 + h;
                    if (s >= PARK0 && qp < NITEMS) park_item(qp);
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int q = s * 2 + h;
                    if (q < NITEMS) issue_item(q);
                }
            }
        };
        static_assert(20 + 3 * (NQT2 - 1) < NSLOTS2 - NWI2 && 2 * NQT2 + NWI2 <= 20, "UPS staging schedule");
        static_assert(CHUNK2 + 2 * NTHREADS + PW + 2 <= PSTRIDE2, "the quad sink must stay inside the buffer's sink");
        auto run_taps = [&](auto MORE) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap - dy * 3;
            const int tn = tap + 1, dyn = tn / 3, dxn = tn - dyn * 3;
            const float* pt = pb + dy * PW + dx;
            const float* wt = wb + tap * CK2 * 32;
            const float* ptn = pb + dyn * PW + dxn;
            const float* wtn = wb + tn * CK2 * 32;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < KSTEPS2; kk += 2) {
                load_ops(a1, b1, wt, pt, kk + 1);
                __builtin_amdgcn_sched_barrier(0);
                mfma_step(a0, b0);
                if (decltype(MORE)::value) slot(tap * KSTEPS2 + kk);
                __builtin_amdgcn_sched_barrier(0);
                if (kk + 2 < KSTEPS2) load_ops(a0, b0, wt, pt, kk + 2);
                else if (tap < 8) load_ops(a0, b0, wtn, ptn, 0);
                __builtin_amdgcn_sched_barrier(0);
                mfma_step(a1, b1);
                if (decltype(MORE)::value) slot(tap * KSTEPS2 + kk + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        };
        if (more) run_taps(std::true_type{}); else run_taps(std::false_type{});
        __syncthreads();
    }
    if (p.dbg & 8) st2 = __builtin_amdgcn_s_memtime();
    conv_epilogue<1, R>(p, acc, smem, n, oy0, ox0, co0, wave, lane);
    if ((p.dbg & 8) && tid == 0 && p.stamps) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long st3 = __builtin_amdgcn_s_memtime();
        unsigned long long* o = p.stamps + (size_t)blockIdx.x * 4;
        o[0] = st0; o[1] = st1; o[2] = st2; o[3] = st3;
    }
}

template <int MT>
constexpr size_t conv_fwd_lds_bytes() { return (size_t)(2 * CHUNK + 2 * 9 * CK * MT * 32 + 4 * NTHREADS) * sizeof(float); }

// ---- forward for small problems: one output row per workgroup, K split over the waves -------------
// A batch of 32x32 training crops is 256 workgroups of the 4x32 tiling (16 crops; 32 for the two crops per GPU of
// BASELINE config #3): at most one wave per SIMD, each running its 288 dependent k-steps alone -- 13 us of workgroup
// life however few pixels there are.  Here a workgroup is ONE output row of 32 pixels x 32 output channels and its
// four waves split the input channels; partial sums meet in LDS.  Four times the workgroups with a quarter of the
// serial chain each.  Operands come straight from global memory / L2 (one dword per lane and MFMA operand, both
// coalesced: weights [tap][ci][co] along co, pixels along x) through buffer descriptors whose out-of-range reads
// return the zero padding; no staging, no barriers in the loop.  The loads bound it: a dword-per-lane buffer load
// occupies the texture path for 16 cycles per wave (stamps: 9200 cycles for the 4 x 144 loads of a CU's workgroup,
// the 72 MFMAs of a wave are 4600).  Tried and dropped: channel / tap offsets as the instruction's scalar offset
// (45 % slower), one dwordx3 load per patch row (dword-aligned multi-dword buffer loads return the first dword in
// every component on this part), a second accumulator (no change: the chain is not what the wave waits for).
constexpr int ROW_VARIANT = 12;

__global__ __launch_bounds__(NTHREADS) void conv3x3_rowsplit_kernel(const ConvParams p)
{
    constexpr int KSB = 8;      // k-steps per block of up-front operand loads (144 registers; with 4 and more waves per
                                // SIMD the kernel is L1/TA bound and loses to the LDS-staged tiling from ~200 tiles on)
    __shared__ float part[4][16][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, kh = lane >> 5;
    int bid = blockIdx.x;
    const int cg = bid % p.cgroups; bid /= p.cgroups;
    const int tx = bid % p.tilesX; bid /= p.tilesX;
    const int oy = bid % p.H, n = bid / p.H;
    const int ox0 = tx * 32, co0 = cg * 32;

    const int chPerWave = p.cinPad >> 2;                     // cinPad is a multiple of 16
    const int c0 = wave * chPerWave;
    const rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x + (size_t)n * p.xImage), 0,
                                                         (int)((size_t)p.Cin * p.xPlane * 4), 0x00020000);
    const rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, (int)((size_t)9 * p.cinPad * p.coutPad * 4), 0x00020000);
    // per-lane pixel offsets of the 9 taps inside a channel plane (BAD_OFFSET = zero padding)
    unsigned xoff[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int iy = oy + t / 3 - 1, ix = ox0 + j + t % 3 - 1;
        xoff[t] = ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) ? (unsigned)(iy * p.W + ix) * 4u : BAD_OFFSET;
    }
    const unsigned planeBytes = (unsigned)p.xPlane * 4u;
    const unsigned wTap = (unsigned)(p.cinPad * p.coutPad) * 4u, wRow = (unsigned)p.coutPad * 4u;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;

    // a block of KS k-steps (2 channels each): ALL operand loads are issued before the first MFMA, so the wave pays
    // one L2 round trip per block instead of one per k-step (there is no other wave on the SIMD to hide it)
    auto block = [&](auto ks_tag, int c) {
        constexpr int KS = decltype(ks_tag)::value;
        float a[KS][9], b[KS][9];
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            const unsigned ci = (unsigned)(c + 2 * k + kh);
            const unsigned cx = ci * planeBytes;
            const unsigned cw = ci * wRow + (unsigned)(co0 + j) * 4u;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                a[k][t] = buf_load(wrs, cw + (unsigned)t * wTap);
                b[k][t] = buf_load(xrs, xoff[t] == BAD_OFFSET ? BAD_OFFSET : cx + xoff[t]);
            }
        }
#pragma unroll
        for (int k = 0; k < KS; ++k)
#pragma unroll
            for (int t = 0; t < 9; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k][t], b[k][t], acc, 0, 0, 0);
    };
    unsigned long long st0 = 0, st1 = 0, st2 = 0;
    if (p.dbg & 8) st0 = __builtin_amdgcn_s_memtime();
    int c = c0, rem = chPerWave >> 1;
    for (; rem >= KSB; rem -= KSB, c += 2 * KSB) block(std::integral_constant<int, KSB>{}, c);
    for (; rem >= 2; rem -= 2, c += 4) block(std::integral_constant<int, 2>{}, c);
    if (p.dbg & 8) { st1 = __builtin_amdgcn_s_memtime(); }
#pragma unroll
    for (int i = 0; i < 16; ++i) part[wave][i][lane] = acc[i];
    __syncthreads();
    // wave w finishes accumulator registers 4w .. 4w+3: D row (cout) = (reg & 3) + 8 * (reg >> 2) + 4 * kh, col (pixel) = j
    const int ox = ox0 + j;
    const rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(p.y + (size_t)n * p.yImage, 0, (int)((size_t)p.Cout * p.yPlane * 4), 0x00020000);
    const rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.residual ? p.residual + (size_t)n * p.rImage : p.y), 0,
                                                         p.residual ? (int)((size_t)p.Cout * p.rPlane * 4) : 0, 0x00020000);
    const unsigned pix = ox < p.W ? (unsigned)(oy * p.W + ox) * 4u : BAD_OFFSET;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int reg = wave * 4 + q;
        const int co = co0 + (reg & 3) + 8 * (reg >> 2) + 4 * kh;
        float v = (part[0][reg][lane] + part[1][reg][lane]) + (part[2][reg][lane] + part[3][reg][lane]);
        v += p.bias[min(co, p.Cout - 1)];
        if (p.act == ISR_ACT_RELU) v = v > 0.f ? v : 0.f;
        else if (p.act == ISR_ACT_LEAKY) v = v > 0.f ? v : v * p.slope;
        const bool ok = pix != BAD_OFFSET && co < p.Cout;
        if (p.residual) {
            const float r = buf_load(rrs, ok ? pix + (unsigned)co * (unsigned)p.rPlane * 4u : BAD_OFFSET);
            if (p.act == ISR_ACT_GATE) v = r > 0.f ? v : 0.f; else v += r;
        }
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), yrs, ok ? (int)(pix + (unsigned)co * (unsigned)p.yPlane * 4u) : (int)BAD_OFFSET, 0, 0);
    }
    if ((p.dbg & 8) && tid == 0 && p.stamps) {
        st2 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long st3 = __builtin_amdgcn_s_memtime();
        unsigned long long* o = p.stamps + (size_t)blockIdx.x * 4;
        o[0] = st0; o[1] = st1; o[2] = st2; o[3] = st3;
    }
}

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

constexpr int WG_MAX_SEG = 32;
struct WGradParams {
    // K runs over the pixels of `segments` tensor pairs of N images each (the frames of a training clip share the
    // weights, so one launch can take all of them: train.py defers the weight gradients to the end of the backward)
    const float* x[WG_MAX_SEG];     // each [N][Cin][H][W]
    const float* gz[WG_MAX_SEG];    // each [N][Cout][H][W]
    float* slabs;       // [G][9][64][64]
    float* bslabs;      // [G][64] partial sums of gz per output channel (bias gradient), or NULL
    int N, Cin, H, W, Cout;
    int ci0, co0;       // channel group handled by this launch
    int tilesX, tilesY, ntiles;
    const float* scale; // split-operand kernel only: { 2^S, 2^-S } with max |gz| 2^S in [2^13, 2^14)
    ISR_DIAG_MEMBER(int, dbg, 0);            // diagnostics (isrDebugSetAblation; conv3x3_wgrad_split2_kernel): 1 no MFMAs, 2 no split / park, 4 no fetch
};

// Staging is branch-free and software pipelined: the next tile's 8 + 51 elements per thread are fetched through
// buffer descriptors (out-of-image / out-of-range channels read 0) into registers BEFORE the tile's MFMA block and
// parked in LDS after it.  TAPS = 9: one workgroup accumulates all taps (144 accumulator registers per wave);
// TAPS = 3: three workgroups share a tile set, one row of the 3x3 each -- three times the workgroups for small
// problems (the 32x32 training crops are 128 tiles for 256 CUs).
// KSPLIT (at most 32 output channels in this launch, e.g. the 64 -> 6 output layer): instead of idling on an empty
// second channel block, the waves m = 1 take rows 2..3 of every tile and the waves m = 0 rows 0..1, each pair writes
// its own slab (g and G + g) -- half the MFMAs per workgroup.
template <int TAPS, bool KSPLIT = false>
__global__ __launch_bounds__(NTHREADS, 1) void conv3x3_wgrad_kernel(const WGradParams p)
{
    __shared__ float gzs[64 * GZ_STRIDE];
    __shared__ float xps[64 * XP_PLANE];
    constexpr int TG = 9 / TAPS;                       // workgroups per slab
    constexpr int NGZ = 64 * WG_PX / NTHREADS;         // 32 gz elements per thread and tile
    constexpr int NXE = (64 * (WG_TH + 2) * XP_W + NTHREADS - 1) / NTHREADS;   // 51 x elements
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = wave >> 1, nn = wave & 1;
    const int j = lane & 31, kh = lane >> 5;
    const int g = blockIdx.x / TG, tap0 = (blockIdx.x - g * TG) * TAPS, nslab = gridDim.x / TG;

    f32x16 acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;

    // per-thread element descriptors, tile invariant: gz element i = (channel c, pixel px) -> LDS slot + (ry, rx);
    // x element i = (channel c, patch row r, patch column col)
    unsigned xdesc[NXE];
#pragma unroll
    for (int i = 0; i < NXE; ++i) {
        const int e = tid + i * NTHREADS;
        const int c = e / ((WG_TH + 2) * XP_W), rem = e - c * ((WG_TH + 2) * XP_W);
        const int r = rem / XP_W, col = rem - r * XP_W;
        xdesc[i] = e < 64 * (WG_TH + 2) * XP_W ? (unsigned)(c | (r << 8) | (col << 16)) : 0xFFFFFFFFu;
    }
    const size_t planeBytes = (size_t)p.H * p.W * 4;
    const int tilesPerImage = p.tilesX * p.tilesY;
    float gv[NGZ], xv[NXE];

    auto fetch = [&](int tile) {
        const int ng = tile / tilesPerImage;             // image index over all segments
        const int t2 = tile - ng * tilesPerImage;
        const int seg = ng / p.N, n = ng - seg * p.N;
        const int ty = t2 / p.tilesX, tx = t2 - ty * p.tilesX;
        const int oy0 = ty * WG_TH, ox0 = tx * WG_TW;
        const int gzc = p.Cout - p.co0, xc = p.Cin - p.ci0;
        const rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.gz[seg] + ((size_t)n * p.Cout + p.co0) * p.H * p.W), 0,
                                                             (int)((gzc < 64 ? gzc : 64) * planeBytes), 0x00020000);
        const rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x[seg] + ((size_t)n * p.Cin + p.ci0) * p.H * p.W), 0,
                                                             (int)((xc < 64 ? xc : 64) * planeBytes), 0x00020000);
#pragma unroll
        for (int i = 0; i < NGZ; ++i) {
            const int e = tid + i * NTHREADS;
            const int c = e >> 7, px = e & 127, ry = px >> 5, rx = px & 31;
            const int gy = oy0 + ry, gx = ox0 + rx;
            const bool ok = gy < p.H && gx < p.W;
            gv[i] = buf_load(grs, ok ? (unsigned)((c * p.H + gy) * p.W + gx) * 4u : BAD_OFFSET);
        }
#pragma unroll
        for (int i = 0; i < NXE; ++i) {
            const unsigned d = xdesc[i];
            const int c = d & 255, r = (d >> 8) & 255, col = (d >> 16) & 255;
            const int gy = oy0 + r - 1, gx = ox0 + col - 1;
            const bool ok = d != 0xFFFFFFFFu && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
            xv[i] = buf_load(xrs, ok ? (unsigned)((c * p.H + gy) * p.W + gx) * 4u : BAD_OFFSET);
        }
    };
    auto park = [&]() {
#pragma unroll
        for (int i = 0; i < NGZ; ++i) {
            const int e = tid + i * NTHREADS;
            gzs[(e >> 7) * GZ_STRIDE + (e & 127)] = gv[i];
        }
#pragma unroll
        for (int i = 0; i < NXE; ++i) {
            const unsigned d = xdesc[i];
            if (d != 0xFFFFFFFFu) xps[(d & 255) * XP_PLANE + ((d >> 8) & 255) * XP_W + ((d >> 16) & 255)] = xv[i];
        }
    };

    float bsum = 0.0f;                                   // this lane's share of sum(gz[co = m*32+j]) (pixels kh, kh+2, ..)
    int tile = g;
    if (tile < p.ntiles) { fetch(tile); park(); }
    __syncthreads();
    for (; tile < p.ntiles; tile += nslab) {
        const bool more = tile + nslab < p.ntiles;
        if (more) fetch(tile + nslab);
        __builtin_amdgcn_sched_barrier(0);               // the loads stay above the MFMA block
        const float* ga = &gzs[((KSPLIT ? 0 : m * 32) + j) * GZ_STRIDE + kh];
        const float* xb = &xps[(nn * 32 + j) * XP_PLANE + kh];
#pragma unroll
        for (int r2 = 0; r2 < (KSPLIT ? WG_TH / 2 : WG_TH); ++r2) {
            const int ry = KSPLIT ? m * (WG_TH / 2) + r2 : r2;
#pragma unroll 4
            for (int rx = 0; rx < WG_TW; rx += 2) {
                const float a = ga[ry * WG_TW + rx];
                bsum += a;
#pragma unroll
                for (int t = 0; t < TAPS; ++t) {
                    const int tap = TAPS == 9 ? t : -1;
                    const int dy = TAPS == 9 ? tap / 3 : 0, dx = TAPS == 9 ? tap - dy * 3 : t;
                    const float b = xb[(ry + (TAPS == 9 ? dy : tap0 / 3)) * XP_W + rx + dx];
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
                }
            }
        }
        __syncthreads();                                 // everyone has finished reading this tile
        if (more) park();
        __syncthreads();
    }
    // slab[g][tap][co(64)][ci(64)]
    const int gs = KSPLIT ? g + m * nslab : g;        // KSPLIT: slabs [0, G) hold rows 0..1, [G, 2G) rows 2..3; channels 32.. stay 0
    float* slab = p.slabs + (size_t)gs * 9 * 64 * 64;
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int co = (KSPLIT ? 0 : m * 32) + (i & 3) + 8 * (i >> 2) + 4 * kh;
            slab[((size_t)(tap0 + t) * 64 + co) * 64 + nn * 32 + j] = acc[t][i];
        }
    // bias gradient: the A operand already passes every gz value of the tile set through the waves with nn == 0
    const float btot = bsum + __shfl_xor(bsum, 32, 64);
    if (p.bslabs && nn == 0 && tap0 == 0 && kh == 0) p.bslabs[(size_t)gs * 64 + (KSPLIT ? 0 : m * 32) + j] = btot;
}

// ---- weight gradient with bf16 operands (the opt-in mixed-precision training mode, not the parity step) ----------------
// Same decomposition (M = co, N = ci, K = pixels of 4x32 tiles, one slab per workgroup, the same reduction), but the
// tile is rounded to bf16 on its way into LDS and multiplied by v_mfma_f32_32x32x16_bf16 (16 pixels per instruction):
// A fragment = 8 consecutive pixels of one gz channel (one ds_read_b128 from [co][pixel]), B fragment = 8 consecutive
// pixels of one x channel SHIFTED by the tap: the row is read once as five dwords (10 pixels) and the three horizontal
// taps are dwords 0..3, the 16-bit funnel shifts of neighbouring dwords, and dwords 1..4 -- no shifted copies in LDS.
// At this MFMA rate the kernel is bound by streaming the tile (84 KB per 2300 MFMA cycles).  Needs W % 4 == 0.
typedef __bf16 wbf16x8 __attribute__((ext_vector_type(8)));
constexpr int BG_PITCH = 272;                  // bytes per gz channel row in LDS: 128 pixels x 2 + 16 (bank spread)
constexpr int BX_ROW = 80;                     // bytes per x patch row: 40 pixels (34 used), 16-byte aligned rows
constexpr int BX_PITCH = 6 * BX_ROW + 16;      // 496 bytes per x channel
constexpr int BX_PAIRS = 17;                   // packed pixel pairs per patch row (34 columns)

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi)
{
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    bf2 v; v[0] = (__bf16)lo; v[1] = (__bf16)hi;
    return __builtin_bit_cast(unsigned, v);
}

__global__ __launch_bounds__(NTHREADS, 1) void conv3x3_wgrad_bf16_kernel(const WGradParams p)
{
    __shared__ __attribute__((aligned(16))) unsigned char gzs[64 * BG_PITCH];
    __shared__ __attribute__((aligned(16))) unsigned char xps[64 * BX_PITCH];
    constexpr int NGQ = 64 * WG_PX / 4 / NTHREADS;                              // 8 gz quads per thread and tile
    constexpr int NXP = (64 * (WG_TH + 2) * BX_PAIRS + NTHREADS - 1) / NTHREADS;  // 26 x pairs
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = wave >> 1, nn = wave & 1;
    const int j = lane & 31, kh = lane >> 5;
    const int g = blockIdx.x, nslab = gridDim.x;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;
    float bs[NGQ];                                                             // fp32 sums of gz per thread (channel tid/32 + 8 i)
#pragma unroll
    for (int i = 0; i < NGQ; ++i) bs[i] = 0.0f;

    const size_t planeBytes = (size_t)p.H * p.W * 4;
    const int tilesPerImage = p.tilesX * p.tilesY;
    u32x4 gq[NGQ];
    float xlo[NXP], xhi[NXP];

    auto fetch = [&](int tile) {
        const int ng = tile / tilesPerImage;
        const int t2 = tile - ng * tilesPerImage;
        const int seg = ng / p.N, n = ng - seg * p.N;
        const int ty = t2 / p.tilesX, tx = t2 - ty * p.tilesX;
        const int oy0 = ty * WG_TH, ox0 = tx * WG_TW;
        const int gzc = p.Cout - p.co0, xc = p.Cin - p.ci0;
        const rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.gz[seg] + ((size_t)n * p.Cout + p.co0) * p.H * p.W), 0,
                                                             (int)((gzc < 64 ? gzc : 64) * planeBytes), 0x00020000);
        const rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x[seg] + ((size_t)n * p.Cin + p.ci0) * p.H * p.W), 0,
                                                             (int)((xc < 64 ? xc : 64) * planeBytes), 0x00020000);
#pragma unroll
        for (int i = 0; i < NGQ; ++i) {                                        // quad q = (channel q / 32, 4 pixels)
            const int q = tid + i * NTHREADS;
            const int c = q >> 5, px = (q & 31) * 4, ry = px >> 5, rx = px & 31;
            const int gy = oy0 + ry, gx = ox0 + rx;
            const bool ok = gy < p.H && gx < p.W;                              // W % 4 == 0: a quad is inside or outside as a whole
            gq[i] = __builtin_amdgcn_raw_buffer_load_b128(grs, (int)(ok ? (unsigned)((c * p.H + gy) * p.W + gx) * 4u : BAD_OFFSET), 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NXP; ++i) {                                        // pair e = (channel, patch row, patch columns 2 pr, 2 pr + 1)
            const int e = tid + i * NTHREADS;
            const int c = e / ((WG_TH + 2) * BX_PAIRS), rem = e - c * ((WG_TH + 2) * BX_PAIRS);
            const int r = rem / BX_PAIRS, pr = rem - r * BX_PAIRS;
            const int gy = oy0 + r - 1, gx = ox0 + 2 * pr - 1;
            const bool in = e < 64 * (WG_TH + 2) * BX_PAIRS && (unsigned)gy < (unsigned)p.H;
            const unsigned off = (unsigned)((c * p.H + gy) * p.W + gx) * 4u;
            xlo[i] = buf_load(xrs, (in && (unsigned)gx < (unsigned)p.W) ? off : BAD_OFFSET);
            xhi[i] = buf_load(xrs, (in && (unsigned)(gx + 1) < (unsigned)p.W) ? off + 4u : BAD_OFFSET);
        }
    };
    auto park = [&]() {
#pragma unroll
        for (int i = 0; i < NGQ; ++i) {
            const int q = tid + i * NTHREADS;
            const float4 f = __builtin_bit_cast(float4, gq[i]);
            bs[i] += (f.x + f.y) + (f.z + f.w);
            uint2 v; v.x = pack_bf16(f.x, f.y); v.y = pack_bf16(f.z, f.w);
            *reinterpret_cast<uint2*>(gzs + (q >> 5) * BG_PITCH + (q & 31) * 8) = v;
        }
#pragma unroll
        for (int i = 0; i < NXP; ++i) {
            const int e = tid + i * NTHREADS;
            if (e < 64 * (WG_TH + 2) * BX_PAIRS) {
                const int c = e / ((WG_TH + 2) * BX_PAIRS), rem = e - c * ((WG_TH + 2) * BX_PAIRS);
                const int r = rem / BX_PAIRS, pr = rem - r * BX_PAIRS;
                *reinterpret_cast<unsigned*>(xps + c * BX_PITCH + r * BX_ROW + pr * 4) = pack_bf16(xlo[i], xhi[i]);
            }
        }
    };

    int tile = g;
    if (tile < p.ntiles) { fetch(tile); park(); }
    __syncthreads();
    for (; tile < p.ntiles; tile += nslab) {
        const bool more = tile + nslab < p.ntiles;
        if (more) fetch(tile + nslab);
        __builtin_amdgcn_sched_barrier(0);
        const unsigned char* ga = gzs + (m * 32 + j) * BG_PITCH + kh * 16;
        const unsigned char* xb = xps + (nn * 32 + j) * BX_PITCH + kh * 16;
#pragma unroll
        for (int ry = 0; ry < WG_TH; ++ry) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {                                   // k-step = 16 pixels of the row: columns 16 ks + 8 kh ..
                const wbf16x8 a = __builtin_bit_cast(wbf16x8, *reinterpret_cast<const u32x4*>(ga + (ry * 32 + ks * 16) * 2));
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const unsigned char* row = xb + (ry + dy) * BX_ROW + ks * 32;
                    const u32x4 d = *reinterpret_cast<const u32x4*>(row);
                    const unsigned d4 = *reinterpret_cast<const unsigned*>(row + 16);
                    u32x4 f1, f2;
                    f1.x = __builtin_amdgcn_alignbit(d.y, d.x, 16); f1.y = __builtin_amdgcn_alignbit(d.z, d.y, 16);
                    f1.z = __builtin_amdgcn_alignbit(d.w, d.z, 16); f1.w = __builtin_amdgcn_alignbit(d4, d.w, 16);
                    f2.x = d.y; f2.y = d.z; f2.z = d.w; f2.w = d4;
                    acc[dy * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(wbf16x8, d), acc[dy * 3 + 0], 0, 0, 0);
                    acc[dy * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(wbf16x8, f1), acc[dy * 3 + 1], 0, 0, 0);
                    acc[dy * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(wbf16x8, f2), acc[dy * 3 + 2], 0, 0, 0);
                }
            }
        }
        __syncthreads();
        if (more) park();
        __syncthreads();
    }
    float* slab = p.slabs + (size_t)g * 9 * 64 * 64;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int co = m * 32 + (i & 3) + 8 * (i >> 2) + 4 * kh;
            slab[((size_t)t * 64 + co) * 64 + nn * 32 + j] = acc[t][i];
        }
    // bias gradient from the fp32 values: thread t's sum i belongs to channel t / 32 + 8 i; reduce over the 32 lanes
    if (p.bslabs) {
#pragma unroll
        for (int i = 0; i < NGQ; ++i) {
            float v = bs[i];
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
            if ((lane & 31) == 0) p.bslabs[(size_t)g * 64 + (tid >> 5) + 8 * i] = v;
        }
    }
}

// ---- weight gradient on split operands: fp32-equivalent accuracy on the fp16 matrix pipe (the training default) --------
// The decomposition and tile pipeline of conv3x3_wgrad_bf16_kernel, with every operand carried as two fp16 numbers
// (csrc/sr_conv_split.hip): gz, whose magnitude is unknown, is first scaled by a power of two taken from max |gz| over
// the launch's tensors (wgrad_absmax / wgrad_scale kernels; undone exactly by the slab reduction) and split as
// hi = RN16(g), lo = RN16(g - hi); x is split as hi = RN16(x), lo' = RN16((x - hi) 2^11) and meets gz_hi 2^-11 (an
// exponent shift of the A fragment in registers).  A product is gz_lo x_hi + (gz_hi 2^-11) x_lo' + gz_hi x_hi: three
// v_mfma_f32_32x32x16_f16 instead of eight v_mfma_f32_32x32x2_f32 at twice the cycles each -> 5.3x fewer matrix cycles.
typedef _Float16 wf16x8 __attribute__((ext_vector_type(8)));
constexpr int WS_LDS_BYTES = 2 * (64 * BG_PITCH + 64 * BX_PITCH);          // hi and lo planes of gz and x: 98 304 B

__device__ __forceinline__ void split_pair(float a, float b, float lo_scale, unsigned& hi, unsigned& lo)
{
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    h2 vh, vl;
    vh[0] = (_Float16)a; vh[1] = (_Float16)b;
    vl[0] = (_Float16)((a - (float)vh[0]) * lo_scale); vl[1] = (_Float16)((b - (float)vh[1]) * lo_scale);
    hi = __builtin_bit_cast(unsigned, vh); lo = __builtin_bit_cast(unsigned, vl);
}

__global__ __launch_bounds__(NTHREADS, 1) void conv3x3_wgrad_split_kernel(const WGradParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char wsl[];
    unsigned char* gzh = wsl;
    unsigned char* gzl = gzh + 64 * BG_PITCH;
    unsigned char* xph = gzl + 64 * BG_PITCH;
    unsigned char* xpl = xph + 64 * BX_PITCH;
    constexpr int NGQ = 64 * WG_PX / 4 / NTHREADS;                              // 8 gz quads per thread and tile
    constexpr int NXP = (64 * (WG_TH + 2) * BX_PAIRS + NTHREADS - 1) / NTHREADS;  // 26 x pairs
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = wave >> 1, nn = wave & 1;
    const int j = lane & 31, kh = lane >> 5;
    const int g = blockIdx.x, nslab = gridDim.x;
    const float gscale = p.scale[0];

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;
    float bs[NGQ];                                                             // fp32 sums of gz per thread (channel tid/32 + 8 i)
#pragma unroll
    for (int i = 0; i < NGQ; ++i) bs[i] = 0.0f;

    const size_t planeBytes = (size_t)p.H * p.W * 4;
    const int tilesPerImage = p.tilesX * p.tilesY;
    u32x4 gq[NGQ];
    float xlo[NXP], xhi[NXP];

    auto fetch = [&](int tile) {
        const int ng = tile / tilesPerImage;
        const int t2 = tile - ng * tilesPerImage;
        const int seg = ng / p.N, n = ng - seg * p.N;
        const int ty = t2 / p.tilesX, tx = t2 - ty * p.tilesX;
        const int oy0 = ty * WG_TH, ox0 = tx * WG_TW;
        const int gzc = p.Cout - p.co0, xc = p.Cin - p.ci0;
        const rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.gz[seg] + ((size_t)n * p.Cout + p.co0) * p.H * p.W), 0,
                                                             (int)((gzc < 64 ? gzc : 64) * planeBytes), 0x00020000);
        const rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x[seg] + ((size_t)n * p.Cin + p.ci0) * p.H * p.W), 0,
                                                             (int)((xc < 64 ? xc : 64) * planeBytes), 0x00020000);
#pragma unroll
        for (int i = 0; i < NGQ; ++i) {                                        // quad q = (channel q / 32, 4 pixels)
            const int q = tid + i * NTHREADS;
            const int c = q >> 5, px = (q & 31) * 4, ry = px >> 5, rx = px & 31;
            const int gy = oy0 + ry, gx = ox0 + rx;
            const bool ok = gy < p.H && gx < p.W;                              // W % 4 == 0: a quad is inside or outside as a whole
            gq[i] = __builtin_amdgcn_raw_buffer_load_b128(grs, (int)(ok ? (unsigned)((c * p.H + gy) * p.W + gx) * 4u : BAD_OFFSET), 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NXP; ++i) {                                        // pair e = (channel, patch row, patch columns 2 pr, 2 pr + 1)
            const int e = tid + i * NTHREADS;
            const int c = e / ((WG_TH + 2) * BX_PAIRS), rem = e - c * ((WG_TH + 2) * BX_PAIRS);
            const int r = rem / BX_PAIRS, pr = rem - r * BX_PAIRS;
            const int gy = oy0 + r - 1, gx = ox0 + 2 * pr - 1;
            const bool in = e < 64 * (WG_TH + 2) * BX_PAIRS && (unsigned)gy < (unsigned)p.H;
            const unsigned off = (unsigned)((c * p.H + gy) * p.W + gx) * 4u;
            xlo[i] = buf_load(xrs, (in && (unsigned)gx < (unsigned)p.W) ? off : BAD_OFFSET);
            xhi[i] = buf_load(xrs, (in && (unsigned)(gx + 1) < (unsigned)p.W) ? off + 4u : BAD_OFFSET);
        }
    };
    auto park = [&]() {
#pragma unroll
        for (int i = 0; i < NGQ; ++i) {
            const int q = tid + i * NTHREADS;
            const float4 f = __builtin_bit_cast(float4, gq[i]);
            bs[i] += (f.x + f.y) + (f.z + f.w);
            uint2 vh, vl;
            split_pair(f.x * gscale, f.y * gscale, 1.0f, vh.x, vl.x);
            split_pair(f.z * gscale, f.w * gscale, 1.0f, vh.y, vl.y);
            *reinterpret_cast<uint2*>(gzh + (q >> 5) * BG_PITCH + (q & 31) * 8) = vh;
            *reinterpret_cast<uint2*>(gzl + (q >> 5) * BG_PITCH + (q & 31) * 8) = vl;
        }
#pragma unroll
        for (int i = 0; i < NXP; ++i) {
            const int e = tid + i * NTHREADS;
            if (e < 64 * (WG_TH + 2) * BX_PAIRS) {
                const int c = e / ((WG_TH + 2) * BX_PAIRS), rem = e - c * ((WG_TH + 2) * BX_PAIRS);
                const int r = rem / BX_PAIRS, pr = rem - r * BX_PAIRS;
                unsigned vh, vl;
                split_pair(xlo[i], xhi[i], 2048.0f, vh, vl);
                *reinterpret_cast<unsigned*>(xph + c * BX_PITCH + r * BX_ROW + pr * 4) = vh;
                *reinterpret_cast<unsigned*>(xpl + c * BX_PITCH + r * BX_ROW + pr * 4) = vl;
            }
        }
    };

    int tile = g;
    if (tile < p.ntiles) { fetch(tile); park(); }
    __syncthreads();
    for (; tile < p.ntiles; tile += nslab) {
        const bool more = tile + nslab < p.ntiles;
        if (more) fetch(tile + nslab);
        __builtin_amdgcn_sched_barrier(0);
        const int ga = (m * 32 + j) * BG_PITCH + kh * 16;
        const int xb = (nn * 32 + j) * BX_PITCH + kh * 16;
#pragma unroll
        for (int ry = 0; ry < WG_TH; ++ry) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {                                   // k-step = 16 pixels of the row: columns 16 ks + 8 kh ..
                const wf16x8 ah = __builtin_bit_cast(wf16x8, *reinterpret_cast<const u32x4*>(gzh + ga + (ry * 32 + ks * 16) * 2));
                const wf16x8 al = __builtin_bit_cast(wf16x8, *reinterpret_cast<const u32x4*>(gzl + ga + (ry * 32 + ks * 16) * 2));
                const wf16x8 as = ah * (_Float16)0.00048828125f;              // gz_hi 2^-11: partner of the scaled x_lo'
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const int row = xb + (ry + dy) * BX_ROW + ks * 32;
                    const u32x4 dh = *reinterpret_cast<const u32x4*>(xph + row);
                    const unsigned dh4 = *reinterpret_cast<const unsigned*>(xph + row + 16);
                    const u32x4 dl = *reinterpret_cast<const u32x4*>(xpl + row);
                    const unsigned dl4 = *reinterpret_cast<const unsigned*>(xpl + row + 16);
                    u32x4 h1, h2, l1, l2;
                    h1.x = __builtin_amdgcn_alignbit(dh.y, dh.x, 16); h1.y = __builtin_amdgcn_alignbit(dh.z, dh.y, 16);
                    h1.z = __builtin_amdgcn_alignbit(dh.w, dh.z, 16); h1.w = __builtin_amdgcn_alignbit(dh4, dh.w, 16);
                    h2.x = dh.y; h2.y = dh.z; h2.z = dh.w; h2.w = dh4;
                    l1.x = __builtin_amdgcn_alignbit(dl.y, dl.x, 16); l1.y = __builtin_amdgcn_alignbit(dl.z, dl.y, 16);
                    l1.z = __builtin_amdgcn_alignbit(dl.w, dl.z, 16); l1.w = __builtin_amdgcn_alignbit(dl4, dl.w, 16);
                    l2.x = dl.y; l2.y = dl.z; l2.z = dl.w; l2.w = dl4;
#define ISR_WS3(T, BH, BL)                                                                                              \
                    acc[T] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, __builtin_bit_cast(wf16x8, BH), acc[T], 0, 0, 0);   \
                    acc[T] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as, __builtin_bit_cast(wf16x8, BL), acc[T], 0, 0, 0);   \
                    acc[T] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, __builtin_bit_cast(wf16x8, BH), acc[T], 0, 0, 0);
                    ISR_WS3(dy * 3 + 0, dh, dl)
                    ISR_WS3(dy * 3 + 1, h1, l1)
                    ISR_WS3(dy * 3 + 2, h2, l2)
#undef ISR_WS3
                }
            }
        }
        __syncthreads();
        if (more) park();
        __syncthreads();
    }
    float* slab = p.slabs + (size_t)g * 9 * 64 * 64;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int co = m * 32 + (i & 3) + 8 * (i >> 2) + 4 * kh;
            slab[((size_t)t * 64 + co) * 64 + nn * 32 + j] = acc[t][i];
        }
    // bias gradient from the fp32 values: thread t's sum i belongs to channel t / 32 + 8 i; reduce over the 32 lanes
    if (p.bslabs) {
#pragma unroll
        for (int i = 0; i < NGQ; ++i) {
            float v = bs[i];
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
            if ((lane & 31) == 0) p.bslabs[(size_t)g * 64 + (tid >> 5) + 8 * i] = v;
        }
    }
}

// ---- the same sums with the staging on its own waves ----------------------------------------------------------------------
// conv3x3_wgrad_split_kernel runs ONE wave per SIMD (144 accumulator registers + 92 of staging: 507 registers), so the split of the
// next tile (a few hundred VALU instructions per thread), its LDS writes and the latency of every operand read sit in series with
// the MFMAs: 8.7 us per 128-pixel tile for 3.4 us of matrix work.  Here a workgroup is EIGHT waves, two per SIMD: waves 0..3 own the
// accumulators and do nothing but read operands and issue MFMAs, waves 4..7 fetch, split and park -- the matrix pipe and the vector
// pipe of a SIMD work for different waves at the same time.  Both roles fit 256 registers (accumulators + operands | two register
// sets of staging).  LDS is double buffered, which a 4-row tile does not fit twice: the unit of the pipeline is HALF a tile (2 rows
// x 32 pixels, 61 KB per buffer); a workgroup walks the tiles of conv3x3_wgrad_split_kernel in the same order, each as its two
// halves, so every accumulator sees the same products in the same order -- slabs and weight gradients are bit-identical.  (The bias
// sums are grouped differently per thread: equal to rounding.)  One barrier per half-tile.
constexpr int W2_THREADS = 512;
constexpr int W2_TH = 2;
constexpr int W2_GP = W2_TH * WG_TW * 2 + 16;                 // bytes per gz channel: 64 pixels x 2 + 16 (bank spread: 36 dwords)
constexpr int W2_XP = (W2_TH + 2) * BX_ROW + 16;              // bytes per x channel: 4 patch rows + 16 (84 dwords)
constexpr int W2_GZ = 64 * W2_GP, W2_X = 64 * W2_XP;
constexpr int W2_BUF = 2 * W2_GZ + 2 * W2_X;                  // hi and lo planes of gz and x: 61 440 B
constexpr int W2_LDS_BYTES = 2 * W2_BUF;                      // 122 880 B
constexpr int W2_NGQ = 64 * W2_TH * WG_TW / 4 / 256;          // 4 gz quads per staging thread and half-tile
constexpr int W2_NXP = 64 * (W2_TH + 2) * BX_PAIRS / 256;     // 17 x pairs (4352 = 17 x 256 exactly)
static_assert(64 * (W2_TH + 2) * BX_PAIRS == W2_NXP * 256, "x pairs divide evenly over the staging threads");

struct W2Stage { u32x4 gq[W2_NGQ]; float xlo[W2_NXP], xhi[W2_NXP]; };

__global__ __launch_bounds__(W2_THREADS, 1) void conv3x3_wgrad_split2_kernel(const WGradParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char w2l[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool consumer = wave < 4;
    const int m = (wave >> 1) & 1, nn = wave & 1;
    const int j = lane & 31, kh = lane >> 5;
    const int pt = tid & 255;                                                  // staging thread
    const int g = blockIdx.x, nslab = gridDim.x;
    const int nhalf = g < p.ntiles ? 2 * ((p.ntiles - g + nslab - 1) / nslab) : 0;
    const float gscale = p.scale[0];

    const size_t planeBytes = (size_t)p.H * p.W * 4;
    const int tilesPerImage = p.tilesX * p.tilesY;

    auto fetch = [&](W2Stage& st, int it) {
        const int tile = g + (it >> 1) * nslab;
        const int ng = tile / tilesPerImage;
        const int t2 = tile - ng * tilesPerImage;
        const int seg = ng / p.N, n = ng - seg * p.N;
        const int ty = t2 / p.tilesX, tx = t2 - ty * p.tilesX;
        const int oy0 = ty * WG_TH + (it & 1) * W2_TH, ox0 = tx * WG_TW;
        const int gzc = p.Cout - p.co0, xc = p.Cin - p.ci0;
        const rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.gz[seg] + ((size_t)n * p.Cout + p.co0) * p.H * p.W), 0,
                                                             (int)((gzc < 64 ? gzc : 64) * planeBytes), 0x00020000);
        const rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x[seg] + ((size_t)n * p.Cin + p.ci0) * p.H * p.W), 0,
                                                             (int)((xc < 64 ? xc : 64) * planeBytes), 0x00020000);
#pragma unroll
        for (int i = 0; i < W2_NGQ; ++i) {                                     // quad q = (channel q / 16, row, 4 pixels)
            const int q = pt + i * 256;
            const int c = q >> 4, ry = (q >> 3) & 1, rx = (q & 7) * 4;
            const int gy = oy0 + ry, gx = ox0 + rx;
            const bool ok = gy < p.H && gx < p.W;                              // W % 4 == 0: a quad is inside or outside as a whole
            st.gq[i] = __builtin_amdgcn_raw_buffer_load_b128(grs, (int)(ok ? (unsigned)((c * p.H + gy) * p.W + gx) * 4u : BAD_OFFSET), 0, 0);
        }
#pragma unroll
        for (int i = 0; i < W2_NXP; ++i) {                                     // pair e = (channel, patch row, patch columns 2 pr, 2 pr + 1)
            const int e = pt + i * 256;
            const int c = e / ((W2_TH + 2) * BX_PAIRS), rem = e - c * ((W2_TH + 2) * BX_PAIRS);
            const int r = rem / BX_PAIRS, pr = rem - r * BX_PAIRS;
            const int gy = oy0 + r - 1, gx = ox0 + 2 * pr - 1;
            const bool in = (unsigned)gy < (unsigned)p.H;
            const unsigned off = (unsigned)((c * p.H + gy) * p.W + gx) * 4u;
            st.xlo[i] = buf_load(xrs, (in && (unsigned)gx < (unsigned)p.W) ? off : BAD_OFFSET);
            st.xhi[i] = buf_load(xrs, (in && (unsigned)(gx + 1) < (unsigned)p.W) ? off + 4u : BAD_OFFSET);
        }
    };
    auto park = [&](const W2Stage& st, float (&bs)[W2_NGQ], int buf) {
        unsigned char* gzh = w2l + buf * W2_BUF;
        unsigned char* gzl = gzh + W2_GZ;
        unsigned char* xph = gzl + W2_GZ;
        unsigned char* xpl = xph + W2_X;
#pragma unroll
        for (int i = 0; i < W2_NGQ; ++i) {
            const int q = pt + i * 256;
            const float4 f = __builtin_bit_cast(float4, st.gq[i]);
            bs[i] += (f.x + f.y) + (f.z + f.w);
            uint2 vh, vl;
            split_pair(f.x * gscale, f.y * gscale, 1.0f, vh.x, vl.x);
            split_pair(f.z * gscale, f.w * gscale, 1.0f, vh.y, vl.y);
            *reinterpret_cast<uint2*>(gzh + (q >> 4) * W2_GP + (q & 15) * 8) = vh;
            *reinterpret_cast<uint2*>(gzl + (q >> 4) * W2_GP + (q & 15) * 8) = vl;
        }
#pragma unroll
        for (int i = 0; i < W2_NXP; ++i) {
            const int e = pt + i * 256;
            const int c = e / ((W2_TH + 2) * BX_PAIRS), rem = e - c * ((W2_TH + 2) * BX_PAIRS);
            const int r = rem / BX_PAIRS, pr = rem - r * BX_PAIRS;
            unsigned vh, vl;
            split_pair(st.xlo[i], st.xhi[i], 2048.0f, vh, vl);
            *reinterpret_cast<unsigned*>(xph + c * W2_XP + r * BX_ROW + pr * 4) = vh;
            *reinterpret_cast<unsigned*>(xpl + c * W2_XP + r * BX_ROW + pr * 4) = vl;
        }
    };
    auto compute = [&](f32x16 (&acc)[9], int buf) {
        const unsigned char* gzh = w2l + buf * W2_BUF;
        const unsigned char* gzl = gzh + W2_GZ;
        const unsigned char* xph = gzl + W2_GZ;
        const unsigned char* xpl = xph + W2_X;
        const int ga = (m * 32 + j) * W2_GP + kh * 16;
        const int xb = (nn * 32 + j) * W2_XP + kh * 16;
#pragma unroll
        for (int ry = 0; ry < W2_TH; ++ry) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {                                   // k-step = 16 pixels of the row: columns 16 ks + 8 kh ..
                const wf16x8 ah = __builtin_bit_cast(wf16x8, *reinterpret_cast<const u32x4*>(gzh + ga + (ry * 32 + ks * 16) * 2));
                const wf16x8 al = __builtin_bit_cast(wf16x8, *reinterpret_cast<const u32x4*>(gzl + ga + (ry * 32 + ks * 16) * 2));
                const wf16x8 as = ah * (_Float16)0.00048828125f;              // gz_hi 2^-11: partner of the scaled x_lo'
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const int row = xb + (ry + dy) * BX_ROW + ks * 32;
                    const u32x4 dh = *reinterpret_cast<const u32x4*>(xph + row);
                    const unsigned dh4 = *reinterpret_cast<const unsigned*>(xph + row + 16);
                    const u32x4 dl = *reinterpret_cast<const u32x4*>(xpl + row);
                    const unsigned dl4 = *reinterpret_cast<const unsigned*>(xpl + row + 16);
                    u32x4 h1, h2, l1, l2;
                    h1.x = __builtin_amdgcn_alignbit(dh.y, dh.x, 16); h1.y = __builtin_amdgcn_alignbit(dh.z, dh.y, 16);
                    h1.z = __builtin_amdgcn_alignbit(dh.w, dh.z, 16); h1.w = __builtin_amdgcn_alignbit(dh4, dh.w, 16);
                    h2.x = dh.y; h2.y = dh.z; h2.z = dh.w; h2.w = dh4;
                    l1.x = __builtin_amdgcn_alignbit(dl.y, dl.x, 16); l1.y = __builtin_amdgcn_alignbit(dl.z, dl.y, 16);
                    l1.z = __builtin_amdgcn_alignbit(dl.w, dl.z, 16); l1.w = __builtin_amdgcn_alignbit(dl4, dl.w, 16);
                    l2.x = dl.y; l2.y = dl.z; l2.z = dl.w; l2.w = dl4;
#define ISR_WS3(T, BH, BL)                                                                                              \
                    acc[T] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, __builtin_bit_cast(wf16x8, BH), acc[T], 0, 0, 0);   \
                    acc[T] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as, __builtin_bit_cast(wf16x8, BL), acc[T], 0, 0, 0);   \
                    acc[T] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, __builtin_bit_cast(wf16x8, BH), acc[T], 0, 0, 0);
                    ISR_WS3(dy * 3 + 0, dh, dl)
                    ISR_WS3(dy * 3 + 1, h1, l1)
                    ISR_WS3(dy * 3 + 2, h2, l2)
#undef ISR_WS3
                }
                __builtin_amdgcn_sched_barrier(0);                             // operands are fetched one k-step ahead at most (256 registers)
            }
        }
    };

    // half-tile `it` is computed from buffer it & 1 while the staging waves park half-tile it + 1 (requested one iteration ago) into
    // the other buffer and then request half-tile it + 2: ONE register set, the requests fly while the staging waves wait at the
    // barrier for the MFMAs.  The two roles are separate loops (same number of barriers in each: nhalf + 1), so that the register
    // allocation is the larger of the two roles and not their sum.
    if (consumer) {
        f32x16 acc[9];
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;
        __syncthreads();
        for (int it = 0; it < nhalf; ++it) {
            if (!(p.dbg & 1)) compute(acc, it & 1);
            __syncthreads();
        }
        float* slab = p.slabs + (size_t)g * 9 * 64 * 64;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int co = m * 32 + (i & 3) + 8 * (i >> 2) + 4 * kh;
                slab[((size_t)t * 64 + co) * 64 + nn * 32 + j] = acc[t][i];
            }
    } else {
        // one register set (a second one, requests two iterations ahead, measured the same: the staging waves are bound by their
        // own work -- 900 vector instructions and 38 memory requests per thread and half-tile -- not by the latency)
        W2Stage st;
        float bs[W2_NGQ];                                                      // fp32 sums of gz (channel pt / 16 + 16 i)
#pragma unroll
        for (int i = 0; i < W2_NGQ; ++i) bs[i] = 0.0f;
        const bool doPark = !(p.dbg & 2), doFetch = !(p.dbg & 4);
        if (nhalf > 0) {
            fetch(st, 0);
            park(st, bs, 0);
            fetch(st, 1);                                                      // nhalf is even
        }
        __syncthreads();
        for (int it = 0; it < nhalf; ++it) {
            if (it + 1 < nhalf) {
                if (doPark) park(st, bs, (it + 1) & 1);
                if (it + 2 < nhalf && doFetch) fetch(st, it + 2);
            }
            __syncthreads();
        }
        // bias gradient from the fp32 values: staging thread t's sum i belongs to channel t / 16 + 16 i; reduce over the 16 lanes
        if (p.bslabs) {
#pragma unroll
            for (int i = 0; i < W2_NGQ; ++i) {
                float v = bs[i];
#pragma unroll
                for (int off = 8; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
                if ((lane & 15) == 0) p.bslabs[(size_t)g * 64 + (pt >> 4) + 16 * i] = v;
            }
        }
    }
}

// max |gz| over the tensors of a launch: one partial per workgroup, then { 2^S, 2^-S } with max |gz| 2^S in [2^13, 2^14)
struct AbsMaxParams { const float* t[WG_MAX_SEG]; int segments; long long count; };
__global__ __launch_bounds__(256) void wgrad_absmax_kernel(const AbsMaxParams p, float* __restrict__ partial)
{
    __shared__ float red[256];
    float mx = 0.0f;
    const long long quads = p.count >> 2;
    for (int s = 0; s < p.segments; ++s) {
        const float4* src = reinterpret_cast<const float4*>(p.t[s]);
        for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < quads; q += (long long)gridDim.x * 256) {
            const float4 f = src[q];
            mx = fmaxf(fmaxf(mx, fmaxf(fabsf(f.x), fabsf(f.y))), fmaxf(fabsf(f.z), fabsf(f.w)));
        }
        if (blockIdx.x == 0 && threadIdx.x < (p.count & 3)) mx = fmaxf(mx, fabsf(p.t[s][(quads << 2) + threadIdx.x]));
    }
    red[threadIdx.x] = mx;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
__global__ __launch_bounds__(256) void wgrad_scale_kernel(const float* __restrict__ partial, int n, float* __restrict__ scale)
{
    __shared__ float red[256];
    float mx = 0.0f;
    for (int i = threadIdx.x; i < n; i += 256) mx = fmaxf(mx, partial[i]);
    red[threadIdx.x] = mx;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        int S = 0;
        if (red[0] > 0.0f && red[0] < 3.0e38f) {
            S = 13 - ilogbf(red[0]);
            S = S < -100 ? -100 : (S > 100 ? 100 : S);
        }
        scale[0] = ldexpf(1.0f, S);
        scale[1] = ldexpf(1.0f, -S);
    }
}

// ... the same pair from maxima the PRODUCERS of the gz tensors left behind (bit patterns of max |gz|, one word per tensor: the
// split-operand convolutions note it in their epilogues -- SplitConvParams::absmax, Block2Params::zmax / ymax): no pass over gz
struct MaxFlagParams { const unsigned* f[WG_MAX_SEG]; int segments, words; };
__global__ __launch_bounds__(1024) void wgrad_scale_flags_kernel(const MaxFlagParams p, float* __restrict__ scale)
{
    __shared__ unsigned red[16];
    unsigned m = 0u;
    for (int i = threadIdx.x; i < p.words; i += 1024) {
#pragma unroll 8
        for (int s = 0; s < p.segments; ++s) { const unsigned t = p.f[s][i]; m = t > m ? t : m; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned t = (unsigned)__shfl_xor((int)m, o, 64); m = t > m ? t : m; }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 16; ++k) m = red[k] > m ? red[k] : m;
        const float mx = __builtin_bit_cast(float, m);
        int S = 0;
        if (mx > 0.0f && mx < 3.0e38f) {
            S = 13 - ilogbf(mx);
            S = S < -100 ? -100 : (S > 100 ? 100 : S);
        }
        scale[0] = ldexpf(1.0f, S);
        scale[1] = ldexpf(1.0f, -S);
    }
}

// dw[co][ci][tap] = sum over the G slabs, in a fixed order (bitwise reproducible run to run).  A workgroup owns 64
// consecutive slab elements; its four waves take the slabs g % 4 == wave with four loads in flight each and the
// four partial sums are combined through LDS -- 576 workgroups x 16 independent loads instead of 144 x 8, the
// reduction of 256 slabs (38 MB) is latency bound otherwise.  Workgroup 576 reduces the 64 bias sums the same way.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slabs, int G, float* __restrict__ dw,
                                    int Cout, int Cin, int co0, int ci0, const float* __restrict__ bslabs, float* __restrict__ db,
                                    const float* __restrict__ scale = nullptr,      // split kernel: slabs hold sums scaled by scale[0]
                                    int accumulate = 0)                             // bit 0: dw += (bit 1: db +=) instead of = : straight into .grad
{
    __shared__ float part[4][64];
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const bool isBias = blockIdx.x == 9 * 64;
    if (isBias && !bslabs) return;
    const size_t stride = isBias ? 64 : (size_t)9 * 64 * 64;
    const float* src = (isBias ? bslabs : slabs + (size_t)blockIdx.x * 64) + lane;
    float s[4] = { 0.f, 0.f, 0.f, 0.f };
    int g = q;
    for (; g + 12 < G; g += 16) {
#pragma unroll
        for (int k = 0; k < 4; ++k) s[k] += src[(size_t)(g + 4 * k) * stride];
    }
    for (int k = 0; g < G; g += 4, ++k) s[k] += src[(size_t)g * stride];
    part[q][lane] = (s[0] + s[1]) + (s[2] + s[3]);
    __syncthreads();
    if (q != 0) return;
    const float total = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
    if (isBias) {
        if (co0 + lane < Cout) db[co0 + lane] = (accumulate & 2) ? __fadd_rn(db[co0 + lane], total) : total;
        return;
    }
    const int e = blockIdx.x * 64 + lane;                    // over [9][64][64]
    const int ci = e & 63, co = (e >> 6) & 63, tap = e >> 12;
    if (co0 + co >= Cout || ci0 + ci >= Cin) return;
    // (product and sum rounded separately: the same bits as this kernel followed by `grad += dw`)
    const float gval = scale ? __fmul_rn(total, scale[1]) : total;
    float* dst = dw + ((size_t)(co0 + co) * Cin + (ci0 + ci)) * 9 + tap;
    *dst = (accumulate & 1) ? __fadd_rn(*dst, gval) : gval;
}

constexpr int WGRAD_MAX_SLABS = 512;

}  // namespace

// ---- optional per-dispatch timing (bench.py): start/stop events ride on the dispatch packet itself
// (hipExtLaunchKernelGGL), so no extra barrier packets or cache flushes perturb the stream.
struct ProfileRecord { int variant; double flops; hipEvent_t e0, e1; };
static bool g_profile = false;
static bool g_profile_small = false;      // isrProfileEnable(2): also the frame's small kernels (variants >= ISR_VARIANT_TRUNK_PACK)
static std::vector<ProfileRecord> g_records;
static std::vector<hipEvent_t> g_event_pool;
static size_t g_pool_used = 0;
static hipEvent_t pool_event()
{
    if (g_pool_used == g_event_pool.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        g_event_pool.push_back(e);
    }
    return g_event_pool[g_pool_used++];
}

void isr_profile_record(int variant, double flops, hipEvent_t* e0, hipEvent_t* e1)
{
    if (!g_profile || (variant >= ISR_VARIANT_TRUNK_PACK && variant <= ISR_VARIANT_UPS_FRAME && !g_profile_small)) return;
    *e0 = pool_event(); *e1 = pool_event();
    g_records.push_back({ variant, flops, *e0, *e1 });
}

[[maybe_unused]] static int g_conv_dbg = 0;
static int g_wgrad_split_form = isr_diag_env_int("ISR_WGRAD_FORM", 2) == 1 ? 1 : 2;
static int g_conv_tile = 0;   // 0: automatic, 1: always 4x32 tiles, 2: always 16x32 tiles, 3: one row per workgroup (tools / tests)
static long long g_row_threshold = 160;   // automatic: rows when the 4x32 tiling has at most this many workgroups
                                          // (HIP-graph chain of 64 -> 64 layers on N crops of 32x32, rows vs 4x32 tiles:
                                          //  N=2 7.5 vs 15.1 us, N=4 8.4 vs 15.2, N=8 12.1 vs 15.1, N=12 16.1 vs 16.0, N=16 19.5 vs 16.1)
static int g_conv_algo = 1;   // 0: one workgroup per CU (64 channels), 1: two per CU (32 channels each)
[[maybe_unused]] static unsigned long long* g_conv_stamps = nullptr;

extern "C" {

// Per-dispatch profiling of isrConv3x3Forward (see include/isr_sr_kernels.h)
int isrProfileEnable(int on)
{
    g_profile = on != 0;
    g_profile_small = on == 2;
    g_records.clear();
    g_pool_used = 0;
    return 0;
}

int isrProfileCount(void) { return (int)g_records.size(); }

int isrProfileGet(int i, int* variant, double* flops, float* ms)
{
    if (i < 0 || i >= (int)g_records.size()) return -1;
    const ProfileRecord& r = g_records[i];
    float t = 0.f;
    if (!r.e0 || !r.e1 || hipEventElapsedTime(&t, r.e0, r.e1) != hipSuccess) return -2;
    *variant = r.variant; *flops = r.flops; *ms = t;
    return 0;
}

#ifdef ISR_DIAG
void isrDebugSetAblation(int bits) { g_conv_dbg = bits; }
// split-operand weight gradient: 2 = staging on its own waves (conv3x3_wgrad_split2_kernel, the default), 1 = one wave per SIMD
void isrDebugSetWgradSplitForm(int form) { g_wgrad_split_form = form == 1 ? 1 : 2; }
int isrDebugWgradSplitForm(void) { return g_wgrad_split_form; }
void isrDebugSetForwardAlgo(int a) { g_conv_algo = a; }
void isrDebugSetForwardTile(int t) { g_conv_tile = t; }   // not part of the public header
void isrDebugSetRowThreshold(long long n) { g_row_threshold = n; }
void isrDebugSetStampBuffer(void* p) { g_conv_stamps = (unsigned long long*)p; }
#endif

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
    if (N <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return -1;
    const long long xp = upsample2x ? (long long)(H / 2) * (W / 2) : (long long)H * W, yp = (long long)H * W;
    return isrConv3x3ForwardStrided(x, wprep, bias, residual, y, N, Cin, H, W, Cout, act, slope, upsample2x,
                                    xp, xp * Cin, yp, yp * Cout, yp, yp * Cout, stream);
}

int isrConv3x3ForwardStrided(const float* x, const float* wprep, const float* bias, const float* residual, float* y,
                             int N, int Cin, int H, int W, int Cout, int act, float slope, int upsample2x,
                             long long xPlane, long long xImage, long long yPlane, long long yImage,
                             long long rPlane, long long rImage, void* stream)
{
    if (!x || !wprep || !y || N <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return -1;
    if (upsample2x && ((H & 1) || (W & 1))) return -1;
    if (act < ISR_ACT_NONE || act > ISR_ACT_GATE) return -1;
    if (act == ISR_ACT_GATE && !residual) return -1;
    {
        const long long rowsIn = (long long)(upsample2x ? H / 2 : H) * (upsample2x ? W / 2 : W);
        if (xPlane < rowsIn || yPlane < (long long)H * W || (residual && rPlane < (long long)H * W)) return -1;
        // buffer descriptors address one image with 32-bit byte offsets
        if (xPlane * Cin * 4 > 0x7fffffffLL || yPlane * Cout * 4 > 0x7fffffffLL || (residual && rPlane * Cout * 4 > 0x7fffffffLL)) return -1;
    }
    ConvParams p;
    p.x = x; p.w = wprep; p.bias = bias; p.residual = residual; p.y = y;
    p.N = N; p.Cin = Cin; p.H = H; p.W = W; p.Cout = Cout;
    p.Hin = upsample2x ? H / 2 : H; p.Win = upsample2x ? W / 2 : W;
    p.xPlane = (int)xPlane; p.yPlane = (int)yPlane; p.rPlane = (int)(residual ? rPlane : yPlane);
    p.xImage = xImage; p.yImage = yImage; p.rImage = rImage;
    p.cinPad = isrConvCinPad(Cin); p.coutPad = isrConvCoutPad(Cout);
    p.tilesX = (W + TW - 1) / TW; p.tilesY = (H + TH - 1) / TH;
    p.act = act; p.slope = slope;
    ISR_DIAG_SET(p.dbg, g_conv_dbg);
    ISR_DIAG_SET(p.stamps, g_conv_stamps);
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
    p.cgroups = 1;
    if (g_conv_algo == 1) {
        // two workgroups per CU, one 32-channel group each; the grid covers all groups of all tiles
        static bool attr2_done = false;
        if (!attr2_done) {
            (void)hipFuncSetAttribute((const void*)conv3x3_fwd2_kernel<true, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)conv_fwd2_lds_bytes<4>());
            (void)hipFuncSetAttribute((const void*)conv3x3_fwd2_kernel<false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)conv_fwd2_lds_bytes<4>());
            attr2_done = true;
        }
        p.co0 = 0;
        p.cgroups = p.coutPad / 32;
        if (nwg * p.cgroups > 0x7fffffffLL) return -1;
        // small problems: 4x32 tiles (one row per wave) when the 16x32 tiling would leave most of the 2 x #CU slots empty
        const long long tilesY4 = (H + 3) / 4, nwgSmall = (long long)N * p.tilesX * tilesY4 * p.cgroups;
        const bool small = (g_conv_tile == 1) || (g_conv_tile == 0 && nwg * p.cgroups < 384 && nwgSmall <= 0x7fffffffLL);
        // smaller still (at most one 4x32 workgroup per CU): one output row per workgroup, K split over its waves
        const long long nwgRow = (long long)N * H * p.tilesX * p.cgroups;
        const bool rows = !upsample2x && nwgRow <= 0x7fffffffLL &&
                          ((g_conv_tile == 3) || (g_conv_tile == 0 && small && nwgSmall <= g_row_threshold));
        if (rows) {
            hipEvent_t e0 = nullptr, e1 = nullptr;
            if (g_profile) {
                e0 = pool_event(); e1 = pool_event();
                g_records.push_back({ ROW_VARIANT, 2.0 * 9 * Cin * Cout * (double)N * H * W, e0, e1 });
            }
            ISR_LAUNCH(conv3x3_rowsplit_kernel, dim3((unsigned)nwgRow), block, 0, s, e0, e1, p);
            return hipGetLastError() == hipSuccess ? 0 : -2;
        }
        if (small) {
            p.tilesY = (int)tilesY4;
            hipEvent_t e0 = nullptr, e1 = nullptr;
            if (g_profile) {
                e0 = pool_event(); e1 = pool_event();
                g_records.push_back({ 10 + (upsample2x ? 1 : 0), 2.0 * 9 * Cin * Cout * (double)N * H * W, e0, e1 });
            }
            const dim3 grid1((unsigned)nwgSmall);
            if (upsample2x) ISR_LAUNCH((conv3x3_fwd2_kernel<true, 1>), grid1, block, conv_fwd2_lds_bytes<1>(), s, e0, e1, p);
            else ISR_LAUNCH((conv3x3_fwd2_kernel<false, 1>), grid1, block, conv_fwd2_lds_bytes<1>(), s, e0, e1, p);
            return hipGetLastError() == hipSuccess ? 0 : -2;
        }
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (g_profile) {
            e0 = pool_event(); e1 = pool_event();
            g_records.push_back({ 8 + (upsample2x ? 1 : 0), 2.0 * 9 * Cin * Cout * (double)N * H * W, e0, e1 });
        }
        const dim3 grid2((unsigned)(nwg * p.cgroups));
        if (upsample2x) ISR_LAUNCH((conv3x3_fwd2_kernel<true, 4>), grid2, block, conv_fwd2_lds_bytes<4>(), s, e0, e1, p);
        else ISR_LAUNCH((conv3x3_fwd2_kernel<false, 4>), grid2, block, conv_fwd2_lds_bytes<4>(), s, e0, e1, p);
        return hipGetLastError() == hipSuccess ? 0 : -2;
    }
    // one launch per group of up to 64 output channels (2 M tiles per wave)
    for (int co0 = 0; co0 < p.coutPad; co0 += 64) {
        p.co0 = co0;
        const int mt = (p.coutPad - co0 == 32) ? 1 : 2;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (g_profile) {
            e0 = pool_event(); e1 = pool_event();
            const int cg = Cout - co0 < 64 ? Cout - co0 : 64;
            g_records.push_back({ mt * 2 + (upsample2x ? 1 : 0), 2.0 * 9 * Cin * cg * (double)N * H * W, e0, e1 });
        }
        const size_t lds = mt == 1 ? conv_fwd_lds_bytes<1>() : conv_fwd_lds_bytes<2>();
        if (mt == 1) {
            if (upsample2x) ISR_LAUNCH((conv3x3_fwd_kernel<1, true>), grid, block, lds, s, e0, e1, p);
            else ISR_LAUNCH((conv3x3_fwd_kernel<1, false>), grid, block, lds, s, e0, e1, p);
        } else {
            if (upsample2x) ISR_LAUNCH((conv3x3_fwd_kernel<2, true>), grid, block, lds, s, e0, e1, p);
            else ISR_LAUNCH((conv3x3_fwd_kernel<2, false>), grid, block, lds, s, e0, e1, p);
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
    return (long long)WGRAD_MAX_SLABS * (9 * 64 * 64 + 64) * sizeof(float) + 4096;     // + partial maxima and the scale pair of the split kernel
}

int isrConvWeightGradMaxSegments(void) { return WG_MAX_SEG; }

int isrConv3x3WeightGradSegments(const float* const* xs, const float* const* gzs, int segments, float* dw, float* db, void* workspace,
                                 int N, int Cin, int H, int W, int Cout, void* stream)
{
    if (!xs || !gzs || segments <= 0 || segments > WG_MAX_SEG || !dw || !workspace || N <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0)
        return -1;
    hipStream_t s = (hipStream_t)stream;
    WGradParams p;
    ISR_DIAG_SET(p.dbg, g_conv_dbg);
    for (int k = 0; k < WG_MAX_SEG; ++k) {
        p.x[k] = k < segments ? xs[k] : nullptr;
        p.gz[k] = k < segments ? gzs[k] : nullptr;
        if (k < segments && (!xs[k] || !gzs[k])) return -1;
    }
    p.slabs = (float*)workspace;
    p.scale = nullptr;
    float* bslabs = p.slabs + (size_t)WGRAD_MAX_SLABS * 9 * 64 * 64;
    p.N = N; p.Cin = Cin; p.H = H; p.W = W; p.Cout = Cout;
    p.tilesX = (W + WG_TW - 1) / WG_TW; p.tilesY = (H + WG_TH - 1) / WG_TH;
    const long long nt = (long long)N * segments * p.tilesX * p.tilesY;
    if (nt > 0x7fffffffLL) return -1;
    p.ntiles = (int)nt;
    if ((long long)(Cin < 64 ? Cin : 64) * H * W * 4 > 0x7fffffffLL || (long long)(Cout < 64 ? Cout : 64) * H * W * 4 > 0x7fffffffLL) return -1;
    // one slab per workgroup; with many tiles one workgroup per CU (fewer slabs to reduce, same balance)
    const int G = p.ntiles >= 1024 ? 256 : (p.ntiles < WGRAD_MAX_SLABS ? p.ntiles : WGRAD_MAX_SLABS);
    const bool split = G <= 192;        // few tiles: one workgroup per row of the 3x3 instead of one per tile set
    for (int co0 = 0; co0 < Cout; co0 += 64)
        for (int ci0 = 0; ci0 < Cin; ci0 += 64) {
            p.co0 = co0; p.ci0 = ci0;
            p.bslabs = (db && ci0 == 0) ? bslabs : nullptr;
            const bool ksplit = !split && Cout - co0 <= 32 && G <= WGRAD_MAX_SLABS / 2;
            if (split) hipLaunchKernelGGL(conv3x3_wgrad_kernel<3>, dim3(3 * G), dim3(NTHREADS), 0, s, p);
            else if (ksplit) hipLaunchKernelGGL((conv3x3_wgrad_kernel<9, true>), dim3(G), dim3(NTHREADS), 0, s, p);
            else hipLaunchKernelGGL(conv3x3_wgrad_kernel<9>, dim3(G), dim3(NTHREADS), 0, s, p);
            hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(9 * 64 + 1), dim3(256), 0, s,
                               p.slabs, ksplit ? 2 * G : G, dw, Cout, Cin, co0, ci0, (const float*)p.bslabs, db);
        }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int isrConv3x3WeightGradSegmentsBf16(const float* const* xs, const float* const* gzs, int segments, float* dw, float* db, void* workspace,
                                     int N, int Cin, int H, int W, int Cout, void* stream)
{
    if (!xs || !gzs || segments <= 0 || segments > WG_MAX_SEG || !dw || !workspace || N <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0)
        return -1;
    if (W & 3) return -3;                          // the gz tile is fetched as aligned groups of four pixels
    hipStream_t s = (hipStream_t)stream;
    WGradParams p;
    ISR_DIAG_SET(p.dbg, g_conv_dbg);
    for (int k = 0; k < WG_MAX_SEG; ++k) {
        p.x[k] = k < segments ? xs[k] : nullptr;
        p.gz[k] = k < segments ? gzs[k] : nullptr;
        if (k < segments && (!xs[k] || !gzs[k] || ((uintptr_t)gzs[k] & 15))) return -1;
    }
    p.slabs = (float*)workspace;
    p.scale = nullptr;
    float* bslabs = p.slabs + (size_t)WGRAD_MAX_SLABS * 9 * 64 * 64;
    p.N = N; p.Cin = Cin; p.H = H; p.W = W; p.Cout = Cout;
    p.tilesX = (W + WG_TW - 1) / WG_TW; p.tilesY = (H + WG_TH - 1) / WG_TH;
    const long long nt = (long long)N * segments * p.tilesX * p.tilesY;
    if (nt > 0x7fffffffLL) return -1;
    p.ntiles = (int)nt;
    if ((long long)(Cin < 64 ? Cin : 64) * H * W * 4 > 0x7fffffffLL || (long long)(Cout < 64 ? Cout : 64) * H * W * 4 > 0x7fffffffLL) return -1;
    const int G = p.ntiles >= 512 ? 256 : (p.ntiles < WGRAD_MAX_SLABS ? p.ntiles : WGRAD_MAX_SLABS);
    for (int co0 = 0; co0 < Cout; co0 += 64)
        for (int ci0 = 0; ci0 < Cin; ci0 += 64) {
            p.co0 = co0; p.ci0 = ci0;
            p.bslabs = (db && ci0 == 0) ? bslabs : nullptr;
            hipLaunchKernelGGL(conv3x3_wgrad_bf16_kernel, dim3(G), dim3(NTHREADS), 0, s, p);
            hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(9 * 64 + 1), dim3(256), 0, s,
                               p.slabs, G, dw, Cout, Cin, co0, ci0, (const float*)p.bslabs, db);
        }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int isrConv3x3WeightGradSegmentsSplit(const float* const* xs, const float* const* gzs, int segments, float* dw, float* db, void* workspace,
                                      int N, int Cin, int H, int W, int Cout, void* stream)
{
    return isrConv3x3WeightGradSegmentsSplitMax(xs, gzs, nullptr, 0, segments, dw, db, workspace, N, Cin, H, W, Cout, stream);
}

static int g_wgrad_accumulate = 0;      // isrSetWeightGradAccumulate: taken (and cleared) by the next ...SegmentsSplit[Max] call
void isrSetWeightGradAccumulate(int bits) { g_wgrad_accumulate = bits & 3; }

int isrConv3x3WeightGradSegmentsSplitMax(const float* const* xs, const float* const* gzs, const void* const* gzmax, int maxWords, int segments,
                                         float* dw, float* db, void* workspace, int N, int Cin, int H, int W, int Cout, void* stream)
{
    if (!xs || !gzs || segments <= 0 || segments > WG_MAX_SEG || !dw || !workspace || N <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0)
        return -1;
    const int accumulate = g_wgrad_accumulate;
    g_wgrad_accumulate = 0;
    if (W & 3) return -3;                          // the gz tile is fetched as aligned groups of four pixels
    hipStream_t s = (hipStream_t)stream;
    WGradParams p;
    ISR_DIAG_SET(p.dbg, g_conv_dbg);
    AbsMaxParams ap;
    for (int k = 0; k < WG_MAX_SEG; ++k) {
        p.x[k] = k < segments ? xs[k] : nullptr;
        p.gz[k] = k < segments ? gzs[k] : nullptr;
        ap.t[k] = p.gz[k];
        if (k < segments && (!xs[k] || !gzs[k] || ((uintptr_t)gzs[k] & 15))) return -1;
    }
    ap.segments = segments;
    ap.count = (long long)N * Cout * H * W;
    p.slabs = (float*)workspace;
    float* bslabs = p.slabs + (size_t)WGRAD_MAX_SLABS * 9 * 64 * 64;
    float* partial = bslabs + (size_t)WGRAD_MAX_SLABS * 64;       // 512 partial maxima, then the scale pair
    float* scale = partial + 512;
    p.scale = scale;
    p.N = N; p.Cin = Cin; p.H = H; p.W = W; p.Cout = Cout;
    p.tilesX = (W + WG_TW - 1) / WG_TW; p.tilesY = (H + WG_TH - 1) / WG_TH;
    const long long nt = (long long)N * segments * p.tilesX * p.tilesY;
    if (nt > 0x7fffffffLL) return -1;
    p.ntiles = (int)nt;
    if ((long long)(Cin < 64 ? Cin : 64) * H * W * 4 > 0x7fffffffLL || (long long)(Cout < 64 ? Cout : 64) * H * W * 4 > 0x7fffffffLL) return -1;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)conv3x3_wgrad_split_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)conv3x3_wgrad_split2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, W2_LDS_BYTES);
        attr_done = true;
    }
    const long long quads = ap.count >> 2;
    const int AB = (int)(quads < 512LL * 256 ? (quads + 255) / 256 > 0 ? (quads + 255) / 256 : 1 : 512);
    if (gzmax) {
        MaxFlagParams mp;
        for (int k = 0; k < WG_MAX_SEG; ++k) {
            mp.f[k] = k < segments ? (const unsigned*)gzmax[k] : nullptr;
            if (k < segments && !gzmax[k]) return -1;
        }
        if (maxWords <= 0) return -1;
        mp.segments = segments; mp.words = maxWords;
        hipLaunchKernelGGL(wgrad_scale_flags_kernel, dim3(1), dim3(1024), 0, s, mp, scale);
    } else {
        hipLaunchKernelGGL(wgrad_absmax_kernel, dim3(AB), dim3(256), 0, s, ap, partial);
        hipLaunchKernelGGL(wgrad_scale_kernel, dim3(1), dim3(256), 0, s, (const float*)partial, AB, scale);
    }
    const int G = p.ntiles >= 512 ? 256 : (p.ntiles < WGRAD_MAX_SLABS ? p.ntiles : WGRAD_MAX_SLABS);
    for (int co0 = 0; co0 < Cout; co0 += 64)
        for (int ci0 = 0; ci0 < Cin; ci0 += 64) {
            p.co0 = co0; p.ci0 = ci0;
            p.bslabs = (db && ci0 == 0) ? bslabs : nullptr;
            // (dispatch-packet events when profiling is on, like the forward kernels: the training bench line's weight-gradient family)
            hipEvent_t pe0 = nullptr, pe1 = nullptr;
            isr_profile_record(ISR_VARIANT_WGRAD_SPLIT, 2.0 * 9 * (Cin - ci0 < 64 ? Cin - ci0 : 64) * (Cout - co0 < 64 ? Cout - co0 : 64) * (double)segments * N * H * W, &pe0, &pe1);
            if (g_wgrad_split_form == 2) {
                if (pe0 || pe1) hipExtLaunchKernelGGL(conv3x3_wgrad_split2_kernel, dim3(G), dim3(W2_THREADS), W2_LDS_BYTES, s, pe0, pe1, 0, p);
                else hipLaunchKernelGGL(conv3x3_wgrad_split2_kernel, dim3(G), dim3(W2_THREADS), W2_LDS_BYTES, s, p);
            } else {
                if (pe0 || pe1) hipExtLaunchKernelGGL(conv3x3_wgrad_split_kernel, dim3(G), dim3(NTHREADS), WS_LDS_BYTES, s, pe0, pe1, 0, p);
                else hipLaunchKernelGGL(conv3x3_wgrad_split_kernel, dim3(G), dim3(NTHREADS), WS_LDS_BYTES, s, p);
            }
            hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(9 * 64 + 1), dim3(256), 0, s,
                               p.slabs, G, dw, Cout, Cin, co0, ci0, (const float*)p.bslabs, db, (const float*)scale, accumulate);
        }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int isrConv3x3WeightGrad(const float* x, const float* gz, float* dw, float* db, void* workspace,
                         int N, int Cin, int H, int W, int Cout, void* stream)
{
    return isrConv3x3WeightGradSegments(&x, &gz, 1, dw, db, workspace, N, Cin, H, W, Cout, stream);
}

}  // extern "C"

// ---- small-Cout convolution on 4x4x1 MFMA blocks ---------------------------------------------------
// The last layer of EnhanceNet maps 64 -> 6 channels (enhancenet.py:124).  On the 32-row MFMA tile
// that wastes 26/32 of the matrix pipe.  v_mfma_f32_4x4x1_16B_f32 instead multiplies sixteen
// independent (4 x 1) x (1 x 4) blocks: with A = the weights of 4 output channels (the same in every
// block: lane l holds channel l % 4) and B = 64 consecutive pixels of one input row (lane l = pixel l),
// one instruction is D[ch][pixel] += w[ch] * x[pixel] for 4 channels x 64 pixels -- the conv's
// natural layout, no padding beyond 6 -> 8 channels, and the result registers are already
// [channel][64 contiguous pixels].  Same rate as every other f32 MFMA (512 flops / 8 cycles).
//
// Workgroup = 4 waves = 16 x 64 output pixels, wave w owns rows 4w..4w+3 and both channel groups
// (8 accumulators of 4 registers).  Input: 4-channel chunks of the haloed 18 x 66 patch, double
// buffered in LDS, the B operand is a plain ds_read_b32 of 64 consecutive floats (row rr, column
// lane + dx); each of those registers feeds every (dy, r) with dy + r = rr and both channel groups
// (2..6 MFMAs).  A operands: per input channel 18 values per lane, kept in LDS as
// [chunk][c][lane % 4][(dx, dy, group)] and read with ds_read_b128 (4 distinct addresses per wave).
// Same fused epilogue semantics as the big kernel (bias, activation, residual).
namespace {

constexpr int SM_TH = 16, SM_TW = 64;          // output tile
constexpr int SM_PH = SM_TH + 2, SM_PW = SM_TW + 2;
constexpr int SM_CK = 4;                       // input channels per LDS chunk
constexpr int SM_PLANE = SM_PH * SM_PW;        // 1188 floats
constexpr int SM_CHUNK = SM_CK * SM_PLANE;     // 4752
constexpr int SM_NEL = (SM_CHUNK + 255) / 256; // 19 elements per thread and chunk
constexpr int SM_PBUF = SM_NEL * 256;          // 4864: patch buffer, the tail is a sink for masked-off lanes
constexpr int SM_WQ = 20;                      // floats per (c, lane % 4): 18 weights [dx][dy][group] + 2 pad (16-byte rows)
constexpr int SM_WCH = SM_CK * 4 * SM_WQ;      // 320 weight floats per chunk = 80 float4
constexpr int SM_WBUF = 4 * 256;               // every thread writes one float4 (80 real ones)

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct SmallConvParams {
    const float* x; const float* wq;           // wq: [chunks][4][4][20], see prepare_weights_small_kernel
    const float* bias8; const float* residual; float* y;
    int N, Cin, H, W, Cout;
    int tilesX, tilesY;
    int act; float slope;
    long long xPlane, xImage;                  // channel / batch strides of x in floats (y and residual are packed)
    int finish;                                // 1: instead of storing y, finish the frame per pixel (fin; N == 1, Cout == 6)
    FinishParams fin;
};

__global__ __launch_bounds__(256, 2) void conv3x3_small_cout_kernel(const SmallConvParams p)
{
    __shared__ __attribute__((aligned(16))) float patch[2][SM_PBUF];
    __shared__ __attribute__((aligned(16))) float wl[2][SM_WBUF];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int tilesPerImage = p.tilesX * p.tilesY;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int n = bid / tilesPerImage;
    const int t = bid - n * tilesPerImage;
    const int ty = t / p.tilesX, tx = t - ty * p.tilesX;
    const int oy0 = ty * SM_TH, ox0 = tx * SM_TW;

    const int planeIn = (int)p.xPlane;
    const float* ximg = p.x + (size_t)n * p.xImage;
    const int nchunks = (p.Cin + SM_CK - 1) / SM_CK;

    unsigned plan[SM_NEL];                     // byte offsets inside a chunk's 4 planes (BAD_OFFSET -> 0)
#pragma unroll
    for (int i = 0; i < SM_NEL; ++i) {
        const int e = tid + i * 256;
        const int c = e / SM_PLANE, rem = e - c * SM_PLANE;
        const int r = rem / SM_PW, col = rem - r * SM_PW;
        const int gy = oy0 + r - 1, gx = ox0 + col - 1;
        const bool ok = e < SM_CHUNK && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
        plan[i] = ok ? (unsigned)((c * planeIn + gy * p.W + gx) * 4) : BAD_OFFSET;
    }
    auto chunk_rsrc = [&](int chunk) -> rsrc_t {
        const int left = p.Cin - chunk * SM_CK;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ximg + (size_t)chunk * SM_CK * planeIn), 0,
                                                 left > 0 ? left * planeIn * 4 : 0, 0x00020000);
    };
    const float4* wq4 = reinterpret_cast<const float4*>(p.wq) + min(tid, SM_WCH / 4 - 1);

    f32x4 acc[2][4];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[g][r] = (f32x4){ 0.f, 0.f, 0.f, 0.f };

    {   // prologue: chunk 0
        const rsrc_t rs = chunk_rsrc(0);
        float v[SM_NEL];
        const float4 wv = wq4[0];
#pragma unroll
        for (int i = 0; i < SM_NEL; ++i) v[i] = buf_load(rs, plan[i]);
#pragma unroll
        for (int i = 0; i < SM_NEL; ++i) patch[0][tid + i * 256] = v[i];
        reinterpret_cast<float4*>(wl[0])[tid] = wv;
    }
    __syncthreads();

    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int buf = chunk & 1;
        const bool more = chunk + 1 < nchunks;
        float sv[SM_NEL];
        float4 wv;
        if (more) {
            const rsrc_t rsn = chunk_rsrc(chunk + 1);
#pragma unroll
            for (int i = 0; i < SM_NEL; ++i) sv[i] = buf_load(rsn, plan[i]);
            wv = wq4[(chunk + 1) * (SM_WCH / 4)];
        }
        __builtin_amdgcn_sched_barrier(0);       // keep the loads above the MFMA block (hipcc would sink them to the stores)
        const float* pw = patch[buf] + (4 * wave) * SM_PW + lane;
        const float* aw = wl[buf] + (lane & 3) * SM_WQ;
#pragma unroll
        for (int c = 0; c < SM_CK; ++c) {
            float a[SM_WQ];
#pragma unroll
            for (int q = 0; q < SM_WQ / 4; ++q) {
                const float4 a4 = *reinterpret_cast<const float4*>(aw + c * 4 * SM_WQ + 4 * q);
                a[4 * q] = a4.x; a[4 * q + 1] = a4.y; a[4 * q + 2] = a4.z; a[4 * q + 3] = a4.w;
            }
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                float b[6];
#pragma unroll
                for (int rr = 0; rr < 6; ++rr) b[rr] = pw[c * SM_PLANE + rr * SM_PW + dx];
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int g = 0; g < 2; ++g)
                            acc[g][r] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[(dx * 3 + dy) * 2 + g], b[dy + r], acc[g][r], 0, 0, 0);
            }
        }
        if (more) {
#pragma unroll
            for (int i = 0; i < SM_NEL; ++i) patch[buf ^ 1][tid + i * 256] = sv[i];
            reinterpret_cast<float4*>(wl[buf ^ 1])[tid] = wv;
        }
        __syncthreads();
    }

    // D register i of group g = channel 4g + i, lane = pixel column
    const int ox = ox0 + lane;
    const size_t plane = (size_t)p.H * p.W;
    if (p.finish) {
        // the frame's last layer: clamp / normalise / shade right here instead of a 50 MB round trip through memory
        // and another launch (isrFinishFrame); the six channels of a pixel sit in this lane's accumulators
        if (ox < p.W) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int oy = oy0 + 4 * wave + r;
                if (oy >= p.H) break;
                float v[6];
#pragma unroll
                for (int co = 0; co < 6; ++co) v[co] = acc[co >> 2][r][co & 3] + p.bias8[co];
                isr_finish_pixel(p.fin, ox, oy, v);
            }
        }
        return;
    }
    if (ox < p.W) {
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int co = 4 * g + i;
                if (co >= p.Cout) break;
                const float bias = p.bias8[co];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int oy = oy0 + 4 * wave + r;
                    if (oy >= p.H) break;
                    float v = acc[g][r][i] + bias;
                    if (p.act == ISR_ACT_RELU) v = v > 0.f ? v : 0.f;
                    else if (p.act == ISR_ACT_LEAKY) v = v > 0.f ? v : v * p.slope;
                    const size_t o = ((size_t)n * p.Cout + co) * plane + (size_t)oy * p.W + ox;
                    if (p.residual) v += p.residual[o];
                    p.y[o] = v;
                }
            }
    }
}

// wq[chunk][c][q][k]: k = (dx*3 + dy)*2 + g < 18 -> w[4g + q][chunk*4 + c][dy][dx], zero padded
__global__ void prepare_weights_small_kernel(const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ wq,
                                             float* __restrict__ bias8, int Cout, int Cin)
{
    const int nchunks = (Cin + SM_CK - 1) / SM_CK;
    const int total = nchunks * SM_WCH;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int chunk = e / SM_WCH, rem = e - chunk * SM_WCH;
        const int c = rem / (4 * SM_WQ), q = (rem / SM_WQ) & 3, k = rem % SM_WQ;
        const int dx = k / 6, dy = (k % 6) >> 1, g = k & 1;
        const int co = 4 * g + q, ci = chunk * SM_CK + c;
        wq[e] = (k < 18 && co < Cout && ci < Cin) ? w[((co * Cin + ci) * 3 + dy) * 3 + dx] : 0.f;
    }
    if (blockIdx.x == 0 && threadIdx.x < 8) bias8[threadIdx.x] = (bias && (int)threadIdx.x < Cout) ? bias[threadIdx.x] : 0.f;
}

}  // namespace

extern "C" {

int isrConvSmallCinPad(int Cin) { return ((Cin + 7) / 8) * 8; }

long long isrConvSmallWeightFloats(int Cin) { return (long long)((Cin + SM_CK - 1) / SM_CK) * SM_WCH; }

int isrConvSmallPrepare(const float* w, const float* bias, float* w8, float* bias8, int Cout, int Cin, void* stream)
{
    if (!w || !w8 || !bias8 || Cout <= 0 || Cout > 8 || Cin <= 0) return -1;
    const int total = (int)isrConvSmallWeightFloats(Cin);
    hipLaunchKernelGGL(prepare_weights_small_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, bias, w8, bias8, Cout, Cin);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int isrConv3x3SmallCout(const float* x, const float* w8, const float* bias8, const float* residual, float* y,
                        int N, int Cin, int H, int W, int Cout, int act, float slope, void* stream)
{
    return isrConv3x3SmallCoutStrided(x, w8, bias8, residual, y, N, Cin, H, W, Cout, act, slope,
                                      (long long)H * W, (long long)Cin * H * W, stream);
}

int isrConv3x3SmallCoutStrided(const float* x, const float* w8, const float* bias8, const float* residual, float* y,
                               int N, int Cin, int H, int W, int Cout, int act, float slope,
                               long long xPlane, long long xImage, void* stream)
{
    if (!x || !w8 || !bias8 || !y || N <= 0 || Cin <= 0 || Cout <= 0 || Cout > 8 || H <= 0 || W <= 0) return -1;
    if (xPlane < (long long)H * W || (long long)Cin * xPlane * 4 >= (1LL << 31)) return -1;
    if (act < ISR_ACT_NONE || act > ISR_ACT_LEAKY) return -1;
    SmallConvParams p;
    p.x = x; p.wq = w8; p.bias8 = bias8; p.residual = residual; p.y = y;
    p.N = N; p.Cin = Cin; p.H = H; p.W = W; p.Cout = Cout;
    p.tilesX = (W + SM_TW - 1) / SM_TW; p.tilesY = (H + SM_TH - 1) / SM_TH;
    p.act = act; p.slope = slope;
    p.xPlane = xPlane; p.xImage = xImage;
    p.finish = 0;
    const long long nwg = (long long)N * p.tilesX * p.tilesY;
    if (nwg > 0x7fffffffLL) return -1;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (g_profile) {
        e0 = pool_event(); e1 = pool_event();
        g_records.push_back({ 6, 2.0 * 9 * Cin * Cout * (double)N * H * W, e0, e1 });
    }
    ISR_LAUNCH(conv3x3_small_cout_kernel, dim3((unsigned)nwg), dim3(256), 0, (hipStream_t)stream, e0, e1, p);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int isrConvSmallFinishFrame(const float* x, const float* w8, const float* bias8, const float* net_input, float* next_prev, float* rgb,
                            int Cin, int h, int w, long long xPlane, const float* shading24, int exponent, float ao_strength,
                            int inverse_ao, int enable_specular, void* stream)
{
    if (!x || !w8 || !bias8 || !net_input || !next_prev || Cin <= 0 || h <= 0 || w <= 0 || (rgb && !shading24)) return -1;
    const int H = 4 * h, W = 4 * w;
    if (xPlane < (long long)H * W || (long long)Cin * xPlane * 4 >= (1LL << 31)) return -1;
    SmallConvParams p;
    p.x = x; p.wq = w8; p.bias8 = bias8; p.residual = nullptr; p.y = nullptr;
    p.N = 1; p.Cin = Cin; p.H = H; p.W = W; p.Cout = 6;
    p.tilesX = (W + SM_TW - 1) / SM_TW; p.tilesY = (H + SM_TH - 1) / SM_TH;
    p.act = ISR_ACT_NONE; p.slope = 0.f;
    p.xPlane = xPlane; p.xImage = (long long)Cin * xPlane;
    p.finish = 1;
    isr_fill_finish_params(p.fin, nullptr, net_input, next_prev, rgb, h, w, shading24, exponent, ao_strength, inverse_ao, enable_specular);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (g_profile) {
        e0 = pool_event(); e1 = pool_event();
        g_records.push_back({ 6, 2.0 * 9 * Cin * 6 * (double)H * W, e0, e1 });
    }
    ISR_LAUNCH(conv3x3_small_cout_kernel, dim3((unsigned)(p.tilesX * p.tilesY)), dim3(256), 0, (hipStream_t)stream, e0, e1, p);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // extern "C"
