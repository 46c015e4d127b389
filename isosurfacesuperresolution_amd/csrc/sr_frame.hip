// Frame-assembly kernels around the EnhanceNet convolutions (inference path): everything the
// reference does with ~100 small PyTorch launches per frame between the renderer and the first
// convolution, and between the last convolution and the displayed image, in two passes over memory.
//
//  isrAssembleInput  = LoadedModel.inference's input assembly (inference/loadedmodel.py:84-118):
//                      (mask*2-1, normal, depth) from the HWC G-buffer, VideoTools.warp_upscale of
//                      the previous high-res frame by the (hole-filled) low-res flow
//                      (models/videotools.py:51-87) and VideoTools.flatten_high (:8-25).
//  isrFinishFrame    = EnhanceNet._recon_image's residual (models/enhancenet.py:51-90), the
//                      clamp / normalise of mainGUI.py:594-599 and ScreenSpaceShading.forward
//                      (utils/shading.py:148-191).
//
// Both are HBM-streaming kernels (each ~100-125 MB of traffic at 1080p).
#include "sr_diag.h"
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "../../include/isr_sr_kernels.h"
#include "sr_finish.h"
#include "sr_profile.h"
#include "sr_warp_exact.h"
#include "sr_split_common.h"

namespace {

__device__ __forceinline__ void src_index(int dst, float scale, int in_size, int& i0, int& i1, float& l1) { isr_src_index_rn(dst, scale, in_size, i0, i1, l1); }
__device__ __forceinline__ float bilerp_rn(float hy, float hx, float ly, float lx, float a, float b, float c, float d) { return isr_bilerp_rn(hy, hx, ly, lx, a, b, c, d); }
__device__ __forceinline__ float pixel_grid(int i, int n) { return isr_pixel_grid(i, n); }

struct AssembleParams {
    const float* gbuf;      // [h][w][12]
    const float* flow;      // [2][h][w] hole-filled low-res flow (may be NULL when prev == NULL)
    const float* prev;      // [6][4h][4w] previous output or NULL
    float* out;             // [101][h][w]
    int h, w;
    int init_mode;          // prev == NULL: 0 zero, 1 unshaded constants, 2 upsampled input (+ones)
    int ao_inverted;
    int row0;               // first row of the launch (isrAssembleInputRows: a rank that needs only its strip + halo)
    int col0, col1;         // columns [col0, col1) of the launch (isrAssembleInputRect: a screen TILE + halo)
    // PACK (isrAssembleInputPacked): the input goes straight into the dataflow trunk's workspace, PACKED-SPLIT (sr_conv_trunk.hip) --
    // [hi | lo'][groups][psPlane units of 8 channels] -- and only channels 0 .. 4 (what the frame's finishing reads) to `out` as well
    u32x4* ps; int groups; long long psPlane;
    u32x4* fps; u32x4* tps;          // the launch's other two packed-split tensors: their planes' zero units are (re)written here
    unsigned* done; int ntiles;      // the tiles' progress counters: back to zero for the trunk launch that follows
};

constexpr int ASM_STRIDE = 105;      // floats per pixel of the PACK form's LDS tile (odd: lanes = pixels read conflict-free)

// one thread per (low-res pixel, dx): neighbouring lanes read neighbouring hi-res columns (the gathers of the
// previous frame coalesce; one thread per low-res pixel read 4x its algorithmic bytes); loops over dy and 6 channels.
// PACK: the 64 pixels x 101 channels of the workgroup are collected in LDS and leave as packed-split units (the same split16x
// the trunk's own packing pass applied: bit-identical operands), one pass over memory less and one kernel boundary less per frame.
template <bool PACK>
__device__ __forceinline__ void assemble_pixel(const AssembleParams& p, int x, int y, int dx, float* vals)
{
    const size_t plane = (size_t)p.h * p.w;
    const size_t pix = (size_t)y * p.w + x;
    const float* g = p.gbuf + pix * 12;
    const float4 g0 = *reinterpret_cast<const float4*>(g);        // r g b mask
    const float4 g1 = *reinterpret_cast<const float4*>(g + 4);    // nx ny nz depth
    if (dx == 0) {
        const float cur[5] = { g0.w * 2.0f - 1.0f, g1.x, g1.y, g1.z, g1.w };
#pragma unroll
        for (int c = 0; c < 5; ++c) {
            p.out[c * plane + pix] = cur[c];
            if (PACK) vals[c] = cur[c];
        }
    }
    const int H = 4 * p.h, W = 4 * p.w;
    const size_t hplane = (size_t)H * W;
    float* const op = p.out + 5 * plane + pix;
    // flattened channel f = c * 16 + dy * 4 + dx of the previous frame -> network channel 5 + f
    auto put = [&](int f, float v) {
        if (PACK) vals[5 + f] = v;
        else op[(size_t)f * plane] = v;
    };
    if (!p.prev) {
        if (p.init_mode == 0) {
            for (int c = 0; c < 6; ++c)
                for (int dy = 0; dy < 4; ++dy) put(c * 16 + dy * 4 + dx, 0.0f);
        } else if (p.init_mode == 1) {
            const float defaults[6] = { -1.f, 0.f, 0.f, 1.f, 0.5f, p.ao_inverted ? 0.f : 1.f };
            for (int c = 0; c < 6; ++c)
                for (int dy = 0; dy < 4; ++dy) put(c * 16 + dy * 4 + dx, defaults[c]);
        } else {
            // "input": bilinear x4 of (mask*2-1, normal, depth), remaining channel = 1
            for (int dy = 0; dy < 4; ++dy) {
                    int y0, y1, x0, x1; float ly, lx;
                    src_index(4 * y + dy, 0.25f, p.h, y0, y1, ly);
                    src_index(4 * x + dx, 0.25f, p.w, x0, x1, lx);
                    const float hy = 1.f - ly, hx = 1.f - lx;
                    const float* a = p.gbuf + ((size_t)y0 * p.w + x0) * 12;
                    const float* b = p.gbuf + ((size_t)y0 * p.w + x1) * 12;
                    const float* c_ = p.gbuf + ((size_t)y1 * p.w + x0) * 12;
                    const float* d = p.gbuf + ((size_t)y1 * p.w + x1) * 12;
                    for (int c = 0; c < 5; ++c) {
                        const int ch = 3 + c;
                        float va = a[ch], vb = b[ch], vc = c_[ch], vd = d[ch];
                        if (c == 0) { va = va * 2.f - 1.f; vb = vb * 2.f - 1.f; vc = vc * 2.f - 1.f; vd = vd * 2.f - 1.f; }
                        put(c * 16 + dy * 4 + dx, hy * (hx * va + lx * vb) + ly * (hx * vc + lx * vd));
                    }
                    put(5 * 16 + dy * 4 + dx, 1.0f);
                }
        }
        return;
    }
    const float* fx = p.flow;
    const float* fy = p.flow + plane;
    const float sx_scale = 0.5f * (float)(W - 1), sy_scale = 0.5f * (float)(H - 1);
    for (int dy = 0; dy < 4; ++dy) {
#pragma clang fp contract(off)
        const int Y = 4 * y + dy;
        int y0, y1; float ly;
        src_index(Y, 0.25f, p.h, y0, y1, ly);
        {
            const int X = 4 * x + dx;
            int x0, x1; float lx;
            src_index(X, 0.25f, p.w, x0, x1, lx);
            const float hy = 1.f - ly, hx = 1.f - lx;
            // flow scaled by (-2, +2) (exact) then bilinearly upsampled (videotools.py:65-70)
            const float flx = bilerp_rn(hy, hx, ly, lx, fx[y0 * p.w + x0] * -2.0f, fx[y0 * p.w + x1] * -2.0f, fx[y1 * p.w + x0] * -2.0f, fx[y1 * p.w + x1] * -2.0f);
            const float fly = bilerp_rn(hy, hx, ly, lx, fy[y0 * p.w + x0] * 2.0f, fy[y0 * p.w + x1] * 2.0f, fy[y1 * p.w + x0] * 2.0f, fy[y1 * p.w + x1] * 2.0f);
            // grid = linspace(-1, 1)[X] + flow ; sample position with align_corners=True
            const float gx = pixel_grid(X, W) + flx;
            const float gy = pixel_grid(Y, H) + fly;
            const float gx1 = gx + 1.0f, gy1 = gy + 1.0f;
            const float sx = gx1 * sx_scale, sy = gy1 * sy_scale;
            const float fx0 = floorf(sx), fy0 = floorf(sy);
            const int ix0 = (int)fminf(fmaxf(fx0, -2.f), (float)W), iy0 = (int)fminf(fmaxf(fy0, -2.f), (float)H);
            const float wx1 = sx - fx0, wy1 = sy - fy0;
            const float wx0 = 1.f - wx1, wy0 = 1.f - wy1;
            const bool vx0 = (unsigned)ix0 < (unsigned)W, vx1 = (unsigned)(ix0 + 1) < (unsigned)W;
            const bool vy0 = (unsigned)iy0 < (unsigned)H, vy1 = (unsigned)(iy0 + 1) < (unsigned)H;
            const float w00 = wx0 * wy0, w01 = wx1 * wy0, w10 = wx0 * wy1, w11 = wx1 * wy1;
            const long long b00 = (long long)iy0 * W + ix0;
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                const float* q = p.prev + (size_t)c * hplane;
                float v00 = (vy0 && vx0) ? q[b00] : 0.f, v01 = (vy0 && vx1) ? q[b00 + 1] : 0.f;
                float v10 = (vy1 && vx0) ? q[b00 + W] : 0.f, v11 = (vy1 && vx1) ? q[b00 + W + 1] : 0.f;
                if (c == 0) {   // special mask: [-1,1] -> [0,1] before sampling, back after (zero padding == -1)
                    const float h00 = v00 * 0.5f, h01 = v01 * 0.5f, h10 = v10 * 0.5f, h11 = v11 * 0.5f;
                    v00 = (vy0 && vx0) ? h00 + 0.5f : 0.f; v01 = (vy0 && vx1) ? h01 + 0.5f : 0.f;
                    v10 = (vy1 && vx0) ? h10 + 0.5f : 0.f; v11 = (vy1 && vx1) ? h11 + 0.5f : 0.f;
                }
                // ((v00 w00 + v01 w01) + v10 w10) + v11 w11, one rounding per operation
                const float t00 = v00 * w00, t01 = v01 * w01, t10 = v10 * w10, t11 = v11 * w11;
                float r = t00 + t01;
                r = r + t10;
                r = r + t11;
                if (c == 0) { r = r * 2.0f; r = r - 1.0f; }
                put(c * 16 + dy * 4 + dx, r);
            }
        }
    }
}

template <bool PACK>
__global__ __launch_bounds__(256) void assemble_input_kernel(const AssembleParams p)
{
    __shared__ float tile[PACK ? 64 * ASM_STRIDE : 1];
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int x = p.col0 + (t >> 2), dx = t & 3;
    const int y = p.row0 + blockIdx.y;
    if (x < p.col1) assemble_pixel<PACK>(p, x, y, dx, tile + (threadIdx.x >> 2) * ASM_STRIDE);
    if (!PACK) return;
    const u32x4 zero = {0u, 0u, 0u, 0u};
    if (blockIdx.x == 0 && blockIdx.y == 0) {
        // what trunk_pack_input_kernel does besides packing: the zero unit that ends every plane of the launch's three packed-split
        // tensors, and the tiles' progress counters back to zero
        const size_t npix = (size_t)p.h * p.w;
        for (int i = threadIdx.x; i < p.ntiles; i += 256) p.done[i] = 0u;
        for (int i = threadIdx.x; i < 2 * p.groups; i += 256) p.ps[(size_t)i * p.psPlane + npix] = zero;
        for (int i = threadIdx.x; i < 16; i += 256) { p.fps[(size_t)i * p.psPlane + npix] = zero; p.tps[(size_t)i * p.psPlane + npix] = zero; }
    }
    __syncthreads();
    const int xb = p.col0 + blockIdx.x * 64;
    for (int it = threadIdx.x; it < 64 * p.groups; it += 256) {
        const int pl = it & 63, g = it >> 6;
        if (xb + pl >= p.col1) continue;
        const float* v = tile + pl * ASM_STRIDE + g * 8;
        f16x8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            _Float16 a, b;
            split16x(g * 8 + e < 101 ? v[e] : 0.0f, a, b);
            hi[e] = a; lo[e] = b;
        }
        const size_t pix = (size_t)y * p.w + xb + pl;
        p.ps[(size_t)g * p.psPlane + pix] = __builtin_bit_cast(u32x4, hi);
        p.ps[(size_t)(p.groups + g) * p.psPlane + pix] = __builtin_bit_cast(u32x4, lo);
    }
}

__global__ __launch_bounds__(256) void finish_frame_kernel(const FinishParams p)
{
    const int X = blockIdx.x * blockDim.x + threadIdx.x;
    const int Y = blockIdx.y;
    const int H = 4 * p.h, W = 4 * p.w;
    if (X >= W) return;
    const size_t hplane = (size_t)H * W;
    const size_t pix = (size_t)Y * W + X;
    float v[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) v[c] = p.raw[(size_t)c * hplane + pix];
    isr_finish_pixel(p, X, Y, v);
}


// ---- flow hole filling: push-pull pyramid in ONE launch -----------------------------------------
// Same arithmetic as inference/flowfill.py (mask-weighted 2x2 pulls down to 1x1, bilinear pushes
// back up, known pixels kept), which the PyTorch version spends ~150 tiny launches on.  The whole
// pyramid of a 480x270 frame is 1.4 MB and L2 resident, so a single 1024-thread workgroup walks
// it level by level with __syncthreads() in between -- no inter-kernel gaps, no grid sync.
struct FillLevel { int h, w; size_t off; };     // v: [2][h][w] at off, m: [h][w] at off + 2*h*w

// hy (hx a + lx b) + ly (hx c + lx d) with the roundings spelled out (seven, no FMA: inference/flowfill.py defines the fill in
// elementwise operations and both forms of the kernel -- three launches / one -- produce its bits)
__device__ __forceinline__ float fill_bilerp(float hy, float hx, float ly, float lx, float a, float b, float c, float d)
{
    return bilerp_rn(hy, hx, ly, lx, a, b, c, d);
}

// mode 0: pull level 0 -> 1 over the whole grid; mode 1: one workgroup pulls levels 2..n and pushes
// back down to level 1; mode 2: push level 1 -> 0 over the whole grid (the two big levels are
// bandwidth work for every CU, the small ones latency work for one).
__global__ __launch_bounds__(1024) void flow_fill_kernel(const float* __restrict__ gbuf, float* __restrict__ out,
                                                          float* __restrict__ ws, int h, int w, int mode)
{
    __shared__ FillLevel lv[20];
    __shared__ int nlev;
    const int tid = threadIdx.x;
    const int gtid = blockIdx.x * blockDim.x + tid, gstride = gridDim.x * blockDim.x;
    if (mode == 3) {                                  // a 1 x 1 image is its own coarsest level: flow * valid (flowfill.py: levels[-1])
        if (gtid == 0) { const bool known = gbuf[3] != 0.f; out[0] = known ? gbuf[8] : 0.f; out[1] = known ? gbuf[9] : 0.f; }
        return;
    }
    if (tid == 0) {
        int ch = h, cw = w, n = 0;
        size_t off = 0;
        lv[0].h = h; lv[0].w = w; lv[0].off = 0;      // level 0 lives in gbuf / out, not in ws
        while (!(ch <= 1 && cw <= 1) && n < 18) {
            ch = (ch + 1) / 2; cw = (cw + 1) / 2;
            ++n;
            lv[n].h = ch; lv[n].w = cw; lv[n].off = off;
            off += (size_t)3 * ch * cw;
        }
        nlev = n;
    }
    __syncthreads();
    // pull
    const int pull_lo = mode == 0 ? 1 : 2, pull_hi = mode == 0 ? 1 : (mode == 1 ? nlev : 0);
    for (int l = pull_lo; l <= pull_hi; ++l) {
        const int ph = lv[l - 1].h, pw = lv[l - 1].w, ch = lv[l].h, cw = lv[l].w;
        float* v = ws + lv[l].off;
        float* m = v + (size_t)2 * ch * cw;
        const float* pv = ws + lv[l - 1].off;
        const float* pm = pv + (size_t)2 * ph * pw;
        for (int e = gtid; e < ch * cw; e += gstride) {
            const int y = e / cw, x = e - y * cw;
            float sm = 0.f, sx = 0.f, sy = 0.f;
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const int yy = 2 * y + dy, xx = 2 * x + dx;
                    if (yy < ph && xx < pw) {
                        if (l == 1) {
                            const float* g = gbuf + ((size_t)yy * pw + xx) * 12;
                            const float valid = g[3] != 0.f ? 1.f : 0.f;
                            sm += valid; sx += g[8] * valid; sy += g[9] * valid;
                        } else {
                            sm += pm[yy * pw + xx]; sx += pv[yy * pw + xx]; sy += pv[ph * pw + yy * pw + xx];
                        }
                    }
                }
            const float ms = sm * 0.25f, vx = sx * 0.25f, vy = sy * 0.25f;     // avg_pool2d over the zero-padded 2x2
            const float den = ms > 1e-12f ? ms : 1e-12f;
            v[e] = ms > 0.f ? vx / den : 0.f;
            v[ch * cw + e] = ms > 0.f ? vy / den : 0.f;
            m[e] = ms > 0.f ? 1.f : 0.f;
        }
        __syncthreads();
    }
    // push: the coarsest level is complete; every finer level keeps its known pixels
    const int push_hi = mode == 1 ? nlev - 1 : (mode == 2 ? 0 : -1), push_lo = mode == 1 ? 1 : 0;
    for (int l = push_hi; l >= push_lo; --l) {
        const int fh = lv[l].h, fw = lv[l].w, ch = lv[l + 1].h, cw = lv[l + 1].w;
        const float* cv = ws + lv[l + 1].off;
        float* fv = l == 0 ? out : ws + lv[l].off;
        const float* fm = l == 0 ? nullptr : fv + (size_t)2 * fh * fw;
        const float sy_ = (float)ch / (float)fh, sx_ = (float)cw / (float)fw;
        for (int e = gtid; e < fh * fw; e += gstride) {
            const int y = e / fw, x = e - y * fw;
            bool known; float kx = 0.f, ky = 0.f;
            if (l == 0) {
                const float* g = gbuf + (size_t)e * 12;
                known = g[3] != 0.f; kx = g[8]; ky = g[9];
            } else {
                known = fm[e] > 0.f; kx = fv[e]; ky = fv[fh * fw + e];
            }
            float ox = kx, oy = ky;
            if (!known) {
                int y0, y1, x0, x1; float ly, lx;
                src_index(y, sy_, ch, y0, y1, ly);
                src_index(x, sx_, cw, x0, x1, lx);
                const float hy = 1.f - ly, hx = 1.f - lx;
                ox = fill_bilerp(hy, hx, ly, lx, cv[y0 * cw + x0], cv[y0 * cw + x1], cv[y1 * cw + x0], cv[y1 * cw + x1]);
                const float* c2 = cv + ch * cw;
                oy = fill_bilerp(hy, hx, ly, lx, c2[y0 * cw + x0], c2[y0 * cw + x1], c2[y1 * cw + x0], c2[y1 * cw + x1]);
            }
            fv[e] = ox; fv[fh * fw + e] = oy;
        }
        __syncthreads();
    }
}


// ---- the same pyramid in ONE launch of one workgroup per 64 x 64 tile --------------------------------------------------------
// The 2 x 2 pulls are aligned, so a 64 x 64 tile owns its part of levels 1 .. 6 outright: phase A pulls them in LDS (and writes them
// to the workspace, where neighbours read their halos later).  What is left above level 6 is one value per tile: the workgroup that
// arrives last (a ticket) pulls those few levels to 1 x 1 and pushes them back down to level 6 in LDS, then raises a flag.  Phase B
// pushes from level 6 back to level 0 per tile, on regions that grow by the bilinear footprint at every coarser level (1 + half the
// finer halo: 5 x 5 tiles' values at level 6, 34 x 34 at level 1), recomputing the halo instead of waiting for neighbours level by
// level.  Per pixel and level the arithmetic is flow_fill_kernel's, expression for expression: the results are bit-identical.
// Hand-off between workgroups (MI355X_MICROARCH.md, inter-workgroup visibility): agent-scope (sc1, write-through) stores, every wave
// drains them, barrier, ONE lane takes the ticket / raises the flag; readers poll with agent-scope loads and load the handed-off
// values with agent-scope loads.  Every workgroup must get a slot while others spin: the host refuses more than 256 tiles, and the
// spin has a deadline on the 100 MHz clock -- a launch that runs into it reports through the error word and never hangs.
constexpr int F1_TILE = 64, F1_LEVELS = 6, F1_THREADS = 256;
constexpr int F1_TOP_MAX = 1024;                  // tiles (= cells of level 6) the last workgroup takes in LDS
constexpr int F1_SMEM = 8192;                     // floats of LDS: phase A 3 x 1365, top 3 x ~1400, phase B 2 x 2444

struct FillOneParams {
    const float* gbuf; float* out; float* ws;     // as flow_fill_kernel
    float* top;                                   // [2][tiles] filled values of the tiles' level (6, or the last if there are fewer)
    unsigned* sync;                               // [0] ticket (0 .. ntiles - 1 inside a launch, zero between launches), [1] flag = number of completed launches (wraps)
    unsigned* error;                              // set to 1 by a launch that gave up waiting
    int h, w, tilesX, tilesY;
    unsigned long long timeoutTicks;
    ISR_DIAG_MEMBER(int, fault, 0);                                    // diagnostics (isrDebugSetFlowFillFault): the last workgroup never raises the flag
};

__device__ __forceinline__ void st_agent(float* q, float v) { __hip_atomic_store(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_agent(const float* q) { return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ __launch_bounds__(F1_THREADS) void flow_fill_one_kernel(const FillOneParams p)
{
    __shared__ FillLevel lv[20];
    __shared__ int nlev, s_last, s_ok;
    __shared__ unsigned s_epoch;
    __shared__ int offs[20];                       // last workgroup: LDS offset of level l
    __shared__ int ya[F1_LEVELS + 1], yb[F1_LEVELS + 1], xa[F1_LEVELS + 1], xb[F1_LEVELS + 1], fo[F1_LEVELS + 2];     // phase B regions
    __shared__ float smem[F1_SMEM];
    const int tid = threadIdx.x;
    const int tx = blockIdx.x % p.tilesX, ty = blockIdx.x / p.tilesX;
    const int ntiles = p.tilesX * p.tilesY;
    const int h = p.h, w = p.w;
    if (tid == 0) {
        int ch = h, cw = w, n = 0;
        size_t off = 0;
        lv[0].h = h; lv[0].w = w; lv[0].off = 0;
        while (!(ch <= 1 && cw <= 1) && n < 18) {
            ch = (ch + 1) / 2; cw = (cw + 1) / 2;
            ++n;
            lv[n].h = ch; lv[n].w = cw; lv[n].off = off;
            off += (size_t)3 * ch * cw;
        }
        nlev = n;
    }
    __syncthreads();
    const int L = nlev < F1_LEVELS ? nlev : F1_LEVELS;               // the tiles' own levels: 1 .. L

    // ---- phase A: pull levels 1 .. L of this tile -------------------------------------------------------------------------------
    {
        int base = 0, pbase = 0;                                     // LDS: level l at smem[base ..]: vx | vy | m, side x side each
        for (int l = 1; l <= L; ++l) {
            const int side = F1_TILE >> l, pside = side * 2;
            const int ph = lv[l - 1].h, pw = lv[l - 1].w, ch = lv[l].h, cw = lv[l].w;
            float* const v = p.ws + lv[l].off;
            float* const m = v + (size_t)2 * ch * cw;
            float* const mine = smem + base;
            const float* const prev = smem + pbase;
            for (int e = tid; e < side * side; e += F1_THREADS) {
                const int ly = e / side, lx = e - ly * side;
                const int y = ty * side + ly, x = tx * side + lx;
                float rx = 0.f, ry = 0.f, rm = 0.f;
                if (y < ch && x < cw) {
                    float sm = 0.f, sx = 0.f, sy = 0.f;
#pragma unroll
                    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                        for (int dx = 0; dx < 2; ++dx) {
                            const int yy = 2 * y + dy, xx = 2 * x + dx;
                            if (yy < ph && xx < pw) {
                                if (l == 1) {
                                    const float* g = p.gbuf + ((size_t)yy * pw + xx) * 12;
                                    const float valid = g[3] != 0.f ? 1.f : 0.f;
                                    sm += valid; sx += g[8] * valid; sy += g[9] * valid;
                                } else {
                                    const int q = (2 * ly + dy) * pside + 2 * lx + dx;
                                    sm += prev[2 * pside * pside + q]; sx += prev[q]; sy += prev[pside * pside + q];
                                }
                            }
                        }
                    const float ms = sm * 0.25f, vx = sx * 0.25f, vy = sy * 0.25f;
                    const float den = ms > 1e-12f ? ms : 1e-12f;
                    rx = ms > 0.f ? vx / den : 0.f;
                    ry = ms > 0.f ? vy / den : 0.f;
                    rm = ms > 0.f ? 1.f : 0.f;
                    const int ge = y * cw + x;
                    st_agent(v + ge, rx); st_agent(v + ch * cw + ge, ry); st_agent(m + ge, rm);
                }
                mine[e] = rx; mine[side * side + e] = ry; mine[2 * side * side + e] = rm;
            }
            __syncthreads();
            pbase = base; base += 3 * side * side;
        }
    }
    // ---- ticket: the last workgroup finishes the pyramid above level L ---------------------------------------------------------
    // The launch's epoch is the flag's value when the launch began + 1 (read BEFORE this workgroup's ticket: the flag only moves after
    // the last ticket of a launch has been taken); the ticket counts 0 .. ntiles - 1 within ONE launch and is put back to zero by the
    // last workgroup.  Nothing here grows without bound: a viewer that runs for 2^32 launches wraps the flag, and the signed
    // comparison below is wrap-safe (the earlier form derived epoch and "last" from one ever-growing ticket, which lost its
    // alignment at the wrap unless ntiles was a power of two).
    if (tid == 0) s_epoch = __hip_atomic_load(p.sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(p.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = old == (unsigned)(ntiles - 1);
    }
    __syncthreads();
    const unsigned epoch = s_epoch;
    if (s_last) {
        // levels L .. nlev in LDS: level l at tl[toff(l) ..]: vx | vy | m, lv[l].h x lv[l].w each
        float* const tl = smem;
        if (tid == 0) {
            int o = 0;
            for (int l = L; l <= nlev; ++l) { offs[l] = o; o += 3 * lv[l].h * lv[l].w; }
        }
        __syncthreads();
        {
            const int ch = lv[L].h, cw = lv[L].w, cells = ch * cw;
            const float* const v = p.ws + lv[L].off;
            for (int e = tid; e < 3 * cells; e += F1_THREADS) tl[offs[L] + e] = ld_agent(v + e);      // vx | vy | m are contiguous in the workspace too
        }
        __syncthreads();
        for (int l = L + 1; l <= nlev; ++l) {
            const int ph = lv[l - 1].h, pw = lv[l - 1].w, ch = lv[l].h, cw = lv[l].w;
            float* const v = tl + offs[l];
            float* const m = v + 2 * ch * cw;
            const float* const pv = tl + offs[l - 1];
            const float* const pm = pv + 2 * ph * pw;
            for (int e = tid; e < ch * cw; e += F1_THREADS) {
                const int y = e / cw, x = e - y * cw;
                float sm = 0.f, sx = 0.f, sy = 0.f;
#pragma unroll
                for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx) {
                        const int yy = 2 * y + dy, xx = 2 * x + dx;
                        if (yy < ph && xx < pw) { sm += pm[yy * pw + xx]; sx += pv[yy * pw + xx]; sy += pv[ph * pw + yy * pw + xx]; }
                    }
                const float ms = sm * 0.25f, vx = sx * 0.25f, vy = sy * 0.25f;
                const float den = ms > 1e-12f ? ms : 1e-12f;
                v[e] = ms > 0.f ? vx / den : 0.f;
                v[ch * cw + e] = ms > 0.f ? vy / den : 0.f;
                m[e] = ms > 0.f ? 1.f : 0.f;
            }
            __syncthreads();
        }
        for (int l = nlev - 1; l >= L; --l) {
            const int fh = lv[l].h, fw = lv[l].w, ch = lv[l + 1].h, cw = lv[l + 1].w;
            const float* const cv = tl + offs[l + 1];
            float* const fv = tl + offs[l];
            const float* const fm = fv + 2 * fh * fw;
            const float sy_ = (float)ch / (float)fh, sx_ = (float)cw / (float)fw;
            for (int e = tid; e < fh * fw; e += F1_THREADS) {
                const int y = e / fw, x = e - y * fw;
                const bool known = fm[e] > 0.f;
                float ox = fv[e], oy = fv[fh * fw + e];
                if (!known) {
                    int y0, y1, x0, x1; float ly, lx;
                    src_index(y, sy_, ch, y0, y1, ly);
                    src_index(x, sx_, cw, x0, x1, lx);
                    const float hy = 1.f - ly, hx = 1.f - lx;
                    ox = fill_bilerp(hy, hx, ly, lx, cv[y0 * cw + x0], cv[y0 * cw + x1], cv[y1 * cw + x0], cv[y1 * cw + x1]);
                    const float* c2 = cv + ch * cw;
                    oy = fill_bilerp(hy, hx, ly, lx, c2[y0 * cw + x0], c2[y0 * cw + x1], c2[y1 * cw + x0], c2[y1 * cw + x1]);
                }
                fv[e] = ox; fv[fh * fw + e] = oy;
            }
            __syncthreads();
        }
        {
            const int cells = lv[L].h * lv[L].w;
            for (int e = tid; e < 2 * cells; e += F1_THREADS) st_agent(p.top + e, tl[offs[L] + e]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_store(p.sync, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);          // every ticket of this launch is taken: the next launch counts from zero
            if (!p.fault) __hip_atomic_store(p.sync + 1, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // ---- everyone: wait for the top of the pyramid ---------------------------------------------------------------------------
    if (tid == 0) {
        const unsigned long long deadline = __builtin_amdgcn_s_memrealtime() + p.timeoutTicks;
        int ok = 1;
        while ((int)(__hip_atomic_load(p.sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - epoch) < 0) {
            __builtin_amdgcn_s_sleep(4);
            if (__builtin_amdgcn_s_memrealtime() > deadline) { ok = 0; break; }
        }
        s_ok = ok;
    }
    __syncthreads();
    if (!s_ok) {
        if (tid == 0) atomicMax(p.error, 1u);
        return;
    }
    // ---- phase B: push level L -> 0 on this tile's regions ----------------------------------------------------------------------
    if (tid == 0) {
        ya[0] = ty * F1_TILE; yb[0] = min(h, ya[0] + F1_TILE);
        xa[0] = tx * F1_TILE; xb[0] = min(w, xa[0] + F1_TILE);
        fo[1] = 0;
        for (int l = 0; l < L; ++l) {
            ya[l + 1] = max(0, (ya[l] >> 1) - 1); yb[l + 1] = min(lv[l + 1].h, ((yb[l] + 1) >> 1) + 1);
            xa[l + 1] = max(0, (xa[l] >> 1) - 1); xb[l + 1] = min(lv[l + 1].w, ((xb[l] + 1) >> 1) + 1);
            fo[l + 2] = fo[l + 1] + 2 * (yb[l + 1] - ya[l + 1]) * (xb[l + 1] - xa[l + 1]);  // level l + 1 at smem[fo[l + 1] ..]: vx | vy
        }
    }
    __syncthreads();
    {   // level L on its region: the last workgroup's values
        const int rh = yb[L] - ya[L], rw = xb[L] - xa[L], cw = lv[L].w, cells = lv[L].h * lv[L].w;
        float* const f = smem + fo[L];
        for (int e = tid; e < rh * rw; e += F1_THREADS) {
            const int ry = e / rw, rx = e - ry * rw;
            const int ge = (ya[L] + ry) * cw + xa[L] + rx;
            f[e] = ld_agent(p.top + ge); f[rh * rw + e] = ld_agent(p.top + cells + ge);
        }
    }
    __syncthreads();
    for (int l = L - 1; l >= 0; --l) {
        const int fh = lv[l].h, fw = lv[l].w, ch = lv[l + 1].h, cw = lv[l + 1].w;
        const int rh = yb[l] - ya[l], rw = xb[l] - xa[l];
        const int crh = yb[l + 1] - ya[l + 1], crw = xb[l + 1] - xa[l + 1];
        const float* const cvx = smem + fo[l + 1];
        const float* const cvy = cvx + crh * crw;
        float* const f = l == 0 ? nullptr : smem + fo[l];
        const float* const pv = p.ws + lv[l].off;                       // (l > 0) the pulled values of this level: vx | vy | m
        const float sy_ = (float)ch / (float)fh, sx_ = (float)cw / (float)fw;
        for (int e = tid; e < rh * rw; e += F1_THREADS) {
            const int ry = e / rw, rx = e - ry * rw;
            const int y = ya[l] + ry, x = xa[l] + rx;
            const int ge = y * fw + x;
            bool known; float kx, ky;
            if (l == 0) {
                const float* g = p.gbuf + (size_t)ge * 12;
                known = g[3] != 0.f; kx = g[8]; ky = g[9];
            } else {
                known = ld_agent(pv + 2 * fh * fw + ge) > 0.f; kx = ld_agent(pv + ge); ky = ld_agent(pv + fh * fw + ge);
            }
            float ox = kx, oy = ky;
            if (!known) {
                int y0, y1, x0, x1; float ly, lx;
                src_index(y, sy_, ch, y0, y1, ly);
                src_index(x, sx_, cw, x0, x1, lx);
                const float hy = 1.f - ly, hx = 1.f - lx;
                // (inside the region by construction; the clamps keep a mistake in that from reading outside the LDS image)
                const int r0 = min(max(y0 - ya[l + 1], 0), crh - 1) * crw, r1 = min(max(y1 - ya[l + 1], 0), crh - 1) * crw;
                const int c0 = min(max(x0 - xa[l + 1], 0), crw - 1), c1 = min(max(x1 - xa[l + 1], 0), crw - 1);
                ox = fill_bilerp(hy, hx, ly, lx, cvx[r0 + c0], cvx[r0 + c1], cvx[r1 + c0], cvx[r1 + c1]);
                oy = fill_bilerp(hy, hx, ly, lx, cvy[r0 + c0], cvy[r0 + c1], cvy[r1 + c0], cvy[r1 + c1]);
            }
            if (l == 0) { p.out[ge] = ox; p.out[(size_t)fh * fw + ge] = oy; }
            else { f[e] = ox; f[rh * rw + e] = oy; }
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" {

int isrAssembleInput(const float* gbuffer_hwc12, const float* flow_filled, const float* prev_high, float* net_input,
                     int h, int w, int init_mode, int ao_inverted, void* stream)
{
    if (!gbuffer_hwc12 || !net_input || h <= 0 || w <= 0) return -1;
    if (prev_high && !flow_filled) return -1;
    if (init_mode < 0 || init_mode > 2) return -1;
    return isrAssembleInputRows(gbuffer_hwc12, flow_filled, prev_high, net_input, h, w, init_mode, ao_inverted, 0, h, stream);
}

int isrAssembleInputRows(const float* gbuffer_hwc12, const float* flow_filled, const float* prev_high, float* net_input,
                         int h, int w, int init_mode, int ao_inverted, int row0, int row1, void* stream)
{
    return isrAssembleInputRect(gbuffer_hwc12, flow_filled, prev_high, net_input, h, w, init_mode, ao_inverted, row0, row1, 0, w, stream);
}

int isrAssembleInputRect(const float* gbuffer_hwc12, const float* flow_filled, const float* prev_high, float* net_input,
                         int h, int w, int init_mode, int ao_inverted, int row0, int row1, int col0, int col1, void* stream)
{
    if (!gbuffer_hwc12 || !net_input || h <= 0 || w <= 0 || row0 < 0 || row1 > h || row0 >= row1 || col0 < 0 || col1 > w || col0 >= col1) return -1;
    if (prev_high && !flow_filled) return -1;
    if (init_mode < 0 || init_mode > 2) return -1;
    AssembleParams p = { gbuffer_hwc12, flow_filled, prev_high, net_input, h, w, init_mode, ao_inverted, row0, col0, col1, nullptr, 0, 0, nullptr, nullptr, nullptr, 0 };
    ISR_LAUNCH_PROFILED(ISR_VARIANT_ASSEMBLE, assemble_input_kernel<false>, dim3((4 * (col1 - col0) + 255) / 256, row1 - row0), dim3(256), 0, (hipStream_t)stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

/* isrAssembleInput straight into the dataflow trunk's workspace: the 101-channel input leaves PACKED-SPLIT (where isrTrunkDataflow's own
 * packing pass would have put it: isrTrunkDataflowInputLayout) and the launch that follows is isrTrunkDataflowPrepacked.  Of `net_input`
 * ([101][h][w] as for isrAssembleInput) only channels 0 .. 4 are written -- what the frame's finishing reads; the rest stays UNDEFINED. */
int isrAssembleInputPacked(const float* gbuffer_hwc12, const float* flow_filled, const float* prev_high, float* net_input,
                           int h, int w, int init_mode, int ao_inverted, void* trunk_workspace, void* stream)
{
    if (!gbuffer_hwc12 || !net_input || !trunk_workspace || h <= 0 || w <= 0) return -1;
    if (prev_high && !flow_filled) return -1;
    if (init_mode < 0 || init_mode > 2) return -1;
    long long off[3]; int groups0 = 0, tiles = 0;
    if (isrTrunkDataflowInputLayout(101, h, w, off, &groups0, &tiles) != 0) return -3;
    char* const ws = (char*)trunk_workspace;
    AssembleParams p = { gbuffer_hwc12, flow_filled, prev_high, net_input, h, w, init_mode, ao_inverted, 0, 0, w,
                         (u32x4*)(ws + off[0]), groups0, (long long)(((long long)h * w + 8) & ~7LL), (u32x4*)(ws + off[1]), (u32x4*)(ws + off[2]),
                         (unsigned*)(ws + 16), tiles };
    ISR_LAUNCH_PROFILED(ISR_VARIANT_ASSEMBLE, assemble_input_kernel<true>, dim3((4 * w + 255) / 256, h), dim3(256), 0, (hipStream_t)stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

long long isrFlowFillWorkspace(int h, int w)
{
    long long floats = 0;
    while (!(h <= 1 && w <= 1)) { h = (h + 1) / 2; w = (w + 1) / 2; floats += 3LL * h * w; }
    return (floats + 16 + 2 * F1_TOP_MAX + 16) * (long long)sizeof(float);     // pyramid | spare | isrFlowFillOne: top values, sync words
}

int isrFlowFill(const float* gbuffer_hwc12, float* flow_out, void* workspace, int h, int w, void* stream)
{
    return isrFlowFillEx(gbuffer_hwc12, flow_out, workspace, h, w, 1024, stream);
}

int isrFlowFillEx(const float* gbuffer_hwc12, float* flow_out, void* workspace, int h, int w, int threads, void* stream)
{
    if (!gbuffer_hwc12 || !flow_out || !workspace || h <= 0 || w <= 0) return -1;
    if (threads != 64 && threads != 128 && threads != 256 && threads != 512 && threads != 1024) return -1;
    const int per = 4 * threads;
    const int big = (h * w + per - 1) / per;
    hipStream_t st = (hipStream_t)stream;
    if (h == 1 && w == 1) {
        hipLaunchKernelGGL(flow_fill_kernel, dim3(1), dim3(64), 0, st, gbuffer_hwc12, flow_out, (float*)workspace, h, w, 3);
        return hipGetLastError() == hipSuccess ? 0 : -2;
    }
    hipLaunchKernelGGL(flow_fill_kernel, dim3(big > 0 ? big : 1), dim3(threads), 0, st, gbuffer_hwc12, flow_out, (float*)workspace, h, w, 0);
    // the pyramid in between is ONE workgroup whatever `threads` says: 1024 threads on a single CU cost the network
    // nothing measurable and finish nine times sooner than 256
    hipLaunchKernelGGL(flow_fill_kernel, dim3(1), dim3(1024), 0, st, gbuffer_hwc12, flow_out, (float*)workspace, h, w, 1);
    hipLaunchKernelGGL(flow_fill_kernel, dim3(4 * big > 0 ? 4 * big : 1), dim3(threads), 0, st, gbuffer_hwc12, flow_out, (float*)workspace, h, w, 2);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

static unsigned* g_fill_error_word = nullptr;
void isrSetFlowFillErrorWord(unsigned* word) { g_fill_error_word = word; }
/* Tests of the timeout path: with `on` the workgroup that finishes the pyramid's top never raises its flag, so every workgroup's wait runs
 * into the deadline, `timeoutTicks` of the 100 MHz clock (0: the default 50 ms).  (With the fault the ticket is still put back to zero, so the
 * next launch on the same workspace works; after a REAL timeout -- a workgroup that never arrived -- the ticket is out of step and the caller
 * zero-fills the workspace: ops._fill_failed.) */
[[maybe_unused]] static int g_fill_fault = 0;
static unsigned long long g_fill_timeout_ticks = 5000000ull;
#ifdef ISR_DIAG
void isrDebugSetFlowFillFault(int on, unsigned long long timeoutTicks) { g_fill_fault = on ? 1 : 0; g_fill_timeout_ticks = timeoutTicks ? timeoutTicks : 5000000ull; }
int isrDebugFlowFillState(void) { return (g_fill_fault || g_fill_timeout_ticks != 5000000ull) ? 1 : 0; }
#endif

int isrFlowFillOneSupported(int h, int w)
{
    if (h <= 0 || w <= 0 || (h == 1 && w == 1)) return 0;     // (a single pixel has no pyramid)
    const long long tiles = (long long)((h + F1_TILE - 1) / F1_TILE) * ((w + F1_TILE - 1) / F1_TILE);
    return tiles <= 256 ? 1 : 0;              // every workgroup must find a slot while others wait (and the top fits the LDS: F1_TOP_MAX)
}

int isrFlowFillOne(const float* gbuffer_hwc12, float* flow_out, void* workspace, int h, int w, void* stream)
{
    if (!gbuffer_hwc12 || !flow_out || !workspace || !isrFlowFillOneSupported(h, w)) return -1;
    long long floats = 0;
    for (int ch = h, cw = w; !(ch <= 1 && cw <= 1);) { ch = (ch + 1) / 2; cw = (cw + 1) / 2; floats += 3LL * ch * cw; }
    FillOneParams p;
    p.gbuf = gbuffer_hwc12; p.out = flow_out; p.ws = (float*)workspace;
    p.h = h; p.w = w; p.tilesX = (w + F1_TILE - 1) / F1_TILE; p.tilesY = (h + F1_TILE - 1) / F1_TILE;
    p.top = p.ws + floats + 16;
    p.sync = reinterpret_cast<unsigned*>(p.top + 2 * F1_TOP_MAX);
    p.error = g_fill_error_word ? g_fill_error_word : p.sync + 2;
    p.timeoutTicks = g_fill_timeout_ticks;        // 50 ms of the 100 MHz clock unless a test shortened it
    ISR_DIAG_SET(p.fault, g_fill_fault);
    ISR_LAUNCH_PROFILED(ISR_VARIANT_FLOW_FILL, flow_fill_one_kernel, dim3(p.tilesX * p.tilesY), dim3(F1_THREADS), 0, (hipStream_t)stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int isrFinishFrame(const float* raw, const float* net_input, float* next_prev, float* rgb, int h, int w,
                   const float* shading24, int exponent, float ao_strength, int inverse_ao, int enable_specular, void* stream)
{
    if (!raw || !net_input || !next_prev || h <= 0 || w <= 0 || (rgb && !shading24)) return -1;
    FinishParams p;
    isr_fill_finish_params(p, raw, net_input, next_prev, rgb, h, w, shading24, exponent, ao_strength, inverse_ao, enable_specular);
    ISR_LAUNCH_PROFILED(ISR_VARIANT_FINISH, finish_frame_kernel, dim3((4 * w + 255) / 256, 4 * h), dim3(256), 0, (hipStream_t)stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // extern "C"
