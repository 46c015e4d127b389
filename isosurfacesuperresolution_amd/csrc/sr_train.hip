// Training-side kernels of the super-resolution path that are NOT convolutions (gfx950): everything here is
// HBM/launch bound elementwise work that the reference runs as chains of small PyTorch launches.
//
//  * x2 bilinear upsampling (align_corners=False) forward / backward -- nn.Upsample(scale_factor=2, 'bilinear')
//    of SuperresolutionNetwork/models/enhancenet.py:116,119 and its autograd adjoint;
//  * LossNetUnshaded (SuperresolutionNetwork/losses/lossnet_unshaded.py:236-388, l1 / mse / temp-l2 terms on
//    mask / normal / ao / depth / colour): forward = ONE pass over gt / pred / prev that produces every term's
//    sum, backward = ONE pass that writes d loss / d pred and d loss / d prev.  The module path
//    (losses/lossnet_unshaded.py in this package) issues ~150 launches forward and ~200 backward per frame.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/isr_sr_kernels.h"
#include "sr_finish.h"
#include "sr_warp_exact.h"

namespace {

// ---------------------------------------------------------------------------------------------------------
// x2 bilinear upsampling
// ---------------------------------------------------------------------------------------------------------
// y[p][Y][X] = hy*(hx*x[y0][x0] + lx*x[y0][x1]) + ly*(hx*x[y1][x0] + lx*x[y1][x1])   (ATen's association)
// V output pixels per thread: 4 where the output rows are whole quads (even w), 2 for an odd w (the rows are whole pairs: W = 2 w)
template <int V>
__global__ __launch_bounds__(256) void upsample2x_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int h, int w, long long quads)
{
    const int W = 2 * w, H = 2 * h, QW = W / V;
    for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < quads; q += (long long)gridDim.x * 256) {
        const int qx = (int)(q % QW);
        const long long r = q / QW;
        const int Y = (int)(r % H);
        const long long plane = r / H;
        int y0, y1; float ly;
        isr_src_index(Y, 0.5f, h, y0, y1, ly);
        const float hy = 1.f - ly;
        const float* r0 = x + (plane * h + y0) * w;
        const float* r1 = x + (plane * h + y1) * w;
        float o[V];
#pragma unroll
        for (int k = 0; k < V; ++k) {
            int x0, x1; float lx;
            isr_src_index(V * qx + k, 0.5f, w, x0, x1, lx);
            const float hx = 1.f - lx;
            o[k] = hy * (hx * r0[x0] + lx * r0[x1]) + ly * (hx * r1[x0] + lx * r1[x1]);
        }
        float* dst = y + (plane * H + Y) * W + V * qx;
        if constexpr (V == 4) *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
        else { dst[0] = o[0]; dst[1] = o[1]; }
    }
}

// weight with which output index `o` reads input index `i` along one axis of length n (0 if it does not)
__device__ __forceinline__ float up2_weight(int o, int i, int n)
{
    int i0, i1; float l1;
    isr_src_index(o, 0.5f, n, i0, i1, l1);
    return (i0 == i ? 1.f - l1 : 0.f) + (i1 == i ? l1 : 0.f);
}

// adjoint as a gather (deterministic, no atomics): input pixel (iy, ix) is read by output rows 2iy-1 .. 2iy+2
__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(const float* __restrict__ gy, float* __restrict__ gx, int h, int w, long long count)
{
    const int W = 2 * w, H = 2 * h;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long long)gridDim.x * 256) {
        const int ix = (int)(i % w);
        const long long r = i / w;
        const int iy = (int)(r % h);
        const long long plane = r / h;
        float wx[4], acc = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int X = 2 * ix - 1 + k;
            wx[k] = (X >= 0 && X < W) ? up2_weight(X, ix, w) : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int Y = 2 * iy - 1 + j;
            if (Y < 0 || Y >= H) continue;
            const float wy = up2_weight(Y, iy, h);
            const float* row = gy + (plane * H + Y) * W;
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int X = 2 * ix - 1 + k;
                if (X >= 0 && X < W) s += wx[k] * row[X];
            }
            acc += wy * s;
        }
        gx[i] = acc;
    }
}

// The same two maps for the shapes of a training step (w a multiple of 4, fewer than 2^31 elements), laid out for the memory system:
// no 64-bit divisions, the x2 weights in closed form (0.25 / 0.75, 1 at the borders -- what up2_weight evaluates to), a thread per
// 2 x 4 output block (forward: 12 loads for 8 outputs) / per 4 input pixels (backward: the 4 x 10 window as 8 quad + 8 single loads
// instead of 64 single ones).  Every output is the same expression in the same order as in the kernels above: same bits.
__global__ __launch_bounds__(256) void upsample2x_fwd4_kernel(const float* __restrict__ x, float* __restrict__ y, int h, int w, unsigned count)
{
    // a thread = input row iy, input columns 4 q .. 4 q + 3 -> the 2 x 8 output block below them: the 3 x 6 input window is read as
    // three quads + six border values (18 loads for 16 outputs, four quad stores)
    const unsigned W = 2u * w, qw = (unsigned)w / 4u;
    const unsigned t = blockIdx.x * 256u + threadIdx.x;
    if (t >= count) return;
    const unsigned q = t % qw, r = t / qw, iy = r % (unsigned)h, plane = r / (unsigned)h;
    const int ix0 = (int)(4 * q);
    const float* src = x + (size_t)plane * h * w;
    float* dst = y + ((size_t)plane * 2 * h + 2 * iy) * W + 2 * ix0;
    const int ym = iy > 0 ? (int)iy - 1 : 0, yp = (int)iy < h - 1 ? (int)iy + 1 : (int)iy;
    const int xm = ix0 > 0 ? ix0 - 1 : 0, xp = ix0 + 4 < w ? ix0 + 4 : w - 1;
    float v[3][6];
    const int rows[3] = { ym, (int)iy, yp };
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float* row = src + (size_t)rows[k] * w;
        const float4 c = *reinterpret_cast<const float4*>(row + ix0);
        v[k][0] = row[xm]; v[k][1] = c.x; v[k][2] = c.y; v[k][3] = c.z; v[k][4] = c.w; v[k][5] = row[xp];
    }
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        // output row 2 iy + d blends window rows (d, d + 1), output column 2 ix0 + e window columns ((e + 1) / 2, (e + 1) / 2 + 1): what
        // isr_src_index returns, except at the image border, where it clamps both taps onto the border pixel and returns weight 0 for
        // the second -- the window holds the border pixel in both slots there, so the value is the same (finite inputs)
        int y0, y1; float ly;
        isr_src_index((int)(2 * iy) + d, 0.5f, h, y0, y1, ly);
        const float hy = 1.f - ly;
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            int x0, x1; float lx;
            isr_src_index(2 * ix0 + e, 0.5f, w, x0, x1, lx);
            const float hx = 1.f - lx;
            const int c0 = (e + 1) >> 1, c1 = c0 + 1;
            o[e] = hy * (hx * v[d][c0] + lx * v[d][c1]) + ly * (hx * v[d + 1][c0] + lx * v[d + 1][c1]);
        }
        *reinterpret_cast<float4*>(dst + (size_t)d * W) = make_float4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<float4*>(dst + (size_t)d * W + 4) = make_float4(o[4], o[5], o[6], o[7]);
    }
}

__global__ __launch_bounds__(256) void upsample2x_bwd4_kernel(const float* __restrict__ gy, float* __restrict__ gx, int h, int w, unsigned count)
{
    const unsigned W = 2u * w, H = 2u * h, qw = (unsigned)w / 4u;
    const unsigned t = blockIdx.x * 256u + threadIdx.x;                      // (plane, iy, 4 input pixels)
    if (t >= count) return;
    const unsigned q = t % qw, r = t / qw, iy = r % (unsigned)h, plane = r / (unsigned)h;
    const int ix0 = (int)(4 * q);
    float acc[4] = { 0.f, 0.f, 0.f, 0.f };
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int Y = 2 * (int)iy - 1 + j;
        if (Y < 0 || Y >= (int)H) continue;
        // output row Y reads input row iy with 0.25 (the far rows), 0.75 (the near ones), 1 where the other tap is clamped onto it
        const float wy = (j == 0 || j == 3) ? 0.25f : ((j == 1 && iy == 0) || (j == 2 && iy == (unsigned)h - 1)) ? 1.0f : 0.75f;
        const float* row = gy + ((size_t)plane * H + Y) * W + 2 * ix0;      // row[-1 .. 8] are this thread's ten columns
        const float4 a = *reinterpret_cast<const float4*>(row), b = *reinterpret_cast<const float4*>(row + 4);
        const float left = ix0 > 0 ? row[-1] : 0.f, right = ix0 + 4 < w ? row[8] : 0.f;
        const float v[10] = { left, a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, right };
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int ix = ix0 + e;
            float s = 0.f;
            if (ix > 0) s += 0.25f * v[2 * e];
            s += (ix == 0 ? 1.0f : 0.75f) * v[2 * e + 1];
            s += (ix == w - 1 ? 1.0f : 0.75f) * v[2 * e + 2];
            if (ix < w - 1) s += 0.25f * v[2 * e + 3];
            acc[e] += wy * s;
        }
    }
    *reinterpret_cast<float4*>(gx + ((size_t)plane * h + iy) * w + ix0) = make_float4(acc[0], acc[1], acc[2], acc[3]);
}

// ---------------------------------------------------------------------------------------------------------
// Residual reconstruction of the network output (enhancenet.py:65-78, reconType='residual'):
// out[:, c] = y[:, c] + bilinear_x4(x[:, c]) for c < k, out[:, c] = y[:, c] beyond -- one launch instead of slice,
// F.interpolate, add and cat; the adjoint w.r.t. x is a gather over the 8 x 8 output pixels that read an input pixel.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void recon_residual_fwd_kernel(const float* __restrict__ y, const float* __restrict__ x, float* __restrict__ out,
                                                                 int cout, int cin, int k, int h, int w)
{
    const int W = 4 * w, H = 4 * h;
    const int X = blockIdx.x * 256 + threadIdx.x, Y = blockIdx.y, n = blockIdx.z;
    if (X >= W) return;
    int y0, y1, x0, x1; float ly, lx;
    isr_src_index(Y, 0.25f, h, y0, y1, ly);
    isr_src_index(X, 0.25f, w, x0, x1, lx);
    const float hy = 1.f - ly, hx = 1.f - lx;
    const size_t hplane = (size_t)H * W, lplane = (size_t)h * w;
    const size_t pix = (size_t)Y * W + X;
    for (int c = 0; c < cout; ++c) {
        float v = y[((size_t)n * cout + c) * hplane + pix];
        if (c < k) {
            const float* q = x + ((size_t)n * cin + c) * lplane;
            v += hy * (hx * q[y0 * w + x0] + lx * q[y0 * w + x1]) + ly * (hx * q[y1 * w + x0] + lx * q[y1 * w + x1]);
        }
        out[((size_t)n * cout + c) * hplane + pix] = v;
    }
}

__device__ __forceinline__ float up4_weight(int o, int i, int n)
{
    int i0, i1; float l1;
    isr_src_index(o, 0.25f, n, i0, i1, l1);
    return (i0 == i ? 1.f - l1 : 0.f) + (i1 == i ? l1 : 0.f);
}

// gx[n][c][iy][ix], c < k: sum over the output pixels (4 iy - 2 .. 4 iy + 5) x (4 ix - 2 .. 4 ix + 5) that blend this input
// pixel.  Eight neighbouring lanes take one output row each and add up through DPP-sized shuffles (a single thread walking
// the 64 taps is a chain of eight load latencies: 20 us for a launch that moves 4 MB).  Channels >= k are zeroed by the caller.
__global__ __launch_bounds__(256) void recon_residual_bwd_kernel(const float* __restrict__ gy, float* __restrict__ gx,
                                                                 int cout, int cin, int k, int h, int w, long long count8)
{
    const int W = 4 * w, H = 4 * h;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const int j = (int)(t & 7);
    const long long i = t >> 3;                       // (n, c < k, iy, ix)
    float part = 0.f;
    int ix = 0, iy = 0, c = 0, n = 0;
    const bool live = t < count8;
    if (live) {
        ix = (int)(i % w);
        long long r = i / w;
        iy = (int)(r % h); r /= h;
        c = (int)(r % k); n = (int)(r / k);
        const int Y = 4 * iy - 2 + j;
        if (Y >= 0 && Y < H) {
            const float wy = up4_weight(Y, iy, h);
            const float* row = gy + (((size_t)n * cout + c) * H + Y) * (size_t)W;
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int X = 4 * ix - 2 + q;
                if (X >= 0 && X < W) s += up4_weight(X, ix, w) * row[X];
            }
            part = wy * s;
        }
    }
    // fixed-order sum of the eight rows (bitwise reproducible)
    part += __shfl_xor(part, 1);
    part += __shfl_xor(part, 2);
    part += __shfl_xor(part, 4);
    if (live && j == 0) gx[(((size_t)n * cin + c) * h + iy) * (size_t)w + ix] = part;
}

// ---------------------------------------------------------------------------------------------------------
// LossNetUnshaded
// ---------------------------------------------------------------------------------------------------------
constexpr int LOSS_TERMS = 15;        // kind * 5 + target;  kind: 0 mse, 1 l1, 2 temp-l2;  target: 0 mask 1 normal 2 ao 3 depth 4 colour
constexpr int LOSS_SLOTS = 16;        // + the weighted total

struct LossParams {
    const float* gt;      // [N][6][H][W]
    const float* pred;
    const float* prev;    // may be NULL (no temp-l2 terms)
    int N, H, W, pad;
    float weight[LOSS_TERMS];
    unsigned enabled;     // bit t: term t is evaluated
    float ambmat[3], diffmat[3], light[3], bg[3];
    float ao_strength;
    int inverse_ao;
    float* partial;       // [blocks][LOSS_SLOTS]
    float* values;        // [LOSS_SLOTS]: per-term means, [15] = sum of weight * mean
    int blocks;
    const float* gout;    // backward: d / d values (only [15] is used)
    float* gpred;
    float* gprev;         // may be NULL
};

struct Fields {           // what the loss terms compare, and what their derivatives need
    float m, n[3], ao, d, col[3];
    float nh[3], len, ndl, aof, t, cb[3], colpre[3];
    bool a_in, t_in;
};

__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.f), 1.f); }

__device__ __forceinline__ void eval_fields(const LossParams& P, const float (&x)[6], float gate, Fields& f)
{
    f.m = x[0];
    f.len = sqrtf(x[1] * x[1] + x[2] * x[2] + x[3] * x[3]);
    const float den = fmaxf(f.len, 1e-7f);
#pragma unroll
    for (int k = 0; k < 3; ++k) { f.nh[k] = x[1 + k] / den; f.n[k] = f.nh[k] * gate; }
    f.d = x[4] * gate;
    f.ao = x[5] * gate;
    const float a = P.inverse_ao ? 1.0f - x[5] : x[5];
    f.a_in = a >= 0.f && a <= 1.f;
    f.aof = P.ao_strength * clamp01(a) + (1.0f - P.ao_strength);
    f.ndl = P.light[0] * x[1] + P.light[1] * x[2] + P.light[2] * x[3];      // the shader takes the normal as it is
    const float tt = x[0] * 0.5f + 0.5f;
    f.t_in = tt >= 0.f && tt <= 1.f;
    f.t = clamp01(tt);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        f.cb[k] = P.ambmat[k] + P.diffmat[k] * fabsf(f.ndl);
        f.colpre[k] = P.bg[k] + f.t * (f.cb[k] * f.aof - P.bg[k]);
        f.col[k] = clamp01(f.colpre[k]);
    }
}

// upstream derivative w.r.t. the fields -> derivative w.r.t. the six channels
struct FieldGrad { float m, n[3], ao, d, col[3]; };

__device__ __forceinline__ void backprop_fields(const LossParams& P, const Fields& f, float gate, const FieldGrad& g, float (&gx)[6])
{
    gx[0] = g.m;
    gx[4] = gate * g.d;
    gx[5] = gate * g.ao;
    float dn[3] = {gate * g.n[0], gate * g.n[1], gate * g.n[2]};
    if (f.len > 1e-7f) {
        const float dot = f.nh[0] * dn[0] + f.nh[1] * dn[1] + f.nh[2] * dn[2];
#pragma unroll
        for (int k = 0; k < 3; ++k) gx[1 + k] = (dn[k] - f.nh[k] * dot) / f.len;
    } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) gx[1 + k] = dn[k] / 1e-7f;
    }
    float dt = 0.f, daof = 0.f, dabs = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float gp = (f.colpre[k] >= 0.f && f.colpre[k] <= 1.f) ? g.col[k] : 0.f;
        dt += gp * (f.cb[k] * f.aof - P.bg[k]);
        const float dc = gp * f.t;
        daof += dc * f.cb[k];
        dabs += dc * f.aof * P.diffmat[k];
    }
    if (f.t_in) gx[0] += 0.5f * dt;
    if (f.a_in) gx[5] += daof * P.ao_strength * (P.inverse_ao ? -1.f : 1.f);
    const float sg = f.ndl > 0.f ? 1.f : (f.ndl < 0.f ? -1.f : 0.f);
#pragma unroll
    for (int k = 0; k < 3; ++k) gx[1 + k] += dabs * sg * P.light[k];
}

__device__ __forceinline__ float sgn(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }

// one thread = four horizontally adjacent pixels of one image
template <bool BACKWARD>
__global__ __launch_bounds__(256) void loss_unshaded_kernel(LossParams P)
{
    const int QW = P.W / 4;
    const long long quadsPerImage = (long long)QW * P.H;
    const long long quads = quadsPerImage * P.N;
    const size_t plane = (size_t)P.H * P.W;
    float sums[LOSS_TERMS];
#pragma unroll
    for (int t = 0; t < LOSS_TERMS; ++t) sums[t] = 0.f;

    float coef[LOSS_TERMS];      // backward: weight * d mean / d sum, times the upstream gradient
    if (BACKWARD) {
        const float go = P.gout[LOSS_TERMS];
        const float n1 = (float)((double)P.N * P.H * P.W), n3 = 3.0f * n1;
#pragma unroll
        for (int t = 0; t < LOSS_TERMS; ++t) {
            const int target = t % 5;
            const float cnt = (target == 1 || target == 4) ? n3 : n1;
            coef[t] = ((P.enabled >> t) & 1u) ? go * P.weight[t] / cnt : 0.f;
        }
    }

    for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < quads; q += (long long)gridDim.x * 256) {
        const int img = (int)(q / quadsPerImage);
        const long long r = q % quadsPerImage;
        const int Y = (int)(r / QW), X0 = 4 * (int)(r % QW);
        const size_t base = (size_t)img * 6 * plane + (size_t)Y * P.W + X0;
        float4 g4[6], p4[6], v4[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            g4[c] = *reinterpret_cast<const float4*>(P.gt + base + c * plane);
            p4[c] = *reinterpret_cast<const float4*>(P.pred + base + c * plane);
            v4[c] = P.prev ? *reinterpret_cast<const float4*>(P.prev + base + c * plane) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float4 gp4[6], gv4[6];
        const bool rowInside = Y >= P.pad && Y < P.H - P.pad;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int X = X0 + k;
            const bool inside = rowInside && X >= P.pad && X < P.W - P.pad;
            float g[6], p[6], v[6], gp[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, gv[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                g[c] = reinterpret_cast<const float*>(&g4[c])[k];
                p[c] = reinterpret_cast<const float*>(&p4[c])[k];
                v[c] = reinterpret_cast<const float*>(&v4[c])[k];
            }
            // a zeroed border pixel gives identical fields for gt / pred / prev: no contribution, no gradient
            if (inside) {
                const float gate = clamp01(g[0] * 0.5f + 0.5f);
                Fields fg, fp, fv;
                eval_fields(P, g, gate, fg);
                eval_fields(P, p, gate, fp);
                if (P.prev) eval_fields(P, v, gate, fv);
                // differences b - a per field component: [0] mask, [1..3] normal, [4] ao, [5] depth, [6..8] colour
                float dg[9] = {fp.m - fg.m, fp.n[0] - fg.n[0], fp.n[1] - fg.n[1], fp.n[2] - fg.n[2], fp.ao - fg.ao, fp.d - fg.d,
                               fp.col[0] - fg.col[0], fp.col[1] - fg.col[1], fp.col[2] - fg.col[2]};
                float dv[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if (P.prev) {
                    dv[0] = fp.m - fv.m; dv[1] = fp.n[0] - fv.n[0]; dv[2] = fp.n[1] - fv.n[1]; dv[3] = fp.n[2] - fv.n[2];
                    dv[4] = fp.ao - fv.ao; dv[5] = fp.d - fv.d;
                    dv[6] = fp.col[0] - fv.col[0]; dv[7] = fp.col[1] - fv.col[1]; dv[8] = fp.col[2] - fv.col[2];
                }
                // component -> target
                constexpr int tgt[9] = {0, 1, 1, 1, 2, 3, 4, 4, 4};
                if (!BACKWARD) {
#pragma unroll
                    for (int c = 0; c < 9; ++c) {
                        sums[0 + tgt[c]] += dg[c] * dg[c];
                        sums[5 + tgt[c]] += fabsf(dg[c]);
                        sums[10 + tgt[c]] += dv[c] * dv[c];
                    }
                } else {
                    float up[9], uv[9];
#pragma unroll
                    for (int c = 0; c < 9; ++c) {
                        const float tv = 2.f * coef[10 + tgt[c]] * dv[c];
                        up[c] = 2.f * coef[0 + tgt[c]] * dg[c] + coef[5 + tgt[c]] * sgn(dg[c]) + tv;
                        uv[c] = -tv;
                    }
                    FieldGrad fgp = {up[0], {up[1], up[2], up[3]}, up[4], up[5], {up[6], up[7], up[8]}};
                    backprop_fields(P, fp, gate, fgp, gp);
                    if (P.gprev) {
                        FieldGrad fgv = {uv[0], {uv[1], uv[2], uv[3]}, uv[4], uv[5], {uv[6], uv[7], uv[8]}};
                        backprop_fields(P, fv, gate, fgv, gv);
                    }
                }
            }
            if (BACKWARD) {
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    reinterpret_cast<float*>(&gp4[c])[k] = gp[c];
                    reinterpret_cast<float*>(&gv4[c])[k] = gv[c];
                }
            }
        }
        if (BACKWARD) {
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                *reinterpret_cast<float4*>(P.gpred + base + c * plane) = gp4[c];
                if (P.gprev) *reinterpret_cast<float4*>(P.gprev + base + c * plane) = gv4[c];
            }
        }
    }

    if (!BACKWARD) {
        __shared__ float red[4][LOSS_SLOTS];
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int t = 0; t < LOSS_TERMS; ++t) {
            float s = sums[t];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
            if (lane == 0) red[wave][t] = s;
        }
        __syncthreads();
        if (threadIdx.x < LOSS_TERMS)
            P.partial[(size_t)blockIdx.x * LOSS_SLOTS + threadIdx.x] =
                (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    }
}

// one workgroup of 16 waves: wave t sums term t over the blocks in a fixed order, thread 0 forms the total
__global__ __launch_bounds__(1024) void loss_finalize_kernel(LossParams P)
{
    __shared__ float mean[LOSS_SLOTS];
    const int lane = threadIdx.x & 63, t = threadIdx.x >> 6;
    if (t < LOSS_TERMS) {
        float s = 0.f;
        for (int b = lane; b < P.blocks; b += 64) s += P.partial[(size_t)b * LOSS_SLOTS + t];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if (lane == 0) {
            const int target = t % 5;
            const double cnt = (double)P.N * P.H * P.W * ((target == 1 || target == 4) ? 3.0 : 1.0);
            mean[t] = ((P.enabled >> t) & 1u) ? (float)((double)s / cnt) : 0.f;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float total = 0.f;
        for (int k = 0; k < LOSS_TERMS; ++k) {
            P.values[k] = mean[k];
            if ((P.enabled >> k) & 1u) total += P.weight[k] * mean[k];
        }
        P.values[LOSS_TERMS] = total;
    }
}

int fill_loss_params(LossParams& P, const float* gt, const float* pred, const float* prev, int N, int H, int W, int pad,
                     const float* weights15, unsigned enabled, const float* shading12, float ao_strength, int inverse_ao)
{
    if (!gt || !pred || !weights15 || !shading12 || N <= 0 || H <= 0 || W <= 0 || (W & 3) || pad < 0) return -1;
    if (2 * pad >= H || 2 * pad >= W) return -1;
    P.gt = gt; P.pred = pred; P.prev = prev;
    P.N = N; P.H = H; P.W = W; P.pad = pad;
    for (int t = 0; t < LOSS_TERMS; ++t) P.weight[t] = weights15[t];
    P.enabled = enabled & ((1u << LOSS_TERMS) - 1u);
    if (!prev) P.enabled &= (1u << 10) - 1u;
    for (int k = 0; k < 3; ++k) {
        P.ambmat[k] = shading12[k]; P.diffmat[k] = shading12[3 + k]; P.light[k] = shading12[6 + k]; P.bg[k] = shading12[9 + k];
    }
    P.ao_strength = ao_strength; P.inverse_ao = inverse_ao;
    long long quads = (long long)N * H * (W / 4);
    long long blocks = (quads + 255) / 256;
    P.blocks = (int)(blocks > 1024 ? 1024 : blocks);
    P.partial = nullptr; P.values = nullptr; P.gout = nullptr; P.gpred = nullptr; P.gprev = nullptr;
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
// Recurrent network input of a training clip (mainVideoUnshaded.py:436-447,463-466), frames t > 0:
//   previous_output = cat(clamp(p0,-1,1), normalize(p1..3), clamp(p4,0,1), clamp(p5,0,1))     p = raw prediction t-1
//   warped          = VideoTools.warp_upscale(previous_output, flow[t-1], 4, special_mask=True)
//   net_input       = cat(input[t], VideoTools.flatten_high(warped, 4))
// forward: one gather pass that writes `warped` (the loss's temp-l2 partner) and `net_input`;
// backward: one scatter pass (float atomics, as in PyTorch's grid_sampler backward) into the gradient of
// previous_output, one pass through clamp / normalize to the gradient of p.
// ---------------------------------------------------------------------------------------------------------
struct RecurParams {
    const float* raw;        // [B][6][H][W] raw prediction of the previous frame
    const float* input;      // [B][5][h][w] (batch stride inStride)
    const float* flow;       // [B][2][h][w] (batch stride flowStride)
    float* netin;            // [B][101][h][w]
    float* warped;           // [B][6][H][W]
    long long inStride, flowStride;
    int B, h, w;
    const float* gnetin;     // backward: [B][101][h][w] or NULL
    const float* gwarped;    // [B][6][H][W] or NULL
    float* gout;             // [B][6][H][W] gradient of previous_output (zeroed before the scatter)
    float* graw;             // [B][6][H][W]
};

struct WarpTaps { size_t b00; int ix0, iy0; float w00, w01, w10, w11; bool v00, v01, v10, v11; };

// (operation for operation models/videotools.py: warp_upscale -- see sr_warp_exact.h)
__device__ __forceinline__ WarpTaps warp_taps(const float* fx, const float* fy, int h, int w, int Y, int X)
{
#pragma clang fp contract(off)
    const int H = 4 * h, W = 4 * w;
    int y0, y1, x0, x1; float ly, lx;
    isr_src_index_rn(Y, 0.25f, h, y0, y1, ly);
    isr_src_index_rn(X, 0.25f, w, x0, x1, lx);
    const float hy = 1.f - ly, hx = 1.f - lx;
    // flow scaled by (-2, +2), then bilinearly upsampled (videotools.py:65-70)
    const float flx = isr_bilerp_rn(hy, hx, ly, lx, fx[y0 * w + x0] * -2.0f, fx[y0 * w + x1] * -2.0f, fx[y1 * w + x0] * -2.0f, fx[y1 * w + x1] * -2.0f);
    const float fly = isr_bilerp_rn(hy, hx, ly, lx, fy[y0 * w + x0] * 2.0f, fy[y0 * w + x1] * 2.0f, fy[y1 * w + x0] * 2.0f, fy[y1 * w + x1] * 2.0f);
    // grid = linspace(-1, 1) + flow; bilinear sampler with align_corners=True and zero padding
    const float gx = isr_pixel_grid(X, W) + flx;
    const float gy = isr_pixel_grid(Y, H) + fly;
    const float gx1 = gx + 1.0f, gy1 = gy + 1.0f;
    const float sx = gx1 * (0.5f * (float)(W - 1)), sy = gy1 * (0.5f * (float)(H - 1));
    const float fx0 = floorf(sx), fy0 = floorf(sy);
    const int ix0 = (int)fminf(fmaxf(fx0, -2.f), (float)W), iy0 = (int)fminf(fmaxf(fy0, -2.f), (float)H);
    const float wx1 = sx - fx0, wy1 = sy - fy0;
    const float wx0 = 1.f - wx1, wy0 = 1.f - wy1;
    const bool vx0 = (unsigned)ix0 < (unsigned)W, vx1 = (unsigned)(ix0 + 1) < (unsigned)W;
    const bool vy0 = (unsigned)iy0 < (unsigned)H, vy1 = (unsigned)(iy0 + 1) < (unsigned)H;
    WarpTaps t;
    t.b00 = (size_t)((long long)iy0 * W + ix0);
    t.ix0 = ix0; t.iy0 = iy0;
    t.w00 = wx0 * wy0; t.w01 = wx1 * wy0; t.w10 = wx0 * wy1; t.w11 = wx1 * wy1;
    t.v00 = vy0 && vx0; t.v01 = vy0 && vx1; t.v10 = vy1 && vx0; t.v11 = vy1 && vx1;
    return t;
}

// previous_output at one pixel from the raw prediction (mask already mapped to [0,1] for the special-mask warp)
__device__ __forceinline__ void prev_output_at(const float* raw, size_t hplane, size_t at, float (&o)[6])
{
    const float m = fminf(fmaxf(raw[at], -1.f), 1.f);
    const float nx = raw[hplane + at], ny = raw[2 * hplane + at], nz = raw[3 * hplane + at];
    const float den = fmaxf(sqrtf(nx * nx + ny * ny + nz * nz), 1e-7f);
    o[0] = m * 0.5f + 0.5f;
    o[1] = nx / den; o[2] = ny / den; o[3] = nz / den;
    o[4] = clamp01(raw[4 * hplane + at]);
    o[5] = clamp01(raw[5 * hplane + at]);
}

// one thread per (low-res pixel, dx), loop over dy: as assemble_input_kernel of the inference path
__global__ __launch_bounds__(256) void recurrent_input_fwd_kernel(const RecurParams p)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int x = t >> 2, dx = t & 3;
    const int y = blockIdx.y, b = blockIdx.z;
    if (x >= p.w) return;
    const int H = 4 * p.h, W = 4 * p.w;
    const size_t plane = (size_t)p.h * p.w, hplane = (size_t)H * W;
    const size_t pix = (size_t)y * p.w + x;
    float* out = p.netin + (size_t)b * 101 * plane;
    if (dx == 0) {
        const float* in = p.input + (size_t)b * p.inStride;
#pragma unroll
        for (int c = 0; c < 5; ++c) out[c * plane + pix] = in[c * plane + pix];
    }
    const float* fx = p.flow + (size_t)b * p.flowStride;
    const float* fy = fx + plane;
    const float* raw = p.raw + (size_t)b * 6 * hplane;
    float* wout = p.warped + (size_t)b * 6 * hplane;
    float* o = out + 5 * plane + pix;
    const int X = 4 * x + dx;
    for (int dy = 0; dy < 4; ++dy) {
        const int Y = 4 * y + dy;
        const WarpTaps tp = warp_taps(fx, fy, p.h, p.w, Y, X);
        float a[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, bq[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        float c_[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, d[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (tp.v00) prev_output_at(raw, hplane, tp.b00, a);
        if (tp.v01) prev_output_at(raw, hplane, tp.b00 + 1, bq);
        if (tp.v10) prev_output_at(raw, hplane, tp.b00 + W, c_);
        if (tp.v11) prev_output_at(raw, hplane, tp.b00 + W + 1, d);
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            float r;
            {
#pragma clang fp contract(off)
                const float t00 = a[c] * tp.w00, t01 = bq[c] * tp.w01, t10 = c_[c] * tp.w10, t11 = d[c] * tp.w11;
                r = t00 + t01;
                r = r + t10;
                r = r + t11;
                if (c == 0) { r = r * 2.0f; r = r - 1.0f; }      // special mask: zero padding means "mask = -1"
            }
            o[(size_t)(c * 16 + dy * 4 + dx) * plane] = r;
            wout[(size_t)c * hplane + (size_t)Y * W + X] = r;
        }
    }
}

// Backward of the warp: every high-res source pixel scatters its gradient to the four pixels its bilinear sample read.
// The atomics are pre-combined in LDS: a workgroup takes a 16 x 64 tile of source pixels (one per thread) and adds their
// taps into a 24 x 72 window (the tile + 4 pixels around it) of LDS per channel; taps that land outside the window (flow
// of more than 4 high-res pixels) go to memory directly; the window is then added to memory once per element -- 10 k
// coalesced global atomics per tile instead of 24.6 k neighbouring ones: 45 us per launch on a 16 x 128^2 batch against 125
// for one global atomic per tap.  (Floating-point sums in an unspecified order, like PyTorch's grid-sampler backward.)
constexpr int SC_TH = 16, SC_TW = 64, SC_R = 4;
constexpr int SC_WH = SC_TH + 2 * SC_R, SC_WW = SC_TW + 2 * SC_R, SC_WPIX = SC_WH * SC_WW;   // 24 x 72 = 1728

__global__ __launch_bounds__(1024) void recurrent_input_scatter_lds_kernel(const RecurParams p)
{
    __shared__ float win[6 * SC_WPIX];
    const int tid = threadIdx.x;
    const int W = 4 * p.w, H = 4 * p.h;
    const int X0 = blockIdx.x * SC_TW, Y0 = blockIdx.y * SC_TH, b = blockIdx.z;
    const size_t plane = (size_t)p.h * p.w, hplane = (size_t)H * W;
    const float* fx = p.flow + (size_t)b * p.flowStride;
    const float* fy = fx + plane;
    const float* gnb = p.gnetin ? p.gnetin + (size_t)b * 101 * plane + 5 * plane : nullptr;
    const float* gw = p.gwarped ? p.gwarped + (size_t)b * 6 * hplane : nullptr;
    float* go = p.gout + (size_t)b * 6 * hplane;
    for (int i = tid; i < 6 * SC_WPIX; i += 1024) win[i] = 0.f;
    __syncthreads();
    {                                                  // one source pixel per thread
        const int q = tid;
        const int Y = Y0 + q / SC_TW, X = X0 + (q % SC_TW);
        const bool inside = Y < H && X < W;
        const int y = Y >> 2, dy = Y & 3, x = X >> 2, dx = X & 3;
        float g[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            g[c] = 0.f;
            if (inside && gnb) g[c] += gnb[(size_t)(c * 16 + dy * 4 + dx) * plane + (size_t)y * p.w + x];
            if (inside && gw) g[c] += gw[(size_t)c * hplane + (size_t)Y * W + X];
        }
        const WarpTaps tp = warp_taps(fx, fy, p.h, p.w, inside ? Y : 0, inside ? X : 0);
        const int wy = tp.iy0 - (Y0 - SC_R), wx = tp.ix0 - (X0 - SC_R);
        auto tap = [&](bool valid, int oy, int ox, float wgt) {
            if (!valid || !inside) return;
            const int ty = wy + oy, tx = wx + ox;
            if ((unsigned)ty < (unsigned)SC_WH && (unsigned)tx < (unsigned)SC_WW) {
#pragma unroll
                for (int c = 0; c < 6; ++c)
                    __hip_atomic_fetch_add(&win[c * SC_WPIX + ty * SC_WW + tx], g[c] * wgt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
                const size_t at = tp.b00 + (size_t)oy * W + ox;
#pragma unroll
                for (int c = 0; c < 6; ++c) unsafeAtomicAdd(go + (size_t)c * hplane + at, g[c] * wgt);
            }
        };
        tap(tp.v00, 0, 0, tp.w00);
        tap(tp.v01, 0, 1, tp.w01);
        tap(tp.v10, 1, 0, tp.w10);
        tap(tp.v11, 1, 1, tp.w11);
    }
    __syncthreads();
    for (int i = tid; i < 6 * SC_WPIX; i += 1024) {
        const float v = win[i];
        if (v == 0.f) continue;
        const int c = i / SC_WPIX, e = i - c * SC_WPIX;
        const int ty = Y0 - SC_R + e / SC_WW, tx = X0 - SC_R + (e % SC_WW);
        if ((unsigned)ty < (unsigned)H && (unsigned)tx < (unsigned)W) unsafeAtomicAdd(go + (size_t)c * hplane + (size_t)ty * W + tx, v);
    }
}

__global__ __launch_bounds__(256) void zero_fill_floats_kernel(float* __restrict__ dst, long long count)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long long)gridDim.x * 256) dst[i] = 0.f;
}

__global__ __launch_bounds__(256) void zero_fill_kernel(float4* __restrict__ dst, long long quads)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < quads; i += (long long)gridDim.x * 256)
        dst[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// gradient of previous_output -> gradient of the raw prediction (clamps pass inside their closed range, as
// torch.clamp does; normalize as in backprop_fields)
__global__ __launch_bounds__(256) void recurrent_input_post_kernel(const RecurParams p, long long pixels)
{
    const size_t hplane = (size_t)16 * p.h * p.w;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < pixels; i += (long long)gridDim.x * 256) {
        const size_t b = (size_t)(i / (long long)hplane), at = (size_t)(i % (long long)hplane);
        const float* raw = p.raw + b * 6 * hplane + at;
        const float* go = p.gout + b * 6 * hplane + at;
        float* gr = p.graw + b * 6 * hplane + at;
        const float m = raw[0], nx = raw[hplane], ny = raw[2 * hplane], nz = raw[3 * hplane], dp = raw[4 * hplane], ao = raw[5 * hplane];
        gr[0] = (m >= -1.f && m <= 1.f) ? go[0] : 0.f;
        const float g1 = go[hplane], g2 = go[2 * hplane], g3 = go[3 * hplane];
        const float len = sqrtf(nx * nx + ny * ny + nz * nz);
        if (len > 1e-7f) {
            const float hx = nx / len, hy = ny / len, hz = nz / len;
            const float dot = hx * g1 + hy * g2 + hz * g3;
            gr[hplane] = (g1 - hx * dot) / len; gr[2 * hplane] = (g2 - hy * dot) / len; gr[3 * hplane] = (g3 - hz * dot) / len;
        } else {
            gr[hplane] = g1 / 1e-7f; gr[2 * hplane] = g2 / 1e-7f; gr[3 * hplane] = g3 / 1e-7f;
        }
        gr[4 * hplane] = (dp >= 0.f && dp <= 1.f) ? go[4 * hplane] : 0.f;
        gr[5 * hplane] = (ao >= 0.f && ao <= 1.f) ? go[5 * hplane] : 0.f;
    }
}

} // namespace

// One Adam step (torch.optim.Adam: no weight decay, no amsgrad) over FLAT parameter / gradient / moment buffers: the whole
// network's 911 046 parameters in one launch instead of ~100 per-tensor launches of the framework's optimizer.  `step` holds the
// number of steps taken so far (device, so that a HIP graph replays the right bias corrections); it is read here and
// incremented by adam_flat_count_kernel afterwards.  lr_dev (may be NULL) overrides lr: a learning rate a scheduler changes
// between graph replays.
__global__ __launch_bounds__(256) void adam_flat_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                         float* __restrict__ v, long long n, const float* lr_dev, float lr, float b1, float b2,
                                                         float eps, const float* __restrict__ step)
{
    const float t = step[0] + 1.0f;
    const float bc1 = 1.0f - powf(b1, t), bc2s = sqrtf(1.0f - powf(b2, t));
    const float step_size = (lr_dev ? lr_dev[0] : lr) / bc1;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float gi = g[i];
        const float mi = m[i] + (gi - m[i]) * (1.0f - b1);                  // exp_avg.lerp_(grad, 1 - beta1)
        const float vi = v[i] * b2 + (1.0f - b2) * gi * gi;                // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
        m[i] = mi; v[i] = vi;
        p[i] -= step_size * (mi / (sqrtf(vi) / bc2s + eps));
    }
}

__global__ void adam_flat_count_kernel(float* step) { step[0] += 1.0f; }

extern "C" {

int isrAdamFlatStep(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long n, const float* lr_dev, float lr,
                    float beta1, float beta2, float eps, float* step, void* stream)
{
    if (!params || !grads || !exp_avg || !exp_avg_sq || !step || n <= 0) return -1;
    const long long blocks = (n + 255) / 256;
    hipLaunchKernelGGL(adam_flat_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, (hipStream_t)stream,
                       params, grads, exp_avg, exp_avg_sq, n, lr_dev, lr, beta1, beta2, eps, (const float*)step);
    hipLaunchKernelGGL(adam_flat_count_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int isrUpsample2xForward(const float* x, float* y, long long planes, int h, int w, void* stream)
{
    if (!x || !y || planes <= 0 || h <= 0 || w <= 0) return -1;
    if ((w & 1) || (((uintptr_t)y) & 15)) {                                   // odd width (or a view that is not 16-byte aligned): pairs of output pixels, scalar stores
        const long long pairs = planes * (2LL * h) * w;
        long long blocks = (pairs + 255) / 256;
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(upsample2x_fwd_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, y, h, w, pairs);
        return hipGetLastError() == hipSuccess ? 0 : -2;
    }
    const long long quads = planes * (2LL * h) * (2 * w / 4);
    if (!(w & 3) && quads < 0x7fffffffLL && !(((uintptr_t)y | (uintptr_t)x) & 15)) {
        const unsigned count = (unsigned)(quads / 4);                         // a thread per four input pixels
        hipLaunchKernelGGL(upsample2x_fwd4_kernel, dim3((count + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, y, h, w, count);
        return hipGetLastError() == hipSuccess ? 0 : -2;
    }
    long long blocks = (quads + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(upsample2x_fwd_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, y, h, w, quads);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int isrUpsample2xBackward(const float* gy, float* gx, long long planes, int h, int w, void* stream)
{
    if (!gy || !gx || planes <= 0 || h <= 0 || w <= 0) return -1;
    const long long count = planes * h * w;
    if (!(w & 3) && count < 0x7fffffffLL && !(((uintptr_t)gy | (uintptr_t)gx) & 15)) {
        const unsigned n4 = (unsigned)(count / 4);
        hipLaunchKernelGGL(upsample2x_bwd4_kernel, dim3((n4 + 255) / 256), dim3(256), 0, (hipStream_t)stream, gy, gx, h, w, n4);
        return hipGetLastError() == hipSuccess ? 0 : -2;
    }
    long long blocks = (count + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(upsample2x_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, gy, gx, h, w, count);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

long long isrLossUnshadedWorkspace(void) { return (long long)1024 * LOSS_SLOTS * sizeof(float); }

int isrLossUnshadedForward(const float* gt, const float* pred, const float* prev, int N, int H, int W, int pad,
                           const float* weights15, unsigned enabled, const float* shading12, float ao_strength, int inverse_ao,
                           void* workspace, float* values16, void* stream)
{
    LossParams P;
    if (!workspace || !values16) return -1;
    if (int rc = fill_loss_params(P, gt, pred, prev, N, H, W, pad, weights15, enabled, shading12, ao_strength, inverse_ao)) return rc;
    P.partial = (float*)workspace; P.values = values16;
    hipLaunchKernelGGL(loss_unshaded_kernel<false>, dim3(P.blocks), dim3(256), 0, (hipStream_t)stream, P);
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, P);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int isrLossUnshadedBackward(const float* gt, const float* pred, const float* prev, int N, int H, int W, int pad,
                            const float* weights15, unsigned enabled, const float* shading12, float ao_strength, int inverse_ao,
                            const float* gvalues16, float* gpred, float* gprev, void* stream)
{
    LossParams P;
    if (!gvalues16 || !gpred) return -1;
    if (gprev && !prev) return -1;
    if (int rc = fill_loss_params(P, gt, pred, prev, N, H, W, pad, weights15, enabled, shading12, ao_strength, inverse_ao)) return rc;
    P.gout = gvalues16; P.gpred = gpred; P.gprev = gprev;
    hipLaunchKernelGGL(loss_unshaded_kernel<true>, dim3(P.blocks), dim3(256), 0, (hipStream_t)stream, P);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int isrRecurrentInputForward(const float* prev_raw, const float* input, const float* flow, float* net_input, float* warped,
                             int B, int h, int w, long long inputBatchStride, long long flowBatchStride, void* stream)
{
    if (!prev_raw || !input || !flow || !net_input || !warped || B <= 0 || h <= 0 || w <= 0 || B > 65535 || h > 65535) return -1;
    if (inputBatchStride < 5LL * h * w || flowBatchStride < 2LL * h * w) return -1;
    RecurParams p = {};
    p.raw = prev_raw; p.input = input; p.flow = flow; p.netin = net_input; p.warped = warped;
    p.inStride = inputBatchStride; p.flowStride = flowBatchStride; p.B = B; p.h = h; p.w = w;
    hipLaunchKernelGGL(recurrent_input_fwd_kernel, dim3((4 * w + 255) / 256, h, B), dim3(256), 0, (hipStream_t)stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int isrRecurrentInputBackward(const float* prev_raw, const float* flow, const float* g_net_input, const float* g_warped,
                              float* scratch, float* g_prev_raw, int B, int h, int w, long long flowBatchStride, void* stream)
{
    if (!prev_raw || !flow || !scratch || !g_prev_raw || B <= 0 || h <= 0 || w <= 0 || B > 65535 || h > 65535) return -1;
    if (flowBatchStride < 2LL * h * w) return -1;
    RecurParams p = {};
    p.raw = prev_raw; p.flow = flow; p.flowStride = flowBatchStride; p.B = B; p.h = h; p.w = w;
    p.gnetin = g_net_input; p.gwarped = g_warped; p.gout = scratch; p.graw = g_prev_raw;
    hipStream_t s = (hipStream_t)stream;
    const long long pixels = 16LL * B * h * w;
    // (a kernel, not hipMemsetAsync: captured in a HIP graph the memset node did not take effect on replay)
    {
        const long long quads = pixels * 6 / 4;              // pixels is a multiple of 16
        long long zb = (quads + 255) / 256;
        if (zb > 4096) zb = 4096;
        hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)zb), dim3(256), 0, s, reinterpret_cast<float4*>(scratch), quads);
    }
    if (g_net_input || g_warped)
        hipLaunchKernelGGL(recurrent_input_scatter_lds_kernel, dim3((4 * w + SC_TW - 1) / SC_TW, (4 * h + SC_TH - 1) / SC_TH, B), dim3(SC_TH * SC_TW), 0, s, p);
    long long blocks = (pixels + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(recurrent_input_post_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p, pixels);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int isrReconResidualForward(const float* y, const float* x, float* out, int N, int Cout, int Cin, int k, int h, int w, void* stream)
{
    if (!y || !x || !out || N <= 0 || Cout <= 0 || Cin <= 0 || k < 0 || k > Cout || k > Cin || h <= 0 || w <= 0 || N > 65535 || 4LL * h > 65535) return -1;
    hipLaunchKernelGGL(recon_residual_fwd_kernel, dim3((4 * w + 255) / 256, 4 * h, N), dim3(256), 0, (hipStream_t)stream, y, x, out, Cout, Cin, k, h, w);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int isrReconResidualBackward(const float* gy, float* gx, int N, int Cout, int Cin, int k, int h, int w, void* stream)
{
    if (!gy || !gx || N <= 0 || Cout <= 0 || Cin <= 0 || k < 0 || k > Cout || k > Cin || h <= 0 || w <= 0) return -1;
    // (a kernel, not hipMemsetAsync: captured in a HIP graph the memset node did not take effect on replay)
    {
        const long long total = (long long)N * Cin * h * w;
        long long zb = (total + 255) / 256;
        if (zb > 4096) zb = 4096;
        hipLaunchKernelGGL(zero_fill_floats_kernel, dim3((unsigned)zb), dim3(256), 0, (hipStream_t)stream, gx, total);
    }
    const long long count8 = (long long)N * k * h * w * 8;
    if (count8 == 0) return hipGetLastError() == hipSuccess ? 0 : -2;
    const long long blocks = (count8 + 255) / 256;
    if (blocks > 0x7fffffffLL) return -1;
    hipLaunchKernelGGL(recon_residual_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, gy, gx, Cout, Cin, k, h, w, count8);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

} // extern "C"
