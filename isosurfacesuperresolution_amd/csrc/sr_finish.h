// End of an inference frame, per high-resolution pixel (shared by finish_frame_kernel and the fused epilogue of the
// final 64 -> 6 convolution): residual reconstruction, clamp / normalise, screen-space shading.
#pragma once
#include <hip/hip_runtime.h>

// bilinear source coordinate, align_corners=False (ATen area_pixel_compute_source_index)
__device__ __forceinline__ void isr_src_index(int dst, float scale, int in_size, int& i0, int& i1, float& l1)
{
    float s = ((float)dst + 0.5f) * scale - 0.5f;
    s = s < 0.f ? 0.f : s;
    i0 = (int)s;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l1 = s - (float)i0;
}

struct FinishParams {
    const float* raw;       // [6][H][W] network output before the residual reconstruction (unused by the fused epilogue)
    const float* net_in;    // [Cin][h][w] network input (first 5 channels are used)
    float* next_prev;       // [6][H][W] clamped / normalised frame (next frame's "previous")
    float* rgb;             // [3][H][W] shaded colour (may be NULL)
    int h, w;
    float ambient[3], diffuse[3], specular[3], light[3], material[3], background[3];
    int exponent;
    float ao_strength;
    int inverse_ao, enable_specular;
};

// v[0..5]: the conv output at high-resolution pixel (X, Y)
__device__ __forceinline__ void isr_finish_pixel(const FinishParams& p, int X, int Y, float (&v)[6])
{
    const int H = 4 * p.h, W = 4 * p.w;
    const size_t hplane = (size_t)H * W, lplane = (size_t)p.h * p.w;
    const size_t pix = (size_t)Y * W + X;
    int y0, y1, x0, x1; float ly, lx;
    isr_src_index(Y, 0.25f, p.h, y0, y1, ly);
    isr_src_index(X, 0.25f, p.w, x0, x1, lx);
    const float hy = 1.f - ly, hx = 1.f - lx;
#pragma unroll
    for (int c = 0; c < 5; ++c) {       // out[:, :5] += bilinear_resize(x[:, :5])   (enhancenet.py:65-78)
        const float* q = p.net_in + (size_t)c * lplane;
        v[c] += hy * (hx * q[y0 * p.w + x0] + lx * q[y0 * p.w + x1]) + ly * (hx * q[y1 * p.w + x0] + lx * q[y1 * p.w + x1]);
    }
    // mainGUI.py:594-599
    const float mask = fminf(fmaxf(v[0], -1.f), 1.f);
    const float nlen = fmaxf(sqrtf(v[1] * v[1] + v[2] * v[2] + v[3] * v[3]), 1e-7f);
    const float nx = v[1] / nlen, ny = v[2] / nlen, nz = v[3] / nlen;
    const float depth = fminf(fmaxf(v[4], 0.f), 1.f);
    const float ao = fminf(fmaxf(v[5], 0.f), 1.f);
    p.next_prev[0 * hplane + pix] = mask;
    p.next_prev[1 * hplane + pix] = nx;
    p.next_prev[2 * hplane + pix] = ny;
    p.next_prev[3 * hplane + pix] = nz;
    p.next_prev[4 * hplane + pix] = depth;
    p.next_prev[5 * hplane + pix] = ao;
    if (!p.rgb) return;
    // utils/shading.py:148-191
    const float a = p.inverse_ao ? 1.0f - ao : ao;
    const float aof = p.ao_strength * fminf(fmaxf(a, 0.f), 1.f) + (1.0f - p.ao_strength);
    const float ndl = p.light[0] * nx + p.light[1] * ny + p.light[2] * nz;
    float spec = 0.f;
    if (p.enable_specular) {
        const float rz = 2.f * ndl * nz - p.light[2];           // eye direction is (0,0,1) everywhere
        const float base = fminf(fmaxf(rz, 0.f), 1.f);
        float pw = 1.f;
        for (int e = 0; e < p.exponent; ++e) pw *= base;
        spec = ((float)(p.exponent + 2) / (2.0f * 3.14159265358979323846f)) * pw;
    }
    const float t = fminf(fmaxf(mask * 0.5f + 0.5f, 0.f), 1.f);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float col = p.ambient[k] * p.material[k] + (p.diffuse[k] * p.material[k]) * fabsf(ndl) + spec * p.specular[k];
        col *= aof;
        col = p.background[k] + t * (col - p.background[k]);
        p.rgb[(size_t)k * hplane + pix] = fminf(fmaxf(col, 0.f), 1.f);
    }
}

inline void isr_fill_finish_params(FinishParams& p, const float* raw, const float* net_input, float* next_prev, float* rgb, int h, int w,
                                   const float* shading24, int exponent, float ao_strength, int inverse_ao, int enable_specular)
{
    p.raw = raw; p.net_in = net_input; p.next_prev = next_prev; p.rgb = rgb; p.h = h; p.w = w;
    for (int k = 0; k < 3; ++k) {
        p.ambient[k] = rgb ? shading24[k] : 0.f; p.diffuse[k] = rgb ? shading24[3 + k] : 0.f;
        p.specular[k] = rgb ? shading24[6 + k] : 0.f; p.light[k] = rgb ? shading24[9 + k] : 0.f;
        p.material[k] = rgb ? shading24[12 + k] : 0.f; p.background[k] = rgb ? shading24[15 + k] : 0.f;
    }
    p.exponent = exponent; p.ao_strength = ao_strength; p.inverse_ao = inverse_ao; p.enable_specular = enable_specular;
}
