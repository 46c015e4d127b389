// The temporal input path's arithmetic, shared by the inference kernels (sr_frame.hip) and the training kernels (sr_train.hip).
#pragma once
#include <hip/hip_runtime.h>

// The temporal input path -- flow hole filling, resize of the flow, warp of the previous frame -- is DEFINED operation by operation in
// the package's module path (inference/flowfill.py, models/videotools.py: elementwise torch operations, one IEEE rounding each) and
// computed here with the same operations in the same order and NO contraction into FMAs (`#pragma clang fp contract(off)` in every
// function that takes part): same inputs, same bits.  Why it matters: the reference's warp goes through normalised coordinates, a
// rounding of 6e-8 there is 6e-5 pixels at 1080p and 1e-4 in the warped value across a silhouette edge; two fp32 evaluations that
// round differently hand the network inputs that differ by that much, and the recurrence multiplies it frame by frame
// (tests/test_recurrence_gpu.py, DESIGN "temporal input path").
__device__ __forceinline__ void isr_src_index_rn(int dst, float scale, int in_size, int& i0, int& i1, float& l1)
{
#pragma clang fp contract(off)
    float s = ((float)dst + 0.5f) * scale;      // (dst + 0.5) * scale - 0.5, clamped at 0 (ATen area_pixel_compute_source_index)
    s = s - 0.5f;
    s = s < 0.f ? 0.f : s;
    i0 = (int)s;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l1 = s - (float)i0;
}

// hy (hx a + lx b) + ly (hx c + lx d): seven roundings in this order (models/videotools.py: bilinear_taps)
__device__ __forceinline__ float isr_bilerp_rn(float hy, float hx, float ly, float lx, float a, float b, float c, float d)
{
#pragma clang fp contract(off)
    const float ha = hx * a, lb = lx * b, hc = hx * c, ld = lx * d;
    const float t0 = ha + lb, t1 = hc + ld;
    const float u0 = hy * t0, u1 = ly * t1;
    return u0 + u1;
}

// linspace(-1, 1, n)[i] as models/videotools.py: pixel_grid defines it: 2 i / (n - 1) - 1 in double (multiply, divide, subtract),
// rounded once to float
__device__ __forceinline__ float isr_pixel_grid(int i, int n)
{
#pragma clang fp contract(off)
    const double twice = (double)i * 2.0;
    const double q = twice / (double)(n > 1 ? n - 1 : 1);
    return (float)(q - 1.0);
}

