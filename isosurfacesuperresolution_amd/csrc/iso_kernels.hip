// Isosurface ray-march kernels for MI355X (gfx950, wave64).
//
// What is computed (values): the reference's CPU tracer -- hierarchical DDA over 4096/128/8-voxel
// nodes, voxel DDA with zero-crossing test and 5 bisection steps, node-centred trilinear
// interpolation, +-1 voxel central-difference normal, two-sided Phong, camera-space flow
// (CPURenderer/IsoVolumeRayTracer.h:37-46,81-114,274-309,502-551; PhongShader.h:27-38;
//  CPURenderer.cpp:726-737).  Layout of the result: GPURendererDirect's interleaved HWC
// 12-channel buffer (GPURendererDirect/render_kernel.cu:254-265).
//
// How (MI355X-native, nothing in common with the CUDA/GVDB kernel): the volume lives in HBM as
// 9^3 "apron bricks" (8^3 voxels + the +1 layer trilinear needs), one wave64 owns one 8x8 pixel
// tile, ray state is fp64 (78 TF/s vector fp64 on CDNA4 makes the reference's double DDA
// affordable), samples are fp32.  This TU is compiled with -ffp-contract=off: the hit mask must be
// bit-identical to the IEEE CPU restatement, so no FMA contraction is allowed here.
//
// Variant 0: every lane gathers its 8 corners from the brick in global memory (L2/MALL resident).
// Variant 1: the wave cooperatively stages the brick most lanes need into LDS (ballot vote),
//            lanes whose ray is inside that brick march out of LDS; see iso_render_lds below.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <float.h>
#include <stdint.h>

#include "iso_params.h"

namespace {

struct Ray {
    double ex, ey, ez, dx, dy, dz, ix, iy, iz, t0, t1;
};

struct DDA {
    double t0, t1, nx, ny, nz, dx, dy, dz;
    int vx, vy, vz, sx, sy, sz;
};

__device__ __forceinline__ void ray_at(const Ray& r, double t, double& px, double& py, double& pz)
{
    px = r.ex + r.dx * t;
    py = r.ey + r.dy * t;
    pz = r.ez + r.dz * t;
}

// TP/openvdb/math/DDA.h:79-103
template <int LOG2DIM>
__device__ __forceinline__ void dda_init(DDA& d, const Ray& r)
{
    constexpr int DIM = 1 << LOG2DIM;
    d.t0 = r.t0;
    d.t1 = r.t1;
    double px, py, pz;
    ray_at(r, d.t0, px, py, pz);
    d.vx = ((int)floor(px)) & (~(DIM - 1));
    d.vy = ((int)floor(py)) & (~(DIM - 1));
    d.vz = ((int)floor(pz)) & (~(DIM - 1));
#define ISO_AXIS(V, S, N, D, P, DIR, INV)                                   \
    if (DIR == 0.0) { S = 0; N = DBL_MAX; D = DBL_MAX; }                      \
    else if (INV > 0) { S = DIM; N = d.t0 + ((double)(V + DIM) - P) * INV; D = (double)S * INV; } \
    else { S = -DIM; N = d.t0 + ((double)V - P) * INV; D = (double)S * INV; }
    ISO_AXIS(d.vx, d.sx, d.nx, d.dx, px, r.dx, r.ix)
    ISO_AXIS(d.vy, d.sy, d.ny, d.dy, py, r.dy, r.iy)
    ISO_AXIS(d.vz, d.sz, d.nz, d.dz, pz, r.dz, r.iz)
#undef ISO_AXIS
}

// TP/openvdb/math/DDA.h:139 with Math.h:623-626
__device__ __forceinline__ double dda_next(const DDA& d)
{
    double a = d.t1 < d.nx ? d.t1 : d.nx;
    double b = d.ny < d.nz ? d.ny : d.nz;
    return b < a ? b : a;
}

// TP/openvdb/math/DDA.h:111-117 with the MinIndex tie table of Math.h:893-902
__device__ __forceinline__ bool dda_step(DDA& d)
{
    const int key = ((d.nx < d.ny) << 2) + ((d.nx < d.nz) << 1) + (d.ny < d.nz);
    if (key >= 6) { d.t0 = d.nx; d.nx += d.dx; d.vx += d.sx; }
    else if (key == 1 || key == 3) { d.t0 = d.ny; d.ny += d.dy; d.vy += d.sy; }
    else { d.t0 = d.nz; d.nz += d.dz; d.vz += d.sz; }
    return d.t0 <= d.t1;
}

// ---- volume access -------------------------------------------------------------------------

// Coordinates are GLOBAL index coordinates throughout; P.org (a multiple of 8, zero unless the volume is one tile of a
// larger one) is subtracted only where a table or a brick of the locally stored region is addressed.
__device__ __forceinline__ float voxel_value(const IsoRenderParams& P, int x, int y, int z)
{
    x -= P.org[0]; y -= P.org[1]; z -= P.org[2];
    if ((unsigned)x >= (unsigned)P.nx || (unsigned)y >= (unsigned)P.ny || (unsigned)z >= (unsigned)P.nz) return 0.0f;
    const int s = P.slot[((z >> 3) * P.nby + (y >> 3)) * P.nbx + (x >> 3)];
    if (s < 0) return 0.0f;
    return P.bricks[(size_t)s * ISO_BRICK_STRIDE + ((z & 7) * 9 + (y & 7)) * 9 + (x & 7)];
}

// TP/openvdb/math/Stencils.h:335-354 -- nested lerps z, then y, then x, all in float
__device__ __forceinline__ float trilerp(float V000, float V001, float V010, float V011,
                                         float V100, float V101, float V110, float V111,
                                         float u, float v, float w)
{
    float A = V000 + (V001 - V000) * w;
    float B = V010 + (V011 - V010) * w;
    float C = A + (B - A) * v;
    A = V100 + (V101 - V100) * w;
    B = V110 + (V111 - V110) * w;
    float D = A + (B - A) * v;
    return C + (D - C) * u;
}

// the corners (x, x+1) are neighbours in the brick: four 8-byte loads (dword aligned) instead of eight 4-byte ones
typedef float float2u __attribute__((ext_vector_type(2), aligned(4)));
__device__ __forceinline__ float interp_from_brick(const float* __restrict__ b, int lx, int ly, int lz, float u, float v, float w)
{
    const float* q = b + (lz * 9 + ly) * 9 + lx;
    const float2u c00 = *reinterpret_cast<const float2u*>(q), c01 = *reinterpret_cast<const float2u*>(q + 81);
    const float2u c10 = *reinterpret_cast<const float2u*>(q + 9), c11 = *reinterpret_cast<const float2u*>(q + 90);
    return trilerp(c00.x, c01.x, c10.x, c11.x, c00.y, c01.y, c10.y, c11.y, u, v, w);
}

// Trilinear sample at index-space position (px,py,pz); Stencils.h:110-114 (cell = floor of the
// double position, fractions from the float-narrowed position).
__device__ __forceinline__ float interp_global(const IsoRenderParams& P, double px, double py, double pz)
{
    const int cx = (int)floor(px), cy = (int)floor(py), cz = (int)floor(pz);
    const float u = (float)px - (float)cx;
    const float v = (float)py - (float)cy;
    const float w = (float)pz - (float)cz;
    const int lx = cx - P.org[0], ly = cy - P.org[1], lz = cz - P.org[2];
    if ((unsigned)lx < (unsigned)P.nx && (unsigned)ly < (unsigned)P.ny && (unsigned)lz < (unsigned)P.nz) {
        const int s = P.slot[((lz >> 3) * P.nby + (ly >> 3)) * P.nbx + (lx >> 3)];
        if (s < 0) return 0.0f;   // all 27 candidate corners are zero: 0 + (0-0)*w ... == +0
        return interp_from_brick(P.bricks + (size_t)s * ISO_BRICK_STRIDE, lx & 7, ly & 7, lz & 7, u, v, w);
    }
    // cell on or outside the low/high border of the grid: per-corner fetch (background 0 outside)
    return trilerp(voxel_value(P, cx, cy, cz), voxel_value(P, cx, cy, cz + 1),
                   voxel_value(P, cx, cy + 1, cz), voxel_value(P, cx, cy + 1, cz + 1),
                   voxel_value(P, cx + 1, cy, cz), voxel_value(P, cx + 1, cy, cz + 1),
                   voxel_value(P, cx + 1, cy + 1, cz), voxel_value(P, cx + 1, cy + 1, cz + 1), u, v, w);
}

// IsoVolumeRayTracer.h:66-71 (float - double -> double -> float)
__device__ __forceinline__ float interp_value(const IsoRenderParams& P, const Ray& r, double t)
{
    double px, py, pz;
    ray_at(r, t, px, py, pz);
    return (float)((double)interp_global(P, px, py, pz) - P.iso);
}

// Per-leaf / per-node flags for the frame's isovalue (iso_march_flags, refreshed by the host whenever the isovalue or the
// volume changes): bit 0 = the leaf / node exists, bit 1 = it exists and must be marched.
// In a tile, P.leaf holds "leaf exists AND is owned by this tile": leaves of the halo are walked past like empty space.
// Min/max skipping, exact: every sample the voxel DDA of a leaf can take reads voxels of [8b-1, 8b+9]^3 only (cells
// 8b-1 .. 8b+8: a position may sit a rounding error outside the leaf's faces), and a trilinear value stays inside the
// range of its 8 corners up to ~11 ulp of the seven float lerps.  If the isovalue lies outside [min, max] of that
// neighbourhood by more than the pad, (value - iso) has one strict sign along the whole march, the reference's
// `v0 * v1 <= 0` never fires, and stepping over the leaf is the same computation.  Long rays that cross the thin
// low-density fringe or the dense core without meeting the surface were the tail the whole frame waited for.
// The same one level up: node1Range = (min, max) over the ranges of the node's existing leaves.  If the isovalue lies
// outside it, no leaf of the node can be marched, and since the leaf-level DDA is re-initialised per node
// (IsoVolumeRayTracer.h:37-46) stepping over the whole node changes nothing downstream.
__device__ __forceinline__ bool range_may_cross(const float* mm, double iso)
{
    const double lo = (double)mm[0], hi = (double)mm[1];
    const double pad = 4e-6 * fmax(fabs(lo), fabs(hi));
    return !(iso < lo - pad || iso > hi + pad);
}

__global__ __launch_bounds__(256) void iso_march_flags(const uint8_t* __restrict__ exists, const float* __restrict__ range, int n, double iso,
                                                       uint8_t* __restrict__ flags)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) flags[i] = exists[i] ? (uint8_t)(1 | (range_may_cross(range + 2 * (size_t)i, iso) ? 2 : 0)) : (uint8_t)0;
}

__device__ __forceinline__ int leaf_flags(const IsoRenderParams& P, int x, int y, int z)
{
    x -= P.org[0]; y -= P.org[1]; z -= P.org[2];
    if ((unsigned)x >= (unsigned)P.nx || (unsigned)y >= (unsigned)P.ny || (unsigned)z >= (unsigned)P.nz) return 0;
    return P.leafMarch[((z >> 3) * P.nby + (y >> 3)) * P.nbx + (x >> 3)];
}

// node tables are indexed by GLOBAL 128^3 node coordinates relative to the first node the stored region overlaps (P.n1o)
__device__ __forceinline__ int node1_flags(const IsoRenderParams& P, int x, int y, int z)
{
    const int ax = (x >> 7) - P.n1o[0], ay = (y >> 7) - P.n1o[1], az = (z >> 7) - P.n1o[2];
    if ((unsigned)ax >= (unsigned)P.n1x || (unsigned)ay >= (unsigned)P.n1y || (unsigned)az >= (unsigned)P.n1z) return 0;
    return P.node1March[(az * P.n1y + ay) * P.n1x + ax];
}

__device__ __forceinline__ bool has_node2(const IsoRenderParams& P, int x, int y, int z)
{
    return P.any_leaf && x == 0 && y == 0 && z == 0;
}

// Diagnostics (tools/lab/raymarch_stats.py): what one ray did.  The product kernels pass NoTrace, which compiles to nothing.
struct NoTrace { __device__ __forceinline__ void sample(int) {} __device__ __forceinline__ void leaf(bool) {} };
struct Trace {
    int samples = 0, leaves = 0, skipped = 0;
    __device__ __forceinline__ void sample(int n) { samples += n; }
    __device__ __forceinline__ void leaf(bool marched) { if (marched) ++leaves; else ++skipped; }
};

// IsoVolumeRayTracer.h:81-114
template <typename TR>
__device__ __forceinline__ bool hits_voxel(const IsoRenderParams& P, const Ray& ray, double& time, TR& tr)
{
    DDA d;
    dda_init<0>(d, ray);
    double t0 = d.t0;
    float v0 = interp_value(P, ray, t0);
    tr.sample(1);
    do {
        double t1 = dda_next(d);
        float v1 = interp_value(P, ray, t1);
        tr.sample(1);
        if (v0 * v1 <= 0.0f) {
            double t = 0.5 * (t0 + t1);
            for (int i = 0; i < 5; ++i) {
                float v2 = interp_value(P, ray, t);
                if (v0 * v2 <= 0.0f) t1 = t;
                else { t0 = t; v0 = v2; }
                t = 0.5 * (t0 + t1);
            }
            tr.sample(5);
            time = t;
            return true;
        }
        t0 = t1;
        v0 = v1;
    } while (dda_step(d));
    return false;
}
__device__ __forceinline__ bool hits_voxel(const IsoRenderParams& P, const Ray& ray, double& time)
{
    NoTrace nt;
    return hits_voxel(P, ray, time, nt);
}

// IsoVolumeRayTracer.h:37-46 for node sizes 4096, 128, 8
template <typename TR>
__device__ bool hits_hierarchy(const IsoRenderParams& P, Ray& ray, double& time, TR& tr)
{
    DDA d2;
    dda_init<12>(d2, ray);
    do {
        if (has_node2(P, d2.vx, d2.vy, d2.vz)) {
            ray.t0 = d2.t0; ray.t1 = dda_next(d2);
            DDA d1;
            dda_init<7>(d1, ray);
            do {
                if (node1_flags(P, d1.vx, d1.vy, d1.vz) & 2) {
                    ray.t0 = d1.t0; ray.t1 = dda_next(d1);
                    DDA d0;
                    dda_init<3>(d0, ray);
                    do {
                        const int lf = leaf_flags(P, d0.vx, d0.vy, d0.vz);
                        if (lf) {
                            const bool march = (lf & 2) != 0;
                            tr.leaf(march);
                            if (march) {
                                ray.t0 = d0.t0; ray.t1 = dda_next(d0);
                                if (hits_voxel(P, ray, time, tr)) return true;
                            }
                        }
                    } while (dda_step(d0));
                }
            } while (dda_step(d1));
        }
    } while (dda_step(d2));
    return false;
}
__device__ bool hits_hierarchy(const IsoRenderParams& P, Ray& ray, double& time)
{
    NoTrace nt;
    return hits_hierarchy(P, ray, time, nt);
}

// ---- the same traversal as ONE flat per-lane state machine ------------------------------------------------------------
// hits_hierarchy() nests four loops; a wave runs them in lock step, so every round of the leaf loop lasts as long as the
// longest voxel march any of its 64 rays does in that round while the rays whose leaf is skipped wait: the heaviest tile of
// the bench frame costs 1.13 M cycles although its busiest ray, marched alone, costs 0.26 M (tools/lab/raymarch_lone.py).
// Here every lane carries a state and each iteration of the single loop does at most WALK hierarchy steps and then ONE
// trilinear sample -- the leaf's first sample, a march sample or a bisection sample, all through the same code -- so a wave
// lasts about as long as its busiest ray.  Same operations on the same values in the same per-ray order: bit-identical.
// A level's DDA keeps only what changes: the per-axis step and delta are +-DIM and DIM * |1/dir| (DDA.h:79-103), recomputed
// from the ray where an axis is stepped; the end of a level's interval is the parent's dda_next.  Level 2 (4096^3 nodes)
// has one node, so it is a prologue.
struct Lvl { double t0, nx, ny, nz; int vx, vy, vz; };

template <int LOG2DIM>
__device__ __forceinline__ void lvl_init(Lvl& d, const Ray& r, double t0)
{
    constexpr int DIM = 1 << LOG2DIM;
    d.t0 = t0;
    double px, py, pz;
    ray_at(r, d.t0, px, py, pz);
    d.vx = ((int)floor(px)) & (~(DIM - 1));
    d.vy = ((int)floor(py)) & (~(DIM - 1));
    d.vz = ((int)floor(pz)) & (~(DIM - 1));
    // DDA.h:79-103 without branches: the far face for a positive direction, the near one otherwise
#define ISO_AXIS(V, N, P, DIR, INV)                                          \
    {                                                                         \
        const double n_ = d.t0 + ((double)(V + (INV > 0 ? DIM : 0)) - P) * INV; \
        N = DIR == 0.0 ? DBL_MAX : n_;                                        \
    }
    ISO_AXIS(d.vx, d.nx, px, r.dx, r.ix)
    ISO_AXIS(d.vy, d.ny, py, r.dy, r.iy)
    ISO_AXIS(d.vz, d.nz, pz, r.dz, r.iz)
#undef ISO_AXIS
}

template <int DIM>
__device__ __forceinline__ void lvl_axis(double dir, double inv, int& S, double& D)
{
    // (+-DIM) * inv with the sign of inv == DIM * |inv| exactly (a product's magnitude does not depend on the signs), so the
    // delta needs no int->double conversion and no branch: |inv| is a source modifier, DIM == 1 needs no multiply at all
    S = dir == 0.0 ? 0 : (inv > 0 ? DIM : -DIM);
    D = dir == 0.0 ? DBL_MAX : (double)DIM * fabs(inv);
}

__device__ __forceinline__ double lvl_next(const Lvl& d, double t1)
{
    double a = t1 < d.nx ? t1 : d.nx;
    double b = d.ny < d.nz ? d.ny : d.nz;
    return b < a ? b : a;
}

template <int LOG2DIM>
__device__ __forceinline__ bool lvl_step(Lvl& d, const Ray& r, double t1)
{
    constexpr int DIM = 1 << LOG2DIM;
    const int key = ((d.nx < d.ny) << 2) + ((d.nx < d.nz) << 1) + (d.ny < d.nz);
    int S; double D;
    if (key >= 6) { lvl_axis<DIM>(r.dx, r.ix, S, D); d.t0 = d.nx; d.nx += D; d.vx += S; }
    else if (key == 1 || key == 3) { lvl_axis<DIM>(r.dy, r.iy, S, D); d.t0 = d.ny; d.ny += D; d.vy += S; }
    else { lvl_axis<DIM>(r.dz, r.iz, S, D); d.t0 = d.nz; d.nz += D; d.vz += S; }
    return d.t0 <= t1;
}

enum { FS_STEP1 = 0, FS_NODE1 = 1, FS_LEAF = 2, FS_ENTER = 3, FS_MARCH = 4, FS_BISECT = 5, FS_MISS = 10, FS_HIT = 11 };

template <int WALK, typename TR>
__device__ __forceinline__ bool hits_flat(const IsoRenderParams& P, const Ray& ray, double& time, TR& tr)
{
    double T2;                       // end of the level-2 node's interval = d1.t1
    Lvl d1;
    {
        DDA d2;
        dda_init<12>(d2, ray);
        bool found = false;
        do {
            if (has_node2(P, d2.vx, d2.vy, d2.vz)) { found = true; break; }
        } while (dda_step(d2));
        if (!found) return false;    // no other level-2 node exists: once the walk below leaves this one, the ray has missed
        T2 = dda_next(d2);
        lvl_init<7>(d1, ray, d2.t0);
    }
    Lvl d0, dv;
    d0.t0 = d0.nx = d0.ny = d0.nz = 0.0; d0.vx = d0.vy = d0.vz = 0;
    dv = d0;
    double t1_0 = 0.0, t1_v = 0.0;   // ends of the current node's / leaf's interval (dda_next of the parent level)
    double b1 = 0.0;                 // bisection: [dv.t0, b1]
    float v0 = 0.0f;
    int st = FS_NODE1;
    while (st < FS_MISS) {
#pragma unroll
        for (int k = 0; k < WALK; ++k) {
            if (st == FS_STEP1) st = lvl_step<7>(d1, ray, T2) ? FS_NODE1 : FS_MISS;
            if (st == FS_NODE1) {
                if (node1_flags(P, d1.vx, d1.vy, d1.vz) & 2) {
                    t1_0 = lvl_next(d1, T2);
                    lvl_init<3>(d0, ray, d1.t0);
                    st = FS_LEAF;
                } else st = FS_STEP1;
            } else if (st == FS_LEAF) {
                const int lf = leaf_flags(P, d0.vx, d0.vy, d0.vz);
                const bool march = (lf & 2) != 0;
                if (lf) tr.leaf(march);
                if (march) {
                    t1_v = lvl_next(d0, t1_0);
                    lvl_init<0>(dv, ray, d0.t0);
                    st = FS_ENTER;
                } else if (!lvl_step<3>(d0, ray, t1_0)) st = FS_STEP1;
            }
        }
        if (st >= FS_ENTER && st < FS_MISS) {
            double t;
            if (st == FS_ENTER) t = dv.t0;
            else if (st == FS_MARCH) t = lvl_next(dv, t1_v);
            else t = 0.5 * (dv.t0 + b1);
            const float v = interp_value(P, ray, t);
            tr.sample(1);
            if (st == FS_ENTER) { v0 = v; st = FS_MARCH; }
            else if (st == FS_MARCH) {
                if (v0 * v <= 0.0f) { b1 = t; st = FS_BISECT; }             // the crossing lies in [dv.t0, t]
                else {
                    v0 = v;
                    if (!lvl_step<0>(dv, ray, t1_v)) st = lvl_step<3>(d0, ray, t1_0) ? FS_LEAF : FS_STEP1;
                }
            } else {
                if (v0 * v <= 0.0f) b1 = t;
                else { dv.t0 = t; v0 = v; }
                if (++st == FS_BISECT + 5) { time = 0.5 * (dv.t0 + b1); st = FS_HIT; }
            }
        }
    }
    return st == FS_HIT;
}

// Two samples of one ray with their gathers in flight together (interp_value twice: same operations, same results).
// B is fetched only where hasB.
struct SamplePos { int cx, cy, cz, idx, loc; float u, v, w; bool inb; };

__device__ __forceinline__ void sample_pos(const IsoRenderParams& P, const Ray& r, double t, SamplePos& q)
{
    double px, py, pz;
    ray_at(r, t, px, py, pz);
    q.cx = (int)floor(px); q.cy = (int)floor(py); q.cz = (int)floor(pz);
    q.u = (float)px - (float)q.cx;
    q.v = (float)py - (float)q.cy;
    q.w = (float)pz - (float)q.cz;
    const int lx = q.cx - P.org[0], ly = q.cy - P.org[1], lz = q.cz - P.org[2];
    q.inb = (unsigned)lx < (unsigned)P.nx && (unsigned)ly < (unsigned)P.ny && (unsigned)lz < (unsigned)P.nz;
    q.idx = ((lz >> 3) * P.nby + (ly >> 3)) * P.nbx + (lx >> 3);
    q.loc = ((lz & 7) * 9 + (ly & 7)) * 9 + (lx & 7);
}

__device__ __forceinline__ float sample_border(const IsoRenderParams& P, const SamplePos& q)
{
    return trilerp(voxel_value(P, q.cx, q.cy, q.cz), voxel_value(P, q.cx, q.cy, q.cz + 1),
                   voxel_value(P, q.cx, q.cy + 1, q.cz), voxel_value(P, q.cx, q.cy + 1, q.cz + 1),
                   voxel_value(P, q.cx + 1, q.cy, q.cz), voxel_value(P, q.cx + 1, q.cy, q.cz + 1),
                   voxel_value(P, q.cx + 1, q.cy + 1, q.cz), voxel_value(P, q.cx + 1, q.cy + 1, q.cz + 1), q.u, q.v, q.w);
}

__device__ __forceinline__ void interp_pair(const IsoRenderParams& P, const Ray& r, double tA, double tB, bool hasB, float& vA, float& vB)
{
    SamplePos a, b;
    sample_pos(P, r, tA, a);
    sample_pos(P, r, tB, b);
    int sA = -1, sB = -1;
    if (a.inb) sA = P.slot[a.idx];
    if (hasB && b.inb) sB = P.slot[b.idx];
    float2u a00, a01, a10, a11, b00, b01, b10, b11;
    a00 = a01 = a10 = a11 = b00 = b01 = b10 = b11 = float2u{0.0f, 0.0f};
    if (sA >= 0) {
        const float* q = P.bricks + (size_t)sA * ISO_BRICK_STRIDE + a.loc;
        a00 = *reinterpret_cast<const float2u*>(q); a01 = *reinterpret_cast<const float2u*>(q + 81);
        a10 = *reinterpret_cast<const float2u*>(q + 9); a11 = *reinterpret_cast<const float2u*>(q + 90);
    }
    if (sB >= 0) {
        const float* q = P.bricks + (size_t)sB * ISO_BRICK_STRIDE + b.loc;
        b00 = *reinterpret_cast<const float2u*>(q); b01 = *reinterpret_cast<const float2u*>(q + 81);
        b10 = *reinterpret_cast<const float2u*>(q + 9); b11 = *reinterpret_cast<const float2u*>(q + 90);
    }
    // an empty brick position reads as zero: 0 + (0 - 0) * w ... == +0, like the early return of interp_global
    float fa = sA >= 0 ? trilerp(a00.x, a01.x, a10.x, a11.x, a00.y, a01.y, a10.y, a11.y, a.u, a.v, a.w) : 0.0f;
    float fb = sB >= 0 ? trilerp(b00.x, b01.x, b10.x, b11.x, b00.y, b01.y, b10.y, b11.y, b.u, b.v, b.w) : 0.0f;
    if (!a.inb) fa = sample_border(P, a);
    if (hasB && !b.inb) fb = sample_border(P, b);
    vA = (float)((double)fa - P.iso);
    vB = (float)((double)fb - P.iso);
}

// hits_flat with the march two voxel steps per iteration: the DDA does not depend on the sampled values, so the sample
// after the next boundary is fetched together with the one at it (its gathers overlap); it is dropped when the first one
// turns out to be the crossing or the leaf's last.  Same operations per ray in the same order otherwise: bit-identical.
template <typename TR>
__device__ __forceinline__ bool hits_flat2(const IsoRenderParams& P, const Ray& ray, double& time, TR& tr)
{
    double T2;
    Lvl d1;
    {
        DDA d2;
        dda_init<12>(d2, ray);
        bool found = false;
        do {
            if (has_node2(P, d2.vx, d2.vy, d2.vz)) { found = true; break; }
        } while (dda_step(d2));
        if (!found) return false;
        T2 = dda_next(d2);
        lvl_init<7>(d1, ray, d2.t0);
    }
    Lvl d0, dv;
    d0.t0 = d0.nx = d0.ny = d0.nz = 0.0; d0.vx = d0.vy = d0.vz = 0;
    dv = d0;
    double t1_0 = 0.0, t1_v = 0.0, b1 = 0.0;
    float v0 = 0.0f;
    int st = FS_NODE1;
    while (st < FS_MISS) {
        if (st == FS_STEP1) st = lvl_step<7>(d1, ray, T2) ? FS_NODE1 : FS_MISS;
        if (st == FS_NODE1) {
            if (node1_flags(P, d1.vx, d1.vy, d1.vz) & 2) {
                t1_0 = lvl_next(d1, T2);
                lvl_init<3>(d0, ray, d1.t0);
                st = FS_LEAF;
            } else st = FS_STEP1;
        } else if (st == FS_LEAF) {
            const int lf = leaf_flags(P, d0.vx, d0.vy, d0.vz);
            const bool march = (lf & 2) != 0;
            if (lf) tr.leaf(march);
            if (march) {
                t1_v = lvl_next(d0, t1_0);
                lvl_init<0>(dv, ray, d0.t0);
                st = FS_ENTER;
            } else if (!lvl_step<3>(d0, ray, t1_0)) st = FS_STEP1;
        }
        if (st >= FS_ENTER && st < FS_MISS) {
            double tA, tB;
            const double tprev = dv.t0;
            bool hasB = true;
            if (st == FS_ENTER) { tA = dv.t0; tB = lvl_next(dv, t1_v); }
            else if (st == FS_MARCH) {
                tA = lvl_next(dv, t1_v);
                hasB = lvl_step<0>(dv, ray, t1_v);                           // taken back below if A is the crossing
                tB = lvl_next(dv, t1_v);
            } else { tA = 0.5 * (dv.t0 + b1); tB = tA; hasB = false; }
            float vA, vB;
            interp_pair(P, ray, tA, tB, hasB, vA, vB);
            bool leafDone = false, stepAfter = false;
            if (st >= FS_BISECT) {
                tr.sample(1);
                if (v0 * vA <= 0.0f) b1 = tA;
                else { dv.t0 = tA; v0 = vA; }
                if (++st == FS_BISECT + 5) { time = 0.5 * (dv.t0 + b1); st = FS_HIT; }
            } else {
                bool checkB = true;
                if (st == FS_ENTER) { v0 = vA; tr.sample(1); st = FS_MARCH; }
                else {
                    tr.sample(1);
                    if (v0 * vA <= 0.0f) { dv.t0 = tprev; b1 = tA; st = FS_BISECT; checkB = false; }
                    else { v0 = vA; if (!hasB) { leafDone = true; checkB = false; } }
                }
                if (checkB) {
                    tr.sample(1);
                    if (v0 * vB <= 0.0f) { b1 = tB; st = FS_BISECT; }
                    else { v0 = vB; stepAfter = true; }
                }
            }
            if (stepAfter && !lvl_step<0>(dv, ray, t1_v)) leafDone = true;
            if (leafDone) st = lvl_step<3>(d0, ray, t1_0) ? FS_LEAF : FS_STEP1;
        }
    }
    return st == FS_HIT;
}

__device__ __forceinline__ double len3(double x, double y, double z) { return sqrt(x * x + y * y + z * z); }

// Vec3::normalize(eps = 1e-7), TP/openvdb/math/Vec3.h:377-385
__device__ __forceinline__ void normalize3(double& x, double& y, double& z)
{
    const double d = len3(x, y, z);
    if (!(fabs(d - 0.0) > 1.0e-7)) return;
    const double r = 1.0 / d;
    x *= r; y *= r; z *= r;
}

// PerspectiveCamera::getRay + Ray::worldToIndex + Ray::clip
// (TP/openvdb/tools/RayTracer.h:444-448,507-516; TP/openvdb/math/Ray.h:177-185,260-293)
__device__ __forceinline__ bool make_ray(const IsoRenderParams& P, int i, int j, Ray& r, double& wdx, double& wdy, double& wdz)
{
    const IsoCamera& c = P.cam;
    const double ds0 = (2 * ((double)i + 0.5) / (double)P.W - 1) * c.sw;
    const double ds1 = (1 - 2 * ((double)j + 0.5) / (double)P.H) * c.sh;
    const double ds2 = -1.0;
    double dx = ds0 * c.J[0][0] + ds1 * c.J[1][0] + ds2 * c.J[2][0];
    double dy = ds0 * c.J[0][1] + ds1 * c.J[1][1] + ds2 * c.J[2][1];
    double dz = ds0 * c.J[0][2] + ds1 * c.J[1][2] + ds2 * c.J[2][2];
    normalize3(dx, dy, dz);
    wdx = dx; wdy = dy; wdz = dz;
    const double sc = 1.0 / (dx * c.d0[0] + dy * c.d0[1] + dz * c.d0[2]);
    const double wt0 = 1e-3 * sc, wt1 = DBL_MAX * sc;
    r.ex = (c.org[0] - P.t[0]) * P.sinv;
    r.ey = (c.org[1] - P.t[1]) * P.sinv;
    r.ez = (c.org[2] - P.t[2]) * P.sinv;
    const double ix = dx * P.sinv, iy = dy * P.sinv, iz = dz * P.sinv;
    const double len = len3(ix, iy, iz);
    r.dx = ix / len; r.dy = iy / len; r.dz = iz / len;
    r.ix = 1.0 / r.dx; r.iy = 1.0 / r.dy; r.iz = 1.0 / r.dz;
    double t0 = len * wt0, t1 = len * wt1;
#define ISO_SLAB(MIN, MAX, E, INV)                                 \
    {                                                              \
        double a = ((double)(MIN) - E) * INV;                      \
        double b = ((double)(MAX) - E) * INV;                      \
        if (a > b) { double s_ = a; a = b; b = s_; }               \
        if (a > t0) t0 = a;                                        \
        if (b < t1) t1 = b;                                        \
        if (t0 > t1) return false;                                 \
    }
    ISO_SLAB(P.bbmin[0], P.bbmax[0], r.ex, r.ix)
    ISO_SLAB(P.bbmin[1], P.bbmax[1], r.ey, r.iy)
    ISO_SLAB(P.bbmin[2], P.bbmax[2], r.ez, r.iz)
#undef ISO_SLAB
    r.t0 = t0; r.t1 = t1;
    return true;
}

// ---- ambient occlusion ---------------------------------------------------------------------
// render_kernel.cu:109-146 (mode 1: ray sampling) with the secondary rays cast by the same
// double-precision hierarchy as the primary ray; tables of GPURendererDirect.cpp:146-189.
template <int FLAT>
__device__ bool cast_world(const IsoRenderParams& P, double ox, double oy, double oz, double dx, double dy, double dz,
                           double& hx, double& hy, double& hz)
{
    Ray r;
    r.ex = (ox - P.t[0]) * P.sinv; r.ey = (oy - P.t[1]) * P.sinv; r.ez = (oz - P.t[2]) * P.sinv;
    const double ix = dx * P.sinv, iy = dy * P.sinv, iz = dz * P.sinv;
    const double len = len3(ix, iy, iz);
    r.dx = ix / len; r.dy = iy / len; r.dz = iz / len;
    r.ix = 1.0 / r.dx; r.iy = 1.0 / r.dy; r.iz = 1.0 / r.dz;
    double t0 = len * 1e-9, t1 = len * DBL_MAX;
#define ISO_SLAB(MIN, MAX, E, INV)                                 \
    {                                                              \
        double a = ((double)(MIN) - E) * INV;                      \
        double b = ((double)(MAX) - E) * INV;                      \
        if (a > b) { double s_ = a; a = b; b = s_; }               \
        if (a > t0) t0 = a;                                        \
        if (b < t1) t1 = b;                                        \
        if (t0 > t1) return false;                                 \
    }
    ISO_SLAB(P.bbmin[0], P.bbmax[0], r.ex, r.ix)
    ISO_SLAB(P.bbmin[1], P.bbmax[1], r.ey, r.iy)
    ISO_SLAB(P.bbmin[2], P.bbmax[2], r.ez, r.iz)
#undef ISO_SLAB
    r.t0 = t0; r.t1 = t1;
    double it;
    bool hit;
    if (FLAT > 0) { NoTrace nt; hit = hits_flat<1>(P, r, it, nt); }        // the secondary rays take the one-sample form (registers)
    else hit = hits_hierarchy(P, r, it);
    if (!hit) return false;
    double px, py, pz;
    ray_at(r, it, px, py, pz);
    hx = px * P.s + P.t[0]; hy = py * P.s + P.t[1]; hz = pz * P.s + P.t[2];
    return true;
}

// the hemisphere frame of a pixel: tangent and bitangent around the normal from the 4 x 4-tiled rotation table
struct AoFrame { double tx, ty, tz, bx, by, bz; };
__device__ __forceinline__ AoFrame ao_frame(const IsoRenderParams& P, double nx, double ny, double nz, int x, int y)
{
    const float* rot = P.aoRot + 4 * ((x % 4) + 4 * (y % 4));
    const double qx = rot[0], qy = rot[1], qz = rot[2];
    const double dn = qx * nx + qy * ny + qz * nz;
    AoFrame f;
    f.tx = qx - nx * dn; f.ty = qy - ny * dn; f.tz = qz - nz * dn;
    normalize3(f.tx, f.ty, f.tz);
    f.bx = ny * f.tz - nz * f.ty; f.by = nz * f.tx - nx * f.tz; f.bz = nx * f.ty - ny * f.tx;
    return f;
}
// world direction of hemisphere sample i
__device__ __forceinline__ void ao_direction(const IsoRenderParams& P, const AoFrame& f, double nx, double ny, double nz, int i,
                                             double& wx, double& wy, double& wz)
{
    double sx = P.aoHemi[4 * i], sy = P.aoHemi[4 * i + 1], sz = P.aoHemi[4 * i + 2];
    normalize3(sx, sy, sz);
    wx = f.tx * sx + f.bx * sy + nx * sz;
    wy = f.ty * sx + f.by * sy + ny * sz;
    wz = f.tz * sx + f.bz * sy + nz * sz;
    normalize3(wx, wy, wz);
}
// contribution of a sample whose ray hit at distance dist: smoothstep(1, 0, r / d)
__device__ __forceinline__ double ao_value(const IsoRenderParams& P, double dist)
{
    double yv = 1.0 - P.aoRadius / dist;
    yv = yv < 0.0 ? 0.0 : (yv > 1.0 ? 1.0 : yv);
    return yv * yv * (3.0 - (2.0 * yv));
}

template <int FLAT>
__device__ double ambient_occlusion(const IsoRenderParams& P, double px, double py, double pz,
                                    double nx, double ny, double nz, int x, int y)
{
    const AoFrame f = ao_frame(P, nx, ny, nz, x, y);
    double ao = 0.0;
    const int n = P.aoSamples > 512 ? 512 : P.aoSamples;
    for (int i = 0; i < n; ++i) {
        double wx, wy, wz;
        ao_direction(P, f, nx, ny, nz, i, wx, wy, wz);
        double hx, hy, hz;
        double value = 1.0;
        if (cast_world<FLAT>(P, px, py, pz, wx, wy, wz, hx, hy, hz)) value = ao_value(P, len3(px - hx, py - hy, pz - hz));
        ao += value;
    }
    return ao / n;
}

// Everything after the hit time is known: position, normal, Phong, depth, flow
// (IsoVolumeRayTracer.h:274-292,300-307,519-548; PhongShader.h:27-38; CPURenderer.cpp:726-737)
template <bool AO, int FLAT = 0>
__device__ __forceinline__ void shade_hit(const IsoRenderParams& P, const Ray& r, double it,
                                          double wdx, double wdy, double wdz, int px_, int py_, float o[12])
{
    double ipx, ipy, ipz;
    ray_at(r, it, ipx, ipy, ipz);
    const double wx = ipx * P.s + P.t[0], wy = ipy * P.s + P.t[1], wz = ipz * P.s + P.t[2];
    double nx = (double)interp_global(P, ipx + 1.0, ipy + 0.0, ipz + 0.0);
    nx -= (double)interp_global(P, ipx - 1.0, ipy - 0.0, ipz - 0.0);
    double ny = (double)interp_global(P, ipx + 0.0, ipy + 1.0, ipz + 0.0);
    ny -= (double)interp_global(P, ipx - 0.0, ipy - 1.0, ipz - 0.0);
    double nz = (double)interp_global(P, ipx + 0.0, ipy + 0.0, ipz + 1.0);
    nz -= (double)interp_global(P, ipx - 0.0, ipy - 0.0, ipz - 1.0);
    normalize3(nx, ny, nz);
    const double wtime = it * len3(r.dx * P.s, r.dy * P.s, r.dz * P.s);

    const double ndl = nx * P.light[0] + ny * P.light[1] + nz * P.light[2];
    const double andl = fabs(ndl);
    double c0 = P.ambient[0], c1 = P.ambient[1], c2 = P.ambient[2];
    c0 += P.diffuse[0] * andl; c1 += P.diffuse[1] * andl; c2 += P.diffuse[2] * andl;
    const double two = 2 * ndl;
    const double rx = two * nx - P.light[0], ry = two * ny - P.light[1], rz = two * nz - P.light[2];
    const double rd = rx * wdx + ry * wdy + rz * wdz;
    double x = rd > 0.0 ? rd : 0.0;
    double pw = 1.0;
    int e = P.exponent;
    if (e < 0) { e = -e; x = 1.0 / x; }
    while (e--) pw *= x;
    c0 += (P.specular[0] * P.spec_c1) * pw;
    c1 += (P.specular[1] * P.spec_c1) * pw;
    c2 += (P.specular[2] * P.spec_c1) * pw;
    o[0] = (float)c0; o[1] = (float)c1; o[2] = (float)c2;
    o[3] = (wtime == 0) ? 0.0f : 1.0f;
    if (wtime > 0) {
        const double (*V)[4] = P.cam.V;
        double n0 = nx * V[0][0] + ny * V[1][0] + nz * V[2][0];
        double n1 = nx * V[0][1] + ny * V[1][1] + nz * V[2][1];
        double n2 = nx * V[0][2] + ny * V[1][2] + nz * V[2][2];
        if (n2 < 0) { n0 = -n0; n1 = -n1; n2 = -n2; }
        o[4] = (float)n0; o[5] = (float)n1; o[6] = (float)n2; o[7] = (float)wtime;
        const double (*L)[4] = P.Vlast;
        const double cx = wx * V[0][0] + wy * V[1][0] + wz * V[2][0] + 1.0 * V[3][0];
        const double cy = wx * V[0][1] + wy * V[1][1] + wz * V[2][1] + 1.0 * V[3][1];
        const double cw = wx * V[0][3] + wy * V[1][3] + wz * V[2][3] + 1.0 * V[3][3];
        const double lx = wx * L[0][0] + wy * L[1][0] + wz * L[2][0] + 1.0 * L[3][0];
        const double ly = wx * L[0][1] + wy * L[1][1] + wz * L[2][1] + 1.0 * L[3][1];
        const double lw = wx * L[0][3] + wy * L[1][3] + wz * L[2][3] + 1.0 * L[3][3];
        o[8] = -(float)(lx / lw - cx / cw);
        o[9] = -(float)(ly / lw - cy / cw);
        if ((AO && P.aoSamples > 0) || P.hitState) {
            // hemisphere around the normal that faces the viewer, origin pulled back by aoBias = 1e-3
            double ax = nx, ay = ny, az = nz;
            if (nx * wdx + ny * wdy + nz * wdz > 0) { ax = -ax; ay = -ay; az = -az; }
            const double ox = wx - 1e-3 * wdx, oy = wy - 1e-3 * wdy, oz = wz - 1e-3 * wdz;
            if (P.hitState) {            // tiled AO: the rays are cast later, by every tile (iso_ao_dist_kernel)
                double* hs = P.hitState + ((size_t)py_ * P.W + px_) * 6;
                hs[0] = ox; hs[1] = oy; hs[2] = oz; hs[3] = ax; hs[4] = ay; hs[5] = az;
            }
            if (AO && P.aoSamples > 0) o[10] = (float)ambient_occlusion<FLAT>(P, ox, oy, oz, ax, ay, az, px_, py_);
        }
    }
}

__device__ __forceinline__ void store_pixel(const IsoRenderParams& P, int i, int j, const float o[12])
{
    float4* dst = reinterpret_cast<float4*>(P.out + ((size_t)j * P.W + i) * 12);
    dst[0] = make_float4(o[0], o[1], o[2], o[3]);
    dst[1] = make_float4(o[4], o[5], o[6], o[7]);
    dst[2] = make_float4(o[8], o[9], o[10], o[11]);
}

// XCD-aware tile order: blocks b and b+8 share an XCD/L2, so give each XCD a contiguous run of
// tiles (neighbouring pixel tiles walk the same bricks).  Bijective for any tile count.
__device__ __forceinline__ int xcd_remap(int bid, int nwg)
{
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// ---- variant 0: per-lane gather ------------------------------------------------------------
// FLAT: 0 = the nested loops of hits_hierarchy, 1 = hits_flat (one sample per iteration), 3 = hits_flat2 (two)
template <bool AO, int FLAT = 3>
__device__ __forceinline__ void render_gather_tile(const IsoRenderParams& P, int vb, int tiles_x, int ntiles, int lane, bool remap = true)
{
    const bool ordered = remap && P.tileOrder != nullptr;
    const long long c0 = P.tileCost ? (long long)__builtin_amdgcn_s_memtime() : 0;
    const int tile = ordered ? (int)P.tileOrder[vb] : (remap ? xcd_remap(vb, ntiles) : vb);
    const int i = (tile % tiles_x) * 8 + (lane & 7);
    const int j = (tile / tiles_x) * 8 + (lane >> 3);
    if (i >= P.W || j >= P.H) return;
    float o[12] = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 1.0f, 0.f };
    const bool inside = i >= P.vp[0] && j >= P.vp[1] && i < P.vp[2] && j < P.vp[3];
    if (inside) {
        o[8] = -0.0f; o[9] = -0.0f;
        Ray r;
        double wdx, wdy, wdz, it;
        bool hit = make_ray(P, i, j, r, wdx, wdy, wdz);
        if (hit) {
            if (FLAT == 3) { NoTrace nt; hit = hits_flat2(P, r, it, nt); }
            else if (FLAT > 0) { NoTrace nt; hit = hits_flat<1>(P, r, it, nt); }
            else hit = hits_hierarchy(P, r, it);
        }
        if (hit) shade_hit<AO, FLAT>(P, r, it, wdx, wdy, wdz, i, j, o);
    }
    store_pixel(P, i, j, o);
    // the wave's cost for the next frame's dispatch order: its lanes leave the traversal together, any lane may write
    if (P.tileCost && remap && lane == 0)
        P.tileCost[tile] = (unsigned)((long long)__builtin_amdgcn_s_memtime() - c0);
}

// ---- ray-cast AO of a tiled volume, exactly ------------------------------------------------------------------------
// One wave per 8 x 8 pixels; a lane casts the aoSamples rays of its pixel (origin and normal from the composite of the tiles'
// hit-state exports) against THIS tile's leaves and writes each ray's hit distance (+inf: none here).  Same frame, directions,
// traversal and distance arithmetic as ambient_occlusion(); the first hit of a ray in the unsplit volume lies in a leaf that
// exactly one tile owns, which computes it bit for bit, every other tile a later one: min over the tiles = the unsplit distance.
__global__ __launch_bounds__(64) void iso_ao_dist_kernel(const IsoRenderParams P, const double* __restrict__ hitState,
                                                          const float* __restrict__ gbuf, double* __restrict__ dist)
{
    const int tiles_x = (P.W + 7) >> 3, ntiles = tiles_x * ((P.H + 7) >> 3);
    const int tile = xcd_remap(blockIdx.x, ntiles), lane = threadIdx.x;
    const int i = (tile % tiles_x) * 8 + (lane & 7), j = (tile / tiles_x) * 8 + (lane >> 3);
    if (i >= P.W || j >= P.H) return;
    const size_t pix = (size_t)j * P.W + i;
    if (gbuf[pix * 12 + 3] != 1.0f) return;
    const double* hs = hitState + pix * 6;
    const double px = hs[0], py = hs[1], pz = hs[2], nx = hs[3], ny = hs[4], nz = hs[5];
    const AoFrame f = ao_frame(P, nx, ny, nz, i, j);
    const int n = P.aoSamples > 512 ? 512 : P.aoSamples;
    for (int s = 0; s < n; ++s) {
        double wx, wy, wz, hx, hy, hz;
        ao_direction(P, f, nx, ny, nz, s, wx, wy, wz);
        double d = __builtin_huge_val();
        if (cast_world<3>(P, px, py, pz, wx, wy, wz, hx, hy, hz)) d = len3(px - hx, py - hy, pz - hz);
        dist[pix * n + s] = d;
    }
}

__global__ __launch_bounds__(256) void iso_ao_finish_kernel(const IsoRenderParams P, const double* __restrict__ dist, float* __restrict__ gbuf)
{
    const size_t pix = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (pix >= (size_t)P.W * P.H || gbuf[pix * 12 + 3] != 1.0f) return;
    const int n = P.aoSamples > 512 ? 512 : P.aoSamples;
    double ao = 0.0;
    for (int s = 0; s < n; ++s) {
        const double d = dist[pix * n + s];
        ao += d == __builtin_huge_val() ? 1.0 : ao_value(P, d);
    }
    gbuf[pix * 12 + 10] = (float)(ao / n);
}

// ---- cost-ordered dispatch: one workgroup sorts the (cost, tile) pairs of the previous frame (bitonic, in LDS) -------------
__global__ __launch_bounds__(1024) void iso_tile_order_kernel(const unsigned* __restrict__ cost, unsigned short* __restrict__ order,
                                                               int n, int mode, int slots)
{
    __shared__ unsigned long long key[ISO_ORDER_MAX_TILES];
    const int tid = threadIdx.x;
    // key = (cost + 1) << 16 | tile; the padding (key 0) sorts behind every real tile
    for (int k = tid; k < ISO_ORDER_MAX_TILES; k += 1024)
        key[k] = k < n ? ((((unsigned long long)cost[k] + 1ull) << 16) | (unsigned)k) : 0ull;
    __syncthreads();
    for (int size = 2; size <= ISO_ORDER_MAX_TILES; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int k = tid; k < ISO_ORDER_MAX_TILES / 2; k += 1024) {
                const int lo = 2 * k - (k & (stride - 1)), hi = lo + stride;
                const bool desc = (lo & size) == 0;                          // overall descending
                const unsigned long long a = key[lo], b = key[hi];
                if ((a < b) == desc) { key[lo] = b; key[hi] = a; }
            }
            __syncthreads();
        }
    // key[0 .. n) now holds the real tiles, heaviest first
    for (int k = tid; k < n; k += 1024) {
        int src = k;
        if (mode == 2 && k >= slots) src = n - 1 - (k - slots);             // second waves: lightest first
        if (mode == 2 && k < slots) src = k;
        order[k] = (unsigned short)(key[src] & 0xffffu);
    }
}

template <bool AO, int FLAT>
__global__ __launch_bounds__(64) void iso_render_gather(const IsoRenderParams P)
{
    const int tiles_x = (P.W + 7) >> 3, ntiles = tiles_x * ((P.H + 7) >> 3);
    render_gather_tile<AO, FLAT>(P, blockIdx.x, tiles_x, ntiles, threadIdx.x);
}

// The same kernel with the per-frame part of the parameter block (camera, previous camera, light) read from DEVICE memory through
// the constant address space: the loads are scalar (s_load), the values live where the kernel arguments would -- and the launch itself
// no longer changes from frame to frame, so a captured HIP graph of the frame replays it (pipeline.py, frame graph).
typedef const double __attribute__((address_space(4))) iso_cdouble_t;
__global__ __launch_bounds__(64) void iso_render_gather_block(const IsoRenderParams P0, const IsoFrameBlock* fb)
{
    IsoRenderParams P = P0;
    const iso_cdouble_t* src = (const iso_cdouble_t*)(const void*)fb;
    constexpr int NC = sizeof(IsoCamera) / 8;
    double* cam = reinterpret_cast<double*>(&P.cam);
#pragma unroll
    for (int i = 0; i < NC; ++i) cam[i] = src[i];
#pragma unroll
    for (int i = 0; i < 16; ++i) (&P.Vlast[0][0])[i] = src[NC + i];
#pragma unroll
    for (int i = 0; i < 3; ++i) P.light[i] = src[NC + 16 + i];
    const int tiles_x = (P.W + 7) >> 3, ntiles = tiles_x * ((P.H + 7) >> 3);
    render_gather_tile<false, 3>(P, blockIdx.x, tiles_x, ntiles, threadIdx.x);
}

__global__ __launch_bounds__(64) void iso_write_block_kernel(const IsoFrameBlock b, IsoFrameBlock* dst)
{
    const unsigned* s = reinterpret_cast<const unsigned*>(&b);
    unsigned* d = reinterpret_cast<unsigned*>(dst);
    for (int i = threadIdx.x; i < (int)(sizeof(IsoFrameBlock) / 4); i += 64) d[i] = s[i];
}

// ---- diagnostics: variant 0 with per-tile clocks and per-ray step counts (never on the product path) ------------------
// out[tile] = { cycles of the wave (s_memtime), samples of its busiest ray, leaves marched / skipped by that ray,
//               sum of samples over its rays, rays that hit }
template <int FLAT>
__global__ __launch_bounds__(64) void iso_render_stats(const IsoRenderParams P, long long* __restrict__ out)
{
    const long long c0 = (long long)__builtin_amdgcn_s_memtime();
    const int tiles_x = (P.W + 7) >> 3, ntiles = tiles_x * ((P.H + 7) >> 3);
    const int tile = xcd_remap(blockIdx.x, ntiles), lane = threadIdx.x;
    const int i = (tile % tiles_x) * 8 + (lane & 7), j = (tile / tiles_x) * 8 + (lane >> 3);
    Trace tr;
    int hit = 0;
    if (i < P.W && j < P.H && i >= P.vp[0] && j >= P.vp[1] && i < P.vp[2] && j < P.vp[3]) {
        Ray r;
        double wdx, wdy, wdz, it;
        float o[12] = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 1.0f, 0.f };
        bool h = make_ray(P, i, j, r, wdx, wdy, wdz);
        if (h) h = FLAT == 3 ? hits_flat2(P, r, it, tr) : (FLAT > 0 ? hits_flat<1>(P, r, it, tr) : hits_hierarchy(P, r, it, tr));
        if (h) {
            shade_hit<false>(P, r, it, wdx, wdy, wdz, i, j, o);
            hit = 1;
        }
        store_pixel(P, i, j, o);
    }
    int best = tr.samples, lv = tr.leaves, sk = tr.skipped, sum = tr.samples, hits = hit;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int ob = __shfl_xor(best, off), ol = __shfl_xor(lv, off), os = __shfl_xor(sk, off);
        if (ob > best) { best = ob; lv = ol; sk = os; }
        sum += __shfl_xor(sum, off);
        hits += __shfl_xor(hits, off);
    }
    const long long c1 = (long long)__builtin_amdgcn_s_memtime();
    if (lane == 0) {
        long long* q = out + (size_t)tile * 6;
        q[0] = c1 - c0; q[1] = best; q[2] = lv; q[3] = sk; q[4] = sum; q[5] = hits;
    }
}

// ---- variant 3: the slot table of the volume in LDS ------------------------------------------------------------
// Every sample is a dependent pair of gathers: slot[brick position] -> 8 corners inside that brick.  For volumes of up
// to 32 768 brick positions (256^3) the whole slot table is 128 KB: one workgroup of eight waves (= eight 8x8 pixel
// tiles) per CU copies it into LDS once, and the first gather of every sample never leaves the CU.  The traversal code
// is the same (it sees a parameter block whose `slot` pointer addresses LDS), so the results are bit-identical.
template <bool AO>
__global__ __launch_bounds__(512) void iso_render_gather_ldsslot(const IsoRenderParams P)
{
    extern __shared__ __attribute__((aligned(16))) int32_t slotLds[];
    const int nb = P.nbx * P.nby * P.nbz;
    {
        const int4* src = reinterpret_cast<const int4*>(P.slot);
        int4* dst = reinterpret_cast<int4*>(slotLds);
        const int quads = nb >> 2;
        for (int q = threadIdx.x; q < quads; q += 512) dst[q] = src[q];
        for (int k = (quads << 2) + threadIdx.x; k < nb; k += 512) slotLds[k] = P.slot[k];
    }
    __syncthreads();
    IsoRenderParams Q = P;
    Q.slot = slotLds;
    const int tiles_x = (P.W + 7) >> 3, ntiles = tiles_x * ((P.H + 7) >> 3);
    // the eight tiles of a workgroup are neighbours in the image, and an XCD gets a contiguous run of workgroups
    const int vb = xcd_remap(blockIdx.x, gridDim.x) * 8 + (threadIdx.x >> 6);
    if (vb < ntiles) render_gather_tile<AO>(Q, vb, tiles_x, ntiles, threadIdx.x & 63, false);
}

// ---- variant 2: the same code in a 128-register budget ----------------------------------------
// For rendering frame t+1 on a side stream under the SR network of frame t: the fused conv holds
// 226 VGPR + 128 AGPR = 360 of a SIMD's 512 registers, so a 168-register ray-march wave can never
// sit beside it and the two kernels only time-slice CUs.  At <= 128 registers (a few cold values
// spilled to scratch) one ray-march wave fits beside each conv wave and fills its stall slots --
// provided there is never a second one on the same SIMD (2 x 128 + 360 > 512 would keep the next
// conv workgroup off that CU), hence the wave cap: 4 x #CUs waves, striding over the tiles.
// Same instructions on the same values: results are bit-identical to variant 0.
template <bool AO, int FLAT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void iso_render_gather_slim(const IsoRenderParams P)
{
    // workgroups of FOUR independent waves: the hardware spreads the waves of a workgroup over the four SIMDs of its
    // CU, so 'one workgroup per CU' is 'one wave per SIMD', and 256 workgroups dispatch 4x faster than 1024
    const int lane = threadIdx.x & 63;
    // The grid is capped and the waves PULL tiles: tile costs differ by several x (background vs. deep volume),
    // and with a fixed stride the frame waits for the unluckiest wave.  One queue per XCD (its contiguous share
    // of the image, as in xcd_remap, so neighbouring tiles keep sharing an L2); a wave whose own queue is empty
    // steals from the next XCD's.  Every wave ends after 8 failed fetches: the grid always drains.
    if (threadIdx.x == 0) atomicAdd(P.resident, 4u);      // "this workgroup's four waves have their slots" -- see
                                                          // iso_gate_kernel; one atomic per workgroup: 1024 on one address serialise
    const int tiles_x = (P.W + 7) >> 3, ntiles = tiles_x * ((P.H + 7) >> 3);
    const int q = ntiles >> 3, r = ntiles & 7, xcd = blockIdx.x & 7;
    for (int s = 0; s < 8; ++s) {
        const int k = (xcd + s) & 7;
        const unsigned cnt = (unsigned)(q + (k < r ? 1 : 0));
        for (;;) {
            unsigned i = 0;
            if (lane == 0) i = atomicAdd(&P.tileQueue[k], 1u);
            i = (unsigned)__builtin_amdgcn_readfirstlane((int)i);
            if (i >= cnt) break;
            render_gather_tile<AO, FLAT>(P, (int)i * 8 + k, tiles_x, ntiles, lane);
        }
    }
}

// WHERE the capped ray-march waves land decides what the overlap costs: dispatched into an idle GPU they spread
// one per SIMD; dispatched while another kernel is ramping up they pile onto the CUs that happen to be free, and
// every CU with two of them on a SIMD is lost to the conv workgroups for the whole render (measured: the 480x270
// conv layers 84 -> 130-200 us).  The frame pipeline therefore puts this one-wave kernel on the MAIN stream right
// after it has enqueued the render on the side stream: it holds the main stream back until all ray-march waves have
// reported in (or a timeout passes -- it must never wait for a render that is not coming).
__global__ __launch_bounds__(64) void iso_gate_kernel(const unsigned* resident, unsigned target, long long timeoutTicks)
{
    if (threadIdx.x != 0) return;
    const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();      // 100 MHz
    for (;;) {
        const unsigned seen = __atomic_load_n(resident, __ATOMIC_RELAXED);
        if ((int)(seen - target) >= 0) break;
        if ((long long)__builtin_amdgcn_s_memrealtime() - t0 > timeoutTicks) break;
        __builtin_amdgcn_s_sleep(8);
    }
}

// ---- variant 1: wave-cooperative LDS brick cache --------------------------------------------
// The hierarchy walk of hits_hierarchy() as a resumable per-lane state machine, so that the wave
// can meet between leaves: every lane walks its node/leaf DDAs to its next occupied leaf, the wave
// votes (ballot) for the leaf of its first waiting lane, stages that 9^3 apron brick into LDS with
// coalesced 16-byte loads, and the lanes inside that leaf run the voxel DDA on LDS; samples whose
// cell falls outside the staged brick (entry / exit faces) fall back to the global gather, so the
// arithmetic -- and therefore the hit mask -- is identical to variant 0 and to the CPU restatement.
struct Walk {
    DDA d2, d1, d0;
    int lvl;          // 2, 1, 0: level whose current cell has not been examined yet
};

// advance to the next occupied leaf; returns false when the ray has left the volume.
// On success ray.t0/t1 hold the leaf's span (IsoVolumeRayTracer.h:42) and (lx,ly,lz) its brick index.
__device__ __forceinline__ bool walk_next_leaf(const IsoRenderParams& P, Walk& w, Ray& ray, int& brick)
{
    for (;;) {
        if (w.lvl == 2) {
            if (has_node2(P, w.d2.vx, w.d2.vy, w.d2.vz)) {
                ray.t0 = w.d2.t0; ray.t1 = dda_next(w.d2);
                dda_init<7>(w.d1, ray);
                w.lvl = 1;
            } else if (!dda_step(w.d2)) return false;
        } else if (w.lvl == 1) {
            if (node1_flags(P, w.d1.vx, w.d1.vy, w.d1.vz) & 2) {
                ray.t0 = w.d1.t0; ray.t1 = dda_next(w.d1);
                dda_init<3>(w.d0, ray);
                w.lvl = 0;
            } else if (!dda_step(w.d1)) {
                if (!dda_step(w.d2)) return false;
                w.lvl = 2;
            }
        } else {
            if (leaf_flags(P, w.d0.vx, w.d0.vy, w.d0.vz) & 2) {
                ray.t0 = w.d0.t0; ray.t1 = dda_next(w.d0);
                brick = (((w.d0.vz - P.org[2]) >> 3) * P.nby + ((w.d0.vy - P.org[1]) >> 3)) * P.nbx + ((w.d0.vx - P.org[0]) >> 3);
                return true;
            }
            if (!dda_step(w.d0)) {
                if (!dda_step(w.d1)) {
                    if (!dda_step(w.d2)) return false;
                    w.lvl = 2;
                } else w.lvl = 1;
            }
        }
    }
}

// the leaf just visited produced no hit: `while (mDDA.step())` of each enclosing level
__device__ __forceinline__ bool walk_leave_leaf(Walk& w)
{
    if (dda_step(w.d0)) { w.lvl = 0; return true; }
    if (dda_step(w.d1)) { w.lvl = 1; return true; }
    if (dda_step(w.d2)) { w.lvl = 2; return true; }
    return false;
}

__device__ __forceinline__ float interp_cached(const IsoRenderParams& P, const float* __restrict__ lds, int cachedBrick,
                                               double px, double py, double pz)
{
    const int cx = (int)floor(px), cy = (int)floor(py), cz = (int)floor(pz);
    const int lx = cx - P.org[0], ly = cy - P.org[1], lz = cz - P.org[2];
    if ((unsigned)lx < (unsigned)P.nx && (unsigned)ly < (unsigned)P.ny && (unsigned)lz < (unsigned)P.nz &&
        ((lz >> 3) * P.nby + (ly >> 3)) * P.nbx + (lx >> 3) == cachedBrick) {
        const float u = (float)px - (float)cx;
        const float v = (float)py - (float)cy;
        const float w = (float)pz - (float)cz;
        return interp_from_brick(lds, cx & 7, cy & 7, cz & 7, u, v, w);
    }
    return interp_global(P, px, py, pz);
}

__device__ __forceinline__ bool hits_voxel_cached(const IsoRenderParams& P, const float* lds, int cachedBrick,
                                                  const Ray& ray, double& time)
{
    auto value = [&](double t) -> float {
        double px, py, pz;
        ray_at(ray, t, px, py, pz);
        return (float)((double)interp_cached(P, lds, cachedBrick, px, py, pz) - P.iso);
    };
    DDA d;
    dda_init<0>(d, ray);
    double t0 = d.t0;
    float v0 = value(t0);
    do {
        double t1 = dda_next(d);
        float v1 = value(t1);
        if (v0 * v1 <= 0.0f) {
            double t = 0.5 * (t0 + t1);
            for (int i = 0; i < 5; ++i) {
                float v2 = value(t);
                if (v0 * v2 <= 0.0f) t1 = t;
                else { t0 = t; v0 = v2; }
                t = 0.5 * (t0 + t1);
            }
            time = t;
            return true;
        }
        t0 = t1;
        v0 = v1;
    } while (dda_step(d));
    return false;
}

template <bool AO>
__global__ __launch_bounds__(64) void iso_render_lds(const IsoRenderParams P)
{
    __shared__ __attribute__((aligned(16))) float brickLds[ISO_BRICK_STRIDE];
    const int tiles_x = (P.W + 7) >> 3, tiles_y = (P.H + 7) >> 3;
    const int tile = xcd_remap(blockIdx.x, tiles_x * tiles_y);
    const int lane = threadIdx.x;
    const int i = (tile % tiles_x) * 8 + (lane & 7);
    const int j = (tile / tiles_x) * 8 + (lane >> 3);
    const bool in_image = i < P.W && j < P.H;
    float o[12] = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 1.0f, 0.f };
    const bool inside = in_image && i >= P.vp[0] && j >= P.vp[1] && i < P.vp[2] && j < P.vp[3];
    Ray r;
    Walk w;
    double wdx = 0, wdy = 0, wdz = 0, it = 0;
    int brick = -1;
    bool hit = false;
    // state: 0 = walking to the next leaf, 1 = waiting at a leaf, 2 = finished
    int state = 2;
    if (inside) {
        o[8] = -0.0f; o[9] = -0.0f;
        if (make_ray(P, i, j, r, wdx, wdy, wdz)) {
            dda_init<12>(w.d2, r);
            w.lvl = 2;
            state = 0;
        }
    }
    for (;;) {
        if (state == 0) state = walk_next_leaf(P, w, r, brick) ? 1 : 2;
        const unsigned long long waiting = __ballot(state == 1);
        if (!waiting) break;                                   // ballot-based early out of the tile
        const int leader = __ffsll((long long)waiting) - 1;
        const int chosen = __shfl(brick, leader);
        const int slot = P.slot[chosen];                       // wave-uniform; a leaf always has a slot
        const float4* src = reinterpret_cast<const float4*>(P.bricks + (size_t)slot * ISO_BRICK_STRIDE);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int q = lane + 64 * k;
            if (q < ISO_BRICK_STRIDE / 4) reinterpret_cast<float4*>(brickLds)[q] = src[q];
        }
        __syncthreads();
        if (state == 1 && brick == chosen) {
            if (hits_voxel_cached(P, brickLds, chosen, r, it)) { hit = true; state = 2; }
            else state = walk_leave_leaf(w) ? 0 : 2;
        }
        __syncthreads();                                       // everyone done reading before the next stage
    }
    if (hit) shade_hit<AO>(P, r, it, wdx, wdy, wdz, i, j, o);
    if (in_image) store_pixel(P, i, j, o);
}

// ---- brick builder -------------------------------------------------------------------------
// One 64-lane workgroup per 8^3 brick position.  flag9: any non-zero among the 9^3 apron values
// (brick must be stored); leaf: any non-zero among the 8^3 own voxels (OpenVDB leaf exists);
// bbox6 / maxbits: active-voxel bbox and maximum (grid->evalMinMax, CPURenderer.cpp:501-502).
__device__ __forceinline__ unsigned int float_order_bits(float f)
{
    unsigned int u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(64) void iso_brick_flags(const float* __restrict__ dense, int nx, int ny, int nz,
                                                     int nbx, int nby, int nbz,
                                                     uint8_t* flag9, uint8_t* leaf, int* bbox6, unsigned int* maxbits)
{
    const int b = blockIdx.x;
    const int bx = b % nbx, by = (b / nbx) % nby, bz = b / (nbx * nby);
    const int lane = threadIdx.x;
    bool any9 = false, any8 = false;
    int mnx = INT32_MAX, mny = INT32_MAX, mnz = INT32_MAX, mxx = INT32_MIN, mxy = INT32_MIN, mxz = INT32_MIN;
    unsigned int mb = 0;
    for (int k = lane; k < ISO_BRICK_VALUES; k += 64) {
        const int lx = k % 9, ly = (k / 9) % 9, lz = k / 81;
        const int x = bx * 8 + lx, y = by * 8 + ly, z = bz * 8 + lz;
        float f = 0.0f;
        if (x < nx && y < ny && z < nz) f = dense[((size_t)z * ny + y) * nx + x];
        if (f != 0.0f) {
            any9 = true;
            if (lx < 8 && ly < 8 && lz < 8) {
                any8 = true;
                mnx = min(mnx, x); mny = min(mny, y); mnz = min(mnz, z);
                mxx = max(mxx, x); mxy = max(mxy, y); mxz = max(mxz, z);
                mb = max(mb, float_order_bits(f));
            }
        }
    }
    const unsigned long long m9 = __ballot(any9), m8 = __ballot(any8);
    if (m8) {
        for (int off = 32; off > 0; off >>= 1) {
            mnx = min(mnx, __shfl_xor(mnx, off)); mny = min(mny, __shfl_xor(mny, off)); mnz = min(mnz, __shfl_xor(mnz, off));
            mxx = max(mxx, __shfl_xor(mxx, off)); mxy = max(mxy, __shfl_xor(mxy, off)); mxz = max(mxz, __shfl_xor(mxz, off));
            mb = max(mb, (unsigned int)__shfl_xor((int)mb, off));
        }
    }
    if (lane == 0) {
        flag9[b] = m9 ? 1 : 0;
        leaf[b] = m8 ? 1 : 0;
        if (m8) {
            atomicMin(&bbox6[0], mnx); atomicMin(&bbox6[1], mny); atomicMin(&bbox6[2], mnz);
            atomicMax(&bbox6[3], mxx); atomicMax(&bbox6[4], mxy); atomicMax(&bbox6[5], mxz);
            atomicMax(maxbits, mb);
        }
    }
}

__global__ __launch_bounds__(64) void iso_brick_fill(const float* __restrict__ dense, int nx, int ny, int nz,
                                                    int nbx, int nby, int nbz,
                                                    const int32_t* __restrict__ slot, float* __restrict__ bricks)
{
    const int b = blockIdx.x;
    const int s = slot[b];
    if (s < 0) return;
    const int bx = b % nbx, by = (b / nbx) % nby, bz = b / (nbx * nby);
    float* dst = bricks + (size_t)s * ISO_BRICK_STRIDE;
    for (int k = threadIdx.x; k < ISO_BRICK_STRIDE; k += 64) {
        float f = 0.0f;
        if (k < ISO_BRICK_VALUES) {
            const int lx = k % 9, ly = (k / 9) % 9, lz = k / 81;
            const int x = bx * 8 + lx, y = by * 8 + ly, z = bz * 8 + lz;
            if (x < nx && y < ny && z < nz) f = dense[((size_t)z * ny + y) * nx + x];
        }
        dst[k] = f;
    }
}

// range[b] = (min, max) over the voxels [8b-1, 8b+9]^3 of brick position b; outside the grid counts as 0
__global__ __launch_bounds__(64) void iso_leaf_range(const float* __restrict__ dense, int nx, int ny, int nz,
                                                    int nbx, int nby, int nbz, float* __restrict__ range)
{
    const int b = blockIdx.x;
    const int bx = b % nbx, by = (b / nbx) % nby, bz = b / (nbx * nby);
    float lo = 3.0e38f, hi = -3.0e38f;
    for (int k = threadIdx.x; k < 11 * 11 * 11; k += 64) {
        const int lx = k % 11, ly = (k / 11) % 11, lz = k / 121;
        const int x = bx * 8 - 1 + lx, y = by * 8 - 1 + ly, z = bz * 8 - 1 + lz;
        float f = 0.0f;
        if ((unsigned)x < (unsigned)nx && (unsigned)y < (unsigned)ny && (unsigned)z < (unsigned)nz) f = dense[((size_t)z * ny + y) * nx + x];
        lo = fminf(lo, f); hi = fmaxf(hi, f);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, off));
        hi = fmaxf(hi, __shfl_xor(hi, off));
    }
    if (threadIdx.x == 0) { range[2 * (size_t)b] = lo; range[2 * (size_t)b + 1] = hi; }
}

// nodeRange[n] = (min of the leaf minima, max of the leaf maxima) over the existing leaves of 128^3 node n
__global__ __launch_bounds__(64) void iso_node_range(const uint8_t* __restrict__ leaf, const float* __restrict__ leafRange,
                                                    int nbx, int nby, int nbz, int ox, int oy, int oz, int n1x, int n1y, int n1ox, int n1oy, int n1oz,
                                                    float* __restrict__ nodeRange)
{
    const int n = blockIdx.x;
    const int ax = n % n1x, ay = (n / n1x) % n1y, az = n / (n1x * n1y);
    float lo = 3.0e38f, hi = -3.0e38f;
    for (int k = threadIdx.x; k < 4096; k += 64) {
        // global brick coordinates of the node's k-th leaf position, then local to the stored region
        const int bx = ((ax + n1ox) << 4) + (k & 15) - (ox >> 3), by = ((ay + n1oy) << 4) + ((k >> 4) & 15) - (oy >> 3),
                  bz = ((az + n1oz) << 4) + (k >> 8) - (oz >> 3);
        if ((unsigned)bx >= (unsigned)nbx || (unsigned)by >= (unsigned)nby || (unsigned)bz >= (unsigned)nbz) continue;
        const size_t b = ((size_t)bz * nby + by) * nbx + bx;
        if (!leaf[b]) continue;
        lo = fminf(lo, leafRange[2 * b]); hi = fmaxf(hi, leafRange[2 * b + 1]);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, off));
        hi = fmaxf(hi, __shfl_xor(hi, off));
    }
    if (threadIdx.x == 0) { nodeRange[2 * (size_t)n] = lo; nodeRange[2 * (size_t)n + 1] = hi; }
}

// sparse loads (.vbx brick lists): the tables of the few existing positions are scattered into memset tables
__global__ __launch_bounds__(256) void iso_scatter_tables(int n, const long long* __restrict__ index, const int32_t* __restrict__ slotv,
                                                         const uint8_t* __restrict__ leafv, const float* __restrict__ rangev,
                                                         int32_t* __restrict__ slot, uint8_t* __restrict__ leaf, float* __restrict__ range)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long long b = index[i];
    slot[b] = slotv[i];
    leaf[b] = leafv[i];
    range[2 * b] = rangev[2 * i];
    range[2 * b + 1] = rangev[2 * i + 1];
}

}  // namespace

void iso_launch_node_range(const uint8_t* leaf, const float* leafRange, int nbx, int nby, int nbz, const int org[3],
                           int n1x, int n1y, int n1z, const int n1o[3], float* nodeRange, void* stream)
{
    hipLaunchKernelGGL(iso_node_range, dim3(n1x * n1y * n1z), dim3(64), 0, (hipStream_t)stream, leaf, leafRange, nbx, nby, nbz,
                       org[0], org[1], org[2], n1x, n1y, n1o[0], n1o[1], n1o[2], nodeRange);
}

void iso_launch_scatter_tables(int n, const long long* index, const int32_t* slotv, const uint8_t* leafv, const float* rangev,
                               int32_t* slot, uint8_t* leaf, float* range, void* stream)
{
    if (n > 0)
        hipLaunchKernelGGL(iso_scatter_tables, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, index, slotv, leafv, rangev, slot, leaf, range);
}

void iso_launch_leaf_range(const float* dense, int nx, int ny, int nz, int nbx, int nby, int nbz, float* range, void* stream)
{
    hipLaunchKernelGGL(iso_leaf_range, dim3(nbx * nby * nbz), dim3(64), 0, (hipStream_t)stream, dense, nx, ny, nz, nbx, nby, nbz, range);
}

void iso_launch_march_flags(const uint8_t* exists, const float* range, int n, double iso, uint8_t* flags, void* stream)
{
    if (n > 0) hipLaunchKernelGGL(iso_march_flags, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, exists, range, n, iso, flags);
}

void iso_launch_render(const IsoRenderParams& p, int variant, void* stream, void* startEvent, void* stopEvent, int waveCap)
{
    const int tiles = ((p.W + 7) >> 3) * ((p.H + 7) >> 3);
    const dim3 grid(tiles), block(64);
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t e0 = (hipEvent_t)startEvent, e1 = (hipEvent_t)stopEvent;
    // the AO loop is a separate instantiation: it costs registers the SR-mode render (aosamples=0) should not pay
    if (variant == 1) {
        if (p.aoSamples > 0) hipExtLaunchKernelGGL(iso_render_lds<true>, grid, block, 0, st, e0, e1, 0, p);
        else hipExtLaunchKernelGGL(iso_render_lds<false>, grid, block, 0, st, e0, e1, 0, p);
    } else if (variant == 3 && (long long)p.nbx * p.nby * p.nbz <= 32768) {
        // one workgroup of eight tiles per launch slot; tiles are XCD-remapped inside render_gather_tile, so the eight
        // tiles of a workgroup are spread -- keep them neighbours instead: the remap is applied to the workgroup
        const size_t lds = (size_t)p.nbx * p.nby * p.nbz * sizeof(int32_t);
        static bool attr = false;
        if (!attr) {
            (void)hipFuncSetAttribute((const void*)iso_render_gather_ldsslot<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
            (void)hipFuncSetAttribute((const void*)iso_render_gather_ldsslot<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
            attr = true;
        }
        const dim3 g8((tiles + 7) / 8), b8(512);
        if (p.aoSamples > 0) hipExtLaunchKernelGGL(iso_render_gather_ldsslot<true>, g8, b8, lds, st, e0, e1, 0, p);
        else hipExtLaunchKernelGGL(iso_render_gather_ldsslot<false>, g8, b8, lds, st, e0, e1, 0, p);
    } else if (variant == 2) {
        const int waves = waveCap > 0 && waveCap < tiles ? (waveCap + 7) & ~7 : (tiles + 7) & ~7;   // 0 = one wave per tile
        const dim3 capped(waves / 4), block4(256);                                                  // 4 waves per workgroup
        (void)hipMemsetAsync(p.tileQueue, 0, 8 * sizeof(unsigned), st);
        // the one-sample flat traversal: 116 registers on its own, 21 cold values spilled under the 128-register cap
        // (the two-sample form spills 77 and the frame is slower with it: 379 vs 393 frames/s)
        if (p.aoSamples > 0) hipExtLaunchKernelGGL((iso_render_gather_slim<true, 1>), capped, block4, 0, st, e0, e1, 0, p);
        else hipExtLaunchKernelGGL((iso_render_gather_slim<false, 1>), capped, block4, 0, st, e0, e1, 0, p);
    } else if (variant == 4) {
        if (p.aoSamples > 0) hipExtLaunchKernelGGL((iso_render_gather<true, 0>), grid, block, 0, st, e0, e1, 0, p);
        else hipExtLaunchKernelGGL((iso_render_gather<false, 0>), grid, block, 0, st, e0, e1, 0, p);
    } else if (variant == 5) {
        if (p.aoSamples > 0) hipExtLaunchKernelGGL((iso_render_gather<true, 1>), grid, block, 0, st, e0, e1, 0, p);
        else hipExtLaunchKernelGGL((iso_render_gather<false, 1>), grid, block, 0, st, e0, e1, 0, p);
    } else {
        if (p.aoSamples > 0) hipExtLaunchKernelGGL((iso_render_gather<true, 3>), grid, block, 0, st, e0, e1, 0, p);
        else hipExtLaunchKernelGGL((iso_render_gather<false, 3>), grid, block, 0, st, e0, e1, 0, p);
    }
}

void iso_launch_render_from_block(const IsoRenderParams& p, const IsoFrameBlock* deviceBlock, void* stream)
{
    const int tiles = ((p.W + 7) >> 3) * ((p.H + 7) >> 3);
    hipLaunchKernelGGL(iso_render_gather_block, dim3(tiles), dim3(64), 0, (hipStream_t)stream, p, deviceBlock);
}

void iso_launch_write_block(const IsoFrameBlock& block, IsoFrameBlock* deviceDst, void* stream)
{
    hipLaunchKernelGGL(iso_write_block_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, block, deviceDst);
}

void iso_launch_ao_distances(const IsoRenderParams& p, const double* hitState, const float* gbuf, double* dist, void* stream)
{
    const int tiles = ((p.W + 7) >> 3) * ((p.H + 7) >> 3);
    hipLaunchKernelGGL(iso_ao_dist_kernel, dim3(tiles), dim3(64), 0, (hipStream_t)stream, p, hitState, gbuf, dist);
}

void iso_launch_ao_finish(const IsoRenderParams& p, const double* dist, float* gbuf, void* stream)
{
    const size_t n = (size_t)p.W * p.H;
    hipLaunchKernelGGL(iso_ao_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, dist, gbuf);
}

void iso_launch_tile_order(const unsigned* cost, unsigned short* order, int n, int mode, int slots, void* stream)
{
    if (n <= 0 || n > ISO_ORDER_MAX_TILES) return;
    hipLaunchKernelGGL(iso_tile_order_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, cost, order, n, mode, slots);
}

void iso_launch_render_stats(const IsoRenderParams& p, int variant, long long* out, void* stream)
{
    const int tiles = ((p.W + 7) >> 3) * ((p.H + 7) >> 3);
    if (variant == 4) hipLaunchKernelGGL(iso_render_stats<0>, dim3(tiles), dim3(64), 0, (hipStream_t)stream, p, out);
    else if (variant == 5) hipLaunchKernelGGL(iso_render_stats<1>, dim3(tiles), dim3(64), 0, (hipStream_t)stream, p, out);
    else hipLaunchKernelGGL(iso_render_stats<3>, dim3(tiles), dim3(64), 0, (hipStream_t)stream, p, out);
}

void iso_launch_gate(const unsigned* resident, unsigned target, int timeoutUs, void* stream)
{
    hipLaunchKernelGGL(iso_gate_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, resident, target, (long long)timeoutUs * 100);
}

void iso_launch_brick_flags(const float* dense, int nx, int ny, int nz, int nbx, int nby, int nbz,
                            uint8_t* flag9, uint8_t* leaf, int* bbox6, unsigned int* maxbits, void* stream)
{
    hipLaunchKernelGGL(iso_brick_flags, dim3(nbx * nby * nbz), dim3(64), 0, (hipStream_t)stream,
                       dense, nx, ny, nz, nbx, nby, nbz, flag9, leaf, bbox6, maxbits);
}

void iso_launch_brick_fill(const float* dense, int nx, int ny, int nz, int nbx, int nby, int nbz,
                           const int32_t* slot, float* bricks, void* stream)
{
    hipLaunchKernelGGL(iso_brick_fill, dim3(nbx * nby * nbz), dim3(64), 0, (hipStream_t)stream,
                       dense, nx, ny, nz, nbx, nby, nbz, slot, bricks);
}
