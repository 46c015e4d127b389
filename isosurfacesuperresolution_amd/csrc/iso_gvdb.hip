// `semantics=gvdb`: the G-buffer with the ARITHMETIC of the reference's CUDA renderer (right
// column; 8(f) rank 3) -- what the released networks were trained on -- on the same brick store as the default
// (CPU-renderer) semantics.
//
//   sampling   cell-centred trilinear (voxel i at i + 0.5), GPURendererDirect/render_kernel.cu:172 (tex3D)
//   march      fixed 0.05-voxel steps inside occupied 8^3 bricks, 10 bisections, the OUTSIDE end is the hit (:159-199)
//   isovalue   absolute (GPURendererDirect.cpp:364);  world scale: longest edge -> 0.5 (:276-278)
//   camera     GVDB's Camera3D: image half-width tangent tan(fov/2)/2, near .1, far 5000 (gvdb_camera.cpp:425-489)
//   outputs    Phong with (e+2)/(2*3.41) and the eye direction (:232-237), flow = .5 * delta NDC (:239-245),
//              depth = NDC z (:247), outward view-space normal, no flip (:249), ray-cast AO (:109-146), shadow = 1
//
// The texture unit's fixed-point filter weights and --use_fast_math are not reproducible: this
// kernel computes in IEEE float (the translation unit is built with -ffp-contract=off) and is checked bit for bit
// against the CPU restatement of the same algorithm in oracle/iso_oracle_gvdb.c, which documents the deviations
// (single-level brick DDA instead of GVDB's 5-level tree walk; no 256-iteration cap on that walk).
// One wave64 = one 8x8 pixel tile, XCD-aware tile order, as in iso_kernels.hip.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "iso_params.h"

namespace {

constexpr float GV_PSTEP = 0.05f;     // GPURendererDirect.cpp:365
constexpr float GV_EPS = 0.001f;      // gvdb_volume_gvdb.cpp:116
constexpr int GV_MAX_ITER = 256;      // cuda_gvdb_raycast.cuh:37

struct f3 { float x, y, z; };
__device__ __forceinline__ f3 mk(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ f3 add(f3 a, f3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ f3 sub(f3 a, f3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ f3 scale(f3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ f3 cross(f3 a, f3 b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
__device__ __forceinline__ f3 normalize(f3 a) { const float l = sqrtf(dot(a, a)); return mk(a.x / l, a.y / l, a.z / l); }
__device__ __forceinline__ f3 safe_normalize(f3 a)          // render_kernel.cu:150-155
{
    const float l = sqrtf(dot(a, a));
    if (!(l > 1e-6f)) return mk(0.f, 0.f, 0.f);
    return mk(a.x / l, a.y / l, a.z / l);
}

// (x, y, z): GLOBAL index coordinates; a tile of a larger volume stores the region that starts at P.org (zero for a whole volume)
__device__ __forceinline__ float voxel(const IsoRenderParams& P, int x, int y, int z)
{
    x -= P.org[0]; y -= P.org[1]; z -= P.org[2];
    if ((unsigned)x >= (unsigned)P.nx || (unsigned)y >= (unsigned)P.ny || (unsigned)z >= (unsigned)P.nz) return 0.0f;
    const int s = P.slot[((z >> 3) * P.nby + (y >> 3)) * P.nbx + (x >> 3)];
    if (s < 0) return 0.0f;
    return P.bricks[(size_t)s * ISO_BRICK_STRIDE + ((z & 7) * 9 + (y & 7)) * 9 + (x & 7)];
}

// tex3D with linear filtering at grid-local position q; the 8 corners of an in-grid cell live in ONE stored
// brick (own voxels + the +1 apron), so the common case is a single slot lookup
__device__ __forceinline__ float tex(const IsoRenderParams& P, f3 q)
{
    const float fx = q.x - 0.5f, fy = q.y - 0.5f, fz = q.z - 0.5f;
    const float cx = floorf(fx), cy = floorf(fy), cz = floorf(fz);
    const int ix = (int)cx, iy = (int)cy, iz = (int)cz;
    const float a = fx - cx, b = fy - cy, c = fz - cz;
    float v000, v100, v010, v110, v001, v101, v011, v111;
    const int lx = ix - P.org[0], ly = iy - P.org[1], lz = iz - P.org[2];       // stored (tile-local) index; org is a multiple of 8
    if ((unsigned)lx < (unsigned)P.nx && (unsigned)ly < (unsigned)P.ny && (unsigned)lz < (unsigned)P.nz) {
        const int s = P.slot[((lz >> 3) * P.nby + (ly >> 3)) * P.nbx + (lx >> 3)];
        if (s < 0) return 0.0f;                              // all 8 corners are zero: 0 + a * 0 ... == +0
        const float* w = P.bricks + (size_t)s * ISO_BRICK_STRIDE + ((lz & 7) * 9 + (ly & 7)) * 9 + (lx & 7);
        v000 = w[0]; v100 = w[1]; v010 = w[9]; v110 = w[10];
        v001 = w[81]; v101 = w[82]; v011 = w[90]; v111 = w[91];
    } else {
        v000 = voxel(P, ix, iy, iz); v100 = voxel(P, ix + 1, iy, iz);
        v010 = voxel(P, ix, iy + 1, iz); v110 = voxel(P, ix + 1, iy + 1, iz);
        v001 = voxel(P, ix, iy, iz + 1); v101 = voxel(P, ix + 1, iy, iz + 1);
        v011 = voxel(P, ix, iy + 1, iz + 1); v111 = voxel(P, ix + 1, iy + 1, iz + 1);
    }
    const float x00 = v000 + a * (v100 - v000), x10 = v010 + a * (v110 - v010);
    const float x01 = v001 + a * (v101 - v001), x11 = v011 + a * (v111 - v011);
    const float y0 = x00 + b * (x10 - x00), y1 = x01 + b * (x11 - x01);
    return y0 + c * (y1 - y0);
}

__device__ __forceinline__ f3 gradient(const IsoRenderParams& P, f3 q)   // cuda_gvdb_raycast.cuh:132-141
{
    f3 g;
    g.x = tex(P, mk(q.x - 0.5f, q.y, q.z)) - tex(P, mk(q.x + 0.5f, q.y, q.z));
    g.y = tex(P, mk(q.x, q.y - 0.5f, q.z)) - tex(P, mk(q.x, q.y + 0.5f, q.z));
    g.z = tex(P, mk(q.x, q.y, q.z - 0.5f)) - tex(P, mk(q.x, q.y, q.z + 0.5f));
    return g;
}

// render_kernel.cu:159-199 (CUSTOM: with the bisection) / cuda_gvdb_raycast.cuh:255-277 (plain, for the AO rays)
template <bool CUSTOM>
__device__ __forceinline__ bool march_brick(const IsoRenderParams& P, float iso, f3 vmin, float t, f3 pos, f3 dir, f3& hit, f3& grad)
{
    f3 p = sub(add(pos, scale(dir, t)), vmin);
    const f3 pstart = p;
    float tcur = 0.0f;
    for (int iter = 0; iter < GV_MAX_ITER && p.x >= 0 && p.y >= 0 && p.z >= 0 && p.x < 8.0f && p.y < 8.0f && p.z < 8.0f; ++iter) {
        if (tex(P, add(p, vmin)) >= iso) {
            if (CUSTOM) {
                float lo = tcur - GV_PSTEP, hi = tcur;
                for (int i = 0; i < 10; ++i) {
                    const float mid = 0.5f * (lo + hi);
                    p = add(pstart, scale(dir, mid));
                    if (tex(P, add(p, vmin)) >= iso) hi = mid; else lo = mid;
                }
                p = add(pstart, scale(dir, lo));
            }
            hit = add(p, vmin);
            grad = gradient(P, hit);
            return true;
        }
        p = add(p, scale(dir, GV_PSTEP));
        tcur += GV_PSTEP;
    }
    return false;
}

// rayCast (cuda_gvdb_raycast.cuh:504-575) on a single-level DDA over the bricks of the bounding box
template <bool CUSTOM>
__device__ __forceinline__ bool ray_cast(const IsoRenderParams& P, float iso, f3 pos, f3 dir, f3& hit, f3& grad)
{
    if (!P.any_leaf) return false;
    const f3 bmin = mk((float)P.bbmin[0], (float)P.bbmin[1], (float)P.bbmin[2]);
    const f3 bmax = mk((float)P.bbmax[0], (float)P.bbmax[1], (float)P.bbmax[2]);
    const float h0 = (bmin.x - pos.x) / dir.x, h1 = (bmax.x - pos.x) / dir.x;
    const float h2 = (bmin.y - pos.y) / dir.y, h3 = (bmax.y - pos.y) / dir.y;
    const float h4 = (bmin.z - pos.z) / dir.z, h5 = (bmax.z - pos.z) / dir.z;
    float tin = fmaxf(fmaxf(fminf(h0, h1), fminf(h2, h3)), fminf(h4, h5));
    const float tout = fminf(fminf(fmaxf(h0, h1), fmaxf(h2, h3)), fmaxf(h4, h5));
    tin = (tin < 0.f) ? 0.0f : tin;
    if (tout < tin || tout < 0.f) return false;
    float tx = tin + GV_EPS;
    const float tmax = tout - GV_EPS;
    const f3 pstep = mk(dir.x > 0.f ? 1.f : -1.f, dir.y > 0.f ? 1.f : -1.f, dir.z > 0.f ? 1.f : -1.f);
    f3 p = scale(sub(add(pos, scale(dir, tx)), bmin), 0.125f);
    const f3 tdel = mk(fabsf(8.0f / dir.x), fabsf(8.0f / dir.y), fabsf(8.0f / dir.z));
    const f3 fl = mk(floorf(p.x), floorf(p.y), floorf(p.z));
    f3 tside = mk(((fl.x - p.x + 0.5f) * pstep.x + 0.5f) * tdel.x + tx,
                  ((fl.y - p.y + 0.5f) * pstep.y + 0.5f) * tdel.y + tx,
                  ((fl.z - p.z + 0.5f) * pstep.z + 0.5f) * tdel.z + tx);
    p = fl;
    const int ox = P.bbmin[0] >> 3, oy = P.bbmin[1] >> 3, oz = P.bbmin[2] >> 3;
    const float rx = (float)((P.bbmax[0] - P.bbmin[0]) >> 3), ry = (float)((P.bbmax[1] - P.bbmin[1]) >> 3), rz = (float)((P.bbmax[2] - P.bbmin[2]) >> 3);
    for (int iter = 0; iter < 4096 && p.x >= 0 && p.y >= 0 && p.z >= 0 && p.x < rx && p.y < ry && p.z < rz && tx <= tmax; ++iter) {
        const float mx = (float)((tside.x < tside.y) & (tside.x <= tside.z));
        const float my = (float)((tside.y < tside.z) & (tside.y <= tside.x));
        const float mz = (float)((tside.z < tside.x) & (tside.z <= tside.y));
        const float ty = mx != 0.f ? tside.x : (my != 0.f ? tside.y : tside.z);
        const int bx = ox + (int)p.x, by = oy + (int)p.y, bz = oz + (int)p.z;                  // global brick coordinates
        // a tile of a larger volume (P.org != 0 or a smaller table) walks the same bricks of the GLOBAL box and marches the ones it
        // owns (its `leaf` table says "exists AND owned"); what a march computes depends on the brick's entry time and voxels only,
        // so the tile that owns the first brick with a hit produces the unsplit pixel
        const int tbx = bx - (P.org[0] >> 3), tby = by - (P.org[1] >> 3), tbz = bz - (P.org[2] >> 3);
        const bool stored = (unsigned)tbx < (unsigned)P.nbx && (unsigned)tby < (unsigned)P.nby && (unsigned)tbz < (unsigned)P.nbz;
        const size_t bi = stored ? ((size_t)tbz * P.nby + tby) * P.nbx + tbx : 0;
        // max skipping, exact: a march through this brick reads voxels of [8b-1, 8b+9]^3 only; if their maximum (plus the
        // rounding allowance of the float lerps) is below the isovalue no sample can reach it and the march finds nothing
        if (stored && P.leaf[bi] && !(iso > P.leafRange[2 * bi + 1] + 4e-6f * fmaxf(fabsf(P.leafRange[2 * bi]), fabsf(P.leafRange[2 * bi + 1])))) {
            const f3 vmin = mk((float)(bx * 8), (float)(by * 8), (float)(bz * 8));
            if (march_brick<CUSTOM>(P, iso, vmin, tx + GV_EPS, pos, dir, hit, grad)) return true;
        }
        tx = ty;
        tside = mk(tside.x + mx * tdel.x, tside.y + my * tdel.y, tside.z + mz * tdel.z);
        p = mk(p.x + mx * pstep.x, p.y + my * pstep.y, p.z + mz * pstep.z);
    }
    return false;
}

__device__ __forceinline__ void mat4_apply(const float* m, f3 w, float (&out)[4])
{
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i] = m[4 * i + 0] * w.x + m[4 * i + 1] * w.y + m[4 * i + 2] * w.z + m[4 * i + 3] * 1.0f;
}

__device__ float ambient_occlusion(const IsoRenderParams& P, const IsoGvdbFrame& F, f3 pos, f3 normal, int x, int y)
{
    const float* nz = P.aoRot + 4 * ((x % 4) + 4 * (y % 4));
    const f3 noise = mk(nz[0], nz[1], nz[2]);
    const f3 tangent = normalize(sub(noise, scale(normal, dot(noise, normal))));
    const f3 bitangent = cross(normal, tangent);
    float ao = 0.0f;
    const int n = P.aoSamples;
    for (int i = 0; i < n; ++i) {
        const f3 st = normalize(mk(P.aoHemi[4 * i], P.aoHemi[4 * i + 1], P.aoHemi[4 * i + 2]));
        const f3 sw = mk(dot(mk(tangent.x, bitangent.x, normal.x), st),
                         dot(mk(tangent.y, bitangent.y, normal.y), st),
                         dot(mk(tangent.z, bitangent.z, normal.z), st));
        f3 h, g;
        float value = 1.0f;
        if (ray_cast<false>(P, F.iso, pos, sw, h, g)) {
            const f3 d = sub(pos, h);
            const float dist = sqrtf(dot(d, d));
            float yv = 1.0f - F.aoRadius / dist;                       // smoothstep(1, 0, r / d)
            yv = yv < 0.0f ? 0.0f : (yv > 1.0f ? 1.0f : yv);
            value = yv * yv * (3.0f - (2.0f * yv));
        }
        ao += value;
    }
    return ao / (float)n;
}

__device__ __forceinline__ int xcd_remap(int bid, int nwg)
{
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

template <bool AO>
__global__ __launch_bounds__(64) void iso_render_gvdb(const IsoRenderParams P, const IsoGvdbFrame F)
{
    const int tiles_x = (P.W + 7) >> 3, ntiles = tiles_x * ((P.H + 7) >> 3);
    const int tile = xcd_remap(blockIdx.x, ntiles);
    const int lane = threadIdx.x;
    const int x = (tile % tiles_x) * 8 + (lane & 7);
    const int y = (tile / tiles_x) * 8 + (lane >> 3);
    if (x >= P.W || y >= P.H) return;
    float o[12] = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 1.0f, 1.0f };   // ao, shadow: render_kernel.cu:219-220
    if (x >= P.vp[0] && y >= P.vp[1] && x < P.vp[2] && y < P.vp[3]) {
        const f3 rpos = mk(F.rpos[0], F.rpos[1], F.rpos[2]);
        const float u = ((float)x + 0.5f) / (float)P.W, w = ((float)y + 0.5f) / (float)P.H;
        const f3 camu = mk(F.camu[0], F.camu[1], F.camu[2]), camv = mk(F.camv[0], F.camv[1], F.camv[2]), cams = mk(F.cams[0], F.cams[1], F.cams[2]);
        const f3 rdir = normalize(add(add(scale(camu, u), scale(camv, w)), cams));
        f3 hit, g;
        if (ray_cast<true>(P, F.iso, rpos, rdir, hit, g)) {
            const f3 n = safe_normalize(g);
            const f3 light = mk(F.light[0], F.light[1], F.light[2]);
            o[3] = 1.0f;
            const f3 eye = normalize(sub(rpos, hit));
            const float ndl = dot(n, light);
            const f3 R = normalize(sub(light, scale(n, 2.0f * ndl)));
            float s = dot(R, eye);
            s = s > 0.0f ? s : 0.0f;
            float pw = 1.0f;
            for (int k = 0; k < F.exponent; ++k) pw *= s;
            const float andl = fabsf(ndl);
            o[0] = F.ambient[0] + F.diffuse[0] * andl + F.specular[0] * F.spec_c * pw;
            o[1] = F.ambient[1] + F.diffuse[1] * andl + F.specular[1] * F.spec_c * pw;
            o[2] = F.ambient[2] + F.diffuse[2] * andl + F.specular[2] * F.spec_c * pw;
            const f3 world = mk(F.scale * hit.x + F.tr[0], F.scale * hit.y + F.tr[1], F.scale * hit.z + F.tr[2]);
            float sc[4], sn[4];
            mat4_apply(F.cur, world, sc);
            mat4_apply(F.nxt, world, sn);
            const float cx = sc[0] / sc[3], cy = sc[1] / sc[3], cz = sc[2] / sc[3];
            const float nx_ = sn[0] / sn[3], ny_ = sn[1] / sn[3];
            o[8] = 0.5f * (cx - nx_); o[9] = 0.5f * (cy - ny_);
            o[7] = cz;
            o[4] = F.vrot[0] * n.x + F.vrot[1] * n.y + F.vrot[2] * n.z;
            o[5] = F.vrot[3] * n.x + F.vrot[4] * n.y + F.vrot[5] * n.z;
            o[6] = F.vrot[6] * n.x + F.vrot[7] * n.y + F.vrot[8] * n.z;
            if (AO) o[10] = ambient_occlusion(P, F, sub(hit, scale(rdir, 1e-3f)), n, x, y);
        }
    }
    float4* dst = reinterpret_cast<float4*>(P.out + ((size_t)y * P.W + x) * 12);
    dst[0] = make_float4(o[0], o[1], o[2], o[3]);
    dst[1] = make_float4(o[4], o[5], o[6], o[7]);
    dst[2] = make_float4(o[8], o[9], o[10], o[11]);
}

}  // namespace

void iso_launch_render_gvdb(const IsoRenderParams& p, const IsoGvdbFrame& f, void* stream, void* startEvent, void* stopEvent)
{
    const int tiles = ((p.W + 7) >> 3) * ((p.H + 7) >> 3);
    const dim3 grid(tiles), block(64);
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t e0 = (hipEvent_t)startEvent, e1 = (hipEvent_t)stopEvent;
    if (p.aoSamples > 0) hipExtLaunchKernelGGL(iso_render_gvdb<true>, grid, block, 0, st, e0, e1, 0, p, f);
    else hipExtLaunchKernelGGL(iso_render_gvdb<false>, grid, block, 0, st, e0, e1, 0, p, f);
}
