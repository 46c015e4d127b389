// Host side of libGPURendererDirect.so: process-global renderer state, the reference's
// string-keyed parameter protocol, volume upload (dense -> 9^3 apron bricks + occupancy
// hierarchy, built on the GPU), camera/matrix set-up in fp64, launch and timing.
//
// Replaces GPURendererDirect/GPURendererDirect.cpp:34-446 (exports at :228-246, :248-285,
// :393-428, :430-446).  Camera mathematics follow the CPU renderer, whose values this library
// reproduces: TP/openvdb/tools/RayTracer.h:404-531, TP/openvdb/math/Mat.h:758-774,
// TP/openvdb/math/Mat4.h:531-672, CPURenderer/CPURenderer.cpp:448-458,484-507.
// Compiled with -ffp-contract=off (bit-exact camera constants vs. the CPU restatement).
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../../include/gpu_renderer_direct.h"
#include "iso_params.h"
#include "vbx_reader.h"

namespace {

struct Args {   // GPURendererDirect.cpp:103-128
    int resolutionX = 512, resolutionY = 512;
    double cameraFov = 45;
    double cameraOrigin[3] = { 0, 0, -1 };
    double cameraLookAt[3] = { 0, 0, 0 };
    double cameraUp[3] = { 0, 1, 0 };
    int noShading = 0;
    int viewport[4] = { 0, 0, 512, 512 };
    double isovalue = 0.0;
    double materialDiffuse[3] = { 0.7, 0.7, 0.7 };
    double materialSpecular[3] = { 1, 1, 1 };
    double materialAmbient[3] = { 0.1, 0.1, 0.1 };
    int materialSpecularExponent = 32;
    bool cameraLight = true;
    double lightDirection[3] = { 0, 0, 0 };
    int aoSamples = 32;
    float aoRadius = 0.01f;
};

struct Volume {
    bool loaded = false;
    int nx = 0, ny = 0, nz = 0, nbx = 0, nby = 0, nbz = 0, n1x = 0, n1y = 0, n1z = 0;
    int nslots = 0, nleaf = 0;
    int org[3] = { 0, 0, 0 };    // global index of stored voxel (0,0,0); non-zero only for a tile of a larger volume
    int n1o[3] = { 0, 0, 0 };
    bool tile = false;
    int bbmin[3] = { 0, 0, 0 }, bbmax[3] = { 0, 0, 0 };
    float maxValue = 0.f;
    double s = 1, sinv = 1, t[3] = { 0, 0, 0 };
    float* bricks = nullptr;
    int32_t* slot = nullptr;
    uint8_t* leaf = nullptr;
    float* leafRange = nullptr;  // (min, max) per brick position, see iso_kernels.hip: leaf_may_cross
    uint8_t* node1 = nullptr;
    float* node1Range = nullptr; // (min, max) over the ranges of a 128^3 node's existing leaves
    uint8_t* marchFlags = nullptr;   // per leaf, then per 128^3 node: exists / must be marched at marchIso (iso_march_flags)
    double marchIso = 0;
    bool marchValid = false;
};

struct State {
    bool initialised = false;
    Args args;
    double lastOrigin[3] = { 0, 0, -1 }, lastLookAt[3] = { 0, 0, 0 };
    Volume vol;
    int variant = 0;
    int waveCap = 0;             // see isoSetWaveCap
    int semantics = 0;           // 0: reference CPU renderer (default), 1: reference CUDA renderer (setParameter("semantics", "gvdb"))
    float* aoHemi = nullptr;     // device copies of the AO tables
    float* aoRot = nullptr;
    unsigned* tileQueue = nullptr;   // 8 per-XCD work counters of kernel variant 2 (+ 1 resident-wave counter)
    unsigned residentTarget = 0;     // waves launched by all variant-2 renders so far (what the counter will reach)
    unsigned gatedTarget = 0;        // residentTarget at the last isoGateResident: a gate with nothing new to wait for is a no-op
    // cost-ordered dispatch of the default kernel (isoSetTileOrderMode): the waves of frame t write their clock cycles, one
    // workgroup sorts them on the render's stream, frame t + 1 dispatches its tiles in that order (same pixels, other order)
    int orderMode = 0;               // 0 off, 1 heaviest first, 2 heaviest first then the lightest as the SIMDs' second waves
    unsigned* tileCost = nullptr;    // [ISO_ORDER_MAX_TILES]
    unsigned short* tileOrder = nullptr;   // [ISO_ORDER_MAX_TILES]
    int orderW = 0, orderH = 0;      // resolution the stored order belongs to (0: none yet)
    double* hitState = nullptr;      // isoSetHitStateBuffer: caller-owned [H][W][6] doubles the renders export their AO ray set-up to
    long long* statsOut = nullptr;   // diagnostics: isoDebugSetStatsBuffer
    bool profile = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;   // one pair per profiled frame
};
State g;

#define HIP_OK(expr)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            std::fprintf(stderr, "GPURendererDirect: HIP error '%s' in %s (%s:%d)\n",        \
                         hipGetErrorString(e_), #expr, __FILE__, __LINE__);                  \
            return false;                                                                    \
        }                                                                                    \
    } while (0)

void freeVolume(Volume& v)
{
    if (v.bricks) (void)hipFree(v.bricks);
    if (v.slot) (void)hipFree(v.slot);
    if (v.leaf) (void)hipFree(v.leaf);
    if (v.leafRange) (void)hipFree(v.leafRange);
    if (v.node1) (void)hipFree(v.node1);
    if (v.node1Range) (void)hipFree(v.node1Range);
    if (v.marchFlags) (void)hipFree(v.marchFlags);
    v = Volume();
}

// A volume under construction and the upload's device temporaries: released on every early return (HIP_OK returns false from the
// middle of an upload; a sparse or hostile .vbx that fails half way must not leak device memory on each loadGrid that reports -2).
// handOver(): the volume is complete and finalizeVolume -- which frees it itself when it fails -- takes it from here.
struct UploadGuard {
    Volume& v;
    void** tmp[4] = { nullptr, nullptr, nullptr, nullptr };
    bool armed = true;
    explicit UploadGuard(Volume& vol) : v(vol) {}
    void watch(int i, void** q) { tmp[i] = q; }
    void freeTemporaries() { for (void**& q : tmp) if (q && *q) { (void)hipFree(*q); *q = nullptr; } }
    void handOver() { freeTemporaries(); armed = false; }
    ~UploadGuard() { freeTemporaries(); if (armed) freeVolume(v); }
};

// ---- "a,b,c" parsing (GPURendererDirect.cpp:60-85); strict arity, no exceptions ------------
bool splitNumbers(const char* s, int n, double* out)
{
    const char* p = s;
    for (int k = 0; k < n; ++k) {
        char* end = nullptr;
        out[k] = std::strtod(p, &end);
        if (end == p) return false;
        while (*end == ' ' || *end == '\t') ++end;
        if (k + 1 < n) {
            if (*end != ',') return false;
            p = end + 1;
        } else if (*end != '\0' && *end != '\n' && *end != '\r') {
            return false;
        }
    }
    return true;
}

// ---- fp64 vector helpers with OpenVDB's operation order ------------------------------------
inline double len3(const double a[3]) { return std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }
inline void unit3(const double a[3], double o[3])
{
    const double l = len3(a);
    o[0] = a[0] / l; o[1] = a[1] / l; o[2] = a[2] / l;
}
inline void cross3(const double a[3], const double b[3], double o[3])
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
inline void normalize3(double a[3])
{
    const double d = len3(a);
    if (!(std::fabs(d - 0.0) > 1.0e-7)) return;
    const double r = 1.0 / d;
    a[0] *= r; a[1] *= r; a[2] *= r;
}

void buildCamera(IsoCamera& c, const double origin[3], const double lookAt[3], const double up[3],
                 double fov, int W, int H)
{
    const double aperture = 0.01;
    const double focal = aperture / (2.0 * (std::tan(fov * M_PI / 360.0)));
    c.sw = 0.5 * aperture / focal;
    c.sh = c.sw * double(H) / double(W);
    const double dir[3] = { origin[0] - lookAt[0], origin[1] - lookAt[1], origin[2] - lookAt[2] };
    double fwd[3], upn[3], hor[3], upv[3], tmp[3];
    unit3(dir, fwd);
    unit3(up, upn);
    cross3(upn, fwd, tmp); unit3(tmp, hor);
    cross3(fwd, hor, tmp); unit3(tmp, upv);
    for (int k = 0; k < 3; ++k) {
        c.J[0][k] = hor[k]; c.J[1][k] = upv[k]; c.J[2][k] = fwd[k];
        c.org[k] = origin[k];
        c.d0[k] = 0.0 * hor[k] + 0.0 * upv[k] + (-1.0) * fwd[k];
    }
    const double (*m)[3] = c.J;
    const double m0011 = m[0][0] * m[1][1], m0012 = m[0][0] * m[1][2], m0110 = m[0][1] * m[1][0];
    const double m0210 = m[0][2] * m[1][0], m0120 = m[0][1] * m[2][0], m0220 = m[0][2] * m[2][0];
    double detA = m0011 * m[2][2] - m0012 * m[2][1] - m0110 * m[2][2]
                + m0210 * m[2][1] + m0120 * m[1][2] - m0220 * m[1][1];
    detA = 1.0 / detA;
    double (*inv)[4] = c.V;
    inv[0][0] = detA * ( m[1][1] * m[2][2] - m[1][2] * m[2][1]);
    inv[0][1] = detA * (-m[0][1] * m[2][2] + m[0][2] * m[2][1]);
    inv[0][2] = detA * ( m[0][1] * m[1][2] - m[0][2] * m[1][1]);
    inv[1][0] = detA * (-m[1][0] * m[2][2] + m[1][2] * m[2][0]);
    inv[1][1] = detA * ( m[0][0] * m[2][2] - m0220);
    inv[1][2] = detA * ( m0210 - m0012);
    inv[2][0] = detA * ( m[1][0] * m[2][1] - m[1][1] * m[2][0]);
    inv[2][1] = detA * ( m0120 - m[0][0] * m[2][1]);
    inv[2][2] = detA * ( m0011 - m0110);
    const double* o = c.org;
    inv[3][0] = -(o[0] * inv[0][0] + o[1] * inv[1][0] + o[2] * inv[2][0]);
    inv[3][1] = -(o[0] * inv[0][1] + o[1] * inv[1][1] + o[2] * inv[2][1]);
    inv[3][2] = -(o[0] * inv[0][2] + o[1] * inv[1][2] + o[2] * inv[2][2]);
    inv[0][3] = 0.0; inv[1][3] = 0.0; inv[2][3] = 0.0; inv[3][3] = 1.0;
}

// AO tables of GPURendererDirect.cpp:146-189, generated with an explicit minstd_rand0 (seed 1): the
// reference's unseeded std::default_random_engine is implementation defined.
unsigned lcgNext(unsigned& st)
{
    st = unsigned((static_cast<unsigned long long>(st) * 16807ULL) % 2147483647ULL);
    return st;
}
float lcgUniform(unsigned& st) { return float(lcgNext(st) - 1u) * (1.0f / 2147483646.0f); }

bool uploadAoTables()
{
    std::vector<float> hemi(512 * 4), rot(16 * 4);
    unsigned st = 1u;
    for (int i = 0; i < 512; ++i) {
        const float u1 = lcgUniform(st), u2 = lcgUniform(st);
        const float r = sqrtf(u1);
        const float theta = float(2 * M_PI * u2);
        const float x = r * cosf(theta), y = r * sinf(theta);
        float scale = lcgUniform(st);
        scale = float(0.1 + 0.9 * scale * scale);
        hemi[4 * i + 0] = x * scale; hemi[4 * i + 1] = y * scale;
        hemi[4 * i + 2] = sqrtf(1 - u1) * scale; hemi[4 * i + 3] = 0.f;
    }
    for (int i = 0; i < 16; ++i) {
        const float x = lcgUniform(st) * 2 - 1, y = lcgUniform(st) * 2 - 1;
        const float linv = 1.0f / sqrtf(x * x + y * y);
        rot[4 * i + 0] = x * linv; rot[4 * i + 1] = y * linv; rot[4 * i + 2] = 0.f; rot[4 * i + 3] = 0.f;
    }
    HIP_OK(hipMalloc(&g.aoHemi, hemi.size() * sizeof(float)));
    HIP_OK(hipMalloc(&g.aoRot, rot.size() * sizeof(float)));
    if (!g.tileQueue) {
        HIP_OK(hipMalloc(&g.tileQueue, 16 * sizeof(unsigned)));
        HIP_OK(hipMemset(g.tileQueue, 0, 16 * sizeof(unsigned)));
        g.residentTarget = 0;
    }
    HIP_OK(hipMemcpy(g.aoHemi, hemi.data(), hemi.size() * sizeof(float), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(g.aoRot, rot.data(), rot.size() * sizeof(float), hipMemcpyHostToDevice));
    return true;
}

float orderBitsToFloat(unsigned int u)
{
    u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

// Placement of a tile inside a larger (multi-GPU) volume.  A tile walks the GLOBAL ray:
// world map, isovalue scale and the node-level bounding box the ray is clipped to come from the global volume, the
// 4096/128/8 DDAs run in global index coordinates, and the tile merely owns a subset of the leaves (the others are
// stepped over like empty space).  The reference re-initialises the voxel DDA per leaf from the leaf's own span
// (CPURenderer/IsoVolumeRayTracer.h:37-46), so what a ray computes inside a leaf does not depend on which other leaves
// exist: the tile that owns the first leaf with a crossing produces bit for bit the pixel of the unsplit render, and a
// nearest-hit composite of the tiles IS the unsplit image.
struct TileInfo {
    int origin[3];                 // global index of local voxel (0,0,0), multiple of 8
    int gmin[3], gmax[3];          // global active-voxel bbox
    float globalMax;
    int clipLo[3], clipHi[3];      // leaves owned by this tile, global index coordinates, [lo, hi), lo a multiple of 8
};

// Node-level bbox, world map and installation of a volume whose device tables are complete.
// bbox: active-voxel bbox of the stored data (min xyz, max xyz); lmin / lmax: extreme leaf coordinates (bricks).
bool finalizeVolume(Volume& v, int bbox[6], float maxValue, int lmin[3], int lmax[3], const TileInfo* tile)
{
    if (v.nleaf == 0 && !tile) {   // the reference throws on empty grids (IsoVolumeRayTracer.h:188-190)
        freeVolume(v);
        return false;
    }
    if (v.nleaf == 0) {            // an empty tile of a larger volume renders nothing
        for (int k = 0; k < 3; ++k) { lmin[k] = 0; lmax[k] = -1; }
    }
    for (int k = 0; k < 3; ++k) {   // IsoVolumeRayTracer.h:195-197
        v.bbmin[k] = lmin[k] * 8;
        v.bbmax[k] = lmax[k] * 8 + 7 + 1;
    }
    v.maxValue = maxValue;
    // CPURenderer.cpp:448-458 with unit voxels: scale longest active-bbox edge to 1, centre at 0
    double ext[3], cen[3];
    if (tile) {
        for (int k = 0; k < 3; ++k) {
            bbox[k] = tile->gmin[k]; bbox[3 + k] = tile->gmax[k];
            // the ray is clipped to the GLOBAL node-level box: the leaf holding the extreme active voxel bounds it
            v.bbmin[k] = tile->gmin[k] & ~7;
            v.bbmax[k] = (tile->gmax[k] & ~7) + 7 + 1;
        }
        v.maxValue = tile->globalMax;
    }
    for (int k = 0; k < 3; ++k) {
        const double lo = double(bbox[k]), hi = double(bbox[3 + k]);
        ext[k] = hi - lo;
        cen[k] = (lo + hi) * 0.5;
    }
    double m = ext[0];
    if (ext[1] > m) m = ext[1];
    if (ext[2] > m) m = ext[2];
    if (!(m > 0)) { freeVolume(v); return false; }
    const double scale = 1.0 / m;
    v.s = 1.0 * scale;
    v.sinv = 1.0 / v.s;
    for (int k = 0; k < 3; ++k) v.t[k] = (-cen[k]) * scale;
    {
        const size_t n1 = size_t(v.n1x) * v.n1y * v.n1z;
        if (hipMalloc(&v.node1Range, n1 * 2 * sizeof(float)) != hipSuccess) { freeVolume(v); return false; }
        iso_launch_node_range(v.leaf, v.leafRange, v.nbx, v.nby, v.nbz, v.org, v.n1x, v.n1y, v.n1z, v.n1o, v.node1Range, nullptr);
        if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) { freeVolume(v); return false; }
    }
    v.loaded = true;
    freeVolume(g.vol);
    g.vol = v;
    // GPURendererDirect.cpp:280-281
    for (int k = 0; k < 3; ++k) { g.lastOrigin[k] = g.args.cameraOrigin[k]; g.lastLookAt[k] = g.args.cameraLookAt[k]; }
    return true;
}

// Dense device volume -> bricks, occupancy hierarchy, bbox, max, world map.
bool uploadFromDevice(const float* dense, int nx, int ny, int nz, const TileInfo* tile = nullptr)
{
    if (nx <= 0 || ny <= 0 || nz <= 0 || nx > 4096 || ny > 4096 || nz > 4096) return false;
    Volume v;
    v.nx = nx; v.ny = ny; v.nz = nz;
    v.nbx = (nx + 7) / 8; v.nby = (ny + 7) / 8; v.nbz = (nz + 7) / 8;
    const int dims[3] = { nx, ny, nz };
    if (tile) {
        v.tile = true;
        for (int k = 0; k < 3; ++k) {
            // the stored region must sit on the global leaf grid, inside the one 4096^3 level-2 node, and hold what a
            // march through an owned leaf [8b, 8b+8) can read: voxels [8b-2, 8b+10] (trilinear +1, gradient +-1)
            if (tile->origin[k] < 0 || (tile->origin[k] & 7) || (tile->clipLo[k] & 7) || tile->origin[k] + dims[k] > 4096) return false;
            if (tile->clipLo[k] < tile->origin[k] || tile->clipHi[k] > tile->origin[k] + dims[k]) return false;
            v.org[k] = tile->origin[k];
            v.n1o[k] = tile->origin[k] >> 7;
        }
    }
    v.n1x = ((v.org[0] + nx - 1) >> 7) - v.n1o[0] + 1;
    v.n1y = ((v.org[1] + ny - 1) >> 7) - v.n1o[1] + 1;
    v.n1z = ((v.org[2] + nz - 1) >> 7) - v.n1o[2] + 1;
    const size_t nb = size_t(v.nbx) * v.nby * v.nbz;
    uint8_t *dFlag9 = nullptr;
    int* dBBox = nullptr;
    unsigned int* dMax = nullptr;
    UploadGuard guard(v);
    guard.watch(0, (void**)&dFlag9); guard.watch(1, (void**)&dBBox); guard.watch(2, (void**)&dMax);
    HIP_OK(hipMalloc(&dFlag9, nb));
    HIP_OK(hipMalloc(&v.leaf, nb));
    HIP_OK(hipMalloc(&dBBox, 6 * sizeof(int)));
    HIP_OK(hipMalloc(&dMax, sizeof(unsigned int)));
    const int bboxInit[6] = { INT32_MAX, INT32_MAX, INT32_MAX, INT32_MIN, INT32_MIN, INT32_MIN };
    const unsigned int zero = 0;
    HIP_OK(hipMemcpy(dBBox, bboxInit, sizeof(bboxInit), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dMax, &zero, sizeof(zero), hipMemcpyHostToDevice));
    iso_launch_brick_flags(dense, nx, ny, nz, v.nbx, v.nby, v.nbz, dFlag9, v.leaf, dBBox, dMax, nullptr);
    HIP_OK(hipGetLastError());
    std::vector<uint8_t> flag9(nb), leaf(nb);
    int bbox[6];
    unsigned int maxbits = 0;
    HIP_OK(hipMemcpy(flag9.data(), dFlag9, nb, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(leaf.data(), v.leaf, nb, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(bbox, dBBox, sizeof(bbox), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(&maxbits, dMax, sizeof(maxbits), hipMemcpyDeviceToHost));
    guard.freeTemporaries();

    // slot table (exclusive scan of flag9), leaf-level bbox, 128^3 node occupancy
    std::vector<int32_t> slot(nb);
    std::vector<uint8_t> node1(size_t(v.n1x) * v.n1y * v.n1z, 0);
    int lmin[3] = { INT32_MAX, INT32_MAX, INT32_MAX }, lmax[3] = { INT32_MIN, INT32_MIN, INT32_MIN };
    int nslots = 0, nleaf = 0;
    for (int z = 0; z < v.nbz; ++z)
        for (int y = 0; y < v.nby; ++y)
            for (int x = 0; x < v.nbx; ++x) {
                const size_t b = (size_t(z) * v.nby + y) * v.nbx + x;
                slot[b] = flag9[b] ? nslots++ : -1;
                if (leaf[b] && tile) {   // leaves of the halo belong to a neighbour
                    const int c[3] = { v.org[0] + x * 8, v.org[1] + y * 8, v.org[2] + z * 8 };
                    for (int k = 0; k < 3; ++k)
                        if (c[k] < tile->clipLo[k] || c[k] >= tile->clipHi[k]) leaf[b] = 0;
                }
                if (leaf[b]) {
                    ++nleaf;
                    const int gx = v.org[0] + x * 8, gy = v.org[1] + y * 8, gz = v.org[2] + z * 8;
                    node1[(size_t((gz >> 7) - v.n1o[2]) * v.n1y + ((gy >> 7) - v.n1o[1])) * v.n1x + ((gx >> 7) - v.n1o[0])] = 1;
                    if (x < lmin[0]) lmin[0] = x; if (x > lmax[0]) lmax[0] = x;
                    if (y < lmin[1]) lmin[1] = y; if (y > lmax[1]) lmax[1] = y;
                    if (z < lmin[2]) lmin[2] = z; if (z > lmax[2]) lmax[2] = z;
                }
            }
    if (nleaf == 0 && !tile) return false;   // the reference throws on empty grids (IsoVolumeRayTracer.h:188-190); the guard frees
    if (nleaf == 0) {            // an empty tile of a larger volume renders nothing
        for (int k = 0; k < 3; ++k) { lmin[k] = 0; lmax[k] = -1; }
    }
    v.nslots = nslots; v.nleaf = nleaf;
    HIP_OK(hipMalloc(&v.slot, nb * sizeof(int32_t)));
    HIP_OK(hipMalloc(&v.node1, node1.size()));
    HIP_OK(hipMalloc(&v.bricks, size_t(nslots > 0 ? nslots : 1) * ISO_BRICK_STRIDE * sizeof(float)));
    HIP_OK(hipMemcpy(v.slot, slot.data(), nb * sizeof(int32_t), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(v.node1, node1.data(), node1.size(), hipMemcpyHostToDevice));
    if (tile) HIP_OK(hipMemcpy(v.leaf, leaf.data(), nb, hipMemcpyHostToDevice));   // existence AND ownership
    iso_launch_brick_fill(dense, nx, ny, nz, v.nbx, v.nby, v.nbz, v.slot, v.bricks, nullptr);
    HIP_OK(hipMalloc(&v.leafRange, nb * 2 * sizeof(float)));
    iso_launch_leaf_range(dense, nx, ny, nz, v.nbx, v.nby, v.nbz, v.leafRange, nullptr);
    HIP_OK(hipGetLastError());
    HIP_OK(hipDeviceSynchronize());
    guard.handOver();
    return finalizeVolume(v, bbox, orderBitsToFloat(maxbits), lmin, lmax, tile);
}

// Sparse brick list (a .vbx file: 8^3 bricks at arbitrary positions of a box of up to 4096^3) -> the same device
// structures as uploadFromDevice, without ever materialising the box: host work and host memory are proportional to
// the number of bricks; the device holds the bricks plus the three per-position tables of the box (9 bytes per 8^3
// position: 1.2 GB for a box of 4096^3, memset on the device and filled by a scatter of the existing positions).
bool uploadFromBricks(const VbxBricks& vb)
{
    if (vb.bd != 8) return false;
    const int nx = vb.dims[0], ny = vb.dims[1], nz = vb.dims[2];
    if (nx <= 0 || ny <= 0 || nz <= 0 || nx > 4096 || ny > 4096 || nz > 4096) return false;
    Volume v;
    v.nx = nx; v.ny = ny; v.nz = nz;
    v.nbx = (nx + 7) / 8; v.nby = (ny + 7) / 8; v.nbz = (nz + 7) / 8;
    v.n1x = ((nx - 1) >> 7) + 1; v.n1y = ((ny - 1) >> 7) + 1; v.n1z = ((nz - 1) >> 7) + 1;
    const long long nbx = v.nbx, nby = v.nby;
    const size_t nb = size_t(v.nbx) * v.nby * v.nbz;
    auto lin = [&](int bx, int by, int bz) { return (long long)((bz * nby + by) * nbx + bx); };
    // bricks that hold a non-zero voxel, by position; active bbox and maximum over their voxels
    std::unordered_map<long long, const float*> have;
    have.reserve(vb.count() * 2);
    int bbox[6] = { INT32_MAX, INT32_MAX, INT32_MAX, INT32_MIN, INT32_MIN, INT32_MIN };
    float maxv = -3.0e38f;
    for (size_t k = 0; k < vb.count(); ++k) {
        const float* d = vb.data.data() + k * 512;
        const int px = vb.pos[3 * k], py = vb.pos[3 * k + 1], pz = vb.pos[3 * k + 2];
        bool any = false;
        for (int i = 0; i < 512; ++i) {
            const float f = d[i];
            if (f == 0.0f) continue;
            any = true;
            const int x = px + (i & 7), y = py + ((i >> 3) & 7), z = pz + (i >> 6);
            if (x < bbox[0]) bbox[0] = x; if (x > bbox[3]) bbox[3] = x;
            if (y < bbox[1]) bbox[1] = y; if (y > bbox[4]) bbox[4] = y;
            if (z < bbox[2]) bbox[2] = z; if (z > bbox[5]) bbox[5] = z;
            if (f > maxv) maxv = f;
        }
        if (any) have[lin(px >> 3, py >> 3, pz >> 3)] = d;
    }
    // positions whose 9^3 apron can hold a non-zero value: a non-zero brick and its seven lower neighbours
    std::vector<long long> cand;
    cand.reserve(have.size() * 8);
    for (const auto& kv : have) {
        const long long b = kv.first;
        const int bx = int(b % nbx), by = int((b / nbx) % nby), bz = int(b / (nbx * nby));
        for (int o = 0; o < 8; ++o) {
            const int x = bx - (o & 1), y = by - ((o >> 1) & 1), z = bz - (o >> 2);
            if (x >= 0 && y >= 0 && z >= 0) cand.push_back(lin(x, y, z));
        }
    }
    std::sort(cand.begin(), cand.end());
    cand.erase(std::unique(cand.begin(), cand.end()), cand.end());
    std::vector<long long> index;
    std::vector<int32_t> slotv;
    std::vector<uint8_t> leafv;
    std::vector<float> rangev, bricks;
    std::vector<uint8_t> node1(size_t(v.n1x) * v.n1y * v.n1z, 0);
    int lmin[3] = { INT32_MAX, INT32_MAX, INT32_MAX }, lmax[3] = { INT32_MIN, INT32_MIN, INT32_MIN };
    int nslots = 0, nleaf = 0;
    for (const long long b : cand) {
        const int bx = int(b % nbx), by = int((b / nbx) % nby), bz = int(b / (nbx * nby));
        // the 27 bricks around the position
        const float* nbr[3][3][3];
        for (int dz = -1; dz <= 1; ++dz)
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    const int x = bx + dx, y = by + dy, z = bz + dz;
                    const float* q = nullptr;
                    if (x >= 0 && y >= 0 && z >= 0 && x < v.nbx && y < v.nby && z < v.nbz) {
                        auto it = have.find(lin(x, y, z));
                        if (it != have.end()) q = it->second;
                    }
                    nbr[dz + 1][dy + 1][dx + 1] = q;
                }
        // voxel at offset (lx, ly, lz) in [-1, 9] from the position's corner; 0 outside the box / between bricks
        auto at = [&](int lx, int ly, int lz) -> float {
            const float* q = nbr[(lz + 8) >> 3][(ly + 8) >> 3][(lx + 8) >> 3];
            return q ? q[((lz & 7) * 8 + (ly & 7)) * 8 + (lx & 7)] : 0.0f;
        };
        bool any9 = false;
        float tmp[ISO_BRICK_STRIDE];
        for (int k = 0; k < ISO_BRICK_STRIDE; ++k) tmp[k] = 0.0f;
        for (int lz = 0; lz < 9; ++lz)
            for (int ly = 0; ly < 9; ++ly)
                for (int lx = 0; lx < 9; ++lx) {
                    const float f = at(lx, ly, lz);
                    tmp[(lz * 9 + ly) * 9 + lx] = f;
                    any9 = any9 || f != 0.0f;
                }
        const bool leaf = nbr[1][1][1] != nullptr;
        if (!any9 && !leaf) continue;
        float lo = 3.0e38f, hi = -3.0e38f;
        if (leaf) {   // iso_leaf_range: every value a march through the leaf can read, [8b - 1, 8b + 9]^3
            for (int lz = -1; lz < 10; ++lz)
                for (int ly = -1; ly < 10; ++ly)
                    for (int lx = -1; lx < 10; ++lx) {
                        const float f = at(lx, ly, lz);
                        lo = f < lo ? f : lo; hi = f > hi ? f : hi;
                    }
            ++nleaf;
            node1[(size_t(bz >> 4) * v.n1y + (by >> 4)) * v.n1x + (bx >> 4)] = 1;
            if (bx < lmin[0]) lmin[0] = bx; if (bx > lmax[0]) lmax[0] = bx;
            if (by < lmin[1]) lmin[1] = by; if (by > lmax[1]) lmax[1] = by;
            if (bz < lmin[2]) lmin[2] = bz; if (bz > lmax[2]) lmax[2] = bz;
        } else {
            lo = hi = 0.0f;
        }
        index.push_back(b);
        leafv.push_back(leaf ? 1 : 0);
        rangev.push_back(lo); rangev.push_back(hi);
        if (any9) {
            slotv.push_back(nslots++);
            bricks.insert(bricks.end(), tmp, tmp + ISO_BRICK_STRIDE);
        } else {
            slotv.push_back(-1);
        }
    }
    if (nleaf == 0) return false;
    v.nslots = nslots; v.nleaf = nleaf;
    long long* dIndex = nullptr; int32_t* dSlot = nullptr; uint8_t* dLeaf = nullptr; float* dRange = nullptr;
    UploadGuard guard(v);
    guard.watch(0, (void**)&dIndex); guard.watch(1, (void**)&dSlot); guard.watch(2, (void**)&dLeaf); guard.watch(3, (void**)&dRange);
    HIP_OK(hipMalloc(&v.slot, nb * sizeof(int32_t)));
    HIP_OK(hipMalloc(&v.leaf, nb));
    HIP_OK(hipMalloc(&v.leafRange, nb * 2 * sizeof(float)));
    HIP_OK(hipMalloc(&v.node1, node1.size()));
    HIP_OK(hipMalloc(&v.bricks, size_t(nslots > 0 ? nslots : 1) * ISO_BRICK_STRIDE * sizeof(float)));
    HIP_OK(hipMemset(v.slot, 0xFF, nb * sizeof(int32_t)));              // -1: nothing stored
    HIP_OK(hipMemset(v.leaf, 0, nb));
    HIP_OK(hipMemset(v.leafRange, 0, nb * 2 * sizeof(float)));
    HIP_OK(hipMemcpy(v.node1, node1.data(), node1.size(), hipMemcpyHostToDevice));
    if (nslots > 0) HIP_OK(hipMemcpy(v.bricks, bricks.data(), bricks.size() * sizeof(float), hipMemcpyHostToDevice));
    const int n = int(index.size());
    HIP_OK(hipMalloc(&dIndex, size_t(n) * sizeof(long long)));
    HIP_OK(hipMalloc(&dSlot, size_t(n) * sizeof(int32_t)));
    HIP_OK(hipMalloc(&dLeaf, size_t(n)));
    HIP_OK(hipMalloc(&dRange, size_t(n) * 2 * sizeof(float)));
    HIP_OK(hipMemcpy(dIndex, index.data(), size_t(n) * sizeof(long long), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dSlot, slotv.data(), size_t(n) * sizeof(int32_t), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dLeaf, leafv.data(), size_t(n), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dRange, rangev.data(), size_t(n) * 2 * sizeof(float), hipMemcpyHostToDevice));
    iso_launch_scatter_tables(n, dIndex, dSlot, dLeaf, dRange, v.slot, v.leaf, v.leafRange, nullptr);
    HIP_OK(hipGetLastError());
    HIP_OK(hipDeviceSynchronize());
    guard.handOver();
    return finalizeVolume(v, bbox, maxv, lmin, lmax, nullptr);
}

// ---- semantics=gvdb: the CUDA renderer's camera / transform, as constants of one frame --------------------------
// gvdb_camera.cpp:431-441 (gluLookAt basis)
void gvdbBasis(const double origin[3], const double lookat[3], const double up[3], double side[3], double upv[3], double back[3])
{
    double d[3] = { lookat[0] - origin[0], lookat[1] - origin[1], lookat[2] - origin[2] };
    double l = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    for (int k = 0; k < 3; ++k) d[k] /= l;
    side[0] = d[1] * up[2] - d[2] * up[1]; side[1] = d[2] * up[0] - d[0] * up[2]; side[2] = d[0] * up[1] - d[1] * up[0];
    l = std::sqrt(side[0] * side[0] + side[1] * side[1] + side[2] * side[2]);
    for (int k = 0; k < 3; ++k) side[k] /= l;
    upv[0] = side[1] * d[2] - side[2] * d[1]; upv[1] = side[2] * d[0] - side[0] * d[2]; upv[2] = side[0] * d[1] - side[1] * d[0];
    l = std::sqrt(upv[0] * upv[0] + upv[1] * upv[1] + upv[2] * upv[2]);
    for (int k = 0; k < 3; ++k) { upv[k] /= l; back[k] = -d[k]; }
}

// proj * view, row-major; P00 = 2 near / (tan(fov/2) near), near .1, far 5000 (gvdb_camera.cpp:59-60,447-455)
void gvdbViewProj(const double origin[3], const double lookat[3], const double up[3], double fovDeg, double aspect, float out[16])
{
    double s[3], u[3], b[3];
    gvdbBasis(origin, lookat, up, s, u, b);
    const double nr = 0.1, fr = 5000.0;
    const double sx = std::tan(fovDeg * (M_PI / 180.0) / 2.0) * nr, sy = sx / aspect;
    const double P[4][4] = { { 2.0 * nr / sx, 0, 0, 0 }, { 0, 2.0 * nr / sy, 0, 0 },
                             { 0, 0, -(fr + nr) / (fr - nr), -(2.0 * fr * nr) / (fr - nr) }, { 0, 0, -1.0, 0 } };
    const double V[4][4] = { { s[0], s[1], s[2], -(s[0] * origin[0] + s[1] * origin[1] + s[2] * origin[2]) },
                             { u[0], u[1], u[2], -(u[0] * origin[0] + u[1] * origin[1] + u[2] * origin[2]) },
                             { b[0], b[1], b[2], -(b[0] * origin[0] + b[1] * origin[1] + b[2] * origin[2]) },
                             { 0, 0, 0, 1 } };
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double a = 0;
            for (int k = 0; k < 4; ++k) a += P[i][k] * V[k][j];
            out[4 * i + j] = float(a);
        }
}

void buildGvdbFrame(IsoGvdbFrame& f, const Args& a, const Volume& v, const double lastOrigin[3], const double lastLookAt[3])
{
    // loadGrid: SetTransform(-centre, 0.5 / longest edge) of the object bounds (GPURendererDirect.cpp:266-278)
    double ext = 0, cen[3];
    for (int k = 0; k < 3; ++k) {
        const double lo = v.bbmin[k], hi = v.bbmax[k];
        if (hi - lo > ext) ext = hi - lo;
        cen[k] = (lo + hi) * 0.5;
    }
    const double scale = 0.5 / ext;
    f.scale = float(scale);
    for (int k = 0; k < 3; ++k) {
        f.tr[k] = float(-cen[k] * scale);
        f.rpos[k] = float(a.cameraOrigin[k] / scale + cen[k]);       // campos * invxform
    }
    const double aspect = double(a.resolutionX) / double(a.resolutionY);
    gvdbViewProj(a.cameraOrigin, a.cameraLookAt, a.cameraUp, a.cameraFov, aspect, f.cur);
    gvdbViewProj(lastOrigin, lastLookAt, a.cameraUp, a.cameraFov, aspect, f.nxt);
    double s[3], u[3], b[3];
    gvdbBasis(a.cameraOrigin, a.cameraLookAt, a.cameraUp, s, u, b);
    for (int k = 0; k < 3; ++k) { f.vrot[k] = float(s[k]); f.vrot[3 + k] = float(u[k]); f.vrot[6 + k] = float(b[k]); }
    // corner rays tl / tr / bl (gvdb_camera.cpp:598-601,654-665): view-space directions (x / P00, y / P11, -1)
    const double hx = std::tan(a.cameraFov * (M_PI / 180.0) / 2.0) / 2.0, hy = hx / aspect;
    for (int k = 0; k < 3; ++k) {
        const double tl = -hx * s[k] + hy * u[k] - b[k];
        const double tr = hx * s[k] + hy * u[k] - b[k];
        const double bl = -hx * s[k] - hy * u[k] - b[k];
        f.cams[k] = float(tl); f.camu[k] = float(tr - tl); f.camv[k] = float(bl - tl);
    }
    double L[3];
    for (int k = 0; k < 3; ++k) L[k] = a.cameraLight ? (a.cameraLookAt[k] - a.cameraOrigin[k]) : a.lightDirection[k];
    const double ll = std::sqrt(L[0] * L[0] + L[1] * L[1] + L[2] * L[2]);
    for (int k = 0; k < 3; ++k) {
        f.light[k] = float(L[k] / ll);
        f.ambient[k] = float(a.materialAmbient[k]); f.diffuse[k] = float(a.materialDiffuse[k]); f.specular[k] = float(a.materialSpecular[k]);
    }
    f.iso = float(a.isovalue);                                       // absolute (GPURendererDirect.cpp:364)
    f.exponent = a.materialSpecularExponent;
    f.spec_c = float(a.materialSpecularExponent + 2) / (2.0f * 3.41f);   // render_kernel.cu:236
    f.aoRadius = a.aoRadius;
}

// The traversal reads one byte per leaf / node: "exists" and "its value range holds the isovalue".  The table belongs to
// one isovalue; a change (rare: the user moves the slider) waits for every render in flight, refills it and waits again,
// so that renders on any stream see a finished table.
bool updateMarchFlags(Volume& v, double iso, hipStream_t stream)
{
    if (v.marchValid && v.marchIso == iso) return true;
    const size_t nb = size_t(v.nbx) * v.nby * v.nbz, n1 = size_t(v.n1x) * v.n1y * v.n1z;
    if (!v.marchFlags) HIP_OK(hipMalloc(&v.marchFlags, nb + n1));
    else HIP_OK(hipDeviceSynchronize());
    iso_launch_march_flags(v.leaf, v.leafRange, int(nb), iso, v.marchFlags, stream);
    iso_launch_march_flags(v.node1, v.node1Range, int(n1), iso, v.marchFlags + nb, stream);
    HIP_OK(hipGetLastError());
    HIP_OK(hipStreamSynchronize(stream));
    v.marchIso = iso;
    v.marchValid = true;
    return true;
}

// Everything a kernel needs to know about the volume, the camera and the shading, from the current Args (launchFrame, AO passes)
bool buildParams(IsoRenderParams& p, float* out, hipStream_t stream)
{
    const Args& a = g.args;
    const Volume& v = g.vol;
    if (!g.initialised || !v.loaded || !out || a.resolutionX <= 0 || a.resolutionY <= 0) return false;
    std::memset(&p, 0, sizeof(p));
    IsoCamera last;
    buildCamera(p.cam, a.cameraOrigin, a.cameraLookAt, a.cameraUp, a.cameraFov, a.resolutionX, a.resolutionY);
    buildCamera(last, g.lastOrigin, g.lastLookAt, a.cameraUp, a.cameraFov, a.resolutionX, a.resolutionY);
    std::memcpy(p.Vlast, last.V, sizeof(p.Vlast));
    p.s = v.s; p.sinv = v.sinv;
    for (int k = 0; k < 3; ++k) p.t[k] = v.t[k];
    // CPURenderer.cpp:501-503,515-516: relative isovalue, narrowed through float
    p.iso = double(float(a.isovalue * double(v.maxValue)));
    double light[3];
    for (int k = 0; k < 3; ++k) light[k] = a.cameraLight ? (a.cameraLookAt[k] - a.cameraOrigin[k]) : a.lightDirection[k];
    normalize3(light);
    for (int k = 0; k < 3; ++k) {
        p.light[k] = light[k];
        p.ambient[k] = a.materialAmbient[k];
        p.diffuse[k] = a.materialDiffuse[k];
        p.specular[k] = a.materialSpecular[k];
        p.bbmin[k] = v.bbmin[k]; p.bbmax[k] = v.bbmax[k];
    }
    p.exponent = a.materialSpecularExponent;
    p.spec_c1 = (a.materialSpecularExponent + 2) / (2 * M_PI);
    p.W = a.resolutionX; p.H = a.resolutionY;
    for (int k = 0; k < 4; ++k) p.vp[k] = a.viewport[k];
    p.nx = v.nx; p.ny = v.ny; p.nz = v.nz;
    p.nbx = v.nbx; p.nby = v.nby; p.nbz = v.nbz;
    p.n1x = v.n1x; p.n1y = v.n1y; p.n1z = v.n1z;
    for (int k = 0; k < 3; ++k) { p.org[k] = v.org[k]; p.n1o[k] = v.n1o[k]; }
    p.any_leaf = v.nleaf > 0;
    p.bricks = v.bricks; p.slot = v.slot; p.leaf = v.leaf; p.leafRange = v.leafRange; p.node1 = v.node1; p.node1Range = v.node1Range;
    if (g.semantics != 1) {
        if (!updateMarchFlags(g.vol, p.iso, stream)) return false;
        p.leafMarch = v.marchFlags;
        p.node1March = v.marchFlags + size_t(v.nbx) * v.nby * v.nbz;
    }
    p.out = out;
    p.aoSamples = a.aoSamples < 0 ? 0 : (a.aoSamples > 512 ? 512 : a.aoSamples);   // GPURendererDirect.cpp:350
    p.aoRadius = double(a.aoRadius);
    p.aoHemi = g.aoHemi; p.aoRot = g.aoRot;
    p.tileQueue = g.tileQueue;
    p.resident = g.tileQueue + 8;
    p.hitState = g.hitState;
    return true;
}

bool launchFrame(float* out, hipStream_t stream)
{
    IsoRenderParams p;
    if (!buildParams(p, out, stream)) return false;
    const Args& a = g.args;
    const Volume& v = g.vol;
    const int tiles_all = ((p.W + 7) >> 3) * ((p.H + 7) >> 3);
    const bool ordering = g.orderMode > 0 && g.variant == 0 && g.semantics != 1 && !g.statsOut && tiles_all <= ISO_ORDER_MAX_TILES;
    if (ordering) {
        if (!g.tileCost) {
            HIP_OK(hipMalloc(&g.tileCost, ISO_ORDER_MAX_TILES * sizeof(unsigned)));
            HIP_OK(hipMalloc(&g.tileOrder, ISO_ORDER_MAX_TILES * sizeof(unsigned short)));
        }
        p.tileCost = g.tileCost;
        p.tileOrder = (g.orderW == p.W && g.orderH == p.H) ? g.tileOrder : nullptr;
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (g.profile) {
        if (hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) g.events.emplace_back(e0, e1);
        else e0 = e1 = nullptr;
    }
    if (g.semantics == 1) {
        // (tiles too, round 4: the brick DDA runs over the GLOBAL box, a tile marches the bricks it owns; ray-cast AO inside this
        // launch would see the tile's own bricks only, as in the default semantics: render tiles with aosamples = 0)
        IsoGvdbFrame f;
        buildGvdbFrame(f, a, v, g.lastOrigin, g.lastLookAt);
        iso_launch_render_gvdb(p, f, stream, e0, e1);
    } else if (g.statsOut) {
        iso_launch_render_stats(p, g.variant, g.statsOut, stream);
    } else {
        iso_launch_render(p, g.variant, stream, e0, e1, g.waveCap);
        if (ordering) {
            int dev = 0, cus = 256;
            (void)hipGetDevice(&dev);
            if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
            iso_launch_tile_order(g.tileCost, g.tileOrder, tiles_all, g.orderMode, 4 * cus, stream);
            g.orderW = p.W; g.orderH = p.H;
        }
        if (g.variant == 2) {
            const int tiles = ((p.W + 7) >> 3) * ((p.H + 7) >> 3);
            g.residentTarget += unsigned(g.waveCap > 0 && g.waveCap < tiles ? (g.waveCap + 7) & ~7 : (tiles + 7) & ~7);
        }
    }
    if (hipGetLastError() != hipSuccess) return false;
    // GPURendererDirect.cpp:440-442: the camera just rendered becomes the flow reference
    for (int k = 0; k < 3; ++k) { g.lastOrigin[k] = a.cameraOrigin[k]; g.lastLookAt[k] = a.cameraLookAt[k]; }
    return true;
}

bool endsWith(const std::string& s, const char* suffix)
{
    const size_t n = std::strlen(suffix);
    return s.size() >= n && s.compare(s.size() - n, n, suffix) == 0;
}

}  // namespace

extern "C" {

int isoSetHitStateBuffer(unsigned long long devicePtr)
{
    g.hitState = reinterpret_cast<double*>(devicePtr);
    return 0;
}

int isoAoDistancesAsync(unsigned long long hitStatePtr, unsigned long long gbufPtr, unsigned long long distPtr, void* stream)
{
    if (!g.initialised || !g.vol.loaded || !hitStatePtr || !gbufPtr || !distPtr || g.semantics == 1) return -1;
    IsoRenderParams p;
    if (!buildParams(p, reinterpret_cast<float*>(gbufPtr), static_cast<hipStream_t>(stream))) return -1;
    if (p.aoSamples <= 0) return -1;
    p.hitState = nullptr;
    iso_launch_ao_distances(p, reinterpret_cast<const double*>(hitStatePtr), reinterpret_cast<const float*>(gbufPtr),
                            reinterpret_cast<double*>(distPtr), stream);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int isoAoFinishAsync(unsigned long long distPtr, unsigned long long gbufPtr, void* stream)
{
    if (!g.initialised || !g.vol.loaded || !gbufPtr || !distPtr) return -1;
    IsoRenderParams p;
    if (!buildParams(p, reinterpret_cast<float*>(gbufPtr), static_cast<hipStream_t>(stream))) return -1;
    if (p.aoSamples <= 0) return -1;
    iso_launch_ao_finish(p, reinterpret_cast<const double*>(distPtr), reinterpret_cast<float*>(gbufPtr), stream);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int isoSetTileOrderMode(int mode)
{
    if (mode < 0 || mode > 2) return -1;
    g.orderMode = mode;
    g.orderW = g.orderH = 0;
    return 0;
}

int initGVDB(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        std::fprintf(stderr, "GPURendererDirect: no HIP device available\n");
        return -1;
    }
    if (hipFree(nullptr) != hipSuccess) return -1;   // create the context on the current device
    if (!g.initialised) {
        g.args = Args();
        if (!uploadAoTables()) return -1;
        g.initialised = true;
    }
    return 0;
}

int loadGrid(const char* filename)
{
    if (!filename) return -1;
    const std::string name(filename);
    if (!endsWith(name, ".vbx")) {
        std::printf("Error: Input must end in .vbx\n");
        return -1;
    }
    if (!g.initialised) return -2;
    try {
        VbxBricks vb;
        std::string err;
        if (!vbx_read_bricks(name.c_str(), vb, err)) {
            std::printf("Unable to load VBX file: %s\n", err.c_str());
            return -2;
        }
        if (vb.bd == 8) return uploadFromBricks(vb) ? 0 : -2;      // the reference's <5,5,5,4,3> trees: bricks as they are
        // other brick sizes are re-bricked from a dense copy (refused beyond 2^31 voxels)
        std::vector<float> dense;
        int nx = 0, ny = 0, nz = 0;
        vb = VbxBricks();
        if (!vbx_read_dense(name.c_str(), dense, nx, ny, nz, err)) {
            std::printf("Unable to load VBX file: %s\n", err.c_str());
            return -2;
        }
        return isoLoadDenseHost(dense.data(), nx, ny, nz);
    } catch (...) {      // nothing may unwind through the C boundary
        std::printf("Unable to load VBX file: out of memory\n");
        return -2;
    }
}

int setParameter(const char* cmd_, const char* value)
{
    if (!cmd_ || !value) return -1;
    const std::string cmd(cmd_);
    Args& a = g.args;
    double d[4];
    if (cmd == "fov" || cmd == "cameraFoV") {
        if (!splitNumbers(value, 1, d)) return -1;
        a.cameraFov = d[0];
    } else if (cmd == "cameraOrigin") {
        if (!splitNumbers(value, 3, d)) return -1;
        for (int k = 0; k < 3; ++k) a.cameraOrigin[k] = d[k];
    } else if (cmd == "cameraLookAt") {
        if (!splitNumbers(value, 3, d)) return -1;
        for (int k = 0; k < 3; ++k) a.cameraLookAt[k] = d[k];
    } else if (cmd == "cameraUp") {
        if (!splitNumbers(value, 3, d)) return -1;
        for (int k = 0; k < 3; ++k) a.cameraUp[k] = d[k];
    } else if (cmd == "resolution") {
        if (!splitNumbers(value, 2, d)) return -1;
        a.resolutionX = int(d[0]); a.resolutionY = int(d[1]);
    } else if (cmd == "isovalue") {
        if (!splitNumbers(value, 1, d)) return -1;
        a.isovalue = d[0];
    } else if (cmd == "unshaded") {
        if (!splitNumbers(value, 1, d)) return -1;
        a.noShading = int(d[0]);
    } else if (cmd == "aosamples") {
        if (!splitNumbers(value, 1, d)) return -1;
        a.aoSamples = int(d[0]);
    } else if (cmd == "aoradius") {
        if (!splitNumbers(value, 1, d)) return -1;
        a.aoRadius = float(d[0]);
    } else if (cmd == "viewport") {
        if (!splitNumbers(value, 4, d)) return -1;
        for (int k = 0; k < 4; ++k) a.viewport[k] = int(d[k]);
    } else if (cmd == "ambient" || cmd == "diffuse" || cmd == "specular") {   // additive
        if (!splitNumbers(value, 3, d)) return -1;
        double* dst = cmd == "ambient" ? a.materialAmbient : (cmd == "diffuse" ? a.materialDiffuse : a.materialSpecular);
        for (int k = 0; k < 3; ++k) dst[k] = d[k];
    } else if (cmd == "exponent") {                                            // additive
        if (!splitNumbers(value, 1, d)) return -1;
        a.materialSpecularExponent = int(d[0]);
    } else if (cmd == "light") {                                               // additive
        if (std::strcmp(value, "camera") == 0) a.cameraLight = true;
        else {
            if (!splitNumbers(value, 3, d)) return -1;
            a.cameraLight = false;
            for (int k = 0; k < 3; ++k) a.lightDirection[k] = d[k];
        }
    } else if (cmd == "semantics") {                                           // additive, see include/gpu_renderer_direct.h
        if (std::strcmp(value, "cpu") == 0) g.semantics = 0;
        else if (std::strcmp(value, "gvdb") == 0) g.semantics = 1;
        else return -1;
    } else {
        std::printf("Unknown command: '%s', exit\n", cmd_);
        return -1;
    }
    return 0;
}

float render(unsigned long long devicePtr)
{
    if (hipDeviceSynchronize() != hipSuccess) return -1.f;
    const auto start = std::chrono::high_resolution_clock::now();
    if (!launchFrame(reinterpret_cast<float*>(devicePtr), nullptr)) return -1.f;
    if (hipDeviceSynchronize() != hipSuccess) {
        std::fprintf(stderr, "GPURendererDirect: kernel failed: %s\n", hipGetErrorString(hipGetLastError()));
        return -1.f;
    }
    const auto finish = std::chrono::high_resolution_clock::now();
    return float(std::chrono::duration<double>(finish - start).count());
}

int isoRenderAsync(unsigned long long devicePtr, void* stream)
{
    return launchFrame(reinterpret_cast<float*>(devicePtr), static_cast<hipStream_t>(stream)) ? 0 : -1;
}

int isoFrameBlockBytes(void) { return (int)sizeof(IsoFrameBlock); }

int isoWriteFrameBlockAsync(unsigned long long deviceBlock, void* stream)
{
    if (!deviceBlock || (deviceBlock & 7)) return -1;
    IsoRenderParams p;
    float dummy = 0.0f;       // buildParams wants an output pointer; nothing is launched with it
    if (g.semantics == 1 || !buildParams(p, &dummy, static_cast<hipStream_t>(stream))) return -1;
    IsoFrameBlock b;
    b.cam = p.cam;
    std::memcpy(b.Vlast, p.Vlast, sizeof(b.Vlast));
    for (int k = 0; k < 3; ++k) b.light[k] = p.light[k];
    iso_launch_write_block(b, reinterpret_cast<IsoFrameBlock*>(deviceBlock), stream);
    if (hipGetLastError() != hipSuccess) return -1;
    // as after a render: the camera just described becomes the flow reference of the next frame (GPURendererDirect.cpp:440-442)
    for (int k = 0; k < 3; ++k) { g.lastOrigin[k] = g.args.cameraOrigin[k]; g.lastLookAt[k] = g.args.cameraLookAt[k]; }
    return 0;
}

int isoRenderFromBlockAsync(unsigned long long devicePtr, unsigned long long deviceBlock, void* stream)
{
    if (!deviceBlock || (deviceBlock & 7)) return -1;
    IsoRenderParams p;
    if (g.semantics == 1 || g.args.aoSamples > 0 || g.statsOut) return -1;     // the plain SR-mode render only
    if (!buildParams(p, reinterpret_cast<float*>(devicePtr), static_cast<hipStream_t>(stream))) return -1;
    p.tileCost = nullptr; p.tileOrder = nullptr;
    iso_launch_render_from_block(p, reinterpret_cast<const IsoFrameBlock*>(deviceBlock), stream);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int isoLoadDenseDevice(unsigned long long devicePtr, int nx, int ny, int nz)
{
    if (!g.initialised || !devicePtr) return -2;
    return uploadFromDevice(reinterpret_cast<const float*>(devicePtr), nx, ny, nz) ? 0 : -2;
}

int isoLoadDenseHost(const float* hostData, int nx, int ny, int nz)
{
    if (!g.initialised || !hostData || nx <= 0 || ny <= 0 || nz <= 0) return -2;
    float* dense = nullptr;
    const size_t bytes = size_t(nx) * ny * nz * sizeof(float);
    if (hipMalloc(&dense, bytes) != hipSuccess) return -2;
    if (hipMemcpy(dense, hostData, bytes, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(dense); return -2; }
    const bool ok = uploadFromDevice(dense, nx, ny, nz);
    (void)hipFree(dense);
    return ok ? 0 : -2;
}

static bool fillTileInfo(TileInfo& t, const int origin[3], const int globalActiveMin[3], const int globalActiveMax[3], float globalMax,
                         const int clipLo[3], const int clipHi[3])
{
    if (!origin || !globalActiveMin || !globalActiveMax || !clipLo || !clipHi) return false;
    for (int k = 0; k < 3; ++k) {
        t.origin[k] = origin[k]; t.gmin[k] = globalActiveMin[k]; t.gmax[k] = globalActiveMax[k];
        t.clipLo[k] = clipLo[k]; t.clipHi[k] = clipHi[k];
    }
    t.globalMax = globalMax;
    return true;
}

int isoLoadDenseTileDevice(unsigned long long devicePtr, int nx, int ny, int nz, const int origin[3],
                           const int globalActiveMin[3], const int globalActiveMax[3], float globalMax,
                           const int clipLo[3], const int clipHi[3])
{
    TileInfo t;
    if (!g.initialised || !devicePtr || nx <= 0 || ny <= 0 || nz <= 0 || !fillTileInfo(t, origin, globalActiveMin, globalActiveMax, globalMax, clipLo, clipHi)) return -2;
    return uploadFromDevice(reinterpret_cast<const float*>(devicePtr), nx, ny, nz, &t) ? 0 : -2;
}

int isoLoadDenseTileHost(const float* hostData, int nx, int ny, int nz, const int origin[3],
                         const int globalActiveMin[3], const int globalActiveMax[3], float globalMax,
                         const int clipLo[3], const int clipHi[3])
{
    TileInfo t;
    if (!g.initialised || !hostData || nx <= 0 || ny <= 0 || nz <= 0 || !fillTileInfo(t, origin, globalActiveMin, globalActiveMax, globalMax, clipLo, clipHi)) return -2;
    float* dense = nullptr;
    const size_t bytes = size_t(nx) * ny * nz * sizeof(float);
    if (hipMalloc(&dense, bytes) != hipSuccess) return -2;
    if (hipMemcpy(dense, hostData, bytes, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(dense); return -2; }
    const bool ok = uploadFromDevice(dense, nx, ny, nz, &t);
    (void)hipFree(dense);
    return ok ? 0 : -2;
}

int isoGetVolumeInfo(int info[12], float* out_max)
{
    const Volume& v = g.vol;
    if (!v.loaded) return -1;
    info[0] = v.nx; info[1] = v.ny; info[2] = v.nz;
    info[3] = v.nslots; info[4] = v.nleaf;
    for (int k = 0; k < 3; ++k) { info[5 + k] = v.bbmin[k]; info[8 + k] = v.bbmax[k]; }
    info[11] = int((size_t(v.nslots) * ISO_BRICK_STRIDE * sizeof(float)) >> 20);
    if (out_max) *out_max = v.maxValue;
    return 0;
}

int isoSetWaveCap(int waves)
{
    if (waves < 0) return -1;
    g.waveCap = waves;
    return 0;
}

int isoGateResident(void* stream, int timeoutUs)
{
    if (!g.initialised || !g.tileQueue || timeoutUs < 0) return -1;
    // no variant-2 render was enqueued since the last gate: there is nothing to wait for (a gate that spun until its
    // timeout every frame would stall the caller's stream, e.g. after a failed prefetch)
    if (g.residentTarget == g.gatedTarget) return 0;
    g.gatedTarget = g.residentTarget;
    iso_launch_gate(g.tileQueue + 8, g.residentTarget, timeoutUs, stream);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int isoSetKernelVariant(int variant)
{
    if (variant < 0 || variant > 5) return -1;
    g.variant = variant;
    return 0;
}

// Diagnostics, not part of the public header: while a device buffer of 6 * tiles int64 is set, frames are rendered by the
// instrumented variant-0 kernel, which also writes per-tile clocks and per-ray step counts (tools/lab/raymarch_stats.py).
void isoDebugSetStatsBuffer(unsigned long long devicePtr) { g.statsOut = reinterpret_cast<long long*>(devicePtr); }

int isoProfileEnable(int on)
{
    for (auto& e : g.events) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    g.events.clear();
    g.profile = on != 0;
    return 0;
}

int isoProfileCount(void) { return (int)g.events.size(); }

int isoProfileGet(int i, float* ms)
{
    if (i < 0 || i >= (int)g.events.size() || !ms) return -1;
    return hipEventElapsedTime(ms, g.events[i].first, g.events[i].second) == hipSuccess ? 0 : -2;
}

int isoSetLastCamera(const double origin[3], const double lookAt[3])
{
    if (!origin || !lookAt) return -1;
    for (int k = 0; k < 3; ++k) { g.lastOrigin[k] = origin[k]; g.lastLookAt[k] = lookAt[k]; }
    return 0;
}

int isoVbxInfo(const char* path, int dims[3])
{
    if (!path || !dims) return -2;
    try {
        VbxBricks vb;
        std::string err;
        if (!vbx_read_bricks(path, vb, err)) return -2;
        for (int k = 0; k < 3; ++k) dims[k] = vb.dims[k];
        return 0;
    } catch (...) {
        return -2;
    }
}

int isoVbxReadDense(const char* path, float* hostOut)
{
    if (!path || !hostOut) return -2;
    try {
        std::vector<float> dense;
        std::string err;
        int nx, ny, nz;
        if (!vbx_read_dense(path, dense, nx, ny, nz, err)) return -2;
        std::memcpy(hostOut, dense.data(), dense.size() * sizeof(float));
        return 0;
    } catch (...) {
        return -2;
    }
}

void isoShutdown(void)
{
    freeVolume(g.vol);
    if (g.aoHemi) (void)hipFree(g.aoHemi);
    if (g.aoRot) (void)hipFree(g.aoRot);
    if (g.tileQueue) (void)hipFree(g.tileQueue);
    g.tileQueue = nullptr;
    if (g.tileCost) (void)hipFree(g.tileCost);
    if (g.tileOrder) (void)hipFree(g.tileOrder);
    g.tileCost = nullptr; g.tileOrder = nullptr; g.orderW = g.orderH = 0;
    g.aoHemi = g.aoRot = nullptr;
    g.initialised = false;
}

}  // extern "C"
