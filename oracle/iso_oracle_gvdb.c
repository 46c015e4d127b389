/*
 * iso_oracle_gvdb.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the ARITHMETIC of the reference's CUDA renderer (the "GPURendererDirect column" of
 * SURVEY.md 8(a.3)): cell-centred trilinear sampling, fixed 0.05-voxel march inside occupied 8^3 bricks with
 * 10 bisection steps, absolute isovalue, longest edge -> 0.5 world units, GVDB camera (image half-width
 * tangent tan(fov/2)/2), NDC depth, outward view-space normals, 0.5 * delta-NDC flow, ray-cast AO, shadow 1.
 *
 * PARITY UNPINNED, and necessarily so: the reference samples through the texture unit (9-bit filter
 * weights) under --use_fast_math, which no software can reproduce bit for bit.  What is
 * restated here, in IEEE float, is the algorithm; deviations that follow from not building GVDB's 5-level
 * tree are listed where they occur.  The product's `semantics=gvdb` kernel must match THIS file bit for
 * bit on the hit mask and to 1e-4 elsewhere.
 *
 * Followed: GPURendererDirect/render_kernel.cu:109-266 (kernel, AO, brick march),
 * GPURendererDirect/GPURendererDirect.cpp:266-281,319-365 (transform, matrices, iso/step),
 * third-party/include/gvdb/cuda_gvdb_raycast.cuh:132-141,255-277,504-575 (gradient, plain brick march, rayCast),
 * cuda_gvdb_dda.cuh:38-60 (DDA macros), cuda_gvdb_geom.cuh:61-74,96-109 (view ray, ray/box),
 * gvdb_camera.cpp:425-489,547-602,654-665 (matrices), gvdb_volume_gvdb.cpp:6210-6230 (SetTransform).
 */
#include "iso_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "iso_oracle_priv.h"

#define GV_NOHIT 1.0e10f          /* cuda_gvdb_scene.cuh:33 */
#define GV_MAX_ITER 256           /* cuda_gvdb_raycast.cuh:37 */
#define GV_PSTEP 0.05f            /* GPURendererDirect.cpp:365 */
#define GV_EPS 0.001f             /* gvdb_volume_gvdb.cpp:116 */

typedef struct { float x, y, z; } f3;

static inline f3 f3_make(float x, float y, float z) { f3 r = { x, y, z }; return r; }
static inline f3 f3_add(f3 a, f3 b) { return f3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline f3 f3_sub(f3 a, f3 b) { return f3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline f3 f3_scale(f3 a, float s) { return f3_make(a.x * s, a.y * s, a.z * s); }
static inline float f3_dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline f3 f3_cross(f3 a, f3 b) { return f3_make(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
static inline f3 f3_normalize(f3 a) { float l = sqrtf(f3_dot(a, a)); return f3_make(a.x / l, a.y / l, a.z / l); }
static inline f3 f3_safe_normalize(f3 a)            /* render_kernel.cu:150-155 */
{
    float l = sqrtf(f3_dot(a, a));
    if (!(l > 1e-6f)) return f3_make(0.f, 0.f, 0.f);
    return f3_make(a.x / l, a.y / l, a.z / l);
}

/* everything the per-pixel code needs, prepared once per frame in double and narrowed to float */
typedef struct {
    f3 rpos;                    /* camera position in grid-local (voxel) coordinates: campos * invxform */
    f3 cams, camu, camv;        /* corner-ray basis, cuda_gvdb_geom.cuh:66-74 */
    float cur[4][4], nxt[4][4]; /* proj * view of the current / "next" camera, row-major */
    float vrot[3][3];           /* rotation rows of the view matrix (side, up, -dir) */
    float scale, tr[3];         /* grid-local -> world: w = scale * p + tr   (SetTransform: S * PT) */
    f3 bmin, bmax;              /* object bounds in voxels (brick aligned) */
    f3 light;                   /* normalised */
    float iso;
} gv_frame;

static void look_basis(const double origin[3], const double lookat[3], const double up[3], double side[3], double upv[3], double back[3])
{
    double d[3] = { lookat[0] - origin[0], lookat[1] - origin[1], lookat[2] - origin[2] };   /* gvdb_camera.cpp:431-441 */
    double l = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    for (int k = 0; k < 3; ++k) d[k] /= l;
    side[0] = d[1] * up[2] - d[2] * up[1]; side[1] = d[2] * up[0] - d[0] * up[2]; side[2] = d[0] * up[1] - d[1] * up[0];
    l = sqrt(side[0] * side[0] + side[1] * side[1] + side[2] * side[2]);
    for (int k = 0; k < 3; ++k) side[k] /= l;
    upv[0] = side[1] * d[2] - side[2] * d[1]; upv[1] = side[2] * d[0] - side[0] * d[2]; upv[2] = side[0] * d[1] - side[1] * d[0];
    l = sqrt(upv[0] * upv[0] + upv[1] * upv[1] + upv[2] * upv[2]);
    for (int k = 0; k < 3; ++k) { upv[k] /= l; back[k] = -d[k]; }
}

/* proj * view, row-major; gluPerspective-like with P00 = 2 near / (tan(fov/2) near) (gvdb_camera.cpp:447-455) */
static void view_proj(const double origin[3], const double lookat[3], const double up[3], double fov_deg, double aspect, float out[4][4])
{
    double s[3], u[3], b[3];
    look_basis(origin, lookat, up, s, u, b);
    const double nr = 0.1, fr = 5000.0;                 /* gvdb_camera.cpp:59-60 */
    double sx = tan(fov_deg * (M_PI / 180.0) / 2.0) * nr, sy = sx / aspect;
    double P[4][4] = { { 2.0 * nr / sx, 0, 0, 0 }, { 0, 2.0 * nr / sy, 0, 0 },
                       { 0, 0, -(fr + nr) / (fr - nr), -(2.0 * fr * nr) / (fr - nr) }, { 0, 0, -1.0, 0 } };
    double V[4][4] = { { s[0], s[1], s[2], -(s[0] * origin[0] + s[1] * origin[1] + s[2] * origin[2]) },
                       { u[0], u[1], u[2], -(u[0] * origin[0] + u[1] * origin[1] + u[2] * origin[2]) },
                       { b[0], b[1], b[2], -(b[0] * origin[0] + b[1] * origin[1] + b[2] * origin[2]) },
                       { 0, 0, 0, 1 } };
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double a = 0;
            for (int k = 0; k < 4; ++k) a += P[i][k] * V[k][j];
            out[i][j] = (float)a;
        }
}

static void frame_build(const iso_volume* v, const iso_params* p, gv_frame* f)
{
    double ext = 0, cen[3];
    for (int k = 0; k < 3; ++k) {
        double lo = v->nbox_min[k], hi = v->nbox_max[k];
        if (hi - lo > ext) ext = hi - lo;
        cen[k] = (lo + hi) * 0.5;
    }
    const double scale = 0.5 / ext;                     /* GPURendererDirect.cpp:276-278 */
    f->scale = (float)scale;
    for (int k = 0; k < 3; ++k) f->tr[k] = (float)(-cen[k] * scale);
    f->rpos = f3_make((float)(p->origin[0] / scale + cen[0]), (float)(p->origin[1] / scale + cen[1]), (float)(p->origin[2] / scale + cen[2]));
    f->bmin = f3_make((float)v->nbox_min[0], (float)v->nbox_min[1], (float)v->nbox_min[2]);
    f->bmax = f3_make((float)v->nbox_max[0], (float)v->nbox_max[1], (float)v->nbox_max[2]);
    const double aspect = (double)p->width / (double)p->height;
    view_proj(p->origin, p->lookat, p->up, p->fov_deg, aspect, f->cur);
    view_proj(p->last_origin, p->last_lookat, p->up, p->fov_deg, aspect, f->nxt);
    double s[3], u[3], b[3];
    look_basis(p->origin, p->lookat, p->up, s, u, b);
    for (int k = 0; k < 3; ++k) { f->vrot[0][k] = (float)s[k]; f->vrot[1][k] = (float)u[k]; f->vrot[2][k] = (float)b[k]; }
    /* corner rays: inverse projection of (+-1, +-1) is the view-space direction (x / P00, y / P11, -1), rotated back
     * to the world by the transpose of the view rotation (gvdb_camera.cpp:598-601,654-665; the common scale of the
     * three points drops out in normalize()) */
    const double hx = tan(p->fov_deg * (M_PI / 180.0) / 2.0) / 2.0, hy = hx / aspect;
    double tl[3], trr[3], bl[3];
    for (int k = 0; k < 3; ++k) {
        tl[k] = -hx * s[k] + hy * u[k] - b[k];
        trr[k] = hx * s[k] + hy * u[k] - b[k];
        bl[k] = -hx * s[k] - hy * u[k] - b[k];
    }
    f->cams = f3_make((float)tl[0], (float)tl[1], (float)tl[2]);
    f->camu = f3_make((float)(trr[0] - tl[0]), (float)(trr[1] - tl[1]), (float)(trr[2] - tl[2]));
    f->camv = f3_make((float)(bl[0] - tl[0]), (float)(bl[1] - tl[1]), (float)(bl[2] - tl[2]));
    double L[3];
    if (p->light_from_camera) for (int k = 0; k < 3; ++k) L[k] = p->lookat[k] - p->origin[k];
    else for (int k = 0; k < 3; ++k) L[k] = p->light_dir[k];
    double ll = sqrt(L[0] * L[0] + L[1] * L[1] + L[2] * L[2]);
    f->light = f3_make((float)(L[0] / ll), (float)(L[1] / ll), (float)(L[2] / ll));
    f->iso = (float)p->isovalue;                        /* absolute: SetVolumeRange(args.isovalue, ..), :364 */
}

static inline float voxel(const iso_volume* v, int x, int y, int z)
{
    if ((unsigned)x >= (unsigned)v->nx || (unsigned)y >= (unsigned)v->ny || (unsigned)z >= (unsigned)v->nz) return 0.0f;
    return v->data[((size_t)z * v->ny + y) * v->nx + x];
}

/* tex3D with linear filtering at grid-local position q: voxel i is centred at i + 0.5 (cell-centred); the apron of
 * a brick holds its neighbours' voxels, i.e. the global grid with 0 where nothing is stored.  Exact float lerps
 * instead of the texture unit's fixed-point weights. */
static inline float tex(const iso_volume* v, f3 q)
{
    const float fx = q.x - 0.5f, fy = q.y - 0.5f, fz = q.z - 0.5f;
    const float cx = floorf(fx), cy = floorf(fy), cz = floorf(fz);
    const int ix = (int)cx, iy = (int)cy, iz = (int)cz;
    const float a = fx - cx, b = fy - cy, c = fz - cz;
    const float v000 = voxel(v, ix, iy, iz), v100 = voxel(v, ix + 1, iy, iz);
    const float v010 = voxel(v, ix, iy + 1, iz), v110 = voxel(v, ix + 1, iy + 1, iz);
    const float v001 = voxel(v, ix, iy, iz + 1), v101 = voxel(v, ix + 1, iy, iz + 1);
    const float v011 = voxel(v, ix, iy + 1, iz + 1), v111 = voxel(v, ix + 1, iy + 1, iz + 1);
    const float x00 = v000 + a * (v100 - v000), x10 = v010 + a * (v110 - v010);
    const float x01 = v001 + a * (v101 - v001), x11 = v011 + a * (v111 - v011);
    const float y0 = x00 + b * (x10 - x00), y1 = x01 + b * (x11 - x01);
    return y0 + c * (y1 - y0);
}

static inline f3 gradient(const iso_volume* v, f3 q)      /* cuda_gvdb_raycast.cuh:132-141, without its normalize */
{
    f3 g;
    g.x = tex(v, f3_make(q.x - 0.5f, q.y, q.z)) - tex(v, f3_make(q.x + 0.5f, q.y, q.z));
    g.y = tex(v, f3_make(q.x, q.y - 0.5f, q.z)) - tex(v, f3_make(q.x, q.y + 0.5f, q.z));
    g.z = tex(v, f3_make(q.x, q.y, q.z - 0.5f)) - tex(v, f3_make(q.x, q.y, q.z + 0.5f));
    return g;
}

/* March one occupied brick (origin vmin, 8 voxels wide) from ray parameter t.  custom != 0: the kernel's own
 * version with the bisection (render_kernel.cu:159-199); custom == 0: GVDB's plain one used by the AO rays
 * (cuda_gvdb_raycast.cuh:255-277).  Returns 1 and the hit position / raw gradient. */
static int march_brick(const iso_volume* v, float iso, f3 vmin, float t, f3 pos, f3 dir, int custom, f3* hit, f3* grad)
{
    f3 p = f3_sub(f3_add(pos, f3_scale(dir, t)), vmin);
    const f3 pstart = p;
    float tcur = 0.0f;
    for (int iter = 0; iter < GV_MAX_ITER && p.x >= 0 && p.y >= 0 && p.z >= 0 && p.x < 8.0f && p.y < 8.0f && p.z < 8.0f; ++iter) {
        if (tex(v, f3_add(p, vmin)) >= iso) {
            if (custom) {
                float lo = tcur - GV_PSTEP, hi = tcur;
                for (int i = 0; i < 10; ++i) {
                    const float mid = 0.5f * (lo + hi);
                    p = f3_add(pstart, f3_scale(dir, mid));
                    if (tex(v, f3_add(p, vmin)) >= iso) hi = mid; else lo = mid;
                }
                p = f3_add(pstart, f3_scale(dir, lo));
            }
            *hit = f3_add(p, vmin);
            *grad = gradient(v, *hit);
            return 1;
        }
        p = f3_add(p, f3_scale(dir, GV_PSTEP));
        tcur += GV_PSTEP;
    }
    return 0;
}

/* rayCast (cuda_gvdb_raycast.cuh:504-575) restated on a single-level DDA over the 8^3 bricks of the bounding box
 * instead of GVDB's <5,5,5,4,3> tree: occupied bricks are visited in the same order, so the first brick whose
 * march reports a sample >= iso is the same one.  Not restated: the 256-iteration cap of the tree walk and the
 * extra epsilon nudges when descending a tree level (entry parameters can differ by ~1e-3 voxel). */
static int ray_cast(const iso_volume* v, float iso, const gv_frame* f, f3 pos, f3 dir, int custom, f3* hit, f3* grad)
{
    if (!v->any_leaf) return 0;
    /* rayBoxIntersect, cuda_gvdb_geom.cuh:96-109 */
    float h0 = (f->bmin.x - pos.x) / dir.x, h1 = (f->bmax.x - pos.x) / dir.x;
    float h2 = (f->bmin.y - pos.y) / dir.y, h3 = (f->bmax.y - pos.y) / dir.y;
    float h4 = (f->bmin.z - pos.z) / dir.z, h5 = (f->bmax.z - pos.z) / dir.z;
    float tin = fmaxf(fmaxf(fminf(h0, h1), fminf(h2, h3)), fminf(h4, h5));
    float tout = fminf(fminf(fmaxf(h0, h1), fmaxf(h2, h3)), fmaxf(h4, h5));
    tin = (tin < 0.f) ? 0.0f : tin;
    if (tout < tin || tout < 0.f) return 0;
    float tx = tin + GV_EPS;
    const float tmax = tout - GV_EPS;
    const f3 pstep = f3_make(dir.x > 0.f ? 1.f : -1.f, dir.y > 0.f ? 1.f : -1.f, dir.z > 0.f ? 1.f : -1.f);   /* isign3 */
    /* PREPARE_DDA at the brick level (cell size 8, grid origin = bmin) */
    f3 p = f3_scale(f3_sub(f3_add(pos, f3_scale(dir, tx)), f->bmin), 0.125f);
    const f3 tdel = f3_make(fabsf(8.0f / dir.x), fabsf(8.0f / dir.y), fabsf(8.0f / dir.z));
    f3 fl = f3_make(floorf(p.x), floorf(p.y), floorf(p.z));
    f3 tside = f3_make(((fl.x - p.x + 0.5f) * pstep.x + 0.5f) * tdel.x + tx,
                       ((fl.y - p.y + 0.5f) * pstep.y + 0.5f) * tdel.y + tx,
                       ((fl.z - p.z + 0.5f) * pstep.z + 0.5f) * tdel.z + tx);
    p = fl;
    const int ox = v->nbox_min[0] >> 3, oy = v->nbox_min[1] >> 3, oz = v->nbox_min[2] >> 3;
    const int rx = (v->nbox_max[0] - v->nbox_min[0]) >> 3, ry = (v->nbox_max[1] - v->nbox_min[1]) >> 3, rz = (v->nbox_max[2] - v->nbox_min[2]) >> 3;
    for (int iter = 0; iter < 4096 && p.x >= 0 && p.y >= 0 && p.z >= 0 && p.x < (float)rx && p.y < (float)ry && p.z < (float)rz && tx <= tmax; ++iter) {
        /* NEXT_DDA */
        const float mx = (float)((tside.x < tside.y) & (tside.x <= tside.z));
        const float my = (float)((tside.y < tside.z) & (tside.y <= tside.x));
        const float mz = (float)((tside.z < tside.x) & (tside.z <= tside.y));
        const float ty = mx != 0.f ? tside.x : (my != 0.f ? tside.y : tside.z);
        const int bx = ox + (int)p.x, by = oy + (int)p.y, bz = oz + (int)p.z;
        if (v->leaf[((size_t)bz * v->by + by) * v->bx + bx]) {
            const f3 vmin = f3_make((float)(bx * 8), (float)(by * 8), (float)(bz * 8));
            if (march_brick(v, iso, vmin, tx + GV_EPS, pos, dir, custom, hit, grad)) return 1;
        }
        /* STEP_DDA */
        tx = ty;
        tside = f3_make(tside.x + mx * tdel.x, tside.y + my * tdel.y, tside.z + mz * tdel.z);
        p = f3_make(p.x + mx * pstep.x, p.y + my * pstep.y, p.z + mz * pstep.z);
    }
    return 0;
}

static inline void mat4_apply(const float m[4][4], f3 w, float out[4])
{
    for (int i = 0; i < 4; ++i) out[i] = m[i][0] * w.x + m[i][1] * w.y + m[i][2] * w.z + m[i][3] * 1.0f;
}

static float ambient_occlusion(const iso_volume* v, const iso_params* p, const gv_frame* f, const float* hemi, const float* rot,
                               f3 pos, f3 normal, int x, int y)
{
    if (p->ao_samples <= 0) return 1.0f;
    const float* nz = rot + 4 * ((x % 4) + 4 * (y % 4));
    const f3 noise = f3_make(nz[0], nz[1], nz[2]);
    const f3 tangent = f3_normalize(f3_sub(noise, f3_scale(normal, f3_dot(noise, normal))));
    const f3 bitangent = f3_cross(normal, tangent);
    float ao = 0.0f;
    const int n = p->ao_samples > 512 ? 512 : p->ao_samples;
    for (int i = 0; i < n; ++i) {
        const f3 st = f3_normalize(f3_make(hemi[4 * i], hemi[4 * i + 1], hemi[4 * i + 2]));
        const f3 sw = f3_make(f3_dot(f3_make(tangent.x, bitangent.x, normal.x), st),
                              f3_dot(f3_make(tangent.y, bitangent.y, normal.y), st),
                              f3_dot(f3_make(tangent.z, bitangent.z, normal.z), st));
        f3 h, g;
        float value = 1.0f;
        if (ray_cast(v, f->iso, f, pos, sw, 0, &h, &g)) {
            const f3 d = f3_sub(pos, h);
            const float dist = sqrtf(f3_dot(d, d));
            float yv = 1.0f - (float)p->ao_radius / dist;           /* smoothstep(1, 0, r / d) */
            yv = yv < 0.0f ? 0.0f : (yv > 1.0f ? 1.0f : yv);
            value = yv * yv * (3.0f - (2.0f * yv));
        }
        ao += value;
    }
    return ao / (float)n;
}

static void render_pixel(const iso_volume* v, const iso_params* p, const gv_frame* f, const float* hemi, const float* rot,
                         int x, int y, float* o)
{
    for (int k = 0; k < 12; ++k) o[k] = 0.0f;
    o[10] = 1.0f; o[11] = 1.0f;                                   /* ao, shadow: render_kernel.cu:219-220 */
    if (!(x >= p->viewport[0] && y >= p->viewport[1] && x < p->viewport[2] && y < p->viewport[3])) return;
    const float u = ((float)x + 0.5f) / (float)p->width, w = ((float)y + 0.5f) / (float)p->height;
    const f3 rdir = f3_normalize(f3_add(f3_add(f3_scale(f->camu, u), f3_scale(f->camv, w)), f->cams));
    f3 hit, g;
    if (!ray_cast(v, f->iso, f, f->rpos, rdir, 1, &hit, &g)) return;
    const f3 n = f3_safe_normalize(g);
    o[3] = 1.0f;
    /* shading, render_kernel.cu:232-237 (pow(x, int) restated as repeated multiplication) */
    const f3 eye = f3_normalize(f3_sub(f->rpos, hit));
    const float ndl = f3_dot(n, f->light);
    const f3 R = f3_normalize(f3_sub(f->light, f3_scale(n, 2.0f * ndl)));
    float s = f3_dot(R, eye);
    s = s > 0.0f ? s : 0.0f;
    float pw = 1.0f;
    for (int k = 0; k < p->specular_exponent; ++k) pw *= s;
    const float sc = (float)(p->specular_exponent + 2) / (2.0f * 3.41f);
    const float andl = fabsf(ndl);
    o[0] = (float)p->ambient[0] + (float)p->diffuse[0] * andl + (float)p->specular[0] * sc * pw;
    o[1] = (float)p->ambient[1] + (float)p->diffuse[1] * andl + (float)p->specular[1] * sc * pw;
    o[2] = (float)p->ambient[2] + (float)p->diffuse[2] * andl + (float)p->specular[2] * sc * pw;
    /* flow / depth / normal, :239-249 */
    const f3 world = f3_make(f->scale * hit.x + f->tr[0], f->scale * hit.y + f->tr[1], f->scale * hit.z + f->tr[2]);
    float sc_[4], sn_[4];
    mat4_apply(f->cur, world, sc_);
    mat4_apply(f->nxt, world, sn_);
    const float cx = sc_[0] / sc_[3], cy = sc_[1] / sc_[3], cz = sc_[2] / sc_[3];
    const float nx_ = sn_[0] / sn_[3], ny_ = sn_[1] / sn_[3];
    o[8] = 0.5f * (cx - nx_); o[9] = 0.5f * (cy - ny_);
    o[7] = cz;
    o[4] = f->vrot[0][0] * n.x + f->vrot[0][1] * n.y + f->vrot[0][2] * n.z;
    o[5] = f->vrot[1][0] * n.x + f->vrot[1][1] * n.y + f->vrot[1][2] * n.z;
    o[6] = f->vrot[2][0] * n.x + f->vrot[2][1] * n.y + f->vrot[2][2] * n.z;
    o[10] = ambient_occlusion(v, p, f, hemi, rot, f3_sub(hit, f3_scale(rdir, 1e-3f)), n, x, y);
}

int iso_render_gvdb(const iso_volume* v, const iso_params* p, float* out, int threads)
{
    gv_frame f;
    frame_build(v, p, &f);
    static float hemi[512 * 4], rot[16 * 4];
    static int tables_ready = 0;
    if (!tables_ready) { iso_ao_tables(hemi, rot); tables_ready = 1; }
    const int W = p->width, H = p->height;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (int j = 0; j < H; ++j)
        for (int i = 0; i < W; ++i)
            render_pixel(v, p, &f, hemi, rot, i, j, out + ((size_t)j * W + i) * 12);
    return 0;
}
