/*
 * iso_oracle.c -- TEST INFRASTRUCTURE ONLY.  CPU restatement of the reference's CPU
 * isosurface ray tracer.  See iso_oracle.h for the "parity unpinned" statement.
 *
 * Every function cites the reference code it follows (paths relative to /root/reference,
 * TP/ = third-party/include/).  Build with -ffp-contract=off: the operation order below is
 * the contract the HIP kernel is checked against bit-for-bit (hit mask) .
 *
 * Conventions: OpenVDB matrices act on row vectors; ray quantities are
 * double; grid values and the trilinear interpolation are float.
 */
#include "iso_oracle.h"

#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "iso_oracle_priv.h"

/* ------------------------------------------------------------------ volume */

iso_volume* iso_volume_create(const float* dense, int nx, int ny, int nz)
{
    if (!dense || nx <= 0 || ny <= 0 || nz <= 0 || nx > 4096 || ny > 4096 || nz > 4096) return NULL;
    iso_volume* v = (iso_volume*)calloc(1, sizeof(iso_volume));
    size_t n = (size_t)nx * ny * nz;
    v->nx = nx; v->ny = ny; v->nz = nz;
    v->data = (float*)malloc(n * sizeof(float));
    memcpy(v->data, dense, n * sizeof(float));
    v->bx = (nx + 7) / 8; v->by = (ny + 7) / 8; v->bz = (nz + 7) / 8;
    v->mx = (nx + 127) / 128; v->my = (ny + 127) / 128; v->mz = (nz + 127) / 128;
    v->leaf = (unsigned char*)calloc((size_t)v->bx * v->by * v->bz, 1);
    v->touched = (unsigned char*)calloc((size_t)v->bx * v->by * v->bz, 1);
    v->node1 = (unsigned char*)calloc((size_t)v->mx * v->my * v->mz, 1);
    int amin[3] = { INT32_MAX, INT32_MAX, INT32_MAX }, amax[3] = { INT32_MIN, INT32_MIN, INT32_MIN };
    float maxv = -FLT_MAX;
    /* A voxel is active iff it was written, i.e. iff it differs from the background 0
     * (harness contract; FloatGrid::evalMinMax runs over active values,
     * CPURenderer.cpp:501-502). */
    for (int z = 0; z < nz; ++z)
        for (int y = 0; y < ny; ++y) {
            const float* row = v->data + ((size_t)z * ny + y) * nx;
            for (int x = 0; x < nx; ++x) {
                float f = row[x];
                if (f != 0.0f) {
                    v->leaf[((size_t)(z >> 3) * v->by + (y >> 3)) * v->bx + (x >> 3)] = 1;
                    v->node1[((size_t)(z >> 7) * v->my + (y >> 7)) * v->mx + (x >> 7)] = 1;
                    if (x < amin[0]) amin[0] = x; if (x > amax[0]) amax[0] = x;
                    if (y < amin[1]) amin[1] = y; if (y > amax[1]) amax[1] = y;
                    if (z < amin[2]) amin[2] = z; if (z > amax[2]) amax[2] = z;
                    if (f > maxv) maxv = f;
                }
            }
        }
    /* node-level bbox: union of the full 8^3 boxes of existing leaves, then max += 1
     * (IsoVolumeRayTracer.h:195-197, TP/openvdb/tree/LeafNode.h:1484-1494) */
    int lmin[3] = { INT32_MAX, INT32_MAX, INT32_MAX }, lmax[3] = { INT32_MIN, INT32_MIN, INT32_MIN };
    int nleaf = 0;
    for (int z = 0; z < v->bz; ++z)
        for (int y = 0; y < v->by; ++y)
            for (int x = 0; x < v->bx; ++x)
                if (v->leaf[((size_t)z * v->by + y) * v->bx + x]) {
                    ++nleaf;
                    if (x < lmin[0]) lmin[0] = x; if (x > lmax[0]) lmax[0] = x;
                    if (y < lmin[1]) lmin[1] = y; if (y > lmax[1]) lmax[1] = y;
                    if (z < lmin[2]) lmin[2] = z; if (z > lmax[2]) lmax[2] = z;
                }
    v->nleaf = nleaf;
    v->any_leaf = nleaf > 0;
    if (!v->any_leaf) { /* the reference throws on empty grids (IsoVolumeRayTracer.h:188-190) */
        iso_volume_free(v);
        return NULL;
    }
    for (int k = 0; k < 3; ++k) {
        v->nbox_min[k] = lmin[k] * 8;
        v->nbox_max[k] = lmax[k] * 8 + 7 + 1;
        v->abox_min[k] = amin[k];
        v->abox_max[k] = amax[k];
    }
    v->max_value = maxv;
    /* grid normalisation, CPURenderer.cpp:448-458 with unit voxel size.
     * Transform::indexToWorld(CoordBBox) lives in OpenVDB 6.0.1's Transform.cc, which is not
     * vendored under /root/reference; its published behaviour is BBoxd(min.asVec3d(), max.asVec3d())
     * mapped corner-wise, so the world extents are (max - min) without the +1 of integer boxes. */
    double ext[3], cen[3];
    for (int k = 0; k < 3; ++k) {
        double lo = (double)amin[k], hi = (double)amax[k];
        ext[k] = hi - lo;
        cen[k] = (lo + hi) * 0.5;                 /* BBox<Vec3d>::getCenter, TP/openvdb/math/BBox.h:272-275 */
    }
    double m = ext[0];
    if (ext[1] > m) m = ext[1];
    if (ext[2] > m) m = ext[2];
    double scale = 1.0 / m;
    /* postTranslate(-centre) then postScale(scale) on a unit UniformScaleMap */
    v->s = 1.0 * scale;
    v->sinv = 1.0 / v->s;
    for (int k = 0; k < 3; ++k) v->t[k] = (-cen[k]) * scale;
    return v;
}

/* Tile of a larger volume (tests of the multi-GPU tiled render): same rules as the product's
 * isoLoadDenseTileHost.  The tile walks the GLOBAL ray -- world map, isovalue scale and node-level bbox of
 * the global volume, DDAs in global index coordinates -- and owns the leaves inside [clip_lo, clip_hi);
 * all other leaves are stepped over.  Because the reference re-initialises the voxel DDA per leaf from the
 * leaf's span (IsoVolumeRayTracer.h:37-46), the tile owning the first leaf with a crossing computes exactly
 * the unsplit pixel.  origin and clip_lo must be multiples of 8.  An all-zero tile is allowed. */
iso_volume* iso_volume_create_tile(const float* dense, int nx, int ny, int nz, const int origin[3],
                                   const int gmin[3], const int gmax[3], float global_max,
                                   const int clip_lo[3], const int clip_hi[3])
{
    const int dims[3] = { nx, ny, nz };
    for (int k = 0; k < 3; ++k) {
        if (origin[k] < 0 || (origin[k] & 7) || (clip_lo[k] & 7) || origin[k] + dims[k] > 4096) return NULL;
        if (clip_lo[k] < origin[k] || clip_hi[k] > origin[k] + dims[k]) return NULL;
    }
    size_t n = (size_t)nx * ny * nz;
    int any = 0;
    for (size_t i = 0; i < n && !any; ++i) any = dense[i] != 0.0f;
    iso_volume* v;
    if (any) v = iso_volume_create(dense, nx, ny, nz);
    else {
        float* tmp = (float*)malloc(n * sizeof(float));
        memcpy(tmp, dense, n * sizeof(float));
        tmp[0] = 1.0f;                       /* build the containers, then forget the dummy voxel */
        v = iso_volume_create(tmp, nx, ny, nz);
        free(tmp);
        if (v) { v->data[0] = 0.0f; memset(v->leaf, 0, (size_t)v->bx * v->by * v->bz); }
    }
    if (!v) return NULL;
    for (int k = 0; k < 3; ++k) { v->org[k] = origin[k]; v->n1o[k] = origin[k] >> 7; }
    /* node1 over GLOBAL 128^3 node coordinates; leaf = exists AND owned */
    free(v->node1);
    v->mx = ((origin[0] + nx - 1) >> 7) - v->n1o[0] + 1;
    v->my = ((origin[1] + ny - 1) >> 7) - v->n1o[1] + 1;
    v->mz = ((origin[2] + nz - 1) >> 7) - v->n1o[2] + 1;
    v->node1 = (unsigned char*)calloc((size_t)v->mx * v->my * v->mz, 1);
    int nleaf = 0;
    for (int z = 0; z < v->bz; ++z)
        for (int y = 0; y < v->by; ++y)
            for (int x = 0; x < v->bx; ++x) {
                unsigned char* lf = &v->leaf[((size_t)z * v->by + y) * v->bx + x];
                const int c[3] = { origin[0] + x * 8, origin[1] + y * 8, origin[2] + z * 8 };
                for (int k = 0; k < 3; ++k) if (c[k] < clip_lo[k] || c[k] >= clip_hi[k]) *lf = 0;
                if (*lf) {
                    ++nleaf;
                    v->node1[((size_t)((c[2] >> 7) - v->n1o[2]) * v->my + ((c[1] >> 7) - v->n1o[1])) * v->mx + ((c[0] >> 7) - v->n1o[0])] = 1;
                }
            }
    v->nleaf = nleaf;
    v->any_leaf = nleaf > 0;
    double ext[3], cen[3];
    for (int k = 0; k < 3; ++k) { double lo = (double)gmin[k], hi = (double)gmax[k]; ext[k] = hi - lo; cen[k] = (lo + hi) * 0.5; }
    double m = ext[0];
    if (ext[1] > m) m = ext[1];
    if (ext[2] > m) m = ext[2];
    double scale = 1.0 / m;
    v->s = 1.0 * scale;
    v->sinv = 1.0 / v->s;
    v->max_value = global_max;
    for (int k = 0; k < 3; ++k) {
        v->t[k] = (-cen[k]) * scale;
        v->abox_min[k] = gmin[k]; v->abox_max[k] = gmax[k];
        v->nbox_min[k] = gmin[k] & ~7;               /* the leaf holding the extreme active voxel bounds the global box */
        v->nbox_max[k] = (gmax[k] & ~7) + 7 + 1;
    }
    return v;
}

void iso_volume_free(iso_volume* v)
{
    if (!v) return;
    free(v->data); free(v->leaf); free(v->node1); free(v->touched); free(v);
}

void iso_volume_info(const iso_volume* v, int info[13], double st[4], float* max_value)
{
    for (int k = 0; k < 3; ++k) {
        info[k] = v->nbox_min[k]; info[3 + k] = v->nbox_max[k];
        info[6 + k] = v->abox_min[k]; info[9 + k] = v->abox_max[k];
    }
    info[12] = v->nleaf;
    st[0] = v->s; st[1] = v->t[0]; st[2] = v->t[1]; st[3] = v->t[2];
    *max_value = v->max_value;
}

void iso_params_default(iso_params* p)
{
    /* GPURendererDirect.cpp:103-128 (binding-surface defaults) */
    memset(p, 0, sizeof(*p));
    p->width = 512; p->height = 512; p->fov_deg = 45.0;
    p->origin[2] = -1.0; p->up[1] = 1.0;
    p->last_origin[2] = -1.0;
    p->isovalue = 0.0;
    for (int k = 0; k < 3; ++k) { p->ambient[k] = 0.1; p->diffuse[k] = 0.7; p->specular[k] = 1.0; }
    p->specular_exponent = 32;
    p->light_from_camera = 1;
    p->viewport[2] = 512; p->viewport[3] = 512;
    p->ao_samples = 0; p->ao_radius = 0.01;
}

/* ------------------------------------------------------------------ small vector helpers
 * (TP/openvdb/math/Vec3.h:216-231,245-250,377-385,396-403) */

static double v3_dot(const double a[3], const double b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static double v3_len(const double a[3]) { return sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }
static void v3_cross(const double a[3], const double b[3], double o[3])
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
/* Vec3::unit(): component-wise division by the length */
static void v3_unit(const double a[3], double o[3])
{
    double l = v3_len(a);
    o[0] = a[0] / l; o[1] = a[1] / l; o[2] = a[2] / l;
}
/* Vec3::normalize(eps=1e-7): multiply by the reciprocal; leave untouched when ~0 */
static void v3_normalize(double a[3])
{
    double d = v3_len(a);
    if (!(fabs(d - 0.0) > 1.0e-7)) return;
    double r = 1.0 / d;
    a[0] *= r; a[1] *= r; a[2] *= r;
}

/* ------------------------------------------------------------------ camera
 * TP/openvdb/tools/RayTracer.h:404-531, TP/openvdb/math/Mat.h:758-774 */

typedef struct {
    double J[3][3];      /* rows: horizontal, up, forward */
    double org[3];
    double d0[3];        /* base ray direction = (0,0,-1) * J */
    double sw, sh;
    double V[4][4];      /* inverse camera matrix (row-vector convention) */
} camera_t;

static void camera_build(camera_t* c, const double origin[3], const double lookat[3], const double up[3],
                         double fov_deg, int W, int H)
{
    const double aperture = 0.01;                                      /* CPURenderer.cpp:484 */
    double focal = aperture / (2.0 * (tan(fov_deg * M_PI / 360.0)));  /* RayTracer.h:527-530 */
    c->sw = 0.5 * aperture / focal;                                    /* RayTracer.h:498 */
    c->sh = c->sw * (double)H / (double)W;                             /* RayTracer.h:411 */
    double dir[3] = { origin[0] - lookat[0], origin[1] - lookat[1], origin[2] - lookat[2] };
    double fwd[3], hor[3], upv[3], upn[3], tmp[3];
    v3_unit(dir, fwd);
    v3_unit(up, upn);
    v3_cross(upn, fwd, tmp); v3_unit(tmp, hor);
    v3_cross(fwd, hor, tmp); v3_unit(tmp, upv);
    for (int k = 0; k < 3; ++k) { c->J[0][k] = hor[k]; c->J[1][k] = upv[k]; c->J[2][k] = fwd[k]; c->org[k] = origin[k]; }
    /* initRay: dir = transform3x3((0,0,-1)) (TP/openvdb/math/Mat4.h:1115-1121) */
    for (int k = 0; k < 3; ++k) c->d0[k] = 0.0 * c->J[0][k] + 0.0 * c->J[1][k] + (-1.0) * c->J[2][k];
    /* Mat4::inverse(), affine branch (TP/openvdb/math/Mat4.h:554-562,590-606,662-672) */
    double (*m)[3] = c->J;
    double m0011 = m[0][0] * m[1][1], m0012 = m[0][0] * m[1][2], m0110 = m[0][1] * m[1][0];
    double m0210 = m[0][2] * m[1][0], m0120 = m[0][1] * m[2][0], m0220 = m[0][2] * m[2][0];
    double detA = m0011 * m[2][2] - m0012 * m[2][1] - m0110 * m[2][2]
                + m0210 * m[2][1] + m0120 * m[1][2] - m0220 * m[1][1];
    detA = 1.0 / detA;
    double (*inv)[4] = c->V;
    inv[0][0] = detA * ( m[1][1] * m[2][2] - m[1][2] * m[2][1]);
    inv[0][1] = detA * (-m[0][1] * m[2][2] + m[0][2] * m[2][1]);
    inv[0][2] = detA * ( m[0][1] * m[1][2] - m[0][2] * m[1][1]);
    inv[1][0] = detA * (-m[1][0] * m[2][2] + m[1][2] * m[2][0]);
    inv[1][1] = detA * ( m[0][0] * m[2][2] - m0220);
    inv[1][2] = detA * ( m0210 - m0012);
    inv[2][0] = detA * ( m[1][0] * m[2][1] - m[1][1] * m[2][0]);
    inv[2][1] = detA * ( m0120 - m[0][0] * m[2][1]);
    inv[2][2] = detA * ( m0011 - m0110);
    const double* o = c->org;
    inv[3][0] = -(o[0] * inv[0][0] + o[1] * inv[1][0] + o[2] * inv[2][0]);
    inv[3][1] = -(o[0] * inv[0][1] + o[1] * inv[1][1] + o[2] * inv[2][1]);
    inv[3][2] = -(o[0] * inv[0][2] + o[1] * inv[1][2] + o[2] * inv[2][2]);
    inv[0][3] = 0.0; inv[1][3] = 0.0; inv[2][3] = 0.0; inv[3][3] = 1.0;
}

/* ------------------------------------------------------------------ ray + DDA
 * TP/openvdb/math/Ray.h:87-96,135,260-293 ; TP/openvdb/math/DDA.h:79-139 */

typedef struct { double eye[3], dir[3], inv[3], t0, t1; } ray_t;

static inline void ray_at(const ray_t* r, double t, double p[3])
{
    p[0] = r->eye[0] + r->dir[0] * t;
    p[1] = r->eye[1] + r->dir[1] * t;
    p[2] = r->eye[2] + r->dir[2] * t;
}

static int ray_clip(ray_t* r, const int bmin[3], const int bmax[3])
{
    double t0 = r->t0, t1 = r->t1;
    for (int i = 0; i < 3; ++i) {
        double a = ((double)bmin[i] - r->eye[i]) * r->inv[i];
        double b = ((double)bmax[i] - r->eye[i]) * r->inv[i];
        if (a > b) { double s = a; a = b; b = s; }
        if (a > t0) t0 = a;
        if (b < t1) t1 = b;
        if (t0 > t1) return 0;
    }
    r->t0 = t0; r->t1 = t1;
    return 1;
}

typedef struct { double t0, t1, delta[3], next[3]; int voxel[3], step[3]; } dda_t;

static inline void dda_init(dda_t* d, const ray_t* r, int log2dim)
{
    const int DIM = 1 << log2dim;
    d->t0 = r->t0; d->t1 = r->t1;
    double pos[3];
    ray_at(r, d->t0, pos);
    for (int a = 0; a < 3; ++a) {
        d->voxel[a] = ((int)floor(pos[a])) & (~(DIM - 1));
        if (r->dir[a] == 0.0) {
            d->step[a] = 0; d->next[a] = DBL_MAX; d->delta[a] = DBL_MAX;
        } else if (r->inv[a] > 0) {
            d->step[a] = DIM;
            d->next[a] = d->t0 + ((double)(d->voxel[a] + DIM) - pos[a]) * r->inv[a];
            d->delta[a] = (double)d->step[a] * r->inv[a];
        } else {
            d->step[a] = -DIM;
            d->next[a] = d->t0 + ((double)d->voxel[a] - pos[a]) * r->inv[a];
            d->delta[a] = (double)d->step[a] * r->inv[a];
        }
    }
}

static inline double dda_next(const dda_t* d)
{
    double a = d->t1 < d->next[0] ? d->t1 : d->next[0];
    double b = d->next[1] < d->next[2] ? d->next[1] : d->next[2];
    return b < a ? b : a;       /* std::min(std::min(a,b),std::min(c,d)), TP/openvdb/math/Math.h:623-626 */
}

static inline int dda_step(dda_t* d)
{
    /* MinIndex tie table, TP/openvdb/math/Math.h:893-902 */
    static const int hash[8] = { 2, 1, 9, 1, 2, 9, 0, 0 };
    int key = ((d->next[0] < d->next[1]) << 2) + ((d->next[0] < d->next[2]) << 1) + (d->next[1] < d->next[2]);
    int ax = hash[key];
    d->t0 = d->next[ax];
    d->next[ax] += d->delta[ax];
    d->voxel[ax] += d->step[ax];
    return d->t0 <= d->t1;
}

/* ------------------------------------------------------------------ sampling
 * TP/openvdb/math/Stencils.h:110-114,335-354,402-411 */

typedef struct { long long samples, steps; } counters_t;

/* coordinates are GLOBAL index coordinates; v->org is zero unless the volume is a tile */
static inline float grid_value(const iso_volume* v, int x, int y, int z)
{
    x -= v->org[0]; y -= v->org[1]; z -= v->org[2];
    if ((unsigned)x >= (unsigned)v->nx || (unsigned)y >= (unsigned)v->ny || (unsigned)z >= (unsigned)v->nz) return 0.0f;
    return v->data[((size_t)z * v->ny + y) * v->nx + x];
}

static inline void touch(const iso_volume* v, int x, int y, int z)
{
    x -= v->org[0]; y -= v->org[1]; z -= v->org[2];
    if ((unsigned)x >= (unsigned)v->nx || (unsigned)y >= (unsigned)v->ny || (unsigned)z >= (unsigned)v->nz) return;
    unsigned char* p = &v->touched[((size_t)(z >> 3) * v->by + (y >> 3)) * v->bx + (x >> 3)];
    if (!*p) *p = 1;
}

static inline float interp(const iso_volume* v, const double p[3], counters_t* cn)
{
    int cx = (int)floor(p[0]), cy = (int)floor(p[1]), cz = (int)floor(p[2]);
    float V000 = grid_value(v, cx, cy, cz),         V001 = grid_value(v, cx, cy, cz + 1);
    float V010 = grid_value(v, cx, cy + 1, cz),     V011 = grid_value(v, cx, cy + 1, cz + 1);
    float V100 = grid_value(v, cx + 1, cy, cz),     V101 = grid_value(v, cx + 1, cy, cz + 1);
    float V110 = grid_value(v, cx + 1, cy + 1, cz), V111 = grid_value(v, cx + 1, cy + 1, cz + 1);
    if (cn) {
        cn->samples++;
        touch(v, cx, cy, cz); touch(v, cx + 1, cy + 1, cz + 1);
        touch(v, cx + 1, cy, cz); touch(v, cx, cy + 1, cz); touch(v, cx, cy, cz + 1);
        touch(v, cx + 1, cy + 1, cz); touch(v, cx + 1, cy, cz + 1); touch(v, cx, cy + 1, cz + 1);
    }
    /* position narrowed double->float by Vec3<float>(Vec3<double>) (Vec3.h:101-107) */
    float u = (float)p[0] - (float)cx;
    float w_ = (float)p[1] - (float)cy;
    float w = (float)p[2] - (float)cz;
    float A = V000 + (V001 - V000) * w;
    float B = V010 + (V011 - V010) * w;
    float C = A + (B - A) * w_;
    A = V100 + (V101 - V100) * w;
    B = V110 + (V111 - V110) * w;
    float D = A + (B - A) * w_;
    return C + (D - C) * u;
}

/* IsoVolumeRayTracer.h:66-71: float - double -> double -> float */
static inline float interp_value(const iso_volume* v, const ray_t* r, double iso, double t, counters_t* cn)
{
    double p[3];
    ray_at(r, t, p);
    return (float)((double)interp(v, p, cn) - iso);
}

/* ------------------------------------------------------------------ hierarchy
 * IsoVolumeRayTracer.h:37-46 (node levels) and :81-114 (voxel level) */

static inline int has_leaf(const iso_volume* v, const int c[3])
{
    const int x = c[0] - v->org[0], y = c[1] - v->org[1], z = c[2] - v->org[2];
    if ((unsigned)x >= (unsigned)v->nx || (unsigned)y >= (unsigned)v->ny || (unsigned)z >= (unsigned)v->nz) return 0;
    return v->leaf[((size_t)(z >> 3) * v->by + (y >> 3)) * v->bx + (x >> 3)];
}
static inline int has_node1(const iso_volume* v, const int c[3])
{
    const int x = (c[0] >> 7) - v->n1o[0], y = (c[1] >> 7) - v->n1o[1], z = (c[2] >> 7) - v->n1o[2];
    if ((unsigned)x >= (unsigned)v->mx || (unsigned)y >= (unsigned)v->my || (unsigned)z >= (unsigned)v->mz) return 0;
    return v->node1[((size_t)z * v->my + y) * v->mx + x];
}
static inline int has_node2(const iso_volume* v, const int c[3])
{
    return v->any_leaf && c[0] == 0 && c[1] == 0 && c[2] == 0;   /* dims <= 4096: one level-2 node at the origin */
}

static int hits_voxel(const iso_volume* v, ray_t* ray, double iso, double* time, counters_t* cn)
{
    dda_t d;
    dda_init(&d, ray, 0);
    double t0 = d.t0;
    float v0 = interp_value(v, ray, iso, t0, cn);
    do {
        double t1 = dda_next(&d);
        float v1 = interp_value(v, ray, iso, t1, cn);
        if (v0 * v1 <= 0.0f) {                       /* ZeroCrossing, Math.h:710-713 */
            double t = 0.5 * (t0 + t1);
            for (int i = 0; i < 5; ++i) {            /* BINARY_STEPS, IsoVolumeRayTracer.h:93 */
                float v2 = interp_value(v, ray, iso, t, cn);
                if (v0 * v2 <= 0.0f) t1 = t;
                else { t0 = t; v0 = v2; }
                t = 0.5 * (t0 + t1);
            }
            *time = t;
            return 1;
        }
        t0 = t1; v0 = v1;
        if (cn) cn->steps++;
    } while (dda_step(&d));
    return 0;
}

static int hits_leaf(const iso_volume* v, ray_t* ray, double iso, double* time, counters_t* cn)
{
    dda_t d;
    dda_init(&d, ray, 3);
    do {
        if (has_leaf(v, d.voxel)) {
            ray->t0 = d.t0; ray->t1 = dda_next(&d);
            if (hits_voxel(v, ray, iso, time, cn)) return 1;
        }
    } while (dda_step(&d));
    return 0;
}

static int hits_node1(const iso_volume* v, ray_t* ray, double iso, double* time, counters_t* cn)
{
    dda_t d;
    dda_init(&d, ray, 7);
    do {
        if (has_node1(v, d.voxel)) {
            ray->t0 = d.t0; ray->t1 = dda_next(&d);
            if (hits_leaf(v, ray, iso, time, cn)) return 1;
        }
    } while (dda_step(&d));
    return 0;
}

static int hits_node2(const iso_volume* v, ray_t* ray, double iso, double* time, counters_t* cn)
{
    dda_t d;
    dda_init(&d, ray, 12);
    do {
        if (has_node2(v, d.voxel)) {
            ray->t0 = d.t0; ray->t1 = dda_next(&d);
            if (hits_node1(v, ray, iso, time, cn)) return 1;
        }
    } while (dda_step(&d));
    return 0;
}

/* ------------------------------------------------------------------ ambient occlusion
 * GPURendererDirect.cpp:146-189 (tables) and render_kernel.cu:109-146 (mode 1, ray sampling),
 * with the secondary rays cast by the CPU tracer's own hierarchy (doubles) instead of GVDB's.
 * The reference has no CPU implementation of this (CPURenderer.cpp:736 writes 1), so this
 * restatement is the only oracle the HIP kernel's AO has. */

static unsigned lcg_next(unsigned* st)     /* minstd_rand0: x <- 16807 x mod (2^31 - 1) */
{
    *st = (unsigned)(((unsigned long long)*st * 16807ULL) % 2147483647ULL);
    return *st;
}
static float lcg_uniform(unsigned* st)     /* [0,1) */
{
    return (float)(lcg_next(st) - 1u) * (1.0f / 2147483646.0f);
}

void iso_ao_tables(float* hemi, float* rot)
{
    unsigned st = 1u;
    for (int i = 0; i < 512; ++i) {
        float u1 = lcg_uniform(&st), u2 = lcg_uniform(&st);
        float r = sqrtf(u1);
        float theta = (float)(2 * M_PI * u2);
        float x = r * cosf(theta), y = r * sinf(theta);
        float scale = lcg_uniform(&st);
        scale = (float)(0.1 + 0.9 * scale * scale);
        hemi[4 * i + 0] = x * scale; hemi[4 * i + 1] = y * scale;
        hemi[4 * i + 2] = sqrtf(1 - u1) * scale; hemi[4 * i + 3] = 0.f;
    }
    for (int i = 0; i < 16; ++i) {
        float x = lcg_uniform(&st) * 2 - 1, y = lcg_uniform(&st) * 2 - 1;
        float linv = 1.0f / sqrtf(x * x + y * y);
        rot[4 * i + 0] = x * linv; rot[4 * i + 1] = y * linv; rot[4 * i + 2] = 0.f; rot[4 * i + 3] = 0.f;
    }
}

/* world-space ray (origin o, unit direction d) -> first hit position; returns 0 on a miss */
static int cast_world(const iso_volume* v, const double o[3], const double d[3], double iso, double hitw[3], counters_t* cn)
{
    ray_t r;
    double di[3];
    for (int k = 0; k < 3; ++k) { r.eye[k] = (o[k] - v->t[k]) * v->sinv; di[k] = d[k] * v->sinv; }
    double len = v3_len(di);
    for (int k = 0; k < 3; ++k) { r.dir[k] = di[k] / len; r.inv[k] = 1.0 / r.dir[k]; }
    r.t0 = len * 1e-9; r.t1 = len * DBL_MAX;       /* Ray's default time span, Ray.h:83-84 */
    if (!ray_clip(&r, v->nbox_min, v->nbox_max)) return 0;
    double it;
    if (!hits_node2(v, &r, iso, &it, cn)) return 0;
    double ip[3];
    ray_at(&r, it, ip);
    for (int k = 0; k < 3; ++k) hitw[k] = ip[k] * v->s + v->t[k];
    return 1;
}

static double ambient_occlusion(const iso_volume* v, const iso_params* p, const float* hemi, const float* rot,
                                const double pos[3], const double nrm[3], int x, int y, double iso, counters_t* cn)
{
    if (p->ao_samples <= 0) return 1.0;
    const float* nz = rot + 4 * ((x % 4) + 4 * (y % 4));
    double noise[3] = { nz[0], nz[1], nz[2] };
    double dn = v3_dot(noise, nrm);
    double tan_[3] = { noise[0] - nrm[0] * dn, noise[1] - nrm[1] * dn, noise[2] - nrm[2] * dn };
    v3_normalize(tan_);
    double bit[3];
    v3_cross(nrm, tan_, bit);
    double ao = 0.0;
    int n = p->ao_samples > 512 ? 512 : p->ao_samples;
    for (int i = 0; i < n; ++i) {
        double st[3] = { hemi[4 * i], hemi[4 * i + 1], hemi[4 * i + 2] };
        v3_normalize(st);
        double sw[3];
        for (int k = 0; k < 3; ++k) sw[k] = tan_[k] * st[0] + bit[k] * st[1] + nrm[k] * st[2];
        v3_normalize(sw);
        double h[3];
        double value = 1.0;
        if (cast_world(v, pos, sw, iso, h, cn)) {
            double dd[3] = { pos[0] - h[0], pos[1] - h[1], pos[2] - h[2] };
            double dist = v3_len(dd);
            double yv = 1.0 - p->ao_radius / dist;            /* smoothstep(1, 0, r/d), cuda_math.cuh:1507-1511 */
            yv = yv < 0.0 ? 0.0 : (yv > 1.0 ? 1.0 : yv);
            value = yv * yv * (3.0 - (2.0 * yv));
        }
        ao += value;
    }
    return ao / n;
}

/* ------------------------------------------------------------------ per-pixel driver
 * IsoVolumeRayTracer.h:294-309 (intersectsWS), :274-292 (gradient), :502-551 (operator()),
 * PhongShader.h:27-38, CPURenderer.cpp:726-737 (channel packing) */

static void render_pixel(const iso_volume* v, const iso_params* p, const camera_t* cam, const camera_t* nxt,
                         double iso, const double light[3], const float* hemi, const float* rot,
                         int i, int j, float* o, counters_t* cn, long long* hits)
{
    for (int k = 0; k < 12; ++k) o[k] = 0.0f;
    o[8] = -0.0f; o[9] = -0.0f;          /* -static_cast<float>(0) , CPURenderer.cpp:734-735 */
    o[10] = 1.0f; o[11] = 0.0f;          /* CPURenderer.cpp:736-737 */
    if (!(i >= p->viewport[0] && j >= p->viewport[1] && i < p->viewport[2] && j < p->viewport[3])) {
        o[8] = 0.0f; o[9] = 0.0f;
        return;
    }
    const int W = p->width, H = p->height;
    /* PerspectiveCamera::getRay, RayTracer.h:507-516 ; rasterToScreen :444-448 */
    double ds[3];
    ds[0] = (2 * ((double)i + 0.5) / (double)W - 1) * cam->sw;
    ds[1] = (1 - 2 * ((double)j + 0.5) / (double)H) * cam->sh;
    ds[2] = -1.0;
    double dir[3];
    for (int k = 0; k < 3; ++k) dir[k] = ds[0] * cam->J[0][k] + ds[1] * cam->J[1][k] + ds[2] * cam->J[2][k];
    v3_normalize(dir);
    double sc = 1.0 / v3_dot(dir, cam->d0);
    double wt0 = 1e-3 * sc, wt1 = DBL_MAX * sc;
    /* Ray::applyInverseMap with a ScaleTranslateMap, Ray.h:177-185, Maps.h:1287-1305 */
    ray_t r;
    double di[3];
    for (int k = 0; k < 3; ++k) {
        r.eye[k] = (cam->org[k] - v->t[k]) * v->sinv;
        di[k] = dir[k] * v->sinv;
    }
    double len = v3_len(di);
    for (int k = 0; k < 3; ++k) { r.dir[k] = di[k] / len; r.inv[k] = 1.0 / r.dir[k]; }
    r.t0 = len * wt0; r.t1 = len * wt1;
    if (!ray_clip(&r, v->nbox_min, v->nbox_max)) return;
    double it;
    if (!hits_node2(v, &r, iso, &it, cn)) return;
    (*hits)++;
    double ip[3];
    ray_at(&r, it, ip);
    double world[3];
    for (int k = 0; k < 3; ++k) world[k] = ip[k] * v->s + v->t[k];
    /* gradient: central differences of the trilinear interpolant at +-1 voxel */
    double n[3];
    for (int k = 0; k < 3; ++k) {
        double pp[3] = { ip[0], ip[1], ip[2] }, pm[3] = { ip[0], ip[1], ip[2] };
        pp[k] = ip[k] + 1.0; pm[k] = ip[k] - 1.0;
        for (int q = 0; q < 3; ++q) if (q != k) { pp[q] = ip[q] + 0.0; pm[q] = ip[q] - 0.0; }
        double a = (double)interp(v, pp, cn);
        a -= (double)interp(v, pm, cn);
        n[k] = a;
    }
    v3_normalize(n);
    double js[3] = { r.dir[0] * v->s, r.dir[1] * v->s, r.dir[2] * v->s };
    double wtime = it * v3_len(js);
    /* Phong (two sided) */
    double ndl = v3_dot(n, light);
    double col[3], refl[3];
    for (int k = 0; k < 3; ++k) col[k] = p->ambient[k];
    double andl = fabs(ndl);
    for (int k = 0; k < 3; ++k) col[k] += p->diffuse[k] * andl;
    double two = 2 * ndl;
    for (int k = 0; k < 3; ++k) refl[k] = two * n[k] - light[k];
    double rd = v3_dot(refl, dir);
    double base = rd > 0.0 ? rd : 0.0;                      /* Max(0.0, x) = std::max */
    double pw = 1.0;
    { int e = p->specular_exponent; double x = base; if (e < 0) { e = -e; x = 1.0 / x; } while (e--) pw *= x; }  /* Pow(T,int), Math.h:518-527 */
    double c1 = (p->specular_exponent + 2) / (2 * M_PI);
    for (int k = 0; k < 3; ++k) col[k] += (p->specular[k] * c1) * pw;
    o[0] = (float)col[0]; o[1] = (float)col[1]; o[2] = (float)col[2];
    o[3] = (wtime == 0) ? 0.0f : 1.0f;                      /* IsoVolumeRayTracer.h:528-529 */
    if (wtime > 0) {
        double n2[3];
        for (int k = 0; k < 3; ++k) n2[k] = n[0] * cam->V[0][k] + n[1] * cam->V[1][k] + n[2] * cam->V[2][k];
        if (n2[2] < 0) { n2[0] = -n2[0]; n2[1] = -n2[1]; n2[2] = -n2[2]; }
        o[4] = (float)n2[0]; o[5] = (float)n2[1]; o[6] = (float)n2[2]; o[7] = (float)wtime;
        double sc4[4], sn4[4];
        for (int k = 0; k < 4; ++k) {
            sc4[k] = world[0] * cam->V[0][k] + world[1] * cam->V[1][k] + world[2] * cam->V[2][k] + 1.0 * cam->V[3][k];
            sn4[k] = world[0] * nxt->V[0][k] + world[1] * nxt->V[1][k] + world[2] * nxt->V[2][k] + 1.0 * nxt->V[3][k];
        }
        double cx = sc4[0] / sc4[3], cy = sc4[1] / sc4[3];
        double nx_ = sn4[0] / sn4[3], ny_ = sn4[1] / sn4[3];
        o[8] = -(float)(nx_ - cx);
        o[9] = -(float)(ny_ - cy);
        if (p->ao_samples > 0) {
            /* hemisphere around the normal that faces the viewer; start 1e-3 world units back along
             * the primary ray (aoBias, render_kernel.cu:38,251) */
            double na[3] = { n[0], n[1], n[2] };
            if (v3_dot(n, dir) > 0) { na[0] = -na[0]; na[1] = -na[1]; na[2] = -na[2]; }
            double pos[3] = { world[0] - 1e-3 * dir[0], world[1] - 1e-3 * dir[1], world[2] - 1e-3 * dir[2] };
            o[10] = (float)ambient_occlusion(v, p, hemi, rot, pos, na, i, j, iso, cn);
        }
    }
}

int iso_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

int iso_render(const iso_volume* v, const iso_params* p, float* out, long long stats[4], int threads)
{
    camera_t cam, nxt;
    camera_build(&cam, p->origin, p->lookat, p->up, p->fov_deg, p->width, p->height);
    camera_build(&nxt, p->last_origin, p->last_lookat, p->up, p->fov_deg, p->width, p->height);
    /* isovalue = args.isovalue * maxValue, narrowed to float by the intersector's ctor call
     * (CPURenderer.cpp:501-503,515-516) */
    double iso = (double)(float)(p->isovalue * (double)v->max_value);
    double light[3];
    if (p->light_from_camera) for (int k = 0; k < 3; ++k) light[k] = p->lookat[k] - p->origin[k];
    else for (int k = 0; k < 3; ++k) light[k] = p->light_dir[k];
    v3_normalize(light);
    static float hemi[512 * 4], rot[16 * 4];
    static int tables_ready = 0;
    if (!tables_ready) { iso_ao_tables(hemi, rot); tables_ready = 1; }
    memset(v->touched, 0, (size_t)v->bx * v->by * v->bz);
    long long hits = 0, samples = 0, steps = 0;
    const int W = p->width, H = p->height;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : hits, samples, steps)
#endif
    for (int j = 0; j < H; ++j) {
        counters_t cn = { 0, 0 };
        long long h = 0;
        for (int i = 0; i < W; ++i)
            render_pixel(v, p, &cam, &nxt, iso, light, hemi, rot, i, j, out + ((size_t)j * W + i) * 12, stats ? &cn : NULL, &h);
        hits += h; samples += cn.samples; steps += cn.steps;
    }
    if (stats) {
        long long nb = 0;
        size_t n = (size_t)v->bx * v->by * v->bz;
        for (size_t k = 0; k < n; ++k) nb += v->touched[k];
        stats[0] = hits; stats[1] = samples; stats[2] = nb; stats[3] = steps;
    }
    return 0;
}
