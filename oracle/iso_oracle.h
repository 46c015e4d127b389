/*
 * iso_oracle.h -- TEST INFRASTRUCTURE ONLY (never shipped, never on the product path).
 *
 * CPU restatement of the reference's CPU isosurface ray tracer
 * (CPURenderer/IsoVolumeRayTracer.h + the OpenVDB 6.0.1 headers it runs on).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * link or call this.
 *
 * PARITY UNPINNED: the reference ships no golden vectors for this path
 *  and its CPU renderer cannot be compiled in this image
 * without writing stand-ins for missing headers/libraries (the vendored boost
 * lacks boost/preprocessor/debug/error.hpp; OpenVDB/TBB have no .cc/.so), so
 * the restatement is checked against analytic answers only (tests/test_oracle_iso.py).
 */
#ifndef ISO_ORACLE_H
#define ISO_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct iso_volume iso_volume;

/* Dense fp32 grid, C-order [z][y][x]; value 0 == OpenVDB background / inactive voxel.
 * Mirrors a FloatGrid filled by setValue() for every non-zero voxel (no pruning), then
 * normalised as CPURenderer.cpp:448-458 does.  Dimensions must be <= 4096 per axis
 * (one level-2 node of the 5-4-3 tree).  The data are copied. Returns NULL on error. */
iso_volume* iso_volume_create(const float* dense, int nx, int ny, int nz);
iso_volume* iso_volume_create_tile(const float* dense, int nx, int ny, int nz, const int origin[3],
                                   const int gmin[3], const int gmax[3], float global_max,
                                   const int clip_lo[3], const int clip_hi[3]);
void iso_volume_free(iso_volume* v);

/* info[0..2] node-level bbox min, [3..5] node-level bbox max (IsoVolumeRayTracer.h:195-197),
 * [6..8] active-voxel bbox min, [9..11] max, [12] number of leaf bricks */
void iso_volume_info(const iso_volume* v, int info[13], double map_scale_trans[4], float* max_value);

typedef struct {
    int width, height;
    double fov_deg;
    double origin[3], lookat[3], up[3];
    double last_origin[3], last_lookat[3]; /* "next" camera of CPURenderer.cpp:717-719 */
    double isovalue;                        /* relative: multiplied by the grid max (CPURenderer.cpp:501-503) */
    double ambient[3], diffuse[3], specular[3];
    int specular_exponent;
    int light_from_camera;                  /* CPURenderer.cpp:504-507 */
    double light_dir[3];
    int viewport[4];                        /* minX,minY,maxX,maxY (render_kernel.cu:222) */
    int ao_samples;                         /* 0 -> AO == 1 (CPURenderer.cpp:736); else ray-cast AO, render_kernel.cu:109-146 */
    double ao_radius;                       /* world units of the unit-cube normalisation */
} iso_params;

/* The 512-entry cosine-hemisphere table and the 4x4 rotation table of
 * GPURendererDirect.cpp:146-189, generated with an explicit minstd_rand0 (seed 1) instead of the
 * reference's unseeded std::default_random_engine (implementation defined). hemi: [512][4], rot: [16][4]. */
void iso_ao_tables(float* hemi, float* rot);

void iso_params_default(iso_params* p);

/* Renders H*W*12 floats, HWC interleaved, channel order r,g,b,mask,nx,ny,nz,depth,fx,fy,ao,shadow
 * (values: CPURenderer.cpp:726-737; layout: render_kernel.cu:254-265).
 * stats[0] = hit pixels, stats[1] = trilinear samples taken, stats[2] = distinct 8^3 bricks
 * whose voxels were read (SURVEY.md 8(d) "N_bricks_touched"), stats[3] = voxel-DDA steps.
 * threads <= 0 -> all cores (OpenMP over image rows, the analogue of the reference's
 * tbb::parallel_for over rows, IsoVolumeRayTracer.h:494-498).  Returns 0. */
int iso_render(const iso_volume* v, const iso_params* p, float* out_hwc, long long stats[4], int threads);

/* The same frame with the arithmetic of the reference's CUDA renderer (right column): see
 * iso_oracle_gvdb.c.  isovalue is ABSOLUTE here; stats are not collected.  Returns 0. */
int iso_render_gvdb(const iso_volume* v, const iso_params* p, float* out_hwc, int threads);

int iso_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
