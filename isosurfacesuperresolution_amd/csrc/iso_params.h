// Parameter block shared by the renderer host code and the ray-march kernels.
#pragma once
#include <stdint.h>

#define ISO_BRICK 8
#define ISO_APRON_DIM 9                 // 8 voxels + 1 apron voxel on the high side
#define ISO_BRICK_VALUES 729            // 9^3
#define ISO_BRICK_STRIDE 736            // floats per stored brick (2944 B, 16-B aligned)

struct IsoCamera {
    double J[3][3];   // rows: horizontal, up, forward (TP/openvdb/math/Mat.h:758-774)
    double org[3];
    double d0[3];     // base ray direction (0,0,-1)*J
    double sw, sh;    // half frame width / height (TP/openvdb/tools/RayTracer.h:411,498)
    double V[4][4];   // inverse camera matrix, row-vector convention
};

struct IsoRenderParams {
    IsoCamera cam;
    double Vlast[4][4];          // inverse matrix of the previously rendered camera (flow)
    double s, sinv, t[3];        // index -> world: w = i*s + t
    double iso;                  // absolute isovalue, already narrowed through float
    double light[3];
    double ambient[3], diffuse[3], specular[3];
    double spec_c1;              // (e+2)/(2*pi)
    int exponent;
    int W, H;
    int vp[4];
    int nx, ny, nz;              // voxel dims
    int nbx, nby, nbz;           // 8^3 brick grid
    int n1x, n1y, n1z;           // 128^3 node grid (of the stored region)
    int org[3];                  // global index of stored voxel (0,0,0): multiple of 8; non-zero only for a tile of a larger volume
    int n1o[3];                  // global 128^3 node coordinate of node1[0][0][0] (= org >> 7)
    int bbmin[3], bbmax[3];      // node-level bbox (max already +1), GLOBAL index coordinates
    int any_leaf;
    const float* bricks;         // [slot][ISO_BRICK_STRIDE], local index (z*9+y)*9+x
    const int32_t* slot;         // [nbz][nby][nbx] -> slot or -1 (all 9^3 values zero)
    const uint8_t* leaf;         // [nbz][nby][nbx] leaf node exists
    const float* leafRange;      // [nbz][nby][nbx][2] min / max of every value a march through that leaf can read
    const uint8_t* node1;        // [n1z][n1y][n1x]
    const float* node1Range;     // [n1z][n1y][n1x][2] min / max over the ranges of the node's existing leaves
    const uint8_t* leafMarch;    // [nbz][nby][nbx] bit 0: leaf exists, bit 1: exists and its range holds the isovalue (iso_march_flags)
    const uint8_t* node1March;   // [n1z][n1y][n1x] the same per 128^3 node
    float* out;                  // [H][W][12]
    int aoSamples;               // 0 -> AO channel == 1
    double aoRadius;             // world units
    const float* aoHemi;         // [512][4] cosine-hemisphere table
    const float* aoRot;          // [16][4] per-pixel (x%4, y%4) rotation vectors
    unsigned* tileQueue;         // variant 2: 8 per-XCD tile counters, zeroed before the launch
    unsigned* resident;          // variant 2: every wave adds 1 when it starts (never reset; see iso_launch_gate)
    // cost-ordered dispatch (variant 0, at most ISO_ORDER_MAX_TILES tiles): tileCost[tile] receives the wave's clock cycles,
    // tileOrder[block] (or NULL: the XCD-aware scan order) says which tile a workgroup renders -- a permutation built from the
    // PREVIOUS frame's costs (iso_launch_tile_order), so the output is unchanged bit for bit
    unsigned* tileCost;
    const unsigned short* tileOrder;
    // exact ray-cast AO of an object-space TILED volume (iso_launch_ao_*): the render exports, per hit pixel, the AO
    // rays' origin and the viewer-facing normal in double precision, every tile then casts every pixel's AO rays against its
    // OWN leaves and writes the hit distances, the minimum over the tiles is the unsplit ray's distance
    double* hitState;            // [H][W][6] or NULL: (origin xyz, normal xyz) exactly as ambient_occlusion() receives them
};
constexpr int ISO_ORDER_MAX_TILES = 4096;

// What changes from frame to frame when only the camera moves: a launch that reads this block from DEVICE memory (isoRenderFromBlockAsync)
// is the same launch every frame -- it can be captured in a HIP graph and replayed (isoWriteFrameBlockAsync refreshes the block).
struct IsoFrameBlock {
    IsoCamera cam;
    double Vlast[4][4];
    double light[3];
};

// Per-frame constants of the `semantics=gvdb` kernel (iso_gvdb.hip), prepared in double and narrowed to float
struct IsoGvdbFrame {
    float rpos[3];               // camera position in grid-local (voxel) coordinates
    float cams[3], camu[3], camv[3];   // corner-ray basis (cuda_gvdb_geom.cuh:66-74)
    float cur[16], nxt[16];      // proj * view of the current / previously rendered camera, row-major
    float vrot[9];               // rotation rows of the view matrix (side, up, -dir)
    float scale, tr[3];          // grid-local -> world: w = scale * p + tr
    float light[3];
    float iso;                   // absolute
    float ambient[3], diffuse[3], specular[3];
    float spec_c;                // (e + 2) / (2 * 3.41)
    int exponent;
    float aoRadius;
};

// launchers (iso_kernels.hip, iso_gvdb.hip)
// waveCap: variant 2 only -- launch at most this many one-wave workgroups (0 = one per 8x8 tile)
void iso_launch_render(const IsoRenderParams& p, int variant, void* stream, void* startEvent, void* stopEvent, int waveCap);
// variant 0 without AO with the camera part of `p` taken from a device-resident block; no dispatch-packet events (graph capture)
void iso_launch_render_from_block(const IsoRenderParams& p, const IsoFrameBlock* deviceBlock, void* stream);
// *deviceDst = block, by a one-wave kernel that carries the block as its argument: ordered in `stream` like any launch, no staging buffer to race on
void iso_launch_write_block(const IsoFrameBlock& block, IsoFrameBlock* deviceDst, void* stream);
void iso_launch_render_gvdb(const IsoRenderParams& p, const IsoGvdbFrame& f, void* stream, void* startEvent, void* stopEvent);
// order[0 .. n) = a permutation of the tiles from cost[0 .. n): mode 1 heaviest first; mode 2 heaviest first for the first
// `slots` workgroups (one per SIMD), then the LIGHTEST first, so that a SIMD's second wave is light where its first is heavy
void iso_launch_tile_order(const unsigned* cost, unsigned short* order, int n, int mode, int slots, void* stream);
// dist[H][W][aoSamples] (double): distance of AO sample s of pixel (i, j) to its first hit among this volume's (tile's) leaves,
// +inf if none; pixels with gbuf mask != 1 are left untouched.  hitState: composite of the tiles' exports (P.hitState layout)
void iso_launch_ao_distances(const IsoRenderParams& p, const double* hitState, const float* gbuf, double* dist, void* stream);
// gbuf[.][10] = mean over the samples of smoothstep(1, 0, aoRadius / dist) (1 for +inf), summed in sample order as ambient_occlusion() does
void iso_launch_ao_finish(const IsoRenderParams& p, const double* dist, float* gbuf, void* stream);
// diagnostics: variant 0 with per-tile clocks and step counts, out[tiles][6] (see iso_render_stats)
void iso_launch_render_stats(const IsoRenderParams& p, int variant, long long* out, void* stream);
// One wave on `stream` that spins until *resident has reached `target` (wrap-safe) or `timeoutUs` have passed.
void iso_launch_gate(const unsigned* resident, unsigned target, int timeoutUs, void* stream);
void iso_launch_brick_flags(const float* dense, int nx, int ny, int nz, int nbx, int nby, int nbz,
                            uint8_t* flag9, uint8_t* leaf, int* bbox6, unsigned int* maxbits, void* stream);
void iso_launch_leaf_range(const float* dense, int nx, int ny, int nz, int nbx, int nby, int nbz, float* range, void* stream);
// flags[i] = exists[i] ? 1 | (range i may hold iso ? 2 : 0) : 0 -- the tables the traversal reads, per isovalue
void iso_launch_march_flags(const uint8_t* exists, const float* range, int n, double iso, uint8_t* flags, void* stream);
void iso_launch_node_range(const uint8_t* leaf, const float* leafRange, int nbx, int nby, int nbz, const int org[3],
                           int n1x, int n1y, int n1z, const int n1o[3], float* nodeRange, void* stream);
void iso_launch_scatter_tables(int n, const long long* index, const int32_t* slotv, const uint8_t* leafv, const float* rangev,
                               int32_t* slot, uint8_t* leaf, float* range, void* stream);
void iso_launch_brick_fill(const float* dense, int nx, int ny, int nz, int nbx, int nby, int nbz,
                           const int32_t* slot, float* bricks, void* stream);
