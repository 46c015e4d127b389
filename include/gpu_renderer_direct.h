/*
 * gpu_renderer_direct.h -- C-ABI of libGPURendererDirect.so (MI355X / gfx950).
 *
 * Drop-in boundary for the reference's GPURendererDirect DLL.  The first four entry points are
 * exactly what the reference's ctypes binding binds
 * (SuperresolutionNetwork/inference/renderer.py:78-110); everything below "additive" is new and
 * optional.  Plain pointers and sizes only -- no torch types cross this boundary.
 *
 * Semantics: the *values* written by render() follow the reference's CPU renderer
 * (CPURenderer/IsoVolumeRayTracer.h, CPURenderer.cpp:468-567,726-737); the *layout* and the
 * call protocol follow GPURendererDirect (render_kernel.cu:254-265).  See DESIGN.md section 2.
 */
#ifndef GPU_RENDERER_DIRECT_H
#define GPU_RENDERER_DIRECT_H

#ifdef __cplusplus
extern "C" {
#endif

/* ---- the reference's four exports -------------------------------------------------------- */

/* Replaces GPURendererDirect/GPURendererDirect.cpp:228-246.  Creates the process-global renderer
 * state on the current HIP device.  Returns 0 (or -1 if no HIP device is usable). */
int initGVDB(void);

/* Replaces GPURendererDirect.cpp:248-285.  Loads a GVDB .vbx volume: 0 ok, -1 if the name does
 * not end in ".vbx", -2 if loading fails (unreadable / malformed / truncated file, sizes that exceed the file, no
 * memory -- never an exception or abort).  The file's 8^3 bricks are uploaded as they are: memory is proportional
 * to the bricks, not to the box they span (bricks 4000 voxels apart cost two bricks).  Clears the previous volume and resets the "last
 * camera" used for the flow channels (GPURendererDirect.cpp:280-281). */
int loadGrid(const char* filename);

/* Replaces GPURendererDirect.cpp:393-428.  Commands: fov|cameraFoV (1 double),
 * cameraOrigin|cameraLookAt|cameraUp (3 doubles), resolution (2 ints), isovalue (1 double),
 * unshaded (1 int, parsed and ignored as in the reference), aosamples (1 int), aoradius (1 float),
 * viewport (4 ints minX,minY,maxX,maxY).  Values are comma separated decimals.
 * Returns 0, or -1 for an unknown command or (unlike the reference, which lets a C++ exception
 * escape) for a value of the wrong arity.
 * Additive commands: ambient|diffuse|specular (3 doubles), exponent (1 int),
 * light ("camera" or 3 doubles), and
 *   semantics ("cpu" | "gvdb"): which of the reference's two renderers the numbers follow.
 *     "cpu"  (default): CPURenderer -- node-centred sampling, voxel DDA + 5 bisections, isovalue relative to the
 *            grid maximum, world = longest edge 1.0, depth = distance, camera-space normal flipped to z >= 0,
 *            flow = camera-space delta, AO only when aosamples > 0, shadow 0.
 *     "gvdb": GPURendererDirect's CUDA kernel -- cell-centred sampling, 0.05-voxel march + 10 bisections,
 *            ABSOLUTE isovalue, world = longest edge 0.5, GVDB camera (half-width tangent tan(fov/2)/2),
 *            depth = NDC z, outward view-space normal, flow = 0.5 * delta NDC, ray-cast AO, shadow 1: the
 *            G-buffer statistics the released networks were trained on (computed in IEEE float; the texture
 *            unit's filtering is not bit-reproducible). */
int setParameter(const char* cmd, const char* value);

/* Replaces GPURendererDirect.cpp:430-446.  Synchronous.  Writes resolutionY*resolutionX*12 fp32
 * (HWC; channels r,g,b,mask,nx,ny,nz,depth,flowx,flowy,ao,shadow) to caller-owned DEVICE memory
 * and then stores the current camera as "last".  Returns seconds (launch + sync) or -1. */
float render(unsigned long long devicePtr);

/* ---- additive exports -------------------------------------------------------------------- */

/* Load a dense fp32 volume ([z][y][x], value 0 = empty) from host or device memory.
 * Same post-conditions as loadGrid.  0 ok, -2 on failure (empty volume, dims > 4096, no memory). */
int isoLoadDenseHost(const float* hostData, int nx, int ny, int nz);
int isoLoadDenseDevice(unsigned long long devicePtr, int nx, int ny, int nz);

/* Load one tile of a larger volume (multi-GPU object-space decomposition): the dense
 * data are the tile plus a halo (>= 2 voxels below and 3 above the owned region, or up to the volume's
 * border), `origin` is the global index of local voxel (0,0,0) and a multiple of 8.  World normalisation,
 * isovalue scale and the box rays are clipped to come from the GLOBAL active bbox / maximum; the ray is
 * walked in global index coordinates exactly as in the unsplit volume and the tile processes only the
 * leaves inside [clipLo, clipHi) (global index coordinates, clipLo a multiple of 8), so the pixels it
 * produces are bit for bit those of the unsplit render wherever its leaves hold the first crossing, and
 * the nearest-hit composite of all tiles equals the unsplit image.  An all-zero tile is valid.  Ray-cast
 * AO inside render() sees only the tile's own leaves: render with aosamples=0 and use the exact tiled-AO passes
 * (isoSetHitStateBuffer / isoAoDistancesAsync / isoAoFinishAsync below).  semantics=gvdb renders tiles too (round 4: the brick DDA
 * covers the global box, a tile marches the bricks it owns, aosamples=0), but its nearest-DEPTH composite is only approximately the
 * unsplit image: that renderer's hit is the outside end of a bisection that begins one 0.05-voxel step before the first sample >= iso,
 * so a brick whose first sample is already inside the surface reports a hit in FRONT of its entry, possibly in front of the hit of
 * the previous brick -- and where those two bricks belong to different tiles the smaller depth wins instead of the earlier brick
 * (hit mask exact; 0 .. 0.1 % of the pixels off by <= 0.02 in tests/test_render_gpu.py).  Exact would mean selecting by brick order.
 * 0 ok, -2 on failure (misaligned origin / clipLo, region outside the stored data, no memory). */
int isoLoadDenseTileHost(const float* hostData, int nx, int ny, int nz, const int origin[3],
                         const int globalActiveMin[3], const int globalActiveMax[3], float globalMax,
                         const int clipLo[3], const int clipHi[3]);
/* ... the same from DEVICE memory (a tile that a solver or an earlier pass left on the GPU: no staging through the host). */
int isoLoadDenseTileDevice(unsigned long long devicePtr, int nx, int ny, int nz, const int origin[3],
                           const int globalActiveMin[3], const int globalActiveMax[3], float globalMax,
                           const int clipLo[3], const int clipHi[3]);

/* Launch the frame on `stream` (a hipStream_t, may be NULL) without synchronising; the "last
 * camera" bookkeeping is identical to render().  Returns 0 or -1. */
int isoRenderAsync(unsigned long long devicePtr, void* stream);

/* The frame as a REPLAYABLE launch (additive; pipeline.py's frame graph).  What changes when only the camera moves -- the camera, the
 * previous camera (flow reference) and the light -- lives in a device block of isoFrameBlockBytes() bytes (8-byte aligned):
 *   isoWriteFrameBlockAsync(block, stream): fills it from the CURRENT parameters (setParameter) by a one-wave kernel on `stream`
 *     (ordered like any launch; no host staging buffer) and makes the current camera the "last camera", exactly as render() does;
 *   isoRenderFromBlockAsync(devicePtr, block, stream): the default-semantics render without ambient occlusion (the SR-mode call of
 *     mainGUI.py:690-701: aosamples = 0) whose kernel READS that block; everything else (volume, resolution, viewport, isovalue,
 *     material) is taken from the current parameters at the time of the call.  The launch is identical from frame to frame, so a
 *     stream capture that contains it can be replayed with a refreshed block.  Does not touch the "last camera".
 * Pixels are bit for bit those of isoRenderAsync with the same parameters.  0 ok, -1 not available (no volume, semantics=gvdb,
 * aosamples > 0, misaligned block). */
int isoFrameBlockBytes(void);
int isoWriteFrameBlockAsync(unsigned long long deviceBlock, void* stream);
int isoRenderFromBlockAsync(unsigned long long devicePtr, unsigned long long deviceBlock, void* stream);

/* info: [0..2] volume dims, [3] stored bricks, [4] leaf bricks, [5..7] node bbox min,
 * [8..10] node bbox max, [11] bytes of brick storage (MiB), out_max = grid max value. */
int isoGetVolumeInfo(int info[12], float* out_max);

/* Kernel variant (all bit-identical): 0 = one lane per ray, the traversal of IsoVolumeRayTracer.h:37-114 as one flat
 * per-lane state machine marching two voxel boundaries per iteration (default); 1 = wave-cooperative LDS brick cache
 * (nested loops); 2 = the flat traversal, one sample per iteration, in a 128-register budget with a capped, tile-pulling
 * grid (rendering under another kernel, see isoSetWaveCap); 3 = variant 0 with the slot table of a <= 256^3 volume in
 * LDS (falls back to 0 for larger volumes); 4 = the reference's four nested loops in lock step (what 0 replaced: 2.3x
 * slower); 5 = the flat traversal with one sample per iteration.  Returns 0, -1 for an unknown variant. */
int isoSetKernelVariant(int variant);

/* Optional per-frame kernel timing for benchmarks: while enabled, each ray-march dispatch carries a
 * start/stop event pair on its own packet.  isoProfileGet(i, &ms) after synchronising. */
int isoProfileEnable(int on);
int isoProfileCount(void);
int isoProfileGet(int i, float* ms);

/* Variant 2 only: launch at most `waves` one-wave workgroups, which pull the 8x8 pixel tiles from per-XCD queues
 * (0 = one per tile, the default).  4 x the CU count keeps one ray-march wave per SIMD, which is what
 * lets the next SR conv workgroup land on every CU while a frame renders on a side stream.
 * Returns 0, or -1 for a negative cap. */
int isoSetWaveCap(int waves);
/* Additive: cost-ordered dispatch of the default kernel (variant 0) for images of at most 4096 tiles of 8x8 pixels.  The waves
 * of a frame record their clock cycles, one workgroup sorts the tiles on the render's stream, and the NEXT frame dispatches its
 * tiles in that order: 1 = heaviest first, 2 = heaviest first for the first 4 x #CUs workgroups, then the lightest (a SIMD's
 * second wave is light where its first is heavy).  A pure permutation of the work: the G-buffer does not change.  0 = off
 * (default: the XCD-aware scan order).  Returns 0, or -1 for an unknown mode. */
int isoSetTileOrderMode(int mode);

/* Additive: EXACT ray-cast ambient occlusion for a volume split into tiles (isoLoadDenseTileHost).  An AO ray ends at its first
 * hit anywhere in the volume (render_kernel.cu:109-146: computeAmbientOcclusion calls rayCast without a range), so no halo makes a
 * tile's own AO right.  Instead, per frame:
 *   1. every tile renders with aosamples = 0 and isoSetHitStateBuffer(ptr): besides the G-buffer it exports, per pixel it hits,
 *      the AO rays' origin and the viewer-facing normal in double precision ([H][W][6] doubles, caller-owned device memory;
 *      0 switches the export off);
 *   2. the tiles' G-buffers AND hit states are composited by nearest hit (the winner's values are the unsplit pixel's);
 *   3. with aosamples = n set, every tile runs isoAoDistancesAsync(hitState, gbuf, dist): each hit pixel's n rays against the
 *      tile's OWN leaves, dist[H][W][n] doubles = distance to the first hit there or +inf;
 *   4. the element-wise MINIMUM of the tiles' dist arrays (one all-reduce) is each ray's distance in the unsplit volume;
 *   5. isoAoFinishAsync(dist, gbuf) writes channel 10 = mean of smoothstep(1, 0, aoradius / dist) in sample order.
 * Same frame, directions, traversal and arithmetic as the unsplit kernel: the result equals it bit for bit.  0 ok, -1 on
 * missing volume / null pointers / aosamples = 0 / semantics = gvdb. */
int isoSetHitStateBuffer(unsigned long long devicePtr);
int isoAoDistancesAsync(unsigned long long hitStatePtr, unsigned long long gbufPtr, unsigned long long distPtr, void* stream);
int isoAoFinishAsync(unsigned long long distPtr, unsigned long long gbufPtr, void* stream);

/* Enqueues a one-wave kernel on `stream` that returns once every wave of the most recently launched variant-2
 * render has started (or after `timeoutUs`).  Put on the stream of the SR network right after the render was
 * enqueued on its side stream, it makes the ray-march waves land on an idle GPU -- one per SIMD -- instead of racing
 * the network's next kernel for slots.  Returns 0, -1 before initGVDB. */
int isoGateResident(void* stream, int timeoutUs);

/* Overrides the "last camera" the flow channels (8, 9) are measured against (render() / isoRenderAsync() set it to the
 * camera they rendered; GPURendererDirect.cpp:440-442).  For callers that rendered a frame ahead and then discard it:
 * the flow reference must go back to the last frame that was actually displayed.  0 ok, -1 on null pointers. */
int isoSetLastCamera(const double origin[3], const double lookAt[3]);

/* Host-only helpers around the .vbx reader (no GPU needed): volume dims [x,y,z] of the dense box
 * spanned by the stored bricks, and the dense fp32 data [z][y][x] itself. 0 ok, -2 on failure. */
int isoVbxInfo(const char* path, int dims[3]);
int isoVbxReadDense(const char* path, float* hostOut);

/* Release all device memory held by the renderer. */
void isoShutdown(void);

#ifdef __cplusplus
}
#endif
#endif
