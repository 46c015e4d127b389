// The 1080p TAIL of EnhanceNet in two launches instead of three kernels and two 531 MB round trips:
//     postblock.6 (conv3x3 64 -> 64 + ReLU)  ->  postblock.8 (conv3x3 64 -> 6)  ->  residual reconstruction, clamp /
//     normalise, screen-space shading            (SuperresolutionNetwork/models/enhancenet.py:119-125,51-90; mainGUI.py:594-603)
//
// The 64-channel output y6 of postblock.6 never goes to memory.  The last convolution is linear in y6, so it is taken apart
// by TAP:   out[c][q] = b[c] + sum_t z[t][c][q + d_t],     z[t][c][p] = sum_k W8[c][k][t] * y6[k][p]      (t = 3 dy + dx, d_t = (dy - 1, dx - 1))
// -- z is a 1x1 convolution 64 -> 54 of y6, per pixel, with NO halo: it is computed where y6 is, in the registers of the
// wave that has just finished the pixel's 64 accumulators.  The MFMA D layout (lane = pixel, 16 registers = 16 channels)
// IS a B-operand layout for a following MFMA whose K index runs over those channels, provided the A operand (the
// re-laid weights of the last layer) uses the same K order: the ReLU'd values are split into (hi, lo') fp16 pairs in
// registers and 48 further MFMAs per wave (11 % on top of the layer's 432) produce the 54 tap-partials on the same
// split-operand arithmetic (three products, fp32 accumulation) as every other layer.  Only z (54 fp32 planes) is
// written; a second, streaming kernel adds each pixel's nine shifted partials in a FIXED order (bit-reproducible and
// independent of the tiling) and finishes the frame.  Every z element is read exactly once.
//
// Against the three-kernel tail (conv3x3_split_stream_kernel, conv3x3_small_cout_kernel with its fused finish): the
// fp32 4x4x1-MFMA kernel (0.20 ms, 72 TFLOP/s) and its 606 MB read are gone, the 531 MB write of y6 becomes 448 MB of z.
#include "sr_diag.h"
#include <cstdlib>
#include "sr_split_common.h"

namespace {

constexpr int TZ_ROWS = 54;                                                  // 9 taps x 6 output channels
constexpr int TZ_UNITS = 4 * 2 * 2 * 64;                                     // [k-step q][part][lane half][row m (64, >= 54 zero)] 16-byte units
constexpr int SQ_QPR = (ST_W + 8) / 4;                                       // 10 quads per patch row
constexpr int SQ_UNITS = 2 * SP_H * SQ_QPR;                                  // 200 (channel group, patch row, quad) units per k-step
constexpr int SQ_SLOT = 2 * SP_PIX;                                          // 16-byte units of one k-step slot of the hi (or lo) patch
constexpr int T_LDS_BYTES = S_LDS_BYTES + 256;                               // + postblock.6's bias: 80 640 B, two workgroups per CU

struct TailParams {
    SplitConvParams c;           // postblock.6: x, wq, bias, H, W, ... (y / residual unused)
    const u32x4* wz;             // header {2^S, 2^-S, S, 0} + TZ_UNITS units: the last layer's weights by (tap, channel) row
    float* z;                    // two-kernel form: [54][zPlane] fp32 tap-partials
    int zPlane;
    // fused form (FUSED = true): pixels whose nine partials all lie in their own tile (or outside the image) are finished here;
    // the others -- the SEAM pixels on the tiles' rims -- get their partials from up to four tiles, each of which writes the
    // ones it owns into the pixel's record [9 taps][6 channels]; tail_seam_finish_kernel adds them in the same tap order
    FinishParams fin;
    const float* bias8;
    float* srec;                 // S form: [H][tilesX][6 kinds][18 groups] addends of the rows' end pixels
    float* rowrec;               // records of the pixels with Y % 8 in {0, 7}: [2 tilesY][W][54]
    float* colrec;               // records of the other pixels with X % 32 in {0, 31}: [H][2 tilesX][54]
    // PSIN = true: the input arrives PACKED-SPLIT (SplitConvParams::ps of the producing layer): xps[hi | lo][8 groups][xpsPlane
    // units]; k-steps are staged by LDS-DMA, 22 wave-wide 1 KB pieces each, pixels outside the image from `zero` (16 zero bytes)
    const u32x4* xps; int xpsPlane;
    const u32x4* zero;
};

// Row of the z product that holds tap t = 3 dy + dx of output channel c: [dx][dy][c], so that the three horizontal taps of one (dy, c)
// are 18 rows apart (the S form below adds them with constant offsets).  tail_prepare_kernel lays the weights out accordingly.
__host__ __device__ __forceinline__ constexpr int tail_zrow(int t, int c) { return (t % 3) * 18 + (t / 3) * 6 + c; }
constexpr int TS_GROUPS = 18;                                                // (dy, c) pairs: the planes of the S form
constexpr int TS_STRIDE = ST_W + 2;                                          // slab row: pixels -1 .. 32 of the wave's row (the two pad columns are zero)
constexpr int TS_REC = 6 * TS_GROUPS;                                        // floats of one (image row, tile) edge record: kinds A .. F x 18 groups

__device__ __forceinline__ bool tail_in_image(int x, int y, int W, int H) { return (unsigned)x < (unsigned)W && (unsigned)y < (unsigned)H; }

// a pixel is finished by the tile that owns it iff none of its nine taps reads an in-image pixel of another tile
__device__ __forceinline__ bool tail_is_seam(int x, int y, int W, int H)
{
    const int tx0 = x & ~(ST_W - 1), ty0 = y & ~(ST_H - 1);
    bool seam = false;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int px = x + t % 3 - 1, py = y + t / 3 - 1;
        seam = seam || (tail_in_image(px, py, W, H) && !(px >= tx0 && px < tx0 + ST_W && py >= ty0 && py < ty0 + ST_H));
    }
    return seam;
}

__device__ __forceinline__ float* tail_record(const TailParams& tp, int x, int y)
{
    const int W = tp.c.W;
    if ((y & 7) == 0 || (y & 7) == 7)
        return tp.rowrec + ((size_t)((y >> 3) * 2 + ((y & 7) == 7 ? 1 : 0)) * W + x) * TZ_ROWS;
    return tp.colrec + ((size_t)y * (2 * tp.c.tilesX) + (x >> 5) * 2 + ((x & 31) == 31 ? 1 : 0)) * TZ_ROWS;
}

// The streaming split-operand convolution of sr_conv_split.hip (persistent workgroups, 8 x 32-pixel x 64-channel tiles,
// next k-step's operands in flight under the MFMAs) with the z stage between its last k-step and its epilogue.
constexpr int ZW_OFF = S_WUNITS - TZ_UNITS;                                  // the z weights sit at the END of the weight buffer: the fused
                                                                             // form's z tile (54 x 8 x 32 fp32 = 55 296 B) grows from the patch into its start
// FORM 0: the 54 tap-partial planes go to memory.  FORM 1 (= the old FUSED): a tile's partials are combined in LDS.  FORM 2, the
// default: each wave adds the three HORIZONTAL taps of every (dy, c) for its own row of 32 pixels through its LDS slab (whose two
// pad columns are zero) and stores 18 S planes instead of 54 z planes; the two pixels at the row's ends, whose sums need a value
// of the neighbouring tile, get their three addends from small per-(row, tile) records and are added -- in the SAME order
// (z[dx=0] + z[dx=1]) + z[dx=2] -- by the finishing kernel: every pixel's arithmetic is independent of where tile borders fall.
template <int FORM, bool PSIN>
__global__ __launch_bounds__(S_THREADS, 2) void conv3x3_split_tail_kernel(const TailParams tp)
{
    constexpr bool FUSED = FORM == 1;
    constexpr bool SFORM = FORM == 2 || FORM == 4;                           // horizontal tap sums through the wave's slab (4: + the vertical sums of the tile's inner rows)
    const SplitConvParams& p = tp.c;
    extern __shared__ u32x4 patch[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    u32x4* wbuf = patch + S_PUNITS;
    const int ntiles = p.tilesY * p.tilesX;
    const int xcd = blockIdx.x & 7, jw = blockIdx.x >> 3;
    const int njw = ((int)gridDim.x >> 3) + (xcd < ((int)gridDim.x & 7) ? 1 : 0);
    const int tq = ntiles >> 3, trm = ntiles & 7;
    const int tstart = xcd < trm ? xcd * (tq + 1) : trm * (tq + 1) + (xcd - trm) * tq;
    const int tcount = tq + (xcd < trm ? 1 : 0);
    if (jw >= tcount) return;
    const unsigned planeBytes = (unsigned)p.xPlane * 4u;

    struct Tile { int oy0, ox0; };
    // Tile list order: BANDS of `band` tile rows walked column by column, so that the ~64 workgroups of an XCD, which take consecutive
    // list entries at any time, cover a 2-D block of tiles (16 columns x 4 rows) and find each other's halo rows and columns in their
    // L2 instead of fetching them again (row-major order: a tile row of 60 tiles is 5 MB of patches, the row above has left the 4 MB L2
    // by the time the row below wants its two shared image rows).  p.dbg bit 7: the row-major order (A/B runs).
    const int band = (p.dbg & 128) ? 1 : 4;
    auto decode = [&](int t) {
        const int b = tstart + t;
        const int per = band * p.tilesX, bi = b / per, r0 = b - bi * per;
        const int rows = min(band, p.tilesY - bi * band);
        Tile r;
        r.ox0 = (r0 / rows) * ST_W;
        r.oy0 = (bi * band + r0 % rows) * ST_H;
        return r;
    };
    const bool staging = tid < SQ_UNITS;
    const int ug = tid / (SP_H * SQ_QPR), urem = tid - ug * (SP_H * SQ_QPR);
    const int ur = urem / SQ_QPR, uq = urem - ur * SQ_QPR;
    const rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, PSIN ? 0 : (int)((size_t)64 * p.xPlane * 4), 0x00020000);
    u32x4 v[PSIN ? 1 : 8];
    // packed-split input: k-step ks of tile t into slot `slot` -- per part (hi, lo) 680 units [2 groups][10 x 34 pixels] = 11 pieces.
    // LDS-DMA through a BUFFER DESCRIPTOR (buffer_load_dwordx4 ... lds): the lane's 32-bit offset inside the patch is loop invariant (one
    // register per piece), the tile's and k-step's position goes into the scalar offset, and a lane whose pixel lies outside the image
    // gets the out-of-range offset, for which the hardware writes ZEROS into LDS (tools/probes/buffer_lds_oob_probe.hip) -- the
    // convolution's zero padding without a zero unit.  Round 3 used global_load_lds with a 64-bit address per lane and piece: the six
    // pointer pairs were spilled, and every reload ended in an `s_waitcnt vmcnt(0)` that waited for the PREVIOUS pieces' DMA to land.
    // The descriptor starts one image row + one pixel BEFORE the tensor, so that the patch's halo offsets are never negative (the lanes
    // that would read in front of the tensor are exactly the out-of-image ones, which are masked).
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    unsigned pvoff[PSIN ? 6 : 1], prc[PSIN ? 6 : 1];
    rsrc_t xprs = xrs;
    if constexpr (PSIN) {
        xprs = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(tp.xps) - (p.W + 1), 0, (int)(((size_t)2 * 8 * tp.xpsPlane + p.W + 1) * 16), 0x00020000);
#pragma unroll
        for (int d = 0; d < 6; ++d) {
            const int piece = wv + 4 * d;
            const int pc = piece % 11, off = pc * 64 + lane;
            const int gg = off / SP_PIX, pix = off - gg * SP_PIX;
            const int r = pix / SP_W, c = pix - r * SP_W;
            pvoff[d] = (piece < 22 && off < SQ_SLOT) ? ((unsigned)gg * (unsigned)tp.xpsPlane + (unsigned)(r * p.W + c)) * 16u : BAD_OFFSET;
            prc[d] = (unsigned)r | ((unsigned)c << 8);
        }
    }
    auto dma_kstep = [&](const Tile& t, int ks, int slot) {
        if constexpr (PSIN) {
            const bool interior = t.oy0 >= 1 && t.oy0 + ST_H + 1 <= p.H && t.ox0 >= 1 && t.ox0 + ST_W + 1 <= p.W;
#pragma unroll
            for (int d = 0; d < 6; ++d) {
                const int piece = wv + 4 * d;
                if (piece >= 22) continue;
                const int part = piece / 11, pc = piece - part * 11;
                unsigned vo = pvoff[d];
                if (!interior) {
                    const int iy = t.oy0 + (int)(prc[d] & 255u) - 1, ix = t.ox0 + (int)(prc[d] >> 8) - 1;
                    vo = ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) ? vo : BAD_OFFSET;
                }
                const unsigned soff = ((unsigned)(part * 8 + 2 * ks) * (unsigned)tp.xpsPlane + (unsigned)(t.oy0 * p.W + t.ox0)) * 16u;
                if (pc * 64 + lane < SQ_SLOT)        // (the last piece is 40 units: the other lanes must not write behind the slot)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(xprs, (isr_lvoid_t*)(patch + part * S_PART + slot * SQ_SLOT + pc * 64), 16, (int)vo, (int)soff, 0, 0);
            }
        }
    };
    // stage(t, ks, slot): start moving k-step ks of tile t towards patch slot `slot` -- fp32 input: loads into registers, parked
    // (split, written to LDS) later by park_loads(slot); packed-split input: LDS-DMA straight into the slot
    auto issue_loads = [&](const Tile& t, int ks, int slot) {
        if constexpr (PSIN) {
            dma_kstep(t, ks, slot);
        } else {
            const int iy = t.oy0 + ur - 1, ix = t.ox0 - 4 + 4 * uq;
            const bool ok = staging && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const unsigned base = (unsigned)(ks * 16 + ug * 8) * planeBytes + (unsigned)(iy * p.W + ix) * 4u;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                v[e] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)(ok ? base : BAD_OFFSET), (int)((unsigned)e * planeBytes), 0);
        }
    };
    auto park_loads = [&](int slot) {
        if constexpr (!PSIN) {
            if (!staging) return;
            f16x8 h0, h1, h2, h3, l0, l1, l2, l3;
    #pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float4 f = __builtin_bit_cast(float4, v[e]);
                _Float16 a, b;
                split16x(f.x, a, b); h0[e] = a; l0[e] = b;
                split16x(f.y, a, b); h1[e] = a; l1[e] = b;
                split16x(f.z, a, b); h2[e] = a; l2[e] = b;
                split16x(f.w, a, b); h3[e] = a; l3[e] = b;
            }
            u32x4* dst = patch + slot * SQ_SLOT + ug * SP_PIX + ur * SP_W + 4 * uq - 3;
            if (uq > 0) { dst[0] = __builtin_bit_cast(u32x4, h0); dst[S_PART] = __builtin_bit_cast(u32x4, l0); }
            if (uq > 0 && uq < SQ_QPR - 1) {
                dst[1] = __builtin_bit_cast(u32x4, h1); dst[S_PART + 1] = __builtin_bit_cast(u32x4, l1);
                dst[2] = __builtin_bit_cast(u32x4, h2); dst[S_PART + 2] = __builtin_bit_cast(u32x4, l2);
            }
            if (uq < SQ_QPR - 1) { dst[3] = __builtin_bit_cast(u32x4, h3); dst[S_PART + 3] = __builtin_bit_cast(u32x4, l3); }
        }
    };
    // weights of one k-step: thread t moves unit (tap i, part t / 128, t % 128), i = 0..8: byte 16 t + 16384 i + 4096 ks of the image
    const rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(p.wq + 1), 0, 9 * 4 * 256 * 16, 0x00020000);
    u32x4 wreg[9];
    auto wfetch = [&](int ks) {
#pragma unroll
        for (int i = 0; i < 9; ++i) wreg[i] = __builtin_amdgcn_raw_buffer_load_b128(wrs, tid * 16, i * 16384 + ks * 4096, 0);
    };
    u32x4* wdst = wbuf + (tid >> 7) * S_WPART + (tid & 127);
    auto wpark = [&]() {
#pragma unroll
        for (int i = 0; i < 9; ++i) wdst[i * 128] = wreg[i];
    };
    // constants of the z stage: bias of postblock.6 (64 floats in LDS behind the weight buffer), both output scales
    const float unscale = reinterpret_cast<const float*>(p.wq)[1];
    const float zunscale = reinterpret_cast<const float*>(tp.wz)[1];
    float* bias_lds = reinterpret_cast<float*>(wbuf + S_WUNITS);
    if (tid < 64) bias_lds[tid] = p.bias ? p.bias[tid] : 0.0f;
    const rsrc_t zrs = __builtin_amdgcn_make_buffer_rsrc(tp.z, 0, (int)((size_t)(SFORM ? TS_GROUPS : TZ_ROWS) * tp.zPlane * 4), 0x00020000);
    const rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(tp.srec, 0, SFORM ? (int)((size_t)p.H * p.tilesX * TS_REC * 4) : 0, 0x00020000);
    u32x4 zw[4];                                                             // this thread's 4 of the 1024 units of the z weights

    Tile cur = decode(jw);
    issue_loads(cur, 0, 0);
    wfetch(0);
    park_loads(0);
    wpark();
    __syncthreads();
    int slot = 0;
    for (int t = jw; t < tcount; t += njw) {
        const bool more = t + njw < tcount;
        Tile nxt = cur;
        if (more) nxt = decode(t + njw);
        f32x16 acc[2][2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[cb][r][i] = 0.0f;
#pragma unroll 1
        for (int ks = 0; ks < 3; ++ks) {
            issue_loads(cur, ks + 1, slot ^ 1);
            wfetch(ks + 1);
            split_kstep(acc, wbuf + h * 64 + j, patch + slot * SQ_SLOT + h * SP_PIX + (wave * 2) * SP_W + j, true);
            __syncthreads();
            park_loads(slot ^ 1);
            wpark();
            __syncthreads();
            slot ^= 1;
        }
        if (more) {
            issue_loads(nxt, 0, slot ^ 1);
            if (FORM != 4) wfetch(0);                                        // (V form: requested behind the z stage -- 36 registers the epilogue needs, see below)
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) zw[i] = tp.wz[1 + tid + i * S_THREADS];   // in flight under the last k-step
        split_kstep(acc, wbuf + h * 64 + j, patch + slot * SQ_SLOT + h * SP_PIX + (wave * 2) * SP_W + j, true);
        __syncthreads();                                                     // weight buffer and patch idle
#pragma unroll
        for (int i = 0; i < 4; ++i) wbuf[ZW_OFF + tid + i * S_THREADS] = zw[i];
        __syncthreads();
        // ---- z stage, one output row at a time: y6 = relu(acc 2^-S + bias) as (hi, lo') B fragments straight from the D
        // layout.  k-step q of the z product covers y6 channels 32 (q >> 1) + 16 (q & 1) + (e & 3) + 8 (e >> 2) + 4 h, e = 0..7:
        // registers 8 (q & 1) .. + 7 of acc[q >> 1][r]; the prepared weights use the same order.
        // transposition slab of this wave (8 KB).  With packed-split input the next tile's first k-step is landing in slot ^ 1 by DMA
        // meanwhile: the slabs then live in the two halves of the slot just multiplied and in the front of the weight buffer
        unsigned ymag = 0u;
        float sreg[2][9];                                                    // V form: S[g = 9 h + k] of this lane's pixel, both rows of the wave
        float* tr = PSIN ? reinterpret_cast<float*>(wave == 0 ? patch + slot * SQ_SLOT : wave == 1 ? patch + S_PART + slot * SQ_SLOT
                                                              : wbuf + (wave - 2) * 512)
                         : reinterpret_cast<float*>(patch) + wave * (64 * 32);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            f16x8 zh[4], zl[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 b0 = *reinterpret_cast<const float4*>(bias_lds + 32 * (q >> 1) + 16 * (q & 1) + 4 * h);
                const float4 b1 = *reinterpret_cast<const float4*>(bias_lds + 32 * (q >> 1) + 16 * (q & 1) + 8 + 4 * h);
                const float bq[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float val = acc[q >> 1][r][8 * (q & 1) + e] * unscale + bq[e];
                    val = val > 0.f ? val : 0.f;
                    _Float16 a, b;
                    split16x(val, a, b);
                    zh[q][e] = a; zl[q][e] = b;
                    ymag = isr_umax(ymag, isr_mag(val));
                }
            }
            const int oy = cur.oy0 + wave * 2 + r;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                f32x16 zacc;
#pragma unroll
                for (int i = 0; i < 16; ++i) zacc[i] = 0.0f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f16x8 ah = __builtin_bit_cast(f16x8, wbuf[ZW_OFF + ((q * 2 + 0) * 2 + h) * 64 + mb * 32 + j]);
                    const f16x8 al = __builtin_bit_cast(f16x8, wbuf[ZW_OFF + ((q * 2 + 1) * 2 + h) * 64 + mb * 32 + j]);
                    const f16x8 as = ah * (_Float16)0.00048828125f;          // w_hi 2^-11: partner of the scaled lo'
                    zacc = mfma16(al, zh[q], zacc);
                    zacc = mfma16(as, zl[q], zacc);
                    zacc = mfma16(ah, zh[q], zacc);
                }
                // D row (z row within the block) = (i & 3) + 8 (i >> 2) + 4 h, column = pixel j
                if (FUSED) {                                                 // ... into the workgroup's z tile [54][8 rows][32 pixels]
                    float* zt = reinterpret_cast<float*>(patch);
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int m = mb * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                        if (m < TZ_ROWS) zt[(m * ST_H + wave * 2 + r) * ST_W + j] = zacc[i] * zunscale;
                    }
                } else if (SFORM) {                                          // ... into this wave's slab [row][pixel -1 .. 32]; rows >= 56 do not exist
                    const float zs = (cur.ox0 + j < p.W) ? zunscale : 0.0f; // a pixel beyond the image's right edge contributes nothing to its neighbour
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        if (mb == 0 || (i >> 2) < 3) tr[(mb * 32 + (i & 3) + 8 * (i >> 2) + 4 * h) * TS_STRIDE + j + 1] = zacc[i] * zs;
                } else {                                                     // ... into this wave's transposition slab
#pragma unroll
                    for (int i = 0; i < 16; ++i) tr[(mb * 32 + (i & 3) + 8 * (i >> 2) + 4 * h) * 32 + j] = zacc[i] * zunscale;
                }
            }
            if (FUSED) continue;
            if (SFORM) {
                if (lane < TZ_ROWS) { tr[lane * TS_STRIDE] = 0.0f; tr[lane * TS_STRIDE + TS_STRIDE - 1] = 0.0f; }      // the pad columns
                __builtin_amdgcn_s_waitcnt(0xC07F);                          // lgkmcnt(0): same-wave hand-off through LDS
                // lane (j, h) adds groups g = 9 h + k, k = 0 .. 8: S[g][pixel j] = (z[dx 0][j - 1] + z[dx 1][j]) + z[dx 2][j + 1]
                const float* sb = tr + (9 * h) * TS_STRIDE + j;
                const bool live = oy < p.H && cur.ox0 + j < p.W;
                const unsigned sv = live ? ((unsigned)(9 * h) * (unsigned)tp.zPlane + (unsigned)(oy * p.W + cur.ox0 + j)) * 4u : BAD_OFFSET;
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    const float sum = (sb[k * TS_STRIDE] + sb[(TS_GROUPS + k) * TS_STRIDE + 1]) + sb[(2 * TS_GROUPS + k) * TS_STRIDE + 2];
                    if (FORM == 4) sreg[r][k] = sum;                         // V form: kept for the vertical sums below
                    else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, sum), zrs, (int)sv, k * tp.zPlane * 4, 0);
                }
                // the addends of the two end pixels (and of the neighbours' end pixels), 6 x 18 floats per (row, tile):
                //   A z[dx 0] @ 30, B z[dx 1] @ 31, C z[dx 0] @ 31, D z[dx 1] @ 0, E z[dx 2] @ 1, F z[dx 2] @ 0   (pixel of this row)
#pragma unroll
                for (int rr = 0; rr < 2; ++rr) {
                    const int idx = lane + 64 * rr;
                    const int kind = idx / TS_GROUPS, g = idx - kind * TS_GROUPS;
                    const int zdx = kind == 0 || kind == 2 ? 0 : (kind == 1 || kind == 3 ? 1 : 2);
                    const int col = kind == 0 ? 31 : (kind == 1 || kind == 2) ? 32 : (kind == 4 ? 2 : 1);        // slab column = pixel + 1
                    const float val = idx < TS_REC ? tr[(zdx * TS_GROUPS + g) * TS_STRIDE + col] : 0.0f;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), rrs,
                                                          (int)((idx < TS_REC && oy < p.H) ? ((unsigned)(oy * p.tilesX + cur.ox0 / ST_W) * TS_REC + (unsigned)idx) * 4u : BAD_OFFSET), 0, 0);
                }
                __builtin_amdgcn_s_waitcnt(0xC07F);                          // reads done before the next row overwrites the slab
                continue;
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);                              // lgkmcnt(0): same-wave hand-off through LDS
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int qq = lane + 64 * k;                                // float4 index: z row = qq / 8, pixel group = qq % 8
                const int m = qq >> 3, px = cur.ox0 + (qq & 7) * 4;
                const bool ok = oy < p.H && px < p.W && m < ((p.dbg & 64) ? 18 : TZ_ROWS);   // (dbg 64: timing experiment, a third of the planes)
                const float4 val = reinterpret_cast<const float4*>(tr)[qq];
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, val), zrs,
                                                       (int)(ok ? ((unsigned)m * (unsigned)tp.zPlane + (unsigned)(oy * p.W + px)) * 4u : BAD_OFFSET), 0, 0);
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);                              // reads done before the next row overwrites the slab
        }
        isr_range_note(p.absmax, ymag);
        if (FORM == 4) {
            // ---- V form: the vertical sums of the tile's rows where all three source rows lie in the tile.
            // Every wave leaves the S values of its two rows in its (now dead) slab, [row][group][pixel]; after a barrier lane (j, h) takes
            // the pixel j of tile row 2 wave + h and adds in the finishing kernel's order, ((bias + S[dy 0] @ Y - 1) + S[dy 1] @ Y) + S[dy 2] @ Y + 1:
            //   rows 1 .. 6: the whole sum -> plane c (the finishing kernel only finishes the pixel);
            //   row 7:       (bias + S0 @ 6) + S1 @ 7 -> plane c, the finishing kernel adds S2 of the tile below's row 0;
            //   row 0:       S1 @ 0 -> plane c, S2 @ 1 -> plane 6 + c (row = tile row), the finishing kernel starts from bias + S0 of the tile above's row 7;
            //   exports for the neighbours: S2 @ 0 and S0 @ 7 -> plane 12 + c, rows 2 ty and 2 ty + 1.
            // Per pixel the same additions in the same order as the S form: bit-identical, independent of where tile borders fall.  6 + 3 / 8 x 12 plane rows
            // per image row instead of 18; the first / last pixel of a tile's 32 still come from the end-pixel records (their S needs the neighbour tile).
            if (more) wfetch(0);                                             // the next tile's first weights travel under the exchange
            int jv = j, hv = h;                                              // (laundered: hoisted out of the tile loop the ~60 slab addresses below spill)
            asm volatile("" : "+v"(jv), "+v"(hv));
            {
                float* sw = tr + (9 * hv) * 32 + jv;
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int k = 0; k < 9; ++k) sw[(r * TS_GROUPS + k) * 32] = sreg[r][k];
            }
            __syncthreads();
            __builtin_amdgcn_sched_barrier(0);
            auto slab_of = [&](int w) -> const float* {
                return PSIN ? reinterpret_cast<const float*>(w == 0 ? patch + slot * SQ_SLOT : w == 1 ? patch + S_PART + slot * SQ_SLOT : wbuf + (w - 2) * 512)
                            : reinterpret_cast<const float*>(patch) + w * (64 * 32);
            };
            // lane (jv, hv): pixel jv, channels 3 hv .. 3 hv + 2, the wave's two rows one after the other -- row index, branches and slab bases are
            // wave-uniform (a lane-dependent row made every read a four-way select and every wave run all three row kinds)
            float b3[3];                                                     // the last layer's bias opens the sums
#pragma unroll
            for (int c = 0; c < 3; ++c) b3[c] = tp.bias8[3 * hv + c];
            const int wvu = __builtin_amdgcn_readfirstlane(wave), ty = cur.oy0 / ST_H, X = cur.ox0 + jv;
#pragma unroll
            for (int rloc = 0; rloc < 2; ++rloc) {
                const int rr = wvu * 2 + rloc, Y = cur.oy0 + rr;            // (scalar)
                const bool live = Y < p.H && X < p.W;
                const unsigned pv = live ? (unsigned)(Y * p.W + X) * 4u : BAD_OFFSET;
                const float* own = slab_of(wvu) + (rloc * TS_GROUPS + 3 * hv) * 32 + jv;           // S[g][jv] of row rr at own[g * 32] (g relative to 3 hv)
                const float* up = (rloc == 1 ? slab_of(wvu) : slab_of(wvu - 1 < 0 ? 0 : wvu - 1) + TS_GROUPS * 32) + (3 * hv) * 32 + jv;     // row rr - 1
                const float* dn = (rloc == 0 ? slab_of(wvu) + TS_GROUPS * 32 : slab_of(wvu + 1 > 3 ? 3 : wvu + 1)) + (3 * hv) * 32 + jv;    // row rr + 1
                float vsum[3];
                if (rr == 0) {
                    const unsigned e1 = live ? (unsigned)(ty * p.W + X) * 4u : BAD_OFFSET, e2 = live ? (unsigned)(2 * ty * p.W + X) * 4u : BAD_OFFSET;
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        vsum[c] = own[(6 + c) * 32];
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, dn[(12 + c) * 32]), zrs, (int)e1, (6 + 3 * hv + c) * tp.zPlane * 4, 0);
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, own[(12 + c) * 32]), zrs, (int)e2, (12 + 3 * hv + c) * tp.zPlane * 4, 0);
                    }
                } else if (rr == ST_H - 1) {
                    const unsigned e2 = live ? (unsigned)((2 * ty + 1) * p.W + X) * 4u : BAD_OFFSET;
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        vsum[c] = (b3[c] + up[c * 32]) + own[(6 + c) * 32];
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, own[c * 32]), zrs, (int)e2, (12 + 3 * hv + c) * tp.zPlane * 4, 0);
                    }
                } else {
                    const bool below = Y + 1 < p.H;                          // (a row outside the image is the convolution's zero padding: skipped)
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        vsum[c] = (b3[c] + up[c * 32]) + own[(6 + c) * 32];
                        if (below) vsum[c] += dn[(12 + c) * 32];
                    }
                }
#pragma unroll
                for (int c = 0; c < 3; ++c) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vsum[c]), zrs, (int)pv, (3 * hv + c) * tp.zPlane * 4, 0);
            }
        }
        __syncthreads();
        if (FUSED) {
            const float* zt = reinterpret_cast<const float*>(patch);
            const int W = p.W, H = p.H;
            {   // pixels this tile can finish on its own: bias, then the nine partials in tap order
                const int row = tid >> 5, col = tid & 31;
                const int X = cur.ox0 + col, Y = cur.oy0 + row;
                if (X < W && Y < H && !tail_is_seam(X, Y, W, H)) {
                    float v[6];
#pragma unroll
                    for (int c = 0; c < 6; ++c) v[c] = tp.bias8[c];
#pragma unroll
                    for (int t = 0; t < 9; ++t) {
                        const int pc = col + t % 3 - 1, pr = row + t / 3 - 1;
                        if (tail_in_image(cur.ox0 + pc, cur.oy0 + pr, W, H)) {        // in the image = in this tile, for these pixels
#pragma unroll
                            for (int c = 0; c < 6; ++c) v[c] += zt[(tail_zrow(t, c) * ST_H + pr) * ST_W + pc];
                        }
                    }
                    isr_finish_pixel(tp.fin, X, Y, v);
                }
            }
            // seam pixels of this tile's rim and of the frame around it: the partials this tile owns go into their records
            for (int u = tid; u < SP_PIX; u += S_THREADS) {
                const int er = u / SP_W, ec = u - er * SP_W;
                const int qx = cur.ox0 - 1 + ec, qy = cur.oy0 - 1 + er;
                if (!tail_in_image(qx, qy, W, H) || !tail_is_seam(qx, qy, W, H)) continue;
                float* rec = tail_record(tp, qx, qy);
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int pc = ec - 1 + t % 3 - 1, pr = er - 1 + t / 3 - 1;     // the tap's pixel, tile relative
                    if ((unsigned)pc < (unsigned)ST_W && (unsigned)pr < (unsigned)ST_H && tail_in_image(cur.ox0 + pc, cur.oy0 + pr, W, H)) {
#pragma unroll
                        for (int c = 0; c < 6; ++c) rec[t * 6 + c] = zt[(tail_zrow(t, c) * ST_H + pr) * ST_W + pc];
                    }
                }
            }
            __syncthreads();
        }
        if (more) {
            park_loads(slot ^ 1);
            wpark();
            __syncthreads();
        }
        slot ^= 1;
        cur = nxt;
    }
}

#ifdef ISR_DIAG      // (form 0's second launch: the diagnostics build only)
// Second launch: out[c][q] = b[c] + sum over the nine taps, in tap order, of z[t][c][q + d_t]  (taps that fall outside the
// image are the convolution's zero padding), then the frame's finishing code.  One thread per high-resolution pixel.
struct TailFinishParams {
    FinishParams fin;
    const float* z;
    int zPlane;
    const float* bias8;          // the last layer's bias (6 floats, device)
    int taps;                    // 9 (timing experiments: fewer)
};

__global__ __launch_bounds__(256) void tail_combine_finish_kernel(const TailFinishParams p)
{
    const int H = 4 * p.fin.h, W = 4 * p.fin.w;
    const int X = blockIdx.x * blockDim.x + threadIdx.x, Y = blockIdx.y;
    if (X >= W) return;
    float v[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) v[c] = p.bias8[c];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        if (t >= p.taps) break;
        const int py = Y + t / 3 - 1, px = X + t % 3 - 1;
        if ((unsigned)py < (unsigned)H && (unsigned)px < (unsigned)W) {
            const float* zp = p.z + (size_t)tail_zrow(t, 0) * p.zPlane + (size_t)py * W + px;
#pragma unroll
            for (int c = 0; c < 6; ++c) v[c] += zp[(size_t)c * p.zPlane];
        }
    }
    isr_finish_pixel(p.fin, X, Y, v);
}
#endif

// S form, second launch: one thread per high-resolution pixel, a block = 256 consecutive pixels of row Y (eight tiles).
//     out = ((bias + S[dy 0] @ Y - 1) + S[dy 1] @ Y) + S[dy 2] @ Y + 1        (rows outside the image skipped: the convolution's zero padding)
// S comes from the planes, except for the first and last pixel of every tile's 32, whose S the block first rebuilds from the
// records -- (z0 + z1) + z2 with z0 (first pixel) / z2 (last pixel) out of the neighbouring tile's record, zero at the image's
// border -- into LDS (16 pixels x 18 groups), so that every thread runs the same code and the stores stay coalesced.
struct TailSFinishParams {
    FinishParams fin;
    const float* s;              // [18][zPlane]
    const float* rec;            // [H][tilesX][6][18]
    int zPlane, tilesX;
    const float* bias8;
};

__global__ __launch_bounds__(256) void tail_s_finish_kernel(const TailSFinishParams p)
{
    __shared__ float sedge[16][TS_GROUPS + 1];
    const int H = 4 * p.fin.h, W = 4 * p.fin.w;
    const int X0 = blockIdx.x * 256, Y = blockIdx.y, tid = threadIdx.x;
    for (int t = tid; t < 16 * TS_GROUPS; t += 256) {
        const int ep = t / TS_GROUPS, g = t - ep * TS_GROUPS;
        const int tile = X0 / ST_W + (ep >> 1), side = ep & 1;
        const int py = Y + g / 6 - 1, X = tile * ST_W + side * (ST_W - 1);
        float val = 0.0f;
        if (tile < p.tilesX && X < W && (unsigned)py < (unsigned)H) {
            const float* own = p.rec + ((size_t)py * p.tilesX + tile) * TS_REC + g;
            float z0, z1, z2;
            if (side == 0) {
                z0 = tile > 0 ? own[2 * TS_GROUPS - TS_REC] : 0.0f;                             // C of the tile to the left
                z1 = own[3 * TS_GROUPS]; z2 = own[4 * TS_GROUPS];                               // D, E
            } else {
                z0 = own[0]; z1 = own[TS_GROUPS];                                               // A, B
                z2 = (tile + 1 < p.tilesX && X + 1 < W) ? own[5 * TS_GROUPS + TS_REC] : 0.0f;   // F of the tile to the right
            }
            val = (z0 + z1) + z2;
        }
        sedge[ep][g] = val;
    }
    __syncthreads();
    const int X = X0 + tid;
    if (X >= W) return;
    const int j = X & (ST_W - 1);
    const bool edge = j == 0 || j == ST_W - 1;
    const int ep = (tid >> 5) * 2 + (j == ST_W - 1 ? 1 : 0);
    float v[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) v[c] = p.bias8[c];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int py = Y + dy - 1;
        if ((unsigned)py < (unsigned)H) {
            const float* sp = p.s + (size_t)(dy * 6) * p.zPlane + (size_t)py * W + X;
            float sv[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) sv[c] = sp[(size_t)c * p.zPlane];
            if (edge) {
#pragma unroll
                for (int c = 0; c < 6; ++c) sv[c] = sedge[ep][dy * 6 + c];
            }
#pragma unroll
            for (int c = 0; c < 6; ++c) v[c] += sv[c];
        }
    }
    isr_finish_pixel(p.fin, X, Y, v);
}

#ifdef ISR_DIAG      // (form 4's second launch: the diagnostics build only)
// V form, second launch: as tail_s_finish_kernel, but only the tiles' first and last rows still have additions to make (and the first /
// last pixel of a tile's 32 come from the end-pixel records, all three rows of them, as before).
__global__ __launch_bounds__(256) void tail_v_finish_kernel(const TailSFinishParams p)
{
    __shared__ float sedge[16][TS_GROUPS + 1];
    const int H = 4 * p.fin.h, W = 4 * p.fin.w;
    const int X0 = blockIdx.x * 256, Y = blockIdx.y, tid = threadIdx.x;
    for (int t = tid; t < 16 * TS_GROUPS; t += 256) {
        const int ep = t / TS_GROUPS, g = t - ep * TS_GROUPS;
        const int tile = X0 / ST_W + (ep >> 1), side = ep & 1;
        const int py = Y + g / 6 - 1, X = tile * ST_W + side * (ST_W - 1);
        float val = 0.0f;
        if (tile < p.tilesX && X < W && (unsigned)py < (unsigned)H) {
            const float* own = p.rec + ((size_t)py * p.tilesX + tile) * TS_REC + g;
            float z0, z1, z2;
            if (side == 0) {
                z0 = tile > 0 ? own[2 * TS_GROUPS - TS_REC] : 0.0f;                             // C of the tile to the left
                z1 = own[3 * TS_GROUPS]; z2 = own[4 * TS_GROUPS];                               // D, E
            } else {
                z0 = own[0]; z1 = own[TS_GROUPS];                                               // A, B
                z2 = (tile + 1 < p.tilesX && X + 1 < W) ? own[5 * TS_GROUPS + TS_REC] : 0.0f;   // F of the tile to the right
            }
            val = (z0 + z1) + z2;
        }
        sedge[ep][g] = val;
    }
    __syncthreads();
    const int X = X0 + tid;
    if (X >= W) return;
    const int j = X & (ST_W - 1);
    const bool edge = j == 0 || j == ST_W - 1;
    const int ep = (tid >> 5) * 2 + (j == ST_W - 1 ? 1 : 0);
    float v[6];
    if (edge) {
#pragma unroll
        for (int c = 0; c < 6; ++c) v[c] = p.bias8[c];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            if ((unsigned)(Y + dy - 1) < (unsigned)H) {
#pragma unroll
                for (int c = 0; c < 6; ++c) v[c] += sedge[ep][dy * 6 + c];
            }
        }
    } else {
        const int r = Y & (ST_H - 1), ty = Y / ST_H;
        const float* vp = p.s + (size_t)Y * W + X;
        if (r == 0) {
#pragma unroll
            for (int c = 0; c < 6; ++c) v[c] = p.bias8[c];
            if (Y > 0) {
                const float* e2 = p.s + (size_t)12 * p.zPlane + (size_t)(2 * (ty - 1) + 1) * W + X;     // S[dy 0] of the tile above's last row
#pragma unroll
                for (int c = 0; c < 6; ++c) v[c] += e2[(size_t)c * p.zPlane];
            }
#pragma unroll
            for (int c = 0; c < 6; ++c) v[c] += vp[(size_t)c * p.zPlane];                                // S[dy 1] of this row
            if (Y + 1 < H) {
                const float* e1 = p.s + (size_t)6 * p.zPlane + (size_t)ty * W + X;                       // S[dy 2] of the row below
#pragma unroll
                for (int c = 0; c < 6; ++c) v[c] += e1[(size_t)c * p.zPlane];
            }
        } else {
#pragma unroll
            for (int c = 0; c < 6; ++c) v[c] = vp[(size_t)c * p.zPlane];
            if (r == ST_H - 1 && Y + 1 < H) {
                const float* e2 = p.s + (size_t)12 * p.zPlane + (size_t)(2 * (ty + 1)) * W + X;          // S[dy 2] of the tile below's first row
#pragma unroll
                for (int c = 0; c < 6; ++c) v[c] += e2[(size_t)c * p.zPlane];
            }
        }
    }
    isr_finish_pixel(p.fin, X, Y, v);
}
#endif

#ifdef ISR_DIAG      // (form 1's second launch: the diagnostics build only)
// Fused form, second launch: the seam pixels (30 % of the image at 8 x 32 tiles) from their records, same order of additions.
__global__ __launch_bounds__(256) void tail_seam_finish_kernel(const TailParams tp)
{
    const int W = tp.c.W, H = tp.c.H;
    const int nrow = 2 * tp.c.tilesY * W, ncs = 2 * tp.c.tilesX;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    int X, Y;
    if (idx < nrow) {
        const int rs = idx / W;
        X = idx - rs * W; Y = (rs >> 1) * ST_H + (rs & 1) * (ST_H - 1);
    } else {
        const int k = idx - nrow;
        if (k >= H * ncs) return;
        Y = k / ncs;
        const int cs = k - Y * ncs;
        X = (cs >> 1) * ST_W + (cs & 1) * (ST_W - 1);
        if ((Y & 7) == 0 || (Y & 7) == 7) return;                            // a row-seam pixel: handled above
    }
    if (!tail_in_image(X, Y, W, H) || !tail_is_seam(X, Y, W, H)) return;
    const float* rec = tail_record(tp, X, Y);
    float v[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) v[c] = tp.bias8[c];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        if (tail_in_image(X + t % 3 - 1, Y + t / 3 - 1, W, H)) {
#pragma unroll
            for (int c = 0; c < 6; ++c) v[c] += rec[t * 6 + c];
        }
    }
    isr_finish_pixel(tp.fin, X, Y, v);
}
#endif

// w8 [6][64][3][3] fp32 -> header + [q][part][h][m] units: element e of (q, h) is y6 channel 32 (q >> 1) + 16 (q & 1) + (e & 3) +
// 8 (e >> 2) + 4 h, row m = tail_zrow(t, c) holds w8[c][.][t] 2^S (rows >= 54 zero); part 0 = hi, 1 = lo
__global__ __launch_bounds__(256) void tail_prepare_kernel(const float* __restrict__ w8, u32x4* __restrict__ wz)
{
    __shared__ float red[256];
    float m = 0.0f;
    for (int i = threadIdx.x; i < 6 * 64 * 9; i += 256) m = fmaxf(m, fabsf(w8[i]));
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    int S = 0;
    const float mx = red[0];
    if (mx > 0.0f && mx < 3.0e38f) {
        S = 13 - ilogbf(mx);
        S = S < -100 ? -100 : (S > 100 ? 100 : S);
    }
    const float scale = ldexpf(1.0f, S);
    if (threadIdx.x == 0) {
        u32x4 hdr;
        hdr.x = __builtin_bit_cast(unsigned, scale);
        hdr.y = __builtin_bit_cast(unsigned, ldexpf(1.0f, -S));
        hdr.z = (unsigned)S; hdr.w = 0u;
        wz[0] = hdr;
    }
    for (int u = threadIdx.x; u < 4 * 2 * 64; u += 256) {                     // (q, h, m)
        const int mrow = u & 63, hh = (u >> 6) & 1, q = u >> 7;
        const int dxr = mrow / 18, gr = mrow - dxr * 18;                      // row [dx][dy][c] (tail_zrow)
        const int t = 3 * (gr / 6) + dxr, c = gr % 6;
        f16x8 qh, ql;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = 32 * (q >> 1) + 16 * (q & 1) + (e & 3) + 8 * (e >> 2) + 4 * hh;
            _Float16 a, b;
            split16(mrow < TZ_ROWS ? w8[((size_t)c * 64 + k) * 9 + t] * scale : 0.0f, a, b);
            qh[e] = a; ql[e] = b;
        }
        wz[1 + ((q * 2 + 0) * 2 + hh) * 64 + mrow] = __builtin_bit_cast(u32x4, qh);
        wz[1 + ((q * 2 + 1) * 2 + hh) * 64 + mrow] = __builtin_bit_cast(u32x4, ql);
    }
}

} // namespace

extern "C" {

long long isrConvTailWeightBytes(void) { return 16 + (long long)TZ_UNITS * 16; }

// 0 (default): 54 z planes + a streaming combine kernel.  1: the partials of a tile are combined in LDS, pixels whose nine
// partials lie in their own tile are finished inside the convolution kernel, the tiles' rims go through per-pixel records and
// tail_seam_finish_kernel -- bit-identical output, 0.9 GB less traffic per 1080p frame, and SLOWER: the convolution kernel
// runs two 256-register waves per SIMD at the board's power limit, and 54 LDS reads + the finishing code's scattered loads,
// divisions and nine stores per pixel inside it cost 0.25 ms where the streaming kernel needs 0.10 (0.75 vs 0.51 + 0.10 ms)
static int g_tail_fused = isr_diag_env_int("ISR_TAIL_FORM", 2);     // (ISR_TAIL_FORM: A/B runs) 2: the S form (default); 0: 54 planes; 1: combined in LDS; 3: timing experiment (a third of form 0's planes, wrong output)
__device__ u32x4 g_tail_zero_unit[4];       // zero initialised: the source of out-of-image units of the LDS-DMA staging
#ifdef ISR_DIAG
void isrDebugSetTailFused(int on) { g_tail_fused = on; }      // not part of the public header
int isrDebugTailState(void) { return g_tail_fused != 2 ? 1 : 0; }
#endif

static long long tail_row_floats(int H, int W) { return 2LL * ((H + ST_H - 1) / ST_H) * W * TZ_ROWS; }
static long long tail_col_floats(int H, int W) { return (long long)H * 2 * ((W + ST_W - 1) / ST_W) * TZ_ROWS; }

long long isrConvTailWorkspaceBytes(int h, int w)
{
    if (h <= 0 || w <= 0) return -1;
    const long long H = 4LL * h, W = 4LL * w;
    const long long planes = (long long)TZ_ROWS * (H * W + W) * 4;           // two-kernel form: one extra row between planes (see ops.empty_planes)
    const long long records = (tail_row_floats((int)H, (int)W) + tail_col_floats((int)H, (int)W)) * 4;
    // (the S form needs 18 planes + [H][tilesX][108] floats of end-pixel records: less than the 54 planes)
    return planes > records ? planes : records;
}

int isrConvTailPrepare(const float* w8, void* wz, void* stream)
{
    if (!w8 || !wz) return -1;
    hipLaunchKernelGGL(tail_prepare_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, w8, (u32x4*)wz);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int isrConvTailSupported(const float* x, int h, int w, long long xPlane)
{
    const long long H = 4LL * h, W = 4LL * w;
    if (h <= 0 || w <= 0 || !x) return 0;
    if (((uintptr_t)x & 15) != 0 || (xPlane & 3) != 0 || xPlane < H * W) return 0;
    if (xPlane * 64 * 4 > 0x7fffffffLL || (H * W + W) * TZ_ROWS * 4 > 0x7fffffffLL) return 0;
    return 1;
}

static int tail_launch(const void* x, int packed, const void* wq6, const float* bias6, const void* wz, const float* bias8, void* workspace,
                       const float* net_input, float* next_prev, float* rgb, int h, int w, long long xPlane,
                       const float* shading24, int exponent, float ao_strength, int inverse_ao, int enable_specular, void* stream);

int isrConvTailFinishFrame(const float* x, const void* wq6, const float* bias6, const void* wz, const float* bias8, void* workspace,
                           const float* net_input, float* next_prev, float* rgb, int h, int w, long long xPlane,
                           const float* shading24, int exponent, float ao_strength, int inverse_ao, int enable_specular, void* stream)
{
    return tail_launch(x, 0, wq6, bias6, wz, bias8, workspace, net_input, next_prev, rgb, h, w, xPlane, shading24, exponent, ao_strength,
                       inverse_ao, enable_specular, stream);
}

int isrConvTailFinishFramePacked(const void* xps, const void* wq6, const float* bias6, const void* wz, const float* bias8, void* workspace,
                                 const float* net_input, float* next_prev, float* rgb, int h, int w, long long xpsPlane,
                                 const float* shading24, int exponent, float ao_strength, int inverse_ao, int enable_specular, void* stream)
{
    return tail_launch(xps, 1, wq6, bias6, wz, bias8, workspace, net_input, next_prev, rgb, h, w, xpsPlane, shading24, exponent, ao_strength,
                       inverse_ao, enable_specular, stream);
}

static int tail_launch(const void* xin, int packed, const void* wq6, const float* bias6, const void* wz, const float* bias8, void* workspace,
                       const float* net_input, float* next_prev, float* rgb, int h, int w, long long xPlane,
                       const float* shading24, int exponent, float ao_strength, int inverse_ao, int enable_specular, void* stream)
{
    const float* x = (const float*)xin;
    unsigned* const rangeFlag = isr_take_range_flag();       // taken first: an error return must not leave it armed
    if (!x || !wq6 || !wz || !bias8 || !workspace || !net_input || !next_prev || (rgb && !shading24)) return -1;
    if (!packed && !isrConvTailSupported(x, h, w, xPlane)) return -3;
    const int H = 4 * h, W = 4 * w;
    if (packed && (((uintptr_t)xin & 15) != 0 || xPlane < (long long)H * W || xPlane * 16 * 16 > 0x7fffffffLL
                   || ((long long)H * W + W) * TZ_ROWS * 4 > 0x7fffffffLL)) return -3;
    TailParams tp;
    SplitConvParams& p = tp.c;
    p.x = x; p.wq = (const u32x4*)wq6; p.bias = bias6; p.residual = nullptr; p.y = nullptr;
    p.N = 1; p.Cin = 64; p.H = H; p.W = W; p.Cout = 64;
    p.Hin = H; p.Win = W;
    p.xPlane = (int)xPlane; p.yPlane = 0; p.rPlane = 0;
    p.xImage = 64 * xPlane; p.yImage = 0; p.rImage = 0;
    p.ksteps = 4; p.coutPad = 64; p.cgroups = 1;
    p.tilesX = (W + ST_W - 1) / ST_W; p.tilesY = (H + ST_H - 1) / ST_H;
    p.act = ISR_ACT_RELU; p.slope = 0.0f;
    ISR_DIAG_SET(p.stamps, nullptr); ISR_DIAG_SET(p.dbg, (g_tail_fused == 3 ? 64 : 0) | (isr_diag_env_int("ISR_TAIL_ROWMAJOR", 0) ? 128 : 0)); p.quads = 1;
    tp.wz = (const u32x4*)wz;
    tp.z = (float*)workspace;
    tp.zPlane = H * W + W;
    tp.xps = nullptr; tp.xpsPlane = 0; tp.zero = nullptr;
    p.ps = nullptr; p.psPlane = 0; p.xps = nullptr; p.xpsPlane = 0; p.zero = nullptr;
    p.absmax = rangeFlag;                  // here: the largest |y6|, the 64-channel intermediate that is split in registers
    if (packed) {
        static u32x4* zero = nullptr;
        if (!zero && hipGetSymbolAddress((void**)&zero, HIP_SYMBOL(g_tail_zero_unit)) != hipSuccess) return -2;
        tp.xps = (const u32x4*)xin; tp.xpsPlane = (int)xPlane; tp.zero = zero;
        p.x = nullptr; p.xPlane = 0; p.xImage = 0;
    }
    static int slots = 0;
    if (!slots) {
        int dev = 0, cus = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        slots = 2 * cus;
#ifdef ISR_DIAG
        (void)hipFuncSetAttribute((const void*)conv3x3_split_tail_kernel<0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, T_LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)conv3x3_split_tail_kernel<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, T_LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)conv3x3_split_tail_kernel<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, T_LDS_BYTES);
#endif
        (void)hipFuncSetAttribute((const void*)conv3x3_split_tail_kernel<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, T_LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)conv3x3_split_tail_kernel<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, T_LDS_BYTES);
#ifdef ISR_DIAG
        (void)hipFuncSetAttribute((const void*)conv3x3_split_tail_kernel<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, T_LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)conv3x3_split_tail_kernel<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, T_LDS_BYTES);
#endif
    }
    const long long ntiles = (long long)p.tilesX * p.tilesY;
    const long long want = ntiles < slots ? ((ntiles + 7) / 8) * 8 : slots;
    hipStream_t s = (hipStream_t)stream;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    isr_fill_finish_params(tp.fin, nullptr, net_input, next_prev, rgb, h, w, shading24, exponent, ao_strength, inverse_ao, enable_specular);
    tp.bias8 = bias8;
    tp.rowrec = (float*)workspace;
    tp.colrec = tp.rowrec + tail_row_floats(H, W);
    const bool fused = g_tail_fused == 1 && !packed;
    // algorithmic flops: postblock.6 and the final 64 -> 6 layer, whose arithmetic this launch carries
    isr_profile_record(ISR_VARIANT_SPLIT_TAIL, 2.0 * 9 * 64 * (64 + 6) * (double)H * W, &e0, &e1);
    tp.srec = tp.z + (size_t)TS_GROUPS * tp.zPlane;
    const dim3 tgrid((unsigned)want), tblock(S_THREADS);
#define TAIL_LAUNCH(FORM, PS)                                                                                                          \
    do {                                                                                                                               \
        if (e0 || e1) hipExtLaunchKernelGGL((conv3x3_split_tail_kernel<FORM, PS>), tgrid, tblock, T_LDS_BYTES, s, e0, e1, 0, tp);      \
        else hipLaunchKernelGGL((conv3x3_split_tail_kernel<FORM, PS>), tgrid, tblock, T_LDS_BYTES, s, tp);                             \
    } while (0)
#ifdef ISR_DIAG     // the forms beside the default (0: 54 planes, 1: finishing inside the kernel, 4: vertical sums inside the kernel; all measured slower) exist in the diagnostics build only
    if (fused) {
        TAIL_LAUNCH(1, false);
        const long long threads = 2LL * p.tilesY * W + (long long)H * 2 * p.tilesX;
        ISR_LAUNCH_PROFILED(ISR_VARIANT_TAIL_FINISH, tail_seam_finish_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, tp);
    } else if (g_tail_fused == 2 || g_tail_fused == 4) {
        TailSFinishParams fp;
        fp.fin = tp.fin;
        fp.s = tp.z; fp.rec = tp.srec; fp.zPlane = tp.zPlane; fp.tilesX = p.tilesX; fp.bias8 = bias8;
        if (g_tail_fused == 4) {
            if (packed) TAIL_LAUNCH(4, true); else TAIL_LAUNCH(4, false);
            ISR_LAUNCH_PROFILED(ISR_VARIANT_TAIL_FINISH, tail_v_finish_kernel, dim3((unsigned)((W + 255) / 256), (unsigned)H), dim3(256), 0, s, fp);
        } else {
            if (packed) TAIL_LAUNCH(2, true); else TAIL_LAUNCH(2, false);
            ISR_LAUNCH_PROFILED(ISR_VARIANT_TAIL_FINISH, tail_s_finish_kernel, dim3((unsigned)((W + 255) / 256), (unsigned)H), dim3(256), 0, s, fp);
        }
    } else {
        if (packed) TAIL_LAUNCH(0, true); else TAIL_LAUNCH(0, false);
        TailFinishParams fp;
        fp.fin = tp.fin;
        fp.z = tp.z; fp.zPlane = tp.zPlane; fp.bias8 = bias8; fp.taps = g_tail_fused == 3 ? 3 : 9;
        ISR_LAUNCH_PROFILED(ISR_VARIANT_TAIL_FINISH, tail_combine_finish_kernel, dim3((W + 255) / 256, H), dim3(256), 0, s, fp);
    }
#else
    {
        (void)fused;
        TailSFinishParams fp;
        fp.fin = tp.fin;
        fp.s = tp.z; fp.rec = tp.srec; fp.zPlane = tp.zPlane; fp.tilesX = p.tilesX; fp.bias8 = bias8;
        if (packed) TAIL_LAUNCH(2, true); else TAIL_LAUNCH(2, false);
        ISR_LAUNCH_PROFILED(ISR_VARIANT_TAIL_FINISH, tail_s_finish_kernel, dim3((unsigned)((W + 255) / 256), (unsigned)H), dim3(256), 0, s, fp);
    }
#endif
#undef TAIL_LAUNCH
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

} // extern "C"
