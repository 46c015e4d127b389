// One residual block of EnhanceNet's trunk,  y = x + conv2(relu(conv1(x)))  (64 -> 64 -> 64 channels,
// SuperresolutionNetwork/models/enhancenet.py:18-33,108-112,139-141), as ONE launch on the split-operand arithmetic of
// sr_conv_split.hip -- the same products in the same order as two launches of conv3x3_split_stream_kernel, so the result is
// bit-identical to them.
//
// Why: at 480 x 270 a trunk layer is ONE round of 510 workgroups on 512 slots: 11.5 us of MFMA issue inside a 45 us launch --
// first staging, epilogue and launch / drain are exposed chip-wide, 21 times per frame (0.93 of a 2.4 ms frame).  Round 2's
// fused block kept the intermediate tile in LDS: 152 KB, ONE workgroup per CU, every barrier and conversion phase idled the
// matrix pipe -- slower than two launches.  Here the intermediate t = relu(conv1(x)) of a tile never needs to be LDS resident:
//   * a workgroup computes conv1 on the 10 x 34 pixels its 8 x 32 output tile needs (1-pixel halo recomputed: the region is
//     walked as 12 flat blocks of 32 positions with the patch's row stride, 1.41x the tile's MFMAs -- the matrix pipe is 29 %
//     busy in this part of the network, the time is elsewhere), splits relu(.) into (hi, lo') fp16 pairs IN REGISTERS and writes
//     them as LDS-ready 16-byte units into a per-workgroup scratch in global memory (87 KB, rewritten for every tile: L2);
//   * conv2 streams its k-steps from that scratch by LDS-DMA (global_load_lds_dwordx4: no registers, no conversion, no
//     ds_write), two slots; the first k-step's units are written into slot 0 straight from the registers.
// LDS stays at 80 KB -> two workgroups per CU, one's barriers and epilogues under the other's MFMAs, as in the layer kernels.
// Per block: one launch and one first-staging / epilogue pair less, t (33 MB written, 44 MB read back through the memory system
// by another launch) stays next to the CU that made it.
#include "sr_diag.h"
#include "sr_split_common.h"

namespace {

constexpr int B_THREADS = 256;
constexpr int R1_H = ST_H + 2, R1_W = ST_W + 2;                              // 10 x 34: the pixels of t the output tile needs
constexpr int P1_H = ST_H + 4, P1_W = ST_W + 4;                              // 12 x 36: the patch of x conv1 reads for them
[[maybe_unused]] constexpr int P1_PIX = P1_H * P1_W;                                          // 432
constexpr int P1_SEG = 464;                                                  // units per (part, group) of the conv1 slot: 432 + the overrun of the last flat block's taps
[[maybe_unused]] constexpr int P1_UNITS = 4 * P1_SEG;                                         // [hi | lo][2 groups][P1_SEG] = 1856 units
constexpr int R1_PIX = R1_H * R1_W;                                          // 340 = SP_PIX: conv2's patch IS the region of t
constexpr int B_BLOCKS = 3;                                                  // flat blocks of 32 positions per wave: 4 x 3 x 32 = 384 >= 10 x 36
constexpr int SCR_UNITS = 2 * 8 * R1_PIX;                                    // scratch per workgroup: [hi | lo][8 groups][340] = 5440 units = 87 040 B
constexpr int Q1_QPR = (P1_W + 4) / 4;                                       // 10 aligned quads per conv1 patch row: columns ox0 - 4 .. ox0 + 35
constexpr int Q1_UNITS = 2 * P1_H * Q1_QPR;                                  // 240 staging units (group, row, quad) per k-step
constexpr int C2_SLOT = 2 * R1_PIX;                                          // conv2 patch as in the layer kernels: hi part = two slots of [2 groups][340] units,
constexpr int C2_PART = 2 * C2_SLOT;                                         // lo part the same, C2_PART = S_PART units further on (split_kstep's addressing)
constexpr int B_PATCH_UNITS = 2 * C2_PART;                                   // 2720 >= P1_UNITS, and >= the 2048 units the epilogue transposes through
static_assert(C2_PART == S_PART && C2_SLOT == 2 * SP_PIX && R1_W == SP_W, "conv2 re-uses split_kstep");
constexpr int B_LDS_BYTES = (S_WUNITS + B_PATCH_UNITS) * 16 + 512;           // weights 36 864 + patch 43 520 + two biases = 80 896

struct BlockParams {
    const float* x; float* y;
    const u32x4* wq1; const u32x4* wq2;
    const float* bias1; const float* bias2;
    u32x4* scratch;               // gridDim.x * SCR_UNITS units
    int H, W, xPlane, yPlane, tilesX, tilesY;
    unsigned* absmax;             // range guard (SplitConvParams::absmax): over the intermediate t AND the output
    ISR_DIAG_MEMBER(unsigned long long*, stamps, nullptr);   // diagnostics: 8 s_memrealtime stamps (100 MHz, chip-wide clock) per workgroup, first tile; or NULL
    ISR_DIAG_MEMBER(int, dbg, 0);                      // diagnostics: 1 skip conv1's MFMAs, 2 skip conv2's MFMAs, 4 skip the scratch stores, 8 skip the DMA
};

typedef u32x2 uint2_t;
__device__ __forceinline__ void dma16(const u32x4* src, u32x4* dst_wave_base) { isr_dma16(src, dst_wave_base); }

__global__ __launch_bounds__(B_THREADS, 2) void resblock_split_kernel(const BlockParams p)
{
    extern __shared__ u32x4 lds[];
    u32x4* wbuf = lds;                                                       // one k-step of weights (conv1's, then conv2's)
    u32x4* patch = lds + S_WUNITS;                                           // conv1: one slot of P1_UNITS; conv2: two slots of C2_SLOT
    float* bias_lds = reinterpret_cast<float*>(lds + S_WUNITS + B_PATCH_UNITS);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int ntiles = p.tilesY * p.tilesX;
    const int xcd = blockIdx.x & 7, jw = blockIdx.x >> 3;
    const int njw = ((int)gridDim.x >> 3) + (xcd < ((int)gridDim.x & 7) ? 1 : 0);
    const int tq = ntiles >> 3, trm = ntiles & 7;
    const int tstart = xcd < trm ? xcd * (tq + 1) : trm * (tq + 1) + (xcd - trm) * tq;
    const int tcount = tq + (xcd < trm ? 1 : 0);
    if (jw >= tcount) return;
    const unsigned planeBytes = (unsigned)p.xPlane * 4u;
    u32x4* scr = p.scratch + (size_t)blockIdx.x * SCR_UNITS;
    if (tid < 128) bias_lds[tid] = tid < 64 ? (p.bias1 ? p.bias1[tid] : 0.0f) : (p.bias2 ? p.bias2[tid - 64] : 0.0f);

    struct Tile { int oy0, ox0; };
    auto decode = [&](int t) {
        int b = tstart + t;
        Tile r;
        r.ox0 = (b % p.tilesX) * ST_W; b /= p.tilesX;
        r.oy0 = b * ST_H;
        return r;
    };
    // ---- conv1 staging: fp32 x -> (hi, lo') units, through registers (one (group, row, quad) unit per thread and k-step) ----
    const bool staging = tid < Q1_UNITS;
    const int ug = tid / (P1_H * Q1_QPR), urem = tid - ug * (P1_H * Q1_QPR);
    const int ur = urem / Q1_QPR, uq = urem - ur * Q1_QPR;
    const rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)((size_t)64 * p.xPlane * 4), 0x00020000);
    u32x4 v[8];
    auto issue_loads = [&](const Tile& t, int ks) {
        const int iy = t.oy0 + ur - 2, ix = t.ox0 - 4 + 4 * uq;
        const bool ok = staging && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        const unsigned base = (unsigned)(ks * 16 + ug * 8) * planeBytes + (unsigned)(iy * p.W + ix) * 4u;
#pragma unroll
        for (int e = 0; e < 8; ++e)
            v[e] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)(ok ? base : BAD_OFFSET), (int)((unsigned)e * planeBytes), 0);
    };
    auto park_loads = [&]() {
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
        // quad q holds image columns ox0 - 4 + 4q .. + 3 = patch columns 4q - 2 .. 4q + 1: quad 0 contributes its last two,
        // quad 9 its first two
        u32x4* dst = patch + ug * P1_SEG + ur * P1_W + 4 * uq - 2;
        if (uq > 0) {
            dst[0] = __builtin_bit_cast(u32x4, h0); dst[2 * P1_SEG] = __builtin_bit_cast(u32x4, l0);
            dst[1] = __builtin_bit_cast(u32x4, h1); dst[2 * P1_SEG + 1] = __builtin_bit_cast(u32x4, l1);
        }
        if (uq < Q1_QPR - 1) {
            dst[2] = __builtin_bit_cast(u32x4, h2); dst[2 * P1_SEG + 2] = __builtin_bit_cast(u32x4, l2);
            dst[3] = __builtin_bit_cast(u32x4, h3); dst[2 * P1_SEG + 3] = __builtin_bit_cast(u32x4, l3);
        }
    };
    // weights of one k-step: 2304 units [part][tap][lane half][64 couts]; thread t moves unit (tap i, part t / 128, t % 128) for
    // i = 0..8: in the prepared image that is byte 16 t + 16384 i + 4096 ks -- ONE address register, the rest scalar offsets
    const rsrc_t w1rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(p.wq1 + 1), 0, 9 * 4 * 256 * 16, 0x00020000);
    const rsrc_t w2rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(p.wq2 + 1), 0, 9 * 4 * 256 * 16, 0x00020000);
    u32x4 wreg[9];
    auto wfetch = [&](rsrc_t wrs, int ks) {
#pragma unroll
        for (int i = 0; i < 9; ++i) wreg[i] = __builtin_amdgcn_raw_buffer_load_b128(wrs, tid * 16, i * 16384 + ks * 4096, 0);
    };
    u32x4* wdst = wbuf + (tid >> 7) * S_WPART + (tid & 127);
    auto wpark = [&]() {
#pragma unroll
        for (int i = 0; i < 9; ++i) wdst[i * 128] = wreg[i];
    };
    // ---- conv2 staging: k-step ks of t from the scratch into slot `slot`, 22 wave-wide 1 KB pieces dealt to the 4 waves ----
    auto dma_kstep = [&](int ks, int slot) {
#pragma unroll
        for (int d = 0; d < 6; ++d) {
            const int piece = wave + 4 * d;                                  // 0 .. 23: 11 pieces per part (680 units), pieces 22, 23 do not exist
            const int part = piece / 11, off = (piece - part * 11) * 64 + lane;   // unit of the part's slot: [group of the k-step][pixel]
            if (piece < 22 && off < C2_SLOT) {
                const int gg = off / R1_PIX, pix = off - gg * R1_PIX;
                dma16(scr + (part * 8 + 2 * ks + gg) * R1_PIX + pix, patch + part * C2_PART + slot * C2_SLOT + (piece - part * 11) * 64);
            }
        }
    };
    const float unscale1 = reinterpret_cast<const float*>(p.wq1)[1];
    SplitConvParams p2;                                                      // conv2's epilogue: y = acc 2^-S + bias2 + x
    p2.x = nullptr; p2.wq = p.wq2; p2.bias = p.bias2; p2.residual = p.x; p2.y = p.y;
    p2.N = 1; p2.Cin = 64; p2.H = p.H; p2.W = p.W; p2.Cout = 64;
    p2.xPlane = p.xPlane; p2.yPlane = p.yPlane; p2.rPlane = p.xPlane;
    p2.xImage = 0; p2.yImage = 0; p2.rImage = 0;
    p2.ksteps = 4; p2.coutPad = 64; p2.cgroups = 1; p2.tilesX = p.tilesX; p2.tilesY = p.tilesY;
    p2.act = ISR_ACT_NONE; p2.slope = 0.0f; p2.Hin = p.H; p2.Win = p.W; p2.quads = 1; ISR_DIAG_SET(p2.dbg, 0); ISR_DIAG_SET(p2.stamps, nullptr); p2.ps = nullptr; p2.psPlane = 0; p2.xps = nullptr; p2.xpsPlane = 0; p2.zero = nullptr; p2.absmax = p.absmax; p2.slotmax = nullptr;

    // The two workgroups of a CU share its SIMDs, and issue arbitration prefers the OLDER wave: the first-dispatched workgroup
    // of a CU runs near full speed, the second (in practice blockIdx >= half the grid) gets the leftovers and finishes ~20 us
    // later with the CU half idle -- the launch lasts as long as the losers.  prio experiments (p.dbg): 16 younger half at priority
    // 1 from conv2 on, 32 always, 64 alternating per k-step with the older half
    const bool younger = (int)blockIdx.x >= ((int)gridDim.x >> 1);
    if ((p.dbg & 32) && younger) __builtin_amdgcn_s_setprio(1);
    unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (p.stamps) st[0] = __builtin_amdgcn_s_memrealtime();
    Tile cur = decode(jw);
    issue_loads(cur, 0);
    wfetch(w1rs, 0);
    park_loads();
    wpark();
    __syncthreads();
    if (p.stamps) st[1] = __builtin_amdgcn_s_memrealtime();
    for (int t = jw; t < tcount; t += njw) {
        const bool more = t + njw < tcount;
        Tile nxt = cur;
        if (more) nxt = decode(t + njw);
        // ================= conv1 on the 10 x 34 region, as flat blocks 3 wave .. 3 wave + 2 of 32 positions =================
        f32x16 acc1[2][B_BLOCKS];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int b = 0; b < B_BLOCKS; ++b)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc1[cb][b][i] = 0.0f;
#pragma unroll 1
        for (int ks = 0; ks < 4; ++ks) {
            if (p.dbg & 64) { if (((ks & 1) != 0) == younger) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
            if (ks < 3) { issue_loads(cur, ks + 1); wfetch(w1rs, ks + 1); }
            else wfetch(w2rs, 0);                                           // conv2's first weights travel under conv1's last k-step
            if (!(p.dbg & 1)) {
                const u32x4* wl = wbuf + h * 64 + j;
                const u32x4* bl = patch + h * P1_SEG + wave * (B_BLOCKS * 32) + j;
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int dy = tap / 3, dx = tap - dy * 3;
                    const f16x8 a0h = __builtin_bit_cast(f16x8, wl[tap * 128]);
                    const f16x8 a0l = __builtin_bit_cast(f16x8, wl[S_WPART + tap * 128]);
                    const f16x8 a1h = __builtin_bit_cast(f16x8, wl[tap * 128 + 32]);
                    const f16x8 a1l = __builtin_bit_cast(f16x8, wl[S_WPART + tap * 128 + 32]);
                    const f16x8 a0s = a0h * (_Float16)0.00048828125f;       // w_hi 2^-11: partner of the scaled x_lo'
                    const f16x8 a1s = a1h * (_Float16)0.00048828125f;
#pragma unroll
                    for (int b = 0; b < B_BLOCKS; ++b) {
                        const f16x8 bh = __builtin_bit_cast(f16x8, bl[b * 32 + dy * P1_W + dx]);
                        const f16x8 bo = __builtin_bit_cast(f16x8, bl[2 * P1_SEG + b * 32 + dy * P1_W + dx]);
                        acc1[0][b] = mfma16(a0l, bh, acc1[0][b]);
                        acc1[0][b] = mfma16(a0s, bo, acc1[0][b]);
                        acc1[0][b] = mfma16(a0h, bh, acc1[0][b]);
                        acc1[1][b] = mfma16(a1l, bh, acc1[1][b]);
                        acc1[1][b] = mfma16(a1s, bo, acc1[1][b]);
                        acc1[1][b] = mfma16(a1h, bh, acc1[1][b]);
                    }
                }
            }
            __syncthreads();                                                 // slot and weight buffer free
            if (ks < 3) { park_loads(); wpark(); __syncthreads(); }
        }
        // ---- t = relu(acc 2^-S + bias1) inside the image, 0 outside (conv2's zero padding), as (hi, lo') units: to the scratch,
        // and the first k-step's two groups straight into conv2's slot 0 --------------------------------------------------------
        if (p.stamps && t == jw) st[2] = __builtin_amdgcn_s_memrealtime();
        wpark();                                                             // conv2 k-step 0 weights
        const rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(scr, 0, SCR_UNITS * 16, 0x00020000);
        unsigned tmag = 0u;
#pragma unroll
        for (int b = 0; b < B_BLOCKS; ++b) {
            const int q = (wave * B_BLOCKS + b) * 32 + j;
            const int r = q / P1_W, c = q - r * P1_W;
            const bool valid = r < R1_H && c < R1_W;
            const int Y = cur.oy0 - 1 + r, X = cur.ox0 - 1 + c;
            const bool inimg = valid && (unsigned)Y < (unsigned)p.H && (unsigned)X < (unsigned)p.W;
            const int pix = r * R1_W + c;
            // this lane's 8 bytes of the unit (group, pixel): byte offset 16 pix + 8 h inside the group's plane; the plane is a
            // wave-uniform (scalar) offset, so a store needs no address arithmetic of its own
            const unsigned voff = (valid && !(p.dbg & 4)) ? (unsigned)(pix * 16 + 8 * h) : BAD_OFFSET;
            char* lbase = reinterpret_cast<char*>(patch) + pix * 16 + 8 * h;
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int gi = 0; gi < 4; ++gi) {
                    const float4 bq = *reinterpret_cast<const float4*>(bias_lds + cb * 32 + 8 * gi + 4 * h);
                    const float bb[4] = {bq.x, bq.y, bq.z, bq.w};
                    f16x4 th, tl;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float val = acc1[cb][b][4 * gi + e] * unscale1 + bb[e];
                        val = (inimg && val > 0.f) ? val : 0.f;
                        _Float16 a, bo;
                        split16x(val, a, bo);
                        th[e] = a; tl[e] = bo;
                        tmag = isr_umax(tmag, isr_mag(val));
                    }
                    const int g = cb * 4 + gi;
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(uint2_t, th), srs, (int)voff, g * R1_PIX * 16, 0);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(uint2_t, tl), srs, (int)voff, (8 + g) * R1_PIX * 16, 0);
                    if (cb == 0 && gi < 2 && valid) {
                        *reinterpret_cast<f16x4*>(lbase + gi * R1_PIX * 16) = th;
                        *reinterpret_cast<f16x4*>(lbase + (C2_PART + gi * R1_PIX) * 16) = tl;
                    }
                }
        }
        isr_range_note(p.absmax, tmag);
        __syncthreads();                                                     // scratch stores done and visible to the workgroup; slot 0 and weights in place
        if (p.stamps && t == jw) st[3] = __builtin_amdgcn_s_memrealtime();
        // ================= conv2 on the 8 x 32 tile, k-steps streamed from the scratch ==========================================
        f32x16 acc2[2][2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc2[cb][r][i] = 0.0f;
        if ((p.dbg & 16) && younger) __builtin_amdgcn_s_setprio(1);
#pragma unroll 1
        for (int ks = 0; ks < 4; ++ks) {
            const int slot = ks & 1;
            if (p.dbg & 64) { if (((ks & 1) != 0) == younger) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
            if (ks < 3) { if (!(p.dbg & 8)) dma_kstep(ks + 1, slot ^ 1); wfetch(w2rs, ks + 1); }
            else if (more) { issue_loads(nxt, 0); wfetch(w1rs, 0); }
            if (!(p.dbg & 2)) split_kstep(acc2, wbuf + h * 64 + j, patch + slot * C2_SLOT + h * R1_PIX + (wave * 2) * R1_W + j, true);
            __syncthreads();                                                 // DMA landed (vmcnt(0) precedes the barrier), readers done
            if (ks < 3) { wpark(); __syncthreads(); }
        }
        if (p.stamps && t == jw) st[4] = __builtin_amdgcn_s_memrealtime();
        split_epilogue<true>(p2, acc2, patch, 0, cur.oy0, cur.ox0, 0, true, lane, wave, j, h);
        if (p.stamps && t == jw) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            st[5] = __builtin_amdgcn_s_memrealtime();
            if (tid == 0) {
                unsigned long long* o = p.stamps + (size_t)blockIdx.x * 8;
                for (int k = 0; k < 6; ++k) o[k] = st[k];
            }
        }
        __syncthreads();
        if (more) {
            park_loads();
            wpark();
            __syncthreads();
        }
        cur = nxt;
    }
}

} // namespace

extern "C" {

[[maybe_unused]] static unsigned long long* g_block_stamps = nullptr;
[[maybe_unused]] static int g_block_dbg = 0;
#ifdef ISR_DIAG
void isrDebugSetBlockStampBuffer(unsigned long long* buf) { g_block_stamps = buf; }   // not part of the public header
void isrDebugSetBlockAblation(int bits) { g_block_dbg = bits; }
int isrDebugBlockState(void) { return (g_block_stamps ? 1 : 0) | (g_block_dbg ? 2 : 0); }
#endif

static int block_slots()
{
    static int slots = 0;
    if (!slots) {
        int dev = 0, cus = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        slots = 2 * cus;
        (void)hipFuncSetAttribute((const void*)resblock_split_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, B_LDS_BYTES);
    }
    return slots;
}

long long isrResBlockSplitWorkspaceBytes(void) { return (long long)block_slots() * SCR_UNITS * 16; }

int isrResBlockSplitSupported(const float* x, int H, int W, long long xPlane, long long yPlane)
{
    if (!x || H <= 0 || W <= 0) return 0;
    if ((W & 3) != 0 || ((uintptr_t)x & 15) != 0 || (xPlane & 3) != 0 || (yPlane & 3) != 0) return 0;
    if (xPlane < (long long)H * W || yPlane < (long long)H * W) return 0;
    if (xPlane * 64 * 4 > 0x7fffffffLL || yPlane * 64 * 4 > 0x7fffffffLL) return 0;
    return 1;
}

int isrResBlockSplit(const float* x, const void* wq1, const float* bias1, const void* wq2, const float* bias2, float* y, void* workspace,
                     int H, int W, long long xPlane, long long yPlane, void* stream)
{
    unsigned* const rangeFlag = isr_take_range_flag();       // taken first: an error return must not leave it armed
    if (!x || !wq1 || !wq2 || !y || !workspace) return -1;
    if (!isrResBlockSplitSupported(x, H, W, xPlane, yPlane) || ((uintptr_t)y & 15) != 0) return -3;
    BlockParams p;
    p.x = x; p.y = y; p.wq1 = (const u32x4*)wq1; p.wq2 = (const u32x4*)wq2; p.bias1 = bias1; p.bias2 = bias2;
    p.scratch = (u32x4*)workspace;
    p.H = H; p.W = W; p.xPlane = (int)xPlane; p.yPlane = (int)yPlane;
    p.tilesX = (W + ST_W - 1) / ST_W; p.tilesY = (H + ST_H - 1) / ST_H;
    ISR_DIAG_SET(p.stamps, g_block_stamps); ISR_DIAG_SET(p.dbg, g_block_dbg);
    p.absmax = rangeFlag;
    const int slots = block_slots();
    const long long ntiles = (long long)p.tilesX * p.tilesY;
    const long long want = ntiles < slots ? ((ntiles + 7) / 8) * 8 : slots;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    isr_profile_record(ISR_VARIANT_SPLIT_BLOCK, 2.0 * 2.0 * 9 * 64 * 64 * (double)H * W, &e0, &e1);
    hipStream_t s = (hipStream_t)stream;
    if (e0 || e1) hipExtLaunchKernelGGL(resblock_split_kernel, dim3((unsigned)want), dim3(B_THREADS), B_LDS_BYTES, s, e0, e1, 0, p);
    else hipLaunchKernelGGL(resblock_split_kernel, dim3((unsigned)want), dim3(B_THREADS), B_LDS_BYTES, s, p);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

} // extern "C"
