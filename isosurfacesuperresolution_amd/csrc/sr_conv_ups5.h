// conv3x3(U(x)) for EnhanceNet's two upsampling layers (SuperresolutionNetwork/models/enhancenet.py:113-124) with the staging of k-step
// g + 1 IN THE MULTIPLYING WAVE'S OWN INSTRUCTION STREAM, between the MFMAs of k-step g ("form 5").  Included by
// sr_conv_split.hip (same translation unit as sr_conv_ups3.h); the arithmetic -- interpolation, split, products, their order -- is
// conv3x3_split_kernel<true>'s: bit-identical (tests/test_ups_gpu.py).
//
// sr_conv_ups3.h runs a workgroup's phases one after the other -- park the low-resolution region, interpolate + split + pack it, then
// three tap rows of MFMAs -- and relies on the other two workgroups of the CU to multiply meanwhile.  Here the patch is double buffered:
// while the MFMAs of k-step g read slot g & 1, the SAME waves park k-step g + 1's low-resolution region (tap row 0), interpolate, split
// and pack it into slot (g + 1) & 1 (tap rows 1 and 2) and request k-step g + 2's region (tap row 2): every tap row is one basic block of
// 36 MFMAs and ~1/3 of a k-step's staging, for the scheduler to interleave (tools/lab/mfma_valu_overlap.hip: up to ~4 vector instructions
// per MFMA ride along in the same wave).  The weights of a tap row (hi and lo planes; the partner of the scaled x_lo' is made with
// v_pk_mul_f16 as in the tile kernel) go through registers into the OTHER of two weight slots at the end of the row before: ONE barrier
// per tap row (ups3: two).  LDS: 2 x 21.8 KB patch slots + 2 x 12.3 KB weight slots + the 7.2 KB fp32 copy = 75.3 KB: two workgroups per
// CU, <= 256 registers.
#pragma once
#include <type_traits>
#include "sr_split_common.h"

namespace {

#ifndef U5_FENCE
#define U5_FENCE 1
#endif
constexpr int U5_PART = 2 * SP_PIX;                                          // one k-step of the patch: 2 channel groups; hi, then lo' at + U5_PART
constexpr int U5_PUNITS = 2 * U5_PART;                                       // 1360 units = 21 760 B per slot
constexpr int U5_WROW = 3 * 128;                                             // one tap row of one part: 3 taps x [lane half][64 couts]
constexpr int U5_WUNITS = 2 * U5_WROW;                                       // hi then lo: 768 units = 12 288 B per slot, 3 per thread
constexpr int U5_LR_CS = 113;                                                // channel stride of the fp32 copy (as sr_conv_ups3.h)
constexpr int U5_TMP_UNITS = (16 * U5_LR_CS * 4 + 15) / 16 + 16;             // 452 units = 7 232 B, + 64 floats nobody reads (switched-off parking stores)
constexpr int U5_LDS_BYTES = (2 * U5_PUNITS + 2 * U5_WUNITS + U5_TMP_UNITS) * 16;      // 75 328

__global__ __launch_bounds__(S_THREADS, 2) void conv3x3_split_ups5_kernel(const SplitConvParams p)
{
    extern __shared__ u32x4 lds5[];
    u32x4* const patch = lds5;                                               // [2 slots][U5_PUNITS]; the fp32 epilogue's scratch afterwards
    u32x4* const wbuf = lds5 + 2 * U5_PUNITS;                                // [2 slots][U5_WUNITS]
    float* const tmp = reinterpret_cast<float*>(lds5 + 2 * U5_PUNITS + 2 * U5_WUNITS);     // [16][113] fp32: the low-resolution region being interpolated
    // (not const: re-derived from an opaque copy of the thread index at the top of every k-step, see the loop)
    int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int j = lane & 31, h = lane >> 5;
    int bid;
    {   // an XCD (= an L2) gets a contiguous range of tiles
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    }
    const int tx = bid % p.tilesX; bid /= p.tilesX;
    const int ty = bid % p.tilesY, n = bid / p.tilesY;
    const int oy0 = ty * ST_H, ox0 = tx * ST_W;
    const rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x + (size_t)n * p.xImage), 0,
                                                         (int)((size_t)p.Cin * p.xPlane * 4), 0x00020000);
    const unsigned planeBytes = (unsigned)p.xPlane * 4u;

    // weights of tap row `step` = 3 ks + dy: thread t moves unit (tap 3 dy + i, part t / 128, t % 128), i = 0..2
    const rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(p.wq + 1), 0, 9 * p.ksteps * 4096, 0x00020000);
    const int nsteps = 3 * p.ksteps;
    u32x4 wreg[3];
    auto wfetch = [&](int step) {
        const int ks = step / 3, dy = step - 3 * ks;
#pragma unroll
        for (int i = 0; i < 3; ++i)
            wreg[i] = __builtin_amdgcn_raw_buffer_load_b128(wrs, (int)(step < nsteps ? (unsigned)tid * 16u : BAD_OFFSET), ((3 * dy + i) * p.ksteps + ks) * 4096, 0);
    };
    auto wpark = [&](int slot) {
        u32x4* const d = wbuf + slot * U5_WUNITS + (tid >> 7) * U5_WROW + (tid & 127);
#pragma unroll
        for (int i = 0; i < 3; ++i) d[i * 128] = wreg[i];
    };

    // ---- staging of one k-step (16 channels): low-resolution region -> fp32 copy -> interpolate, split, pack (sr_conv_ups3.h) --------
    constexpr int LR_H = ST_H / 2 + 2, LR_W = ST_W / 2 + 2;                 // 6 x 18 low-res pixels: rows oy0/2 - 1 .., cols ox0/2 - 1 ..
    constexpr int LQ = (ST_W / 2 + 8) / 4;                                   // 6 aligned quads per row
    constexpr int LUNITS = 16 * LR_H * LQ;                                   // (channel, row, quad) = 576: 2.25 per thread
    constexpr int QR = SP_H / 2, QC = SP_W / 2, UQ = QR * QC;               // 5 x 17 quads of 2 x 2 patch pixels: 340 staging units
    const int ly0 = oy0 / 2 - 1, lx0 = ox0 / 2 - 1;
    u32x4 v[3];
    auto lfetch = [&](int cin0) {                                            // requests only (channels beyond Cin: out of range = zeros, never parked)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int u = tid + k * S_THREADS;
            const int c = u / (LR_H * LQ), rem = u - c * (LR_H * LQ);
            const int r = rem / LQ, q = rem - r * LQ;
            const int iy = ly0 + r, ix = ox0 / 2 - 4 + 4 * q;
            const bool ok = u < LUNITS && cin0 < p.Cin && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
            v[k] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)(ok ? (unsigned)(cin0 + c) * planeBytes + (unsigned)(iy * p.Win + ix) * 4u
                                                                       : BAD_OFFSET), 0, 0);
        }
    };
    auto lpark = [&](int k0, int k1) {
#pragma unroll
        for (int k = k0; k < k1; ++k) {
            const int u = tid + k * S_THREADS;
            const int c = u / (LR_H * LQ), rem = u - c * (LR_H * LQ);
            const int r = rem / LQ, q = rem - r * LQ;
            const float4 f = __builtin_bit_cast(float4, v[k]);
            // (no branches: a store that must not happen goes to the lane's own dump float -- a branch per store would cut the tap
            // row's basic block into pieces)
            float* const dst = tmp + c * U5_LR_CS + r * LR_W + 4 * q - 3;     // quad q holds low-res patch columns 4q - 3 .. 4q
            float* const dump = tmp + 16 * U5_LR_CS + lane;
            const bool live = u < LUNITS;
            *((live && q > 0) ? dst : dump) = f.x;
            *((live && q > 0 && q < LQ - 1) ? dst + 1 : dump) = f.y;
            *((live && q > 0 && q < LQ - 1) ? dst + 2 : dump) = f.z;
            *((live && q < LQ - 1) ? dst + 3 : dump) = f.w;
        }
    };
    const bool interior = oy0 >= 2 && oy0 + ST_H + 2 <= p.H && ox0 >= 2 && ox0 + ST_W + 2 <= p.W;
    // staging unit u (< 4 UQ): (quad u >> 2, four-channel group u & 3) of the fp32 copy -> patch slot `dst16`, in THREE parts (one per
    // tap of the row it rides along with: channels 0 - 1, channels 2 - 3, the eight stores), its eight half-vectors carried in `st`
    struct UnitState { f16x4 h00, h01, h10, h11, l00, l01, l10, l11; };
    auto unit = [&](int part, UnitState& st, int u, _Float16* dst16, auto interiorTag) {
        constexpr bool INTERIOR = decltype(interiorTag)::value;
        const int g4 = u & 3, q = u >> 2;                                    // neighbouring lanes: the 4 four-channel groups of one quad
        const int kr = q / QC, kc = q - kr * QC;
        if (part < 2) {
            const int e0 = 2 * part;
            if (INTERIOR) {
                const float* ta = tmp + (g4 * 4) * U5_LR_CS + kr * LR_W + kc;
#pragma unroll
                for (int e = e0; e < e0 + 2; ++e) {
                    const float a0 = ta[e * U5_LR_CS], a1 = ta[e * U5_LR_CS + 1];
                    const float b0 = ta[e * U5_LR_CS + LR_W], b1 = ta[e * U5_LR_CS + LR_W + 1];
                    const float al = isr_blend(0.75f, a0, 0.25f, a1), ar = isr_blend(0.25f, a0, 0.75f, a1);
                    const float bl = isr_blend(0.75f, b0, 0.25f, b1), br = isr_blend(0.25f, b0, 0.75f, b1);
                    _Float16 vh, vl;
                    split16x(isr_blend(0.75f, al, 0.25f, bl), vh, vl); st.h00[e] = vh; st.l00[e] = vl;
                    split16x(isr_blend(0.75f, ar, 0.25f, br), vh, vl); st.h01[e] = vh; st.l01[e] = vl;
                    split16x(isr_blend(0.25f, al, 0.75f, bl), vh, vl); st.h10[e] = vh; st.l10[e] = vl;
                    split16x(isr_blend(0.25f, ar, 0.75f, br), vh, vl); st.h11[e] = vh; st.l11[e] = vl;
                }
            } else {
                const int Yu = oy0 - 1 + 2 * kr, Xl = ox0 - 1 + 2 * kc;
                const bool oku = (unsigned)Yu < (unsigned)p.H, okl = (unsigned)Xl < (unsigned)p.W;
                int y0, y1, x0, x1, t0, t1; float lyu, lyd, lxl, lxr, t;
                isr_src_index(oku ? Yu : Yu + 1, 0.5f, p.Hin, y0, y1, t);     // both rows of the pair blend these two source rows
                isr_src_index(okl ? Xl : Xl + 1, 0.5f, p.Win, x0, x1, t);
                isr_src_index(Yu, 0.5f, p.Hin, t0, t1, lyu);
                isr_src_index(Yu + 1, 0.5f, p.Hin, t0, t1, lyd);
                isr_src_index(Xl, 0.5f, p.Win, t0, t1, lxl);
                isr_src_index(Xl + 1, 0.5f, p.Win, t0, t1, lxr);
                const float hyu = 1.f - lyu, hyd = 1.f - lyd, hxl = 1.f - lxl, hxr = 1.f - lxr;
                y0 = min(max(y0 - ly0, 0), LR_H - 1); y1 = min(max(y1 - ly0, 0), LR_H - 1);
                x0 = min(max(x0 - lx0, 0), LR_W - 1); x1 = min(max(x1 - lx0, 0), LR_W - 1);
                const float* ta = tmp + (g4 * 4) * U5_LR_CS + y0 * LR_W;
                const float* tb = tmp + (g4 * 4) * U5_LR_CS + y1 * LR_W;
#pragma unroll
                for (int e = e0; e < e0 + 2; ++e) {
                    const float a0 = ta[e * U5_LR_CS + x0], a1 = ta[e * U5_LR_CS + x1];
                    const float b0 = tb[e * U5_LR_CS + x0], b1 = tb[e * U5_LR_CS + x1];
                    const float al = isr_blend(hxl, a0, lxl, a1), ar = isr_blend(hxr, a0, lxr, a1);
                    const float bl = isr_blend(hxl, b0, lxl, b1), br = isr_blend(hxr, b0, lxr, b1);
                    _Float16 vh, vl;
                    split16x(isr_blend(hyu, al, lyu, bl), vh, vl); st.h00[e] = vh; st.l00[e] = vl;
                    split16x(isr_blend(hyu, ar, lyu, br), vh, vl); st.h01[e] = vh; st.l01[e] = vl;
                    split16x(isr_blend(hyd, al, lyd, bl), vh, vl); st.h10[e] = vh; st.l10[e] = vl;
                    split16x(isr_blend(hyd, ar, lyd, br), vh, vl); st.h11[e] = vh; st.l11[e] = vl;
                }
            }
            return;
        }
        if (!INTERIOR) {
            const int Yu = oy0 - 1 + 2 * kr, Xl = ox0 - 1 + 2 * kc;
            const bool oku = (unsigned)Yu < (unsigned)p.H, okd = (unsigned)(Yu + 1) < (unsigned)p.H;
            const bool okl = (unsigned)Xl < (unsigned)p.W, okr = (unsigned)(Xl + 1) < (unsigned)p.W;
            const f16x4 z = {(_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f};
            if (!(oku && okl)) { st.h00 = z; st.l00 = z; }
            if (!(oku && okr)) { st.h01 = z; st.l01 = z; }
            if (!(okd && okl)) { st.h10 = z; st.l10 = z; }
            if (!(okd && okr)) { st.h11 = z; st.l11 = z; }
        }
        // 16-byte unit (8-channel group g4 / 2, pixel) holds 8 halves: this 4-channel group is its half (g4 & 1)
        _Float16* d = dst16 + ((size_t)((g4 >> 1) * SP_PIX + (2 * kr) * SP_W + 2 * kc)) * 8 + (g4 & 1) * 4;
        *reinterpret_cast<f16x4*>(d) = st.h00;
        *reinterpret_cast<f16x4*>(d + 8) = st.h01;
        *reinterpret_cast<f16x4*>(d + SP_W * 8) = st.h10;
        *reinterpret_cast<f16x4*>(d + SP_W * 8 + 8) = st.h11;
        *reinterpret_cast<f16x4*>(d + U5_PART * 8) = st.l00;
        *reinterpret_cast<f16x4*>(d + U5_PART * 8 + 8) = st.l01;
        *reinterpret_cast<f16x4*>(d + (U5_PART + SP_W) * 8) = st.l10;
        *reinterpret_cast<f16x4*>(d + (U5_PART + SP_W) * 8 + 8) = st.l11;
    };
    auto unit_whole = [&](int u, _Float16* dst16, auto interiorTag) {
        UnitState st;
        unit(0, st, u, dst16, interiorTag); unit(1, st, u, dst16, interiorTag); unit(2, st, u, dst16, interiorTag);
    };
    // the second pass of a k-step's 340 units: units 256 .. 339 on waves 0 and 1 (lanes beyond the last unit repeat unit 339: the same
    // bytes to the same place, no divergence inside the tap row)
    int u2 = min(tid + S_THREADS, 4 * UQ - 1);
    const bool second = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6) < 2;      // (wave-uniform, in a scalar register)

    f32x16 acc[2][2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[cb][r][i] = 0.0f;

    // one tap row of MFMAs on patch slot `ps` / weight slot `ws`; `work()` is this row's share of the staging, in the same basic block
    auto row = [&](int dy, const u32x4* ps, const u32x4* wsl, auto work) {
        const u32x4* wl = wsl + h * 64 + j;
        const u32x4* bl = ps + h * SP_PIX + (wave * 2 + dy) * SP_W + j;
        {   // (no ablation switch around the MFMAs here: a branch would end the basic block the staging is to be scheduled into)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const f16x8 a0h = __builtin_bit_cast(f16x8, wl[dx * 128]);
                const f16x8 a0l = __builtin_bit_cast(f16x8, wl[U5_WROW + dx * 128]);
                const f16x8 a1h = __builtin_bit_cast(f16x8, wl[dx * 128 + 32]);
                const f16x8 a1l = __builtin_bit_cast(f16x8, wl[U5_WROW + dx * 128 + 32]);
                const f16x8 a0s = a0h * (_Float16)0.00048828125f;           // w_hi 2^-11: partner of the scaled x_lo'
                const f16x8 a1s = a1h * (_Float16)0.00048828125f;
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const f16x8 bh = __builtin_bit_cast(f16x8, bl[r * SP_W + dx]);
                    const f16x8 bo = __builtin_bit_cast(f16x8, bl[U5_PART + r * SP_W + dx]);
                    acc[0][r] = mfma16(a0l, bh, acc[0][r]);
                    acc[0][r] = mfma16(a0s, bo, acc[0][r]);
                    acc[0][r] = mfma16(a0h, bh, acc[0][r]);
                    acc[1][r] = mfma16(a1l, bh, acc[1][r]);
                    acc[1][r] = mfma16(a1s, bo, acc[1][r]);
                    acc[1][r] = mfma16(a1h, bh, acc[1][r]);
                }
                work(dx);                                                    // a third of the row's staging share, scheduled among this tap's 12 MFMAs
                if (U5_FENCE) __builtin_amdgcn_sched_barrier(0);            // ... and not further up: the next tap's fragments are not fetched early
            }
        }
    };
    auto nothing = [](int) {};

    // ---- prologue: k-step 0's patch (nobody multiplies yet: the other workgroup of the CU does), tap row 0's weights -----------------
    lfetch(0);
    wfetch(0);
    lpark(0, 3);
    __syncthreads();
    if (!(p.dbg & 2)) {
        _Float16* const d0 = reinterpret_cast<_Float16*>(patch);
        if (interior) { unit_whole(tid, d0, std::true_type()); if (second) unit_whole(u2, d0, std::true_type()); }
        else { unit_whole(tid, d0, std::false_type()); if (second) unit_whole(u2, d0, std::false_type()); }
    }
    wpark(0);
    if (p.ksteps > 1) lfetch(16);
    wfetch(1);
    __syncthreads();                                                         // patch slot 0 and weight slot 0 complete, the fp32 copy free

#pragma unroll 1
    for (int ks = 0; ks < p.ksteps; ++ks) {
        const u32x4* const pcur = patch + (ks & 1) * U5_PUNITS;
        _Float16* const dnext = reinterpret_cast<_Float16*>(patch + ((ks + 1) & 1) * U5_PUNITS);
        const bool more = ks + 1 < p.ksteps;
        const bool stage = more && !(p.dbg & 2);
        UnitState ust;
        // Everything derived from the thread index -- fragment, staging and parking addresses of eight inlined tap rows -- is loop
        // invariant; hoisted out of the k-loop it does not fit beside the accumulators (64 registers spilled, reloaded inside the rows:
        // 1.15 ms for the two launches against 0.67).  An opaque copy per k-step makes the compiler derive it again, ~60 instructions.
        asm volatile("" : "+v"(tid));
        lane = tid & 63; wave = tid >> 6; j = lane & 31; h = lane >> 5;
        u2 = min(tid + S_THREADS, 4 * UQ - 1);
        // tap row 0: + park k-step ks + 1's low-resolution region (requested one k-step ago)
        if (more) row(0, pcur, wbuf + ((3 * ks) & 1) * U5_WUNITS, [&](int dx) { lpark(dx, dx + 1); });
        else row(0, pcur, wbuf + ((3 * ks) & 1) * U5_WUNITS, nothing);
        wpark((3 * ks + 1) & 1);
        wfetch(3 * ks + 2);
        __syncthreads();
        // tap row 1: + the first 256 staging units of k-step ks + 1
        if (!stage) row(1, pcur, wbuf + ((3 * ks + 1) & 1) * U5_WUNITS, nothing);
        else if (interior) row(1, pcur, wbuf + ((3 * ks + 1) & 1) * U5_WUNITS, [&](int dx) { unit(dx, ust, tid, dnext, std::true_type()); });
        else row(1, pcur, wbuf + ((3 * ks + 1) & 1) * U5_WUNITS, [&](int dx) { unit(dx, ust, tid, dnext, std::false_type()); });
        wpark((3 * ks + 2) & 1);
        wfetch(3 * ks + 3);
        __syncthreads();
        // tap row 2: + the other 84 units (waves 0 and 1) and the request for k-step ks + 2's region
        if (!stage || !second) row(2, pcur, wbuf + ((3 * ks + 2) & 1) * U5_WUNITS, nothing);
        else if (interior) row(2, pcur, wbuf + ((3 * ks + 2) & 1) * U5_WUNITS, [&](int dx) { unit(dx, ust, u2, dnext, std::true_type()); });
        else row(2, pcur, wbuf + ((3 * ks + 2) & 1) * U5_WUNITS, [&](int dx) { unit(dx, ust, u2, dnext, std::false_type()); });
        if (ks + 2 < p.ksteps) lfetch(16 * (ks + 2));
        if (more) {
            wpark((3 * ks + 3) & 1);
            wfetch(3 * ks + 4);
        }
        __syncthreads();                                                     // patch slot (ks + 1) & 1 complete; slot ks & 1, the fp32 copy and the weight slots' turn free
    }

    if (p.dbg & 8) {
        if (acc[0][0][0] == 123.456f) p.ps[0] = u32x4{1u, 2u, 3u, 4u};       // (keeps the accumulators alive)
    } else if (p.ps) split_epilogue_ps(p, acc, oy0, ox0, 0, true, wave, j, h);
    else split_epilogue<true>(p, acc, patch, n, oy0, ox0, 0, true, lane, wave, j, h);
}

} // namespace

// ISR_UPS_FORM=5 takes 64 -> 64 layers (as both of EnhanceNet's are); the fp32 epilogue is compiled for quads only, as in sr_conv_ups3.h
static bool isr_split_ups5_takes(const SplitConvParams& p)
{
    if (p.Cin <= 0 || (p.Cin & 15) || p.coutPad != 64 || p.Cout != 64 || p.cgroups != 1 || p.xps || p.stamps) return false;
    return p.ps || !((p.W | p.yPlane | p.rPlane) & 3);
}

static int isr_launch_split_ups5(const SplitConvParams& p, unsigned nwg, hipStream_t s, hipEvent_t e0, hipEvent_t e1)
{
    if (!isr_split_ups5_takes(p)) return -1;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)conv3x3_split_ups5_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, U5_LDS_BYTES); attr = true; }
    if (e0 || e1) hipExtLaunchKernelGGL(conv3x3_split_ups5_kernel, dim3(nwg), dim3(S_THREADS), U5_LDS_BYTES, s, e0, e1, 0, p);
    else hipLaunchKernelGGL(conv3x3_split_ups5_kernel, dim3(nwg), dim3(S_THREADS), U5_LDS_BYTES, s, p);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
