/*
 * isr_sr_kernels.h -- C-ABI of libisr_sr.so: the hand-written gfx950 kernels behind the
 * super-resolution network (EnhanceNet) of the reference.
 *
 * The reference has no native interface on this path: its convolutions are `nn.Conv2d`
 * modules dispatched to cuDNN through PyTorch (SuperresolutionNetwork/models/enhancenet.py:92-125,
 * forward :136-144).  These entry points are what a maintainer binds in place of those library
 * calls: plain device pointers, sizes and a hipStream_t -- no torch types.
 *
 * All tensors are fp32, NCHW, contiguous, resident on the current HIP device.
 * Every function returns 0 on success, a negative value on invalid arguments or launch failure.
 */
#ifndef ISR_SR_KERNELS_H
#define ISR_SR_KERNELS_H

#ifdef __cplusplus
extern "C" {
#endif

/* activation codes */
#define ISR_ACT_NONE 0
#define ISR_ACT_RELU 1
#define ISR_ACT_LEAKY 2   /* LeakyReLU / single-parameter PReLU with slope `slope` */
#define ISR_ACT_GATE 3    /* isrConv3x3Forward[Strided] only: y = residual > 0 ? conv + bias : 0 -- the ReLU backward of a
                             conv -> ReLU pair folded into the data-gradient launch (residual = the ReLU's output) */

/* Padded sizes of the kernel-side weight layout [9][cinPad][coutPad]. */
int isrConvCinPad(int Cin);
int isrConvCoutPad(int Cout);

/* Re-lays PyTorch conv weights w[Cout][Cin][3][3] into wprep[9][cinPad][coutPad] (zero padded).
 * transpose_flip = 0: forward weights,  tap = ky*3+kx, wprep[tap][ci][co] = w[co][ci][ky][kx].
 * transpose_flip = 1: data-gradient weights (the roles of Cin/Cout swap and the taps flip):
 *   wprep[tap][co][ci] = w[co][ci][2-ky][2-kx], sized [9][isrConvCinPad(Cout)][isrConvCoutPad(Cin)]. */
int isrConvPrepareWeights(const float* w, float* wprep, int Cout, int Cin, int transpose_flip, void* stream);

/* Fused 3x3 convolution, stride 1, zero padding 1 (replaces nn.Conv2d(...,3,padding=1) + nn.ReLU
 * (+ the residual add of enhancenet.py:141, + the preceding nn.Upsample(x2, bilinear) of :116,:119)):
 *     y = act(conv3x3(U(x), w) + bias) + residual
 * x: [N][Cin][H/u][W/u] with u = 2 if upsample2x (bilinear, align_corners=False) else 1;
 * wprep: from isrConvPrepareWeights(.., 0); bias: [Cout] or NULL; residual: [N][Cout][H][W] or NULL;
 * y: [N][Cout][H][W].  H, W are the OUTPUT sizes. */
int isrConv3x3Forward(const float* x, const float* wprep, const float* bias, const float* residual, float* y,
                      int N, int Cin, int H, int W, int Cout, int act, float slope, int upsample2x, void* stream);

/* The same convolution on tensors whose channel planes are not packed: x / y / residual are addressed as
 * base + n * image + c * plane + row * cols + col (strides in floats; plane >= rows * cols; one image must
 * stay below 2 GiB).  Lets a caller pad the plane size of the 1080p activations: with a plane of exactly
 * 1920 x 1080 x 4 B the 64 planes alias in the memory system and the conv loses ~15 %. */
int isrConv3x3ForwardStrided(const float* x, const float* wprep, const float* bias, const float* residual, float* y,
                             int N, int Cin, int H, int W, int Cout, int act, float slope, int upsample2x,
                             long long xPlane, long long xImage, long long yPlane, long long yImage,
                             long long rPlane, long long rImage, void* stream);

/* The same fused convolution for Cout <= 8 (EnhanceNet's final 64 -> 6 layer, enhancenet.py:124) on
 * 4x4x1 MFMA blocks (4 channels x 64 pixels per instruction) instead of the 32-row MFMA tile.
 * isrConvSmallPrepare re-lays w[Cout][Cin][3][3] into w8 (isrConvSmallWeightFloats(Cin) floats, a layout
 * private to the kernel) and bias into bias8[8] (zero padded); no upsampling variant.
 * isrConvSmallCinPad is kept for callers that sized buffers with it (Cin rounded up to 8). */
int isrConvSmallCinPad(int Cin);
long long isrConvSmallWeightFloats(int Cin);
int isrConvSmallPrepare(const float* w, const float* bias, float* w8, float* bias8, int Cout, int Cin, void* stream);
int isrConv3x3SmallCout(const float* x, const float* w8, const float* bias8, const float* residual, float* y,
                        int N, int Cin, int H, int W, int Cout, int act, float slope, void* stream);
/* ... with a strided input (see isrConv3x3ForwardStrided); y and residual are packed. */
int isrConv3x3SmallCoutStrided(const float* x, const float* w8, const float* bias8, const float* residual, float* y,
                               int N, int Cin, int H, int W, int Cout, int act, float slope,
                               long long xPlane, long long xImage, void* stream);

/* SPLIT-OPERAND form of the same fused convolution: fp32-equivalent accuracy on the fp16 matrix pipe (the default
 * inference path; what it replaces is the cuDNN convolution behind nn.Conv2d in
 * SuperresolutionNetwork/models/enhancenet.py:92-125, as isrConv3x3Forward does).  Every fp32 operand v is split into
 * two fp16 numbers hi = RN16(v), lo = RN16(v - hi) (22 significand bits) and x*w becomes the three products
 * x_hi*w_hi + x_hi*w_lo + x_lo*w_hi on v_mfma_f32_32x32x16_f16, accumulated in fp32; the weights are pre-scaled by a
 * per-layer power of two (undone exactly in the epilogue) so that w_lo stays a normal fp16 number.  Error vs an fp64
 * convolution: that of the exact fp32 kernel.  Inputs must satisfy |x| < 65520 (larger values become inf, loudly).
 * Same tensors / strides / act / residual / upsample2x conventions as isrConv3x3ForwardF16 below; wq: prepared by
 * isrConvSplitPrepare into isrConvSplitWeightBytes(Cin, Cout) bytes of device memory (layout private to the kernel). */
long long isrConvSplitWeightBytes(int Cin, int Cout);
int isrConvSplitPrepare(const float* w, void* wq, int Cout, int Cin, void* stream);
/* The same preparation for up to isrConvSplitPrepareManyMax() layers in two launches: per layer the forward image
 * (wqForward[l], isrConvSplitWeightBytes(cin, cout) bytes, may be NULL) and the image of the data-gradient convolution
 * w'[ci][co][ky][kx] = w[co][ci][2-ky][2-kx] (wqBackward[l], isrConvSplitWeightBytes(cout, cin) bytes, may be NULL).  Training
 * re-prepares every layer after every optimizer step.  Pointer arrays are host arrays.  0 ok, -1 bad arguments, -2 launch failure. */
int isrConvSplitPrepareManyMax(void);
int isrConvSplitPrepareMany(int n, const float* const* w, void* const* wqForward, void* const* wqBackward, const int* cout, const int* cin, void* stream);
int isrConv3x3ForwardSplit(const float* x, const void* wq, const float* bias, const float* residual, float* y,
                           int N, int Cin, int H, int W, int Cout, int act, float slope, int upsample2x,
                           long long xPlane, long long xImage, long long yPlane, long long yImage,
                           long long rPlane, long long rImage, void* stream);

/* HALF-PRECISION FAST MODE of the same fused convolution (inference; not the parity path -- SURVEY.md 8(d) lets a
 * reduced-precision mode be reported separately, judged by PSNR): operands rounded to fp16 (saturating) on their way
 * into LDS, v_mfma_f32_32x32x16_f16 with fp32 accumulation, fp32 tensors in memory as above (strides in floats, rows packed).  wq: prepared by
 * isrConvF16Prepare into isrConvF16WeightBytes(Cin, Cout) bytes of device memory (layout private to the kernel).
 * act: ISR_ACT_NONE / RELU / LEAKY. */
long long isrConvF16WeightBytes(int Cin, int Cout);
int isrConvF16Prepare(const float* w, void* wq, int Cout, int Cin, void* stream);
int isrConv3x3ForwardF16(const float* x, const void* wq, const float* bias, const float* residual, float* y,
                         int N, int Cin, int H, int W, int Cout, int act, float slope, int upsample2x,
                         long long xPlane, long long xImage, long long yPlane, long long yImage,
                         long long rPlane, long long rImage, void* stream);
/* The same kernel with bf16 operands (same buffer sizes and calling convention; wq from isrConvBf16Prepare): the
 * mixed-precision TRAINING mode uses it for the forward and data-gradient convolutions, whose operands (gradients)
 * span the fp32 exponent range -- fp16 would flush the small ones to zero.  Both variants also accept
 * act = ISR_ACT_GATE (y = residual > 0 ? conv + bias : 0). */
int isrConvBf16Prepare(const float* w, void* wq, int Cout, int Cin, void* stream);
int isrConv3x3ForwardBf16(const float* x, const void* wq, const float* bias, const float* residual, float* y,
                          int N, int Cin, int H, int W, int Cout, int act, float slope, int upsample2x,
                          long long xPlane, long long xImage, long long yPlane, long long yImage,
                          long long rPlane, long long rImage, void* stream);
/* upsample2x = 1 (x: [N][Cin][H/2][W/2], fused bilinear x2) needs W/2 % 4 == 0, plane / image strides % 4 == 0 and a
 * 16-byte aligned x (else -3): this predicate says so beforehand. */
int isrConvF16SupportsUpsample(long long x_address, int Win, long long xPlane, long long xImage);

/* Weight gradient of the same convolution: dw[Cout][Cin][3][3] = sum_{n,y,x} gz[n][co][y][x] * x[n][ci][y+ky-1][x+kx-1]
 * and db[Cout] = sum gz.  x: [N][Cin][H][W], gz: [N][Cout][H][W] (gradient w.r.t. the pre-activation).
 * workspace: at least isrConvWeightGradWorkspace(...) bytes of device memory. dw/db are overwritten. */
long long isrConvWeightGradWorkspace(int N, int Cin, int H, int W, int Cout);
int isrConv3x3WeightGrad(const float* x, const float* gz, float* dw, float* db, void* workspace,
                         int N, int Cin, int H, int W, int Cout, void* stream);

/* The same weight (and bias) gradient summed over `segments` (<= isrConvWeightGradMaxSegments()) pairs of tensors
 * xs[k]: [N][Cin][H][W], gzs[k]: [N][Cout][H][W] in ONE pass: the T frames of a training clip
 * (SuperresolutionNetwork/mainVideoUnshaded.py:416-466) share the network's weights, so instead of T launches per
 * layer followed by T-1 accumulations the clip's weight gradient is one launch over all frames' pixels.  xs / gzs
 * are HOST arrays of device pointers (copied into the kernel arguments). */
int isrConvWeightGradMaxSegments(void);
int isrConv3x3WeightGradSegments(const float* const* xs, const float* const* gzs, int segments, float* dw, float* db, void* workspace,
                                 int N, int Cin, int H, int W, int Cout, void* stream);
/* ...Split: the same sums on SPLIT operands -- every gz and x value as two fp16 numbers, three fp16 MFMAs per product,
 * fp32 accumulation (see isrConv3x3ForwardSplit): the fp32 kernel's accuracy against fp64 at 5.3x fewer matrix cycles;
 * gz is scaled by a power of two taken from its maximum over the launch (two small kernels) and unscaled exactly by the
 * slab reduction.  The training default for layers with many tiles.  W % 4 == 0 and 16-byte aligned gzs[k] (else -3 / -1).
 * ...Bf16: the same sums with bf16 MFMA operands (fp32 accumulation and results; bias gradient from the fp32 values): the
 * weight-gradient half of the opt-in mixed-precision training mode.  W % 4 == 0 and 16-byte aligned gzs[k] (else -3 / -1). */
int isrConv3x3WeightGradSegmentsSplit(const float* const* xs, const float* const* gzs, int segments, float* dw, float* db, void* workspace,
                                      int N, int Cin, int H, int W, int Cout, void* stream);
/* Per-wave maxima of the NEXT isrConv3x3ForwardSplit launch (training: a data gradient that will be a weight gradient's gz operand):
 * `words` -> a zeroed device array of `capacity` words; the launch, if its kernel form supports it and 4 x its workgroups fit, leaves in
 * word 4 w + v the bit pattern of the largest |value| wave v of workgroup w stored.  isrTakeMaxSlotWords() -> the number of words the
 * last armed launch used (0: not supported / did not fit -- make the pass over the tensor instead). */
void isrSetMaxSlots(void* words, int capacity);
int isrTakeMaxSlotWords(void);
/* ...SplitMax: as ...Split, with the maxima of the gz tensors supplied: gzmax[k] (HOST array of device pointers) -> maxWords words
 * whose maximum is the bit pattern of max |gzs[k]|, as left by the kernel that produced the tensor (isrResBlockSmall's zmax /
 * ymax, isrSetMaxSlots: one word per wave); the pass over gz that ...Split makes to find its scale is skipped.  gzmax == NULL: exactly ...Split. */
int isrConv3x3WeightGradSegmentsSplitMax(const float* const* xs, const float* const* gzs, const void* const* gzmax, int maxWords, int segments,
                                         float* dw, float* db, void* workspace, int N, int Cin, int H, int W, int Cout, void* stream);
/* isrSetWeightGradAccumulate(bits) arms the NEXT ...SegmentsSplit / ...SplitMax call: bit 0 -> dw += the gradient, bit 1 -> db += (instead
 * of =), i.e. the slab reduction adds straight into a parameter's .grad; the same bits as the call followed by `grad += dw`. */
void isrSetWeightGradAccumulate(int bits);
int isrConv3x3WeightGradSegmentsBf16(const float* const* xs, const float* const* gzs, int segments, float* dw, float* db, void* workspace,
                                     int N, int Cin, int H, int W, int Cout, void* stream);

/* x2 bilinear upsampling, align_corners=False (nn.Upsample(scale_factor=2, mode='bilinear') of
 * SuperresolutionNetwork/models/enhancenet.py:116,119 when it is not fused into the following convolution, i.e. in
 * training) and its adjoint.  x / gx: [planes][h][w], y / gy: [planes][2h][2w], packed; w even.  The adjoint is a
 * gather (no atomics: bitwise reproducible). */
int isrUpsample2xForward(const float* x, float* y, long long planes, int h, int w, void* stream);
int isrUpsample2xBackward(const float* gy, float* gx, long long planes, int h, int w, void* stream);

/* Residual reconstruction of the network output (SuperresolutionNetwork/models/enhancenet.py:65-78, reconType='residual'):
 * out[n][c] = y[n][c] + bilinear_x4(x[n][c]) (align_corners=False) for c < k, out[n][c] = y[n][c] for k <= c < Cout -- the
 * slice / F.interpolate / add / cat of the reference as one launch.  y, out: [N][Cout][4h][4w], x: [N][Cin][h][w], packed.
 * The backward w.r.t. y is the identity; isrReconResidualBackward writes the one w.r.t. x: gx [N][Cin][h][w], the adjoint of
 * the resize for channels < k (a gather: bitwise reproducible), zero for the others.  0 ok, -1 bad arguments, -2 launch failure. */
int isrReconResidualForward(const float* y, const float* x, float* out, int N, int Cout, int Cin, int k, int h, int w, void* stream);
int isrReconResidualBackward(const float* gy, float* gx, int N, int Cout, int Cin, int k, int h, int w, void* stream);

/* LossNetUnshaded (SuperresolutionNetwork/losses/lossnet_unshaded.py:236-388) for the l1 / mse / temp-l2 terms on
 * mask / normal / ao / depth / colour, fused: one pass forward (+ a one-workgroup reduction), one pass backward.
 * gt, pred, prev: [N][6][H][W] (mask, normal xyz, depth, ao), W % 4 == 0; prev = warped previous prediction or NULL
 * (then the temp-l2 terms are skipped).  A border of `pad` pixels is treated as zeroed in all three (the module's
 * `pad`).  weights15 / enabled bit: index kind * 5 + target, kind 0 mse, 1 l1, 2 temp-l2, target 0 mask, 1 normal,
 * 2 ao, 3 depth, 4 colour.  shading12: ambient*material, diffuse*material, light direction, background (specular
 * is off in the loss shader, lossnet_unshaded.py:116-126).  values16 (device): the 15 term means (0 for terms that
 * are not enabled) and [15] = sum of weight * mean.  workspace: isrLossUnshadedWorkspace() bytes.
 * Backward: gvalues16 (device) is d L / d values16, of which only [15] is read; writes gpred (and gprev unless
 * NULL), both [N][6][H][W]. */
long long isrLossUnshadedWorkspace(void);
int isrLossUnshadedForward(const float* gt, const float* pred, const float* prev, int N, int H, int W, int pad,
                           const float* weights15, unsigned enabled, const float* shading12, float ao_strength, int inverse_ao,
                           void* workspace, float* values16, void* stream);
int isrLossUnshadedBackward(const float* gt, const float* pred, const float* prev, int N, int H, int W, int pad,
                            const float* weights15, unsigned enabled, const float* shading12, float ao_strength, int inverse_ao,
                            const float* gvalues16, float* gpred, float* gprev, void* stream);

/* Recurrent network input of a training clip for frames t > 0 (SuperresolutionNetwork/mainVideoUnshaded.py:436-447,
 * 463-466 with models/videotools.py:8-25,51-87), fused:
 *   previous_output = cat(clamp(p0,-1,1), normalize(p1..3), clamp(p4,0,1), clamp(p5,0,1))   p = prev_raw [B][6][4h][4w]
 *   warped    [B][6][4h][4w] = warp_upscale(previous_output, flow, 4, special_mask=True)
 *   net_input [B][101][h][w] = cat(input, flatten_high(warped, 4))
 * input: [B][5][h][w], flow: [B][2][h][w], both with an explicit batch stride in floats (views of [B][T][..] clips).
 * Backward: g_net_input / g_warped (either may be NULL) -> g_prev_raw; scratch: B*6*4h*4w floats (the gradient of
 * previous_output, accumulated with float atomics like PyTorch's grid_sampler backward). */
int isrRecurrentInputForward(const float* prev_raw, const float* input, const float* flow, float* net_input, float* warped,
                             int B, int h, int w, long long inputBatchStride, long long flowBatchStride, void* stream);
int isrRecurrentInputBackward(const float* prev_raw, const float* flow, const float* g_net_input, const float* g_warped,
                              float* scratch, float* g_prev_raw, int B, int h, int w, long long flowBatchStride, void* stream);

/* One step of Adam (torch.optim.Adam as mainVideoUnshaded.py:287-289 uses it: betas (0.9, 0.999), eps 1e-8, no weight decay) over
 * FLAT buffers of n floats: params -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps) with the moments updated first;
 * step[0] (device float) = steps taken so far, incremented by the call.  lr_dev (device, may be NULL) overrides lr. */
int isrAdamFlatStep(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long n, const float* lr_dev, float lr,
                    float beta1, float beta2, float eps, float* step, void* stream);

/* gz = gy * act'(y) for the activations above (y is the post-activation output, before the residual add). */
int isrActBackward(const float* gy, const float* y, float* gz, long long count, int act, float slope, void* stream);

/* Input assembly of one inference frame (replaces inference/loadedmodel.py:84-118,
 * models/videotools.py:8-25,51-87 and utils/initial_image.py:5-54 as ~60 PyTorch launches):
 * net_input[101][h][w] = cat(mask*2-1, normal, depth  from the renderer's HWC G-buffer,
 *                           flatten_high(warp_upscale(prev_high, flow, 4, special_mask=True), 4)).
 * flow_filled: [2][h][w] hole-filled flow; prev_high: [6][4h][4w] or NULL, in which case init_mode
 * selects initialImage: 0 "zero", 1 "unshaded", 2 "input". */
int isrAssembleInput(const float* gbuffer_hwc12, const float* flow_filled, const float* prev_high, float* net_input,
                     int h, int w, int init_mode, int ao_inverted, void* stream);
/* ... rows [row0, row1) of it only (the other rows of net_input are left as they are): a rank that super-resolves a screen strip
 * assembles its strip + halo instead of the whole frame (parallel_sr.py). */
int isrAssembleInputRows(const float* gbuffer_hwc12, const float* flow_filled, const float* prev_high, float* net_input,
                         int h, int w, int init_mode, int ao_inverted, int row0, int row1, void* stream);
/* ... and of a rectangle rows [row0, row1) x columns [col0, col1) only: a rank that super-resolves a screen TILE + halo
 * (parallel_sr.StripSuperResolution with a rows x columns grid). */
int isrAssembleInputRect(const float* gbuffer_hwc12, const float* flow_filled, const float* prev_high, float* net_input,
                         int h, int w, int init_mode, int ao_inverted, int row0, int row1, int col0, int col1, void* stream);
/* ... the whole frame straight into the dataflow trunk's workspace: the 101 channels leave PACKED-SPLIT, where isrTrunkDataflow's own packing
 * pass would have put them (isrTrunkDataflowInputLayout(101, h, w)), together with that pass's housekeeping (the planes' zero units, the
 * tiles' progress counters); the launch that follows on the same stream is isrTrunkDataflowPrepacked.  Same values, same split: the
 * trunk's result is bit-identical.  Of net_input ([101][h][w] as above) only channels 0 .. 4 are written (what isrFinishFrame / the fused
 * tail read back); channels 5 .. 100 stay UNDEFINED.  -3: the trunk does not take this size. */
int isrAssembleInputPacked(const float* gbuffer_hwc12, const float* flow_filled, const float* prev_high, float* net_input,
                           int h, int w, int init_mode, int ao_inverted, void* trunk_workspace, void* stream);

/* Hole filling of the low-res flow (channels 8,9 of the HWC G-buffer) where the mask (channel 3) is 0:
 * mask-weighted push-pull pyramid, the on-device replacement of the reference's CPU OpenCV
 * cv.inpaint call (inference/loadedmodel.py:77-82).  flow_out: [2][h][w].  Three launches. */
long long isrFlowFillWorkspace(int h, int w);
int isrFlowFill(const float* gbuffer_hwc12, float* flow_out, void* workspace, int h, int w, void* stream);
/* ... with `threads` (64..1024, a power of two) per workgroup of the two grid-wide launches instead of 1024 (the
 * single-workgroup pyramid pass in between always uses 1024).  256 = one wave per SIMD: the three
 * launches then fit beside the SR network's conv workgroups when the fill of the NEXT frame runs on a side stream
 * (1024-thread workgroups evict conv workgroups from their CU and cost the network more than the fill takes). */
int isrFlowFillEx(const float* gbuffer_hwc12, float* flow_out, void* workspace, int h, int w, int threads, void* stream);
/* The same result (bit for bit) in ONE launch: a workgroup per 64 x 64 tile pulls its own part of levels 1 .. 6 in LDS, the workgroup
 * that arrives last finishes the few cells above, every tile pushes back down on regions widened by the bilinear footprint
 * (csrc/sr_frame.hip: flow_fill_one_kernel).  `workspace` (isrFlowFillWorkspace bytes) must have been ZERO-FILLED ONCE before its first
 * use here (launch tickets live in it) and serves one stream at a time.  isrFlowFillOneSupported: images of at most 256 tiles (the
 * workgroups wait for one another inside the launch); otherwise -1 and the caller uses isrFlowFillEx.  A workgroup that waited
 * longer than 50 ms gives up and sets the error word (isrSetFlowFillErrorWord; default: a word of the workspace) -- never a hang. */
int isrFlowFillOneSupported(int h, int w);
int isrFlowFillOne(const float* gbuffer_hwc12, float* flow_out, void* workspace, int h, int w, void* stream);
void isrSetFlowFillErrorWord(unsigned* word);

/* End of one inference frame: raw[6][4h][4w] (EnhanceNet output before its residual reconstruction)
 * -> next_prev = cat(clamp(mask +- 1), normalize(normal), clamp(depth, ao in [0,1])) after
 * out[:5] += bilinear x4 of net_input[:5] (models/enhancenet.py:65-78, mainGUI.py:594-599), and
 * rgb[3][4h][4w] = ScreenSpaceShading(next_prev) (utils/shading.py:148-191; rgb may be NULL).
 * shading24: ambient[3], diffuse[3], specular[3], light(normalised)[3], material[3], background[3] (host). */
int isrFinishFrame(const float* raw, const float* net_input, float* next_prev, float* rgb, int h, int w,
                   const float* shading24, int exponent, float ao_strength, int inverse_ao, int enable_specular, void* stream);

/* The frame's last layer and isrFinishFrame in ONE launch: conv3x3(x [Cin][4h][4w], 64 -> 6) + bias feeds the per-pixel
 * finishing code directly (no [6][4h][4w] round trip through memory).  w8 / bias8 from isrConvSmallPrepare for Cout = 6;
 * x may have padded channel planes (xPlane floats apart); the other arguments are those of isrFinishFrame. */
int isrConvSmallFinishFrame(const float* x, const float* w8, const float* bias8, const float* net_input, float* next_prev, float* rgb,
                            int Cin, int h, int w, long long xPlane, const float* shading24, int exponent, float ao_strength,
                            int inverse_ao, int enable_specular, void* stream);

/* isrConv3x3ForwardSplit for one image whose result is written PACKED-SPLIT instead of fp32: every output value already as
 * the (hi, lo') fp16 pair the next split-operand layer multiplies, eight channels of a pixel per 16-byte unit,
 * ps[part: hi | lo][Cout / 8][psPlane units] (unit index y W + x; Cout a multiple of 8; no residual).  The consumer
 * (isrConvTailFinishFramePacked) stages it by LDS-DMA -- the numbers are those a consumer of the fp32 tensor would form on
 * its way in, so results do not change.  ps: 2 * (Cout / 8) * psPlane * 16 bytes, 16-byte aligned.  Returns as above. */
int isrConv3x3ForwardSplitPacked(const float* x, const void* wq, const float* bias, void* ps, int Cin, int H, int W, int Cout,
                                 int act, float slope, int upsample2x, long long xPlane, long long psPlane, void* stream);

/* ... and the consumer side for plain layers: the split-operand convolution of a PACKED-SPLIT input xps (Cin / 8 groups, xpsPlane
 * units per plane, one image), result as isrConv3x3ForwardSplit (y fp32, planes yPlane floats apart, optional residual with rPlane)
 * or, with packed_out != 0, packed-split again (y = ps, yPlane = its plane stride in units, no residual). */
int isrConv3x3ForwardSplitFromPacked(const void* xps, const void* wq, const float* bias, const float* residual, void* y, int packed_out,
                                     int Cin, int H, int W, int Cout, int act, float slope, long long xpsPlane, long long yPlane, long long rPlane,
                                     void* stream);

/* The whole low-resolution trunk of one image as ONE persistent dataflow launch (csrc/sr_conv_trunk.hip):
 *   F = relu(conv3x3(x [cin0][H][W], w[0]) + b[0]);  nblocks times  F += conv3x3(relu(conv3x3(F, w[2k+1]) + b[2k+1]), w[2k+2]) + b[2k+2]
 * (models/enhancenet.py:92-112,136-141), split-operand arithmetic, bit-identical to the per-layer launches of
 * isrConv3x3ForwardSplit.  Workgroup w owns tile w of 16 x 32 pixels through all layers and starts a layer when its 3 x 3
 * neighbourhood has finished the previous one; one workgroup per CU, all resident at once.  Up to isrTrunkDataflowMaxTiles() tiles a
 * workgroup owns ONE tile (residual stream in registers, half of every hand-over through LDS); larger images (ISR_TRUNK_MT, default on)
 * take trunk_mt_kernel: a workgroup owns every #CUs-th tile and keeps the residual stream in y.  isrTrunkDataflowSupported says whether
 * these tensors qualify.  Between the layers the activations live in the workspace in the packed-split format.
 *   x: planes xPlane floats apart (rows contiguous); y (the result F): [64][H][W], planes `plane` floats apart;
 *   wq[l]: isrConvSplitPrepare images, bias[l]: [64] device or NULL, l = 0 .. 2 nblocks;
 *   workspace: isrTrunkDataflowWorkspaceBytes(cin0, H, W) bytes of device memory, 256-byte aligned, ZERO-FILLED by the caller before
 *   its first use.  Bytes 16 .. hold the tiles' progress counters and then an error word: 1 + layer if a tile's wait timed out after
 *   50 ms in ANY launch since the caller last reset it (the launches never clear it).
 * 0 ok, -1 bad arguments, -2 launch failure, -3 unsupported shape / alignment / too many tiles. */
int isrTrunkDataflowMaxTiles(void);
long long isrTrunkDataflowWorkspaceBytes(int cin0, int H, int W);
int isrTrunkDataflowSupported(const float* x, int cin0, int H, int W, long long xPlane, long long plane);
int isrTrunkDataflow(const float* x, int cin0, long long xPlane, float* y, long long plane, const void* const* wq,
                     const float* const* bias, int nblocks, int H, int W, void* workspace, void* stream);
/* The same launch for an input that is ALREADY packed-split in the workspace (isrAssembleInputPacked on the same stream): no packing pass.
 * isrTrunkDataflowInputLayout: byte offsets into the workspace of the launch's three packed-split tensors (input, F, T; planes of
 * ((H W + 8) & ~7) 16-byte units, [hi | lo'][groups]), the input's channel groups of 8, the number of tiles. */
int isrTrunkDataflowInputLayout(int cin0, int H, int W, long long* offsets3, int* groups0, int* tiles);
int isrTrunkDataflowPrepacked(int cin0, float* y, long long plane, const void* const* wq, const float* const* bias, int nblocks, int H, int W,
                              void* workspace, void* stream);
/* Where every LATER isrTrunkDataflow launch reports a timed-out wait instead of the workspace's own word (same encoding, sticky until
 * the caller clears it; NULL restores the workspace word): one device word the caller can mirror to the host once per frame together
 * with the range-guard words (isrSetRangeFlag), so that a disturbed launch is noticed a frame later, not whenever somebody looks. */
void isrSetTrunkErrorWord(unsigned* word);

/* One residual block of the trunk, y = x + conv2(relu(conv1(x) + bias1)) + bias2 (64 -> 64 -> 64 channels, one image;
 * models/enhancenet.py:18-33,108-112,139-141), in ONE launch on the split-operand arithmetic: the same products in the same
 * order as two calls of isrConv3x3ForwardSplit (bit-identical result).  The intermediate tensor stays in a per-workgroup
 * scratch (`workspace`, isrResBlockSplitWorkspaceBytes() bytes) as LDS-ready (hi, lo) units and is streamed back by LDS-DMA.
 *   x, y: [64][H][W] fp32, 16-byte aligned, planes xPlane / yPlane floats apart (multiples of 4), W a multiple of 4;
 *   wq1 / wq2: isrConvSplitPrepare(w [64][64][3][3]); bias1 / bias2: [64] device (may be NULL).
 * isrResBlockSplitSupported -> 1 if these tensors can be taken.  0 ok, -1 bad arguments, -2 launch failure, -3 unsupported. */
/* Two chained 64 -> 64 convolutions of a BATCH OF SMALL IMAGES (W <= 32, e.g. the 32 x 32 crops of mainVideoUnshaded.py's training
 * step) in ONE launch -- a residual block, z = relu(conv(x, wa) + ba), y = conv(z, wb) + bb + x (models/enhancenet.py:18-33,139-141;
 * gate == NULL), or its data gradient, z = gate > 0 ? conv(x, wa) : 0, y = conv(z, wb) + x (x = the gradient of the block's output,
 * gate = the saved relu output, wa / wb = the flipped / transposed images of the block's second / first convolution).  Bit-identical
 * to two isrConv3x3ForwardSplit launches (the intermediate's halo rows are recomputed per 2-row tile).
 *   x, gate, z, y: packed [N][64][H][W] fp32, x and y 16-byte aligned, W a multiple of 4; wa / wb: isrConvSplitPrepare(w [64][64][3][3]);
 *   ba / bb: [64] device or NULL.  isrResBlockSmallSupported -> 1 for shapes this takes (at least 64 two-row tiles).
 *   zmax / ymax (both or neither): device arrays of 4 N ceil(H / 2) words that receive, per wave, the bit pattern of max |z| / max |y| --
 *   what isrConv3x3WeightGradSegmentsSplitMax wants for a gz operand.
 * 0 ok, -1 bad arguments, -2 launch failure, -3 unsupported shape. */
int isrResBlockSmallSupported(int N, int H, int W);
int isrResBlockSmall(const float* x, const void* wa, const float* ba, const float* gate, const void* wb, const float* bb, float* z, float* y,
                     int N, int H, int W, void* zmax, void* ymax, void* stream);

long long isrResBlockSplitWorkspaceBytes(void);
int isrResBlockSplitSupported(const float* x, int H, int W, long long xPlane, long long yPlane);
int isrResBlockSplit(const float* x, const void* wq1, const float* bias1, const void* wq2, const float* bias2, float* y, void* workspace,
                     int H, int W, long long xPlane, long long yPlane, void* stream);

/* The 1080p TAIL of the network in two launches (csrc/sr_conv_tail.hip): relu(conv3x3(x [64][4h][4w], 64 -> 64) + bias6)
 * -> conv3x3(., 64 -> 6) + bias8 -> isrFinishFrame, i.e. models/enhancenet.py:119-125 (postblock.6 .. postblock.8) followed by
 * _recon_image (:51-90) and the viewer's clamp / normalise / shading (mainGUI.py:594-603), on the split-operand arithmetic of
 * isrConv3x3ForwardSplit.  The 64-channel tensor between the two convolutions never exists in memory: the last layer is
 * evaluated per tap as a 1x1 product on the first convolution's accumulators (54 partial planes in `workspace`), and the second
 * launch adds every pixel's nine shifted partials in a fixed order and finishes the frame.
 *   wq6: isrConvSplitPrepare(w6 [64][64][3][3]); wz: isrConvTailPrepare(w8 [6][64][3][3]) into isrConvTailWeightBytes() bytes;
 *   bias6 [64] (may be NULL), bias8 [6]: device; workspace: isrConvTailWorkspaceBytes(h, w) bytes of device memory;
 *   x: 16-byte aligned, channel planes xPlane floats apart (a multiple of 4); w must be a multiple of 4.
 * isrConvTailSupported(x, h, w, xPlane) -> 1 if the launch can take these tensors (else use isrConv3x3ForwardSplit +
 * isrConvSmallFinishFrame).  Returns 0 ok, -1 bad arguments, -2 launch failure, -3 unsupported shape / alignment. */
long long isrConvTailWeightBytes(void);
long long isrConvTailWorkspaceBytes(int h, int w);
int isrConvTailPrepare(const float* w8, void* wz, void* stream);
int isrConvTailSupported(const float* x, int h, int w, long long xPlane);
int isrConvTailFinishFrame(const float* x, const void* wq6, const float* bias6, const void* wz, const float* bias8, void* workspace,
                           const float* net_input, float* next_prev, float* rgb, int h, int w, long long xPlane,
                           const float* shading24, int exponent, float ao_strength, int inverse_ao, int enable_specular, void* stream);
/* ... with the 64-channel input PACKED-SPLIT (isrConv3x3ForwardSplitPacked, 64 channels, [4h][4w], xpsPlane units per plane). */
int isrConvTailFinishFramePacked(const void* xps, const void* wq6, const float* bias6, const void* wz, const float* bias8, void* workspace,
                                 const float* net_input, float* next_prev, float* rgb, int h, int w, long long xpsPlane,
                                 const float* shading24, int exponent, float ao_strength, int inverse_ao, int enable_specular, void* stream);

/* PHASE-DECOMPOSED x2-upsampling convolution (csrc/sr_conv_upsp.h): y = act(conv3x3(U2(x), w) + bias) for EnhanceNet's two
 * upsampling layers (SuperresolutionNetwork/models/enhancenet.py:113-124: nn.Upsample(scale_factor=2, mode='bilinear') + Conv2d(64, 64, 3)),
 * 64 -> 64 channels, one image, input and output PACKED-SPLIT (the layout of isrConv3x3ForwardSplitPacked: [2 parts][8 groups][plane
 * units]).  U2 is linear, so the result at the high-resolution pixel (2y + py, 2x + px) is a 3 x 3 convolution of the LOW-resolution
 * image with one of four effective weight sets: no interpolation at run time, the operands are copied into LDS as the producer left
 * them.  The output's one-pixel frame (where the convolution's zero padding drops taps) is computed exactly by a small kernel of its
 * own from the fp32 weights `w`.  Same accuracy against an fp64 convolution as isrConv3x3ForwardSplit(upsample2x = 1), not the same bits.
 *   isrConvUpsPhasePrepare(w [64][64][3][3] fp32, wq, scratch, stream): wq = isrConvUpsPhaseWeightBytes() bytes (the four stacked
 *     images, one scale), scratch = isrConvUpsPhaseScratchBytes() bytes (the effective weights in fp32; may be freed once the stream
 *     has passed this call);
 *   isrConvUpsPhase(xps [64][h][wd] packed, wq, w, bias [64] or NULL, ps [64][2h][2wd] packed, h, wd, act (NONE / RELU / LEAKY), slope,
 *     xpsPlane, psPlane (units), stream).  isrConvUpsPhaseSupported(64, 64, h, wd, xpsPlane, psPlane) -> 1 if the launch takes the shape.
 * isrPackSplit: an fp32 tensor [C][H][W] (planes xPlane floats apart, C a multiple of 8) as a packed-split tensor (what a producer's
 * packed epilogue writes).  isrTrunkDataflowPackedResult: where isrTrunkDataflow leaves its result packed-split inside its workspace.
 * Return codes as isrConv3x3ForwardSplit. */
long long isrConvUpsPhaseWeightBytes(void);
long long isrConvUpsPhaseScratchBytes(void);
int isrConvUpsPhasePrepare(const float* w, void* wq, void* scratch, void* stream);
int isrConvUpsPhaseSupported(int Cin, int Cout, int h, int wd, long long xpsPlane, long long psPlane);
int isrConvUpsPhase(const void* xps, const void* wq, const float* w, const float* bias, void* ps, int h, int wd, int act, float slope,
                    long long xpsPlane, long long psPlane, void* stream);
int isrPackSplit(const float* x, void* ps, int C, int H, int W, long long xPlane, long long psPlane, void* stream);
int isrTrunkDataflowPackedResult(int cin0, int H, int W, long long* offsetBytes, long long* planeUnits);
void isrSetTrunkPackedResult(int on);      /* 1: isrTrunkDataflow also writes that packed-split result (default 0) */
/* Rows of the 16 x 32 tile per wave in the one-tile form of isrTrunkDataflow: 2 (default; eight waves, two per SIMD) or 4 (four waves of
 * 4 rows x 64 channels, one per SIMD, accumulators in AGPRs).  Same bits either way; 4 is the slower one on MI355X.
 * Environment: ISR_TRUNK_ROWS. */
void isrSetTrunkRows(int rows);

/* Optional per-dispatch timing of isrConv3x3Forward for benchmarks: while enabled, every forward
 * dispatch carries a start/stop event pair on its own packet (no extra stream operations).
 * isrProfileEnable(1) clears the records and starts recording, (0) stops; (2) also records the frame's small kernels
 * (input assembly, trunk input packing, flow fill, tail finishing: variants 25-29, zero flops).  After synchronising the
 * stream, record i gives variant = 2*MT + upsample (MT = 1|2 M tiles of 32 output channels), the
 * algorithmic FLOPs 2*9*Cin*Cout*N*H*W of the dispatch and its duration in ms. */
int isrProfileEnable(int on);
int isrProfileCount(void);
int isrProfileGet(int i, int* variant, double* flops, float* ms);

#ifdef __cplusplus
}
#endif
#endif
