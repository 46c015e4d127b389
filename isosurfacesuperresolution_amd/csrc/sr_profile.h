// Per-dispatch timing shared by the translation units of libisr_sr.so (bench.py reads it through isrProfile*):
// while profiling is enabled, a launcher asks for a start/stop event pair that rides on its dispatch packet
// (hipExtLaunchKernelGGL) and registers the launch's kernel variant and algorithmic flops.
#pragma once
#include <hip/hip_runtime.h>

constexpr int ISR_VARIANT_SPLIT = 13;       // conv3x3_split_kernel<false>
constexpr int ISR_VARIANT_SPLIT_UPS = 14;   // conv3x3_split_kernel<true>
constexpr int ISR_VARIANT_SPLIT_STREAM = 15;   // conv3x3_split_stream_kernel
constexpr int ISR_VARIANT_SPLIT_WIDE = 16;     // conv3x3_split_wide_kernel
constexpr int ISR_VARIANT_SPLIT_ROWS2 = 17;    // conv3x3_split_rows2_kernel
constexpr int ISR_VARIANT_SPLIT_TAIL = 18;     // conv3x3_split_tail_kernel (sr_conv_tail.hip)
constexpr int ISR_VARIANT_SPLIT_BLOCK = 19;    // resblock_split_kernel (sr_conv_block.hip)
constexpr int ISR_VARIANT_SPLIT_TRUNK = 20;    // trunk_dataflow_kernel (sr_conv_trunk.hip)
constexpr int ISR_VARIANT_SPLIT_UPS3 = 21;     // conv3x3_split_ups3_kernel (sr_conv_ups3.h)
constexpr int ISR_VARIANT_SPLIT_BLOCK2 = 22;   // conv3x3_split_block2_kernel (sr_conv_block2.h)
constexpr int ISR_VARIANT_SPLIT_UPS4 = 23;     // conv3x3_split_ups4_kernel (sr_conv_ups4.h)
constexpr int ISR_VARIANT_SPLIT_TRUNK_MT = 24; // trunk_mt_kernel (sr_conv_trunk.hip)
// the frame's small kernels (no matrix work: flops = 0); registered so that bench.py can say how much of a frame is BETWEEN kernels
constexpr int ISR_VARIANT_TRUNK_PACK = 25;     // trunk_pack_input_kernel (sr_conv_trunk.hip)
constexpr int ISR_VARIANT_ASSEMBLE = 26;       // assemble_input_kernel (sr_frame.hip)
constexpr int ISR_VARIANT_TAIL_FINISH = 27;    // tail_s_finish_kernel / tail_seam_finish_kernel / tail_combine_finish_kernel (sr_conv_tail.hip)
constexpr int ISR_VARIANT_FLOW_FILL = 28;      // flow_fill_one_kernel / flow_fill_kernel (sr_frame.hip; the frame pipeline runs it on the render stream)
constexpr int ISR_VARIANT_FINISH = 29;         // finish_frame_kernel (sr_frame.hip)
constexpr int ISR_VARIANT_UPS_FRAME = 30;      // ups_frame_kernel (sr_conv_upsp.h): the one-pixel frame of a phase-decomposed upsampling layer
constexpr int ISR_VARIANT_WGRAD_SPLIT = 32;    // conv3x3_wgrad_split2_kernel / conv3x3_wgrad_split_kernel (sr_conv3x3.hip): the split-operand weight gradient of one 64 x 64 channel block
constexpr int ISR_VARIANT_SPLIT_UPSP = 31;     // conv3x3_split_upsp_kernel (sr_conv_upsp.h); NOT a "small" kernel: recorded at level 1

// Sets *e0 / *e1 to an event pair (and records the launch) when profiling is on, leaves them untouched otherwise.
void isr_profile_record(int variant, double flops, hipEvent_t* e0, hipEvent_t* e1);

// A launch that carries the event pair on its dispatch packet while profiling is on (kernels without template commas in their name).
#define ISR_LAUNCH_PROFILED(variant, kernel, grid, block, lds, stream, ...)                                              \
    do {                                                                                                                \
        hipEvent_t pe0_ = nullptr, pe1_ = nullptr;                                                                      \
        isr_profile_record(variant, 0.0, &pe0_, &pe1_);                                                                 \
        if (pe0_ || pe1_) hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, pe0_, pe1_, 0, __VA_ARGS__);          \
        else hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                         \
    } while (0)

// Range guard (SplitConvParams::absmax): isrSetRangeFlag(ptr) arms the NEXT launch of a split-operand kernel (any translation
// unit) with a device word that receives the bit pattern of the largest |value| it stores; the launcher takes (and clears) it.
unsigned* isr_take_range_flag();
