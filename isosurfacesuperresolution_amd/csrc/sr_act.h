// The activation of an epilogue as a compile-time constant (round 6).  Included by sr_split_common.h and sr_conv3x3.hip.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/isr_sr_kernels.h"

// The activation as a compile-time constant: with `p.act` tested per value the compiler kept the test -- three scalar compares and
// taken branches around every one of a wave's 64 output values (round 6, the disassembly of the 1080p layer's epilogue: 148
// instructions per group of four values).  The epilogues switch ONCE and run straight-line code.
template <int ACT>
__device__ __forceinline__ float isr_activate(float v, float slope)
{
    if (ACT == ISR_ACT_RELU) return v > 0.f ? v : 0.f;
    if (ACT == ISR_ACT_LEAKY) return v > 0.f ? v : v * slope;
    return v;
}

// `f` called with the activation as a type (`[&](auto A) { constexpr int ACT = decltype(A)::value; ... }`): ONE switch in front of an
// epilogue's unrolled loops instead of one per value
template <int V> struct isr_act_tag { static constexpr int value = V; };
template <typename F>
__device__ __forceinline__ void isr_with_act(int act, F&& f)
{
    if (act == ISR_ACT_RELU) f(isr_act_tag<ISR_ACT_RELU>{});
    else if (act == ISR_ACT_LEAKY) f(isr_act_tag<ISR_ACT_LEAKY>{});
    else if (act == ISR_ACT_GATE) f(isr_act_tag<ISR_ACT_GATE>{});
    else f(isr_act_tag<ISR_ACT_NONE>{});
}

