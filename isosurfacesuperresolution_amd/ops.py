"""Functional bridge between the PyTorch modules and the hand-written HIP kernels.

``conv3x3`` is the one hot op of the super-resolution network (23 of the 24 weight layers and
99.9 % of its FLOPs):

    y = act(conv3x3(U(x), w, padding=1) + bias) + residual         U = optional bilinear x2

* CUDA/HIP tensors: one launch of ``isrConv3x3Forward`` from libisr_sr.so (fp32 MFMA) through the
  C-ABI of ``include/isr_sr_kernels.h``.  Autograd is provided by hand-written kernels as well:
  the data gradient re-uses the forward kernel with flipped/transposed weights, the weight
  gradient is ``isrConv3x3WeightGrad``.  There is no fallback: if the library is missing this raises.
* CPU tensors: the reference's own CPU path, i.e. plain PyTorch ops (BASELINE config #1).
"""
import contextlib
import ctypes
import os
import warnings
import weakref

import torch
import torch.nn.functional as F

from . import _native

ACT_CODES = {'none': 0, 'relu': 1, 'leaky': 2, 'gate': 3}     # 'gate': internal (data gradient through a ReLU)
_lib = None
_warned = set()


def _warn_once(key, message):
    """A device tensor took a PyTorch (MIOpen / ATen) route instead of a kernel of this package: say so, once."""
    if key not in _warned:
        _warned.add(key)
        warnings.warn(message, RuntimeWarning, stacklevel=3)


def _bind(lib):
    """ctypes signatures of include/isr_sr_kernels.h on a loaded build of the library (product or diagnostics: the same ABI)."""
    vp, ci, cf, ll = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_longlong
    lib.isrConvCinPad.argtypes = [ci]; lib.isrConvCinPad.restype = ci
    lib.isrConvCoutPad.argtypes = [ci]; lib.isrConvCoutPad.restype = ci
    lib.isrConvPrepareWeights.argtypes = [vp, vp, ci, ci, ci, vp]; lib.isrConvPrepareWeights.restype = ci
    lib.isrConv3x3Forward.argtypes = [vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, cf, ci, vp]
    lib.isrConv3x3Forward.restype = ci
    lib.isrConv3x3ForwardStrided.argtypes = [vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, cf, ci, ll, ll, ll, ll, ll, ll, vp]
    lib.isrConv3x3ForwardStrided.restype = ci
    lib.isrConvWeightGradWorkspace.argtypes = [ci, ci, ci, ci, ci]; lib.isrConvWeightGradWorkspace.restype = ll
    lib.isrConv3x3WeightGrad.argtypes = [vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, vp]; lib.isrConv3x3WeightGrad.restype = ci
    lib.isrConvWeightGradMaxSegments.argtypes = []; lib.isrConvWeightGradMaxSegments.restype = ci
    lib.isrConv3x3WeightGradSegments.argtypes = [vp, vp, ci, vp, vp, vp, ci, ci, ci, ci, ci, vp]
    lib.isrConv3x3WeightGradSegments.restype = ci
    lib.isrConv3x3WeightGradSegmentsBf16.argtypes = lib.isrConv3x3WeightGradSegments.argtypes
    lib.isrConv3x3WeightGradSegmentsBf16.restype = ci
    lib.isrConv3x3WeightGradSegmentsSplit.argtypes = lib.isrConv3x3WeightGradSegments.argtypes
    lib.isrConv3x3WeightGradSegmentsSplit.restype = ci
    lib.isrActBackward.argtypes = [vp, vp, vp, ll, ci, cf, vp]; lib.isrActBackward.restype = ci
    lib.isrResBlockSmallSupported.argtypes = [ci, ci, ci]; lib.isrResBlockSmallSupported.restype = ci
    lib.isrResBlockSmall.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, vp, vp, vp]; lib.isrResBlockSmall.restype = ci
    lib.isrConv3x3WeightGradSegmentsSplitMax.argtypes = [vp, vp, vp, ci, ci, vp, vp, vp, ci, ci, ci, ci, ci, vp]
    lib.isrConv3x3WeightGradSegmentsSplitMax.restype = ci
    lib.isrSetMaxSlots.argtypes = [vp, ci]; lib.isrSetMaxSlots.restype = None
    lib.isrSetWeightGradAccumulate.argtypes = [ci]; lib.isrSetWeightGradAccumulate.restype = None
    lib.isrTakeMaxSlotWords.argtypes = []; lib.isrTakeMaxSlotWords.restype = ci
    lib.isrAssembleInput.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, vp]; lib.isrAssembleInput.restype = ci
    lib.isrAssembleInputRows.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, vp]; lib.isrAssembleInputRows.restype = ci
    lib.isrAssembleInputRect.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, vp]; lib.isrAssembleInputRect.restype = ci
    lib.isrAssembleInputPacked.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, vp, vp]; lib.isrAssembleInputPacked.restype = ci
    lib.isrConvSmallCinPad.argtypes = [ci]; lib.isrConvSmallCinPad.restype = ci
    lib.isrConvSmallWeightFloats.argtypes = [ci]; lib.isrConvSmallWeightFloats.restype = ll
    lib.isrConvSmallPrepare.argtypes = [vp, vp, vp, vp, ci, ci, vp]; lib.isrConvSmallPrepare.restype = ci
    lib.isrConv3x3SmallCout.argtypes = [vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, cf, vp]; lib.isrConv3x3SmallCout.restype = ci
    lib.isrConv3x3SmallCoutStrided.argtypes = [vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, cf, ll, ll, vp]
    lib.isrConv3x3SmallCoutStrided.restype = ci
    lib.isrFlowFillWorkspace.argtypes = [ci, ci]; lib.isrFlowFillWorkspace.restype = ll
    lib.isrFlowFill.argtypes = [vp, vp, vp, ci, ci, vp]; lib.isrFlowFill.restype = ci
    lib.isrFlowFillEx.argtypes = [vp, vp, vp, ci, ci, ci, vp]; lib.isrFlowFillEx.restype = ci
    lib.isrFlowFillOne.argtypes = [vp, vp, vp, ci, ci, vp]; lib.isrFlowFillOne.restype = ci
    lib.isrFlowFillOneSupported.argtypes = [ci, ci]; lib.isrFlowFillOneSupported.restype = ci
    lib.isrSetFlowFillErrorWord.argtypes = [vp]; lib.isrSetFlowFillErrorWord.restype = None
    lib.isrFinishFrame.argtypes = [vp, vp, vp, vp, ci, ci, vp, ci, cf, ci, ci, vp]; lib.isrFinishFrame.restype = ci
    lib.isrConvSmallFinishFrame.argtypes = [vp, vp, vp, vp, vp, vp, ci, ci, ci, ll, vp, ci, cf, ci, ci, vp]
    lib.isrConvSmallFinishFrame.restype = ci
    lib.isrUpsample2xForward.argtypes = [vp, vp, ll, ci, ci, vp]; lib.isrUpsample2xForward.restype = ci
    lib.isrUpsample2xBackward.argtypes = [vp, vp, ll, ci, ci, vp]; lib.isrUpsample2xBackward.restype = ci
    lib.isrReconResidualForward.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci, ci, vp]; lib.isrReconResidualForward.restype = ci
    lib.isrReconResidualBackward.argtypes = [vp, vp, ci, ci, ci, ci, ci, ci, vp]; lib.isrReconResidualBackward.restype = ci
    lib.isrLossUnshadedWorkspace.argtypes = []; lib.isrLossUnshadedWorkspace.restype = ll
    lib.isrLossUnshadedForward.argtypes = [vp, vp, vp, ci, ci, ci, ci, vp, ctypes.c_uint, vp, cf, ci, vp, vp, vp]
    lib.isrLossUnshadedForward.restype = ci
    lib.isrLossUnshadedBackward.argtypes = [vp, vp, vp, ci, ci, ci, ci, vp, ctypes.c_uint, vp, cf, ci, vp, vp, vp, vp]
    lib.isrLossUnshadedBackward.restype = ci
    lib.isrRecurrentInputForward.argtypes = [vp, vp, vp, vp, vp, ci, ci, ci, ll, ll, vp]; lib.isrRecurrentInputForward.restype = ci
    lib.isrRecurrentInputBackward.argtypes = [vp, vp, vp, vp, vp, vp, ci, ci, ci, ll, vp]; lib.isrRecurrentInputBackward.restype = ci
    lib.isrConvF16WeightBytes.argtypes = [ci, ci]; lib.isrConvF16WeightBytes.restype = ll
    lib.isrConvF16Prepare.argtypes = [vp, vp, ci, ci, vp]; lib.isrConvF16Prepare.restype = ci
    lib.isrConv3x3ForwardF16.argtypes = [vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, cf, ci, ll, ll, ll, ll, ll, ll, vp]
    lib.isrConvF16SupportsUpsample.argtypes = [ll, ci, ll, ll]; lib.isrConvF16SupportsUpsample.restype = ci
    lib.isrConvBf16Prepare.argtypes = [vp, vp, ci, ci, vp]; lib.isrConvBf16Prepare.restype = ci
    lib.isrConv3x3ForwardBf16.argtypes = lib.isrConv3x3ForwardF16.argtypes; lib.isrConv3x3ForwardBf16.restype = ci
    lib.isrConv3x3ForwardF16.restype = ci
    lib.isrConvSplitWeightBytes.argtypes = [ci, ci]; lib.isrConvSplitWeightBytes.restype = ll
    lib.isrConvSplitPrepare.argtypes = [vp, vp, ci, ci, vp]; lib.isrConvSplitPrepare.restype = ci
    lib.isrConvSplitPrepareManyMax.argtypes = []; lib.isrConvSplitPrepareManyMax.restype = ci
    lib.isrConvSplitPrepareMany.argtypes = [ci, vp, vp, vp, vp, vp, vp]; lib.isrConvSplitPrepareMany.restype = ci
    lib.isrConv3x3ForwardSplit.argtypes = lib.isrConv3x3ForwardF16.argtypes; lib.isrConv3x3ForwardSplit.restype = ci
    lib.isrConvTailWeightBytes.argtypes = []; lib.isrConvTailWeightBytes.restype = ll
    lib.isrConvTailWorkspaceBytes.argtypes = [ci, ci]; lib.isrConvTailWorkspaceBytes.restype = ll
    lib.isrConvTailPrepare.argtypes = [vp, vp, vp]; lib.isrConvTailPrepare.restype = ci
    lib.isrConvTailSupported.argtypes = [vp, ci, ci, ll]; lib.isrConvTailSupported.restype = ci
    lib.isrConvTailFinishFrame.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ll, vp, ci, cf, ci, ci, vp]
    lib.isrConvTailFinishFrame.restype = ci
    lib.isrConv3x3ForwardSplitPacked.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, ci, cf, ci, ll, ll, vp]
    lib.isrConv3x3ForwardSplitPacked.restype = ci
    lib.isrConv3x3ForwardSplitFromPacked.argtypes = [vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, cf, ll, ll, ll, vp]
    lib.isrConv3x3ForwardSplitFromPacked.restype = ci
    lib.isrConvTailFinishFramePacked.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ll, vp, ci, cf, ci, ci, vp]
    lib.isrConvTailFinishFramePacked.restype = ci
    lib.isrTrunkDataflowMaxTiles.argtypes = []; lib.isrTrunkDataflowMaxTiles.restype = ci
    lib.isrTrunkDataflowWorkspaceBytes.argtypes = [ci, ci, ci]; lib.isrTrunkDataflowWorkspaceBytes.restype = ll
    lib.isrTrunkDataflowSupported.argtypes = [vp, ci, ci, ci, ll, ll]; lib.isrTrunkDataflowSupported.restype = ci
    lib.isrTrunkDataflow.argtypes = [vp, ci, ll, vp, ll, vp, vp, ci, ci, ci, vp, vp]; lib.isrTrunkDataflow.restype = ci
    lib.isrTrunkDataflowPrepacked.argtypes = [ci, vp, ll, vp, vp, ci, ci, ci, vp, vp]; lib.isrTrunkDataflowPrepacked.restype = ci
    lib.isrTrunkDataflowPackedResult.argtypes = [ci, ci, ci, vp, vp]; lib.isrTrunkDataflowPackedResult.restype = ci
    lib.isrSetTrunkPackedResult.argtypes = [ci]; lib.isrSetTrunkPackedResult.restype = None
    lib.isrConvUpsPhaseWeightBytes.argtypes = []; lib.isrConvUpsPhaseWeightBytes.restype = ll
    lib.isrConvUpsPhaseScratchBytes.argtypes = []; lib.isrConvUpsPhaseScratchBytes.restype = ll
    lib.isrConvUpsPhasePrepare.argtypes = [vp, vp, vp, vp]; lib.isrConvUpsPhasePrepare.restype = ci
    lib.isrConvUpsPhaseSupported.argtypes = [ci, ci, ci, ci, ll, ll]; lib.isrConvUpsPhaseSupported.restype = ci
    lib.isrConvUpsPhase.argtypes = [vp, vp, vp, vp, vp, ci, ci, ci, cf, ll, ll, vp]; lib.isrConvUpsPhase.restype = ci
    lib.isrPackSplit.argtypes = [vp, vp, ci, ci, ci, ll, ll, vp]; lib.isrPackSplit.restype = ci
    lib.isrResBlockSplitWorkspaceBytes.argtypes = []; lib.isrResBlockSplitWorkspaceBytes.restype = ll
    lib.isrResBlockSplitSupported.argtypes = [vp, ci, ci, ll, ll]; lib.isrResBlockSplitSupported.restype = ci
    lib.isrResBlockSplit.argtypes = [vp, vp, vp, vp, vp, vp, vp, ci, ci, ll, ll, vp]; lib.isrResBlockSplit.restype = ci
    lib.isrAdamFlatStep.argtypes = [vp, vp, vp, vp, ll, vp, cf, cf, cf, cf, vp, vp]; lib.isrAdamFlatStep.restype = ci
    lib.isrSetRangeFlag.argtypes = [vp]; lib.isrSetRangeFlag.restype = None
    lib.isrSetTrunkErrorWord.argtypes = [vp]; lib.isrSetTrunkErrorWord.restype = None
    lib.isrSetTrunkRows.argtypes = [ci]; lib.isrSetTrunkRows.restype = None
    lib.isrProfileEnable.argtypes = [ci]; lib.isrProfileEnable.restype = ci
    lib.isrProfileCount.argtypes = []; lib.isrProfileCount.restype = ci
    lib.isrProfileGet.argtypes = [ci, vp, vp, vp]; lib.isrProfileGet.restype = ci
    if hasattr(lib, "isrDebugSplitState"):            # the diagnostics build (csrc/sr_diag.h): its switches on top
        lib.isrDebugSetFlowFillFault.argtypes = [ci, ctypes.c_ulonglong]; lib.isrDebugSetFlowFillFault.restype = None
        lib.isrDebugSetTrunkFault.argtypes = [ci, ctypes.c_ulonglong]; lib.isrDebugSetTrunkFault.restype = None
        lib.isrDebugSetTrunkMultiTile.argtypes = [ci]; lib.isrDebugSetTrunkMultiTile.restype = None
        for name in ("isrDebugSetSplitAlgo", "isrDebugSetSplitUpsForm", "isrDebugSetSplitSlots", "isrDebugSetSplitSmall", "isrDebugSetSplitAblation",
                     "isrDebugSetTailFused", "isrDebugSetForwardTile", "isrDebugSetForwardAlgo", "isrDebugSetWgradSplitForm", "isrDebugSetAblation",
                     "isrDebugSetTrunkAblation", "isrDebugSetBlockAblation", "isrDebugSetF16Ablation"):
            getattr(lib, name).argtypes = [ci]; getattr(lib, name).restype = None
        for name in ("isrDebugSplitState", "isrDebugTailState", "isrDebugBlockState", "isrDebugTrunkState", "isrDebugFlowFillState",
                     "isrDebugSplitUpsForm", "isrDebugWgradSplitForm"):
            getattr(lib, name).argtypes = []; getattr(lib, name).restype = ci
        for name in ("isrDebugSetSplitStampBuffer", "isrDebugSetTrunkStampBuffer", "isrDebugSetBlockStampBuffer", "isrDebugSetF16StampBuffer",
                     "isrDebugSetStampBuffer"):
            getattr(lib, name).argtypes = [vp]; getattr(lib, name).restype = None
    return lib


def _sr():
    global _lib
    if _lib is None:
        _lib = _bind(_native.load(_native.SR_LIB))
    return _lib


def is_diagnostics_library():
    """Is the library this process launches on the DIAGNOSTICS build (isrDebug* switches, stamp buffers, fault injection, the experimental
    kernel forms)?  The product build -- what a deployment ships and what bench.py / the driver measure -- has none of them."""
    return hasattr(_sr(), "isrDebugSplitState")


@contextlib.contextmanager
def diagnostics_library():
    """Run the enclosed launches on the DIAGNOSTICS build of the kernels (lib/libisr_sr_diag.so, `make diag`), then go back to the product
    build: the tests of the timeout paths (fault injection) and of kernel forms only a switch selects, and tools/.  The two builds are
    the same sources and the same ABI; weight images and workspaces made by one are valid for the other, the per-launch globals (range
    flag, error words, trunk rows) are set by ``ops`` at every launch.  Single-threaded by contract, like the libraries."""
    global _lib
    saved = _sr()
    if hasattr(saved, "isrDebugSplitState"):          # already there (ISR_SR_DIAG=1 / ISR_SR_LIB=...)
        yield saved
        return
    torch.cuda.synchronize()
    diag = _bind(_native.load(_native.SR_DIAG_LIB))
    assert hasattr(diag, "isrDebugSplitState"), "%s is not a diagnostics build (make -C csrc diag)" % _native.SR_DIAG_LIB
    was_profiling = _profile_on
    _lib = diag
    try:
        if was_profiling:
            diag.isrProfileEnable(1)
        yield diag
    finally:
        torch.cuda.synchronize()
        if was_profiling:
            diag.isrProfileEnable(0)
        _lib = saved


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


# Inference re-uses the re-laid-out weights for as long as the very same tensor object is alive and
# unmodified (a data_ptr alone is not an identity: the caching allocator recycles addresses).
# ``weight._version`` does not see every change: a HIP-graph replay of a training step (train.GraphedTrainStep) runs the
# captured optimizer kernels on the parameters' memory without touching the Python-side version counter.  Whoever changes
# weights behind autograd's back therefore calls ``invalidate_weight_images()``; every cached image carries the epoch
# it was made in and is rebuilt when the epoch has moved on.
_wcache = {}
_images_epoch = 0


def invalidate_weight_images():
    """Declare every cached kernel-layout weight image (exact, small-Cout, fp16 / bf16, split) stale."""
    global _images_epoch
    _images_epoch += 1


# ---- routing epoch: what a CAPTURED frame (pipeline.SuperResolutionPipeline, ISR_FRAME_GRAPH) has baked in ----------------------
# A captured HIP graph holds raw device pointers and the routing decisions of the moment it was captured: which kernel form ran
# (dataflow trunk or per-layer, one-launch or three-launch flow fill), which guard word a launch reports to, which workspace it
# spins in.  Every code path that changes one of these behind a graph's back moves this epoch (``_trunk_failed``,
# ``_fill_failed``, ``range_reset``, ``set_device_shared``); the pipeline puts it -- with the switches themselves and the weights'
# (version, address) -- into the signature that decides whether its captured frames are still valid.
_routing_epoch = 0


def routing_epoch():
    return _routing_epoch


def _routing_changed():
    global _routing_epoch
    _routing_epoch += 1


# Several processes (or several independent streams of kernels) share ONE device: the all-resident spin kernels -- the dataflow
# trunk (one workgroup per CU, neighbours wait for each other) and the one-launch flow fill (every workgroup waits for the last) --
# assume that all workgroups of a launch are resident at once; beside another process's grid they may not be, and then they wait
# for each other until the 50 ms deadline.  With the hint set those forms are refused everywhere (``trunk_supported``,
# ``fill_flow_gbuffer``) and the per-layer / three-launch forms run: bit-identical results.  ISR_DEVICE_SHARED=1 or
# ``set_device_shared(True)`` (bench.py: BENCH_SHARE_DEVICE=1, in every mode).
DEVICE_SHARED = os.environ.get("ISR_DEVICE_SHARED", "0") == "1"


def set_device_shared(on):
    global DEVICE_SHARED
    on = bool(on)
    if on != DEVICE_SHARED:
        DEVICE_SHARED = on
        _routing_changed()


def spin_kernel_forms():
    """Which of the all-resident forms a launch may take right now (bench lines and tests report it)."""
    return {"trunk_dataflow": bool(TRUNK_DATAFLOW and not DEVICE_SHARED), "flow_fill_one": bool(FLOW_FILL_ONE and not DEVICE_SHARED)}


def prepare_weights(weight, transpose_flip=False):
    """PyTorch [Cout,Cin,3,3] -> kernel layout [9][cinPad][coutPad] (device tensor)."""
    lib = _sr()
    cout, cin = weight.shape[0], weight.shape[1]
    key = (id(weight), bool(transpose_flip))
    hit = _wcache.get(key)
    if hit is not None:
        ref, version, ptr, wp, epoch = hit
        if ref() is weight and version == weight._version and ptr == weight.data_ptr() and epoch == _images_epoch:
            return wp
    w = weight.detach().contiguous()
    if transpose_flip:
        cin_pad, cout_pad = lib.isrConvCinPad(cout), lib.isrConvCoutPad(cin)
    else:
        cin_pad, cout_pad = lib.isrConvCinPad(cin), lib.isrConvCoutPad(cout)
    wp = torch.empty(9 * cin_pad * cout_pad, dtype=torch.float32, device=weight.device)
    rc = lib.isrConvPrepareWeights(_ptr(w), _ptr(wp), cout, cin, 1 if transpose_flip else 0, _stream())
    if rc != 0:
        raise RuntimeError("isrConvPrepareWeights failed (%d)" % rc)
    if len(_wcache) > 512:
        for k in [k for k, v in _wcache.items() if v[0]() is None]:
            del _wcache[k]
    _wcache[key] = (weakref.ref(weight), weight._version, weight.data_ptr(), wp, _images_epoch)
    return wp


# Optional per-dispatch timing for bench.py: the library attaches start/stop events to the
# dispatch packets themselves (isrProfile*), which does not add stream operations.
VARIANT_NAMES = {6: "conv3x3_small_cout_kernel", 2: "conv3x3_fwd_kernel<1,false>", 3: "conv3x3_fwd_kernel<1,true>",
                 4: "conv3x3_fwd_kernel<2,false>", 5: "conv3x3_fwd_kernel<2,true>",
                 8: "conv3x3_fwd2_kernel<false,4>", 9: "conv3x3_fwd2_kernel<true,4>",
                 10: "conv3x3_fwd2_kernel<false,1>", 11: "conv3x3_fwd2_kernel<true,1>", 12: "conv3x3_rowsplit_kernel",
                 13: "conv3x3_split_kernel<false>", 14: "conv3x3_split_kernel<true>",
                 15: "conv3x3_split_stream_kernel", 16: "conv3x3_split_wide_kernel", 17: "conv3x3_split_rows2_kernel",
                 18: "conv3x3_split_tail_kernel", 19: "resblock_split_kernel", 20: "trunk_dataflow_kernel",
                 21: "conv3x3_split_ups3_kernel", 22: "conv3x3_split_block2_kernel", 23: "conv3x3_split_ups4_kernel",
                 24: "trunk_mt_kernel",
                 # the frame's small kernels (zero algorithmic flops; registered for bench.py's gap accounting)
                 25: "trunk_pack_input_kernel", 26: "assemble_input_kernel", 27: "tail_finish_kernel", 28: "flow_fill_one_kernel",
                 29: "finish_frame_kernel", 30: "ups_frame_kernel", 31: "conv3x3_split_upsp_kernel",
                 32: "conv3x3_wgrad_split_kernel"}


def debug_switches():
    """Bit mask of the libraries' process-global diagnostic switches (``isrDebugSet*``: ablations that skip MFMAs, forced kernel
    forms, grid caps, stamp buffers) that are not in their default position; 0 on a clean process.  bench.py reports it and
    refuses to print a headline measured with an ablation active."""
    lib = _sr()
    if not hasattr(lib, "isrDebugSplitState"):        # the product build has no switches (csrc/sr_diag.h): 0 by construction
        return 0
    return (int(lib.isrDebugSplitState()) | (int(lib.isrDebugTailState()) << 8) | (int(lib.isrDebugBlockState()) << 12) | (int(lib.isrDebugTrunkState()) << 16)
            | (int(lib.isrDebugFlowFillState()) << 28))


_profile_on = False


def profile_enable(on, small_kernels=False):
    """``small_kernels``: also the frame's kernels without matrix work (assembly, packing, flow fill, finishing)."""
    global _profile_on
    _profile_on = bool(on)
    _sr().isrProfileEnable((2 if small_kernels else 1) if on else 0)


def profile_is_on():
    """Per-dispatch profiling puts start / stop events on the kernels' dispatch packets: not something to capture into a graph."""
    return _profile_on


def profile_records():
    """[(kernel_name, algorithmic_flops, milliseconds)] of every forward dispatch since profile_enable(True);
    call after synchronising."""
    lib = _sr()
    out = []
    v, f, ms = ctypes.c_int(), ctypes.c_double(), ctypes.c_float()
    for i in range(lib.isrProfileCount()):
        if lib.isrProfileGet(i, ctypes.byref(v), ctypes.byref(f), ctypes.byref(ms)) != 0:
            raise RuntimeError("isrProfileGet failed")
        out.append((VARIANT_NAMES[v.value], f.value, ms.value))
    return out


_small_cache = {}


def _prepare_small(weight, bias):
    """(w8, bias8) device tensors for the Cout <= 8 kernel, cached like prepare_weights."""
    lib = _sr()
    key = id(weight)
    hit = _small_cache.get(key)
    bptr = bias.data_ptr() if bias is not None else 0
    bver = bias._version if bias is not None else 0
    if hit is not None:
        ref, ver, ptr, bp, bv, w8, b8, epoch = hit
        if ref() is weight and ver == weight._version and ptr == weight.data_ptr() and bp == bptr and bv == bver and epoch == _images_epoch:
            return w8, b8
    cout, cin = weight.shape[0], weight.shape[1]
    w8 = torch.empty(lib.isrConvSmallWeightFloats(cin), dtype=torch.float32, device=weight.device)
    b8 = torch.empty(8, dtype=torch.float32, device=weight.device)
    rc = lib.isrConvSmallPrepare(_ptr(weight.detach().contiguous()), _ptr(bias.detach().contiguous()) if bias is not None else None,
                                 _ptr(w8), _ptr(b8), cout, cin, _stream())
    if rc != 0:
        raise RuntimeError("isrConvSmallPrepare failed (%d)" % rc)
    if len(_small_cache) > 64:
        _small_cache.clear()
    _small_cache[key] = (weakref.ref(weight), weight._version, weight.data_ptr(), bptr, bver, w8, b8, _images_epoch)
    return w8, b8


def _launch_small(x, weight, bias, residual, act, slope):
    lib = _sr()
    n, cin, h, w = x.shape
    cout = weight.shape[0]
    w8, b8 = _prepare_small(weight, bias)
    x, xp, xi = _plane_strides(x)
    y = torch.empty((n, cout, h, w), dtype=torch.float32, device=x.device)
    rc = lib.isrConv3x3SmallCoutStrided(_ptr(x), _ptr(w8), _ptr(b8), _ptr(residual), _ptr(y), n, cin, h, w, cout,
                                        ACT_CODES[act], float(slope), xp, xi, _stream())
    if rc != 0:
        raise RuntimeError("isrConv3x3SmallCout failed (%d)" % rc)
    return y


# Channel-plane padding of large activations.  A [64][1080][1920] fp32 tensor has planes of exactly
# 2025 x 4 KiB; when the 1080p conv reads 64 such planes and writes 64 more (same rows of every plane at the
# same time) the accesses alias in the memory system and the layer drops from 125 to 105 TFLOP/s
# (tools/lab/bench_conv_pad.py: any padding >= 16 floats on BOTH tensors cures it, padding one of them does not).
# Activations the conv kernels allocate themselves therefore get one extra row between channel planes once a
# plane reaches PLANE_PAD_MIN_BYTES; every consumer in this package takes the strides from the tensor
# (`_plane_strides`), anything else sees an ordinary strided view.
PLANE_PAD_MIN_BYTES = 4 << 20
PLANE_PAD_ROWS = 1


def plane_pad(h, w):
    return PLANE_PAD_ROWS * ((w + 3) // 4) * 4 if h * w * 4 >= PLANE_PAD_MIN_BYTES else 0


def empty_planes(n, c, h, w, device):
    """[n, c, h, w] fp32 tensor with packed rows and (possibly) padded channel planes."""
    plane = h * w + plane_pad(h, w)
    if plane == h * w:
        return torch.empty((n, c, h, w), dtype=torch.float32, device=device)
    return torch.empty(n * c * plane, dtype=torch.float32, device=device).as_strided((n, c, h, w), (c * plane, plane, w, 1))


def _plane_strides(t):
    """(tensor, plane stride, image stride) with packed rows; copies only if the rows are not packed."""
    n, c, h, w = t.shape
    if t.stride(3) == 1 and t.stride(2) == w and t.stride(1) >= h * w and (n == 1 or t.stride(0) >= c * t.stride(1)):
        return t, t.stride(1), t.stride(0) if n > 1 else c * t.stride(1)
    t = t.contiguous()
    return t, h * w, c * h * w


def _launch_forward(x, wprep, bias, residual, cin, cout, act, slope, upsample2x, packed=False):
    """``packed``: the result is an ordinary contiguous tensor (the autograd paths hand raw pointers of saved outputs
    to kernels that index them flat); otherwise large channel planes are padded (``empty_planes``)."""
    lib = _sr()
    n, _, hin, win = x.shape
    h, w = (hin * 2, win * 2) if upsample2x else (hin, win)
    x, xp, xi = _plane_strides(x)
    rp = ri = 0
    if residual is not None:
        residual, rp, ri = _plane_strides(residual)
    y = torch.empty((n, cout, h, w), dtype=torch.float32, device=x.device) if packed else empty_planes(n, cout, h, w, x.device)
    rc = lib.isrConv3x3ForwardStrided(_ptr(x), _ptr(wprep), _ptr(bias), _ptr(residual), _ptr(y),
                                      n, cin, h, w, cout, ACT_CODES[act], float(slope), 1 if upsample2x else 0,
                                      xp, xi, y.stride(1), cout * y.stride(1), rp, ri, _stream())
    if rc != 0:
        raise RuntimeError("isrConv3x3ForwardStrided failed (%d)" % rc)
    return y


# ---- fp16 fast mode (inference only, NOT the parity path) ------------------------------------------------------
# FAST_F16 = True routes the no-grad ``conv3x3`` of layers with more than 8 output channels through
# ``isrConv3x3ForwardF16`` (operands rounded to fp16, fp32 accumulation, fp32 tensors).  Off by default: the 1e-4
# parity with the reference's CPU path is a property of the fp32 kernels.  bench.py reports it separately with its PSNR.
FAST_F16 = False
_f16_cache = {}


def _prepare_lp(weight, transpose_flip=False, bf16=False):
    """Weights in the low-precision kernel's layout (fp16, or bf16 for the training mode); transpose_flip: the
    data-gradient weights w'[ci][co][ky][kx] = w[co][ci][2-ky][2-kx]."""
    lib = _sr()
    key = (id(weight), bool(transpose_flip), bool(bf16))
    hit = _f16_cache.get(key)
    if hit is not None:
        ref, version, ptr, wq, epoch = hit
        if ref() is weight and version == weight._version and ptr == weight.data_ptr() and epoch == _images_epoch:
            return wq
    w = weight.detach()
    if transpose_flip:
        w = w.flip(2, 3).transpose(0, 1)
    w = w.contiguous()
    cout, cin = w.shape[0], w.shape[1]
    wq = torch.empty(lib.isrConvF16WeightBytes(cin, cout), dtype=torch.uint8, device=weight.device)
    rc = (lib.isrConvBf16Prepare if bf16 else lib.isrConvF16Prepare)(_ptr(w), _ptr(wq), cout, cin, _stream())
    if rc != 0:
        raise RuntimeError("isrConv%sPrepare failed (%d)" % ("Bf16" if bf16 else "F16", rc))
    if len(_f16_cache) > 512:
        for k in [k for k, v in _f16_cache.items() if v[0]() is None]:
            del _f16_cache[k]
    _f16_cache[key] = (weakref.ref(weight), weight._version, weight.data_ptr(), wq, _images_epoch)
    return wq


def _prepare_f16(weight):
    return _prepare_lp(weight)


def _launch_lp(x, wq, bias, residual, cout, act, slope, upsample2x, bf16, packed=False):
    """One launch of isrConv3x3ForwardF16 / Bf16 (x: fp32 NCHW, channel planes may be padded; ``packed`` as in
    ``_launch_forward``)."""
    lib = _sr()
    x, xp, xi = _plane_strides(x)
    fuse = False
    if upsample2x:
        fuse = bool(lib.isrConvF16SupportsUpsample(x.data_ptr(), x.shape[3], xp, xi))
        if not fuse:          # unaligned low-res rows: the resize runs as its own kernel first
            x, xp, xi = _plane_strides(bilinear_upsample2x(x))
    n, cin = x.shape[0], x.shape[1]
    h, w = (2 * x.shape[2], 2 * x.shape[3]) if fuse else (x.shape[2], x.shape[3])
    rp = ri = 0
    if residual is not None:
        residual, rp, ri = _plane_strides(residual)
    y = torch.empty((n, cout, h, w), dtype=torch.float32, device=x.device) if packed else empty_planes(n, cout, h, w, x.device)
    fn = lib.isrConv3x3ForwardBf16 if bf16 else lib.isrConv3x3ForwardF16
    rc = fn(_ptr(x), _ptr(wq), _ptr(bias), _ptr(residual), _ptr(y), n, cin, h, w, cout, ACT_CODES[act], float(slope),
            1 if fuse else 0, xp, xi, y.stride(1), cout * y.stride(1), rp, ri, _stream())
    if rc != 0:
        raise RuntimeError("isrConv3x3Forward%s failed (%d)" % ("Bf16" if bf16 else "F16", rc))
    return y


def conv3x3_f16(x, weight, bias=None, act='none', slope=0.01, residual=None, upsample2x=False):
    """``conv3x3`` in the fp16 fast mode (no autograd): y = act(conv3x3(U(x), fp16(w)) + bias) + residual."""
    if act not in ('none', 'relu', 'leaky'):
        raise ValueError("unknown activation %r" % (act,))
    with torch.no_grad():
        return _launch_lp(x, _prepare_lp(weight), bias.contiguous() if bias is not None else None, residual, weight.shape[0],
                          act, slope, upsample2x, False)


# ---- range guard of the split-operand path, and the per-frame guard words ---------------------------------------------------
# The split of an ACTIVATION into (hi, lo') fp16 numbers overflows at |x| >= 65520 (inf / NaN downstream: loud, not silent).  With
# the reference's networks and G-buffer inputs in [-1, 5] this cannot happen; a user's checkpoint is one badly scaled layer away
# (inference/loadedmodel.py:70-120 of the reference: arbitrary checkpoints, arbitrary sequences).  So every split-operand launch
# leaves the largest |value| it stored in a device word (SplitConvParams::absmax, one atomic per wave) and tensors remember which
# word describes them.  Producers whose output came within a factor two of the limit are HOT: the CONSUMER of a hot tensor (and,
# conservatively, everything downstream of it in that frame) runs on the exact fp32 kernels (sr_conv3x3.hip), which have the full
# fp32 range.  Two ways the host learns about it, neither a synchronisation inside a frame:
#   * a model's FIRST frame: ``range_check_due`` -> ``refresh_range_flags`` reads the words (one synchronisation) and the frame is
#     computed again with the routing -- repeated, because a fused launch only says THAT something inside it was hot and the
#     per-layer pass that replaces it says where: the first frame of any checkpoint is right;
#   * EVERY later frame, one frame late (``guards_publish`` / ``guards_poll``): the frame ends with an asynchronous copy of the
#     words into pinned host memory, the next frame starts by reading that copy -- a plain load.  A layer that turns hot at frame t
#     is routed from frame t + 1 on (a fused segment is then taken apart conservatively: all its layers count as hot).  The
#     threshold (3e4) is a factor two below the overflow, so values that GROW into the limit are caught while they are still exact.
# The last word of the buffer is the dataflow trunk's error word (``isrSetTrunkErrorWord``): a launch that gave up waiting for a
# neighbour raises at the start of the NEXT frame and switches the dataflow form off for the process.
RANGE_GUARD = True
RANGE_LIMIT = 3.0e4
_RANGE_SLOTS = 512
_TRUNK_ERROR_SLOT = _RANGE_SLOTS - 1
_FILL_ERROR_SLOT = _RANGE_SLOTS - 2       # ... and the word before it the one-launch flow fill's (``isrSetFlowFillErrorWord``), same rule
HOT = "hot"                      # range key of a tensor of unknown / large range (e.g. produced by an exact kernel on the guarded path)
_range = {}                      # device -> state, see _range_state


def _range_state(device):
    device = torch.device(device)
    if device.index is None:
        device = torch.device(device.type, torch.cuda.current_device())
    st = _range.get(device)
    if st is None:
        st = {"buf": torch.zeros(_RANGE_SLOTS, dtype=torch.int32, device=device), "slots": {}, "hot": set(), "frames": 0,
              "members": {},                # fused segment key -> keys of the layers it covers
              "mirror": torch.zeros(_RANGE_SLOTS, dtype=torch.int32).pin_memory(), "event": torch.cuda.Event(), "pending": False}
        _range[device] = st
    return st


def _arm_range(key, device, members=None):
    """Arm the NEXT split-operand launch with the flag word of producer ``key`` (no-op when the guard is off).  ``members``: the
    per-layer keys a fused launch stands for.  Out of words: the tensor counts as HOT (its consumer takes the exact kernel) -- an
    unguarded launch is never silently accepted."""
    if not RANGE_GUARD:
        return None
    st = _range_state(device)
    idx = st["slots"].get(key)
    if idx is None:
        if len(st["slots"]) >= _FILL_ERROR_SLOT:
            _warn_once("range_slots", "range guard: all %d flag words are in use (models loaded without ops.range_reset()?); "
                       "further layers' consumers run on the exact fp32 kernels" % _FILL_ERROR_SLOT)
            return HOT
        idx = st["slots"][key] = len(st["slots"])
    if members is not None:
        st["members"][key] = tuple(members)
    _sr().isrSetRangeFlag(ctypes.c_void_p(st["buf"].data_ptr() + 4 * idx))
    return key


def range_is_hot(key, device):
    """Must a consumer of the tensor tagged ``key`` avoid the split kernels?"""
    if key is None or not RANGE_GUARD:
        return False
    return key == HOT or key in _range_state(device)["hot"]


def any_hot(device):
    if not RANGE_GUARD:
        return False
    device = torch.device(device)
    if device.index is None:
        device = torch.device(device.type, torch.cuda.current_device())
    return device in _range and bool(_range[device]["hot"])


def _mark_hot(st, words, conservative):
    """``words``: the guard words as a float32 view (host).  -> the producers that became hot.  ``conservative``: a fused segment
    that is hot makes every layer it covers hot at once (nobody recomputes the frame to find out which one it was)."""
    new = set()
    n = len(st["slots"])
    words = words.numpy() if torch.is_tensor(words) else words
    if n == 0 or bool((words[:n] < RANGE_LIMIT).all()):              # the common case in one vector compare (slots are handed out 0, 1, 2, ...)
        return new
    for key, idx in st["slots"].items():
        v = float(words[idx])
        if not (v < RANGE_LIMIT) and key not in st["hot"]:           # NaN compares false: hot
            st["hot"].add(key)
            new.add(key)
            if conservative:
                for m in st["members"].get(key, ()):
                    if m not in st["hot"]:
                        st["hot"].add(m)
                        new.add(m)
    return new


def refresh_range_flags(device=None):
    """Read the producers' maxima (ONE host synchronisation), mark producers at or above RANGE_LIMIT (or non-finite) hot (the
    words are running maxima and stay: hot is a one-way state until ``range_reset``).  Returns the set of producers that became hot in this call.  (Also a moment the dataflow trunk's error
    word is looked at, ``trunk_check``.)"""
    trunk_check()
    new = set()
    for dev, st in list(_range.items()):
        if device is not None and dev != _range_device(device):
            continue
        if not st["slots"]:
            continue
        vals = st["buf"].cpu().view(torch.float32)
        new |= _mark_hot(st, vals, conservative=False)
        st["pending"] = False
    return new


def _range_device(device):
    device = torch.device(device)
    return device if device.index is not None else torch.device(device.type, torch.cuda.current_device())


def guards_publish(device, record=True):
    """End of a frame: copy the guard words (range maxima + the trunk's error word) into pinned host memory, asynchronously on
    the current stream.  ``record=False`` inside a stream capture (an event recorded there is a graph node, not something the host
    can query): the caller marks the replay's end with ``guards_mark``."""
    st = _range_state(device)      # (with RANGE_GUARD off the range words stay zero; the two error words of the spin kernels are still mirrored)
    st["mirror"].copy_(st["buf"], non_blocking=True)
    if record:
        guards_mark(device)


def guards_mark(device):
    st = _range_state(device)
    st["event"].record(torch.cuda.current_stream())
    st["pending"] = True


def guards_poll(device):
    """Start of a frame: look at what the PREVIOUS frame published -- a plain read of pinned memory, no synchronisation (if that
    copy has not landed yet it is looked at a frame later).  Marks new hot producers (returned) and raises if a dataflow-trunk
    launch timed out."""
    st = _range_state(device)
    if not st["pending"] or not st["event"].query():
        return set()
    return _guards_look(st)


def _guards_look(st):
    st["pending"] = False
    words = st["mirror"]
    err = int(words[_TRUNK_ERROR_SLOT])
    if err:
        _trunk_failed(st, err)
    if int(words[_FILL_ERROR_SLOT]):
        _fill_failed(st)
    if not RANGE_GUARD:
        return set()
    return _mark_hot(st, words.view(torch.float32), conservative=True)


def guards_flush(device):
    """END of a sequence (``SuperResolutionPipeline.reset()`` / ``close()``, the end of an offline render): the frame-late look of
    ``guards_poll`` done NOW for the last frame -- waits for that frame's guard-word copy (one event synchronisation, the frame's
    work is what it waits for) and raises / marks exactly like ``guards_poll``.  Without it the last frame of a sequence is never
    followed by a poll."""
    device = _range_device(device)
    st = _range.get(device)
    if st is None or not st["pending"]:
        return set()
    st["event"].synchronize()
    return _guards_look(st)


def range_reset():
    """Forget every maximum, every hot producer and every producer's word (a new model was loaded)."""
    for st in _range.values():
        st["buf"][:_FILL_ERROR_SLOT].zero_()
        st["hot"].clear()
        st["slots"].clear()
        st["members"].clear()
        st["frames"] = 0
        st["pending"] = False
    _routing_changed()           # guard words are handed out anew: a captured launch would keep writing the old ones


def range_check_due(device):
    """Frame pipelines call this once per frame: True for the FIRST frame since the last reset (the synchronous check that makes a
    checkpoint's first frame right); every later frame is covered by ``guards_publish`` / ``guards_poll``."""
    if not RANGE_GUARD:
        return False
    st = _range_state(device)
    st["frames"] += 1
    return st["frames"] == 1


# ---- split-operand mode: fp32-equivalent accuracy on the fp16 matrix pipe (inference; THE default parity path) ----------
# SPLIT_F16 = True routes the no-grad ``conv3x3`` of layers with more than 8 output channels through
# ``isrConv3x3ForwardSplit`` (csrc/sr_conv_split.hip): every operand is split into two fp16 numbers (22 significand
# bits), a product is three fp16 MFMAs accumulated in fp32 -- 5.3x fewer matrix cycles than the fp32 MFMA at an error
# against fp64 that matches the exact fp32 kernel's (tests/test_conv_gpu.py).  SPLIT_F16 = False restores the exact
# k-ordered fmaf-chain kernels of sr_conv3x3.hip (training always uses those).
SPLIT_F16 = True
_split_cache = {}


def _prepare_split(weight, transpose_flip=False):
    """Weights as scaled (hi, lo) fp16 pairs in the split kernel's layout, cached like ``prepare_weights``;
    transpose_flip: the data-gradient weights w'[ci][co][ky][kx] = w[co][ci][2-ky][2-kx]."""
    lib = _sr()
    key = (id(weight), bool(transpose_flip))
    hit = _split_cache.get(key)
    if hit is not None:
        ref, version, ptr, wq, epoch = hit
        if ref() is weight and version == weight._version and ptr == weight.data_ptr() and epoch == _images_epoch:
            return wq
    w = weight.detach()
    if transpose_flip:
        w = w.flip(2, 3).transpose(0, 1)
    w = w.contiguous()
    cout, cin = w.shape[0], w.shape[1]
    wq = torch.empty(lib.isrConvSplitWeightBytes(cin, cout), dtype=torch.uint8, device=weight.device)
    rc = lib.isrConvSplitPrepare(_ptr(w), _ptr(wq), cout, cin, _stream())
    if rc != 0:
        raise RuntimeError("isrConvSplitPrepare failed (%d)" % rc)
    if len(_split_cache) > 512:
        for k in [k for k, v in _split_cache.items() if v[0]() is None]:
            del _split_cache[k]
    _split_cache[key] = (weakref.ref(weight), weight._version, weight.data_ptr(), wq, _images_epoch)
    return wq


def _split_cached(weight, transpose_flip):
    hit = _split_cache.get((id(weight), bool(transpose_flip)))
    return hit is not None and hit[0]() is weight and hit[1] == weight._version and hit[2] == weight.data_ptr() and hit[4] == _images_epoch


def prepare_split_many(weights):
    """Bring the cached split-kernel images (forward AND data gradient) of all given [Cout, Cin, 3, 3] weights up to date in
    two launches (``isrConvSplitPrepareMany``).  A training step changes every weight, and preparing them one by one costs two
    launches per image plus a flip and a copy for each data-gradient twin -- ~140 small launches per step for EnhanceNet."""
    lib = _sr()
    stale = [w for w in weights if w.is_cuda and w.dtype == torch.float32 and w.dim() == 4 and tuple(w.shape[2:]) == (3, 3)
             and w.is_contiguous() and not (_split_cached(w, False) and _split_cached(w, True))]
    most = lib.isrConvSplitPrepareManyMax()
    for k in range(0, len(stale), most):
        part = stale[k:k + most]
        n = len(part)
        fwd = [torch.empty(lib.isrConvSplitWeightBytes(w.shape[1], w.shape[0]), dtype=torch.uint8, device=w.device) for w in part]
        bwd = [torch.empty(lib.isrConvSplitWeightBytes(w.shape[0], w.shape[1]), dtype=torch.uint8, device=w.device) for w in part]
        pw = (ctypes.c_void_p * n)(*[w.data_ptr() for w in part])
        pf = (ctypes.c_void_p * n)(*[t.data_ptr() for t in fwd])
        pb = (ctypes.c_void_p * n)(*[t.data_ptr() for t in bwd])
        co = (ctypes.c_int * n)(*[w.shape[0] for w in part])
        ci = (ctypes.c_int * n)(*[w.shape[1] for w in part])
        rc = lib.isrConvSplitPrepareMany(n, pw, pf, pb, co, ci, _stream())
        if rc != 0:
            raise RuntimeError("isrConvSplitPrepareMany failed (%d)" % rc)
        for w, f, b in zip(part, fwd, bwd):
            _split_cache[(id(w), False)] = (weakref.ref(w), w._version, w.data_ptr(), f, _images_epoch)
            _split_cache[(id(w), True)] = (weakref.ref(w), w._version, w.data_ptr(), b, _images_epoch)


def _split_fits(x, cout, upsample2x):
    """The split kernels address one image through 32-bit buffer offsets: input and output of a launch must each stay
    below 2 GiB (the caller falls back to the exact fp32 kernels otherwise)."""
    h, w = (2 * x.shape[2], 2 * x.shape[3]) if upsample2x else (x.shape[2], x.shape[3])
    xplane = max(x.stride(1), x.shape[2] * x.shape[3]) if x.stride(3) == 1 else x.shape[2] * x.shape[3]
    return x.shape[1] * xplane * 4 < 2 ** 31 and cout * (h * w + plane_pad(h, w)) * 4 < 2 ** 31


def conv3x3_split(x, weight, bias=None, act='none', slope=0.01, residual=None, upsample2x=False):
    """``conv3x3`` on the split-operand kernel (no autograd): y = act(conv3x3(U(x), w) + bias) + residual."""
    if act not in ('none', 'relu', 'leaky'):
        raise ValueError("unknown activation %r" % (act,))
    with torch.no_grad():
        key = _arm_range(id(weight), x.device)
        y = _launch_split(x, _prepare_split(weight), bias.contiguous() if bias is not None else None, residual, weight.shape[0],
                          act, slope, upsample2x)
        y._isr_range_key = key
        return y


def _launch_split(x, wq, bias, residual, cout, act, slope, upsample2x, packed=False):
    """One launch of isrConv3x3ForwardSplit (x: fp32 NCHW, channel planes may be padded; ``packed`` as in ``_launch_forward``)."""
    lib = _sr()
    x, xp, xi = _plane_strides(x)
    fuse = False
    if upsample2x:
        fuse = bool(lib.isrConvF16SupportsUpsample(x.data_ptr(), x.shape[3], xp, xi))
        if not fuse:          # unaligned low-res rows: the resize runs as its own kernel first
            x, xp, xi = _plane_strides(bilinear_upsample2x(x))
    n, cin = x.shape[0], x.shape[1]
    h, w = (2 * x.shape[2], 2 * x.shape[3]) if fuse else (x.shape[2], x.shape[3])
    rp = ri = 0
    if residual is not None:
        residual, rp, ri = _plane_strides(residual)
    y = torch.empty((n, cout, h, w), dtype=torch.float32, device=x.device) if packed else empty_planes(n, cout, h, w, x.device)
    rc = lib.isrConv3x3ForwardSplit(_ptr(x), _ptr(wq), _ptr(bias), _ptr(residual), _ptr(y), n, cin, h, w, cout,
                                    ACT_CODES[act], float(slope), 1 if fuse else 0, xp, xi, y.stride(1), cout * y.stride(1),
                                    rp, ri, _stream())
    if rc != 0:
        raise RuntimeError("isrConv3x3ForwardSplit failed (%d)" % rc)
    return y


# ---- mixed-precision training mode (opt-in, NOT the parity path) ------------------------------------------------------
# TRAIN_BF16 = True runs the forward and data-gradient convolutions of the autograd functions below with bf16 MFMA
# operands (fp32 accumulation, fp32 tensors, fp32 master weights; the weight gradients and everything else stay
# fp32).  bf16 rather than fp16: gradients span the fp32 exponent range.  Layers with at most 8 input or output
# channels (the 64 -> 6 output layer and its data gradient) keep their fp32 kernels.
TRAIN_BF16 = False
# Forward and data-gradient convolutions of the training graph on the split-operand kernels (csrc/sr_conv_split.hip:
# three fp16 MFMAs per product on (hi, lo) operand pairs, fp32 accumulation -- the fp32 kernels' accuracy against fp64,
# NOT a reduced-precision mode) for layers with at least TRAIN_SPLIT_MIN_TILES tiles of 8x32 pixels; smaller layers and
# all weight gradients stay on the exact fp32 MFMA kernels.
TRAIN_SPLIT = True
# GATE_FUSION: a conv3x3 whose input is the output of a ReLU conv3x3 applies that ReLU's backward in the epilogue of its own
# data gradient (one isrActBackward launch and a 3-tensor round trip less per such pair).  Parameter and input gradients are
# bit-identical either way; the gradient of the INTERMEDIATE activation is then the post-gate value -- a tensor hook or
# retain_grad() on it switches the fusion off for that pair (tests/test_train_kernels_gpu.py), torch.autograd.grad on it does not.
GATE_FUSION = True
TRAIN_SPLIT_MIN_TILES = 256
TRAIN_SPLIT_FEW_INPUTS = os.environ.get("ISR_TRAIN_SPLIT_FEW_INPUTS", "1") != "0"
TRAIN_SPLIT_MIN_TILES2 = 128      # small images: 2-row tiles (the library picks that form below 256 tiles of 8x32 pixels)


def _train_conv(x, weight, transpose_flip, bias, residual, act):
    """Forward (or, with transpose_flip, data-gradient) convolution of the training graph.  Results are PACKED
    tensors: they are saved for backward and their raw pointers go to ``isrActBackward`` and the weight-gradient
    kernels, which index [N, C, H, W] flat (inference outputs may have padded channel planes, these never do)."""
    cout, cin = (weight.shape[1], weight.shape[0]) if transpose_flip else (weight.shape[0], weight.shape[1])
    # the low-precision kernel is a streaming kernel with 8x32-pixel x 64-channel tiles: it pays from a few hundred
    # tiles on (the 128x128 layers of a crop batch); below that the fp32 kernels' finer decompositions are faster
    tiles = x.shape[0] * ((x.shape[2] + 7) // 8) * ((x.shape[3] + 31) // 32)
    flops = 2.0 * 9 * cin * cout * x.shape[0] * x.shape[2] * x.shape[3]
    if TRAIN_BF16 and cout > 8 and cin > 8 and tiles >= 256:
        _tally("bf16", flops)
        return _launch_lp(x, _prepare_lp(weight, transpose_flip, True), bias, residual, cout, act, 0.0, False, True, packed=True)
    # the split-operand kernel (fp32-equivalent accuracy, 2.3x the fp32 MFMA kernel) once a layer has enough 8x32-pixel
    # tiles to fill the persistent grid: the 64^2 / 128^2 post-block layers of a crop batch, 54 % of the step's flops
    # ... or, for a batch of small crops, enough 2-row tiles for the small-image form (conv3x3_split_rows2_kernel)
    tiles2 = x.shape[0] * ((x.shape[2] + 1) // 2) * ((x.shape[3] + 31) // 32)
    # (few INPUT channels are fine -- the 6 -> 64 data gradient of the output layer is one k-step of mostly zero padding, a store-bound
    # launch -- few OUTPUT channels are not: the 32-row MFMA tile would be mostly padding, conv3x3_small_cout_kernel takes those)
    if TRAIN_SPLIT and cout > 8 and (cin > 8 or TRAIN_SPLIT_FEW_INPUTS) and (tiles >= TRAIN_SPLIT_MIN_TILES or tiles2 >= TRAIN_SPLIT_MIN_TILES2) \
            and x.shape[3] % 4 == 0 and _split_fits(x, cout, False):
        _tally("split", flops)
        slot = _gmax_slot(x.device, _GMAX_CONV_WORDS) if transpose_flip else None    # a data gradient: some layer's gz later on
        if slot is not None:
            _sr().isrSetMaxSlots(ctypes.c_void_p(slot), _GMAX_CONV_WORDS)
        y = _launch_split(x, _prepare_split(weight, transpose_flip), bias, residual, cout, act, 0.0, False, packed=True)
        if slot is not None:
            words = _sr().isrTakeMaxSlotWords()
            if words > 0:
                _gmax_tag(y, slot, words)
        return y
    _tally("exact", flops)
    return _launch_forward(x, prepare_weights(weight, transpose_flip=transpose_flip), bias, residual, cin, cout, act, 0.0, False,
                           packed=True)


# Optional tally of the algorithmic flops a step dispatches to each kernel family ("split": three fp16 MFMAs per product,
# ceiling 2500 / 3 TFLOP/s; "exact": fp32 MFMA, ceiling 157.3): bench.py --mode train prices its roofline with it.
FLOP_TALLY = None


def _tally(family, flops):
    if FLOP_TALLY is not None:
        FLOP_TALLY[family] = FLOP_TALLY.get(family, 0.0) + float(flops)


_workspace = {}


def _wgrad_workspace(device, nbytes):
    ws = _workspace.get(device)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _workspace[device] = ws
    return ws


def _weight_grad(xs, gzs, weight, has_bias):
    """(dw, db) summed over the pairs (xs[k], gzs[k]) -- ``isrConv3x3WeightGradSegments``, one pass per <= 32 pairs."""
    lib = _sr()
    cout, cin = weight.shape[0], weight.shape[1]
    n, _, h, w = xs[0].shape
    ws = _wgrad_workspace(weight.device, lib.isrConvWeightGradWorkspace(n, cin, h, w, cout))
    most = lib.isrConvWeightGradMaxSegments()
    gw = gb = None
    for k in range(0, len(xs), most):
        part_x, part_g = xs[k:k + most], gzs[k:k + most]
        dw = torch.empty_like(weight, memory_format=torch.contiguous_format)
        db = torch.empty(cout, dtype=torch.float32, device=weight.device) if has_bias else None
        px = (ctypes.c_void_p * len(part_x))(*[t.data_ptr() for t in part_x])
        pg = (ctypes.c_void_p * len(part_g))(*[t.data_ptr() for t in part_g])
        # mixed-precision mode: bf16 operands once there are enough 4x32-pixel tiles to stream (the kernel is memory bound)
        tiles = n * len(part_x) * ((h + 3) // 4) * ((w + 31) // 32)
        big = w % 4 == 0 and tiles >= 1024 and all(t.data_ptr() % 16 == 0 for t in part_g)
        # (the split kernel also takes the 64 -> 6 output layer: its 32-row MFMA tile is mostly padding there, but at a
        # fifth of the matrix cycles it still beats the fp32 K-split kernel)
        fn = lib.isrConv3x3WeightGradSegmentsBf16 if (TRAIN_BF16 and big and cout > 8) else (
            lib.isrConv3x3WeightGradSegmentsSplit if (TRAIN_SPLIT and big) else lib.isrConv3x3WeightGradSegments)
        _tally("bf16" if (TRAIN_BF16 and big and cout > 8) else ("split" if (TRAIN_SPLIT and big) else "exact"),
               2.0 * 9 * cin * cout * n * len(part_x) * h * w)
        maxima = [_gmax_of(t) for t in part_g] if fn is lib.isrConv3x3WeightGradSegmentsSplit else [None]
        if all(m is not None for m in maxima) and len(set(m[1] for m in maxima)) == 1:
            pm = (ctypes.c_void_p * len(part_g))(*[m[0] for m in maxima])
            rc = lib.isrConv3x3WeightGradSegmentsSplitMax(px, pg, pm, maxima[0][1], len(part_x), _ptr(dw), _ptr(db), _ptr(ws), n, cin, h, w, cout, _stream())
        else:
            rc = fn(px, pg, len(part_x), _ptr(dw), _ptr(db), _ptr(ws), n, cin, h, w, cout, _stream())
        if rc != 0:
            raise RuntimeError("isrConv3x3WeightGradSegments failed (%d)" % rc)
        gw = dw if gw is None else gw + dw
        gb = db if gb is None or db is None else gb + db
    return gw, gb


# Deferred weight gradients.  The frames of a training clip run the SAME layers one after the other, and backward
# through time visits them again in reverse: per layer that is T weight-gradient launches over few pixels each (a
# 32x32 crop batch fills a quarter of the GPU), T slab reductions and T-1 accumulations into .grad.  Inside
# ``deferred_weight_gradients()`` the backward of ``conv3x3`` only records (input, output gradient) and returns no
# weight gradient; leaving the context runs ONE weight-gradient pass per layer over all recorded frames and adds the
# result to ``weight.grad`` / ``bias.grad`` -- the same sums in another order.
_deferred = None
# The split-operand weight gradient scales gz by a power of two taken from max |gz| over the launch's tensors: a pass over every
# gradient tensor.  Inside deferred_weight_gradients() the kernels that PRODUCE those tensors -- the fused small-image block
# (isrResBlockSmall, backward direction: both outputs) and the persistent split-operand convolution (isrSetMaxSlots) -- leave the
# maxima of their outputs in words of a per-step pool instead (one word per wave, no contention), the tensors are tagged with the words' address, and a layer whose gz tensors all carry a valid tag skips the pass
# (isrConv3x3WeightGradSegmentsSplitMax).  The same maxima, the same scale: bit-identical gradients.
GMAX_FROM_PRODUCERS = os.environ.get("ISR_GMAX_FROM_PRODUCERS", "1") != "0"
_GMAX_WORDS = 1 << 20
_GMAX_CONV_WORDS = 8192    # isrSetMaxSlots capacity: 4 words per workgroup (up to four per CU on the one-workgroup-per-tile form)
_gmax = None              # {"pool": int32 tensor, "next": int} of the active deferred_weight_gradients() context


def _gmax_slot(device, words):
    """Address of `words` zeroed words for the maxima of a gradient tensor, or None (outside the context / switched off / pool used up)."""
    global _gmax
    if _deferred is None or not GMAX_FROM_PRODUCERS:
        return None
    if _gmax is None:
        _gmax = {"pool": torch.zeros(_GMAX_WORDS, dtype=torch.int32, device=device), "next": 0}
    if _gmax["pool"].device != device or _gmax["next"] + words > _GMAX_WORDS:
        return None
    _gmax["next"] += words
    return _gmax["pool"].data_ptr() + 4 * (_gmax["next"] - words)


def _gmax_tag(t, slot, words):
    t._isr_gmax = (slot, words, t.data_ptr(), t._version, _gmax["pool"])      # the pool lives as long as a tag points into it


def _gmax_of(t):
    tag = getattr(t, '_isr_gmax', None)
    if tag is None or tag[2] != t.data_ptr() or tag[3] != t._version or _gmax is None or tag[4] is not _gmax["pool"]:
        return None
    return tag[0], tag[1]


class deferred_weight_gradients:
    def __enter__(self):
        global _deferred, _gmax
        if _deferred is not None:
            raise RuntimeError("deferred_weight_gradients() does not nest")
        _deferred = {}
        _gmax = None
        return self

    def __exit__(self, exc_type, exc, tb):
        global _deferred, _gmax
        pending, _deferred = _deferred, None
        try:
            return self._finish(pending, exc_type)
        finally:
            _gmax = None

    @staticmethod
    def _finish(pending, exc_type):
        if exc_type is not None:
            return False
        for (_, shape), (weight, bias, xs, gzs) in pending.items():
            if _weight_grad_into_grad(xs, gzs, weight, bias):
                continue
            gw, gb = _weight_grad(xs, gzs, weight, bias is not None)
            for param, g in ((weight, gw), (bias, gb)):
                if param is None or g is None or not param.requires_grad:
                    continue
                if param.grad is None:
                    param.grad = g.view_as(param)
                else:
                    param.grad.add_(g.view_as(param))
        return False


# The deferred pass writes straight into .grad: a layer that takes the split-operand kernel in one piece (<= 32 frames) has its slab
# reduction ADD to the parameters' gradients (isrSetWeightGradAccumulate) instead of producing dw / db for a `grad += dw` launch per
# parameter (47 launches per step).  Same sums, same roundings (tests/test_train_kernels_gpu.py).  (Tried: the weight-gradient kernels
# of all layers back to back and ONE reduction launch at the end -- 71 launches fewer, the same step time: the slabs, 38 MB per layer,
# then come back from memory instead of the caches.)
WGRAD_INTO_GRAD = os.environ.get("ISR_WGRAD_INTO_GRAD", "1") != "0"


def _weight_grad_into_grad(xs, gzs, weight, bias):
    """The layer's weight (and bias) gradient added to ``weight.grad`` / ``bias.grad`` by the reduction itself; False: not eligible."""
    lib = _sr()
    cout, cin = weight.shape[0], weight.shape[1]
    n, _, h, w = xs[0].shape
    tiles = n * len(xs) * ((h + 3) // 4) * ((w + 31) // 32)
    if not (WGRAD_INTO_GRAD and TRAIN_SPLIT and not TRAIN_BF16 and len(xs) <= lib.isrConvWeightGradMaxSegments()
            and w % 4 == 0 and tiles >= 1024 and all(t.data_ptr() % 16 == 0 for t in gzs)):
        return False
    bits = 0
    for k, param in enumerate((weight, bias)):
        if param is None:
            continue
        if not param.requires_grad or (param.grad is not None and not param.grad.is_contiguous()):
            return False
        if param.grad is None:
            param.grad = torch.empty_like(param, memory_format=torch.contiguous_format)
        else:
            bits |= 1 << k
    ws = _wgrad_workspace(weight.device, lib.isrConvWeightGradWorkspace(n, cin, h, w, cout))
    px = (ctypes.c_void_p * len(xs))(*[t.data_ptr() for t in xs])
    pg = (ctypes.c_void_p * len(gzs))(*[t.data_ptr() for t in gzs])
    maxima = [_gmax_of(t) for t in gzs]
    if all(m is not None for m in maxima) and len(set(m[1] for m in maxima)) == 1:
        pm, words = (ctypes.c_void_p * len(gzs))(*[m[0] for m in maxima]), maxima[0][1]
    else:
        pm, words = None, 0
    _tally("split", 2.0 * 9 * cin * cout * n * len(xs) * h * w)
    lib.isrSetWeightGradAccumulate(bits)
    rc = lib.isrConv3x3WeightGradSegmentsSplitMax(px, pg, pm, words, len(xs), _ptr(weight.grad), _ptr(bias.grad if bias is not None else None),
                                                  _ptr(ws), n, cin, h, w, cout, _stream())
    if rc != 0:
        raise RuntimeError("isrConv3x3WeightGradSegmentsSplitMax failed (%d)" % rc)
    return True


def _has_grad_hooks(p):
    return p is not None and bool(getattr(p, '_backward_hooks', None) or getattr(p, '_post_accumulate_grad_hooks', None))


def _weight_grad_or_defer(weight, bias, has_bias, x, gz):
    """Inside deferred_weight_gradients(): record the pair and return (None, None); else compute (dw, db) now.

    Deferred gradients are written to ``param.grad`` when the context exits, AFTER ``loss.backward()`` and without
    passing through autograd's accumulation: tensor hooks and post-accumulate-grad hooks (DDP-style reducers,
    per-parameter clipping / logging) would never fire and ``torch.autograd.grad(..., weights)`` would see None.  A
    parameter that carries such hooks is therefore NOT deferred (its gradient is computed here, per frame, and flows
    through autograd as usual); ``torch.autograd.grad`` on convolution weights must be called outside the context."""
    if _deferred is not None and weight.is_leaf and (bias is None or bias.is_leaf) \
            and not _has_grad_hooks(weight) and not _has_grad_hooks(bias):
        entry = _deferred.setdefault((id(weight), tuple(x.shape)), (weight, bias, [], []))
        entry[2].append(x)
        entry[3].append(gz)
        return None, None
    return _weight_grad([x], [gz], weight, has_bias)


# A residual block of a batch of SMALL images (the 32 x 32 training crops: too few 8 x 32 tiles for the streaming kernel) as ONE
# launch forward and ONE backward (csrc/sr_conv_block2.h: both convolutions, the intermediate's halo rows recomputed per 2-row
# tile) instead of two each: bit-identical tensors, one chain of launch latencies instead of two.
TRAIN_BLOCK2 = os.environ.get("ISR_TRAIN_BLOCK2", "1") != "0"


def _block2_supported(x, w1, w2):
    n, c, h, w = x.shape
    if not (TRAIN_BLOCK2 and TRAIN_SPLIT and not TRAIN_BF16 and c == 64 and tuple(w1.shape) == (64, 64, 3, 3) and tuple(w2.shape) == (64, 64, 3, 3)):
        return False
    if n * ((h + 7) // 8) * ((w + 31) // 32) >= TRAIN_SPLIT_MIN_TILES or n * ((h + 1) // 2) < TRAIN_SPLIT_MIN_TILES2:
        return False                        # enough 8 x 32 tiles for the streaming kernel / too few rows to fill the GPU
    return bool(_sr().isrResBlockSmallSupported(n, h, w))


def _block2(x, wa, ba, gate, wb, bb, transpose_flip):
    """(z, y): z = relu(conv(x, wa) + ba) (or, with ``gate``: conv(x, wa) where gate > 0, else 0), y = conv(z, wb) + bb + x;
    x, gate: packed [N, 64, H, W]."""
    n, _, h, w = x.shape
    z, y = torch.empty_like(x), torch.empty_like(x)
    _tally("split", 2 * 2.0 * 9 * 64 * 64 * n * h * w)
    words = 4 * n * ((h + 1) // 2)                                     # one per wave of the launch
    zs = _gmax_slot(x.device, words) if transpose_flip else None      # data gradients: both outputs are some layer's gz
    ys = _gmax_slot(x.device, words) if (transpose_flip and zs is not None) else None
    if ys is None:
        zs = None
    rc = _sr().isrResBlockSmall(_ptr(x), _ptr(_prepare_split(wa, transpose_flip)), _ptr(ba), _ptr(gate), _ptr(_prepare_split(wb, transpose_flip)),
                                _ptr(bb), _ptr(z), _ptr(y), n, h, w, ctypes.c_void_p(zs), ctypes.c_void_p(ys), _stream())
    if rc != 0:
        raise RuntimeError("isrResBlockSmall failed (%d)" % rc)
    if zs is not None:
        _gmax_tag(z, zs, words)
        _gmax_tag(y, ys, words)
    return z, y


class _ResidualBlockFunction(torch.autograd.Function):
    """y = x + conv2(relu(conv1(x))) -- EnhanceNet's residual block (enhancenet.py:18-33,141) as ONE autograd node of
    two fused launches forward and two backward: the data gradient of conv2 is gated by relu's output in its epilogue
    (ISR_ACT_GATE) and the data gradient of conv1 adds the skip path's gradient in its epilogue, where separate
    nodes need an activation-backward launch and an accumulation launch in between."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        x = x.contiguous()
        if _block2_supported(x, w1, w2):
            t, y = _block2(x, w1, b1.contiguous() if b1 is not None else None, None, w2, b2.contiguous() if b2 is not None else None, False)
        else:
            t = _train_conv(x, w1, False, b1.contiguous() if b1 is not None else None, None, 'relu')
            y = _train_conv(t, w2, False, b2.contiguous() if b2 is not None else None, x, 'none')
        ctx.params = (w1, b1, w2, b2)
        ctx.save_for_backward(x, t)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, t = ctx.saved_tensors
        w1, b1, w2, b2 = ctx.params
        gy = gy.contiguous()
        gx = None
        if ctx.needs_input_grad[0] and _block2_supported(gy, w1, w2):
            gz1, gx = _block2(gy, w2, None, t, w1, None, True)
        else:
            gz1 = _train_conv(gy, w2, True, None, t, 'gate')
            if ctx.needs_input_grad[0]:
                gx = _train_conv(gz1, w1, True, None, gy, 'none')
        gw1 = gb1 = gw2 = gb2 = None
        if ctx.needs_input_grad[3] or (b2 is not None and ctx.needs_input_grad[4]):
            gw2, gb2 = _weight_grad_or_defer(w2, b2 if (b2 is not None and b2.requires_grad) else None, b2 is not None, t, gy)
        if ctx.needs_input_grad[1] or (b1 is not None and ctx.needs_input_grad[2]):
            gw1, gb1 = _weight_grad_or_defer(w1, b1 if (b1 is not None and b1.requires_grad) else None, b1 is not None, x, gz1)
        return gx, gw1, gb1, gw2, gb2


# ---- the whole low-resolution trunk as ONE dataflow launch (csrc/sr_conv_trunk.hip) -----------------------------------------
# preblock + the residual blocks of a single image whose tiles all fit on the GPU at once (<= #CUs tiles of 16 x 32 pixels, one
# workgroup per CU: the 480 x 270 frame has 255): workgroup w owns tile w through every layer and waits only for its 3 x 3
# neighbourhood's progress.  Bit-identical to the per-layer launches.  Every tile must be RESIDENT: a co-runner that holds a CU (a
# second process on the device, another stream's kernel) can keep a tile out, its neighbours' waits then run into the 50 ms
# deadline, the launch ends with an incomplete output and the error word set.  That word is one of the per-frame guard words:
# ``guards_poll`` at the start of the next frame raises and switches the dataflow form off for the process (``trunk_check()`` is the
# synchronous look: bench.py after its timed region, tests).  Do not use it on a shared device (bench.py: BENCH_SHARE_DEVICE).
TRUNK_DATAFLOW = os.environ.get("ISR_TRUNK_DATAFLOW", "1") != "0"
_trunk_ws = {}


def trunk_supported(x, convs):
    """x [1, Cin, h, w]; convs: [(weight, bias)] = preblock, then conv1 / conv2 of every block."""
    if not (TRUNK_DATAFLOW and not DEVICE_SHARED and SPLIT_F16 and not FAST_F16 and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[0] == 1):
        return False
    if torch.is_grad_enabled() and (x.requires_grad or any(w.requires_grad for w, _ in convs)):
        return False
    if len(convs) < 1 or len(convs) % 2 != 1 or len(convs) > 24 or tuple(convs[0][0].shape[2:]) != (3, 3) or convs[0][0].shape[0] != 64:
        return False
    if any(tuple(w.shape) != (64, 64, 3, 3) for w, _ in convs[1:]) or convs[0][0].shape[1] != x.shape[1]:
        return False
    if any_hot(x.device) or range_is_hot(getattr(x, '_isr_range_key', None), x.device):
        return False
    xs, xp, _ = _plane_strides(x)
    return xs is x and bool(_sr().isrTrunkDataflowSupported(_ptr(x), x.shape[1], x.shape[2], x.shape[3], xp, x.shape[2] * x.shape[3] + plane_pad(x.shape[2], x.shape[3])))


def _trunk_workspace(device, cin, h, w):
    """The dataflow trunk's workspace for this size on the current stream: header (zero unit, the tiles' progress counters, error
    word) + the packed-split input, F and T tensors of the launch; zero-filled once."""
    key = (device, cin, h, w, torch.cuda.current_stream().cuda_stream)
    ws = _trunk_ws.get(key)
    if ws is None:
        ws = torch.zeros(_sr().isrTrunkDataflowWorkspaceBytes(cin, h, w) // 4, dtype=torch.int32, device=device)
        _trunk_ws[key] = ws
    return ws


def trunk_dataflow(x, convs):
    """f = relu(conv(x, w0) + b0); f = f + conv(relu(conv(f, w1) + b1), w2) + b2; ...  in ONE launch (``isrTrunkDataflow``)."""
    lib = _sr()
    prepacked = getattr(x, '_isr_prepacked', None)
    x, xp, _ = _plane_strides(x)
    _, cin, h, w = x.shape
    ws = _trunk_workspace(x.device, cin, h, w)
    if prepacked is not None and prepacked is not ws:
        raise RuntimeError("trunk_dataflow: the input was assembled into another workspace (another stream?) than this launch uses")
    f = empty_planes(1, 64, h, w, x.device)
    wq = [_prepare_split(wt) for wt, _ in convs]
    bs = [b.detach().contiguous() if b is not None else None for _, b in convs]
    n = len(convs)
    pw = (ctypes.c_void_p * n)(*[q.data_ptr() for q in wq])
    pb = (ctypes.c_void_p * n)(*[(b.data_ptr() if b is not None else None) for b in bs])
    st = _range_state(x.device)
    lib.isrSetTrunkErrorWord(ctypes.c_void_p(st["buf"].data_ptr() + 4 * _TRUNK_ERROR_SLOT))
    f._isr_range_key = _arm_range(("trunk", id(convs[0][0])), x.device, members=[id(wt) for wt, _ in convs])
    lib.isrSetTrunkPackedResult(1 if UPS_PHASE else 0)
    if prepacked is not None:
        # assemble_input_packed left the input packed-split in the workspace: no packing pass
        rc = lib.isrTrunkDataflowPrepacked(cin, _ptr(f), f.stride(1), pw, pb, (n - 1) // 2, h, w, _ptr(ws), _stream())
    else:
        rc = lib.isrTrunkDataflow(_ptr(x), cin, xp, _ptr(f), f.stride(1), pw, pb, (n - 1) // 2, h, w, _ptr(ws), _stream())
    if rc != 0:
        raise RuntimeError("isrTrunkDataflow failed (%d)" % rc)
    # the same result PACKED-SPLIT, inside the workspace (valid until the next launch on it): what the phase-decomposed upsampling
    # layer stages by LDS-DMA (conv3x3_ups_phase)
    off, plane = ctypes.c_longlong(), ctypes.c_longlong()
    if UPS_PHASE and lib.isrTrunkDataflowPackedResult(cin, h, w, ctypes.byref(off), ctypes.byref(plane)) == 0:
        ps = PackedSplit(ws[off.value // 4: off.value // 4 + 2 * 8 * plane.value * 4], 64, h, w, plane.value)
        ps.range_key = f._isr_range_key
        f._isr_packed = ps
    return f


def _trunk_failed(st, err):
    global TRUNK_DATAFLOW
    st["buf"][_TRUNK_ERROR_SLOT] = 0           # sticky on the device: an error of ANY launch since the last look was still there
    TRUNK_DATAFLOW = False                      # whoever catches this goes on with the per-layer kernels
    _routing_changed()                          # ... and a captured frame that holds the dataflow launch is stale (pipeline._graph_signature)
    raise RuntimeError("trunk_dataflow_kernel: a tile timed out waiting for its neighbours at layer %d in a launch since the last "
                       "look (that launch's output is incomplete; a workgroup was not resident -- is the device shared?).  The "
                       "dataflow trunk is now off for this process (ops.TRUNK_DATAFLOW)" % (err - 1))


def trunk_check():
    """Host read of the dataflow launches' error word (ONE synchronisation): raises if a tile ever gave up waiting for a
    neighbour -- which would mean a workgroup was not resident (more tiles than the GPU holds) or the launch was disturbed."""
    for st in list(_range.values()):
        err = int(st["buf"][_TRUNK_ERROR_SLOT].item())
        if err:
            _trunk_failed(st, err)


# Inference: the whole block in ONE launch (csrc/sr_conv_block.hip) -- bit-identical to the two split-operand launches
BLOCK_FUSION = os.environ.get("ISR_BLOCK_FUSION", "0") == "1"
BLOCK_PACKED = os.environ.get("ISR_BLOCK_PACKED", "1") != "0"       # two launches per block, the intermediate packed-split (see residual_block)
BLOCK_PACKED_MIN_TILES = 256
BLOCK_FUSION_MIN_TILES = 256       # below that the persistent grid is not filled; the per-layer kernels' small-image forms take over
_block_ws = {}


def _block_supported(x, w1, w2):
    if not (BLOCK_FUSION and SPLIT_F16 and not FAST_F16 and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[0] == 1):
        return False
    if any_hot(x.device):
        return False
    if tuple(w1.shape) != (64, 64, 3, 3) or tuple(w2.shape) != (64, 64, 3, 3) or x.shape[1] != 64:
        return False
    if ((x.shape[2] + 7) // 8) * ((x.shape[3] + 31) // 32) < BLOCK_FUSION_MIN_TILES:
        return False
    xs, xp, _ = _plane_strides(x)
    return xs is x and bool(_sr().isrResBlockSplitSupported(_ptr(x), x.shape[2], x.shape[3], xp, x.shape[2] * x.shape[3] + plane_pad(x.shape[2], x.shape[3])))


def residual_block_fused(x, w1, b1, w2, b2):
    """x + conv3x3(relu(conv3x3(x, w1, b1)), w2, b2) for one [1, 64, H, W] image in one launch of ``isrResBlockSplit``."""
    lib = _sr()
    x, xp, _ = _plane_strides(x)
    _, _, h, w = x.shape
    key = (x.device, torch.cuda.current_stream().cuda_stream)
    ws = _block_ws.get(key)
    if ws is None:
        ws = torch.empty(lib.isrResBlockSplitWorkspaceBytes(), dtype=torch.uint8, device=x.device)
        _block_ws[key] = ws
    y = empty_planes(1, 64, h, w, x.device)
    y._isr_range_key = _arm_range(("block", id(w1)), x.device, members=[id(w1), id(w2)])      # covers the intermediate and the output
    rc = lib.isrResBlockSplit(_ptr(x), _ptr(_prepare_split(w1)), _ptr(b1.detach().contiguous() if b1 is not None else None),
                              _ptr(_prepare_split(w2)), _ptr(b2.detach().contiguous() if b2 is not None else None), _ptr(y), _ptr(ws),
                              h, w, xp, y.stride(1), _stream())
    if rc != 0:
        raise RuntimeError("isrResBlockSplit failed (%d)" % rc)
    return y


def residual_block(x, w1, b1, w2, b2):
    """x + conv3x3(relu(conv3x3(x, w1, b1)), w2, b2), 64 -> 64 -> 64 channels."""
    needs_grad = torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (x, w1, b1, w2, b2))
    if x.is_cuda and needs_grad and x.dtype == torch.float32 and w1.shape[0] == w1.shape[1] == w2.shape[0] == w2.shape[1]:
        return _ResidualBlockFunction.apply(x, w1, b1, w2, b2)
    if not needs_grad and _block_supported(x, w1, w2):
        return residual_block_fused(x, w1, b1, w2, b2)
    if not needs_grad and BLOCK_PACKED and SPLIT_F16 and not FAST_F16 and x.is_cuda and x.dtype == torch.float32 and x.shape[0] == 1 \
            and tuple(w1.shape) == (64, 64, 3, 3) == tuple(w2.shape) and x.shape[3] % 4 == 0 \
            and ((x.shape[2] + 7) // 8) * ((x.shape[3] + 31) // 32) >= BLOCK_PACKED_MIN_TILES and packed_supported(x, w1, False):
        # the intermediate of the block travels packed-split: conv1 stores (hi, lo') units straight from its accumulators (no LDS
        # transposition), conv2 stages them by LDS-DMA (no conversion) -- the same numbers, bit-identical output
        t = conv3x3_split_packed(x, w1, b1, act='relu')
        return conv3x3_split_from_packed(t, w2, b2, residual=x)
    return conv3x3(conv3x3(x, w1, b1, act='relu'), w2, b2, residual=x)


class _Conv3x3Function(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, residual, act, slope):
        # x is the output of a ReLU convolution (tagged by conv3x3 below): the data gradient of THIS layer can then apply
        # that ReLU's backward in its epilogue (ISR_ACT_GATE on x > 0) and the producer skips its isrActBackward launch
        ctx.x_relu = bool(getattr(x, '_isr_relu_out', False)) and x.is_contiguous()
        x = x.contiguous()
        res = residual.contiguous() if residual is not None else None
        b = bias.contiguous() if bias is not None else None
        cout, cin = weight.shape[0], weight.shape[1]
        if cout <= 8 and cin * x.shape[2] * x.shape[3] * 4 < 2 ** 31:
            _tally("exact", 2.0 * 9 * cin * cout * x.shape[0] * x.shape[2] * x.shape[3])
            y = _launch_small(x, weight, b, res, act, slope)      # the 64 -> 6 output layer: 4x4x1 MFMA blocks
        elif act == 'leaky':
            _tally("exact", 2.0 * 9 * cin * cout * x.shape[0] * x.shape[2] * x.shape[3])
            y = _launch_forward(x, prepare_weights(weight), b, res, cin, cout, act, slope, False, packed=True)
        else:
            y = _train_conv(x, weight, False, b, res, act)
        ctx.act, ctx.slope = act, slope
        ctx.has_bias, ctx.has_res = bias is not None, residual is not None
        ctx.bias = bias if (bias is not None and bias.requires_grad) else None
        ctx.weight = weight
        if act != 'none' and residual is not None:
            raise RuntimeError("conv3x3: activation together with a fused residual is inference-only")
        ctx.save_for_backward(x, weight, y if act != 'none' else None)
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = _sr()
        x, weight, y = ctx.saved_tensors
        gy = gy.contiguous()
        cout, cin = weight.shape[0], weight.shape[1]
        if ctx.act == 'relu' and getattr(gy, '_isr_gated_by', None) == (y.data_ptr(), y._version, gy._version):
            gz = gy              # the consumer's data gradient already multiplied by (y > 0), see below
        elif ctx.act != 'none':
            gz = torch.empty_like(gy)
            rc = lib.isrActBackward(_ptr(gy), _ptr(y), _ptr(gz), gy.numel(), ACT_CODES[ctx.act], float(ctx.slope), _stream())
            if rc != 0:
                raise RuntimeError("isrActBackward failed (%d)" % rc)
        else:
            gz = gy
        gx = gw = gb = gres = None
        if ctx.needs_input_grad[0]:
            # data gradient = the same fused kernel on flipped / transposed weights
            # (a hook or retain_grad() on x wants dL/dx itself, not dL/dx already multiplied by (x > 0): no fusion then.  What
            # cannot be seen from here -- torch.autograd.grad(loss, x) on the intermediate activation -- gets the gated value)
            if ctx.x_relu and GATE_FUSION and not _has_grad_hooks(x) and not x.retains_grad:
                gx = _train_conv(gz, weight, True, None, x, 'gate')
                gx._isr_gated_by = (x.data_ptr(), x._version, gx._version)   # read by the backward of the layer that produced x
            else:
                gx = _train_conv(gz, weight, True, None, None, 'none')
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            gw, gb = _weight_grad_or_defer(ctx.weight, ctx.bias, ctx.has_bias, x, gz)
        if ctx.has_res and ctx.needs_input_grad[3]:
            gres = gy
        return gx, gw, gb, gres, None, None


def _act_cpu(z, act, slope):
    if act == 'relu':
        return F.relu(z)
    if act == 'leaky':
        return F.leaky_relu(z, slope)
    return z


def conv3x3(x, weight, bias=None, act='none', slope=0.01, residual=None, upsample2x=False):
    """See module docstring.  x [N,Cin,h,w], weight [Cout,Cin,3,3] -> [N,Cout,H,W]."""
    if act not in ('none', 'relu', 'leaky'):
        raise ValueError("unknown activation %r" % (act,))
    if not x.is_cuda:
        if upsample2x:
            x = F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False)
        y = _act_cpu(F.conv2d(x, weight, bias, padding=1), act, slope)
        return y + residual if residual is not None else y
    if x.dtype != torch.float32 or weight.dtype != torch.float32:
        raise TypeError("conv3x3 (HIP) computes in fp32")
    needs_grad = torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad or
                                              (bias is not None and bias.requires_grad) or
                                              (residual is not None and residual.requires_grad))
    if not needs_grad:
        cout, cin = weight.shape[0], weight.shape[1]
        if FAST_F16 and cout > 8:
            return conv3x3_f16(x, weight, bias, act, slope, residual, upsample2x)
        hot_in = range_is_hot(getattr(x, '_isr_range_key', None), x.device) or \
            (residual is not None and getattr(residual, '_isr_range_key', None) == HOT)
        if SPLIT_F16 and cout > 8 and _split_fits(x, cout, upsample2x) and not hot_in:
            return conv3x3_split(x, weight, bias, act, slope, residual, upsample2x)
        if hot_in and cout > 8:
            # the input (or something upstream of it) came close to the fp16 range of the split: exact fp32 kernel, and whatever
            # consumes ITS output stays exact too (its range is not tracked)
            y = _launch_forward(x, prepare_weights(weight), bias.contiguous() if bias is not None else None, residual,
                                cin, cout, act, slope, upsample2x)
            y._isr_range_key = HOT
            return y
        if cout <= 8 and not upsample2x and cin * x.shape[2] * x.shape[3] * 4 < 2 ** 31:
            return _launch_small(x, weight, bias, residual.contiguous() if residual is not None else None, act, slope)
        if cout <= 8 and not upsample2x:
            _warn_once("small_cout_2g", "conv3x3: %d x %d x %d input exceeds the small-Cout kernel's 2 GiB addressing; the layer "
                       "runs on the general fp32 MFMA kernel (output channels padded to 32)" % (cin, x.shape[2], x.shape[3]))
        return _launch_forward(x, prepare_weights(weight),
                               bias.contiguous() if bias is not None else None,
                               residual,
                               cin, cout, act, slope, upsample2x)
    if upsample2x:   # training: the resize stays its own differentiable op (isrUpsample2xForward / Backward)
        x = bilinear_upsample2x(x)
    if residual is not None and act != 'none':
        return _Conv3x3Function.apply(x, weight, bias, None, act, slope) + residual
    y = _Conv3x3Function.apply(x, weight, bias, residual, act, slope)
    if act == 'relu' and residual is None:
        y._isr_relu_out = True      # a consumer conv3x3 may fold this ReLU's backward into its data gradient
    return y


# ---- training-side elementwise kernels (csrc/sr_train.hip) ---------------------------------------
class _Upsample2xFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        n, c, h, w = x.shape
        y = torch.empty((n, c, 2 * h, 2 * w), dtype=torch.float32, device=x.device)
        rc = _sr().isrUpsample2xForward(_ptr(x), _ptr(y), n * c, h, w, _stream())
        if rc != 0:
            raise RuntimeError("isrUpsample2xForward failed (%d)" % rc)
        return y

    @staticmethod
    def backward(ctx, gy):
        gy = gy.contiguous()
        n, c, H, W = gy.shape
        gx = torch.empty((n, c, H // 2, W // 2), dtype=torch.float32, device=gy.device)
        rc = _sr().isrUpsample2xBackward(_ptr(gy), _ptr(gx), n * c, H // 2, W // 2, _stream())
        if rc != 0:
            raise RuntimeError("isrUpsample2xBackward failed (%d)" % rc)
        return gx


def bilinear_upsample2x(x):
    """``F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False)`` with a hand-written forward and
    (gather, deterministic) backward on the GPU; PyTorch's launch takes ~340 us for a 64-channel 64^2 -> 128^2 batch
    of 16 that moves 84 MB."""
    if not x.is_cuda:
        return F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False)
    if x.dtype != torch.float32:
        raise TypeError("bilinear_upsample2x (HIP) computes in fp32")
    return _Upsample2xFunction.apply(x)


class _ReconResidualFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, outputs, inputs, k):
        outputs, inputs = outputs.contiguous(), inputs.contiguous()
        n, cout, H, W = outputs.shape
        cin, h, w = inputs.shape[1], inputs.shape[2], inputs.shape[3]
        out = torch.empty_like(outputs)
        rc = _sr().isrReconResidualForward(_ptr(outputs), _ptr(inputs), _ptr(out), n, cout, cin, k, h, w, _stream())
        if rc != 0:
            raise RuntimeError("isrReconResidualForward failed (%d)" % rc)
        ctx.dims = (n, cout, cin, k, h, w)
        return out

    @staticmethod
    def backward(ctx, g):
        n, cout, cin, k, h, w = ctx.dims
        g = g.contiguous()
        gin = None
        if ctx.needs_input_grad[1]:
            gin = torch.empty((n, cin, h, w), dtype=torch.float32, device=g.device)
            rc = _sr().isrReconResidualBackward(_ptr(g), _ptr(gin), n, cout, cin, k, h, w, _stream())
            if rc != 0:
                raise RuntimeError("isrReconResidualBackward failed (%d)" % rc)
        return g, gin, None


def recon_residual_supported(outputs, inputs, k):
    return (outputs.is_cuda and outputs.dtype == torch.float32 and inputs.dtype == torch.float32 and outputs.dim() == 4
            and outputs.shape[0] == inputs.shape[0] and outputs.shape[2] == 4 * inputs.shape[2] and outputs.shape[3] == 4 * inputs.shape[3]
            and 0 < k <= min(outputs.shape[1], inputs.shape[1]) and outputs.shape[0] <= 65535 and outputs.shape[2] <= 65535)


def recon_residual(outputs, inputs, k):
    """``outputs`` with the bilinearly x4-resized first ``k`` channels of ``inputs`` added to its first ``k`` channels
    (EnhanceNet's residual reconstruction) in one launch; differentiable w.r.t. both."""
    return _ReconResidualFunction.apply(outputs, inputs, k)


def _batch_view(t, c, h, w):
    """(tensor, batch stride in floats) of a [B, c, h, w] view whose images are packed (a frame of a [B, T, ..] clip)."""
    if t.stride(3) == 1 and t.stride(2) == w and t.stride(1) == h * w and (t.shape[0] == 1 or t.stride(0) >= c * h * w):
        return t, (t.stride(0) if t.shape[0] > 1 else c * h * w)
    t = t.contiguous()
    return t, c * h * w


class _RecurrentInputFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, prev_raw, input_low, flow_low):
        prev_raw = prev_raw.contiguous()
        b, _, h, w = input_low.shape
        inp, istride = _batch_view(input_low, 5, h, w)
        flo, fstride = _batch_view(flow_low, 2, h, w)
        netin = torch.empty((b, 101, h, w), dtype=torch.float32, device=prev_raw.device)
        warped = torch.empty((b, 6, 4 * h, 4 * w), dtype=torch.float32, device=prev_raw.device)
        rc = _sr().isrRecurrentInputForward(_ptr(prev_raw), _ptr(inp), _ptr(flo), _ptr(netin), _ptr(warped), b, h, w,
                                            istride, fstride, _stream())
        if rc != 0:
            raise RuntimeError("isrRecurrentInputForward failed (%d)" % rc)
        ctx.save_for_backward(prev_raw, flo)
        ctx.fstride = fstride
        return netin, warped

    @staticmethod
    def backward(ctx, g_netin, g_warped):
        prev_raw, flo = ctx.saved_tensors
        b, _, H, W = prev_raw.shape
        g_netin = g_netin.contiguous() if g_netin is not None else None
        g_warped = g_warped.contiguous() if g_warped is not None else None
        scratch = torch.empty_like(prev_raw)
        g_raw = torch.empty_like(prev_raw)
        rc = _sr().isrRecurrentInputBackward(_ptr(prev_raw), _ptr(flo), _ptr(g_netin), _ptr(g_warped), _ptr(scratch), _ptr(g_raw),
                                             b, H // 4, W // 4, ctx.fstride, _stream())
        if rc != 0:
            raise RuntimeError("isrRecurrentInputBackward failed (%d)" % rc)
        return g_raw, None, None


def recurrent_input(prev_raw, input_low, flow_low):
    """Network input of frame t > 0 of a training clip and the warped previous frame the temp-l2 loss compares against:
    (net_input [B,101,h,w], warped [B,6,4h,4w]) from the RAW prediction of frame t-1 [B,6,4h,4w], the frame's low-res
    input [B,5,h,w] and the flow [B,2,h,w]; differentiable w.r.t. prev_raw.  Replaces clamp / normalize / cat,
    ``VideoTools.warp_upscale(.., special_mask=True)``, ``VideoTools.flatten_high`` and the final cat (~25 launches
    forward, ~35 backward) by one launch forward and three backward."""
    return _RecurrentInputFunction.apply(prev_raw, input_low, flow_low)


def recurrent_input_supported(prev_raw, input_low, flow_low, upscale):
    return (upscale == 4 and prev_raw.is_cuda and prev_raw.dtype == torch.float32 and input_low.dtype == torch.float32
            and prev_raw.shape[1] == 6 and input_low.shape[1] == 5 and not flow_low.requires_grad and not input_low.requires_grad
            and prev_raw.shape[2] == 4 * input_low.shape[2] and prev_raw.shape[3] == 4 * input_low.shape[3])


LOSS_KINDS = ('mse', 'l1', 'temp-l2')
LOSS_TARGETS = ('mask', 'normal', 'ao', 'depth', 'color')
_loss_ws = {}


class _LossUnshadedFunction(torch.autograd.Function):
    """values[16] = the 15 term means + the weighted total, see ``isrLossUnshadedForward``."""

    @staticmethod
    def forward(ctx, gt, pred, prev, cfg):
        lib = _sr()
        gt, pred = gt.contiguous(), pred.contiguous()
        prev = prev.contiguous() if prev is not None else None
        n, _, h, w = pred.shape
        key = (pred.device, torch.cuda.current_stream().cuda_stream)
        ws = _loss_ws.get(key)
        if ws is None:
            ws = torch.empty(lib.isrLossUnshadedWorkspace(), dtype=torch.uint8, device=pred.device)
            _loss_ws[key] = ws
        values = torch.empty(16, dtype=torch.float32, device=pred.device)
        rc = lib.isrLossUnshadedForward(_ptr(gt), _ptr(pred), _ptr(prev), n, h, w, cfg['pad'], cfg['weights'], cfg['enabled'],
                                        cfg['shading'], cfg['ao'], cfg['inverse_ao'], _ptr(ws), _ptr(values), _stream())
        if rc != 0:
            raise RuntimeError("isrLossUnshadedForward failed (%d)" % rc)
        ctx.cfg = cfg
        ctx.save_for_backward(gt, pred, prev)
        return values

    @staticmethod
    def backward(ctx, gvalues):
        lib = _sr()
        gt, pred, prev = ctx.saved_tensors
        cfg = ctx.cfg
        n, _, h, w = pred.shape
        gvalues = gvalues.contiguous()
        gpred = torch.empty_like(pred)
        gprev = torch.empty_like(prev) if prev is not None and ctx.needs_input_grad[2] else None
        rc = lib.isrLossUnshadedBackward(_ptr(gt), _ptr(pred), _ptr(prev), n, h, w, cfg['pad'], cfg['weights'], cfg['enabled'],
                                         cfg['shading'], cfg['ao'], cfg['inverse_ao'], _ptr(gvalues), _ptr(gpred), _ptr(gprev),
                                         _stream())
        if rc != 0:
            raise RuntimeError("isrLossUnshadedBackward failed (%d)" % rc)
        return None, gpred, gprev, None


def loss_unshaded_config(weight_dict, padding, shading):
    """Host-side constants of the fused loss: weight_dict {(kind, target): weight}, shading = the loss module's
    ScreenSpaceShading (specular must be off)."""
    weights = [0.0] * 15
    enabled = 0
    for (kind, target), wgt in weight_dict.items():
        kind = 'mse' if kind in ('l2', 'l2_loss') else kind
        t = LOSS_KINDS.index(kind) * 5 + LOSS_TARGETS.index(target)
        weights[t] = float(wgt)
        enabled |= 1 << t
    v = shading.packed_parameters()          # ambient, diffuse, specular, light, material, background
    amb, diff, light, mat, bg = v[0:3], v[3:6], v[9:12], v[12:15], v[15:18]
    sh = [a * m for a, m in zip(amb, mat)] + [d * m for d, m in zip(diff, mat)] + list(light) + list(bg)
    return {'pad': int(padding), 'weights': (ctypes.c_float * 15)(*weights), 'enabled': enabled,
            'shading': (ctypes.c_float * 12)(*sh), 'ao': float(shading._ao), 'inverse_ao': int(bool(shading.inverse_ao))}


def loss_unshaded(gt, pred, prev, cfg):
    """-> values[16] (differentiable w.r.t. pred and prev through values[15], the weighted total)."""
    return _LossUnshadedFunction.apply(gt, pred, prev, cfg)


@contextlib.contextmanager
def graph_capture(graph, **kw):
    """``torch.cuda.graph(graph, **kw)`` with the CYCLIC garbage collector out of the way.

    A global-mode stream capture makes most HIP calls illegal on every thread until it ends.  If a cyclic collection runs
    INSIDE the capture and finalises GPU objects that an earlier caller dropped in a reference cycle (a HIP graph, a
    stream or event of a SuperResolutionPipeline, a cached workspace), their destructors call hipFree / hipEventDestroy /
    hipGraphDestroy from C++ ``noexcept`` code while the capture is open: the runtime reports an error, the destructor
    throws, ``std::terminate`` -- "Fatal Python error: Aborted" with the main thread "Garbage-collecting" (one full GPU
    test run of round 3 died exactly there, in ``GraphedTrainStep.__init__``).  torch 2.10's ``graph.__enter__`` does not
    collect by itself any more.  So: drain the device, collect NOW (destructors run at a quiescent point), and keep the
    collector off until the capture has ended.

    What this does NOT cover (the caller's contract): it removes the cyclic collector, the path that produced the abort.  An
    object whose reference count drops to zero INSIDE the capture body (a stream, event, graph or tensor the body itself lets
    go of) is still finalised there and then, and a collection or a free triggered from ANOTHER thread during the capture is
    not held back either.  Capture bodies in this package (``train.GraphedTrainStep``, ``SuperResolutionPipeline._frame_graph``)
    therefore allocate their streams / events / workspaces before the capture (the eager warm-up pass) and hold them in
    attributes for the graph's lifetime; multi-threaded hosts must not free GPU objects while a capture is open."""
    import gc
    torch.cuda.synchronize()
    gc.collect()
    torch.cuda.synchronize()
    was_enabled = gc.isenabled()
    gc.disable()
    try:
        with torch.cuda.graph(graph, **kw):
            yield graph
    finally:
        if was_enabled:
            gc.enable()


def adam_flat_step(params, grads, exp_avg, exp_avg_sq, step, lr, beta1, beta2, eps):
    """One Adam step over flat fp32 buffers (``isrAdamFlatStep``); ``lr``: float or one-element device tensor; ``step``: one-element
    float32 device tensor counting the steps taken (incremented).  CPU tensors: the same arithmetic with PyTorch ops."""
    if not params.is_cuda:
        t = float(step.item()) + 1.0
        lr_v = float(lr.item()) if torch.is_tensor(lr) else float(lr)
        exp_avg.lerp_(grads, 1.0 - beta1)
        exp_avg_sq.mul_(beta2).addcmul_(grads, grads, value=1.0 - beta2)
        denom = (exp_avg_sq.sqrt() / (1.0 - beta2 ** t) ** 0.5).add_(eps)
        params.addcdiv_(exp_avg, denom, value=-lr_v / (1.0 - beta1 ** t))
        step.add_(1.0)
        return
    lr_dev = lr if torch.is_tensor(lr) else None
    rc = _sr().isrAdamFlatStep(_ptr(params), _ptr(grads), _ptr(exp_avg), _ptr(exp_avg_sq), params.numel(), _ptr(lr_dev),
                               0.0 if lr_dev is not None else float(lr), float(beta1), float(beta2), float(eps), _ptr(step), _stream())
    if rc != 0:
        raise RuntimeError("isrAdamFlatStep failed (%d)" % rc)


# ---- fused frame assembly (inference) ---------------------------------------------------------
INIT_MODES = {"zero": 0, "unshaded": 1, "input": 2}


def assemble_input(gbuffer_hwc, flow_filled, prev_high, initial_image="zero", ao_inverted=False, out=None, rows=None, cols=None):
    """Renderer G-buffer [h,w,12] (+ hole-filled flow [1,2,h,w], previous frame [1,6,4h,4w] or None)
    -> network input [1,101,h,w] in one launch (``isrAssembleInput``).  ``rows`` = (row0, row1): only those rows are written
    (``isrAssembleInputRows``; the rest of the result is uninitialised unless ``out`` was given)."""
    assert gbuffer_hwc.is_cuda and gbuffer_hwc.is_contiguous() and gbuffer_hwc.shape[-1] == 12
    h, w = gbuffer_hwc.shape[0], gbuffer_hwc.shape[1]
    if out is None:
        out = torch.empty((1, 101, h, w), dtype=torch.float32, device=gbuffer_hwc.device)
    elif getattr(out, '_isr_prepacked', None) is not None:
        del out._isr_prepacked                      # a tensor assemble_input_packed handed out before: its planes are written now
    if prev_high is not None:
        prev_high = prev_high.contiguous()
        flow_filled = flow_filled.contiguous()
        assert prev_high.shape == (1, 6, 4 * h, 4 * w) and flow_filled.shape == (1, 2, h, w)
    r0, r1 = (0, h) if rows is None else (int(rows[0]), int(rows[1]))
    c0, c1 = (0, w) if cols is None else (int(cols[0]), int(cols[1]))          # ``cols``: likewise columns (a screen tile + halo)
    rc = _sr().isrAssembleInputRect(_ptr(gbuffer_hwc), _ptr(flow_filled) if prev_high is not None else None,
                                    _ptr(prev_high), _ptr(out), h, w, INIT_MODES[initial_image], 1 if ao_inverted else 0, r0, r1, c0, c1, _stream())
    if rc != 0:
        raise RuntimeError("isrAssembleInput failed (%d)" % rc)
    return out


# the frame's input assembly writes the dataflow trunk's packed-split input itself (isrAssembleInputPacked): one pass over memory and
# one kernel boundary less per frame than assembling fp32 planes and packing them
ASSEMBLE_PACKED = os.environ.get("ISR_ASSEMBLE_PACKED", "1") != "0"


def assemble_input_packed(gbuffer_hwc, flow_filled, prev_high, convs, initial_image="zero", ao_inverted=False):
    """``assemble_input`` for a frame whose trunk runs as the dataflow launch (``convs``: what ``trunk_supported`` takes): the network
    input goes PACKED-SPLIT straight into that launch's workspace.  Returns x [1,101,h,w] of which only channels 0 .. 4 are written
    (what the frame's finishing reads) -- ``x._isr_prepacked`` holds the workspace, ``trunk_dataflow`` then skips its packing pass and
    ``forward_features`` refuses any other route -- or None when the dataflow trunk would not take this frame (assemble the plain way).
    Same values through the same split as the packing pass: the frame is bit-identical."""
    assert gbuffer_hwc.is_cuda and gbuffer_hwc.is_contiguous() and gbuffer_hwc.shape[-1] == 12
    h, w = gbuffer_hwc.shape[0], gbuffer_hwc.shape[1]
    if not ASSEMBLE_PACKED:
        return None
    out = torch.empty((1, 101, h, w), dtype=torch.float32, device=gbuffer_hwc.device)
    if not trunk_supported(out, convs):
        return None
    if prev_high is not None:
        prev_high = prev_high.contiguous()
        flow_filled = flow_filled.contiguous()
        assert prev_high.shape == (1, 6, 4 * h, 4 * w) and flow_filled.shape == (1, 2, h, w)
    ws = _trunk_workspace(out.device, 101, h, w)
    rc = _sr().isrAssembleInputPacked(_ptr(gbuffer_hwc), _ptr(flow_filled) if prev_high is not None else None, _ptr(prev_high), _ptr(out),
                                      h, w, INIT_MODES[initial_image], 1 if ao_inverted else 0, _ptr(ws), _stream())
    if rc == -3:
        return None
    if rc != 0:
        raise RuntimeError("isrAssembleInputPacked failed (%d)" % rc)
    out._isr_prepacked = ws
    return out


_fill_ws = {}
_FILL_WS_MAX = 16                               # (device, size, stream) workspaces kept; a frame pipeline uses one or two


FLOW_FILL_ONE = os.environ.get("ISR_FLOW_FILL_ONE", "1") != "0"       # the push-pull pyramid as ONE launch (isrFlowFillOne) where it applies


def _fill_failed(st):
    global FLOW_FILL_ONE
    st["buf"][_FILL_ERROR_SLOT] = 0
    FLOW_FILL_ONE = False                       # whoever catches this goes on with the three-launch form
    _routing_changed()                          # a captured frame that holds the one-launch fill is stale (pipeline._graph_signature)
    # A launch that gave up leaves its workspace's ticket / flag words out of step.  The workspaces are NOT freed (a captured graph
    # or a launch still in flight on the render stream may point into them): they are zero-filled again at a quiescent point.
    torch.cuda.synchronize()
    for ws in _fill_ws.values():
        ws.zero_()
    torch.cuda.synchronize()
    raise RuntimeError("flow_fill_one_kernel: a workgroup waited 50 ms for the top of the pyramid in a launch since the last look "
                       "(that frame's filled flow is incomplete; is the device shared?).  The three-launch form is used from now on.")


def fill_flow_gbuffer(gbuffer_hwc, out=None, stream=None, threads=1024, one_launch=None):
    """Hole-filled flow [1,2,h,w] straight from the renderer's G-buffer [h,w,12] (``isrFlowFillOne``: one launch, a workgroup
    per 64 x 64 tile; ``isrFlowFillEx``: three launches, for images of more than 256 tiles or ``one_launch=False`` /
    ISR_FLOW_FILL_ONE=0 -- bit-identical).  ``out`` / ``stream``: caller-owned result tensor and HIP stream (the frame pipeline
    fills the flow of the next frame on its render stream); the pyramid workspace is per (device, size, stream);
    ``threads``: workgroup size of the three-launch form's two grid-wide passes."""
    lib = _sr()
    h, w = gbuffer_hwc.shape[0], gbuffer_hwc.shape[1]
    st = torch.cuda.current_stream() if stream is None else stream
    key = (gbuffer_hwc.device, h, w, st.cuda_stream)
    ws = _fill_ws.pop(key, None)
    if ws is None:
        if len(_fill_ws) >= _FILL_WS_MAX:
            # bounded: the key holds a raw stream handle (streams come and go in a long-lived viewer).  The least recently used workspace is
            # dropped at a quiescent point -- a launch in flight or a captured frame may still point into it, so: synchronise, and tell the
            # frame pipelines that what they captured is stale
            torch.cuda.synchronize()
            _fill_ws.pop(next(iter(_fill_ws)))
            _routing_changed()
        with torch.cuda.stream(st):             # zero-filled once, in stream order with its first use: the one-launch form's tickets live in it
            ws = torch.zeros(lib.isrFlowFillWorkspace(h, w), dtype=torch.uint8, device=gbuffer_hwc.device)
    _fill_ws[key] = ws                          # (re-inserted: most recently used last)
    if out is None:
        out = torch.empty((1, 2, h, w), dtype=torch.float32, device=gbuffer_hwc.device)
    one = (FLOW_FILL_ONE and not DEVICE_SHARED) if one_launch is None else one_launch
    if one and lib.isrFlowFillOneSupported(h, w):
        # the error word is one of the per-frame guard words whether or not the RANGE guard is on (a spin kernel's timeout must
        # never go to a word that nobody reads)
        lib.isrSetFlowFillErrorWord(ctypes.c_void_p(_range_state(gbuffer_hwc.device)["buf"].data_ptr() + 4 * _FILL_ERROR_SLOT))
        rc = lib.isrFlowFillOne(_ptr(gbuffer_hwc), _ptr(out), _ptr(ws), h, w, ctypes.c_void_p(st.cuda_stream))
    else:
        rc = lib.isrFlowFillEx(_ptr(gbuffer_hwc), _ptr(out), _ptr(ws), h, w, int(threads), ctypes.c_void_p(st.cuda_stream))
    if rc != 0:
        raise RuntimeError("isrFlowFill failed (%d)" % rc)
    return out


def final_conv_finish(features, weight, bias, net_input, shading=None):
    """The network's last layer (64 -> 6, no activation) and ``finish_frame`` in one launch
    (``isrConvSmallFinishFrame``): features [1,Cin,4h,4w] (channel planes may be padded), net_input [1,>=5,h,w]
    -> (next_prev [1,6,4h,4w], rgb [1,3,4h,4w] or None)."""
    assert features.is_cuda and features.shape[0] == 1 and weight.shape[0] == 6
    lib = _sr()
    w8, b8 = _prepare_small(weight, bias)
    features, xp, _ = _plane_strides(features)
    net_input = net_input.contiguous()
    _, cin, H, W = features.shape
    h, w = H // 4, W // 4
    nxt = torch.empty((1, 6, H, W), dtype=torch.float32, device=features.device)
    rgb, params = None, None
    exponent, ao, inv, spec = 1, 0.0, 0, 0
    if shading is not None:
        rgb = torch.empty((1, 3, H, W), dtype=torch.float32, device=features.device)
        params = (ctypes.c_float * 18)(*shading.packed_parameters())
        exponent, ao = int(shading._specular_exponent), float(shading._ao)
        inv, spec = int(bool(shading.inverse_ao)), int(bool(shading.enable_specular))
    rc = lib.isrConvSmallFinishFrame(_ptr(features), _ptr(w8), _ptr(b8), _ptr(net_input), _ptr(nxt), _ptr(rgb), cin, h, w, xp,
                                     params, exponent, ao, inv, spec, _stream())
    if rc != 0:
        raise RuntimeError("isrConvSmallFinishFrame failed (%d)" % rc)
    return nxt, rgb


# ---- the fused 1080p tail (csrc/sr_conv_tail.hip): postblock.6 + postblock.8 + frame finish without the 64-channel round trip ----
TAIL_FUSION = True
_tail_cache = {}
_tail_ws = {}


def _prepare_tail(weight8):
    """The last layer's weights [6, 64, 3, 3] by (tap, channel) row in the z stage's operand order, cached like ``prepare_weights``."""
    lib = _sr()
    key = id(weight8)
    hit = _tail_cache.get(key)
    if hit is not None:
        ref, version, ptr, wz, epoch = hit
        if ref() is weight8 and version == weight8._version and ptr == weight8.data_ptr() and epoch == _images_epoch:
            return wz
    wz = torch.empty(lib.isrConvTailWeightBytes(), dtype=torch.uint8, device=weight8.device)
    rc = lib.isrConvTailPrepare(_ptr(weight8.detach().contiguous()), _ptr(wz), _stream())
    if rc != 0:
        raise RuntimeError("isrConvTailPrepare failed (%d)" % rc)
    if len(_tail_cache) > 64:
        _tail_cache.clear()
    _tail_cache[key] = (weakref.ref(weight8), weight8._version, weight8.data_ptr(), wz, _images_epoch)
    return wz


def tail_supported(features, weight6, weight8):
    """Can ``tail_conv_finish`` take this frame?  (64 -> 64 -> 6 channels, one image, 16-byte aligned rows; the default
    split-operand inference mode.)"""
    if not (TAIL_FUSION and SPLIT_F16 and not FAST_F16 and features.is_cuda and features.dtype == torch.float32):
        return False
    if any_hot(features.device):          # range guard: a layer of this model needs the exact kernels -- per-layer routing only
        return False
    if features.dim() != 4 or features.shape[0] != 1 or features.shape[1] != 64 or features.shape[2] % 4 or features.shape[3] % 4:
        return False
    if tuple(weight6.shape) != (64, 64, 3, 3) or tuple(weight8.shape) != (6, 64, 3, 3):
        return False
    f, xp, _ = _plane_strides(features)
    return f is features and bool(_sr().isrConvTailSupported(_ptr(features), features.shape[2] // 4, features.shape[3] // 4, xp))


class PackedSplit:
    """An activation tensor [1, C, H, W] in PACKED-SPLIT form: every value already split into the (hi, lo') fp16 pair the
    split-operand kernels multiply, eight channels of one pixel per 16-byte unit, ``data[part: hi | lo][C / 8][plane units]``
    (csrc/sr_split_common.h, SplitConvParams::ps).  Produced by ``conv3x3_split_packed``, consumed by ``tail_conv_finish``:
    the consumer stages its k-steps with plain 16-byte copies (LDS-DMA) -- same numbers as converting an fp32 tensor on the
    way in, without the conversions, the staging registers and the producer's LDS transposition."""

    def __init__(self, data, channels, h, w, plane):
        self.data, self.channels, self.h, self.w, self.plane = data, channels, h, w, plane
        self.range_key = None      # range guard: the producer whose flag word describes these values

    def to_float(self):
        """The fp32 tensor this stands for, hi + lo' 2^-11 (tests)."""
        u = self.data.view(torch.float16).view(2, self.channels // 8, self.plane, 8)[:, :, :self.h * self.w].float()
        v = u[0] + u[1] * (2.0 ** -11)
        return v.permute(0, 2, 1).reshape(1, self.channels, self.h, self.w)


TAIL_PACKED = os.environ.get("ISR_TAIL_PACKED", "1") != "0"       # hand postblock.4's output to the fused tail packed-split (frame pipeline)


def conv3x3_split_packed(x, weight, bias=None, act='relu', slope=0.01, upsample2x=False):
    """``conv3x3_split`` (no autograd, one image, Cout a multiple of 8) whose result is written PACKED-SPLIT."""
    lib = _sr()
    x, xp, _ = _plane_strides(x)
    assert x.shape[0] == 1 and weight.shape[0] % 8 == 0
    cin, cout = weight.shape[1], weight.shape[0]
    h, w = (2 * x.shape[2], 2 * x.shape[3]) if upsample2x else (x.shape[2], x.shape[3])
    if upsample2x and not lib.isrConvF16SupportsUpsample(x.data_ptr(), x.shape[3], xp, cin * xp):
        raise ValueError("conv3x3_split_packed: the fused upsampling needs 16-byte aligned low-resolution rows")
    plane = h * w + plane_pad(h, w)
    data = torch.empty(2 * (cout // 8) * plane * 4, dtype=torch.int32, device=x.device)
    key = _arm_range(id(weight), x.device)
    rc = lib.isrConv3x3ForwardSplitPacked(_ptr(x), _ptr(_prepare_split(weight)), _ptr(bias.detach().contiguous() if bias is not None else None),
                                          _ptr(data), cin, h, w, cout, ACT_CODES[act], float(slope), 1 if upsample2x else 0, xp, plane, _stream())
    if rc != 0:
        raise RuntimeError("isrConv3x3ForwardSplitPacked failed (%d)" % rc)
    out = PackedSplit(data, cout, h, w, plane)
    out.range_key = key
    return out


def conv3x3_split_from_packed(xp, weight, bias=None, act='none', slope=0.01, residual=None, packed_out=False):
    """The split-operand convolution of a PACKED-SPLIT input (``PackedSplit``, staged by LDS-DMA): act(conv3x3(x) + bias)
    + residual as an fp32 tensor, or (``packed_out``, no residual) packed-split again."""
    lib = _sr()
    cout, cin = weight.shape[0], weight.shape[1]
    assert isinstance(xp, PackedSplit) and xp.channels == cin and cin % 8 == 0
    h, w = xp.h, xp.w
    rp = 0
    if residual is not None:
        residual, rp, _ = _plane_strides(residual)
    if packed_out:
        assert residual is None and cout % 8 == 0
        plane = h * w + plane_pad(h, w)
        out = torch.empty(2 * (cout // 8) * plane * 4, dtype=torch.int32, device=xp.data.device)
        yplane = plane
    else:
        out = empty_planes(1, cout, h, w, xp.data.device)
        yplane = out.stride(1)
    key = _arm_range(id(weight), xp.data.device)
    rc = lib.isrConv3x3ForwardSplitFromPacked(_ptr(xp.data), _ptr(_prepare_split(weight)), _ptr(bias.detach().contiguous() if bias is not None else None),
                                              _ptr(residual), _ptr(out), 1 if packed_out else 0, cin, h, w, cout, ACT_CODES[act], float(slope),
                                              xp.plane, yplane, rp, _stream())
    if rc != 0:
        raise RuntimeError("isrConv3x3ForwardSplitFromPacked failed (%d)" % rc)
    if packed_out:
        out = PackedSplit(out, cout, h, w, yplane)
        out.range_key = key
    else:
        out._isr_range_key = key
    return out


# ---- phase-decomposed upsampling layers (csrc/sr_conv_upsp.h): conv3x3(U2(x)) without interpolation at run time ------------------
# U2 (bilinear x2) is linear: the convolution at the high-resolution pixel (2y + py, 2x + px) is a 3 x 3 convolution of the LOW-
# resolution image with one of four effective weight sets.  Input and output are PACKED-SPLIT; the output's one-pixel frame (where the
# convolution's zero padding bites) is computed by a small exact kernel of its own.  Not bit-identical to the interpolate-then-convolve
# kernels (different roundings), the same distance from an fp64 convolution (tests/test_upsp_gpu.py).
# OPT-IN (ISR_UPS_PHASE=1): built, parity-green and MEASURED SLOWER in round 5 -- 1080p layer 0.60 ms + 0.08 ms frame kernel against
# 0.47 ms of conv3x3_split_ups3_kernel.  Why (profiles/r05_upsp_ablation.md): with every operand arriving by LDS-DMA the launch's
# MFMAs + operand reads alone take 0.345 ms at the clock the matrix pipe really holds (~1.5 GHz under load, not the 2.4 GHz the
# roofline's peak is priced at), the operand DMA adds 0.08 ms of LDS write traffic beside 0.67 LDS reads per MFMA, and the four
# epilogues per tile with their stride-2 pixel stores 0.17 ms; the interpolating kernel was never more than 35 % above that floor.
UPS_PHASE = os.environ.get("ISR_UPS_PHASE", "0") == "1"
_upsp_cache = {}


def _prepare_ups_phase(weight):
    """The stacked image of the four effective weight sets of a [64, 64, 3, 3] weight (``isrConvUpsPhasePrepare``), cached per weight version."""
    lib = _sr()
    key = id(weight)
    hit = _upsp_cache.get(key)
    if hit is not None:
        ref, version, ptr, wq, epoch = hit
        if ref() is weight and version == weight._version and ptr == weight.data_ptr() and epoch == _images_epoch:
            return wq
    w = weight.detach().contiguous()
    wq = torch.empty(lib.isrConvUpsPhaseWeightBytes(), dtype=torch.uint8, device=weight.device)
    scratch = torch.empty(lib.isrConvUpsPhaseScratchBytes(), dtype=torch.uint8, device=weight.device)
    rc = lib.isrConvUpsPhasePrepare(_ptr(w), _ptr(wq), _ptr(scratch), _stream())
    if rc != 0:
        raise RuntimeError("isrConvUpsPhasePrepare failed (%d)" % rc)
    if len(_upsp_cache) > 64:
        for k in [k for k, v in _upsp_cache.items() if v[0]() is None]:
            del _upsp_cache[k]
    _upsp_cache[key] = (weakref.ref(weight), weight._version, weight.data_ptr(), wq, _images_epoch)
    return wq


def pack_split(x):
    """fp32 [1, C, H, W] -> ``PackedSplit`` (what a producer's packed epilogue would have written; ``isrPackSplit``)."""
    x, xp, _ = _plane_strides(x)
    _, c, h, w = x.shape
    assert x.shape[0] == 1 and c % 8 == 0
    plane = h * w + plane_pad(h, w)
    data = torch.empty(2 * (c // 8) * plane * 4, dtype=torch.int32, device=x.device)
    rc = _sr().isrPackSplit(_ptr(x), _ptr(data), c, h, w, xp, plane, _stream())
    if rc != 0:
        raise RuntimeError("isrPackSplit failed (%d)" % rc)
    out = PackedSplit(data, c, h, w, plane)
    out.range_key = getattr(x, '_isr_range_key', None)
    return out


def ups_phase_supported(xp, weight):
    """Can ``conv3x3_ups_phase`` take this layer (a 64 -> 64 convolution of the x2-upsampled packed-split tensor ``xp``)?"""
    if not (UPS_PHASE and SPLIT_F16 and not FAST_F16 and isinstance(xp, PackedSplit) and xp.channels == 64 and tuple(weight.shape) == (64, 64, 3, 3)):
        return False
    if any_hot(xp.data.device) or range_is_hot(xp.range_key, xp.data.device):
        return False
    H, W = 2 * xp.h, 2 * xp.w
    return bool(_sr().isrConvUpsPhaseSupported(64, 64, xp.h, xp.w, xp.plane, H * W + plane_pad(H, W)))


def conv3x3_ups_phase(xp, weight, bias=None, act='relu', slope=0.01):
    """act(conv3x3(U2(x), weight) + bias) of a packed-split x [64, h, w] -> packed-split [64, 2h, 2w] (``isrConvUpsPhase``)."""
    lib = _sr()
    assert isinstance(xp, PackedSplit) and xp.channels == 64 and tuple(weight.shape) == (64, 64, 3, 3)
    H, W = 2 * xp.h, 2 * xp.w
    plane = H * W + plane_pad(H, W)
    dev = xp.data.device
    data = torch.empty(2 * 8 * plane * 4, dtype=torch.int32, device=dev)
    wq = _prepare_ups_phase(weight)
    key = _arm_range(id(weight), dev)
    rc = lib.isrConvUpsPhase(_ptr(xp.data), _ptr(wq), _ptr(weight.detach().contiguous()), _ptr(bias.detach().contiguous() if bias is not None else None),
                             _ptr(data), xp.h, xp.w, ACT_CODES[act], float(slope), xp.plane, plane, _stream())
    if rc != 0:
        raise RuntimeError("isrConvUpsPhase failed (%d)" % rc)
    out = PackedSplit(data, 64, H, W, plane)
    out.range_key = key
    return out


def packed_supported(x, weight, upsample2x):
    """Can ``conv3x3_split_packed`` + the packed tail take this layer?"""
    if not (TAIL_PACKED and x.is_cuda and x.dtype == torch.float32 and x.shape[0] == 1 and weight.shape[0] == 64):
        return False
    if any_hot(x.device):
        return False
    x2, xp, _ = _plane_strides(x)
    if x2 is not x:
        return False
    h, w = (2 * x.shape[2], 2 * x.shape[3]) if upsample2x else (x.shape[2], x.shape[3])
    if h % 4 or w % 4 or 16 * 16 * (h * w + plane_pad(h, w)) >= 2 ** 31:
        return False
    return (not upsample2x) or bool(_sr().isrConvF16SupportsUpsample(x.data_ptr(), x.shape[3], xp, x.shape[1] * xp))


def tail_conv_finish(features, weight6, bias6, weight8, bias8, net_input, shading=None, out=None):
    """features [1,64,4h,4w] (the output of postblock.4; channel planes may be padded) -> relu(conv3x3(., weight6) + bias6)
    -> conv3x3(., weight8) + bias8 -> ``finish_frame``: (next_prev [1,6,4h,4w], rgb [1,3,4h,4w] or None) in two launches
    (``isrConvTailFinishFrame``); the 64-channel tensor between the two convolutions never exists in memory.
    ``out``: (next_prev, rgb) tensors to write into (the frame graph's static output buffers)."""
    lib = _sr()
    packed = isinstance(features, PackedSplit)
    if packed:
        H, W, xp, dev = features.h, features.w, features.plane, features.data.device
        assert features.channels == 64
    else:
        features, xp, _ = _plane_strides(features)
        _, _, H, W = features.shape
        dev = features.device
    net_input = net_input.contiguous()
    h, w = H // 4, W // 4
    wq6 = _prepare_split(weight6)
    wz = _prepare_tail(weight8)
    key = (dev, h, w, torch.cuda.current_stream().cuda_stream)
    ws = _tail_ws.get(key)
    if ws is None:
        ws = torch.empty(lib.isrConvTailWorkspaceBytes(h, w), dtype=torch.uint8, device=dev)
        _tail_ws[key] = ws
    nxt = out[0] if out is not None else torch.empty((1, 6, H, W), dtype=torch.float32, device=dev)
    assert tuple(nxt.shape) == (1, 6, H, W) and nxt.is_contiguous() and nxt.dtype == torch.float32
    rgb, params = None, None
    exponent, ao, inv, spec = 1, 0.0, 0, 0
    if shading is not None:
        rgb = out[1] if out is not None else torch.empty((1, 3, H, W), dtype=torch.float32, device=dev)
        assert tuple(rgb.shape) == (1, 3, H, W) and rgb.is_contiguous()
        params = (ctypes.c_float * 18)(*shading.packed_parameters())
        exponent, ao = int(shading._specular_exponent), float(shading._ao)
        inv, spec = int(bool(shading.inverse_ao)), int(bool(shading.enable_specular))
    b6 = bias6.detach().contiguous() if bias6 is not None else None
    b8 = bias8.detach().contiguous() if bias8 is not None else torch.zeros(6, dtype=torch.float32, device=dev)
    _arm_range(("tail", id(weight6)), dev, members=[id(weight6)])         # the largest |y6|: the intermediate is split in registers inside the kernel
    fn = lib.isrConvTailFinishFramePacked if packed else lib.isrConvTailFinishFrame
    rc = fn(_ptr(features.data if packed else features), _ptr(wq6), _ptr(b6), _ptr(wz), _ptr(b8), _ptr(ws), _ptr(net_input), _ptr(nxt), _ptr(rgb),
                                    h, w, xp, params, exponent, ao, inv, spec, _stream())
    if rc != 0:
        raise RuntimeError("isrConvTailFinishFrame failed (%d)" % rc)
    return nxt, rgb


def finish_frame(raw, net_input, shading=None):
    """Conv output before reconstruction [1,6,4h,4w] + network input [1,>=5,h,w] -> (next_prev [1,6,4h,4w],
    rgb [1,3,4h,4w] or None) in one launch (``isrFinishFrame``).  ``shading``: a utils.ScreenSpaceShading."""
    raw = raw.contiguous()
    net_input = net_input.contiguous()
    _, _, H, W = raw.shape
    h, w = H // 4, W // 4
    nxt = torch.empty_like(raw)
    rgb = None
    params = None
    exponent, ao, inv, spec = 1, 0.0, 0, 0
    if shading is not None:
        rgb = torch.empty((1, 3, H, W), dtype=torch.float32, device=raw.device)
        vals = shading.packed_parameters()
        params = (ctypes.c_float * 18)(*vals)
        exponent, ao = int(shading._specular_exponent), float(shading._ao)
        inv, spec = int(bool(shading.inverse_ao)), int(bool(shading.enable_specular))
    rc = _sr().isrFinishFrame(_ptr(raw), _ptr(net_input), _ptr(nxt), _ptr(rgb), h, w, params, exponent, ao, inv, spec, _stream())
    if rc != 0:
        raise RuntimeError("isrFinishFrame failed (%d)" % rc)
    return nxt, rgb
