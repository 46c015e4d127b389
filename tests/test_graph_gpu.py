"""The frame as one HIP graph (pipeline.SuperResolutionPipeline(graph=True)): steady-state frames replay a captured
graph -- input assembly, dataflow trunk, the fork to the side stream (next frame's ray-march from the device camera block + flow fill),
upsampling layers, fused tail, guard mirror -- and must be the eager frames bit for bit
(SuperresolutionNetwork/mainComparisonVideo3.py:430-530: the per-frame call sequence)."""
import argparse

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
OPT = argparse.Namespace(upsample='bilinear', reconType='residual', useBN=False, numResidualLayers=10)


def _pipes(low=(96, 56)):
    from isosurfacesuperresolution_amd import models, volumes as V
    from isosurfacesuperresolution_amd.inference import DirectRenderer, LoadedModel
    from isosurfacesuperresolution_amd.pipeline import SuperResolutionPipeline, default_shading
    torch.manual_seed(4)
    net = models.createNetwork('EnhanceNet', 4, 101, [0, 1, 2, 3, 4], 6, OPT)
    with torch.no_grad():
        for b in net.blocks:
            b[0].weight.mul_(0.4); b[2].weight.mul_(0.4)
    r = DirectRenderer()
    r.load_dense(V.ejecta(64))
    lm = LoadedModel.from_model(net, "cuda", parameters={"initialImage": "zero"})
    mk = lambda graph: SuperResolutionPipeline(r, lm, default_shading("cuda", 30.0), low, graph=graph)
    return r, mk, V


def _run(pipe, V, seq):
    """seq: list of (frame index, next index or None); returns clones of (rgb, raw, gbuffer)."""
    out = []
    pipe.renderer.set_last_camera(V.quantize3(V.orbit_camera(seq[0][0] - 1)))      # the flow reference of the first frame (the renderer is shared)
    for k, nk in seq:
        rgb, raw = pipe.frame(V.orbit_camera(k), V.orbit_camera(nk) if nk is not None else None)
        torch.cuda.synchronize()
        out.append((rgb.clone(), raw.clone(), pipe.gbuffer.clone()))
    return out


def test_render_from_camera_block_equals_the_ordinary_render():
    r, _, V = _pipes()
    for c, v in (("cameraLookAt", "0,0,0"), ("cameraUp", "0,1,0"), ("cameraFoV", "30.000"), ("isovalue", "0.340"), ("aosamples", "0"),
                 ("resolution", "96,56"), ("viewport", "0,0,96,56")):
        r.send_command(c, v)
    a, b = (torch.empty((56, 96, 12), device="cuda") for _ in range(2))
    block = torch.zeros(r.frame_block_bytes(), dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream()
    for k in range(4):
        r.set_last_camera(V.quantize3(V.orbit_camera(k - 1)))
        r.send_command("cameraOrigin", V.fmt3(V.orbit_camera(k)))
        r.render_async(a, s)
        r.set_last_camera(V.quantize3(V.orbit_camera(k - 1)))
        r.write_frame_block(block, s)                       # makes camera k the flow reference, as the render above did
        r.render_from_block(b, block, s)
        torch.cuda.synchronize()
        assert a[..., 3].sum().item() > 100 and torch.equal(a, b), k


def test_graph_frames_equal_eager_frames_bit_for_bit():
    _, mk, V = _pipes()
    # a sequence with a steady run, a camera jump whose prefetched frame is discarded, a frame without a successor, and a reset
    seq = [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 9), (7, 8), (8, 9), (9, None), (10, 11), (11, 12), (12, 13)]
    eager = mk(False)
    ref = _run(eager, V, seq[:9]); eager.reset(); ref += _run(eager, V, seq[9:])
    pipe = mk(True)
    assert pipe.graph
    got = _run(pipe, V, seq[:9]); pipe.reset(); got += _run(pipe, V, seq[9:])
    assert pipe.graph_replays >= 3                               # slots 1 and 0 captured at frames 1 and 2, replays afterwards
    for i, ((rgb_a, raw_a, g_a), (rgb_b, raw_b, g_b)) in enumerate(zip(ref, got)):
        assert torch.equal(g_a, g_b), i
        assert torch.equal(raw_a, raw_b), (i, (raw_a - raw_b).abs().max().item())
        assert torch.equal(rgb_a, rgb_b), i
    assert ref[3][0].std().item() > 0.01


def test_graph_is_recaptured_when_static_parameters_change():
    _, mk, V = _pipes()
    pipe, eager = mk(True), mk(False)
    seq = [(k, k + 1) for k in range(5)]
    _run(pipe, V, seq)
    n0 = pipe.graph_replays
    assert n0 >= 2
    for p in (pipe, eager):
        p.shading.ambient_light_color(np.array([0.3, 0.2, 0.1]))        # baked into the captured finishing kernel's arguments
        p.set_static(fov=30.0, isovalue=0.30)
        p.reset()
    a = _run(eager, V, seq)
    b = _run(pipe, V, seq)
    for (rgb_a, raw_a, _), (rgb_b, raw_b, _) in zip(a, b):
        assert torch.equal(raw_a, raw_b) and torch.equal(rgb_a, rgb_b)
    assert pipe.graph_replays > n0


def test_a_timeout_under_graph_mode_drops_the_captured_frames_and_the_fallback_forms_take_over(diag_lib):
    """a captured frame holds the dataflow-trunk launch, its workspace pointers and guard words.  A launch that times
    out (device-induced fault) raises at the next frame's start; the captured frames are dropped before the error leaves, the
    next frames run on the per-layer kernels (eagerly, then captured again) and equal the eager pipeline bit for bit."""
    from isosurfacesuperresolution_amd import ops
    lib = ops._sr()
    _, mk, V = _pipes()
    seq = [(k, k + 1) for k in range(4)]
    eager = mk(False)
    pipe = mk(True)
    try:
        _run(pipe, V, seq)
        assert pipe.graph_replays >= 1 and all(g is not None for g in pipe._graphs)
        lib.isrDebugSetTrunkFault(5, 200000)                     # tile 5 never publishes, 2 ms deadline: the REPLAYED launch reads these
        # (the fault switches are launch parameters: a captured launch keeps the values of its capture -- so capture again with them)
        pipe._graphs, pipe._graph_sig = [None, None], None
        pipe.frame(V.orbit_camera(4), V.orbit_camera(5))         # eager warm-up + capture of the faulty launch
        lib.isrDebugSetTrunkFault(-1, 0)
        torch.cuda.synchronize()
        with pytest.raises(RuntimeError, match="timed out waiting for its neighbours"):
            pipe.frame(V.orbit_camera(5), V.orbit_camera(6))
        assert ops.TRUNK_DATAFLOW is False and pipe._graphs == [None, None]
        # go on: a new sequence from frame 4 (the frame that was lost) on the fallback forms
        pipe.reset()
        eager.reset()
        a = _run(eager, V, [(4, 5), (5, 6), (6, 7), (7, 8)])
        b = _run(pipe, V, [(4, 5), (5, 6), (6, 7), (7, 8)])
        for (rgb_a, raw_a, _), (rgb_b, raw_b, _) in zip(a, b):
            assert torch.equal(raw_a, raw_b) and torch.equal(rgb_a, rgb_b)
        assert all(g is not None for g in pipe._graphs)          # captured again, without the dataflow trunk
    finally:
        lib.isrDebugSetTrunkFault(-1, 0)
        ops.TRUNK_DATAFLOW = True
        torch.cuda.synchronize()
        ops._range_state("cuda")["buf"][ops._TRUNK_ERROR_SLOT] = 0


def test_an_in_place_weight_update_invalidates_the_captured_frames():
    """load_state_dict bumps the weights' versions; the eager path then builds new split images and drops the old
    ones, which a captured frame would keep reading.  The signature carries every parameter's (version, address)."""
    _, mk, V = _pipes()
    pipe, eager = mk(True), mk(False)
    seq = [(k, k + 1) for k in range(4)]
    _run(pipe, V, seq)
    assert all(g is not None for g in pipe._graphs)
    net = pipe.model.model
    with torch.no_grad():
        sd = {k: v * 0.9 for k, v in net.state_dict().items()}
    net.load_state_dict(sd)
    pipe.reset(); eager.reset()
    a = _run(eager, V, seq)
    b = _run(pipe, V, seq)
    for (rgb_a, raw_a, _), (rgb_b, raw_b, _) in zip(a, b):
        assert torch.equal(raw_a, raw_b) and torch.equal(rgb_a, rgb_b)
