"""The dataflow trunk (csrc/sr_conv_trunk.hip): preblock + ten residual blocks of one image in ONE persistent launch with
per-tile progress counters, against the per-layer launches -- same products in the same order: EQUAL bit for bit
(SuperresolutionNetwork/models/enhancenet.py:92-112,136-141)."""
import argparse

import pytest
import torch

pytestmark = pytest.mark.gpu
OPT = argparse.Namespace(upsample='bilinear', reconType='residual', useBN=False, numResidualLayers=10)


def _net(seed):
    from isosurfacesuperresolution_amd import models
    torch.manual_seed(seed)
    return models.createNetwork('EnhanceNet', 4, 101, [0, 1, 2, 3, 4], 6, OPT).cuda().eval()


def _per_layer(net, x):
    from isosurfacesuperresolution_amd import ops
    ops.TRUNK_DATAFLOW = False
    try:
        with torch.no_grad():
            return net.forward_features(x, last_three=False)
    finally:
        ops.TRUNK_DATAFLOW = True


@pytest.mark.parametrize("h,w", [(8, 32), (9, 36), (41, 100), (270, 480), (128, 512)])
def test_dataflow_trunk_is_bit_identical_to_the_per_layer_launches(h, w):
    from isosurfacesuperresolution_amd import ops
    net = _net(h + w)
    x = torch.rand(1, 101, h, w, device="cuda") * 2 - 0.5
    pre = net.preblock[0]
    convs = [(pre.weight, pre.bias)] + [(m.weight, m.bias) for b in net.blocks for m in (b[0], b[2])]
    with torch.no_grad():
        assert ops.trunk_supported(x, convs)
        f = ops.trunk_dataflow(x, convs)
        f2 = ops.trunk_dataflow(x, convs)
        ref = x
        ref = ops.conv3x3_split(x, pre.weight, pre.bias, act='relu')
        for b in net.blocks:
            ref = ops.conv3x3_split(ops.conv3x3_split(ref, b[0].weight, b[0].bias, act='relu'), b[2].weight, b[2].bias, residual=ref)
    torch.cuda.synchronize()
    ops.trunk_check()
    assert torch.equal(f, ref), (f - ref).abs().max().item()
    assert torch.equal(f, f2)


def test_network_uses_the_dataflow_trunk_and_is_unchanged(diag_lib):
    from isosurfacesuperresolution_amd import ops
    net = _net(3)
    x = torch.rand(1, 101, 270, 480, device="cuda")
    ops.profile_enable(True)
    with torch.no_grad():
        f = net.forward_features(x, last_three=False)
    torch.cuda.synchronize()
    names = [n for n, _, _ in ops.profile_records()]
    ops.profile_enable(False)
    assert names.count("trunk_dataflow_kernel") == 1 and not any(n.startswith("conv3x3_split_kernel<false>") for n in names)
    assert torch.equal(f, _per_layer(net, x))
    ops.trunk_check()
    # more tiles than CUs (960 x 540 has 1020): the multi-tile form of the dataflow trunk (or, with it off, the per-layer kernels)
    big = torch.rand(1, 101, 540, 960, device="cuda")
    pre = net.preblock[0]
    convs = [(pre.weight, pre.bias)] + [(m.weight, m.bias) for b in net.blocks for m in (b[0], b[2])]
    with torch.no_grad():
        assert ops.trunk_supported(big, convs)
        ops._sr().isrDebugSetTrunkMultiTile(0)
        try:
            assert not ops.trunk_supported(big, convs)
        finally:
            ops._sr().isrDebugSetTrunkMultiTile(1)


def test_dataflow_trunk_under_uneven_load_and_repeated_launches():
    """The hand-off between workgroups must hold when the chip is busy with something else and the consumers' caches are warm:
    the trunk runs forty times back to back while another stream keeps the GPU loaded with bandwidth-heavy work; every result
    is compared word for word."""
    from isosurfacesuperresolution_amd import ops
    net = _net(5)
    x = torch.rand(1, 101, 270, 480, device="cuda")
    pre = net.preblock[0]
    convs = [(pre.weight, pre.bias)] + [(m.weight, m.bias) for b in net.blocks for m in (b[0], b[2])]
    with torch.no_grad():
        ref = ops.conv3x3_split(x, pre.weight, pre.bias, act='relu')
        for b in net.blocks:
            ref = ops.conv3x3_split(ops.conv3x3_split(ref, b[0].weight, b[0].bias, act='relu'), b[2].weight, b[2].bias, residual=ref)
    side = torch.cuda.Stream()
    junk = torch.rand(64 * 1024 * 1024, device="cuda")
    outs = []
    with torch.no_grad():
        for k in range(40):
            if k % 2 == 0:
                with torch.cuda.stream(side):
                    for _ in range(3):
                        junk.mul_(1.0001).add_(0.5)
            outs.append(ops.trunk_dataflow(x, convs))
    torch.cuda.synchronize()
    ops.trunk_check()
    bad = [k for k, o in enumerate(outs) if not torch.equal(o, ref)]
    assert not bad, "launches %s differ (max |diff| %.3g)" % (bad, max((outs[k] - ref).abs().max().item() for k in bad))


@pytest.mark.parametrize("cin,nblocks,h,w", [(5, 1, 33, 70), (16, 0, 16, 32), (40, 2, 50, 64), (101, 3, 270, 480)])
def test_dataflow_trunk_with_other_depths_and_input_widths(cin, nblocks, h, w):
    """Fewer input channel groups than the 8 of the inner tensors (the packing kernel still has to end every plane of all three
    packed-split tensors with its zero unit), no blocks at all (the preblock's result goes straight to y), and consecutive launches
    on DIFFERENT inputs through the same workspace (nothing of the previous launch may survive in it)."""
    from isosurfacesuperresolution_amd import ops
    g = torch.Generator().manual_seed(cin * 100 + nblocks)
    convs = [(((torch.rand(64, cin if k == 0 else 64, 3, 3, generator=g) - 0.5) * 0.15).cuda(), ((torch.rand(64, generator=g) - 0.5) * 0.2).cuda())
             for k in range(1 + 2 * nblocks)]
    with torch.no_grad():
        for trial in range(3):
            x = ((torch.rand(1, cin, h, w, generator=g) - 0.4) * (1 + trial)).cuda()
            assert ops.trunk_supported(x, convs)
            f = ops.trunk_dataflow(x, convs)
            ref = ops.conv3x3_split(x, convs[0][0], convs[0][1], act='relu')
            for k in range(nblocks):
                t = ops.conv3x3_split(ref, convs[2 * k + 1][0], convs[2 * k + 1][1], act='relu')
                ref = ops.conv3x3_split(t, convs[2 * k + 2][0], convs[2 * k + 2][1], residual=ref)
            torch.cuda.synchronize()
            ops.trunk_check()
            assert torch.equal(f, ref), (trial, (f - ref).abs().max().item())


def test_error_word_of_the_dataflow_trunk_is_sticky():
    """A tile that gives up on a neighbour leaves 1 + layer in the error word (the last of the per-frame guard words,
    isrSetTrunkErrorWord); later launches must not clear it, the check that reports it does -- and switches the dataflow form off."""
    from isosurfacesuperresolution_amd import ops
    g = torch.Generator().manual_seed(9)
    convs = [(((torch.rand(64, 16 if k == 0 else 64, 3, 3, generator=g) - 0.5) * 0.1).cuda(), None) for k in range(3)]
    x = torch.rand(1, 16, 40, 70, generator=g).cuda()
    try:
        with torch.no_grad():
            ops.trunk_dataflow(x, convs)
            torch.cuda.synchronize()
            ops.trunk_check()
            st = ops._range_state(x.device)
            st["buf"][ops._TRUNK_ERROR_SLOT] = 7   # as if layer 6 had timed out in some earlier launch
            ops.trunk_dataflow(x, convs)
            ops.trunk_dataflow(x, convs)
            torch.cuda.synchronize()
            with pytest.raises(RuntimeError, match="layer 6"):
                ops.trunk_check()
            assert ops.TRUNK_DATAFLOW is False     # whoever catches the error continues on the per-layer kernels
            ops.trunk_check()                      # reported once, then reset
    finally:
        ops.TRUNK_DATAFLOW = True


def test_a_timeout_induced_on_the_device_raises_at_the_start_of_the_next_frame(diag_lib):
    """the timeout path itself (deadline -> LDS flag -> atomicMax(error) -> early return), executed by the
    kernel: a diagnostic switch makes ONE tile never publish its progress and shortens the deadline to 2 ms.  The frame with the
    fault returns without any host synchronisation; the NEXT frame's start reads the guard words the faulty frame mirrored into
    pinned memory, raises, and leaves the dataflow form off -- the frame after that runs on the per-layer kernels and is right."""
    from isosurfacesuperresolution_amd import ops, volumes as V
    from isosurfacesuperresolution_amd.inference import DirectRenderer, LoadedModel
    from isosurfacesuperresolution_amd.pipeline import SuperResolutionPipeline, default_shading
    lib = ops._sr()
    net = _net(21)
    r = DirectRenderer()
    r.load_dense(V.ejecta(64))
    lm = LoadedModel.from_model(net, "cuda", parameters={"initialImage": "zero"})
    pipe = SuperResolutionPipeline(r, lm, default_shading("cuda", 30.0), (96, 56), temporal=False)      # 4 x 3 tiles of 16 x 32
    pipe.set_static(fov=30.0, isovalue=0.34)
    try:
        for k in range(3):
            pipe.frame(V.orbit_camera(k))                            # clean frames (the first one takes the synchronous check)
        torch.cuda.synchronize()
        lib.isrDebugSetTrunkFault(5, 200000)                         # tile 5 never publishes; deadline 2 ms after the kernel's start
        assert ops.debug_switches() != 0                             # bench.py would refuse to report in this state
        _, raw_bad = pipe.frame(V.orbit_camera(7))                   # returns: nothing inside a frame waits for the host
        raw_bad = raw_bad.clone()
        lib.isrDebugSetTrunkFault(-1, 0)
        torch.cuda.synchronize()
        assert int(ops._range_state(raw_bad.device)["buf"][ops._TRUNK_ERROR_SLOT].item()) >= 2     # set by the KERNEL: 1 + layer
        with pytest.raises(RuntimeError, match="timed out waiting for its neighbours"):
            pipe.frame(V.orbit_camera(7))                            # the next frame, not frame 256
        assert ops.TRUNK_DATAFLOW is False
        _, raw_again = pipe.frame(V.orbit_camera(7))                 # per-layer kernels: bit-identical to the dataflow form
        torch.cuda.synchronize()
        raw_again = raw_again.clone()
        assert not torch.equal(raw_bad, raw_again)                   # the faulty launch's output was incomplete (whatever its buffer held before)
        ops.TRUNK_DATAFLOW = True
        _, raw_flow = pipe.frame(V.orbit_camera(7))                  # the dataflow form again, undisturbed
        torch.cuda.synchronize()
        assert torch.equal(raw_again, raw_flow)
        pipe.frame(V.orbit_camera(3))                                # and nothing is left pending
    finally:
        lib.isrDebugSetTrunkFault(-1, 0)
        ops.TRUNK_DATAFLOW = True
        torch.cuda.synchronize()
        ops._range_state("cuda")["buf"][ops._TRUNK_ERROR_SLOT] = 0


@pytest.mark.parametrize("h,w", [(270, 960), (300, 500), (540, 960), (272, 481)])
def test_multi_tile_trunk_is_bit_identical_to_the_per_layer_launches(h, w):
    """trunk_mt_kernel (images of more tiles than CUs: a workgroup owns every 256th tile, the residual stream lives in the result
    tensor): the network's trunk equals the 21 per-layer launches bit for bit, twice in a row through one workspace."""
    from isosurfacesuperresolution_amd import ops
    net = _net(h + w)
    x = torch.rand(1, 101, h, w, device="cuda") * 2 - 0.5
    pre = net.preblock[0]
    convs = [(pre.weight, pre.bias)] + [(m.weight, m.bias) for b in net.blocks for m in (b[0], b[2])]
    ops.profile_enable(True)
    with torch.no_grad():
        assert ops.trunk_supported(x, convs)
        f = ops.trunk_dataflow(x, convs)
        f2 = ops.trunk_dataflow(x * 0.5, convs)
        f3 = ops.trunk_dataflow(x, convs)
    torch.cuda.synchronize()
    names = [n for n, _, _ in ops.profile_records()]
    ops.profile_enable(False)
    assert names.count("trunk_mt_kernel") == 3
    ops.trunk_check()

    def layers(inp):
        with torch.no_grad():
            r = ops.conv3x3_split(inp, pre.weight, pre.bias, act='relu')
            for b in net.blocks:
                r = ops.conv3x3_split(ops.conv3x3_split(r, b[0].weight, b[0].bias, act='relu'), b[2].weight, b[2].bias, residual=r)
        return r
    ref = layers(x)
    assert torch.equal(f, ref), (f - ref).abs().max().item()
    assert torch.equal(f3, ref) and not torch.equal(f2, ref)
    assert torch.equal(f2, layers(x * 0.5))


@pytest.mark.parametrize("cin,nblocks,h,w", [(101, 10, 270, 480), (5, 1, 33, 70), (16, 0, 16, 32), (40, 2, 50, 64), (24, 3, 300, 520)])
def test_multi_tile_form_forced_on_every_size_equals_the_one_tile_form(cin, nblocks, h, w, diag_lib):
    from isosurfacesuperresolution_amd import ops
    g = torch.Generator().manual_seed(cin * 100 + nblocks)
    convs = [(((torch.rand(64, cin if k == 0 else 64, 3, 3, generator=g) - 0.5) * 0.15).cuda(), ((torch.rand(64, generator=g) - 0.5) * 0.2).cuda())
             for k in range(1 + 2 * nblocks)]
    lib = ops._sr()
    with torch.no_grad():
        for trial in range(2):
            x = ((torch.rand(1, cin, h, w, generator=g) - 0.4) * (1 + trial)).cuda()
            ref = ops.conv3x3_split(x, convs[0][0], convs[0][1], act='relu')
            for k in range(nblocks):
                t = ops.conv3x3_split(ref, convs[2 * k + 1][0], convs[2 * k + 1][1], act='relu')
                ref = ops.conv3x3_split(t, convs[2 * k + 2][0], convs[2 * k + 2][1], residual=ref)
            lib.isrDebugSetTrunkMultiTile(2)
            try:
                f = ops.trunk_dataflow(x, convs)
            finally:
                lib.isrDebugSetTrunkMultiTile(1)
            torch.cuda.synchronize()
            ops.trunk_check()
            assert torch.equal(f, ref), (trial, (f - ref).abs().max().item())


@pytest.mark.parametrize("cin,nblocks,h,w", [(101, 10, 270, 480), (5, 1, 33, 70), (16, 0, 16, 32), (40, 2, 50, 64)])
def test_four_rows_per_wave_form_equals_the_default_form(cin, nblocks, h, w):
    """isrSetTrunkRows(4) / ISR_TRUNK_ROWS=4: four waves per workgroup, each 4 rows x 64 channels (one wave per SIMD, product-major MFMA
    order); per accumulator the same products in the same order, so the same bits.  Opt-in: measured slower."""
    from isosurfacesuperresolution_amd import ops
    g = torch.Generator().manual_seed(cin * 10 + nblocks)
    convs = [(((torch.rand(64, cin if k == 0 else 64, 3, 3, generator=g) - 0.5) * 0.15).cuda(), ((torch.rand(64, generator=g) - 0.5) * 0.2).cuda())
             for k in range(1 + 2 * nblocks)]
    lib = ops._sr()
    with torch.no_grad():
        x = ((torch.rand(1, cin, h, w, generator=g) - 0.4)).cuda()
        ref = ops.trunk_dataflow(x, convs).clone()
        lib.isrSetTrunkRows(4)
        try:
            f = ops.trunk_dataflow(x, convs).clone()
        finally:
            lib.isrSetTrunkRows(2)
        torch.cuda.synchronize()
        ops.trunk_check()
        assert torch.equal(f, ref), (f - ref).abs().max().item()


def test_multi_tile_trunk_gives_up_loudly_when_a_tile_never_publishes(diag_lib):
    """The multi-tile form's waits have the same deadline and error word as the one-tile form's: a tile that never publishes
    (isrDebugSetTrunkFault, deadline 2 ms) ends the launch with the word set by the kernel -- no hang -- and the next launch is right."""
    from isosurfacesuperresolution_amd import ops
    lib = ops._sr()
    g = torch.Generator().manual_seed(5)
    convs = [(((torch.rand(64, 24 if k == 0 else 64, 3, 3, generator=g) - 0.5) * 0.15).cuda(), ((torch.rand(64, generator=g) - 0.5) * 0.2).cuda())
             for k in range(5)]
    x = ((torch.rand(1, 24, 300, 520, generator=g) - 0.4)).cuda()                 # 19 x 17 = 323 tiles: more than CUs
    st = ops._range_state(x.device)
    try:
        with torch.no_grad():
            good = ops.trunk_dataflow(x, convs)
            torch.cuda.synchronize()
            lib.isrDebugSetTrunkFault(40, 200000)
            ops.trunk_dataflow(x, convs)
            lib.isrDebugSetTrunkFault(-1, 0)
            torch.cuda.synchronize()
            assert int(st["buf"][ops._TRUNK_ERROR_SLOT].item()) >= 2
            st["buf"][ops._TRUNK_ERROR_SLOT] = 0
            again = ops.trunk_dataflow(x, convs)
            torch.cuda.synchronize()
            assert torch.equal(good, again)
            ops.trunk_check()
    finally:
        lib.isrDebugSetTrunkFault(-1, 0)
        torch.cuda.synchronize()
        st["buf"][ops._TRUNK_ERROR_SLOT] = 0


def test_multi_round_launch_outlives_its_timeout_because_the_deadline_is_per_wait(diag_lib):
    """a healthy multi-round launch must not fail because it LASTS longer than the limit.  1 020 tiles = 4 rounds
    (~2.5 ms); the limit is shortened to 0.2 ms per round (0.8 ms per wait): counted from the kernel's start -- the earlier form --
    every wait after 0.2 ms would have given up; per wait nothing comes near it.  Result bit-identical, error word clean."""
    from isosurfacesuperresolution_amd import ops
    lib = ops._sr()
    g = torch.Generator().manual_seed(9)
    convs = [(((torch.rand(64, 24 if k == 0 else 64, 3, 3, generator=g) - 0.5) * 0.15).cuda(), ((torch.rand(64, generator=g) - 0.5) * 0.2).cuda())
             for k in range(21)]
    x = ((torch.rand(1, 24, 540, 960, generator=g) - 0.4)).cuda()
    st = ops._range_state(x.device)
    try:
        with torch.no_grad():
            good = ops.trunk_dataflow(x, convs)
            torch.cuda.synchronize()
            ops.trunk_check()
            lib.isrDebugSetTrunkFault(-1, 20000)                # no faulty tile, 0.2 ms x rounds
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            short = ops.trunk_dataflow(x, convs)
            t1.record()
            lib.isrDebugSetTrunkFault(-1, 0)
            torch.cuda.synchronize()
            assert t0.elapsed_time(t1) > 0.4                     # the launch did last longer than the 0.2 ms the old form allowed
            assert int(st["buf"][ops._TRUNK_ERROR_SLOT].item()) == 0
            assert torch.equal(short, good)
    finally:
        lib.isrDebugSetTrunkFault(-1, 0)
        torch.cuda.synchronize()
        st["buf"][ops._TRUNK_ERROR_SLOT] = 0
