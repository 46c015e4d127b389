"""The flow hole filling as ONE launch (csrc/sr_frame.hip: flow_fill_one_kernel, ``isrFlowFillOne``) against the three-launch form
(``isrFlowFillEx``) -- bit for bit -- and against the module path (inference/flowfill.py, the restatement of the reference's
``cv.inpaint`` replacement, inference/loadedmodel.py:77-82)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _gbuffer(h, w, seed, density):
    g = torch.Generator().manual_seed(seed)
    gb = torch.rand((h, w, 12), generator=g) * 2 - 1
    if density == "blobs":                       # an object in front of a background: connected holes, as a rendered frame has
        yy, xx = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
        m = torch.zeros(h, w)
        for k in range(4):
            cy, cx = float(torch.rand((), generator=g)) * h, float(torch.rand((), generator=g)) * w
            r = (0.08 + 0.25 * float(torch.rand((), generator=g))) * max(h, w)
            m = torch.maximum(m, (((yy - cy) ** 2 + (xx - cx) ** 2) < r * r).float())
    elif density == "none":
        m = torch.zeros(h, w)
    elif density == "all":
        m = torch.ones(h, w)
    elif density == "one":
        m = torch.zeros(h, w); m[h // 3, w // 2] = 1.0
    else:
        m = (torch.rand((h, w), generator=g) < float(density)).float()
    gb[..., 3] = m
    return gb.cuda()


SIZES = [(270, 480), (540, 960), (37, 53), (64, 64), (65, 129), (16, 480), (300, 7), (1, 1), (2, 3), (33, 31), (128, 1024)]


@pytest.mark.parametrize("h,w", SIZES)
@pytest.mark.parametrize("density", ["blobs", 0.02, 0.5, "one", "none", "all"])
def test_one_launch_equals_three_launches(h, w, density):
    from isosurfacesuperresolution_amd import ops
    assert ops._sr().isrFlowFillOneSupported(h, w) == (0 if (h, w) == (1, 1) else 1)
    gb = _gbuffer(h, w, 7 * h + w, density)
    three = ops.fill_flow_gbuffer(gb, one_launch=False)
    one = ops.fill_flow_gbuffer(gb, one_launch=True)
    torch.cuda.synchronize()
    assert torch.equal(one, three)
    if (h, w) == (1, 1):                             # its own coarsest level: flow * valid (inference/flowfill.py)
        assert torch.equal(one.flatten().cpu(), (gb[0, 0, 8:10] * gb[0, 0, 3]).cpu())
    if density not in ("none",):
        known = gb[..., 3] != 0
        assert torch.equal(one[0, 0][known], gb[..., 8][known]) and torch.equal(one[0, 1][known], gb[..., 9][known])


def test_one_launch_matches_the_module_path():
    from isosurfacesuperresolution_amd import ops
    from isosurfacesuperresolution_amd.inference.flowfill import fill_flow
    gb = _gbuffer(270, 480, 3, "blobs")
    one = ops.fill_flow_gbuffer(gb, one_launch=True)
    low = gb.permute(2, 0, 1).unsqueeze(0)
    ref = fill_flow(low[:, 8:10].cpu(), (low[:, 3:4] != 0).cpu())
    assert (one.cpu() - ref).abs().max().item() <= 1e-5


def test_launch_after_launch_on_one_workspace_and_beside_a_busy_stream():
    """The tickets only ever grow: forty launches back to back on one workspace, inputs alternating, every result checked; then the
    same on a side stream while convolutions keep every CU busy (workgroups of the fill get their slots one by one)."""
    from isosurfacesuperresolution_amd import ops
    a, b = _gbuffer(270, 480, 11, "blobs"), _gbuffer(270, 480, 12, 0.1)
    ra, rb = ops.fill_flow_gbuffer(a, one_launch=False).clone(), ops.fill_flow_gbuffer(b, one_launch=False).clone()
    outs = []
    for k in range(40):
        outs.append(ops.fill_flow_gbuffer(a if k % 2 == 0 else b, one_launch=True).clone())
    torch.cuda.synchronize()
    for k, o in enumerate(outs):
        assert torch.equal(o, ra if k % 2 == 0 else rb), "launch %d" % k
    side = torch.cuda.Stream()
    x = torch.rand(1, 64, 540, 960, device="cuda")
    wgt, bias = torch.rand(64, 64, 3, 3, device="cuda") * 0.05, torch.zeros(64, device="cuda")
    torch.cuda.synchronize()
    outs = []
    for k in range(12):
        y = ops.conv3x3(x, wgt, bias, act='relu')
        y = ops.conv3x3(y, wgt, bias, act='relu')
        outs.append(ops.fill_flow_gbuffer(a if k % 2 == 0 else b, stream=side, one_launch=True, out=torch.empty((1, 2, 270, 480), device="cuda")))
    torch.cuda.synchronize()
    for k, o in enumerate(outs):
        assert torch.equal(o, ra if k % 2 == 0 else rb), "launch %d beside the convolutions" % k
    assert int(ops._range_state(a.device)["buf"][ops._FILL_ERROR_SLOT].item()) == 0


def test_images_of_more_than_256_tiles_take_the_three_launch_form():
    from isosurfacesuperresolution_amd import ops
    assert ops._sr().isrFlowFillOneSupported(1100, 1100) == 0
    gb = _gbuffer(1100, 1100, 5, "blobs")
    assert torch.equal(ops.fill_flow_gbuffer(gb), ops.fill_flow_gbuffer(gb, one_launch=False))


def test_a_fill_that_times_out_reports_through_the_guard_word_and_the_three_launch_form_takes_over(diag_lib):
    """isrDebugSetFlowFillFault: the workgroup that finishes the top of the pyramid never raises the flag, the deadline is 2 ms -- every
    workgroup gives up (no hang), the KERNEL sets the guard word, the next frame's poll raises and switches the one-launch form off."""
    from isosurfacesuperresolution_amd import ops
    lib = ops._sr()
    gb = _gbuffer(270, 480, 21, "blobs")
    good = ops.fill_flow_gbuffer(gb, one_launch=False).clone()
    dev = gb.device
    try:
        assert torch.equal(ops.fill_flow_gbuffer(gb, one_launch=True), good)
        ops.guards_publish(dev); torch.cuda.synchronize(); ops.guards_poll(dev)            # clean so far
        lib.isrDebugSetFlowFillFault(1, 200000)
        assert ops.debug_switches() != 0
        ops.fill_flow_gbuffer(gb, one_launch=True)
        lib.isrDebugSetFlowFillFault(0, 0)
        ops.guards_publish(dev)
        torch.cuda.synchronize()
        assert int(ops._range_state(dev)["buf"][ops._FILL_ERROR_SLOT].item()) == 1
        with pytest.raises(RuntimeError, match="flow_fill_one_kernel"):
            ops.guards_poll(dev)
        assert ops.FLOW_FILL_ONE is False
        assert torch.equal(ops.fill_flow_gbuffer(gb), good)                               # the default is the three-launch form now
        ops.FLOW_FILL_ONE = True
        assert torch.equal(ops.fill_flow_gbuffer(gb), good)                               # switched on again: fresh workspace, right again
    finally:
        lib.isrDebugSetFlowFillFault(0, 0)
        ops.FLOW_FILL_ONE = True
        torch.cuda.synchronize()
        ops._range_state(dev)["buf"][ops._FILL_ERROR_SLOT] = 0


def test_the_flag_epoch_wraps_and_the_ticket_returns_to_zero():
    """the hand-off words must not grow without bound.  The ticket counts 0 .. ntiles - 1 inside ONE launch and is zero
    between launches (whatever ntiles is: 40 tiles here, not a power of two); the flag counts completed launches and the wait
    compares it wrap-safely -- started two launches before 2^32 the fill stays right across the wrap."""
    from isosurfacesuperresolution_amd import ops
    gb = _gbuffer(270, 480, 11, "blobs")
    good = ops.fill_flow_gbuffer(gb, one_launch=False).clone()
    assert torch.equal(ops.fill_flow_gbuffer(gb, one_launch=True), good)
    torch.cuda.synchronize()
    ws = ops._fill_ws[(gb.device, 270, 480, torch.cuda.current_stream().cuda_stream)].view(torch.int32)
    sync = ws.numel() - 16                                      # [0] ticket, [1] flag (csrc/sr_frame.hip: isrFlowFillOne)
    assert int(ws[sync].item()) == 0 and int(ws[sync + 1].item()) >= 1
    ws[sync + 1] = -2                                           # 0xFFFFFFFE completed launches
    for k in range(5):
        assert torch.equal(ops.fill_flow_gbuffer(gb, one_launch=True), good), k
        torch.cuda.synchronize()
        assert int(ws[sync].item()) == 0
    assert int(ws[sync + 1].item()) == 3                        # ... FFFFFFFF, 0, 1, 2, 3
    ops.guards_publish(gb.device); torch.cuda.synchronize(); ops.guards_poll(gb.device)     # no launch gave up


@pytest.mark.parametrize("h,w", [(270, 480), (135, 240), (37, 53), (65, 129), (300, 7), (2, 3)])
@pytest.mark.parametrize("density", ["blobs", 0.02, 0.5])
def test_fill_is_bit_identical_to_the_module_definition(h, w, density):
    """inference/flowfill.py DEFINES the fill in elementwise operations (one rounding each); both kernel forms compute its bits --
    against the definition evaluated on the CPU and by torch's own device kernels.  (The filled flow positions the warp of the
    previous frame: 1e-9 of flow is 1e-6 pixels at 1080p, and the recurrence amplifies what reaches the network.)"""
    from isosurfacesuperresolution_amd import ops
    from isosurfacesuperresolution_amd.inference.flowfill import fill_flow
    gb = _gbuffer(h, w, 13 * h + w, density)
    gb[..., 8:10] *= 0.03
    low = gb.permute(2, 0, 1).unsqueeze(0)
    ref_cpu = fill_flow(low[:, 8:10].cpu(), (low[:, 3:4] != 0).cpu())
    ref_dev = fill_flow(low[:, 8:10], low[:, 3:4] != 0)
    assert torch.equal(ref_dev.cpu(), ref_cpu)
    for one in (True, False):
        if one and not ops._sr().isrFlowFillOneSupported(h, w):
            continue
        got = ops.fill_flow_gbuffer(gb, one_launch=one)
        torch.cuda.synchronize()
        assert torch.equal(got.cpu(), ref_cpu), (one, (got.cpu() - ref_cpu).abs().max().item())
