"""CPU tests of the host-side pieces: C-ABI exports, .vbx I/O, parameter protocol (no GPU calls),
flow fill, loss, training step and the 2-rank (gloo) data-parallel form."""
import argparse
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from isosurfacesuperresolution_amd import _native, inference, losses, models, train, vbx, volumes as V

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def libs():
    _native.build()
    return ctypes.CDLL(_native.RENDERER_LIB), ctypes.CDLL(_native.SR_LIB)


def _declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return re.findall(r"^\s*(?:int|float|void|long long)\s+(\w+)\s*\(", text, flags=re.M)


def test_c_abi_exports_every_declared_symbol(libs):
    rend, sr = libs
    names_r = _declared("gpu_renderer_direct.h")
    names_s = _declared("isr_sr_kernels.h")
    assert {"initGVDB", "loadGrid", "setParameter", "render"} <= set(names_r)
    assert {"isrConv3x3Forward", "isrConv3x3WeightGrad", "isrConvPrepareWeights"} <= set(names_s)
    for n in names_r:
        assert hasattr(rend, n), n
    for n in names_s:
        assert hasattr(sr, n), n


def test_product_library_has_no_diagnostics_and_the_diagnostics_build_has_the_same_abi(libs):
    """the library a deployment ships (lib/libisr_sr.so) exports nothing named isrDebug* -- checked on the dynamic
    symbol table, not through ctypes -- and the diagnostics build of the same sources (lib/libisr_sr_diag.so) exports every entry point
    of the public header plus the switches; ``ops`` tells the two apart."""
    import subprocess
    from isosurfacesuperresolution_amd import _native, ops
    product = os.path.join(_native.LIBDIR, "libisr_sr.so")
    assert os.path.exists(product) and os.path.exists(_native.SR_DIAG_LIB)

    def exported(path):
        out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
        return {l.split()[-1] for l in out.splitlines() if " T " in l}
    prod, diag = exported(product), exported(_native.SR_DIAG_LIB)
    assert not [n for n in prod if n.startswith("isrDebug")], sorted(n for n in prod if n.startswith("isrDebug"))
    switches = {n for n in diag if n.startswith("isrDebug")}
    assert {"isrDebugSetTrunkFault", "isrDebugSetFlowFillFault", "isrDebugSetSplitUpsForm", "isrDebugSplitState"} <= switches
    assert {n for n in prod if n.startswith("isr")} == {n for n in diag if n.startswith("isr")} - switches      # the same ABI otherwise
    for n in _declared("isr_sr_kernels.h"):
        assert n in prod and n in diag, n
    assert not hasattr(ops._bind(_native.load(product)), "isrDebugSplitState")
    assert hasattr(ops._bind(_native.load(_native.SR_DIAG_LIB)), "isrDebugSplitState")


def test_no_sixteen_byte_store_with_an_sgpr_soffset_in_the_built_libraries(libs, tmp_path):
    """Round 6 (tools/probes/store_valu_overwrite_probe.hip): a 16-byte buffer store reads its data registers during the issue slots behind
    it; the compiler pads those slots only when the store's soffset is NOT an SGPR, and with an SGPR soffset gfx950 still needs one wait
    state -- a vector write into a data register there replaces lanes 12-15 of every 16 in memory.  The sources pass soffset = 0 to every
    16-byte store; this test reads the device code of the built libraries and fails if a > 8-byte store with a register soffset is back."""
    import re
    import shutil
    import subprocess
    from isosurfacesuperresolution_amd import _native
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("no llvm-objdump in this image")
    wide = re.compile(r"buffer_store_dwordx[34]\s")
    register_soffset = re.compile(r"s\[\d+:\d+\],\s*(s\d+|m0|ttmp\d+|vcc_lo|vcc_hi)\b")
    seen = 0
    for lib in (os.path.join(_native.LIBDIR, "libGPURendererDirect.so"), os.path.join(_native.LIBDIR, "libisr_sr.so"), _native.SR_DIAG_LIB):
        work = tmp_path / os.path.basename(lib)
        work.mkdir()
        shutil.copy(lib, work)                      # (--offloading writes the bundles next to the file it reads)
        subprocess.run([objdump, "--offloading", os.path.basename(lib)], cwd=work, capture_output=True, check=True)
        objects = [f for f in os.listdir(work) if f.endswith("gfx950")]
        assert objects, "no gfx950 code object in %s" % lib
        for obj in objects:
            text = subprocess.run([objdump, "-d", obj], cwd=work, capture_output=True, text=True, check=True).stdout
            for line in text.splitlines():
                if wide.search(line):
                    seen += 1
                    assert not register_soffset.search(line), "%s: %s" % (os.path.basename(lib), line.strip())
    assert seen > 50                                # (the epilogues' wide stores were found and looked at)


def test_padding_helpers(libs):
    _, sr = libs
    assert sr.isrConvCinPad(101) == 112 and sr.isrConvCinPad(64) == 64 and sr.isrConvCinPad(6) == 16
    assert sr.isrConvCoutPad(6) == 32 and sr.isrConvCoutPad(64) == 64 and sr.isrConvCoutPad(101) == 128


def test_set_parameter_protocol_without_gpu(libs):
    rend, _ = libs
    rend.setParameter.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
    ok = [(b"cameraOrigin", b"1.000,0.500,-2.000"), (b"cameraLookAt", b"0,0,0"), (b"cameraUp", b"0,1,0"),
          (b"cameraFoV", b"30.000"), (b"fov", b"45"), (b"resolution", b"480,270"), (b"isovalue", b"0.340"),
          (b"unshaded", b"1"), (b"aosamples", b"0"), (b"aoradius", b"0.010"), (b"viewport", b"0,0,480,270"),
          (b"light", b"camera"), (b"light", b"0.1,0.2,1"), (b"exponent", b"16")]
    for k, v in ok:
        assert rend.setParameter(k, v) == 0, (k, v)
    for k, v in [(b"bogus", b"1"), (b"cameraOrigin", b"1,2"), (b"cameraOrigin", b"1,2,3,4"), (b"resolution", b"x,y"),
                 (b"viewport", b"1,2,3")]:
        assert rend.setParameter(k, v) == -1, (k, v)
    rend.loadGrid.argtypes = [ctypes.c_char_p]
    assert rend.loadGrid(b"/tmp/volume.vdb") == -1          # must end in .vbx (GPURendererDirect.cpp:255-259)


def test_vbx_round_trip(libs, tmp_path):
    rend, _ = libs
    vol = V.ejecta(64)
    vol[:8] = 0                                            # make the brick box not start at the origin
    path = str(tmp_path / "v.vbx")
    nb = vbx.write_vbx(path, vol)
    assert nb == int(vbx.dense_blocks_nonzero(vol, 8, 8, 8).sum())
    dims = (ctypes.c_int * 3)()
    assert rend.isoVbxInfo(path.encode(), dims) == 0
    out = np.zeros((dims[2], dims[1], dims[0]), np.float32)
    assert rend.isoVbxReadDense(path.encode(), out.ctypes.data_as(ctypes.c_void_p)) == 0
    nzv = np.argwhere(vol != 0)
    lo, hi = (nzv.min(0) // 8) * 8, (nzv.max(0) // 8 + 1) * 8
    assert np.array_equal(out, vol[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]])
    assert rend.isoVbxInfo(b"/nonexistent.vbx", dims) == -2
    (tmp_path / "bad.vbx").write_bytes(b"\x01\x0b" + b"\0" * 20)
    assert rend.isoVbxInfo(str(tmp_path / "bad.vbx").encode(), dims) == -2


def test_camera_helper():
    cam = inference.Camera(480, 270, [0, 1, -1.7])
    o = cam.getOrigin()
    assert np.allclose(o, [0, 1, -1.7], atol=1e-12)
    cam.orientation = inference.Orientation.Zp
    assert cam.getUp() == [0, 0, 1]
    cam.startMove(); cam.move(1000, 1000)
    assert abs(cam.currentPitch) <= np.radians(80) + 1e-12
    cam.zoom(2)
    assert abs(cam.currentDistance - cam.baseDistance * 1.21) < 1e-9
    assert inference.Material(0.3).isovalue == 0.3


def test_flow_fill_keeps_known_and_fills_holes():
    torch.manual_seed(0)
    flow = torch.rand(1, 2, 27, 48) - 0.5
    valid = torch.zeros(1, 1, 27, 48, dtype=torch.bool)
    valid[:, :, 8:20, 10:30] = True
    out = inference.fill_flow(flow, valid)
    assert torch.equal(out[valid.expand_as(out)], flow[valid.expand_as(flow)])
    inside = flow[:, :, 8:20, 10:30]
    assert out.min() >= inside.min() - 1e-6 and out.max() <= inside.max() + 1e-6     # convex combinations only
    const = inference.fill_flow(torch.full((1, 2, 9, 9), 0.25) * valid[:, :, :9, 6:15].float(), valid[:, :, :9, 6:15])
    assert torch.allclose(const, torch.full_like(const, 0.25), atol=1e-6)
    none = inference.fill_flow(flow, torch.zeros_like(valid))
    assert none.abs().max() == 0


OPT = argparse.Namespace(upsample='bilinear', reconType='residual', useBN=False, numResidualLayers=10,
                         losses="l1:mask:1,l1:ao:1,l1:normal:10,l1:depth:10,temp-l2:color:0.1",
                         lossAO=0.0, lossAmbient=0.1, lossDiffuse=0.9, lossSpecular=0.0)


def test_loss_known_answers():
    crit = losses.LossNetUnshaded('cpu', 5, 6, 16, 2, OPT)
    gt = torch.zeros(1, 6, 16, 16)
    gt[:, 0] = 1.0; gt[:, 3] = 1.0; gt[:, 4] = 0.5; gt[:, 5] = 1.0
    total, vals = crit(gt, gt.clone(), gt[:, :5], None, gt.clone())
    assert total.item() == 0 and all(v == 0 for v in vals.values())
    assert ('mse', 'color') in vals                       # always evaluated for PSNR (lossnet_unshaded.py:37-38)
    pred = gt.clone()
    pred[:, 0] = 0.0                                      # mask off by 1 inside the 12x12 un-padded window
    total, vals = crit(gt, pred, gt[:, :5], None, gt.clone())
    inner = 12 * 12 / (16 * 16)
    assert abs(vals[('l1', 'mask')] - inner) < 1e-6
    # colour: gt shades to ambient+diffuse = 1.0 inside, pred mask 0 -> lerp(bg=0, 1, 0.5) = 0.5
    assert abs(vals[('mse', 'color')] - 0.25 * inner) < 1e-6
    assert abs(total.item() - (inner + 0.1 * 0.0)) < 1e-5 or abs(total.item() - (inner + 0.1 * vals[('temp-l2', 'color')])) < 1e-5
    with pytest.raises(ValueError):
        losses.LossNetUnshaded('cpu', 5, 6, 16, 2, argparse.Namespace(**{**vars(OPT), "losses": "l1:bogus:1"}))
    with pytest.raises(NotImplementedError):
        losses.LossNetUnshaded('cpu', 5, 6, 16, 2, argparse.Namespace(**{**vars(OPT), "losses": "perceptual:color:1"}))


def _clip_batch(B=2, T=3, h=8, seed=5):
    g = torch.Generator().manual_seed(seed)
    inp = torch.rand(B, T, 5, h, h, generator=g)
    inp[:, :, 0] = inp[:, :, 0] * 2 - 1
    flow = (torch.rand(B, T, 2, h, h, generator=g) - 0.5) * 0.05
    tgt = torch.rand(B, T, 6, 4 * h, 4 * h, generator=g)
    tgt[:, :, 0] = tgt[:, :, 0] * 2 - 1
    return inp, flow, tgt


def test_train_step_decreases_loss_and_backprops_through_time():
    torch.manual_seed(124)
    net = models.createNetwork('EnhanceNet', 4, 101, [0, 1, 2, 3, 4], 6, OPT)
    crit = losses.LossNetUnshaded('cpu', 5, 6, 32, 4, OPT)
    opt, sched = train.make_optimizer(net, lr=1e-4)
    batch = _clip_batch()
    l0 = train.train_step(net, crit, opt, batch, initial_image="zero")
    w_before = net.preblock[0].weight.detach().clone()
    for _ in range(3):
        l1 = train.train_step(net, crit, opt, batch, initial_image="zero")
    assert l1 < l0
    assert not torch.equal(w_before, net.preblock[0].weight)
    # recurrence is NOT detached: frame 0's prediction influences frame 1's loss
    net.zero_grad()
    loss, _ = train.clip_loss(net, crit, *batch, initial_image="zero")
    g_full = torch.autograd.grad(loss, net.postblock[8].weight)[0]
    assert torch.isfinite(g_full).all() and g_full.abs().sum() > 0


def test_two_rank_gloo_data_parallel_matches_single_process(tmp_path):
    script = tmp_path / "ddp.py"
    script.write_text('''
import argparse, os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from isosurfacesuperresolution_amd import models, losses, train
sys.path.insert(0, os.path.join(%r, "tests"))
from test_host_cpu import OPT, _clip_batch
dist.init_process_group("gloo")
rank = dist.get_rank()
torch.manual_seed(1000 + rank)            # different init per rank: the trainer must broadcast rank 0's
net = models.createNetwork('EnhanceNet', 4, 101, [0, 1, 2, 3, 4], 6, OPT)
crit = losses.LossNetUnshaded('cpu', 5, 6, 32, 4, OPT)
opt = torch.optim.SGD(net.parameters(), lr=0.05)   # plain SGD: Adam's first step is sign(g)*lr, which
tr = train.DataParallelTrainer(net, crit, opt)      # amplifies 1e-9 reduction-order noise on near-zero gradients
init = {k: v.clone() for k, v in net.state_dict().items()}
batch = _clip_batch(B=2, T=2, h=8, seed=9)
tr.step(tr.shard(batch), initial_image="zero")
if rank == 0:
    torch.save({"init": init, "final": {k: v.clone() for k, v in net.state_dict().items()}}, %r)
sd = [None, None]
flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
gathered = [torch.zeros_like(flat) for _ in range(2)]
dist.all_gather(gathered, flat)
assert torch.equal(gathered[0], gathered[1]), "ranks diverged"
dist.destroy_process_group()
''' % (ROOT, ROOT, str(tmp_path / "ddp_out.pt")))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                           "--master-addr", "127.0.0.1", "--master-port", "29613", str(script)],
                          env=env, stdout=subprocess.DEVNULL, timeout=300)
    ddp = torch.load(tmp_path / "ddp_out.pt")
    # single process, global batch of 2, starting from rank 0's (broadcast) weights
    net = models.createNetwork('EnhanceNet', 4, 101, [0, 1, 2, 3, 4], 6, OPT)
    net.load_state_dict(ddp["init"])
    crit = losses.LossNetUnshaded('cpu', 5, 6, 32, 4, OPT)
    opt = torch.optim.SGD(net.parameters(), lr=0.05)
    train.train_step(net, crit, opt, _clip_batch(B=2, T=2, h=8, seed=9), initial_image="zero")
    # compare the two parameter updates as vectors (individual ReLU pre-activations that sit within
    # rounding of zero may switch differently for batch 1 vs batch 2 kernels, cf. test_conv_gpu)
    upd_single = torch.cat([(v - ddp["init"][k]).reshape(-1) for k, v in net.state_dict().items()])
    upd_ddp = torch.cat([(ddp["final"][k] - ddp["init"][k]).reshape(-1) for k in net.state_dict()])
    assert upd_single.abs().max() > 1e-4
    rel = (upd_single - upd_ddp).norm() / upd_single.norm()
    assert rel < 1e-2, rel


def test_tiled_render_two_rank_gloo_composite(tmp_path, oracle):
    """Object-space split over 2 ranks: local render of each tile (oracle, CPU), ONE all-gather over
    gloo, nearest-hit composite == the single-volume render."""
    script = tmp_path / "tiled.py"
    script.write_text('''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r)
from oracle import iso_oracle as O
from isosurfacesuperresolution_amd import volumes as V, parallel_render as PR
dist.init_process_group("gloo")
rank = dist.get_rank()
vol = V.ejecta(64)
gmin, gmax, gmaxval = PR.global_stats(vol)
box = PR.tile_boxes(vol.shape, (2, 1, 1))[rank]
tile = PR.make_tile(lambda z0, z1, y0, y1, x0, x1: vol[z0:z1, y0:y1, x0:x1], vol.shape, box, gmin, gmax, gmaxval)
tv = O.OracleVolume(tile["data"], tile=tile)
p = O.make_params(80, 48, origin=V.quantize3(V.orbit_camera(21)), fov=30.0, isovalue=0.34)
def local_render(t):
    img, _ = O.render(tv, p, threads=2)
    t.copy_(torch.from_numpy(img))
tr = PR.TiledRenderer(None, tile, render_fn=local_render)
out = tr.render(80, 48, device="cpu")
if rank == 0:
    np.save(%r, out.numpy())
dist.destroy_process_group()
''' % (ROOT, str(tmp_path / "tiled.npy")))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                           "--master-addr", "127.0.0.1", "--master-port", "29617", str(script)],
                          env=env, stdout=subprocess.DEVNULL, timeout=300)
    comp = np.load(tmp_path / "tiled.npy")
    vol = V.ejecta(64)
    p = oracle.make_params(80, 48, origin=V.quantize3(V.orbit_camera(21)), fov=30.0, isovalue=0.34)
    full, _ = oracle.render(oracle.OracleVolume(vol), p)
    assert full[..., 3].sum() > 100
    assert np.array_equal(comp, full)      # tiles walk the global ray: the composite IS the unsplit render, all 12 channels


def _strip_sequence(seed=5, frames=3, h=70, w=40):
    """A short synthetic low-resolution G-buffer sequence [frames][h][w][12] with holes in mask and flow."""
    g = torch.Generator().manual_seed(seed)
    seq = []
    for _ in range(frames):
        t = torch.rand(h, w, 12, generator=g)
        t[..., 3] = (t[..., 3] > 0.35).float()
        t[..., 4:7] = t[..., 4:7] * 2 - 1
        t[..., 8:10] = (t[..., 8:10] - 0.5) * 0.08
        seq.append(t)
    return seq


def test_strip_super_resolution_three_rank_gloo_matches_single_process(tmp_path):
    """One frame's SR split over 3 ranks by screen strips with a 24-px halo + ONE all-gather per frame
     == the single-process temporal sequence (uneven strips: 70 rows over 3 ranks)."""
    script = tmp_path / "strips.py"
    script.write_text('''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
sys.path.insert(0, os.path.join(%r, "tests"))
from isosurfacesuperresolution_amd import models, parallel_sr
from isosurfacesuperresolution_amd.inference import LoadedModel
from isosurfacesuperresolution_amd.pipeline import default_shading
from test_host_cpu import OPT, _strip_sequence
dist.init_process_group("gloo")
torch.manual_seed(77)
net = models.createNetwork('EnhanceNet', 4, 101, [0, 1, 2, 3, 4], 6, OPT).eval()
lm = LoadedModel.from_model(net, "cpu", parameters={"initialImage": "zero"})
sr = parallel_sr.StripSuperResolution(lm, default_shading("cpu", 30.0))
outs = [sr.frame(g) for g in _strip_sequence()]
if dist.get_rank() == 2:                       # any rank holds the full frames
    torch.save([(rgb, raw) for rgb, raw in outs], %r)
dist.destroy_process_group()
''' % (ROOT, ROOT, str(tmp_path / "strips.pt")))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3",
                           "--master-addr", "127.0.0.1", "--master-port", "29621", str(script)],
                          env=env, stdout=subprocess.DEVNULL, timeout=300)
    from isosurfacesuperresolution_amd import parallel_sr
    from isosurfacesuperresolution_amd.pipeline import default_shading
    split = torch.load(tmp_path / "strips.pt")
    torch.manual_seed(77)
    net = models.createNetwork('EnhanceNet', 4, 101, [0, 1, 2, 3, 4], 6, OPT).eval()
    lm = inference.LoadedModel.from_model(net, "cpu", parameters={"initialImage": "zero"})
    single = parallel_sr.StripSuperResolution(lm, default_shading("cpu", 30.0))     # world == 1: the whole frame
    for g, (rgb_s, raw_s) in zip(_strip_sequence(), split):
        rgb, raw = single.frame(g)
        assert raw.shape == raw_s.shape == (1, 6, 280, 160)
        assert (raw - raw_s).abs().max().item() <= 1e-5
        assert (rgb - rgb_s).abs().max().item() <= 1e-5
    # and the single-rank strip path is the reference module path (LoadedModel.inference + clamp/normalise)
    ref_prev = None
    single.reset()
    for g in _strip_sequence():
        low = g.permute(2, 0, 1).unsqueeze(0)
        ref = lm.inference(low, ref_prev)
        ref = torch.cat([ref[:, 0:1].clamp(-1, 1), inference_normalize(ref[:, 1:4]), ref[:, 4:].clamp(0, 1)], dim=1)
        _, raw = single.frame(g)
        assert (raw - ref).abs().max().item() <= 1e-5
        ref_prev = ref


def test_tile_bounds_and_best_grid():
    from isosurfacesuperresolution_amd import parallel_sr as P
    # tiles partition the image, column cuts on multiples of 8
    for (h, w, grid) in ((540, 960, (4, 2)), (540, 960, (2, 4)), (70, 40, (2, 2)), (135, 240, (8, 1)), (271, 483, (3, 2))):
        cover = torch.zeros(h, w, dtype=torch.int32)
        for r in range(grid[0] * grid[1]):
            y0, y1, x0, x1 = P.tile_bounds(h, w, grid, r)
            assert x0 % 8 == 0 and (x1 % 8 == 0 or x1 == w)
            cover[y0:y1, x0:x1] += 1
        assert bool((cover == 1).all())
    assert P.tile_bounds(540, 960, (8, 1), 3)[:2] == P.strip_bounds(540, 8, 3)
    # what the slowest rank computes: strips 1.71x its share at 8 ranks, the best grid 1.31x
    share = 540 * 960 / 8
    assert abs(P.extended_area(540, 960, (8, 1)) / share - 1.72) < 0.02
    assert P.best_grid(8, 540, 960) == (2, 4) and abs(P.extended_area(540, 960, (2, 4)) / share - 1.31) < 0.02
    assert abs(P.extended_area(540, 960, (4, 2)) / share - 1.42) < 0.02
    assert P.best_grid(2, 540, 960) in ((2, 1), (1, 2)) and P.best_grid(1, 540, 960) == (1, 1)


def test_tile_super_resolution_two_by_two_gloo_matches_single_process(tmp_path):
    """The same frame sequence split over a 2 x 2 grid of screen tiles (halo in y AND x, ONE all-gather of padded rectangles per
    frame) == the single-process temporal sequence."""
    script = tmp_path / "tiles.py"
    script.write_text('''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
sys.path.insert(0, os.path.join(%r, "tests"))
from isosurfacesuperresolution_amd import models, parallel_sr
from isosurfacesuperresolution_amd.inference import LoadedModel
from isosurfacesuperresolution_amd.pipeline import default_shading
from test_host_cpu import OPT, _strip_sequence
dist.init_process_group("gloo")
torch.manual_seed(77)
net = models.createNetwork('EnhanceNet', 4, 101, [0, 1, 2, 3, 4], 6, OPT).eval()
lm = LoadedModel.from_model(net, "cpu", parameters={"initialImage": "zero"})
sr = parallel_sr.StripSuperResolution(lm, default_shading("cpu", 30.0), grid=(2, 2))
outs = [sr.frame(g) for g in _strip_sequence(frames=2, h=70, w=96)]
if dist.get_rank() == 3:
    torch.save([(rgb, raw) for rgb, raw in outs], %r)
dist.destroy_process_group()
''' % (ROOT, ROOT, str(tmp_path / "tiles.pt")))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=4",
                           "--master-addr", "127.0.0.1", "--master-port", "29623", str(script)],
                          env=env, stdout=subprocess.DEVNULL, timeout=300)
    from isosurfacesuperresolution_amd import parallel_sr
    from isosurfacesuperresolution_amd.pipeline import default_shading
    split = torch.load(tmp_path / "tiles.pt")
    torch.manual_seed(77)
    net = models.createNetwork('EnhanceNet', 4, 101, [0, 1, 2, 3, 4], 6, OPT).eval()
    lm = inference.LoadedModel.from_model(net, "cpu", parameters={"initialImage": "zero"})
    single = parallel_sr.StripSuperResolution(lm, default_shading("cpu", 30.0))
    for g, (rgb_s, raw_s) in zip(_strip_sequence(frames=2, h=70, w=96), split):
        rgb, raw = single.frame(g)
        assert raw.shape == raw_s.shape == (1, 6, 280, 384)
        assert (raw - raw_s).abs().max().item() <= 1e-5
        assert (rgb - rgb_s).abs().max().item() <= 1e-5


def inference_normalize(v):
    from isosurfacesuperresolution_amd.utils import ScreenSpaceShading
    return ScreenSpaceShading.normalize(v, dim=1)


def test_ssim_matches_reference_fixture():
    from isosurfacesuperresolution_amd import utils
    G = np.load(os.path.join(ROOT, "tests", "golden", "sr_reference.npz"))
    a, b = torch.from_numpy(G["psnr_a"]), torch.from_numpy(G["psnr_b"])
    assert abs(utils.SSIM()(a, b).item() - float(G["ssim"])) < 1e-6
    assert abs(utils.SSIM()(a, a).item() - 1.0) < 1e-6
    big = torch.rand(1, 3, 176, 176)
    assert abs(utils.MSSSIM()(big, big).item() - 1.0) < 1e-5


def test_raw_import_and_vbx_conversion(tmp_path, libs):
    from isosurfacesuperresolution_amd import volume_io
    vol = V.ejecta(64)
    dat = str(tmp_path / "vol.dat")
    volume_io.export_raw(dat, vol, "USHORT")
    back = volume_io.import_raw(dat, 1, 0.02)
    assert back.shape == vol.shape and np.abs(back - vol).max() <= 0.02 + 1e-4     # quantisation + re-threshold
    half = volume_io.import_raw(dat, 2, 0.02)
    assert half.shape == (32, 32, 32)
    with open(str(tmp_path / "vol.raw"), "r+b") as f:           # a header in front of the payload is skipped
        payload = f.read()
    with open(str(tmp_path / "vol.raw"), "wb") as f:
        f.write(b"HDR!" * 4 + payload)
    assert np.array_equal(volume_io.import_raw(dat, 1, 0.02), back)
    nb = volume_io.convert_to_vbx(dat, str(tmp_path / "vol.vbx"))
    assert nb > 0
    with pytest.raises(ValueError):
        volume_io.import_raw(str(tmp_path / "vol.raw"))


def test_dataset_video_contract(tmp_path):
    from isosurfacesuperresolution_amd import dataset_video as D
    rng = np.random.default_rng(0)
    for i in range(3):
        low = rng.random((4, 5, 48, 64), dtype=np.float32)
        low[:, 0] = 1.0
        if i == 1:
            low[:, 0:3, :, :32] = 0.0                       # left half of clip 1 is empty -> crops must avoid it
        np.save(tmp_path / ("low_%05d.npy" % i), low)
        np.save(tmp_path / ("high_%05d.npy" % i), rng.random((4, 6, 192, 256), dtype=np.float32))
        np.save(tmp_path / ("flow_%05d.npy" % i), rng.random((4, 2, 48, 64), dtype=np.float32))
    dd = D.collect_samples(str(tmp_path), 40, seed=1)
    assert dd.num_frames == 4 and dd.input_channels == 5 and dd.output_channels == 6
    assert [s.index for s in dd.samples] == sorted(s.index for s in dd.samples)
    for s in dd.samples:
        if s.index == 1:
            cover = dd.images_low[1][0, 0:3, s.crop_low[0]:s.crop_low[1], s.crop_low[2]:s.crop_low[3]].sum(0) > 0
            assert cover.sum() >= 512
    train_set, test_set = D.DatasetFromSamples(dd, False, 0.2), D.DatasetFromSamples(dd, True, 0.2)
    assert len(train_set) == 32 and len(test_set) == 8
    low, flow, high = train_set[0]
    assert low.shape == (4, 5, 32, 32) and flow.shape == (4, 2, 32, 32) and high.shape == (4, 6, 128, 128)
    s0 = dd.samples[0]
    assert torch.equal(high, torch.from_numpy(dd.images_high[s0.index][:, :, 4 * s0.crop_low[0]:4 * s0.crop_low[1], 4 * s0.crop_low[2]:4 * s0.crop_low[3]]))
    l2, h2, f2 = D.data_augmentation(low.numpy(), high.numpy(), flow.numpy(), 1, enabled=True)
    assert np.array_equal(l2, low.numpy()[:, :, ::-1]) and np.array_equal(f2[:, 0], -flow.numpy()[:, 0, ::-1])
    batch = next(iter(torch.utils.data.DataLoader(train_set, batch_size=4)))
    assert batch[0].shape == (4, 4, 5, 32, 32)


def test_vbx_sparse_and_malformed_files(libs, tmp_path):
    """A genuinely sparse .vbx (two bricks 4000 voxels apart in every axis) is read as a brick LIST: a few hundred KB,
    not the 4008^3 box it spans (a dense copy of that box is refused, not attempted); files whose pool / atlas sizes
    exceed the file, or that are cut short, give -2 -- no exception crosses the C boundary."""
    import psutil
    rend, _ = libs
    b = np.full((8, 8, 8), 0.7, np.float32)
    path = str(tmp_path / "far.vbx")
    assert vbx.write_vbx_bricks(path, {(0, 0, 0): b, (4000, 4000, 4000): b}) == 2
    dims = (ctypes.c_int * 3)()
    rss = psutil.Process().memory_info().rss
    assert rend.isoVbxInfo(path.encode(), dims) == 0 and list(dims) == [4008, 4008, 4008]
    assert psutil.Process().memory_info().rss - rss < 64 << 20
    out = np.zeros(8, np.float32)
    assert rend.isoVbxReadDense(path.encode(), out.ctypes.data_as(ctypes.c_void_p)) == -2
    good = str(tmp_path / "good.vbx")
    vbx.write_vbx(good, V.ejecta(32))
    raw = bytearray(open(good, "rb").read())
    assert rend.isoVbxInfo(good.encode(), dims) == 0
    # header layout (vbx.py): 2 + 48 + 4 + 1 + 8 bytes, then the grid header: name 256, 3 bytes, voxelsize 12, then leafcnt ...
    topo = 2 + 48 + 4 + 1 + 8 + 256 + 3 + 12 + 4 + 12 + 4 + 4 + 8 + 1 + 4 + 1 + 12 + 12 + 4 + 8   # first level record
    for field, value in ((5, 0x7fffffff), (6, 0x7fffffff)):               # cnt0 / width0 of level 0
        bad = bytearray(raw)
        bad[topo + 4 * field: topo + 4 * field + 4] = int(value).to_bytes(4, "little")
        (tmp_path / "bad.vbx").write_bytes(bytes(bad))
        assert rend.isoVbxInfo(str(tmp_path / "bad.vbx").encode(), dims) == -2
    for cut in (len(raw) // 2, len(raw) - 4, topo + 8):
        (tmp_path / "cut.vbx").write_bytes(bytes(raw[:cut]))
        assert rend.isoVbxInfo(str(tmp_path / "cut.vbx").encode(), dims) == -2


class _OracleBackend:
    """A CPU stand-in for DirectRenderer behind inference.Renderer (tests only): same command strings, frames by the
    oracle into a caller-owned [H, W, 12] CPU tensor."""

    def __init__(self, oracle):
        self.o = oracle
        self.args = {"cameraOrigin": [0, 0, -1], "cameraLookAt": [0, 0, 0], "cameraUp": [0, 1, 0], "cameraFoV": [45.0],
                     "resolution": [512, 512], "isovalue": [0.0], "viewport": [0, 0, 512, 512], "aosamples": [32], "aoradius": [0.01]}
        self.material = {}
        self.last = None
        self.log = []

    def load_dense(self, vol):
        self.vol = self.o.OracleVolume(vol)

    def send_command(self, cmd, value):
        self.log.append((cmd, value))
        cmd = {"fov": "cameraFoV"}.get(cmd, cmd)
        if cmd in self.args:
            self.args[cmd] = [float(v) for v in value.split(",")]
        elif cmd in ("diffuse", "specular", "ambient"):
            self.material[cmd] = [float(v) for v in value.split(",")]
        elif cmd == "exponent":
            self.material["specular_exponent"] = int(value)
        elif cmd not in ("unshaded", "light"):
            return -1
        return 0

    def render_direct(self, tensor):
        a = self.args
        w, h = int(a["resolution"][0]), int(a["resolution"][1])
        p = self.o.make_params(w, h, origin=a["cameraOrigin"], lookat=a["cameraLookAt"], up=a["cameraUp"], fov=a["cameraFoV"][0],
                               isovalue=a["isovalue"][0], last_origin=self.last, viewport=[int(v) for v in a["viewport"]],
                               ao_samples=int(a["aosamples"][0]), ao_radius=a["aoradius"][0], **self.material)
        img, _ = self.o.render(self.vol, p, threads=2)
        tensor.copy_(torch.from_numpy(img))
        self.last = list(a["cameraOrigin"])
        return 0.125


def test_pipe_renderer_call_sequence(oracle):
    """``inference.Renderer`` (the reference's pipe flavour, renderer.py:16-76) driven exactly the way
    ``mainPSNR2_AllAngles.py:184-276`` drives it: construct with (exe, file, Material, Camera), ``aoradius=...\\\\n``, per
    sample the camera commands + ``aosamples=0\\\\n``, ``resolution=W,H\\\\n`` + ``render\\\\n`` + ``read_image(W, H)`` for
    the ground truth, then the same at W/4 x H/4 -- planar [12, H, W] frames, and the time of the last frame."""
    vol = V.sphere64()
    backend = _OracleBackend(oracle)
    RES = (96, 64)
    r = inference.Renderer("renderer.exe", vol, inference.Material(0.5), inference.Camera(RES[0], RES[1]), backend=backend, device="cpu")
    r.send_command("aoradius=%5.3f\n" % 0.01)
    images = []
    for org in ((0.0, 0.6, -1.9), (1.2, 0.4, 1.5)):
        r.send_command("aosamples=0\n")
        r.send_command("cameraOrigin=%5.3f,%5.3f,%5.3f\n" % org)
        r.send_command("cameraLookAt=%5.3f,%5.3f,%5.3f\n" % (0, 0, 0))
        r.send_command("cameraUp=%5.3f,%5.3f,%5.3f\n" % (0, 1, 0))
        r.send_command("resolution=%d,%d\n" % RES)
        r.send_command("render\n")
        gt = r.read_image(RES[0], RES[1])
        assert gt.shape == (12, RES[1], RES[0]) and gt.dtype == np.float32 and r.get_time() == 0.125
        r.send_command("resolution=%d,%d\n" % (RES[0] // 4, RES[1] // 4))
        r.render()
        low = r.read_image(RES[0] // 4, RES[1] // 4)
        assert low.shape == (12, RES[1] // 4, RES[0] // 4)
        images.append((gt, low))
    gt, low = images[1]
    assert gt[3].sum() > 300 and set(np.unique(gt[3])) <= {0.0, 1.0}
    assert np.allclose(np.linalg.norm(gt[4:7], axis=0)[gt[3] == 1], 1.0, atol=1e-5)
    assert (gt[10] == 1).all() and (gt[11] == 0).all()                      # AO off, no shadow
    assert np.abs(gt[8:10][:, gt[3] == 1]).max() > 0                        # flow vs the previous camera
    # the planar frame is the interleaved frame of the same commands
    p = oracle.make_params(RES[0], RES[1], origin=(1.2, 0.4, 1.5), fov=45.0, isovalue=0.5, last_origin=(0.0, 0.6, -1.9),
                           diffuse=[0.7, 0.2, 0.2], specular=[0.1, 0.1, 0.1], specular_exponent=16)
    ref, _ = oracle.render(oracle.OracleVolume(vol), p, threads=2)
    assert np.array_equal(gt, ref.transpose(2, 0, 1))
    # resolution resets the viewport (pipe mode), the material went through the command protocol
    assert ("viewport", "0,0,%d,%d" % (RES[0] // 4, RES[1] // 4)) in backend.log and ("exponent", "16") in backend.log
    # protocol errors: reading without a pending frame, wrong size, unknown command, use after close
    with pytest.raises(RuntimeError):
        r.read_image(*RES)
    r.render()
    with pytest.raises(RuntimeError):
        r.read_image(*RES)                                                  # the waiting frame is 24x16
    with pytest.raises(RuntimeError):
        r.send_command("bogus=1\n")
    r.close()
    with pytest.raises(RuntimeError):
        r.render()


def test_losses_surface_and_checkpoint_unpickler(tmp_path):
    """``losses.LossNet`` exists and says why it is out of scope; checkpoints are read through an allow-list unpickler:
    the reference-format fixture loads, a pickle that names an arbitrary callable is refused."""
    import zipfile
    assert {"LossBuilder", "LossNetUnshaded", "LossNet"} <= set(dir(losses))
    with pytest.raises(NotImplementedError):
        losses.LossNet("cpu", 5, 6, 128, 16, None)
    here = os.path.join(os.path.dirname(__file__), "golden")
    with zipfile.ZipFile(os.path.join(here, "ref_checkpoint.zip")) as z:
        z.extractall(tmp_path)
    lm = inference.LoadedModel(str(tmp_path / "model_epoch_12.pth"), "cpu", 4)
    assert type(lm.model).__module__ == "isosurfacesuperresolution_amd.models.enhancenet"

    class Evil:
        def __reduce__(self):
            return (os.system, ("true",))
    torch.save({"model": Evil()}, str(tmp_path / "evil.pth"))
    import pickle
    with pytest.raises(pickle.UnpicklingError):
        inference.LoadedModel(str(tmp_path / "evil.pth"), "cpu", 4)
    # names are matched exactly: nothing that is merely reachable from an allowed package resolves (attribute walks such
    # as torch + "os.system", re-exports, torch's own shell runner, loaders), and a protocol-4 pickle that names one fails
    from isosurfacesuperresolution_amd.inference.loadedmodel import _CheckpointPickle
    import io
    up = _CheckpointPickle.Unpickler(io.BytesIO(b""))
    for module, name in [("torch", "os.system"), ("torch", "os.getcwd"), ("torch.utils.collect_env", "run"), ("torch", "serialization.load"),
                         ("torch", "hub.load"), ("torch", "load"), ("torch._C", "_cuda_init"), ("torch.nn.modules.module", "warnings"),
                         ("torch.optim.adam", "Tensor"), ("builtins", "getattr"), ("builtins", "eval"), ("numpy", "load"),
                         ("isosurfacesuperresolution_amd.models.enhancenet", "torch")]:
        with pytest.raises(pickle.UnpicklingError):
            up.find_class(module, name)
    payload = b"\x80\x04\x95\x1b\x00\x00\x00\x00\x00\x00\x00\x8c\x05torch\x8c\tos.getcwd\x93)R."     # torch / 'os.getcwd', called
    with pytest.raises(pickle.UnpicklingError):
        _CheckpointPickle.Unpickler(io.BytesIO(payload)).load()


def test_fit_driver_epochs_checkpoint_restore_and_loadedmodel(tmp_path):
    """The driver around the step (mainVideoUnshaded.py:344-375,397-473,639-726,799-826): two epochs on dataset_video
    clips with StepLR stepped at the epoch's start, a test pass per epoch (per-term losses + PSNR), one checkpoint per
    epoch with the reference's keys; ``inference.LoadedModel`` loads the checkpoint and reproduces the trained
    network's output; ``restore`` continues from the newest checkpoint with the optimizer / scheduler state."""
    from isosurfacesuperresolution_amd import dataset_video as D
    rng = np.random.default_rng(5)
    clips = tmp_path / "clips"
    clips.mkdir()
    for i in range(2):
        low = rng.random((3, 5, 40, 40), dtype=np.float32)
        low[:, 0] = 1.0
        high = rng.random((3, 6, 160, 160), dtype=np.float32)
        np.save(clips / ("low_%05d.npy" % i), low)
        np.save(clips / ("high_%05d.npy" % i), high)
        np.save(clips / ("flow_%05d.npy" % i), ((rng.random((3, 2, 40, 40), dtype=np.float32) - 0.5) * 0.02).astype(np.float32))
    dd = D.collect_samples(str(clips), 5, seed=3)
    train_loader = torch.utils.data.DataLoader(D.DatasetFromSamples(dd, False, 0.2), batch_size=2, shuffle=False)
    test_loader = torch.utils.data.DataLoader(D.DatasetFromSamples(dd, True, 0.2), batch_size=2, shuffle=False)
    torch.manual_seed(124)
    net = models.createNetwork('EnhanceNet', 4, 101, [0, 1, 2, 3, 4], 6, OPT)
    crit = losses.LossNetUnshaded('cpu', 5, 6, 128, 16, OPT)
    params = dict(vars(OPT), initialImage="zero", upscale_factor=4)
    modeldir = str(tmp_path / "run00000")
    logs = []
    net, hist = train.fit(net, crit, train_loader, test_loader, modeldir, 2, params, device="cpu", lr=2e-4, lr_step=1, lr_gamma=0.5,
                          initial_image="zero", log=logs.append)
    assert [h['epoch'] for h in hist] == [1, 2]
    # StepLR(step 1, gamma .5) stepped at the START of every epoch: epoch 1 already trains at lr/2 (mainVideoUnshaded.py:399)
    assert hist[0]['lr'] == pytest.approx(1e-4) and hist[1]['lr'] == pytest.approx(5e-5)
    assert hist[1]['train_loss'] < hist[0]['train_loss']
    t = hist[1]['test']
    assert {'total_loss', 'psnr', "('mse', 'color')", "('l1', 'mask')", "('temp-l2', 'color')"} <= set(t)
    assert t['psnr'] == pytest.approx(10 * np.log10(1 / t["('mse', 'color')"]), abs=0.5)     # mean of logs vs log of mean: close for 3 frames
    assert os.path.basename(hist[1]['checkpoint']) == "model_epoch_2.pth" and os.path.exists(hist[0]['checkpoint'])
    ck = torch.load(hist[1]['checkpoint'], weights_only=False)
    assert set(ck) == {'epoch', 'model', 'parameters', 'optimizer', 'scheduler'} and ck['epoch'] == 3
    assert ck['parameters']['initialImage'] == "zero"
    # the viewer's loader takes it: same class, same output as the trained network
    lm = inference.LoadedModel(hist[1]['checkpoint'], "cpu", 4)
    assert lm.name == "model_epoch_2" and lm.initial_image_mode == "zero" and lm.input_channels == 101
    x = torch.rand(1, 101, 12, 10, generator=torch.Generator().manual_seed(9))
    with torch.no_grad():
        assert torch.equal(lm.model(x)[0], net.eval()(x)[0])
    # restore: newest checkpoint, optimizer + scheduler state taken from it; the loop starts AT the restored epoch number
    assert train.find_restore_epoch(modeldir) == 2
    net2, hist2 = train.fit(None, crit, train_loader, test_loader, modeldir, 3, params, device="cpu", restore=True,
                            initial_image="zero", log=logs.append)
    assert [h['epoch'] for h in hist2] == [2, 3]
    assert hist2[0]['lr'] == pytest.approx(2.5e-5)          # the restored scheduler had already seen two steps
    assert os.path.exists(train.checkpoint_path(modeldir, 3))
    assert any("Restore training" in str(l) for l in logs)
    with pytest.raises(FileNotFoundError):
        train.load_checkpoint(str(tmp_path / "nothing"))


def test_flat_adam_checkpoint_restore_continues_like_an_uninterrupted_run(tmp_path):
    """``fit(flat_adam=True)`` checkpoints pickle the optimizer OBJECT (mainVideoUnshaded.py:799-811).  FlatAdam's
    moments and step count live in ``optimizer.state`` under Adam's keys and its member parameters travel with the pickle, so
    a restored run takes exactly the steps the uninterrupted run takes; ``state_dict`` / ``load_state_dict`` carry them too."""
    g = torch.Generator().manual_seed(4)
    inp = torch.rand(1, 2, 5, 16, 16, generator=g); inp[:, :, 0] = inp[:, :, 0] * 2 - 1
    flow = (torch.rand(1, 2, 2, 16, 16, generator=g) - 0.5) * 0.05
    tgt = torch.rand(1, 2, 6, 64, 64, generator=g); tgt[:, :, 0] = tgt[:, :, 0] * 2 - 1
    loader = [(inp, flow, tgt)]
    opt2 = argparse.Namespace(**dict(vars(OPT), numResidualLayers=10))
    crit = losses.LossNetUnshaded('cpu', 5, 6, 64, 8, opt2)
    params = dict(vars(opt2), initialImage="zero", upscale_factor=4)

    def fresh():
        torch.manual_seed(124)
        return models.createNetwork('EnhanceNet', 4, 101, [0, 1, 2, 3, 4], 6, opt2)
    # uninterrupted: three epochs of one step each
    ref, _ = train.fit(fresh(), crit, loader, None, str(tmp_path / "a"), 3, params, device="cpu", lr=1e-3, flat_adam=True,
                       initial_image="zero", log=lambda *_: None)
    # interrupted after two epochs; the checkpoint of epoch 2 holds the model and the optimizer OBJECT
    run_b = str(tmp_path / "b")
    train.fit(fresh(), crit, loader, None, run_b, 2, params, device="cpu", lr=1e-3, flat_adam=True, initial_image="zero", log=lambda *_: None)
    ck, start = train.load_checkpoint(run_b)
    opt = ck['optimizer']
    assert isinstance(opt, train.FlatAdam) and start == 2
    opt.check_views()                                              # parameters AND gradients are views of the flat buffers again
    assert float(opt.steps.item()) == 2.0 and opt.exp_avg.abs().max().item() > 0 and opt.exp_avg_sq.abs().max().item() > 0
    assert all(p is q for p, q in zip(opt.members, ck['model'].parameters()))
    # one more step from the restored state == the third step of the uninterrupted run (same batch, same lr: lr_step 500)
    train.train_step(ck['model'], crit, opt, loader[0], initial_image="zero")
    for p, q in zip(ref.parameters(), ck['model'].parameters()):
        assert torch.allclose(p, q, rtol=0, atol=1e-7), (p - q).abs().max().item()
    # fit(restore=True) takes the same route (and would have raised AttributeError before)
    net_c, hist = train.fit(None, crit, loader, None, run_b, 3, params, device="cpu", restore=True, initial_image="zero", log=lambda *_: None)
    assert [h['epoch'] for h in hist] == [2, 3] and all(np.isfinite(h['train_loss']) for h in hist)
    # state_dict round trip into a fresh FlatAdam: moments, step count, lr
    sd = opt.state_dict()
    other = train.FlatAdam(list(fresh().parameters()), lr=5e-4)
    live_avg = other.exp_avg
    other.load_state_dict(sd)
    assert other.exp_avg is live_avg and torch.equal(other.exp_avg, opt.exp_avg) and torch.equal(other.exp_avg_sq, opt.exp_avg_sq)
    assert float(other.steps.item()) == float(opt.steps.item()) and other.param_groups[0]['lr'] == pytest.approx(opt.param_groups[0]['lr'])
    other.check_views()
    # a model moved after construction: rebind() keeps values and moments
    net_d = fresh()
    od = train.FlatAdam(list(net_d.parameters()), lr=1e-3)
    train.train_step(net_d, crit, od, loader[0], initial_image="zero")
    w = [p.detach().clone() for p in net_d.parameters()]
    m = od.exp_avg.clone()
    for p in net_d.parameters():
        p.data = p.data.clone()                                     # what model.to(other device) does to the views
    assert not od.views_intact()
    od.rebind().check_views()
    assert all(torch.equal(a, b) for a, b in zip(w, net_d.parameters())) and torch.equal(m, od.exp_avg) and float(od.steps.item()) == 1.0


def test_prefetched_composite_hands_out_frames_in_request_order_on_the_cpu():
    """parallel_render.PrefetchedComposite without a GPU: start / take bookkeeping (two slots, a frame started ahead is the
    one taken, a frame never started is produced on demand), composite == parallel_render.composite of the same buffers."""
    import torch
    from isosurfacesuperresolution_amd import parallel_render as PR
    calls = []

    def render_fn(tensor, key, stream):
        assert stream is None
        calls.append(key)
        g = torch.Generator().manual_seed(100 + key)
        tensor.copy_(torch.rand(tensor.shape, generator=g))
        tensor[..., 3] = (tensor[..., 3] > 0.5).float()

    src = PR.PrefetchedComposite(render_fn, 6, 8, "cpu")
    src.record(True)
    a = src.take(0).clone()
    src.start(1)
    src.start(1)                                   # a second request for a frame in flight is a no-op
    b = src.take(1).clone()
    c = src.take(5).clone()                        # never started: produced now
    assert calls == [0, 1, 5]
    for key, got in ((0, a), (1, b), (5, c)):
        ref = torch.empty(6, 8, 12)
        render_fn(ref, key, None)
        assert torch.equal(got, PR.composite(ref.unsqueeze(0)))
    assert len(src.timeline) == 3 and all(v >= 0 for v in src.phase_ms())
    # a key is handed out once: asking again for a frame that was already taken renders it again (a looping sequence of period
    # <= 2, or the same index after the camera path changed, must not get the composite that still sits in a slot)
    calls.clear()
    d = src.take(5).clone()
    src.start(0)
    e = src.take(0).clone()
    f = src.take(0).clone()
    assert calls == [5, 0, 0]
    assert torch.equal(d, c) and torch.equal(e, a) and torch.equal(f, a)
