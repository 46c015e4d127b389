"""CPU: the package's SR modules against golden vectors produced by importing the reference's own
Python modules (tests/golden/make_sr_fixtures.py -> tests/golden/sr_reference.npz)."""
import argparse
import os

import numpy as np
import pytest
import torch

from isosurfacesuperresolution_amd import models, utils
from isosurfacesuperresolution_amd.models import VideoTools

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "sr_reference.npz"))
OPT = argparse.Namespace(upsample='bilinear', reconType='residual', useBN=False, numResidualLayers=10)


@pytest.fixture(scope="module")
def net():
    # orthogonal_ runs a LAPACK QR whose rounding depends on the thread count; the fixtures were
    # generated single-threaded
    nt = torch.get_num_threads()
    torch.set_num_threads(1)
    torch.manual_seed(0)
    m = models.createNetwork('EnhanceNet', 4, 101, [0, 1, 2, 3, 4], 6, OPT).eval()
    torch.set_num_threads(nt)
    return m


def test_enhancenet_init_matches_reference(net):
    # same construction order + same initialisers => identical weights for the same seed (S1)
    assert int(G["net_param_count"]) == sum(p.numel() for p in net.parameters()) == 911046
    assert list(G["net_state_keys"]) == list(net.state_dict().keys())
    sums = np.array([p.double().sum().item() for p in net.state_dict().values()])
    np.testing.assert_allclose(sums, G["net_param_sums"], rtol=0, atol=1e-9)
    assert abs(sum(p.abs().double().sum().item() for p in net.parameters()) - 38031.676260) < 1e-3   # SURVEY KA1 (fp32 sum there)


def test_enhancenet_forward_matches_reference(net):
    torch.manual_seed(1)
    x = torch.rand(1, 101, 8, 8)
    with torch.no_grad():
        y, raw = net(x)
    np.testing.assert_allclose(y.numpy(), G["net_y"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(raw.numpy(), G["net_raw"], rtol=0, atol=1e-5)
    torch.manual_seed(1)
    x16 = torch.rand(1, 101, 16, 16)
    with torch.no_grad():
        y16, r16 = net(x16)
    ka2 = np.array([y16.mean().item(), y16.abs().mean().item(), r16.abs().mean().item()])
    np.testing.assert_allclose(ka2, G["net_ka2"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(ka2, [0.6064585, 1.4393197, 1.2635158], atol=1e-6)


def test_create_network_names():
    with pytest.raises(ValueError):
        models.createNetwork('nope', 4, 101, [0], 6, OPT)
    assert isinstance(models.createNetwork('enhancenet', 4, 101, [0, 1, 2, 3, 4], 6, OPT), models.EnhanceNet)


def test_flatten_high():
    img = torch.from_numpy(G["vt_img"])
    f = VideoTools.flatten_high(img, 4)
    assert np.array_equal(f.numpy(), G["vt_flatten"])
    assert f.shape == (1, 96, 8, 8) and f[0, 17, 3, 5] == img[0, 1, 12, 21]


@pytest.mark.parametrize("key,flow_key,special", [
    ("vt_warp_special", "vt_flow", True), ("vt_warp_plain", "vt_flow", False), ("vt_warp_big", "vt_flow_big", True)])
def test_warp_upscale(key, flow_key, special):
    """Two forms of the same function.  ``warp_upscale_library`` makes the reference's own calls (F.interpolate + F.grid_sample on a
    linspace grid, videotools.py:51-87) and reproduces the reference-generated fixture to 1e-6: that pins the SEMANTICS (flow
    scaling, align_corners, padding, the special mask).  ``warp_upscale`` -- the package's definition since round 5, the one the HIP
    kernels reproduce bit for bit -- spells the same function out in elementwise operations: in fp64 the two forms agree to 1e-12
    (same function), in fp32 each is a differently rounded evaluation of it and the normalised-coordinate formulation turns a 6e-8
    rounding of the grid into (W - 1) / 2 times that in pixels, so the fixture itself is 3-5e-6 from the fp64 value here (W = 32;
    1e-4 at 1080p, see models/videotools.py).  The explicit form must be as close to fp64 as the reference's own output is."""
    img, flow = torch.from_numpy(G["vt_img"]), torch.from_numpy(G[flow_key])
    lib = VideoTools.warp_upscale_library(img, flow, 4, special_mask=special)
    np.testing.assert_allclose(lib.numpy(), G[key], rtol=0, atol=1e-6)
    w = VideoTools.warp_upscale(img, flow, 4, special_mask=special)
    w64 = VideoTools.warp_upscale(img.double(), flow.double(), 4, special_mask=special)
    lib64 = VideoTools.warp_upscale_library(img.double(), flow.double(), 4, special_mask=special)
    assert (w64 - lib64).abs().max().item() <= 1e-12
    ref_err = np.abs(G[key].astype(np.float64) - w64.numpy()).max()            # the reference's own fp32 output against fp64
    own_err = (w.double() - w64).abs().max().item()
    assert own_err <= 2.0 * ref_err + 1e-6, (own_err, ref_err)
    np.testing.assert_allclose(w.numpy(), G[key], rtol=0, atol=2e-5)             # and never further from it than the conditioning explains


def test_warp_zero_flow_is_identity():
    img = torch.from_numpy(G["vt_img"])
    w_lib = VideoTools.warp_upscale_library(img, torch.zeros(1, 2, 8, 8), 4, special_mask=True)
    np.testing.assert_allclose(w_lib.numpy(), G["vt_warp_zero_flow"], rtol=0, atol=1e-6)
    w = VideoTools.warp_upscale(img, torch.zeros(1, 2, 8, 8), 4, special_mask=True)
    np.testing.assert_allclose(w.numpy(), G["vt_warp_zero_flow"], rtol=0, atol=1e-5)     # (both forms are ~3.7e-6 from the identity: fp32 grid)
    assert (w - img).abs().max() < 1e-5          # align_corners=True semantics
    assert (VideoTools.warp_upscale(img.double(), torch.zeros(1, 2, 8, 8, dtype=torch.float64), 4, special_mask=True) - img.double()).abs().max() < 1e-12
    w_ref = VideoTools.warp_upscale(img, torch.from_numpy(G["vt_flow"]), 4, special_mask=True)
    np.testing.assert_allclose([w_ref.mean().item(), w_ref.abs().mean().item()], [0.3714283, 0.4452816], atol=1e-6)   # KA3


def _shader():
    sh = utils.ScreenSpaceShading('cpu')
    sh.fov(30)
    sh.ambient_light_color(np.array([0.1, 0.1, 0.1]))
    sh.diffuse_light_color(np.array([1.0, 1.0, 1.0]))
    sh.specular_light_color(np.array([0.2, 0.2, 0.2]))
    sh.specular_exponent(16)
    sh.light_direction(np.array([0.1, 0.1, 1.0]))
    sh.material_color(np.array([1.0, 0.3, 0.0]))
    sh.ambient_occlusion(1.0)
    sh.background(np.array([0.2, 0.4, 0.6]))
    return sh


def test_screen_space_shading():
    sh = _shader()
    g = torch.from_numpy(G["sh_in"])
    np.testing.assert_allclose(sh(g).numpy(), G["sh_out"], rtol=0, atol=1e-6)
    sh.inverse_ao = True
    sh.ambient_occlusion(0.6)
    np.testing.assert_allclose(sh(g).numpy(), G["sh_out_invao"], rtol=0, atol=1e-6)
    sh.inverse_ao = False
    sh.enable_specular = False
    np.testing.assert_allclose(sh(g[:, 0:5]).numpy(), G["sh_out_nospec"], rtol=0, atol=1e-6)
    z = torch.zeros(1, 3, 2, 2)
    z[0, :, 0, 0] = torch.tensor([3.0, 0.0, 4.0])
    np.testing.assert_allclose(utils.ScreenSpaceShading.normalize(z, dim=1).numpy(), G["sh_normalize"], atol=1e-7)


def test_initial_image():
    low = torch.from_numpy(G["ii_low"])
    np.testing.assert_allclose(utils.initialImage(low, 6, 'input', False, 4).numpy(), G["ii_input"], atol=1e-6)
    assert np.array_equal(utils.initialImage(low, 6, 'unshaded', False, 4).numpy(), G["ii_unshaded"])
    assert np.array_equal(utils.initialImage(low, 6, 'unshaded', True, 4).numpy(), G["ii_unshaded_inv"])
    z = utils.initialImage(low, 6, 'zero', False, 4)
    assert list(z.shape) == list(G["ii_zero_shape"]) and z.abs().sum() == 0
    with pytest.raises(ValueError):
        utils.initialImage(low, 6, 'bogus', False, 4)


def test_psnr():
    a, b, m = (torch.from_numpy(G[k]) for k in ("psnr_a", "psnr_b", "psnr_m"))
    np.testing.assert_allclose(utils.PSNR()(a, b).numpy(), G["psnr_plain"], rtol=1e-6)
    np.testing.assert_allclose(utils.PSNR()(a, b, m).numpy(), G["psnr_masked"], rtol=1e-6)


def test_mean_variance_against_numpy():
    # the reference's only unit test (utils/mv.py:31-52)
    rng = np.random.default_rng(0)
    for n in (1, 2, 5, 20, 1000):
        xs = rng.random(n)
        mv = utils.MeanVariance()
        for v in xs:
            mv.append(v)
        assert mv.count() == n
        assert abs(mv.mean() - xs.mean()) < 1e-9 and abs(mv.var() - xs.var()) < 1e-9


def test_reference_checkpoint_loads_and_reproduces_the_reference_output(tmp_path):
    """A checkpoint written the way the reference's training script writes it (the whole pickled
    `models.enhancenet.EnhanceNet` object + option dict + optimizer + scheduler; generated by
    tests/golden/make_checkpoint_fixture.py from the reference package) loads through `inference.LoadedModel`
    (module aliases, weights_only=False) and gives the reference network's own output."""
    import zipfile
    from isosurfacesuperresolution_amd import inference
    here = os.path.join(os.path.dirname(__file__), "golden")
    with zipfile.ZipFile(os.path.join(here, "ref_checkpoint.zip")) as z:
        z.extractall(tmp_path)
    io = np.load(os.path.join(here, "ref_checkpoint_io.npz"))
    lm = inference.LoadedModel(str(tmp_path / "model_epoch_12.pth"), "cpu", 4)
    assert lm.name == "model_epoch_12" and lm.unshaded and lm.input_channels == 101
    assert lm.initial_image_mode == "zero" and lm.inverse_ao is False
    assert sum(p.numel() for p in lm.model.parameters()) == int(io["param_count"]) == 911046
    assert type(lm.model).__module__.startswith("isosurfacesuperresolution_amd")      # resolved to THIS package's class
    with torch.no_grad():
        y, raw = lm.model(torch.from_numpy(io["x"]))
    assert np.abs(y.numpy() - io["y"]).max() <= 1e-5 and np.abs(raw.numpy() - io["raw"]).max() <= 1e-5
    # and through the viewer's per-frame entry point (first frame: initial image, no previous output)
    low = torch.rand(1, 12, 12, 10, generator=torch.Generator().manual_seed(3))
    out = lm.inference(low, None)
    assert out.shape == (1, 6, 48, 40) and torch.isfinite(out).all()
