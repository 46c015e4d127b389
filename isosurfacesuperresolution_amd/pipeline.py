"""One frame of the hot path: ray-march the low-resolution G-buffer, super-resolve it 4x with the
temporal EnhanceNet, shade it in screen space.

This is the call sequence of the reference's scripted 1080p video
(``SuperresolutionNetwork/mainComparisonVideo3.py:430-530``: nine ``send_command`` calls,
``render_direct`` into a ``[H, W, 12]`` device tensor, ``LoadedModel.inference``, clamp/normalise,
``ScreenSpaceShading``) and of the interactive viewer (``mainGUI.py:642-720,573-603``), with all
stages enqueued on one HIP stream and no host synchronisation inside the frame.
"""
import numpy as np
import torch

from . import ops
from .inference.flowfill import fill_flow
from .utils import ScreenSpaceShading
from .volumes import fmt3


def default_shading(device, fov=30.0):
    """Shading set-up of ``mainComparisonVideo3.py`` (light from the camera, white-ish material)."""
    sh = ScreenSpaceShading(device)
    sh.fov(fov)
    sh.ambient_light_color(np.array([0.1, 0.1, 0.1]))
    sh.diffuse_light_color(np.array([0.8, 0.8, 0.8]))
    sh.specular_light_color(np.array([0.02, 0.02, 0.02]))
    sh.specular_exponent(16)
    sh.light_direction(np.array([0.1, 0.1, 1.0]))
    sh.material_color(np.array([1.0, 1.0, 1.0]))
    sh.ambient_occlusion(1.0)
    sh.background(np.array([1.0, 1.0, 1.0]))
    return sh


class SuperResolutionPipeline:
    def __init__(self, renderer, model, shading, low_res, upscale=4, temporal=True, device="cuda", fused=True):
        self.renderer = renderer
        self.model = model            # inference.LoadedModel
        self.shading = shading
        self.low_w, self.low_h = low_res
        self.upscale = upscale
        self.temporal = temporal
        self.device = device
        self.gbuffer = torch.empty((self.low_h, self.low_w, 12), dtype=torch.float32, device=device)
        self.previous = None
        # fused=True: input assembly and frame finishing run as two HIP kernels (ops.assemble_input /
        # ops.finish_frame); fused=False: the module-level PyTorch path (LoadedModel.inference etc.)
        self.fused = fused and upscale == 4 and getattr(model.model, 'recon_type', None) == 'residual' \
            and getattr(model.model, 'channel_mask', None) is not None and len(model.model.channel_mask) == 5
        self.set_static(fov=shading.get_fov(), isovalue=0.5)

    def set_static(self, fov, isovalue, lookat=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0)):
        r = self.renderer
        r.send_command("cameraLookAt", fmt3(lookat))
        r.send_command("cameraUp", fmt3(up))
        r.send_command("cameraFoV", "%.3f" % fov)
        r.send_command("isovalue", "%5.3f" % float(isovalue))
        r.send_command("aoradius", "%5.3f" % 0.01)
        r.send_command("aosamples", "0")       # SR mode renders without AO (mainGUI.py:690-691)
        r.send_command("resolution", "%d,%d" % (self.low_w, self.low_h))
        r.send_command("viewport", "%d,%d,%d,%d" % (0, 0, self.low_w, self.low_h))

    def reset(self):
        self.previous = None

    def render_low(self, origin):
        self.renderer.send_command("cameraOrigin", fmt3(origin))
        self.renderer.render_async(self.gbuffer, torch.cuda.current_stream())
        return self.gbuffer.permute(2, 0, 1).unsqueeze(0)

    def superresolve(self, low):
        raw = self.model.inference(low, self.previous if self.temporal else None)
        # mainGUI.py:594-599 -- this tensor is also next frame's "previous high-res" input
        raw = torch.cat([torch.clamp(raw[:, 0:1], -1, +1),
                         ScreenSpaceShading.normalize(raw[:, 1:4], dim=1),
                         torch.clamp(raw[:, 4:], 0, 1)], dim=1)
        self.previous = raw
        return raw

    def frame_fused(self, origin):
        with torch.no_grad():
            self.render_low(origin)
            g = self.gbuffer
            prev = self.previous if self.temporal else None
            flow = None
            if prev is not None:
                flow = ops.fill_flow_gbuffer(g)
            x = ops.assemble_input(g, flow, prev, self.model.initial_image_mode, self.model.inverse_ao)
            feat = self.model.model.forward_features(x)
            self.shading.inverse_ao = self.model.inverse_ao
            raw, rgb = ops.finish_frame(feat, x, self.shading)
            self.previous = raw
        return rgb, raw

    def frame(self, origin):
        """origin: camera position. Returns (rgb [1,3,4h,4w], raw [1,6,4h,4w]) on the device."""
        if self.fused:
            return self.frame_fused(origin)
        with torch.no_grad():
            low = self.render_low(origin)
            raw = self.superresolve(low)
            self.shading.inverse_ao = self.model.inverse_ao
            rgb = self.shading(raw)
        return rgb, raw
