"""Super-resolving ONE frame on several GPUs: horizontal screen strips with a halo.

A temporally coherent sequence cannot be sharded over frames (frame t consumes frame t-1), and at 4K the network
is ~20 ms of a frame, so the remaining axis is the image itself.  The network is purely convolutional with a
receptive field of about 23 low-resolution pixels (1 pre-block conv + 20 block convs at 1x, the post-block convs
at 2x/4x and the bilinear footprints), so a rank that runs it on its strip plus a 24-pixel halo gets exactly the
full-frame values inside the strip.  Per frame and rank:

1. replicated (cheap, exact): the flow hole filling of the whole frame (its pyramid spans the image); the assembly of the
   network input for the rank's rows + halo only -- the warp of the previous high-resolution output samples across strip
   borders, which is why every rank holds the whole previous frame;
2. the convolutional trunk + reconstruction + clamp/normalise + shading on rows ``[y0 - halo, y1 + halo)``;
3. ONE ``all_gather_into_tensor`` of the cropped strips (raw 6 + rgb 3 channels in one buffer), after which every
   rank holds the full output -- which is also the next frame's "previous" input.

The reference has no multi-GPU code; its per-frame sequence is ``inference/loadedmodel.py:75-97`` +
``mainGUI.py:594-603``.  ``tests/test_host_cpu.py`` runs two gloo ranks against the single-process result and
``tests/test_conv_gpu.py`` checks on the GPU that strips computed one after the other reproduce the full frame bit
for bit.
"""
import torch
import torch.distributed as dist

from . import ops
from .inference.flowfill import fill_flow
from .models import VideoTools
from .pipeline import fused_path_ok, run_network
from .utils import ScreenSpaceShading, initialImage

HALO = 24


def strip_bounds(height, world, rank):
    """Rows [y0, y1) of the low-resolution image owned by ``rank`` (balanced to within one row)."""
    return rank * height // world, (rank + 1) * height // world


class StripSuperResolution:
    def __init__(self, model, shading, process_group=None, halo=HALO):
        """model: inference.LoadedModel; shading: utils.ScreenSpaceShading."""
        self.model = model
        self.shading = shading
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        self.halo = halo
        self.upscale = model.upscale_factor
        self.previous = None

    def reset(self):
        self.previous = None

    # -- step 1 (replicated) ---------------------------------------------------------------------------------
    def network_input(self, gbuffer, rows=None):
        """gbuffer [H, W, 12] -> the network's input [1, 5 + 6*upscale^2, H, W].  ``rows`` = (e0, e1): on the fused path only those
        rows are assembled (a rank's strip + halo; the flow hole filling stays global: its pyramid spans the image)."""
        net, lm = self.model.model, self.model
        fused = gbuffer.is_cuda and self.upscale == 4
        if fused:
            flow = ops.fill_flow_gbuffer(gbuffer) if self.previous is not None else None
            return ops.assemble_input(gbuffer, flow, self.previous, lm.initial_image_mode, lm.inverse_ao, rows=rows)
        low = gbuffer.permute(2, 0, 1).unsqueeze(0)
        mask = low[:, 3:4]
        inp = torch.cat((mask * 2 - 1, low[:, 4:8]), dim=1)
        if self.previous is None:
            warped = initialImage(inp, net.output_channels, lm.initial_image_mode, lm.inverse_ao, self.upscale)
        else:
            warped = VideoTools.warp_upscale(self.previous, fill_flow(low[:, 8:10], mask != 0), self.upscale,
                                             special_mask=True)
        return torch.cat((inp, VideoTools.flatten_high(warped, self.upscale)), dim=1)

    # -- step 2 (this rank's rows) -----------------------------------------------------------------------------
    def compute_strip(self, x, rank=None, world=None, after_trunk=None):
        """Network + post-processing on the strip of ``rank`` (default: this process).  x: full-frame network input.
        Returns (raw [1,6,u*rows,u*W], rgb [1,3,u*rows,u*W]) for the strip's own rows (halo cropped).
        ``after_trunk``: called once when the low-resolution trunk is enqueued (``parallel_render.PrefetchedComposite.start``)."""
        rank = self.rank if rank is None else rank
        world = self.world if world is None else world
        h = x.shape[2]
        y0, y1 = strip_bounds(h, world, rank)
        e0, e1 = max(0, y0 - self.halo), min(h, y1 + self.halo)
        xs = x[:, :, e0:e1]
        net = self.model.model
        u = self.upscale
        if x.is_cuda and fused_path_ok(self.model, u):
            # the frame pipeline's network path on the strip (``pipeline.run_network``): dataflow trunk when the strip's tiles fit
            # the CUs, three-workgroup upsampling layers, packed-split hand-over, fused 1080p tail + finish -- one code path for a
            # whole frame and for a strip; per-pixel arithmetic does not depend on the tiling, so strips stay bit-identical to the
            # unsplit frame (tests/test_conv_gpu.py)
            xs = xs if (e0 == 0 and e1 == h) else xs.contiguous()
            ops.guards_poll(xs.device)                     # the previous frame's guard words (pipeline.frame_fused)
            raw, rgb = run_network(self.model, self.shading, xs, after_trunk=after_trunk)
            if ops.range_check_due(xs.device):
                for _ in range(4):                         # range guard, first frame: a hot layer reroutes its consumers; recompute
                    if not ops.refresh_range_flags(xs.device):
                        break
                    raw, rgb = run_network(self.model, self.shading, xs)
            ops.guards_publish(xs.device)
            a, b = (y0 - e0) * u, (y1 - e0) * u
            return raw[:, :, a:b], rgb[:, :, a:b]
        raw, _ = net._recon_image(xs, net.forward_features(xs))
        if after_trunk is not None:
            after_trunk()
        raw = torch.cat([torch.clamp(raw[:, 0:1], -1, +1),
                         ScreenSpaceShading.normalize(raw[:, 1:4], dim=1),
                         torch.clamp(raw[:, 4:], 0, 1)], dim=1)
        raw = raw[:, :, (y0 - e0) * u:(y1 - e0) * u]
        self.shading.inverse_ao = self.model.inverse_ao
        return raw, self.shading(raw)

    # -- step 3 ------------------------------------------------------------------------------------------------
    def frame(self, gbuffer, after_trunk=None):
        """gbuffer: the full low-resolution G-buffer [H, W, 12], identical on every rank (a replicated render or
        the composite of ``parallel_render.TiledRenderer``).  Returns (rgb, raw) of the full frame on every rank."""
        with torch.no_grad():
            h = gbuffer.shape[0]
            y0, y1 = strip_bounds(h, self.world, self.rank)
            x = self.network_input(gbuffer, rows=(max(0, y0 - self.halo), min(h, y1 + self.halo)))      # what compute_strip reads
            raw, rgb = self.compute_strip(x, after_trunk=after_trunk)
            if self.world > 1:
                h, w, u = gbuffer.shape[0], gbuffer.shape[1], self.upscale
                rows = [strip_bounds(h, self.world, r) for r in range(self.world)]
                most = max(b - a for a, b in rows) * u
                mine = torch.zeros((9, most, w * u), dtype=raw.dtype, device=raw.device)
                mine[0:6, :raw.shape[2]] = raw[0]
                mine[6:9, :rgb.shape[2]] = rgb[0]
                everyone = torch.empty((self.world * 9, most, w * u), dtype=raw.dtype, device=raw.device)
                dist.all_gather_into_tensor(everyone, mine, group=self.group)
                everyone = everyone.view(self.world, 9, most, w * u)
                # one pass each (the concatenation writes the contiguous result; no intermediate 9-channel frame)
                raw = torch.cat([everyone[r, 0:6, :(b - a) * u] for r, (a, b) in enumerate(rows)], dim=1).unsqueeze(0)
                rgb = torch.cat([everyone[r, 6:9, :(b - a) * u] for r, (a, b) in enumerate(rows)], dim=1).unsqueeze(0)
            self.previous = raw
        return rgb, raw
