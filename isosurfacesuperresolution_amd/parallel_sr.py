"""Super-resolving ONE frame on several GPUs: screen tiles -- horizontal strips or a (rows x columns) grid -- with a halo
(SURVEY.md 8(e), row 4).

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

Grid.  With strips only, 8 ranks at 960 x 540 own 68 rows each and compute 68 + 48 halo rows: 1.71 x their share.  A 4 x 2 grid
(``grid=(4, 2)``, or ``grid="auto"``: the factorisation of the world size with the smallest largest extended tile) computes
(135 + 48) x (480 + 24) = 1.42 x; a 2 x 4 grid 1.31 x.  Column cuts fall on multiples of 8 low-resolution pixels (the kernels stage
aligned groups of four).  The kernels' per-pixel arithmetic does not depend on where a tensor's borders fall -- in y or in x -- so a
tile's interior is the unsplit frame's bit for bit (``tests/test_conv_gpu.py``); the all-gather moves equal-sized padded rectangles.

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


COLUMN_ALIGN = 8


def tile_bounds(height, width, grid, rank):
    """(y0, y1, x0, x1) of the low-resolution tile of ``rank`` in a ``grid`` = (rows, columns) of tiles, ranks row-major.  Row cuts as
    ``strip_bounds``; column cuts on multiples of ``COLUMN_ALIGN``."""
    gr, gc = grid
    ry, rx = divmod(rank, gc)
    y0, y1 = strip_bounds(height, gr, ry)

    def cut(k):
        return width if k >= gc else min(width, (k * width // gc) // COLUMN_ALIGN * COLUMN_ALIGN)
    return y0, y1, cut(rx), cut(rx + 1)


def extended_area(height, width, grid, halo=HALO):
    """Largest (tile + halo) area over the ranks of a grid, in low-resolution pixels: what the slowest rank computes."""
    worst = 0
    for r in range(grid[0] * grid[1]):
        y0, y1, x0, x1 = tile_bounds(height, width, grid, r)
        if y1 <= y0 or x1 <= x0:
            return float("inf")
        worst = max(worst, (min(height, y1 + halo) - max(0, y0 - halo)) * (min(width, x1 + halo) - max(0, x0 - halo)))
    return worst


def best_grid(world, height, width, halo=HALO):
    """The (rows, columns) factorisation of ``world`` whose largest extended tile is smallest (strips for 1-2 ranks)."""
    grids = [(world // c, c) for c in range(1, world + 1) if world % c == 0]
    return min(grids, key=lambda g: (extended_area(height, width, g, halo), g[1]))


class StripSuperResolution:
    def __init__(self, model, shading, process_group=None, halo=HALO, grid=None):
        """model: inference.LoadedModel; shading: utils.ScreenSpaceShading.  ``grid``: None = horizontal strips (world x 1),
        (rows, columns) with rows * columns == world size, or "auto" (``best_grid`` of the first frame's size)."""
        self.grid = grid
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
    def network_input(self, gbuffer, rows=None, cols=None):
        """gbuffer [H, W, 12] -> the network's input [1, 5 + 6*upscale^2, H, W].  ``rows`` = (e0, e1), ``cols`` = (f0, f1): on the fused
        path only that rectangle is assembled (a rank's tile + halo; the flow hole filling stays global: its pyramid spans the image)."""
        net, lm = self.model.model, self.model
        fused = gbuffer.is_cuda and self.upscale == 4
        if fused:
            flow = ops.fill_flow_gbuffer(gbuffer) if self.previous is not None else None
            return ops.assemble_input(gbuffer, flow, self.previous, lm.initial_image_mode, lm.inverse_ao, rows=rows, cols=cols)
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
    def _grid(self, h, w, world=None):
        world = self.world if world is None else world
        if self.grid is None:
            return (world, 1)
        if self.grid == "auto":
            return best_grid(world, h, w, self.halo)
        assert self.grid[0] * self.grid[1] == world, "grid %s does not match %d ranks" % (self.grid, world)
        return tuple(self.grid)

    def compute_strip(self, x, rank=None, world=None, after_trunk=None, grid=None):
        """Network + post-processing on the tile of ``rank`` (default: this process).  x: full-frame network input (rows outside the
        rank's tile + halo may be unassembled).  Returns (raw [1,6,u*rows,u*cols], rgb [1,3,u*rows,u*cols]) for the tile's own pixels
        (halo cropped).  ``after_trunk``: called once when the low-resolution trunk is enqueued
        (``parallel_render.PrefetchedComposite.start``).  ``grid``: (rows, columns) of tiles (default: this object's; strips if None)."""
        rank = self.rank if rank is None else rank
        world = self.world if world is None else world
        h, w = x.shape[2], x.shape[3]
        grid = self._grid(h, w, world) if grid is None else grid
        y0, y1, x0, x1 = tile_bounds(h, w, grid, rank)
        e0, e1 = max(0, y0 - self.halo), min(h, y1 + self.halo)
        f0, f1 = max(0, x0 - self.halo), min(w, x1 + self.halo)
        xs = x[:, :, e0:e1, f0:f1]
        net = self.model.model
        u = self.upscale
        if x.is_cuda and fused_path_ok(self.model, u):
            # the frame pipeline's network path on the tile (``pipeline.run_network``): dataflow trunk when the tile's 16 x 32 pieces fit
            # the CUs, three-workgroup upsampling layers, packed-split hand-over, fused 1080p tail + finish -- one code path for a
            # whole frame and for a tile; per-pixel arithmetic does not depend on the tiling, so tiles stay bit-identical to the
            # unsplit frame (tests/test_conv_gpu.py)
            xs = xs if (e0 == 0 and e1 == h and f0 == 0 and f1 == w) else xs.contiguous()
            ops.guards_poll(xs.device)                     # the previous frame's guard words (pipeline.frame_fused)
            raw, rgb = run_network(self.model, self.shading, xs, after_trunk=after_trunk)
            if ops.range_check_due(xs.device):
                for _ in range(4):                         # range guard, first frame: a hot layer reroutes its consumers; recompute
                    if not ops.refresh_range_flags(xs.device):
                        break
                    raw, rgb = run_network(self.model, self.shading, xs)
            ops.guards_publish(xs.device)
            a, b = (y0 - e0) * u, (y1 - e0) * u
            c, d = (x0 - f0) * u, (x1 - f0) * u
            return raw[:, :, a:b, c:d], rgb[:, :, a:b, c:d]
        raw, _ = net._recon_image(xs, net.forward_features(xs))
        if after_trunk is not None:
            after_trunk()
        raw = torch.cat([torch.clamp(raw[:, 0:1], -1, +1),
                         ScreenSpaceShading.normalize(raw[:, 1:4], dim=1),
                         torch.clamp(raw[:, 4:], 0, 1)], dim=1)
        raw = raw[:, :, (y0 - e0) * u:(y1 - e0) * u, (x0 - f0) * u:(x1 - f0) * u]
        self.shading.inverse_ao = self.model.inverse_ao
        return raw, self.shading(raw)

    # -- step 3 ------------------------------------------------------------------------------------------------
    def frame(self, gbuffer, after_trunk=None):
        """gbuffer: the full low-resolution G-buffer [H, W, 12], identical on every rank (a replicated render or
        the composite of ``parallel_render.TiledRenderer``).  Returns (rgb, raw) of the full frame on every rank."""
        with torch.no_grad():
            h, w, u = gbuffer.shape[0], gbuffer.shape[1], self.upscale
            grid = self._grid(h, w)
            y0, y1, x0, x1 = tile_bounds(h, w, grid, self.rank)
            x = self.network_input(gbuffer, rows=(max(0, y0 - self.halo), min(h, y1 + self.halo)),
                                   cols=(max(0, x0 - self.halo), min(w, x1 + self.halo)))               # what compute_strip reads
            raw, rgb = self.compute_strip(x, after_trunk=after_trunk, grid=grid)
            if self.world > 1:
                tiles = [tile_bounds(h, w, grid, r) for r in range(self.world)]
                most_h = max(b - a for a, b, _, _ in tiles) * u
                most_w = max(d - c for _, _, c, d in tiles) * u
                mine = torch.zeros((9, most_h, most_w), dtype=raw.dtype, device=raw.device)        # equal-sized padded rectangles
                mine[0:6, :raw.shape[2], :raw.shape[3]] = raw[0]
                mine[6:9, :rgb.shape[2], :rgb.shape[3]] = rgb[0]
                everyone = torch.empty((self.world * 9, most_h, most_w), dtype=raw.dtype, device=raw.device)
                dist.all_gather_into_tensor(everyone, mine, group=self.group)
                everyone = everyone.view(self.world, 9, most_h, most_w)

                def assemble(c0, c1):          # one pass per tile row (the concatenations write the contiguous result)
                    rows = []
                    for ry in range(grid[0]):
                        parts = []
                        for rx in range(grid[1]):
                            r = ry * grid[1] + rx
                            a, b, c, d = tiles[r]
                            parts.append(everyone[r, c0:c1, :(b - a) * u, :(d - c) * u])
                        rows.append(parts[0] if len(parts) == 1 else torch.cat(parts, dim=2))
                    return (rows[0] if len(rows) == 1 else torch.cat(rows, dim=1)).unsqueeze(0)
                raw, rgb = assemble(0, 6), assemble(6, 9)
            self.previous = raw
        return rgb, raw
