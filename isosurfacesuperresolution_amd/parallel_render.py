"""Object-space tiled rendering of volumes that are split over the GPUs of a node
(BASELINE config #5: 1024^3 as 2x2x2 tiles of 512^3).

The reference has nothing like this (single device).  Every rank ray-marches the FULL low-res image
against its own tile (loaded with ``DirectRenderer.load_tile``), then the 12-channel G-buffers are
exchanged with ONE all-gather (RCCL over xGMI: 24.9 MB per GPU at 960x540) and composited per pixel by
nearest hit.

The composite is BIT-IDENTICAL to the render of the unsplit volume: a tile walks the global ray (global
world map, isovalue scale, clip box and 4096/128/8-voxel DDAs in global index coordinates) and only
*processes* the leaves it owns.  The reference's tracer re-initialises its voxel DDA per leaf from the
leaf's own span (``CPURenderer/IsoVolumeRayTracer.h:37-46``), so what happens inside a leaf does not
depend on which other leaves exist; the tile that owns the first leaf with a crossing therefore computes
exactly the unsplit pixel, every other tile a later hit or none, and the minimum depth selects it.
"""
import time

import numpy as np
import torch
import torch.distributed as dist

HALO = 8      # multiples of the 8^3 leaf keep the tile's leaf grid aligned with the global one


def global_stats(volume):
    """Active-voxel bbox (x,y,z order) and maximum of a dense [z][y][x] volume."""
    nzv = np.argwhere(volume != 0)
    lo, hi = nzv.min(0)[::-1], nzv.max(0)[::-1]
    return [int(v) for v in lo], [int(v) for v in hi], float(volume.max())


def tile_boxes(shape, splits):
    """[(lo(x,y,z), hi(x,y,z))] of an (sz, sy, sx) split of a [z][y][x] volume, z-major rank order."""
    nz, ny, nx = shape
    sz, sy, sx = splits
    # interior edges on the 8^3 leaf grid: a leaf belongs to exactly one tile
    edges = lambda n, s: [0] + [min(n, 8 * round(i * n / s / 8)) for i in range(1, s)] + [n]
    ex, ey, ez = edges(nx, sx), edges(ny, sy), edges(nz, sz)
    boxes = []
    for k in range(sz):
        for j in range(sy):
            for i in range(sx):
                boxes.append(((ex[i], ey[j], ez[k]), (ex[i + 1], ey[j + 1], ez[k + 1])))
    return boxes


def make_tile(volume_slice_fn, shape, box, gmin, gmax, gmaxval):
    """Tile descriptor for ``DirectRenderer.load_tile``.  ``volume_slice_fn(z0,z1,y0,y1,x0,x1)`` returns the
    dense data of that index range (so a rank can generate or read only what it owns)."""
    nz, ny, nx = shape
    (x0, y0, z0), (x1, y1, z1) = box
    ox, oy, oz = max(0, x0 - HALO), max(0, y0 - HALO), max(0, z0 - HALO)
    ex, ey, ez = min(nx, x1 + HALO), min(ny, y1 + HALO), min(nz, z1 + HALO)
    return {"data": volume_slice_fn(oz, ez, oy, ey, ox, ex), "origin": (ox, oy, oz), "gmin": gmin, "gmax": gmax,
            "gmaxval": gmaxval, "clip_lo": (x0, y0, z0), "clip_hi": (x1, y1, z1)}


def partition_volume(volume, splits=(2, 2, 2)):
    """All tiles of an in-memory volume (tests / single-process use)."""
    gmin, gmax, gmaxval = global_stats(volume)
    fn = lambda z0, z1, y0, y1, x0, x1: volume[z0:z1, y0:y1, x0:x1]
    return [make_tile(fn, volume.shape, box, gmin, gmax, gmaxval) for box in tile_boxes(volume.shape, splits)]


def generate_tiles(field, splits=(2, 2, 2), ranks=None, reduce_max=None, reduce_min=None):
    """Tiles of a volume that exists only as a generator (``volumes.EjectaField``): each tile evaluates
    its own box plus halo once; the normalising maximum and the active bounding box are reductions
    over the tiles' own values (between ranks: ``reduce_max`` / ``reduce_min`` callables over numpy
    arrays, e.g. an all-reduce; in one process the loop below).  ``ranks``: tile indices to build
    (default all).  Returns the tile descriptors in ``tile_boxes`` order (None for skipped ranks)."""
    n = field.n
    shape = (n, n, n)
    boxes = tile_boxes(shape, splits)
    ranks = list(range(len(boxes))) if ranks is None else list(ranks)
    raws, metas = {}, {}
    for k in ranks:
        (x0, y0, z0), (x1, y1, z1) = boxes[k]
        ox, oy, oz = max(0, x0 - HALO), max(0, y0 - HALO), max(0, z0 - HALO)
        ex, ey, ez = min(n, x1 + HALO), min(n, y1 + HALO), min(n, z1 + HALO)
        raws[k] = field.raw((oz, ez, oy, ey, ox, ex))
        metas[k] = (ox, oy, oz)
    gmaxraw = np.array([max(float(r.max()) for r in raws.values())], dtype=np.float64)
    if reduce_max is not None:
        gmaxraw = reduce_max(gmaxraw)
    lo = np.full(3, np.iinfo(np.int64).max, dtype=np.int64)
    hi = np.full(3, -1, dtype=np.int64)
    vmax = np.zeros(1, dtype=np.float64)
    data = {}
    for k in ranks:
        d = field.finalize(raws.pop(k), gmaxraw[0])
        data[k] = d
        (x0, y0, z0), (x1, y1, z1) = boxes[k]
        ox, oy, oz = metas[k]
        own = d[z0 - oz:z1 - oz, y0 - oy:y1 - oy, x0 - ox:x1 - ox]
        for axis, base in ((2, x0), (1, y0), (0, z0)):             # (x, y, z) order
            other = tuple(a for a in (0, 1, 2) if a != axis)
            nzi = np.flatnonzero(own.any(axis=other))
            if nzi.size:
                lo[2 - axis] = min(lo[2 - axis], base + int(nzi[0]))
                hi[2 - axis] = max(hi[2 - axis], base + int(nzi[-1]))
        vmax[0] = max(vmax[0], float(own.max()))
    if reduce_max is not None:
        hi, vmax = reduce_max(hi), reduce_max(vmax)
    if reduce_min is not None:
        lo = reduce_min(lo)
    tiles = [None] * len(boxes)
    for k in ranks:
        tiles[k] = {"data": data[k], "origin": metas[k], "gmin": [int(v) for v in lo], "gmax": [int(v) for v in hi],
                    "gmaxval": float(np.float32(vmax[0])), "clip_lo": boxes[k][0], "clip_hi": boxes[k][1]}
    return tiles


def assemble(tiles, shape):
    """The dense volume the tiles were cut from (tests, single-GPU rehearsal of config #5)."""
    vol = np.zeros(shape, dtype=np.float32)
    for t in tiles:
        (x0, y0, z0), (x1, y1, z1) = t["clip_lo"], t["clip_hi"]
        ox, oy, oz = t["origin"]
        vol[z0:z1, y0:y1, x0:x1] = t["data"][z0 - oz:z1 - oz, y0 - oy:y1 - oy, x0 - ox:x1 - ox]
    return vol


def composite(gbuffers, extra=None):
    """Nearest-hit composite of per-tile G-buffers [T, H, W, 12] -> [H, W, 12] (mask ch 3, depth ch 7).  ``extra`` [T, H, W, C]
    (any dtype, e.g. the tiles' hit-state exports) is composited with the same per-pixel winner: returns (out, extra_out)."""
    mask = gbuffers[..., 3] > 0
    depth = torch.where(mask, gbuffers[..., 7], torch.full_like(gbuffers[..., 7], float("inf")))
    win = depth.argmin(dim=0)                                        # [H, W]
    idx = win.unsqueeze(0).unsqueeze(-1).expand(1, *gbuffers.shape[1:])
    out = torch.gather(gbuffers, 0, idx)[0]
    if extra is None:
        return out
    eidx = win.unsqueeze(0).unsqueeze(-1).expand(1, *extra.shape[1:])
    return out, torch.gather(extra, 0, eidx)[0]


class TiledRenderer:
    """One rank of the tiled render: local ray-march, all-gather, composite (identical on all ranks)."""

    def __init__(self, renderer, tile, process_group=None, render_fn=None):
        """``render_fn(tensor[H,W,12])`` overrides the local render (CPU tests drive the exchange and the
        composite over gloo with the oracle as the local renderer)."""
        self.renderer = renderer
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.render_fn = render_fn
        if renderer is not None:
            renderer.load_tile(tile)

    def render(self, width, height, device="cuda", ao_samples=0):
        """The composited frame.  ``ao_samples > 0``: channel 10 is the ray-cast ambient occlusion of the WHOLE volume, exactly
        (``render_with_ao``); 0: AO == 1 as in SR mode."""
        if ao_samples > 0 and self.renderer is not None:
            return self.render_with_ao(width, height, ao_samples, device)
        local = torch.empty((height, width, 12), dtype=torch.float32, device=device)
        if self.render_fn is not None:
            self.render_fn(local)
        else:
            self.renderer.render_direct(local)
        if self.world == 1:
            return local
        gathered = torch.empty((self.world * height, width, 12), dtype=torch.float32, device=device)
        dist.all_gather_into_tensor(gathered, local, group=self.group)       # concatenated along dim 0
        return composite(gathered.view(self.world, height, width, 12))

    def render_with_ao(self, width, height, ao_samples, device="cuda"):
        """Ray-cast AO across tiles, bit-identical to the unsplit render.  An AO ray ends at its first hit ANYWHERE in the volume
        (``render_kernel.cu:109-146``), so no halo makes a tile's own AO right; instead every tile casts every hit pixel's rays
        against its own leaves and the minimum over the tiles is each ray's true distance (``gpu_renderer_direct.h``,
        isoSetHitStateBuffer ...).  Per frame: all-gather of the G-buffers and of the hit states (48 B / pixel), one all-reduce
        (MIN) of width x height x ao_samples doubles."""
        r = self.renderer
        local = torch.empty((height, width, 12), dtype=torch.float32, device=device)
        state = torch.zeros((height, width, 6), dtype=torch.float64, device=device)
        r.send_command("aosamples", "0")
        r.set_hit_state_buffer(state)
        try:
            r.render_direct(local)
        finally:
            r.set_hit_state_buffer(None)
        if self.world > 1:
            g = torch.empty((self.world * height, width, 12), dtype=torch.float32, device=device)
            h = torch.empty((self.world * height, width, 6), dtype=torch.float64, device=device)
            dist.all_gather_into_tensor(g, local, group=self.group)
            dist.all_gather_into_tensor(h, state, group=self.group)
            local, state = composite(g.view(self.world, height, width, 12), h.view(self.world, height, width, 6))
            local, state = local.contiguous(), state.contiguous()
        r.send_command("aosamples", "%d" % ao_samples)
        d = torch.empty((height, width, ao_samples), dtype=torch.float64, device=device)
        r.ao_distances(state, local, d)
        if self.world > 1:
            torch.cuda.synchronize()
            dist.all_reduce(d, op=dist.ReduceOp.MIN, group=self.group)
        r.ao_finish(d, local)
        torch.cuda.synchronize()
        return local


class PrefetchedComposite:
    """The composited G-buffer of frame t + 1 produced on a side HIP stream -- this rank's ray-march, the all-gather, the
    nearest-hit composite -- while the caller's stream super-resolves frame t (the frame pipeline's render(t+1) || SR(t),
    ``pipeline.py``, for the tiled path).  A frame's image does not depend on the previous frame's network output, only on the
    camera path, so the order of results is unchanged: ``take(key)`` returns exactly what the serial sequence returns.

    ``render_fn(tensor[H,W,12], key, stream)`` enqueues the local render of frame ``key`` on ``stream`` (None on the CPU).
    Collectives: the all-gather is issued from the side stream through the SAME process group as every other collective of the
    job; torch's group runs its collectives on one internal stream in the order the host issued them, and this class issues
    them at the same point of the frame on every rank, so the two all-gathers of a frame (G-buffers of t + 1, strips of t) are
    totally ordered and identical on all ranks -- no second communicator, nothing that can interleave differently per rank.
    On a CPU device (gloo rehearsals) everything runs in the caller's thread, one step after the other."""

    def __init__(self, render_fn, height, width, device, process_group=None):
        self.render_fn = render_fn
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.device = torch.device(device)
        self.cuda = self.device.type == "cuda"
        self.local = torch.empty((height, width, 12), dtype=torch.float32, device=device)
        self.gathered = torch.empty((self.world, height, width, 12), dtype=torch.float32, device=device)
        self.slots = [torch.empty((height, width, 12), dtype=torch.float32, device=device) for _ in range(2)]
        self.side = torch.cuda.Stream(device=self.device) if self.cuda else None
        self.ready = [None, None]          # per slot: event on the side stream when the composite is complete
        self.free = [None, None]           # per slot: event on the consumer's stream after which the slot may be overwritten
        self.keys = [None, None]
        self.turn = 0                      # the slot the next start() fills
        self.taken = None
        self.timeline = None               # list of (e0, e1, e2, e3) event tuples while record(True)

    def record(self, on):
        self.timeline = [] if on else None

    def _produce(self, key, slot, stream):
        if self.timeline is None:
            ev = lambda: None
        elif self.cuda:
            ev = lambda: stream.record_event(torch.cuda.Event(enable_timing=True))
        else:
            ev = time.perf_counter
        e0 = ev()
        self.render_fn(self.local, key, stream)
        e1 = ev()
        if self.world > 1:
            dist.all_gather_into_tensor(self.gathered.view(self.world * self.local.shape[0], *self.local.shape[1:]), self.local, group=self.group)
        else:
            self.gathered[0].copy_(self.local)
        e2 = ev()
        self.slots[slot].copy_(composite(self.gathered))
        e3 = ev()
        if e0 is not None:
            self.timeline.append((e0, e1, e2, e3))

    def start(self, key, after_current=True):
        """Enqueue frame ``key``.  ``after_current``: the side stream first waits for what the caller's stream holds now (the
        frame pipeline's release point: "when the trunk has ended")."""
        if key in self.keys:
            return
        slot = self.turn
        self.turn ^= 1
        self.keys[slot] = key
        if not self.cuda:
            self._produce(key, slot, None)
            return
        cur = torch.cuda.current_stream(self.device)
        if after_current:
            self.side.wait_stream(cur)
        if self.free[slot] is not None:
            self.side.wait_event(self.free[slot])
        with torch.cuda.stream(self.side):
            self._produce(key, slot, self.side)
            self.ready[slot] = self.side.record_event()

    def take(self, key):
        """The composite of frame ``key`` (started earlier, or now); the caller's stream waits for it.  The tensor stays valid until
        the second ``take`` after this one."""
        if self.cuda and self.taken is not None:
            self.free[self.taken] = torch.cuda.current_stream(self.device).record_event()      # the previous frame's consumer is enqueued
        if key not in self.keys:
            self.start(key, after_current=True)            # nothing ahead: the serial order, after everything enqueued so far
        slot = self.keys.index(key)
        # a composite is handed out ONCE: the key is forgotten here, so a later request for the same key (a looping sequence, the
        # same frame index after the camera path or isovalue changed) renders again instead of returning what sits in the slot --
        # its flow channels were measured against the camera that preceded it THEN
        self.keys[slot] = None
        if self.cuda:
            torch.cuda.current_stream(self.device).wait_event(self.ready[slot])
        self.taken = slot
        return self.slots[slot]

    def phase_ms(self):
        """Summed (render, all-gather, composite) milliseconds of the recorded frames (after a device synchronisation)."""
        out = [0.0, 0.0, 0.0]
        for e0, e1, e2, e3 in self.timeline or ():
            if self.cuda:
                out[0] += e0.elapsed_time(e1); out[1] += e1.elapsed_time(e2); out[2] += e2.elapsed_time(e3)
            else:
                out[0] += (e1 - e0) * 1e3; out[1] += (e2 - e1) * 1e3; out[2] += (e3 - e2) * 1e3
        return out
