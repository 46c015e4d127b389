"""One frame of the hot path: ray-march the low-resolution G-buffer, super-resolve it 4x with the
temporal EnhanceNet, shade it in screen space.

This is the call sequence of the reference's scripted 1080p video
(``SuperresolutionNetwork/mainComparisonVideo3.py:430-530``: nine ``send_command`` calls,
``render_direct`` into a ``[H, W, 12]`` device tensor, ``LoadedModel.inference``, clamp/normalise,
``ScreenSpaceShading``) and of the interactive viewer (``mainGUI.py:642-720,573-603``), with all
stages enqueued on one HIP stream and no host synchronisation inside the frame.
"""
import os

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


def run_network(model, shading, x, after_trunk=None, prefetch_point="trunk", out=None):
    """The network and the frame's finishing on the fused HIP path: input [1,101,h,w] -> (raw [1,6,4h,4w] clamped / normalised,
    rgb [1,3,4h,4w]).  ``model``: inference.LoadedModel; ``after_trunk``: callable run once at ``prefetch_point`` (the frame
    pipeline releases the next frame's ray-march there).  Used by ``SuperResolutionPipeline`` for whole frames and by
    ``parallel_sr.StripSuperResolution`` for a rank's strip (``enhancenet.py:92-144`` + ``mainGUI.py:594-603``).
    ``out``: (raw, rgb) buffers to write into -- honoured on the fused-tail route only (returns None if another route was taken
    while ``out`` was given: the frame graph then stays off)."""
    net = model.model
    shading.inverse_ao = model.inverse_ao
    last = net.postblock[8]
    six = net.postblock[6]
    f4 = None
    tail = last.weight.shape[0] == 6 and ops.TAIL_FUSION and ops.SPLIT_F16 and not ops.FAST_F16 and not ops.any_hot(x.device)
    four = net.postblock[4]
    if tail and ops.TAIL_PACKED and tuple(six.weight.shape) == (64, 64, 3, 3) and tuple(last.weight.shape) == (6, 64, 3, 3):
        at = prefetch_point
        f2 = net.forward_features(x, last_three=False, after_trunk=after_trunk if at == "trunk" else None, packed_tail=True)
        if at == "trunk":
            after_trunk = None
        if at == "ups1" and after_trunk is not None:
            after_trunk(); after_trunk = None
        if isinstance(f2, ops.PackedSplit):
            # the phase-decomposed route (csrc/sr_conv_upsp.h): trunk -> postblock.1 -> postblock.4 -> tail, packed-split all the way
            if not ops.ups_phase_supported(f2, four.weight):
                raise RuntimeError("run_network: postblock.4 does not take the packed-split tensor postblock.1 produced")
            f4 = ops.conv3x3_ups_phase(f2, four.weight, four.bias, act='relu')
        elif ops.packed_supported(f2, four.weight, True):
            # postblock.4 writes its output packed-split (already the (hi, lo') units postblock.6 multiplies), the tail
            # stages them by LDS-DMA: no conversion on the way in, no LDS transposition on the way out
            f4 = ops.conv3x3_split_packed(f2, four.weight, four.bias, act='relu', upsample2x=True)
        else:
            f4 = ops.conv3x3(f2, four.weight, four.bias, act='relu', upsample2x=True)
    elif tail:
        f4 = net.forward_features(x, last_two=False, after_trunk=after_trunk)
        after_trunk = None
    if f4 is not None and after_trunk is not None and prefetch_point != "tail":
        after_trunk(); after_trunk = None                # "ups2": beside the fused tail
    if isinstance(f4, ops.PackedSplit) or (f4 is not None and ops.tail_supported(f4, six.weight, last.weight)):
        # postblock.6, postblock.8 and the frame's finishing in two launches; the 64-channel 1080p tensor between the
        # two convolutions never goes to memory (csrc/sr_conv_tail.hip)
        raw, rgb = ops.tail_conv_finish(f4, six.weight, six.bias, last.weight, last.bias, x, shading, out=out)
        model._isr_static_out = out is not None          # the fused tail wrote straight into the caller's buffers
        out = None
    elif f4 is not None:
        f6 = ops.conv3x3(f4, six.weight, six.bias, act='relu')
        raw, rgb = ops.final_conv_finish(f6, last.weight, last.bias, x, shading)
    elif last.weight.shape[0] == 6:
        # the last layer's epilogue finishes the frame (one launch, no [6,4h,4w] round trip)
        raw, rgb = ops.final_conv_finish(net.forward_features(x, last_layer=False, after_trunk=after_trunk), last.weight, last.bias, x, shading)
        after_trunk = None
    else:
        raw, rgb = ops.finish_frame(net.forward_features(x, after_trunk=after_trunk), x, shading)
        after_trunk = None
    if after_trunk is not None:
        after_trunk()
    if out is not None:          # a route that allocates its own outputs: the caller's static buffers were not written
        model._isr_static_out = False
        out[0].copy_(raw); out[1].copy_(rgb)
        raw, rgb = out
    return raw, rgb


def fused_path_ok(model, upscale=4):
    """The fused frame kernels take this network: 4x, residual reconstruction over the five input channels."""
    net = model.model
    return upscale == 4 and getattr(net, 'recon_type', None) == 'residual' \
        and getattr(net, 'channel_mask', None) is not None and len(net.channel_mask) == 5


class SuperResolutionPipeline:
    def __init__(self, renderer, model, shading, low_res, upscale=4, temporal=True, device="cuda", fused=True, graph=None):
        """``graph`` (default: ISR_FRAME_GRAPH=1): steady-state frames -- temporal sequence, next frame's camera known -- replay as
        ONE HIP graph per frame (``_frame_graph``).  The returned tensors are then the pipeline's static output buffers: valid
        until the frame after next overwrites them."""
        self.renderer = renderer
        self.model = model            # inference.LoadedModel
        self.shading = shading
        self.low_w, self.low_h = low_res
        self.upscale = upscale
        self.temporal = temporal
        self.device = device
        self.gbuffer = torch.empty((self.low_h, self.low_w, 12), dtype=torch.float32, device=device)
        # render(t+1) || SR(t): the ray-marcher of the NEXT frame runs on a side stream while the
        # MFMA-bound network of the current one owns the main stream (the renderer does not depend on
        # the network's output).  Two G-buffers, two events per buffer.
        self._gbuffers = [self.gbuffer, torch.empty_like(self.gbuffer)]
        # the hole-filled flow of a prefetched frame is computed on the render stream as well (it only needs the G-buffer)
        self._flows = [torch.empty((1, 2, self.low_h, self.low_w), dtype=torch.float32, device=device) for _ in range(2)]
        self._flow_ready = [False, False]
        # (ISR_SIDE_PRIORITY: experiment -- a stream of another priority class is served by another hardware queue than the main stream)
        self._render_stream = torch.cuda.Stream(device=device, priority=int(os.environ.get("ISR_SIDE_PRIORITY", "0"))) if str(device).startswith("cuda") else None
        self._ready = [torch.cuda.Event(), torch.cuda.Event()] if self._render_stream else None
        self._consumed = [torch.cuda.Event(), torch.cuda.Event()] if self._render_stream else None
        self._frame_start = torch.cuda.Event() if self._render_stream else None
        self._slot = 0
        self._prefetched = None           # (origin tuple, slot) of a render already in flight
        self._displayed = None            # origin of the last frame handed to the network: the flow reference
        # a prefetched frame is rendered by the 128-register ray-marcher (kernel variant 2) with one wave per
        # SIMD, so that it sits beside the conv waves instead of displacing them (csrc/iso_kernels.hip)
        self.side_waves = 4 * torch.cuda.get_device_properties(device).multi_processor_count if self._render_stream else 0
        # kernel variant of the render that runs under the network.  None = the foreground variant: since the flat traversal
        # (0.21 ms alone) the plain kernel beside the network gave 400 frames/s in round 2; the capped 128-register kernel (2), which
        # round 1's 0.45 ms nested-loop kernel needed, gave 384 (tools/lab/side_sweep.sh)
        # (round 3) A ray-march wave shares a SIMD's 512 registers with the convolution waves.  Beside the TILE upsampling kernel (two
        # waves of 184) the 124-register one-sample traversal (variant 5) fits where the 168-register default takes the place of one of
        # them: 503-508 against 491-497 frames/s.  Beside the three-workgroups-per-CU upsampling kernel (3 x 168, the default since)
        # nothing fits either way and the faster default variant wins: 505-510 against 502-504.  ISR_SIDE_VARIANT picks one (0 / unset:
        # the foreground variant).
        self.side_variant = int(os.environ.get("ISR_SIDE_VARIANT", "0"))
        if self.side_variant == 0:
            self.side_variant = None
        self.prefetch_after_trunk = os.environ.get("ISR_PREFETCH_AFTER_TRUNK", "1") != "0"     # measured: 516-519 against 497-505 frames/s with the render beside the trunk
        # where in the network the next frame's render is released (experiment switch): trunk | ups1 | ups2 | tail = when the trunk,
        # the first / second upsampling layer, or everything but the last launch has ended
        self.prefetch_point = os.environ.get("ISR_PREFETCH_POINT", "trunk")
        # the flow hole filling of the prefetched frame: on the side stream behind its render (1), or on the main stream in front of the
        # next frame's input assembly (0)
        self.flow_fill_on_side = os.environ.get("ISR_FLOW_FILL_ON_SIDE", "1") != "0"
        self.flow_fill_threads = int(os.environ.get("ISR_FLOW_FILL_THREADS", "256"))      # workgroup size of its two full-grid passes (1024: +0.6 % on one box, nothing on another and one 2.4 ms outlier run: a 1024-thread workgroup has to find 16 free wave slots on ONE CU beside the convolutions)
        self._trunk_done = torch.cuda.Event() if torch.cuda.is_available() else None
        self.previous = None
        self.foreground_variant = 0       # kernel variant of frames rendered on the main stream
        # fused=True: input assembly and frame finishing run as two HIP kernels (ops.assemble_input /
        # ops.finish_frame); fused=False: the module-level PyTorch path (LoadedModel.inference etc.)
        self.fused = fused and fused_path_ok(model, upscale)
        # ---- the frame as one HIP graph ------------------------------------------------------------------------
        self.graph = (os.environ.get("ISR_FRAME_GRAPH", "0") == "1") if graph is None else bool(graph)
        self.graph = self.graph and self.fused and self._render_stream is not None
        self._graphs = [None, None]       # per G-buffer slot: the frame whose G-buffer sits in slot s (and renders the next one into s ^ 1)
        self._graph_sig = None
        self._graph_pool = None
        self._static_version = 0
        self.graph_replays = 0
        if self.graph:
            H, W = upscale * self.low_h, upscale * self.low_w
            self._out_raw = [torch.empty((1, 6, H, W), dtype=torch.float32, device=device) for _ in range(2)]
            self._out_rgb = [torch.empty((1, 3, H, W), dtype=torch.float32, device=device) for _ in range(2)]
            self._cam_blocks = [torch.zeros(renderer.frame_block_bytes(), dtype=torch.uint8, device=device) for _ in range(2)]
            self._graph_stream = torch.cuda.Stream(device=device)
            self._fork, self._join = torch.cuda.Event(), torch.cuda.Event()
        self.set_static(fov=shading.get_fov(), isovalue=0.5)

    def set_static(self, fov, isovalue, lookat=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0)):
        r = self.renderer
        self._static_version = getattr(self, "_static_version", 0) + 1          # captured frames hold these parameters: capture again
        self._lookat = tuple(float(v) for v in fmt3(lookat).split(","))
        r.send_command("cameraLookAt", fmt3(lookat))
        r.send_command("cameraUp", fmt3(up))
        r.send_command("cameraFoV", "%.3f" % fov)
        r.send_command("isovalue", "%5.3f" % float(isovalue))
        r.send_command("aoradius", "%5.3f" % 0.01)
        r.send_command("aosamples", "0")       # SR mode renders without AO (mainGUI.py:690-691)
        r.send_command("resolution", "%d,%d" % (self.low_w, self.low_h))
        r.send_command("viewport", "%d,%d,%d,%d" % (0, 0, self.low_w, self.low_h))

    def reset(self, flush=True):
        """Start a new temporal sequence.  ``flush=True`` (default) BLOCKS the host until the in-flight frame's guard words have landed
        (one event synchronisation) and MAY RAISE -- after the sequence state has been cleared, so the pipeline is usable either way;
        an interactive viewer that resets on every camera jump passes ``flush=False``: no wait, no exception here, the last frame's
        words are looked at by the next frame's ``guards_poll`` instead (they stay published).  Also the END of the sequence before: the guards watch every frame one frame late
        (``ops.guards_publish`` at a frame's end, ``ops.guards_poll`` at the next frame's start), so the sequence's LAST frame is
        looked at here -- a dataflow-trunk / flow-fill launch of that frame that timed out raises RuntimeError from this call (the
        frame it produced is incomplete), a layer that came close to the split operands' range is routed to the exact kernels from
        the next sequence on.  Offline renderers call ``reset()`` or ``close()`` after their last frame."""
        self.previous = None
        self._drop_prefetched()
        if flush and self.fused and str(self.device).startswith("cuda"):
            ops.guards_flush(self.device)

    def close(self):
        """End of use: the last frame's guard look (see ``reset``) and the captured frames dropped."""
        try:
            self.reset()
        finally:
            self._graphs = [None, None]
            self._graph_sig = None

    def _drop_prefetched(self):
        """Forget a frame that was rendered ahead but is not going to be displayed.  The renderer's "last camera" is
        already that frame's camera; the flow of whatever is rendered next must be measured against the last frame
        that was DISPLAYED, so the reference is put back (``isoSetLastCamera``)."""
        if self._prefetched is not None:
            self._prefetched = None
            if self._displayed is not None:
                self.renderer.set_last_camera(self._displayed, self._lookat)

    def prefetch(self, origin, after=None):
        """Start rendering the G-buffer of ``origin`` on the side stream (used by ``frame(..., next_origin=)``).
        ``after``: an event the render waits for on top of the buffer hand-over (e.g. the end of the trunk)."""
        slot = self._slot ^ 1
        rs = self._render_stream
        if after is not None:
            rs.wait_event(after)
        rs.wait_event(self._consumed[slot])              # the network has finished reading that buffer ...
        rs.wait_event(self._frame_start)                 # ... and so has anything the caller enqueued before this frame()
                                                         # (``pipe.gbuffer`` stays valid until the next frame() call)
        self.renderer.send_command("cameraOrigin", fmt3(origin))
        if self.side_variant is not None:
            self.renderer.set_kernel_variant(self.side_variant)
            self.renderer.set_wave_cap(self.side_waves)
        self.renderer.render_async(self._gbuffers[slot], rs)
        self.renderer.set_kernel_variant(self.foreground_variant)
        self._flow_ready[slot] = self.fused and self.temporal and self.flow_fill_on_side
        if self._flow_ready[slot]:
            ops.fill_flow_gbuffer(self._gbuffers[slot], out=self._flows[slot], stream=rs, threads=self.flow_fill_threads)
        self._ready[slot].record(rs)
        self._prefetched = (tuple(origin), slot)

    def _acquire_gbuffer(self, origin):
        """G-buffer of ``origin`` on the current stream: the prefetched one if it matches, else rendered now."""
        cur = torch.cuda.current_stream()
        if self._prefetched is not None and self._prefetched[0] == tuple(origin):
            slot = self._prefetched[1]
            cur.wait_event(self._ready[slot])
            self._prefetched = None
        else:
            self._drop_prefetched()       # ``next_origin`` of the previous call was not honoured: that frame is discarded
            slot = self._slot
            self._flow_ready[slot] = False
            self.renderer.send_command("cameraOrigin", fmt3(origin))
            self.renderer.render_async(self._gbuffers[slot], cur)
        self._slot = slot
        self.gbuffer = self._gbuffers[slot]
        self._displayed = tuple(float(v) for v in fmt3(origin).split(","))   # what the renderer parsed
        return self.gbuffer

    def render_low(self, origin):
        self._drop_prefetched()
        self._displayed = tuple(float(v) for v in fmt3(origin).split(","))
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

    def _assemble(self, gbuffer, flow, prev):
        """The network input of a frame: packed-split straight into the dataflow trunk's workspace where that launch will take it
        (ops.assemble_input_packed), fp32 planes otherwise."""
        net = self.model.model
        if self.fused and hasattr(net, 'trunk_convs'):
            x = ops.assemble_input_packed(gbuffer, flow, prev, net.trunk_convs(), self.model.initial_image_mode, self.model.inverse_ao)
            if x is not None:
                return x
        return ops.assemble_input(gbuffer, flow, prev, self.model.initial_image_mode, self.model.inverse_ao)

    def _network(self, x, after_trunk=None, out=None):
        """Network input [1,101,h,w] -> (raw [1,6,4h,4w] clamped / normalised, rgb [1,3,4h,4w])."""
        return run_network(self.model, self.shading, x, after_trunk=after_trunk, prefetch_point=self.prefetch_point, out=out)

    # ---- one HIP graph per frame ---------------------------------------------------------------------------------------------
    # Steady state of a temporal sequence whose next camera is known: frame t's G-buffer and hole-filled flow sit in slot s (rendered
    # under frame t - 1), its "previous" is frame t - 1's output in the static buffer s ^ 1.  Everything frame t launches -- input
    # assembly, dataflow trunk, the fork to the side stream (render of frame t + 1 from the camera block + its flow fill), the
    # upsampling layers, fused tail, guard-word mirror, the join -- has fixed arguments and is captured ONCE per slot; per frame the
    # host sends the next camera, refreshes the camera block (one small launch on this stream) and replays.  Kernel boundaries
    # inside a graph need no host: ~2 us instead of the 4-15 us of eager launches with event records between them.
    def _graph_signature(self):
        """Everything a captured frame has baked in besides its static buffers: shading constants, the model object, the weight
        images (``_images_epoch`` + every parameter's (version, address): an in-place update makes the eager path build new images
        and drop the ones a graph points at), the renderer's static parameters, and the ROUTING -- which kernel forms a launch takes
        (``ops.TRUNK_DATAFLOW``, ``ops.FLOW_FILL_ONE``, the device-shared hint) and ``ops.routing_epoch()``, which moves whenever a
        guard failure switched a form off, re-zeroed a workspace or ``range_reset`` handed the guard words out anew."""
        sh = self.shading
        weights = tuple((p._version, p.data_ptr()) for p in self.model.model.parameters())      # (an address can change without the version moving: every frame, ~50 pairs; graph mode is opt-in)
        return (tuple(sh.packed_parameters()), int(sh._specular_exponent), float(sh._ao), bool(self.model.inverse_ao), bool(sh.enable_specular),
                self.model.initial_image_mode, id(self.model.model), ops._images_epoch, self._static_version, self.flow_fill_threads,
                ops.TRUNK_DATAFLOW, ops.FLOW_FILL_ONE, ops.DEVICE_SHARED, ops.SPLIT_F16, ops.FAST_F16, ops.UPS_PHASE, ops.ASSEMBLE_PACKED, ops.routing_epoch(), weights)

    def _graph_ready(self, origin, next_origin):
        if not (self.graph and next_origin is not None and self.temporal and self.previous is not None and self._prefetched is not None):
            return False
        slot = self._prefetched[1]
        return (self._prefetched[0] == tuple(origin) and self._flow_ready[slot] and self.previous is self._out_raw[slot ^ 1]
                and getattr(self.model, "_isr_static_out", False) and self.side_variant is None and self.prefetch_after_trunk
                and self.prefetch_point == "trunk" and self.flow_fill_on_side and not ops.any_hot(self.device)
                and not ops.profile_is_on() and self.foreground_variant == 0)

    def _graph_body(self, slot):
        """Frame of slot ``slot`` on the current stream (eagerly once as the warm-up, then under capture)."""
        nxt = slot ^ 1
        cur, rs = torch.cuda.current_stream(), self._render_stream
        x = self._assemble(self._gbuffers[slot], self._flows[slot], self._out_raw[nxt])

        def start_next():
            self._fork.record(cur)
            rs.wait_event(self._fork)
            self.renderer.render_from_block(self._gbuffers[nxt], self._cam_blocks[nxt], rs)
            ops.fill_flow_gbuffer(self._gbuffers[nxt], out=self._flows[nxt], stream=rs, threads=self.flow_fill_threads)
            self._join.record(rs)
        run_network(self.model, self.shading, x, after_trunk=start_next, prefetch_point="trunk", out=(self._out_raw[slot], self._out_rgb[slot]))
        ops.guards_publish(self.device, record=False)
        cur.wait_event(self._join)

    def _frame_graph(self, origin, next_origin):
        slot = self._prefetched[1]
        nxt = slot ^ 1
        cur = torch.cuda.current_stream()
        try:
            ops.guards_poll(self.device)
        except RuntimeError:
            # a spin kernel of the previous (replayed) frame timed out: that form is now off (ops.routing_epoch moved).  The captured
            # frames hold exactly that launch -- drop them BEFORE the error leaves, so that whoever catches it and goes on gets
            # frames on the fallback forms (eagerly first, then captured again)
            self._graphs, self._graph_sig = [None, None], None
            raise
        sig = self._graph_signature()
        if sig != self._graph_sig:
            self._graphs, self._graph_sig = [None, None], sig
        self.renderer.send_command("cameraOrigin", fmt3(next_origin))
        self.renderer.write_frame_block(self._cam_blocks[nxt], cur)      # next frame's camera (flow reference: this frame's), ordered before the replay
        cur.wait_event(self._ready[slot])                                # (an EAGER render of this slot, if that is where it came from)
        if self._graphs[slot] is None:
            gs = self._graph_stream
            gs.wait_stream(cur)
            with torch.cuda.stream(gs):
                self._graph_body(slot)                                   # this frame, and the warm-up of everything the capture touches
            cur.wait_stream(gs)
            g = torch.cuda.CUDAGraph()
            kw = {"pool": self._graph_pool} if self._graph_pool is not None else {}
            with ops.graph_capture(g, stream=gs, **kw):
                self._graph_body(slot)
            if self._graph_pool is None:
                self._graph_pool = g.pool()
            self._graphs[slot] = g
        else:
            self._graphs[slot].replay()
            self.graph_replays += 1
        ops.guards_mark(self.device)
        self._slot = slot
        self.gbuffer = self._gbuffers[slot]
        self._displayed = tuple(float(v) for v in fmt3(origin).split(","))
        self._prefetched = (tuple(next_origin), nxt)
        self._flow_ready[nxt] = True
        self.previous = self._out_raw[slot]
        return self._out_rgb[slot], self._out_raw[slot]

    def frame_fused(self, origin, next_origin=None):
        if self._graph_ready(origin, next_origin):
            with torch.no_grad():
                return self._frame_graph(origin, next_origin)
        with torch.no_grad():
            # what the previous frame's kernels reported (a plain read of pinned memory): a layer that came close to the split
            # operands' range is routed to the exact kernels from THIS frame on; a dataflow-trunk launch that timed out raises
            ops.guards_poll(self.device)
            if next_origin is not None:
                self._frame_start.record(torch.cuda.current_stream())
            g = self._acquire_gbuffer(origin)
            start_next = None
            if next_origin is not None and self.prefetch_after_trunk:
                # the next frame's render starts when the trunk has ENDED (an event): the dataflow trunk is ONE round of workgroups, one
                # per CU, that must all be resident and wait for each other tile by tile -- a guest that delays one tile delays its
                # neighbours -- while the 1080p kernels behind it run many rounds and absorb a guest
                def start_next():
                    self._trunk_done.record(torch.cuda.current_stream())
                    self.prefetch(next_origin, after=self._trunk_done)
            elif next_origin is not None:
                # the next frame's render goes first.  With the capped 128-register kernel (side_variant = 2) this stream then
                # waits until its waves sit one per SIMD on the momentarily idle GPU (a no-op for the other variants).
                self.prefetch(next_origin)
                self.renderer.gate_resident(torch.cuda.current_stream())
            prev = self.previous if self.temporal else None
            flow = None
            if prev is not None:
                flow = self._flows[self._slot] if self._flow_ready[self._slot] else ops.fill_flow_gbuffer(g)
            x = self._assemble(g, flow, prev)
            self._consumed[self._slot].record(torch.cuda.current_stream())   # G-buffer no longer needed
            out = (self._out_raw[self._slot], self._out_rgb[self._slot]) if self.graph else None
            raw, rgb = self._network(x, after_trunk=start_next, out=out)
            if ops.range_check_due(x.device):
                # range guard (ops.RANGE_GUARD), a model's FIRST frame: a layer's output came close to the fp16 range of the split
                # operands -- its consumers run on the exact fp32 kernels from now on; this frame is computed again with that routing.
                # (Again, because the fused launches only say THAT something inside them was hot: the per-layer pass that replaces
                # them says where.)  Every later frame is watched one frame late: guards_publish here, guards_poll at the next start.
                for _ in range(4):
                    if not ops.refresh_range_flags(x.device):
                        break
                    if getattr(x, '_isr_prepacked', None) is not None:
                        # the per-layer pass reads the fp32 planes: assemble them (the packed form wrote channels 0 .. 4 only)
                        x = ops.assemble_input(g, flow, prev, self.model.initial_image_mode, self.model.inverse_ao)
                        self._consumed[self._slot].record(torch.cuda.current_stream())
                    raw, rgb = self._network(x, out=out)
            ops.guards_publish(x.device)
            self.previous = raw
        return rgb, raw

    def frame(self, origin, next_origin=None):
        """origin: camera position; ``next_origin`` (optional) starts the next frame's ray-march on a side
        stream so that it overlaps this frame's network.  Returns (rgb [1,3,4h,4w], raw [1,6,4h,4w])."""
        if self.fused:
            return self.frame_fused(origin, next_origin)
        with torch.no_grad():
            low = self.render_low(origin)
            raw = self.superresolve(low)
            self.shading.inverse_ao = self.model.inverse_ao
            rgb = self.shading(raw)
        return rgb, raw
