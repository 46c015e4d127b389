"""ctypes binding of libGPURendererDirect.so.

Mirrors ``SuperresolutionNetwork/inference/renderer.py:78-117`` (class ``DirectRenderer``:
``load``, ``send_command``, ``render_direct``, ``get_time``, ``close``), ``Material``
(``:9-15``) and the pipe flavour ``Renderer`` (``:16-76``: ``send_command``, ``render``,
``read_image``, ``get_time``, ``close``) -- the latter over the same in-process library instead of
a child process.  Additive: ``load_dense`` (dense numpy / device tensor volumes), ``render_async``,
return codes are surfaced instead of being dropped.
"""
import collections
import ctypes
import os

import torch

from .. import _native
from .camera import Camera


class Material:
    """inference/renderer.py:9-15"""

    def __init__(self, iso):
        self.isovalue = iso
        self.diffuseColor = [0.7, 0.2, 0.2]
        self.specularColor = [0.1, 0.1, 0.1]
        self.specularExponent = 16
        self.light = 'camera'


class DirectRenderer:
    def __init__(self, renderer=None):
        """``renderer``: path of libGPURendererDirect.so (default: the in-tree build)."""
        if renderer is None:
            renderer = _native.RENDERER_LIB
        assert isinstance(renderer, str)
        if renderer == _native.RENDERER_LIB:
            self.lib = _native.load(renderer)
        else:
            assert os.path.exists(renderer)
            _native._preload_hip_runtime()
            self.lib = ctypes.cdll.LoadLibrary(renderer)
        self.time = 0
        lib = self.lib
        lib.initGVDB.argtypes = []
        lib.initGVDB.restype = ctypes.c_int
        lib.loadGrid.argtypes = [ctypes.c_char_p]
        lib.loadGrid.restype = ctypes.c_int
        lib.setParameter.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
        lib.setParameter.restype = ctypes.c_int
        lib.render.argtypes = [ctypes.c_ulonglong]
        lib.render.restype = ctypes.c_float
        lib.isoLoadDenseHost.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        lib.isoLoadDenseHost.restype = ctypes.c_int
        lib.isoLoadDenseDevice.argtypes = [ctypes.c_ulonglong, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        lib.isoLoadDenseDevice.restype = ctypes.c_int
        lib.isoLoadDenseTileHost.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_float] + [ctypes.c_void_p] * 2
        lib.isoLoadDenseTileHost.restype = ctypes.c_int
        lib.isoLoadDenseTileDevice.argtypes = [ctypes.c_ulonglong, ctypes.c_int, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_float] + [ctypes.c_void_p] * 2
        lib.isoLoadDenseTileDevice.restype = ctypes.c_int
        lib.isoRenderAsync.argtypes = [ctypes.c_ulonglong, ctypes.c_void_p]
        lib.isoRenderAsync.restype = ctypes.c_int
        lib.isoGetVolumeInfo.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        lib.isoGetVolumeInfo.restype = ctypes.c_int
        lib.isoSetKernelVariant.argtypes = [ctypes.c_int]
        lib.isoSetKernelVariant.restype = ctypes.c_int
        lib.isoGateResident.argtypes = [ctypes.c_void_p, ctypes.c_int]
        lib.isoGateResident.restype = ctypes.c_int
        lib.isoSetWaveCap.argtypes = [ctypes.c_int]
        lib.isoSetWaveCap.restype = ctypes.c_int
        lib.isoSetTileOrderMode.argtypes = [ctypes.c_int]
        lib.isoSetTileOrderMode.restype = ctypes.c_int
        lib.isoSetHitStateBuffer.argtypes = [ctypes.c_ulonglong]
        lib.isoSetHitStateBuffer.restype = ctypes.c_int
        lib.isoAoDistancesAsync.argtypes = [ctypes.c_ulonglong, ctypes.c_ulonglong, ctypes.c_ulonglong, ctypes.c_void_p]
        lib.isoAoDistancesAsync.restype = ctypes.c_int
        lib.isoAoFinishAsync.argtypes = [ctypes.c_ulonglong, ctypes.c_ulonglong, ctypes.c_void_p]
        lib.isoAoFinishAsync.restype = ctypes.c_int
        lib.isoProfileEnable.argtypes = [ctypes.c_int]
        lib.isoProfileEnable.restype = ctypes.c_int
        lib.isoProfileCount.argtypes = []
        lib.isoProfileCount.restype = ctypes.c_int
        lib.isoProfileGet.argtypes = [ctypes.c_int, ctypes.c_void_p]
        lib.isoProfileGet.restype = ctypes.c_int
        lib.isoShutdown.argtypes = []
        lib.isoShutdown.restype = None
        lib.isoFrameBlockBytes.argtypes = []; lib.isoFrameBlockBytes.restype = ctypes.c_int
        lib.isoWriteFrameBlockAsync.argtypes = [ctypes.c_ulonglong, ctypes.c_void_p]; lib.isoWriteFrameBlockAsync.restype = ctypes.c_int
        lib.isoRenderFromBlockAsync.argtypes = [ctypes.c_ulonglong, ctypes.c_ulonglong, ctypes.c_void_p]; lib.isoRenderFromBlockAsync.restype = ctypes.c_int
        lib.isoSetLastCamera.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        lib.isoSetLastCamera.restype = ctypes.c_int
        if lib.initGVDB() != 0:
            raise RuntimeError("initGVDB failed: no usable HIP device")

    def load(self, filename: str):
        return self.lib.loadGrid(ctypes.c_char_p(filename.encode("ascii")))

    def load_dense(self, volume):
        """Additive: dense fp32 volume [z][y][x] as a numpy array or a device tensor."""
        if hasattr(volume, "data_ptr"):
            assert volume.is_cuda and volume.is_contiguous() and volume.dim() == 3
            nz, ny, nx = volume.shape
            import torch
            torch.cuda.synchronize()
            rc = self.lib.isoLoadDenseDevice(ctypes.c_ulonglong(volume.data_ptr()), nx, ny, nz)
        else:
            import numpy as np
            volume = np.ascontiguousarray(volume, dtype=np.float32)
            nz, ny, nx = volume.shape
            rc = self.lib.isoLoadDenseHost(volume.ctypes.data, nx, ny, nz)
        if rc != 0:
            raise RuntimeError("loading the dense volume failed (rc=%d)" % rc)
        return rc

    def load_tile(self, tile):
        """Additive: one tile of a larger volume (see ``parallel_render.partition_volume``).  ``tile['data']``: a numpy array
        [z][y][x] (``isoLoadDenseTileHost``) or a float32 CUDA tensor of that shape (``isoLoadDenseTileDevice``: no host staging)."""
        import numpy as np
        i3 = lambda v: (ctypes.c_int * 3)(*[int(a) for a in v])
        meta = (i3(tile['origin']), i3(tile['gmin']), i3(tile['gmax']), ctypes.c_float(tile['gmaxval']), i3(tile['clip_lo']), i3(tile['clip_hi']))
        if torch.is_tensor(tile['data']) and tile['data'].is_cuda:
            data = tile['data'].to(torch.float32).contiguous()
            nz, ny, nx = data.shape
            rc = self.lib.isoLoadDenseTileDevice(ctypes.c_ulonglong(data.data_ptr()), nx, ny, nz, *meta)
            torch.cuda.synchronize()
            if rc != 0:
                raise RuntimeError("loading the volume tile failed (rc=%d)" % rc)
            return rc
        data = np.ascontiguousarray(tile['data'], dtype=np.float32)
        nz, ny, nx = data.shape
        rc = self.lib.isoLoadDenseTileHost(data.ctypes.data, nx, ny, nz, *meta)
        if rc != 0:
            raise RuntimeError("loading the volume tile failed (rc=%d)" % rc)
        return rc

    def send_command(self, cmd, value):
        assert isinstance(cmd, str)
        assert isinstance(value, str)
        return self.lib.setParameter(ctypes.c_char_p(cmd.encode("ascii")),
                                     ctypes.c_char_p(value.encode("ascii")))

    def render_direct(self, tensor):
        time = self.lib.render(ctypes.c_ulonglong(tensor.data_ptr()))
        self.time = float(time)
        return self.time

    def render_async(self, tensor, stream=None):
        """Additive: enqueue the frame on ``stream`` (a torch.cuda.Stream or None) without syncing."""
        handle = ctypes.c_void_p(stream.cuda_stream) if stream is not None else None
        rc = self.lib.isoRenderAsync(ctypes.c_ulonglong(tensor.data_ptr()), handle)
        if rc != 0:
            raise RuntimeError("isoRenderAsync failed")

    def frame_block_bytes(self):
        return int(self.lib.isoFrameBlockBytes())

    def write_frame_block(self, block, stream=None):
        """Additive: the per-frame camera block (a device uint8 tensor of ``frame_block_bytes()``) from the current parameters,
        written by a launch on ``stream``; the current camera becomes the flow reference as after a render."""
        handle = ctypes.c_void_p(stream.cuda_stream) if stream is not None else None
        if self.lib.isoWriteFrameBlockAsync(ctypes.c_ulonglong(block.data_ptr()), handle) != 0:
            raise RuntimeError("isoWriteFrameBlockAsync failed")

    def render_from_block(self, tensor, block, stream=None):
        """Additive: the SR-mode render (aosamples = 0) with the camera read from ``block`` -- the same launch every frame, so a
        captured HIP graph can replay it."""
        handle = ctypes.c_void_p(stream.cuda_stream) if stream is not None else None
        if self.lib.isoRenderFromBlockAsync(ctypes.c_ulonglong(tensor.data_ptr()), ctypes.c_ulonglong(block.data_ptr()), handle) != 0:
            raise RuntimeError("isoRenderFromBlockAsync failed")

    def volume_info(self):
        info = (ctypes.c_int * 12)()
        mx = ctypes.c_float()
        if self.lib.isoGetVolumeInfo(info, ctypes.byref(mx)) != 0:
            return None
        return {"dims": list(info[0:3]), "bricks": info[3], "leaves": info[4],
                "node_bbox_min": list(info[5:8]), "node_bbox_max": list(info[8:11]),
                "brick_mib": info[11], "max_value": mx.value}

    def profile_enable(self, on):
        """Additive: per-frame kernel timing carried on the dispatch packets (no extra stream ops)."""
        self.lib.isoProfileEnable(1 if on else 0)

    def profile_times_ms(self):
        ms = ctypes.c_float()
        out = []
        for i in range(self.lib.isoProfileCount()):
            if self.lib.isoProfileGet(i, ctypes.byref(ms)) != 0:
                raise RuntimeError("isoProfileGet failed")
            out.append(ms.value)
        return out

    def set_last_camera(self, origin, lookat=(0.0, 0.0, 0.0)):
        """Additive: see isoSetLastCamera (the flow reference after a frame rendered ahead was discarded)."""
        d3 = lambda v: (ctypes.c_double * 3)(*[float(a) for a in v])
        return self.lib.isoSetLastCamera(d3(origin), d3(lookat))

    def gate_resident(self, stream, timeout_us=100):
        """Additive: see isoGateResident.  ``stream``: a torch.cuda.Stream."""
        return self.lib.isoGateResident(ctypes.c_void_p(stream.cuda_stream), int(timeout_us))

    # ---- exact AO of a tiled volume (include/gpu_renderer_direct.h: isoSetHitStateBuffer ...) ------------------------------
    def set_hit_state_buffer(self, tensor):
        """tensor: [H, W, 6] float64 on the device (the renders export their AO ray set-up into it), or None to switch it off."""
        if tensor is not None:
            assert tensor.is_cuda and tensor.is_contiguous() and tensor.dtype.is_floating_point and tensor.element_size() == 8
        return self.lib.isoSetHitStateBuffer(ctypes.c_ulonglong(tensor.data_ptr() if tensor is not None else 0))

    def ao_distances(self, hit_state, gbuffer, dist, stream=None):
        """dist [H, W, aosamples] float64 <- distance of every hit pixel's AO rays to their first hit among THIS volume's leaves."""
        handle = ctypes.c_void_p(stream.cuda_stream) if stream is not None else None
        rc = self.lib.isoAoDistancesAsync(ctypes.c_ulonglong(hit_state.data_ptr()), ctypes.c_ulonglong(gbuffer.data_ptr()),
                                          ctypes.c_ulonglong(dist.data_ptr()), handle)
        if rc != 0:
            raise RuntimeError("isoAoDistancesAsync failed (aosamples must be > 0, a volume loaded)")

    def ao_finish(self, dist, gbuffer, stream=None):
        """gbuffer[..., 10] <- the ambient occlusion from the (tile-minimum) distances."""
        handle = ctypes.c_void_p(stream.cuda_stream) if stream is not None else None
        rc = self.lib.isoAoFinishAsync(ctypes.c_ulonglong(dist.data_ptr()), ctypes.c_ulonglong(gbuffer.data_ptr()), handle)
        if rc != 0:
            raise RuntimeError("isoAoFinishAsync failed")

    def set_tile_order_mode(self, mode):
        """Additive: see isoSetTileOrderMode (cost-ordered dispatch of the default kernel from the previous frame's tile costs)."""
        return self.lib.isoSetTileOrderMode(int(mode))

    def set_wave_cap(self, waves):
        """Additive: see isoSetWaveCap (variant 2, side-stream rendering under the SR network)."""
        return self.lib.isoSetWaveCap(int(waves))

    def set_kernel_variant(self, variant):
        return self.lib.isoSetKernelVariant(int(variant))

    def get_time(self):
        """Returns the time of the last render pass in seconds"""
        return self.time

    def close(self):
        pass  # No-op, as in the reference


class Renderer:
    """The reference's pipe-flavour renderer (``inference/renderer.py:16-76``; protocol of
    ``CPURenderer/CPURenderer.cpp:688-787``) with the same methods and the same command strings, served by the
    in-process HIP library instead of a ``CPURenderer.exe ... PIPE`` child: ``send_command("cmd=value\\n")`` or
    ``send_command("cmd", value)``, ``render()`` / ``send_command("render\\n")``, then
    ``read_image(resX, resY, channels=12) -> np.float32[12, resY, resX]`` (planar, the pipe's layout) and
    ``get_time()`` (the trailing float of a pipe frame).  This is what ``mainPSNR2_AllAngles.py:184-276`` drives.

    ``renderer``: path of libGPURendererDirect.so (anything that is not an existing file selects the in-tree
    build -- the reference passes the path of an EXE here).  ``inputfile``: a ``.vbx`` path, or (additive) a dense
    numpy volume ``[z][y][x]``.  ``backend`` (additive): an object with ``DirectRenderer``'s methods
    (``send_command`` / ``render_direct`` / ``load`` / ``load_dense``); tests drive the protocol on CPU tensors with it.
    As in pipe mode, ``resolution`` also resets the viewport to the whole image, unknown commands end the session
    (here: ``RuntimeError``), and every frame is synchronous."""

    def __init__(self, renderer, inputfile, material, camera, backend=None, device="cuda"):
        assert isinstance(renderer, str)
        assert isinstance(material, Material)
        assert isinstance(camera, Camera)
        if backend is None:
            backend = DirectRenderer(renderer if os.path.isfile(renderer) and renderer.endswith(".so") else None)
        self.backend = backend
        self.device = device
        if isinstance(inputfile, str):
            rc = backend.load(inputfile)
            if rc != 0:
                raise RuntimeError("Renderer: cannot load %r (rc=%d)" % (inputfile, rc))
        else:
            backend.load_dense(inputfile)
        self.resX, self.resY = int(camera.resX), int(camera.resY)
        f3 = lambda v: "%5.3f,%5.3f,%5.3f" % (v[0], v[1], v[2])
        # the EXE's command line (renderer.py:26-43)
        for cmd, value in (("resolution", "%d,%d" % (self.resX, self.resY)), ("cameraOrigin", f3(camera.getOrigin())),
                           ("cameraLookAt", f3(camera.getLookAt())), ("cameraUp", f3(camera.getUp())),
                           ("isovalue", str(material.isovalue)), ("unshaded", "0"), ("diffuse", f3(material.diffuseColor)),
                           ("specular", f3(material.specularColor)), ("exponent", str(material.specularExponent)),
                           ("light", material.light), ("aoradius", "0.01")):
            self._apply(cmd, value)
        self._frames = collections.deque()
        self._buffer = None
        self.time = 0
        self.closed = False

    def _apply(self, cmd, value):
        if self.backend.send_command(cmd, value) != 0:
            raise RuntimeError("Unknown command: %r (the reference's pipe renderer exits here)" % cmd)
        if cmd == "resolution":
            self.resX, self.resY = (int(v) for v in value.split(","))
            self.backend.send_command("viewport", "0,0,%d,%d" % (self.resX, self.resY))

    def send_command(self, cmd, value=None):
        if self.closed:
            raise RuntimeError("Renderer is closed")
        if value is not None:
            cmd = cmd + "=" + str(value) + "\n"
        for line in cmd.split("\n"):
            line = line.strip()
            if not line:
                continue
            if line == "exit":
                self.closed = True
            elif line == "render":
                self._render()
            else:
                name, sep, val = line.partition("=")
                if not sep:
                    raise RuntimeError("Unknown command format: %r" % line)
                self._apply(name, val)

    def _render(self):
        import torch
        if self._buffer is None or tuple(self._buffer.shape[:2]) != (self.resY, self.resX):
            self._buffer = torch.empty((self.resY, self.resX, 12), dtype=torch.float32, device=self.device)
        seconds = self.backend.render_direct(self._buffer)
        planar = self._buffer.permute(2, 0, 1).contiguous().cpu().numpy()
        self._frames.append((planar, float(seconds)))

    def render(self):
        self.send_command("render\n")

    def read_image(self, resX, resY, channels=12):
        if not self._frames:
            raise RuntimeError("read_image: no rendered frame is waiting (the pipe would block forever)")
        image, seconds = self._frames.popleft()
        if image.shape != (channels, resY, resX):
            raise RuntimeError("read_image(%d, %d, %d): the frame waiting in the pipe is %s" % (resX, resY, channels, image.shape))
        self.time = seconds
        return image

    def close(self):
        self.closed = True

    def get_time(self):
        """Returns the time of the last render pass in seconds"""
        return self.time
