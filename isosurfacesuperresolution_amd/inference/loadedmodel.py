"""``LoadedModel``: a trained generator plus the per-frame input assembly of the viewer.

Public surface of ``SuperresolutionNetwork/inference/loadedmodel.py`` (attributes ``name, model,
unshaded, inverse_ao, initial_image_mode, input_channels``; ``inference(current_low, prev_high)``)
for the unshaded networks of the hot path.  Differences, all additive or forced by the platform:

* flow hole filling happens on the GPU (``flowfill.fill_flow``) instead of a CPU OpenCV call
  (``loadedmodel.py:77-82``) -- no device round trip inside the frame;
* ``LoadedModel.from_model`` wraps an in-memory network (no checkpoint files ship with the
  reference, README.md:68);
* reference checkpoints pickle whole ``models.enhancenet.EnhanceNet`` objects
  (``mainVideoUnshaded.py:801``), so they cannot be read with ``weights_only=True``.  They are read through a
  RESTRICTED unpickler instead of a plain ``torch.load(weights_only=False)``: ``models.*`` / ``utils.*`` resolve to
  this package's modules (whatever top-level ``models`` the host process may have), tensors / storages / containers /
  optimizer and scheduler state resolve to torch and the standard library, and anything else -- the arbitrary
  callables a malicious pickle would name -- is refused.  A checkpoint is still code-adjacent data: load files you trust.
"""
import importlib
import os.path
import pickle

import torch


class _CheckpointPickle:
    """``pickle_module`` for ``torch.load``: an allow-list ``find_class``."""
    __name__ = "isosurfacesuperresolution_amd.checkpoint_pickle"
    _ALIASES = {"models": "isosurfacesuperresolution_amd.models", "utils": "isosurfacesuperresolution_amd.utils",
                "losses": "isosurfacesuperresolution_amd.losses"}
    _ALLOWED_ROOTS = ("torch", "collections", "argparse", "numpy", "isosurfacesuperresolution_amd")
    _ALLOWED_BUILTINS = {"set", "frozenset", "dict", "list", "tuple", "slice", "range", "complex", "int", "float", "bool",
                         "str", "bytes", "bytearray", "object", "getattr"}

    class Unpickler(pickle.Unpickler):
        def find_class(self, module, name):
            cls = _CheckpointPickle
            root = module.split(".")[0]
            if root in cls._ALIASES:
                module = cls._ALIASES[root] + module[len(root):]
            elif root in ("builtins", "__builtin__"):
                if name not in cls._ALLOWED_BUILTINS or name == "getattr":
                    raise pickle.UnpicklingError("checkpoint refers to builtins.%s: refused" % name)
                return super().find_class(module, name)
            elif root not in cls._ALLOWED_ROOTS:
                raise pickle.UnpicklingError("checkpoint refers to %s.%s: only torch / numpy / this package's "
                                             "models, utils and losses are loaded" % (module, name))
            if root == "torch" and (name in ("load", "save") or module.startswith(("torch.utils.cpp_extension", "torch.hub",
                                                                                      "torch.distributed", "torch.multiprocessing"))):
                raise pickle.UnpicklingError("checkpoint refers to %s.%s: refused" % (module, name))
            mod = importlib.import_module(module)
            obj = mod
            for part in name.split("."):
                obj = getattr(obj, part)
            return obj

    load = staticmethod(pickle.load)
    dump = staticmethod(pickle.dump)
    Pickler = pickle.Pickler

from ..models import VideoTools
from ..utils import initialImage
from .flowfill import fill_flow


class LoadedModel:
    def __init__(self, name, device, upscale_factor):
        self.name = os.path.splitext(os.path.basename(name))[0]
        self.device = device
        self.upscale_factor = upscale_factor
        checkpoint = torch.load(name, map_location=device, weights_only=False, pickle_module=_CheckpointPickle)
        parameters = checkpoint.get('parameters', dict())
        if not isinstance(parameters, dict):
            parameters = vars(parameters)
        self._setup(checkpoint['model'], parameters)

    @classmethod
    def from_model(cls, model, device, upscale_factor=4, parameters=None, name="model"):
        self = cls.__new__(cls)
        self.name = name
        self.device = device
        self.upscale_factor = upscale_factor
        self._setup(model, dict(parameters or {}))
        return self

    def _setup(self, model, parameters):
        self.parameters = parameters
        self.model = model
        self.model.to(self.device)
        self.model.train(False)
        first = self.model
        while True:   # first leaf module = first convolution (loadedmodel.py:26-34)
            children = list(first.children())
            if not children:
                break
            first = children[0]
        self.input_channels = first.in_channels
        r2 = self.upscale_factor ** 2
        self.unshaded = self.input_channels == 5 + 6 * r2 or bool(self.parameters.get('unshaded', False))
        if not self.unshaded:
            raise NotImplementedError("only the unshaded (mask/normal/depth/ao) networks are on the hot path")
        self.initial_image_mode = self.parameters.get('initialImage', 'input')
        self.inverse_ao = self.parameters.get('aoInverted', False)

    def inference(self, current_low, prev_high):
        """current_low [1,12,h,w] renderer output (r,g,b,mask,nx,ny,nz,depth,fx,fy,ao,shadow);
        prev_high [1,6,4h,4w] previous network output or None.  Returns [1,6,4h,4w]."""
        with torch.no_grad():
            mask = current_low[:, 3:4]
            inp = torch.cat((mask * 2 - 1, current_low[:, 4:8]), dim=1)
            if prev_high is None:
                previous_warped = initialImage(inp, 6, self.initial_image_mode, self.inverse_ao,
                                               self.upscale_factor).to(self.device)
            else:
                flow = fill_flow(current_low[:, 8:10], mask != 0)
                previous_warped = VideoTools.warp_upscale(prev_high.to(self.device), flow,
                                                          self.upscale_factor, special_mask=True)
            flat = VideoTools.flatten_high(previous_warped, self.upscale_factor)
            prediction, _ = self.model(torch.cat((inp, flat), dim=1))
        return prediction
