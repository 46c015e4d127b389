"""MI355X-native isosurface ray-marching + temporal super-resolution.

Sub-packages mirror the reference's Python module API: ``models``, ``utils``, ``losses``,
``inference``; ``ops`` binds the HIP kernels, ``pipeline`` is the per-frame driver, ``train`` the
training step and its data-parallel form, ``volumes`` the synthetic stand-in data.
"""
from . import volumes  # noqa: F401
from . import utils, models, losses, inference  # noqa: F401
from . import ops, pipeline, train  # noqa: F401
