"""``losses``: the package surface of ``SuperresolutionNetwork/losses/__init__.py`` -- ``LossBuilder``, ``LossNetUnshaded`` (the hot path's
criterion, ``lossnet_unshaded.py``) and ``LossNet``."""
from .lossnet_unshaded import LossBuilder, LossNetUnshaded


class LossNet:
    """The loss of the SHADED networks (``SuperresolutionNetwork/losses/lossnet.py``, the older ``mainVideo.py`` trainer): part of the
    reference's package surface (``losses/__init__.py:2``) but not on the accelerated hot path (SURVEY.md section 2, rows 12 and 17:
    superseded trainer, VGG perceptual terms that need a weight download).  An explicit out-of-scope marker: code written against the
    reference fails with a clear message rather than an ``ImportError``."""

    def __init__(self, *args, **kwargs):
        raise NotImplementedError(
            "losses.LossNet (shaded RGB networks of mainVideo.py) is outside the accelerated hot path; "
            "the unshaded temporal networks of mainVideoUnshaded.py use losses.LossNetUnshaded")
