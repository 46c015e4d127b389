"""``losses.LossNet`` -- the loss of the SHADED networks (``SuperresolutionNetwork/losses/lossnet.py``, used by the
older ``mainVideo.py`` trainer).  The name is part of the reference's ``losses`` package surface
(``losses/__init__.py:2``), but the shaded networks are not on the accelerated hot path (SURVEY.md section 2,
rows 12 and 17: superseded trainer, VGG perceptual terms that need a weight download).  It is kept as an explicit
out-of-scope marker so that code written against the reference fails with a clear message rather than an
``ImportError``; the unshaded hot path uses ``LossNetUnshaded``."""


class LossNet:
    def __init__(self, *args, **kwargs):
        raise NotImplementedError(
            "losses.LossNet (shaded RGB networks of mainVideo.py) is outside the accelerated hot path; "
            "the unshaded temporal networks of mainVideoUnshaded.py use losses.LossNetUnshaded")
