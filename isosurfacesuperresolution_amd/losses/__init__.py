from .lossbuilder import LossBuilder
from .lossnet_unshaded import LossNetUnshaded
from .lossnet import LossNet
