from .mv import MeanVariance
from .shading import ScreenSpaceShading
from .initial_image import initialImage
from .psnr import PSNR
from .ssim import SSIM, MSSSIM
