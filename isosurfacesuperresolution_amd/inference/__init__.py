from .camera import Camera, Orientation
from .renderer import Material, DirectRenderer
