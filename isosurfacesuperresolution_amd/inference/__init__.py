from .camera import Camera, Orientation
from .renderer import Material, DirectRenderer, Renderer
from .loadedmodel import LoadedModel
from .flowfill import fill_flow
