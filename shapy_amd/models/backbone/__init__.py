"""HRNet-W48 backbone on the gfx950 matrix cores (mirror of human_shape.models.backbone)."""
from .build import build_backbone
from .hrnet import HighResolutionNet

__all__ = ['build_backbone', 'HighResolutionNet']
