from .build import build_backbone
