from .build import build_model
from .body_heads import BODY_HEAD_REGISTRY
