from .registry import BODY_HEAD_REGISTRY
from .body_heads import *
from .build import build as build_body_head
