"""Model boundary of the reference: ``build_model(exp_cfg)['network']``
(regressor/human_shape/models/build.py:14-36)."""
from .body_heads import BODY_HEAD_REGISTRY
from .build import build_model

__all__ = ['build_model', 'BODY_HEAD_REGISTRY']
