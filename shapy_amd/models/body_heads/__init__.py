"""Body heads selectable through ``exp_cfg.network.type`` (body_heads/registry.py:3)."""
from .body_heads import SMPLHRegressor, SMPLRegressor, SMPLXRegressor
from .build import build as build_body_head
from .registry import BODY_HEAD_REGISTRY

__all__ = ['BODY_HEAD_REGISTRY', 'SMPLRegressor', 'SMPLHRegressor', 'SMPLXRegressor',
           'build_body_head']
