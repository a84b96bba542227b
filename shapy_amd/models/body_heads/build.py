"""build_body_head (reference: models/body_heads/build.py:5-32)."""
from .registry import BODY_HEAD_REGISTRY


def build(exp_cfg):
    network_cfg = exp_cfg.get('network', {})
    body_cfg = exp_cfg.get('body_model', {})
    network_type = network_cfg.get('type', 'smplx')
    key = {'SMPLRegressor': 'smpl', 'SMPLHRegressor': 'smplh', 'SMPLXRegressor': 'smplx'}.get(
        network_type)
    if key is None:
        raise ValueError(f'Unknown network type: {network_type}')
    loss_cfg = exp_cfg.get('losses', {}).get('body', {})
    return BODY_HEAD_REGISTRY.get(network_type)(
        body_cfg, network_cfg=network_cfg.get(key, {}), loss_cfg=loss_cfg)
