"""build_model -- the plug-in boundary used by demo.py / evaluate.py
(reference: regressor/human_shape/models/build.py:14-36)."""
from .body_heads import build_body_head, BODY_HEAD_REGISTRY


def build_model(exp_cfg):
    network_cfg = exp_cfg.get('network', {})
    net_type = network_cfg.get('type', 'expose')
    if net_type in BODY_HEAD_REGISTRY:
        network = build_body_head(exp_cfg)
    else:
        raise ValueError(f'Unknown network type: {net_type}')
    if exp_cfg.get('use_adv_training', False):
        raise NotImplementedError
    return {'network': network, 'discriminator': None, 'discriminator_loss': None}
