"""build_backbone (reference: models/backbone/build.py:7-27)."""
from .hrnet import build as build_hr_net


def build_backbone(backbone_cfg):
    backbone_type = backbone_cfg.get('type', 'resnet50')
    if 'hrnet' in backbone_type:
        backbone = build_hr_net(backbone_cfg, pretrained=True)
        return backbone, backbone.get_output_dim()
    if 'resnet' in backbone_type:
        raise NotImplementedError(
            'ResNet backbones are out of scope: SHAPY_A uses HRNet '
            '(configs/b2a_expose_hrnet_demo.yaml:196-199)')
    raise ValueError('Unknown backbone type: {}'.format(backbone_type))
