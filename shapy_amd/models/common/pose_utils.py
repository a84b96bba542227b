"""Pose parameterisations (reference: models/common/pose_utils.py:74-153,254-280,414-477)."""
from dataclasses import dataclass, fields
from typing import Optional

import torch
import torch.nn as nn

from ... import _lib


def _decode(x, pose_type, per):
    _lib.require_cuda(x, 'pose')
    lib = _lib.load()
    B = x.shape[0]
    flat = x.reshape(-1, per).contiguous().float()
    out = torch.empty(flat.shape[0], 3, 3, dtype=torch.float32, device=x.device)
    _lib.check(lib.shapy_pose_decode_f32(_lib.ptr(flat), pose_type, _lib.ptr(out), flat.shape[0],
                                         _lib.current_stream()), 'shapy_pose_decode_f32')
    return out.view(B, -1, 3, 3)


class ContinuousRotReprDecoder(nn.Module):
    """6-D continuous representation -> rotation matrices by Gram-Schmidt
    (pose_utils.py:74-153)."""

    def __init__(self, num_angles, dtype=torch.float32, mean=None, **kwargs):
        super().__init__()
        self.num_angles = num_angles
        self.dtype = dtype
        if isinstance(mean, dict):
            mean = mean.get('cont_rot_repr', None)
        if mean is None:
            mean = torch.tensor([1.0, 0.0, 0.0, 1.0, 0.0, 0.0], dtype=dtype).unsqueeze(
                dim=0).expand(num_angles, -1).contiguous().view(-1)
        if not torch.is_tensor(mean):
            mean = torch.tensor(mean)
        mean = mean.reshape(-1, 6)
        if mean.shape[0] < num_angles:
            mean = mean.repeat(num_angles // mean.shape[0] + 1, 1).contiguous()[:num_angles]
        elif mean.shape[0] > num_angles:
            mean = mean[:num_angles]
        self.register_buffer('mean', mean.reshape(-1).to(dtype))

    def get_type(self):
        return 'cont_rot_repr'

    def get_param_dim(self):
        return 6

    def get_dim_size(self):
        return self.num_angles * 6

    def get_mean(self):
        return self.mean.clone()

    def forward(self, module_input):
        return _decode(module_input, _lib.POSE_CONT6D, 6)


class AADecoder(nn.Module):
    """Axis-angle -> rotation matrices by Rodrigues (pose_utils.py:225-280)."""

    def __init__(self, num_angles, dtype=torch.float32, mean=None, **kwargs):
        super().__init__()
        self.num_angles = num_angles
        self.dtype = dtype
        if isinstance(mean, dict):
            mean = mean.get('aa', None)
        if mean is None:
            mean = torch.zeros([num_angles * 3], dtype=dtype)
        if not torch.is_tensor(mean):
            mean = torch.tensor(mean, dtype=dtype)
        self.register_buffer('mean', mean.reshape(-1))

    def get_type(self):
        return 'aa'

    def get_param_dim(self):
        return 3

    def get_dim_size(self):
        return self.num_angles * 3

    def get_mean(self):
        return self.mean.clone()

    def forward(self, module_input):
        return _decode(module_input, _lib.POSE_AXIS_ANGLE, 3)


@dataclass
class PoseParameterization:
    dim: int
    ind_dim: int
    decoder: Optional[nn.Module] = None
    mean: Optional[torch.Tensor] = None
    regressor: Optional[nn.Module] = None

    def keys(self):
        return [f.name for f in fields(PoseParameterization)]

    def __getitem__(self, key):
        return getattr(self, key)


def build_pose_parameterization(num_angles, type='aa', num_pca_comps=12, latent_dim_size=32,
                                append_params=True, **kwargs):
    """pose_utils.py:443-477."""
    if type == 'aa':
        decoder, ind_dim = AADecoder(num_angles=num_angles, **kwargs), 3
    elif type in ('cont_rot_repr', 'cont-rot-repr'):
        decoder, ind_dim = ContinuousRotReprDecoder(num_angles, **kwargs), 6
    elif type in ('pca', 'rot_mats'):
        raise NotImplementedError(f'pose parameterization {type} is not used by SHAPY_A')
    else:
        raise ValueError(f'Unknown pose parameterization: {type}')
    return PoseParameterization(decoder=decoder, dim=decoder.get_dim_size(), ind_dim=ind_dim,
                                mean=decoder.get_mean())
