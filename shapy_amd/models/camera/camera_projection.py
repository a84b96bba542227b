"""Camera (reference: models/camera/camera_projection.py:16-89,173-213)."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import _lib
from ..body_models import KeypointTensor

DEFAULT_FOCAL_LENGTH = 5000


class CameraParams(object):
    KEYS = ['translation', 'rotation', 'scale', 'focal_length', 'scale_first']
    attributes = KEYS

    def __init__(self, translation=None, rotation=None, scale=None, scale_first=False,
                 focal_length=None):
        self.translation = translation
        self.rotation = rotation
        self.scale = scale
        self.focal_length = focal_length
        self.scale_first = scale_first

    def keys(self):
        return [key for key in self.KEYS if getattr(self, key) is not None]

    def get(self, key, default=None):
        return getattr(self, key, default)

    def __getitem__(self, key):
        return getattr(self, key)


class WeakPerspectiveCamera(nn.Module):
    """Scaled orthographic camera (camera_projection.py:173-213)."""

    def __init__(self, scale_first=False, **kwargs):
        super().__init__()
        self.scale_first = scale_first

    def forward(self, points, scale, translation, **kwargs):
        assert translation.shape[-1] == 2, 'Translation shape must be -1x2'
        assert scale.shape[-1] == 1, 'Scale shape must be -1x1'
        src = points
        pts = points._t if isinstance(points, KeypointTensor) else points
        _lib.require_cuda(pts, 'points')
        lib = _lib.load()
        B, N = pts.shape[:2]
        pts = pts.contiguous().float()
        out = torch.empty(B, N, 2, dtype=torch.float32, device=pts.device)
        _lib.check(lib.shapy_weak_persp_project_f32(
            _lib.ptr(pts), _lib.ptr(scale.reshape(-1).contiguous().float()),
            _lib.ptr(translation.reshape(-1, 2).contiguous().float()), _lib.ptr(out), B, N,
            int(self.scale_first), _lib.current_stream()), 'shapy_weak_persp_project_f32')
        if isinstance(src, KeypointTensor):
            out = KeypointTensor.from_obj(out, src)
        return out


def build_cam_proj(camera_cfg, dtype=torch.float32):
    """camera_projection.py:44-89."""
    camera_type = camera_cfg.get('type', 'weak-persp')
    pos = camera_cfg.get('pos_func')
    if pos == 'softplus':
        scale_func = F.softplus
    elif pos == 'exp':
        scale_func = torch.exp
    elif pos in ('none', 'None'):
        scale_func = lambda x: x
    else:
        raise ValueError(f'Unknown positive scaling function: {pos}')
    if camera_type.lower() == 'weak-persp':
        wp = camera_cfg.get('weak_persp', {})
        mean_scale = wp.get('mean_scale', 0.9)
        if pos == 'softplus':
            mean_scale = np.log(np.exp(mean_scale) - 1)
        elif pos == 'exp':
            mean_scale = np.log(mean_scale)
        return {'camera': WeakPerspectiveCamera(scale_first=wp.get('scale_first', False)),
                'mean': torch.tensor([mean_scale, 0.0, 0.0], dtype=torch.float32),
                'scale_func': scale_func, 'dim': 3, 'pos_func': pos}
    if camera_type.lower() == 'persp':
        raise NotImplementedError('perspective camera is not used by SHAPY_A')
    raise ValueError(f'Unknown camera type: {camera_type}')
