"""Batch pre-processing on the GPU: full images + person boxes -> normalised crops [B,3,S,S]."""
import ctypes

import numpy as np
import torch

from .. import _lib
from .keypoints import crop_window


def crop_and_normalize(images, centers, scales, crop_size, mean=(0.485, 0.456, 0.406),
                       std=(0.229, 0.224, 0.225), device='cuda'):
    """images: list of uint8 [H,W,3] arrays; centers/scales as produced by the dataset
    (bbox_to_center_scale).  Equivalent to Crop + ToTensor + Normalize of the reference
    (data/transforms/transforms.py:521-573,613-624,710-733) for every image, in one launch."""
    lib = _lib.load()
    B = len(images)
    offs, hw, boxes, total = [], [], [], 0
    for img, c, s in zip(images, centers, scales):
        if img.dtype != np.uint8 or img.ndim != 3 or img.shape[2] != 3:
            raise ValueError('images must be uint8 [H,W,3]')
        offs.append(total)
        total += img.size
        hw.append(img.shape[:2])
        boxes.append(crop_window(c, s, [crop_size, crop_size]))
    flat = np.concatenate([np.ascontiguousarray(i).reshape(-1) for i in images])
    d_img = torch.from_numpy(flat).to(device)
    d_off = torch.tensor(offs, dtype=torch.int64, device=device)
    d_hw = torch.tensor(np.asarray(hw, np.int32), device=device)
    d_box = torch.tensor(np.asarray(boxes, np.int32), device=device)
    out = torch.empty(B, 3, crop_size, crop_size, dtype=torch.float32, device=device)
    m = (ctypes.c_float * 3)(*[float(x) for x in mean])
    s = (ctypes.c_float * 3)(*[float(x) for x in std])
    _lib.check(lib.shapy_crop_resize_normalize_u8(
        _lib.ptr(d_img), _lib.ptr(d_off), _lib.ptr(d_hw), _lib.ptr(d_box), _lib.ptr(out), B,
        crop_size, m, s, _lib.current_stream()), 'shapy_crop_resize_normalize_u8')
    return out
