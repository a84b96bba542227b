"""OpenPose-keypoint image folder (reference: data/datasets/openpose.py:28-246).

Host-side plumbing only: lists ``<data_folder>/<img_folder>/*`` with their
``<keyp_folder>/<name>[_keypoints].json``, decodes the image (PIL) and derives the person box
from the confident keypoints.  Cropping / resizing / normalisation is NOT done here: it runs on
the GPU for the whole batch (datasets/preprocess.py -> csrc/preprocess.hip).

JPEGs are decoded with PIL (libjpeg); the reference uses jpeg4py (libjpeg-turbo) with a cv2
fallback (utils/img_utils.py:12-58).  Decoders may differ by +-1 LSB per pixel.
"""
import os
import os.path as osp

import numpy as np

from .keypoints import bbox_to_center_scale, keyps_to_bbox, part_indices, read_keypoints
from .structures import Target

_EXIF_ORIENTATION = 0x0112


def read_img_u8(img_fn):
    """RGB uint8 [H,W,3] with the EXIF orientation handling of img_utils.py:22-48."""
    from PIL import Image
    im = Image.open(img_fn)
    orientation = None
    try:
        exif = im._getexif()
        if exif:
            orientation = exif.get(_EXIF_ORIENTATION)
    except Exception:
        pass
    img = np.asarray(im.convert('RGB'))
    if orientation == 2:
        img = np.fliplr(img)
    elif orientation == 3:
        img = np.rot90(img, k=2)
    elif orientation == 4:
        img = np.fliplr(np.rot90(img, k=2))
    elif orientation == 5:
        img = np.fliplr(np.rot90(img, axes=(1, 0)))
    elif orientation == 6:
        img = np.rot90(img, axes=(1, 0))
    elif orientation == 7:
        img = np.fliplr(np.rot90(img))
    elif orientation == 8:
        img = np.rot90(img)
    return np.ascontiguousarray(img)


class OpenPose(object):
    def __init__(self, data_folder='data/openpose', img_folder='images', keyp_folder='keypoints',
                 split='test', use_face_contour=False, body_thresh=0.1, hand_thresh=0.2,
                 face_thresh=0.4, head_thresh=None, body_dset_factor=1.2, binarization=True,
                 **kwargs):
        if head_thresh is not None:          # config key name (datasets_defaults.py:74)
            face_thresh = head_thresh
        self.data_folder = osp.expandvars(osp.expanduser(data_folder))
        self.img_folder = osp.join(self.data_folder, img_folder)
        self.keyp_folder = osp.join(self.data_folder, keyp_folder)
        self.body_thresh, self.hand_thresh, self.face_thresh = body_thresh, hand_thresh, face_thresh
        self.body_dset_factor = body_dset_factor
        self.binarization = binarization
        idx = part_indices()
        self.body_idxs = idx['body']
        self.left_hand_idxs, self.right_hand_idxs = idx['left_hand'], idx['right_hand']
        self.face_idxs = idx['face'] if use_face_contour else idx['face'][:-17]
        self.img_paths, keypoints = [], []
        for img_fname in sorted(os.listdir(self.img_folder)):
            fname, _ = osp.splitext(img_fname)
            keyp_path = osp.join(self.keyp_folder, f'{fname}_keypoints.json')
            if not osp.exists(keyp_path):
                keyp_path = osp.join(self.keyp_folder, f'{fname}.json')
                if not osp.exists(keyp_path):
                    continue
            kp = read_keypoints(keyp_path)
            if kp is None:
                continue
            self.img_paths += [osp.join(self.img_folder, img_fname)] * kp.shape[0]
            keypoints.append(kp)
        self.keypoints = (np.concatenate(keypoints, axis=0) if keypoints
                          else np.zeros((0, 135, 3), np.float32))

    def __len__(self):
        return len(self.img_paths)

    def _filter_conf(self, keypoints2d):
        """openpose.py:150-191: per-part confidence thresholds (+ binarisation)."""
        kp = keypoints2d.copy()
        kp[:, -1] = np.clip(kp[:, -1], 0, 1)
        for idxs, thr in ((self.body_idxs, self.body_thresh), (self.face_idxs, self.face_thresh),
                          (self.left_hand_idxs, self.hand_thresh),
                          (self.right_hand_idxs, self.hand_thresh)):
            conf = kp[idxs, -1]
            if thr > 0:
                conf[conf < thr] = 0.0
            if self.binarization:
                conf = (conf >= thr).astype(kp.dtype) if thr > 0 else (conf > 0).astype(kp.dtype)
            kp[idxs, -1] = conf
        return kp

    def __getitem__(self, index):
        """-> (image uint8 [H,W,3], Target) or (None, None) when the person has < 6 keypoints."""
        img_fn = self.img_paths[index]
        img = read_img_u8(img_fn)
        kp = self._filter_conf(self.keypoints[index])
        center, scale, bbox_size = bbox_to_center_scale(
            keyps_to_bbox(kp[:, :-1], kp[:, -1]), dset_scale_factor=self.body_dset_factor)
        if center is None:
            return None, None
        target = Target(center=center, scale=scale, bbox_size=bbox_size, orig_center=center,
                        orig_bbox_size=bbox_size, keypoints=kp, fname=osp.split(img_fn)[1],
                        img_shape=img.shape)
        return img, target


def batches(dataset, batch_size, rank=0, world=1):
    """Yields lists of (image, target) of the rank's contiguous shard (parallel.shard_range)."""
    from ..parallel import shard_range
    lo, hi = shard_range(len(dataset), rank, world)
    cur = []
    for i in range(lo, hi):
        img, tgt = dataset[i]
        if img is None:
            continue
        cur.append((img, tgt))
        if len(cur) == batch_size:
            yield cur
            cur = []
    if cur:
        yield cur
