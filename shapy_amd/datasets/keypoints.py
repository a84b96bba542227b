"""OpenPose keypoint reading and person boxes (host logic, NumPy).

Reference: data/utils/keypoints.py:75-120 (read_keypoints), data/utils/bbox.py:54-97
(keyps_to_bbox, bbox_to_center_scale), utils/transf_utils.py:9-66 (get_transform, transform).
"""
import json
import os.path as osp

import numpy as np

_DATA = osp.join(osp.dirname(osp.dirname(osp.abspath(__file__))), 'data')


def part_indices():
    """Index lists of the openpose25_v1 keypoint layout (body / hands / face), a data table
    derived from the reference's keypoint name lists (data/utils/keypoint_names.py)."""
    with open(osp.join(_DATA, 'openpose25_v1_parts.json')) as f:
        d = json.load(f)
    return {k: np.asarray(v, dtype=np.int64) for k, v in d.items() if isinstance(v, list)}


def read_keypoints(keypoint_fn):
    """-> [num_people, 135, 3] float32 (x, y, confidence) or None."""
    with open(keypoint_fn) as f:
        data = json.load(f)
    people = []
    for person in data['people']:
        body = np.array(person['pose_keypoints_2d'], dtype=np.float32).reshape(-1, 3)

        def part(key, n):
            v = person.get(key, [])
            if len(v) < 1:
                v = [0] * (n * 3)
            return np.array(v, dtype=np.float32).reshape(-1, 3)
        lh, rh = part('hand_left_keypoints_2d', 21), part('hand_right_keypoints_2d', 21)
        face = part('face_keypoints_2d', 70)[:-2]
        people.append(np.concatenate([body, lh, rh, face], axis=0))
    if not people:
        return None
    return np.stack(people)


def keyps_to_bbox(keypoints, conf, min_valid_keypoints=6, scale=1.0):
    valid = keypoints[conf > 0]
    if len(valid) < min_valid_keypoints:
        return None
    xmin, ymin = np.amin(valid, axis=0)
    xmax, ymax = np.amax(valid, axis=0)
    width, height = (xmax - xmin) * scale, (ymax - ymin) * scale
    xc, yc = 0.5 * (xmax + xmin), 0.5 * (ymax + ymin)
    bbox = np.stack([xc - 0.5 * width, yc - 0.5 * height, xc + 0.5 * width,
                     yc + 0.5 * height]).astype(np.float32)
    return bbox if (bbox[2] - bbox[0]) * (bbox[3] - bbox[1]) > 0 else None


def bbox_to_center_scale(bbox, dset_scale_factor=1.0, ref_bbox_size=200):
    if bbox is None:
        return None, None, None
    bbox = bbox.reshape(-1)
    bbox_size = dset_scale_factor * max(bbox[2] - bbox[0], bbox[3] - bbox[1])
    scale = bbox_size / ref_bbox_size
    center = np.stack([(bbox[0] + bbox[2]) * 0.5, (bbox[1] + bbox[3]) * 0.5]).astype(np.float32)
    return center, scale, bbox_size


def get_transform(center, scale, res):
    """transf_utils.py:9-36 without rotation."""
    h = 200 * scale
    t = np.zeros((3, 3), dtype=np.float32)
    t[0, 0] = float(res[1]) / h
    t[1, 1] = float(res[0]) / h
    t[0, 2] = res[1] * (-float(center[0]) / h + .5)
    t[1, 2] = res[0] * (-float(center[1]) / h + .5)
    t[2, 2] = 1
    return t.astype(np.float32)


def transform(pt, center, scale, res, invert=0):
    """transf_utils.py:41-49."""
    t = get_transform(center, scale, res)
    if invert:
        t = np.linalg.inv(t)
    new_pt = np.array([pt[0] - 1, pt[1] - 1, 1.], dtype=np.float32).T
    new_pt = np.dot(t, new_pt)
    return new_pt[:2].astype(int) + 1


def crop_window(center, scale, res):
    """Integer crop window (ul_x, ul_y, br_x, br_y) of crop() (transf_utils.py:53-58)."""
    ul = np.array(transform([1, 1], center, scale, res, invert=1)) - 1
    br = np.array(transform([res[0] + 1, res[1] + 1], center, scale, res, invert=1)) - 1
    return np.array([ul[0], ul[1], br[0], br[1]], dtype=np.int32)
