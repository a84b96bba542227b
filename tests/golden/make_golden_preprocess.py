"""Golden vectors for the input pre-processing (n1/n2 rows): the reference's own
read_keypoints / keyps_to_bbox / bbox_to_center_scale / transform / crop run on the sample
images shipped with the reference.  cv2 is not installed, so cv2.resize inside the reference's
crop() is provided by oracle/preprocess_np.resize_bilinear_cv2 (the one unpinned sub-step);
everything else in these vectors is produced by reference code.

    python tests/golden/make_golden_preprocess.py
"""
import importlib
import os.path as osp
import sys

import numpy as np

HERE = osp.dirname(osp.abspath(__file__))
ROOT = osp.dirname(osp.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_loader                                    # noqa: E402
from oracle import preprocess_np                     # noqa: E402
from shapy_amd.datasets import OpenPose              # noqa: E402


def main():
    ref_loader.load_reference()
    cv2 = sys.modules['cv2']
    cv2.INTER_LINEAR = 1
    cv2.resize = lambda img, dsize, interpolation=None: preprocess_np.resize_bilinear_cv2(img, dsize)
    bbox = importlib.import_module('human_shape.data.utils.bbox')
    kpu = importlib.import_module('human_shape.data.utils.keypoints')
    tu = importlib.import_module('human_shape.utils.transf_utils')
    ds = OpenPose(data_folder=osp.join(HERE, 'samples'), img_folder='images', keyp_folder='openpose',
                  body_thresh=0.05, hand_thresh=0.2, head_thresh=0.3, use_face_contour=True)
    out = {}
    for i in range(len(ds)):
        img, tgt = ds[i]
        name = tgt.get_field('fname').split('.')[0]
        # reference functions on the same (thresholded) keypoints
        kp_ref = kpu.read_keypoints(osp.join(HERE, 'samples', 'openpose', name + '.json'))
        assert np.array_equal(kp_ref[0], ds.keypoints[i])
        kp = tgt.get_field('keypoints')
        c, s, b = bbox.bbox_to_center_scale(
            bbox.keyps_to_bbox(kp[:, :-1], kp[:, -1], img_size=img.shape), dset_scale_factor=1.2)
        out[f'{name}.center'] = c; out[f'{name}.scale'] = np.float64(s); out[f'{name}.bbox_size'] = np.float64(b)
        imgf = np.clip(img.astype(np.float32) / 255.0, 0, 1)
        for res in (224, 256):
            ul = np.array(tu.transform([1, 1], c, s, [res, res], invert=1)) - 1
            br = np.array(tu.transform([res + 1, res + 1], c, s, [res, res], invert=1)) - 1
            out[f'{name}.window{res}'] = np.array([ul[0], ul[1], br[0], br[1]], np.int32)
            crop = tu.crop(imgf, c, s, [res, res])
            out[f'{name}.crop{res}_sub'] = crop[::8, ::8].astype(np.float32)
            out[f'{name}.crop{res}_cs'] = np.array([crop.astype(np.float64).sum(),
                                                     (crop.astype(np.float64) ** 2).sum()])
        print(name, img.shape, c, s, b, out[f'{name}.window256'])
    np.savez(osp.join(HERE, 'preprocess_golden.npz'), **out)


if __name__ == '__main__':
    main()
