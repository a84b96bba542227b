"""ORACLE (test infrastructure only -- never imported by shapy_amd/).

NumPy restatement of the reference's per-image input pipeline:
  read_img /255 + clip           regressor/human_shape/utils/img_utils.py:60-64
  crop() window copy, zero fill  regressor/human_shape/utils/transf_utils.py:53-84
  cv2.resize INTER_LINEAR        transf_utils.py:95 -- third-party OpenCV (opencv-python,
      requirements.txt), absent from the image and from /root/reference: its float32 bilinear
      algorithm is restated (half-pixel centres; source index clamped to the window;
      horizontal pass then vertical pass).  That sub-step is "parity unpinned"; everything
      around it is pinned by running the reference's own crop()/transform() with this resize
      plugged in (tests/golden/make_golden_preprocess.py).
  ToTensor + Normalize           data/transforms/transforms.py:613-624,710-733
"""
import numpy as np

f32 = np.float32


def resize_bilinear_cv2(src, dsize):
    """src [h,w,c] float32 -> [dsize[1], dsize[0], c] like cv2.resize(src, dsize, INTER_LINEAR)."""
    h, w = src.shape[:2]
    dw, dh = dsize

    def taps(n_src, n_dst):
        f = (np.arange(n_dst) + 0.5) * (n_src / n_dst) - 0.5
        s = np.floor(f).astype(np.int64)
        a = (f - s).astype(f32)
        lo = s < 0
        a[lo] = 0; s[lo] = 0
        hi = s >= n_src - 1
        a[hi] = 0; s[hi] = n_src - 1
        return s, np.minimum(s + 1, n_src - 1), a
    sx, sx1, ax = taps(w, dw)
    sy, sy1, ay = taps(h, dh)
    ax = ax[None, :, None]
    rows0 = src[sy][:, sx] * (f32(1) - ax) + src[sy][:, sx1] * ax
    rows1 = src[sy1][:, sx] * (f32(1) - ax) + src[sy1][:, sx1] * ax
    ay = ay[:, None, None]
    return (rows0 * (f32(1) - ay) + rows1 * ay).astype(f32)


def crop(img, window, res):
    """transf_utils.crop (no rotation) for a float image and the integer window (ul, br)."""
    ulx, uly, brx, bry = [int(v) for v in window]
    new_img = np.zeros([bry - uly, brx - ulx, img.shape[2]], dtype=img.dtype)
    new_x = max(0, -ulx), min(brx, img.shape[1]) - ulx
    new_y = max(0, -uly), min(bry, img.shape[0]) - uly
    old_x = max(0, ulx), min(img.shape[1], brx)
    old_y = max(0, uly), min(img.shape[0], bry)
    new_img[new_y[0]:new_y[1], new_x[0]:new_x[1]] = img[old_y[0]:old_y[1], old_x[0]:old_x[1]]
    return resize_bilinear_cv2(new_img, (res, res))


def preprocess(img_u8, window, res, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
    img = np.clip(img_u8.astype(f32) / f32(255.0), 0, 1)
    c = np.clip(crop(img, window, res), 0, 1).transpose(2, 0, 1)
    return ((c - np.asarray(mean, f32)[:, None, None]) / np.asarray(std, f32)[:, None, None]).astype(f32)
