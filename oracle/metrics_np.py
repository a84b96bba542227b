"""CPU oracle of the evaluator metrics (TEST INFRASTRUCTURE ONLY -- never imported by the product).

numpy restatement of regressor/human_shape/utils/metrics.py: ``point_error`` (:31-56), the four
alignments (:84-277) and ``v2vhdError.__call__`` (:419-460).  Pinned against the reference's own
classes run on the CPU in this container: tests/golden/metrics_golden.npz
(tests/golden/make_golden_metrics.py).
"""
import numpy as np


def point_error(est, gt):
    """metrics.py:31-56."""
    d = np.asarray(est) - np.asarray(gt)
    return np.sqrt((d * d).sum(axis=-1))


def align(est, gt, kind):
    """Aligned estimate for ``kind`` in none / translation / scale / procrustes.
    est, gt: [B,P,3].  metrics.py:84-277 work on the transposed [B,3,P] layout; the algebra
    below is the same written for row vectors."""
    est = np.asarray(est)
    gt = np.asarray(gt)
    if kind in ('none', 'no'):
        return est
    mu1 = est.mean(axis=1, keepdims=True)
    mu2 = gt.mean(axis=1, keepdims=True)
    if kind == 'translation':                      # :248-277
        return est + (mu2 - mu1)
    x1, x2 = est - mu1, gt - mu2
    var1 = (x1 ** 2).sum(axis=(1, 2))
    if kind == 'scale':                            # :184-232
        s = np.sqrt((x2 ** 2).sum(axis=(1, 2)) / var1)
        return s[:, None, None] * est + (mu2 - s[:, None, None] * mu1)
    if kind == 'procrustes':                       # :100-170
        K = np.einsum('bpi,bpj->bij', x1, x2)      # X1 X2^T in the reference's layout
        U, _, Vh = np.linalg.svd(K)
        V = np.transpose(Vh, (0, 2, 1))
        Z = np.tile(np.eye(3)[None], (len(K), 1, 1))
        Z[:, 2, 2] = np.sign(np.linalg.det(U @ Vh))
        R = V @ Z @ np.transpose(U, (0, 2, 1))
        s = np.trace(R @ K, axis1=1, axis2=2) / var1
        t = mu2[:, 0] - s[:, None] * np.einsum('bmn,bn->bm', R, mu1[:, 0])
        return s[:, None, None] * np.einsum('bmn,bpn->bpm', R, est) + t[:, None]
    raise ValueError(kind)


def aligned_point_error(est, gt, kind):
    """PointError(alignment)(est, gt) (metrics.py:335-365)."""
    return point_error(align(est, gt, kind), gt)


def p2p_error(in_reg, tgt_reg, input_verts, target_verts, do_align=True):
    """v2vhdError.__call__ (metrics.py:419-460): in_reg / tgt_reg are scipy.sparse P x V
    matrices, vertices [B,V,3]; float64 throughout (evaluation.py:253-255)."""
    pin = np.asarray(input_verts, np.float64)
    pta = np.asarray(target_verts, np.float64)
    a = np.stack([in_reg @ v for v in pin])
    c = np.stack([tgt_reg @ v for v in pta])
    t = c.mean(axis=1) - a.mean(axis=1) if do_align else np.zeros((len(a), 3))
    err = np.sqrt(((a + t[:, None] - c) ** 2).sum(-1))
    return err.mean(1), err


def measurement_error(est, gt):
    """Evaluator._compute_measurement_error (evaluation.py:265-296): |gt - est| over the
    samples whose ground truth is positive.  est, gt: dict name -> [B]."""
    out = {}
    for k, v in est.items():
        if k not in gt:
            continue
        g = np.asarray(gt[k], np.float64)
        m = g > 0
        if m.any():
            out[k] = np.abs(g[m] - np.asarray(v)[m])
    return out
