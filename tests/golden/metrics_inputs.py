"""Seeded inputs shared by make_golden_metrics.py (reference outputs) and the metric tests."""
import numpy as np
import scipy.sparse as sp

from shapy_amd.utils import synthetic as syn

B, V, V2, P = 5, 10475, 6890, 2000


def _rot(a):
    cx, cy, cz = np.cos(a)
    sx, sy, sz = np.sin(a)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def point_regressor(rng, faces, n_verts, n_points):
    """P x V barycentric sampler: one random point on a random face per row."""
    f = faces[rng.integers(0, len(faces), n_points)]
    w = rng.dirichlet(np.ones(3), n_points)
    rows = np.repeat(np.arange(n_points), 3)
    return sp.coo_matrix((w.reshape(-1), (rows, f.reshape(-1))), shape=(n_points, n_verts)).tocsr()


def make_inputs():
    r = syn.rng_for(0, 'metrics')
    model = syn.make_synthetic_smplx(0)
    base = model['v_template'][None].astype(np.float32)
    est = (base + 0.02 * r.standard_normal((B, V, 3))).astype(np.float32)
    # ground truth: rotated / scaled / shifted copies with noise, so every alignment matters
    Rm = np.stack([_rot(a) for a in r.uniform(-0.3, 0.3, (B, 3))]).astype(np.float32)
    s = r.uniform(0.9, 1.1, (B, 1, 1)).astype(np.float32)
    gt = (s * np.einsum('bij,bpj->bpi', Rm, base.repeat(B, 0)) +
          r.uniform(-0.2, 0.2, (B, 1, 3)) + 0.01 * r.standard_normal((B, V, 3))).astype(np.float32)
    joints = r.standard_normal((B, 24, 3)).astype(np.float32)
    joints_gt = (joints + 0.05 * r.standard_normal((B, 24, 3))).astype(np.float32)
    # P2P: SMPL-X topology on the estimate side, a different topology on the target side
    reg_in = point_regressor(r, np.asarray(model['f']), V, P)
    reg_tg = point_regressor(r, r.integers(0, V2, (12000, 3)), V2, P)
    tgt = 0.5 * r.standard_normal((B, V2, 3)) + r.uniform(-0.1, 0.1, (B, 1, 3))
    return dict(est=est, gt=gt, joints=joints, joints_gt=joints_gt, reg_in=reg_in, reg_tg=reg_tg,
                tgt=tgt)
