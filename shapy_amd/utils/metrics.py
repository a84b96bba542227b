"""Evaluator metrics on the GPU -- interface of regressor/human_shape/utils/metrics.py.

``point_error`` (:31-56), ``NoAlignment`` / ``ProcrustesAlignment`` / ``ScaleAlignment`` /
``TranslationAlignment`` (:59-277), ``build_alignment``, ``PointError`` (:335-365) and
``v2vhdError`` (:367-460) keep their names, constructor arguments and call signatures.  The
reference computes them with numpy / torch-sparse on the CPU of rank 0; here each is one HIP
launch (``csrc/metrics.hip``) on the tensors the forward pass left in HBM, and the results stay
``torch`` CUDA tensors (``.cpu().numpy()`` gives the reference's arrays).  numpy / CPU inputs
are accepted and moved to the current GPU -- there is no CPU implementation.
"""
import pickle

import numpy as np
import torch

from .. import _lib

_MODES = {'none': 0, 'translation': 1, 'scale': 2, 'procrustes': 3}


def _as_points(x, dtype=torch.float32):
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(np.ascontiguousarray(x))
    if not torch.is_tensor(x):
        raise TypeError(f'expected a tensor or an array of points, got {type(x)}')
    if not x.is_cuda:
        if not torch.cuda.is_available():
            raise _lib.ShapyHipError('the metrics run on the GPU only (no CPU fallback)')
        x = x.cuda()
    x = x.detach().to(dtype)
    if x.dim() == 2:
        x = x[None]
    if x.dim() != 3 or x.shape[-1] != 3:
        raise ValueError(f'points must be [B,P,3], got {tuple(x.shape)}')
    return x.contiguous()


def _aligned(est, gt, mode, want_err=True, want_mean=False, want_aligned=False):
    lib = _lib.load()
    est, gt = _as_points(est), _as_points(gt)
    if est.shape != gt.shape:
        raise ValueError(f'shape mismatch: {tuple(est.shape)} vs {tuple(gt.shape)}')
    B, P, _ = est.shape
    err = est.new_empty(B, P) if want_err else None
    mean = est.new_empty(B) if want_mean else None
    ali = torch.empty_like(est) if want_aligned else None
    _lib.check(lib.shapy_aligned_point_error_f32(
        _lib.ptr(est), _lib.ptr(gt), B, P, mode, _lib.ptr(err), _lib.ptr(mean), _lib.ptr(ali),
        _lib.current_stream()), 'shapy_aligned_point_error_f32')
    return err, mean, ali, gt


def point_error(input_points, target_points):
    """``sqrt(sum((a - b)**2, -1))`` -> [B,P] (metrics.py:31-56)."""
    return _aligned(input_points, target_points, 0)[0]


def mpjpe(input_joints, target_joints):
    """Per-joint position error [B,J] -- despite the name the reference does not average
    (metrics.py:56-78)."""
    return _aligned(input_joints, target_joints, 0)[0]


def vertex_to_vertex_error(input_vertices, target_vertices):
    """metrics.py:81-82."""
    return _aligned(input_vertices, target_vertices, 0)[0]


class _Alignment:
    _name = 'none'

    def __repr__(self):
        return type(self).__name__

    @property
    def name(self):
        return self._name

    @property
    def mode(self):
        return _MODES[self._name]

    def error(self, est, gt, per_point=True):
        """Fused alignment + error: [B,P] per-point errors, or their mean [B]."""
        err, mean, _, _ = _aligned(est, gt, self.mode, want_err=per_point, want_mean=not per_point)
        return err if per_point else mean

    def __call__(self, S1, S2):
        """Returns (aligned S1, S2) like the reference's alignment objects."""
        _, _, ali, gt = _aligned(S1, S2, self.mode, want_err=False, want_aligned=True)
        return ali, gt


class NoAlignment(_Alignment):
    _name = 'none'


class ProcrustesAlignment(_Alignment):
    _name = 'procrustes'


class ScaleAlignment(_Alignment):
    _name = 'scale'


class TranslationAlignment(_Alignment):
    _name = 'translation'


class RootAlignment(_Alignment):
    """Subtracts the mean of the root joints from both point sets (metrics.py:280-316)."""
    _name = 'root'

    def __init__(self, root=None, **kwargs):
        self.root = [0] if root is None else list(root)

    def set_root(self, new_root):
        self.root = list(new_root)

    @property
    def mode(self):
        return 0

    def __call__(self, est, gt):
        est, gt = _as_points(est), _as_points(gt)
        idx = torch.as_tensor(self.root, dtype=torch.long, device=est.device)
        return (est - est[:, idx].mean(dim=1, keepdim=True),
                gt - gt[:, idx].mean(dim=1, keepdim=True))


def build_alignment(name, **kwargs):
    """metrics.py:319-332."""
    if name == 'procrustes':
        return ProcrustesAlignment()
    if name == 'root':
        return RootAlignment(**kwargs)
    if name == 'scale':
        return ScaleAlignment()
    if name == 'translation':
        return TranslationAlignment()
    if name in ('no', 'none'):
        return NoAlignment()
    raise ValueError(f'Unknown alignment type: {name}')


class PointError:
    """``PointError(alignment)(est, gt)`` -> per-point error [B,P] (metrics.py:335-365);
    alignment and error are one launch."""

    def __init__(self, alignment_object, name=''):
        self._alignment = alignment_object
        self._name = name

    @property
    def name(self):
        return self._name

    def __repr__(self):
        return f'PointError: Alignment = {self._alignment}'

    def set_root(self, new_root):
        if hasattr(self._alignment, 'set_root'):
            self._alignment.set_root(new_root)

    def set_alignment(self, alignment_object):
        self._alignment = alignment_object

    def __call__(self, est_points, gt_points):
        if isinstance(self._alignment, RootAlignment):
            est_points, gt_points = self._alignment(est_points, gt_points)
        return _aligned(est_points, gt_points, self._alignment.mode)[0]


class v2vhdError(torch.nn.Module):  # noqa: N801  (reference class name)
    """Point-to-point error for meshes of different topology (metrics.py:367-460): a fixed set
    of P points is regressed from both meshes with sparse P x V matrices, optionally translated
    onto each other, and their distances averaged.  ``__call__`` returns ``(error.mean(1),
    error)`` in float64 like the reference."""

    def __init__(self, input_point_regressor_path='', target_point_regressor_path='', align=True,
                 input_point_regressor=None, target_point_regressor=None):
        super().__init__()
        self.align = align
        if input_point_regressor is None:
            with open(input_point_regressor_path, 'rb') as f:
                input_point_regressor = pickle.load(f)
        if target_point_regressor is None:
            with open(target_point_regressor_path, 'rb') as f:
                target_point_regressor = pickle.load(f)
        for tag, mat in (('input', input_point_regressor), ('target', target_point_regressor)):
            rp, ci, va = self.to_csr(mat)
            self.register_buffer(f'{tag}_rowptr', rp)
            self.register_buffer(f'{tag}_col', ci)
            self.register_buffer(f'{tag}_val', va)
            setattr(self, f'{tag}_shape', tuple(mat.shape))
        if self.input_shape[0] != self.target_shape[0]:
            raise ValueError('both point regressors must produce the same number of points')

    @staticmethod
    def to_csr(point_regressor):
        """scipy.sparse matrix -> (rowptr int32, col int32, val float64) tensors; duplicates
        summed, like the coalescing sparse mm of the reference (metrics.py:394-412)."""
        m = point_regressor.tocsr().astype(np.float64)
        m.sum_duplicates()
        return (torch.from_numpy(m.indptr.astype(np.int32)),
                torch.from_numpy(m.indices.astype(np.int32)),
                torch.from_numpy(m.data.astype(np.float64)))

    def forward(self, input_points, target_points):
        lib = _lib.load()
        pin = _as_points(input_points, torch.float64)
        pta = _as_points(target_points, torch.float64)
        if pin.shape[0] != pta.shape[0]:
            raise ValueError('batch size mismatch')
        P, V1 = self.input_shape
        V2 = self.target_shape[1]
        if pin.shape[1] != V1 or pta.shape[1] != V2:
            raise ValueError(f'expected {V1} input / {V2} target vertices, got '
                             f'{pin.shape[1]} / {pta.shape[1]}')
        dev = pin.device
        if self.input_val.device != dev:
            self.to(dev)
        B = pin.shape[0]
        err = torch.empty(B, P, dtype=torch.float64, device=dev)
        mean = torch.empty(B, dtype=torch.float64, device=dev)
        _lib.check(lib.shapy_p2p_error_f64(
            _lib.ptr(self.input_rowptr), _lib.ptr(self.input_col), _lib.ptr(self.input_val),
            _lib.ptr(self.target_rowptr), _lib.ptr(self.target_col), _lib.ptr(self.target_val),
            _lib.ptr(pin), _lib.ptr(pta), B, P, V1, V2, int(bool(self.align)), _lib.ptr(err),
            _lib.ptr(mean), _lib.current_stream()), 'shapy_p2p_error_f64')
        return mean, err
