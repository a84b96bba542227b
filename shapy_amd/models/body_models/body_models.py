"""SMPL-X layer on gfx950 kernels.

Drop-in for ``SMPLX`` in regressor/human_shape/models/body_models/body_models.py:497-767:
same constructor arguments (``model_folder``, ``ext``, ``betas.num``, ``expression.num``,
``use_face_contour``, ``j14_regressor_path``, ...), same registered buffers (so checkpoints
that carry ``model.*`` buffers load), same ``forward`` signature and output dict.

Data layout on the device (prepared once, float64 folds done on the host):
  * shapedirs / posedirs are stored K-contiguous and row-padded ([V*3, 32] and [V*3, 496]) so
    that the two blend-shape sums are GEMMs on the f32 MFMA kernel with M = batch:
        v_shaped = v_template + S c            (lbs.py:163, 218-239)  bias epilogue
        v_posed  = v_shaped + P^T pose_feat    (lbs.py:171-182)       residual epilogue
    posedirs (61 MB) is streamed from HBM exactly once per batch;
  * the joint regressor is folded through the shape basis: J = J_t + (J_reg S) c, which
    removes the [55,V] x [B,V,3] product (lbs.py:167, 199-215) from the per-batch work;
  * lbs_weights are stored transposed [55, V] so the skinning kernel reads them coalesced.
"""
import os
import os.path as osp
import pickle
from collections import defaultdict

import numpy as np
import torch
import torch.nn as nn

from ... import _lib
from ...utils.versioning import VersionedWeights
from ...utils.synthetic import keypoint_names as _smplx_keypoint_names
from .utils import KeypointTensor, find_joint_kin_chain, to_tensor

J14_NAMES = ['right_ankle', 'right_knee', 'right_hip', 'left_hip', 'left_knee', 'left_ankle',
             'right_wrist', 'right_elbow', 'right_shoulder', 'left_shoulder', 'left_elbow',
             'left_wrist', 'neck', 'head']
J9_NAMES = ['right_wrist', 'right_elbow', 'right_shoulder', 'left_shoulder', 'left_elbow',
            'left_wrist', 'spine', 'jaw ', 'head']


def _to_np(array, dtype=np.float32):
    if 'scipy.sparse' in str(type(array)):
        array = array.todense()
    return np.array(array, dtype=dtype)


def _round_up(x, m):
    return (x + m - 1) // m * m


class SMPLX(VersionedWeights, nn.Module):
    NUM_BODY_JOINTS = 21
    NUM_HAND_JOINTS = 15
    NUM_FACE_JOINTS = 3
    NUM_JOINTS = NUM_BODY_JOINTS + 2 * NUM_HAND_JOINTS + NUM_FACE_JOINTS
    SHAPE_SPACE_DIM = 300
    EXPRESSION_SPACE_DIM = 100
    NECK_IDX = 12
    HEAD_IDX = 15
    NAME = 'smplx'

    def __init__(self, model_folder, is_training=False, data_struct=None, betas=None,
                 expression=None, use_face_contour=False, gender='neutral',
                 dtype=torch.float32, ext='npz', extra_joint_path='', v_template_path='',
                 head_verts_ids_path='', **kwargs):
        super().__init__()
        if dtype != torch.float32:
            raise NotImplementedError('the HIP SMPL-X layer computes in float32')
        if extra_joint_path:
            raise NotImplementedError('extra_joint_path is empty in every SHAPY config')
        if data_struct is None:
            path = osp.join(model_folder, f'SMPLX_{gender.upper()}.{ext}')
            if ext == 'npz':
                data_struct = dict(np.load(path, allow_pickle=True))
            else:
                with open(path, 'rb') as f:
                    data_struct = dict(pickle.load(f, encoding='latin1'))
        ds = data_struct
        self.gender = gender
        self.dtype = dtype
        self.use_face_contour = use_face_contour
        self._num_betas = (betas or {'num': 10}).get('num', 10)
        self._num_expression_coeffs = (expression or {'num': 10}).get('num', 10)

        self.faces = _to_np(ds['f'], dtype=np.int64)
        self.register_buffer('faces_tensor', to_tensor(self.faces, dtype=torch.long))
        v_template_path = osp.expandvars(v_template_path or '')
        if osp.exists(v_template_path):
            raise NotImplementedError('v_template_path meshes need trimesh (not in this image)')
        self.register_buffer('v_template', to_tensor(_to_np(ds['v_template'])))
        head_verts_ids_path = osp.expandvars(head_verts_ids_path or '')
        head_ids = np.load(head_verts_ids_path) if osp.exists(head_verts_ids_path) else []
        self.register_buffer('head_vertices_ids', torch.tensor(head_ids, dtype=torch.long))

        num_betas = min(self._num_betas, self.SHAPE_SPACE_DIM)
        shapedirs = np.asarray(ds['shapedirs'])
        self.register_buffer('shapedirs', to_tensor(_to_np(shapedirs[:, :, :num_betas])))
        self.register_buffer('J_regressor', to_tensor(_to_np(ds['J_regressor'])))
        num_pose_basis = ds['posedirs'].shape[-1]
        posedirs = np.reshape(ds['posedirs'], [-1, num_pose_basis]).T
        self.register_buffer('posedirs', to_tensor(_to_np(posedirs)))
        parents = to_tensor(_to_np(ds['kintree_table'][0], dtype=np.int64), dtype=torch.long)
        parents[0] = -1
        self.register_buffer('parents', parents)
        self.register_buffer('lbs_weights', to_tensor(_to_np(ds['weights'])))

        self.register_buffer('lmk_faces_idx',
                             torch.tensor(np.asarray(ds['lmk_faces_idx'], np.int64)))
        self.register_buffer('lmk_bary_coords', to_tensor(_to_np(ds['lmk_bary_coords'])))
        self.register_buffer('dynamic_lmk_faces_idx',
                             torch.tensor(np.asarray(ds['dynamic_lmk_faces_idx'], np.int64)))
        self.register_buffer('dynamic_lmk_bary_coords',
                             to_tensor(_to_np(ds['dynamic_lmk_bary_coords'])))
        neck_kin_chain = find_joint_kin_chain(self.HEAD_IDX, parents.tolist())
        self.register_buffer('neck_kin_chain', torch.tensor(neck_kin_chain, dtype=torch.long))
        e0 = self.SHAPE_SPACE_DIM
        self.register_buffer('expr_dirs', to_tensor(_to_np(
            shapedirs[:, :, e0:e0 + self._num_expression_coeffs])))

        self._keypoint_names = self.build_keypoint_names()
        j14_regressor_path = osp.expandvars(kwargs.get('j14_regressor_path', '') or '')
        self.use_joint_regressor = osp.exists(j14_regressor_path)
        if self.use_joint_regressor:
            if j14_regressor_path.endswith('.pkl'):
                with open(j14_regressor_path, 'rb') as f:
                    j14 = pickle.load(f, encoding='latin1')
            else:
                j14 = np.load(j14_regressor_path)
            target_names = J14_NAMES if j14.shape[0] == 14 else J9_NAMES
            source = [i for i, n in enumerate(self.keypoint_names) if n in target_names]
            target = [target_names.index(self.keypoint_names[i]) for i in source]
            self.register_buffer('source_idxs', torch.tensor(source, dtype=torch.long))
            self.register_buffer('target_idxs', torch.tensor(target, dtype=torch.long))
            self.register_buffer('extra_joint_regressor',
                                 torch.from_numpy(np.asarray(j14)).to(torch.float32))
        self._dev = {}
        self.register_load_state_dict_post_hook(lambda m, k: m.invalidate())

    # ---- reference properties ----
    name = property(lambda self: self.NAME)
    num_betas = property(lambda self: self._num_betas)
    num_expression_coeffs = property(lambda self: self._num_expression_coeffs)
    num_body_joints = property(lambda self: self.NUM_BODY_JOINTS)
    num_hand_joints = property(lambda self: self.NUM_HAND_JOINTS)
    num_joints = property(lambda self: self.J_regressor.shape[0])
    num_verts = property(lambda self: self.v_template.shape[0])
    keypoint_names = property(lambda self: self._keypoint_names)
    connections = property(lambda self: None)
    parts = property(lambda self: None)
    part_connections = property(lambda self: None)

    def get_num_verts(self):
        return self.v_template.shape[0]

    def get_num_faces(self):
        return self.faces.shape[0]

    def get_head_vertices_ids(self):
        return self.head_vertices_ids

    def build_keypoint_names(self):
        names = _smplx_keypoint_names()
        if not self.use_face_contour:
            names = [n for n in names if 'contour' not in n]
        return names

    def extra_repr(self):
        return '\n'.join([f'Gender: {self.gender.upper()}',
                          f'Number of joints: {self.J_regressor.shape[0]}',
                          f'Betas: {self.num_betas}',
                          f'Number of Expression Coefficients: {self.num_expression_coeffs}',
                          f'Use face contour: {self.use_face_contour}'])

    # ---- device-side model ----
    def invalidate(self):
        self._dev = {}
        self._drop_version_cache()

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._dev = {}
        self._drop_version_cache()
        return out

    def __getstate__(self):
        st = self.__dict__.copy()
        st['_dev'] = {}
        st['_ver_tensors'] = None
        return st

    def _device_model(self, device):
        ver = self._weights_version()          # in-place edits of the model buffers
        if ver != self.__dict__.get('_dev_ver'):
            self._dev = {}
            self.__dict__['_dev_ver'] = ver
        key = str(device)
        dm = self._dev.get(key)
        if dm is not None:
            return dm
        V, J = self.v_template.shape[0], self.J_regressor.shape[0]
        nb, ne = self.shapedirs.shape[-1], self.expr_dirs.shape[-1]
        NB = nb + ne
        NBpad = _round_up(NB, 16)
        P = self.posedirs.shape[0]
        Ppad = _round_up(P, 16)
        N = V * 3
        Npad = _round_up(N, 128)
        d64 = lambda t: t.detach().double().cpu()
        S = torch.cat([d64(self.shapedirs), d64(self.expr_dirs)], dim=-1)       # V,3,NB
        Jreg = d64(self.J_regressor)
        J_t = Jreg @ d64(self.v_template)                                       # J,3
        J_s = torch.einsum('jv,vkl->jkl', Jreg, S)                              # J,3,NB
        sh_t = torch.zeros(Npad, NBpad, dtype=torch.float32)
        sh_t[:N, :NB] = S.reshape(N, NB).float()
        pd_t = torch.zeros(Npad, Ppad, dtype=torch.float32)
        pd_t[:N, :P] = self.posedirs.detach().float().cpu().t()
        f32 = lambda t: t.detach().float().contiguous().to(device)
        i32 = lambda t: t.detach().to(torch.int32).contiguous().to(device)
        t = dict(
            parents=i32(self.parents), J_template=f32(J_t), J_shapedirs=f32(J_s),
            v_template=f32(self.v_template.reshape(-1)), shapedirs_t=sh_t.to(device),
            posedirs_t=pd_t.to(device), lbs_weights_t=f32(self.lbs_weights.t()),
            faces=i32(self.faces_tensor), lmk_faces_idx=i32(self.lmk_faces_idx),
            lmk_bary=f32(self.lmk_bary_coords), dyn_lmk_faces_idx=i32(self.dynamic_lmk_faces_idx),
            dyn_lmk_bary=f32(self.dynamic_lmk_bary_coords), neck_kin_chain=i32(self.neck_kin_chain))
        m = _lib.ShapySmplxModel()
        m.V, m.J, m.NB, m.P, m.Ppad, m.NBpad = V, J, NB, P, Ppad, NBpad
        m.n_static_lmk = self.lmk_faces_idx.shape[0]
        m.n_dyn_lmk = self.dynamic_lmk_faces_idx.shape[1]
        m.n_dyn_rows = self.dynamic_lmk_faces_idx.shape[0]
        m.n_neck = self.neck_kin_chain.shape[0]
        for k, v in t.items():
            setattr(m, k, v.data_ptr())
        dm = dict(struct=m, tensors=t, V=V, J=J, NB=NB, NBpad=NBpad, P=P, Ppad=Ppad, nb=nb, ne=ne)
        self._dev[key] = dm
        return dm

    def _gemm(self, lib, stream, inp, K, wgt, N, out, bias=None, res=None):
        d = _lib.ShapyConv()
        d.in_ = inp.data_ptr(); d.wgt = wgt.data_ptr()
        d.bias = bias.data_ptr() if bias is not None else None
        d.res = res.data_ptr() if res is not None else None
        d.out = out.data_ptr()
        d.B = inp.shape[0]; d.Hi = d.Wi = d.Ho = d.Wo = 1
        d.Cin = K; d.in_ld = K; d.Cout = N
        d.ksize = 1; d.stride = 1; d.pad = 0
        d.out_ld = N; d.out_coff = 0; d.res_ld = N; d.res_coff = 0
        d.relu = 0; d.ups = 1; d.tile = 0
        _lib.check(lib.shapy_conv2d(ctypes_byref(d), stream), 'shapy_conv2d (blend shapes)')

    def forward_shape(self, betas=None):
        """SMPL.forward_shape (body_models.py:296-306)."""
        out = self.forward(betas=betas, get_skin=False, return_shaped=True, _shape_only=True)
        return {'vertices': out['v_shaped'], 'betas': betas, 'v_shaped': out['v_shaped']}

    def forward(self, global_rot=None, body_pose=None, left_hand_pose=None, right_hand_pose=None,
                jaw_pose=None, betas=None, expression=None, transl=None, leye_pose=None,
                reye_pose=None, get_skin=True, return_full_pose=False, return_shaped=True,
                _shape_only=False, **kwargs):
        """SMPLX.forward (body_models.py:628-767): poses are rotation matrices [B,k,3,3]."""
        device = self.shapedirs.device
        _lib.require_cuda(self.shapedirs, 'SMPLX buffers')
        model_vars = [betas, global_rot, body_pose, transl, left_hand_pose, right_hand_pose,
                      jaw_pose, leye_pose, reye_pose, expression]
        B = 1
        for var in model_vars:
            if var is not None:
                B = max(B, len(var))
        dm = self._device_model(device)
        f32 = dict(dtype=torch.float32, device=device)

        def eye(n):
            return torch.eye(3, **f32).view(1, 1, 3, 3).expand(B, n, -1, -1)
        parts = [(global_rot, 1), (body_pose, self.NUM_BODY_JOINTS), (jaw_pose, 1),
                 (leye_pose, 1), (reye_pose, 1), (left_hand_pose, self.NUM_HAND_JOINTS),
                 (right_hand_pose, self.NUM_HAND_JOINTS)]
        # joints after the last given part are identity inside the kernel (n_pose)
        last = max([i for i, (p, _) in enumerate(parts) if p is not None], default=-1)
        given = parts[:last + 1]
        n_pose = sum(n for _, n in given)
        # one launch for the whole argument glue (shapy_smplx_prepare_f32) instead of eye / cat /
        # zeros / slice assignments / clone: 8-10 tiny torch kernels per call
        lib = _lib.load()
        keep = []

        def dev_f32(t, shape):
            t = t.reshape(shape)
            if t.shape[0] != B:
                if t.shape[0] != 1:
                    raise ValueError(f'batch size mismatch: {t.shape[0]} vs {B}')
                t = t.expand(B, *t.shape[1:])
            if t.dtype != torch.float32 or not t.is_contiguous() or t.device != device:
                t = t.to(**f32).contiguous()
            keep.append(t)
            return t
        ptrs = (ctypes_vp() * 7)()
        cnts = (ctypes_i32() * 7)()
        for q, (p_, n) in enumerate(given):
            cnts[q] = n
            ptrs[q] = None if p_ is None else dev_f32(p_, (-1, n, 3, 3)).data_ptr()
        NBpad, nb, ne = dm['NBpad'], dm['nb'], dm['ne']
        has_expr = expression is not None
        # one arena: the coefficient rows FIRST -- they are the A operand of the blend-shape GEMMs, whose
        # kernel wants 16-byte-aligned inputs (conv_prepare), and B * NBpad is a multiple of 4 floats; the
        # pose slice behind them starts aligned too and nothing needs to follow it
        n_co = B * NBpad * (2 if has_expr else 1)
        assert NBpad % 4 == 0
        arena = torch.empty(n_co + B * n_pose * 9, **f32)
        coeffs = arena[:B * NBpad].view(B, NBpad)
        coeffs_shape = arena[B * NBpad:n_co].view(B, NBpad) if has_expr else None
        full_pose = arena[n_co:].view(B, n_pose, 3, 3)
        # (reshape by the tensor's OWN batch: a [1, nb] row broadcasts to B in dev_f32)
        betas_t = None if betas is None else dev_f32(betas, (betas.shape[0], -1))
        if betas_t is not None and betas_t.shape[1] != nb:
            raise ValueError(f'betas: expected {nb} coefficients, got {betas_t.shape[1]}')
        expr_t = dev_f32(expression, (expression.shape[0], -1)) if has_expr else None
        if has_expr and expr_t.shape[1] != ne:
            raise ValueError(f'expression: expected {ne} coefficients, got {expr_t.shape[1]}')
        _lib.check(lib.shapy_smplx_prepare_f32(
            ptrs, cnts, len(given), _lib.ptr(betas_t), nb, _lib.ptr(expr_t), ne if has_expr else 0, NBpad,
            _lib.ptr(full_pose) if n_pose else None, _lib.ptr(coeffs), _lib.ptr(coeffs_shape), B,
            _lib.current_stream()), 'shapy_smplx_prepare_f32')
        out = self.forward_prepared(full_pose, coeffs, coeffs_shape=coeffs_shape, transl=transl,
                                    get_skin=get_skin, return_shaped=return_shaped,
                                    _shape_only=_shape_only)
        if return_full_pose and not _shape_only:
            J = dm['J']
            out['full_pose'] = torch.cat([full_pose, eye(J - n_pose)], dim=1) if n_pose < J \
                else full_pose
        return out

    def forward_prepared(self, pose, coeffs, coeffs_shape=None, camera=None, transl=None,
                         get_skin=True, return_shaped=True, _shape_only=False):
        """The numeric core of ``forward`` on prepared inputs -- no torch glue kernels:
          pose    [B, n_pose, 3, 3] contiguous float32 rotation matrices of the first n_pose
                  joints (global, body, jaw, eyes, hands order); the rest are identity
          coeffs  [B, NBpad] float32: betas, then expression coefficients, zero padded
          coeffs_shape  the same with the expression part zeroed, or None when there is no
                  expression (``v_shaped`` is then the first GEMM's result)
          camera  optional [B,3] raw weak-perspective parameters: the landmark kernel then also
                  returns ``proj_joints`` = softplus(c0) * (xy + c[1:3]) and ``cam_scale`` [B,1]
        ``shapy_head_prepare_f32`` produces pose / coeffs / camera for the regressor in one
        launch (HMRLikeRegressor.forward)."""
        device = self.shapedirs.device
        lib = _lib.load()
        stream = _lib.current_stream()
        dm = self._device_model(device)
        m = dm['struct']
        V, J = dm['V'], dm['J']
        B, n_pose = coeffs.shape[0], pose.shape[1]
        f32 = dict(dtype=torch.float32, device=device)
        # ONE allocation for everything the layer writes and ONE C call for all its launches
        # (shapy_smplx_forward_f32): between ctypes calls the host used to lose ~40 us per layer
        n_out = J + m.n_static_lmk + (m.n_dyn_lmk if self.use_face_contour else 0)
        # the fused projection is only valid when nothing edits the joints afterwards
        fuse_cam = (camera is not None and not self.use_joint_regressor and transl is None
                    and not _shape_only)
        sizes = [('v_shaped_full', (B, V, 3)), ('v_shaped', (B, V, 3) if coeffs_shape is not None else None)]
        if not _shape_only:
            sizes += [('rot', (B, J, 3, 3)), ('pf', (B, dm['Ppad'])), ('A', (B, J, 12)),
                      ('posed', (B, J, 3)), ('dyn_row', (B,)), ('v_posed', (B, V, 3)),
                      ('vertices', (B, V, 3)), ('joints', (B, n_out, 3)),
                      ('proj', (B, n_out, 2) if fuse_cam else None),
                      ('scale', (B, 1) if fuse_cam else None)]
        total, offs = 0, {}
        for name, shp in sizes:
            if shp is None:
                continue
            n = 1
            for q in shp:
                n *= q
            offs[name] = (total, n, shp)
            total += (n + 3) // 4 * 4                       # 16-byte aligned slices
        arena = torch.empty(total, **f32)

        def view(name):
            if name not in offs:
                return None
            o, n, shp = offs[name]
            return arena[o:o + n].view(shp)
        bufs = {name: view(name) for name, _ in sizes}
        dyn_row = bufs.get('dyn_row')
        if dyn_row is not None:
            dyn_row = dyn_row.view(torch.int32)
        p = _lib.ptr
        _lib.check(lib.shapy_smplx_forward_f32(
            ctypes_byref(m), p(pose), _lib.POSE_ROTMAT, n_pose, p(coeffs), p(coeffs_shape), p(camera) if fuse_cam else None,
            p(bufs['v_shaped_full']), p(bufs.get('v_shaped')), p(bufs.get('rot')), p(bufs.get('pf')),
            p(bufs.get('A')), p(bufs.get('posed')), p(dyn_row), p(bufs.get('v_posed')),
            p(bufs.get('vertices')), p(bufs.get('joints')), p(bufs.get('proj')), p(bufs.get('scale')),
            B, int(self.use_face_contour), int(_shape_only), stream), 'shapy_smplx_forward_f32')
        v_shaped_full = bufs['v_shaped_full']
        v_shaped = v_shaped_full if coeffs_shape is None else bufs['v_shaped']
        output = defaultdict(lambda: None, faces=self.faces)
        if return_shaped:
            output['v_shaped'] = v_shaped
        if _shape_only:
            return output
        vertices, joints, proj, scale = bufs['vertices'], bufs['joints'], bufs.get('proj'), bufs.get('scale')

        if self.use_joint_regressor:
            Jn = self.extra_joint_regressor.shape[0]
            reg = torch.empty(B, Jn, 3, **f32)
            _lib.check(lib.shapy_joint_regress_f32(
                _lib.ptr(self.extra_joint_regressor.contiguous()), _lib.ptr(vertices),
                _lib.ptr(reg), B, V, Jn, stream), 'shapy_joint_regress_f32')
            joints[:, self.source_idxs] = reg[:, self.target_idxs]
        if transl is not None:
            joints += transl.unsqueeze(dim=1)
            vertices += transl.unsqueeze(dim=1)

        def kpt(t):
            return KeypointTensor(t, source=self.name, keypoint_names=self.keypoint_names,
                                  part_indices=self.parts, connections=self.connections,
                                  part_connections=self.part_connections)
        output['joints'] = kpt(joints)
        if fuse_cam:
            output['proj_joints'] = kpt(proj)
            output['cam_scale'] = scale
        if get_skin:
            output['vertices'] = vertices
        return output


def ctypes_byref(x):
    import ctypes
    return ctypes.byref(x)


def ctypes_vp():
    import ctypes
    return ctypes.c_void_p


def ctypes_i32():
    import ctypes
    return ctypes.c_int32
