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
        Ppad = _round_up(P, 32)     # 128-byte K chunks of the pose-blend GEMM (csrc/conv_igemm.hip: KQ = 8)
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
        """SMPLX.forward (body_models.py:628-767): poses are rotation matrices [B,k,3,3].

        The call is host-bound (B = 4 and B = 64 take the same time), so the path below avoids torch
        ops: float32 device tensors whose joints are contiguous go to the glue kernel as they are --
        pointer + batch stride; slices like ``rot[:, 1:]`` are not copied, batch-1 tensors are
        broadcast with stride 0 -- and everything else takes the converting route (``dev``)."""
        device = self.shapedirs.device
        _lib.require_cuda(self.shapedirs, 'SMPLX buffers')
        parts = ((global_rot, 1), (body_pose, self.NUM_BODY_JOINTS), (jaw_pose, 1), (leye_pose, 1),
                 (reye_pose, 1), (left_hand_pose, self.NUM_HAND_JOINTS), (right_hand_pose, self.NUM_HAND_JOINTS))
        # the reference views every part as reshape(-1, n, 3, 3) (body_models.py:676-690): a part given as
        # [B * n, 3, 3] or [B, n * 9] is the same pose
        parts = tuple((p_ if p_ is None or (p_.dim() == 4 and tuple(p_.shape[1:]) == (n, 3, 3))
                       else p_.reshape(-1, n, 3, 3), n) for p_, n in parts)
        B = 1
        for var in (betas, transl, expression):
            if var is not None and var.shape[0] > B:
                B = var.shape[0]
        last = -1                                   # joints after the last given part are identity in the kernel
        for i, (p_, _) in enumerate(parts):
            if p_ is not None:
                last = i
                if p_.shape[0] > B:
                    B = p_.shape[0]
        dm = self._device_model(device)
        lib = _lib.load()
        keep = []

        def dev(t, inner):
            """(pointer, floats between bodies) of t viewed as [B or 1, *inner]."""
            n = 1
            for q in inner:
                n *= q
            ok = (t.dtype == torch.float32 and t.device == device and t.dim() == len(inner) + 1
                  and tuple(t.shape[1:]) == inner)
            if ok:                                  # inner dims contiguous?
                st, expect = t.stride(), 1
                for d in range(len(inner), 0, -1):
                    if t.shape[d] != 1 and st[d] != expect:
                        ok = False
                        break
                    expect *= t.shape[d]
            if not ok:
                t = t.reshape(-1, *inner).to(dtype=torch.float32, device=device).contiguous()
                keep.append(t)
            if t.shape[0] == B and B > 1:
                return t.data_ptr(), t.stride(0)
            if t.shape[0] != 1:
                raise ValueError(f'batch size mismatch: {t.shape[0]} vs {B}')
            return t.data_ptr(), 0
        ptrs = (ctypes_vp() * 7)()
        cnts = (ctypes_i32() * 7)()
        strides = (ctypes_i64() * 7)()
        n_pose = 0
        for q in range(last + 1):
            p_, n = parts[q]
            cnts[q] = n
            n_pose += n
            if p_ is not None:
                ptrs[q], strides[q] = dev(p_, (n, 3, 3))
        NBpad, nb, ne = dm['NBpad'], dm['nb'], dm['ne']
        has_expr = expression is not None
        # one arena: the coefficient rows FIRST -- they are the A operand of the blend-shape GEMMs, whose
        # kernel wants 16-byte-aligned inputs (conv_prepare), and B * NBpad is a multiple of 4 floats; the
        # pose slice behind them starts aligned too and nothing needs to follow it
        n_co = B * NBpad * (2 if has_expr else 1)
        arena = torch.empty(n_co + B * n_pose * 9, dtype=torch.float32, device=device)
        if betas is not None and betas.shape[-1] != nb and betas.numel() != betas.shape[0] * nb:
            raise ValueError(f'betas: expected {nb} coefficients, got {betas.numel() // betas.shape[0]}')
        if has_expr and expression.numel() != expression.shape[0] * ne:
            raise ValueError(f'expression: expected {ne} coefficients, got {expression.numel() // expression.shape[0]}')
        bp, bst = dev(betas, (nb,)) if betas is not None else (None, 0)
        ep, est = dev(expression, (ne,)) if has_expr else (None, 0)
        base = arena.data_ptr()
        _lib.check(lib.shapy_smplx_prepare_f32(
            ptrs, cnts, strides, last + 1, bp, bst, nb, ep, est, ne if has_expr else 0, NBpad,
            base + 4 * n_co if n_pose else None, base, base + 4 * B * NBpad if has_expr else None, B,
            _lib.current_stream()), 'shapy_smplx_prepare_f32')
        out = self._run(dm, B, n_pose, base + 4 * n_co, base, base + 4 * B * NBpad if has_expr else None,
                        None, transl, get_skin, return_shaped, _shape_only, (arena,))
        if return_full_pose and not _shape_only:
            J = dm['J']
            full_pose = arena[n_co:].view(B, n_pose, 3, 3)
            if n_pose < J:
                eye = torch.eye(3, dtype=torch.float32, device=device).view(1, 1, 3, 3).expand(B, J - n_pose, -1, -1)
                out['full_pose'] = torch.cat([full_pose, eye], dim=1)
            else:
                out['full_pose'] = full_pose
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
        dm = self._device_model(self.shapedirs.device)
        return self._run(dm, coeffs.shape[0], pose.shape[1], pose.data_ptr(), coeffs.data_ptr(),
                         None if coeffs_shape is None else coeffs_shape.data_ptr(), camera, transl,
                         get_skin, return_shaped, _shape_only, (pose, coeffs, coeffs_shape))

    def _run(self, dm, B, n_pose, pose_ptr, coeffs_ptr, coeffs_shape_ptr, camera, transl, get_skin,
             return_shaped, _shape_only, _keep):
        """forward_prepared on raw device pointers (`_keep`: the tensors behind them)."""
        device = self.shapedirs.device
        lib = _lib.load()
        stream = _lib.current_stream()
        m = dm['struct']
        V, J = dm['V'], dm['J']
        f32 = dict(dtype=torch.float32, device=device)
        has_cs = coeffs_shape_ptr is not None
        # ONE allocation for everything the layer writes and ONE C call for all its launches
        # (shapy_smplx_forward_f32): between ctypes calls the host used to lose ~40 us per layer
        n_out = J + m.n_static_lmk + (m.n_dyn_lmk if self.use_face_contour else 0)
        # the fused projection is only valid when nothing edits the joints afterwards
        fuse_cam = (camera is not None and not self.use_joint_regressor and transl is None
                    and not _shape_only)
        # (v_shaped directly behind v_shaped_full, no alignment gap: with adjacent coefficient rows the
        # library then runs both shape GEMMs as one launch with M = 2 B)
        sizes = [('v_shaped_pair', (2 * B, V, 3) if has_cs else (B, V, 3))]
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

        def view(name, skip=0):
            if name not in offs:
                return None
            o, n, shp = offs[name]
            if skip:                                     # second half of the pair
                shp = (shp[0] - skip,) + tuple(shp[1:])
                o += skip * shp[1] * shp[2]
            st, acc = [], 1
            for d in reversed(shp):
                st.append(acc)
                acc *= d
            return arena.as_strided(shp, st[::-1], o)     # one torch op per returned tensor
        # raw pointers for everything the layer only uses internally (a torch view costs ~2 us of host
        # time each and the call is host-bound); views only for what the caller gets back
        base = arena.data_ptr()

        def p(name):
            return ctypes_vp()(base + 4 * offs[name][0]) if name in offs else None
        pair_off = offs['v_shaped_pair'][0]
        p_vsf = ctypes_vp()(base + 4 * pair_off)
        p_vs = ctypes_vp()(base + 4 * (pair_off + B * V * 3)) if has_cs else None
        _lib.check(lib.shapy_smplx_forward_f32(
            ctypes_byref(m), pose_ptr, _lib.POSE_ROTMAT, n_pose, coeffs_ptr, coeffs_shape_ptr,
            _lib.ptr(camera) if fuse_cam else None,
            p_vsf, p_vs, p('rot'), p('pf'), p('A'), p('posed'), p('dyn_row'), p('v_posed'),
            p('vertices'), p('joints'), p('proj'), p('scale'),
            B, int(self.use_face_contour), int(_shape_only), stream), 'shapy_smplx_forward_f32')
        output = defaultdict(lambda: None, faces=self.faces)
        if return_shaped:
            output['v_shaped'] = view('v_shaped_pair', B if has_cs else 0)
        if _shape_only:
            return output
        vertices, joints, proj, scale = view('vertices'), view('joints'), view('proj'), view('scale')

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


def ctypes_i64():
    import ctypes
    return ctypes.c_int64
