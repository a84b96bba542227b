"""HMRLikeRegressor -- the hot-path orchestrator on gfx950.

Drop-in for regressor/human_shape/models/common/iterative_regressor.py:39-870 (inference
path): same constructor arguments, same registered buffers (``<name>_idxs``, ``<name>_mean``,
``param_mean``), same ``forward(images, targets, compute_losses, cond, extra_features)``
and the same output dict (``stage_00..``, ``stage_keys``, ``num_stages``, ``features``,
``proj_joints``, ``camera_parameters``, ``measurements``, ``losses``).

Everything numeric runs in libshapy_hip.so; torch is used for buffer allocation, slicing
views and the output containers only.  The training-only parts of the reference (losses,
priors; iterative_regressor.py:251-581) are out of scope.
"""
import os.path as osp
import pickle
from collections import defaultdict

import numpy as np
import torch
import torch.nn as nn

from ... import _lib
from ...measurements import BodyMeasurements
from ..backbone import build_backbone
from ..body_models import KeypointTensor
from ..camera import CameraParams, build_cam_proj
from .networks import build_regressor


class HMRLikeRegressor(nn.Module):
    def __init__(self, body_model_cfg, network_cfg, loss_cfg, dtype=torch.float32):
        super().__init__()
        self.pose_last_stage = network_cfg.get('pose_last_stage', True)
        camera_cfg = network_cfg.get('camera', {})
        camera_data = build_cam_proj(camera_cfg, dtype=dtype)
        self.projection = camera_data['camera']
        self.camera_scale_func = camera_data['scale_func']
        camera_space = {'dim': camera_data['dim'], 'mean': camera_data['mean']}

        self.model = self._build_model(body_model_cfg)
        pose_space = self._build_pose_space(body_model_cfg)
        blendshape_space = self._build_blendshape_space(body_model_cfg)
        appearance_space = self._build_appearance_space(body_model_cfg)
        self.pose_space, self.blendshape_space = pose_space, blendshape_space
        self.appearance_space = appearance_space
        param_dict = dict(**pose_space, **blendshape_space, **appearance_space)
        param_dict['camera'] = camera_space

        mean_lst, start = [], 0
        for name, desc in param_dict.items():
            indices = torch.tensor(list(range(start, start + desc['dim'])), dtype=torch.long)
            self.register_buffer(f'{name}_idxs', indices)
            mean_lst.append(desc['mean'].view(-1))
            start += desc['dim']
            self.register_buffer(f'{name}_mean', desc['mean'])
        self.param_names = list(param_dict.keys())
        self._slices = {}
        s = 0
        for name, desc in param_dict.items():
            self._slices[name] = (s, s + desc['dim'])
            s += desc['dim']
        param_mean = torch.cat(mean_lst).view(1, -1)
        self._param_dim = param_mean.numel()
        self.register_buffer('param_mean', param_mean)

        backbone_cfg = network_cfg.get('backbone', {})
        self.backbone, feat_dims = build_backbone(backbone_cfg)
        self.feature_key = network_cfg.get('feature_key', 'avg_pooling')
        self._feat_dim = feat_dims[self.feature_key]
        self.regressor, self._num_stages = build_regressor(
            network_cfg, self._feat_dim, self._param_dim, param_mean=param_mean)

        expand = lambda p: osp.expandvars(p or '')
        meas_definition_path = expand(network_cfg.get('meas_definition_path', ''))
        meas_vertices_path = expand(network_cfg.get('meas_vertices_path', ''))
        compute_measurements = network_cfg.get('compute_measurements', False)
        self.compute_measurements = bool(
            compute_measurements and osp.exists(meas_definition_path) and
            osp.exists(meas_vertices_path))
        if self.compute_measurements:
            self.body_measurements = BodyMeasurements(
                {'meas_definition_path': meas_definition_path,
                 'meas_vertices_path': meas_vertices_path})

        use_b2a = network_cfg.get('use_b2a', False)
        b2a_m = expand(network_cfg.get('b2a_males_checkpoint', ''))
        b2a_f = expand(network_cfg.get('b2a_females_checkpoint', ''))
        self.use_b2a = bool(use_b2a and osp.exists(b2a_m) and osp.exists(b2a_f))
        if self.use_b2a:                                   # iterative_regressor.py:146-172
            from ..attributes import B2A
            self.b2a_males = B2A.load_from_checkpoint(b2a_m)
            self.b2a_females = B2A.load_from_checkpoint(b2a_f)
            for p in list(self.b2a_males.parameters()) + list(self.b2a_females.parameters()):
                p.requires_grad = False
        use_a2b = network_cfg.get('use_a2b', False)
        self.num_attributes = network_cfg.get('num_attributes', False)
        self.use_a2b = bool(use_a2b and osp.exists(expand(network_cfg.get('a2b_males_checkpoint', '')))
                            and osp.exists(expand(network_cfg.get('a2b_females_checkpoint', ''))))
        if self.use_a2b:
            raise NotImplementedError(
                'the A2B refinement head (normalising flows / MLP Lightning modules) is a separate '
                'demo of the reference, not part of the image->shape hot path (SURVEY.md section 2)')
        self._faces_i32 = {}

    # ---- reference properties ----
    param_dim = property(lambda self: self._param_dim)
    feat_dim = property(lambda self: self._feat_dim)
    num_stages = property(lambda self: self._num_stages)
    num_betas = property(lambda self: self.model.num_betas)

    @property
    def num_expression_coeffs(self):
        return getattr(self.model, 'num_expression_coeffs', 0)

    def get_mean(self):
        return self.param_mean

    def _build_pose_space(self, body_model_cfg):
        mean_pose_path = osp.expandvars(self.curr_model_cfg.mean_pose_path or '')
        self.mean_poses_dict = {}
        if osp.exists(mean_pose_path):
            with open(mean_pose_path, 'rb') as f:
                self.mean_poses_dict = pickle.load(f)
        return {}

    def _build_appearance_space(self, body_model_cfg, dtype=torch.float32):
        return {}

    def _build_blendshape_space(self, body_model_cfg, dtype=torch.float32):
        return {}

    def flat_params_to_dict(self, param_tensor):
        """iterative_regressor.py:241-249 (the index buffers are contiguous ranges, so the
        index_select is a slice)."""
        out = {}
        for name in self.param_names:
            a, b = self._slices[name]
            out[name] = param_tensor[:, a:b]
        return out

    accepts_next_images = True       # forward(..., next_images=): evaluation.Evaluator.run looks one batch ahead

    def compute_features(self, images, extra_features=None, next_images=None):
        if next_images is not None:             # software pipelining of consecutive batches (backbone/prefetch.py)
            return self.backbone(images, prefetch=next_images)[self.feature_key]
        return self.backbone(images)[self.feature_key]

    def _faces(self, device):
        key = str(device)
        if key not in self._faces_i32:
            self._faces_i32[key] = self.model.faces_tensor.to(device=device, dtype=torch.int32).contiguous()
        return self._faces_i32[key]

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._faces_i32 = {}
        self.__dict__.pop('_fast_layout', None)
        return out

    def __getstate__(self):
        st = self.__dict__.copy()
        st['_faces_i32'] = {}
        return st

    def _fast_head_layout(self):
        """Parameter layout for the one-launch decode (shapy_head_prepare_f32), or None when the
        configuration needs the generic path: the pose parameters must be [global_rot |
        body_pose] back to back with the same parameterisation, only the last stage is posed,
        and betas / camera are plain slices."""
        lay = self.__dict__.get('_fast_layout', False)
        if lay is not False:
            return lay
        lay = None
        dec = [n for n in self.param_names if hasattr(self, f'{n}_decoder')]
        if (self.pose_last_stage and dec == ['global_rot', 'body_pose']
                # ONLY these four: anything else in param_names (expression, jaw / hand poses,
                # transl) is an argument of self.model(**merged_params) on the generic path and
                # would be dropped silently here; the betas slice must be the model's own width
                # (the generic path raises on a mismatch instead of truncating)
                and set(self.param_names) == {'global_rot', 'body_pose', 'betas', 'camera'}
                and self._slices['betas'][1] - self._slices['betas'][0]
                == min(self.model.num_betas, self.model.SHAPE_SPACE_DIM)
                and self._slices['global_rot'][1] == self._slices['body_pose'][0]
                and hasattr(self.model, 'forward_prepared')):
            g, b = self.global_rot_decoder, self.body_pose_decoder
            types = {'cont_rot_repr': (_lib.POSE_CONT6D, 6), 'aa': (_lib.POSE_AXIS_ANGLE, 3)}
            if g.get_type() == b.get_type() and g.get_type() in types:
                pose_type, per = types[g.get_type()]
                a0, a1 = self._slices['global_rot'][0], self._slices['body_pose'][1]
                import torch.nn.functional as F
                from ..camera.camera_projection import WeakPerspectiveCamera
                lay = dict(pose_type=pose_type, pose_off=a0, n_joints=(a1 - a0) // per,
                           betas=self._slices['betas'], cam_off=self._slices['camera'][0],
                           fuse_camera=(self.camera_scale_func is F.softplus
                                        and isinstance(self.projection, WeakPerspectiveCamera)
                                        and not self.projection.scale_first
                                        and self._slices['camera'][1] - self._slices['camera'][0] == 3))
                if lay['n_joints'] != 1 + self.model.NUM_BODY_JOINTS:
                    lay = None
        self.__dict__['_fast_layout'] = lay
        return lay

    def _decode_all_stages(self, lay):
        out3 = self.regressor.last_output                       # [S,B,P]
        S, B, P = out3.shape
        dm = self.model._device_model(out3.device)
        nj = lay['n_joints']
        f32 = dict(dtype=torch.float32, device=out3.device)
        rot_all = torch.empty(S, B, nj, 3, 3, **f32)
        coeffs = torch.empty(B, dm['NBpad'], **f32)
        cam = torch.empty(B, 3, **f32)
        b0, b1 = lay['betas']
        assert b1 - b0 == dm['nb'], (b0, b1, dm['nb'])     # _fast_head_layout checked it
        _lib.check(_lib.load().shapy_head_prepare_f32(
            _lib.ptr(out3), S, B, P, lay['pose_off'], nj, lay['pose_type'], b0, b1 - b0,
            dm['NBpad'], lay['cam_off'], _lib.ptr(rot_all), _lib.ptr(coeffs), _lib.ptr(cam),
            _lib.current_stream()), 'shapy_head_prepare_f32')
        param_dicts = []
        for s in range(S):
            curr = self.flat_params_to_dict(out3[s])
            d = {}
            for key, val in curr.items():
                if key == 'global_rot':
                    d[key], d['raw_global_rot'] = rot_all[s][:, :1], val
                elif key == 'body_pose':
                    d[key], d['raw_body_pose'] = rot_all[s][:, 1:], val
                else:
                    d[key] = val
            param_dicts.append(d)
        return param_dicts, rot_all[S - 1], coeffs, cam

    def forward(self, images, targets=None, compute_losses=True, cond=None, extra_features=None,
                next_images=None, **kwargs):
        """``next_images`` (not in the reference): the NEXT batch of the loop, already on the device; its stem +
        layer1 run under this batch's head (bit-identical outputs; backbone/prefetch.py)."""
        batch_size = len(images)
        features = self.compute_features(images, extra_features=extra_features, next_images=next_images)
        regr_output = self.regressor(features, cond=cond, extra_features=extra_features)
        parameters = [regr_output] if torch.is_tensor(regr_output) else regr_output[0]

        fused_camera = None
        fast = self._fast_head_layout()
        if fast is not None and torch.is_tensor(getattr(self.regressor, 'last_output', None)) \
                and len(parameters) == self.regressor.last_output.shape[0] \
                and parameters[0].data_ptr() == self.regressor.last_output.data_ptr():
            # one launch for the decode of every stage + the SMPL-X argument glue
            # (shapy_head_prepare_f32); raw_* are views of the parameter tensor (the reference
            # returns clones of the same values, iterative_regressor.py:653-656)
            param_dicts, rot_last, coeffs, cam = self._decode_all_stages(fast)
            num_stages = len(param_dicts)
            merged_params = param_dicts[-1]
            model_output = self.model.forward_prepared(
                rot_last, coeffs, camera=cam if fast['fuse_camera'] else None, get_skin=True,
                return_shaped=True)
            if model_output.get('proj_joints') is not None:
                fused_camera = (model_output.pop('proj_joints'), model_output.pop('cam_scale'))
        else:
            param_dicts = []
            for params in parameters:
                curr = self.flat_params_to_dict(params)
                out_dict = {}
                for key, val in curr.items():
                    if hasattr(self, f'{key}_decoder'):
                        out_dict[key] = getattr(self, f'{key}_decoder')(val)
                        out_dict[f'raw_{key}'] = val.clone()
                    else:
                        out_dict[key] = val
                param_dicts.append(out_dict)
            num_stages = len(param_dicts)
            if self.pose_last_stage:
                merged_params = param_dicts[-1]
            else:
                merged_params = {key: torch.cat([pd[key] for pd in param_dicts if pd[key] is not None],
                                                dim=0) for key in param_dicts[0].keys()}
            model_output = self.model(get_skin=True, return_shaped=True, **merged_params)

        out_params = defaultdict(lambda: dict())
        for key in model_output:
            val = model_output[key]
            if isinstance(val, KeypointTensor):
                out_list = torch.split(val._t, batch_size, dim=0)
                if len(out_list) == num_stages:
                    for ii, value in enumerate(out_list):
                        out_params[f'stage_{ii:02d}'][key] = KeypointTensor.from_obj(value, val)
                else:
                    out_params[f'stage_{num_stages - 1:02d}'][key] = KeypointTensor.from_obj(
                        out_list[-1], val)
            elif torch.is_tensor(val):
                out_list = torch.split(val, batch_size, dim=0)
                if len(out_list) == num_stages:
                    for ii, value in enumerate(out_list):
                        out_params[f'stage_{ii:02d}'][key] = value
                else:
                    out_params[f'stage_{num_stages - 1:02d}'][key] = out_list[-1]

        camera_params = param_dicts[-1]['camera']
        translation = camera_params[:, 1:3]
        if fused_camera is not None:          # computed by the landmark kernel
            proj_joints, scale = fused_camera
        else:
            scale = self.camera_scale_func(camera_params[:, 0].reshape(-1, 1))
            est_joints3d = out_params[f'stage_{num_stages - 1:02d}']['joints']
            proj_joints = self.projection(est_joints3d, scale=scale, translation=translation)

        out_params['proj_joints'] = proj_joints
        out_params['num_stages'] = num_stages
        out_params['features'] = features
        out_params['camera_parameters'] = CameraParams(
            translation=translation, scale=scale,
            scale_first=getattr(self.projection, 'scale_first', False))

        stage_keys = []
        for n in range(num_stages):
            stage_key = f'stage_{n:02d}'
            stage_keys.append(stage_key)
            out_params[stage_key]['faces'] = model_output['faces']
            out_params[stage_key].update(param_dicts[n])

        if self.compute_measurements:
            last_stage_key = f'stage_{num_stages - 1:02d}'
            # iterative_regressor.py:742-756 gathers v_shaped[:, faces] ([B,F,3,3]) first; the
            # fused kernel reads v_shaped + the int32 face table directly.
            v_shaped = out_params[last_stage_key]['v_shaped']
            measurements = self.body_measurements.measure_vertices(
                v_shaped, self._faces(v_shaped.device))['measurements']
            meas_dict = {name: d['tensor'] for name, d in measurements.items()}
            out_params[last_stage_key].update(measurements=meas_dict)
            out_params.update(measurements=meas_dict)

        out_params['stage_keys'] = stage_keys
        out_params[stage_keys[-1]]['proj_joints'] = proj_joints

        if self.use_b2a:                                   # iterative_regressor.py:761-776
            genders = [x.get_field('gender') if x.has_field('gender') else None
                       for x in targets] if targets else [None] * batch_size
            genders = np.array([g.lower()[0] if (g is not None and g != '') else 'n'
                                for g in genders])
            betas = parameters[-1][:, self._slices['betas'][0]:self._slices['betas'][1]]
            attributes = torch.zeros(betas.shape[0], self.b2a_males.b2a.output_dim,
                                     device=betas.device)
            for g, module in (('m', self.b2a_males), ('f', self.b2a_females)):
                idx = torch.from_numpy(np.where(genders == g)[0]).to(betas.device)
                if len(idx):
                    attributes[idx] = module(betas[idx])
            out_params['attributes'] = attributes
        if self.training and compute_losses:
            raise NotImplementedError('training losses are out of scope (SURVEY.md section 2)')
        out_params['losses'] = {}
        return out_params
