"""Body heads (reference: models/body_heads/body_heads.py:28-131,225-253)."""
import math
import os.path as osp

import numpy as np
import torch

from ..body_models import build_body_model
from ..common.iterative_regressor import HMRLikeRegressor
from ..common.pose_utils import build_pose_parameterization
from .registry import BODY_HEAD_REGISTRY

__all__ = ['SMPLRegressor', 'SMPLHRegressor', 'SMPLXRegressor']


@BODY_HEAD_REGISTRY.register()
class SMPLRegressor(HMRLikeRegressor):
    def __init__(self, body_model_cfg, network_cfg, loss_cfg, dtype=torch.float32):
        super().__init__(body_model_cfg, network_cfg, loss_cfg)

    def _build_model(self, body_model_cfg):
        self.body_model_cfg = body_model_cfg
        model = build_body_model(body_model_cfg)
        self.model_type = model.name
        self.curr_model_cfg = body_model_cfg.get(self.model_type, {})
        return model

    def _build_pose_space(self, body_model_cfg):
        param_desc = super()._build_pose_space(body_model_cfg)
        global_rot_desc = build_pose_parameterization(1, **self.curr_model_cfg.global_rot)
        self.global_rot_decoder = global_rot_desc.decoder
        body_pose_desc = build_pose_parameterization(
            num_angles=self.model.num_body_joints,
            mean=self.mean_poses_dict.get('body_pose', None), **self.curr_model_cfg.body_pose)
        self.body_pose_decoder = body_pose_desc.decoder
        global_rot_type = body_model_cfg.get('global_rot', {}).get('param_type', 'cont_rot_repr')
        # rotate the model 180 degrees about x (body_heads.py:103-108)
        global_rot_mean = global_rot_desc.mean
        if global_rot_type == 'aa':
            global_rot_mean[0] = math.pi
        elif global_rot_type == 'cont_rot_repr':
            global_rot_mean[3] = -1
        param_desc.update({'global_rot': global_rot_desc, 'body_pose': body_pose_desc})
        return param_desc

    def _build_blendshape_space(self, body_model_cfg, dtype=torch.float32):
        desc = super()._build_blendshape_space(body_model_cfg, dtype=dtype)
        num_betas = self.model.num_betas
        shape_mean_path = osp.expandvars(body_model_cfg.get('shape_mean_path', '') or '')
        if osp.exists(shape_mean_path):
            shape_mean = torch.from_numpy(np.load(shape_mean_path, allow_pickle=True)).to(
                dtype=dtype).reshape(1, -1)[:, :num_betas].reshape(-1)
        else:
            shape_mean = torch.zeros([num_betas], dtype=dtype)
        desc['betas'] = {'dim': num_betas, 'mean': shape_mean}
        return desc


@BODY_HEAD_REGISTRY.register()
class SMPLHRegressor(SMPLRegressor):
    def __init__(self, body_model_cfg, network_cfg, loss_cfg, dtype=torch.float32):
        self.predict_hands = network_cfg.get('predict_hands', True)
        if self.predict_hands:
            raise NotImplementedError('predict_hands=True is not used by SHAPY_A '
                                      '(configs/b2a_expose_hrnet_demo.yaml:182)')
        super().__init__(body_model_cfg, network_cfg, loss_cfg, dtype=dtype)


@BODY_HEAD_REGISTRY.register()
class SMPLXRegressor(SMPLHRegressor):
    def __init__(self, body_model_cfg, network_cfg, loss_cfg, dtype=torch.float32):
        self.predict_face = network_cfg.get('predict_face', True)
        if self.predict_face:
            raise NotImplementedError('predict_face=True is not used by SHAPY_A '
                                      '(configs/b2a_expose_hrnet_demo.yaml:183)')
        super().__init__(body_model_cfg, network_cfg, loss_cfg, dtype=dtype)
