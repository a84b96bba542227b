"""build_body_model (reference: models/body_models/build.py:10-25)."""
import os.path as osp

from .body_models import SMPLX


def build_body_model(body_model_cfg):
    model_type = body_model_cfg.get('type', 'smplx')
    model_folder = osp.expandvars(body_model_cfg.get('model_folder', 'data/models'))
    curr_model_cfg = body_model_cfg.get(model_type, {})
    model_path = osp.join(model_folder, model_type)
    if model_type.lower() == 'smplx':
        return SMPLX(model_path, **curr_model_cfg)
    if model_type.lower() in ('smpl', 'smplh'):
        raise NotImplementedError(
            f'{model_type}: only the SMPL-X body model is on the SHAPY hot path (SURVEY.md 8a)')
    raise ValueError(f'Unknown model type {model_type}, exiting!')
