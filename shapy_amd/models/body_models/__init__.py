"""SMPL-X layer (blend shapes, joints, kinematic chain, skinning, landmarks) on the GPU."""
from .build import build_body_model
from .body_models import SMPLX
from .utils import KeypointTensor, find_joint_kin_chain

__all__ = ['SMPLX', 'build_body_model', 'KeypointTensor', 'find_joint_kin_chain']
