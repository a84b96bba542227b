from .body_models import SMPLX
from .build import build_body_model
from .utils import KeypointTensor, find_joint_kin_chain
