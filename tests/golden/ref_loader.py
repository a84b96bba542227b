"""Import the reference's own modules from /root/reference on CPU (fixture generation only).

This file is *test infrastructure*: it runs only in the build container, where the
read-only reference tree is mounted.  Nothing under ``shapy_amd/`` imports it and
nothing on the GPU box needs it -- the vectors it helps to produce are committed
under ``tests/golden/``.

The reference (``regressor/human_shape``) imports a number of packages that are not in
this image (loguru, omegaconf, yacs, fvcore, trimesh, torchvision, open3d, ...).  We
pre-seed ``sys.modules`` with inert stand-ins so that the reference's *arithmetic*
modules import unchanged:

  * ``torchvision.models.resnet.{BasicBlock,Bottleneck}`` are third-party code that is
    not under /root/reference (torchvision==0.8.2, reference requirements.txt:24; call
    sites hrnet.py:13,196-199,369-370).  Their published semantics are restated here;
    no reference test pins them -> that sub-boundary is "parity unpinned".
  * ``body_measurements.BodyMeasurements`` is imported from the real reference file,
    with ``mesh_mesh_intersection.MeshMeshIntersection`` (CUDA-only extension) replaced
    by the oracle's brute-force C restatement.
"""
import importlib
import importlib.util
import os
import os.path as osp
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REF_ROOT = os.environ.get('SHAPY_REFERENCE', '/root/reference')
REG_ROOT = osp.join(REF_ROOT, 'regressor')
HS_ROOT = osp.join(REG_ROOT, 'human_shape')
MMI_ROOT = osp.join(REF_ROOT, 'mesh-mesh-intersection')


def available():
    return osp.isdir(HS_ROOT)


class _Logger:
    def __getattr__(self, name):
        def f(*a, **k):
            return None
        return f


class AttrDict(dict):
    """Minimal stand-in for omegaconf.DictConfig: attribute + .get access."""

    def __init__(self, d=None, **kw):
        super().__init__()
        d = dict(d or {}, **kw)
        for k, v in d.items():
            self[k] = AttrDict(v) if isinstance(v, dict) and not isinstance(v, AttrDict) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _pkg(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    sys.modules[name] = m
    return m


# ---- torchvision residual blocks (third-party; restated from the published semantics) ----
def _conv3x3(i, o, s=1):
    return nn.Conv2d(i, o, 3, s, 1, bias=False)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, **kw):
        super().__init__()
        self.conv1 = _conv3x3(inplanes, planes, stride)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = _conv3x3(planes, planes)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        out += identity
        return self.relu(out)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, **kw):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = _conv3x3(planes, planes, stride)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        out += identity
        return self.relu(out)


_INSTALLED = False


def install_stubs(intersect_fn=None):
    """Seed sys.modules so the reference's model code imports on this image."""
    global _INSTALLED
    if _INSTALLED:
        return
    _INSTALLED = True
    if REG_ROOT not in sys.path:
        sys.path.insert(0, REG_ROOT)

    _mod('loguru', logger=_Logger())
    _mod('yacs')
    _mod('yacs.config', CfgNode=AttrDict)
    _mod('omegaconf', DictConfig=AttrDict, OmegaConf=AttrDict)

    class Registry(dict):
        def __init__(self, name=''):
            super().__init__()
            self._name = name

        def register(self, obj=None):
            if obj is None:
                def deco(o):
                    self[o.__name__] = o
                    return o
                return deco
            self[obj.__name__] = obj
            return obj

    _mod('fvcore')
    _mod('fvcore.common')
    _mod('fvcore.common.registry', Registry=Registry)
    for name in ('trimesh', 'open3d', 'jpeg4py', 'cv2', 'pyrender', 'PIL'):
        if name not in sys.modules:
            try:
                importlib.import_module(name)
            except Exception:
                _mod(name)
    _mod('torchvision')
    _mod('torchvision.models')
    _mod('torchvision.models.resnet', BasicBlock=BasicBlock, Bottleneck=Bottleneck,
         ResNet=object, model_urls={})

    # ---- human_shape namespace packages (skip the heavy __init__ files) ----
    _pkg('human_shape', HS_ROOT)
    utils = _pkg('human_shape.utils', osp.join(HS_ROOT, 'utils'))
    for sub in ('typing', 'rotation_utils', 'bool_utils', 'data_structs', 'timer',
                'torch_utils'):
        m = importlib.import_module(f'human_shape.utils.{sub}')
        for k in getattr(m, '__all__', [k for k in dir(m) if not k.startswith('_')]):
            setattr(utils, k, getattr(m, k))

    def to_np(array, dtype=np.float32):
        if 'scipy.sparse' in str(type(array)):
            array = array.todense()
        return np.array(array, dtype=dtype)

    def binarize(array, thresh=-1, dtype=np.float32):
        return (array >= thresh).astype(dtype) if thresh > 0 else (array > 0).astype(dtype)
    utils.to_np = to_np
    utils.binarize = binarize

    _pkg('human_shape.data', osp.join(HS_ROOT, 'data'))
    dutils = _pkg('human_shape.data.utils', osp.join(HS_ROOT, 'data', 'utils'))
    for sub in ('keypoints', 'keypoint_names'):
        m = importlib.import_module(f'human_shape.data.utils.{sub}')
        for k in dir(m):
            if not k.startswith('_'):
                setattr(dutils, k, getattr(m, k))
    _mod('human_shape.data.structures', StructureList=list)
    _mod('human_shape.losses', build_loss=lambda *a, **k: None,
         build_adv_loss=lambda *a, **k: None, build_prior=lambda *a, **k: None)

    class _Placeholder:
        pass
    _mod('attributes', A2B=_Placeholder, B2A=_Placeholder)

    # ---- measurement module: real reference consumer + oracle intersection op ----
    class MeshMeshIntersection(nn.Module):
        def __init__(self, max_collisions=32):
            super().__init__()
            self.max_collisions = max_collisions

        def forward(self, query_triangles, target_triangles, print_timings=False):
            if intersect_fn is None:
                raise RuntimeError('no intersection oracle installed')
            f, b = intersect_fn(query_triangles.detach().cpu().numpy(),
                                target_triangles.detach().cpu().numpy(),
                                self.max_collisions)
            return torch.from_numpy(f), torch.from_numpy(b)

    _mod('mesh_mesh_intersection', MeshMeshIntersection=MeshMeshIntersection)
    spec = importlib.util.spec_from_file_location(
        'body_measurements',
        osp.join(MMI_ROOT, 'body_measurements', 'body_measurements.py'))
    bm = importlib.util.module_from_spec(spec)
    sys.modules['body_measurements'] = bm
    spec.loader.exec_module(bm)

    _pkg('human_shape.models', osp.join(HS_ROOT, 'models'))


def load_reference(intersect_fn=None):
    """Returns a namespace with the reference's classes/functions used on the path."""
    install_stubs(intersect_fn)
    ns = types.SimpleNamespace()
    ns.hrnet = importlib.import_module('human_shape.models.backbone.hrnet')
    ns.networks = importlib.import_module('human_shape.models.common.networks')
    ns.pose_utils = importlib.import_module('human_shape.models.common.pose_utils')
    ns.rotation_utils = importlib.import_module('human_shape.utils.rotation_utils')
    ns.lbs = importlib.import_module('human_shape.models.body_models.lbs')
    ns.body_models = importlib.import_module('human_shape.models.body_models.body_models')
    ns.camera = importlib.import_module('human_shape.models.camera.camera_projection')
    ns.body_heads = importlib.import_module('human_shape.models.body_heads')
    ns.iterative_regressor = importlib.import_module(
        'human_shape.models.common.iterative_regressor')
    ns.body_measurements = sys.modules['body_measurements']
    ns.AttrDict = AttrDict
    # the training-only loss builders are out of scope
    ns.iterative_regressor.HMRLikeRegressor._build_losses = lambda self, cfg: None
    return ns
