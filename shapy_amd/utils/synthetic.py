"""Seeded synthetic inputs for the SHAPY hot path (no licensed assets needed).

There is no SMPL-X model file and no SHAPY_A checkpoint in the image (licensed
downloads, reference ``data/download_data.sh``), so parity tests, ``smoke()`` and
``bench.py`` run on

  * a synthetic ``SMPLX_NEUTRAL.npz`` with the *real* SMPL-X topology (faces and a real
    ``v_shaped`` taken from the reference's shipped sample
    ``samples/shapy_fit_for_virtual_measurements/img_00.npz``), the real kinematic tree and
    seeded smooth blend shapes -- the keys are the ones read at
    ``regressor/human_shape/models/body_models/body_models.py:112-166,433-437,543-597``;
  * seeded O(1)-scale network weights.  The reference's own init (``hrnet.py:500-516``,
    conv ~ N(0, 0.001^2)) makes the backbone output ~0 and a parity test vacuous
    (SURVEY.md F8), so convs get He-normal weights and BN gets random affine + running
    statistics.

Every tensor is drawn from ``numpy.random.Generator(PCG64)`` seeded by
``(seed, crc32(tensor_name))`` so that the values depend only on the tensor's name and
shape: the same function fills the reference module (fixture generation), the CPU oracle
and the HIP path, on any machine.
"""
import os.path as osp
import zlib

import numpy as np

DATA_DIR = osp.join(osp.dirname(osp.dirname(osp.abspath(__file__))), 'data')

# SMPL-X kinematic tree (kintree_table[0]); joints: 0-21 body, 22 jaw, 23/24 eyes,
# 25-39 left hand, 40-54 right hand.
SMPLX_PARENTS = [
    -1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19,
    15, 15, 15,
    20, 25, 26, 20, 28, 29, 20, 31, 32, 20, 34, 35, 20, 37, 38,
    21, 40, 41, 21, 43, 44, 21, 46, 47, 21, 49, 50, 21, 52, 53]

NUM_VERTS = 10475
NUM_FACES = 20908
NUM_JOINTS = 55


def rng_for(seed, name):
    return np.random.Generator(np.random.PCG64([int(seed), zlib.crc32(name.encode())]))


def load_topology():
    """faces int32 [20908,3] and 4 real SHAPY v_shaped meshes f32 [4,10475,3]."""
    d = np.load(osp.join(DATA_DIR, 'smplx_topology.npz'))
    return d['faces'], d['v_shaped']


def keypoint_names():
    with open(osp.join(DATA_DIR, 'smplx_keypoint_names.txt')) as f:
        return [l.strip() for l in f if l.strip()]


def make_synthetic_smplx(seed=0, num_shape=400):
    """Returns a dict with the keys of SMPLX_NEUTRAL.npz (float32/int64 arrays)."""
    faces, meshes = load_topology()
    vt = meshes[0].astype(np.float64)
    V = vt.shape[0]
    ctr = vt.mean(0)
    g = lambda n: rng_for(seed, 'smplx.' + n)

    # smooth shape blend shapes: local affine warps modulated by a low-frequency wave
    r = g('shapedirs')
    M = r.normal(size=(num_shape, 3, 3))
    k = r.normal(size=(num_shape, 3)) * 3.0
    ph = r.uniform(0, 2 * np.pi, size=(num_shape,))
    amp = 0.03 * (0.6 + 0.4 * r.uniform(size=(num_shape,)))
    shapedirs = np.empty((V, 3, num_shape), np.float32)
    rel = vt - ctr
    for l in range(num_shape):
        w = np.cos(rel @ k[l] + ph[l])
        shapedirs[:, :, l] = (amp[l] * (rel @ M[l].T) * w[:, None]).astype(np.float32)

    posedirs = (g('posedirs').standard_normal((V, 3, 486), dtype=np.float32) *
                np.float32(1e-3))

    # joint regressor: each joint = normalised gaussian blob around a seed vertex
    r = g('J_regressor')
    centres = r.choice(V, size=NUM_JOINTS, replace=False)
    J_regressor = np.zeros((NUM_JOINTS, V), np.float32)
    for j, c in enumerate(centres):
        d2 = ((vt - vt[c]) ** 2).sum(1)
        idx = np.argsort(d2)[:32]
        w = np.exp(-d2[idx] / (0.04 ** 2)) + 1e-3
        J_regressor[j, idx] = (w / w.sum()).astype(np.float32)
    joints = J_regressor.astype(np.float64) @ vt

    # skinning weights: 4 nearest joints, soft assignment
    d2 = ((vt[:, None, :] - joints[None]) ** 2).sum(-1)            # V x 55
    near = np.argsort(d2, axis=1)[:, :4]
    wv = np.exp(-np.take_along_axis(d2, near, 1) / (0.08 ** 2)) + 1e-4
    wv /= wv.sum(1, keepdims=True)
    weights = np.zeros((V, NUM_JOINTS), np.float32)
    np.put_along_axis(weights, near, wv.astype(np.float32), 1)

    r = g('landmarks')
    F = faces.shape[0]

    def bary(shape):
        b = r.dirichlet(np.ones(3), size=shape)
        return b.astype(np.float32)

    kintree = np.stack([np.array(SMPLX_PARENTS, np.int64), np.arange(NUM_JOINTS)])
    kintree[0, 0] = 2 ** 32 - 1            # as stored in the real file (uint32 -1)
    r2 = g('hands')
    return {
        'f': faces.astype(np.int64),
        'v_template': vt.astype(np.float32),
        'shapedirs': shapedirs,
        'posedirs': posedirs,
        'J_regressor': J_regressor,
        'kintree_table': kintree,
        'weights': weights,
        'lmk_faces_idx': r.integers(0, F, size=51).astype(np.int64),
        'lmk_bary_coords': bary(51),
        'dynamic_lmk_faces_idx': r.integers(0, F, size=(79, 17)).astype(np.int64),
        'dynamic_lmk_bary_coords': bary((79, 17)),
        'hands_meanl': (r2.standard_normal(45) * 0.1).astype(np.float32),
        'hands_meanr': (r2.standard_normal(45) * 0.1).astype(np.float32),
        'hands_componentsl': np.linalg.qr(r2.standard_normal((45, 45)))[0].astype(np.float32),
        'hands_componentsr': np.linalg.qr(r2.standard_normal((45, 45)))[0].astype(np.float32),
    }


def write_synthetic_smplx(folder, seed=0):
    """Writes <folder>/smplx/SMPLX_NEUTRAL.npz (the layout body_models/build.py:21-25 expects)."""
    import os
    out = osp.join(folder, 'smplx')
    os.makedirs(out, exist_ok=True)
    path = osp.join(out, 'SMPLX_NEUTRAL.npz')
    if not osp.exists(path):
        np.savez(path, **make_synthetic_smplx(seed))
    return path


# ------------------------------------------------------------------------------------------
# network weights
# ------------------------------------------------------------------------------------------
#: BN scale of the *last* BN of every residual block / fuse term.  <1 keeps the residual
#: stream O(1) through the ~45 sequential blocks of HRNet-W48 (He-init alone doubles the
#: variance per block).
RESIDUAL_GAMMA = 0.2
#: synthetic gain of the regressor's output layer (reference default 0.01,
#: networks.py:378-382, would make every prediction ~= the mean and the test vacuous).
OUTPUT_GAIN = 0.12


def synth_tensor(seed, name, shape, kind, **kw):
    """kind in {'conv','bn_gamma','bn_beta','bn_mean','bn_var','bias','linear','linear_out'}."""
    r = rng_for(seed, name)
    shape = tuple(int(s) for s in shape)
    if kind == 'conv':
        fan_out = shape[0] * int(np.prod(shape[2:]))
        return (r.standard_normal(shape, dtype=np.float32) *
                np.float32(np.sqrt(2.0 / fan_out)))
    if kind == 'bn_gamma':
        return (r.uniform(0.5, 1.5, size=shape) * kw.get('scale', 1.0)).astype(np.float32)
    if kind in ('bn_beta', 'bn_mean'):
        return (r.standard_normal(shape) * 0.1).astype(np.float32)
    if kind == 'bn_var':
        return r.uniform(0.5, 1.5, size=shape).astype(np.float32)
    if kind == 'bias':
        return (r.standard_normal(shape) * 0.01).astype(np.float32)
    if kind == 'linear':
        bound = 1.0 / np.sqrt(kw['fan_in'])
        return r.uniform(-bound, bound, size=shape).astype(np.float32)
    if kind == 'linear_out':
        fan_out, fan_in = kw['fan_out'], kw['fan_in']
        bound = OUTPUT_GAIN * np.sqrt(6.0 / (fan_in + fan_out))
        return r.uniform(-bound, bound, size=shape).astype(np.float32)
    raise ValueError(kind)


def _is_residual_tail_bn(name):
    """Last BN of a BasicBlock (bn2) / Bottleneck (bn3), and the BN closing a fuse term."""
    parts = name.split('.')
    leaf = parts[-1]
    in_block = ('branches' in parts or 'layer1' in parts or 'conv_layers' in parts)
    if in_block and leaf in ('bn2',) and 'layer1' not in parts and 'conv_layers' not in parts:
        return True
    if in_block and leaf == 'bn3':
        return True
    if 'fuse_layers' in parts:
        # fuse_layers.i.j.1 (up) or fuse_layers.i.j.k.1 with k the last conv of the chain
        return True
    return False


def synthetic_state_dict(spec, seed=0):
    """spec: iterable of (name, shape) in reference ``state_dict`` naming.  Returns name->ndarray.

    The kind of every tensor is inferred from its name and rank alone:
      conv weight (4-D), BN weight/bias/running_* (1-D, siblings), conv bias, Linear.
    """
    spec = list(spec)
    names = {n for n, _ in spec}
    out = {}
    for name, shape in spec:
        prefix, leaf = name.rsplit('.', 1)
        is_bn = (prefix + '.running_mean') in names
        if leaf == 'num_batches_tracked':
            out[name] = np.zeros((), np.int64)
        elif is_bn:
            if leaf == 'weight':
                scale = RESIDUAL_GAMMA if _is_residual_tail_bn(prefix) else 1.0
                out[name] = synth_tensor(seed, name, shape, 'bn_gamma', scale=scale)
            elif leaf == 'bias':
                out[name] = synth_tensor(seed, name, shape, 'bn_beta')
            elif leaf == 'running_mean':
                out[name] = synth_tensor(seed, name, shape, 'bn_mean')
            elif leaf == 'running_var':
                out[name] = synth_tensor(seed, name, shape, 'bn_var')
            else:
                raise ValueError(name)
        elif len(shape) == 4:
            out[name] = synth_tensor(seed, name, shape, 'conv')
        elif len(shape) == 2:
            if 'output_layer' in name:
                out[name] = synth_tensor(seed, name, shape, 'linear_out',
                                         fan_in=shape[1], fan_out=shape[0])
            else:
                out[name] = synth_tensor(seed, name, shape, 'linear', fan_in=shape[1])
        elif len(shape) == 1 and leaf == 'bias':
            wname = prefix + '.weight'
            wshape = dict(spec).get(wname)
            if wshape is not None and len(wshape) == 2:
                out[name] = synth_tensor(seed, name, shape, 'linear', fan_in=wshape[1])
            else:
                out[name] = synth_tensor(seed, name, shape, 'bias')
        else:
            raise ValueError(f'cannot infer kind of {name} {shape}')
    return out


def fill_module_synthetic(module, seed=0, prefix='', only_prefixes=('backbone.', 'regressor.module.')):
    """Overwrites (in place) every conv/BN/linear tensor of ``module`` whose state_dict key
    starts with one of ``only_prefixes``.  Works for the reference module and for ours."""
    import torch
    sd = module.state_dict()
    spec = [(k, tuple(v.shape)) for k, v in sd.items()
            if any((prefix + k).startswith(p) for p in only_prefixes)]
    vals = synthetic_state_dict([(prefix + k, s) for k, s in spec], seed)
    with torch.no_grad():
        for k, _ in spec:
            sd[k].copy_(torch.from_numpy(np.asarray(vals[prefix + k])))
    return module


def synthetic_images(batch, size=224, seed=0):
    """Synthetic crops with post-normalisation statistics (zero mean, ~unit variance), NCHW
    f32: a per-image random low-frequency field (so that different images give visibly
    different features -- white noise alone is averaged away by the strided convs) plus
    0.3 * N(0,1) pixel noise."""
    r = rng_for(seed, f'images.{batch}.{size}')
    yy, xx = np.meshgrid(np.linspace(0, 1, size, dtype=np.float32),
                         np.linspace(0, 1, size, dtype=np.float32), indexing='ij')
    img = 0.3 * r.standard_normal((batch, 3, size, size), dtype=np.float32)
    nwaves = 6
    freq = r.uniform(-4, 4, size=(batch, 3, nwaves, 2)).astype(np.float32)
    phase = r.uniform(0, 2 * np.pi, size=(batch, 3, nwaves)).astype(np.float32)
    amp = (r.uniform(0.3, 1.0, size=(batch, 3, nwaves)) / np.sqrt(nwaves / 2)).astype(np.float32)
    for w in range(nwaves):
        arg = (2 * np.pi * (freq[:, :, w, 0, None, None] * yy + freq[:, :, w, 1, None, None] * xx)
               + phase[:, :, w, None, None])
        img += amp[:, :, w, None, None] * np.cos(arg).astype(np.float32)
    return np.ascontiguousarray(img, dtype=np.float32)
