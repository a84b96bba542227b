"""CPU-only tests of the host side: config system, C-ABI symbols, plan builder, DP helpers."""
import os
import os.path as osp
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = osp.dirname(osp.dirname(osp.abspath(__file__)))


# ---- config -------------------------------------------------------------------------------
def test_config_defaults_yaml_dotlist():
    from shapy_amd.config import merge_config, default_config, ConfigNode
    cfg = merge_config([osp.join(ROOT, 'configs/b2a_expose_hrnet_demo.yaml')],
                       ['network.smplx.num_stages=2', 'output_folder=/tmp/x',
                        'datasets.pose.transforms.crop_size=224'])
    assert cfg.network.type == 'SMPLXRegressor'
    assert cfg.network.smplx.num_stages == 2 and cfg.output_folder == '/tmp/x'
    assert cfg.network.smplx.get('feature_key') == 'concat'
    assert cfg.network.smplx.mlp.activation.type == 'none'            # yaml over default
    assert cfg.network.smplx.mlp.activation.inplace is True           # default survives merge
    assert cfg.body_model.smplx.use_face_contour is True
    assert cfg.network.smplx.backbone.hrnet.stage3.num_channels == [48, 96, 192]
    assert cfg.datasets.pose.transforms.crop_size == 224
    assert default_config().datasets.pose.transforms.crop_size == 256   # reference default
    d = dict(**cfg.network.smplx.camera)                                # ** splatting works
    assert d['pos_func'] == 'softplus'
    with pytest.raises(AttributeError):
        cfg.network.nonexistent
    c2 = cfg.copy()
    c2.network.smplx.num_stages = 5
    assert cfg.network.smplx.num_stages == 2
    assert isinstance(cfg.network.smplx, ConfigNode)


def test_cmd_parser():
    from shapy_amd.config import parse_args
    cfg = parse_args(['--exp-cfg', osp.join(ROOT, 'configs/b2a_expose_hrnet_eval_shape.yaml'),
                      '--exp-opts', 'use_cuda=False', '--num-gpus', '8', '--backend', 'gloo'])
    assert cfg.num_gpus == 8 and cfg.backend == 'gloo' and cfg.use_cuda is False


# ---- C-ABI --------------------------------------------------------------------------------
def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    from shapy_amd import _lib
    hdr = open(osp.join(ROOT, 'include', 'shapy_hip.h')).read()
    declared = set(re.findall(r'\b(shapy_[a-z0-9_]+)\s*\(', hdr))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.shapy_abi_version() == 8
    assert lib.shapy_build_arch() == b'gfx950'
    # struct layouts agree with the C header (sizeof through a tiny C program)
    src = '#include <stdio.h>\n#include "shapy_hip.h"\nint main(){printf("%zu %zu %zu", ' \
          'sizeof(ShapyConv), sizeof(ShapyOp), sizeof(ShapySmplxModel));return 0;}'
    exe = '/tmp/shapy_sizeof'
    subprocess.run(['gcc', '-x', 'c', '-', '-I', osp.join(ROOT, 'include'), '-o', exe],
                   input=src.encode(), check=True)
    sizes = [int(x) for x in subprocess.check_output([exe]).split()]
    import ctypes
    assert sizes == [ctypes.sizeof(_lib.ShapyConv), ctypes.sizeof(_lib.ShapyOp),
                     ctypes.sizeof(_lib.ShapySmplxModel)]


def test_product_path_fails_loudly_without_gpu():
    from shapy_amd._lib import ShapyHipError
    from shapy_amd.models.common.pose_utils import ContinuousRotReprDecoder
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    with pytest.raises(ShapyHipError):
        ContinuousRotReprDecoder(1)(torch.zeros(1, 6))


def test_product_never_imports_oracle():
    bad = []
    for dp, _, files in os.walk(osp.join(ROOT, 'shapy_amd')):
        for f in files:
            if f.endswith('.py'):
                txt = open(osp.join(dp, f)).read()
                if re.search(r'^\s*(from|import)\s+oracle\b', txt, re.M):
                    bad.append(osp.join(dp, f))
    assert not bad, bad


# ---- HRNet plan ---------------------------------------------------------------------------
@pytest.fixture(scope='module')
def hrnet():
    from shapy_amd.config import default_config
    from shapy_amd.models.backbone.hrnet import HighResolutionNet
    return HighResolutionNet(default_config().network.smplx.backbone.hrnet)


def test_plan_macs_match_survey(hrnet):
    P = hrnet._build_plan(224, 224)
    macs = sum(o['Ho'] * o['Wo'] * o['Cout'] * o['Cin'] * o['ksize'] ** 2
               for o in P.ops if o['type'] != 2)
    assert macs == 18_466_524_160           # SURVEY.md 8(d), hooked from the reference module
    assert sum(1 for o in P.ops if o['type'] == 0) == 330 and len(P.ops) == 332
    P256 = hrnet._build_plan(256, 256)
    macs256 = sum(o['Ho'] * o['Wo'] * o['Cout'] * o['Cin'] * o['ksize'] ** 2
                  for o in P256.ops if o['type'] != 2)
    assert abs(2 * macs256 / 1e9 - 48.239) < 0.01


def _executor_order(ops):
    """Reachability enforced by csrc/hrnet_ops.hip for a compiled op list, re-derived from what the
    executor itself sees (lane, barrier_before, group, sig, wait) -- NOT from the plan's own
    dependency sets: reach[v] = bit set of the ops that have finished when v starts."""
    n = len(ops)
    reach = [0] * n
    lane_tail, barrier_mask, ev = {}, 0, {}
    i = 0
    while i < n:
        o = ops[i]
        g = max(1, o['group'])
        members = range(i, i + g)
        if o['barrier_before']:
            barrier_mask, ev = (1 << i) - 1, dict(ev)
        base = barrier_mask | lane_tail.get(o['lane'], 0)
        for w in o['wait']:                      # waits are issued once, before the (group) launch
            if w >= 0:
                base |= ev[w]
        tail = 0
        for k in members:
            reach[k] = base
            tail |= base | (1 << k)
        for k in members:
            if ops[k]['sig'] >= 0:
                ev[ops[k]['sig']] = tail if g > 1 else reach[k] | (1 << k)
        lane_tail[o['lane']] = tail
        i += g
    return reach


@pytest.mark.parametrize('dag,group,ksplit', [(True, False, {}), (True, False, {(384, 4): 2}),
                                              (True, False, {(384, 4): 4, (192, 16): 2}),
                                              (True, False, {(384, 4): 4, (192, 16): 4, (96, 49): 2}),
                                              (False, False, {(384, 4): 2}), (False, True, {(384, 4): 2})])
def test_plan_orders_every_memory_hazard(hrnet, dag, group, ksplit):
    """The executor's order (lanes, barriers, dependency events, launch groups) covers every hazard
    of the PACKED workspace: whenever two ops touch overlapping memory and at least one of them
    writes, one of them has finished before the other starts.  Checked for the event-driven plan
    (dag), the round-2 barrier plan and the grouped single-stream plan, with and without split-K
    layers (whose slabs are buffers of the same workspace)."""
    from shapy_amd import _lib
    keep = hrnet._dag_eff, hrnet.group_branches, hrnet.conv_algo, hrnet.wino4_min_hw, hrnet.wino4_ksplit
    try:
        hrnet._dag_eff, hrnet.group_branches, hrnet.conv_algo, hrnet.wino4_min_hw = dag, group, 'winograd4', 7
        hrnet.wino4_ksplit = ksplit
        small = (96, 49) in ksplit               # the B <= 8 bucket: its implicit-GEMM split layers as well
        if small:
            from shapy_amd.models.backbone.hrnet import DEFAULT_DIRECT_KSPLIT_BY_BATCH
            hrnet.direct_ksplit = dict(DEFAULT_DIRECT_KSPLIT_BY_BATCH[0][1])
        P = hrnet._build_plan(224, 224)
        waits = P.sync_plan()
        total = P.allocate()
    finally:
        hrnet.direct_ksplit = None
        hrnet._dag_eff, hrnet.group_branches, hrnet.conv_algo, hrnet.wino4_min_hw, hrnet.wino4_ksplit = keep
    n_direct = sum(1 for o in P.ops if o.get('scrb') is not None and not o['tile'] & _lib.TILE_WINO4)
    assert (n_direct > 20) == small              # head 1x1 GEMMs + stride-2 fuse / transition / subsample convs
    n_split = sum(1 for o in P.ops if o.get('scrb') is not None) - n_direct
    # 384 -> 384 @7x7: 3 modules x 8 convs, 192 -> 192 @14x14: 7 modules x 8; none inside launch groups
    assert n_split == (0 if not ksplit else 0 if group else
                       24 + (56 if (192, 16) in ksplit else 0) + (64 if (96, 49) in ksplit else 0))
    cnts = sorted((o['cnt_off'], o['cnt_n']) for o in P.ops if o.get('scrb') is not None)
    assert all(c0 + n <= c1 for (c0, n), (c1, _) in zip(cnts, cnts[1:]))      # disjoint counter slices
    assert not cnts or cnts[-1][0] + cnts[-1][1] == P.cnt_ints
    ops = P.ops
    reach = _executor_order(ops)
    if dag:
        assert sum(1 for w in waits if w) > 30 and sum(o['barrier_before'] for o in ops) < 20
    # memory accesses: (op, write?, first float, end float, channel window inside a pixel row)
    acc = []
    for i, o in enumerate(ops):
        for key, is_w, c0, cn in (('inb', False, 0, None), ('scrb', True, 0, None),
                                  ('resb', False, o['res_coff'], o['Cout']),
                                  ('outb', True, o['out_coff'], o['Cout'])):
            b = o.get(key)
            if b is None:
                continue
            assert b.off % 4 == 0 and b.off + b.size <= total
            acc.append((i, is_w, b.off, b.off + b.size, id(b), c0, c0 + (b.C if cn is None else cn)))
    acc.sort(key=lambda a: a[2])
    checked = 0
    for x in range(len(acc)):
        i, wi, s0, e0, bi, ci0, ci1 = acc[x]
        for y in range(x + 1, len(acc)):
            j, wj, s1, e1, bj, cj0, cj1 = acc[y]
            if s1 >= e0:
                break
            if i == j or not (wi or wj):
                continue
            if bi == bj and not (ci0 < cj1 and cj0 < ci1):
                continue                                  # disjoint channel slices of one buffer
            u, v = min(i, j), max(i, j)
            assert (reach[v] >> u) & 1, (ops[u].get('name'), ops[v].get('name'), ops[u]['lane'], ops[v]['lane'])
            checked += 1
    assert checked > 1000
    assert total * 4 / 1e6 < 20.0            # MB per 224x224 image: packing works (no reuse: 112 MB)


@pytest.mark.parametrize('B', [1, 16, 64])
def test_prologue_cut_is_self_contained(hrnet, B):
    """forward(x, prefetch=): the ops in front of the plan's first barrier (stem, conv2, layer1) run as a serial
    prologue in a workspace of their own (backbone/prefetch.py).  That is only valid when they read nothing but the
    input and what they wrote themselves, hold no split-K scratch (its counters belong to the caller's stream) and
    no op of the rest waits for one of their events -- for every batch bucket's plan."""
    from shapy_amd.models.backbone.prefetch import ProloguePrefetch, cut_of
    keep = hrnet._dag_eff, hrnet.conv_algo, hrnet.wino4_min_hw
    try:
        hrnet._dag_eff, hrnet.conv_algo, hrnet.wino4_min_hw = True, 'winograd4', 7
        hrnet._ksplit_eff, hrnet._direct_ksplit_eff = hrnet.ksplit_policy(B), hrnet.direct_ksplit_policy(B, False)
        P = hrnet._build_plan(224, 224)
        P.sync_plan()
        P.allocate()
    finally:
        hrnet._ksplit_eff = hrnet._direct_ksplit_eff = None
        hrnet._dag_eff, hrnet.conv_algo, hrnet.wino4_min_hw = keep
    assert cut_of(P, 1) == 15 and cut_of(P, 2) == 35 and cut_of(P, 99) == 0
    for nth, cut, name in ((1, 15, 'transition1.0'), (2, 35, 'transition2.2.0')):
        assert P.ops[cut]['name'] == name and P.ops[cut]['barrier_before']
        written = set()
        for o in P.ops[:cut]:
            for key in ('inb', 'resb'):
                assert o.get(key) is None or id(o[key]) in written, o.get('name')
            written.add(id(o['outb']))
            if o.get('scrb') is not None:
                written.add(id(o['scrb']))
        # what the rest reads of the prologue: the outputs of its last module only (layer1's output; the two
        # branch outputs of stage 2's fuse layers)
        live = {id(o[key]) for o in P.ops[cut:] for key in ('inb', 'resb') if o.get(key) is not None} & written
        assert 1 <= len(live) <= 2 * nth
    # the B <= 8 bucket splits the 96-channel F(4x4) layers of stage 2: split-K layers inside the deeper prologue
    # (every workspace has arrival counters of its own)
    assert any(o.get('scrb') is not None for o in P.ops[:35]) == (B <= 8)
    cut = 15
    # a rest that waits for a prologue event has no cut
    P.ops[cut + 1]['wait'] = [P.ops[1]['sig'], -1, -1]
    assert P.ops[1]['sig'] >= 0 and cut_of(P, 1) == 0
    # the stash: same memory + same version + same plan + same workspace entry + same stream, else nothing
    pf, eng, ent = ProloguePrefetch(), {}, {}
    x = torch.zeros(2, 3, 32, 32)
    pf.pending = dict(key=pf.key(x, 7), eng=eng, ent=ent, sk=7, arena=1, done=None, x=x, cut=15)
    assert pf.take(x[:], eng, ent, 7)['arena'] == 1 and pf.pending is None and pf.used == 1
    for other in (dict(x=x.clone()), dict(eng={}), dict(ent={}), dict(sk=8), dict(edit=True)):
        pf.pending = dict(key=pf.key(x, 7), eng=eng, ent=ent, sk=-1, arena=1, done=None, x=x, cut=15)
        if other.get('edit'):
            x.add_(1)
        assert pf.take(other.get('x', x), other.get('eng', eng), other.get('ent', ent), other.get('sk', 7)) is None
        assert pf.pending is None
    assert pf.used == 1 and not pf.usable(x, x)            # (host tensors are never prefetched)


@pytest.mark.parametrize('size', [64, 256])
def test_event_driven_plan_at_other_input_sizes(hrnet, size):
    """The dependency events fit their 64 slots and at most three waits per op at the sizes the
    reference uses besides 224 (256: expose configs; 64: the smallest legal input), and the
    executor's order still covers every hazard of the packed workspace."""
    keep = hrnet._dag_eff, hrnet.conv_algo
    try:
        hrnet._dag_eff, hrnet.conv_algo = True, 'winograd4'
        P = hrnet._build_plan(size, size)
        waits = P.sync_plan()
        total = P.allocate()
    finally:
        hrnet._dag_eff, hrnet.conv_algo = keep
    ops = P.ops
    assert max(len(w) for w in waits) <= 3 and max(o['sig'] for o in ops) < 64
    reach = _executor_order(ops)
    spans = []
    for i, o in enumerate(ops):
        for key, is_w in (('inb', False), ('scrb', True), ('resb', False), ('outb', True)):
            b = o.get(key)
            if b is not None:
                spans.append((i, is_w, b.off, b.off + b.size, id(b)))
    n_checked = 0
    for x in range(len(spans)):
        i, wi, s0, e0, bi = spans[x]
        for y in range(x + 1, len(spans)):
            j, wj, s1, e1, bj = spans[y]
            if i == j or bi == bj or not (wi or wj) or s1 >= e0 or s0 >= e1:
                continue                      # (same-buffer channel slices: the 224 test)
            u, v = min(i, j), max(i, j)
            assert (reach[v] >> u) & 1, (ops[u].get('name'), ops[v].get('name'))
            n_checked += 1
    assert n_checked > 100 and total > 0


def test_bn_fold_matches_conv_bn_eval():
    from shapy_amd.models.backbone.hrnet import _fold
    torch.manual_seed(0)
    conv = torch.nn.Conv2d(8, 6, 3, 2, 1, bias=True)
    bn = torch.nn.BatchNorm2d(6).eval()
    bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 1.5)
    bn.weight.data.uniform_(0.5, 1.5); bn.bias.data.normal_()
    w, b = _fold(conv, bn)
    x = torch.randn(2, 8, 9, 9)
    ref = bn(conv(x))
    w_oihw = torch.from_numpy(w).permute(0, 3, 1, 2)
    out = torch.nn.functional.conv2d(x, w_oihw, torch.from_numpy(b), 2, 1)
    assert (ref - out).abs().max() < 1e-5


def test_regressor_collapse_is_the_same_affine_map():
    from shapy_amd.models.common.networks import MLP
    torch.manual_seed(0)
    mlp = MLP(2193, 145, layers=[1024, 1024], activation={'type': 'none'},
              normalization={'type': 'none'}, dropout=0.5, gain=0.3).eval()
    assert sorted(mlp.state_dict()) == ['layer_000.0.bias', 'layer_000.0.weight',
                                        'layer_001.0.bias', 'layer_001.0.weight',
                                        'output_layer.bias', 'output_layer.weight']
    W, b = mlp.collapse()
    x = torch.randn(3, 2193, dtype=torch.float64)
    y = x
    for lin in mlp.linears():
        y = y @ lin.weight.double().t() + lin.bias.double()
    assert (x @ W.t() + b - y).abs().max() < 1e-9
    with pytest.raises(NotImplementedError):
        MLP(8, 4, layers=[8], activation={'type': 'relu'}, normalization={'type': 'none'}).collapse()


def test_train_mode_is_refused(hrnet):
    hrnet.train()
    try:
        with pytest.raises(RuntimeError, match='eval mode'):
            hrnet(torch.zeros(1, 3, 64, 64))
    finally:
        hrnet.eval()


def test_weight_caches_are_keyed_on_parameter_versions(hrnet):
    """an in-place edit of a parameter or buffer must invalidate the folded blob /
    graphs (HighResolutionNet._compile compares _weights_version())."""
    v0 = hrnet._weights_version()
    assert hrnet._weights_version() == v0
    with torch.no_grad():
        hrnet.conv1.weight.mul_(1.0)                 # any in-place op bumps the version counter
    v1 = hrnet._weights_version()
    assert v1 != v0
    hrnet.bn1.running_mean.add_(0.0)
    assert hrnet._weights_version() != v1
    hrnet.invalidate()
    assert hrnet.__dict__['_ver_tensors'] is None and hrnet._engine == {}


def test_weight_cache_key_sees_replaced_parameters_but_not_dot_data_edits(hrnet):
    """replacing a Parameter object moves the key (registration epoch + tensor ids);
    an edit through ``p.data`` does NOT bump ``p._version`` -- documented: it needs invalidate()."""
    import torch.nn as nn
    v0 = hrnet._weights_version()
    old = hrnet.conv2.weight
    hrnet.conv2.weight = nn.Parameter(old.detach().clone())      # same values, new object
    v1 = hrnet._weights_version()
    assert v1 != v0
    assert any(t is hrnet.conv2.weight for t in hrnet.__dict__['_ver_tensors'])   # list re-walked
    hrnet.conv2.weight = old
    # the documented blind spot: .data has its own version counter
    v2 = hrnet._weights_version()
    hrnet.conv1.weight.data.mul_(1.0)
    assert hrnet._weights_version() == v2
    hrnet._engine['sentinel'] = 1
    hrnet.invalidate()                                           # what the docs prescribe after it
    assert hrnet._engine == {}


def test_winograd_plan_covers_the_3x3_stride1_layers(hrnet):
    from shapy_amd.utils import winograd
    hrnet.conv_algo = 'winograd'
    P = hrnet._build_plan(64, 64)
    n3 = [o for o in P.ops if o['type'] == 0 and o['ksize'] == 3 and o['stride'] == 1]
    nw = [o for o in P.ops if o.get('wino_off', -1) >= 0]
    assert len(nw) == len(n3) and len(nw) > 200          # every 3x3 / stride-1 conv of HRNet-W48
    hrnet.conv_algo = 'direct'
    assert all(o.get('wino_off', -1) < 0 for o in hrnet._build_plan(64, 64).ops)
    hrnet.conv_algo = 'winograd'
    # filter transform: U[4i+j] = (G g G^T)[i][j], K-chunked layout
    w = np.random.default_rng(0).standard_normal((48, 3, 3, 32)).astype(np.float32)
    u = winograd.transform_filters(w)
    assert u.shape == (16, 2, 48, 16)
    g = w[5, :, :, 19].astype(np.float64)
    np.testing.assert_allclose(u[:, 1, 5, 3].reshape(4, 4), winograd.G @ g @ winograd.G.T, rtol=1e-6)


def test_winograd4_plan_filters_and_kernel_indexing(hrnet):
    """conv_algo = 'winograd4': F(4x4,3x3) on the large maps only, flagged by SHAPY_TILE_WINO4;
    the filter layout; the NumPy restatement against a float64 direct convolution; and the
    thread-by-thread emulation of csrc/conv_wino4.hip's indexing (tools/wino4_emulate.py)."""
    import importlib.util
    from shapy_amd import _lib
    from shapy_amd.utils import winograd
    keep = hrnet.conv_algo, hrnet.wino4_min_hw
    hrnet.conv_algo = 'winograd4'
    try:
        hrnet.wino4_min_hw = 28
        P28 = hrnet._build_plan(224, 224)
        hrnet.wino4_min_hw = 14
        P = hrnet._build_plan(224, 224)
    finally:
        hrnet.conv_algo = 'winograd'
    count = lambda plan: (
        sum(1 for o in plan.ops if o['type'] == 0 and o['tile'] & _lib.TILE_WINO4),
        sum(1 for o in plan.ops if o.get('wino_off', -1) >= 0 and not o['tile'] & _lib.TILE_WINO4))
    assert count(P28) == (129, 89)                      # 48 @56x56, 96 @28x28, transition1
    assert count(P) == (185, 33)                        # + 192 @14x14; the 7x7 maps keep F(2x2)
    w4 = [o for o in P.ops if o['type'] == 0 and o['tile'] & _lib.TILE_WINO4]
    assert all(min(o['Hi'], o['Wi']) >= hrnet.wino4_min_hw and o['Cout'] % 48 == 0 and
               o['wino_off'] >= 0 for o in w4)
    assert not any(o['tile'] & _lib.TILE_WINO4 for o in hrnet._build_plan(224, 224).ops)
    hrnet.conv_algo, hrnet.wino4_min_hw = keep
    rng = np.random.default_rng(0)
    w = (rng.standard_normal((48, 3, 3, 32)) / 17).astype(np.float32)
    u = winograd.transform_filters4(w)
    assert u.shape == (36, 2, 48, 16)
    g = w[5, :, :, 19].astype(np.float64)
    np.testing.assert_allclose(u[:, 1, 5, 3].reshape(6, 6), winograd.G4 @ g @ winograd.G4.T,
                               rtol=1e-6, atol=1e-7)
    spec = importlib.util.spec_from_file_location('wino4_emulate',
                                                  osp.join(ROOT, 'tools', 'wino4_emulate.py'))
    emu = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(emu)
    x = rng.standard_normal((2, 9, 7, 32)).astype(np.float32)
    b = rng.standard_normal(48).astype(np.float32)
    y = winograd.conv_reference4(x, u, b)
    assert np.abs(y - emu.direct_conv(x, w, b)).max() < 2e-5
    emu.check(2, 12, 20, 48, 48, True, True)             # two workgroups, 3 chunks, residual
    emu.check(1, 7, 9, 32, 96, True, False)              # partial edge tiles, two N tiles
    emu.check(1, 14, 14, 16, 48, False, False, coff=16)  # channel-offset epilogue
    # the per-layer kernel's four-multiplying-wave workgroup (round 6): staging by all waves, 27-item split, exchange
    emu.check(2, 12, 20, 48, 48, True, True, four_waves=True)
    emu.check(1, 7, 9, 32, 96, True, False, four_waves=True)


def test_split_k_policy_marks_the_7x7_branch_and_nothing_else(hrnet):
    """Default plan: the 24 convs of the 384-channel branch on the 7x7 maps carry SHAPY_TILE_W4_KSPLIT(2),
    a slab of 2 x 4 tiles x 16 pixels x 384 channels per image and 48 counters per image each; the choice
    does not depend on the batch (the plan has no batch) and is the same at 256 x 256 (8x8 maps: 4 tiles)."""
    from shapy_amd import _lib
    from shapy_amd.models.backbone.hrnet import DEFAULT_WINO4_KSPLIT
    keep = hrnet.conv_algo, hrnet.wino4_ksplit
    try:
        hrnet.conv_algo, hrnet.wino4_ksplit = 'winograd4', dict(DEFAULT_WINO4_KSPLIT)
        plans = {size: hrnet._build_plan(size, size) for size in (224, 256)}
    finally:
        hrnet.conv_algo, hrnet.wino4_ksplit = keep
    for size, P in plans.items():
        split = [o for o in P.ops if o['type'] == 0 and (o['tile'] >> 21) & 3]
        assert len(split) == 24 and {(o['Cin'], o['Cout'], o['Hi']) for o in split} == {(384, 384, size // 32)}
        assert all((o['tile'] >> 21) & 3 == 1 and o['tile'] & _lib.TILE_WINO4 for o in split)
        assert all(o['scrb'].size == 2 * 4 * 16 * 384 for o in split)
        assert P.cnt_ints == 24 * 2 * 24
    # by batch bucket when no override is set: more slices (and the 14x14 / 28x28 branches) for small batches
    assert keep[1] is None or isinstance(keep[1], dict)
    hrnet.wino4_ksplit = None
    try:
        assert hrnet.ksplit_policy(64) == hrnet.ksplit_policy(None) == hrnet.ksplit_policy(33) == {(384, 4): 2}
        assert hrnet.ksplit_policy(32) == hrnet.ksplit_policy(9) == {(384, 4): 4, (192, 16): 2}
        assert hrnet.ksplit_policy(8) == hrnet.ksplit_policy(1) == {(384, 4): 4, (192, 16): 4, (96, 49): 2}
        hrnet.conv_algo = 'winograd4'
        hrnet._ksplit_eff = hrnet.ksplit_policy(1)
        Ps = hrnet._build_plan(224, 224)
    finally:
        hrnet._ksplit_eff = None
        hrnet.conv_algo, hrnet.wino4_ksplit = keep
    by_s = {}
    for o in Ps.ops:
        sl = ((o['tile'] >> 21) & 3) + 1 if o['type'] == 0 else 1
        if sl > 1:
            by_s[(o['Cin'], o['Hi'], sl)] = by_s.get((o['Cin'], o['Hi'], sl), 0) + 1
    # 24 convs @7x7 in four slices, 56 @14x14 in four, 64 @28x28 in two (4 tiles x 49 = 49 tiles per image)
    assert by_s == {(384, 7, 4): 24, (192, 14, 4): 56, (96, 28, 2): 64}
    assert _lib.w4_split_sizes(7, 7, 384, 2) == (2 * 4 * 16 * 384, 48)
    assert _lib.w4_split_sizes(14, 14, 192, 2) == (2 * 16 * 16 * 192, 24)
    with pytest.raises(ValueError):
        _lib.tile_w4_ksplit(5)


def test_head_gemms_can_take_the_bf16x6_arithmetic(hrnet):
    """Opt-in (x6_gemm_min_batch > 0; off by default): compute_dtype 'f32', B >= x6_gemm_min_batch: the head's
    fifteen wide 1x1 GEMMs (Cin, Cout >= 512) carry
    SHAPY_TILE_X6 and their weights are the three bf16 planes [Cout][3][Kp]; no other layer does, no smaller batch
    does; bench.py counts those layers with the matrix-pipe time they need."""
    import types
    from shapy_amd import _lib
    sys.path.insert(0, ROOT)
    import bench
    keep = hrnet.conv_algo, hrnet.wino4_min_hw
    try:
        hrnet.conv_algo, hrnet.wino4_min_hw = 'winograd4', 7
        P0 = hrnet._build_plan(224, 224)
        hrnet._x6_eff = True
        P1 = hrnet._build_plan(224, 224)
    finally:
        hrnet._x6_eff = False
        hrnet.conv_algo, hrnet.wino4_min_hw = keep
    assert not any(o['tile'] & _lib.TILE_X6 for o in P0.ops if o['type'] == _lib.OP_CONV)
    x6 = [o for o in P1.ops if o['type'] == _lib.OP_CONV and o['tile'] & _lib.TILE_X6]
    assert len(x6) == 15 and all(o['name'].startswith('conv_layers.') and o['ksize'] == 1 for o in x6)
    assert sorted({(o['Cin'], o['Cout']) for o in x6}) == [(512, 2048), (1536, 512), (1536, 2048), (2048, 512),
                                                           (2048, 2048)]
    assert all(o.get('wino_off', -1) < 0 and o.get('scrb') is None for o in x6)
    # planes: 6 bytes per weight instead of 4
    grow = sum(o['Cin'] * o['Cout'] * 2 for o in x6)
    assert 0 <= P1.wbytes - P0.wbytes - grow < 16 * len(x6)
    net = types.SimpleNamespace(backbone=hrnet)
    f0 = bench.executed_mfma_flop_per_image(net, 224, plan=P0)
    f32_part, bf16_part = bench.executed_mfma_flop_per_image(net, 224, plan=P1, parts=True)
    macs_x6 = sum(49 * o['Cin'] * o['Cout'] for o in x6)
    assert bf16_part == 12 * macs_x6 and f32_part == f0 - 2 * macs_x6
    f1 = bench.executed_mfma_flop_per_image(net, 224, plan=P1)
    assert abs(f1 - (f32_part + 2 * macs_x6 * 6 * 157.3 / 2500.0)) < 1.0 and f1 < f0
    assert hrnet.x6_gemm_min_batch == int(os.environ.get('SHAPY_X6_GEMM_MIN_BATCH', '0'))      # off by default


def test_bench_flop_accounting_algorithmic_vs_executed(hrnet):
    """bench.py: `roofline.achieved` counts the direct-convolution FLOPs of SURVEY.md 8(d) whatever
    the algorithm; `executed_mfma` what the matrix cores run (F(2x2): 16 products per 2x2 tile,
    F(4x4): 36 per 4x4 tile)."""
    import types
    sys.path.insert(0, ROOT)
    import bench
    net = types.SimpleNamespace(backbone=hrnet)
    keep = hrnet.conv_algo, hrnet.wino4_min_hw
    try:
        got = {}
        for algo in ('direct', 'winograd', 'winograd4'):
            hrnet.conv_algo, hrnet.wino4_min_hw = algo, 14
            got[algo] = (bench.conv_flop_per_image(net, 224), bench.executed_mfma_flop_per_image(net, 224))
    finally:
        hrnet.conv_algo, hrnet.wino4_min_hw = keep
    assert {a for a, _ in got.values()} == {bench.CONV_FLOP_PER_IMAGE_224}
    assert got['direct'][1] == bench.CONV_FLOP_PER_IMAGE_224
    assert 1.7 < got['winograd'][0] / got['winograd'][1] < 1.8          # 89 % of the MACs / 2.25
    assert 2.15 < got['winograd4'][0] / got['winograd4'][1] < 2.25      # F(4x4) from 14 px


def test_regressor_stage_collapse_matches_the_iteration():
    """The fully collapsed regressor (W_all, b_all of shapy_regressor_collapsed_f32) against the
    layer-by-layer float32 iteration of the oracle (networks.py:536-592)."""
    from oracle import body_np
    from shapy_amd.models.common.networks import MLP, IterativeRegression
    torch.manual_seed(1)
    F, P = 64, 23
    mlp = MLP(F + P, P, layers=[48, 48], activation={'type': 'none'},
              normalization={'type': 'none'}, dropout=0.5, gain=1.0).eval()
    with torch.no_grad():
        mlp.output_layer.bias.normal_()
    mean = torch.randn(1, P)
    it = IterativeRegression(mlp, mean, num_stages=3).eval()
    pk = it._pack(torch.device('cpu'), F)
    feat = torch.randn(5, F)
    got = (feat.double() @ pk['W_all'].double().t() + pk['b_all'].double()).view(5, 4, P)
    layers = [(l.weight.detach().numpy(), l.bias.detach().numpy()) for l in mlp.linears()]
    want = body_np.iterative_regression(feat.numpy(), mean.numpy().reshape(-1), layers, 3)
    for s_ in range(3):
        assert np.abs(got[:, s_].numpy() - want[s_]).max() < 5e-5 * np.abs(want[s_]).max()
    d0 = want[0] - mean.numpy()                      # block 3: the stage-0 delta
    assert np.abs(got[:, 3].numpy() - d0).max() < 5e-5 * np.abs(d0).max()
    # an in-place edit of a weight invalidates the packed copy (staleness key)
    with torch.no_grad():
        mlp.output_layer.bias.add_(1.0)
    pk2 = it._pack(torch.device('cpu'), F)
    assert pk2 is not pk and (pk2['b_all'] - pk['b_all']).abs().max() > 0.5


def test_full_module_state_dict_layout(golden_dir):
    """Same keys and shapes as the reference SMPLXRegressor (checkpoint compatibility)."""
    import __graft_entry__ as ge
    net, _ = ge.make_network(device='cpu')
    ref = {}
    with open(osp.join(golden_dir, 'state_dict_keys.txt')) as f:
        for line in f:
            k, s = line.strip().split(' ', 1)
            ref[k] = tuple(eval(s))
    mine = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    assert mine == ref
    assert net.param_mean.shape == (1, 145)
    assert net.betas_idxs.tolist() == list(range(132, 142))
    import copy
    net2 = copy.deepcopy(net)          # the reference's Evaluator deep-copies the model
    assert sorted(net2.state_dict()) == sorted(net.state_dict())


# ---- data parallel helpers (gloo, world_size 2) -----------------------------------------------
def _dp_worker(rank, world, port, q):
    import torch.distributed as dist
    from shapy_amd import parallel
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world))
    r, w = parallel.init_distributed('gloo')
    n = 7
    a, b = parallel.shard_range(n, r, w)
    full = torch.arange(n * 10, dtype=torch.float32).view(n, 10)
    g = parallel.gather_variable(full[a:b])
    ok1 = torch.equal(g, full)
    gat = parallel.BetasGatherer(w)
    outs = [gat(full[r * 3:(r + 1) * 3] + k) for k in range(3)]      # three "steps"
    last = gat.wait()
    ok2 = (all(torch.equal(o, full[:6] + k) for k, o in enumerate(outs)) and last is outs[-1]
           and gat.wait() is None and gat.issued == 3 and gat.deferred_waits == 2
           and torch.equal(gat.gather(full[r * 3:(r + 1) * 3]), full[:6]))
    q.put((rank, ok1, ok2, (a, b)))
    dist.destroy_process_group()


class _FakeRccl:
    """librccl stand-in for the CPU: records what the construction of a communicator asks of it."""

    def __init__(self, rank):
        self.rank, self.calls = rank, []

    def ncclGetUniqueId(self, uid_ref):
        import ctypes
        ctypes.memmove(uid_ref, bytes(range(100, 228)), 128)
        self.calls.append('id')
        return 0

    def ncclCommInitRank(self, comm_ref, world, uid, rank):
        self.calls.append(('init', world, rank, bytes(uid.internal)))
        return 0

    def ncclCommDestroy(self, comm):
        return 0

    def ncclGetErrorString(self, rc):
        return b'fake'


def _rccl_init_worker(rank, world, port, q, force):
    """BetasGatherer._init_rccl on gloo ranks with librccl stubbed: (a) every rank healthy -> the id
    drawn on rank 0 reaches every rank's ncclCommInitRank; (b) ONE rank fails before the collective
    phase -> nobody enters it (no id broadcast, no ncclCommInitRank) and all ranks agree on the fallback."""
    import torch.distributed as dist
    from shapy_amd import parallel, rccl
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    if force:
        os.environ['SHAPY_RCCL_FORCE_FAIL'] = force
    dist.init_process_group('gloo', init_method='env://')
    fake = _FakeRccl(rank)
    rccl._load = lambda: fake
    g = parallel.BetasGatherer(world)
    g._fallback_group = lambda: 'c10d-fallback-group'          # (no NCCL backend without GPUs)
    # a sub-group whose rank 0 is NOT the world's rank 0 exercises the broadcast source as well
    ok = g._init_rccl()
    q.put((rank, ok, g.mode, g.group, list(fake.calls)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('force', ['', 'rank1', 'rank0'])
def test_rccl_two_phase_init_on_gloo_with_stubbed_librccl(force):
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000 + {'': 0, 'rank1': 1, 'rank0': 2}[force]
    procs = [ctx.Process(target=_rccl_init_worker, args=(r, 2, port, q, force)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)       # a hang here = mismatched collectives
    for p in procs:
        p.join(60)
    uid = bytes(range(100, 228))
    if not force:
        assert [r[1:4] for r in res] == [(True, 'lane', None)] * 2
        assert res[0][4] == ['id', ('init', 2, 0, uid)] and res[1][4] == [('init', 2, 1, uid)]
    else:
        assert [r[1:4] for r in res] == [(False, 'work', 'c10d-fallback-group')] * 2
        assert not any(c[0] == 'init' for r in res for c in r[4] if isinstance(c, tuple))


def test_rccl_broadcast_source_is_the_groups_first_rank():
    """RcclComm.connect broadcasts the id from the GROUP's rank 0 as a global rank."""
    import inspect
    from shapy_amd import rccl
    src = inspect.getsource(rccl.RcclComm.connect)
    assert 'get_global_rank(self.group, 0)' in src and 'src=src' in src


def test_dp_sharding_and_allgather_gloo():
    import torch.multiprocessing as mp
    from shapy_amd.parallel import shard_range
    assert [shard_range(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert [shard_range(256, r, 8) for r in range(8)][-1] == (224, 256)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(r[1] and r[2] for r in res), res
    assert sorted(r[3] for r in res) == [(0, 4), (4, 7)]


def test_bf16x3_split_is_exact():
    """utils/split.py: h + m + l == a bit for bit, planes zero-padded to a multiple of 32."""
    from shapy_amd.utils.split import join_bf16x3, split_bf16x3
    r = np.random.default_rng(0)
    a = (r.standard_normal((5, 3, 48)) * np.exp(r.uniform(-20, 20, (5, 3, 48)))).astype(np.float32)
    a[0, 0, :6] = [0.0, -0.0, 3.0e38, -1.5e-30, 1.0 + 2.0 ** -23, 65504.0]
    planes = split_bf16x3(a)
    assert planes.shape == (5, 3, 3, 64) and planes.dtype == np.uint16
    assert not planes[..., 48:].any()
    back = join_bf16x3(planes, 48)
    assert np.array_equal(back, a)


def test_f32x6_plan_packs_the_same_weights_as_planes(hrnet):
    """The f32x6 plan stores every conv weight as [Cout, 3, Kp] bf16 planes whose sum is,
    bit for bit, the float32 weight of the f32 plan (same ops, same offsets elsewhere)."""
    from shapy_amd.utils.split import join_bf16x3
    Pf = hrnet._build_plan(64, 64)
    Px = hrnet._build_plan(64, 64, x6=True)
    assert len(Pf.ops) == len(Px.ops)
    blob_f = np.frombuffer(b''.join(Pf.wchunks), np.uint8)
    blob_x = np.frombuffer(b''.join(Px.wchunks), np.uint8)
    checked = 0
    for of, ox in zip(Pf.ops, Px.ops):
        for k in ('type', 'Cin', 'Cout', 'ksize', 'stride', 'Hi', 'Wi', 'relu', 'ups'):
            assert of[k] == ox[k]
        if of['type'] != 0 or checked >= 12:
            continue
        K = of['ksize'] ** 2 * of['Cin']
        Kp = (K + 31) // 32 * 32
        w = blob_f[of['wgt_off'] * 4:][:of['Cout'] * K * 4].view(np.float32).reshape(of['Cout'], K)
        planes = blob_x[ox['wgt_off'] * 4:][:of['Cout'] * 3 * Kp * 2].view(np.uint16)
        planes = planes.reshape(of['Cout'], 3, Kp)
        assert np.array_equal(join_bf16x3(planes, K), w)
        assert not planes[:, :, K:].any()
        # the biases stay float32 in both plans
        bf = blob_f[of['bias_off'] * 4:][:of['Cout'] * 4].view(np.float32)
        bx = blob_x[ox['bias_off'] * 4:][:of['Cout'] * 4].view(np.float32)
        assert np.array_equal(bf, bx)
        checked += 1
    assert checked == 12


def test_bf16_plan_keeps_the_48_channel_branch_unpadded():
    """bf16 rows are addressed in 8-channel (16-byte) slots; the 48-channel branch is NOT padded to
    64 channels any more (csrc/conv_igemm.hip: flat-K kernel for Cin % 32 != 0)."""
    import __graft_entry__ as ge
    net, _ = ge.make_network(model_folder='/tmp/shapy_synth_models', device='cpu')
    plan = net.backbone._build_plan(64, 64, bf16=True)
    convs = [o for o in plan.ops if o['type'] == 0]
    assert any(o['Cin'] == 48 and o['Cout'] == 48 and o['ksize'] == 3 for o in convs)
    for o in convs:
        assert o['Cin'] % 8 == 0 and o['Cout'] % 8 == 0 and o['in_ld'] % 8 == 0
        if o['Cin'] % 32:                      # flat-K kernel: no upsample epilogue, Cin >= 32
            assert o['ups'] == 1 and o['Cin'] >= 32, o
    f32 = net.backbone._build_plan(64, 64)
    # (same layers; the float32 plan lists the branch convs level by level: grouped launches)
    assert sorted((o['Cin'], o['Cout']) for o in convs) == \
        sorted((o['Cin'], o['Cout']) for o in f32.ops if o['type'] == 0)


# ---- bench.py's N-rank control flow without GPUs ---------------------------
@pytest.mark.parametrize('launcher', ['self_spawn', 'torchrun'])
def test_bench_two_ranks_end_to_end_on_gloo_with_stub_forward(launcher):
    """`bench.py --gpus 2` end to end on the gloo backend with a stub CPU forward: self-spawn
    command line + env:// rendezvous on 127.0.0.1 (and the driver's own torch.distributed.run
    form), barriers, the deferred all-gather of the betas checked shard by shard, max-over-ranks
    timing, teardown BEFORE rank 0's host-side work, exactly ONE JSON line from rank 0."""
    import json
    import socket
    import subprocess
    import sys
    root = osp.dirname(osp.dirname(osp.abspath(__file__)))
    tail = ['--gpus', '2', '--cpu-stub', '--steps', '3', '--warmup', '2', '--batch', '4', '--size', '32']
    if launcher == 'self_spawn':
        cmd = [sys.executable, osp.join(root, 'bench.py')] + tail
    else:
        with socket.socket() as s:
            s.bind(('127.0.0.1', 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
               '--master-addr', '127.0.0.1', '--master-port', str(port),
               osp.join(root, 'bench.py')] + tail
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d['stub'] is True and d['n_gpus'] == 2 and d['rccl_ranks'] == 2 and d['steps'] == 3
    assert d['config']['global_batch'] == 8 and d['scaling'] == 'weak'
    assert len(d['per_rank']['images_per_sec']) == 2
    # 2 warm-up + 3 timed gathers; all but the one joined by wait() after the warm-up and the last
    # one were joined by the NEXT step's call (the overlap bench.py relies on)
    assert d['per_rank']['allgather'] == {'issued': 5, 'joined_by_next_step': 3}
    assert abs(d['value'] - 8 * 1e3 / d['ms_per_step']) < 1e-6 * d['value']


def test_bench_refuses_a_world_size_mismatch():
    import subprocess
    import sys
    root = osp.dirname(osp.dirname(osp.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE='1', RANK='0')
    r = subprocess.run([sys.executable, osp.join(root, 'bench.py'), '--gpus', '2', '--cpu-stub'],
                       capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode != 0 and 'WORLD_SIZE=1' in (r.stderr + r.stdout)


def test_winograd_guard_keeps_the_callers_layer_overrides(hrnet):
    """The guard's demotions are dropped when the weights change; entries the CALLER put into
    layer_algo (or changed afterwards) survive; the fixed calibration probe is the same tensor on
    every call (every rank compiles the same plan)."""
    keep = dict(hrnet.layer_algo), dict(hrnet._guard_demotions)
    try:
        hrnet.layer_algo = {'stage2.0.branches.0.0.conv1': 'direct',              # the caller's
                            'stage4.0.branches.3.0.conv1': 'winograd'}
        hrnet._guard_demotions = {}
        for name, to in (('stage3.0.branches.1.0.conv2', 'winograd'), ('stage3.1.branches.2.1.conv1', 'direct'),
                         ('stage4.0.branches.3.0.conv1', 'direct')):             # ... over a caller's entry
            hrnet._guard_demote(name, to)                                          # what calibrate() does
        hrnet._guard_demote('stage4.0.branches.3.0.conv1', 'direct')              # (a second pass)
        assert hrnet.layer_algo['stage4.0.branches.3.0.conv1'] == 'direct'
        hrnet.layer_algo['stage3.0.branches.1.0.conv2'] = 'direct'                # caller tightens one
        hrnet._drop_guard_demotions()
        assert hrnet.layer_algo == {'stage2.0.branches.0.0.conv1': 'direct',
                                    'stage3.0.branches.1.0.conv2': 'direct',
                                    'stage4.0.branches.3.0.conv1': 'winograd'}    # the caller's value is back
        assert hrnet._guard_demotions == {}
        x = torch.zeros(3, 3, 64, 96)
        p1, p2 = hrnet._guard_probe(x), hrnet._guard_probe(x)
        assert p1.shape == (4, 3, 64, 96) and torch.equal(p1, p2) and float(p1.std()) > 0.5
        hrnet.wino_guard_probe = 'batch'
        assert hrnet._guard_probe(x).shape == (3, 3, 64, 96)
    finally:
        hrnet.wino_guard_probe = 'fixed'
        hrnet.layer_algo, hrnet._guard_demotions = keep


def test_rccl_binding_loads_and_bench_counts_launch_groups(hrnet):
    """shapy_amd/rccl.py binds the librccl.so that ships with torch (symbols only: no communicator
    without a GPU); bench.launches_per_forward counts a launch group as ONE launch."""
    import bench
    from shapy_amd import parallel, rccl
    lib = rccl._load()
    for sym in ('ncclGetUniqueId', 'ncclCommInitRank', 'ncclAllGather', 'ncclCommDestroy'):
        assert hasattr(lib, sym)
    assert ctypes_sizeof_unique_id() == 128
    g = parallel.BetasGatherer(1)
    assert g.mode == 'lane' and g(torch.ones(2, 10)).shape == (2, 10)      # one rank, not forced: identity
    with pytest.raises(ValueError):
        parallel.BetasGatherer(2, mode='nonsense')
    keep = hrnet.conv_algo, hrnet.wino4_min_hw, hrnet.group_branches
    try:
        hrnet.conv_algo, hrnet.wino4_min_hw = 'winograd4', 7
        hrnet.group_branches = False
        n_single = bench.launches_per_forward(hrnet._build_plan(224, 224))
        hrnet.group_branches = True
        P = hrnet._build_plan(224, 224)
        n_group = bench.launches_per_forward(P)
    finally:
        hrnet.conv_algo, hrnet.wino4_min_hw, hrnet.group_branches = keep
    assert n_single == len(P.ops) == 332
    assert n_group == n_single - sum(o['group'] - 1 for o in P.ops if o['group'] > 1) == 188


def ctypes_sizeof_unique_id():
    import ctypes
    from shapy_amd import rccl
    return ctypes.sizeof(rccl._UniqueId)


def test_grouped_branch_levels_plan(hrnet):
    """conv_algo='winograd4' with group_branches: the branch convs of every HighResolutionModule are
    listed level by level on lane 0, the first op of a level carries the group size, the ops of a
    group are independent (nobody reads or writes what another one writes) and never share memory."""
    keep = hrnet.conv_algo, hrnet.wino4_min_hw, hrnet.group_branches
    try:
        hrnet.conv_algo, hrnet.wino4_min_hw, hrnet.group_branches = 'winograd4', 7, True
        P = hrnet._build_plan(224, 224)
        P.allocate()
        ops = P.ops
        groups = [(i, o['group']) for i, o in enumerate(ops) if o['group'] > 1]
        assert sorted(collections_counter(g for _, g in groups).items()) == [(2, 8), (3, 32), (4, 24)]
        for i, n in groups:
            members = ops[i:i + n]
            assert all(m['type'] == 0 and m['lane'] == 0 for m in members)
            assert all(m['group'] == 0 and not m['barrier_before'] for m in members[1:])
            assert all(m.get('wino_off', -1) >= 0 and m['ksize'] == 3 and m['stride'] == 1 for m in members)
            for a in range(n):
                for b in range(n):
                    if a == b:
                        continue
                    wa = members[a]['outb']
                    for key in ('inb', 'resb', 'outb'):
                        other = members[b][key]
                        if other is None:
                            continue
                        assert other is not wa
                        assert wa.off + wa.size <= other.off or other.off + other.size <= wa.off
        # the same layers as the ungrouped plan, each exactly once
        hrnet.group_branches = False
        Q = hrnet._build_plan(224, 224)
        assert sorted(o['name'] for o in ops if o['type'] == 0) == sorted(o['name'] for o in Q.ops if o['type'] == 0)
        assert all(o['group'] == 0 for o in Q.ops)
    finally:
        hrnet.conv_algo, hrnet.wino4_min_hw, hrnet.group_branches = keep


def collections_counter(it):
    import collections
    return collections.Counter(it)


def test_w4g_schedule_native(tmp_path):
    """The task scheduling of the persistent grouped F(4x4) kernel (host LPT schedule + the per-slot
    task decode the kernel runs) lives in a HIP-free header: compile THAT header with g++ and replay
    every (XCD, slot) of HRNet's level groups at B = 1 / 3 / 64 / 334 and of odd shapes -- every task
    exactly once, on its own XCD, slots balanced (tests/native/w4g_sched_test.cpp)."""
    import shutil
    import subprocess
    root = osp.dirname(osp.dirname(osp.abspath(__file__)))
    cxx = shutil.which('g++') or shutil.which('c++')
    if cxx is None:
        pytest.skip('no host C++ compiler')
    exe = str(tmp_path / 'w4g_sched_test')
    subprocess.check_call([cxx, '-O2', '-std=c++17', '-I', osp.join(root, 'shapy_amd', 'csrc'),
                           osp.join(root, 'tests', 'native', 'w4g_sched_test.cpp'), '-o', exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and 'W4G SCHEDULE OK' in r.stdout, r.stdout[-2000:]
