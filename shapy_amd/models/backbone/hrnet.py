"""HRNet-W48 (SHAPY variant) on hand-written gfx950 kernels.

Drop-in for ``regressor/human_shape/models/backbone/hrnet.py``: same constructor config
(``cfg.stage1..4``, ``pretrained_layers``, ``use_old_impl``), same ``state_dict`` key layout
(``conv1.weight``, ``layer1.0.conv1.weight``, ``stage3.2.fuse_layers.1.0.0.0.weight``, ...) so
the reference's checkpoints load with ``load_state_dict``, same ``forward`` contract
(``[B,3,H,W]`` NCHW f32 -> ``{'concat': [B,2048]}``), same ``get_output_dim`` /
``load_weights``.

The ``torch.nn`` modules below are *parameter containers only* -- none of their ``forward``
methods is ever called.  ``forward`` compiles the tree once into

  * one weight blob: BatchNorm folded into conv weight/bias in float64, OIHW -> OHWI
    (K-contiguous rows for the MFMA B operand), and
  * a flat op list (``ShapyOp[]``) over a liveness-packed NHWC activation workspace,

and hands both to ``shapy_hrnet_run`` (csrc/hrnet_ops.hip), which launches the
implicit-GEMM MFMA kernel (csrc/conv_igemm.hip) per conv with the residual add, ReLU, the
nearest-upsample + add of the fuse layers and the channel concat fused into its epilogue.
The independent branches of a HighResolutionModule are issued on separate HIP streams.
"""
import ctypes
import os
import os.path as osp

import numpy as np
import torch
import torch.nn as nn

from ... import _lib
from ...utils.versioning import VersionedWeights
from ...utils import winograd

BN_MOMENTUM = 0.1
#: product defaults of HighResolutionNet.conv_algo / .wino4_min_hw (see there)
DEFAULT_CONV_ALGO = 'winograd4'
DEFAULT_WINO4_MIN_HW = 7
#: F(4x4) split-K (csrc/conv_wino4.hip) by batch bucket: (largest batch of the bucket | None, {(Cin, most 4x4
#: tiles per image): K slices}).  Measured on MI355X, backbone ms at B = 1 / 8 / 32 / 64
#: (profiles/r05h_ksplit_latency_sweep.txt): no split 5.22 / 5.87 / 7.90 / 12.42; the three policies below
#: 3.61 / 4.09 / 7.14 / 12.39 -- small batches are chains of latency-bound launches whose length is the K
#: depth of a layer, large ones are throughput-bound and lose with every extra prologue / epilogue.
DEFAULT_WINO4_KSPLIT_BY_BATCH = (
    (8, {(384, 4): 4, (192, 16): 4, (96, 49): 2}),
    (32, {(384, 4): 4, (192, 16): 2}),
    (None, {(384, 4): 2}),
)
DEFAULT_WINO4_KSPLIT = DEFAULT_WINO4_KSPLIT_BY_BATCH[-1][1]        # the headline batch's policy
#: split-K of implicit-GEMM layers by batch bucket: {(Cin, ksize, most output pixels per image): K slices}.
#: float32 (profiles/r05l_direct_ksplit_sweep.txt): the head's 1x1 GEMMs (K = 2,048: 64 chunks) and the stride-2
#: fuse convs on the small maps shorten the B <= 8 chain: backbone B = 1 3.56 -> 3.24 ms, B = 8 4.06 -> 3.86;
#: nothing from B = 32 on.  bf16 storage: no policy helps at bs 32 / 64 (its launches are not bound by their K
#: loops: 384 @7x7 23 -> 19 us isolated, nil end to end; 192 @14x14 16 -> 23 us) -- none.
DEFAULT_DIRECT_KSPLIT_BY_BATCH = (
    (8, {(2048, 1, 49): 4, (1536, 1, 49): 4, (512, 1, 49): 2, (192, 3, 49): 4, (96, 3, 196): 2, (96, 3, 49): 2}),
    (None, {}),
)
DEFAULT_DIRECT_KSPLIT_BY_BATCH_BF16 = ((None, {}),)


# ------------------------------------------------------------------------------------------
# parameter containers (names follow torchvision 0.8.2 BasicBlock / Bottleneck, which the
# reference instantiates at hrnet.py:13,196-199,369-370)
# ------------------------------------------------------------------------------------------
class _Container(nn.Module):
    def forward(self, *a, **k):
        raise RuntimeError('parameter container: executed by the HIP engine, not by torch')


class BasicBlock(_Container):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride


class Bottleneck(_Container):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = downsample
        self.stride = stride


blocks_dict = {'BASIC': BasicBlock, 'BOTTLENECK': Bottleneck}


class HighResolutionModule(_Container):
    """Parameter layout of hrnet.py:29-170."""

    def __init__(self, num_branches, block, num_blocks, num_inchannels, num_channels,
                 fuse_method, multi_scale_output=True):
        super().__init__()
        if not (num_branches == len(num_blocks) == len(num_channels) == len(num_inchannels)):
            raise ValueError('NUM_BRANCHES <> NUM_BLOCKS / NUM_CHANNELS / NUM_INCHANNELS')
        self.num_inchannels = list(num_inchannels)
        self.fuse_method = fuse_method
        self.num_branches = num_branches
        self.multi_scale_output = multi_scale_output
        branches = []
        for i in range(num_branches):
            downsample = None
            if self.num_inchannels[i] != num_channels[i] * block.expansion:
                downsample = nn.Sequential(
                    nn.Conv2d(self.num_inchannels[i], num_channels[i] * block.expansion, 1, 1,
                              bias=False),
                    nn.BatchNorm2d(num_channels[i] * block.expansion, momentum=BN_MOMENTUM))
            layers = [block(self.num_inchannels[i], num_channels[i], 1, downsample)]
            self.num_inchannels[i] = num_channels[i] * block.expansion
            for _ in range(1, num_blocks[i]):
                layers.append(block(self.num_inchannels[i], num_channels[i]))
            branches.append(nn.Sequential(*layers))
        self.branches = nn.ModuleList(branches)
        self.fuse_layers = self._make_fuse_layers()

    def _make_fuse_layers(self):
        if self.num_branches == 1:
            return None
        nb, ch = self.num_branches, self.num_inchannels
        fuse_layers = []
        for i in range(nb if self.multi_scale_output else 1):
            fuse_layer = []
            for j in range(nb):
                if j > i:
                    fuse_layer.append(nn.Sequential(
                        nn.Conv2d(ch[j], ch[i], 1, 1, 0, bias=False),
                        nn.BatchNorm2d(ch[i]),
                        nn.Upsample(scale_factor=2 ** (j - i), mode='nearest')))
                elif j == i:
                    fuse_layer.append(None)
                else:
                    convs = []
                    for k in range(i - j):
                        if k == i - j - 1:
                            convs.append(nn.Sequential(
                                nn.Conv2d(ch[j], ch[i], 3, 2, 1, bias=False),
                                nn.BatchNorm2d(ch[i])))
                        else:
                            convs.append(nn.Sequential(
                                nn.Conv2d(ch[j], ch[j], 3, 2, 1, bias=False),
                                nn.BatchNorm2d(ch[j]), nn.ReLU(True)))
                    fuse_layer.append(nn.Sequential(*convs))
            fuse_layers.append(nn.ModuleList(fuse_layer))
        return nn.ModuleList(fuse_layers)

    def get_num_inchannels(self):
        return self.num_inchannels


from .prefetch import ProloguePrefetch, cut_of, ops_from
from .plan import _Buf, _Plan  # noqa: E402,F401  (the op list, its ordering rules and the workspace packing)


def _fold(conv, bn):
    """conv (+bias) followed by eval-mode BN -> (weight OHWI, bias) in float64 -> float32.
    BN: y = (x - mean) / sqrt(var + eps) * gamma + beta  (eps 1e-5)."""
    w = conv.weight.detach().double().cpu()
    cout = w.shape[0]
    b = conv.bias.detach().double().cpu() if conv.bias is not None else torch.zeros(cout, dtype=torch.float64)
    if bn is not None:
        scale = bn.weight.detach().double().cpu() / torch.sqrt(
            bn.running_var.detach().double().cpu() + bn.eps)
        w = w * scale.view(-1, 1, 1, 1)
        b = (b - bn.running_mean.detach().double().cpu()) * scale + bn.bias.detach().double().cpu()
    w = w.permute(0, 2, 3, 1).contiguous()          # OIHW -> OHWI
    return w.float().numpy(), b.float().numpy()


class HighResolutionNet(VersionedWeights, nn.Module):

    def __init__(self, cfg, **kwargs):
        super().__init__()
        self.inplanes = 64
        self.use_old_impl = bool(cfg.get('use_old_impl', False))
        if self.use_old_impl:
            raise NotImplementedError('use_old_impl=True is not used by any SHAPY config')
        self.conv1 = nn.Conv2d(3, 64, 3, 2, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(64, momentum=BN_MOMENTUM)
        self.conv2 = nn.Conv2d(64, 64, 3, 2, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(64, momentum=BN_MOMENTUM)

        self.stage1_cfg = cfg.get('stage1', {})
        num_channels = self.stage1_cfg['num_channels'][0]
        block = blocks_dict[self.stage1_cfg['block']]
        num_blocks = self.stage1_cfg['num_blocks'][0]
        self.layer1 = self._make_layer(block, num_channels, num_blocks)
        stage1_out_channel = block.expansion * num_channels

        self.stage2_cfg = cfg.get('stage2', {})
        block = blocks_dict[self.stage2_cfg.get('block')]
        num_channels = [c * block.expansion for c in self.stage2_cfg.get('num_channels', (32, 64))]
        stage2_num_channels = num_channels
        self.transition1 = self._make_transition_layer([stage1_out_channel], num_channels)
        self.stage2, pre = self._make_stage(self.stage2_cfg, num_channels)

        self.stage3_cfg = cfg.get('stage3')
        block = blocks_dict[self.stage3_cfg['block']]
        num_channels = [c * block.expansion for c in self.stage3_cfg['num_channels']]
        stage3_num_channels = num_channels
        self.transition2 = self._make_transition_layer(pre, num_channels)
        self.stage3, pre = self._make_stage(self.stage3_cfg, num_channels)

        self.stage4_cfg = cfg.get('stage4')
        block = blocks_dict[self.stage4_cfg['block']]
        num_channels = [c * block.expansion for c in self.stage4_cfg['num_channels']]
        self.transition3 = self._make_transition_layer(pre, num_channels)
        self.stage4, pre = self._make_stage(self.stage4_cfg, num_channels)
        stage4_num_channels = num_channels
        self.output_channels_dim = pre
        self.pretrained_layers = list(cfg['pretrained_layers'])

        in_dims = 4 * 384                       # hrnet.py:279
        self.subsample_4 = self._make_subsample_layer(stage4_num_channels[0], 3)
        self.subsample_3 = self._make_subsample_layer(stage2_num_channels[-1], 2)
        self.subsample_2 = self._make_subsample_layer(stage3_num_channels[-1], 1)
        self.conv_layers = self._make_conv_layer(in_dims, 5)
        self.init_weights()
        self._engine = {}
        self.multi_stream = True
        self.tile_overrides = {}
        self.tile_flags = 0          # OR-ed into every conv's tile id (tuning knobs)
        #: replay the forward as one hipGraph (csrc/capi.hip): True (the captured barrier plan), False, or 'auto' =
        #: True for batches up to graph_max_batch = 0, i.e. never: the eager event-driven forward is faster than the
        #: replay at every batch size (B = 1: 5.5 vs 6.3 ms, B = 64: 12.8 vs 14.0)
        self.use_graph = 'auto'
        self.graph_max_batch = 0
        #: 'f32' = exact-f32 MFMA (parity path); 'f32x6' = float32 storage, products from the
        #: exact 3-way bf16 split on the bf16 matrix cores (float32-class accuracy);
        #: 'bf16' = bf16 weights/activations, f32 accumulate
        self.compute_dtype = 'f32'
        #: opt-in (0 = never, the default): in 'f32' forwards of at least this many images the head's wide 1x1 GEMMs
        #: (Cin, Cout >= 512) take the bf16x6 arithmetic layer by layer (SHAPY_TILE_X6) -- 2048 -> 2048 at M = 3,136:
        #: 161 vs 253 us, bs 64 +2.3 % (profiles/r05q_*); the default keeps every f32 product on the f32 matrix cores
        self.x6_gemm_min_batch = int(os.environ.get('SHAPY_X6_GEMM_MIN_BATCH', '0'))
        self._x6_eff = False
        #: float32 convolution algorithm of the 3x3 / stride-1 layers (float32-class results, same 1e-4 parity tests):
        #:   'winograd4' (default) = Winograd F(4x4,3x3) (csrc/conv_wino4.hip: 36 multiplies per 4x4 outputs, 48-channel
        #:       N tiles) on maps of at least wino4_min_hw pixels a side, F(2x2,3x3) on the rest
        #:   'winograd' = F(2x2,3x3) (csrc/conv_wino.hip, 2.25x fewer MFMAs than direct) wherever the kernel applies
        #:   'direct' = implicit GEMM for every layer (the exact-f32 fmaf chain of the reference's sum order)
        #:   'auto' = F(2x2) only on maps of at least wino_min_hw pixels a side
        #: (SHAPY_CONV_ALGO / SHAPY_WINO4_MIN_HW override the defaults for A/B runs of whole suites and benches)
        self.conv_algo = os.environ.get('SHAPY_CONV_ALGO', DEFAULT_CONV_ALGO)
        self.wino_min_hw = 14
        self.wino4_min_hw = int(os.environ.get('SHAPY_WINO4_MIN_HW', DEFAULT_WINO4_MIN_HW))
        #: F(4x4) split-K: {(Cin, most 4x4 tiles per image): S} -- a layer with that many input channels
        #: on a map of at most that many tiles runs S workgroups per output tile, each over Cin / S
        #: channels (csrc/conv_wino4.hip).  The policy is chosen per BATCH BUCKET (wino4_ksplit_by_batch:
        #: B <= 8, B <= 32, larger), each bucket with a plan of its own: inside a bucket the features are
        #: bit-identical from batch size to batch size, between buckets the split layers differ by
        #: float32 rounding (another association of the same sums), as any algorithm choice by shape does.
        #: wino4_ksplit = a dict: that policy at every batch size (tests, A/B runs); None = by bucket.
        #: (SHAPY_W4_KSPLIT="384@4:2,192@16:2" sets such a dict; "" = no split anywhere.)
        self.wino4_ksplit_by_batch = tuple((b, dict(p)) for b, p in DEFAULT_WINO4_KSPLIT_BY_BATCH)
        self.wino4_ksplit = None
        if 'SHAPY_W4_KSPLIT' in os.environ:
            self.wino4_ksplit = {}
            for item in filter(None, os.environ['SHAPY_W4_KSPLIT'].split(',')):
                key, sl = item.split(':')
                cin, tmax = key.split('@')
                self.wino4_ksplit[(int(cin), int(tmax))] = int(sl)
        self._ksplit_eff = None          # the policy of the plan being built (None: the largest bucket's)
        #: split-K of the implicit-GEMM layers, same idea: {(Cin, ksize, most output pixels per image): S} per
        #: batch bucket, float32 and bf16 storage apart (direct_ksplit = a dict pins one policy)
        self.direct_ksplit_by_batch = tuple((b, dict(p)) for b, p in DEFAULT_DIRECT_KSPLIT_BY_BATCH)
        self.direct_ksplit_by_batch_bf16 = tuple((b, dict(p)) for b, p in DEFAULT_DIRECT_KSPLIT_BY_BATCH_BF16)
        self.direct_ksplit = None
        self._direct_ksplit_eff = None

        #: conv_algo='winograd4': the convs at the same depth of a HighResolutionModule's parallel
        #: branches as ONE persistent grouped launch (csrc/conv_wino4g.hip) instead of one launch
        #: per branch on its own stream
        #: 'auto' (default) = only when the forward runs on ONE stream: on four streams the per-layer
        #: launches overlap the fuse layers of a module with the first convs of the next and are the
        #: faster configuration at every batch size measured (B = 64: 4,745 vs 4,662 images/s; one
        #: stream: 14.1 vs 16.3 ms per step in favour of the groups; profiles/r03m_*)
        self.group_branches = {'1': True, '0': False}.get(os.environ.get('SHAPY_GROUP_BRANCHES', ''), 'auto')
        #: order of the terms of a fuse output (False: ascending branch index as in rounds 1-5)
        self.fuse_near_first = True
        #: multi-stream plans: explicit dependencies (events) instead of a join between the branches
        #: and the fuse layers of a module (False: the round-2 barrier plan)
        self.dag = True
        #: Winograd numerics guard: per-layer demotions {op name: 'winograd' (F(2x2)) | 'direct'}: the
        #: caller's own entries AND what ``calibrate`` added for the CURRENT weights (the guard never
        #: removes a caller's entry); wino_guard = run the calibration on the first float32 batch after
        #: the weights changed (SHAPY_WINO_GUARD=0 disables)
        self.layer_algo = {}
        self._guard_demotions = {}       # name -> (the guard's value, the caller's entry it replaced | None)
        self.wino_guard = os.environ.get('SHAPY_WINO_GUARD', '1') != '0'
        self.wino_budget = 2e-5          # rms(winograd - direct) / rms(direct) per layer
        #: probe of the automatic calibration: 'fixed' (default) = four SEEDED synthetic crops of the
        #: input's size -- the same on every rank and in every run, so that all ranks compile the same
        #: plan and features stay bit-identical between them; 'batch' = up to 8 images of the first
        #: batch.  ``calibrate(x)`` on images of your own is always available.
        self.wino_guard_probe = os.environ.get('SHAPY_WINO_GUARD_PROBE', 'fixed')
        #: runtime tripwire: > 0 = every that many forwards two images of the LIVE batch are re-checked
        #: layer by layer (one extra pass over the op list with host syncs: ~30 ms) and layers over
        #: budget are demoted; 0 = off (default: the guard judges weights, not images)
        self.wino_guard_recheck_every = int(os.environ.get('SHAPY_WINO_GUARD_RECHECK', '0'))
        self._n_forward = 0
        self._prefetch = ProloguePrefetch()
        #: forward(x, prefetch=next_x): up to this size the prologue reaches to stage 2 and goes BEFORE the own launches
        self.prefetch_before_max_batch = 16
        self._capture_warned = False
        self.calibration_report = None
        self._calibrated_ver = None
        self._engine_ver = None
        self.register_load_state_dict_post_hook(lambda m, k: m.invalidate())

    # ---- construction helpers (mirror hrnet.py:301-424) ----
    def _make_transition_layer(self, pre, cur):
        layers = []
        for i in range(len(cur)):
            if i < len(pre):
                if cur[i] != pre[i]:
                    layers.append(nn.Sequential(nn.Conv2d(pre[i], cur[i], 3, 1, 1, bias=False),
                                                nn.BatchNorm2d(cur[i]), nn.ReLU(inplace=True)))
                else:
                    layers.append(None)
            else:
                convs = []
                for j in range(i + 1 - len(pre)):
                    inch = pre[-1]
                    outch = cur[i] if j == i - len(pre) else inch
                    convs.append(nn.Sequential(nn.Conv2d(inch, outch, 3, 2, 1, bias=False),
                                               nn.BatchNorm2d(outch), nn.ReLU(inplace=True)))
                layers.append(nn.Sequential(*convs))
        return nn.ModuleList(layers)

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                nn.Conv2d(self.inplanes, planes * block.expansion, 1, stride, bias=False),
                nn.BatchNorm2d(planes * block.expansion, momentum=BN_MOMENTUM))
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes))
        return nn.Sequential(*layers)

    def _make_conv_layer(self, in_channels=2048, num_layers=3, num_filters=2048):
        layers = []
        for _ in range(num_layers):
            downsample = nn.Conv2d(in_channels, num_filters, 1, 1, bias=False)
            layers.append(Bottleneck(in_channels, num_filters // 4, downsample=downsample))
            in_channels = num_filters
        return nn.Sequential(*layers)

    def _make_subsample_layer(self, in_channels=96, num_layers=3, stride=2):
        layers = []
        for _ in range(num_layers):
            layers.append(nn.Conv2d(in_channels, 2 * in_channels, 3, stride, 1))
            in_channels *= 2
            layers.append(nn.BatchNorm2d(in_channels, momentum=BN_MOMENTUM))
            layers.append(nn.ReLU(inplace=True))
        return nn.Sequential(*layers)

    def _make_stage(self, layer_config, num_inchannels, multi_scale_output=True):
        block = blocks_dict[layer_config['block']]
        modules = []
        for i in range(layer_config['num_modules']):
            reset = not (not multi_scale_output and i == layer_config['num_modules'] - 1)
            modules.append(HighResolutionModule(
                layer_config['num_branches'], block, layer_config['num_blocks'],
                num_inchannels, layer_config['num_channels'], layer_config['fuse_method'],
                reset))
            num_inchannels = modules[-1].get_num_inchannels()
        return nn.Sequential(*modules), num_inchannels

    def get_output_dim(self):
        base = {f'layer{i + 1}': v for i, v in enumerate(self.output_channels_dim)}
        out = dict(base)
        for k in base:
            out[f'{k}_avg_pooling'] = base[k]
        out['concat'] = 2048
        return out

    def init_weights(self):
        """hrnet.py:500-516."""
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.normal_(m.weight, std=0.001)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def load_weights(self, pretrained=''):
        """hrnet.py:518-534."""
        pretrained = osp.expandvars(pretrained or '')
        if osp.isfile(pretrained):
            sd = torch.load(pretrained, map_location=torch.device('cpu'))
            need = {k: v for k, v in sd.items()
                    if k.split('.')[0] in self.pretrained_layers or self.pretrained_layers[0] == '*'}
            self.load_state_dict(need, strict=False)
        elif pretrained:
            raise ValueError('{} is not exist!'.format(pretrained))

    # ---- engine ----
    def invalidate(self):
        """Drops the folded weight blob, the op lists and the captured hipGraphs.  Called
        automatically when a parameter / buffer version changes (``_compile``), after
        ``load_state_dict`` and after ``.to()`` / ``.cuda()``."""
        self._engine = {}
        self.__dict__.pop('_xform_cache', None)
        self._calibrated_ver = None
        self._drop_guard_demotions()
        self._drop_version_cache()

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._engine = {}
        self.__dict__.pop('_xform_cache', None)
        self._drop_version_cache()
        return out

    def __getstate__(self):
        st = self.__dict__.copy()
        st.pop('_xform_cache', None)
        st['_engine'], st['_ver_tensors'], st['_prefetch'] = {}, None, ProloguePrefetch()
        return st

    _dag_eff = False

    def _group_on(self):
        g = self.group_branches
        return (not self.multi_stream) if g == 'auto' else bool(g)

    def _use_wino(self, ks, st, pad, cin, cout, Hi, Wi, ups):
        if self.conv_algo == 'direct' or not winograd.eligible(ks, st, pad, cin, cout, ups):
            return False
        return self.conv_algo in ('winograd', 'winograd4') or min(Hi, Wi) >= self.wino_min_hw

    def _use_wino4(self, ks, st, pad, cin, cout, Hi, Wi, ups):
        if cout % 48:                 # the F(4x4) kernel's N tile (layer1's 64 -> 64, the head's 512 -> 512: F(2x2))
            return False
        return (self.conv_algo == 'winograd4' and min(Hi, Wi) >= self.wino4_min_hw
                and winograd.eligible4(ks, st, pad, cin, cout, ups))

    def ksplit_policy(self, B=None):
        """The split-K policy of a forward with batch size B (None: the largest bucket's)."""
        if self.wino4_ksplit is not None:
            return self.wino4_ksplit
        for max_b, pol in self.wino4_ksplit_by_batch:
            if max_b is None or (B is not None and B <= max_b):
                return pol
        return self.wino4_ksplit_by_batch[-1][1]

    def direct_ksplit_policy(self, B=None, bf16=False):
        """{(Cin, ksize, most output pixels per image): S} for the implicit-GEMM layers of a forward with batch B."""
        if self.direct_ksplit is not None:
            return self.direct_ksplit
        table = self.direct_ksplit_by_batch_bf16 if bf16 else self.direct_ksplit_by_batch
        for max_b, pol in table:
            if max_b is None or (B is not None and B <= max_b):
                return pol
        return table[-1][1]

    def _direct_ksplit(self, cin, ks, out_pixels, bf16):
        pol = self._direct_ksplit_eff if self._direct_ksplit_eff is not None else self.direct_ksplit_policy(None, bf16)
        eps = 8 if bf16 else 4
        if cin % (4 * eps):                    # the flat-K bf16 kernel has no split form
            return 1
        best = 1
        for (c, k, pmax), sl in pol.items():
            if c == cin and k == ks and out_pixels <= pmax:
                best = max(best, int(sl))
        # every slice needs at least one K chunk, whichever chunk length the library picks (up to 8 slots)
        n_min = ks * ks * (cin // (8 * eps)) if cin % (8 * eps) == 0 else ks * ks * (cin // (4 * eps))
        while best > 1 and (best - 1) * -(-n_min // best) >= n_min:
            best -= 1
        return best

    def _xform(self, name, kind, w):
        """Winograd-transformed filters of a layer (float64 transform on the host), memoised across the plans of
        one weight version: the batch buckets compile the same 209 transforms otherwise (seconds each time)."""
        cache = self.__dict__.setdefault('_xform_cache', {})
        ver = self._engine_ver
        if cache.get('ver') != ver:
            cache.clear()
            cache['ver'] = ver
        key = (name, kind, w.shape)
        if key not in cache or not name:
            u = winograd.transform_filters4(w) if kind == 4 else winograd.transform_filters(w)
            if not name:
                return u
            cache[key] = u
        return cache[key]

    def _ksplit(self, cin, Hi, Wi):
        """K slices of an F(4x4) layer under the policy of the plan being built; 1 = no split."""
        t = ((Hi + 3) // 4) * ((Wi + 3) // 4)
        best = 1
        pol = self._ksplit_eff if self._ksplit_eff is not None else self.ksplit_policy(None)
        for (c, tmax), sl in pol.items():
            if c == cin and t <= tmax and (cin // 16) % sl == 0:
                best = max(best, int(sl))
        return best

    def _build_plan(self, H, W, bf16=False, x6=False):
        P = _Plan(bf16, x6)
        ov = self.tile_overrides
        if self.conv_algo not in ('direct', 'winograd', 'winograd4', 'auto'):
            raise ValueError(f'unknown conv_algo {self.conv_algo!r}')

        def conv(conv_m, bn, inb, Hi, Wi, outb=None, res=None, relu=False, ups=1, lane=0,
                 out_ld=None, out_coff=0, res_ld=None, res_coff=0, name='', group=0,
                 group_member=False):
            ks, st, pad = conv_m.kernel_size[0], conv_m.stride[0], conv_m.padding[0]
            cin, cout = conv_m.in_channels, conv_m.out_channels
            cin_p, cout_p = P.padc(cin), P.padc(cout)
            assert inb.C == cin_p, (name, inb.C, cin_p)
            Ho, Wo = (Hi + 2 * pad - ks) // st + 1, (Wi + 2 * pad - ks) // st + 1
            w, b = _fold(conv_m, bn)
            if (cin_p, cout_p) != (cin, cout):
                wp = np.zeros((cout_p, ks, ks, cin_p), np.float32)
                wp[:cout, :, :, :cin] = w
                bp = np.zeros(cout_p, np.float32)
                bp[:cout] = b
                w, b = wp, bp
            if outb is None:
                outb = P.buf(Ho * ups, Wo * ups, cout_p)
            wino_off, wino_flag = -1, 0
            scrb, cnt_off, cnt_n = None, -1, 0
            forced = self.layer_algo.get(name)       # Winograd guard (calibrate): per-layer demotion
            if forced == 'direct':
                pass
            elif forced == 'winograd' and not (bf16 or x6) and winograd.eligible(ks, st, pad, cin_p, cout_p, ups):
                wino_off = P.add_weights(self._xform(name, 2, w))
            elif not (bf16 or x6) and self._use_wino4(ks, st, pad, cin_p, cout_p, Hi, Wi, ups):
                wino_off = P.add_weights(self._xform(name, 4, w))
                wino_flag = _lib.TILE_WINO4
                # split-K (never inside a persistent grouped launch, which has no such form)
                sl = 1 if group_member else self._ksplit(cin_p, Hi, Wi)
                if sl > 1:
                    slab, cnt_n = _lib.w4_split_sizes(Hi, Wi, cout_p, sl)
                    scrb = P.buf(1, 1, slab)
                    cnt_off, P.cnt_ints = P.cnt_ints, P.cnt_ints + cnt_n
                    wino_flag |= _lib.tile_w4_ksplit(sl)
            elif not (bf16 or x6) and self._use_wino(ks, st, pad, cin_p, cout_p, Hi, Wi, ups):
                wino_off = P.add_weights(self._xform(name, 2, w))
            if wino_off < 0 and not x6 and ups == 1 and not group_member:
                # implicit-GEMM layers (every conv in bf16; strided / 1x1 / demoted layers in f32): split-K for
                # the K-deep ones on small maps (csrc/conv_igemm.hip; never the flat-K bf16 kernel)
                sl = self._direct_ksplit(cin_p, ks, Ho * Wo, bf16)
                if sl > 1:
                    slab, cnt_n = _lib.igemm_split_sizes(Ho, Wo, cout_p, sl)
                    scrb = P.buf(1, 1, slab * (2 if bf16 else 1))    # float32 partials in a bf16-element arena
                    cnt_off, P.cnt_ints = P.cnt_ints, P.cnt_ints + cnt_n
                    wino_flag |= _lib.tile_w4_ksplit(sl)
            lx6 = (self._x6_eff and not (bf16 or x6) and wino_off < 0 and scrb is None and ks == 1 and st == 1
                   and ups == 1 and cin_p >= 512 and cout_p >= 512 and cin_p % 32 == 0)
            wino_flag |= _lib.TILE_X6 if lx6 else 0
            P.op(type=_lib.OP_CONV, lane=lane, inb=inb, outb=outb, resb=res, scrb=scrb, cnt_off=cnt_off,
                 cnt_n=cnt_n,
                 Hi=Hi, Wi=Wi, Cin=cin_p,
                 in_ld=inb.C, Ho=Ho, Wo=Wo, Cout=cout_p, ksize=ks, stride=st, pad=pad,
                 out_ld=out_ld or outb.C, out_coff=out_coff,
                 res_ld=(res_ld or (res.C if res is not None else 0)), res_coff=res_coff,
                 relu=int(relu), ups=ups,
                 tile=_lib.TILES[ov.get(name, 'auto')] | self.tile_flags | wino_flag,
                 wgt_off=P.add_conv_weights(w, x6=lx6), bias_off=P.add_weights(b), wino_off=wino_off,
                 name=name, group=group)
            return outb, Ho, Wo

        # stem (hrnet.py:427-432)
        H1, W1 = H // 2, W // 2
        w, b = _fold(self.conv1, self.bn1)
        s1 = P.buf(H1, W1, 64)
        P.op(type=_lib.OP_STEM, lane=0, inb=None, outb=s1, resb=None, Hi=H, Wi=W, Cin=3, in_ld=0,
             Ho=H1, Wo=W1, Cout=64, ksize=3, stride=2, pad=1, out_ld=64, out_coff=0, res_ld=0,
             res_coff=0, relu=1, ups=1, tile=0, wgt_off=P.add_weights(w.reshape(64, 27)),
             bias_off=P.add_weights(b), wino_off=-1)   # the stem keeps float32 weights in both modes
        x, Hc, Wc = conv(self.conv2, self.bn2, s1, H1, W1, relu=True, name='conv2')

        def bottleneck(m, x, Hc, Wc, lane=0, side_lane=None, name=''):
            idt = x
            if m.downsample is not None:
                if isinstance(m.downsample, nn.Conv2d):       # conv_layers: bare 1x1 conv
                    if side_lane is not None:
                        P.barrier()
                    idt, _, _ = conv(m.downsample, None, x, Hc, Wc,
                                     lane=side_lane if side_lane is not None else lane,
                                     name=name + '.downsample')
                else:
                    # event-driven plans: the projection of layer1's first Bottleneck (64 -> 256, 82 us
                    # at HBM speed) runs beside conv1 / conv2 on lane 1; conv3 waits for its event
                    idt, _, _ = conv(m.downsample[0], m.downsample[1], x, Hc, Wc,
                                     lane=1 if (self._dag_eff and lane == 0 and side_lane is None) else lane,
                                     name=name + '.downsample')
            t, _, _ = conv(m.conv1, m.bn1, x, Hc, Wc, relu=True, lane=lane, name=name + '.conv1')
            t, _, _ = conv(m.conv2, m.bn2, t, Hc, Wc, relu=True, lane=lane, name=name + '.conv2')
            if side_lane is not None:
                P.barrier()
            o, _, _ = conv(m.conv3, m.bn3, t, Hc, Wc, res=idt, relu=True, lane=lane,
                           name=name + '.conv3')
            return o

        for bi, m in enumerate(self.layer1):
            x = bottleneck(m, x, Hc, Wc, name=f'layer1.{bi}')

        def seq_conv_bn_relu(seq, x, Hc, Wc, lane, name):
            # Sequential(conv, bn, relu) or Sequential(Sequential(conv,bn,relu), ...)
            if isinstance(seq[0], nn.Conv2d):
                return conv(seq[0], seq[1], x, Hc, Wc, relu=True, lane=lane, name=name)
            for q, s in enumerate(seq):
                x, Hc, Wc = conv(s[0], s[1], x, Hc, Wc, relu=True, lane=lane, name=f'{name}.{q}')
            return x, Hc, Wc

        def module(m, xs, last_out=None, name=''):
            """xs: list of (buf, H, W).  HighResolutionModule.forward (hrnet.py:175-193)."""
            nb = m.num_branches
            ys = []
            depth = len(m.branches[0])
            grouped = (self._group_on() and not (bf16 or x6) and 2 <= nb <= 4
                       and all(len(br) == depth for br in m.branches)
                       and all(self._use_wino4(3, 1, 1, c.in_channels, c.out_channels, xs[i][1],
                                               xs[i][2], 1)
                               for i in range(nb) for blk in m.branches[i]
                               for c in (blk.conv1, blk.conv2)))
            if grouped:
                # level-major: the convs at the same depth of the nb branches are independent ->
                # ONE persistent F(4x4) launch per level on the main stream (csrc/conv_wino4g.hip)
                # instead of nb launches on nb streams; the reference walks branch by branch
                # (hrnet.py:175-179)
                P.barrier()
                cur = [xs[i][0] for i in range(nb)]
                for bi in range(depth):
                    ts = []
                    for i in range(nb):
                        blk = m.branches[i][bi]
                        t, _, _ = conv(blk.conv1, blk.bn1, cur[i], xs[i][1], xs[i][2], relu=True,
                                       lane=0, name=f'{name}.branches.{i}.{bi}.conv1',
                                       group=nb if i == 0 else 0, group_member=True)
                        ts.append(t)
                    for i in range(nb):
                        blk = m.branches[i][bi]
                        cur[i], _, _ = conv(blk.conv2, blk.bn2, ts[i], xs[i][1], xs[i][2],
                                            res=cur[i], relu=True, lane=0,
                                            name=f'{name}.branches.{i}.{bi}.conv2',
                                            group=nb if i == 0 else 0, group_member=True)
                ys = [(cur[i], xs[i][1], xs[i][2]) for i in range(nb)]
            else:
                for i in range(nb):
                    x, Hc, Wc = xs[i]
                    for bi, blk in enumerate(m.branches[i]):
                        t, _, _ = conv(blk.conv1, blk.bn1, x, Hc, Wc, relu=True, lane=i,
                                       name=f'{name}.branches.{i}.{bi}.conv1')
                        x, _, _ = conv(blk.conv2, blk.bn2, t, Hc, Wc, res=x, relu=True, lane=i,
                                       name=f'{name}.branches.{i}.{bi}.conv2')
                    ys.append((x, Hc, Wc))
            # dag: no join between the branches and the fuse layers -- every fuse conv waits for
            # exactly the tensors it reads (events, _Plan.sync_plan) -- and the leading convs of the
            # stride-2 chains (fuse_layers[i][j], i - j >= 2) run on the lane of their SOURCE branch
            # (free as soon as that branch is done), so that the output of the smallest map no longer
            # queues six convs on one stream
            dag = self._dag_eff and not grouped
            if not dag:
                P.barrier()
            lead = {}                       # (i, j) -> (tensor, H, W) behind the chain's leading convs
            if dag:
                # enqueue order = stream order: the leading convs of the stride-2 chains first, so
                # that they sit in front of their lane's accumulating convs (which wait for the
                # OTHER branches)
                for i in range(len(m.fuse_layers)):
                    for j in range(i - 1):
                        fl = m.fuse_layers[i][j]
                        t, Ht, Wt = ys[j]
                        for k in range(i - j - 1):
                            t, Ht, Wt = conv(fl[k][0], fl[k][1], t, Ht, Wt, relu=True,
                                             lane=j, name=f'{name}.fuse_layers.{i}.{j}.{k}')
                        lead[(i, j)] = (t, Ht, Wt)
            outs = []
            for i in range(len(m.fuse_layers)):
                xi, Hi_, Wi_ = ys[i]
                terms = [j for j in range(nb) if j != i]
                if self.fuse_near_first:
                    # stride-2 terms from the NEAREST branch first: its input is that branch's own output (ready when
                    # the branch ends, no leading convs) and its conv is the widest (192 -> 384 at 14x14: 4.2 GFLOP);
                    # the chain from branch 0 arrives last and ends in the narrowest conv.  In ascending order the wide
                    # conv ran last on the lane that then starts the next module's 7x7 branch 340-430 us late
                    # (profiles/r06g_module_tails.txt).  Same terms, another summation order (float32 rounding).
                    terms = [j for j in range(i - 1, -1, -1)] + [j for j in range(i + 1, nb)]
                use_last = last_out is not None and i == nb - 1
                outb = last_out[0] if use_last else P.buf(Hi_, Wi_, xi.C)
                o_ld = last_out[1] if use_last else xi.C
                o_co = last_out[2] if use_last else 0
                for ti, j in enumerate(terms):
                    first, last = ti == 0, ti == len(terms) - 1
                    res = xi if first else outb
                    r_ld = xi.C if first else o_ld
                    r_co = 0 if first else o_co
                    xj, Hj, Wj = ys[j]
                    fl = m.fuse_layers[i][j]
                    nm = f'{name}.fuse_layers.{i}.{j}'
                    if j > i:
                        conv(fl[0], fl[1], xj, Hj, Wj, outb=outb, res=res, relu=last,
                             ups=2 ** (j - i), lane=i, out_ld=o_ld, out_coff=o_co, res_ld=r_ld,
                             res_coff=r_co, name=nm)
                    else:
                        t, Ht, Wt = lead.get((i, j), (xj, Hj, Wj))
                        k0 = i - j - 1 if (i, j) in lead else 0
                        for k in range(k0, i - j):
                            if k == i - j - 1:
                                conv(fl[k][0], fl[k][1], t, Ht, Wt, outb=outb, res=res, relu=last,
                                     lane=i, out_ld=o_ld, out_coff=o_co, res_ld=r_ld,
                                     res_coff=r_co, name=f'{nm}.{k}')
                            else:
                                t, Ht, Wt = conv(fl[k][0], fl[k][1], t, Ht, Wt, relu=True, lane=i,
                                                 name=f'{nm}.{k}')
                outs.append((outb, Hi_, Wi_))
            return outs

        # transition1 (hrnet.py:435-440)
        P.barrier()
        xs = []
        for i, tr in enumerate(self.transition1):
            xs.append((x, Hc, Wc) if tr is None else
                      seq_conv_bn_relu(tr, x, Hc, Wc, i, f'transition1.{i}'))
        ys = xs
        stages = [(self.stage2, self.transition2), (self.stage3, self.transition3),
                  (self.stage4, None)]
        cat = None
        for si, (stage, trans) in enumerate(stages):
            for mi, m in enumerate(stage):
                last = None
                if trans is None and mi == len(stage) - 1:
                    # the last module writes branch 3 straight into the concat buffer (x1)
                    Hl, Wl = ys[-1][1], ys[-1][2]
                    cat = P.buf(Hl, Wl, 4 * 384)
                    last = (cat, 4 * 384, 3 * 384)
                ys = module(m, ys, last_out=last, name=f'stage{si + 2}.{mi}')
            if trans is not None:
                P.barrier()
                nxt = []
                for i, tr in enumerate(trans):
                    if tr is None:
                        nxt.append(ys[i])
                    else:
                        src = ys[i] if i < len(ys) else ys[-1]
                        nxt.append(seq_conv_bn_relu(tr, src[0], src[1], src[2], i,
                                                    f'transition{si + 2}.{i}'))
                ys = nxt

        # head (hrnet.py:477-486): cat[x4, x3, x2, x1] -> 5 bottlenecks -> spatial mean
        P.barrier()

        def subsample(seq, src, lane, coff, name):
            x, Hc, Wc = src
            n = len(seq) // 3
            for q in range(n):
                if q == n - 1:
                    conv(seq[3 * q], seq[3 * q + 1], x, Hc, Wc, outb=cat, relu=True, lane=lane,
                         out_ld=cat.C, out_coff=coff, name=f'{name}.{3 * q}')
                else:
                    x, Hc, Wc = conv(seq[3 * q], seq[3 * q + 1], x, Hc, Wc, relu=True, lane=lane,
                                     name=f'{name}.{3 * q}')
        subsample(self.subsample_4, ys[0], 0, 0, 'subsample_4')
        subsample(self.subsample_3, ys[1], 1, 384, 'subsample_3')
        subsample(self.subsample_2, ys[2], 2, 768, 'subsample_2')
        P.barrier()
        Hc, Wc = ys[3][1], ys[3][2]
        x = cat
        for bi, m in enumerate(self.conv_layers):
            x = bottleneck(m, x, Hc, Wc, lane=0, side_lane=1, name=f'conv_layers.{bi}')
            P.barrier()
        P.op(type=_lib.OP_MEANPOOL, lane=0, inb=x, outb=None, resb=None, Hi=Hc, Wi=Wc, Cin=x.C,
             in_ld=x.C, Ho=1, Wo=1, Cout=x.C, ksize=1, stride=1, pad=0, out_ld=x.C, out_coff=0,
             res_ld=0, res_coff=0, relu=0, ups=1, tile=0, wgt_off=-1, bias_off=-1, wino_off=-1)
        return P

    def _compile(self, H, W, device, graph=False, B=None):
        # event-driven plan: eager multi-stream forwards only (the captured hipGraph keeps the barrier form)
        self._dag_eff = bool(self.dag and self.multi_stream and not graph)
        if self.compute_dtype not in ('f32', 'f32x6', 'bf16'):
            raise ValueError(f'unknown compute_dtype {self.compute_dtype!r}')
        bf16 = self.compute_dtype == 'bf16'
        ver = self._weights_version()
        if ver != self._engine_ver:          # a parameter / buffer was edited in place
            self._engine = {}
            self._engine_ver = ver
        pol = self.ksplit_policy(B) if self.compute_dtype == 'f32' else {}
        dpol = self.direct_ksplit_policy(B, bf16) if self.compute_dtype in ('f32', 'bf16') else {}
        key_w = (H, W, str(device), self.compute_dtype, self.conv_algo, self.wino_min_hw,
                 self.wino4_min_hw, self._group_on(), self._dag_eff,
                 tuple(sorted(self.layer_algo.items())), self.tile_flags,
                 tuple(sorted(self.tile_overrides.items())))
        lx6 = self.compute_dtype == 'f32' and 0 < self.x6_gemm_min_batch <= (B or 0)
        key_w += (lx6, self.fuse_near_first)
        key = key_w + (tuple(sorted(pol.items())) + tuple(sorted(dpol.items())),)
        eng = self._engine.get(key)
        if eng is not None:
            return eng
        self._ksplit_eff, self._direct_ksplit_eff, self._x6_eff = pol, dpol, lx6
        try:
            P = self._build_plan(H, W, bf16, self.compute_dtype == 'f32x6')
        finally:
            self._ksplit_eff = self._direct_ksplit_eff = None
            self._x6_eff = False
        P.sync_plan()
        ws_per_img = P.allocate()
        n = len(P.ops)
        arr = (_lib.ShapyOp * n)()
        for i, o in enumerate(P.ops):
            a = arr[i]
            for f in ('type', 'lane', 'barrier_before', 'Hi', 'Wi', 'Cin', 'in_ld', 'Ho', 'Wo',
                      'Cout', 'ksize', 'stride', 'pad', 'out_ld', 'out_coff', 'res_ld', 'res_coff',
                      'relu', 'ups', 'tile', 'group', 'sig', 'wgt_off', 'bias_off', 'wino_off'):
                setattr(a, f, int(o[f]))
            for q in range(3):
                a.wait[q] = int(o['wait'][q])
            a.in_off = (-2 if o['type'] == _lib.OP_STEM else -1 if o['inb'] is None else o['inb'].off)
            a.out_off = -1 if o['outb'] is None else o['outb'].off
            a.res_off = -1 if o['resb'] is None else o['resb'].off
            a.split_off = -1 if o.get('scrb') is None else o['scrb'].off
            a.split_floats = 0 if o.get('scrb') is None else o['scrb'].size
            a.cnt_off = int(o.get('cnt_off', -1))
            a.cnt_n = int(o.get('cnt_n', 0))
        # the weight blob does not depend on the split policy (same layers, same order, same transforms):
        # plans that differ in nothing else share ONE device copy (1.3 GB with the default algorithm)
        weights = next((e['weights'] for k, e in self._engine.items()
                        if k[:-1] == key_w and e['weights'].numel() == P.wbytes), None)
        if weights is None:
            blob = np.frombuffer(b''.join(P.wchunks), dtype=np.uint8)
            weights = torch.from_numpy(blob.copy()).to(device)
        eng = dict(ops=arr, n_ops=n, weights=weights, ws_per_img=ws_per_img, cnt_per_img=P.cnt_ints,
                   plan=P, ws={}, graphs={}, cuts=(cut_of(P, 1), cut_of(P, 2)) if self._dag_eff else (0, 0),
                   feat_dim=P.ops[-1]['Cin'], esz=2 if bf16 else 4,
                   dtype={'f32': _lib.DTYPE_F32, 'bf16': _lib.DTYPE_BF16,
                          'f32x6': _lib.DTYPE_F32X6}[self.compute_dtype])
        self._engine[key] = eng
        return eng

    @staticmethod
    def _counters(eng, B, device):
        """Zeroed arrival counters of the plan's split-K layers for one workspace (None: no such layer).
        Every completed forward leaves them zero (include/shapy_hip.h: shapy_hrnet_run)."""
        n = eng['cnt_per_img'] * B
        return torch.zeros(n, dtype=torch.int32, device=device) if n else None

    # ---- Winograd numerics guard -----------------------------------------------------------
    def calibrate(self, x, budget=None, demote=True, log=None, graph=False):
        """One-batch calibration of the Winograd layers on probe images ``x`` [B,3,H,W] (cuda).

        F(4x4,3x3) with the points {0, +-1, +-2, inf} multiplies by up to 8 in its transforms and
        amplifies the DC part of its input; with He-initialised weights and O(1) activations that
        costs a factor ~5 over the direct sum (still float32-class), but nothing guarantees it for
        a trained checkpoint with wide BatchNorm scales and large post-ReLU means.  This walks the
        op list ONE LAYER AT A TIME on the engine's own buffers; every Winograd layer is run twice
        on its real input -- as planned, and on the direct kernel into a scratch buffer -- and its
        error  rms(winograd - direct) / rms(direct)  is recorded.  Layers above ``budget``
        (default ``wino_budget``) are demoted -- F(4x4) -> F(2x2) -> direct -- in ``layer_algo``,
        the plan is rebuilt and the pass repeated until nothing changes.  Every demotion is logged.
        Returns the report {'layers': [(name, algo, err_rms, err_max)], 'demoted': {...}}."""
        import logging
        log = log or logging.getLogger('shapy_amd.hrnet').warning
        budget = self.wino_budget if budget is None else budget
        _lib.require_cuda(x, 'images')
        if self.compute_dtype != 'f32':
            raise ValueError('calibrate() applies to the float32 path')
        lib = _lib.load()
        x = x.contiguous().float()
        B, _, H, W = x.shape
        report = None
        for _ in range(3):
            # the product's own engine (same plan key): the pass below runs its ops one at a
            # time on the caller's stream, whatever lanes / groups / events the plan carries
            eng = self._compile(H, W, x.device, graph=graph, B=B)
            layers = self._calibrate_pass(lib, eng, x)
            worst = {}
            for name, algo, e_rms, e_max in layers:
                if e_rms > budget:
                    worst[name] = 'winograd' if algo == 'winograd4' and winograd.eligible(
                        3, 1, 1, *self._op_channels(eng, name)) else 'direct'
            report = {'layers': layers, 'demoted': dict(self.layer_algo), 'budget': budget}
            if not worst or not demote:
                break
            for name, to in worst.items():
                e = next(l for l in layers if l[0] == name)
                log(f'Winograd guard: {name}: {e[1]} error {e[2]:.2e} rms-relative '
                    f'(max {e[3]:.2e}) > budget {budget:.1e} on the probe batch -> {to}')
                self._guard_demote(name, to)
            report['demoted'] = dict(self.layer_algo)
        self.calibration_report = report
        self._calibrated_ver = self._weights_version()
        return report

    def _guard_demote(self, name, to):
        """The guard's write to layer_algo: remembers the caller's own entry for that layer, if any."""
        prev = self._guard_demotions[name][1] if name in self._guard_demotions else self.layer_algo.get(name)
        self._guard_demotions[name] = (to, prev)
        self.layer_algo[name] = to

    def _drop_guard_demotions(self):
        """New weights: what the guard demoted for the OLD ones is void; the caller's own entries stay
        (an entry the guard had overridden gets the caller's value back)."""
        for k, (v, prev) in self._guard_demotions.items():
            if self.layer_algo.get(k) == v:
                if prev is None:
                    del self.layer_algo[k]
                else:
                    self.layer_algo[k] = prev
        self._guard_demotions = {}

    def _guard_probe(self, x):
        """Probe images of the automatic calibration (``wino_guard_probe``)."""
        B, _, H, W = x.shape
        if self.wino_guard_probe == 'batch':
            return x[:min(B, 8)]
        from ...utils import synthetic as syn
        side = max(H, W)
        p = torch.from_numpy(syn.synthetic_images(4, side, 7001)).to(x.device)
        return p[:, :, :H, :W].contiguous()

    @staticmethod
    def _op_channels(eng, name):
        o = next(o for o in eng['plan'].ops if o.get('name') == name)
        return o['Cin'], o['Cout']

    def _calibrate_pass(self, lib, eng, x):
        P, B = eng['plan'], x.shape[0]
        H, W = x.shape[2], x.shape[3]
        # scratch output for the direct re-run: one buffer of the largest conv output behind the arena
        scratch_off = (eng['ws_per_img'] + 7) // 8 * 8
        biggest = max(o['Ho'] * o['Wo'] * o['ups'] ** 2 * o['Cout'] for o in P.ops if o['type'] == _lib.OP_CONV)
        ws = torch.empty((scratch_off + biggest) * B, dtype=torch.float32, device=x.device)
        cnt = self._counters(eng, B, x.device)
        feat = torch.empty(B, eng['feat_dim'], dtype=torch.float32, device=x.device)
        one = (_lib.ShapyOp * 1)()
        out = []

        def run(op):
            rc = lib.shapy_hrnet_run(op, 1, _lib.ptr(eng['weights']), _lib.ptr(x), _lib.ptr(ws),
                                     scratch_off + biggest, _lib.ptr(cnt), eng['cnt_per_img'],
                                     _lib.ptr(feat), B, H, W, 0, eng['dtype'], _lib.current_stream())
            _lib.check(rc, 'shapy_hrnet_run (calibration)')
        for i, o in enumerate(P.ops):
            ctypes.memmove(one, ctypes.byref(eng['ops'][i]), ctypes.sizeof(_lib.ShapyOp))
            op = one[0]
            op.barrier_before, op.lane, op.group = 0, 0, 0
            if o['type'] == _lib.OP_CONV and o.get('wino_off', -1) >= 0:
                algo = 'winograd4' if o['tile'] & _lib.TILE_WINO4 else 'winograd'
                keep = op.tile, op.wino_off, op.out_off, op.out_ld, op.out_coff
                # never Winograd: direct kernel, unsplit (the layer's slab is sized for its Winograd form)
                op.tile = (op.tile & ~(_lib.TILE_WINO4 | (3 << 21))) | 0x2000
                op.wino_off, op.out_off, op.out_ld, op.out_coff = -1, scratch_off, o['Cout'], 0
                run(one)
                op.tile, op.wino_off, op.out_off, op.out_ld, op.out_coff = keep
                run(one)
                n = B * o['Ho'] * o['Wo']
                ref = ws[scratch_off * B:scratch_off * B + n * o['Cout']].view(n, o['Cout'])
                got = ws[o['outb'].off * B:o['outb'].off * B + n * o['out_ld']].view(n, o['out_ld'])[
                    :, o['out_coff']:o['out_coff'] + o['Cout']]
                d = (got - ref).double()
                scale = ref.double().pow(2).mean().sqrt().clamp_min(1e-30)
                out.append((o['name'], algo, float(d.pow(2).mean().sqrt() / scale),
                            float(d.abs().max() / ref.abs().max().clamp_min(1e-30))))
            else:
                run(one)
        return out

    def _forward_graph(self, lib, eng, x):
        B, _, H, W = x.shape
        key = (B, bool(self.multi_stream))
        g = eng['graphs'].get(key)
        if g is None:
            g = eng['graphs'][key] = _CapturedForward(lib, eng, B, H, W, x.device, self.multi_stream)
        return g(x)

    def drop_prefetch(self):
        """Forget a prologue stashed by ``forward(..., prefetch=t)``.  RULE for ``t``: from that call until
        ``forward(t)`` has been issued the tensor may only be written THROUGH TORCH (any in-place torch op bumps its
        version counter and the stale prologue is ignored); whoever writes it another way -- a raw-pointer kernel,
        a DLPack / ``__cuda_array_interface__`` alias, a peer copy -- calls this first (or simply does not pass
        ``prefetch``).  tests/test_gpu_parity.py::test_hrnet_prefetch_stash_rules."""
        self._prefetch.drop()

    def forward(self, x, prefetch=None):
        """``prefetch``: the NEXT batch (same shape, float32, contiguous, ready on the current stream): its stem + layer1
        run under this batch's head and the next ``forward(that tensor)`` skips them (prefetch.py; bit-identical).
        The tensor must not change behind torch's back in between: ``drop_prefetch``."""
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError(f'expected [B,3,H,W], got {tuple(x.shape)}')
        if self.training:
            # BatchNorm is folded from the running statistics: this engine only implements the
            # inference semantics (the reference would use batch statistics in train mode)
            raise RuntimeError('HighResolutionNet runs in eval mode only on the HIP path: call '
                               '.eval() (BatchNorm is folded from its running statistics)')
        _lib.require_cuda(x, 'images')
        lib = _lib.load()
        B, _, H, W = x.shape
        if H % 32 or W % 32:
            raise ValueError('HRNet input height/width must be multiples of 32')
        x = x.contiguous().float()
        use_graph = self.use_graph is True or (self.use_graph == 'auto' and B <= self.graph_max_batch)
        self._n_forward += 1
        if self.wino_guard and self.compute_dtype == 'f32' and self.conv_algo in ('winograd', 'winograd4', 'auto'):
            stale = self._calibrated_ver != self._weights_version()
            recheck = (self.wino_guard_recheck_every > 0 and not stale
                       and self._n_forward % self.wino_guard_recheck_every == 0)
            if (stale or recheck) and torch.cuda.is_current_stream_capturing():
                # the calibration synchronises with the host (per-layer error readback): it cannot
                # run inside a caller's stream capture
                if not self._capture_warned:
                    self._capture_warned = True
                    import logging
                    logging.getLogger('shapy_amd.hrnet').warning(
                        'Winograd guard skipped: forward() called under stream capture before the '
                        'weights were calibrated -- call calibrate() once outside the capture')
            elif stale:
                self._drop_guard_demotions()
                self.calibrate(self._guard_probe(x), graph=use_graph)
            elif recheck:
                self.calibrate(x[:min(B, 2)], graph=use_graph)     # only ever ADDS demotions
        eng = self._compile(H, W, x.device, graph=use_graph, B=B)
        if use_graph:
            return {'concat': self._forward_graph(lib, eng, x)}
        need = eng['ws_per_img'] * B * eng['esz']
        # one workspace (+ split-K counters) per CALLER stream: forwards issued on different streams
        # (several batches in flight) must not share activations; a second one for `prefetch=` (prefetch.py)
        sk = torch.cuda.current_stream().cuda_stream
        ent = eng['ws'].get(sk)
        if ent is None or ent['ws'][0].numel() < need or ent['B'] < B:
            ent = eng['ws'][sk] = dict(ws=[torch.empty(need, dtype=torch.uint8, device=x.device), None],
                                       cnt=[self._counters(eng, B, x.device), None], B=B, cur=0)
        feat = torch.empty(B, eng['feat_dim'], dtype=torch.float32, device=x.device)

        def run(first, n, inp, a, multi, stream):        # ops [first, first + n) in workspace a (its own counters)
            return lib.shapy_hrnet_run(ops_from(eng, first), n, _lib.ptr(eng['weights']), _lib.ptr(inp),
                                       _lib.ptr(ent['ws'][a]), eng['ws_per_img'], _lib.ptr(ent['cnt'][a]),
                                       eng['cnt_per_img'], _lib.ptr(feat), B, H, W, multi, eng['dtype'], stream)
        pf, first = self._prefetch.take(x, eng, ent, sk), 0
        if pf is not None:                    # this batch's stem + layer1 (+ stage 2) ran under the previous batch
            torch.cuda.current_stream().wait_event(pf['done'])
            ent['cur'], first = pf['arena'], pf['cut']
        early = B <= self.prefetch_before_max_batch       # small batches: deeper cut, issued before the rest
        cut = eng['cuts'][1 if early and eng['cuts'][1] else 0]
        ahead = (prefetch is not None and cut > 0 and self.multi_stream and self._prefetch.usable(prefetch, x)
                 and not torch.cuda.is_current_stream_capturing())
        if ahead:
            self._prefetch.second_workspace(ent, need, x.device)
        ev = self._prefetch.ready_event(ent) if ahead else None
        rc = 0
        if ahead and early:
            rc = self._prefetch.issue(lib, run, eng, ent, prefetch, ev, sk, need, cut)
        rc = rc or run(first, eng['n_ops'] - first, x, ent['cur'], int(self.multi_stream), _lib.current_stream())
        if not rc and ahead and not early:
            rc = self._prefetch.issue(lib, run, eng, ent, prefetch, ev, sk, need, cut)
        if rc != 0:
            del eng['ws'][sk]         # a failed forward may leave arrival counters behind: start clean
            self._prefetch.pending = None
        _lib.check(rc, 'shapy_hrnet_run')
        return {'concat': feat}


class _CapturedForward:
    """One hipGraph of the whole backbone for a fixed batch size, with the buffers it has baked
    in (they must stay alive and in place as long as the graph exists)."""

    def __init__(self, lib, eng, B, H, W, device, multi_stream):
        self.lib = lib
        self.x = torch.empty(B, 3, H, W, dtype=torch.float32, device=device)
        self.feat = torch.empty(B, eng['feat_dim'], dtype=torch.float32, device=device)
        self.ws = torch.empty(eng['ws_per_img'] * B * eng['esz'], dtype=torch.uint8, device=device)
        self.cnt = HighResolutionNet._counters(eng, B, device)
        self.weights = eng['weights']
        self.handle = ctypes.c_void_p()
        # capture happens on a private stream inside the library; make the buffers visible to it
        torch.cuda.current_stream().synchronize()
        rc = lib.shapy_hrnet_graph_create(eng['ops'], eng['n_ops'], _lib.ptr(self.weights),
                                          _lib.ptr(self.x), _lib.ptr(self.ws), eng['ws_per_img'],
                                          _lib.ptr(self.cnt), eng['cnt_per_img'], _lib.ptr(self.feat),
                                          B, H, W, int(multi_stream), eng['dtype'],
                                          ctypes.byref(self.handle))
        _lib.check(rc, 'shapy_hrnet_graph_create')

    def __call__(self, x):
        self.x.copy_(x)
        _lib.check(self.lib.shapy_hrnet_graph_launch(self.handle, _lib.current_stream()),
                   'shapy_hrnet_graph_launch')
        return self.feat.clone()          # the next replay overwrites the baked-in output

    def __del__(self):
        try:
            if self.handle:
                self.lib.shapy_hrnet_graph_destroy(self.handle)
        except Exception:                  # interpreter shutdown
            pass


def build(cfg, pretrained=True, **kwargs):
    """hrnet.py:18-26."""
    hr_net_cfg = cfg.get('hrnet')
    model = HighResolutionNet(hr_net_cfg, **kwargs)
    if pretrained:
        model.load_weights(hr_net_cfg.get('pretrained_path'))
    return model
