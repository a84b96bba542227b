"""MLP + IterativeRegression on gfx950.

Drop-in for ``MLP`` / ``IterativeRegression`` / ``build_regressor`` of
regressor/human_shape/models/common/networks.py:308-400,492-592,727-762 (same constructor
arguments, same state_dict keys: ``module.layer_000.0.weight``, ``module.output_layer.weight``,
``mean_param``).

The SHAPY_A regressor has ``activation.type: none``, ``normalization.type: none`` and dropout
(identity in eval mode): three Linear layers chained without a non-linearity are ONE affine
map.  It is collapsed on the host in float64 (W = W3 W2 W1 split into a feature part Wf
[145,2048] and a parameter part Wp [145,145]) and all ``num_stages`` iterations run in one
kernel launch (csrc/body.hip: regressor_affine_kernel): 35x fewer MACs than evaluating the
three layers three times, and no intermediate activations in HBM.
"""
import torch
import torch.nn as nn

from ... import _lib
from ...utils.versioning import VersionedWeights


def build_activation(activ_cfg):
    if activ_cfg is None:
        return None
    activ_type = activ_cfg.get('type', 'relu')
    if activ_type == 'none':
        return None
    if activ_type == 'relu':
        return nn.ReLU(inplace=activ_cfg.get('inplace', False))
    if activ_type == 'leaky-relu':
        return nn.LeakyReLU(inplace=activ_cfg.get('inplace', False),
                            **activ_cfg.get('leaky_relu', {}))
    raise ValueError(f'Unknown activation type: {activ_type}')


def build_norm_layer(input_dim, norm_cfg=None, dim=1):
    if norm_cfg is None:
        return None
    norm_type = norm_cfg.get('type', 'bn')
    if norm_type in ('none', 'None'):
        return None
    if norm_type in ('bn', 'batch-norm'):
        return nn.BatchNorm1d(input_dim, **{k: v for k, v in norm_cfg.get('batch_norm', {}).items()})
    raise ValueError(f'Unknown normalization type: {norm_type}')


class MLP(nn.Module):
    """Parameter container with the reference's layout (networks.py:308-382)."""

    def __init__(self, input_dim, output_dim, layers=None, activation=None, normalization=None,
                 dropout=0.0, gain=0.01, preactivated=False, flatten=True, **kwargs):
        super().__init__()
        layers = list(layers or [])
        activation = activation or {}
        normalization = normalization or {}
        self.flatten = flatten
        self.input_dim, self.output_dim = input_dim, output_dim
        self.num_layers = len(layers)
        self.blocks = []
        self.is_affine = True
        curr = input_dim
        for layer_idx, layer_dim in enumerate(layers):
            activ = build_activation(activation)
            norm_layer = build_norm_layer(layer_dim, norm_cfg=normalization, dim=1)
            if activ is not None or norm_layer is not None:
                self.is_affine = False
            linear = nn.Linear(curr, layer_dim, bias=norm_layer is None)
            curr = layer_dim
            layer = []
            if preactivated:
                layer += [l for l in (norm_layer, activ) if l is not None] + [linear]
            else:
                layer += [linear] + [l for l in (activ, norm_layer) if l is not None]
            if dropout > 0.0:
                layer.append(nn.Dropout(dropout))
            block = nn.Sequential(*layer)
            self.add_module('layer_{:03d}'.format(layer_idx), block)
            self.blocks.append(block)
        self.output_layer = nn.Linear(curr, output_dim)
        # init_weights(..., gain, init_type='xavier', distr='uniform') (networks.py:378-382)
        nn.init.xavier_uniform_(self.output_layer.weight, gain=gain)
        nn.init.zeros_(self.output_layer.bias)

    def linears(self):
        out = [m for blk in self.blocks for m in blk if isinstance(m, nn.Linear)]
        return out + [self.output_layer]

    def collapse(self):
        """(W, b) float64 of the whole affine chain."""
        if not self.is_affine:
            raise NotImplementedError(
                'MLP with activation/normalisation layers is not on the SHAPY_A path '
                '(configs/b2a_expose_hrnet_demo.yaml:200-207: both are "none")')
        W, b = None, None
        for lin in self.linears():
            w = lin.weight.detach().double().cpu()
            bb = lin.bias.detach().double().cpu() if lin.bias is not None else torch.zeros(
                w.shape[0], dtype=torch.float64)
            if W is None:
                W, b = w, bb
            else:
                W, b = w @ W, w @ b + bb
        return W, b

    def forward(self, module_input, **kwargs):
        raise RuntimeError('MLP is evaluated inside IterativeRegression on the HIP path')


class IterativeRegression(VersionedWeights, nn.Module):
    def __init__(self, module, mean_param, num_stages=1, append_params=True, learn_mean=False,
                 detach_mean=False, dim=1, **kwargs):
        super().__init__()
        self.module = module
        self._num_stages = num_stages
        self.dim = dim
        self.append_params = append_params
        self.detach_mean = detach_mean
        self.learn_mean = learn_mean
        if not append_params:
            raise NotImplementedError('append_params=False is not used by SHAPY')
        if learn_mean:
            self.register_parameter('mean_param', nn.Parameter(mean_param, requires_grad=True))
        else:
            self.register_buffer('mean_param', mean_param)
        self._packed = {}
        self.register_load_state_dict_post_hook(lambda m, k: m.invalidate())

    num_stages = property(lambda self: self._num_stages)

    def get_mean(self):
        return self.mean_param.clone()

    def invalidate(self):
        self._packed = {}
        self._drop_version_cache()

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._packed = {}
        self._drop_version_cache()
        return out

    def __getstate__(self):
        st = self.__dict__.copy()
        st['_packed'] = {}
        st['_ver_tensors'] = None
        return st

    def _pack(self, device, F):
        ver = self._weights_version()          # in-place edits of the Linear layers / mean
        if ver != self.__dict__.get('_packed_ver'):
            self._packed = {}
            self.__dict__['_packed_ver'] = ver
        key = (str(device), F)
        pk = self._packed.get(key)
        if pk is None:
            W, b = self.module.collapse()
            P = W.shape[0]
            if W.shape[1] != F + P:
                raise ValueError(f'regressor expects {W.shape[1] - P} features, got {F}')
            Wf64, Wp64 = W[:, :F], W[:, F:]
            # the stages collapsed too (cond is None: p_0 = mean for every body):
            #   p_s = A^s mean + (sum_{k<s} A^k) (Wf feat + b),  A = I + Wp      (float64)
            A = torch.eye(P, dtype=torch.float64) + Wp64
            mean64 = self.mean_param.detach().double().cpu().reshape(-1)[:P]
            Ms, cs, M, Ak = [], [], torch.zeros(P, P, dtype=torch.float64), torch.eye(P, dtype=torch.float64)
            for _ in range(self._num_stages):
                M = M + Ak                     # sum_{k<s} A^k
                Ak = A @ Ak                    # A^s
                Ms.append(M.clone())
                cs.append(Ak @ mean64)
            # block S (one more "stage"): the stage-0 delta p_1 - mean that forward() returns
            W_all = torch.cat([m @ Wf64 for m in Ms] + [Ms[0] @ Wf64], dim=0)
            b_all = torch.cat([c + m @ b for c, m in zip(cs, Ms)] + [cs[0] + Ms[0] @ b - mean64], dim=0)
            pk = dict(Wf=Wf64.float().contiguous().to(device),
                      Wp=Wp64.float().contiguous().to(device),
                      b=b.float().contiguous().to(device), P=P,
                      W_all=W_all.float().contiguous().to(device),
                      b_all=b_all.float().contiguous().to(device))
            self._packed[key] = pk
        return pk

    def forward(self, features, cond=None, **kwargs):
        """-> (parameters: list of [B,P] per stage, deltas: [stage-0 delta])
        (networks.py:536-592)."""
        _lib.require_cuda(features, 'features')
        if self.training and any(isinstance(m, nn.Dropout) and m.p > 0 for m in self.module.modules()):
            raise RuntimeError('IterativeRegression runs in eval mode only on the HIP path: the '
                               'collapsed affine map has no dropout (call .eval())')
        lib = _lib.load()
        B, F = features.shape
        if F % 4:
            raise ValueError('feature dimension must be a multiple of 4')
        features = features.contiguous().float()
        pk = self._pack(features.device, F)
        P = pk['P']
        if cond is None:
            init, per_body = self.mean_param.reshape(-1).contiguous().float(), 0
        else:
            init, per_body = cond.reshape(B, -1)[:, :P].contiguous().float(), 1
        S = self._num_stages
        if cond is None:                      # one launch: all stages are affine in the features
            out = torch.empty(S + 1, B, P, dtype=torch.float32, device=features.device)
            _lib.check(lib.shapy_regressor_collapsed_f32(
                _lib.ptr(features), _lib.ptr(pk['W_all']), _lib.ptr(pk['b_all']), _lib.ptr(out),
                B, F, P, S + 1, _lib.current_stream()), 'shapy_regressor_collapsed_f32')
            self.last_output = out[:S]        # [S,B,P]: the stages are views of one tensor
            return [out[s] for s in range(S)], [out[S]]
        out = torch.empty(S, B, P, dtype=torch.float32, device=features.device)
        self.last_output = out
        _lib.check(lib.shapy_regressor_affine_f32(
            _lib.ptr(features), _lib.ptr(pk['Wf']), _lib.ptr(pk['Wp']), _lib.ptr(pk['b']),
            _lib.ptr(init), _lib.ptr(out), B, F, P, self._num_stages, per_body,
            _lib.current_stream()), 'shapy_regressor_affine_f32')
        parameters = [out[s] for s in range(self._num_stages)]
        deltas = [parameters[0] - (init if per_body else init.unsqueeze(0))]
        return parameters, deltas


def build_regressor(network_cfg, input_dim, output_dim, param_mean):
    """networks.py:727-762."""
    regressor_type = network_cfg.get('type', 'mlp')
    if regressor_type == 'iterative-mlp':
        mlp_cfg = network_cfg.get('mlp', {})
        append_params = network_cfg.get('append_params', True)
        regressor = MLP(input_dim + append_params * param_mean.numel(), output_dim, **mlp_cfg)
        it = IterativeRegression(regressor, param_mean, **network_cfg)
        return it, it.num_stages
    if regressor_type in ('mlp', 'iterative-rnn'):
        raise NotImplementedError(f'regressor type {regressor_type} is not used by SHAPY_A')
    raise ValueError(f'Unknown regressor type: {regressor_type}')
