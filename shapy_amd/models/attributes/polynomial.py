"""B2A: betas -> attribute ratings (reference: attributes/attributes/attributes_betas/
polynomial.py:21-140, b2a.py:25-118; called from iterative_regressor.py:761-776).

``Polynomial`` keeps the reference's parameter/buffer names (``linear.weight``,
``indices_000``, ``indices_001``) so a ``B2A`` Lightning checkpoint's ``state_dict``
(``b2a.linear.weight`` ...) loads without pytorch_lightning."""
from itertools import chain, combinations_with_replacement

import torch
import torch.nn as nn

from ... import _lib


class Polynomial(nn.Module):
    def __init__(self, input_dim, output_dim, degree=2, alpha=0.0):
        super().__init__()
        if degree != 2:
            raise NotImplementedError('the SHAPY B2A checkpoints use degree 2')
        self.input_dim, self.output_dim, self.degree, self.alpha = input_dim, output_dim, degree, alpha
        combos = list(chain.from_iterable(
            combinations_with_replacement(range(input_dim), i) for i in range(1, degree + 1)))
        self.coeff_size = len(combos)
        self.linear = nn.Linear(len(combos), output_dim)
        for ii in range(degree):
            idx = torch.tensor([c for c in combos if len(c) == ii + 1], dtype=torch.long)
            self.register_buffer(f'indices_{ii:03d}', idx)

    def forward(self, x):
        _lib.require_cuda(x, 'betas')
        lib = _lib.load()
        x = x.contiguous().float()
        B = x.shape[0]
        out = torch.empty(B, self.output_dim, dtype=torch.float32, device=x.device)
        _lib.check(lib.shapy_b2a_polynomial_f32(
            _lib.ptr(x), _lib.ptr(self.linear.weight.detach().contiguous()),
            _lib.ptr(self.linear.bias.detach().contiguous()), _lib.ptr(out), B, self.input_dim,
            self.output_dim, _lib.current_stream()), 'shapy_b2a_polynomial_f32')
        return out


class B2A(nn.Module):
    """Inference-only stand-in for the LightningModule (b2a.py:25-118): ``self.b2a`` is the
    network, ``forward(betas)`` its prediction."""

    def __init__(self, num_shape_comps=10, num_outputs=15, degree=2):
        super().__init__()
        self.b2a = Polynomial(num_shape_comps, num_outputs, degree=degree)

    def forward(self, x):
        return self.b2a(x)

    @staticmethod
    def load_from_checkpoint(path, map_location='cpu', **kwargs):
        """Reads ``state_dict`` of a Lightning checkpoint.  ``hyper_parameters`` (an OmegaConf
        object in the released checkpoints) is not needed: the shapes come from the weights."""
        import pickle

        class _Tolerant(pickle.Unpickler):
            def find_class(self, module, name):
                try:
                    return super().find_class(module, name)
                except Exception:
                    return type(name, (), {'__init__': lambda self, *a, **k: None,
                                           '__setstate__': lambda self, s: None})
        pm = type('pm', (), {'Unpickler': _Tolerant, 'load': pickle.load,
                             '__name__': 'tolerant_pickle'})
        try:
            ckpt = torch.load(path, map_location=map_location, weights_only=False)
        except Exception:
            ckpt = torch.load(path, map_location=map_location, weights_only=False, pickle_module=pm)
        sd = ckpt.get('state_dict', ckpt.get('model', ckpt))
        w = sd['b2a.linear.weight'] if 'b2a.linear.weight' in sd else sd['linear.weight']
        na, nm = w.shape
        nb = int(round((-3 + (9 + 8 * nm) ** 0.5) / 2))       # nm = nb + nb (nb + 1) / 2
        obj = B2A(nb, na)
        if 'b2a.linear.weight' in sd:
            obj.load_state_dict({k: v for k, v in sd.items() if k.startswith('b2a.')}, strict=False)
        else:
            obj.b2a.load_state_dict(sd, strict=False)
        return obj.eval()
