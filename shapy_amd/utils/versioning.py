"""Staleness key for caches derived from a module's weights (folded / repacked device blobs,
hipGraphs).

What the key sees (``_weights_version``):
  * every in-place edit that goes through the tensor itself -- ``p.mul_()``, ``p.copy_()`` under
    ``no_grad``, ``init_weights()``, an optimizer step, ``load_state_dict`` -- bumps the tensor's
    autograd version counter, and the key sums the counters of all parameters and buffers;
  * REPLACING a parameter or buffer object (``m.weight = nn.Parameter(...)``,
    ``register_buffer``, ``del m.bias``): torch's global registration hooks bump a process-wide
    epoch; when it has moved the tensor list is re-walked and the ids of the live tensors are part
    of the key.
What it canNOT see: edits through ``p.data`` (``p.data.copy_()``, ``p.data.fill_()``) --
``.data`` returns a tensor with its OWN version counter, so ``p._version`` stays where it was
(checked with torch 2.10) -- and writes through other views of the storage made outside autograd
(``numpy`` views, ``torch.as_strided`` on ``.data``).  After such an edit call ``invalidate()``
on the module (HighResolutionNet / SMPLX / IterativeRegression all have it); a checksum of 438 MB
of weights per forward is not an option on the hot path.

~0.2 ms per call for HRNet-W48's 1,967 tensors (the list is cached; it is dropped by ``_apply`` /
``invalidate`` / pickling and whenever the registration epoch moves)."""
import torch.nn.modules.module as _tm

_EPOCH = [0]


def _bump(*_a, **_k):
    _EPOCH[0] += 1
    return None


# process-wide: any module registering a parameter / buffer / submodule anywhere moves the epoch
# (cheap: these hooks only run on attribute assignment, never in a forward)
_tm.register_module_parameter_registration_hook(_bump)
_tm.register_module_buffer_registration_hook(_bump)
_tm.register_module_module_registration_hook(_bump)


class VersionedWeights:
    """Mixin for ``nn.Module`` subclasses that cache weight-derived state."""

    def _weights_version(self):
        d = self.__dict__
        ts = d.get('_ver_tensors')
        if ts is None or d.get('_ver_epoch') != _EPOCH[0]:
            ts = list(self.parameters()) + list(self.buffers())
            d['_ver_tensors'] = ts
            d['_ver_epoch'] = _EPOCH[0]
            d['_ver_ids'] = hash(tuple(id(t) for t in ts))
        return (sum(t._version for t in ts) + 1_000_003 * len(ts), d['_ver_ids'])

    def _drop_version_cache(self):
        self.__dict__['_ver_tensors'] = None
