"""Staleness key for caches derived from a module's weights (folded / repacked device blobs,
hipGraphs): the sum of the autograd version counters of every parameter and buffer.  Any
in-place edit (``p.data.copy_()``, ``init_weights()``, an optimizer step, ``load_state_dict``)
bumps a counter, so a cache keyed on it is rebuilt instead of silently serving results of the
old weights.  ~0.2 ms per call for HRNet-W48's 1,967 tensors (the list is cached; it is dropped
by ``_apply`` / ``invalidate`` / pickling, the ways tensors get *replaced*)."""


class VersionedWeights:
    """Mixin for ``nn.Module`` subclasses that cache weight-derived state."""

    def _weights_version(self):
        ts = self.__dict__.get('_ver_tensors')
        if ts is None:
            ts = list(self.parameters()) + list(self.buffers())
            self.__dict__['_ver_tensors'] = ts
        return sum(t._version for t in ts) + 1_000_003 * len(ts)

    def _drop_version_cache(self):
        self.__dict__['_ver_tensors'] = None
