"""Dependency-free configuration node.

The reference builds its configuration with OmegaConf structured dataclasses, YAML files
and a dot-list CLI (``regressor/human_shape/config/defaults.py:5,112``,
``config/cmd_parser.py:37-42``).  OmegaConf is not available on the target image, and the
model code only needs three behaviours from a config object: attribute access,
``.get(key, default)`` and ``**cfg`` splatting (e.g. ``iterative_regressor.py:52,592``,
``networks.py:745-746``, ``body_models/build.py:21-25``).  ``ConfigNode`` provides exactly
those on top of ``dict`` plus the three merge sources the reference's entry points use:
defaults (+) YAML file(s) (+) ``key.sub=value`` dot-list.
"""
import copy

import yaml


class ConfigNode(dict):
    def __init__(self, init=None, **kwargs):
        super().__init__()
        init = dict(init or {}, **kwargs)
        for k, v in init.items():
            self[k] = self._wrap(v)

    @classmethod
    def _wrap(cls, v):
        if isinstance(v, ConfigNode):
            return v
        if isinstance(v, dict):
            return cls(v)
        if isinstance(v, (list, tuple)):
            return [cls._wrap(x) for x in v]
        return v

    # attribute access ------------------------------------------------------------------
    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError:
            raise AttributeError(f'Missing key {key}') from None

    def __setattr__(self, key, value):
        self[key] = self._wrap(value)

    def __delattr__(self, key):
        del self[key]

    def __deepcopy__(self, memo):
        return ConfigNode({k: copy.deepcopy(v, memo) for k, v in self.items()})

    def copy(self):
        return copy.deepcopy(self)

    # merging ---------------------------------------------------------------------------
    def merge_with(self, other):
        """Recursive in-place merge, ``other`` wins (OmegaConf.merge_with semantics for
        dict nodes; lists are replaced)."""
        for k, v in other.items():
            if isinstance(v, dict) and isinstance(self.get(k), dict):
                self[k].merge_with(v)
            else:
                self[k] = self._wrap(copy.deepcopy(v))
        return self

    def to_dict(self):
        def unwrap(v):
            if isinstance(v, dict):
                return {k: unwrap(x) for k, x in v.items()}
            if isinstance(v, list):
                return [unwrap(x) for x in v]
            return v
        return unwrap(self)

    @classmethod
    def load(cls, path):
        with open(path, 'r') as f:
            data = yaml.safe_load(f) or {}
        return cls(data)

    @classmethod
    def from_dotlist(cls, items):
        """``['network.smplx.num_stages=2', 'output_folder=out']`` -> nested node.
        Values are parsed as YAML scalars/lists, like ``OmegaConf.from_cli``."""
        root = cls()
        for item in items:
            if '=' not in item:
                raise ValueError(f'Expected key=value, got: {item}')
            key, value = item.split('=', 1)
            node = root
            parts = key.split('.')
            for p in parts[:-1]:
                if p not in node or not isinstance(node[p], dict):
                    node[p] = cls()
                node = node[p]
            node[parts[-1]] = cls._wrap(yaml.safe_load(value) if value != '' else '')
        return root
