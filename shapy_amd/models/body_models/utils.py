"""KeypointTensor and small helpers (reference: models/body_models/utils.py:14-49,123-309)."""
from typing import List

import torch


def find_joint_kin_chain(joint_id: int, kinematic_tree: List) -> List:
    kin_chain = []
    curr_idx = joint_id
    while curr_idx != -1:
        kin_chain.append(curr_idx)
        curr_idx = int(kinematic_tree[curr_idx])
    return kin_chain


def to_tensor(array, dtype=torch.float32):
    if not torch.is_tensor(array):
        return torch.tensor(array, dtype=dtype)
    return array.to(dtype=dtype)


class KeypointTensor(object):
    """A keypoint wrapper carrying keypoint names / connectivity
    (models/body_models/utils.py:123-309).  Attribute access falls through to the tensor."""

    def __init__(self, data, source='smplx', keypoint_names=None, connections=None,
                 part_connections=None, part_indices=None, **kwargs):
        if isinstance(data, KeypointTensor):
            data = data._t
        self._t = torch.as_tensor(data, **kwargs)
        self._source = source
        self._keypoint_names = keypoint_names
        self._connections = connections
        self._part_indices = part_indices
        self._part_connections = part_connections

    @staticmethod
    def from_obj(tensor, obj):
        return KeypointTensor(tensor, source=obj.source, keypoint_names=obj.keypoint_names,
                              connections=obj.connections, part_indices=obj.part_indices,
                              part_connections=obj.part_connections)

    source = property(lambda self: self._source)
    keypoint_names = property(lambda self: self._keypoint_names)
    connections = property(lambda self: self._connections)
    part_indices = property(lambda self: self._part_indices)
    part_connections = property(lambda self: self._part_connections)

    def __repr__(self):
        return f'KeypointTensor:\n{self._t}'

    def __getitem__(self, key):
        return self._t[key]

    def __len__(self):
        return len(self._t)

    def __getattr__(self, name):
        # only reached when normal lookup fails: delegate to the wrapped tensor
        t = object.__getattribute__(self, '_t')
        attr = getattr(t, name)
        if callable(attr):
            def wrapped(*a, **k):
                out = attr(*a, **k)
                if torch.is_tensor(out) and out.shape == t.shape:
                    return KeypointTensor.from_obj(out, self)
                return out
            return wrapped
        return attr

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        src = [a for a in list(args) + list(kwargs.values()) if isinstance(a, KeypointTensor)]
        unwrap = lambda a: a._t if isinstance(a, KeypointTensor) else a
        args = [unwrap(a) if not isinstance(a, (list, tuple)) else type(a)(unwrap(x) for x in a)
                for a in args]
        kwargs = {k: unwrap(v) for k, v in kwargs.items()}
        ret = func(*args, **kwargs)
        if torch.is_tensor(ret) and src and ret.shape == src[0]._t.shape:
            return KeypointTensor.from_obj(ret, src[0])
        return ret
