"""BODY_HEAD_REGISTRY (reference: models/body_heads/registry.py:1-7; fvcore-free)."""


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


BODY_HEAD_REGISTRY = Registry('BODY_HEAD_REGISTRY')
