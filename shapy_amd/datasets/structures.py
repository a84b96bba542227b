"""Per-image target container (reference: data/structures/abstract_structure.py:5-79)."""


class Target(object):
    """Field bag with the reference's AbstractStructure accessors.  The hot path only ever asks a
    target for optional fields (``gender``, iterative_regressor.py:761-776); the demo reads
    ``fname``, ``orig_center``, ``orig_bbox_size`` (demo.py:86-87,307-315)."""

    def __init__(self, **fields):
        self.extra_fields = dict(fields)

    def add_field(self, field, field_data):
        self.extra_fields[field] = field_data

    def get_field(self, field, default=None):
        return self.extra_fields.get(field, default)

    def has_field(self, field):
        return field in self.extra_fields

    def delete_field(self, field):
        self.extra_fields.pop(field, None)

    def to(self, *args, **kwargs):
        for k, v in list(self.extra_fields.items()):
            if hasattr(v, 'to'):
                self.extra_fields[k] = v.to(*args, **kwargs)
        return self
