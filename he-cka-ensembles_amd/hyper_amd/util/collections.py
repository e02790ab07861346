"""Nested-dict plumbing of the reference (hyper/util/collections.py: DefaultOrderedDict :6, flatten_keys :75,
unflatten_keys :113), restated for Python >= 3.10 (the reference imports `collections.Callable`)."""
from collections import OrderedDict


class DefaultOrderedDict(OrderedDict):
    """OrderedDict whose missing keys are created from `default_factory` (itself by default)."""

    def __init__(self, default_factory=None, *a, **kw):
        if default_factory is None:
            default_factory = DefaultOrderedDict
        if not callable(default_factory):
            raise TypeError('first argument must be callable')
        super().__init__(*a, **kw)
        self.default_factory = default_factory

    def __missing__(self, key):
        self[key] = value = self.default_factory()
        return value

    def set_subitems(self, keys, value, items=None):
        node = self if items is None else items
        for k in keys[:-1]:
            node = node[k]
        if keys:
            node[keys[-1]] = value


def flatten_keys(obj, prefix='', include_empty=False):
    """{'a': {'b': 1}, 'c': 2} -> {'a.b': 1, 'c': 2}, order preserved."""
    if obj is None:
        return None
    if not isinstance(obj, dict):
        return obj
    flat, empty = OrderedDict(), OrderedDict()

    def walk(node, pre):
        for key, item in node.items():
            name = '%s.%s' % (pre, key) if pre else key
            if isinstance(item, dict):
                if len(item) == 0:
                    empty[name] = item
                walk(item, name)
            else:
                flat[name] = item

    walk(obj, prefix)
    return (flat, empty) if include_empty else flat


def unflatten_keys(flat_dict):
    out = OrderedDict()
    for compound, value in flat_dict.items():
        node = out
        parts = compound.split('.')
        for k in parts[:-1]:
            node = node.setdefault(k, OrderedDict())
        node[parts[-1]] = value
    return out
