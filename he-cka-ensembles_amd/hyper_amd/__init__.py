"""hyper_amd — MI355X-native drop-in for the diversity-loss hot path of he-cka-ensembles.

Mirrors the reference's plugin API for that path only (`hyper.diversity.*`: model-kernel and
layer-weighting registries, same class names, ctor kwargs, forward signatures and state-dict
keys) on top of libhecka's hand-written gfx950 kernels.  See DESIGN.md / INTEGRATION.md."""
from . import _lib  # noqa: F401

__all__ = ['_lib']
