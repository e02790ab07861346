"""The drop-in boundary executed against the REAL reference (build container only: skipped where /root/reference is
absent - nothing of the reference travels, this file holds no reference text).

`hyper_amd.integration.install()` swaps hyper_amd's classes into the reference's registries.
For the `method:` and `hyper:` block of every shipped `configs/**/{ens,ens+ood,svgd,fsvgd,hyper}/*.yml` this test builds
the objects twice through the reference's OWN `build_method` / `build_generator` - once with the reference's classes,
once after the swap - and asserts that the swapped objects are hyper_amd's and that their `state_dict()` key sets and
shapes equal those of the un-swapped reference objects (checkpoints interchange).  It prints every YAML it built and any
it could not build, with the reason.  The YAML loading restates the `base:` merge and the `__latent_size__` /
`__code_size__` substitution of `hyper/experiments/exp.py:115-125,188-228` (that module itself needs torchvision)."""
import collections
import collections.abc
import copy
import glob
import os
import sys
import types

import pytest
import yaml

REF = '/root/reference'
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'hyper')), reason='the reference is not on this machine')

GROUPS = ('ens', 'ens+ood', 'svgd', 'fsvgd', 'hyper')


def _import_reference():
    """The two harness-side shims of SURVEY.md 8c (collections.Callable, stubs of absent third-party packages)."""
    collections.Callable = collections.abc.Callable
    for name, attrs in {
        'torchmetrics': ['Accuracy'],
        'x_transformers': ['TransformerWrapper', 'Decoder'],
        'x_transformers.x_transformers': ['AttentionLayers', 'ScaledSinusoidalEmbedding', 'AbsolutePositionalEmbedding'],
        'positional_encodings': [],
        'positional_encodings.torch_encodings': ['PositionalEncoding1D', 'Summer'],
    }.items():
        mod = types.ModuleType(name)
        for a in attrs:
            setattr(mod, a, type(a, (), {}))
        sys.modules.setdefault(name, mod)
    if REF not in sys.path:
        sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    import hyper.diversity.methods as methods
    import hyper.generators.base as gbase
    import hyper.generators.ensemble  # noqa: F401
    import hyper.generators.mlp  # noqa: F401
    import hyper.target.lenet  # noqa: F401
    import hyper.target.mlp  # noqa: F401
    import hyper.target.resnet  # noqa: F401
    import hyper.experiments.metrics  # noqa: F401
    return methods.build_method, gbase.build_generator


def _merge(a, b):
    if isinstance(a, dict) and isinstance(b, dict):
        for k, v in b.items():
            a[k] = _merge(a[k], v) if k in a else v
        return a
    return b


def _load(path):
    with open(path) as f:
        cfg = yaml.safe_load(f)
    if 'base' in cfg:
        base = _load(os.path.join(os.path.dirname(path), cfg.pop('base')))
        cfg = _merge(base, cfg)
    return cfg


def _substitute(node, match, value):
    for k in list(node.keys()):
        if node[k] == match:
            node[k] = value
        elif isinstance(node[k], dict):
            _substitute(node[k], match, value)


def _shapes(module):
    return {k: tuple(v.shape) for k, v in module.state_dict().items()}


def _is_amd(obj):
    return type(obj).__module__.startswith('hyper_amd.')


def _yamls():
    out = []
    for g in GROUPS:
        out += glob.glob(os.path.join(REF, 'configs', '**', g, '*.yml'), recursive=True)
    return sorted(set(out))


def test_swapped_registries_build_every_shipped_yaml():
    import torch
    from hyper_amd import integration
    build_method, build_generator = _import_reference()
    built, failed = [], []
    paths = _yamls()
    assert len(paths) >= 25
    for path in paths:
        rel = os.path.relpath(path, os.path.join(REF, 'configs'))
        cfg = _load(path)
        if 'latent_size' in cfg:
            _substitute(cfg, '__latent_size__', int(cfg['latent_size']))
        if 'code_size' in cfg:
            _substitute(cfg, '__code_size__', int(cfg['code_size']))
        hyper_cfg = copy.deepcopy(cfg['hyper'])
        hyper_cfg.pop('weights', None)
        for block, build, conf in (('method', build_method, cfg['method']), ('hyper', build_generator, hyper_cfg)):
            integration.uninstall()
            try:
                torch.manual_seed(0)
                ref_obj = build(copy.deepcopy(conf))
            except Exception as e:  # the reference itself cannot build it here: report, not a failure of the swap
                failed.append((rel, block, 'reference: %s: %s' % (type(e).__name__, e)))
                continue
            assert not _is_amd(ref_obj)
            ref_shapes = _shapes(ref_obj)
            del ref_obj
            integration.install()
            try:
                torch.manual_seed(0)
                amd_obj = build(copy.deepcopy(conf))
            finally:
                integration.uninstall()
            assert _is_amd(amd_obj), (rel, block, type(amd_obj))
            # nested blocks come from the swapped registries too
            for name, sub in amd_obj.named_modules():
                mod = type(sub).__module__
                assert not mod.startswith('hyper.'), (rel, block, name, mod)
            amd_shapes = _shapes(amd_obj)
            assert set(amd_shapes) == set(ref_shapes), (rel, block, sorted(set(amd_shapes) ^ set(ref_shapes))[:8])
            assert amd_shapes == ref_shapes, (rel, block)
            built.append((rel, block, len(ref_shapes)))
            del amd_obj
    print('\nbuilt through the reference\'s build_method / build_generator after hyper_amd.integration.install():')
    for rel, block, n in built:
        print('  %-40s %-7s %5d state-dict entries, keys and shapes equal' % (rel, block, n))
    print('could not be built:')
    for rel, block, why in failed:
        print('  %-40s %-7s %s' % (rel, block, why))
    if not failed:
        print('  (none)')
    assert len(built) >= 2 * len(paths) - len(failed)
    assert not [f for f in failed if not f[2].startswith('reference:')]


def test_install_is_idempotent_and_reversible():
    from hyper_amd import integration
    _import_reference()
    import hyper.diversity.model_kernels as ref_mk
    before = dict(ref_mk.AVAILABLE_KERNELS)
    done = integration.install()
    assert 'he' in done['hyper.diversity.model_kernels']
    assert ref_mk.AVAILABLE_KERNELS['he'].__module__ == 'hyper_amd.diversity.model_kernels'
    integration.install()
    integration.uninstall()
    assert ref_mk.AVAILABLE_KERNELS == before
