"""The reference-side binding, as code: `install()` swaps hyper_amd's classes into the reference's registries.

The reference (Deep-Machine-Vision/he-cka-ensembles) builds every object by a YAML `name:` looked up in a decorator
registry, a module-level dict (`hyper/diversity/model_kernels.py:15-37`, `hyper/diversity/layer_weighting.py:8-29`,
`hyper/diversity/methods.py:18-39`, `hyper/diversity/losses.py:10-30`, `hyper/layers/module.py:12-37`,
`hyper/layers/generators/base.py:15-35`, `hyper/generators/base.py:22-42`, `hyper/experiments/metrics.py:13-30`).
Those dicts are read at build time, so replacing their entries is the whole binding: the reference's own
`build_method` / `build_generator` (called by `build_from_file`, `hyper/experiments/exp.py:128,136`) then construct
hyper_amd objects, nested blocks included, and `train.py` runs unchanged.

    import hyper_amd.integration as amd
    amd.install()            # after `import hyper`, before build_from_file()

Names the reference registers and hyper_amd does not (e.g. `dropout`, `scaled_w_conv2d`, `resnet32`) keep the reference's
classes.  `tests/test_reference_boundary.py` runs this against the real reference in the build container.
"""
import importlib

# (reference module, its registry dict) -> (hyper_amd module, its registry dict)
_REGISTRIES = (
    ('hyper.diversity.model_kernels', 'AVAILABLE_KERNELS', 'hyper_amd.diversity.model_kernels', 'AVAILABLE_KERNELS'),
    ('hyper.diversity.layer_weighting', 'AVAILABLE_WEIGHTING', 'hyper_amd.diversity.layer_weighting', 'AVAILABLE_WEIGHTING'),
    ('hyper.diversity.losses', 'AVAILABLE_LOSSES', 'hyper_amd.diversity.losses', 'AVAILABLE_LOSSES'),
    ('hyper.diversity.methods', 'AVAILABLE_METHODS', 'hyper_amd.diversity.methods', 'AVAILABLE_METHODS'),
    ('hyper.layers.module', 'AVAILABLE_GEN_MODULES', 'hyper_amd.layers.module', 'AVAILABLE_GEN_MODULES'),
    ('hyper.layers.generators.base', 'AVAILABLE_LAYER_GENERATORS', 'hyper_amd.layers.generators', 'AVAILABLE_LAYER_GENERATORS'),
    ('hyper.generators.base', 'AVAILABLE_MODELS', 'hyper_amd.generators.ensemble', 'AVAILABLE_GENERATORS'),
    ('hyper.experiments.metrics', 'AVAILABLE_METRICS', 'hyper_amd.experiments.metrics', 'AVAILABLE_METRICS'),
)

_saved = None


def install(skip=()):
    """Swap every hyper_amd class into the reference registry of the same kind under the same name.

    skip: reference module names to leave alone (e.g. ('hyper.experiments.metrics',)).  Returns
    {reference module: sorted names replaced}.  Idempotent; `uninstall()` restores the reference's entries."""
    global _saved
    import hyper_amd.generators  # noqa: F401  (registers ensemble + the hypernetwork generators)
    import hyper_amd.layers  # noqa: F401
    import hyper_amd.target  # noqa: F401  (registers mlp / lenet5 / resnet18)
    saved = {} if _saved is None else _saved
    done = {}
    for ref_mod, ref_dict, amd_mod, amd_dict in _REGISTRIES:
        if ref_mod in skip:
            continue
        ref = getattr(importlib.import_module(ref_mod), ref_dict)
        amd = getattr(importlib.import_module(amd_mod), amd_dict)
        saved.setdefault(ref_mod, dict(ref))
        ref.update(amd)
        done[ref_mod] = sorted(amd)
    _saved = saved
    return done


def uninstall():
    """Put the reference's own classes back (tests build both sides in one process)."""
    global _saved
    if _saved is None:
        return
    for ref_mod, ref_dict, _, _ in _REGISTRIES:
        if ref_mod in _saved:
            ref = getattr(importlib.import_module(ref_mod), ref_dict)
            ref.clear()
            ref.update(_saved[ref_mod])
    _saved = None
