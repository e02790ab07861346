import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'he-cka-ensembles_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def golden_cases(tag):
    """Model-kernel fixtures (the train-step fixtures `toy_*` / `lenet_*` / `resnet_*` / `hyper_*` and the layer fixtures `conv_*` have
    their own tests)."""
    return sorted(f[:-len('_%s.npz' % tag)] for f in os.listdir(GOLDEN)
                  if f.endswith('_%s.npz' % tag) and not f.startswith(('toy_', 'lenet_', 'resnet_', 'svgd_', 'conv_', 'hyper_', 'uhsic_', 'wsvgd_', 'fsvgd_', 'learned_')))


def load_golden(name, tag):
    import json
    import numpy as np
    z = np.load(os.path.join(GOLDEN, '%s_%s.npz' % (name, tag)), allow_pickle=False)
    rec = {k: z[k] for k in z.files}
    rec['config'] = json.loads(str(rec['config']))
    rec['L'] = sum(1 for k in rec if k.startswith('x') and k[1:].isdigit())
    return rec
