"""Generate golden vectors for the HE-CKA hot path FROM THE REAL REFERENCE.

Run only in the build container (needs /root/reference, which never travels):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/make_golden.py

It imports the reference's own `hyper.diversity.model_kernels` (with the two harness-side
shims SURVEY.md 8c lists: `collections.Callable` for Python >= 3.10 and stub modules for
absent third-party packages), builds each model kernel with the reference's
`build_model_kernel` from a YAML-shaped dict, runs forward + autograd backward on seeded
inputs and stores inputs and outputs as arrays in `tests/golden/<case>.npz`.  The fixtures
hold data only (inputs, expected outputs, the config dict as JSON) - no reference source.
"""
import collections
import collections.abc
import json
import os
import sys
import types

import numpy as np
import torch

REF = '/root/reference'
OUT = os.path.dirname(os.path.abspath(__file__))


def _import_reference():
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
    sys.path.insert(0, REF)
    import hyper.diversity.model_kernels as mk
    return mk


def he(kernel='linear', detach_diag=True, params=None, **kw):
    fk = {'name': 'feature', 'kernel': kernel, 'detach_diag': detach_diag}
    if params:
        fk['params'] = params
    return dict(name='he', feature_kernel=fk, **kw)


def cka(kernel='linear', detach_diag=True, params=None, **kw):
    fk = {'name': 'feature', 'kernel': kernel, 'detach_diag': detach_diag}
    if params:
        fk['params'] = params
    return dict(name='cka', feature_kernel=fk, **kw)


def layers(layer_kernel, weighting, reduction='mean'):
    return {'name': 'model_layers', 'layer_kernel': layer_kernel, 'layer_weighting': weighting, 'reduction': reduction}


MNIST_W = {'name': 'start_end_linear', 'first_layer': 0.387, 'middle_start': 0.746, 'middle_increase': 1.267, 'last_layer': 3.63}
UNI = {'name': 'uniform'}

# name -> (config, M, B, feature shapes per layer, input style)
CASES = {
    'he_lin_mnist': (layers(he(eps=0.0025, arc_eps=0.031), MNIST_W), 5, 12, [(6, 3, 3), (16,), (12,), (8,), (10,)], 'randn'),
    'he_lin_exp_cifar': (layers(he(eps=0.001, arc_eps=0.001, use_exp=True),
                                {'name': 'list', 'weights': [0.0, 0.15, 0.25, 0.0001], 'normalize': False}),
                         4, 9, [(4, 2, 2), (7,), (33,), (10,)], 'relu'),
    'he_lin_s0': (layers(he(he_s=0.0, eps=1e-3, arc_eps=1e-2), UNI), 4, 8, [(5,), (24,)], 'randn'),
    'he_lin_s1': (layers(he(he_s=1.0, eps=1e-3, arc_eps=1e-2), UNI), 4, 8, [(5,), (24,)], 'randn'),
    'he_lin_s3_abs': (layers(he(he_s=3.0, eps=1e-5, arc_eps=3e-3, abs_vals=True), {'name': 'linear', 'start_weight': 1.0, 'weight_increase': 0.5}),
                      6, 16, [(24,), (7,)], 'randn'),
    'he_lin_detach_right_sum': (layers(he(eps=1e-3, arc_eps=1e-2, detach_right=True), {'name': 'exponential', 'start_weight': 0.5, 'weight_increase': 2.0}, 'sum'),
                                5, 10, [(9,), (17,)], 'randn'),
    'he_lin_nodetach': (layers(he(detach_diag=False, eps=1e-4, arc_eps=1e-3), UNI), 4, 8, [(11,)], 'randn'),
    'he_lin_none': (layers(he(eps=0.001, arc_eps=0.001, detach_right=True, reduction='none'), UNI, 'none'), 5, 8, [(6,), (20,)], 'randn'),
    'he_lin_correlated': (layers(he(eps=0.0025, arc_eps=0.031), UNI), 6, 16, [(32,), (64,)], 'correlated'),
    'he_lin_odd': (layers(he(eps=0.0025, arc_eps=0.031), UNI), 3, 37, [(19,), (130,)], 'randn'),
    'he_lin_bench_small': (layers(he(eps=0.0025, arc_eps=0.031), UNI), 6, 64, [(32,), (72,)], 'randn'),
    'cka_lin': (layers(cka(eps=1e-6), UNI), 5, 10, [(9,), (17,)], 'randn'),
    'cka_lin_none': (layers(cka(eps=1e-6, detach_right=True, reduction='none'), {'name': 'list', 'weights': [0.4, 0.6], 'normalize': False}, 'none'),
                     5, 10, [(9,), (17,)], 'randn'),
    'he_cossim': (layers(he('cossim', eps=0.0025, arc_eps=0.031), UNI), 5, 12, [(16,), (10,)], 'randn'),
    'he_rbf_toy': (layers(he('rbf', he_s=5.0, eps=0.000005, arc_eps=0.00305, params={'param': 5.025}),
                          {'name': 'list', 'weights': [0.05, 0.35, 1.0, 0.85]}), 6, 10, [(30,), (40,), (30,), (4,)], 'randn'),
    'cka_rbf': (layers(cka('rbf', params={'param': 1.5}), UNI), 4, 9, [(7,)], 'randn'),
}


def make_inputs(style, M, B, shapes, gen):
    xs = []
    for shp in shapes:
        x = torch.randn(M, B, *shp, generator=gen, dtype=torch.float64)
        if style == 'relu':
            x = torch.relu(x + 0.3)
        elif style == 'correlated':
            x = 0.7 * torch.randn(1, B, *shp, generator=gen, dtype=torch.float64) + 0.3 * x
        xs.append(x)
    return xs


def run_case(mk, name, cfg, M, B, shapes, style, dtype):
    gen = torch.Generator().manual_seed(sum(map(ord, name)))
    xs64 = make_inputs(style, M, B, shapes, gen)
    # the reference mutates a module-global default dict (model_kernels.py:78-80): reset it per case
    from hyper.diversity import kernels as rk
    rk.KERN_PARAMS['rbf'] = {'param': 1.0}
    rk.KERN_PARAMS['cossim'] = {'eps': 1e-6}
    kern = mk.build_model_kernel(cfg).to(dtype)
    xs = [x.to(dtype).requires_grad_(True) for x in xs64]
    feats = collections.OrderedDict((str(i), x) for i, x in enumerate(xs))
    track = collections.OrderedDict()
    out = kern(None, feats, None, track)
    if out.ndim == 0:
        up = None
        out.backward()
    else:
        up = torch.randn(out.shape, generator=gen, dtype=torch.float64).to(dtype)
        (out * up).sum().backward()
    rec = {'config': json.dumps(cfg), 'M': M, 'B': B, 'style': style, 'out': out.detach().numpy()}
    if up is not None:
        rec['upstream'] = up.numpy()
    for i, x in enumerate(xs):
        rec['x%d' % i] = x.detach().numpy()
        rec['dx%d' % i] = x.grad.numpy()
        rec['gram%d' % i] = track['layer_track'][i]['gram'].detach().numpy()
    return rec


TOY = {
    'hyper': {'name': 'ensemble', 'ensemble_size': 6,
              'target': {'name': 'mlp', 'activation': 'crater',
                         'layers': [{'in': 2, 'out': 8}, {'in': 8, 'out': 10}, {'in': 10, 'out': 6}, {'in': 6, 'out': 4}]}},
    'method': {'name': 'repulsive_kernel', 'num': 6, 'gamma': 7.5, 'warmup': 10, 'beta_ind': 2.25,
               'model_kernel': layers(he('rbf', he_s=5.0, eps=0.000005, arc_eps=0.00305, params={'param': 5.025}),
                                      {'name': 'list', 'weights': [0.05, 0.35, 1.0, 0.85]}),
               'ind_loss': {'name': 'cross_entropy'}},
}


def run_train_step(dtype, tag):
    """One training step of the toy four-class configuration (configs/toy/fourclass/ens/he-rbf.yml shape, smaller
    sizes) with the REAL reference: FixedEnsembleModel(MLP) forward, RepulsiveKernel.forward, backward."""
    import hyper.target.mlp  # noqa: F401  (registers `mlp`)
    from hyper.generators.base import build_generator
    import hyper.generators.ensemble  # noqa: F401
    from hyper.diversity.methods import build_method
    from hyper.diversity import kernels as rk
    rk.KERN_PARAMS['rbf'] = {'param': 1.0}
    torch.manual_seed(77)
    hyper = build_generator(TOY['hyper']).to(dtype)
    method = build_method(TOY['method']).to(dtype)
    gen = torch.Generator().manual_seed(5)
    X = torch.randn(12, 2, generator=gen, dtype=torch.float64).to(dtype)
    Y = torch.randint(0, 4, (12,), generator=gen)
    trainer = types.SimpleNamespace(training=True)
    track = hyper(params=list(range(6)), x=X, sample_params=True, ret_params=True, feature_split=True, ood_N=0)
    track = method(trainer, track['params'], [], track, X, Y, step=3)
    track['loss'].backward()
    rec = {'config': json.dumps(TOY), 'X': X.numpy(), 'Y': Y.numpy(), 'step': 3,
           'loss': track['loss'].detach().numpy(), 'ind_loss': track['ind_loss'].detach().numpy(),
           'ind_rep_loss': track['ind_rep_loss'].detach().numpy(), 'pred': track['pred'].detach().numpy(),
           'feat_keys': json.dumps(list(track['feats_ind'].keys()))}
    for k, v in hyper.state_dict().items():
        rec['w:' + k] = v.numpy()
    for k, v in hyper.named_parameters():
        rec['g:' + k] = v.grad.numpy()
    np.savez_compressed(os.path.join(OUT, 'toy_train_step_%s.npz' % tag), **rec)
    print('toy_train_step', tag, 'loss', float(track['loss']), 'rep', float(track['ind_rep_loss']))


MNIST = {
    'hyper': {'name': 'ensemble', 'ensemble_size': 5, 'target': {'name': 'lenet5', 'in_channels': 1, 'num_classes': 10}},
    'method': {'name': 'repulsive_kernel', 'num': 5, 'gamma': 10.5, 'warmup': 600,
               'model_kernel': layers(he(eps=0.0025, arc_eps=0.031), MNIST_W), 'ind_loss': {'name': 'cross_entropy'}},
}


def run_lenet_step(dtype, tag):
    """One training step of BASELINE configs[0] (configs/mnist/base.yml + ens/he-lin.yml: LeNet-5 x5, HE on the
    linear-kernel Grams of the five tracked layers) with the REAL reference, batch of 16 MNIST-shaped images."""
    import hyper.target.lenet  # noqa: F401  (registers `lenet5`)
    from hyper.generators.base import build_generator
    import hyper.generators.ensemble  # noqa: F401
    from hyper.diversity.methods import build_method
    torch.manual_seed(31)
    hyper = build_generator(MNIST['hyper']).to(dtype)
    method = build_method(MNIST['method']).to(dtype)
    gen = torch.Generator().manual_seed(6)
    X = torch.randn(16, 1, 28, 28, generator=gen, dtype=torch.float64).to(dtype)
    Y = torch.randint(0, 10, (16,), generator=gen)
    trainer = types.SimpleNamespace(training=True)
    track = hyper(params=list(range(5)), x=X, sample_params=True, ret_params=True, feature_split=True, ood_N=0)
    track = method(trainer, track['params'], [], track, X, Y, step=700)
    track['loss'].backward()
    rec = {'config': json.dumps(MNIST), 'X': X.numpy(), 'Y': Y.numpy(), 'step': 700,
           'loss': track['loss'].detach().numpy(), 'ind_loss': track['ind_loss'].detach().numpy(),
           'ind_rep_loss': track['ind_rep_loss'].detach().numpy(), 'pred': track['pred'].detach().numpy(),
           'feat_keys': json.dumps(list(track['feats_ind'].keys())),
           'feat_shapes': json.dumps([list(v.shape) for v in track['feats_ind'].values()])}
    for k, v in hyper.state_dict().items():
        rec['w:' + k] = v.numpy()
    for k, v in hyper.named_parameters():
        rec['g:' + k] = v.grad.numpy()
    np.savez_compressed(os.path.join(OUT, 'lenet_train_step_%s.npz' % tag), **rec)
    print('lenet_train_step', tag, 'loss', float(track['loss']), 'rep', float(track['ind_rep_loss']))


CIFAR = {
    'hyper': {'name': 'ensemble', 'ensemble_size': 3,
              'target': {'name': 'resnet18', 'num_classes': 10, 'act': 'relu', 'planes': [8, 8, 16, 16, 32]}},
    'method': {'name': 'repulsive_kernel', 'num': 3, 'gamma': 1.0, 'beta_ind': 5.0, 'warmup': 10,
               'model_kernel': layers(he(eps=0.001, arc_eps=0.001, use_exp=True),
                                      {'name': 'list', 'weights': [0.0, 0.15, 0.25, 0.35, 0.4, 0.45, 0.5, 0.0001],
                                       'normalize': False}),
               'ind_loss': {'name': 'cross_entropy'}},
}


def run_resnet_step(dtype, tag):
    """One training step of BASELINE configs[2] (configs/cifar10/base.yml + ens/he-lin.yml: ResNet18 ensemble with
    batched Conv2d / BatchNorm2d, HE on the eight tracked features) with the REAL reference; narrow planes, three
    members, a batch of 6 16x16 images.  Also records the BatchNorm running statistics after the step."""
    import hyper.target.resnet  # noqa: F401  (registers `resnet18`)
    from hyper.generators.base import build_generator
    import hyper.generators.ensemble  # noqa: F401
    from hyper.diversity.methods import build_method
    torch.manual_seed(41)
    hyper = build_generator(CIFAR['hyper']).to(dtype)
    method = build_method(CIFAR['method']).to(dtype)
    gen = torch.Generator().manual_seed(8)
    X = torch.randn(6, 3, 16, 16, generator=gen, dtype=torch.float64).to(dtype)
    Y = torch.randint(0, 10, (6,), generator=gen)
    rec = {'config': json.dumps(CIFAR), 'X': X.numpy(), 'Y': Y.numpy(), 'step': 20}
    for k, v in hyper.state_dict().items():
        rec['w:' + k] = v.clone().numpy()
    trainer = types.SimpleNamespace(training=True)
    track = hyper(params=list(range(3)), x=X, sample_params=True, ret_params=True, feature_split=True, ood_N=0)
    track = method(trainer, track['params'], [], track, X, Y, step=20)
    track['loss'].backward()
    rec.update({'loss': track['loss'].detach().numpy(), 'ind_loss': track['ind_loss'].detach().numpy(),
                'ind_rep_loss': track['ind_rep_loss'].detach().numpy(), 'pred': track['pred'].detach().numpy(),
                'feat_keys': json.dumps(list(track['feats_ind'].keys())),
                'feat_shapes': json.dumps([list(v.shape) for v in track['feats_ind'].values()])})
    for k, v in hyper.state_dict().items():
        if '.norm_buffer.' in k:
            rec['after:' + k] = v.numpy()
    for k, v in hyper.named_parameters():
        rec['g:' + k] = v.grad.numpy()
    np.savez_compressed(os.path.join(OUT, 'resnet_train_step_%s.npz' % tag), **rec)
    print('resnet_train_step', tag, 'loss', float(track['loss']), 'rep', float(track['ind_rep_loss']),
          list(track['feats_ind'].keys()))


SVGD_CFG = {
    'hyper': {'name': 'ensemble', 'ensemble_size': 5, 'target': {'name': 'lenet5', 'in_channels': 1, 'num_classes': 10}},
    'method': {'name': 'svgd', 'num': 5, 'ind_loss': {'name': 'cross_entropy'},
               'model_kernel': {'name': 'model_layers', 'reduction': 'none',
                                'layer_kernel': cka('cossim', eps=0.00025, detach_right=True, reduction='none'),
                                'layer_weighting': {'name': 'start_end_linear', 'first_layer': 0.0, 'middle_start': 0.5,
                                                    'middle_increase': 1.75, 'last_layer': 6.5}}},
}


def run_svgd_step(dtype, tag):
    """One SVGD update of configs/mnist/svgd/cka-cossim.yml (feature-space CKA kernel matrix, reduction none,
    detach_right) with the REAL reference: method.forward, method.backward -> gradients on the ensemble parameters."""
    import hyper.target.lenet  # noqa: F401
    from hyper.generators.base import build_generator
    import hyper.generators.ensemble  # noqa: F401
    from hyper.diversity.methods import build_method
    from hyper.diversity import kernels as rk
    rk.KERN_PARAMS['cossim'] = {'eps': 1e-6}
    torch.manual_seed(52)
    hyper = build_generator(SVGD_CFG['hyper']).to(dtype)
    method = build_method(SVGD_CFG['method']).to(dtype)
    gen = torch.Generator().manual_seed(9)
    X = torch.randn(12, 1, 28, 28, generator=gen, dtype=torch.float64).to(dtype)
    Y = torch.randint(0, 10, (12,), generator=gen)
    trainer = types.SimpleNamespace(training=True)
    track = hyper(params=list(range(5)), x=X, sample_params=True, ret_params=True, feature_split=True, ood_N=0)
    track = method(trainer, track['params'], [], track, X, Y, step=0)
    loss, _ = method.backward(trainer, track['params'], [], track, X, Y)
    rec = {'config': json.dumps(SVGD_CFG), 'X': X.numpy(), 'Y': Y.numpy(), 'loss': loss.detach().numpy(),
           'pred': track['pred'].detach().numpy()}
    for k, v in hyper.state_dict().items():
        rec['w:' + k] = v.numpy()
    for k, v in hyper.named_parameters():
        rec['g:' + k] = v.grad.numpy()
    np.savez_compressed(os.path.join(OUT, 'svgd_step_%s.npz' % tag), **rec)
    print('svgd_step', tag, 'loss', float(loss))


HYPER_HE = he('cossim', eps=0.0025, arc_eps=0.065)
HYPER_W = {'name': 'start_end_linear', 'first_layer': 0.03, 'middle_start': 0.60, 'middle_increase': 1.80, 'last_layer': 6.8}
HYPER = {
    'hyper': {'name': 'mlp_layer_code_generator', 'latent_size': 38, 'dim_multiplier': 2.84, 'mlp_dims': 4, 'bias': True,
              'sn_coeff': 6.0,
              'layer_code_generator': {
                  'name': 'layer_generator', 'code_size': 26,
                  'default_generators': {
                      'conv2d': {'name': 'mlp_sampled_conv_generator', 'input_size': 26, 'dim_multiplier': 2.755,
                                 'mlp_dims': 1, 'bias': True, 'sn_coeff': 3.0},
                      'linear': {'name': 'mlp_layer_generator', 'input_size': 26, 'dim_multiplier': 3.065, 'mlp_dims': 4,
                                 'norm_last': True, 'affine_last': True, 'bias': True, 'sn_coeff': 3.0},
                      'final_linear': {'name': 'mlp_layer_generator', 'input_size': 26, 'dim_multiplier': 3.011,
                                       'mlp_dims': 2, 'norm_last': False, 'norm_before_last': False, 'affine_last': True,
                                       'bias': True, 'sn_coeff': 3.0}},
                  'target': {'name': 'lenet5', 'in_channels': 1, 'num_classes': 10}}},
    'method': {'name': 'repulsive_kernel', 'num': 4, 'gamma': 10.151, 'gamma_ood': 3.688, 'beta_ood': 0.0673,
               'l2_reg': 0.0717, 'warmup': 2075,
               'model_kernel': layers(HYPER_HE, HYPER_W), 'model_kernel_ood': layers(HYPER_HE, HYPER_W),
               'ind_loss': {'name': 'cross_entropy'}, 'ood_loss': {'name': 'mean_softmax_entropy'}},
}


TOY_RBF = he('rbf', he_s=3, eps=0.0001, arc_eps=0.035, params={'param': 6.0})
TOY_RBF_OOD = dict(name='he', he_s=3, eps=0.0001, arc_eps=0.035, feature_kernel={'name': 'feature', 'kernel': 'rbf', 'params': {'param': 6.0}})
TOY_W = {'name': 'list', 'weights': [0.05, 0.35, 0.85, 1.0]}
HYPER_TOY = {
    'hyper': {'name': 'mlp_layer_code_generator', 'latent_size': 8, 'mlp_dims': [24, 32, 48], 'bias': True, 'sn_coeff': 3.0,
              'layer_code_generator': {
                  'name': 'layer_generator', 'code_size': 12,
                  'default_generators': {
                      'linear': {'name': 'mlp_layer_generator', 'input_size': 12, 'mlp_dims': [14, 28], 'norm_last': True,
                                 'affine_last': True, 'bias': True, 'sn_coeff': 5.0},
                      'final_linear': {'name': 'mlp_layer_generator', 'input_size': 12, 'mlp_dims': [32, 48],
                                       'norm_last': False, 'norm_before_last': False, 'affine_last': True, 'bias': True,
                                       'sn_coeff': 5.0}},
                  'target': {'name': 'mlp', 'activation': 'crater',
                             'layers': [{'in': 2, 'out': 30}, {'in': 30, 'out': 40}, {'in': 40, 'out': 30},
                                        {'in': 30, 'out': 4, 'last': True}]}}},
    'method': {'name': 'repulsive_kernel', 'num': 6, 'gamma': 2.0, 'gamma_ood': 22.0, 'warmup': 300, 'beta_ind': 2.0,
               'beta_ood': 2.0, 'model_kernel': layers(TOY_RBF, TOY_W), 'model_kernel_ood': layers(TOY_RBF_OOD, TOY_W),
               'ind_loss': {'name': 'cross_entropy'}},
}


def run_hyper_toy_step(dtype, tag):
    """configs/toy/fourclass/hyper/mlp.yml as shipped (6 members, 12 inlier + 5 outlier points): list-valued mlp_dims,
    a FinalLinear generator without output norm, rbf-kernel HE on inliers and outliers (no detach_diag on the outlier
    kernel, no OOD prediction loss), warm-up factor < 1."""
    _hyper_step(HYPER_TOY, 'hyper_toy_train_step', dtype, tag, members=6, latent=8, xshape=(17, 2), n_in=12, classes=4,
                step=120, seed=64, has_ood_loss=False)


def run_hyper_step(dtype, tag):
    """configs/mnist/hyper/mlp-he.yml (see _hyper_step)."""
    _hyper_step(HYPER, 'hyper_train_step', dtype, tag, members=4, latent=38, xshape=(14, 1, 28, 28), n_in=10, classes=10,
                step=1500, seed=63, has_ood_loss=True)


def run_hyper_cifar_step(dtype, tag):
    """BASELINE.json configs[4] names a CIFAR-10 hypernetwork; the reference ships no such config, and its
    resnet18 cannot be generated (BatchNorm buffers).  The stand-in tools/train_bench.py times: the MNIST hypernetwork of
    configs/mnist/hyper/mlp-he.yml generating a LeNet-5 with 3 input channels (flatten 16*5*5) on CIFAR-shaped input."""
    import copy
    cfg = copy.deepcopy(HYPER)
    cfg['hyper']['layer_code_generator']['target'] = {'name': 'lenet5', 'in_channels': 3, 'num_classes': 10, 'flatten_dims': 400}
    _hyper_step(cfg, 'hyper_cifar_train_step', dtype, tag, members=4, latent=38, xshape=(14, 3, 32, 32), n_in=10, classes=10,
                step=1500, seed=65, has_ood_loss=True)


def _hyper_step(CFG, name, dtype, tag, members, latent, xshape, n_in, classes, step, seed, has_ood_loss):
    """One training step of configs/mnist/hyper/mlp-he.yml (the `hyper:` and `method:` sections as shipped, 4 members)
    with the REAL reference: latent -> code mixer MLP -> layer generators (spectral-norm Linear + GroupNorm + GELU)
    -> LeNet-5 parameters -> batched forward on 10 inlier + 4 outlier images -> RepulsiveKernel loss (cossim HE on
    inliers and outliers, OOD entropy, L2 on the generated weights) -> backward.  Records the generated parameters,
    a digest of every hypernetwork gradient and the power-iteration buffers after the step; the hypernetwork's state is
    the deterministic fill of tests/golden/fill.py (70 M weights are not committed)."""
    import hyper.target.lenet, hyper.target.mlp  # noqa: F401,E401
    from hyper.generators.base import build_generator
    import hyper.generators.mlp  # noqa: F401
    import hyper.layers.generators  # noqa: F401
    from hyper.diversity.methods import build_method
    from hyper.diversity import kernels as rk
    from hyper.util.collections import flatten_keys
    rk.KERN_PARAMS['cossim'] = {'eps': 1e-6}
    rk.KERN_PARAMS['rbf'] = {'param': 1.0}
    torch.manual_seed(seed)
    hyper = build_generator(CFG['hyper']).to(dtype)
    method = build_method(CFG['method']).to(dtype)
    hyper.train()
    gen = torch.Generator().manual_seed(10)
    ood = xshape[0] - n_in
    Z = torch.randn(members, latent, generator=gen, dtype=torch.float64).to(dtype)
    X = torch.randn(*xshape, generator=gen, dtype=torch.float64).to(dtype)
    Y = torch.randint(0, classes, (n_in,), generator=gen)
    sys.path.insert(0, OUT)
    from fill import fill_state, grad_digest, state_stats
    stats = state_stats(hyper)
    fill_state(hyper, stats)
    rec = {'config': json.dumps(CFG), 'Z': Z.numpy(), 'X': X.numpy(), 'Y': Y.numpy(), 'step': step, 'ood_N': ood, 'seed': seed,
           'stats': json.dumps(stats), 'state_keys': json.dumps(list(hyper.state_dict().keys()))}
    trainer = types.SimpleNamespace(training=True)
    params = hyper.forward_params(Z)
    track = hyper(params=params, x=X, sample_params=False, ret_params=True, feature_split=True, ood_N=ood)
    track = method(trainer, track['params'], [], track, X[:n_in], Y, step=step)
    track['loss'].backward()
    for k, v in flatten_keys(params).items():
        rec['p:' + k] = v.detach().numpy()
    rec.update({'loss': track['loss'].detach().numpy(), 'ind_loss': track['ind_loss'].detach().numpy(),
                'ind_rep_loss': track['ind_rep_loss'].detach().numpy(),
                'ood_rep_loss': track['ood_rep_loss'].detach().numpy(), 'pred': track['pred'].detach().numpy(),
                'feat_keys': json.dumps(list(track['feats_ind'].keys()))})
    if has_ood_loss:
        rec['ood_loss'] = track['ood_loss'].detach().numpy()
    for k, v in hyper.state_dict().items():
        if k.endswith(('weight_u', 'weight_v', 'weight_sigma')):
            rec['after:' + k] = v.numpy()
    for k, v in hyper.named_parameters():
        rec['g:' + k] = grad_digest(k, v.grad)
    np.savez_compressed(os.path.join(OUT, '%s_%s.npz' % (name, tag)), **rec)
    print(name, tag, 'loss', float(track['loss']), 'rep', float(track['ind_rep_loss']),
          float(track['ood_rep_loss']), 'keys', len(hyper.state_dict()))


WSVGD_CFG = {
    'hyper': {'name': 'ensemble', 'ensemble_size': 5, 'target': {'name': 'lenet5', 'in_channels': 1, 'num_classes': 10}},
    'method': {'name': 'svgd', 'num': 5, 'ind_loss': {'name': 'cross_entropy'},
               'model_kernel': {'name': 'weight', 'kernel': 'rbf', 'detach_right': True}},
}


def run_wsvgd_step(dtype, tag):
    """One SVGD update of configs/mnist/svgd/rbf.yml (WEIGHT-space rbf kernel with median bandwidth on the members' 44k
    flattened LeNet parameters, detach_right) with the REAL reference."""
    import hyper.target.lenet  # noqa: F401
    from hyper.generators.base import build_generator
    import hyper.generators.ensemble  # noqa: F401
    from hyper.diversity.methods import build_method
    from hyper.diversity import kernels as rk
    rk.KERN_PARAMS['rbf'] = {'param': 1.0}
    torch.manual_seed(53)
    hyper = build_generator(WSVGD_CFG['hyper']).to(dtype)
    method = build_method(WSVGD_CFG['method']).to(dtype)
    gen = torch.Generator().manual_seed(12)
    X = torch.randn(12, 1, 28, 28, generator=gen, dtype=torch.float64).to(dtype)
    Y = torch.randint(0, 10, (12,), generator=gen)
    trainer = types.SimpleNamespace(training=True)
    track = hyper(params=list(range(5)), x=X, sample_params=True, ret_params=True, feature_split=True, ood_N=0)
    track = method(trainer, track['params'], [], track, X, Y, step=0)
    loss, _ = method.backward(trainer, track['params'], [], track, X, Y)
    kern = method.model_kernel(track['params'], track['feats_ind'], track['pred_ind'], track)
    rec = {'config': json.dumps(WSVGD_CFG), 'X': X.numpy(), 'Y': Y.numpy(), 'loss': loss.detach().numpy(),
           'pred': track['pred'].detach().numpy(), 'kern': kern.detach().numpy()}
    for k, v in hyper.state_dict().items():
        rec['w:' + k] = v.numpy()
    for k, v in hyper.named_parameters():
        rec['g:' + k] = v.grad.numpy()
    np.savez_compressed(os.path.join(OUT, 'wsvgd_step_%s.npz' % tag), **rec)
    print('wsvgd_step', tag, 'loss', float(loss), 'kern', kern.detach().numpy()[0, :3])


FSVGD_CFG = {
    'hyper': TOY['hyper'],
    'method': {'name': 'function_svgd', 'num': 6, 'eta': 0.05, 'exp_std': 0.125, 'ind_loss': {'name': 'cross_entropy'},
               'model_kernel': {'name': 'function', 'detach_right': True, 'kernel': 'rbf'}},
}


def run_fsvgd_step(dtype, tag):
    """One function-space SVGD update of configs/toy/fourclass/fsvgd/rbf.yml with the REAL reference (6-member toy MLP):
    likelihood score, SSGE estimate of the functional prior's score from N(0,1) members (their parameters are recorded:
    the draw is the only random input), rbf `function` kernel with detach_right, gradients pushed through the predictions."""
    import hyper.target.mlp  # noqa: F401
    from hyper.generators.base import build_generator
    import hyper.generators.ensemble  # noqa: F401
    from hyper.diversity.methods import build_method
    from hyper.diversity import kernels as rk
    from hyper.util.collections import flatten_keys
    rk.KERN_PARAMS['rbf'] = {'param': 1.0}
    torch.manual_seed(54)
    hyper = build_generator(FSVGD_CFG['hyper']).to(dtype)
    method = build_method(FSVGD_CFG['method']).to(dtype)
    gen = torch.Generator().manual_seed(13)
    X = torch.randn(12, 2, generator=gen, dtype=torch.float64).to(dtype)
    Y = torch.randint(0, 4, (12,), generator=gen)
    prior = {}
    orig = hyper.forward_params

    def recording(codes, device=None):
        out = orig(codes, device=device)
        if isinstance(codes, str):
            for k, v in flatten_keys(out).items():
                out_v = v.to(dtype)
                prior[k] = out_v
            # hand the reference the same values in the working dtype
            from hyper.util.collections import unflatten_keys
            return unflatten_keys(collections.OrderedDict((k, prior[k]) for k in prior))
        return out
    hyper.forward_params = recording
    trainer = types.SimpleNamespace(training=True, hyper=hyper, ddp=False, ood_N=0)
    track = hyper(params=list(range(6)), x=X, sample_params=True, ret_params=True, feature_split=True, ood_N=0)
    track = method(trainer, track['params'], [], track, X, Y, step=0)
    loss, _ = method.backward(trainer, track['params'], [], track, X, Y)
    rec = {'config': json.dumps(FSVGD_CFG), 'X': X.numpy(), 'Y': Y.numpy(), 'loss': loss.detach().numpy(),
           'pred': track['pred'].detach().numpy()}
    for k, v in prior.items():
        rec['prior:' + k] = v.detach().numpy()
    for k, v in hyper.state_dict().items():
        rec['w:' + k] = v.numpy()
    for k, v in hyper.named_parameters():
        rec['g:' + k] = v.grad.numpy()
    np.savez_compressed(os.path.join(OUT, 'fsvgd_step_%s.npz' % tag), **rec)
    print('fsvgd_step', tag, 'loss', float(loss), 'prior keys', list(prior))


def _learned_he(arc_eps):
    return {'name': 'model_layers',
            'layer_kernel': {'name': 'he', 'learnable': True, 'eps': 0.025, 'arc_eps': arc_eps,
                             'feature_kernel': {'name': 'feature', 'kernel': 'rbf'}},
            'layer_weighting': {'name': 'tensor', 'weights': 4, 'learnable': True}}


LEARNED_CFG = {
    'hyper': TOY['hyper'],
    'method': {'name': 'repulsive_kernel', 'num': 6, 'ind_loss': {'name': 'cross_entropy'},
               'ood_loss': {'name': 'mean_softmax_entropy'}, 'gamma': 2.0, 'gamma_ood': 20.0, 'learnable': True,
               'beta_ind': 2.0, 'beta_ood': 0.5, 'warmup': 8,
               'model_kernel': _learned_he(0.025), 'model_kernel_ood': _learned_he(0.15)},
}


def run_learned_step(dtype, tag):
    """configs/toy/fourclass/ens+ood/he-rbf-learned.yml with the REAL reference (6 members, 12 inlier + 5 outlier points,
    warm-up factor 5/8): learnable eps / arc_eps of both HE kernels, learnable `tensor` layer weights, learnable gamma /
    gamma_ood - gradients of the ensemble AND of every method parameter."""
    import hyper.target.mlp  # noqa: F401
    from hyper.generators.base import build_generator
    import hyper.generators.ensemble  # noqa: F401
    from hyper.diversity.methods import build_method
    from hyper.diversity import kernels as rk
    rk.KERN_PARAMS['rbf'] = {'param': 1.0}
    torch.manual_seed(55)
    hyper = build_generator(LEARNED_CFG['hyper']).to(dtype)
    method = build_method(LEARNED_CFG['method']).to(dtype)
    gen = torch.Generator().manual_seed(14)
    X = torch.randn(17, 2, generator=gen, dtype=torch.float64).to(dtype)
    Y = torch.randint(0, 4, (12,), generator=gen)
    with torch.no_grad():      # distinct layer weights
        for mk in (method.model_kernel, method.model_kernel_ood):
            mk.layer_weighting.weights.copy_(torch.rand(mk.layer_weighting.weights.shape, generator=gen, dtype=torch.float64) + 0.5)
    trainer = types.SimpleNamespace(training=True)
    track = hyper(params=list(range(6)), x=X, sample_params=True, ret_params=True, feature_split=True, ood_N=5)
    track = method(trainer, track['params'], [], track, X[:12], Y, step=5)
    track['loss'].backward()
    rec = {'config': json.dumps(LEARNED_CFG), 'X': X.numpy(), 'Y': Y.numpy(), 'step': 5, 'ood_N': 5,
           'loss': track['loss'].detach().numpy(), 'ind_rep_loss': track['ind_rep_loss'].detach().numpy(),
           'ood_rep_loss': track['ood_rep_loss'].detach().numpy(), 'pred': track['pred'].detach().numpy()}
    for k, v in hyper.state_dict().items():
        rec['w:' + k] = v.numpy()
    for k, v in hyper.named_parameters():
        rec['g:' + k] = v.grad.numpy()
    for k, v in method.state_dict().items():
        rec['mw:' + k] = v.numpy()
    for k, v in method.named_parameters():
        if v.grad is not None:
            rec['mg:' + k] = v.grad.numpy()
    np.savez_compressed(os.path.join(OUT, 'learned_step_%s.npz' % tag), **rec)
    print('learned_step', tag, 'loss', float(track['loss']), [k for k in rec if k.startswith('mg:')])


# name -> (Conv2d kwargs, members, input shape (4-D = shared over members), )
CONV_CASES = {
    'conv_shared_gamma': (dict(in_channels=3, out_channels=5, kernel_size=3, padding=1, act='crater', pooling='max',
                               gamma=1.0, param_scale='gamma'), 4, (6, 3, 9, 8)),
    'conv_member_torch_nobias': (dict(in_channels=4, out_channels=7, kernel_size=3, stride=2, padding=1, bias=False,
                                      act='relu', param_scale='torch'), 3, (3, 5, 4, 11, 10)),
    'conv_dilated_rect': (dict(in_channels=2, out_channels=3, kernel_size=(3, 2), stride=(1, 2), padding=(2, 0),
                               dilation=(2, 1), pooling='avg', param_scale='torch'), 3, (3, 4, 2, 10, 9)),
    'conv_same_even': (dict(in_channels=2, out_channels=4, kernel_size=4, padding='same', param_scale='gamma',
                            gamma='crater'), 2, (5, 2, 7, 7)),
    'conv_reflect_1x1': (dict(in_channels=3, out_channels=2, kernel_size=1, padding=1, padding_mode='reflect',
                              param_scale='torch'), 2, (2, 3, 3, 6, 5)),
    'conv_groups2_member': (dict(in_channels=4, out_channels=6, kernel_size=3, padding=1, groups=2, act='relu',
                                 param_scale='torch'), 3, (3, 5, 4, 8, 7)),
    'conv_depthwise_shared': (dict(in_channels=3, out_channels=6, kernel_size=3, stride=2, padding=1, groups=3, bias=False,
                                   param_scale='gamma', gamma=1.0), 2, (4, 3, 9, 9)),
}


def run_conv_case(name, kwargs, M, xshape, dtype):
    """Reference batched Conv2d (hyper/layers/conv.py) forward + autograd: y, dx, dflat for seeded inputs."""
    from hyper.layers.conv import Conv2d
    gen = torch.Generator().manual_seed(sum(map(ord, name)))
    conv = Conv2d(**kwargs)
    flat = torch.randn(M, conv.single_flat_size(), generator=gen, dtype=torch.float64).to(dtype).requires_grad_(True)
    x = torch.randn(*xshape, generator=gen, dtype=torch.float64).to(dtype).requires_grad_(True)
    tracked, y = conv({'self': flat}, x)
    up = torch.randn(y.shape, generator=gen, dtype=torch.float64).to(dtype)
    (y * up).sum().backward()
    return {'kwargs': json.dumps(kwargs), 'M': M, 'flat': flat.detach().numpy(), 'x': x.detach().numpy(),
            'y': y.detach().numpy(), 'up': up.numpy(), 'dflat': flat.grad.numpy(), 'dx': x.grad.numpy()}


def run_uhsic():
    """`ModelGenerator._uHSIC` (generators/base.py:210-221) of the real reference on seeded zero-diagonal Grams."""
    from hyper.generators.base import ModelGenerator
    gen = torch.Generator().manual_seed(21)
    rec = {}
    for idx, (N, P, Q) in enumerate([(9, 5, 7), (16, 24, 3), (4, 6, 6)]):
        a = torch.randn(N, P, generator=gen, dtype=torch.float64)
        b = 0.5 * a[:, :min(P, Q)].repeat(1, 2)[:, :Q] + torch.randn(N, Q, generator=gen, dtype=torch.float64)
        K, Lm = (a @ a.t()).float(), (b @ b.t()).float()      # the reference's ones() are float32: fp32 Grams only
        K.fill_diagonal_(0.0)
        Lm.fill_diagonal_(0.0)
        rec['K%d' % idx], rec['L%d' % idx] = K.numpy(), Lm.numpy()
        rec['v%d' % idx] = np.array([ModelGenerator._uHSIC(None, K, K), ModelGenerator._uHSIC(None, K, Lm),
                                     ModelGenerator._uHSIC(None, Lm, Lm)])
    np.savez_compressed(os.path.join(OUT, 'uhsic_f32.npz'), **rec)
    print('uhsic', rec['v0'], rec['v2'])


def run_functional():
    """The rarely used options of the functional API (hyper/diversity/kernels.py, hyper/diversity/__init__.py,
    HEModelKernel normalize) run through the REAL reference in float64: readd_diag=False, Y != None Grams, log /
    hyperenergy / learned Grams, a learnable rbf bandwidth, grad_var, EMA median bandwidth, EMA centring, half_space,
    normalize=True (with the min / max energies of the pre-solve placed in the reference's own cache, so that the
    fixture pins the arithmetic and not a 100-step Adam run on device noise).  One npz: `<case>/x`, `/y`, `/up`, `/out`,
    `/dx`, `/dy`, ...."""
    import random
    import torch.nn as nn
    from hyper.diversity import kernels as rk, hyperspherical_energy as ref_he
    import hyper.diversity.model_kernels as rmk
    gen = torch.Generator().manual_seed(77)
    M, N, L, P = 4, 7, 5, 6
    rec = {}

    def rnd(*shape):
        return torch.randn(*shape, generator=gen, dtype=torch.float64)

    def case(name, fn, X, Y=None, extra=None):
        X = X.clone().requires_grad_(True)
        Yv = Y.clone().requires_grad_(True) if Y is not None else None
        out = fn(X, Yv)
        up = rnd(*out.shape)
        (out * up).sum().backward()
        rec[name + '/x'], rec[name + '/up'], rec[name + '/out'], rec[name + '/dx'] = X.detach().numpy(), up.numpy(), out.detach().numpy(), X.grad.numpy()
        if Yv is not None:
            rec[name + '/y'] = Yv.detach().numpy()
            rec[name + '/dy'] = Yv.grad.numpy() if Yv.grad is not None else np.zeros(tuple(Yv.shape))
        for k, v in (extra() if extra else {}).items():
            rec[name + '/' + k] = v
        print('functional', name, tuple(out.shape))

    X, Y = rnd(M, N, P), rnd(M, L, P)
    case('linear_readd_false', lambda x, y: rk.batched_gram(x, None, 'linear', True, False, True), X)
    case('linear_cross', lambda x, y: rk.batched_gram(x, y, 'linear', False, True, False), X, Y)
    case('linear_cross_centered', lambda x, y: rk.batched_gram(x, y, 'linear', False, True, True), X, Y)
    case('log', lambda x, y: rk.batched_gram(x, None, 'log', True, True, True), X)
    case('cossim_readd_false', lambda x, y: rk.batched_gram(x, None, 'cossim', True, False, True), X)
    case('cossim_cross', lambda x, y: rk.batched_gram(x, y, 'cossim', False, True, False), X, Y)
    case('hyperenergy', lambda x, y: rk.batched_gram(x, None, 'hyperenergy', True, True, True, arc_eps=0.005, eps=1e-4, mhe_s=2), X)
    case('rbf_readd_false', lambda x, y: rk.batched_gram(x, None, 'rbf', True, False, True, param=1.7, median=True), X)
    case('rbf_cross_fixed', lambda x, y: rk.batched_gram(x, y, 'rbf', False, True, False, param=1.3, median=False), X, Y)
    case('rbf_cross_median_detach_right', lambda x, y: rk.batched_gram(x, y, 'rbf', False, True, True, param=0.8, median=True, detach_right=True), X, Y)
    # learnable bandwidth: one scalar and one per member
    for nm, p0 in (('rbf_learn_scalar', torch.tensor(1.4, dtype=torch.float64)), ('rbf_learn_members', torch.tensor([0.9, 1.1, 1.6, 2.0], dtype=torch.float64))):
        par = p0.clone().requires_grad_(True)
        case(nm, lambda x, y: rk.batched_gram(x, None, 'rbf', True, True, True, param=par, median=False), X,
             extra=lambda: {'param': par.detach().numpy(), 'dparam': par.grad.numpy()})
    # grad_var: (grams, d sum(grams) / d grad_var)
    Xg = X.clone().requires_grad_(True)
    g, dg = rk.rbf_gram(Xg, None, param=1.2, median=True, detach_diag=True, readd=True, grad_var=Xg)
    rec['rbf_grad_var/x'], rec['rbf_grad_var/out'], rec['rbf_grad_var/grad'] = Xg.detach().numpy(), g.detach().numpy(), dg.numpy()
    # EMA median bandwidth: three calls, the second and third update the moving average with a randomly picked member
    ma = rk.EMA(0.9).train()
    random.seed(3)
    Xs = [X, X * 1.5 + 0.2, X * 0.7]
    for i, Xi in enumerate(Xs):
        case('rbf_ema_%d' % i, lambda x, y: rk.batched_gram(x, None, 'rbf', True, True, True, param=1.1, median=True, median_ma=ma, median_update=True), Xi,
             extra=lambda: {'shadow': ma.shadow.detach().numpy().copy()})
    case('rbf_ema_frozen', lambda x, y: rk.batched_gram(x, None, 'rbf', True, True, True, param=1.1, median=True, median_ma=ma, median_update=False), X)
    # EMA centring
    cma = rk.EMA(0.8).train()
    for i, Xi in enumerate(Xs):
        case('center_ema_%d' % i, lambda x, y: rk.batched_gram(x, None, 'linear', True, True, True, ma=cma, update_ma=True), Xi,
             extra=lambda: {'shadow': cma.shadow.detach().numpy().copy()})
    # learned Gram: a shared feature model in front of the linear kernel
    torch.manual_seed(5)
    model = nn.Linear(P, 5).double()
    case('learned_linear', lambda x, y: rk.batched_gram(x, None, 'learned', True, True, True, model=model, learned_kernel='linear'), X,
         extra=lambda: {'w': model.weight.detach().numpy(), 'b': model.bias.detach().numpy(),
                        'dw': model.weight.grad.numpy().copy(), 'db': model.bias.grad.numpy().copy()})
    # half-space energy on a batch of Grams
    G = rnd(5, 6, 6)
    for nm, kw in (('half_space', {}), ('half_space_detach_right', {'detach_right': True}), ('half_space_exp_all', {'use_exp': True, 'remove_diag': False})):
        case(nm, lambda x, y: ref_he(x, s=2.0, half_space=True, eps=0.0025, arc_eps=0.031, **kw), G)
    # normalize=True: min / max energies from the cache
    kern = rmk.build_model_kernel(he(eps=0.0025, arc_eps=0.031, normalize=True)).double()
    arc, eps = kern.get_smoothing()
    Mn, Bn = 5, 9
    rmk.HE_VALUE_CACHE[(Mn, Bn, arc.item(), eps.item())] = (0.37, 2.9)
    feats = rnd(Mn, Bn, 11)
    case('he_normalize', lambda x, y: kern(None, x, None, {}), feats, extra=lambda: {'minmax': np.array([0.37, 2.9])})
    np.savez_compressed(os.path.join(OUT, 'functional_f64.npz'), **rec)
    print('functional:', len(rec), 'arrays')


def f32_twins():
    """The reference's OWN float32 run of the train-step fixtures (same seeds, same inputs cast to float32): how far the
    reference in float32 sits from the reference in float64 is the yardstick for every test tolerance above 1e-5.  Only
    the outputs are kept (losses, predictions, gradients, running statistics): the inputs and weights are those of the
    float64 fixture."""
    for fn, name in ((run_resnet_step, 'resnet_train_step'), (run_svgd_step, 'svgd_step'), (run_wsvgd_step, 'wsvgd_step'),
                     (run_hyper_step, 'hyper_train_step'), (run_hyper_toy_step, 'hyper_toy_train_step'), (run_hyper_cifar_step, 'hyper_cifar_train_step'),
                     (run_learned_step, 'learned_step'), (run_fsvgd_step, 'fsvgd_step'))[(5 if '--f32-twins-rest' in sys.argv else 0):]:
        fn(torch.float32, 'f32')
        path = os.path.join(OUT, '%s_f32.npz' % name)
        z = np.load(path, allow_pickle=False)
        keep = {k: z[k] for k in z.files if not (k.startswith('w:') or k in ('X', 'Y', 'config'))}
        np.savez_compressed(path, **keep)
        print(name, 'f32 twin:', len(keep), 'arrays,', os.path.getsize(path) // 1024, 'KiB')


def main():
    mk = _import_reference()
    if '--hyper-cifar' in sys.argv:
        run_hyper_cifar_step(torch.float64, 'f64')
        run_hyper_cifar_step(torch.float32, 'f32')
        path = os.path.join(OUT, 'hyper_cifar_train_step_f32.npz')
        z = np.load(path, allow_pickle=False)
        np.savez_compressed(path, **{k: z[k] for k in z.files if not (k.startswith('w:') or k in ('X', 'Y', 'config'))})
        return
    if '--conv-groups' in sys.argv:
        for name in ('conv_groups2_member', 'conv_depthwise_shared'):
            kwargs, M, xshape = CONV_CASES[name]
            rec = run_conv_case(name, kwargs, M, xshape, torch.float64)
            np.savez_compressed(os.path.join(OUT, '%s_f64.npz' % name), **rec)
            print(name, 'y', rec['y'].shape, float(np.abs(rec['y']).mean()))
        return
    if '--functional' in sys.argv:
        run_functional()
        return
    if '--uhsic-only' in sys.argv:
        run_uhsic()
        return
    if '--learned-only' in sys.argv:
        run_learned_step(torch.float64, 'f64')
        return
    if '--fsvgd-only' in sys.argv:
        run_fsvgd_step(torch.float64, 'f64')
        return
    if '--wsvgd-only' in sys.argv:
        run_wsvgd_step(torch.float64, 'f64')
        return
    if '--hyper-only' in sys.argv:
        run_hyper_step(torch.float64, 'f64')
        run_hyper_toy_step(torch.float64, 'f64')
        return
    if '--f32-twins' in sys.argv or '--f32-twins-rest' in sys.argv:
        f32_twins()
        return
    for dtype, tag in ((torch.float64, 'f64'), (torch.float32, 'f32')):
        run_train_step(dtype, tag)
        run_lenet_step(dtype, tag)
    f32_twins()
    run_resnet_step(torch.float64, 'f64')
    run_svgd_step(torch.float64, 'f64')
    run_wsvgd_step(torch.float64, 'f64')
    run_fsvgd_step(torch.float64, 'f64')
    run_learned_step(torch.float64, 'f64')
    run_hyper_step(torch.float64, 'f64')
    run_hyper_toy_step(torch.float64, 'f64')
    run_hyper_cifar_step(torch.float64, 'f64')
    run_uhsic()
    run_functional()
    for name, (kwargs, M, xshape) in CONV_CASES.items():
        rec = run_conv_case(name, kwargs, M, xshape, torch.float64)
        np.savez_compressed(os.path.join(OUT, '%s_f64.npz' % name), **rec)
        print(name, 'y', rec['y'].shape, float(np.abs(rec['y']).mean()))
    for name, (cfg, M, B, shapes, style) in CASES.items():
        for dtype, tag in ((torch.float64, 'f64'), (torch.float32, 'f32')):
            rec = run_case(mk, name, cfg, M, B, shapes, style, dtype)
            if name == 'he_lin_bench_small':   # keep the big one small on disk
                rec = {k: v for k, v in rec.items() if not k.startswith('gram')}
            np.savez_compressed(os.path.join(OUT, '%s_%s.npz' % (name, tag)), **rec)
            print(name, tag, 'out', np.asarray(rec['out']).reshape(-1)[:3])


if __name__ == '__main__':
    main()
