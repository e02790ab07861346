"""Train steps/s of the reference's model configurations (BASELINE.json configs[0..2]) on synthetic data of the dataset's
shape: FixedEnsembleModel forward -> RepulsiveKernel (cross-entropy + HE-CKA feature repulsion) -> backward -> optimizer.

    python tools/train_bench.py mnist|toy|cifar|hyper|hyper_cifar [--impl amd|ref] [--steps K] [--members M] [--batch B]

--impl amd : hyper_amd on cuda:0 (HIP kernels).
--impl ref : the real reference on this machine's CPU cores (build container only: needs /root/reference; shims as in
             tests/golden/make_golden.py).  A reported baseline, never part of the product path.
"""
import argparse
import copy
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'he-cka-ensembles_amd')]
import torch  # noqa: E402


def fk(kernel, **params):
    d = {'name': 'feature', 'detach_diag': True, 'kernel': kernel}
    if params:
        d['params'] = params
    return d


def configs(members):
    mnist_w = {'name': 'start_end_linear', 'first_layer': 0.387, 'middle_start': 0.746, 'middle_increase': 1.267, 'last_layer': 3.63}
    hyper_he = {'name': 'model_layers',
                'layer_weighting': {'name': 'start_end_linear', 'first_layer': 0.03, 'middle_start': 0.60, 'middle_increase': 1.80, 'last_layer': 6.8},
                'layer_kernel': {'name': 'he', 'eps': 0.0025, 'arc_eps': 0.065, 'feature_kernel': fk('cossim')}}
    return {
        # configs/mnist/hyper/mlp-he.yml as shipped (BASELINE configs[4]: hypernetwork, members SAMPLED every step; 300 inlier +
        # 100 outlier images per batch)
        'hyper': dict(members=members or 128, batch=300, ood=100, x=(1, 28, 28), classes=10, opt='adamw', sampled=True, clip=0.05,
                      hyper={'name': 'mlp_layer_code_generator', 'latent_size': 38, 'dim_multiplier': 2.84, 'mlp_dims': 4, 'bias': True, 'sn_coeff': 6.0,
                             'layer_code_generator': {'name': 'layer_generator', 'code_size': 26, 'default_generators': {
                                 'conv2d': {'name': 'mlp_sampled_conv_generator', 'input_size': 26, 'dim_multiplier': 2.755, 'mlp_dims': 1, 'bias': True, 'sn_coeff': 3.0},
                                 'linear': {'name': 'mlp_layer_generator', 'input_size': 26, 'dim_multiplier': 3.065, 'mlp_dims': 4, 'norm_last': True,
                                            'affine_last': True, 'bias': True, 'sn_coeff': 3.0},
                                 'final_linear': {'name': 'mlp_layer_generator', 'input_size': 26, 'dim_multiplier': 3.011, 'mlp_dims': 2, 'norm_last': False,
                                                  'norm_before_last': False, 'affine_last': True, 'bias': True, 'sn_coeff': 3.0}},
                                 'target': {'name': 'lenet5', 'in_channels': 1, 'num_classes': 10}}},
                      method={'name': 'repulsive_kernel', 'gamma': 10.151, 'gamma_ood': 3.688, 'beta_ood': 0.0673, 'l2_reg': 0.0717, 'warmup': 2075,
                              'model_kernel': hyper_he, 'model_kernel_ood': copy.deepcopy(hyper_he),
                              'ind_loss': {'name': 'cross_entropy'}, 'ood_loss': {'name': 'mean_softmax_entropy'}}),
        # BASELINE configs[4] names a CIFAR-10 hypernetwork; the reference ships none and its resnet18 cannot be generated
        # (BatchNorm buffers are "only fixed ensembles"): the MNIST hypernetwork above on CIFAR-shaped input,
        # LeNet-5 with 3 input channels (flatten 16*5*5)
        'hyper_cifar': None,
        # configs/mnist/base.yml + ens/he-lin.yml
        'mnist': dict(members=members or 5, batch=300, x=(1, 28, 28), classes=10, opt='adamw',
                      hyper={'name': 'ensemble', 'target': {'name': 'lenet5', 'in_channels': 1, 'num_classes': 10}},
                      method={'name': 'repulsive_kernel', 'gamma': 10.5, 'warmup': 600, 'ind_loss': {'name': 'cross_entropy'},
                              'model_kernel': {'name': 'model_layers', 'layer_weighting': mnist_w,
                                               'layer_kernel': {'name': 'he', 'eps': 0.0025, 'arc_eps': 0.031, 'feature_kernel': fk('linear')}}}),
        # configs/toy/fourclass/base.yml + ens/he-rbf.yml
        'toy': dict(members=members or 30, batch=100, x=(2,), classes=4, opt='adamw',
                    hyper={'name': 'ensemble', 'target': {'name': 'mlp', 'activation': 'crater', 'layers': [
                        {'in': 2, 'out': 30}, {'in': 30, 'out': 40}, {'in': 40, 'out': 30}, {'in': 30, 'out': 4}]}},
                    method={'name': 'repulsive_kernel', 'gamma': 7.5, 'warmup': 0.0035, 'beta_ind': 2.25, 'ind_loss': {'name': 'cross_entropy'},
                            'model_kernel': {'name': 'model_layers', 'layer_weighting': {'name': 'list', 'weights': [0.05, 0.35, 1.0, 0.85]},
                                             'layer_kernel': {'name': 'he', 'he_s': 5.0, 'eps': 0.000005, 'arc_eps': 0.00305,
                                                              'feature_kernel': fk('rbf', param=5.025)}}}),
        # configs/cifar10/base.yml + ens/he-lin.yml (BASELINE.json quotes 10 members)
        'cifar': dict(members=members or 10, batch=128, x=(3, 32, 32), classes=10, opt='sgd',
                      hyper={'name': 'ensemble', 'target': {'name': 'resnet18', 'num_classes': 10, 'act': 'relu'}},
                      method={'name': 'repulsive_kernel', 'gamma': 1.0, 'beta_ind': 5.0, 'warmup': 10, 'ind_loss': {'name': 'cross_entropy'},
                              'model_kernel': {'name': 'model_layers',
                                               'layer_weighting': {'name': 'list', 'weights': [0.0, 0.15, 0.25, 0.35, 0.4, 0.45, 0.5, 0.0001], 'normalize': False},
                                               'layer_kernel': {'name': 'he', 'eps': 0.001, 'arc_eps': 0.001, 'use_exp': True, 'feature_kernel': fk('linear')}}}),
    }


def _with_cifar_hyper(cfgs):
    c = copy.deepcopy(cfgs['hyper'])
    c['x'] = (3, 32, 32)
    c['hyper']['layer_code_generator']['target'] = {'name': 'lenet5', 'in_channels': 3, 'num_classes': 10, 'flatten_dims': 400}
    cfgs['hyper_cifar'] = c
    return cfgs


def run(config, impl='amd', steps=20, warmup=3, members=0, batch=0, no_fused=False, bf16_features=False, graph=False):
    """A few training steps of one configuration; returns {'ms_per_step', 'steps_per_s', 'loss', 'members', 'batch'}."""
    c = _with_cifar_hyper(configs(members))[config]
    M, B = c['members'], batch or c['batch']
    if not c.get('sampled'):
        c['hyper']['ensemble_size'] = M
    ood = c.get('ood', 0)
    ids = M if c.get('sampled') else list(range(M))
    c['method']['num'] = M
    if impl == 'ref':
        sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
        import make_golden
        make_golden._import_reference()
        import hyper.target.lenet, hyper.target.mlp, hyper.target.resnet  # noqa: F401,E401
        import hyper.generators.ensemble, hyper.generators.mlp, hyper.layers.generators  # noqa: F401,E401
        from hyper.generators.base import build_generator
        from hyper.diversity.methods import build_method
        dev = torch.device('cpu')
    else:
        import hyper_amd.target  # noqa: F401
        from hyper_amd.generators import build_generator
        from hyper_amd.diversity.methods import build_method
        dev = torch.device('cuda:0')
    torch.manual_seed(0)
    hyper = build_generator(c['hyper']).to(dev)
    method = build_method(c['method']).to(dev)
    params = list(hyper.parameters()) + list(method.parameters())
    # fused=True: one multi-tensor kernel per step instead of ~ten foreach passes over every parameter (70 M of them in
    # the hypernetwork); the reference's trainer takes torch's default (foreach)
    opt = (torch.optim.AdamW(params, lr=1e-3, capturable=graph, fused=(impl == 'amd' and not no_fused))
           if c['opt'] == 'adamw'
           else torch.optim.SGD(params, lr=0.01, momentum=0.9, fused=(impl == 'amd' and not no_fused)))
    X = torch.randn(B + ood, *c['x'], device=dev)
    Y = torch.randint(0, c['classes'], (B,), device=dev)
    trainer = types.SimpleNamespace(training=True)

    def step(i):
        opt.zero_grad(set_to_none=True)
        track = hyper(params=ids, x=X, sample_params=True, ret_params=True, feature_split=True, ood_N=ood)
        if bf16_features:
            for key in ('feats_ind', 'feats_ood'):
                if track.get(key) is not None:
                    track[key] = type(track[key])((k, v.bfloat16()) for k, v in track[key].items())
        track = method(trainer, track['params'], [], track, X, Y, step=3000 + i)
        track['loss'].backward()
        if c.get('clip'):     # the YAML's `trainer.clip`, as ParticleMethods.batch_end applies it: a non-finite norm skips the step
            try:
                torch.nn.utils.clip_grad_norm_(hyper.parameters(), c['clip'], norm_type=2, error_if_nonfinite=True)
            except RuntimeError:
                return track['loss']
        opt.step()
        return track['loss']

    if graph:
        from hyper_amd.experiments.graphed import GraphedTrainStep
        graphed = GraphedTrainStep(hyper, method, opt, X, Y, member_ids=ids, step=3000, forward_kwargs={'ood_N': ood})
        step = lambda i: graphed()   # noqa: E731
    sync = torch.cuda.synchronize if dev.type == 'cuda' else (lambda: None)
    for i in range(warmup):
        loss = step(i)
    sync()
    t0 = time.perf_counter()
    for i in range(steps):
        loss = step(i)
    sync()
    dt = (time.perf_counter() - t0) / steps
    return {'ms_per_step': dt * 1e3, 'steps_per_s': 1.0 / dt, 'loss': float(loss), 'members': M, 'batch': B}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('config', choices=['mnist', 'toy', 'cifar', 'hyper', 'hyper_cifar'])
    ap.add_argument('--impl', default='amd', choices=['amd', 'ref'])
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--members', type=int, default=0)
    ap.add_argument('--batch', type=int, default=0)
    ap.add_argument('--no-fused', action='store_true', help='torch default (foreach) AdamW instead of the fused kernel')
    ap.add_argument('--bf16-features', action='store_true', help='tracked features enter the loss as bf16 (BASELINE configs[2])')
    ap.add_argument('--graph', action='store_true', help='capture the step in one HIP graph (hyper_amd.experiments.graphed)')
    args = ap.parse_args()
    r = run(args.config, args.impl, args.steps, args.warmup, args.members, args.batch, args.no_fused, args.bf16_features,
            args.graph)
    print('%s impl=%s M=%d B=%d: %.2f ms/step  %.1f steps/s  loss %.4f%s' % (
        args.config, args.impl + ('+graph' if args.graph else ''), r['members'], r['batch'], r['ms_per_step'],
        r['steps_per_s'], r['loss'], '  (%d CPU threads)' % torch.get_num_threads() if args.impl == 'ref' else ''))


if __name__ == '__main__':
    main()
