"""world_size-2 gloo test of the member-sharded loss (hyper_amd.parallel) on the CPU: the sharding /
collective logic with an oracle-backed compute backend must reproduce the single-process oracle."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, cfgname, out, mode):
    sys.path[:0] = [ROOT, os.path.join(ROOT, 'he-cka-ensembles_amd'), os.path.join(ROOT, 'tests')]
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HECKA_SHARD_MODE=mode,
                      CPU_BACKEND_MULTI='1' if cfgname in ('he_sliced', 'he_blocked') else '0',
                      CPU_BACKEND_BLOCKED='1' if cfgname == 'he_blocked' else '0',
                      HECKA_SHARD_EXCHANGE='p2p' if cfgname == 'he_p2p' else 'a2a')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import cpu_backend
    from hyper_amd.diversity.model_kernels import build_model_kernel
    from hyper_amd.parallel import ShardedModelLayersKernel
    from collections import OrderedDict
    cfg, feats = _case(cfgname)
    M = feats[0].shape[0]
    Ml = M // world
    xs = [f[rank * Ml:(rank + 1) * Ml].clone().requires_grad_(True) for f in feats]
    kern = ShardedModelLayersKernel(build_model_kernel(cfg), backend=cpu_backend)
    from hyper_amd import parallel
    n0 = parallel.COLLECTIVES
    loss = kern(None, OrderedDict((str(i), x) for i, x in enumerate(xs)), None, OrderedDict())
    loss.backward()
    launched = parallel.COLLECTIVES - n0
    if cfgname in ('he_p2p', 'he_blocked'):
        # whatever the number of layers: forward one exchange + one all-gather, backward one exchange - two when the
        # deepest layer's slices travel first (three and more layers)
        assert launched == (4 if len(xs) >= 3 else 3), launched
    elif mode == 'slice':       # per-layer all_to_all_single: L forward + one all-gather + L backward
        assert launched == 2 * len(xs) + 1, launched
    if cfgname == 'he_blocked':
        assert cpu_backend.CALLS == {'gram_multi_one_buffer': 1, 'pairwise_blocked': 1, 'combine_blocked': 2,
                                     'bwd_multi_sliced': 2}, cpu_backend.CALLS
    if cfgname in ('he_cossim', 'he_rbf'):
        assert cpu_backend.CALLS.get('transform_bwd') == len(xs), cpu_backend.CALLS
    if cfgname == 'he_sliced':
        # (three layers: the deepest one's dX as soon as its slices are in, the other two in one call behind it)
        assert cpu_backend.CALLS == {'gram_multi_sliced': 1, 'combine_batched': 1, 'bwd_multi_sliced': 2}, cpu_backend.CALLS
    torch.save({'loss': loss.detach(), 'grads': [x.grad for x in xs]}, '%s.%d' % (out, rank))
    dist.destroy_process_group()


def _case(name):
    torch.manual_seed(21)
    B = 5 if name == 'he_uneven' else 12      # B*B = 25 does not split evenly: slices of 12/13 and 8/8/9 elements
    M = 6
    if name in ('he_sliced', 'he_blocked'):   # slices of whole Gram rows (4 / 2 of 8): the slice-major layouts, no repacking
        M, B = 8, 8
    feats = [torch.randn(M, B, p, dtype=torch.float64) for p in ((9, 20, 14) if name in ('he_sliced', 'he_blocked', 'he_rbf') else (9, 20))]
    lk = {'he': dict(name='he', eps=0.0025, arc_eps=0.031), 'he_uneven': dict(name='he', eps=0.0025, arc_eps=0.031),
          'he_sliced': dict(name='he', eps=0.0025, arc_eps=0.031),
          'he_p2p': dict(name='he', eps=0.0025, arc_eps=0.031),
          'he_blocked': dict(name='he', eps=0.0025, arc_eps=0.031),
          'he_cossim': dict(name='he', eps=0.0025, arc_eps=0.031),          # configs/mnist/hyper/mlp-he.yml's kernel
          'he_rbf': dict(name='he', he_s=5.0, eps=0.000005, arc_eps=0.00305),   # configs/toy/fourclass/ens/he-rbf.yml's
          'he_exp_dr': dict(name='he', eps=0.001, arc_eps=0.001, use_exp=True, detach_right=True),
          'cka': dict(name='cka', eps=1e-6)}[name]
    lk['feature_kernel'] = {'name': 'feature', 'detach_diag': True, 'kernel': 'linear'}
    if name == 'he_cossim':
        lk['feature_kernel']['kernel'] = 'cossim'
    if name == 'he_rbf':
        lk['feature_kernel'].update(kernel='rbf', params={'param': 5.025})
    cfg = {'name': 'model_layers', 'layer_kernel': lk, 'reduction': 'mean',
           'layer_weighting': {'name': 'list', 'weights': [0.3, 1.2, 0.7][:len(feats)]}}
    return cfg, feats


@pytest.mark.parametrize('name', ['he', 'he_exp_dr', 'cka', 'he_uneven'])
@pytest.mark.parametrize('world', [2, 3])
@pytest.mark.parametrize('mode', ['slice', 'gather'])
def test_sharded_matches_single_process_oracle(name, world, mode, tmp_path):
    from oracle import hecka_ref as R
    out = str(tmp_path / 'res')
    port = 29600 + (os.getpid() + world * 7 + len(name) + 13 * len(mode)) % 300
    mp.start_processes(_worker, args=(world, port, name, out, mode), nprocs=world, join=True, start_method='spawn')
    cfg, feats = _case(name)
    xo = [f.clone().requires_grad_(True) for f in feats]
    ref = R.run_config(cfg, xo)
    ref.backward()
    Ml = feats[0].shape[0] // world
    for r in range(world):
        res = torch.load('%s.%d' % (out, r), weights_only=True)
        assert abs(res['loss'].item() - ref.item()) <= 1e-10 * abs(ref.item())
        for g, x in zip(res['grads'], xo):
            exp = x.grad[r * Ml:(r + 1) * Ml]
            assert np.abs(g.numpy() - exp.numpy()).max() <= 1e-9 * np.abs(exp.numpy()).max()


@pytest.mark.parametrize('world', [2, 4])
def test_slice_major_layouts_and_batched_stages_match_oracle(world, tmp_path):
    """The branch of hyper_amd.parallel the GPU path takes when a slice is whole Gram rows: Grams produced directly in
    the all-to-all's send order, W consumed as it arrives, one combine / bounds / dX call for all layers."""
    test_sharded_matches_single_process_oracle('he_sliced', world, 'slice', tmp_path)


@pytest.mark.parametrize('name', ['he_cossim', 'he_rbf'])
@pytest.mark.parametrize('world', [2, 3])
def test_sharded_cossim_and_rbf_layers_match_oracle(name, world, tmp_path):
    """cossim / rbf Grams (the hypernetwork and toy configurations) through the sliced scheme: the row-wise transform and
    its backward stay with the member's owner, its row means travel in cvec's place."""
    test_sharded_matches_single_process_oracle(name, world, 'slice', tmp_path)


@pytest.mark.parametrize('world', [2, 3])
def test_batched_send_receive_exchange_matches(world, tmp_path):
    """HECKA_SHARD_EXCHANGE=p2p: every layer's segments of an exchange in one batched send / receive group instead of
    one all_to_all_single per layer."""
    test_sharded_matches_single_process_oracle('he_p2p', world, 'slice', tmp_path)


@pytest.mark.parametrize('world', [2, 4])
def test_blocked_layouts_one_all_to_all_per_exchange(world, tmp_path):
    """The bench's branch: all layers in one buffer, ONE all_to_all_single per exchange (4 collectives per evaluation),
    the small stages on the received layout [source rank][layer][members][n]."""
    test_sharded_matches_single_process_oracle('he_blocked', world, 'slice', tmp_path)


def test_slice_bounds():
    from hyper_amd.parallel import slice_bounds
    assert slice_bounds(512 * 512, 8, 64) == [i * 32768 for i in range(9)]
    assert slice_bounds(25, 3) == [0, 8, 16, 25]
    b = slice_bounds(40 * 40, 3, 64)                       # 25 chunks of 64 over 3 ranks
    assert b[0] == 0 and b[-1] == 1600 and all(x % 64 == 0 for x in b) and b == sorted(b)
    assert slice_bounds(7 * 7, 2, 64) == [0, 24, 49]        # not a multiple of the granule: element granularity
    assert slice_bounds(64, 4, 64) == [0, 0, 0, 0, 64]      # fewer chunks than ranks: empty slices


# ---- pieces of the member-sharded HYPERNETWORK step (shared generator weights): latent broadcast, the exact member-averaged
# OOD entropy, the flat-bucket gradient all-reduce - with a small shared-weight torch generator standing in for the MLP
def _gen_case():
    torch.manual_seed(5)
    gen = torch.nn.Linear(5, 12).double()
    X = torch.randn(7, 4, dtype=torch.float64)
    Y = torch.randint(0, 3, (7,))
    return gen, X, Y


def _gen_loss(gen, z, X, Y, members_total, entropy, ce_scale):
    import torch.nn.functional as F
    W = gen(z).view(z.shape[0], 4, 3)
    pred = torch.einsum('bi,mio->mbo', X, W)
    ce = F.cross_entropy(pred.reshape(-1, 3), Y.repeat(z.shape[0]))
    return ce_scale * ce + 0.7 * entropy(z.shape[0], None, pred, None)


def _gen_worker(rank, world, port, out):
    sys.path[:0] = [ROOT, os.path.join(ROOT, 'he-cka-ensembles_amd'), os.path.join(ROOT, 'tests')]
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from hyper_amd import parallel
    gen, X, Y = _gen_case()

    class Hyper(object):
        def sample_params(self, size, device=None):
            torch.manual_seed(77 + rank)          # ranks would draw DIFFERENT latents: rank 0's draw must win
            return torch.randn(size, 5, dtype=torch.float64)
    z = parallel.sample_member_latents(Hyper(), 6, 'cpu')
    loss = _gen_loss(gen, z, X, Y, 6, parallel.ShardedMeanSoftmaxEntropy(), 1.0 / world)
    loss.backward()
    parallel.allreduce_gradients(gen)
    torch.save({'z': z, 'grads': [p.grad for p in gen.parameters()]}, '%s.%d' % (out, rank))
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_sharded_generator_pieces_match_single_process(world, tmp_path):
    from hyper_amd.diversity.losses import MeanSoftmaxEntropy
    out = str(tmp_path / 'gen')
    port = 29300 + (os.getpid() + 5 * world) % 200
    mp.start_processes(_gen_worker, args=(world, port, out), nprocs=world, join=True, start_method='spawn')
    gen, X, Y = _gen_case()
    torch.manual_seed(77)
    z = torch.randn(6, 5, dtype=torch.float64)
    _gen_loss(gen, z, X, Y, 6, MeanSoftmaxEntropy(), 1.0).backward()
    res = [torch.load('%s.%d' % (out, r), weights_only=True) for r in range(world)]
    assert torch.equal(torch.cat([r['z'] for r in res]), z)
    for r in res:       # every rank ends with the unsharded step's generator gradients
        for g, p in zip(r['grads'], gen.parameters()):
            assert float((g - p.grad).abs().max()) <= 1e-12 * float(p.grad.abs().max())
