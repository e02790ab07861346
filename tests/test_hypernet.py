"""Hypernetwork sampler side: `mlp_layer_code_generator` -> `layer_generator` -> layer generators,
against one training step each of configs/mnist/hyper/mlp-he.yml and configs/toy/fourclass/hyper/mlp.yml run by the REAL
reference (tests/golden/hyper_train_step_f64.npz, hyper_toy_train_step_f64.npz; tests/golden/make_golden.py::_hyper_step).

CPU part (here): the generators are shared-weight torch modules, so parameter generation is checked in fp64 without a
GPU - state-dict keys, generated LeNet parameters, power-iteration buffers after the forward, RNG-order of the
initialisation.  GPU part (marked gpu): the whole step - generated parameters through the batched LeNet on libhecka,
RepulsiveKernel with inlier + outlier HE terms, backward into the hypernetwork."""
import copy
import json
import os
import sys
from collections import OrderedDict

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(HERE, 'golden')]
from fill import fill_state, grad_digest  # noqa: E402
from conftest import GOLDEN, Yardstick  # noqa: E402

FIXTURES = ['hyper_train_step', 'hyper_toy_train_step',     # configs/mnist/hyper/mlp-he.yml, configs/toy/fourclass/hyper/mlp.yml
            'hyper_cifar_train_step']                      # BASELINE configs[4] stand-in: the MNIST hypernetwork on 3 x 32 x 32 input


def _build(dtype, name='hyper_train_step'):
    import hyper_amd.target  # noqa: F401
    from hyper_amd.generators import build_generator
    gold = np.load(os.path.join(HERE, 'golden', name + '_f64.npz'))
    cfg = json.loads(str(gold['config']))
    torch.manual_seed(int(gold['seed']))
    hyper = build_generator(cfg['hyper']).to(dtype)
    return gold, cfg, hyper


@pytest.mark.parametrize('name', FIXTURES)
def test_generated_parameters_match_reference(name):
    from hyper_amd.util.collections import flatten_keys
    gold, cfg, hyper = _build(torch.float64, name)
    assert list(hyper.state_dict().keys()) == json.loads(str(gold['state_keys']))
    fill_state(hyper, json.loads(str(gold['stats'])))
    hyper.train()
    params = flatten_keys(hyper.forward_params(torch.from_numpy(gold['Z'])))
    want = [k[2:] for k in gold.files if k.startswith('p:')]
    assert list(params.keys()) == want
    for k in want:
        np.testing.assert_allclose(params[k].detach().numpy(), gold['p:' + k], rtol=1e-9, atol=1e-11)
    for k, v in hyper.state_dict().items():     # one power-iteration step happened, in place
        if k.endswith(('weight_u', 'weight_v', 'weight_sigma')):
            np.testing.assert_allclose(v.numpy(), gold['after:' + k], rtol=1e-9, atol=1e-12)
    # eval mode: no power iteration, same u / v -> same weights again
    hyper.eval()
    again = flatten_keys(hyper.forward_params(torch.from_numpy(gold['Z'])))
    before = {k: v.clone() for k, v in hyper.state_dict().items() if k.endswith('weight_u')}
    hyper.forward_params(torch.from_numpy(gold['Z']))
    for k, v in hyper.state_dict().items():
        if k in before:
            assert torch.equal(v, before[k])
    assert all(torch.isfinite(v).all() for v in again.values())


@pytest.mark.parametrize('name', FIXTURES)
def test_initialisation_statistics_match_reference(name):
    """Same construction order and draws as the reference: with the seed the fixture was made with, every state-dict
    entry's initial mean / std equals the reference's (the fixture keeps those scalars)."""
    gold, cfg, hyper = _build(torch.float64, name)
    stats = json.loads(str(gold['stats']))
    for k, v in hyper.state_dict().items():
        mean, std = stats[k]
        assert abs(float(v.mean()) - mean) <= 1e-6 + 1e-5 * abs(mean), k
        assert abs((float(v.std(unbiased=False)) if v.numel() > 1 else 0.0) - std) <= 1e-6 + 1e-5 * abs(std), k


def test_code_layout_and_errors():
    gold, cfg, hyper = _build(torch.float32)
    lg = hyper.layer_gen
    assert lg.get_total_layers() == 5 and lg.get_total_codes() == 6 + 16 + 1 + 1 + 1 and lg.code_size == 26
    shapes = lg.get_parameter_shapes()
    assert tuple(shapes['0']['self']['weight']) == (6, 1, 5, 5) and tuple(shapes['5']['self']['bias']) == (10,)
    with pytest.raises(AssertionError):
        lg.forward_params(torch.zeros(2, 3, 26))
    with pytest.raises(ValueError):
        hyper.forward_layer_params(torch.zeros(2, lg.get_total_codes(), 7))
    z = hyper.sample_params(3)
    assert z.shape == (3, 38)
    rnd = hyper.forward_params('3', device='cpu')
    assert rnd['3']['self'].shape == (3, 256 * 120 + 120)
    assert hyper.sample_random_params(7) == '7'


@pytest.mark.gpu
@pytest.mark.parametrize('name', FIXTURES)
def test_hypernetwork_train_step_matches_reference(name):
    from hyper_amd.diversity.methods import build_method
    import types
    dev = torch.device('cuda:0')
    gold, cfg, hyper = _build(torch.float32, name)
    fill_state(hyper, json.loads(str(gold['stats'])))
    hyper = hyper.to(dev).train()
    method = build_method(cfg['method']).to(dev)
    Z = torch.from_numpy(gold['Z']).float().to(dev)
    X = torch.from_numpy(gold['X']).float().to(dev)
    Y = torch.from_numpy(gold['Y']).to(dev)
    ood = int(gold['ood_N'])
    params = hyper.forward_params(Z)
    track = hyper(params=params, x=X, sample_params=False, ret_params=True, feature_split=True, ood_N=ood)
    assert list(track['feats_ind'].keys()) == json.loads(str(gold['feat_keys']))
    track = method(types.SimpleNamespace(training=True), track['params'], [], track, X[:-ood], Y, step=int(gold['step']))
    track['loss'].backward()
    torch.cuda.synchronize()

    def rel(a, b):
        a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
        return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))
    # every tolerance: max(1e-5, deviation of the reference's own float32 run of that quantity) - conftest.Yardstick
    yard = Yardstick(name)
    yard.check('pred', track['pred'].detach().cpu().numpy())
    for key in ('loss', 'ind_loss', 'ood_loss', 'ind_rep_loss', 'ood_rep_loss'):
        if key not in gold.files:
            continue
        yard.check(key, track[key].item())
    f32 = np.load(os.path.join(GOLDEN, '%s_f32.npz' % name), allow_pickle=False)

    def digest_err(got, want, n):
        # norm and sum against the norm; the 64 samples against the largest of them / the rms
        scale = max(want[0], 1e-300)
        return max(abs(got[0] - want[0]) / scale, abs(got[1] - want[1]) / (scale * np.sqrt(n)),
                   np.abs(got[2:] - want[2:]).max() / max(np.abs(want[2:]).max(), scale / np.sqrt(n)))
    # the gradients are stored as digests (norm, sum, 64 seeded samples): the yardstick of the `g:` group is the worst
    # digest deviation of the reference's own float32 run
    named = dict(hyper.named_parameters())
    yard.group_err['g:'] = max(digest_err(f32['g:' + k], gold['g:' + k], v.numel()) for k, v in named.items())
    worst = 0.0
    for k, v in named.items():
        worst = max(worst, yard.check('g:' + k, None, err=digest_err(grad_digest(k, v.grad), gold['g:' + k], v.numel())))
    print('hypernetwork step: worst gradient digest error %.2e' % worst)


@pytest.mark.gpu
@pytest.mark.parametrize('H,Wd', [(30840, 2294), (513, 517), (700, 3), (64, 4096)])
def test_power_iteration_products_match_fp64(H, Wd):
    """hecka_gemv_t_f32 (W^T u), hecka_gemv_f32 (W v) and the rank-one gradient of u.(W v) (hecka_ger_f32) against fp64."""
    from hyper_amd.net import hip
    dev = torch.device('cuda:0')
    torch.manual_seed(2)
    W = torch.randn(H, Wd, device=dev, requires_grad=True)
    u = torch.randn(H, device=dev)
    v = torch.randn(Wd, device=dev)
    y = hip.matvec_t(W.detach(), u)
    ref = torch.mv(W.detach().double().t(), u.double())
    assert y.shape == ref.shape
    assert float((y.double() - ref).abs().max() / ref.abs().max()) <= 2e-6
    z = hip.matvec(W, v)
    refz = torch.mv(W.detach().double(), v.double())
    assert z.shape == refz.shape
    assert float((z.detach().double() - refz).abs().max() / refz.abs().max()) <= 2e-6
    sigma = torch.dot(u, z)
    (3.0 * sigma).backward()
    refg = 3.0 * torch.outer(u.double(), v.double())
    assert float((W.grad.double() - refg).abs().max() / refg.abs().max()) <= 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize('N,C,G,gelu,affine', [(4, 64, 8, True, True), (6, 512, 32, True, True), (3, 30840, 8, False, True),
                                               (17, 100, 5, True, False), (2, 48, 6, False, True), (5, 96, 1, True, True)])
def test_groupnorm_gelu_kernel_matches_torch_fp64(N, C, G, gelu, affine):
    """hecka_groupnorm_{fwd,bwd}_f32 (GroupNorm + fused GELU of the generator's norm blocks, hyper/net/models.py:60-118)
    against torch's F.group_norm / F.gelu in float64: output, dx, dweight, dbias within 1e-5 of the largest entry."""
    import torch.nn.functional as F
    from hyper_amd.net import hip
    dev = torch.device('cuda:0')
    torch.manual_seed(N * 1000 + C)
    x = (torch.randn(N, C, device=dev) * 2.0 + 0.3).requires_grad_(True)
    w = (torch.randn(C, device=dev) * 0.5 + 1.0).requires_grad_(True) if affine else None
    b = (torch.randn(C, device=dev) * 0.2).requires_grad_(True) if affine else None
    dy = torch.randn(N, C, device=dev)
    y = hip.group_norm(x, G, w, b, 1e-5, gelu=gelu)
    assert y.grad_fn is not None and 'GroupNormAct' in type(y.grad_fn).__name__
    y.backward(dy)
    xd = x.detach().double().requires_grad_(True)
    wd = w.detach().double().requires_grad_(True) if affine else None
    bd = b.detach().double().requires_grad_(True) if affine else None
    yd = F.group_norm(xd, G, wd, bd, 1e-5)
    if gelu:
        yd = F.gelu(yd)
    yd.backward(dy.double())
    pairs = [('y', y.detach(), yd.detach()), ('dx', x.grad, xd.grad)]
    if affine:
        pairs += [('dweight', w.grad, wd.grad), ('dbias', b.grad, bd.grad)]
    for name, got, ref in pairs:
        err = float((got.double() - ref).abs().max() / ref.abs().max())
        assert err <= 1e-5, (name, err)


@pytest.mark.gpu
def test_generator_mlp_runs_on_libhecka_kernels():
    """The generator's norm blocks on the GPU: Linear through the grouped GEMM (BatchedLinear node), GroupNorm + GELU
    through the fused kernel, same state-dict keys as the reference's Sequential(linear, norm[, act])."""
    from hyper_amd.net import hip
    from hyper_amd.net.models import build_norm_block
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    blk = build_norm_block(24, 64, bias=True, act=torch.nn.GELU, norm=True, gamma=None).to(dev)
    assert sorted(blk.state_dict()) == ['linear.bias', 'linear.weight', 'norm.bias', 'norm.weight']
    assert isinstance(blk.linear, hip.Linear) and isinstance(blk.norm, hip.GroupNorm) and blk.norm.fused_gelu
    x = torch.randn(7, 24, device=dev, requires_grad=True)
    y = blk(x)
    y.square().sum().backward()
    ref = copy.deepcopy(blk).double().cpu()
    xd = x.detach().double().cpu().requires_grad_(True)
    yd = ref(xd)
    yd.square().sum().backward()
    assert float((y.detach().cpu().double() - yd.detach()).abs().max()) <= 1e-5 * float(yd.detach().abs().max())
    assert float((x.grad.cpu().double() - xd.grad).abs().max()) <= 2e-5 * float(xd.grad.abs().max())
    for (n, p), (_, q) in zip(blk.named_parameters(), ref.named_parameters()):
        assert float((p.grad.cpu().double() - q.grad).abs().max()) <= 2e-5 * float(q.grad.abs().max()), n


@pytest.mark.gpu
def test_groupnorm_frozen_weight_still_trains_the_bias():
    """nn.GroupNorm gives a trainable bias its gradient when the weight is frozen; so must the fused kernel's node."""
    from hyper_amd.net import hip
    dev = torch.device('cuda:0')
    torch.manual_seed(2)
    gn = hip.GroupNorm(4, 32, fused_gelu=True).to(dev)
    with torch.no_grad():
        gn.weight.normal_(1.0, 0.2)
        gn.bias.normal_(0.0, 0.2)
    gn.weight.requires_grad_(False)
    x = torch.randn(9, 32, device=dev, requires_grad=True)
    gn(x).square().sum().backward()
    assert gn.weight.grad is None and gn.bias.grad is not None
    ref = torch.nn.GroupNorm(4, 32).double()
    ref.load_state_dict({k: v.detach().cpu().double() for k, v in gn.state_dict().items()})
    ref.weight.requires_grad_(False)
    xd = x.detach().cpu().double().requires_grad_(True)
    torch.nn.functional.gelu(ref(xd)).square().sum().backward()
    assert float((gn.bias.grad.cpu().double() - ref.bias.grad).abs().max()) <= 2e-5 * float(ref.bias.grad.abs().max())
    assert float((x.grad.cpu().double() - xd.grad).abs().max()) <= 2e-5 * float(xd.grad.abs().max())


@pytest.mark.gpu
def test_graphed_hypernetwork_step_samples_members_inside_the_capture():
    """GraphedTrainStep with an int member count: latents are drawn inside the captured step (fresh ones every replay),
    the spectral-norm buffers keep iterating, the loss stays finite and the weights move."""
    import hyper_amd.target  # noqa: F401
    from hyper_amd.generators import build_generator
    from hyper_amd.diversity.methods import build_method
    from hyper_amd.experiments.graphed import GraphedTrainStep
    dev = torch.device('cuda:0')
    gold = np.load(os.path.join(HERE, 'golden', 'hyper_toy_train_step_f64.npz'))
    cfg = json.loads(str(gold['config']))
    torch.manual_seed(5)
    hyper = build_generator(cfg['hyper']).to(dev).train()
    method = build_method(cfg['method']).to(dev)
    opt = torch.optim.AdamW(list(hyper.parameters()) + list(method.parameters()), lr=1e-3, capturable=True)
    X = torch.randn(40, 2, device=dev)
    Y = torch.randint(0, 4, (30,), device=dev)
    step = GraphedTrainStep(hyper, method, opt, X, Y, member_ids=6, step=1000, forward_kwargs={'ood_N': 10})
    w0 = hyper.code_gen.input.linear.weight_orig.detach().clone()
    u0 = hyper.code_gen.input.linear.weight_u.clone()
    losses = [float(step()) for _ in range(4)]
    assert all(np.isfinite(losses))
    assert len(set(losses)) == 4                      # new latents every replay
    assert not torch.equal(w0, hyper.code_gen.input.linear.weight_orig)
    assert not torch.equal(u0, hyper.code_gen.input.linear.weight_u)


# ---- BASELINE configs[4] / VERDICT row n2: the hypernetwork step with the SAMPLED members sharded over ranks -------------
def _sharded_hyper_worker(rank, world, port, name, out):
    """One rank of the member-sharded hypernetwork step, rehearsed on one card (gloo rendezvous, collectives staged
    through the host): the rank's slice of the fixture's latents -> generator -> batched target -> sharded method ->
    backward -> ONE all-reduce of the generator gradients."""
    root = os.path.dirname(HERE)
    sys.path[:0] = [root, os.path.join(root, 'he-cka-ensembles_amd'), HERE, os.path.join(HERE, 'golden')]
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import types
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from hyper_amd import parallel
    from hyper_amd.diversity.methods import build_method
    dev = torch.device('cuda:0')
    gold, cfg, hyper = _build(torch.float32, name)
    fill_state(hyper, json.loads(str(gold['stats'])))
    hyper = hyper.to(dev).train()
    Zall = torch.from_numpy(gold['Z']).float().to(dev)
    M = Zall.shape[0]
    ml = M // world

    class _Fixed(object):           # rank 0's draw IS the fixture's latents
        def sample_params(self, size, device=None):
            return Zall.clone() if rank == 0 else torch.zeros_like(Zall)
    Z = parallel.sample_member_latents(_Fixed(), M, dev)
    assert torch.equal(Z, Zall[rank * ml:(rank + 1) * ml])
    method = parallel.shard_method(build_method(dict(cfg['method'], num=ml)).to(dev))
    X = torch.from_numpy(gold['X']).float().to(dev)
    Y = torch.from_numpy(gold['Y']).to(dev)
    ood = int(gold['ood_N'])
    track = hyper(params=hyper.forward_params(Z), x=X, sample_params=False, ret_params=True, feature_split=True, ood_N=ood)
    track = method(types.SimpleNamespace(training=True), track['params'], [], track, X[:-ood], Y, step=int(gold['step']))
    track['loss'].backward()
    parallel.allreduce_gradients(hyper)
    torch.cuda.synchronize()
    res = {'pred': track['pred'].detach().cpu()}
    for key in ('ind_loss', 'ood_loss', 'ind_rep_loss', 'ood_rep_loss'):
        if track.get(key) is not None:
            res[key] = float(track[key].detach())
    if rank == 0:
        res['digests'] = {k: grad_digest(k, v.grad) for k, v in hyper.named_parameters()}
    torch.save(res, '%s.%d' % (out, rank))
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize('name,world', [('hyper_train_step', 2), ('hyper_toy_train_step', 3), ('hyper_cifar_train_step', 2)])
def test_member_sharded_hypernetwork_step_matches_reference(name, world, tmp_path):
    """The sampled members sharded over `world` ranks (each rank: its slice of the latents through the replicated
    generator, its members on the full batch, the inlier and outlier cossim / rbf HE kernels and the member-averaged OOD
    entropy evaluated over ALL members through hyper_amd.parallel) and the generator gradients summed in one bucket:
    predictions, loss terms and every generator gradient equal the REAL reference's unsharded step (the same fixtures and
    yardstick as test_hypernetwork_train_step_matches_reference)."""
    import torch.multiprocessing as mp
    gold = np.load(os.path.join(GOLDEN, name + '_f64.npz'), allow_pickle=False)
    f32 = np.load(os.path.join(GOLDEN, name + '_f32.npz'), allow_pickle=False)
    out = str(tmp_path / 'hyper')
    port = 29400 + (os.getpid() + 13 * world + len(name)) % 200
    mp.start_processes(_sharded_hyper_worker, args=(world, port, name, out), nprocs=world, join=True, start_method='spawn')
    res = [torch.load('%s.%d' % (out, r), weights_only=False) for r in range(world)]
    yard = Yardstick(name)
    yard.check('pred', torch.cat([r['pred'] for r in res]).numpy())
    for key in ('ood_loss', 'ind_rep_loss', 'ood_rep_loss'):      # global terms: the same value on every rank
        if key in gold.files:
            for r in res:
                yard.check(key, r[key])
    yard.check('ind_loss', sum(r['ind_loss'] for r in res))        # the ranks' shares (already / G) add up to the mean

    def digest_err(got, want, n):
        scale = max(want[0], 1e-300)
        return max(abs(got[0] - want[0]) / scale, abs(got[1] - want[1]) / (scale * np.sqrt(n)),
                   np.abs(got[2:] - want[2:]).max() / max(np.abs(want[2:]).max(), scale / np.sqrt(n)))
    _, _, hyper = _build(torch.float32, name)
    sizes = {k: v.numel() for k, v in hyper.named_parameters()}
    yard.group_err['g:'] = max(digest_err(f32['g:' + k], gold['g:' + k], n) for k, n in sizes.items())
    worst = 0.0
    for k, n in sizes.items():
        worst = max(worst, yard.check('g:' + k, None, err=digest_err(res[0]['digests'][k], gold['g:' + k], n)))
    print('sharded hypernetwork step (%d ranks): worst gradient digest error %.2e' % (world, worst))
