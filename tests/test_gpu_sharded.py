"""GPU tests of the member-sharded path (hyper_amd.parallel) with the real kernels.

* hecka_combine_slice_f32 on column slices against the same columns of hecka_combine_f32.
* Two / four ranks rehearsed on ONE card: one process per rank, gloo rendezvous (RCCL refuses two ranks on one
  device; parallel.py stages the collectives through the host for gloo), every stage on the GPU through libhecka.
  Loss and gradients must equal the unsharded hyper_amd result to the parity tolerance (rel 1e-5)."""
import os
import sys
from collections import OrderedDict

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RTOL = 1e-5


def _relerr(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


@pytest.mark.parametrize('M,B', [(64, 128), (32, 64), (12, 40), (5, 7)])
def test_combine_slice_equals_full_combine(M, B):
    """M = 64 takes the streaming kernel (64-column chunks, chunk0 != 0), the others the tiled kernel."""
    from hyper_amd import ops, _lib
    dev = torch.device('cuda:0')
    torch.manual_seed(5)
    A = torch.randn(M, M, device=dev) / M
    C = torch.randn(M, B, B, device=dev)
    gamma = torch.randn(M, B, device=dev)
    W = torch.empty(M, B, B, device=dev)
    L = _lib.lib()
    _lib.check(L.hecka_combine_f32(A.data_ptr(), C.data_ptr(), gamma.data_ptr(), M, M, B, 1, W.data_ptr(), None,
                                   torch.cuda.current_stream().cuda_stream), 'combine')
    N2 = B * B
    from hyper_amd.parallel import slice_bounds
    for world in (1, 2, 3, 8):
        bounds = slice_bounds(N2, world, ops.SLICE_GRANULE)
        assert bounds[0] == 0 and bounds[-1] == N2
        for g in range(world):
            s, e = bounds[g], bounds[g + 1]
            Cs = C.view(M, N2)[:, s:e].contiguous()
            Ws = ops.combine_slice(A, Cs, gamma, B, s, True)
            assert Ws.shape == (M, e - s)
            if e > s:
                assert torch.equal(Ws, W.view(M, N2)[:, s:e]), (world, g)


@pytest.mark.parametrize('M,B,Ps,rows', [(8, 512, (2048, 512, 128), 64), (16, 128, (256, 128), 32), (64, 512, (1024, 512), 0),
                                          (8, 512, (1024, 256), 64), (16, 256, (512, 256), 0),      # every layer % 256: the 256 x 256 eight-wave tile
                                          (8, 256, (384, 128), 128), (128, 400, (256, 864), 0), (8, 384, (256, 128), 0)])
def test_multi_layer_dx_and_slice_major_layouts_are_bit_equal(M, B, Ps, rows, monkeypatch):
    """ONE dX launch for all layers (hecka_gram_bwd_multi_f32) == the per-layer launches, bit for bit (same tile code,
    same accumulation order; see `wide` below for the eight-wave tile) - interior and ragged shapes; and with `rows` > 0 the slice-major layouts of the member-
    sharded path: the Gram written as [B/rows, M, rows, B] is the standard Gram permuted, and dX from a W in that order
    equals dX from the standard W."""
    from hyper_amd import ops
    dev = torch.device('cuda:0')
    if B % 128:
        monkeypatch.setenv('HECKA_RAGGED_X3', '1')
    torch.manual_seed(M + B)
    Xs = [torch.randn(M, B, p, device=dev) * (1 + i) for i, p in enumerate(Ps)]
    Ws = [torch.randn(M, B, B, device=dev) for _ in Ps]
    wmaxs = [W.abs().amax(dim=(1, 2)) for W in Ws]
    xmaxs = [X.abs().amax(dim=(1, 2)) for X in Xs]
    assert ops.gram_bwd_multi_eligible(Ws, Xs, wmaxs, xmaxs)
    monkeypatch.setenv('HECKA_BWD_MULTI', '0')       # per-layer launches on the four-wave tiles ...
    one = [ops.gram_bwd_w(W, X, wm, xm) for W, X, wm, xm in zip(Ws, Xs, wmaxs, xmaxs)]
    monkeypatch.delenv('HECKA_BWD_MULTI')            # ... against one launch (256 x 256 eight-wave tiles where every P % 256 == 0)
    many = ops.gram_bwd_multi(Ws, Xs, wmaxs, xmaxs)
    wide = B % 256 == 0 and all(p % 256 == 0 for p in Ps)
    if wide:
        # the eight-wave launch runs the 16x16x32 tile (gemm_f16x3_k32.hpp): another fp32 summation order than the four-wave
        # 32x32x16 tiles - equal to rounding, and bit-equal to the same tile launched layer by layer ...
        for a, b in zip(one, many):
            assert float((a - b).abs().max()) <= 2e-6 * float(a.abs().max())
        one = [ops.gram_bwd_multi([W], [X], [wm], [xm])[0] for W, X, wm, xm in zip(Ws, Xs, wmaxs, xmaxs)]
        prev = os.environ.get('HECKA_K32')
        monkeypatch.setenv('HECKA_K32', '0')         # ... and with that tile switched off, to the four-wave launches
        for a, b in zip([ops.gram_bwd_w(W, X, wm, xm) for W, X, wm, xm in zip(Ws, Xs, wmaxs, xmaxs)],
                        ops.gram_bwd_multi(Ws, Xs, wmaxs, xmaxs)):
            assert torch.equal(a, b)
        if prev is None:
            monkeypatch.delenv('HECKA_K32')
        else:
            monkeypatch.setenv('HECKA_K32', prev)    # (the suite may be run with the tile switched off as a whole)
    for a, b in zip(one, many):
        assert torch.equal(a, b)
    ref = Ws[0].double() @ Xs[0].double()
    assert float((many[0].double() - ref).abs().max() / ref.abs().max()) <= 1e-5
    gate = torch.zeros(1, device=dev)
    assert all(float(g.abs().max()) == 0.0 for g in ops.gram_bwd_multi(Ws, Xs, wmaxs, xmaxs, gate))
    if rows:
        G = B // rows
        Wsl = [W.view(M, G, rows, B).permute(1, 0, 2, 3).contiguous() for W in Ws]
        assert ops.gram_bwd_multi_eligible(Wsl, Xs, wmaxs, xmaxs, rows)
        for a, b in zip(one, ops.gram_bwd_multi(Wsl, Xs, wmaxs, xmaxs, slice_rows=rows)):
            assert torch.equal(a, b)
        std = ops.centered_gram_multi(Xs)
        sl = ops.centered_gram_multi(Xs, slice_rows=rows)
        for (C, cvec, xmax), (Cs, cvs, xms) in zip(std, sl):
            assert Cs.shape == (G, M, rows, B)
            assert torch.equal(C.view(M, G, rows, B).permute(1, 0, 2, 3), Cs)
            assert torch.equal(cvec, cvs) and torch.equal(xmax, xms)


@pytest.mark.parametrize('M,Ml,B,world', [(64, 64, 128, 4), (64, 32, 512, 8), (12, 12, 40, 2)])
def test_batched_combine_slice_equals_per_layer(M, Ml, B, world):
    """hecka_combine_slice_batched_f32 (one launch for all layers; the streaming kernel at 64 members, layer by layer on
    the tiled kernel otherwise) == hecka_combine_slice_f32 per layer, bit for bit."""
    from hyper_amd import ops
    from hyper_amd.parallel import slice_bounds
    dev = torch.device('cuda:0')
    torch.manual_seed(3)
    L = 3
    bounds = slice_bounds(B * B, world, ops.SLICE_GRANULE)
    s, e = bounds[1], bounds[2]
    A = torch.randn(L, Ml, M, device=dev) / M
    Cs = torch.randn(L, M, e - s, device=dev)
    gamma = torch.randn(L, Ml, B, device=dev)
    Wb = ops.combine_slice_batched(A, Cs, gamma, B, s, True)
    for l in range(L):
        assert torch.equal(Wb[l], ops.combine_slice(A[l], Cs[l], gamma[l], B, s, True)), l


@pytest.mark.parametrize('Ma,B,n_start,n', [(64, 256, 0, 256 * 256), (64, 128, 128 * 32, 128 * 64), (32, 128, 0, 128 * 128),
                                            (64, 192, 64 * 7, 64 * 50)])
def test_fp16_split_combine_matches_fp64_with_members_decades_apart(Ma, B, n_start, n):
    """hecka_combine_slice_batched_f32 with nsq (three fp16 products, per-member power-of-two scales from sqrt(nsq_j),
    per-row scales of the folded A) against fp64 and against the fp32-MFMA form: members' Grams six decades apart with
    coefficients A_ij ~ 1 / (|C_i| |C_j|) as the normalised heads produce them, so every member contributes equally to a
    row of W; whole-Gram, slice (both kernel forms: fixed strip / general chunks) and the 32-row variant."""
    from hyper_amd import ops
    dev = torch.device('cuda:0')
    torch.manual_seed(B + Ma)
    L, M = 2, 64
    spread = 10.0 ** torch.linspace(-3, 3, M, device=dev)
    Cfull = torch.randn(L, M, B * B, device=dev) * spread[None, :, None] * torch.tensor([1.0, 40.0], device=dev)[:, None, None]
    Cfull[1, 5] = 0                                                  # a member with an all-zero Gram
    nrm = Cfull.norm(dim=2).clamp_min(1e-30)
    A = (torch.randn(L, M, M, device=dev) / (nrm[:, :, None] * nrm[:, None, :]))[:, :Ma].contiguous()
    A[0, 3] = 0                                                      # an all-zero row of A
    A[1, :, 5] = 0                                                   # (and no coefficient on the zero Gram)
    nsq = Cfull.double().square().sum(dim=2).float()
    Cs = Cfull[:, :, n_start:n_start + n].contiguous()
    gamma = torch.randn(L, Ma, B, device=dev) * 1e-2 * (A.abs().amax(dim=2) * Cs.abs().amax(dim=2).amax(dim=1, keepdim=True))[:, :, None]
    W3 = ops.combine_slice_batched(A, Cs, gamma, B, n_start, True, nsq)
    W1 = ops.combine_slice_batched(A, Cs, gamma, B, n_start, True)
    a = torch.arange(n_start, n_start + n, device=dev) // B
    b = torch.arange(n_start, n_start + n, device=dev) % B
    Wd = 2.0 * (A.double() @ Cs.double()) + gamma.double()[:, :, a] - gamma.double()[:, :, b]
    Wd[:, :, a == b] = 0
    scale = Wd.abs().amax(dim=2).clamp_min(1e-300)
    e3 = float(((W3.double() - Wd).abs().amax(dim=2) / scale).max())
    e1 = float(((W1.double() - Wd).abs().amax(dim=2) / scale).max())
    assert torch.isfinite(W3).all()
    assert e3 <= 1e-6 and e1 <= 1e-6, (e3, e1)
    assert bool((W3[0, 3] == (gamma[0, 3][a] - gamma[0, 3][b]) * (a != b)).all())       # zero row of A: the gamma terms alone


@pytest.mark.parametrize('Ml,B,rows', [(8, 512, 64), (16, 256, 64), (32, 128, 64), (64, 128, 128)])
def test_blocked_layouts_are_bit_equal_to_the_standard_ones(Ml, B, rows):
    """The layouts of the one-all_to_all-per-exchange path (64 members in blocks of Ml rows, all layers in one buffer):
    pairwise and combine on [G][L][Ml][n] == the same kernels on [L][64][n]; the all-layers Gram written as
    [B/rows][L][M][rows][B] and dX read from it == the per-layer slice-major calls.  Bit for bit."""
    from hyper_amd import ops
    dev = torch.device('cuda:0')
    torch.manual_seed(Ml + B)
    M, G, L, n = 64, 64 // Ml, 3, rows * B
    assert ops.blocked_eligible(M, n, Ml)
    Cstd = torch.randn(L, M, n, device=dev)
    Cblk = Cstd.view(L, G, Ml, n).permute(1, 0, 2, 3).contiguous()
    assert torch.equal(ops.pairwise_self_blocked(Cblk), ops.pairwise_self_batched(Cstd))
    A = torch.randn(L, M, M, device=dev) / M
    gamma = torch.randn(L, M, B, device=dev)
    n_start = n * (1 if B * B >= 2 * n else 0)
    nsq = 1.5 * Cstd.square().sum(dim=2)           # (a slice's norm is below the whole Gram's)
    for nq in (None, nsq):                         # fp32-MFMA products / the fp16-split ones
        Wstd = ops.combine_slice_batched(A, Cstd, gamma, B, n_start, True, nq)
        for l0, Lc in ((0, 1), (1, 2), (0, 3)):
            Wb = ops.combine_slice_blocked(A[l0:l0 + Lc].contiguous(), Cblk, gamma[l0:l0 + Lc].contiguous(), l0, B, n_start, True,
                                           None if nq is None else nq[l0:l0 + Lc])
            assert Wb.shape == (G, Lc, Ml, n)
            assert torch.equal(Wb.permute(1, 0, 2, 3).reshape(Lc, M, n), Wstd[l0:l0 + Lc]), (l0, Lc)
    # Gram / dX with every layer in one buffer (a rank's Ml members)
    # (8 members x 512 samples with deep layers: the launch splits them over K - slabs, then the placing reduction)
    Ps = (4096, 2048, 256) if Ml == 8 else ((512, 256, 128) if B % 256 == 0 else (256, 128, 128))
    Xs = [torch.randn(Ml, B, p, device=dev) * (1 + i) for i, p in enumerate(Ps)]
    per = ops.centered_gram_multi(Xs, slice_rows=rows)
    one, buf = ops.centered_gram_multi(Xs, slice_rows=rows, one_buffer=True)
    assert buf.shape == (B // rows, L, Ml, rows, B)
    for (C, cvec, xmax), (Cv, cv2, xm2) in zip(per, one):
        assert torch.equal(C, Cv) and torch.equal(cvec, cv2) and torch.equal(xmax, xm2)
    Wbuf = torch.randn(B // rows, L, Ml, rows, B, device=dev)
    wmaxs = [Wbuf[:, k].abs().amax(dim=(0, 2, 3)) for k in range(L)]
    xmaxs = [X.abs().amax(dim=(1, 2)) for X in Xs]
    want = ops.gram_bwd_multi([Wbuf[:, k].contiguous() for k in range(L)], Xs, wmaxs, xmaxs, slice_rows=rows)
    got = ops.gram_bwd_multi([Wbuf[:, k] for k in range(L)], Xs, wmaxs, xmaxs, slice_rows=rows, slice_members=L * Ml)
    for a, b in zip(want, got):
        assert torch.equal(a, b)


def _case(name):
    torch.manual_seed(17)
    name = name.split(':')[0]
    M, B, Ps = {'m64': (64, 128, (256, 64, 48)), 'm12': (12, 40, (24, 9)), 'm64b512': (64, 512, (2048, 256)),
                'm64b256': (64, 256, (512, 256, 256)),
                'm12cos': (12, 40, (24, 9, 16)), 'm16rbf': (16, 64, (30, 40))}[name]
    feats = [torch.randn(M, B, p) * (1.0 + 0.5 * i) for i, p in enumerate(Ps)]
    fk = {'name': 'feature', 'detach_diag': True, 'kernel': 'linear'}
    lk = {'name': 'he', 'eps': 0.0025, 'arc_eps': 0.031}
    if name == 'm12cos':        # the hypernetwork's kernel (configs/mnist/hyper/mlp-he.yml)
        fk['kernel'] = 'cossim'
    if name == 'm16rbf':        # the toy configuration's (configs/toy/fourclass/ens/he-rbf.yml)
        fk.update(kernel='rbf', params={'param': 5.025})
        lk = {'name': 'he', 'he_s': 5.0, 'eps': 0.000005, 'arc_eps': 0.00305}
    cfg = {'name': 'model_layers', 'reduction': 'mean', 'layer_kernel': dict(lk, feature_kernel=fk),
           'layer_weighting': {'name': 'list', 'weights': [0.3, 1.2, 0.7][:len(Ps)]}}
    return cfg, feats


def _worker(rank, world, port, name, out, mode):
    sys.path[:0] = [ROOT, os.path.join(ROOT, 'he-cka-ensembles_amd'), os.path.join(ROOT, 'tests')]
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HECKA_SHARD_MODE=mode)
    variant = name.split(':')[1] if ':' in name else ''
    if variant:     # a knob's non-default value (read once, at import)
        os.environ[{'noblocked': 'HECKA_SHARD_BLOCKED', 'nosplit': 'HECKA_SHARD_BWD_SPLIT'}[variant]] = '0'
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from hyper_amd.diversity.model_kernels import build_model_kernel
    from hyper_amd import parallel
    from hyper_amd.parallel import ShardedModelLayersKernel
    dev = torch.device('cuda:0')
    cfg, feats = _case(name)
    Ml = feats[0].shape[0] // world
    xs = [f[rank * Ml:(rank + 1) * Ml].to(dev).requires_grad_(True) for f in feats]
    kern = ShardedModelLayersKernel(build_model_kernel(cfg).to(dev))
    n0 = parallel.COLLECTIVES
    loss = kern(None, OrderedDict((str(i), x) for i, x in enumerate(xs)), None, OrderedDict())
    loss.backward()
    torch.cuda.synchronize()
    if name.startswith('m64b256'):   # three layers: 4 collectives blocked (3 without the split), 2 L + 1 per layer otherwise
        assert parallel.COLLECTIVES - n0 == {'': 4, 'nosplit': 3, 'noblocked': 7}[variant], parallel.COLLECTIVES - n0
    torch.save({'loss': loss.detach().cpu(), 'grads': [x.grad.cpu() for x in xs]}, '%s.%d' % (out, rank))
    dist.destroy_process_group()


@pytest.mark.parametrize('name,world,mode', [('m64', 2, 'slice'), ('m64', 4, 'slice'), ('m12', 3, 'slice'),
                                             ('m64', 2, 'gather'), ('m12cos', 3, 'slice'), ('m16rbf', 2, 'slice'),
                                             ('m64b256', 4, 'slice'), ('m64b256:noblocked', 4, 'slice'),
                                             ('m64b256:nosplit', 4, 'slice'),
                                             ('m64b512', 8, 'slice')])   # the last: the bench's
# member / sample counts at the driver's largest world size (8 members per rank, 32768-column slices, split-K Grams)
def test_ranks_on_one_card_match_unsharded(name, world, mode, tmp_path):
    import torch.multiprocessing as mp
    from hyper_amd.diversity.model_kernels import build_model_kernel
    dev = torch.device('cuda:0')
    cfg, feats = _case(name)
    xs = [f.to(dev).requires_grad_(True) for f in feats]
    ref = build_model_kernel(cfg).to(dev)(None, OrderedDict((str(i), x) for i, x in enumerate(xs)), None, OrderedDict())
    ref.backward()
    out = str(tmp_path / 'res')
    port = 29950 + (os.getpid() + 11 * world + len(name) + len(mode)) % 200
    mp.start_processes(_worker, args=(world, port, name, out, mode), nprocs=world, join=True, start_method='spawn')
    Ml = feats[0].shape[0] // world
    for r in range(world):
        res = torch.load('%s.%d' % (out, r), weights_only=True)
        assert abs(res['loss'].item() - ref.item()) <= RTOL * abs(ref.item())
        for g, x in zip(res['grads'], xs):
            assert _relerr(g.numpy(), x.grad[r * Ml:(r + 1) * Ml].cpu().numpy()) <= RTOL


TRAIN_CFG = {
    'target': {'name': 'lenet5', 'in_channels': 1, 'num_classes': 10},
    'method': {'name': 'repulsive_kernel', 'gamma': 10.5, 'warmup': 600, 'l2_reg': 0.05, 'ind_loss': {'name': 'cross_entropy'},
               'model_kernel': {'name': 'model_layers',
                                'layer_weighting': {'name': 'start_end_linear', 'first_layer': 0.387, 'middle_start': 0.746,
                                                    'middle_increase': 1.267, 'last_layer': 3.63},
                                'layer_kernel': {'name': 'he', 'eps': 0.0025, 'arc_eps': 0.031,
                                                 'feature_kernel': {'name': 'feature', 'detach_diag': True, 'kernel': 'linear'}}}},
}
TRAIN_M, TRAIN_B = 6, 16


def _train_objects(members, seed_members):
    """Ensemble holding `seed_members` (global member ids: every member's weights depend on its id only)."""
    import copy
    import hyper_amd.target  # noqa: F401
    from hyper_amd.generators import build_generator
    from hyper_amd.diversity.methods import build_method
    torch.manual_seed(123)
    full = build_generator({'name': 'ensemble', 'ensemble_size': TRAIN_M, 'target': copy.deepcopy(TRAIN_CFG['target'])})
    part = build_generator({'name': 'ensemble', 'ensemble_size': members, 'target': copy.deepcopy(TRAIN_CFG['target'])})
    sd = full.state_dict()
    part.load_state_dict({k: v[seed_members] for k, v in sd.items()})
    method = build_method(dict(copy.deepcopy(TRAIN_CFG['method']), num=members))
    gen = torch.Generator().manual_seed(3)
    X = torch.randn(TRAIN_B, 1, 28, 28, generator=gen)
    Y = torch.randint(0, 10, (TRAIN_B,), generator=gen)
    return part, method, X, Y


def _train_step(hyper, method, X, Y, members):
    import types
    track = hyper(params=list(range(members)), x=X, sample_params=True, ret_params=True, feature_split=True, ood_N=0)
    track = method(types.SimpleNamespace(training=True), track['params'], [], track, X, Y, step=700)
    track['loss'].backward()
    return track


def _train_worker(rank, world, port, out):
    sys.path[:0] = [ROOT, os.path.join(ROOT, 'he-cka-ensembles_amd'), os.path.join(ROOT, 'tests')]
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from hyper_amd.parallel import shard_method
    dev = torch.device('cuda:0')
    ml = TRAIN_M // world
    ids = list(range(rank * ml, (rank + 1) * ml))
    hyper, method, X, Y = _train_objects(ml, ids)
    hyper, method = hyper.to(dev), shard_method(method.to(dev))
    track = _train_step(hyper, method, X.to(dev), Y.to(dev), ml)
    torch.cuda.synchronize()
    torch.save({'rep': track['ind_rep_loss'].detach().cpu(), 'grads': {k: v.grad.cpu() for k, v in hyper.named_parameters()}},
               '%s.%d' % (out, rank))
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_member_sharded_training_step_matches_unsharded(world, tmp_path):
    """each rank trains ITS members (LeNet-5, full batch) with `shard_method`; the gradients it leaves on
    its parameters equal the unsharded six-member step's gradients of those members, the repulsion value is the global one."""
    import torch.multiprocessing as mp
    dev = torch.device('cuda:0')
    hyper, method, X, Y = _train_objects(TRAIN_M, list(range(TRAIN_M)))
    hyper, method = hyper.to(dev), method.to(dev)
    ref = _train_step(hyper, method, X.to(dev), Y.to(dev), TRAIN_M)
    want = {k: v.grad.cpu() for k, v in hyper.named_parameters()}
    out = str(tmp_path / 'train')
    port = 29700 + (os.getpid() + 7 * world) % 200
    mp.start_processes(_train_worker, args=(world, port, out), nprocs=world, join=True, start_method='spawn')
    ml = TRAIN_M // world
    for r in range(world):
        res = torch.load('%s.%d' % (out, r), weights_only=True)
        assert abs(res['rep'].item() - ref['ind_rep_loss'].item()) <= RTOL * abs(ref['ind_rep_loss'].item())
        for k, g in res['grads'].items():
            assert _relerr(g.numpy(), want[k][r * ml:(r + 1) * ml].numpy()) <= 2e-5, (r, k)
