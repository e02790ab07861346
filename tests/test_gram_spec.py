"""The speculative Gram (hecka_gram_center_spec_multi_f32, include/hecka.h): the all-layers centred Gram WITHOUT the
column-mean pass over the features - a sampled shift t in place of the means, the Gram kernel on x - t, then a row-centring
of the Grams (C = Chat - rowmean(Chat) holds for any t), with the exact two-pass path enqueued behind it on a device flag.
Held to the fp64 value of the reference's formula (center_gram(linear_gram(X)), hyper/diversity/kernels.py:107-127,
335-345) and to the exact path (HECKA_GRAM_SPEC=0); the fallback is driven by features whose tail the sample misses."""
from collections import OrderedDict

import pytest
import torch


def _fp64(X):
    Xd = X.double()
    Kd = Xd @ Xd.transpose(1, 2)
    Cd = Kd - Kd.mean(dim=2, keepdim=True)
    sbar = Xd.mean(dim=1, keepdim=True)
    cd = ((Xd - sbar) * sbar).sum(dim=2)
    return Cd, cd


def _flag():
    from hyper_amd import ops
    ws, idx = ops.LAST_SPEC_WS
    return int(ws.view(torch.int32)[idx].item())


CASES = [
    (64, 512, (1024, 512), 'randn'),        # the bench's member / sample counts
    (256, 256, (512, 256), 'relu'),         # two blocks (one pair tile computed twice), post-ReLU features
    (128, 384, (640, 320, 96), 'mean30'),   # three blocks (B not a multiple of 256), features with a mean 30 x their spread
    (16, 1024, (256, 128, 64, 32), 'decades'),   # eight blocks; members six decades apart
]


@pytest.mark.gpu
@pytest.mark.parametrize('M,B,Ps,kind', CASES)
def test_speculative_gram_matches_fp64_and_the_exact_path(M, B, Ps, kind, monkeypatch):
    import ctypes
    from hyper_amd import _lib, ops
    dev = torch.device('cuda:0')
    torch.manual_seed(M + B)
    Xs = []
    for i, p in enumerate(Ps):
        X = torch.randn(M, B, p, device=dev) * (1 + 0.5 * i)
        if kind == 'relu':
            X = torch.relu(X + 0.5)
        elif kind == 'mean30':
            X = X + 30.0 * (1 + 0.5 * i) * torch.randn(1, 1, p, device=dev).sign()
        elif kind == 'decades':
            X = X * (10.0 ** torch.linspace(-3, 3, M, device=dev)).view(M, 1, 1)
        Xs.append(X.contiguous())
    monkeypatch.setenv('HECKA_GRAM_SPEC', '2')      # wherever the shape allows it, not only where it pays (deep layers)
    order = sorted(range(len(Ps)), key=lambda i: -Ps[i])
    Pc = (ctypes.c_int * len(Ps))(*[Ps[i] for i in order])
    assert _lib.lib().hecka_gram_center_spec_eligible(len(Ps), M, B, Pc) == 1
    ops.LAST_SPEC_WS = None
    spec = ops.centered_gram_multi(Xs)
    assert ops.LAST_SPEC_WS is not None and _flag() == 0          # the speculative path ran and kept its result
    monkeypatch.setenv('HECKA_GRAM_SPEC', '0')
    ops.LAST_SPEC_WS = None
    exact = ops.centered_gram_multi(Xs)
    assert ops.LAST_SPEC_WS is None
    for X, (C, cvec, xmax), (C0, cvec0, xmax0) in zip(Xs, spec, exact):
        Cd, cd = _fp64(X)
        cs = Cd.abs().amax(dim=(1, 2), keepdim=True)                 # per member: members may lie decades apart
        err = float(((C.double() - Cd).abs() / cs).max())
        err0 = float(((C0.double() - Cd).abs() / cs).max())
        assert err <= max(2e-6, 1.5 * err0), (err, err0)            # as close to fp64 as the exact two-pass path
        # cvec is the column offset inside C (C = Kt + 1 cvec^T) and the speculative one inherits the Gram's own arithmetic
        # error through the row means: held to C's scale (as C is), and to its own scale with the slack that implies for
        # zero-mean features, whose cvec is small beside the terms it is formed from
        dv = (cvec.double() - cd).abs()
        assert float((dv / cs.view(-1, 1)).max()) <= 1e-6
        assert float((dv / cd.abs().amax(dim=1, keepdim=True).clamp_min(1e-30)).max()) <= 1e-4
        rowmean = C.mean(dim=2).abs().amax(dim=1)                    # what the row-centring is for
        assert float((rowmean / cs.view(-1)).max()) <= 1e-6
        assert bool((xmax >= xmax0).all()) and bool((xmax <= 2.5 * xmax0 + 1e-30).all())   # a bound on max |X_m|, not loose


@pytest.mark.gpu
@pytest.mark.parametrize('poison', [1e4, float('inf')])
def test_speculative_gram_falls_back_when_the_sample_misses_the_tail(poison, monkeypatch):
    """One feature far beyond 16 x the sampled maximum in a row the sample does not read (row 1; the sample rows are
    multiples of B / 16 = 32): the fp16 high plane would overflow.  The flag goes up on the device and the exact path's
    result - bit for bit - is what the call leaves; an infinite feature takes the same route."""
    from hyper_amd import ops
    dev = torch.device('cuda:0')
    torch.manual_seed(7)
    M, B, Ps = 64, 512, (512, 256, 128, 64)
    Xs = [torch.randn(M, B, p, device=dev) for p in Ps]
    Xs[1][5, 100, 17] = poison
    monkeypatch.setenv('HECKA_GRAM_SPEC', '2')
    ops.LAST_SPEC_WS = None
    spec = ops.centered_gram_multi(Xs)
    assert ops.LAST_SPEC_WS is not None and _flag() == 1
    monkeypatch.setenv('HECKA_GRAM_SPEC', '0')
    exact = ops.centered_gram_multi(Xs)
    for (C, cvec, xmax), (C0, cvec0, xmax0) in zip(spec, exact):
        assert torch.equal(xmax, xmax0)
        assert torch.equal(C.nan_to_num(nan=123.0), C0.nan_to_num(nan=123.0))
        assert torch.equal(cvec.nan_to_num(nan=123.0), cvec0.nan_to_num(nan=123.0))
    if poison == 1e4:
        Cd, _ = _fp64(Xs[1])
        assert float((spec[1][0].double() - Cd).abs().max()) <= 5e-6 * float(Cd.abs().max())


@pytest.mark.gpu
def test_speculative_gram_propagates_nan_like_the_exact_path(monkeypatch):
    """A NaN feature needs no fallback (v_max drops NaNs, so the flag stays down): it poisons its member's whole Gram on
    either path - through the column mean on the exact one, through the row means here - and no other member's."""
    from hyper_amd import ops
    dev = torch.device('cuda:0')
    torch.manual_seed(8)
    Xs = [torch.randn(64, 512, p, device=dev) for p in (512, 256, 128, 64)]
    Xs[2][9, 3, 5] = float('nan')
    monkeypatch.setenv('HECKA_GRAM_SPEC', '2')
    ops.LAST_SPEC_WS = None
    spec = ops.centered_gram_multi(Xs)
    assert ops.LAST_SPEC_WS is not None and _flag() == 0
    monkeypatch.setenv('HECKA_GRAM_SPEC', '0')
    exact = ops.centered_gram_multi(Xs)
    for li, ((C, cvec, _), (C0, cvec0, _)) in enumerate(zip(spec, exact)):
        bad = torch.isnan(C).flatten(1).all(dim=1)
        assert torch.equal(bad, torch.isnan(C0).flatten(1).all(dim=1))
        assert torch.equal(bad, torch.isnan(C).flatten(1).any(dim=1))          # all or nothing per member
        assert bad.tolist() == [li == 2 and m == 9 for m in range(64)]
        assert torch.equal(torch.isnan(cvec).any(dim=1), bad) and torch.equal(torch.isnan(cvec0).any(dim=1), bad)


@pytest.mark.gpu
def test_loss_and_gradients_through_the_speculative_gram(monkeypatch):
    """End to end through build_model_kernel: loss and every dX with the speculative Gram against the exact path (both
    inside 1e-5 of each other - the bench and the parity tests hold the default path to the fp64 oracle)."""
    from hyper_amd.diversity.model_kernels import build_model_kernel
    dev = torch.device('cuda:0')
    cfg = {'name': 'model_layers', 'reduction': 'mean', 'layer_weighting': {'name': 'uniform'},
           'layer_kernel': {'name': 'he', 'eps': 0.0025, 'arc_eps': 0.031,
                            'feature_kernel': {'name': 'feature', 'detach_diag': True, 'kernel': 'linear'}}}
    g = torch.Generator().manual_seed(11)
    base = [torch.relu(torch.randn(64, 512, p, generator=g) + 0.3) for p in (768, 512, 256, 128)]

    def run():
        xs = [b.to(dev).requires_grad_(True) for b in base]
        loss = build_model_kernel(cfg).to(dev)(None, OrderedDict((str(i), x) for i, x in enumerate(xs)), None, OrderedDict())
        loss.backward()
        return float(loss.detach()), [x.grad.clone() for x in xs]

    from hyper_amd import ops
    monkeypatch.setenv('HECKA_GRAM_SPEC', '2')
    ops.LAST_SPEC_WS = None
    l1, g1 = run()
    assert ops.LAST_SPEC_WS is not None and _flag() == 0
    monkeypatch.setenv('HECKA_GRAM_SPEC', '0')
    l0, g0 = run()
    assert abs(l1 - l0) <= 2e-6 * abs(l0)
    for a, b in zip(g1, g0):
        assert float((a - b).abs().max()) <= 5e-6 * float(b.abs().max())


@pytest.mark.gpu
def test_speculative_gram_replays_in_a_hip_graph_including_the_fallback(monkeypatch):
    """The call is stream-ordered and holds no host decision: captured once, the same graph gives the speculative result for
    well-behaved features and the exact path's (flag raised on the device, gated launches) when the replayed input has an
    outlier the sample does not see."""
    from hyper_amd import ops
    dev = torch.device('cuda:0')
    torch.manual_seed(31)
    M, B, Ps = 64, 512, (512, 256, 128, 64)
    monkeypatch.setenv('HECKA_GRAM_SPEC', '2')
    good = [torch.randn(M, B, p, device=dev) for p in Ps]
    bad = [g.clone() for g in good]
    bad[0][11, 77, 9] = 3e4
    static = [g.clone() for g in good]
    ops.centered_gram_multi(static)                    # warm-up outside the capture (module load, allocator)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = ops.centered_gram_multi(static)
    ws, idx = ops.LAST_SPEC_WS
    for src, want_flag in ((good, 0), (bad, 1), (good, 0)):
        for s, x in zip(static, src):
            s.copy_(x)
        graph.replay()
        torch.cuda.synchronize()
        assert int(ws.view(torch.int32)[idx].item()) == want_flag
        for X, (C, cvec, xmax) in zip(src, out):
            Cd, _ = _fp64(X)
            assert float((C.double() - Cd).abs().max()) <= 5e-6 * float(Cd.abs().max())
            assert bool((xmax >= X.abs().amax(dim=(1, 2))).all())


def test_speculative_gram_eligibility_is_a_host_decision():
    """hecka_gram_center_spec_eligible needs no GPU: deep layers on unsplit paired-tile grids only (it must pay for the
    row-centring pass and ~60 us of launches), never for ragged / small / split shapes."""
    import ctypes
    from hyper_amd import _lib
    L = _lib.lib()

    def elig(M, B, Ps):
        return L.hecka_gram_center_spec_eligible(len(Ps), M, B, (ctypes.c_int * len(Ps))(*Ps))
    assert elig(64, 512, (8192, 4096, 2048, 1024, 512)) == 1          # the bench
    assert elig(64, 512, (1024, 512)) == 0                            # shallow: the column-mean pass is cheaper
    assert elig(8, 512, (8192, 4096, 2048, 1024, 512)) == 0           # a member shard: the launch is split over K
    assert elig(5, 300, (864, 256)) == 0 and elig(64, 128, (8192, 4096)) == 0      # ragged / a single block
    assert elig(64, 512, (512, 1024)) == 0                            # layers must come deepest first
    assert L.hecka_gram_center_spec_workspace_bytes(5, 64, 512, (ctypes.c_int * 5)(8192, 4096, 2048, 1024, 512)) > 0
