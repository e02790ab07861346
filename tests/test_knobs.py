"""Every HECKA_* knob that survives in hyper_amd/config.py / libhecka's Config has ONE test that holds its non-default
value to the default's results (INTEGRATION.md 6 names the test per knob).  The knobs other test files already switch
(HECKA_K32, HECKA_RAGGED_X3, HECKA_X16_EXACT, HECKA_NATIVE16, HECKA_GRAM_NOSPLIT, HECKA_BWD_MULTI, HECKA_MULTI_GRAM,
HECKA_FUSED_LAYERS, HECKA_CONV_X3, HECKA_CONV_DIRECT, HECKA_BN_FUSE, HECKA_SHARD_MODE, HECKA_SHARD_EXCHANGE) are not
repeated here.  The environment is read once; tests/conftest.py re-reads it whenever a test changes a HECKA_* variable."""
import copy
import os
from collections import OrderedDict

import pytest
import torch


def test_config_is_read_once_and_reloaded_on_request(monkeypatch):
    """No knob is looked up on the call path: a changed environment shows only after config.reload()."""
    from hyper_amd import config
    assert config.knobs.bwd_multi is True
    os.environ['HECKA_BWD_MULTI'] = '0'          # behind the fixture's back: nothing re-reads
    try:
        assert config.knobs.bwd_multi is True
        config.reload()
        assert config.knobs.bwd_multi is False
    finally:
        del os.environ['HECKA_BWD_MULTI']
        config.reload()
    assert config.knobs.bwd_multi is True
    monkeypatch.setenv('HECKA_SHARD_MODE', 'gather')      # through the fixture: re-read at once
    assert config.knobs.shard_mode == 'gather'
    import subprocess
    import sys
    code = ("import sys; sys.path[:0] = %r; import os; os.environ['HECKA_SHARD_EXCHANGE'] = 'bogus'\n"
            "try:\n    import hyper_amd.config\nexcept ValueError as e:\n    print('refused:', e)" %
            ([os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'he-cka-ensembles_amd')],))
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True)
    assert 'refused' in out.stdout, out.stdout + out.stderr


def _loss_and_grads(M, B, Ps, seed=3, kernel='linear'):
    from hyper_amd.diversity.model_kernels import build_model_kernel
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(seed)
    xs = [(torch.randn(M, B, p, generator=g) * (1 + 0.5 * i)).to(dev).requires_grad_(True) for i, p in enumerate(Ps)]
    cfg = {'name': 'model_layers', 'reduction': 'mean', 'layer_weighting': {'name': 'uniform'},
           'layer_kernel': {'name': 'he', 'eps': 0.0025, 'arc_eps': 0.031,
                            'feature_kernel': {'name': 'feature', 'detach_diag': True, 'kernel': kernel}}}
    loss = build_model_kernel(cfg).to(dev)(None, OrderedDict((str(i), x) for i, x in enumerate(xs)), None, OrderedDict())
    loss.backward()
    return float(loss), [x.grad.clone() for x in xs]


@pytest.mark.gpu
@pytest.mark.parametrize('knob,value,shape,tol', [
    ('HECKA_COMBINE_FIXG', '0', (64, 256, (512, 256)), 0.0),        # column terms re-loaded per chunk: same sums, bit-equal
    ('HECKA_COLMEAN_MULTI', '0', (64, 256, (512, 256)), 0.0),       # one column-mean launch per layer: bit-equal
    ('HECKA_COMBINE_X3', '0', (64, 256, (512, 256)), 2e-6),         # all-layers combine on fp32 MFMAs instead of fp16 splits
    ('HECKA_BWD_TN', '128', (64, 256, (512, 256)), 2e-6),           # four-wave 256 x 128 dX tiles: another summation order
    ('HECKA_PAIRWISE_X6', '1', (64, 256, (512, 256)), 5e-6),        # bf16 six-product pairwise: inside the parity budget
    ('HECKA_SMALL_GRID', '1', (5, 300, (864, 256)), 1e-5),          # ragged small grid forced onto 128-row fp32-MFMA tiles
    ('HECKA_FUSED_RAGGED', '0', (5, 300, (864, 256)), 2e-6),        # ragged layers each in their own autograd node
])
def test_knob_alternative_matches_default(knob, value, shape, tol, monkeypatch):
    M, B, Ps = shape
    base_loss, base_grads = _loss_and_grads(M, B, Ps)
    monkeypatch.setenv(knob, value)
    loss, grads = _loss_and_grads(M, B, Ps)
    if tol == 0.0:
        assert loss == base_loss
        for a, b in zip(grads, base_grads):
            assert torch.equal(a, b)
    else:
        assert abs(loss - base_loss) <= tol * abs(base_loss)
        for a, b in zip(grads, base_grads):
            assert float((a - b).abs().max()) <= tol * float(b.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize('M,B,Ps,grid', [(16, 512, (2048, 1024, 512), '1'), (16, 256, (1024, 512, 256), '8'), (5, 256, (768, 256), '16'),
                                         (3, 1024, (512,), '8'), (8, 512, (256,), '1')])
def test_streamed_dx_launch_is_bit_equal_to_a_workgroup_per_tile(M, B, Ps, grid, monkeypatch):
    """HECKA_BWD_STREAM: the all-layers dX launch as one workgroup per CU walking the tiles (the k-tiles of consecutive
    tiles in one pipeline) against a workgroup per tile - the same products in the same order, so bit for bit; grids capped
    at 8 / 16 workgroups make every workgroup cross many tile and layer boundaries (odd member counts: no XCD map); a
    closed gate writes exact zeros."""
    from hyper_amd import ops
    dev = torch.device('cuda:0')
    torch.manual_seed(M + B)
    Xs = [torch.randn(M, B, p, device=dev) * (1 + i) for i, p in enumerate(Ps)]
    Ws = [torch.randn(M, B, B, device=dev) * (0.5 + torch.rand(M, 1, 1, device=dev)) for _ in Ps]
    wmaxs = [W.abs().amax(dim=(1, 2)).contiguous() for W in Ws]
    xmaxs = [X.abs().amax(dim=(1, 2)).contiguous() for X in Xs]
    monkeypatch.setenv('HECKA_BWD_STREAM', '0')
    want = ops.gram_bwd_multi(Ws, Xs, wmaxs, xmaxs)
    monkeypatch.setenv('HECKA_BWD_STREAM', grid)
    got = ops.gram_bwd_multi(Ws, Xs, wmaxs, xmaxs)
    for a, b, W, X in zip(want, got, Ws, Xs):
        assert torch.equal(a, b)
        ref = W.double() @ X.double()
        assert float((b.double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
    closed = torch.zeros(1, device=dev)
    for g in ops.gram_bwd_multi(Ws, Xs, wmaxs, xmaxs, gate=closed):
        assert not bool(g.any())


@pytest.mark.gpu
@pytest.mark.parametrize('knob,value,stride', [('HECKA_CONV_IMPLICIT', '0', 1), ('HECKA_CONV_KEEP_COLS_GB', '0', 1),
                                               ('HECKA_CONV_TAPMAJOR', '0', 1), ('HECKA_CONV_TAPMAJOR', '0', 2),
                                               ('HECKA_WGRAD_QUAD', '0', 1)])
def test_conv_knob_alternative_matches_default(knob, value, stride, monkeypatch):
    """Implicit-GEMM convolution off (explicit im2col / col2im for every conv) / gathered patches never kept for backward /
    the implicit GEMMs with K ordered (channel, tap) - a range check per gathered element - instead of (tap, channel) / the
    implicit weight gradient gathering its patches element by element instead of four consecutive pixels per load."""
    from hyper_amd.layers.conv import BatchedConv2d
    dev = torch.device('cuda:0')
    torch.manual_seed(5)
    M, N, C, HW = 2, 16, 128, 16
    x = torch.randn(M, N, C, HW, HW, device=dev)
    W = torch.randn(M, C, C, 3, 3, device=dev) * 0.05
    up = torch.randn(M, N, C, HW // stride, HW // stride, device=dev)
    geom = (3, 3, stride, stride, 1, 1, 1, 1)
    res = []
    for setting in (None, value):
        if setting is not None:
            monkeypatch.setenv(knob, setting)
        xx, WW = x.clone().requires_grad_(True), W.clone().requires_grad_(True)
        y = BatchedConv2d.apply(xx, WW, None, 0.7, 1.0, geom)
        (y * up).sum().backward()
        res.append((y.detach(), xx.grad, WW.grad))
    for a, b in zip(*res):
        assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max())


@pytest.mark.gpu
def test_sampler_torch_pieces_match_the_hip_ones(monkeypatch):
    """HECKA_SAMPLER_TORCH=linear,groupnorm,matvec: the generator MLP's pieces as torch expressions (A/B handle)."""
    from hyper_amd.net.models import build_norm_block
    from hyper_amd.net import hip
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    blk = build_norm_block(24, 64, bias=True, act=torch.nn.GELU, norm=True, gamma=None).to(dev)
    x = torch.randn(7, 24, device=dev)
    Wm, v = torch.randn(48, 40, device=dev), torch.randn(40, device=dev)
    outs = []
    for setting in (None, 'linear,groupnorm,matvec'):
        if setting is not None:
            monkeypatch.setenv('HECKA_SAMPLER_TORCH', setting)
        b = copy.deepcopy(blk)
        xx = x.clone().requires_grad_(True)
        y = b(xx)
        y.square().sum().backward()
        outs.append([y.detach(), xx.grad] + [p.grad for p in b.parameters()] + [hip.matvec(Wm, v)])
    for a, b in zip(*outs):
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize('M,B,Ps,rows', [(64, 512, (1024, 512), 0), (8, 512, (4096, 2048, 256), 64), (16, 256, (512, 256), 0),
                                         (4, 768, (256, 128), 0), (3, 1024, (256, 128), 256), (5, 128, (256, 64), 0)])
def test_paired_gram_tiles_match_the_square_ones(M, B, Ps, rows, monkeypatch):
    """HECKA_GRAM_PAIR: the all-layers Gram on paired 128 x 256 tiles (two blocks of a block row per workgroup, the
    diagonal block's rows read from its neighbour's panel) against the 128 x 128 tiles and against fp64 products - block
    counts 1 (no pairing: the square tiles), 2 (one tile computed twice), 4 (the bench), 6, 8; split over K (8 members); the
    slice-major epilogue."""
    from hyper_amd import ops
    dev = torch.device('cuda:0')
    torch.manual_seed(M * B)
    Xs = [(torch.randn(M, B, p, device=dev) + 0.3 * i) * (1 + i) for i, p in enumerate(Ps)]
    pair = ops.centered_gram_multi(Xs, slice_rows=rows)
    monkeypatch.setenv('HECKA_GRAM_PAIR', '0')
    square = ops.centered_gram_multi(Xs, slice_rows=rows)
    for X, (C, cvec, xmax), (C0, cvec0, xmax0) in zip(Xs, pair, square):
        assert torch.equal(xmax, xmax0)
        assert float((C - C0).abs().max()) <= 2e-6 * float(C0.abs().max())
        assert float((cvec - cvec0).abs().max()) <= 2e-6 * float(cvec0.abs().max())
        Xd = X.double()
        Kd = Xd @ Xd.transpose(1, 2)
        Cd = Kd - Kd.mean(dim=2, keepdim=True)
        if rows:
            Cd = Cd.view(M, B // rows, rows, B).permute(1, 0, 2, 3)
        assert float((C.double() - Cd).abs().max()) <= 5e-6 * float(Cd.abs().max())
