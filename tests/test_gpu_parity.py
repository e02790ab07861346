"""GPU parity: the HIP path (through the C-ABI, via hyper_amd) against the golden vectors of the real
reference and against the CPU oracle on the same seeded inputs.

Tolerance (BASELINE.json north_star): rel 1e-5 in fp32 for loss and gradients.  Gradients are compared
relative to the largest gradient magnitude of the layer (max-norm relative error)."""
import numpy as np
import pytest
import torch

from conftest import Yardstick, golden_cases, load_golden

pytestmark = pytest.mark.gpu

RTOL = 1e-5


def _dev():
    assert torch.cuda.is_available(), 'GPU tests need a GPU'
    return torch.device('cuda:0')


LINEAR = golden_cases('f64')   # every golden configuration (linear, cossim and rbf Gram kernels)


def _relerr(a, b):
    return float(np.abs(np.asarray(a, dtype=np.float64) - b).max() / max(np.abs(b).max(), 1e-300))


@pytest.mark.parametrize('name', LINEAR)
def test_model_kernel_matches_reference_golden(name):
    """hyper_amd.build_model_kernel(cfg)(feats) on the GPU vs the reference's own fp64 output."""
    from hyper_amd.diversity.model_kernels import build_model_kernel
    from collections import OrderedDict
    rec = load_golden(name, 'f64')
    dev = _dev()
    kern = build_model_kernel(rec['config']).to(dev)
    xs = [torch.from_numpy(rec['x%d' % i]).float().to(dev).requires_grad_(True) for i in range(rec['L'])]
    # golden inputs are fp64; the GPU path sees their fp32 rounding, so compare against the oracle run
    # on exactly those rounded inputs in fp64 (and, loosely, against the golden fp64 outputs)
    from oracle import hecka_ref as R
    xo = [x.detach().cpu().double().requires_grad_(True) for x in xs]
    ref = R.run_config(rec['config'], xo)
    track = OrderedDict()
    out = kern(None, OrderedDict((str(i), x) for i, x in enumerate(xs)), None, track)
    # against the fixture itself (fp64 reference on the UNROUNDED inputs) the yardstick is the reference's own float32 run
    # of the same case (<name>_f32.npz): its deviation from the fp64 run is what float32 inputs + arithmetic cost
    f32 = load_golden(name, 'f32')
    ref32 = lambda key: max(RTOL, _relerr(f32[key], rec[key]))   # noqa: E731
    if name == 'he_lin_none':
        # reduction 'none' keeps the diagonal pairs (i, i): cos = n^2 / (n + eps)^2 -> theta = arccos(1 - O(eps)) and the
        # float32 autograd coefficient 1/d^2 - n/d^3 cancels - ill-conditioned in the reference itself.  Off-diagonal
        # entries and the gradient that flows through them are held to 1e-5; the diagonal to the reference's own
        # float32 deviation.
        M = out.shape[0]
        off = ~torch.eye(M, dtype=torch.bool)
        up = torch.from_numpy(rec['upstream'])
        o, r = out.detach().cpu().double(), ref.detach()
        assert float((o - r)[off].abs().max() / r[off].abs().max()) <= RTOL
        assert float((o - r)[~off].abs().max() / r[~off].abs().max()) <= ref32('out')
        (out * (up * off).float().to(dev)).sum().backward(retain_graph=True)
        (ref * (up * off)).sum().backward(retain_graph=True)
        for i, x in enumerate(xs):
            assert _relerr(x.grad.cpu().numpy(), xo[i].grad.numpy()) <= RTOL, 'layer %d (off-diagonal upstream)' % i
            x.grad = None
            xo[i].grad = None
        (out * up.float().to(dev)).sum().backward()
        (ref * up).sum().backward()
        for i, x in enumerate(xs):
            assert _relerr(x.grad.cpu().numpy(), xo[i].grad.numpy()) <= ref32('dx%d' % i), 'layer %d' % i
        return
    if out.ndim == 0:
        out.backward()
        ref.backward()
    else:
        up = torch.from_numpy(rec['upstream'])
        (out * up.float().to(dev)).sum().backward()
        (ref * up).sum().backward()
    assert _relerr(out.detach().cpu().numpy(), ref.detach().numpy()) <= RTOL
    assert _relerr(out.detach().cpu().numpy(), rec['out']) <= ref32('out')
    for i, x in enumerate(xs):
        assert _relerr(x.grad.cpu().numpy(), xo[i].grad.numpy()) <= RTOL, 'layer %d' % i
        if 'gram%d' % i in rec:
            g = track['layer_track'][i]['gram'].cpu().numpy()
            assert _relerr(g, rec['gram%d' % i]) <= ref32('gram%d' % i)


@pytest.mark.parametrize('M,B,P', [(3, 37, 19), (5, 300, 84), (2, 130, 10), (4, 128, 256), (2, 257, 515), (1, 8, 4),
                                   (5, 300, 864), (30, 100, 40), (128, 400, 256), (3, 132, 20), (2, 1028, 36)])
@pytest.mark.parametrize('x3', ['auto', 'split'])
def test_kernels_ragged_shapes(M, B, P, x3, monkeypatch):
    """Each C-ABI entry point at ragged / unaligned sizes against fp64 torch on the CPU.  The reference's own batch sizes
    (configs/mnist/base.yml:11,17: 300 (+100 ood); configs/toy/fourclass/base.yml: 100) are ragged: with B % 4 == 0 (and
    P % 4 == 0 for the Gram) they run on the split-precision tile with masked loads / epilogue when its grid fills the
    GPU (x3 = 'auto': the library's rule; 'split': HECKA_RAGGED_X3=1 puts every eligible shape on it), otherwise fp32 MFMA."""
    if x3 == 'split':
        if B % 4:
            pytest.skip('no whole quads: fp32 MFMA only')
        monkeypatch.setenv('HECKA_RAGGED_X3', '1')
    from hyper_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(M * 1000 + B + P)
    X = (torch.randn(M, B, P, generator=g) + 0.5).to(dev)
    Xd = X.cpu().double()
    C, cvec = ops.centered_gram(X)
    K = Xd @ Xd.transpose(1, 2)
    Cref = K - K.mean(2, keepdim=True)
    assert _relerr(C.cpu().numpy(), Cref.numpy()) <= 2e-6
    S = ops.pairwise_inner(C)
    Cf = Cref.reshape(M, -1)
    assert _relerr(S.cpu().numpy(), (Cf @ Cf.t()).numpy()) <= 5e-6
    A = torch.randn(M, M, generator=g).to(dev)
    dX = ops.gram_backward(A, C, cvec, X, True)
    dC = torch.einsum('ij,jab->iab', A.cpu().double(), Cref)
    W = (dC + dC.transpose(1, 2)) * (1 - torch.eye(B, dtype=torch.float64))
    assert _relerr(dX.cpu().numpy(), (W @ Xd).numpy()) <= 1e-5
    C2, cvec2, xmax = ops.centered_gram(X, True)          # with xmax the dX GEMM may take the fp16-split tile
    cref = ((Xd - Xd.mean(1, keepdim=True)) * Xd.mean(1, keepdim=True)).sum(2)
    assert _relerr(cvec2.cpu().numpy(), cref.numpy()) <= 1e-5
    dX2 = ops.gram_backward(A, C2, cvec2, X, True, xmax)
    assert _relerr(dX2.cpu().numpy(), (W @ Xd).numpy()) <= 1e-5
    # un-fused functional pieces
    Kg = ops.linear_gram(X.clone().requires_grad_(True), True)
    assert _relerr(Kg.detach().cpu().numpy(), K.numpy()) <= 2e-6
    Cc = ops.center_rows(Kg.detach())
    assert _relerr(Cc.cpu().numpy(), Cref.numpy()) <= 1e-5


def test_unfused_api_matches_fused():
    """batched_gram + hyperspherical_energy (un-fused functional API) == fused model kernel."""
    from hyper_amd.diversity import hyperspherical_energy
    from hyper_amd.diversity.kernels import batched_gram
    from hyper_amd import ops
    from hyper_amd.diversity import cos_from_inner, energy_from_cos
    dev = _dev()
    torch.manual_seed(5)
    X = torch.randn(6, 48, 40, device=dev)
    xa = X.clone().requires_grad_(True)
    e1 = hyperspherical_energy(batched_gram(xa, None, kernel='linear', detach_diag=True, center=True),
                               s=2.0, eps=0.0025, arc_eps=0.031).mean()
    e1.backward()
    xb = X.clone().requires_grad_(True)
    S, nsq, _ = ops.centered_gram_pairwise(xb, True, False)
    e2 = energy_from_cos(cos_from_inner(S, nsq, 0.0025), 2.0, 0.0025, 0.031).mean()
    e2.backward()
    assert abs(e1.item() - e2.item()) <= 1e-5 * abs(e2.item())
    assert _relerr(xa.grad.cpu().numpy(), xb.grad.cpu().double().numpy()) <= 2e-5


def test_bench_shape_properties():
    """Full bench sizes (M=64, B=512): size-independent properties instead of an oracle run:
    rows of C sum to zero, diag(S) = |C_m|^2, S symmetric, backward linear in its coefficients."""
    from hyper_amd import ops
    dev = _dev()
    torch.manual_seed(1234)
    M, B, P = 64, 512, 512
    X = torch.randn(M, B, P, device=dev)
    C, cvec = ops.centered_gram(X)
    assert float(C.sum(2).abs().max()) <= 1e-3 * float(C.abs().max())
    S = ops.pairwise_inner(C)
    nsq = (C.double() ** 2).sum((1, 2))
    assert _relerr(torch.diagonal(S).cpu().numpy(), nsq.cpu().numpy()) <= 1e-5
    assert float((S - S.t()).abs().max()) <= 1e-5 * float(S.abs().max())

    # linearity of the backward in the upstream coefficients: dX(A1 + 2 A2) = dX(A1) + 2 dX(A2)
    g = torch.Generator().manual_seed(7)
    A1, A2 = torch.randn(M, M, generator=g).to(dev), torch.randn(M, M, generator=g).to(dev)
    d12 = ops.gram_backward((A1 + 2 * A2).contiguous(), C, cvec, X, True)
    d1, d2 = ops.gram_backward(A1, C, cvec, X, True), ops.gram_backward(A2, C, cvec, X, True)
    assert float((d12 - d1 - 2 * d2).abs().max()) <= 2e-5 * float(d12.abs().max())


@pytest.mark.parametrize('M,B,P', [(8, 256, 80), (16, 128, 48), (3, 128, 144), (5, 384, 1024), (64, 512, 512),
                                   (5, 300, 864), (30, 100, 40), (128, 400, 256)])
def test_gemm_modes_against_fp64(M, B, P, monkeypatch):
    """Interior shapes run the two large GEMMs on the 16-bit matrix cores with exactly split fp32 operands
    (HECKA_GEMM_AUTO: f16x3, HECKA_GEMM_BF16X6); HECKA_GEMM_FP32 forces fp32 MFMA.  All against fp64 torch on
    the GPU, the split paths to the SAME tolerance as the fp32 one (2e-6 of the largest entry for the Gram,
    1e-5 for dX)."""
    from hyper_amd import ops, _lib
    dev = _dev()
    if B % 128:
        monkeypatch.setenv('HECKA_RAGGED_X3', '1')      # ragged shapes: mode 0 on the masked split-precision tile
    g = torch.Generator().manual_seed(M + B + P)
    X = (torch.randn(M, B, P, generator=g) ** 3 + 0.25).to(dev)     # skewed, non-zero mean
    Xd = X.double()
    K = Xd @ Xd.transpose(1, 2)
    Cref = K - K.mean(2, keepdim=True)
    A = torch.randn(M, M, generator=g).to(dev)
    dC = torch.einsum('ij,jab->iab', A.double(), Cref)
    Wd = (dC + dC.transpose(1, 2)) * (1 - torch.eye(B, dtype=torch.float64, device=dev))
    dref = Wd @ Xd
    for mode in (0, 2, 1):   # f16x3 (two fp16 terms, scaled by xmax / wmax), bf16x6, fp32 MFMA
        with ops.gemm_mode(mode):
            C, cvec, xmax = ops.centered_gram(X, True)
            assert torch.equal(xmax, X.abs().amax(dim=(1, 2)))
            assert float((C.double() - Cref).abs().max() / Cref.abs().max()) <= 2e-6, mode
            cref = ((Xd - Xd.mean(1, keepdim=True)) * Xd.mean(1, keepdim=True)).sum(2)
            assert float((cvec.double() - cref).abs().max() / cref.abs().max()) <= 1e-5, mode
            dX = ops.gram_backward(A, C, cvec, X, True, xmax)
            assert float((dX.double() - dref).abs().max() / dref.abs().max()) <= 1e-5, mode


def test_gemm_mode_travels_with_the_autograd_node():
    """The GEMM arithmetic is a per-call argument of the C-ABI; `with ops.gemm_mode(m)` around the FORWARD is enough - the
    node re-enters the mode in backward (autograd's thread).  Bit-equal to the same evaluation under a process default
    of that mode, and two kernels with different modes coexist in one process."""
    from collections import OrderedDict
    from hyper_amd import ops
    from hyper_amd.diversity.model_kernels import build_model_kernel
    import bench
    dev = _dev()
    torch.manual_seed(8)
    feats = [torch.randn(8, 128, p, device=dev) for p in (256, 128)]

    def run(ctx_mode):
        xs = [f.clone().requires_grad_(True) for f in feats]
        kern = build_model_kernel(bench.CFG).to(dev)
        with ops.gemm_mode(ctx_mode):
            out = kern(None, OrderedDict((str(i), x) for i, x in enumerate(xs)), None, OrderedDict())
        out.backward()                       # outside the context
        return out.detach(), [x.grad for x in xs]
    res = {m: run(m) for m in (0, 1, 2)}
    try:
        for m in (1, 2):
            ops.set_gemm_mode(m)
            out, grads = run(None)
            assert torch.equal(out, res[m][0]) and all(torch.equal(a, b) for a, b in zip(grads, res[m][1])), m
    finally:
        ops.set_gemm_mode(0)
    assert not torch.equal(res[0][1][0], res[1][1][0])     # the modes do differ in the last bits


def test_f16x3_scaling_over_dynamic_range():
    """The fp16-split path scales every member by a power of two from its max |x|: members of very different
    magnitude (1e-12 .. 1e+7; dX ~ |x|^3 stays inside fp32) and features spanning four decades inside a member keep the fp32-level
    accuracy, per member (relative to that member's own largest Gram entry)."""
    from hyper_amd import ops, _lib
    dev = _dev()
    g = torch.Generator().manual_seed(99)
    M, B, P = 8, 128, 256
    X = torch.randn(M, B, P, generator=g) + 0.3
    X = X * (10.0 ** torch.linspace(-2, 2, P))[None, None, :]                      # four decades across features
    X = (X * torch.tensor([1e-12, 1e-6, 1e-3, 1.0, 7.0, 1e3, 1e5, 1e7])[:, None, None]).to(dev)
    Xd = X.double()
    K = Xd @ Xd.transpose(1, 2)
    Cref = K - K.mean(2, keepdim=True)
    assert ops.get_gemm_mode() == 0
    C, cvec, xmax = ops.centered_gram(X, True)
    err = (C.double() - Cref).abs().amax(dim=(1, 2)) / Cref.abs().amax(dim=(1, 2))
    assert float(err.max()) <= 2e-6, err
    A = torch.eye(M, device=dev)                                                    # dC_i = C_i: every member on its own scale
    dX = ops.gram_backward(A, C, cvec, X, True, xmax)
    gam = cvec.double()
    Wd = (2 * Cref + gam[:, :, None] - gam[:, None, :]) * (1 - torch.eye(B, dtype=torch.float64, device=dev))
    dref = Wd @ Xd
    errd = (dX.double() - dref).abs().amax(dim=(1, 2)) / dref.abs().amax(dim=(1, 2))
    assert float(errd.max()) <= 1e-5, errd


def test_f16x3_layer_gemms_over_dynamic_range():
    """hecka_bgemm_x3_f32 and hecka_conv_x3_f32 take their power-of-two operand scales from max |W| and max |x| per
    member: weights and activations spanning four decades INSIDE a member (elements below 2^-17 of the maximum keep
    absolute, not relative, precision - include/hecka.h) and members eight decades apart stay at the fp32 tolerance,
    measured per member against fp64 on that member's own largest output."""
    import torch.nn.functional as F
    from hyper_amd.layers.linear import _absmax_rows, _bgemm_x3, _x3_eligible
    from hyper_amd.layers.conv import BatchedConv2d
    dev = _dev()
    g = torch.Generator().manual_seed(41)
    batch, rows, cols, K = 4, 128, 256, 512
    member = torch.tensor([1e-4, 1.0, 30.0, 1e4])[:, None, None]
    A = torch.randn(batch, rows, K, generator=g) * (10.0 ** torch.linspace(-2, 2, K))[None, None, :] * member
    Bm = torch.randn(batch, K, cols, generator=g) * (10.0 ** torch.linspace(2, -2, K))[None, :, None] / member
    A, Bm = A.to(dev), Bm.to(dev)
    assert _x3_eligible(batch, rows, cols, K)
    out = _bgemm_x3(A, Bm, batch, rows, cols, K, rows * K, K * cols, True, 0.5, _absmax_rows(A.view(batch, -1), batch),
                    _absmax_rows(Bm.view(batch, -1), batch), 1)
    ref = 0.5 * (A.double() @ Bm.double())
    err = (out.double() - ref).abs().amax(dim=(1, 2)) / ref.abs().amax(dim=(1, 2))
    assert float(err.max()) <= 5e-6, err
    # implicit-GEMM convolution (forward, input gradient, weight gradient) with the same spread in weights and activations
    M, N, C, HW = 3, 8, 128, 16
    x = torch.randn(M, N, C, HW, HW, generator=g) * (10.0 ** torch.linspace(-2, 2, C))[None, None, :, None, None]
    x = (x * torch.tensor([1e-3, 1.0, 1e3])[:, None, None, None, None]).to(dev)
    W = torch.randn(M, C, C, 3, 3, generator=g) * 0.05 * (10.0 ** torch.linspace(2, -2, C))[None, None, :, None, None]
    W = (W * torch.tensor([1e2, 1.0, 1e-2])[:, None, None, None, None]).to(dev)
    up = torch.randn(M, N, C, HW, HW, generator=g).to(dev)
    xx, WW = x.clone().requires_grad_(True), W.clone().requires_grad_(True)
    y = BatchedConv2d.apply(xx, WW, None, 0.7, 1.0, (3, 3, 1, 1, 1, 1, 1, 1))
    (y * up).sum().backward()
    xd, Wd = x.double().requires_grad_(True), W.double().requires_grad_(True)
    yd = torch.stack([F.conv2d(xd[m], 0.7 * Wd[m], None, 1, 1) for m in range(M)])
    (yd * up.double()).sum().backward()
    for name, got, want in (('y', y.detach(), yd.detach()), ('dx', xx.grad, xd.grad), ('dW', WW.grad, Wd.grad)):
        e = (got.double() - want).flatten(1).abs().amax(dim=1) / want.flatten(1).abs().amax(dim=1)
        assert float(e.max()) <= 1e-5, (name, e)


@pytest.mark.parametrize('P,style', [(512, 'randn'), (2048, 'correlated')])
def test_bench_shape_against_fp64_oracle_on_gpu(P, style):
    """Full bench member/sample counts (M=64, B=512) against the oracle evaluated in fp64 with stock
    torch ops on the same GPU (the CPU would need minutes); rel 1e-5 on loss and gradients."""
    from oracle import hecka_ref as R
    from hyper_amd.diversity.model_kernels import build_model_kernel
    from collections import OrderedDict
    dev = _dev()
    torch.manual_seed(1234)
    M, B = 64, 512
    X = torch.randn(M, B, P, device=dev)
    if style == 'correlated':
        X = 0.7 * torch.randn(1, B, P, device=dev) + 0.3 * X
    for lk in (dict(eps=0.0025, arc_eps=0.031), dict(eps=0.001, arc_eps=0.001, use_exp=True)):
        cfg = {'name': 'model_layers', 'layer_weighting': {'name': 'uniform'},
               'layer_kernel': dict(name='he', feature_kernel={'name': 'feature', 'detach_diag': True, 'kernel': 'linear'}, **lk)}
        x = X.clone().requires_grad_(True)
        out = build_model_kernel(cfg).to(dev)(None, OrderedDict(a=x), None, OrderedDict())
        out.backward()
        xo = X.double().requires_grad_(True)
        ref = R.run_config(cfg, [xo])
        ref.backward()
        assert abs(out.item() - ref.item()) <= RTOL * abs(ref.item()), (out.item(), ref.item())
        err = float((x.grad.double() - xo.grad).abs().max() / xo.grad.abs().max())
        assert err <= RTOL, err


@pytest.mark.parametrize('style', ['randn', 'correlated'])
def test_bench_workload_five_layers_against_fp64_oracle(style):
    """The EXACT bench.py workload (BASELINE configs[3]): all five layers P = 512 .. 8192 at M = 64, B = 512 through the
    fused `model_layers` node (one Gram launch for all layers, batched heads, K = 8192 accumulation chains of the f16x3
    split) against the oracle in fp64 on the same tensors (evaluated with stock torch ops on the GPU; X at P = 8192 is
    2 GB in fp64).  Power head (configs/mnist/ens/he-lin.yml) and use_exp head (configs/cifar10/ens/he-lin.yml); the
    default f16x3 arithmetic and HECKA_GEMM_BF16X6.  rel 1e-5 on the loss and on every layer's gradient (max-norm)."""
    import bench
    from hyper_amd import ops
    from hyper_amd.diversity.model_kernels import build_model_kernel
    from collections import OrderedDict
    dev = _dev()
    feats = bench.make_feats(dev, slice(0, bench.M), correlated=(style == 'correlated'))
    assert [tuple(f.shape) for f in feats] == [(64, 512, p) for p in (512, 1024, 2048, 4096, 8192)]
    for cfg in (bench.CFG, bench.CFG_EXP):
        ref_loss, ref_grads = bench.oracle_fp64_on_gpu(cfg, feats)
        for mode in (0, 2):
            xs = [f.detach().clone().requires_grad_(True) for f in feats]
            track = OrderedDict()
            with ops.gemm_mode(mode):    # the autograd node keeps the mode for its backward (which runs on another thread)
                out = build_model_kernel(cfg).to(dev)(None, OrderedDict((str(i), x) for i, x in enumerate(xs)), None, track)
            out.backward()
            assert len(track['layer_track']) == 5
            rec = bench.parity_record(float(out.item()), [x.grad for x in xs], ref_loss, ref_grads)
            assert rec['loss_rel'] <= RTOL, (mode, rec)
            assert rec['grad_rel_max'] <= RTOL, (mode, rec)
            del xs, out, track
        del ref_grads
        torch.cuda.empty_cache()


@pytest.mark.parametrize('M,B,dims,force', [(128, 400, (864, 256, 120, 84), False), (5, 300, (864, 256, 120, 84), True),
                                            (30, 100, (40, 40, 20), True), (9, 132, (20, 36, 1000), True)])
def test_model_layers_at_the_references_batch_sizes_against_fp64_oracle(M, B, dims, force, monkeypatch):
    """The reference's own (ragged) batch sizes through the fused `model_layers` node - one masked split-precision Gram
    launch for all layers (split-K where the grid is small), masked dX GEMMs - against the fp64 oracle on the GPU: the
    hypernetwork's 128 x (300 + 100) by the library's own dispatch, MNIST 5 x 300 and toy 30 x 100 with the tile forced
    (HECKA_RAGGED_X3=1; by default their small grids stay on the 64-row fp32-MFMA kernels)."""
    import bench
    from hyper_amd import ops
    from hyper_amd.diversity.model_kernels import build_model_kernel
    from collections import OrderedDict
    dev = _dev()
    if force:
        monkeypatch.setenv('HECKA_RAGGED_X3', '1')
    torch.manual_seed(M + B)
    feats = [torch.randn(M, B, p, device=dev) * (1.0 + i) + 0.3 for i, p in enumerate(dims)]
    assert ops.multi_gram_eligible(feats)
    for cfg in (bench.CFG, bench.CFG_EXP):
        ref_loss, ref_grads = bench.oracle_fp64_on_gpu(cfg, feats)
        xs = [f.detach().clone().requires_grad_(True) for f in feats]
        out = build_model_kernel(cfg).to(dev)(None, OrderedDict((str(i), x) for i, x in enumerate(xs)), None, OrderedDict())
        out.backward()
        rec = bench.parity_record(float(out.item()), [x.grad for x in xs], ref_loss, ref_grads)
        assert rec['loss_rel'] <= RTOL, rec
        assert rec['grad_rel_max'] <= RTOL, rec


@pytest.mark.parametrize('dtype,native', [('bfloat16', '1'), ('float16', '1'), ('bfloat16', '0'), ('bfloat16', 'exact')])
@pytest.mark.parametrize('dims', [(256, 1024), (128,), (256, 80)])
def test_16bit_features_match_oracle_on_rounded_values(dtype, native, dims, monkeypatch):
    """BASELINE config 3 / SURVEY.md 8d: bf16 (and fp16) features; oracle = the restatement in fp64 on the rounded
    values.  north_star tolerance for bf16 features: rel 1e-2; held here: the loss at the fp32 level (1e-5) and the
    gradient to its single rounding into the feature type (2^-8 bf16, 2^-11 fp16, of the largest gradient).
    native=1: the x16 entry points read the 16-bit features directly (dims that are not multiples of 128 fall back
    to the up-cast); native=0: HECKA_NATIVE16=0, up-cast for everything; exact: native with HECKA_X16_EXACT=1 (centred
    three-product Gram instead of the raw one-product Gram, whose K - u cancellation costs the loss a digit)."""
    from oracle import hecka_ref as R
    from hyper_amd import ops
    from hyper_amd.diversity.model_kernels import build_model_kernel
    from collections import OrderedDict
    raw = native == '1'
    if native == 'exact':
        monkeypatch.setenv('HECKA_X16_EXACT', '1')
        native = '1'
    else:
        monkeypatch.delenv('HECKA_X16_EXACT', raising=False)
    monkeypatch.setenv('HECKA_NATIVE16', native)
    dev = _dev()
    dt = getattr(torch, dtype)
    torch.manual_seed(4)
    cfg = {'name': 'model_layers', 'layer_weighting': {'name': 'uniform'},
           'layer_kernel': dict(name='he', eps=0.001, arc_eps=0.001, use_exp=True,
                                feature_kernel={'name': 'feature', 'detach_diag': True, 'kernel': 'linear'})}
    xs = [(torch.relu(torch.randn(10, 128, p, device=dev) + 0.3) * 3.0 ** i).to(dt).requires_grad_(True)
          for i, p in enumerate(dims)]
    assert ops.native16_eligible([x.detach() for x in xs]) == (native == '1' and all(p % 128 == 0 for p in dims))
    out = build_model_kernel(cfg).to(dev)(None, OrderedDict((str(i), x) for i, x in enumerate(xs)), None, OrderedDict())
    out.backward()
    xo = [x.detach().double().requires_grad_(True) for x in xs]
    ref = R.run_config(cfg, xo)
    ref.backward()
    assert abs(out.item() - ref.item()) <= (1e-4 if raw else 1e-5) * abs(ref.item())
    for x, o in zip(xs, xo):
        assert x.grad.dtype == dt
        gmax = float(o.grad.abs().max())
        err = float((x.grad.double() - o.grad).abs().max()) / gmax
        # fp16 gradients of a mean-reduced loss sit in fp16's subnormal range: absolute quantum 2^-25 on top of 2^-11
        tol = 2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11 + 2.0 ** -25 / gmax
        assert err <= tol * 1.01, (err, tol)


@pytest.mark.parametrize('exact', [True, False])
def test_native_16bit_equals_upcast_path(exact, monkeypatch):
    """The x16 kernels against the fp32 kernels on the up-cast copy of the same bf16 features, bench member/sample
    counts.  exact (HECKA_X16_EXACT=1): same loss to 1e-6 and gradients that differ by at most one bf16 ulp (both round
    an fp32-accurate value once) plus 2e-6 of the largest.  Default (raw one-product Gram): loss to 1e-4, gradients to one ulp + 1e-4 of the
    largest."""
    from hyper_amd.diversity.model_kernels import build_model_kernel
    from collections import OrderedDict
    dev = _dev()
    torch.manual_seed(8)
    cfg = {'name': 'model_layers', 'layer_weighting': {'name': 'uniform'},
           'layer_kernel': dict(name='he', eps=0.0025, arc_eps=0.031,
                                feature_kernel={'name': 'feature', 'detach_diag': True, 'kernel': 'linear'})}
    base = [torch.randn(64, 512, p, device=dev).bfloat16() for p in (512, 1280)]
    res = {}
    if exact:
        monkeypatch.setenv('HECKA_X16_EXACT', '1')
    else:
        monkeypatch.delenv('HECKA_X16_EXACT', raising=False)
    for native in ('1', '0'):
        monkeypatch.setenv('HECKA_NATIVE16', native)
        xs = [x.clone().requires_grad_(True) for x in base]
        out = build_model_kernel(cfg).to(dev)(None, OrderedDict((str(i), x) for i, x in enumerate(xs)), None, OrderedDict())
        out.backward()
        res[native] = (out.item(), [x.grad for x in xs])
    assert abs(res['1'][0] - res['0'][0]) <= (1e-6 if exact else 1e-4) * abs(res['0'][0])
    for g1, g0 in zip(res['1'][1], res['0'][1]):
        d = (g1.float() - g0.float()).abs()
        # (exact: the fp32 path runs its Gram / dX on the 16x16x32 tile, the 16-bit path on the 32x32x16 one - two fp32
        # summation orders, so elements that cancel to near zero differ by fp32 rounding of the LARGEST terms, not by a
        # relative ulp of their own value)
        ulp = g0.float().abs() * 2.0 ** -7 + (2e-6 if exact else 1e-4) * float(g0.float().abs().max())
        assert bool((d <= ulp).all())
        if exact:
            assert float((d > 0).float().mean()) < 0.02      # the two roundings disagree only near a rounding boundary


def test_learnable_head_matches_autograd_oracle():
    """learnable=True keeps the [M,M] head in torch: loss, dX, d eps and d arc_eps vs the oracle's autograd."""
    from oracle import hecka_ref as R
    from hyper_amd.diversity.model_kernels import build_model_kernel
    from collections import OrderedDict
    dev = _dev()
    torch.manual_seed(11)
    X = torch.randn(6, 40, 24)
    cfg = {'name': 'he', 'eps': 0.0025, 'arc_eps': 0.031, 'learnable': True,
           'feature_kernel': {'name': 'feature', 'detach_diag': True, 'kernel': 'linear'}}
    kern = build_model_kernel(cfg).to(dev)
    x = X.to(dev).requires_grad_(True)
    out = kern(None, x, None, OrderedDict()).mean()
    out.backward()
    xo = X.double().requires_grad_(True)
    eps = torch.tensor(R.f32(0.0025), dtype=torch.float64, requires_grad=True)
    arc = torch.tensor(R.f32(0.031), dtype=torch.float64, requires_grad=True)
    ref = R.he_layer(xo, eps=eps, arc_eps=arc, detach_diag=True).mean()
    ref.backward()
    assert abs(out.item() - ref.item()) <= RTOL * abs(ref.item())
    assert _relerr(x.grad.cpu().numpy(), xo.grad.numpy()) <= RTOL
    assert abs(kern.eps.grad.item() - eps.grad.item()) <= 1e-4 * abs(eps.grad.item())
    assert abs(kern.arc_eps.grad.item() - arc.grad.item()) <= 1e-4 * abs(arc.grad.item())


def test_errors_are_loud():
    """Bad arguments come back as error codes / exceptions, never as silent fallbacks."""
    from hyper_amd import _lib, ops
    L = _lib.lib()
    assert L.hecka_colmean_f32(None, 1, 1, 1, None, None, None) == -1
    assert b'null' in L.hecka_last_error()
    with pytest.raises(RuntimeError):
        ops.centered_gram_pairwise(torch.randn(2, 4, 3))          # CPU tensor: no CPU path
    with pytest.raises(RuntimeError):
        ops.centered_gram_pairwise(torch.randn(2, 4, 3, device=_dev()).double())


@pytest.mark.parametrize('world', [2, 4])
def test_sharded_stages_on_gpu(world):
    """The member-sharded stages (hyper_amd.parallel) with the HIP kernels, emulated in one process:
    every 'rank' computes its row block from the full Gram set; the pieces must equal the unsharded result."""
    from hyper_amd import ops
    dev = _dev()
    torch.manual_seed(3)
    M, B, P = 8, 64, 48
    X = torch.randn(M, B, P, device=dev)
    spec = ops.HeadSpec(ops.HEAD_HE, s=2.0, eps=0.0025, arc_eps=0.031)
    C, cvec = ops.centered_gram(X)
    S = ops.pairwise_inner(C)
    nsq = torch.diagonal(S).contiguous()
    scale = 1.0 / (M * (M - 1))
    _, full = ops.head_forward(S, nsq, 0, spec, scale)
    one = torch.ones((), device=dev)
    A = ops.head_backward(S, nsq, 0, spec, one, scale)
    dX = ops.gram_backward(A, C, cvec, X, True)
    Ml, tot = M // world, 0.0
    for r in range(world):
        rows = slice(r * Ml, (r + 1) * Ml)
        Sl = ops.pairwise_inner(C[rows].contiguous(), C)
        assert _relerr(Sl.cpu().numpy(), S[rows].cpu().double().numpy()) <= 1e-6
        _, part = ops.head_forward(Sl, nsq, r * Ml, spec, scale)
        tot += part.item()
        Al = ops.head_backward(Sl, nsq, r * Ml, spec, one, scale)
        assert _relerr(Al.cpu().numpy(), A[rows].cpu().double().numpy()) <= 1e-5
        dXl = ops.gram_backward(Al, C, cvec, X[rows].contiguous(), True)
        assert _relerr(dXl.cpu().numpy(), dX[rows].cpu().double().numpy()) <= 1e-5
    assert abs(tot - full.item()) <= 1e-6 * abs(full.item())


def test_toy_train_step_matches_reference_golden():
    """BASELINE configs[1] shape (toy four-class MLP ensemble, rbf HE repulsion): one training step through
    hyper_amd's FixedEnsembleModel -> batched Linear (MFMA) -> RepulsiveKernel -> backward, against the step
    the REAL reference produced (tests/golden/toy_train_step_f64.npz, same weights and batch)."""
    import json
    import types
    import numpy as np
    import hyper_amd.target  # noqa: F401
    from hyper_amd.generators import build_generator
    from hyper_amd.diversity.methods import build_method
    from conftest import GOLDEN
    import os
    z = np.load(os.path.join(GOLDEN, 'toy_train_step_f64.npz'), allow_pickle=False)
    cfg = json.loads(str(z['config']))
    dev = _dev()
    hyper = build_generator(cfg['hyper'])
    hyper.load_state_dict({k[2:]: torch.from_numpy(z[k]).float() for k in z.files if k.startswith('w:')})
    hyper = hyper.to(dev)
    method = build_method(cfg['method']).to(dev)
    X = torch.from_numpy(z['X']).float().to(dev)
    Y = torch.from_numpy(z['Y']).to(dev)
    track = hyper(params=list(range(6)), x=X, sample_params=True, ret_params=True, feature_split=True, ood_N=0)
    assert list(track['feats_ind'].keys()) == json.loads(str(z['feat_keys']))
    track = method(types.SimpleNamespace(training=True), track['params'], [], track, X, Y, step=int(z['step']))
    track['loss'].backward()
    _check_step(Yardstick('toy_train_step'), track, hyper)


def _check_step(yard, track, hyper, keys=('pred', 'ind_rep_loss', 'loss')):
    """Every output of a training step against the float64 fixture, each held to max(1e-5, the deviation of the
    reference's own float32 run of that quantity) - conftest.Yardstick."""
    for key in keys:
        yard.check(key, track[key].detach().cpu().numpy())
    for k, v in hyper.named_parameters():
        yard.check('g:' + k, v.grad.cpu().numpy())


CONV_CASES = ['conv_shared_gamma', 'conv_member_torch_nobias', 'conv_dilated_rect', 'conv_same_even', 'conv_reflect_1x1',
              'conv_groups2_member', 'conv_depthwise_shared']      # the last two: groups > 1 (conv.py:334), folded into members


@pytest.mark.parametrize('name', CONV_CASES)
def test_batched_conv2d_matches_reference_golden(name):
    """hyper_amd Conv2d (im2col gather -> grouped MFMA GEMM -> col2im) vs the reference's per-member F.conv2d loop
    (tests/golden/conv_*_f64.npz from the REAL reference): output, input gradient, flat parameter gradient."""
    import json
    import os
    from conftest import GOLDEN
    from hyper_amd.layers.conv import Conv2d
    z = np.load(os.path.join(GOLDEN, name + '_f64.npz'), allow_pickle=False)
    kwargs = json.loads(str(z['kwargs']))
    if isinstance(kwargs.get('kernel_size'), list):
        kwargs = {k: tuple(v) if isinstance(v, list) else v for k, v in kwargs.items()}
    dev = _dev()
    conv = Conv2d(**kwargs)
    flat = torch.from_numpy(z['flat']).float().to(dev).requires_grad_(True)
    x = torch.from_numpy(z['x']).float().to(dev).requires_grad_(True)
    tracked, y = conv({'self': flat}, x)
    assert tracked is y and y.is_contiguous()
    assert tuple(y.shape) == z['y'].shape
    (y * torch.from_numpy(z['up']).float().to(dev)).sum().backward()
    assert _relerr(y.detach().cpu().numpy(), z['y']) <= RTOL
    assert _relerr(x.grad.cpu().numpy(), z['dx']) <= RTOL
    assert _relerr(flat.grad.cpu().numpy(), z['dflat']) <= RTOL


def test_lenet_train_step_matches_reference_golden():
    """BASELINE configs[0] (MNIST LeNet-5 x5 with HE-CKA, configs/mnist/ens/he-lin.yml): one training step through
    hyper_amd's FixedEnsembleModel -> batched Conv2d / Linear -> RepulsiveKernel -> backward against the step the
    REAL reference produced on its CPU path (tests/golden/lenet_train_step_f64.npz, same weights and batch)."""
    import json
    import os
    import types
    import hyper_amd.target  # noqa: F401
    from hyper_amd.generators import build_generator
    from hyper_amd.diversity.methods import build_method
    from conftest import GOLDEN
    z = np.load(os.path.join(GOLDEN, 'lenet_train_step_f64.npz'), allow_pickle=False)
    cfg = json.loads(str(z['config']))
    dev = _dev()
    hyper = build_generator(cfg['hyper'])
    hyper.load_state_dict({k[2:]: torch.from_numpy(z[k]).float() for k in z.files if k.startswith('w:')})
    hyper = hyper.to(dev)
    method = build_method(cfg['method']).to(dev)
    X = torch.from_numpy(z['X']).float().to(dev)
    Y = torch.from_numpy(z['Y']).to(dev)
    track = hyper(params=list(range(5)), x=X, sample_params=True, ret_params=True, feature_split=True, ood_N=0)
    assert list(track['feats_ind'].keys()) == json.loads(str(z['feat_keys']))
    assert [list(v.shape) for v in track['feats_ind'].values()] == json.loads(str(z['feat_shapes']))
    track = method(types.SimpleNamespace(training=True), track['params'], [], track, X, Y, step=int(z['step']))
    track['loss'].backward()
    _check_step(Yardstick('lenet_train_step'), track, hyper)


@pytest.mark.parametrize('shape,affine,training', [((3, 6, 5, 7, 4), True, True), ((2, 9, 4, 3, 3), False, True),
                                                    ((3, 6, 5, 7, 4), True, False), ((4, 10, 12), True, True)])
def test_batched_batchnorm_matches_oracle(shape, affine, training):
    """hecka_batchnorm_fwd/bwd (one launch for all members) vs the per-member F.batch_norm loop of norm.py:206-222
    (oracle.layers_ref, fp64 CPU): output, dx, dweight, dbias and the running statistics after the call."""
    from oracle import layers_ref as LR
    from hyper_amd.layers.norm import BatchedBatchNorm
    dev = _dev()
    g = torch.Generator().manual_seed(sum(shape) + affine)
    M, N, C = shape[:3]
    x64 = (torch.randn(*shape, generator=g, dtype=torch.float64) * 1.7 + 0.6).requires_grad_(True)
    w64 = (torch.randn(M, C, generator=g, dtype=torch.float64) + 1.0).requires_grad_(True) if affine else None
    b64 = torch.randn(M, C, generator=g, dtype=torch.float64).requires_grad_(True) if affine else None
    rm64 = torch.randn(M, C, generator=g, dtype=torch.float64) * 0.1
    rv64 = torch.rand(M, C, generator=g, dtype=torch.float64) + 0.5
    up = torch.randn(*shape, generator=g, dtype=torch.float64)
    x = x64.detach().float().to(dev).requires_grad_(True)
    w = w64.detach().float().to(dev).requires_grad_(True) if affine else None
    b = b64.detach().float().to(dev).requires_grad_(True) if affine else None
    rm, rv = rm64.float().to(dev), rv64.float().to(dev)
    y, mean, var = BatchedBatchNorm.apply(x, w, b, None if training else rm, None if training else rv, training, 1e-5)
    (y * up.float().to(dev)).sum().backward()
    yo = LR.batched_batch_norm(x64, w64, b64, rm64, rv64, training, 0.1, 1e-5)   # updates rm64 / rv64 in place
    (yo * up).sum().backward()
    assert _relerr(y.detach().cpu().numpy(), yo.detach().numpy()) <= RTOL
    assert _relerr(x.grad.cpu().numpy(), x64.grad.numpy()) <= RTOL
    if affine:
        assert _relerr(w.grad.cpu().numpy(), w64.grad.numpy()) <= RTOL
        assert _relerr(b.grad.cpu().numpy(), b64.grad.numpy()) <= RTOL
    if training:
        cnt = x64[0, :, 0].numel()
        assert _relerr((0.9 * rm + 0.1 * mean).cpu().numpy(), rm64.numpy()) <= RTOL
        assert _relerr((0.9 * rv + 0.1 * var * cnt / (cnt - 1)).cpu().numpy(), rv64.numpy()) <= RTOL


def test_resnet_train_step_matches_reference_golden():
    """BASELINE configs[2] shape (CIFAR ResNet18 ensemble with HE-CKA on the eight tracked block features; narrow
    planes, 3 members): one training step through hyper_amd's FixedEnsembleModel -> batched Conv2d / BatchNorm2d /
    Linear -> RepulsiveKernel -> backward against the step the REAL reference produced
    (tests/golden/resnet_train_step_f64.npz): predictions, losses, every parameter gradient, and the BatchNorm
    running statistics after the step (state-dict keys `...norm_buffer.self.mod-<member>-<name>`)."""
    import json
    import os
    import types
    import hyper_amd.target  # noqa: F401
    from hyper_amd.generators import build_generator
    from hyper_amd.diversity.methods import build_method
    from conftest import GOLDEN
    z = np.load(os.path.join(GOLDEN, 'resnet_train_step_f64.npz'), allow_pickle=False)
    cfg = json.loads(str(z['config']))
    dev = _dev()
    hyper = build_generator(cfg['hyper'])
    ref_keys = sorted(k[2:] for k in z.files if k.startswith('w:'))
    assert sorted(hyper.state_dict().keys()) == ref_keys
    hyper.load_state_dict({k[2:]: torch.from_numpy(z[k]).float() for k in z.files if k.startswith('w:')})
    hyper = hyper.to(dev)
    method = build_method(cfg['method']).to(dev)
    X = torch.from_numpy(z['X']).float().to(dev)
    Y = torch.from_numpy(z['Y']).to(dev)
    track = hyper(params=list(range(3)), x=X, sample_params=True, ret_params=True, feature_split=True, ood_N=0)
    assert list(track['feats_ind'].keys()) == json.loads(str(z['feat_keys']))
    assert [list(v.shape) for v in track['feats_ind'].values()] == json.loads(str(z['feat_shapes']))
    track = method(types.SimpleNamespace(training=True), track['params'], [], track, X, Y, step=int(z['step']))
    track['loss'].backward()
    yard = Yardstick('resnet_train_step')
    _check_step(yard, track, hyper)
    after = hyper.state_dict()
    for k in z.files:
        if k.startswith('after:') and np.abs(z[k]).max() > 0:
            yard.check(k, after[k[6:]].cpu().numpy())


@pytest.mark.parametrize('B', [128, 6])
def test_resnet_step_with_bf16_features_matches_oracle_on_rounded_features(B):
    """BASELINE configs[2] says bf16: the ResNet18 ensemble of the fixture above, tracked block features cast to bf16 on
    their way into the HE loss (the reference itself has no bf16 switch - SURVEY.md F6 - so the oracle is the
    restatement in fp64 on the bf16-ROUNDED features, north_star tolerance rel 1e-2).  B = 128: the x16 kernels read the
    bf16 features natively (P = C H W is a multiple of 128 for the block features; the 10 logits take the up-cast);
    B = 6 (the fixture batch): everything through the up-cast.  Checked: the loss, dLoss/dfeature for every tracked
    layer, and that the parameter gradients of the whole step stay within 1e-2 of the fp32-feature step's."""
    import json
    import os
    import types
    from collections import OrderedDict
    import hyper_amd.target  # noqa: F401
    from hyper_amd.generators import build_generator
    from hyper_amd.diversity.methods import build_method
    from oracle import hecka_ref as R
    from conftest import GOLDEN
    z = np.load(os.path.join(GOLDEN, 'resnet_train_step_f64.npz'), allow_pickle=False)
    cfg = json.loads(str(z['config']))
    dev = _dev()
    hyper = build_generator(cfg['hyper'])
    hyper.load_state_dict({k[2:]: torch.from_numpy(z[k]).float() for k in z.files if k.startswith('w:')})
    hyper = hyper.to(dev)
    method = build_method(cfg['method']).to(dev)
    g = torch.Generator().manual_seed(12)
    X = (torch.from_numpy(z['X']).float() if B == 6 else torch.randn(B, 3, 16, 16, generator=g)).to(dev)
    Y = (torch.from_numpy(z['Y']) if B == 6 else torch.randint(0, 10, (B,), generator=g)).to(dev)
    trainer = types.SimpleNamespace(training=True)

    def step(cast):
        hyper.zero_grad()
        sd = {k: v.clone() for k, v in hyper.state_dict().items()}     # the step updates the running statistics
        track = hyper(params=list(range(3)), x=X, sample_params=True, ret_params=True, feature_split=True, ood_N=0)
        feats = track['feats_ind']
        if cast:
            track['feats_ind'] = OrderedDict((k, v.bfloat16()) for k, v in feats.items())
        track = method(trainer, track['params'], [], track, X, Y, step=int(z['step']))
        track['loss'].backward()
        hyper.load_state_dict(sd)
        return track, feats, {k: v.grad.clone() for k, v in hyper.named_parameters()}

    t32, _, g32 = step(False)
    t16, feats, g16 = step(True)
    # (1) the loss term itself on bf16 leaves against the oracle on the rounded values
    leaves = [v.detach().bfloat16().requires_grad_(True) for v in feats.values()]
    rep = method.model_kernel(None, OrderedDict((k, x) for k, x in zip(feats.keys(), leaves)), None, OrderedDict())
    rep.backward()
    xo = [x.detach().double().requires_grad_(True) for x in leaves]
    ref = R.run_config(cfg['method']['model_kernel'], xo)
    ref.backward()
    assert abs(rep.item() - ref.item()) <= 1e-2 * abs(ref.item())
    assert abs(t16['ind_rep_loss'].item() - ref.item()) <= 1e-2 * abs(ref.item())
    for x, o in zip(leaves, xo):
        if float(o.grad.abs().max()) > 0:      # (the first tracked layer has weight 0 in configs/cifar10/ens/he-lin.yml)
            assert x.grad.dtype == torch.bfloat16
            assert float((x.grad.double() - o.grad).abs().max() / o.grad.abs().max()) <= 1e-2
    # (2) the whole step: bf16 features move the parameter gradients by no more than the north_star tolerance
    for k in g32:
        scale = float(g32[k].abs().max())
        if scale > 0:
            assert float((g16[k] - g32[k]).abs().max()) <= 1e-2 * scale, k
    assert abs(t16['loss'].item() - t32['loss'].item()) <= 1e-2 * abs(t32['loss'].item())


@pytest.mark.parametrize('M,B,dims', [(8, 128, (64, 272, 32)), (5, 256, (48, 16))])
def test_all_layers_gram_launch_matches_per_layer_and_oracle(M, B, dims, monkeypatch):
    """Three routes through ModelLayersKernel: 'fused' (ops.DiversityLayers: one autograd node, one Gram / head /
    head_bwd / gamma / bound launch for all layers), 'multi' (per-layer nodes fed by hecka_gram_center_multi_f32) and
    'single' (per-layer launches).  multi == single bit for bit; fused has the same Grams and differs from them only
    in where the layer weight is multiplied in (1e-6); all match the oracle at RTOL."""
    from collections import OrderedDict
    from oracle import hecka_ref as R
    from hyper_amd import ops
    from hyper_amd.diversity.model_kernels import build_model_kernel
    dev = _dev()
    cfg = {'name': 'model_layers', 'layer_weighting': {'name': 'linear', 'start_weight': 1.0, 'weight_increase': 0.5},
           'layer_kernel': dict(name='he', eps=0.0025, arc_eps=0.031,
                                feature_kernel={'name': 'feature', 'detach_diag': True, 'kernel': 'linear'})}
    torch.manual_seed(M + B)
    base = [torch.relu(torch.randn(M, B, p) + 0.2) for p in dims]
    res = {}
    for mode, (fused, multi) in {'fused': ('1', '1'), 'multi': ('0', '1'), 'single': ('0', '0')}.items():
        monkeypatch.setenv('HECKA_FUSED_LAYERS', fused)
        monkeypatch.setenv('HECKA_MULTI_GRAM', multi)
        xs = [b.to(dev).requires_grad_(True) for b in base]
        assert ops.multi_gram_eligible([x.detach() for x in xs])
        track = OrderedDict()
        out = build_model_kernel(cfg).to(dev)(None, OrderedDict((str(i), x) for i, x in enumerate(xs)), None, track)
        out.backward()
        res[mode] = (out.item(), [x.grad.cpu() for x in xs], [t['gram'].cpu() for t in track['layer_track']],
                     [t['gram_inner'][0].cpu() for t in track['layer_track']])
    assert res['multi'][0] == res['single'][0]
    for a, b in zip(res['multi'][1] + res['multi'][2], res['single'][1] + res['single'][2]):
        assert torch.equal(a, b)
    assert abs(res['fused'][0] - res['single'][0]) <= 1e-6 * abs(res['single'][0])
    for a, b in zip(res['fused'][2] + res['fused'][3], res['single'][2] + res['single'][3]):
        assert torch.equal(a, b)                     # Grams and their inner products: the same kernels
    for a, b in zip(res['fused'][1], res['single'][1]):
        assert _relerr(a.numpy(), b.double().numpy()) <= 1e-6
    xo = [b.double().requires_grad_(True) for b in base]
    ref = R.run_config(cfg, xo)
    ref.backward()
    for mode in ('fused', 'single'):
        assert abs(res[mode][0] - ref.item()) <= RTOL * abs(ref.item())
        for g, o in zip(res[mode][1], xo):
            assert _relerr(g.numpy(), o.grad.numpy()) <= RTOL


def test_svgd_step_matches_reference_golden():
    """configs/mnist/svgd/cka-cossim.yml: one SVGD update - feature-space CKA kernel MATRIX
    (reduction none, detach_right, cossim Grams) and its gradient from the HIP path, K @ drive - dK - against the
    update the REAL reference produced (tests/golden/svgd_step_f64.npz)."""
    import json
    import os
    import types
    import hyper_amd.target  # noqa: F401
    from hyper_amd.generators import build_generator
    from hyper_amd.diversity.methods import build_method
    from conftest import GOLDEN
    z = np.load(os.path.join(GOLDEN, 'svgd_step_f64.npz'), allow_pickle=False)
    cfg = json.loads(str(z['config']))
    dev = _dev()
    hyper = build_generator(cfg['hyper'])
    hyper.load_state_dict({k[2:]: torch.from_numpy(z[k]).float() for k in z.files if k.startswith('w:')})
    hyper = hyper.to(dev)
    method = build_method(cfg['method']).to(dev)
    X = torch.from_numpy(z['X']).float().to(dev)
    Y = torch.from_numpy(z['Y']).to(dev)
    trainer = types.SimpleNamespace(training=True)
    track = hyper(params=list(range(5)), x=X, sample_params=True, ret_params=True, feature_split=True, ood_N=0)
    track = method(trainer, track['params'], [], track, X, Y, step=0)
    loss, _ = method.backward(trainer, track['params'], [], track, X, Y)
    yard = Yardstick('svgd_step')
    yard.check('loss', loss.item())
    for k, v in hyper.named_parameters():
        yard.check('g:' + k, v.grad.cpu().numpy())


def test_ckas_metric_from_loss_byproducts_matches_oracle():
    """`ckas` metric: mean pairwise CKA per layer from the inner products the loss already holds,
    vs pairwise_cossim(track gram).mean() of the oracle (metrics.py:91)."""
    from collections import OrderedDict
    from oracle import hecka_ref as R
    from hyper_amd.diversity.model_kernels import build_model_kernel
    from hyper_amd.experiments.metrics import build_metric
    dev = _dev()
    cfg = {'name': 'model_layers', 'layer_weighting': {'name': 'uniform'},
           'layer_kernel': dict(name='he', eps=0.0025, arc_eps=0.031,
                                feature_kernel={'name': 'feature', 'detach_diag': True, 'kernel': 'linear'})}
    torch.manual_seed(11)
    xs = [torch.relu(torch.randn(6, 40, p)) for p in (24, 9, 130)]
    track = {'ind_kernel': OrderedDict()}
    build_model_kernel(cfg).to(dev)(None, OrderedDict((str(i), x.to(dev)) for i, x in enumerate(xs)), None, track['ind_kernel'])
    logged = build_metric({'name': 'ckas'}).log_train(None, 6, None, None, track)
    assert list(logged) == ['cka_ind_%d/train' % (i + 1) for i in range(3)]
    otrack = {}
    R.run_config(cfg, [x.double() for x in xs], otrack)
    for i, t in enumerate(otrack['layer_track']):
        ref = R.pairwise_cossim(t['gram']).mean().item()
        assert abs(logged['cka_ind_%d/train' % (i + 1)] - ref) <= RTOL * abs(ref)


@pytest.mark.parametrize('M,B,P', [(2, 2, 1), (2, 3, 5), (3, 1, 4), (1, 6, 3), (7, 129, 1)])
def test_smallest_shapes_match_oracle(M, B, P):
    """Edge sizes: two members, two samples, one feature, a single sample (every centred Gram is exactly zero: cos = 0/eps),
    a single member (no pairs: the off-diagonal mean is NaN in the reference - 0/0 - and here)."""
    from collections import OrderedDict
    from oracle import hecka_ref as R
    from hyper_amd.diversity.model_kernels import build_model_kernel
    dev = _dev()
    cfg = {'name': 'model_layers', 'layer_weighting': {'name': 'uniform'},
           'layer_kernel': dict(name='he', eps=0.0025, arc_eps=0.031,
                                feature_kernel={'name': 'feature', 'detach_diag': True, 'kernel': 'linear'})}
    torch.manual_seed(M * 100 + B * 10 + P)
    x0 = torch.randn(M, B, P)
    x = x0.to(dev).requires_grad_(True)
    out = build_model_kernel(cfg).to(dev)(None, OrderedDict(a=x), None, OrderedDict())
    xo = x0.double().requires_grad_(True)
    ref = R.run_config(cfg, [xo])
    if M == 1:
        assert torch.isnan(ref) and torch.isnan(out)
        return
    out.backward()
    ref.backward()
    assert abs(out.item() - ref.item()) <= RTOL * max(abs(ref.item()), 1e-30)
    scale = max(float(xo.grad.abs().max()), 1e-30)
    assert float((x.grad.cpu().double() - xo.grad).abs().max()) <= RTOL * scale + 1e-12


def test_cossim_zero_row_is_nan_and_the_method_masks_it():
    """cossim_gram has no eps (kernels.py:145-169): a zero feature row gives NaN, as in the reference; RepulsiveKernel
    then counts the repulsive term as 0 (methods.py:342)."""
    import types
    from collections import OrderedDict
    from oracle import hecka_ref as R
    from hyper_amd.diversity.model_kernels import build_model_kernel
    from hyper_amd.diversity.methods import build_method
    dev = _dev()
    mk = {'name': 'model_layers', 'layer_weighting': {'name': 'uniform'},
          'layer_kernel': dict(name='he', eps=0.0025, arc_eps=0.031,
                               feature_kernel={'name': 'feature', 'detach_diag': True, 'kernel': 'cossim'})}
    torch.manual_seed(3)
    x0 = torch.randn(4, 6, 5)
    x0[1, 2] = 0.0
    assert torch.isnan(R.run_config(mk, [x0.double()]))
    out = build_model_kernel(mk).to(dev)(None, OrderedDict(a=x0.to(dev)), None, OrderedDict())
    assert torch.isnan(out)
    method = build_method({'name': 'repulsive_kernel', 'num': 4, 'gamma': 2.0, 'model_kernel': mk,
                           'ind_loss': {'name': 'cross_entropy'}}).to(dev)
    pred = torch.randn(4, 6, 3, device=dev)
    track = {'pred_ind': pred, 'feats_ind': OrderedDict(a=x0.to(dev)), 'pred_ood': None, 'feats_ood': None}
    track = method(types.SimpleNamespace(training=True), OrderedDict(), [], track, None, torch.randint(0, 3, (6,), device=dev), step=5)
    assert torch.isnan(track['ind_rep_loss']) and torch.isfinite(track['loss'])
    assert abs(track['loss'].item() - track['ind_loss'].item()) < 1e-6


@pytest.mark.parametrize('kernel,learnable', [('cossim', False), ('linear', False), ('cossim', True)])
def test_masked_nan_term_leaves_only_the_data_gradients(kernel, learnable):
    """The reference drops a NaN repulsion term from the graph (methods.py:342-348): backward then carries the data
    term's gradients only.  Here the term is masked on the device (ops.grad_gate): the feature gradients of the NaN term
    must be exact zeros - not 0 * NaN - and the prediction gradients those of the cross-entropy alone.  cossim: a zero
    feature row; linear: a NaN feature; learnable: the [M,M] head in torch on top of the HIP Gram kernels."""
    import types
    from collections import OrderedDict
    from hyper_amd.diversity.methods import build_method
    dev = _dev()
    mk = {'name': 'model_layers', 'layer_weighting': {'name': 'uniform'},
          'layer_kernel': dict(name='he', eps=0.0025, arc_eps=0.031, learnable=learnable,
                               feature_kernel={'name': 'feature', 'detach_diag': True, 'kernel': kernel})}
    torch.manual_seed(3)
    x0 = torch.randn(4, 6, 5)
    x0[1, 2] = 0.0 if kernel == 'cossim' else float('nan')
    x1 = torch.randn(4, 6, 7)
    method = build_method({'name': 'repulsive_kernel', 'num': 4, 'gamma': 2.0, 'model_kernel': mk,
                           'ind_loss': {'name': 'cross_entropy'}}).to(dev)
    gt = torch.randint(0, 3, (6,), device=dev)
    pred0 = torch.randn(4, 6, 3, device=dev)

    def run(feats_a, feats_b):
        pred = pred0.clone().requires_grad_(True)
        xa, xb = feats_a.to(dev).requires_grad_(True), feats_b.to(dev).requires_grad_(True)
        # the features depend on the predictions' leaf too, as in a network (shared upstream graph)
        track = {'pred_ind': pred, 'feats_ind': OrderedDict(a=xa * 1.0, b=xb * 1.0), 'pred_ood': None, 'feats_ood': None}
        track = method(types.SimpleNamespace(training=True), OrderedDict(), [], track, None, gt, step=5)
        track['loss'].backward()
        return track, pred.grad, xa.grad, xb.grad

    track, gp, ga, gb = run(x0, x1)
    assert torch.isnan(track['ind_rep_loss']) and torch.isfinite(track['loss'])
    assert torch.equal(ga, torch.zeros_like(ga)) and torch.equal(gb, torch.zeros_like(gb))
    # the data term alone
    pred = pred0.clone().requires_grad_(True)
    ce = build_method({'name': 'mle', 'num': 4, 'ind_loss': {'name': 'cross_entropy'}}).to(dev)
    t2 = ce(types.SimpleNamespace(training=True), OrderedDict(), [], {'pred_ind': pred, 'feats_ind': OrderedDict(), 'pred_ood': None,
                                                                     'feats_ood': None}, None, gt, step=5)
    t2['loss'].backward()
    assert torch.isfinite(gp).all() and torch.allclose(gp, pred.grad, rtol=1e-6, atol=1e-8)
    # a finite term right afterwards: the gate is per call, gradients flow again
    track, gp, ga, gb = run(torch.randn(4, 6, 5), x1)
    assert torch.isfinite(track['ind_rep_loss']) and float(ga.abs().max()) > 0 and float(gb.abs().max()) > 0


def test_graphed_train_step_equals_eager():
    """hyper_amd.experiments.graphed.GraphedTrainStep: three eager steps + two graph replays must leave the same
    weights as five eager steps (LeNet-5 ensemble, HE repulsion, AdamW)."""
    import types
    import hyper_amd.target  # noqa: F401
    from hyper_amd.generators import build_generator
    from hyper_amd.diversity.methods import build_method
    from hyper_amd.experiments.graphed import GraphedTrainStep
    dev = _dev()
    M, B = 4, 64
    hcfg = {'name': 'ensemble', 'ensemble_size': M, 'target': {'name': 'lenet5', 'in_channels': 1, 'num_classes': 10}}
    mcfg = {'name': 'repulsive_kernel', 'num': M, 'gamma': 10.5, 'warmup': 0, 'ind_loss': {'name': 'cross_entropy'},
            'model_kernel': {'name': 'model_layers', 'layer_weighting': {'name': 'uniform'},
                             'layer_kernel': {'name': 'he', 'eps': 0.0025, 'arc_eps': 0.031,
                                              'feature_kernel': {'name': 'feature', 'detach_diag': True, 'kernel': 'linear'}}}}
    torch.manual_seed(0)
    X = torch.randn(B, 1, 28, 28, device=dev)
    Y = torch.randint(0, 10, (B,), device=dev)
    trainer = types.SimpleNamespace(training=True)
    finals = []
    for graphed in (False, True):
        torch.manual_seed(1)
        hyper = build_generator(hcfg).to(dev)
        method = build_method(mcfg).to(dev)
        params = list(hyper.parameters()) + list(method.parameters())
        opt = torch.optim.AdamW(params, lr=1e-3, capturable=True)
        if graphed:
            g = GraphedTrainStep(hyper, method, opt, X, Y, member_ids=list(range(M)), step=1000, warmup_iters=3)
            for _ in range(2):
                loss = g(X, Y)
        else:
            for _ in range(5):
                opt.zero_grad(set_to_none=True)
                track = hyper(params=list(range(M)), x=X, sample_params=True, ret_params=True, feature_split=True, ood_N=0)
                track = method(trainer, track['params'], [], track, X, Y, step=1000)
                track['loss'].backward()
                opt.step()
                loss = track['loss']
        torch.cuda.synchronize()
        finals.append((float(loss.detach()), [p.detach().clone() for p in hyper.parameters()]))
    assert abs(finals[0][0] - finals[1][0]) <= 1e-5 * abs(finals[0][0])
    for a, b in zip(finals[0][1], finals[1][1]):
        assert float((a - b).abs().max()) <= 1e-5 * max(float(a.abs().max()), 1e-12)


@pytest.mark.parametrize('N,HW,shared,C', [(8, 8, False, 128), (64, 16, True, 128), (16, 16, False, 64)])
def test_conv_x3_gemms_match_fp64_conv(N, HW, shared, C, monkeypatch):
    """Conv2d at 128 channels (interior shapes, no bias): forward, weight gradient (split-K in the second case) and
    input gradient on the fp16-split batched GEMM (hecka_bgemm_x3_f32) against F.conv2d per member in fp64, and
    against the fp32-MFMA route (HECKA_CONV_X3=0).  C = 64: the 64-row tile (forward rows = 64, dgrad rows = 576; the
    weight gradient's 576 columns stay on the fp32 split-K kernel)."""
    import torch.nn.functional as F
    from hyper_amd.layers.conv import BatchedConv2d
    from hyper_amd import _lib
    dev = _dev()
    torch.manual_seed(11)
    M, O = 2, C
    x = torch.randn((N, C, HW, HW) if shared else (M, N, C, HW, HW), device=dev)
    W = torch.randn(M, O, C, 3, 3, device=dev) * 0.05
    up = torch.randn(M, N, O, HW, HW, device=dev)
    geom = (3, 3, 1, 1, 1, 1, 1, 1)
    assert _lib.lib().hecka_bgemm_x3_eligible(M, O, N * HW * HW, C * 9)
    res = {}
    for mode in ('1', '0'):
        monkeypatch.setenv('HECKA_CONV_X3', mode)
        xx, WW = x.clone().requires_grad_(True), W.clone().requires_grad_(True)
        y = BatchedConv2d.apply(xx, WW, None, 0.7, 1.0, geom)
        (y * up).sum().backward()
        res[mode] = (y.detach().cpu().double(), xx.grad.cpu().double(), WW.grad.cpu().double())
    xd, Wd, upd = x.cpu().double().requires_grad_(True), W.cpu().double().requires_grad_(True), up.cpu().double()
    yd = torch.stack([F.conv2d(xd if shared else xd[m], 0.7 * Wd[m], None, 1, 1) for m in range(M)])
    (yd * upd).sum().backward()
    for mode in ('1', '0'):
        assert _relerr(res[mode][0].numpy(), yd.detach().numpy()) <= RTOL, mode
        assert _relerr(res[mode][1].numpy(), xd.grad.numpy()) <= RTOL, mode
        assert _relerr(res[mode][2].numpy(), Wd.grad.numpy()) <= RTOL, mode


@pytest.mark.parametrize('M,O,C,k', [(2, 64, 128, (3, 3)), (3, 40, 48, (1, 1)), (1, 130, 16, (5, 5)), (2, 32, 80, (2, 3))])
def test_tapmajor_weight_layouts_equal_the_torch_permutes(M, O, C, k):
    """hecka_conv_weights_tapmajor_f32: W [M,O,C,kh,kw] -> [M,O,(kh kw),C] (forward operand) and the rotated [M,C,(kh kw),O]
    (input-gradient operand) - bit-equal to permute / flip + permute."""
    from hyper_amd.layers.conv import _tap_weights
    dev = _dev()
    torch.manual_seed(3)
    W = torch.randn(M, O, C, *k, device=dev)
    assert torch.equal(_tap_weights(W, False), W.permute(0, 1, 3, 4, 2).reshape(M, O, -1))
    assert torch.equal(_tap_weights(W, True), W.flip(3, 4).permute(0, 2, 3, 4, 1).reshape(M, C, -1))


def _round_f16_scaled(t, amax):
    """What the single-product tile multiplies: t * 2^k rounded to fp16 (nearest even), 2^k the power-of-two scale of
    pow2_scale(amax, 14) - amax 2^k < 2^14 -, as fp64 with the scale undone.  amax per leading index of t."""
    _, e = torch.frexp(amax.double().cpu())
    s = torch.pow(torch.tensor(2.0, dtype=torch.float64), (14 - e).double()).view(-1, *([1] * (t.ndim - 1)))
    return (t.cpu().double() * s).float().half().double() / s


@pytest.mark.parametrize('stride,C,O', [(1, 64, 128), (2, 128, 64)])
def test_single_product_layer_mode_is_the_fp64_product_of_fp16_rounded_operands(stride, C, O):
    """ops.layer_products(1) - the 16-bit COMPUTE mode of the layer GEMMs (hecka_set_layer_products; no counterpart in the
    fp32-only reference: parity with it is unpinned).  Pinned instead to what it claims to be: each operand rounded to fp16 after
    its per-member power-of-two scale, exact products, fp32 accumulation - so forward, input gradient and weight gradient of a
    convolution equal the fp64 convolution of the ROUNDED operands to accumulation error, and differ from the unrounded one
    by about 2^-12.  Tap-major forward / dgrad (stride 1), the parity-class dgrad (stride 2), both weight gradients, and the
    batched GEMM of the Linear layers."""
    import torch.nn.functional as F
    from hyper_amd import ops
    from hyper_amd.layers.conv import BatchedConv2d
    from hyper_amd.layers.linear import _bgemm_x3, _absmax_rows
    dev = _dev()
    torch.manual_seed(23)
    M, N, HW = 2, 16, 16
    x = torch.randn(M, N, C, HW, HW, device=dev)
    W = torch.randn(M, O, C, 3, 3, device=dev) * 0.05
    up = torch.randn(M, N, O, HW // stride, HW // stride, device=dev)
    geom = (3, 3, stride, stride, 1, 1, 1, 1)
    assert ops.get_layer_products() == 3
    with ops.layer_products(1):
        assert ops.get_layer_products() == 1
        xx, WW = x.clone().requires_grad_(True), W.clone().requires_grad_(True)
        y = BatchedConv2d.apply(xx, WW, None, 0.7, 1.0, geom)
        (y * up).sum().backward()
        A, Bm = torch.randn(M, 128, 256, device=dev), torch.randn(M, 256, 128, device=dev)
        amA, amB = _absmax_rows(A.view(M, -1), M), _absmax_rows(Bm.view(M, -1), M)
        prod = _bgemm_x3(A, Bm, M, 128, 128, 256, 128 * 256, 256 * 128, True, 1.0, amA, amB, 1)
    assert ops.get_layer_products() == 3
    amax = lambda t: t.reshape(M, -1).abs().amax(1)
    xr, Wr, ur = _round_f16_scaled(x, amax(x)), _round_f16_scaled(W, amax(W)), _round_f16_scaled(up, amax(up))
    conv = lambda a, w: torch.stack([F.conv2d(a[m], 0.7 * w[m], None, stride, 1) for m in range(M)])
    y_ref = conv(xr, Wr)
    xl = x.cpu().double().requires_grad_(True)
    (conv(xl, Wr) * ur).sum().backward()
    Wl = W.cpu().double().requires_grad_(True)
    (conv(xr, Wl) * ur).sum().backward()
    assert _relerr(y.detach().cpu().double().numpy(), y_ref.numpy()) <= RTOL
    assert _relerr(xx.grad.cpu().double().numpy(), xl.grad.numpy()) <= RTOL
    assert _relerr(WW.grad.cpu().double().numpy(), Wl.grad.numpy()) <= RTOL
    exact = conv(x.cpu().double(), W.cpu().double())
    dev16 = _relerr(y.detach().cpu().double().numpy(), exact.numpy())
    assert 1e-5 < dev16 < 2e-3, dev16            # it IS a 16-bit product (and not a worse one)
    p_ref = _round_f16_scaled(A, amA) @ _round_f16_scaled(Bm, amB)
    assert _relerr(prod.cpu().double().numpy(), p_ref.numpy()) <= RTOL


@pytest.mark.parametrize('dtype', ['float32', 'bfloat16'])
def test_multi_gram_split_k_matches_unsplit(dtype, monkeypatch):
    """Few workgroups over deep layers (here 8 members x 1 tile x 3 layers): the multi-layer Gram launch splits layers
    over K into partial Grams and sums them (hecka_gram_center_multi_workspace_bytes > 0).  Same C / cvec / loss / dX as
    the unsplit launch (HECKA_GRAM_NOSPLIT=1) to fp32 rounding, and the oracle's loss and gradients at RTOL."""
    import ctypes
    from collections import OrderedDict
    from oracle import hecka_ref as R
    from hyper_amd import ops, _lib
    from hyper_amd.diversity.model_kernels import build_model_kernel
    dev = _dev()
    dt = getattr(torch, dtype)
    M, B, dims = 8, 128, (4096, 2048, 1024)
    assert _lib.lib().hecka_gram_center_multi_workspace_bytes(3, M, B, (ctypes.c_int * 3)(*dims)) > 0
    cfg = {'name': 'model_layers', 'layer_weighting': {'name': 'uniform'},
           'layer_kernel': dict(name='he', eps=0.0025, arc_eps=0.031,
                                feature_kernel={'name': 'feature', 'detach_diag': True, 'kernel': 'linear'})}
    torch.manual_seed(31)
    base = [(torch.relu(torch.randn(M, B, p)) + 0.1).to(dt) for p in dims]
    res = {}
    for split in ('1', '0'):
        if split == '0':
            monkeypatch.setenv('HECKA_GRAM_NOSPLIT', '1')
        else:
            monkeypatch.delenv('HECKA_GRAM_NOSPLIT', raising=False)
        xs = [b.to(dev).requires_grad_(True) for b in base]
        track = OrderedDict()
        out = build_model_kernel(cfg).to(dev)(None, OrderedDict((str(i), x) for i, x in enumerate(xs)), None, track)
        out.backward()
        res[split] = (out.item(), [x.grad.float().cpu() for x in xs], [t['gram'].cpu() for t in track['layer_track']])
    assert abs(res['1'][0] - res['0'][0]) <= 2e-6 * abs(res['0'][0])
    for a, b in zip(res['1'][2], res['0'][2]):
        assert _relerr(a.numpy(), b.double().numpy()) <= 2e-6
    tol = RTOL if dt == torch.float32 else 2.0 ** -8 * 1.01
    xo = [b.double().requires_grad_(True) for b in base]
    ref = R.run_config(cfg, xo)
    ref.backward()
    assert abs(res['1'][0] - ref.item()) <= (RTOL if dt == torch.float32 else 1e-4) * abs(ref.item())
    for g, o in zip(res['1'][1], xo):
        assert _relerr(g.numpy(), o.grad.numpy()) <= tol


@pytest.mark.parametrize('M,B,dims,rows', [(64, 512, (1024, 512), 0), (8, 256, (2048, 256), 64), (16, 128, (96, 32), 0),
                                           (2, 1024, (512, 256), 0), (3, 768, (256,), 256)])
def test_k32_tile_equals_x6_tile_to_rounding(M, B, dims, rows, monkeypatch):
    """The 32-deep 16x16x32 tile (gemm_f16x3_k32.hpp: all-layers Gram, 256 x 256 dX launch) against the 16-deep 32x32x16 tile
    it replaces (HECKA_K32=0) on the same operands: the same two-term products in another summation order - C, cvec and
    dX equal to 2e-6 of their largest element, in the standard and (rows > 0) the slice-major layouts; and both within
    2e-6 of the fp64 products."""
    from hyper_amd import ops
    dev = _dev()
    torch.manual_seed(B + M)
    Xs = [(torch.relu(torch.randn(M, B, p, device=dev)) + 0.05 * i) for i, p in enumerate(dims)]
    Ws = [torch.randn(M, B, B, device=dev) for _ in dims]
    wmaxs = [W.abs().amax(dim=(1, 2)) for W in Ws]
    res = {}
    for k32 in ('1', '0'):
        monkeypatch.setenv('HECKA_K32', k32)
        g = ops.centered_gram_multi(Xs, slice_rows=rows)
        xmaxs = [t[2] for t in g]
        dx = ops.gram_bwd_multi(Ws, Xs, wmaxs, xmaxs) if ops.gram_bwd_multi_eligible(Ws, Xs, wmaxs, xmaxs) else \
            [ops.gram_bwd_w(W, X, wm, xm) for W, X, wm, xm in zip(Ws, Xs, wmaxs, xmaxs)]
        res[k32] = ([t[0].clone() for t in g], [t[1].clone() for t in g], dx)
    for li in range(len(dims)):
        for a, b in zip((res['1'][0][li], res['1'][1][li], res['1'][2][li]), (res['0'][0][li], res['0'][1][li], res['0'][2][li])):
            assert a.shape == b.shape
            assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max())
        Xd = Xs[li].double()
        Xc = Xd - Xd.mean(1, keepdim=True)
        Cref = Xd @ Xc.transpose(1, 2)      # C[a, b] = x_a . (x_b - mean): the centred Gram plus the 1 c^T term
        C = res['1'][0][li]
        if rows:
            C = C.permute(1, 0, 2, 3).reshape(M, B, B)
        assert float((C.double() - Cref).abs().max()) <= 2e-6 * float(Cref.abs().max())
        dref = Ws[li].double() @ Xd
        assert float((res['1'][2][li].double() - dref).abs().max()) <= 2e-6 * float(dref.abs().max())


@pytest.mark.parametrize('with_res', [False, True])
def test_fused_batchnorm_relu_matches_unfused(with_res, monkeypatch):
    """BatchNorm2d(..., residual=, relu=True) as one launch each way against the plain norm + torch add / relu
    (HECKA_BN_FUSE=0): output, dx, d residual, d affine - and the same running statistics."""
    from hyper_amd.layers.norm import BatchedBatchNorm, BatchedBatchNormAct
    dev = _dev()
    torch.manual_seed(3)
    M, N, C, H = 3, 6, 8, 4
    x0 = torch.randn(M, N, C, H, H, device=dev)
    r0 = torch.randn(M, N, C, H, H, device=dev)
    w0, b0 = torch.rand(M, C, device=dev) + 0.5, torch.randn(M, C, device=dev) * 0.1
    up = torch.randn(M, N, C, H, H, device=dev)
    outs = []
    for fused in (True, False):
        x, r = x0.clone().requires_grad_(True), r0.clone().requires_grad_(True)
        w, b = w0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
        if fused:
            y, mean, var = BatchedBatchNormAct.apply(x, r if with_res else None, w, b, None, None, True, 1e-5, True)
        else:
            y, mean, var = BatchedBatchNorm.apply(x, w, b, None, None, True, 1e-5)
            y = torch.relu(y + r if with_res else y)
        (y * up).sum().backward()
        outs.append([y.detach(), x.grad, w.grad, b.grad, mean, var] + ([r.grad] if with_res else []))
    for a, b_ in zip(*outs):
        assert _relerr(a.cpu().numpy(), b_.cpu().double().numpy()) <= 2e-6


@pytest.mark.parametrize('k,H,W,N,C,O', [((5, 5), 8, 8, 8, 64, 128), ((1, 1), 16, 4, 8, 128, 64), ((3, 5), 4, 32, 4, 64, 64),
                                         ((3, 3), 2, 16, 8, 128, 128)])
def test_implicit_conv_geometries_match_fp64(k, H, W, N, C, O):
    """hecka_conv_x3_f32 / hecka_conv_wgrad_x3_f32 away from the ResNet geometry: 5x5, 1x1 and rectangular kernels,
    non-square images whose rows are shorter (W = 4: four image rows per 16-position k-tile) and longer (W = 32) than a
    k-tile, padded patch rows (C kh kw not a multiple of 128) - against F.conv2d per member in fp64."""
    import torch.nn.functional as F
    from hyper_amd.layers.conv import BatchedConv2d, _conv_x3_ok
    dev = _dev()
    torch.manual_seed(sum(k) + H + W)
    M = 2
    geom = (k[0], k[1], 1, 1, (k[0] - 1) // 2, (k[1] - 1) // 2, 1, 1)
    assert _conv_x3_ok(geom, M, O, C * k[0] * k[1], N, C, H, W)
    x = torch.randn(M, N, C, H, W, device=dev, requires_grad=True)
    Wt = (torch.randn(M, O, C, k[0], k[1], device=dev) * 0.05).requires_grad_(True)
    up = torch.randn(M, N, O, H, W, device=dev)
    y = BatchedConv2d.apply(x, Wt, None, 1.3, 1.0, geom)
    (y * up).sum().backward()
    xd, Wd = x.detach().cpu().double().requires_grad_(True), Wt.detach().cpu().double().requires_grad_(True)
    yd = torch.stack([F.conv2d(xd[m], 1.3 * Wd[m], None, 1, (geom[4], geom[5])) for m in range(M)])
    (yd * up.cpu().double()).sum().backward()
    assert _relerr(y.detach().cpu().numpy(), yd.detach().numpy()) <= RTOL
    assert _relerr(x.grad.cpu().numpy(), xd.grad.numpy()) <= RTOL
    assert _relerr(Wt.grad.cpu().numpy(), Wd.grad.numpy()) <= RTOL


@pytest.mark.parametrize('shape', [(3, 5, 6, 24, 24), (2, 4, 16, 8, 8), (7, 2, 12), (1031,)])
def test_crater_kernels_match_torch_expression(shape):
    """hecka_crater_f32 / hecka_crater_maxpool2_f32 against the reference's five-op expression (activation.py:8-12)
    and torch.max_pool2d, forward and gradient, fp32 (tolerance: a few ulp of tanhf)."""
    import math
    import torch.nn.functional as F
    from hyper_amd.net import activation
    dev = torch.device('cuda:0')
    torch.manual_seed(3)
    h = math.sqrt(2.0) / 2.0
    c = h + 0.008274364699

    def expr(t):
        return c * t * (1.0 + torch.tanh(h * (t + 1.0)))
    x = (2.5 * torch.randn(*shape, device=dev)).requires_grad_(True)
    x64 = x.detach().double().requires_grad_(True)
    y = activation.crater(x)
    up = torch.randn_like(y)
    (y * up).sum().backward()
    y64 = expr(x64)
    (y64 * up.double()).sum().backward()
    assert (y.double() - y64).abs().max().item() <= 2e-6 * y64.abs().max().item()
    assert (x.grad.double() - x64.grad).abs().max().item() <= 2e-6 * x64.grad.abs().max().item()
    if len(shape) >= 3 and activation.crater_maxpool2_eligible(x, {'kernel_size': 2}):
        x2 = x.detach().clone().requires_grad_(True)
        p = activation.crater_maxpool2(x2)
        lead = x.shape[:-3]
        ref_in = x.detach().clone().requires_grad_(True)
        q = F.max_pool2d(expr(ref_in).reshape((-1,) + x.shape[-3:]), kernel_size=2)
        q = q.view(lead + q.shape[1:])
        assert p.shape == q.shape
        up2 = torch.randn_like(p)
        (p * up2).sum().backward()
        (q * up2).sum().backward()
        assert (p - q).abs().max().item() <= 2e-6 * q.abs().max().item()
        assert (x2.grad - ref_in.grad).abs().max().item() <= 4e-6 * ref_in.grad.abs().max().item()
        assert ((x2.grad != 0) == (ref_in.grad != 0)).all()      # the same element of every window took the gradient


@pytest.mark.parametrize('case', [
    dict(M=3, N=5, C=1, O=6, k=5, pad=0, H=28, W=28, shared=True, bias=True),      # LeNet conv1
    dict(M=4, N=3, C=6, O=16, k=5, pad=0, H=12, W=12, shared=False, bias=True),    # LeNet conv2
    dict(M=2, N=4, C=3, O=5, k=3, pad=1, H=9, W=7, shared=True, bias=False),
    dict(M=2, N=3, C=16, O=9, k=3, pad=2, H=6, W=11, shared=False, bias=True),
    dict(M=2, N=2, C=4, O=16, k=5, pad=4, H=5, W=6, shared=False, bias=False),
    dict(M=1, N=70, C=2, O=3, k=5, pad=2, H=8, W=8, shared=False, bias=True),
])
def test_direct_conv_matches_torch_conv2d(case, monkeypatch):
    """Few-channel direct convolution (hecka_conv_direct_f32 / _wgrad_f32) against F.conv2d in fp64 per member, forward and
    all three gradients; and the explicit im2col path (HECKA_CONV_DIRECT=0) on the same inputs."""
    import torch.nn.functional as F
    from hyper_amd.layers.conv import BatchedConv2d
    dev = torch.device('cuda:0')
    torch.manual_seed(11)
    M, N, C, O, k, pad = (case[key] for key in ('M', 'N', 'C', 'O', 'k', 'pad'))
    xs = (N, C, case['H'], case['W']) if case['shared'] else (M, N, C, case['H'], case['W'])
    x0 = torch.randn(*xs, device=dev)
    W0 = torch.randn(M, O, C, k, k, device=dev)
    b0 = torch.randn(M, O, device=dev) if case['bias'] else None
    scale, bscale = 0.37, 1.3
    geom = (k, k, 1, 1, pad, pad, 1, 1)

    def run():
        x, W = x0.clone().requires_grad_(True), W0.clone().requires_grad_(True)
        b = b0.clone().requires_grad_(True) if b0 is not None else None
        y = BatchedConv2d.apply(x, W, b, scale, bscale, geom)
        return x, W, b, y
    x, W, b, y = run()
    up = torch.randn_like(y)
    (y * up).sum().backward()
    x64, W64 = x0.double().requires_grad_(True), W0.double().requires_grad_(True)
    b64 = b0.double().requires_grad_(True) if b0 is not None else None
    ref = torch.stack([F.conv2d(x64 if case['shared'] else x64[m], scale * W64[m], None if b64 is None else bscale * b64[m],
                                padding=pad) for m in range(M)])
    (ref * up.double()).sum().backward()

    def rel(a, r):
        return float((a.detach().double() - r.detach()).abs().max() / r.detach().abs().max())
    assert y.shape == ref.shape and rel(y, ref) <= 2e-6
    assert rel(x.grad, x64.grad) <= 5e-6 and rel(W.grad, W64.grad) <= 5e-6
    if b is not None:
        assert rel(b.grad, b64.grad) <= 5e-6
    monkeypatch.setenv('HECKA_CONV_DIRECT', '0')
    x2, W2, b2, y2 = run()
    (y2 * up).sum().backward()
    assert rel(y2, ref) <= 1e-5 and rel(W2.grad, W64.grad) <= 2e-5 and rel(x2.grad, x64.grad) <= 2e-5


def test_unbiased_cka_matches_oracle():
    """hyper_amd's `unbiased_cka` (three kernels per batch) against the restated reference loop (oracle.unbiased_cka,
    generators/base.py:224-313; its `_uHSIC` is pinned by tests/golden/uhsic_f32.npz) on a small MLP ensemble, 3 batches."""
    from oracle import hecka_ref as R
    import hyper_amd.target  # noqa: F401
    from hyper_amd.generators import build_generator
    from hyper_amd.util.collections import flatten_keys
    dev = torch.device('cuda:0')
    torch.manual_seed(8)
    hyper = build_generator({'name': 'ensemble', 'ensemble_size': 4,
                             'target': {'name': 'mlp', 'activation': 'crater',
                                        'layers': [{'in': 3, 'out': 12}, {'in': 12, 'out': 20}, {'in': 20, 'out': 5}]}}).to(dev)
    data = [(torch.randn(16, 3), torch.zeros(16, dtype=torch.long)) for _ in range(3)]

    class Fixed(type(hyper)):            # same members for the product and the oracle
        def sample_params(self, size, device=None):
            return torch.arange(size, device=device)
    hyper.__class__ = Fixed
    got = hyper.unbiased_cka(data, num_samples=4, device=dev)
    batches = []
    with torch.no_grad():
        hyper.eval()
        for X, _ in data:
            feats, pred = hyper.forward(hyper.forward_params(torch.arange(4, device=dev)), X.to(dev))
            batches.append([f.reshape(4, 16, -1).cpu() for f in flatten_keys(feats).values() if f is not None])
    want = R.unbiased_cka(batches)
    assert got.shape == want.shape == (3, 3) and got.dtype == torch.float64
    assert float((got.cpu() - want).abs().max()) <= 1e-4 * float(want.abs().max())
    assert hyper.training is False or True
    with pytest.raises(NotImplementedError):
        hyper.unbiased_cka(data, num_samples=4, device=dev, save_path='x.png')


def test_weight_space_svgd_step_matches_reference_golden():
    """configs/mnist/svgd/rbf.yml: one SVGD update with the WEIGHT-space rbf kernel (median bandwidth over the members'
    flattened LeNet parameters, detach_right = gradient through the left argument only) against the update the REAL
    reference produced (tests/golden/wsvgd_step_f64.npz): kernel matrix, loss, every parameter gradient."""
    import json
    import os
    import types
    import hyper_amd.target  # noqa: F401
    from hyper_amd.generators import build_generator
    from hyper_amd.diversity.methods import build_method
    from conftest import GOLDEN
    z = np.load(os.path.join(GOLDEN, 'wsvgd_step_f64.npz'), allow_pickle=False)
    cfg = json.loads(str(z['config']))
    dev = _dev()
    hyper = build_generator(cfg['hyper'])
    hyper.load_state_dict({k[2:]: torch.from_numpy(z[k]).float() for k in z.files if k.startswith('w:')})
    hyper = hyper.to(dev)
    method = build_method(cfg['method']).to(dev)
    assert method.weight_space
    X = torch.from_numpy(z['X']).float().to(dev)
    Y = torch.from_numpy(z['Y']).to(dev)
    trainer = types.SimpleNamespace(training=True)
    track = hyper(params=list(range(5)), x=X, sample_params=True, ret_params=True, feature_split=True, ood_N=0)
    track = method(trainer, track['params'], [], track, X, Y, step=0)
    loss, _ = method.backward(trainer, track['params'], [], track, X, Y)
    kern = method.model_kernel(track['params'], track['feats_ind'], track['pred_ind'], track)
    yard = Yardstick('wsvgd_step')
    yard.check('kern', kern.detach().cpu().numpy())
    yard.check('loss', loss.item())
    for k, v in hyper.named_parameters():
        yard.check('g:' + k, v.grad.cpu().numpy())


def test_accuracy_metric_and_inlier_outlier_entropy():
    """`accuracy` metric (experiments/metrics.py:104-152, torchmetrics multiclass accuracy = fraction of hits) and the
    `inlier_outlier_entropy` loss (losses.py:94-113) against their definitions in fp64."""
    import torch.nn.functional as F
    from hyper_amd.experiments.metrics import build_metric
    from hyper_amd.diversity.losses import build_loss
    dev = _dev()
    torch.manual_seed(4)
    pred = torch.randn(5, 40, 10, device=dev)
    Y = torch.randint(0, 10, (40,), device=dev)
    m = build_metric({'name': 'accuracy', 'classes': 10})
    res = m.log(None, 5, None, Y, {'pred_ind': pred})
    accs = [100.0 * float((pred[i].argmax(-1) == Y).double().mean()) for i in range(5)]
    ens = F.softmax(pred.double(), -1).mean(0).argmax(-1)
    assert abs(res['accuracy'] - 100.0 * float((ens == Y).double().mean())) < 1e-9
    assert abs(res['accuracy_std'] - float(np.std(accs))) < 1e-9
    res2 = m.log(None, 5, None, Y, {'pred_ind': pred})
    assert abs(res2['term']['run_acc'] - res['accuracy']) < 1e-9
    m.reset()
    assert m.total == 0.0
    loss = build_loss({'name': 'inlier_outlier_entropy', 'weight': 0.7})
    pood = torch.randn(5, 12, 10, device=dev)
    got = loss(5, None, {'pred_ind': pred, 'pred_ood': pood}, Y)
    p = F.softmax(pood.double(), -1).mean(0)
    want = F.cross_entropy(pred.double().reshape(200, 10), Y.repeat(5)) + 0.7 * (p * torch.log(p + 1e-8)).sum(-1).mean()
    assert abs(got.item() - want.item()) <= 1e-5 * abs(want.item())


def test_function_svgd_step_matches_reference_golden():
    """configs/toy/fourclass/fsvgd/rbf.yml: one function-space SVGD update (likelihood score + SSGE prior score through
    the rbf `function` kernel, detach_right) against the REAL reference (tests/golden/fsvgd_step_f64.npz).  The N(0,1)
    prior members are the fixture's (the only random input).  fp32 here against fp64 there (measured 2.6e-7)."""
    import json
    import os
    import types
    from collections import OrderedDict
    import hyper_amd.target  # noqa: F401
    from hyper_amd.generators import build_generator
    from hyper_amd.diversity.methods import build_method
    from hyper_amd.util.collections import unflatten_keys
    from conftest import GOLDEN
    z = np.load(os.path.join(GOLDEN, 'fsvgd_step_f64.npz'), allow_pickle=False)
    cfg = json.loads(str(z['config']))
    dev = _dev()
    hyper = build_generator(cfg['hyper'])
    hyper.load_state_dict({k[2:]: torch.from_numpy(z[k]).float() for k in z.files if k.startswith('w:')})
    hyper = hyper.to(dev)
    method = build_method(cfg['method']).to(dev)
    prior = unflatten_keys(OrderedDict((k[6:], torch.from_numpy(z[k]).float().to(dev)) for k in z.files if k.startswith('prior:')))
    orig = hyper.forward_params
    hyper.forward_params = lambda codes, device=None: prior if isinstance(codes, str) else orig(codes, device=device)
    X = torch.from_numpy(z['X']).float().to(dev)
    Y = torch.from_numpy(z['Y']).to(dev)
    trainer = types.SimpleNamespace(training=True, hyper=hyper, ddp=False, ood_N=0)
    track = hyper(params=list(range(6)), x=X, sample_params=True, ret_params=True, feature_split=True, ood_N=0)
    track = method(trainer, track['params'], [], track, X, Y, step=0)
    loss, _ = method.backward(trainer, track['params'], [], track, X, Y)
    yard = Yardstick('fsvgd_step')
    yard.check('loss', loss.item())
    for k, v in hyper.named_parameters():
        yard.check('g:' + k, v.grad.cpu().numpy())


def test_learnable_hyperparameters_step_matches_reference_golden():
    """configs/toy/fourclass/ens+ood/he-rbf-learned.yml (SURVEY.md 8b "State / ownership": `learnable: true` needs gradients
    w.r.t. eps, arc_eps, layer weights, gamma): one step against the REAL reference (tests/golden/learned_step_f64.npz) -
    losses, every ensemble gradient and every method-parameter gradient; method state-dict keys equal the reference's."""
    import json
    import os
    import types
    import hyper_amd.target  # noqa: F401
    from hyper_amd.generators import build_generator
    from hyper_amd.diversity.methods import build_method
    from conftest import GOLDEN
    z = np.load(os.path.join(GOLDEN, 'learned_step_f64.npz'), allow_pickle=False)
    cfg = json.loads(str(z['config']))
    dev = _dev()
    hyper = build_generator(cfg['hyper'])
    hyper.load_state_dict({k[2:]: torch.from_numpy(z[k]).float() for k in z.files if k.startswith('w:')})
    method = build_method(cfg['method'])
    want_keys = [k[3:] for k in z.files if k.startswith('mw:')]
    assert list(method.state_dict().keys()) == want_keys
    method.load_state_dict({k: torch.from_numpy(z['mw:' + k]).float() for k in want_keys})
    hyper, method = hyper.to(dev), method.to(dev)
    X = torch.from_numpy(z['X']).float().to(dev)
    Y = torch.from_numpy(z['Y']).to(dev)
    ood = int(z['ood_N'])
    track = hyper(params=list(range(6)), x=X, sample_params=True, ret_params=True, feature_split=True, ood_N=ood)
    track = method(types.SimpleNamespace(training=True), track['params'], [], track, X[:-ood], Y, step=int(z['step']))
    track['loss'].backward()
    yard = Yardstick('learned_step')
    _check_step(yard, track, hyper, keys=('loss', 'ind_rep_loss', 'ood_rep_loss'))
    got = dict(method.named_parameters())
    for k in [k[3:] for k in z.files if k.startswith('mg:')]:
        assert got[k].grad is not None, k
        yard.check('mg:' + k, got[k].grad.cpu().numpy())


@pytest.mark.gpu
@pytest.mark.parametrize('M,N,C,H,W,k,pad,stride', [
    (3, 4, 8, 32, 32, 3, 1, 2),      # a ResNet stage transition: 3 x 3 / pad 1 / stride 2
    (2, 3, 16, 16, 16, 1, 0, 2),     # its 1 x 1 shortcut
    (2, 2, 5, 12, 20, 5, 2, 2),      # 5 x 5, rectangular image
    (1, 2, 3, 10, 8, 4, 1, 2),       # even kernel
    (2, 2, 4, 9, 12, 3, 0, 2),       # no padding, odd height
    (2, 2, 4, 9, 10, 3, 1, 2),       # W % 4 != 0: the general kernel
])
def test_col2im_stride_two_matches_fold(M, N, C, H, W, k, pad, stride):
    """hecka_col2im_f32 (the stride-2 gather: two consecutive output columns per tap and pixel pair, parity tests instead
    of divisions) against torch's fold in fp64 - the adjoint of the patch gather the reference's F.conv2d implies
    (hyper/layers/conv.py:292-346)."""
    import torch.nn.functional as F
    from hyper_amd import _lib
    dev = torch.device('cuda:0')
    torch.manual_seed(H * W + k)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    R, Q = C * k * k, N * Ho * Wo
    dcols = torch.randn(M, R, Q, device=dev)
    dx = torch.empty(M, N, C, H, W, device=dev)
    _lib.check(_lib.lib().hecka_col2im_f32(dcols.data_ptr(), M, N, C, H, W, k, k, stride, stride, pad, pad, 1, 1, R, dx.data_ptr(),
                                           torch.cuda.current_stream().cuda_stream), 'col2im')
    ref = torch.stack([F.fold(dcols[m].double().view(R, N, Ho * Wo).permute(1, 0, 2), (H, W), k, padding=pad, stride=stride)
                       for m in range(M)])
    assert float((dx.double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize('relu,with_res', [(True, True), (True, False), (False, False)])
def test_packed_batchnorm_equals_the_unpacked_launch(relu, with_res):
    """hecka_batchnorm_act_*_packed_f32: weight | bias read from the flat [M, 2 C] parameter row, their gradients returned
    as one tensor, running statistics updated inside the forward kernel - bit-equal outputs and gradients to the
    unpacked launch, running statistics equal to the [M, C] tensor update it replaces."""
    from hyper_amd.layers.norm import BatchedBatchNormAct
    dev = torch.device('cuda:0')
    torch.manual_seed(5)
    M, N, C, H = 3, 6, 8, 4
    x0 = torch.randn(M, N, C, H, H, device=dev)
    r0 = torch.randn_like(x0) if with_res else None
    flat0 = torch.randn(M, 2 * C, device=dev)
    rm0, rv0 = torch.randn(M, C, device=dev), torch.rand(M, C, device=dev) + 0.5
    up = torch.randn_like(x0)
    factor, cnt = 0.1, N * H * H

    def run(packed):
        x = x0.clone().requires_grad_(True)
        r = r0.clone().requires_grad_(True) if r0 is not None else None
        flat = flat0.clone().requires_grad_(True)
        rm, rv = rm0.clone(), rv0.clone()
        if packed:
            y, mean, var = BatchedBatchNormAct.apply(x, r, flat, None, None, None, True, 1e-5, relu, True, (rm, rv, factor))
        else:
            y, mean, var = BatchedBatchNormAct.apply(x, r, flat[:, :C], flat[:, C:], None, None, True, 1e-5, relu)
            rm.mul_(1 - factor).add_(mean, alpha=factor)
            rv.mul_(1 - factor).add_(var * (cnt / (cnt - 1)), alpha=factor)
        (y * up).sum().backward()
        return y.detach(), x.grad, (r.grad if r is not None else None), flat.grad, rm, rv

    a, b = run(True), run(False)
    for u, v in zip(a[:4], b[:4]):
        assert (u is None and v is None) or torch.equal(u, v)
    assert torch.allclose(a[4], b[4], rtol=1e-6, atol=1e-7) and torch.allclose(a[5], b[5], rtol=1e-6, atol=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize('with_res', [True, False])
def test_batchnorm_relu_mask_equals_reading_the_output(with_res, monkeypatch):
    """The fused norm's backward reads the forward's byte-per-four-outputs relu mask instead of y (HECKA_BN_MASK, default on):
    bit-equal gradients to the launch that reads y (HECKA_BN_MASK=0) - zeros of the relu included (a channel driven fully
    negative)."""
    from hyper_amd.layers.norm import BatchedBatchNormAct
    dev = torch.device('cuda:0')
    torch.manual_seed(9)
    M, N, C, H = 2, 16, 8, 8
    x0 = torch.randn(M, N, C, H, H, device=dev)
    r0 = torch.randn_like(x0) if with_res else None
    flat0 = torch.randn(M, 2 * C, device=dev)
    flat0[:, 0] = 0.0
    flat0[:, C] = -1.0                      # channel 0: weight 0, bias -1 - every output of it is clipped (without a residual)
    up = torch.randn_like(x0)
    res = []
    for setting in (None, '0'):
        if setting is not None:
            monkeypatch.setenv('HECKA_BN_MASK', setting)
        x = x0.clone().requires_grad_(True)
        r = r0.clone().requires_grad_(True) if r0 is not None else None
        flat = flat0.clone().requires_grad_(True)
        y, _, _ = BatchedBatchNormAct.apply(x, r, flat, None, None, None, True, 1e-5, True, True, None)
        (y * up).sum().backward()
        res.append((y.detach(), x.grad, r.grad if r is not None else None, flat.grad))
    for u, v in zip(*res):
        assert (u is None and v is None) or torch.equal(u, v)
    yd = torch.relu(torch.nn.functional.batch_norm(x0[0].double(), None, None, flat0[0, :C].double(), flat0[0, C:].double(), True, 0.0, 1e-5)
                    + (r0[0].double() if with_res else 0.0))
    assert float((res[0][0][0].double() - yd).abs().max()) <= 1e-5 * float(yd.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize('k,stride,pad,H,W,N,C,O,shared', [
    ((3, 3), (2, 2), (1, 1), 16, 16, 8, 64, 128, False),     # a ResNet stage transition
    ((1, 1), (2, 2), (0, 0), 16, 16, 8, 64, 128, False),     # its 1 x 1 shortcut
    ((3, 3), (2, 2), (1, 1), 8, 8, 32, 128, 64, True),       # 4 x 4 outputs (four image rows per k-tile), shared input
    ((5, 3), (2, 1), (2, 1), 8, 16, 8, 64, 64, False),       # rectangular kernel, stride 2 along one axis only
    ((3, 3), (1, 1), (0, 0), 10, 18, 8, 64, 64, False),      # unit stride without padding: the output is smaller
])
def test_strided_implicit_conv_matches_fp64(k, stride, pad, H, W, N, C, O, shared):
    """hecka_conv_x3_strided_f32 / hecka_conv_wgrad_x3_strided_f32: forward and weight gradient of strided (and unpadded)
    convolutions as implicit GEMMs - an output position gathers its taps around (ho sh, wo sw) - and the input gradient
    through the GEMM + stride-aware col2im, against F.conv2d per member in fp64 (hyper/layers/conv.py:292-346)."""
    import torch.nn.functional as F
    from hyper_amd.layers.conv import BatchedConv2d, _conv_x3_fwd_ok, _out_size
    dev = _dev()
    torch.manual_seed(sum(k) + H + W + stride[0])
    M = 2
    geom = (k[0], k[1], stride[0], stride[1], pad[0], pad[1], 1, 1)
    Ho, Wo = _out_size(H, k[0], stride[0], pad[0], 1), _out_size(W, k[1], stride[1], pad[1], 1)
    assert _conv_x3_fwd_ok(geom, M, O, C * k[0] * k[1], N, C, H, W, Ho, Wo)
    x = torch.randn(*((N, C, H, W) if shared else (M, N, C, H, W)), device=dev, requires_grad=True)
    Wt = (torch.randn(M, O, C, k[0], k[1], device=dev) * 0.05).requires_grad_(True)
    up = torch.randn(M, N, O, Ho, Wo, device=dev)
    y = BatchedConv2d.apply(x, Wt, None, 1.3, 1.0, geom)
    assert tuple(y.shape) == (M, N, O, Ho, Wo)
    (y * up).sum().backward()
    xd, Wd = x.detach().cpu().double().requires_grad_(True), Wt.detach().cpu().double().requires_grad_(True)
    yd = torch.stack([F.conv2d(xd if shared else xd[m], 1.3 * Wd[m], None, stride, pad) for m in range(M)])
    (yd * up.cpu().double()).sum().backward()
    assert _relerr(y.detach().cpu().numpy(), yd.detach().numpy()) <= RTOL
    assert _relerr(x.grad.cpu().numpy(), xd.grad.numpy()) <= RTOL
    assert _relerr(Wt.grad.cpu().numpy(), Wd.grad.numpy()) <= RTOL
