"""torch.autograd.Function wrappers around libhecka (raw device pointers + the current HIP stream).

PyTorch is plumbing here: it owns the device memory and the stream; all arithmetic on [M,B,P]
and [M,B,B] tensors happens in the HIP kernels."""
import contextlib
import os
import threading

import torch

from . import _lib
from .config import knobs


# bench.py sets this to {'gram_bwd': []} to collect (start_event, end_event, flops) around the dominant
# kernel's launches on the launch stream; None in normal operation.
PROFILE = None
LAST_BWD_KERNEL = None     # name of the kernel the last dX launch took (bench.py's `roofline.kernel`)
LAST_SPEC_WS = None        # (workspace, index of its fallback flag) of the last speculative Gram call (tests)


# ---- arithmetic of the two large GEMMs (include/hecka.h HECKA_GEMM_*): 0 auto (f16x3), 1 fp32 MFMA, 2 bf16x6 ---------------
# A per-call argument of the C-ABI; here: a process default (HECKA_GEMM_MODE, set_gemm_mode) that `with gemm_mode(m):`
# overrides for the calling thread.  Autograd nodes remember the mode of their forward and use it in backward (which
# runs on another thread), so two kernels in one process may differ.
_MODE_NAMES = (0, 1, 2)
_mode_default = [int(os.environ.get('HECKA_GEMM_MODE', '0'))]
_mode_tls = threading.local()


def set_gemm_mode(mode):
    if int(mode) not in _MODE_NAMES:
        raise ValueError('gemm mode must be 0 (auto: f16x3), 1 (fp32 MFMA) or 2 (bf16x6)')
    _mode_default[0] = int(mode)


def get_gemm_mode():
    m = getattr(_mode_tls, 'mode', None)
    return _mode_default[0] if m is None else m


@contextlib.contextmanager
def gemm_mode(mode):
    if mode is not None and int(mode) not in _MODE_NAMES:
        raise ValueError('gemm mode must be 0 (auto: f16x3), 1 (fp32 MFMA) or 2 (bf16x6)')
    prev = getattr(_mode_tls, 'mode', None)
    _mode_tls.mode = None if mode is None else int(mode)
    try:
        yield
    finally:
        _mode_tls.mode = prev


def set_layer_products(n):
    """Products per multiply of the LAYER GEMMs / convolutions on the fp16-split tile (hecka_set_layer_products): 3 = fp32-grade
    (default), 1 = 16-bit compute (operands rounded to fp16 after their power-of-two scale, fp32 accumulation).  Process-wide
    (the library's setting, also read from HECKA_LAYER_PRODUCTS); the loss path is not touched.  Returns the previous value."""
    old = _lib.lib().hecka_set_layer_products(int(n))
    if old < 0:
        raise ValueError('layer products must be 3 (fp32-grade) or 1 (16-bit compute)')
    return old


def get_layer_products():
    return int(_lib.lib().hecka_get_layer_products())


@contextlib.contextmanager
def layer_products(n):
    prev = set_layer_products(n)
    try:
        yield
    finally:
        set_layer_products(prev)


def mode_aware(cls):
    """Class decorator for autograd Functions that launch the large GEMMs: backward runs on autograd's thread, so the
    node remembers the GEMM mode its forward ran under and re-enters it."""
    fwd, bwd = cls.forward, cls.backward

    def forward(ctx, *args, **kwargs):
        ctx._gemm_mode = get_gemm_mode()
        return fwd(ctx, *args, **kwargs)

    def backward(ctx, *grads):
        with gemm_mode(ctx._gemm_mode):
            return bwd(ctx, *grads)
    forward.__doc__, backward.__doc__ = fwd.__doc__, bwd.__doc__
    cls.forward, cls.backward = staticmethod(forward), staticmethod(backward)
    return cls


# ---- gradient gate ------------------------------------------------------------------------------------------------
# RepulsiveKernel drops a NaN repulsion term from the loss and from the graph (hyper/diversity/methods.py:342-348, a
# host-side `if rep.isnan()`).  Without a host read-back the same has to happen on the device: the method allocates a
# one-element gate tensor, runs the model kernel inside `grad_gate(gate)`, and writes 0 into it when the term came out
# NaN.  Every autograd node built inside the context remembers the gate and hands it to its dX launch
# (hecka_gram_bwd_*), which then writes exact zeros instead of multiplying NaN Grams by a zero upstream.
_gate_tls = threading.local()


@contextlib.contextmanager
def grad_gate(gate):
    prev = getattr(_gate_tls, 'gate', None)
    _gate_tls.gate = gate
    try:
        yield gate
    finally:
        _gate_tls.gate = prev


def current_gate():
    return getattr(_gate_tls, 'gate', None)


def _gp(gate):
    return gate.data_ptr() if gate is not None else None


def _p(t):
    return t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _features_f32(X):
    """bf16 / fp16 features (BASELINE config 3) enter the fp32 kernels through a differentiable up-cast: every
    16-bit value is exact in fp32, the kernels compute exactly as for fp32 input, and autograd casts dX back to
    the feature dtype.  (Native 16-bit loads in the staging path would save the cast pass: next round.)"""
    if X.is_cuda and X.dtype in (torch.bfloat16, torch.float16):
        return X.float()
    return X


_DTYPE_CODE = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}   # HECKA_DT_* (include/hecka.h)


def native16_eligible(Xs):
    """bf16 / fp16 features that the x16 entry points read as they lie in HBM (no up-cast pass; dX comes back in the
    feature type): GPU tensors of ONE 16-bit type with the same members and samples, B % 128 == 0, every P % 128 == 0,
    split-precision GEMM mode.  HECKA_NATIVE16=0 forces the up-cast path."""
    if not Xs or not knobs.native16:
        return False
    dt, M, B = Xs[0].dtype, Xs[0].shape[0], Xs[0].shape[1] if Xs[0].ndim == 3 else 0
    if dt not in (torch.bfloat16, torch.float16) or B % 128 or B == 0:
        return False
    for X in Xs:
        if not (X.is_cuda and X.dtype == dt and X.ndim == 3 and X.shape[0] == M and X.shape[1] == B):
            return False
        if X.shape[2] % 128 or X.shape[2] == 0 or B * X.shape[2] * 2 >= 2 ** 32:
            return False
    return get_gemm_mode() == 0


def _features(X):
    """Features for the fused linear-Gram path: 16-bit tensors stay as they are when the x16 kernels take them,
    everything else goes through _features_f32."""
    if X.dtype in (torch.bfloat16, torch.float16) and native16_eligible([X]):
        return X
    return _features_f32(X)


def _check_features(X, allow16=False):
    if not X.is_cuda:
        raise RuntimeError('hyper_amd kernels need features on the GPU (got %s); there is no CPU path' % X.device)
    if allow16 and X.dtype in (torch.bfloat16, torch.float16) and X.ndim == 3 and native16_eligible([X]):
        return
    if X.dtype != torch.float32:
        raise RuntimeError('hyper_amd kernels compute in fp32 (got %s)' % X.dtype)
    if X.ndim != 3:
        raise RuntimeError('expected features of shape [members, samples, features]')


def centered_gram(X, return_xmax=False, out=None):
    """X [M,B,P] -> (C [M,B,B] = K - rowmean(K), cvec [M,B]) with K = X X^T; no autograd.
    `return_xmax` adds xmax [M] = max |X[m]| (a by-product of the column-mean pass) for gram_backward: with it the
    dX GEMM may use the fp16-split arithmetic (include/hecka.h, HECKA_GEMM_AUTO)."""
    L = _lib.lib()
    if X.dtype != torch.float32:      # 16-bit features: the multi-layer entry point with one layer
        if not native16_eligible([X]):
            raise RuntimeError('16-bit features of shape %s need the up-cast path (ops._features_f32)' % (tuple(X.shape),))
        C, cvec, xmax = centered_gram_multi([X])[0]
        return (C, cvec, xmax) if return_xmax else (C, cvec)
    M, B, P = X.shape
    sbar = torch.empty((M, P), device=X.device, dtype=torch.float32)
    xmax = torch.empty((M,), device=X.device, dtype=torch.float32)
    C = out if out is not None else torch.empty((M, B, B), device=X.device, dtype=torch.float32)
    if tuple(C.shape) != (M, B, B) or C.dtype != torch.float32 or not C.is_contiguous():
        raise RuntimeError('centered_gram: out must be a contiguous fp32 [M,B,B] tensor')
    cvec = torch.empty((M, B), device=X.device, dtype=torch.float32)
    st = _stream()
    _lib.check(L.hecka_colmean_f32(_p(X), M, B, P, _p(sbar), _p(xmax), st), 'colmean')
    _lib.check(L.hecka_gram_center_f32(_p(X), _p(sbar), _p(xmax), M, B, P, _p(C), _p(cvec), get_gemm_mode(), st), 'gram_center')
    return (C, cvec, xmax) if return_xmax else (C, cvec)


def multi_gram_eligible(Xs):
    """All layers can share ONE centred-Gram launch (hecka_gram_center_multi_f32): fp32 GPU features with the same
    members and samples, at most 16 layers, split-precision GEMM mode, and either interior shapes (B % 128 == 0, every
    P % 16 == 0) or - HECKA_GEMM_AUTO only - ragged ones made of whole quads (B % 4 == 0, every P % 4 == 0: the
    reference's batch sizes 300 / 100 / 400); or 16-bit features that native16_eligible accepts."""
    if not 2 <= len(Xs) <= 16:
        return False
    if Xs[0].dtype in (torch.bfloat16, torch.float16):
        return native16_eligible(Xs)
    M, B = Xs[0].shape[0], Xs[0].shape[1]
    for X in Xs:
        if not (X.is_cuda and X.dtype == torch.float32 and X.ndim == 3 and X.shape[0] == M and X.shape[1] == B):
            return False
        if X.shape[2] % 4 or X.shape[2] == 0 or B * X.shape[2] * 4 >= 2 ** 30:
            return False
    interior = B % 128 == 0 and all(X.shape[2] % 16 == 0 for X in Xs)
    if interior:
        return get_gemm_mode() != 1
    # ragged: the masked split-precision tile.  (A single ragged layer takes it only when its grid fills the GPU - the
    # library's per-layer rule; two or more layers in ONE launch pay even on small grids: MNIST 5 x 300, +4 % steps/s.)
    return B % 4 == 0 and get_gemm_mode() == 0 and knobs.ragged_x3 != '0'


def centered_gram_multi(Xs, slice_rows=0, one_buffer=False):
    """[X_l [M,B,P_l]] -> [(C_l, cvec_l, xmax_l)]: the column-mean passes, then every layer's centred Gram in one
    launch, deepest layer first (no autograd).  slice_rows (a power of two dividing B; fp32 features): C_l comes as
    [B / slice_rows, M, slice_rows, B] - the send order of the member-sharded exchange - instead of [M, B, B].
    one_buffer (with slice_rows): all layers are written into ONE buffer [B / slice_rows, L, M, slice_rows, B] - every
    layer's rows for one destination rank contiguous, the send order of a single all_to_all_single for all layers;
    returns (list, buffer), C_l being the view buffer[:, l]."""
    import ctypes
    L = _lib.lib()
    M, B = Xs[0].shape[0], Xs[0].shape[1]
    st = _stream()
    dev = Xs[0].device
    order = sorted(range(len(Xs)), key=lambda i: -Xs[i].shape[2])
    Xs = [X.contiguous() for X in Xs]
    dt = _DTYPE_CODE[Xs[0].dtype]
    if any(X.dtype != Xs[0].dtype for X in Xs):
        raise RuntimeError('centered_gram_multi: all layers need one feature dtype')
    out = []
    cvec_all = torch.empty((len(Xs), M, B), device=dev, dtype=torch.float32)   # one tensor: the batched gamma stage reads it whole
    if one_buffer:
        if not slice_rows:
            raise RuntimeError('centered_gram_multi: one_buffer needs slice_rows')
        Cbuf = torch.empty((B // slice_rows, len(Xs), M, slice_rows, B), device=dev, dtype=torch.float32)
        C_all = [Cbuf[0, li] for li in range(len(Xs))]      # layer l's first slice block: base + l * M * slice_rows * B
    else:
        C_all = torch.empty((len(Xs), M, B, B), device=dev, dtype=torch.float32)   # likewise for the batched pairwise stage
    n = len(Xs)
    ptrs = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])   # noqa: E731
    xmax_all = torch.empty((n, M), device=dev, dtype=torch.float32)
    sbars = [torch.empty((M, X.shape[2]), device=dev, dtype=torch.float32) for X in Xs]
    one_colmean = dt == 0 and all(X.shape[2] % 4 == 0 for X in Xs) and knobs.colmean_multi
    Ps = (ctypes.c_int * n)(*[Xs[i].shape[2] for i in order])
    if (one_colmean and not slice_rows and get_gemm_mode() == 0 and M <= 65535
            and L.hecka_gram_center_spec_eligible(n, M, B, Ps)):
        # the speculative Gram (include/hecka.h): no column-mean pass over X; sampled shift, row-centred Grams, and the
        # exact path enqueued behind it on a device flag
        nbytes = L.hecka_gram_center_spec_workspace_bytes(n, M, B, Ps)
        ws = torch.empty((nbytes // 4,), device=dev, dtype=torch.float32)
        _lib.check(L.hecka_gram_center_spec_multi_f32(
            n, ptrs([Xs[i] for i in order]), ptrs([sbars[i] for i in order]), _p(xmax_all), M, B, Ps,
            ptrs([C_all[i] for i in order]), ptrs([cvec_all[i] for i in order]), _p(ws), nbytes, st), 'gram_center_spec')
        global LAST_SPEC_WS
        LAST_SPEC_WS = (ws, n * 3 * M)      # (tests read the fallback flag: ws.view(int32)[n * 3 * M])
        return [(C_all[li], cvec_all[li], xmax_all[order.index(li)]) for li in range(n)]
    if one_colmean:      # every layer's column means (and maxima) in one launch, deepest layer first
        _lib.check(L.hecka_colmean_multi_f32(n, ptrs([Xs[i] for i in order]), M, B, (ctypes.c_int * n)(*[Xs[i].shape[2] for i in order]),
                                             ptrs([sbars[i] for i in order]), _p(xmax_all), st), 'colmean_multi')
    for li, X in enumerate(Xs):
        P = X.shape[2]
        sbar = sbars[li]
        xmax = xmax_all[order.index(li)] if one_colmean else xmax_all[li]
        C = C_all[li]
        cvec = cvec_all[li]
        if not one_colmean:
            _lib.check(L.hecka_colmean_x16(_p(X), dt, M, B, P, _p(sbar), _p(xmax), st), 'colmean')
        out.append((C, cvec, xmax, sbar))
    nbytes = L.hecka_gram_center_multi_workspace_bytes(n, M, B, Ps)     # split-K slabs when the grid would be small
    ws = torch.empty((nbytes // 4,), device=dev, dtype=torch.float32) if nbytes else None
    _lib.check(L.hecka_gram_center_multi_x16(
        n, ptrs([Xs[i] for i in order]), dt, ptrs([out[i][3] for i in order]), ptrs([out[i][2] for i in order]), M, B,
        Ps, ptrs([out[i][0] for i in order]), ptrs([out[i][1] for i in order]),
        _p(ws) if ws is not None else None, nbytes, int(slice_rows), len(Xs) * M if one_buffer else 0, get_gemm_mode(), st),
        'gram_center_multi')
    if one_buffer:
        return [(Cbuf[:, li], cvec, xmax) for li, (C, cvec, xmax, _) in enumerate(out)], Cbuf
    if slice_rows:
        return [(C.view(B // slice_rows, M, slice_rows, B), cvec, xmax) for C, cvec, xmax, _ in out]
    return [(C, cvec, xmax) for C, cvec, xmax, _ in out]


def pairwise_inner(Ca, Cb=None, out=None):
    """S[i,j] = <Ca[i], Cb[j]> over all trailing dims; no autograd.  out: a contiguous [Ma,Mb] fp32 tensor to fill."""
    L = _lib.lib()
    Cb = Ca if Cb is None else Cb
    Ma, Mb = Ca.shape[0], Cb.shape[0]
    K = Ca[0].numel()
    S = out if out is not None else torch.empty((Ma, Mb), device=Ca.device, dtype=torch.float32)
    if out is not None and (tuple(out.shape) != (Ma, Mb) or not out.is_contiguous() or out.dtype != torch.float32):
        raise RuntimeError('pairwise_inner: out must be a contiguous fp32 [%d,%d] tensor' % (Ma, Mb))
    nbytes = L.hecka_pairwise_workspace_bytes(Ma, Mb, K)
    ws = torch.empty((max(nbytes, 4) // 4,), device=Ca.device, dtype=torch.float32)
    _lib.check(L.hecka_pairwise_f32(_p(Ca), Ma, _p(Cb), Mb, K, _p(S), _p(ws), nbytes, _stream()), 'pairwise')
    return S


def gram_backward(A, C, cvec, X, detach_diag, xmax=None, nsq=None, gate=None):
    """dX_i = (dC_i + dC_i^T, diag masked) X_i with dC_i = sum_j A[i,j] C[j].
    xmax: [Ma] max |X_i| from centered_gram(return_xmax=True), or None.
    nsq: [Mb] squared Frobenius norms of C (diag of the pairwise matrix), or None: with it the bound on |W| for the
    fp16-split dX GEMM comes from hecka_wbound_f32 instead of a pass over W."""
    L = _lib.lib()
    Ma, Mb = A.shape
    B, P = X.shape[1], X.shape[2]
    st = _stream()
    gamma = torch.empty((Ma, B), device=X.device, dtype=torch.float32)
    W = torch.empty((Ma, B, B), device=X.device, dtype=torch.float32)
    _lib.check(L.hecka_coef_gamma_f32(_p(A), _p(cvec), Ma, Mb, B, _p(gamma), st), 'coef_gamma')
    wmax = torch.empty((Ma,), device=X.device, dtype=torch.float32) if xmax is not None else None
    bound = wmax is not None and nsq is not None
    if bound:
        _lib.check(L.hecka_wbound_f32(_p(A), _p(nsq), _p(gamma), Ma, Mb, B, _p(wmax), st), 'wbound')
    _lib.check(L.hecka_combine_f32(_p(A), _p(C), _p(gamma), Ma, Mb, B, int(bool(detach_diag)), _p(W),
                                   _p(wmax) if wmax is not None and not bound else None, st), 'combine')
    return gram_bwd_w(W, X, wmax, xmax, gate)


# --- the stages of gram_backward one by one, for the member-sharded path (hyper_amd/parallel.py), which runs the
# combine on a column slice of every member's Gram and the dX GEMM on the local members' rows of W ---
SLICE_GRANULE = 64   # column slices in multiples of the streaming combine kernel's 64-column chunks


def coef_gamma(A, cvec):
    """gamma [Ma,B] = A [Ma,Mb] @ cvec [Mb,B]."""
    Ma, Mb = A.shape
    B = cvec.shape[1]
    gamma = torch.empty((Ma, B), device=A.device, dtype=torch.float32)
    _lib.check(_lib.lib().hecka_coef_gamma_f32(_p(A), _p(cvec), Ma, Mb, B, _p(gamma), _stream()), 'coef_gamma')
    return gamma


def combine_slice(A, Cs, gamma, B, n_start, detach_diag):
    """Ws [Ma,n] = elements [n_start, n_start + n) of every W_i = 2 sum_j A_ij C_j + gamma_i 1^T - 1 gamma_i^T
    (diag masked) from the same elements Cs [Mb,n] of every C_j."""
    Ma, Mb = A.shape
    n = Cs.shape[1]
    Ws = torch.empty((Ma, n), device=Cs.device, dtype=torch.float32)
    if n:
        _lib.check(_lib.lib().hecka_combine_slice_f32(_p(A), _p(Cs), _p(gamma), Ma, Mb, B, n_start, n,
                                                      int(bool(detach_diag)), _p(Ws), _stream()), 'combine_slice')
    return Ws


def _nsq_arg(nsq_all, Ln, Mb):
    """nsq [L, Mb] for the fp16-split combine (include/hecka.h) - only in the f16x3 GEMM mode; None otherwise."""
    if nsq_all is None or get_gemm_mode() != 0:
        return None
    assert nsq_all.shape == (Ln, Mb) and nsq_all.is_contiguous() and nsq_all.dtype == torch.float32
    return _p(nsq_all)


def combine_slice_batched(A_all, Cs_all, gamma_all, B, n_start, detach_diag, nsq_all=None):
    """combine_slice for all layers in one launch: A_all [L,Ma,Mb], Cs_all [L,Mb,n], gamma_all [L,Ma,B] -> Ws_all [L,Ma,n].
    nsq_all [L,Mb] = ||C_j||_F^2 of the members' WHOLE Grams (the pairwise diagonal): products on the fp16 matrix cores."""
    Ln, Ma, Mb = A_all.shape
    n = Cs_all.shape[2]
    Ws = torch.empty((Ln, Ma, n), device=Cs_all.device, dtype=torch.float32)
    if n:
        _lib.check(_lib.lib().hecka_combine_slice_batched_f32(_p(A_all), _p(Cs_all), _p(gamma_all), _nsq_arg(nsq_all, Ln, Mb), Ln,
                                                              Ma, Mb, B, n_start, n, int(bool(detach_diag)), _p(Ws), _stream()),
                   'combine_slice_batched')
    return Ws


def wbound(A, nsq, gamma):
    """[Ma] upper bounds of max |W_i| (hecka_wbound_f32)."""
    Ma, Mb = A.shape
    out = torch.empty((Ma,), device=A.device, dtype=torch.float32)
    _lib.check(_lib.lib().hecka_wbound_f32(_p(A), _p(nsq), _p(gamma), Ma, Mb, gamma.shape[1], _p(out), _stream()),
               'wbound')
    return out


def gram_bwd_w(W, X, wmax=None, xmax=None, gate=None):
    """dX [Ma,B,P] = W [Ma,B,B] X   (gate: one-element fp32 device tensor, 0 = write zeros; see grad_gate)."""
    Ma, B, P = X.shape
    if (X.dtype == torch.float32 and B % 256 == 0 and P % 256 == 0 and wmax is not None and xmax is not None
            and gram_bwd_multi_eligible([W], [X], [wmax], [xmax])):
        return gram_bwd_multi([W], [X], [wmax], [xmax], gate)[0]      # one layer on the 256 x 256 eight-wave tile
    dX = torch.empty_like(X)
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.check(_lib.lib().hecka_gram_bwd_x16(_p(W), _p(X), _DTYPE_CODE[X.dtype], _p(wmax) if wmax is not None else None,
                                             _p(xmax) if xmax is not None else None, _gp(gate), Ma, B, P, _p(dX),
                                             get_gemm_mode(), _stream()), 'gram_bwd')
    if PROFILE is not None:
        e1.record()
        PROFILE['gram_bwd'].append((e0, e1, 2.0 * Ma * B * B * P))
        global LAST_BWD_KERNEL
        tile = 256 if (B % 256 == 0 and B <= 1024) else 128
        LAST_BWD_KERNEL = 'gram_bwd_x6_kernel<1, false, 0, %d>' % tile
    return dX


def gram_bwd_multi_eligible(Ws, Xs, wmaxs, xmaxs, slice_rows=0):
    """Every layer's dX = W X in ONE launch (hecka_gram_bwd_multi_f32): fp32 features, f16x3 mode with all maxima."""
    import ctypes
    if not 1 <= len(Xs) <= 16 or get_gemm_mode() != 0 or not knobs.bwd_multi:
        return False
    if any(w is None for w in wmaxs) or any(x is None for x in xmaxs):
        return False
    M, B = Xs[0].shape[0], Xs[0].shape[1]
    if any(X.dtype != torch.float32 or X.shape[0] != M or X.shape[1] != B for X in Xs):
        return False
    Ps = (ctypes.c_int * len(Xs))(*[X.shape[2] for X in Xs])
    return bool(_lib.lib().hecka_gram_bwd_multi_eligible(len(Xs), M, B, Ps, int(slice_rows)))


def gram_bwd_multi(Ws, Xs, wmaxs, xmaxs, gate=None, slice_rows=0, slice_members=0):
    """[dX_l] = [W_l X_l] for all layers in one launch, deepest layer first.  slice_rows: W_l lies as
    [B / slice_rows, M, slice_rows, B] (see centered_gram_multi); slice_members = L * M when the W_l are the views
    buffer[:, l] of ONE buffer [B / slice_rows, L, M, slice_rows, B] (only their first elements' addresses are used)."""
    import ctypes
    n = len(Xs)
    M, B = Xs[0].shape[0], Xs[0].shape[1]
    order = sorted(range(n), key=lambda i: -Xs[i].shape[2])
    dXs = [torch.empty_like(X) for X in Xs]
    ptrs = lambda ts: (ctypes.c_void_p * n)(*[ts[i].data_ptr() for i in order])   # noqa: E731
    Ps = (ctypes.c_int * n)(*[Xs[i].shape[2] for i in order])
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.check(_lib.lib().hecka_gram_bwd_multi_f32(n, ptrs(Ws), ptrs(Xs), ptrs(wmaxs), ptrs(xmaxs), _gp(gate), M, B, Ps,
                                                  ptrs(dXs), int(slice_rows), int(slice_members), get_gemm_mode(), _stream()),
               'gram_bwd_multi')
    if PROFILE is not None:
        e1.record()
        PROFILE['gram_bwd'].append((e0, e1, sum(2.0 * M * B * B * X.shape[2] for X in Xs)))
        global LAST_BWD_KERNEL
        interior = B % 128 == 0 and all(X.shape[2] % 128 == 0 for X in Xs)
        tm = 256 if interior and B % 256 == 0 else 128
        wide = tm == 256 and all(X.shape[2] % 256 == 0 for X in Xs) and knobs.bwd_tn == '256'
        k32 = wide and knobs.k32      # the 16x16x32 tile (gemm_f16x3_k32.hpp)
        LAST_BWD_KERNEL = 'gram_bwd_x6_multi_kernel<%d, %s, %s, %d, %s>' % (
            tm, 'false' if interior else 'true', 'true' if slice_rows else 'false', 256 if wide else 128,
            'true' if k32 else 'false')
        if k32 and not slice_rows and knobs.bwd_stream:
            LAST_BWD_KERNEL = 'gram_bwd_stream_kernel'      # one workgroup per CU walks the 256 x 256 tiles
    return dXs


def _stacked(ts):
    """[L, ...] tensor of the given L tensors: their common base when they are its consecutive slices, else a copy."""
    base = ts[0]._base
    if (base is not None and base.is_contiguous() and base.shape[0] == len(ts) and
            all(t._base is base and t.data_ptr() == base[i].data_ptr() and t.shape == base.shape[1:] for i, t in enumerate(ts))):
        return base
    return torch.stack(ts)


def _spec_arrays(specs, scales):
    import ctypes
    n = len(specs)
    return ((ctypes.c_int * n)(*[s.kind for s in specs]), (ctypes.c_double * n)(*[s.s for s in specs]),
            (ctypes.c_double * n)(*[s.eps for s in specs]), (ctypes.c_double * n)(*[s.arc_eps for s in specs]),
            (ctypes.c_int * n)(*[s.flags for s in specs]), (ctypes.c_double * n)(*[float(x) for x in scales]))


def head_forward_batched(S_all, nsq_all, row_offset, specs, scales):
    """S_all [L,Ma,Mb], nsq_all [L,Mb] -> totals [L] = scale_l * sum of layer l's selected values (one launch)."""
    Ln, Ma, Mb = S_all.shape
    totals = torch.empty((Ln,), device=S_all.device, dtype=torch.float32)
    _lib.check(_lib.lib().hecka_head_fwd_batched_f32(_p(S_all), _p(nsq_all), Ln, Ma, Mb, int(row_offset),
                                                    *_spec_arrays(specs, scales), _p(totals), _stream()), 'head_fwd_batched')
    return totals


def head_backward_batched(S_all, nsq_all, row_offset, specs, up_scalar, scales):
    """Coefficient matrices A_all [L,Ma,Mb] for the upstream *up_scalar * scale_l (one launch)."""
    Ln, Ma, Mb = S_all.shape
    A = torch.empty((Ln, Ma, Mb), device=S_all.device, dtype=torch.float32)
    _lib.check(_lib.lib().hecka_head_bwd_batched_f32(_p(S_all), _p(nsq_all), Ln, Ma, Mb, int(row_offset),
                                                    *_spec_arrays(specs, scales),
                                                    _p(up_scalar) if up_scalar is not None else None, _p(A), _stream()),
               'head_bwd_batched')
    return A


def coef_gamma_batched(A_all, cvec_all):
    Ln, Ma, Mb = A_all.shape
    B = cvec_all.shape[2]
    gamma = torch.empty((Ln, Ma, B), device=A_all.device, dtype=torch.float32)
    _lib.check(_lib.lib().hecka_coef_gamma_batched_f32(_p(A_all), _p(cvec_all), Ln, Ma, Mb, B, _p(gamma), _stream()),
               'coef_gamma_batched')
    return gamma


def wbound_batched(A_all, nsq_all, gamma_all):
    Ln, Ma, Mb = A_all.shape
    out = torch.empty((Ln, Ma), device=A_all.device, dtype=torch.float32)
    _lib.check(_lib.lib().hecka_wbound_batched_f32(_p(A_all), _p(nsq_all), _p(gamma_all), Ln, Ma, Mb, gamma_all.shape[2],
                                                  _p(out), _stream()), 'wbound_batched')
    return out


def pairwise_self_batched_eligible(M, K):
    """One launch for all layers: the streaming kernel at M == 64, the LDS-tiled one otherwise."""
    return M > 0 and 0 < K < 2 ** 31 - 64 and M * K * 4 < 2 ** 40


def pairwise_self_batched(C_all):
    """C_all [L,64,...] contiguous -> S_all [L,64,64] = C_l C_l^T, one launch for all layers."""
    Ln, M = C_all.shape[0], C_all.shape[1]
    K = C_all[0, 0].numel()
    lib = _lib.lib()
    S_all = torch.empty((Ln, M, M), device=C_all.device, dtype=torch.float32)
    nbytes = Ln * lib.hecka_pairwise_workspace_bytes(M, M, K)
    ws = torch.empty((max(nbytes, 4) // 4,), device=C_all.device, dtype=torch.float32)
    _lib.check(lib.hecka_pairwise_self_batched_f32(_p(C_all), Ln, M, K, _p(S_all), _p(ws), nbytes, _stream()),
               'pairwise_self_batched')
    return S_all


# --- the small stages on the layouts of the member-sharded exchange (one all_to_all_single for all layers delivers a
# rank's column slices as [source rank][layer][members of that rank][n]; include/hecka.h, hecka_*_blocked_f32) ---
def blocked_eligible(M, n, blk_rows):
    return bool(_lib.lib().hecka_blocked_rows_eligible(int(M), int(n), int(blk_rows))) and get_gemm_mode() == 0


def pairwise_self_blocked(Cbuf):
    """Cbuf [G, L, Ml, n] (G * Ml = 64 members) -> S_all [L, 64, 64]; bit-equal to pairwise_self_batched on [L, 64, n]."""
    G, Ln, Ml, n = Cbuf.shape
    M = G * Ml
    lib = _lib.lib()
    S_all = torch.empty((Ln, M, M), device=Cbuf.device, dtype=torch.float32)
    nbytes = Ln * lib.hecka_pairwise_workspace_bytes(M, M, n)
    ws = torch.empty((max(nbytes, 4) // 4,), device=Cbuf.device, dtype=torch.float32)
    _lib.check(lib.hecka_pairwise_self_blocked_f32(_p(Cbuf), Ln, M, n, Ml, _p(S_all), _p(ws), nbytes, _stream()),
               'pairwise_self_blocked')
    return S_all


def combine_slice_blocked(A_part, Cbuf, gamma_part, l0, B, n_start, detach_diag, nsq_part=None):
    """Layers l0 .. l0 + Lc - 1 of Cbuf [G, Lin, Ml, n] combined with A_part [Lc, 64, 64] / gamma_part [Lc, 64, B] into
    Ws [G, Lc, Ml, n] - the send order of the returning exchange for that group of layers."""
    G, Lin, Ml, n = Cbuf.shape
    Lc, Ma, Mb = A_part.shape
    Ws = torch.empty((G, Lc, Ml, n), device=Cbuf.device, dtype=torch.float32)
    _lib.check(_lib.lib().hecka_combine_slice_blocked_f32(_p(A_part), _p(Cbuf), _p(gamma_part), _nsq_arg(nsq_part, Lc, Mb), Lc, Lin,
                                                          int(l0), Ma, Mb, B, n_start, n, Ml, int(bool(detach_diag)), _p(Ws),
                                                          _stream()),
               'combine_slice_blocked')
    return Ws


def combine_full(A, C, gamma, detach_diag):
    """W [Ma,B,B] = 2 sum_j A_ij C_j + gamma_i 1^T - 1 gamma_i^T (diag masked); no maximum pass."""
    Ma, Mb = A.shape
    B = C.shape[1]
    W = torch.empty((Ma, B, B), device=C.device, dtype=torch.float32)
    _lib.check(_lib.lib().hecka_combine_f32(_p(A), _p(C), _p(gamma), Ma, Mb, B, int(bool(detach_diag)), _p(W), None,
                                           _stream()), 'combine')
    return W


@mode_aware
class DiversityLayers(torch.autograd.Function):
    """ALL tracked layers of a `model_layers` loss in one autograd node (linear centred Grams, he / cka heads, mean / sum
    reduction, host-float layer weights):
        forward : column means, ONE Gram launch, pairwise per layer, ONE head launch -> sum_l w_l reduce(values_l)
        backward: ONE head_bwd launch, ONE gamma launch, ONE bound launch, then combine + dX GEMM per layer.
    Returns (loss, S_all [L,M,M], nsq_all [L,M], C_0 .. C_{L-1}); everything but the loss is a non-differentiable
    by-product for `track` (the `ckas` metric reads S / nsq, hyper/experiments/metrics.py:75-91 reads the Grams)."""

    @staticmethod
    def forward(ctx, cfg, *Xs):
        specs, scales, detach_diag, pre, gspecs = cfg
        Ln, M = len(Xs), Xs[0].shape[0]
        Xs = [X.contiguous() for X in Xs]
        if gspecs is not None:
            # cossim / rbf Grams: one raw Gram + transform per layer into a shared [L,M,B,B] buffer; pairwise and heads batched
            B = Xs[0].shape[1]
            C_all = torch.empty((Ln, M, B, B), device=Xs[0].device, dtype=torch.float32)
            aux = [transformed_gram(X, g, True, out=C_all[k])[1:] for k, (X, g) in enumerate(zip(Xs, gspecs))]
            Cs = [C_all[k] for k in range(Ln)]
            S_all = pairwise_self_batched(C_all)
            nsq_all = torch.diagonal(S_all, dim1=1, dim2=2).contiguous()
            totals = head_forward_batched(S_all, nsq_all, 0, specs, scales)
            ctx.save_for_backward(*Xs, *Cs, *[a[0] for a in aux], *[a[1] for a in aux], *[a[2] for a in aux], S_all, nsq_all)
            ctx.cfg = (specs, scales, bool(detach_diag), Ln)
            ctx.gspecs = list(gspecs)
            ctx.gate = current_gate()
            ctx.mark_non_differentiable(S_all, nsq_all, *Cs)
            ctx.set_materialize_grads(False)
            return (totals.sum(), S_all, nsq_all) + tuple(Cs)
        ctx.gspecs = None
        ctx.gate = current_gate()
        if pre is None:
            pre = centered_gram_multi(Xs)
        Cs = [p[0] for p in pre]
        cv_all = _stacked([p[1] for p in pre])
        S_all = pairwise_self_batched(_stacked(Cs)) if pairwise_self_batched_eligible(M, Cs[0][0].numel()) else None
        if S_all is None:
            S_all = torch.empty((Ln, M, M), device=Xs[0].device, dtype=torch.float32)
            for li, C in enumerate(Cs):
                pairwise_inner(C, out=S_all[li])
        nsq_all = torch.diagonal(S_all, dim1=1, dim2=2).contiguous()
        totals = head_forward_batched(S_all, nsq_all, 0, specs, scales)
        ctx.save_for_backward(*Xs, *Cs, *[p[2] for p in pre], cv_all, S_all, nsq_all)
        ctx.cfg = (specs, scales, bool(detach_diag), Ln)
        ctx.mark_non_differentiable(S_all, nsq_all, *Cs)
        ctx.set_materialize_grads(False)
        return (totals.sum(), S_all, nsq_all) + tuple(Cs)

    @staticmethod
    def backward(ctx, dout, *_unused):
        specs, scales, detach_diag, Ln = ctx.cfg
        t = ctx.saved_tensors
        if ctx.gspecs is not None:
            Xs, Cs, rmeans, gdiags, hbands = (t[k * Ln:(k + 1) * Ln] for k in range(5))
            S_all, nsq_all = t[5 * Ln:]
            if dout is None:
                return (None,) + tuple(torch.zeros_like(X) for X in Xs)
            A_all = head_backward_batched(S_all, nsq_all, 0, specs, dout.contiguous().float(), scales)
            return (None,) + tuple(transformed_gram_backward(A_all[li], Cs[li], rmeans[li], gdiags[li], hbands[li], Xs[li],
                                                             ctx.gspecs[li], ctx.gate) for li in range(Ln))
        Xs, Cs, xmaxs = t[:Ln], t[Ln:2 * Ln], t[2 * Ln:3 * Ln]
        cv_all, S_all, nsq_all = t[3 * Ln:]
        if dout is None:
            return (None,) + tuple(torch.zeros_like(X) for X in Xs)
        dout = dout.contiguous().float()
        A_all = head_backward_batched(S_all, nsq_all, 0, specs, dout, scales)
        gamma_all = coef_gamma_batched(A_all, cv_all)
        wmax_all = wbound_batched(A_all, nsq_all, gamma_all)
        wmaxs = [wmax_all[li] for li in range(Ln)]
        if gram_bwd_multi_eligible(Cs, Xs, wmaxs, xmaxs):      # every W in ONE combine launch, then ONE dX launch
            M, B = Xs[0].shape[0], Xs[0].shape[1]
            W_all = combine_slice_batched(A_all, _stacked(Cs).view(Ln, M, B * B), gamma_all, B, 0, detach_diag, nsq_all)
            return (None,) + tuple(gram_bwd_multi([W_all[li] for li in range(Ln)], Xs, wmaxs, xmaxs, ctx.gate))
        grads = []
        for li in range(Ln):
            W = combine_full(A_all[li], Cs[li], gamma_all[li], detach_diag)
            grads.append(gram_bwd_w(W, Xs[li], wmaxs[li], xmaxs[li], ctx.gate))
        return (None,) + tuple(grads)


def diversity_layers(Xs, specs, scales, detach_diag=True, pre=None, gspecs=None):
    """gspecs: one GramSpec per layer for cossim / rbf Grams (fp32 features), None for the linear path."""
    if gspecs is not None:
        Xs = [_features_f32(X) for X in Xs]
    return DiversityLayers.apply((list(specs), [float(x) for x in scales], bool(detach_diag), pre,
                                  list(gspecs) if gspecs is not None else None), *Xs)


@mode_aware
class CenteredGramPairwise(torch.autograd.Function):
    """X [M,B,P] -> (S [M,M], nsq [M], C [M,B,B]).

    C = centred linear Gram (the reference's track['gram']), S[i,j] = <C_i, C_j>, nsq = diag(S).
    Gradients: dC_m = sum_j A_mj C_j with A = dS (+ dS^T unless `left_only`) + 2 diag(dnsq);
    `left_only` gives S the gradient of C @ C.detach()^T (the reference's `detach_right`).
    `detach_diag` zeroes the Gram diagonal's gradient (kernels.py:76-104).  C itself is returned
    non-differentiable (it is a by-product for metrics)."""

    @staticmethod
    def forward(ctx, X, detach_diag, left_only):
        _check_features(X)
        X = X.contiguous()
        C, cvec, xmax = centered_gram(X, True)
        S = pairwise_inner(C)
        nsq = torch.diagonal(S).clone()
        ctx.save_for_backward(X, C, cvec, xmax)
        ctx.detach_diag, ctx.left_only = bool(detach_diag), bool(left_only)
        ctx.gate = current_gate()
        ctx.mark_non_differentiable(C)
        ctx.set_materialize_grads(False)
        return S, nsq, C

    @staticmethod
    def backward(ctx, dS, dnsq, _dC):
        X, C, cvec, xmax = ctx.saved_tensors
        M = X.shape[0]
        A = torch.zeros((M, M), device=X.device, dtype=torch.float32)
        if dS is not None:
            A = A + (dS if ctx.left_only else dS + dS.t())
        if dnsq is not None:
            A = A + torch.diag_embed(2.0 * dnsq)
        dX = gram_backward(A.contiguous(), C, cvec, X, ctx.detach_diag, xmax, gate=ctx.gate)
        return dX, None, None


def centered_gram_pairwise(X, detach_diag=True, left_only=False):
    return CenteredGramPairwise.apply(_features_f32(X), detach_diag, left_only)


class PairwiseInner(torch.autograd.Function):
    """V [M,K] -> (S [M,M] = V V^T, nsq [M] = diag S) for arbitrary vectorised Grams.
    backward: dV = (dS (+ dS^T unless left_only) + 2 diag(dnsq)) V   (one row-mix GEMM)."""

    @staticmethod
    def forward(ctx, V, left_only):
        if not V.is_cuda or V.dtype != torch.float32:
            raise RuntimeError('hyper_amd kernels need fp32 GPU tensors (got %s on %s)' % (V.dtype, V.device))
        V = V.contiguous()
        S = pairwise_inner(V)
        ctx.save_for_backward(V)
        ctx.left_only = bool(left_only)
        return S, torch.diagonal(S).clone()

    @staticmethod
    def backward(ctx, dS, dnsq):
        (V,) = ctx.saved_tensors
        M, K = V.shape
        A = torch.zeros((M, M), device=V.device, dtype=torch.float32)
        if dS is not None:
            A = A + (dS if ctx.left_only else dS + dS.t())
        if dnsq is not None:
            A = A + torch.diag_embed(2.0 * dnsq)
        A = A.contiguous()
        dV = torch.empty_like(V)
        _lib.check(_lib.lib().hecka_rowmix_f32(_p(A), _p(V), M, M, K, _p(dV), _stream()), 'rowmix')
        return dV, None


def pairwise_inner_ad(V, left_only=False):
    return PairwiseInner.apply(V, left_only)


@mode_aware
class LinearGram(torch.autograd.Function):
    """X [M,B,P] -> K = X X^T (value of the diagonal kept; its gradient dropped when detach_diag).
    backward: dX = (dK + dK^T, diag masked) X."""

    @staticmethod
    def forward(ctx, X, detach_diag):
        _check_features(X)
        X = X.contiguous()
        L = _lib.lib()
        M, B, P = X.shape
        zero = torch.zeros((M, P), device=X.device, dtype=torch.float32)
        K = torch.empty((M, B, B), device=X.device, dtype=torch.float32)
        cvec = torch.empty((M, B), device=X.device, dtype=torch.float32)
        _lib.check(L.hecka_gram_center_f32(_p(X), _p(zero), None, M, B, P, _p(K), _p(cvec), get_gemm_mode(), _stream()), 'gram')
        ctx.save_for_backward(X)
        ctx.detach_diag = bool(detach_diag)
        ctx.gate = current_gate()
        return K

    @staticmethod
    def backward(ctx, dK):
        (X,) = ctx.saved_tensors
        L = _lib.lib()
        M, B, P = X.shape
        dK = dK.contiguous()
        W = torch.empty_like(dK)
        dX = torch.empty_like(X)
        st = _stream()
        _lib.check(L.hecka_symmetrize_f32(_p(dK), M, B, int(ctx.detach_diag), _p(W), st), 'symmetrize')
        _lib.check(L.hecka_gram_bwd_f32(_p(W), _p(X), None, None, _gp(ctx.gate), M, B, P, _p(dX), get_gemm_mode(), st), 'gram_bwd')
        return dX, None


def linear_gram(X, detach_diag=True):
    return LinearGram.apply(_features_f32(X), detach_diag)


def cross_gram(X, Y):
    """K_m = X_m Y_m^T for X [M,N,P], Y [M,L,P]: libhecka's grouped GEMM, differentiable in both operands (the general
    path of diversity/kernels.py: Grams between two sample sets)."""
    from .layers.linear import BatchedLinear
    return BatchedLinear.apply(_features_f32(X).contiguous(), _features_f32(Y).contiguous(), None, 1.0, 1.0)


class CenterRows(torch.autograd.Function):
    """gram - rowmean(gram) over the last dim; self-adjoint, so backward is the same kernel."""

    @staticmethod
    def forward(ctx, G):
        if not G.is_cuda or G.dtype != torch.float32:
            raise RuntimeError('hyper_amd kernels need fp32 GPU tensors (got %s on %s)' % (G.dtype, G.device))
        G = G.contiguous()
        out = torch.empty_like(G)
        cols = G.shape[-1]
        _lib.check(_lib.lib().hecka_center_rows_f32(_p(G), G.numel() // cols, cols, _p(out), None, _stream()),
                   'center_rows')
        return out

    @staticmethod
    def backward(ctx, dC):
        return CenterRows.apply(dC)


def center_rows(G):
    return CenterRows.apply(G)


# ------------------------------------------------------------------------- fused layer op
HEAD_CKA, HEAD_HE = 0, 1
F_USE_EXP, F_ABS_VALS, F_REMOVE_DIAG, F_DETACH_RIGHT = 1, 2, 4, 8


class HeadSpec(object):
    """Non-learnable head hyper-parameters, passed by value to the head kernels."""

    def __init__(self, kind, s=2.0, eps=1e-6, arc_eps=1e-3, use_exp=False, abs_vals=False, remove_diag=True,
                 detach_right=False):
        self.kind, self.s, self.eps, self.arc_eps = int(kind), float(s), abs(float(eps)), abs(float(arc_eps))
        self.flags = ((F_USE_EXP if use_exp else 0) | (F_ABS_VALS if abs_vals else 0) |
                      (F_REMOVE_DIAG if remove_diag else 0) | (F_DETACH_RIGHT if detach_right else 0))
        self.remove_diag = bool(remove_diag)

    def args(self):
        return (self.kind, self.s, self.eps, self.arc_eps, self.flags)


GRAM_LINEAR, GRAM_COSSIM, GRAM_RBF = 0, 1, 2


class GramSpec(object):
    """Which feature Gram a layer uses: linear (fused centred MFMA kernel) or a row-wise transform of the raw
    Gram (cossim, rbf with the median-heuristic bandwidth `param * median / (2 log(B+1))` or `param^2`)."""

    def __init__(self, kind=GRAM_LINEAR, param=1.0, median=True):
        self.kind, self.param, self.median = int(kind), float(param), bool(median)


def raw_gram(X):
    """G = X X^T [M,B,B] (the centred-Gram kernel with zero column means)."""
    L = _lib.lib()
    M, B, P = X.shape
    zero = torch.zeros((M, P), device=X.device, dtype=torch.float32)
    G = torch.empty((M, B, B), device=X.device, dtype=torch.float32)
    scratch = torch.empty((M, B), device=X.device, dtype=torch.float32)
    # no maxima (xmax) on purpose: the cossim / rbf transforms divide by (or difference) each row's OWN squared norm, and
    # the fp16 split scaled by the member's maximum loses the relative precision of rows far below it - tried: a row at
    # 1e-6 of the maximum came out with a zero norm, NaN cosines after some training steps of the hypernetwork.  bf16x6
    # (interior shapes) and fp32 MFMA (ragged) have no range assumption.  (Maxima passes over W and X for the dX GEMM of
    # these layers were tried too: the two extra HBM passes cost what the faster GEMM saves at the hypernetwork's sizes.)
    _lib.check(L.hecka_gram_center_f32(_p(X), _p(zero), None, M, B, P, _p(G), _p(scratch), get_gemm_mode(), _stream()), 'gram')
    return G


def transformed_gram(X, gspec, center, out=None):
    """cossim / rbf Gram (row-centred when `center`): returns (C, rmean, gdiag, hband).  out: [M,B,B] buffer for C."""
    L = _lib.lib()
    M, B, _ = X.shape
    G = raw_gram(X)
    st = _stream()
    hband = torch.empty((M,), device=X.device, dtype=torch.float32)
    if gspec.kind == GRAM_RBF:
        if gspec.median:
            _lib.check(L.hecka_rbf_bandwidth_f32(_p(G), M, B, gspec.param, _p(hband), st), 'rbf_bandwidth')
        else:
            hband.fill_(gspec.param ** 2)
    C = out if out is not None else torch.empty_like(G)
    if tuple(C.shape) != tuple(G.shape) or C.dtype != torch.float32 or not C.is_contiguous():
        raise RuntimeError('transformed_gram: out must be a contiguous fp32 [M,B,B] tensor')
    rmean = torch.empty((M, B), device=X.device, dtype=torch.float32)
    gdiag = torch.empty((M, B), device=X.device, dtype=torch.float32)
    _lib.check(L.hecka_gram_transform_f32(_p(G), M, B, gspec.kind, _p(hband), int(bool(center)), _p(C), _p(rmean),
                                          _p(gdiag), st), 'gram_transform')
    return C, rmean, gdiag, hband


def transformed_gram_backward(A, C, rmean, gdiag, hband, X, gspec, gate=None):
    """dX for a cossim / rbf layer from the coefficient matrix A (dC_i = sum_j A_ij C_j)."""
    L = _lib.lib()
    M, B, P = X.shape
    st = _stream()
    rho = torch.empty((M, B), device=X.device, dtype=torch.float32)
    W = torch.empty((M, B, B), device=X.device, dtype=torch.float32)
    _lib.check(L.hecka_coef_gamma_f32(_p(A), _p(rmean), M, M, B, _p(rho), st), 'coef_gamma')
    _lib.check(L.hecka_combine_f32(_p(A), _p(C), _p(rho), M, M, B, 1, _p(W), None, st), 'combine')
    return transformed_gram_bwd_w(W, C, rmean, gdiag, hband, X, gspec, gate)


def transformed_gram_bwd_w(W, C, rmean, gdiag, hband, X, gspec, gate=None):
    """The member-local end of transformed_gram_backward: W [M,B,B] = dK' + dK'^T (zero diagonal) of these members'
    cossim / rbf Grams C -> dX.  W is overwritten (the transform's backward runs in place).  The member-sharded path
    (hyper_amd/parallel.py) calls it on the rows of W that came back from the exchange."""
    L = _lib.lib()
    M, B, P = X.shape
    st = _stream()
    if not W.is_contiguous() or tuple(W.shape) != (M, B, B):
        raise RuntimeError('transformed_gram_bwd_w: W must be a contiguous [M,B,B] tensor')
    dX = torch.empty_like(X)
    _lib.check(L.hecka_gram_transform_bwd_f32(_p(W), _p(C), _p(rmean), _p(gdiag), _p(hband), M, B, gspec.kind, st),
               'gram_transform_bwd')
    _lib.check(L.hecka_gram_bwd_f32(_p(W), _p(X), None, None, _gp(gate), M, B, P, _p(dX), get_gemm_mode(), st), 'gram_bwd')
    return dX


@mode_aware
class DiversityLayer(torch.autograd.Function):
    """One tracked layer of the diversity loss, entirely in libhecka:
         X [M,B,P] -> centred Gram C -> S = <C_i,C_j> -> cos -> (energy) -> value matrix / reduced scalar.

    reduce=None  : returns (val [M,M], C)   - val_ij for every ordered pair (diagonal included)
    reduce='mean'|'sum': returns (scalar, C) - over the off-diagonal pairs (remove_diag) or all pairs
    backward (from dval [M,M] or the scalar's gradient): head_bwd -> combine -> dX = W X."""

    @staticmethod
    def forward(ctx, X, spec, detach_diag, reduce, gspec, pre=None):
        _check_features(X, allow16=gspec is None or gspec.kind == GRAM_LINEAR)
        X = X.contiguous()
        L = _lib.lib()
        M = X.shape[0]
        if gspec is None or gspec.kind == GRAM_LINEAR:
            C, cvec, xmax = pre if pre is not None else centered_gram(X, True)   # pre: centered_gram_multi's output
            aux = (cvec, xmax)
        else:
            C, rmean, gdiag, hband = transformed_gram(X, gspec, True)
            aux = (rmean, gdiag, hband)
        S = pairwise_inner(C)
        nsq = torch.diagonal(S).contiguous()
        st = _stream()
        if reduce is None:
            out = torch.empty((M, M), device=X.device, dtype=torch.float32)
            _lib.check(L.hecka_head_fwd_f32(_p(S), _p(nsq), M, M, 0, *spec.args(), _p(out), None, 1.0, st), 'head_fwd')
            ctx.scale = 1.0
        else:
            count = (M * (M - 1) if spec.remove_diag else M * M) if reduce == 'mean' else 1
            ctx.scale = 1.0 / count if count else float('nan')   # a single member has no pairs: mean of nothing = NaN
            out = torch.empty((), device=X.device, dtype=torch.float32)
            _lib.check(L.hecka_head_fwd_f32(_p(S), _p(nsq), M, M, 0, *spec.args(), None, _p(out), ctx.scale, st), 'head_fwd')
        ctx.save_for_backward(X, C, S, nsq, *aux)
        ctx.spec, ctx.detach_diag, ctx.reduced, ctx.gspec = spec, bool(detach_diag), reduce is not None, gspec
        ctx.gate = current_gate()
        ctx.mark_non_differentiable(C, S, nsq)
        ctx.set_materialize_grads(False)   # no zeros_like(C) for the non-differentiable by-products
        return out, C, S, nsq              # S [M,M] / nsq [M]: inner products and squared norms of the Grams (metrics)

    @staticmethod
    def backward(ctx, dout, _dC, _dS=None, _dn=None):
        X, C, S, nsq = ctx.saved_tensors[:4]
        aux = ctx.saved_tensors[4:]
        if dout is None:
            return torch.zeros_like(X), None, None, None, None, None
        L = _lib.lib()
        M = X.shape[0]
        A = torch.empty((M, M), device=X.device, dtype=torch.float32)
        dout = dout.contiguous().float()
        up_mat, up_scalar = (None, _p(dout)) if ctx.reduced else (_p(dout), None)
        _lib.check(L.hecka_head_bwd_f32(_p(S), _p(nsq), M, M, 0, *ctx.spec.args(), up_mat, up_scalar, ctx.scale, _p(A),
                                        _stream()), 'head_bwd')
        if len(aux) == 2:
            dX = gram_backward(A, C, aux[0], X, ctx.detach_diag, aux[1], nsq, gate=ctx.gate)
        else:
            dX = transformed_gram_backward(A, C, aux[0], aux[1], aux[2], X, ctx.gspec, ctx.gate)
        return dX, None, None, None, None, None


def diversity_layer(X, spec, detach_diag=True, reduce=None, gspec=None, pre=None):
    linear = gspec is None or gspec.kind == GRAM_LINEAR
    return DiversityLayer.apply(_features(X) if linear else _features_f32(X), spec, detach_diag, reduce, gspec, pre)


_OFFDIAG_INDEX = {}


def offdiag_index(M, device):
    """Flat indices of the off-diagonal entries of an [M,M] matrix in row-major order (what the
    reference's boolean mask `m[~eye]` selects), cached so that no nonzero()/host sync happens per step."""
    key = (M, str(device))
    if key not in _OFFDIAG_INDEX:
        idx = torch.arange(M * M, device=device).view(M, M)
        _OFFDIAG_INDEX[key] = idx[~torch.eye(M, dtype=torch.bool, device=device)].contiguous()
    return _OFFDIAG_INDEX[key]


# ------------------------------------------------------------------------- head kernels, raw
def head_forward(S, nsq, row_offset, spec, scale, want_values=False):
    """S [Ma,Mb], nsq [Mb] -> (values [Ma,Mb] or None, scale * sum of selected values as a 0-dim tensor)."""
    Ma, Mb = S.shape
    val = torch.empty((Ma, Mb), device=S.device, dtype=torch.float32) if want_values else None
    tot = torch.empty((), device=S.device, dtype=torch.float32)
    _lib.check(_lib.lib().hecka_head_fwd_f32(_p(S), _p(nsq), Ma, Mb, int(row_offset), *spec.args(),
                                            _p(val) if want_values else None, _p(tot), float(scale), _stream()), 'head_fwd')
    return val, tot


def head_backward(S, nsq, row_offset, spec, up_scalar, scale, up_mat=None):
    """Coefficient matrix A [Ma,Mb] of dC_i = sum_j A_ij C_j."""
    Ma, Mb = S.shape
    A = torch.empty((Ma, Mb), device=S.device, dtype=torch.float32)
    _lib.check(_lib.lib().hecka_head_bwd_f32(_p(S), _p(nsq), Ma, Mb, int(row_offset), *spec.args(),
                                            _p(up_mat) if up_mat is not None else None,
                                            _p(up_scalar) if up_scalar is not None else None, float(scale), _p(A),
                                            _stream()), 'head_bwd')
    return A


@mode_aware
class TransformedGram(torch.autograd.Function):
    """Un-fused functional API: X -> K' (cossim / rbf Gram, not centred); backward dK' -> dX."""

    @staticmethod
    def forward(ctx, X, gspec, left_only=False):
        _check_features(X)
        X = X.contiguous()
        K, rmean, gdiag, hband = transformed_gram(X, gspec, False)
        ctx.save_for_backward(X, K, gdiag, hband)
        ctx.gspec, ctx.left_only = gspec, bool(left_only)
        ctx.gate = current_gate()
        return K

    @staticmethod
    def backward(ctx, dK):
        X, K, gdiag, hband = ctx.saved_tensors
        L = _lib.lib()
        M, B, P = X.shape
        st = _stream()
        W = torch.empty_like(K)
        dX = torch.empty_like(X)
        if ctx.left_only:
            # K(X, X.detach()) (the reference's `detach_right`): the transform's backward is row-wise - row a of W holds
            # the coefficients of dX_a - so the un-symmetrised upstream gives exactly the left argument's gradient
            W.copy_(dK)
            W.diagonal(dim1=1, dim2=2).zero_()
        else:
            _lib.check(L.hecka_symmetrize_f32(_p(dK.contiguous()), M, B, 1, _p(W), st), 'symmetrize')
        _lib.check(L.hecka_gram_transform_bwd_f32(_p(W), _p(K), None, _p(gdiag), _p(hband), M, B, ctx.gspec.kind, st),
                   'gram_transform_bwd')
        _lib.check(L.hecka_gram_bwd_f32(_p(W), _p(X), None, None, _gp(ctx.gate), M, B, P, _p(dX), get_gemm_mode(), st), 'gram_bwd')
        return dX, None, None


def transformed_gram_ad(X, gspec, left_only=False):
    return TransformedGram.apply(_features_f32(X), gspec, left_only)
