"""Batched-weight Conv2d (hyper/layers/conv.py:16-208 `_ConvNd`, :215-346 `Conv2d`, :434-502 `_conv_forward`).
The reference loops F.conv2d over the ensemble members and stacks the results; here all members go through ONE
grouped GEMM on fp32 MFMA: patches are gathered once (libhecka hecka_im2col_f32, shared across members when the
input is a plain image batch), forward / wgrad / dgrad are hecka_bgemm_f32 calls with the parameter scaling in the
epilogue and the bias carried as an extra weight column against a row of ones, and hecka_col2im_f32 gathers the
input gradient.  The gathered patches are kept for the weight gradient (HECKA_CONV_KEEP_COLS_GB per layer, default 8)
and re-gathered in backward beyond that."""
import math

import numpy as np
import torch
import torch.nn.functional as F
import torch.nn.init as init
from torch.nn.modules.utils import _pair, _reverse_repeat_tuple

from .. import _lib, ops
from ..config import knobs
from ..net import activation
from .linear import _absmax_rows, _bgemm, _bgemm_x3, _x3_eligible
from .norm import cached_absmax
from .module import ParametersGroup, register_gen_module


def _out_size(n, k, s, p, d):
    return (n + 2 * p - d * (k - 1) - 1) // s + 1


def _im2col(x5, geom, ones_row):
    """x5 [Mx,N,C,H,W] -> cols [Mx,R,Q]"""
    Mx, N, C, H, W = x5.shape
    kh, kw, sh, sw, ph, pw, dh, dw = geom
    Q = N * _out_size(H, kh, sh, ph, dh) * _out_size(W, kw, sw, pw, dw)
    R = C * kh * kw + (1 if ones_row else 0)
    cols = torch.empty((Mx, R, Q), device=x5.device, dtype=torch.float32)
    _lib.check(_lib.lib().hecka_im2col_f32(x5.data_ptr(), Mx, N, C, H, W, kh, kw, sh, sw, ph, pw, dh, dw,
                                           int(ones_row), cols.data_ptr(), torch.cuda.current_stream().cuda_stream),
               'im2col')
    return cols


def _col2im(dcols, M, N, C, H, W, geom):
    kh, kw, sh, sw, ph, pw, dh, dw = geom
    dx = torch.empty((M, N, C, H, W), device=dcols.device, dtype=torch.float32)
    _lib.check(_lib.lib().hecka_col2im_f32(dcols.data_ptr(), M, N, C, H, W, kh, kw, sh, sw, ph, pw, dh, dw,
                                           dcols.shape[1], dx.data_ptr(), torch.cuda.current_stream().cuda_stream),
               'col2im')
    return dx


_TABS = {}


def _conv_tab(Cin, H, W, kh, kw, ph, pw, device, rows=None):
    """Tap table of the implicit-GEMM convolution (hecka_conv_x3_f32): for tap k = (c, i, j) the element offset of
    x[c][h + i - ph][w + j - pw] from x[0][h][w], and (di << 16) | (dj & 0xffff).  `rows` pads the table with taps that
    are never in range (the weight-gradient kernel's patch rows beyond C kh kw)."""
    key = (Cin, H, W, kh, kw, ph, pw, str(device), rows)
    if key not in _TABS:
        k = torch.arange(Cin * kh * kw, dtype=torch.int64)
        c, i, j = k // (kh * kw), (k // kw) % kh, k % kw
        di, dj = i - ph, j - pw
        tab = torch.stack([(c * H + di) * W + dj, (di << 16) | (dj & 0xffff)], dim=1)
        if rows is not None and rows > tab.shape[0]:
            pad = torch.tensor([[0, (-30000 << 16)]], dtype=torch.int64).expand(rows - tab.shape[0], 2)
            tab = torch.cat([tab, pad])
        if len(_TABS) > 256:
            _TABS.clear()
        _TABS[key] = tab.to(torch.int32).contiguous().to(device)
    return _TABS[key]


def _tap_tab(Cin, H, W, taps, device):
    """Table of the TAP-MAJOR implicit GEMM (hecka_conv_x3_tapmajor_f32): K ordered (tap, channel), one entry per k-tile of
    16 channels = (byte offset of the tile's first channel, byte offset of the tap, di, dj), plus one entry of padding (the
    kernel fetches one entry ahead).  taps: [(di, dj)] in the order of the weight matrix's K axis."""
    key = ('tap', Cin, H, W, tuple(taps), str(device))
    if key not in _TABS:
        rows = [(cb * 16 * H * W * 4, (di * W + dj) * 4, di, dj) for di, dj in taps for cb in range(Cin // 16)]
        rows.append((0, 0, -30000, -30000))
        tab = (np.asarray(rows, dtype=np.int64) & 0xFFFFFFFF).astype(np.uint32).view(np.int32)
        if len(_TABS) > 256:
            _TABS.clear()
        _TABS[key] = torch.from_numpy(tab).contiguous().to(device)
    return _TABS[key]


def _tap_weights(W, rot):
    """W [M,O,C,kh,kw] -> [M,O,kh kw C] (forward) or the rotated [M,C,kh kw O] (input gradient): the A operand of the tap-major
    implicit GEMM (hecka_conv_weights_tapmajor_f32)."""
    M, O, C, kh, kw = W.shape
    if C * kh * kw > 16384:
        Wt = W.flip(3, 4).permute(0, 2, 3, 4, 1) if rot else W.permute(0, 1, 3, 4, 2)
        return Wt.reshape(M, C if rot else O, -1).contiguous()
    Wc = W.detach().contiguous()
    out = torch.empty((M, C, kh * kw * O) if rot else (M, O, kh * kw * C), device=W.device, dtype=torch.float32)
    _lib.check(_lib.lib().hecka_conv_weights_tapmajor_f32(Wc.data_ptr(), M, O, C, kh * kw, int(rot), out.data_ptr(),
                                                          torch.cuda.current_stream().cuda_stream), 'conv_weights_tapmajor')
    return out


def _tapmajor(Cin):
    """K ordered (tap, channel) - one range check per k-tile instead of one per gathered element: 16 | Cin."""
    return knobs.conv_tapmajor and Cin % 16 == 0


def _conv_x3_ok(geom, M, rows, K, N, Cin, H, W):
    """The implicit-GEMM convolution with the output as large as the input (unit stride, 'same' padding): the forward of
    most layers and every input gradient that runs as a convolution of dy with the rotated weights."""
    kh, kw, sh, sw, ph, pw, dh, dw = geom
    return (knobs.conv_implicit and knobs.conv_x3
            and sh == 1 and sw == 1 and dh == 1 and dw == 1 and 2 * ph == kh - 1 and 2 * pw == kw - 1
            and ops.get_gemm_mode() == 0 and bool(_lib.lib().hecka_conv_x3_eligible(M, rows, K, N, Cin, H, W)))


def _conv_x3_fwd_ok(geom, M, rows, K, N, Cin, H, W, Ho, Wo):
    """The FORWARD convolution as an implicit GEMM: any stride and zero padding, no dilation (the stage transitions of a
    ResNet - 3 x 3 / stride 2 and the 1 x 1 / stride 2 shortcut - besides the same-size layers)."""
    kh, kw, sh, sw, ph, pw, dh, dw = geom
    return (knobs.conv_implicit and knobs.conv_x3 and dh == 1 and dw == 1 and ops.get_gemm_mode() == 0
            and bool(_lib.lib().hecka_conv_x3_strided_eligible(M, rows, K, N, Cin, H, W, Ho, Wo)))


def _conv_x3(A, X, M, rows, K, sX, N, Cin, H, W, tab, alpha, amaxA, amaxB, amaxB_stride, out_hw=None, stride=(1, 1),
             tapmajor=False):
    """Y [M,N,rows,Ho,Wo] = alpha * implicit-GEMM convolution of X [M|1,N,Cin,H,W] with A [M,rows,K] (see include/hecka.h);
    out_hw / stride default to the same-size unit-stride case.  tapmajor: K ordered (tap, channel), tab from _tap_tab."""
    Ho, Wo = out_hw if out_hw is not None else (H, W)
    Y = torch.empty((M, N, rows, Ho, Wo), device=X.device, dtype=torch.float32)
    L = _lib.lib()
    fn = L.hecka_conv_x3_tapmajor_f32 if tapmajor else L.hecka_conv_x3_placed_f32
    _lib.check(fn(A.data_ptr(), X.data_ptr(), Y.data_ptr(), tab.data_ptr(), M, rows, K, rows * K, sX, N, Cin, H, W, Ho, Wo,
                  int(stride[0]), int(stride[1]), Ho, Wo, 1, 1, 0, 0, float(alpha), amaxA.data_ptr(), amaxB.data_ptr(),
                  int(amaxB_stride), torch.cuda.current_stream().cuda_stream), 'conv_x3')
    return Y


def _dgrad_s2_classes(geom, H, W):
    """Parity classes (a, b) of the input pixels of a stride-2 convolution: the taps that reach a class, their shifts on
    the dy grid, and the class's pixel grid.  dx[2u + a][2v + b] = sum_{i in I_a, j in J_b} W[.., i, j] dy[u + (a + ph - i) / 2][v +
    (b + pw - j) / 2]."""
    kh, kw, sh, sw, ph, pw, dh, dw = geom
    out = []
    for a in (0, 1):
        I = [i for i in range(kh) if (a + ph - i) % 2 == 0]
        U = (H - a + 1) // 2
        for b in (0, 1):
            J = [j for j in range(kw) if (b + pw - j) % 2 == 0]
            V = (W - b + 1) // 2
            out.append((a, b, I, J, [(a + ph - i) // 2 for i in I], [(b + pw - j) // 2 for j in J], U, V))
    return out


def _dgrad_s2_ok(geom, M, C, O, N, H, W, Ho, Wo):
    """The input gradient of a stride-2 convolution as four placed implicit GEMMs (hecka_conv_x3_placed_f32)."""
    kh, kw, sh, sw, ph, pw, dh, dw = geom
    if not (knobs.conv_implicit and knobs.conv_x3 and sh == 2 and sw == 2 and dh == 1 and dw == 1 and ops.get_gemm_mode() == 0):
        return False
    L = _lib.lib()
    any_taps = False
    for a, b, I, J, di, dj, U, V in _dgrad_s2_classes(geom, H, W):
        if not I or not J or U == 0 or V == 0:
            continue
        any_taps = True
        if not L.hecka_conv_x3_strided_eligible(M, C, O * len(I) * len(J), N, O, Ho, Wo, U, V):
            return False
    return any_taps and kh * kw > 1      # (a 1 x 1 kernel fills one class in four: the GEMM + zero-filling col2im is as good)


def _dgrad_s2(W, dy, geom, N, H, Wd, Ho, Wo, scale, wmax, dymax):
    M, O, C, kh, kw = W.shape
    classes = _dgrad_s2_classes(geom, H, Wd)
    full = all(I and J and U and V for _, _, I, J, _, _, U, V in classes)
    dx = (torch.empty if full else torch.zeros)((M, N, C, H, Wd), device=dy.device, dtype=torch.float32)
    tap = _tapmajor(O)
    Wp = W.permute(0, 2, 3, 4, 1) if tap else W.permute(0, 2, 1, 3, 4)      # [M, C, kh, kw, O] | [M, C, O, kh, kw]
    L = _lib.lib()
    fn = L.hecka_conv_x3_tapmajor_f32 if tap else L.hecka_conv_x3_placed_f32
    for a, b, I, J, di, dj, U, V in classes:
        if not I or not J or U == 0 or V == 0:
            continue
        K = O * len(I) * len(J)
        if tap:      # (I, J: every second tap - basic slices, one copy)
            A = Wp[:, :, I[0]::2, J[0]::2].reshape(M, C, K).contiguous()
            tab = _tap_tab(O, Ho, Wo, [(ti, tj) for ti in di for tj in dj], dy.device)
        else:
            A = Wp[:, :, :, I[0]::2, J[0]::2].reshape(M, C, K).contiguous()
            key = ('dgrad2', O, Ho, Wo, tuple(di), tuple(dj), str(dy.device))
            if key not in _TABS:
                o = torch.arange(O, dtype=torch.int64).view(O, 1, 1)
                ti = torch.tensor(di, dtype=torch.int64).view(1, -1, 1)
                tj = torch.tensor(dj, dtype=torch.int64).view(1, 1, -1)
                off = ((o * Ho + ti) * Wo + tj).reshape(-1)
                pk = ((ti << 16) | (tj & 0xffff)).expand(O, -1, -1).reshape(-1)
                _TABS[key] = torch.stack([off, pk], dim=1).to(torch.int32).contiguous().to(dy.device)
            tab = _TABS[key]
        _lib.check(fn(A.data_ptr(), dy.data_ptr(), dx.data_ptr(), tab.data_ptr(), M, C, K, C * K, N * O * Ho * Wo, N, O, Ho, Wo,
                      U, V, 1, 1, H, Wd, 2, 2, a, b, float(scale), wmax.data_ptr(), dymax.data_ptr(), 1,
                      torch.cuda.current_stream().cuda_stream), 'conv_x3_placed')
    return dx


def _direct_ok(geom, C, O):
    """Few-channel direct convolution (hecka_conv_direct_f32): unit stride, no dilation, equal zero padding on both axes."""
    kh, kw, sh, sw, ph, pw, dh, dw = geom
    return (knobs.conv_direct and sh == 1 and sw == 1 and dh == 1 and dw == 1 and ph == pw
            and ph <= kh - 1 and bool(_lib.lib().hecka_conv_direct_eligible(C, O, kh, kw)))


def _direct_weights(Wt, outs):
    """[M, taps, outs] -> zero-padded [M, taps, 8 | 16] (the kernel reads OT consecutive floats per tap)."""
    OT = 8 if outs <= 8 else 16
    if outs == OT:
        return Wt.contiguous()
    wt = Wt.new_zeros(Wt.shape[0], Wt.shape[1], OT)
    wt[:, :, :outs] = Wt
    return wt


def _conv_direct(x, x_stride, wt, bias, bias_scale, M, N, C, H, W, O, kh, kw, pad):
    Ho, Wo = H + 2 * pad - kh + 1, W + 2 * pad - kw + 1
    y = torch.empty((M, N, O, Ho, Wo), device=x.device, dtype=torch.float32)
    _lib.check(_lib.lib().hecka_conv_direct_f32(x.data_ptr(), x_stride, wt.data_ptr(), None if bias is None else bias.data_ptr(),
                                                float(bias_scale), M, N, C, H, W, O, kh, kw, pad, y.data_ptr(),
                                                torch.cuda.current_stream().cuda_stream), 'conv_direct')
    return y


@ops.mode_aware
class BatchedConv2d(torch.autograd.Function):
    """y[m] = conv2d(x[m] | x, scale * W[m], bias_scale * b[m]);  x [M,N,C,H,W] or [N,C,H,W] (shared),
    W [M,O,C,kh,kw], b [M,O] | None  ->  y [M,N,O,Ho,Wo].  groups = 1, zero padding."""

    @staticmethod
    def forward(ctx, x, W, b, scale, bias_scale, geom):
        if not (x.is_cuda and W.is_cuda) or x.dtype != torch.float32 or W.dtype != torch.float32:
            raise RuntimeError('hyper_amd Conv2d needs fp32 GPU tensors; there is no CPU path')
        M, O, C, kh, kw = W.shape
        shared = x.ndim == 4
        x5 = (x.unsqueeze(0) if shared else x).contiguous()
        Mx, N, _, H, Wd = x5.shape
        Ho, Wo = _out_size(H, kh, geom[2], geom[4], geom[6]), _out_size(Wd, kw, geom[3], geom[5], geom[7])
        if Ho <= 0 or Wo <= 0:
            raise RuntimeError('Conv2d kernel %s is larger than the padded input %s' % ((kh, kw), (H, Wd)))
        K, Q = C * kh * kw, N * Ho * Wo
        has_bias = b is not None
        ctx.direct = _direct_ok(geom, C, O)
        if ctx.direct:
            # few channels (LeNet-5): one thread per output position, weights from scalar registers, no patch matrix
            wt = _direct_weights((scale * W).permute(0, 2, 3, 4, 1).reshape(M, K, O), O)
            y = _conv_direct(x5, 0 if shared else N * C * H * Wd, wt, b.contiguous() if has_bias else None, bias_scale,
                             M, N, C, H, Wd, O, kh, kw, geom[4])
            ctx.save_for_backward(x5, W, None, None, None)
            ctx.cfg = (scale, bias_scale, geom, shared, has_bias, (N, Ho, Wo))
            return y
        if not has_bias and _conv_x3_fwd_ok(geom, M, O, K, N, C, H, Wd, Ho, Wo):
            # implicit GEMM: no patch matrix in forward (the weight gradient gathers it in backward), y written as NCHW
            tap = _tapmajor(C)
            if tap:      # K ordered (tap, channel): one copy of the weights per step, no range check per gathered element
                Wf = _tap_weights(W, False)
                tab = _tap_tab(C, H, Wd, [(i - geom[4], j - geom[5]) for i in range(kh) for j in range(kw)], x5.device)
            else:
                Wf = W.reshape(M, O, K).contiguous()
                tab = _conv_tab(C, H, Wd, kh, kw, geom[4], geom[5], x5.device)
            xmax = (cached_absmax(x, Mx) if not shared else None)
            if xmax is None:
                xmax = _absmax_rows(x5.reshape(Mx, -1), Mx)
            wmax = _absmax_rows(Wf.reshape(M, -1), M)
            y = _conv_x3(Wf, x5, M, O, K, 0 if shared else N * C * H * Wd, N, C, H, Wd, tab, scale, wmax, xmax,
                         0 if shared else 1, (Ho, Wo), (geom[2], geom[3]), tap)
            ctx.save_for_backward(x5, W, None, xmax, wmax)
            ctx.cfg = (scale, bias_scale, geom, shared, has_bias, (N, Ho, Wo))
            return y
        cols = _im2col(x5, geom, has_bias)
        Wf = W.reshape(M, O, K)
        if has_bias:      # bias rides as weight column K against the row of ones; alpha = scale rescales it
            Wf = torch.cat([Wf, (bias_scale / scale) * b.reshape(M, O, 1)], dim=2)
        Wf = Wf.contiguous()
        R = Wf.shape[2]
        # interior shapes without a bias row (the ResNet stages at 128 and more channels): the GEMMs run on the fp16-split
        # tile, scaled by max |x| (the patches are copies of x) and max |W|
        xmax = wmax = None
        if not has_bias and (_x3_eligible(M, O, Q, R) or _x3_eligible(M, O, R, Q) or _x3_eligible(M, K, Q, O)):
            xmax = cached_absmax(x, Mx) if not shared else None
            if xmax is None:
                xmax = _absmax_rows(x5.reshape(Mx, -1), Mx)
            wmax = _absmax_rows(Wf.reshape(M, -1), M)
        if xmax is not None and _x3_eligible(M, O, Q, R):
            y = _bgemm_x3(Wf, cols, M, O, Q, R, O * R, 0 if shared else R * Q, True, scale, wmax, xmax, 0 if shared else 1)
        else:
            y = _bgemm(Wf, cols, M, O, Q, R, O * R, 0 if shared else R * Q, True, scale)      # [M,O,Q]
        # the gathered patches are kept for the weight gradient while they are small next to 288 GB of HBM (23 GB for
        # all of ResNet18 at M=10, N=128); larger ones are re-gathered in backward
        keep = cols.numel() * 4 <= int(knobs.conv_keep_cols_gb * 2 ** 30)
        ctx.save_for_backward(x5, W, cols if keep else None, xmax, wmax)
        ctx.cfg = (scale, bias_scale, geom, shared, has_bias, (N, Ho, Wo))
        return y.view(M, O, N, Ho, Wo).transpose(1, 2)

    @staticmethod
    def backward(ctx, dy):
        x5, W, cols, xmax, wmax = ctx.saved_tensors
        scale, bias_scale, geom, shared, has_bias, (N, Ho, Wo) = ctx.cfg
        M, O, C, kh, kw = W.shape
        K, Q = C * kh * kw, N * Ho * Wo
        dy = dy.contiguous()
        dx = dW = db = None
        if ctx.direct:
            H, Wd, pad = x5.shape[3], x5.shape[4], geom[4]
            L = _lib.lib()
            if ctx.needs_input_grad[0]:     # the same kernel on dy with the rotated weights, pad' = k - 1 - pad
                wr = _direct_weights((scale * W).flip(3, 4).permute(0, 1, 3, 4, 2).reshape(M, O * kh * kw, C), C)
                dx = _conv_direct(dy, N * O * Ho * Wo, wr, None, 0.0, M, N, O, Ho, Wo, C, kh, kw, kh - 1 - pad)
                dx = dx.sum(0) if shared else dx
            if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
                groups = C * ((O + 7) // 8 if kh == 5 else (O + 15) // 16)
                want = max(1, min(N, -(-512 // (M * groups))))      # about two workgroups per CU (254 VGPRs: two fit)
                splits = -(-N // -(-N // want))
                dwp = torch.empty((splits, M, O, C, kh * kw), device=dy.device, dtype=torch.float32)
                dbp = torch.empty((splits, M, O), device=dy.device, dtype=torch.float32) if has_bias else None
                _lib.check(L.hecka_conv_direct_wgrad_f32(
                    x5.data_ptr(), 0 if shared else N * C * H * Wd, dy.data_ptr(), M, N, C, H, Wd, O, kh, kw, pad, splits,
                    dwp.data_ptr(), None if dbp is None else dbp.data_ptr(), torch.cuda.current_stream().cuda_stream),
                    'conv_direct_wgrad')
                dW = (scale * dwp.sum(0)).view(W.shape)
                if has_bias:
                    db = bias_scale * dbp.sum(0)
            return dx, dW, db, None, None, None
        dymax = None
        if xmax is not None:
            dymax = cached_absmax(dy, M)
            if dymax is None:
                dymax = _absmax_rows(dy.view(M, -1), M)
        H, Wd = x5.shape[3], x5.shape[4]
        if ctx.needs_input_grad[0] and dymax is not None and _conv_x3_ok(geom, M, C, O * kh * kw, N, O, H, Wd):
            # input gradient as an implicit-GEMM convolution of dy with the rotated weights: no dgrad patch matrix, no col2im
            tap = _tapmajor(O)
            if tap:
                Wrot = _tap_weights(W, True)
                tab = _tap_tab(O, H, Wd, [(i - (kh - 1 - geom[4]), j - (kw - 1 - geom[5])) for i in range(kh) for j in range(kw)],
                               dy.device)
            else:
                Wrot = W.flip(3, 4).transpose(1, 2).reshape(M, C, O * kh * kw).contiguous()
                tab = _conv_tab(O, H, Wd, kh, kw, kh - 1 - geom[4], kw - 1 - geom[5], dy.device)
            dx = _conv_x3(Wrot, dy, M, C, O * kh * kw, N * O * H * Wd, N, O, H, Wd, tab, scale, wmax, dymax, 1, tapmajor=tap)
            dx = dx.sum(0) if shared else dx
        if (ctx.needs_input_grad[0] and dx is None and dymax is not None and not has_bias
                and _dgrad_s2_ok(geom, M, C, O, N, H, Wd, Ho, Wo)):
            # stride 2: one placed implicit GEMM per parity class of the input pixels (no structural zeros, no col2im)
            dx = _dgrad_s2(W, dy, geom, N, H, Wd, Ho, Wo, scale, wmax, dymax)
            dx = dx.sum(0) if shared else dx
        need_dgrad = ctx.needs_input_grad[0] and dx is None
        Rp = (K + 127) // 128 * 128
        L = _lib.lib()
        if (ctx.needs_input_grad[1] and not has_bias and dymax is not None and geom[6:] == (1, 1) and knobs.conv_implicit
                and knobs.conv_x3 and ops.get_gemm_mode() == 0
                and L.hecka_conv_wgrad_x3_strided_eligible(M, O, Rp, N, C, H, Wd, Ho, Wo)):
            # weight gradient with both operands implicit: dy as it lies (NCHW), patches gathered in the kernel (any stride
            # and zero padding: output position (ho, wo) reads the input around (ho sh, wo sw))
            dWp = torch.empty((M, O, Rp), device=dy.device, dtype=torch.float32)
            nbytes = L.hecka_conv_wgrad_x3_strided_workspace_bytes(M, O, Rp, N, C, H, Wd, Ho, Wo)
            ws = torch.empty((max(nbytes, 4) // 4,), device=dy.device, dtype=torch.float32)
            _lib.check(L.hecka_conv_wgrad_x3_strided_f32(
                dy.data_ptr(), x5.data_ptr(), dWp.data_ptr(),
                _conv_tab(C, H, Wd, kh, kw, geom[4], geom[5], dy.device, rows=Rp).data_ptr(), M, O, Rp,
                0 if shared else N * C * H * Wd, N, C, H, Wd, Ho, Wo, int(geom[2]), int(geom[3]), float(scale),
                dymax.data_ptr(), xmax.data_ptr(), 0 if shared else 1, ws.data_ptr(), nbytes,
                torch.cuda.current_stream().cuda_stream), 'conv_wgrad_x3')
            dW = dWp[:, :, :K].reshape(W.shape)
        if dW is not None and not need_dgrad:
            return dx, dW, db, None, None, None
        dyq = dy.transpose(1, 2).reshape(M, O, Q).contiguous()
        if dW is None and (ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2])):
            if cols is None:
                cols = _im2col(x5, geom, has_bias)
            R = cols.shape[1]
            if dymax is not None and _x3_eligible(M, O, R, Q):
                dWf = _bgemm_x3(dyq, cols, M, O, R, Q, O * Q, 0 if shared else R * Q, False, scale, dymax, xmax,
                                0 if shared else 1)
            else:
                dWf = _bgemm(dyq, cols, M, O, R, Q, O * Q, 0 if shared else R * Q, False, scale)   # dy cols^T  [M,O,R]
            dW = dWf[:, :, :K].reshape(W.shape)
            if has_bias:
                db = dWf[:, :, K] * (bias_scale / scale)
            del cols
        if need_dgrad:
            Wt = W.reshape(M, O, K).transpose(1, 2).contiguous()                              # [M,K,O]
            if dymax is not None and _x3_eligible(M, K, Q, O):
                dcols = _bgemm_x3(Wt, dyq, M, K, Q, O, K * O, O * Q, True, scale, wmax, dymax, 1)
            else:
                dcols = _bgemm(Wt, dyq, M, K, Q, O, K * O, O * Q, True, scale)                 # W^T dy  [M,K,Q]
            dx = _col2im(dcols, M, N, C, x5.shape[3], x5.shape[4], geom)
            dx = dx.sum(0) if shared else dx
        return dx, dW, db, None, None, None


CONV2D_DEFAULT_POOLING_ARGS = {'kernel_size': 2}
_POOLING = {'max': F.max_pool2d, 'avg': F.avg_pool2d,
            'adaptive_max': F.adaptive_avg_pool2d,       # sic: the reference maps both names to the avg pool
            'adaptive_avg': F.adaptive_avg_pool2d}


class _ConvNd(ParametersGroup):
    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation, transposed, output_padding,
                 groups, bias, padding_mode, act=None, gamma=1.0, param_scale='torch', track=True):
        if groups <= 0:
            raise ValueError('groups must be a positive integer')
        if in_channels % groups != 0:
            raise ValueError('in_channels must be divisible by groups')
        if out_channels % groups != 0:
            raise ValueError('out_channels must be divisible by groups')
        if isinstance(padding, str):
            if padding not in ('same', 'valid'):
                raise ValueError("Invalid padding string %r, should be one of {'same', 'valid'}" % padding)
            if padding == 'same' and any(s != 1 for s in stride):
                raise ValueError("padding='same' is not supported for strided convolutions")
        if padding_mode not in ('zeros', 'reflect', 'replicate', 'circular'):
            raise ValueError("padding_mode must be one of zeros/reflect/replicate/circular, but got padding_mode='%s'"
                             % padding_mode)
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding, self.dilation = kernel_size, stride, padding, dilation
        self.transposed, self.output_padding, self.groups = transposed, output_padding, groups
        self.padding_mode, self.param_scale, self.bias = padding_mode, param_scale, bias
        if isinstance(padding, str):
            self._reversed_padding_repeated_twice = [0, 0] * len(kernel_size)
            if padding == 'same':
                for d, k, i in zip(dilation, kernel_size, range(len(kernel_size) - 1, -1, -1)):
                    total = d * (k - 1)
                    self._reversed_padding_repeated_twice[2 * i] = total // 2
                    self._reversed_padding_repeated_twice[2 * i + 1] = total - total // 2
        else:
            self._reversed_padding_repeated_twice = _reverse_repeat_tuple(padding, 2)
        self.weight_shape = ((in_channels, out_channels // groups) if transposed
                             else (out_channels, in_channels // groups)) + tuple(kernel_size)
        self.bias_shape = (out_channels,)
        self.fan_in = (in_channels // groups) * int(np.prod(kernel_size))
        if param_scale == 'gamma':
            self.invsq = math.sqrt(1 / self.fan_in)
            self.gamma = activation.activation_gamma(gamma)
            self.scale = self.gamma * self.invsq
            self.bias_scale = self.invsq
        elif param_scale == 'torch':
            gain = init.calculate_gain('leaky_relu', math.sqrt(5.0))
            self.scale = math.sqrt(gain / math.sqrt(self.fan_in))
            self.bias_scale = math.sqrt(1.0 / math.sqrt(self.fan_in))
        else:
            raise ValueError('Invalid paramater scaling method %s' % param_scale)
        if isinstance(act, str):
            if hasattr(F, act):
                act = getattr(F, act)
            elif hasattr(activation, act):
                act = getattr(activation, act)
            else:
                raise ValueError('Could not find activation %s in either torch.nn.functional or net.activation' % act)
        self.act = act
        shapes = [('weight', self.weight_shape)] + ([('bias', self.bias_shape)] if bias else [])
        super().__init__(shapes=shapes, track=track)

    def gen_initialized_params(self, dtype=None, device=None, gain=1.0):
        if gain is None or not isinstance(gain, float):
            raise ValueError('Gain must be a float')
        params = self.gen_empty_params(dtype=dtype, device=device)
        with torch.no_grad():
            for p in params.values():
                if self.param_scale == 'gamma':
                    init.normal_(p, 0.0, 1.0)
                else:
                    init.uniform_(p, -math.sqrt(3.0), math.sqrt(3.0))
        return params


@register_gen_module('conv2d')
class Conv2d(_ConvNd):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 padding_mode='zeros', act=None, pooling=None, pooling_kwargs=CONV2D_DEFAULT_POOLING_ARGS, gamma=1.0,
                 param_scale='torch', track=True):
        if isinstance(pooling, str):
            if pooling not in _POOLING:
                raise KeyError('Provided pooling "%s" was not found. Available keys: %s' % (pooling, ', '.join(_POOLING)))
            pooling = _POOLING[pooling]
        self.pooling, self.pooling_kwargs = pooling, pooling_kwargs
        super().__init__(in_channels, out_channels, _pair(kernel_size), _pair(stride),
                         padding if isinstance(padding, str) else _pair(padding), _pair(dilation), False, _pair(0),
                         groups, bias, padding_mode, act, gamma, param_scale, track)

    def _geometry(self):
        if self.padding_mode != 'zeros' or isinstance(self.padding, str):
            # the input is padded explicitly first (conv.py:306-316); 'same' pads asymmetrically for even kernels
            return self.kernel_size + self.stride + (0, 0) + self.dilation
        return self.kernel_size + self.stride + tuple(self.padding) + self.dilation

    def _forward(self, viewed, x):
        W = viewed['weight']
        M = W.shape[0]
        if x.ndim == 4:
            assert x.shape[1] == self.in_channels, 'Input channel mismatch!'
        elif x.ndim == 5:
            assert x.shape[0] == M, 'Conv2d input x has wrong shape %s' % (tuple(x.shape),)
            assert x.shape[2] == self.in_channels, 'Input channel mismatch!'
        else:
            raise RuntimeError('Invalid input shape to batched Conv2d %s expecting ndim to be 4 or 5' % (tuple(x.shape),))
        if self.padding_mode != 'zeros' or (isinstance(self.padding, str) and self.padding == 'same'):
            mode = 'constant' if self.padding_mode == 'zeros' else self.padding_mode
            lead = x.shape[:-3]
            x = F.pad(x.reshape((-1,) + x.shape[-3:]), self._reversed_padding_repeated_twice, mode=mode)
            x = x.reshape(lead + x.shape[-3:])
        b = viewed['bias'] if self.bias else None
        g = self.groups
        if g == 1:
            y = BatchedConv2d.apply(x, W, b, self.scale, self.bias_scale, self._geometry())
        else:
            # grouped convolution (conv.py:334): group k of member m is its own small convolution - fold the groups into
            # the member dimension, run the same batched kernels on M * g "members", unfold
            O, Cg = W.shape[1], W.shape[2]
            if x.ndim == 4:
                x = x.unsqueeze(0).expand((M,) + tuple(x.shape))
            N = x.shape[1]
            xg = x.reshape((M, N, g, Cg) + tuple(x.shape[-2:])).permute(0, 2, 1, 3, 4, 5).reshape((M * g, N, Cg) + tuple(x.shape[-2:]))
            yg = BatchedConv2d.apply(xg.contiguous(), W.reshape((M * g, O // g, Cg) + tuple(W.shape[3:])).contiguous(),
                                     b.reshape(M * g, O // g).contiguous() if b is not None else None, self.scale, self.bias_scale,
                                     self._geometry())
            y = yg.reshape((M, g, N, O // g) + tuple(yg.shape[-2:])).permute(0, 2, 1, 3, 4, 5).reshape((M, N, O) + tuple(yg.shape[-2:]))
        if (self.act is activation.crater and self.pooling is F.max_pool2d
                and activation.crater_maxpool2_eligible(y, self.pooling_kwargs)):
            return activation.crater_maxpool2(y).contiguous()      # act -> 2x2 max pool in one pass
        if self.act is not None:
            y = self.act(y)
        if self.pooling is not None:
            N = y.shape[1]
            y = self.pooling(y.reshape((M * N,) + y.shape[2:]), **self.pooling_kwargs)
            y = y.view((M, N) + y.shape[1:])
        return y.contiguous()

    def extra_repr(self):
        return 'in_channels=%d, out_channels=%d, kernel_size=%s, stride=%s, padding=%s, groups=%d' % (
            self.in_channels, self.out_channels, self.kernel_size, self.stride, self.padding, self.groups)


@register_gen_module('final_conv2d')
class FinalConv2d(Conv2d):
    """Marker subclass: lets a generator treat the output layer differently."""
