"""Batched-weight Linear (hyper/layers/linear.py:13-173): every ensemble member applies ITS OWN weight to its
[samples, in] block.  The reference issues torch.baddbmm; here forward, dgrad and wgrad are grouped GEMMs on
fp32 MFMA (libhecka hecka_bgemm_f32) with the parameter scaling and the bias add folded into the epilogue."""
import math


import torch
import torch.nn.functional as F
import torch.nn.init as init

from .. import _lib, ops
from ..config import knobs
from ..net import activation
from .module import ParametersGroup, register_gen_module


def _bgemm(A, B, batch, rows, cols, K, sA, sB, nn, alpha, bias=None, beta=0.0):
    out = torch.empty((batch, rows, cols), device=A.device, dtype=torch.float32)
    L = _lib.lib()
    nbytes = L.hecka_bgemm_splitk_workspace_bytes(batch, rows, cols, K) if bias is None else 0
    if nbytes:     # small output over a long K (weight gradients): split-K slabs, fixed-order sum
        ws = torch.empty((nbytes // 4,), device=A.device, dtype=torch.float32)
        _lib.check(L.hecka_bgemm_splitk_f32(A.data_ptr(), B.data_ptr(), out.data_ptr(), batch, rows, cols, K, sA, sB,
                                            int(nn), float(alpha), ws.data_ptr(), nbytes,
                                            torch.cuda.current_stream().cuda_stream), 'bgemm_splitk')
        return out
    _lib.check(_lib.lib().hecka_bgemm_f32(A.data_ptr(), B.data_ptr(), out.data_ptr(), batch, rows, cols, K, sA, sB,
                                          int(nn), float(alpha), bias.data_ptr() if bias is not None else None,
                                          float(beta), torch.cuda.current_stream().cuda_stream), 'bgemm')
    return out


def _absmax_rows(V, rows):
    """[rows] maxima of |V| over each of `rows` equal contiguous parts (operand scales of the fp16-split GEMM)."""
    V = V.contiguous()
    out = torch.empty((rows,), device=V.device, dtype=torch.float32)
    _lib.check(_lib.lib().hecka_absmax_rows_f32(V.data_ptr(), rows, V.numel() // rows, out.data_ptr(),
                                                torch.cuda.current_stream().cuda_stream), 'absmax_rows')
    return out


def _x3_eligible(batch, rows, cols, K):
    return (knobs.conv_x3 and ops.get_gemm_mode() == 0      # fp16-split only in HECKA_GEMM_AUTO
            and bool(_lib.lib().hecka_bgemm_x3_eligible(batch, rows, cols, K)))


def _bgemm_x3(A, B, batch, rows, cols, K, sA, sB, nn, alpha, amaxA, amaxB, amaxB_stride):
    """_bgemm on the fp16-split tile (interior shapes; see hecka_bgemm_x3_f32)."""
    out = torch.empty((batch, rows, cols), device=A.device, dtype=torch.float32)
    L = _lib.lib()
    nbytes = L.hecka_bgemm_x3_workspace_bytes(batch, rows, cols, K)
    ws = torch.empty((max(nbytes, 4) // 4,), device=A.device, dtype=torch.float32)
    _lib.check(L.hecka_bgemm_x3_f32(A.data_ptr(), B.data_ptr(), out.data_ptr(), batch, rows, cols, K, sA, sB, int(nn),
                                    float(alpha), amaxA.data_ptr(), amaxB.data_ptr(), int(amaxB_stride), ws.data_ptr(),
                                    nbytes, torch.cuda.current_stream().cuda_stream), 'bgemm_x3')
    return out


@ops.mode_aware
class BatchedLinear(torch.autograd.Function):
    """y[m] = scale * x[m] W[m]^T + bias_scale * b[m];  x [M,N,I] (or [N,I], shared), W [M,O,I], b [M,O] | None."""

    @staticmethod
    def forward(ctx, x, W, b, scale, bias_scale):
        if not (x.is_cuda and W.is_cuda) or x.dtype != torch.float32 or W.dtype != torch.float32:
            raise RuntimeError('hyper_amd Linear needs fp32 GPU tensors; there is no CPU path')
        M, O, I = W.shape
        shared = x.ndim == 2
        x = x.contiguous()
        W = W.contiguous()
        b = b.contiguous() if b is not None else None
        N = x.shape[-2]
        y = _bgemm(x, W, M, N, O, I, 0 if shared else N * I, O * I, False, scale, b, bias_scale)
        ctx.save_for_backward(x, W)
        ctx.cfg = (scale, bias_scale, shared, b is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W = ctx.saved_tensors
        scale, bias_scale, shared, has_bias = ctx.cfg
        M, O, I = W.shape
        N = x.shape[-2]
        dy = dy.contiguous()
        dx = dW = db = None
        if ctx.needs_input_grad[0]:
            dx = _bgemm(dy, W, M, N, I, O, N * O, O * I, True, scale)              # dy [N,O] @ W [O,I]
            if shared:
                dx = dx.sum(0)
        if ctx.needs_input_grad[1]:
            dyt = dy.transpose(1, 2).contiguous()                                   # [M,O,N]
            xt = (x.t() if shared else x.transpose(1, 2)).contiguous()             # [I,N] | [M,I,N]
            dW = _bgemm(dyt, xt, M, O, I, N, O * N, 0 if shared else I * N, False, scale)
        if has_bias and ctx.needs_input_grad[2]:
            db = bias_scale * dy.sum(1)
        return dx, dW, db, None, None


@register_gen_module('linear')
class Linear(ParametersGroup):
    def __init__(self, in_features, out_features, bias=True, act=None, gamma=1.0, param_scale='torch', track=True):
        self.in_features, self.out_features = in_features, out_features
        self.param_scale, self.bias = param_scale, bias
        if param_scale == 'gamma':
            self.invsq = math.sqrt(1 / in_features)
            self.gamma = activation.activation_gamma(gamma)
            self.scale = self.gamma * self.invsq
            self.bias_scale = self.invsq
        elif param_scale == 'torch':
            gain = init.calculate_gain('leaky_relu', math.sqrt(5.0))
            self.scale = math.sqrt(gain / math.sqrt(in_features))
            self.bias_scale = math.sqrt(1.0 / math.sqrt(in_features))
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
        shapes = [('weight', (out_features, in_features))] + ([('bias', (out_features,))] if bias else [])
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

    def _forward(self, viewed, x):
        W = viewed['weight']
        if x.ndim == 2:
            assert x.shape[1] == self.in_features, 'Input has wrong shape %s. Expecting (batch, %d)' % (tuple(x.shape), self.in_features)
        elif x.ndim == 3:
            assert x.shape[0] == W.shape[0] and x.shape[2] == self.in_features, \
                'Input has wrong shape %s. Expecting (%d, B, %d)' % (tuple(x.shape), W.shape[0], self.in_features)
        else:
            raise ValueError('Bad x dimension. Got %d' % x.ndim)
        y = BatchedLinear.apply(x, W, viewed['bias'] if self.bias else None, self.scale, self.bias_scale)
        return y if self.act is None else self.act(y)

    def extra_repr(self):
        return 'in_features=%d, out_features=%d, activation=%s' % (self.in_features, self.out_features, self.act)


@register_gen_module('final_linear')
class FinalLinear(Linear):
    """Marker subclass: lets a generator treat the output layer differently."""
