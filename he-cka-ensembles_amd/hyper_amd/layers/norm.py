"""Batched BatchNorm (hyper/layers/norm.py:12-46 `_NormAffine` / `_NormBuffers`, :49-127 `_NormBase`, :129-224
`_BatchNorm.forward`, BatchNorm1d / BatchNorm2d).  The reference loops F.batch_norm over the ensemble members and stacks;
here all members go through ONE launch (libhecka hecka_batchnorm_fwd_f32 / _bwd_f32: a workgroup per (member, channel)).
On the fused path (hecka_batchnorm_act_*_packed_f32) the kernels read weight | bias from the layer's flat parameter row
and return their gradients as one tensor, and the forward kernel updates the running statistics itself when every member
is selected in order; otherwise the selected members' statistics are updated with a handful of [members, C] tensor ops."""
import torch
import torch.nn as nn

from .. import _lib
from ..config import knobs
from .module import GenModule, ParametersGroup


class _NormAffine(ParametersGroup):
    def __init__(self, num_features):
        super().__init__(shapes=[('weight', (num_features,)), ('bias', (num_features,))])

    def gen_initialized_params(self, dtype=None, device=None, gain=1):
        params = self.gen_empty_params(dtype=dtype, device=device, requires_grad=True)
        with torch.no_grad():
            nn.init.ones_(params['weight'])
            nn.init.zeros_(params['bias'])
        return params


class _NormBuffers(ParametersGroup):
    def __init__(self, num_features):
        super().__init__(shapes=[('mean', (num_features,)), ('var', (num_features,)), ('num_batches_tracked', (1,))])

    def gen_initialized_params(self, dtype=None, device=None, gain=1):
        params = self.gen_empty_params(dtype=dtype, device=device, requires_grad=False)
        nn.init.zeros_(params['mean'])
        nn.init.ones_(params['var'])
        nn.init.zeros_(params['num_batches_tracked'])
        return params


class BatchedBatchNorm(torch.autograd.Function):
    """y[m] = batch_norm(x[m], weight[m], bias[m]);  x [M,N,C,*], weight / bias [M,C] | None.  Returns (y, batch mean,
    biased batch variance) - the statistics are by-products for the running averages and carry no gradient."""

    @staticmethod
    def forward(ctx, x, weight, bias, run_mean, run_var, training, eps):
        if not x.is_cuda or x.dtype != torch.float32:
            raise RuntimeError('hyper_amd BatchNorm needs fp32 GPU tensors; there is no CPU path')
        x = x.contiguous()
        M, N, C = x.shape[:3]
        S = int(x[0, 0, 0].numel())
        w = weight.contiguous() if weight is not None else None
        b = bias.contiguous() if bias is not None else None
        y = torch.empty_like(x)
        mean = torch.empty((M, C), device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        var = torch.empty_like(mean) if training else None
        ptr = lambda t: t.data_ptr() if t is not None else None   # noqa: E731
        _lib.check(_lib.lib().hecka_batchnorm_fwd_f32(
            x.data_ptr(), M, N, C, S, ptr(w), ptr(b), ptr(run_mean), ptr(run_var), int(training), float(eps),
            y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), ptr(var), torch.cuda.current_stream().cuda_stream),
            'batchnorm_fwd')
        ctx.save_for_backward(x, mean, rstd, w)
        ctx.cfg = (training, S, weight is not None, bias is not None)
        if var is None:
            var = mean.new_zeros(())
        ctx.mark_non_differentiable(mean, var)
        return y, mean, var

    @staticmethod
    def backward(ctx, dy, _dm, _dv):
        x, mean, rstd, w = ctx.saved_tensors
        training, S, has_w, has_b = ctx.cfg
        M, N, C = x.shape[:3]
        dy = dy.contiguous()
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw = torch.empty_like(mean) if has_w and ctx.needs_input_grad[1] else None
        db = torch.empty_like(mean) if has_b and ctx.needs_input_grad[2] else None
        ptr = lambda t: t.data_ptr() if t is not None else None   # noqa: E731
        _lib.check(_lib.lib().hecka_batchnorm_bwd_f32(
            dy.data_ptr(), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), ptr(w), M, N, C, S, int(training),
            ptr(dx), ptr(dw), ptr(db), torch.cuda.current_stream().cuda_stream), 'batchnorm_bwd')
        return dx, dw, db, None, None, None, None


def tag_absmax(t, amax):
    """Remember max |t[g]| (a kernel by-product, [G] on the device) on the tensor object; `cached_absmax` gives it to the next
    consumer as long as the tensor has not been written since."""
    t._hecka_absmax = (amax, t._version)


def cached_absmax(t, rows):
    tag = getattr(t, '_hecka_absmax', None)
    if tag is not None and tag[1] == t._version and tag[0].shape[0] == rows and t.shape[0] == rows:
        return tag[0]
    return None


class BatchedBatchNormAct(torch.autograd.Function):
    """y[m] = [relu](batch_norm(x[m], weight[m], bias[m]) [+ res[m]]) in one launch each way (hecka_batchnorm_act_*): the
    two tails of a ResNet block, norm -> relu and norm -> + shortcut -> relu.  Same by-products as BatchedBatchNorm.
    packed: `weight` is the layer's flat parameter row [M, 2 C] (weight | bias) as the batched layers keep it - read in
    place, its gradient returned as one tensor (no slice copies either way); upd = (running mean store, running var
    store, factor): the kernel updates the running statistics itself (then the returned statistics are not needed)."""

    @staticmethod
    def forward(ctx, x, res, weight, bias, run_mean, run_var, training, eps, relu, packed=False, upd=None):
        if not x.is_cuda or x.dtype != torch.float32:
            raise RuntimeError('hyper_amd BatchNorm needs fp32 GPU tensors; there is no CPU path')
        x = x.contiguous()
        res = res.contiguous() if res is not None else None
        M, N, C = x.shape[:3]
        S = int(x[0, 0, 0].numel())
        ptr = lambda t: t.data_ptr() if t is not None else None   # noqa: E731
        if packed:
            w = weight.contiguous()
            assert bias is None and tuple(w.shape) == (M, 2 * C) and w.dtype == torch.float32
            w_ptr, b_ptr, wstride = w.data_ptr(), w.data_ptr() + 4 * C, 2 * C
        else:
            w = weight.contiguous() if weight is not None else None
            b = bias.contiguous() if bias is not None else None
            w_ptr, b_ptr, wstride = ptr(w), ptr(b), C
        y = torch.empty_like(x)
        ymax = torch.empty((M,), device=x.device, dtype=torch.float32)
        mean = torch.empty((M, C), device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        var = torch.empty_like(mean) if training and upd is None else None
        # all the backward needs of y is the relu's sign: one byte per four outputs instead of y in both of its sweeps
        mask = (torch.empty((x.numel() // 4,), device=x.device, dtype=torch.uint8)
                if relu and knobs.bn_mask and any(ctx.needs_input_grad[:4]) else None)
        _lib.check(_lib.lib().hecka_batchnorm_act_fwd_packed_f32(
            x.data_ptr(), ptr(res), M, N, C, S, w_ptr, b_ptr, wstride, ptr(run_mean), ptr(run_var), int(training), float(eps),
            int(bool(relu)), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), ptr(var), ymax.data_ptr(),
            upd[0].data_ptr() if upd is not None else None, upd[1].data_ptr() if upd is not None else None,
            float(upd[2]) if upd is not None else 0.0, ptr(mask), torch.cuda.current_stream().cuda_stream), 'batchnorm_act_fwd')
        tag_absmax(y, ymax)      # the convolution that reads y next takes its operand scale from here
        ctx.save_for_backward(x, mean, rstd, w, y if relu and mask is None else None, mask)
        ctx.cfg = (training, S, weight is not None, bias is not None or packed, res is not None, packed, bool(relu))
        if var is None:
            var = mean if upd is not None else mean.new_zeros(())
        ctx.mark_non_differentiable(mean, var)
        return y, mean, var

    @staticmethod
    def backward(ctx, dy, _dm, _dv):
        x, mean, rstd, w, y, mask = ctx.saved_tensors
        training, S, has_w, has_b, has_res, packed, relu = ctx.cfg
        M, N, C = x.shape[:3]
        dy = dy.contiguous()
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dxmax = torch.empty((M,), device=x.device, dtype=torch.float32) if dx is not None else None
        dres = None
        if has_res and ctx.needs_input_grad[1]:
            dres = torch.empty_like(x) if relu else dy      # no relu: the residual's gradient is dy itself
        ptr = lambda t: t.data_ptr() if t is not None else None   # noqa: E731
        if packed:
            dw = torch.empty((M, 2 * C), device=x.device, dtype=torch.float32) if ctx.needs_input_grad[2] else None
            db = None
            dw_ptr, db_ptr, wstride = ptr(dw), (dw.data_ptr() + 4 * C if dw is not None else None), 2 * C
        else:
            dw = torch.empty_like(mean) if has_w and ctx.needs_input_grad[2] else None
            db = torch.empty_like(mean) if has_b and ctx.needs_input_grad[3] else None
            dw_ptr, db_ptr, wstride = ptr(dw), ptr(db), C
        _lib.check(_lib.lib().hecka_batchnorm_act_bwd_packed_f32(
            dy.data_ptr(), x.data_ptr(), ptr(y), mean.data_ptr(), rstd.data_ptr(), ptr(w), wstride, M, N, C, S, int(training),
            ptr(dx), ptr(dres) if relu else None, dw_ptr, db_ptr, ptr(dxmax), ptr(mask),
            torch.cuda.current_stream().cuda_stream), 'batchnorm_act_bwd')
        if dx is not None:
            tag_absmax(dx, dxmax)    # dx is the output gradient of the convolution in front of this norm
        return dx, dres, dw, db, None, None, None, None, None, None, None


def fused_act_eligible(x):
    """The fused norm (+ residual) + relu launch takes fp32 GPU tensors with a multiple of four positions per channel."""
    return (knobs.bn_fuse and x.is_cuda and x.dtype == torch.float32 and x.dim() >= 3
            and int(x[0, 0, 0].numel()) % 4 == 0)


class _NormBase(GenModule):
    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True, track=True):
        super().__init__(track=track)
        self.num_features, self.eps, self.momentum = num_features, eps, momentum
        self.affine, self.track_running_stats = affine, track_running_stats
        self.affine_module = _NormAffine(num_features) if affine else None
        self.buffer_module = _NormBuffers(num_features) if track_running_stats else None

    def is_generated(self):
        return False

    def define_generated_modules(self):
        mod = super().define_generated_modules()
        if self.affine:
            mod['affine'] = self.affine_module.define_generated_modules()
        if self.track_running_stats:      # fixed ensembles only (as in the reference)
            mod['norm_buffer'] = self.buffer_module.define_generated_modules()
        return mod

    def _check_input_dim(self, input):
        raise NotImplementedError

    def extra_repr(self):
        return '{num_features}, eps={eps}, momentum={momentum}, affine={affine}, ' \
               'track_running_stats={track_running_stats}'.format(**self.__dict__)


class _BatchNorm(_NormBase):
    def forward(self, params, input, residual=None, relu=False):
        """`residual` / `relu` (not in the reference's signature): y = [relu](norm(input) [+ residual]) as ONE launch when
        the shape allows (fused_act_eligible), else the same thing with torch ops behind the plain norm."""
        self._check_input_dim(input)
        fused = fused_act_eligible(input) and (relu or residual is not None or self.affine)
        weight = bias = None
        bufs = params['norm_buffer']['self'] if self.track_running_stats else None
        bn_training = self.training or not self.track_running_stats
        run_mean = run_var = None
        if bufs is not None and not bn_training:
            run_mean, run_var = bufs.gather('mean'), bufs.gather('var')
        # the running statistics updated inside the forward kernel: every member in order, a fixed momentum
        upd = None
        if fused and self.training and bufs is not None and self.momentum is not None and bufs.whole('mean') and bufs.whole('var'):
            upd = (bufs.store('mean'), bufs.store('var'), float(self.momentum))
        flat = params['affine']['self'] if self.affine else None
        packed = (fused and flat is not None and flat.is_cuda and flat.dtype == torch.float32 and flat.dim() == 2
                  and flat.shape[1] == 2 * self.num_features)
        if self.affine and not packed:
            viewed = self.affine_module.from_flat(flat)
            weight, bias = viewed['weight'], viewed['bias']
        if fused:
            y, mean, var = BatchedBatchNormAct.apply(input, residual, flat if packed else weight, None if packed else bias,
                                                     run_mean, run_var, bn_training, self.eps, bool(relu), packed, upd)
        else:
            y, mean, var = BatchedBatchNorm.apply(input, weight, bias, run_mean, run_var, bn_training, self.eps)
            if residual is not None:
                y = y + residual
            if relu:
                y = torch.relu(y)
        if self.training and bufs is not None and upd is None:
            cnt = input.shape[1] * int(input[0, 0, 0].numel())
            with torch.no_grad():
                if self.momentum is None:   # cumulative average (num_batches_tracked is never advanced by the reference)
                    factor = 1.0 / bufs.gather('num_batches_tracked')
                else:
                    factor = self.momentum
                bufs.update('mean', mean, factor)
                bufs.update('var', var * (cnt / max(cnt - 1, 1)), factor)
        return self.track_feature(y), y


class BatchNorm1d(_BatchNorm):
    def _check_input_dim(self, input):
        if input.dim() != 3 and input.dim() != 4:
            raise ValueError('expected 3D or 4D input (got %dD input)' % input.dim())


class BatchNorm2d(_BatchNorm):
    def _check_input_dim(self, input):
        if input.dim() != 5:
            raise ValueError('expected 5D input (got %dD input)' % input.dim())
