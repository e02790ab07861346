"""Training "methods" on the hot path (hyper/diversity/methods.py: registry :18-39, ParticleMethods `mle`
:42-164, SVGD `svgd` :167-226 with feature-space (`model_layers`) kernels, RepulsiveKernel `repulsive_kernel`
:309-360).  Weight-space SVGD (`weight` kernels) and `function_svgd` are not built."""
import copy
import traceback
from collections import OrderedDict

import torch
import torch.nn as nn

from .. import ops
from ..util.collections import flatten_keys
from .losses import build_loss
from .model_kernels import FunctionKernel, WeightKernel, build_model_kernel
from .ssge import SpectralSteinEstimator

AVAILABLE_METHODS = {}


def register_method(name):
    def deco(cls):
        AVAILABLE_METHODS[name] = cls
        return cls
    return deco


def build_method(config):
    config = copy.deepcopy(config)
    if 'name' not in config:
        raise RuntimeError('method config needs a `name` key')
    cls = AVAILABLE_METHODS[config.pop('name')]
    return cls(**cls.from_config(config))


@register_method('mle')
class ParticleMethods(nn.Module):
    def __init__(self, num, model_kernel=None, ind_loss=None, ood_loss=None, beta_ind=1.0, beta_ood=1.0, l2_reg=None,
                 learnable_beta=False):
        super().__init__()
        self.num, self.ind_loss, self.ood_loss, self.model_kernel = num, ind_loss, ood_loss, model_kernel
        self.l2_reg = (l2_reg / 2.0) if l2_reg is not None else None
        self.beta_ind = nn.Parameter(torch.tensor(beta_ind, dtype=torch.float32), requires_grad=learnable_beta)
        self.beta_ood = nn.Parameter(torch.tensor(beta_ood, dtype=torch.float32), requires_grad=learnable_beta)

    def batch_start(self, trainer, X, Y):
        trainer.optim_hyper.zero_grad()
        trainer.optim_method.zero_grad()

    def batch_end(self, trainer, X, Y):
        try:   # non-finite gradients skip the step instead of poisoning the weights
            if trainer.clip is not None:
                torch.nn.utils.clip_grad_norm_(trainer.hyper.parameters(), trainer.clip, norm_type=2, error_if_nonfinite=True)
            trainer.optim_hyper.step()
            trainer.optim_method.step()
        except RuntimeError as err:
            traceback.print_exc()
            print('Clip error! %s' % err)

    def forward(self, trainer, params, shared_params, track, x, gt, ignore_ood_loss=False, step=0):
        track['ind_loss'] = self.ind_loss(self.num, x, track['pred_ind'], gt) if self.ind_loss is not None else None
        track['loss'] = torch.abs(self.beta_ind) * track['ind_loss']
        if not ignore_ood_loss and self.ood_loss is not None:
            if track['pred_ood'] is None:
                raise RuntimeError('OOD loss was defined but no OOD predictions were passed')
            track = self.forward_ood(trainer, params, shared_params, track, x, gt)
        if self.l2_reg is not None:
            l2 = 0.0
            for name, value in flatten_keys(params).items():
                if 'affine' in name or 'bias' in name or 'skip' in name:
                    continue
                l2 = l2 + torch.square(value).view(value.shape[0], -1).mean(dim=1).mean()
            track['loss'] = track['loss'] + self.l2_reg * l2
        return track

    def forward_ood(self, trainer, params, shared_params, track, x, gt):
        track['ood_loss'] = self.ood_loss(self.num, x, track['pred_ood'], gt)
        ood = torch.abs(self.beta_ood) * track['ood_loss']
        track['loss'] = track['loss'] + ood if 'loss' in track else ood
        return track

    @torch.no_grad()
    def test(self, trainer, params, shared_params, track, x, gt):
        return track['loss'], {}

    def backward(self, trainer, params, shared_params, track, x, gt, retain_graph=False, create_graph=False, inputs=None):
        track['loss'].backward(retain_graph=retain_graph, create_graph=create_graph, inputs=inputs)
        return track['loss'], {}

    @staticmethod
    def from_config(config):
        config['model_kernel'] = build_model_kernel(config['model_kernel']) if 'model_kernel' in config else None
        for key in ('ind_loss', 'ood_loss'):
            if key in config:
                config[key] = build_loss(config[key])
        return config


def filter_params(param_dict):
    """Flattened parameter dict without BatchNorm buffers (WeightKernel.filter_params, model_kernels.py:166-173)."""
    return OrderedDict((k, v) for k, v in flatten_keys(param_dict).items() if '.norm_buffer.' not in k)


@register_method('svgd')
class SVGD(ParticleMethods):
    """Stein variational gradient descent on the particles with a FEATURE-space kernel - the [M,M] matrix of a
    `model_layers` kernel with `reduction: none` (configs/*/svgd/{cka,he}-*.yml) - or a WEIGHT-space kernel (`weight`:
    rbf / ... Gram of the members' flattened parameters, configs/*/svgd/rbf*.yml).  Per flat parameter block p:
        grad_p = (K @ d(-loss)/dp - dK.sum()/dp) / num         (kde: d(-loss)/dp - (dK.sum()/dp) / K.sum(1))
    and -grad_p is pushed back through the generator.  The kernel matrix and its gradient come from the HIP path
    (ops.diversity_layer with reduce=None: head_bwd takes the all-ones upstream matrix)."""

    def __init__(self, num, model_kernel, ind_loss, kde=False, beta_ind=1.0):
        super().__init__(num, model_kernel, ind_loss, beta_ind=beta_ind)
        self.weight_space = isinstance(model_kernel, WeightKernel)
        self.kde = kde

    @staticmethod
    def grads(var, grad_var, retain_graph=True, create_graph=False, **kwargs):
        return torch.autograd.grad(var.sum(), grad_var, retain_graph=retain_graph, create_graph=create_graph, **kwargs)

    def backward(self, trainer, params, shared_params, track, x, gt, retain_graph=False, create_graph=False, inputs=None):
        flat_params = filter_params(params)
        all_flat = list(flat_params.values()) + list(shared_params)
        loss = self.beta_ind * track['ind_loss']
        drive = self.grads(-loss, all_flat, retain_graph=True, create_graph=create_graph, allow_unused=True)
        if self.weight_space:      # the kernel sees ONE flat row per member (methods.py:207-209)
            kern = self.model_kernel(WeightKernel.flatten_weights(params), track['feats_ind'], track['pred_ind'], track)
        else:
            kern = self.model_kernel(flat_params, track['feats_ind'], track['pred_ind'], track)
        kgrads = self.grads(kern, all_flat, retain_graph=True, create_graph=create_graph, allow_unused=True)
        for p, gd, gk in zip(all_flat, drive, kgrads):
            if gd is None:
                continue
            if self.kde:
                grad = gd - gk / kern.sum(1, keepdim=True)
            else:
                grad = (torch.matmul(kern, gd) - gk) / self.num
            p.backward(inputs=inputs, gradient=-grad, retain_graph=True, create_graph=create_graph)
        return loss, {}


@register_method('function_svgd')
class FunctionSVGD(ParticleMethods):
    """SVGD in function space (methods.py:229-306, configs/toy/fourclass/fsvgd/*.yml): the particles are the members'
    predictions on the batch; update = K (score + prior score) - dK / num, pushed back through the predictions.  The
    likelihood score is d(-loss / exp_std^2)/dpred; the functional prior's score is estimated (SSGE) from the predictions
    of `num` members with N(0,1) parameters; K is the `function` kernel (the HIP rbf Gram of [members, samples * classes])."""

    def __init__(self, num, model_kernel, ind_loss, exp_std=1.0, eta=0.05, kde=False):
        super().__init__(num, model_kernel, ind_loss)
        if not isinstance(model_kernel, FunctionKernel):
            raise ValueError('Weight kernels/feature kernels are not supported for function SVGD. Use normal SVGD for that')
        self.ssge = SpectralSteinEstimator(eta)
        self.exp_std, self.kde = exp_std, kde

    grads = staticmethod(SVGD.grads)

    def backward(self, trainer, params, shared_params, track, x, gt, retain_graph=False, create_graph=False, inputs=None):
        flat_params = flatten_keys(params)
        loss, pred = track['ind_loss'], track['pred_ind']
        score = self.grads(-loss / (self.exp_std ** 2), pred, retain_graph=True, create_graph=create_graph)[0].reshape(self.num, -1)
        module = trainer.hyper.module if getattr(trainer, 'ddp', False) else trainer.hyper
        prior_params = module.forward_params(module.sample_random_params(self.num, device=pred.device), device=pred.device)
        ood_n = getattr(trainer, 'ood_N', 0)
        _, pred_prior = module.forward(prior_params, x[:-ood_n] if ood_n > 0 else x)
        grad_prior = self.ssge.compute_score_gradients(pred.reshape(self.num, -1), pred_prior.reshape(self.num, -1))
        kern = self.model_kernel(flat_params, track['feats_ind'], pred, track)
        kgrad = self.grads(kern, pred, retain_graph=True, create_graph=create_graph, allow_unused=True)[0].reshape(self.num, -1)
        if self.kde:
            grad = (score + grad_prior) - (kgrad / kern.sum(1, keepdim=True)) / self.num
        else:
            grad = torch.matmul(kern, score + grad_prior) - kgrad / self.num
        pred.backward(gradient=-grad.reshape(pred.shape), retain_graph=retain_graph, create_graph=create_graph)
        return loss, {}


class _MaskNaN(torch.autograd.Function):
    """NaN -> 0 in the value; the upstream gradient of a NaN entry is dropped (exact zero), not multiplied."""

    @staticmethod
    def forward(ctx, rep):
        nan = torch.isnan(rep)
        ctx.save_for_backward(nan)
        return torch.where(nan, torch.zeros_like(rep), rep)

    @staticmethod
    def backward(ctx, g):
        (nan,) = ctx.saved_tensors
        return torch.where(nan, torch.zeros_like(g), g)


@register_method('repulsive_kernel')
class RepulsiveKernel(ParticleMethods):
    """loss = |beta_ind| CE + c_warm |gamma| rep (+ c_warm |gamma_ood| rep_ood); a NaN repulsive term counts as 0."""

    def __init__(self, num, model_kernel, ind_loss, ood_loss=None, gamma=1.0, model_kernel_ood=None, beta_ind=1.0,
                 beta_ood=1.0, gamma_ood=1.0, warmup=0, l2_reg=None, learnable_beta=False, learnable=False):
        super().__init__(num, model_kernel, ind_loss, ood_loss, beta_ind=beta_ind, l2_reg=l2_reg, beta_ood=beta_ood,
                         learnable_beta=learnable_beta)
        self.gamma = nn.Parameter(torch.tensor(gamma, dtype=torch.float32), requires_grad=learnable)
        self.use_ood = ood_loss is not None or model_kernel_ood is not None
        if self.use_ood:
            self.gamma_ood = nn.Parameter(torch.tensor(gamma_ood, dtype=torch.float32), requires_grad=learnable)
        self.model_kernel_ood = model_kernel_ood
        self.warmup = warmup
        if isinstance(model_kernel, WeightKernel):
            raise ValueError('Weight kernels are not supported for feature repulsion')

    def forward(self, trainer, params, shared_params, track, x, gt, ignore_ood_loss=False, step=0):
        track = super().forward(trainer, params, shared_params, track, x, gt, ignore_ood_loss=ignore_ood_loss, step=step)
        flat_params = flatten_keys(params)
        warm = 1.0
        if trainer.training and self.warmup is not None and step is not None and step < self.warmup:
            warm = (1e-8 + step) / self.warmup
        track['ind_kernel'] = OrderedDict()
        rep, masked = self._gated_kernel(self.model_kernel, flat_params, track['feats_ind'], track['pred_ind'], track['ind_kernel'])
        track['ind_rep_loss'] = rep
        track['loss'] = track['loss'] + warm * torch.abs(self.gamma) * masked
        if self.use_ood and self.model_kernel_ood is not None:
            track['ood_kernel'] = OrderedDict()
            rep_ood, masked = self._gated_kernel(self.model_kernel_ood, flat_params, track['feats_ood'], track['pred_ood'], track['ood_kernel'])
            track['ood_rep_loss'] = rep_ood
            track['loss'] = track['loss'] + warm * torch.abs(self.gamma_ood) * masked
        return track

    @staticmethod
    def _gated_kernel(kernel, flat_params, feats, pred, ktrack):
        """(rep, rep with NaN counted as 0).  The reference reads `rep.isnan()` on the host and drops a NaN term from the
        loss AND from the graph (methods.py:342-348), so only the data term's gradients flow.  Here nothing is read
        back: the kernel runs inside ops.grad_gate, its autograd nodes keep the one-element gate, and the gate is set
        to 0 on the device when the term is NaN - every dX launch of the term then writes exact zeros (a zero upstream
        alone would give 0 * NaN = NaN in the feature gradients)."""
        feat0 = next(iter(feats.values()), None) if feats is not None and len(feats) else None
        if feat0 is None or not feat0.is_cuda:
            rep = kernel(flat_params, feats, pred, ktrack)
            return rep, (torch.nan_to_num(rep, nan=0.0, posinf=float('inf'), neginf=float('-inf')) if rep is not None else 0.0)
        gate = torch.ones((1,), device=feat0.device, dtype=torch.float32)
        with ops.grad_gate(gate):
            rep = kernel(flat_params, feats, pred, ktrack)
        if rep is None:
            return rep, 0.0
        with torch.no_grad():
            gate.copy_((~torch.isnan(rep.detach()).any()).to(torch.float32).reshape(1))
        return rep, _MaskNaN.apply(rep)

    @staticmethod
    def from_config(config):
        config = ParticleMethods.from_config(config)
        if 'model_kernel_ood' in config:
            config['model_kernel_ood'] = build_model_kernel(config['model_kernel_ood'])
        return config
