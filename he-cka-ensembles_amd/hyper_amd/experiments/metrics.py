"""The `ckas` metric (hyper/experiments/metrics.py: registry :11-21, `Metric` :24-51, `CKASMetric` :54-101): mean
pairwise CKA of every tracked layer, logged per step.  The reference recomputes pairwise_cossim from each layer's
[M,B,B] Gram and synchronises once per layer; here the inner products <C_i, C_j> that the loss already computed
(`track['layer_track'][l]['gram_inner']`) are reused - a few [M,M] tensor ops - and all layers come back in ONE host
read.  Layers without that by-product (un-fused kernels) fall back to hyper_amd.diversity.pairwise_cossim."""
import torch

from ..diversity import pairwise_cossim

AVAILABLE_METRICS = {}


def register_metric(name):
    def deco(cls):
        AVAILABLE_METRICS[name] = cls
        return cls
    return deco


def build_metric(config):
    config = dict(config)
    return AVAILABLE_METRICS[config.pop('name')](**config)


class Metric:
    def log(self, train_test, trainer, model_bs, X, Y, track, *args, **kwargs):
        return {}

    def log_val(self, *args, **kwargs):
        return self.log(*args, **kwargs)

    def log_train(self, *args, **kwargs):
        return self.log(*args, **kwargs)

    def reset(self):
        pass


def mean_cka_from_inner(S, nsq, eps=1e-6):
    """mean over i != j of S_ij / ((sqrt(nsq_i) + eps)(sqrt(nsq_j) + eps)) = pairwise_cossim(gram, eps).mean()"""
    d = torch.sqrt(nsq.double()) + eps
    cos = S.double() / (d[:, None] * d[None, :])
    M = S.shape[0]
    return (cos.sum() - torch.diagonal(cos).sum()) / (M * (M - 1))


@register_metric('ckas')
class CKASMetric(Metric):
    def __init__(self, track_kernel='ind'):
        self.kernel = track_kernel

    @torch.no_grad()
    def log(self, train_test, trainer, model_bs, X, Y, track, *args, **kwargs):
        if self.kernel is None:
            raise NotImplementedError('CKAS with None kernel is not implemented yet.')
        try:
            track = track['%s_kernel' % self.kernel]
        except KeyError:
            print('Could not report CKAS! %s kernel was not found in the layer track.' % self.kernel)
            return {}
        if 'layer_track' not in track:
            print('Could not report CKAS! Gram matrices were not found in the layer track.')
            return {}
        names, vals = [], []
        for ind, c in enumerate(track['layer_track']):
            if 'gram' not in c:
                continue
            if 'gram_inner' in c and c['gram_inner'][0].shape[0] == c['gram_inner'][0].shape[1]:
                vals.append(mean_cka_from_inner(*c['gram_inner']))
            else:
                vals.append(pairwise_cossim(c['gram']).mean().double())
            names.append('cka_%s_%d/%s' % (self.kernel, ind + 1, train_test))
        if not vals:
            return {}
        return dict(zip(names, torch.stack(vals).tolist()))     # one host read for all layers

    def log_val(self, *args, **kwargs):
        return self.log('val', *args, **kwargs)

    def log_train(self, *args, **kwargs):
        return self.log('train', *args, **kwargs)
