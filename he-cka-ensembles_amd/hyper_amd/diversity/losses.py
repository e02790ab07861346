"""Data losses of the particle methods (hyper/diversity/losses.py:10-113) - tiny, stock torch ops (out of the
kernel scope); kept so that `build_method` works on the shipped YAML method blocks."""
import copy

import torch
import torch.nn as nn
import torch.nn.functional as F

AVAILABLE_LOSSES = {}


def register_loss(name):
    def deco(cls):
        AVAILABLE_LOSSES[name] = cls
        return cls
    return deco


def build_loss(config):
    config = copy.deepcopy(config)
    if 'name' not in config:
        raise RuntimeError('loss config needs a `name` key')
    cls = AVAILABLE_LOSSES[config.pop('name')]
    return cls(**cls.from_config(config))


class LossFunction(nn.Module):
    def forward(self, model_bs, x, pred, gt):
        raise NotImplementedError('Must implement forward method')

    @staticmethod
    def from_config(config):
        return config


@register_loss('cross_entropy')
class BatchedCrossEntropy(LossFunction):
    """Every member is scored on the same labels."""

    def __init__(self, reduction='mean'):
        super().__init__()
        self.reduction = reduction

    def forward(self, model_bs, x, pred, gt):
        Y = gt.repeat(model_bs)
        return F.cross_entropy(pred.reshape(Y.shape[0], -1), Y, reduction=self.reduction)


def _entropy_prob(p, eps=1e-8):
    return -(p * torch.log(p + eps)).sum(-1)


@register_loss('mean_softmax_entropy')
class MeanSoftmaxEntropy(LossFunction):
    """Negative entropy of the member-averaged softmax (maximised on OOD points)."""

    def forward(self, model_bs, x, pred, gt):
        return -_entropy_prob(F.softmax(pred, dim=-1).mean(0)).mean()


@register_loss('inlier_outlier_entropy')
class InlierOutlierEntropy(LossFunction):
    """Cross-entropy on the inliers + weight * (negative entropy of the mean softmax on the outliers); takes the forward
    dictionary (`pred_ind`, `pred_ood`) where the other losses take predictions (losses.py:94-113)."""

    def __init__(self, weight=1.0):
        super().__init__()
        self.weight = weight
        self.ce, self.ood_entropy = BatchedCrossEntropy(), MeanSoftmaxEntropy()

    def forward(self, model_bs, x, forward_data, gt):
        return (self.ce(model_bs, x, forward_data['pred_ind'], gt)
                + self.weight * self.ood_entropy(model_bs, x, forward_data['pred_ood'], None))
