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


def unbiased_hsic_all(feats):
    """Unbiased HSIC (Nguyen et al. 2010.15327 eq. 3; `ModelGenerator._uHSIC`, generators/base.py:210-221) between EVERY
    pair of (layer, member) zero-diagonal linear Grams of one batch, from three kernels instead of the reference's Python
    loop of M^2 L^2 small matmuls with three host reads each: the raw Grams (one launch per layer), their pairwise inner
    products tr(K L) = <K, L> (the loss's own pairwise kernel on [L*M, B*B]) and the row sums.
    feats: list of L tensors [M,B,P_l] (fp32, GPU) -> H [L*M, L*M] float64, index l * M + m."""
    from .. import ops
    M, B = feats[0].shape[0], feats[0].shape[1]
    if B < 4:
        raise ValueError('unbiased HSIC needs at least 4 samples')
    G = torch.cat([ops.raw_gram(ops._features_f32(X.reshape(M, B, -1)).contiguous()) for X in feats])   # [L*M,B,B]
    G.diagonal(dim1=1, dim2=2).zero_()
    T = ops.pairwise_inner(G).double()
    r = G.sum(2).double()                    # K 1
    s = r.sum(1)                             # 1^T K 1
    n = float(B)
    return (T + s[:, None] * s[None, :] / ((n - 1) * (n - 2)) - 2.0 * (r @ r.t()) / (n - 2)) / (n * (n - 3))


@torch.no_grad()
def unbiased_cka(hyper, data, num_samples=5, device='cuda', save_path=None, title=None):
    """Layer-by-layer unbiased CKA between the members of a sampled ensemble, averaged over member pairs
    (`ModelGenerator.unbiased_cka`, generators/base.py:224-352): the same members over every batch of `data`, HSIC terms
    accumulated per batch, CKA from the sums.  As in the reference, entry [a, b] pairs layer max(a, b) of member i with
    layer min(a, b) of member j (i < j).  Returns the [L, L] float64 matrix; the figure the reference can save is left
    to the caller (`save_path` is refused)."""
    from ..util.collections import flatten_keys
    if save_path is not None:
        raise NotImplementedError('unbiased_cka: plotting is not part of hyper_amd; plot the returned matrix')
    was_training = hyper.training
    hyper.eval()
    codes = hyper.sample_params(num_samples, device=device)
    num_train = len(data)
    acc = None
    for X, _ in data:
        X = X.to(device)
        feats, pred = hyper.forward(hyper.forward_params(codes), X)
        feats = [f.reshape(f.shape[0], pred.shape[1], -1) for f in flatten_keys(feats).values() if f is not None]
        L, M = len(feats), feats[0].shape[0]
        H = unbiased_hsic_all(feats).view(L, M, L, M) / num_train
        Hd = torch.diagonal(H.reshape(L * M, L * M)).view(L, M)
        li = torch.arange(L, device=H.device)
        hi, lo = torch.maximum(li[:, None], li[None, :]), torch.minimum(li[:, None], li[None, :])    # [L,L]
        kk = Hd[hi].permute(2, 0, 1)[:, None].expand(M, M, L, L)            # [mi, mj, a, b] = Hd[hi[a,b], mi]
        ll = Hd[lo].permute(2, 0, 1)[None].expand(M, M, L, L)               #                  Hd[lo[a,b], mj]
        kl = H[hi[:, :, None, None], torch.arange(M, device=H.device)[None, None, :, None], lo[:, :, None, None],
               torch.arange(M, device=H.device)[None, None, None, :]].permute(2, 3, 0, 1)     # H[hi, mi, lo, mj]
        cur = torch.stack([kk, kl, ll], dim=-1)
        acc = cur.clone() if acc is None else acc + cur
    if was_training:
        hyper.train()
    cka = (acc[..., 1] + 1e-8) / (torch.sqrt(torch.abs(acc[..., 0])) * torch.sqrt(torch.abs(acc[..., 2])) + 1e-8)
    iu = torch.triu_indices(cka.shape[0], cka.shape[1], offset=1, device=cka.device)
    return cka[iu[0], iu[1]].mean(0)


@register_metric('accuracy')
class AccuracyMetric(Metric):
    """Accuracy of the member-averaged softmax, spread of the members' own accuracies and a running accuracy for the
    progress bar (experiments/metrics.py:104-152; the reference counts with torchmetrics and reads one value per member
    back to the host - here the per-member accuracies are one tensor op and one host read).  Note the reference's `log`
    takes no `train_test` argument for this metric; neither does this one."""

    def __init__(self, classes, term_track=True):
        self.classes, self.term_track = classes, term_track
        self.correct = self.total = 0.0

    def reset(self):
        self.correct = self.total = 0.0

    @torch.no_grad()
    def log(self, trainer, model_bs, X, Y, track, *args, **kwargs):
        pred = track['pred_ind']                                         # [M, B, classes]
        member_acc = 100.0 * (pred.argmax(-1) == Y[None]).double().mean(1)
        ens = torch.softmax(pred, dim=-1).mean(0).argmax(-1)
        hits = ens.eq(Y).sum()
        vals = torch.cat([member_acc, hits.double().reshape(1)]).tolist()  # one host read
        res = {'accuracy': 100.0 * vals[-1] / Y.shape[0], 'accuracy_std': float(torch.tensor(vals[:-1], dtype=torch.float64).std(unbiased=False))}
        if self.term_track:
            self.total += float(Y.shape[0])
            self.correct += vals[-1]
            res['term'] = {'acc': res['accuracy'], 'run_acc': 100.0 * self.correct / self.total}
        return res
