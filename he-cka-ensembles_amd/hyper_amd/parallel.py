"""Member-sharded multi-GPU evaluation of the diversity loss (one process per GPU, torch.distributed;
backend "nccl" is RCCL over xGMI on ROCm).

The reference scales with DDP over the SAMPLE batch and evaluates the loss on each rank's sub-batch
(hyper/experiments/training.py:686-695) - that changes the math.  Here the ensemble MEMBERS
are sharded: rank r owns members [r*M/G, (r+1)*M/G) on the FULL batch and the result equals the
single-GPU loss and gradient up to reduction order.

Per layer, one exchange step:
    local   : C_loc, cvec_loc = centred Gram of the local members                      (MFMA)
    exchange: all_gather(C_loc) -> C_all [M,B,B], all_gather(cvec_loc) -> cvec_all     (RCCL, async: overlaps
              with the next layer's Gram)
    local   : S_loc = <C_loc, C_all> [M/G, M];  nsq via a tiny all_gather of its diagonal
              head -> partial loss; all_reduce(sum) of ONE scalar for all layers
    backward: head_bwd on the local rows -> A_loc [M/G, M]; W_loc = combine(A_loc, C_all); dX_loc = W_loc X_loc
No second exchange: row i of the coefficient matrix needs G_ij and G_ji, and G_ji = G_ij because cos is
symmetric and the upstream of a mean/sum reduction is uniform (with detach_right only G_ij is used).

`backend` is the object that provides the five device stages (default: hyper_amd.ops = libhecka).  The
gloo/CPU tests inject an oracle-backed backend to check the sharding logic without a GPU; the product
path never does."""
import torch
import torch.distributed as dist


class _ShardedLayers(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cfg, *feats):
        backend, group, specs, weights, detach, reduce = cfg
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        Ml = feats[0].shape[0]
        M, r0 = Ml * world, rank * Ml
        grams, works = [], []
        feats = [X.contiguous() for X in feats]
        # every layer's local Gram in one launch when the backend can (a rank's M/G members leave a per-layer launch
        # far below one workgroup per GPU slot); otherwise layer by layer, the exchanges running behind them
        multi = getattr(backend, 'multi_gram_eligible', None)
        pre = backend.centered_gram_multi(feats) if multi is not None and multi(feats) else None
        for li, X in enumerate(feats):
            C, cvec, xmax = pre[li] if pre is not None else backend.centered_gram(X, True)   # xmax: max |X| (dX scaling)
            C_all = torch.empty((M,) + tuple(C.shape[1:]), device=C.device, dtype=C.dtype)
            cv_all = torch.empty((M, cvec.shape[1]), device=C.device, dtype=C.dtype)
            works.append((dist.all_gather_into_tensor(C_all, C, group=group, async_op=True),
                          dist.all_gather_into_tensor(cv_all, cvec, group=group, async_op=True)))
            grams.append((X, C, C_all, cv_all, xmax))
        total = None
        saved = []
        for (X, C, C_all, cv_all, xmax), (w1, w2), spec, wl in zip(grams, works, specs, weights):
            w1.wait()
            w2.wait()
            S = backend.pairwise_inner(C, C_all)                        # [Ml, M]
            nsq_loc = torch.diagonal(S, offset=r0).contiguous()          # S[i, r0+i]
            nsq = torch.empty((M,), device=S.device, dtype=S.dtype)
            dist.all_gather_into_tensor(nsq, nsq_loc, group=group)
            count = (M * (M - 1) if spec.remove_diag else M * M) if reduce == 'mean' else 1
            scale = float(wl) / count
            _, part = backend.head_forward(S, nsq, r0, spec, scale)
            total = part if total is None else total + part
            saved.append((X, C_all, cv_all, S, nsq, spec, scale, xmax))
        total = total.clone()
        dist.all_reduce(total, op=dist.ReduceOp.SUM, group=group)
        ctx.saved, ctx.backend, ctx.r0, ctx.detach = saved, backend, r0, detach
        return total

    @staticmethod
    def backward(ctx, dloss):
        dloss = dloss.contiguous().float()
        grads = []
        for X, C_all, cv_all, S, nsq, spec, scale, xmax in ctx.saved:
            A = ctx.backend.head_backward(S, nsq, ctx.r0, spec, dloss, scale)
            grads.append(ctx.backend.gram_backward(A, C_all, cv_all, X, ctx.detach, xmax))
        return (None,) + tuple(grads)


def sharded_layers_loss(feats, specs, weights, detach_diag=True, reduce='mean', group=None, backend=None):
    """Global `model_layers` loss from this rank's member slice of every layer's features.
    feats: list of [M/G, B, P_l]; specs: one ops.HeadSpec per layer; weights: per-layer floats."""
    if backend is None:
        from . import ops as backend
    if reduce not in ('mean', 'sum'):
        raise ValueError("the sharded loss supports reduction 'mean' or 'sum'")
    return _ShardedLayers.apply((backend, group, list(specs), list(weights), bool(detach_diag), reduce), *feats)


class ShardedModelLayersKernel(torch.nn.Module):
    """Wraps a hyper_amd ModelLayersKernel for member-sharded evaluation: same forward signature, but
    `feats` holds only this rank's members and the returned scalar is the loss over ALL members."""

    def __init__(self, kernel, group=None, backend=None):
        super().__init__()
        self.kernel, self.group, self.backend = kernel, group, backend

    def forward(self, params, feats, pred, track):
        k = self.kernel
        if len(feats) == 0:
            return None
        k.layer_weighting.reset_total()
        k.layer_weighting.save_total(len(feats))
        kerns = list(k.layer_kernel) if k.different_kernels else [k.layer_kernel] * len(feats)
        specs = [kk._spec() for kk in kerns]
        if any(s is None for s in specs):
            raise NotImplementedError('member sharding needs linear, centred, non-learnable he/cka layer kernels')
        detach = {kk.feature_kernel.detach_diag for kk in kerns}
        if len(detach) != 1:
            raise NotImplementedError('member sharding needs one detach_diag setting for all layers')
        weights = [float(k.layer_weighting.get_weight(i, len(feats))) for i in range(len(feats))]
        xs = [kerns[i].feature_kernel.flatten(f) for i, f in enumerate(feats.values())]
        return sharded_layers_loss(xs, specs, weights, detach.pop(), k.reduction, self.group, self.backend)
