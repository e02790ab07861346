"""Member-sharded multi-GPU evaluation of the diversity loss (one process per GPU, torch.distributed;
backend "nccl" is RCCL over xGMI on ROCm).

The reference scales with DDP over the SAMPLE batch and evaluates the loss on each rank's sub-batch
(hyper/experiments/training.py:686-695) - that changes the math.  Here the ensemble MEMBERS
are sharded: rank r owns members [r*M/G, (r+1)*M/G) on the FULL batch and the result equals the
single-GPU loss and gradient up to reduction order.

Two exchange schemes (HECKA_SHARD_MODE = slice (default) | gather):

slice - every stage's work AND the exchanged bytes per rank fall as 1/G.  After the local Grams, an
    all-to-all re-partitions the [M, B*B] matrix of flattened Grams from member rows to COLUMN slices:
    local   : C_loc [M/G, B*B], cvec_loc = centred Grams of the local members              (MFMA, 1/G of the flops)
    exchange: all_to_all -> Cs [M, B*B/G] (this rank's slice of every member's Gram);      (M/G * B*B * 4 bytes out)
              one all_gather of every layer's cvec (M*B floats per layer)
    local   : S_part = Cs Cs^T [M, M] over the slice; ONE all_reduce of all layers' S_part   (L*M*M floats)
              head on the full S (M*M elements, every rank the same: no exchange of the loss)
    backward: head_bwd -> A [M, M] (every rank), gamma = A cvec; Ws = combine(A, Cs) on the slice   (1/G of HBM traffic)
              all_to_all back -> W_loc [M/G, B*B]; dX_loc = W_loc X_loc                     (MFMA, 1/G of the flops)
    The all-to-all of layer l runs behind the stages of layer l+1 (async work handles).
gather - the first scheme: all_gather the local Grams to C_all [M, B, B] on every rank, S rows and W rows for the
    local members against all of C_all.  One exchange, but M*B*B*4 bytes arrive at every rank whatever G is, and the
    pairwise and combine stages do not shrink below reading all of C_all; kept for backends without the slice stages.

`backend` is the object that provides the device stages (default: hyper_amd.ops = libhecka).  The gloo/CPU
tests inject an oracle-backed backend to check the sharding logic without a GPU; the product path never
does.  With the gloo backend and GPU tensors (a rehearsal of several ranks on one card) the collectives are
staged through the host."""
import os

import torch
import torch.distributed as dist


EXCHANGED_BYTES = 0      # bytes this rank has SENT through the collectives below (bench.py reports it per evaluation)


def _sent(nbytes):
    global EXCHANGED_BYTES
    EXCHANGED_BYTES += int(nbytes)


class _Done:
    def wait(self):
        return True


def _staged(group, *ts):
    return dist.get_backend(group) == 'gloo' and any(t.is_cuda for t in ts)


def _all_to_all(out, inp, out_splits, in_splits, group):
    world = dist.get_world_size(group)
    _sent(inp.numel() * inp.element_size() * (world - 1) // world)
    if _staged(group, out, inp):
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(o, inp.cpu(), out_splits, in_splits, group=group)
        out.copy_(o)
        return _Done()
    return dist.all_to_all_single(out, inp, out_splits, in_splits, group=group, async_op=True)


def _all_gather(out, inp, group):
    _sent(inp.numel() * inp.element_size() * (dist.get_world_size(group) - 1))
    if _staged(group, out, inp):
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(o, inp.cpu(), group=group)
        out.copy_(o)
        return _Done()
    return dist.all_gather_into_tensor(out, inp, group=group, async_op=True)


def _all_reduce(t, group):
    world = dist.get_world_size(group)
    _sent(2 * t.numel() * t.element_size() * (world - 1) // world)      # ring: reduce-scatter + all-gather
    if _staged(group, t):
        o = t.cpu()
        dist.all_reduce(o, op=dist.ReduceOp.SUM, group=group)
        t.copy_(o)
        return
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)


def slice_bounds(n, world, granule=1):
    """world+1 boundaries of the column slices of a row of n Gram elements: multiples of `granule` when n is
    (the streaming combine kernel works on 64-column chunks), as even as that allows."""
    q = granule if granule > 0 and n % granule == 0 else 1
    units = n // q
    return [((g * units) // world) * q for g in range(world + 1)]


class _SlicedLayers(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cfg, *feats):
        backend, group, specs, weights, detach, reduce = cfg
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        Ml, B = feats[0].shape[0], feats[0].shape[1]
        M, r0, N2 = Ml * world, rank * Ml, B * B
        bounds = slice_bounds(N2, world, getattr(backend, 'SLICE_GRANULE', 1))
        ns = [bounds[g + 1] - bounds[g] for g in range(world)]
        n_r = ns[rank]
        even = all(n == n_r for n in ns)
        feats = [X.contiguous() for X in feats]
        multi = getattr(backend, 'multi_gram_eligible', None)
        multi = multi is not None and multi(feats)
        # slice-major Grams: when a slice is whole rows (a power of two of them) the Gram epilogue writes C straight in
        # the all-to-all's send order [G, Ml, rows, B] and the dX GEMM reads W in the order it arrives - no repacking
        rows = n_r // B if (even and n_r % B == 0) else 0
        sliced = (world > 1 and multi and rows > 0 and rows & (rows - 1) == 0 and
                  feats[0].dtype in getattr(backend, 'SLICED_DTYPES', (torch.float32,)) and
                  hasattr(backend, 'gram_bwd_multi') and os.environ.get('HECKA_SHARD_REPACK', '0') != '1')
        pre = (backend.centered_gram_multi(feats, slice_rows=rows) if sliced else backend.centered_gram_multi(feats)) if multi else None
        slices, cvecs, xmaxs = [], [], []
        # every layer's column slice in ONE buffer: the batched pairwise stage reads it whole
        Cs_all = None
        for li, X in enumerate(feats):
            if X.shape[0] != Ml or X.shape[1] != B:
                raise ValueError('every layer needs the same members and samples')
            C, cvec, xmax = pre[li] if pre is not None else backend.centered_gram(X, True)
            C2 = C.reshape(Ml, N2)
            if world == 1:
                slices.append((C2, _Done()))
            else:
                if sliced:
                    send = C2.view(-1)
                elif even:      # [Ml, G, n] -> [G, Ml, n] in one strided copy
                    send = C2.view(Ml, world, n_r).permute(1, 0, 2).contiguous().view(-1)
                else:
                    send = torch.cat([C2[:, bounds[g]:bounds[g + 1]].reshape(-1) for g in range(world)])
                if Cs_all is None:
                    Cs_all = torch.empty((len(feats), M, n_r), device=C.device, dtype=C.dtype)
                Cs = Cs_all[li]
                slices.append((Cs, _all_to_all(Cs.view(-1), send, [Ml * n_r] * world, [Ml * n for n in ns], group)))
            cvecs.append(cvec)
            xmaxs.append(xmax)
        L = len(feats)
        cv_loc = torch.stack(cvecs)                                              # [L, Ml, B]
        if world == 1:
            cv_all = cv_loc
        else:
            raw = torch.empty((world, L, Ml, B), device=cv_loc.device, dtype=cv_loc.dtype)
            _all_gather(raw.view(-1), cv_loc.view(-1), group).wait()
            cv_all = raw.permute(1, 0, 2, 3).reshape(L, M, B).contiguous()
        for _, work in slices:
            work.wait()
        elig = getattr(backend, 'pairwise_self_batched_eligible', None)
        if world > 1 and elig is not None and elig(M, n_r):
            S_all = backend.pairwise_self_batched(Cs_all)                        # one launch for all layers
        else:
            S_all = torch.stack([backend.pairwise_inner(Cs) if n_r else
                                 torch.zeros((M, M), device=Cs.device, dtype=Cs.dtype) for Cs, _ in slices])   # [L, M, M]
        if world > 1:
            _all_reduce(S_all, group)
        total, saved, scales = None, [], []
        for spec, wl in zip(specs, weights):
            count = (M * (M - 1) if spec.remove_diag else M * M) if reduce == 'mean' else 1
            scales.append(float(wl) / count if count else float('nan'))
        batched = len(specs) <= 16 and all(hasattr(backend, f) for f in
                                           ('head_forward_batched', 'head_backward_batched', 'coef_gamma_batched'))
        nsq_all = torch.diagonal(S_all, dim1=1, dim2=2).contiguous()
        if batched:                 # one head launch for all layers
            total = backend.head_forward_batched(S_all, nsq_all, 0, specs, scales).sum()
        for li, (X, spec) in enumerate(zip(feats, specs)):
            S, nsq = S_all[li], nsq_all[li]
            if not batched:
                _, part = backend.head_forward(S, nsq, 0, spec, scales[li])
                total = part if total is None else total + part
            saved.append((X, slices[li][0], cv_all[li], S, nsq, spec, scales[li], xmaxs[li]))
        ctx.batch = (S_all, nsq_all, cv_all, specs, scales) if batched else None
        ctx.saved, ctx.backend, ctx.group, ctx.detach = saved, backend, group, detach
        ctx.geom = (world, rank, Ml, B, bounds, ns)
        ctx.sliced = rows if sliced else 0
        ctx.Cs_all = Cs_all
        ctx.gate = getattr(backend, 'current_gate', lambda: None)()     # ops.grad_gate: a masked NaN term gives dX = 0
        ctx.mode = getattr(backend, 'get_gemm_mode', lambda: None)()
        return total.clone()

    @staticmethod
    def backward(ctx, dloss):
        import contextlib
        with getattr(ctx.backend, 'gemm_mode', lambda m: contextlib.nullcontext())(ctx.mode):   # forward's GEMM mode (another thread)
            return _SlicedLayers._backward(ctx, dloss)

    @staticmethod
    def _backward(ctx, dloss):
        be, group = ctx.backend, ctx.group
        world, rank, Ml, B, bounds, ns = ctx.geom
        r0, n_r, N2 = rank * Ml, ns[rank], B * B
        dloss = dloss.contiguous().float() if dloss.is_cuda else dloss.contiguous()
        pending = []
        Ws_all = wmax_all = None
        if ctx.batch is not None:   # one head_bwd and one gamma launch for all layers
            S_all, nsq_all, cv_stack, specs, scales = ctx.batch
            A_all = be.head_backward_batched(S_all, nsq_all, 0, specs, dloss, scales)
            gamma_all = be.coef_gamma_batched(A_all, cv_stack)
            if world > 1 and ctx.Cs_all is not None and hasattr(be, 'combine_slice_batched'):   # ... and one combine launch
                Ws_all = be.combine_slice_batched(A_all, ctx.Cs_all, gamma_all, B, bounds[rank], ctx.detach)
            if hasattr(be, 'wbound_batched') and all(s[7] is not None for s in ctx.saved):
                wmax_all = be.wbound_batched(A_all[:, r0:r0 + Ml].contiguous(), nsq_all, gamma_all[:, r0:r0 + Ml].contiguous())
        # exchanges are issued deepest layer first: its W arrives first and its dX GEMM (half the flops at the bench's
        # widths) runs while the other layers' slices are still on the links
        order = sorted(range(len(ctx.saved)), key=lambda i: -ctx.saved[i][0].shape[2])
        for li in order:
            X, Cs, cv_all, S, nsq, spec, scale, xmax = ctx.saved[li]
            if ctx.batch is not None:
                A, gamma = A_all[li], gamma_all[li]
            else:
                A = be.head_backward(S, nsq, 0, spec, dloss, scale)              # [M, M], the same on every rank
                gamma = be.coef_gamma(A, cv_all)                                 # [M, B]
            Ws = Ws_all[li] if Ws_all is not None else be.combine_slice(A, Cs, gamma, B, bounds[rank], ctx.detach)   # [M, n_r]
            if world == 1:
                recv, work = Ws, _Done()
            else:
                recv = torch.empty((Ml * N2,), device=Ws.device, dtype=Ws.dtype)
                work = _all_to_all(recv, Ws.view(-1), [Ml * n for n in ns], [Ml * n_r] * world, group)
            wmax = None
            if wmax_all is not None:
                wmax = wmax_all[li]
            elif xmax is not None and hasattr(be, 'wbound'):
                wmax = be.wbound(A[r0:r0 + Ml].contiguous(), nsq, gamma[r0:r0 + Ml].contiguous())
            pending.append((X, recv, work, wmax, xmax))
        gate = (ctx.gate,) if ctx.gate is not None else ()

        def in_layer_order(grads):      # pending (and grads) are in `order`; autograd wants the layers' own order
            out = [None] * len(grads)
            for pos, li in enumerate(order):
                out[li] = grads[pos]
            return (None,) + tuple(out)
        if ctx.sliced and be.gram_bwd_multi_eligible([p[1] for p in pending], [p[0] for p in pending], [p[3] for p in pending],
                                                      [p[4] for p in pending], ctx.sliced):
            # W stays as it arrived, [G, Ml, rows, B]: the dX launches read it in that order.  Two launches when there is
            # an exchange to hide: the deepest layer as soon as its slices are in, the rest in one launch behind it
            cut = 1 if (world > 1 and len(pending) >= 3 and os.environ.get('HECKA_SHARD_BWD_SPLIT', '1') != '0') else len(pending)
            grads = []
            for part in (pending[:cut], pending[cut:]):
                if not part:
                    continue
                for p in part:
                    p[2].wait()
                grads += be.gram_bwd_multi([p[1] for p in part], [p[0] for p in part], [p[3] for p in part],
                                           [p[4] for p in part], *gate, slice_rows=ctx.sliced)      # (one layer is a valid launch)
            return in_layer_order(grads)
        Wl = []
        for X, recv, work, wmax, xmax in pending:
            work.wait()
            if world == 1:
                W = recv
            elif all(n == n_r for n in ns):      # [G, Ml, n] -> [Ml, G, n] in one strided copy
                W = recv.view(world, Ml, n_r).permute(1, 0, 2).contiguous()
            else:
                W = torch.empty((Ml, N2), device=recv.device, dtype=recv.dtype)
                off = 0
                for g in range(world):
                    W[:, bounds[g]:bounds[g + 1]] = recv[off:off + Ml * ns[g]].view(Ml, ns[g])
                    off += Ml * ns[g]
            Wl.append(W.view(Ml, B, B))
        multi = getattr(be, 'gram_bwd_multi_eligible', None)
        if multi is not None and multi(Wl, [p[0] for p in pending], [p[3] for p in pending], [p[4] for p in pending]):
            return in_layer_order(be.gram_bwd_multi(Wl, [p[0] for p in pending], [p[3] for p in pending],
                                                    [p[4] for p in pending], *gate))
        return in_layer_order([be.gram_bwd_w(W, X, wmax, xmax, *gate) for W, (X, _, _, wmax, xmax) in zip(Wl, pending)])


class _GatheredLayers(torch.autograd.Function):
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
            works.append((_all_gather(C_all, C, group), _all_gather(cv_all, cvec, group)))
            grams.append((X, C, C_all, cv_all, xmax))
        total = None
        saved = []
        for (X, C, C_all, cv_all, xmax), (w1, w2), spec, wl in zip(grams, works, specs, weights):
            w1.wait()
            w2.wait()
            S = backend.pairwise_inner(C, C_all)                        # [Ml, M]
            nsq_loc = torch.diagonal(S, offset=r0).contiguous()          # S[i, r0+i]
            nsq = torch.empty((M,), device=S.device, dtype=S.dtype)
            _all_gather(nsq, nsq_loc, group).wait()
            count = (M * (M - 1) if spec.remove_diag else M * M) if reduce == 'mean' else 1
            scale = float(wl) / count if count else float('nan')
            _, part = backend.head_forward(S, nsq, r0, spec, scale)
            total = part if total is None else total + part
            saved.append((X, C_all, cv_all, S, nsq, spec, scale, xmax))
        total = total.clone()
        _all_reduce(total, group)
        ctx.saved, ctx.backend, ctx.r0, ctx.detach = saved, backend, r0, detach
        ctx.gate = getattr(backend, 'current_gate', lambda: None)()
        ctx.mode = getattr(backend, 'get_gemm_mode', lambda: None)()
        return total

    @staticmethod
    def backward(ctx, dloss):
        import contextlib
        with getattr(ctx.backend, 'gemm_mode', lambda m: contextlib.nullcontext())(ctx.mode):
            return _GatheredLayers._backward(ctx, dloss)

    @staticmethod
    def _backward(ctx, dloss):
        dloss = dloss.contiguous().float() if dloss.is_cuda else dloss.contiguous()
        grads = []
        for X, C_all, cv_all, S, nsq, spec, scale, xmax in ctx.saved:
            A = ctx.backend.head_backward(S, nsq, ctx.r0, spec, dloss, scale)
            gate = {'gate': ctx.gate} if ctx.gate is not None else {}
            grads.append(ctx.backend.gram_backward(A, C_all, cv_all, X, ctx.detach, xmax, **gate))
        return (None,) + tuple(grads)


def shard_mode(backend):
    mode = os.environ.get('HECKA_SHARD_MODE', 'slice')
    if mode not in ('slice', 'gather'):
        raise ValueError("HECKA_SHARD_MODE must be 'slice' or 'gather'")
    if mode == 'slice' and not all(hasattr(backend, f) for f in ('coef_gamma', 'combine_slice', 'gram_bwd_w')):
        return 'gather'
    return mode


def sharded_layers_loss(feats, specs, weights, detach_diag=True, reduce='mean', group=None, backend=None):
    """Global `model_layers` loss from this rank's member slice of every layer's features.
    feats: list of [M/G, B, P_l]; specs: one ops.HeadSpec per layer; weights: per-layer floats."""
    if backend is None:
        from . import ops as backend
    if reduce not in ('mean', 'sum'):
        raise ValueError("the sharded loss supports reduction 'mean' or 'sum'")
    if hasattr(backend, '_features'):      # 16-bit features: native where the kernels take them, else the exact up-cast
        feats = [backend._features(X) for X in feats]
        if len({X.dtype for X in feats}) > 1:
            feats = [backend._features_f32(X) for X in feats]
    fn = _SlicedLayers if shard_mode(backend) == 'slice' else _GatheredLayers
    return fn.apply((backend, group, list(specs), list(weights), bool(detach_diag), reduce), *feats)


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


class _ScaledLoss(torch.nn.Module):
    def __init__(self, loss, factor):
        super().__init__()
        self.loss, self.factor = loss, factor

    def forward(self, *args, **kwargs):
        return self.factor * self.loss(*args, **kwargs)


def shard_method(method, group=None, backend=None):
    """Member-sharded TRAINING (last sentence): every rank builds the ensemble with ITS members only
    (`ensemble_size` = `num` = M / G, all on the full batch) and calls `shard_method` once on its `repulsive_kernel` /
    `mle` method.  After that rank r's loss is  beta * CE(own members) / G + gamma * rep(ALL members)  and its backward
    leaves on the local parameters exactly the gradients the unsharded step gives those members: the repulsion term is
    the sharded kernel (global value, gradients for the local members), the per-member means of the data term and of
    the L2 term are rescaled from M / G members to M.  No gradient all-reduce exists or is needed - members share no
    weights - so the only exchange is the Gram column slices inside the kernel.  Unlike the reference's DDP (per-rank
    sub-batch Grams) the loss is the single-GPU loss.  Member-coupled OOD losses (`mean_softmax_entropy`
    averages the members' softmax) are refused."""
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if getattr(method, 'ood_loss', None) is not None:
        raise NotImplementedError('shard_method: the OOD prediction loss averages over all members; not sharded')
    for name in ('model_kernel', 'model_kernel_ood'):
        kern = getattr(method, name, None)
        if kern is not None and not isinstance(kern, ShardedModelLayersKernel):
            setattr(method, name, ShardedModelLayersKernel(kern, group, backend))
    if method.ind_loss is not None and not isinstance(method.ind_loss, _ScaledLoss):
        method.ind_loss = _ScaledLoss(method.ind_loss, 1.0 / world)
        if method.l2_reg is not None:
            method.l2_reg = method.l2_reg / world
    return method
