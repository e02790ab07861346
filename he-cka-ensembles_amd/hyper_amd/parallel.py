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
import torch
import torch.distributed as dist

from .config import knobs


EXCHANGED_BYTES = 0      # bytes this rank has SENT through the collectives below (bench.py reports it per evaluation)
COLLECTIVES = 0          # collective launches this rank has issued (one batched exchange = one RCCL group = one launch)
PROFILE = None           # bench.py: a list -> (event before, event after) around every wait on the compute stream


def _sent(nbytes, launches=1):
    global EXCHANGED_BYTES, COLLECTIVES
    EXCHANGED_BYTES += int(nbytes)
    COLLECTIVES += int(launches)


class _Done:
    def wait(self):
        return True


class _Works:
    """Several work handles waited for as one."""

    def __init__(self, works, keep=None, after=None):
        self.works, self.keep, self.after = list(works), keep, after

    def wait(self):
        prof = PROFILE is not None and torch.cuda.is_available()
        if prof:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        for w in self.works:
            w.wait()
        self.works = []
        if self.after is not None:
            self.after()
            self.after = None
        self.keep = None
        if prof:
            e1.record()
            PROFILE.append((e0, e1))
        return True


def _staged(group, *ts):
    return dist.get_backend(group) == 'gloo' and any(t.is_cuda for t in ts)


def exchange_mode():
    """How the layers' segments travel when the blocked layouts (ONE all_to_all_single per exchange, _SlicedLayers) do not
    apply.  HECKA_SHARD_EXCHANGE = a2a (default): one all_to_all_single per layer.  p2p: every layer's segments of an
    exchange in ONE batched send / receive group (torch.distributed.batch_isend_irecv: on RCCL one ncclGroup, i.e. one
    device launch whatever the number of layers, but 2 L (G - 1) Python-level operations on the host)."""
    return knobs.shard_exchange


def _exchange(sends, recvs, send_splits, recv_splits, group):
    """One exchange for a list of layers.  sends[l] / recvs[l]: flat contiguous buffers holding one segment per rank, in rank
    order, of send_splits[g] / recv_splits[g] elements; segment g of sends[l] goes to rank g, segment s of recvs[l] comes
    from rank s.  Returns a handle whose wait() orders the current stream behind the arrival of every segment."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    so = [0]
    for n in send_splits:
        so.append(so[-1] + n)
    ro = [0]
    for n in recv_splits:
        ro.append(ro[-1] + n)
    esz = sends[0].element_size()
    sent = sum(so[-1] - send_splits[rank] for _ in sends) * esz
    if exchange_mode() == 'a2a':
        _sent(sent, len(sends))
        works = []
        for snd, rcv in zip(sends, recvs):
            if _staged(group, snd, rcv):
                o = torch.empty(rcv.shape, dtype=rcv.dtype)
                dist.all_to_all_single(o, snd.cpu(), list(recv_splits), list(send_splits), group=group)
                rcv.copy_(o)
            else:
                works.append(dist.all_to_all_single(rcv, snd, list(recv_splits), list(send_splits), group=group, async_op=True))
        return _Works(works, keep=(sends, recvs))
    _sent(sent, 1)
    staged = _staged(group, *sends, *recvs)
    ops, host = [], []
    peer = (lambda g: g) if group is None else (lambda g: dist.get_global_rank(group, g))
    for li, (snd, rcv) in enumerate(zip(sends, recvs)):
        # this rank's own segment never leaves the device
        if send_splits[rank]:
            rcv[ro[rank]:ro[rank + 1]].copy_(snd[so[rank]:so[rank + 1]])
        for g in range(world):
            if g == rank:
                continue
            if send_splits[g]:
                t = snd[so[g]:so[g + 1]]
                ops.append(dist.P2POp(dist.isend, t.cpu() if staged else t, peer(g), group, tag=li))
            if recv_splits[g]:
                t = rcv[ro[g]:ro[g + 1]]
                if staged:
                    h = torch.empty(t.shape, dtype=t.dtype)
                    host.append((t, h))
                    t = h
                ops.append(dist.P2POp(dist.irecv, t, peer(g), group, tag=li))
    works = dist.batch_isend_irecv(ops) if ops else []
    after = None
    if staged:
        def after():
            for dst, h in host:
                dst.copy_(h)
    return _Works(works, keep=(sends, recvs, ops), after=after)


def _all_gather(out, inp, group):
    _sent(inp.numel() * inp.element_size() * (dist.get_world_size(group) - 1))
    if _staged(group, out, inp):
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(o, inp.cpu(), group=group)
        out.copy_(o)
        return _Done()
    return _Works([dist.all_gather_into_tensor(out, inp, group=group, async_op=True)], keep=(out, inp))


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


def _a2a_equal(out, inp, group):
    """ONE all_to_all_single with equal splits (the blocked layouts: every layer of an exchange in one buffer)."""
    world = dist.get_world_size(group)
    _sent(inp.numel() * inp.element_size() * (world - 1) // world, 1)
    if _staged(group, out, inp):
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(o, inp.cpu(), group=group)
        out.copy_(o)
        return _Done()
    return _Works([dist.all_to_all_single(out, inp, group=group, async_op=True)], keep=(out, inp))


class _SlicedLayers(torch.autograd.Function):
    """Collectives per evaluation, whatever the number of layers (world > 1): forward ONE exchange (every layer's Gram
    rows -> column slices) + ONE all-gather (every layer's partial inner products and centring vectors, summed locally in
    rank order: bit-identical S on every rank); backward TWO exchanges (column slices of W -> rows: the deepest layer,
    whose dX GEMM then runs while the other layers' slices are on the links, then the rest).

    Blocked layouts (the bench's shapes: 64 members, slices of whole Gram rows; `backend.blocked_eligible`): each exchange
    is ONE all_to_all_single.  The Gram epilogue writes all layers into one buffer [G][L][Ml][rows][B] - the send order - the
    received [G][L][Ml][n] is read as it lies by the pairwise and combine kernels (member rows in blocks), combine writes
    [G][L'][Ml][n] for each backward group and the dX GEMM reads the received [G][L'][Ml][rows][B]: no repacking anywhere.
    Other shapes: the layers' segments travel per layer (`_exchange`) with repacking copies where a slice is not whole rows."""

    @staticmethod
    def forward(ctx, cfg, *feats):
        backend, group, specs, weights, detach, reduce, gspecs = cfg
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        Ml, B = feats[0].shape[0], feats[0].shape[1]
        M, r0, N2 = Ml * world, rank * Ml, B * B
        L = len(feats)
        # everything below runs deepest layer first (the order of the all-layers launches and of the backward exchanges)
        order = sorted(range(L), key=lambda i: -feats[i].shape[2])
        feats = [feats[i].contiguous() for i in order]
        specs, weights = [specs[i] for i in order], [weights[i] for i in order]
        gspecs = [gspecs[i] for i in order] if gspecs is not None else None
        bounds = slice_bounds(N2, world, getattr(backend, 'SLICE_GRANULE', 1))
        ns = [bounds[g + 1] - bounds[g] for g in range(world)]
        n_r = ns[rank]
        even = all(n == n_r for n in ns)
        for X in feats:
            if X.shape[0] != Ml or X.shape[1] != B:
                raise ValueError('every layer needs the same members and samples')
        linear = gspecs is None
        multi = getattr(backend, 'multi_gram_eligible', None)
        multi = linear and multi is not None and multi(feats)
        # slice-major Grams: when a slice is whole rows (a power of two of them) the Gram epilogue writes C straight in
        # the exchange's send order [G, Ml, rows, B] and the dX GEMM reads W in the order it arrives - no repacking
        rows = n_r // B if (even and n_r % B == 0) else 0
        sliced = (world > 1 and multi and rows > 0 and rows & (rows - 1) == 0 and
                  feats[0].dtype in getattr(backend, 'SLICED_DTYPES', (torch.float32,)) and
                  hasattr(backend, 'gram_bwd_multi'))
        blocked = (sliced and hasattr(backend, 'blocked_eligible') and backend.blocked_eligible(M, n_r, Ml) and
                   knobs.shard_blocked)
        Cbuf = None
        if blocked:
            pre, Cbuf = backend.centered_gram_multi(feats, slice_rows=rows, one_buffer=True)      # [G, L, Ml, rows, B]
        elif multi:
            pre = backend.centered_gram_multi(feats, slice_rows=rows) if sliced else backend.centered_gram_multi(feats)
        else:
            pre = None
        locs, fulls, cvecs, xmaxs, aux = [], [], [], [], []
        for li, X in enumerate(feats):
            if linear:
                C, cvec, xmax = pre[li] if pre is not None else backend.centered_gram(X, True)
                aux.append(None)
            else:   # cossim / rbf: the row-wise transform of the raw Gram is local to a member; its row means play cvec's part
                C, cvec, gdiag, hband = backend.transformed_gram(X, gspecs[li], True)
                xmax = None
                aux.append((C, cvec, gdiag, hband))
            if not blocked:
                locs.append(C.reshape(Ml, N2) if not sliced else C)
            fulls.append(C)
            cvecs.append(cvec)
            xmaxs.append(xmax)
        cv_loc = getattr(backend, '_stacked', torch.stack)(cvecs)               # [L, Ml, B] (no copy after an all-layers launch)
        if blocked:
            # ONE all_to_all_single for all layers; the received [G][L][Ml][n_r] is read as it lies
            Cs_all = torch.empty((world, L, Ml, n_r), device=Cbuf.device, dtype=Cbuf.dtype)
            _a2a_equal(Cs_all.view(-1), Cbuf.view(-1), group).wait()
            S_all = backend.pairwise_self_blocked(Cs_all)
        else:
            # every layer's column slice in ONE buffer [L, M, n_r]: the batched pairwise / combine stages read it whole
            if world == 1:      # (the Grams of one all-layers launch already are one [L, M, B, B] buffer: no copy)
                Cs_all, work = getattr(backend, '_stacked', torch.stack)(fulls).view(L, M, N2), _Done()
            else:
                Cs_all = torch.empty((L, M, n_r), device=locs[0].device, dtype=locs[0].dtype)
                sends = []
                for C2 in locs:
                    if sliced:
                        sends.append(C2.reshape(-1))
                    elif even:      # [Ml, G, n] -> [G, Ml, n] in one strided copy
                        sends.append(C2.view(Ml, world, n_r).permute(1, 0, 2).contiguous().view(-1))
                    else:
                        sends.append(torch.cat([C2[:, bounds[g]:bounds[g + 1]].reshape(-1) for g in range(world)]))
                work = _exchange(sends, [Cs_all[li].view(-1) for li in range(L)], [Ml * n for n in ns], [Ml * n_r] * world, group)
            work.wait()
            elig = getattr(backend, 'pairwise_self_batched_eligible', None)
            if n_r and elig is not None and elig(M, n_r):
                S_all = backend.pairwise_self_batched(Cs_all)                        # one launch for all layers
            else:
                S_all = torch.stack([backend.pairwise_inner(Cs_all[li]) if n_r else
                                     torch.zeros((M, M), device=Cs_all.device, dtype=Cs_all.dtype) for li in range(L)])   # [L, M, M]
        if world == 1:
            cv_all = cv_loc
        else:
            # ONE all-gather carries every rank's partial S and its members' centring vectors; the partial sums are
            # added in rank order on every rank, so S (and everything derived from it) is bit-identical everywhere
            nS, nc = L * M * M, L * Ml * B
            pack = torch.cat([S_all.reshape(-1), cv_loc.reshape(-1).to(S_all.dtype)])
            raw = torch.empty((world, nS + nc), device=pack.device, dtype=pack.dtype)
            _all_gather(raw.view(-1), pack, group).wait()
            S_all = torch.sum(raw[:, :nS], dim=0).view(L, M, M)      # (one kernel; the same reduction order on every rank)
            cv_all = raw[:, nS:].reshape(world, L, Ml, B).permute(1, 0, 2, 3).reshape(L, M, B).contiguous().to(cv_loc.dtype)
        scales = []
        for spec, wl in zip(specs, weights):
            count = (M * (M - 1) if spec.remove_diag else M * M) if reduce == 'mean' else 1
            scales.append(float(wl) / count if count else float('nan'))
        batched = len(specs) <= 16 and all(hasattr(backend, f) for f in
                                           ('head_forward_batched', 'head_backward_batched', 'coef_gamma_batched'))
        nsq_all = torch.diagonal(S_all, dim1=1, dim2=2).contiguous()
        total = None
        if batched:                 # one head launch for all layers
            total = backend.head_forward_batched(S_all, nsq_all, 0, specs, scales).sum()
        else:
            for li, spec in enumerate(specs):
                _, part = backend.head_forward(S_all[li], nsq_all[li], 0, spec, scales[li])
                total = part if total is None else total + part
        ctx.feats, ctx.xmaxs, ctx.aux, ctx.order = feats, xmaxs, aux, order
        ctx.state = (Cs_all, S_all, nsq_all, cv_all, specs, scales, batched)
        ctx.backend, ctx.group, ctx.detach, ctx.gspecs = backend, group, detach, gspecs
        ctx.geom = (world, rank, Ml, B, bounds, ns)
        ctx.sliced, ctx.blocked = (rows if sliced else 0), blocked
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
        Cs_all, S_all, nsq_all, cv_all, specs, scales, batched = ctx.state
        feats, xmaxs, linear = ctx.feats, ctx.xmaxs, ctx.gspecs is None
        L, M = len(feats), Ml * world
        even = all(n == n_r for n in ns)
        dloss = dloss.contiguous().float() if dloss.is_cuda else dloss.contiguous()
        detach = ctx.detach if linear else True      # the diagonal of a cossim / rbf Gram carries no gradient
        # the [M, M]-sized stages for all layers (the same on every rank)
        if batched:
            A_all = be.head_backward_batched(S_all, nsq_all, 0, specs, dloss, scales)
            gamma_all = be.coef_gamma_batched(A_all, cv_all)
        else:
            A_all = torch.stack([be.head_backward(S_all[li], nsq_all[li], 0, specs[li], dloss, scales[li]) for li in range(L)])
            gamma_all = torch.stack([be.coef_gamma(A_all[li], cv_all[li]) for li in range(L)])
        wmaxs = [None] * L
        if linear and all(x is not None for x in xmaxs):
            if hasattr(be, 'wbound_batched'):
                wmax_all = be.wbound_batched(A_all[:, r0:r0 + Ml].contiguous(), nsq_all, gamma_all[:, r0:r0 + Ml].contiguous())
                wmaxs = [wmax_all[li] for li in range(L)]
            elif hasattr(be, 'wbound'):
                wmaxs = [be.wbound(A_all[li, r0:r0 + Ml].contiguous(), nsq_all[li], gamma_all[li, r0:r0 + Ml].contiguous())
                         for li in range(L)]
        # layers are in depth order: the deepest layer's slices travel first, and its dX GEMM (half the flops at the
        # bench's widths) runs while the other layers' slices are still on the links
        split = world > 1 and L >= 3 and knobs.shard_bwd_split
        parts = [[0], list(range(1, L))] if split else [list(range(L))]
        gate = (ctx.gate,) if ctx.gate is not None else ()
        grads = [None] * L

        def finish():
            out = [None] * L
            for pos, li in enumerate(ctx.order):
                out[li] = grads[pos]
            return (None,) + tuple(out)

        if ctx.blocked:
            # per group: ONE combine launch writing the send order [G][L'][Ml][n_r], ONE all_to_all_single, and the dX
            # launch reading the received [G][L'][Ml][rows][B] as it lies
            pend = []
            for part in parts:
                l0, Lc = part[0], len(part)
                Ws = be.combine_slice_blocked(A_all[l0:l0 + Lc], Cs_all, gamma_all[l0:l0 + Lc], l0, B, bounds[rank], detach,
                                              nsq_all[l0:l0 + Lc])
                Wr = torch.empty((world, Lc, Ml, ctx.sliced, B), device=Ws.device, dtype=Ws.dtype)
                pend.append((part, Wr, _a2a_equal(Wr.view(-1), Ws.view(-1), group)))
            for part, Wr, work in pend:
                work.wait()
                out = be.gram_bwd_multi([Wr[:, k] for k in range(len(part))], [feats[li] for li in part],
                                        [wmaxs[li] for li in part], [xmaxs[li] for li in part], *gate,
                                        slice_rows=ctx.sliced, slice_members=len(part) * Ml)
                for li, g in zip(part, out):
                    grads[li] = g
            return finish()

        # every layer's slice of W, then the exchanges
        if not n_r:
            Ws_all = torch.empty((L, M, 0), device=Cs_all.device, dtype=Cs_all.dtype)
        elif hasattr(be, 'combine_slice_batched'):
            Ws_all = be.combine_slice_batched(A_all, Cs_all, gamma_all, B, bounds[rank], detach, nsq_all)        # [L, M, n_r]
        else:
            Ws_all = torch.stack([be.combine_slice(A_all[li], Cs_all[li], gamma_all[li], B, bounds[rank], detach) for li in range(L)])
        recv = {}
        works = []
        for part in parts:
            if world == 1:
                for li in part:
                    recv[li] = Ws_all[li].reshape(-1)
                works.append(_Done())
                continue
            for li in part:
                recv[li] = torch.empty((Ml * N2,), device=Ws_all.device, dtype=Ws_all.dtype)
            works.append(_exchange([Ws_all[li].reshape(-1) for li in part], [recv[li] for li in part],
                                   [Ml * n_r] * world, [Ml * n for n in ns], group))

        def rows_of(li):     # the local members' W [Ml, B, B] from the segments as they arrived ([G][Ml][n_g])
            r = recv[li]
            if world == 1:
                return r.view(Ml, B, B)
            if even:         # [G, Ml, n] -> [Ml, G, n] in one strided copy
                return r.view(world, Ml, n_r).permute(1, 0, 2).contiguous().view(Ml, B, B)
            W = torch.empty((Ml, N2), device=r.device, dtype=r.dtype)
            off = 0
            for g in range(world):
                W[:, bounds[g]:bounds[g + 1]] = r[off:off + Ml * ns[g]].view(Ml, ns[g])
                off += Ml * ns[g]
            return W.view(Ml, B, B)

        for part, work in zip(parts, works):
            work.wait()
            if not linear:
                for li in part:
                    C, rmean, gdiag, hband = ctx.aux[li]
                    grads[li] = be.transformed_gram_bwd_w(rows_of(li), C, rmean, gdiag, hband, feats[li], ctx.gspecs[li], *gate)
                continue
            Xs, wm, xm = [feats[li] for li in part], [wmaxs[li] for li in part], [xmaxs[li] for li in part]
            multi = getattr(be, 'gram_bwd_multi_eligible', None)
            if ctx.sliced and multi is not None and multi([recv[li] for li in part], Xs, wm, xm, ctx.sliced):
                # W stays as it arrived, [G, Ml, rows, B]: the dX launch reads it in that order (one layer is a valid launch)
                out = be.gram_bwd_multi([recv[li] for li in part], Xs, wm, xm, *gate, slice_rows=ctx.sliced)
            else:
                Wl = [rows_of(li) for li in part]
                if multi is not None and multi(Wl, Xs, wm, xm):
                    out = be.gram_bwd_multi(Wl, Xs, wm, xm, *gate)
                else:
                    out = [be.gram_bwd_w(W, X, a, b, *gate) for W, X, a, b in zip(Wl, Xs, wm, xm)]
            for li, g in zip(part, out):
                grads[li] = g
        return finish()


class _GatheredLayers(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cfg, *feats):
        backend, group, specs, weights, detach, reduce, gspecs = cfg
        if gspecs is not None:
            raise NotImplementedError("cossim / rbf layers are sharded by the 'slice' scheme only")
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
    mode = knobs.shard_mode
    if mode == 'slice' and not all(hasattr(backend, f) for f in ('coef_gamma', 'combine_slice', 'gram_bwd_w')):
        return 'gather'
    return mode


def sharded_layers_loss(feats, specs, weights, detach_diag=True, reduce='mean', group=None, backend=None, gspecs=None):
    """Global `model_layers` loss from this rank's member slice of every layer's features.
    feats: list of [M/G, B, P_l]; specs: one ops.HeadSpec per layer; weights: per-layer floats; gspecs: one ops.GramSpec
    per layer for cossim / rbf Grams (all layers; None = centred linear Grams)."""
    if backend is None:
        from . import ops as backend
    if reduce not in ('mean', 'sum'):
        raise ValueError("the sharded loss supports reduction 'mean' or 'sum'")
    if gspecs is not None:
        if hasattr(backend, '_features_f32'):
            feats = [backend._features_f32(X) for X in feats]
    elif hasattr(backend, '_features'):      # 16-bit features: native where the kernels take them, else the exact up-cast
        feats = [backend._features(X) for X in feats]
        if len({X.dtype for X in feats}) > 1:
            feats = [backend._features_f32(X) for X in feats]
    fn = _SlicedLayers if (gspecs is not None or shard_mode(backend) == 'slice') else _GatheredLayers
    return fn.apply((backend, group, list(specs), list(weights), bool(detach_diag), reduce,
                     list(gspecs) if gspecs is not None else None), *feats)


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
        gspecs = [kk.feature_kernel.gram_spec() for kk in kerns]
        if any(s is None for s in specs) or any(g is None for g in gspecs):
            raise NotImplementedError('member sharding needs centred, non-learnable he / cka layer kernels on linear, '
                                      'cossim or rbf Grams')
        weights = [float(k.layer_weighting.get_weight(i, len(feats))) for i in range(len(feats))]
        xs = [kerns[i].feature_kernel.flatten(f) for i, f in enumerate(feats.values())]
        # one call (= one set of collectives) per kind of Gram and detach_diag setting; a shipped config has one
        groups = {}
        for i, (kk, g) in enumerate(zip(kerns, gspecs)):
            lin = g.kind == 0        # ops.GRAM_LINEAR
            groups.setdefault((lin, bool(kk.feature_kernel.detach_diag) if lin else True), []).append(i)
        total = None
        for (lin, detach), idx in groups.items():
            part = sharded_layers_loss([xs[i] for i in idx], [specs[i] for i in idx], [weights[i] for i in idx], detach,
                                       k.reduction, self.group, self.backend,
                                       None if lin else [gspecs[i] for i in idx])
            total = part if total is None else total + part
        return total


class _ScaledLoss(torch.nn.Module):
    def __init__(self, loss, factor):
        super().__init__()
        self.loss, self.factor = loss, factor

    def forward(self, *args, **kwargs):
        return self.factor * self.loss(*args, **kwargs)


class _AllReduceSum(torch.autograd.Function):
    """Sum over ranks of a quantity every rank then uses in the SAME global loss term: the gradient of that term with
    respect to this rank's contribution is the upstream gradient itself (each rank differentiates its own copy of the
    global term, and only through its own members)."""

    @staticmethod
    def forward(ctx, t, group):
        out = t.detach().clone()
        _all_reduce(out, group)
        return out

    @staticmethod
    def backward(ctx, g):
        return g, None


class ShardedMeanSoftmaxEntropy(torch.nn.Module):
    """`mean_softmax_entropy` (hyper/diversity/losses.py:77: negative entropy of the MEMBER-AVERAGED softmax) when the
    members are sharded: the local softmax sums [samples, classes] are added over the ranks (one tiny all-reduce), so the
    value is the unsharded loss exactly and every rank's backward reaches its own members only."""

    def __init__(self, group=None):
        super().__init__()
        self.group = group

    def forward(self, model_bs, x, pred, gt):
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        psum = torch.nn.functional.softmax(pred, dim=-1).sum(0)
        if world > 1:
            psum = _AllReduceSum.apply(psum, self.group)
        p = psum / (pred.shape[0] * world)
        return -(-(p * torch.log(p + 1e-8)).sum(-1)).mean()


class _ShardedInlierOutlierEntropy(torch.nn.Module):
    """`inlier_outlier_entropy` (losses.py:94-113) under member sharding: cross-entropy of the local members / G + weight *
    the exact member-averaged entropy term."""

    def __init__(self, loss, world, group):
        super().__init__()
        self.loss, self.world, self.ood_entropy = loss, world, ShardedMeanSoftmaxEntropy(group)

    def forward(self, model_bs, x, forward_data, gt):
        return (self.loss.ce(model_bs, x, forward_data['pred_ind'], gt) / self.world
                + self.loss.weight * self.ood_entropy(model_bs, x, forward_data['pred_ood'], None))


def shard_method(method, group=None, backend=None):
    """Member-sharded TRAINING (last sentence; BASELINE configs[4]).  Every rank holds M / G members on the
    full batch - the rank's own members of a fixed ensemble (`ensemble_size` = `num` = M / G), or its slice of the members
    a hypernetwork samples this step (`sample_member_latents`) - and calls `shard_method` once on its `repulsive_kernel` /
    `mle` method (built with `num` = M / G).  After that rank r's loss is
        beta CE(own members) / G + gamma rep(ALL members) [+ gamma_ood rep_ood(ALL members) + beta_ood H(ALL members)]
    and its backward leaves on whatever produced the local members' parameters exactly the part of the unsharded step's
    gradient that flows through those members: the repulsion terms are the sharded kernels (global value, gradients for the
    local members; linear, cossim and rbf Grams), the OOD entropy of the member-averaged softmax is made exact by one tiny
    all-reduce of the softmax sums (`ShardedMeanSoftmaxEntropy`), the per-member means of the data and L2 terms are
    rescaled from M / G members to M.
      * fixed ensembles: members share no weights - no gradient exchange exists or is needed;
      * hypernetworks: the generator's weights are shared by all members, so after `loss.backward()` the generator
        gradients of the ranks are summed with `allreduce_gradients(hyper)` (ONE flat bucket over RCCL).
    Unlike the reference's DDP (per-rank sub-batch Grams; hyper/experiments/training.py:686-695) the loss is
    the single-GPU loss.  Learnable method parameters (gamma, beta) are refused: their gradients mix global and local terms."""
    from .diversity.losses import InlierOutlierEntropy, MeanSoftmaxEntropy
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world > 1 and any(p.requires_grad for n, p in method.named_parameters() if n in ('gamma', 'gamma_ood', 'beta_ind', 'beta_ood')):
        raise NotImplementedError('shard_method: learnable gamma / beta are not sharded')
    ood = getattr(method, 'ood_loss', None)
    if ood is not None and not isinstance(ood, ShardedMeanSoftmaxEntropy):
        if not isinstance(ood, MeanSoftmaxEntropy):
            raise NotImplementedError('shard_method: OOD loss %s is not sharded' % type(ood).__name__)
        method.ood_loss = ShardedMeanSoftmaxEntropy(group)
    for name in ('model_kernel', 'model_kernel_ood'):
        kern = getattr(method, name, None)
        if kern is not None and not isinstance(kern, ShardedModelLayersKernel):
            setattr(method, name, ShardedModelLayersKernel(kern, group, backend))
    if method.ind_loss is not None and not isinstance(method.ind_loss, (_ScaledLoss, _ShardedInlierOutlierEntropy)):
        if isinstance(method.ind_loss, InlierOutlierEntropy):
            method.ind_loss = _ShardedInlierOutlierEntropy(method.ind_loss, world, group)
        else:
            method.ind_loss = _ScaledLoss(method.ind_loss, 1.0 / world)
        if method.l2_reg is not None:
            method.l2_reg = method.l2_reg / world
    return method


def sample_member_latents(hyper, members, device, group=None):
    """This rank's slice of the `members` latent samples of a hypernetwork step: rank 0 draws them all
    (`hyper.sample_params`, hyper/generators/mlp.py:42-44), broadcasts the small [members, latent] tensor, and every rank
    keeps rows [r M / G, (r + 1) M / G) - the same members, in the same order, as the unsharded step would sample."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if members % world:
        raise ValueError('%d sampled members do not shard over %d ranks' % (members, world))
    z = hyper.sample_params(members, device=device)
    if world > 1:
        src = 0 if group is None else dist.get_global_rank(group, 0)
        _sent(z.numel() * z.element_size() if rank == 0 else 0)
        if _staged(group, z):
            h = z.cpu()
            dist.broadcast(h, src, group=group)
            z.copy_(h)
        else:
            dist.broadcast(z, src, group=group)
    ml = members // world
    return z[rank * ml:(rank + 1) * ml]


def allreduce_gradients(module, group=None):
    """Sum the gradients of a module's parameters over the ranks in ONE flat bucket (the hypernetwork's generator weights are
    shared by the members every rank holds; replaces DistributedDataParallel's bucketed all-reduce of the reference,
    hyper/experiments/training.py:693, for the member-sharded step).  Parameters without a gradient count as zero."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    params = [p for p in module.parameters() if p.requires_grad]
    if world == 1 or not params:
        return
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])
    _all_reduce(flat, group)
    off = 0
    for p in params:
        n = p.numel()
        g = flat[off:off + n].view_as(p)
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
        off += n
