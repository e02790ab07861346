"""The two HBM-side stages of the headline workload - all-layers pairwise S = C C^T and all-layers combine
W = 2 A C + gamma terms - timed alone at the bench shape (L = 5, M = 64, B = 512) and at an 8-way member shard's slice,
with their error against fp64.      python tools/small_ab.py [tag]       (HECKA_LIB_PATH selects the library)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'he-cka-ensembles_amd')]
import torch  # noqa: E402
from hyper_amd import ops  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else 'tree'
dev = torch.device('cuda:0')
L, M, B = 5, 64, 512
torch.manual_seed(0)
Cs = []
for l in range(L):
    spread = 10.0 ** torch.linspace(-3, 3, M, device=dev)[:, None, None] if l == 0 else 0.5 + torch.rand(M, 1, 1, device=dev)
    X = torch.randn(M, B, 256, device=dev) * (1.0 + l) * spread          # layer 0: members six decades apart
    Cs.append(ops.centered_gram(X)[0].view(M, B * B))
C_all = torch.stack(Cs)
del Cs
A_all = torch.randn(L, M, M, device=dev) / (C_all.norm(dim=2)[:, :, None] * C_all.norm(dim=2)[:, None, :])
nsq_all = (C_all.double() ** 2).sum(dim=2).float().contiguous()
gamma_all = torch.randn(L, M, B, device=dev) * 1e-3 * (A_all.abs().amax(dim=2, keepdim=True) * C_all.abs().amax(dim=2)[:, :, None])


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for G in (1, 8):
    n = B * B // G
    Cg = C_all[:, :, :n].contiguous()
    S = ops.pairwise_self_batched(Cg)
    Sd = Cg[0].double() @ Cg[0].double().t()
    es = float((S[0].double() - Sd).abs().max() / Sd.abs().max())
    dg = torch.diagonal(S[0]).double()
    ec = float(((S[0].double() - Sd) / torch.sqrt(dg[:, None] * dg[None, :])).abs().max())
    W = ops.combine_slice_batched(A_all, Cg, gamma_all, B, 0, True, nsq_all)
    a = torch.arange(n, device=dev) // B
    b = torch.arange(n, device=dev) % B
    Wd = 2.0 * (A_all[0].double() @ Cg[0].double()) + gamma_all[0].double()[:, a] - gamma_all[0].double()[:, b]
    Wd[:, a == b] = 0
    ew = float(((W[0].double() - Wd).abs().amax(dim=1) / Wd.abs().amax(dim=1)).max())       # per row of W: rows differ by decades
    tp = timed(lambda: ops.pairwise_self_batched(Cg))
    tc = timed(lambda: ops.combine_slice_batched(A_all, Cg, gamma_all, B, 0, True, nsq_all))
    nb = 4.0 * L * M * n
    print('%-10s G=%d  pairwise %6.1f us (%4.0f GB/s) err %.1e cos-err %.1e   combine %6.1f us (%4.0f GB/s) err %.1e' %
          (tag, G, tp, nb / tp / 1e3, es, ec, tc, 2 * nb / tc / 1e3, ew), flush=True)
