"""What a workgroup of the all-layers dX launch pays outside its k-loop (first loads, epilogue): the same flops as
(M = 64, B = 512) run as (M = 16, B = 1024) - half the workgroups, each with twice the k-tiles.
    python tools/dx_fill_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'he-cka-ensembles_amd')]
import torch  # noqa: E402
from hyper_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
DIMS = [512, 1024, 2048, 4096, 8192]


def timed(fn, n=12, warm=3):
    for _ in range(warm):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2] * 1e3


res = {}
for M, B in ((64, 512), (16, 1024), (256, 256)):
    torch.manual_seed(0)
    Xs = [torch.randn(M, B, p, device=dev) for p in DIMS]
    Ws = [torch.randn(M, B, B, device=dev) for _ in DIMS]
    wmaxs = [W.abs().amax(dim=(1, 2)).contiguous() for W in Ws]
    xmaxs = [X.abs().amax(dim=(1, 2)).contiguous() for X in Xs]
    t = timed(lambda: ops.gram_bwd_multi(Ws, Xs, wmaxs, xmaxs))
    wgs = sum(M * (B // 256) * (p // 256) for p in DIMS)
    res[(M, B)] = (t, wgs)
    print('M=%3d B=%4d: dX %8.1f us, %5d workgroups of %2d k-tiles' % (M, B, t, wgs, B // 32), flush=True)
    del Xs, Ws
(t1, w1), (t2, w2), (t3, w3) = res[(64, 512)], res[(16, 1024)], res[(256, 256)]
# t = rounds * (o + n tk) with rounds = wgs / 256:  two equations in o (per-workgroup overhead) and tk (per k-tile)
r1, r2 = w1 / 256.0, w2 / 256.0
tk = (t2 / r2 - t1 / r1) / 16.0
o = t1 / r1 - 16.0 * tk
print('per workgroup: %.2f us outside the k-loop, %.2f us per k-tile (from B = 512 and 1024)' % (o, tk))
