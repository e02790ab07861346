"""Times the two GEMM launches of the headline workload (Gram of five layers, dX of five layers) on random operands through
hyper_amd.ops, for whatever library HECKA_LIB_PATH names - the csrc/lab ablation builds (HECKA_ABL) are run through this.
    HECKA_LIB_PATH=... python tools/tile_ablation.py [tag]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'he-cka-ensembles_amd')]
import torch  # noqa: E402
from hyper_amd import ops  # noqa: E402

M, B, DIMS = 64, 512, [512, 1024, 2048, 4096, 8192]
dev = torch.device('cuda:0')
torch.manual_seed(3)
Xs = [torch.randn(M, B, p, device=dev) for p in DIMS]
Ws = [torch.randn(M, B, B, device=dev) for _ in DIMS]
wmaxs = [W.abs().amax(dim=(1, 2)).contiguous() for W in Ws]


def timed(fn, n=12, warm=3):
    for _ in range(warm):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2] * 1e3


out = ops.centered_gram_multi(Xs)
xmaxs = [o[2] for o in out]
# (accuracy of the Gram against fp64 on the narrowest and the deepest layer, so that a variant build is not timed blind)
errs = []
for li in (0, len(DIMS) - 1):
    Xd = Xs[li][:4].double()
    Kd = Xd @ Xd.transpose(1, 2)
    Cd = Kd - Kd.mean(dim=2, keepdim=True)
    errs.append(float((out[li][0][:4].double() - Cd).abs().max() / Cd.abs().max()))
tg = timed(lambda: ops.centered_gram_multi(Xs))          # (includes the column-mean launch)
tb = timed(lambda: ops.gram_bwd_multi(Ws, Xs, wmaxs, xmaxs))
print('%-10s colmean+gram %8.1f us   dX %8.1f us   gram err %.1e / %.1e' % (sys.argv[1] if len(sys.argv) > 1 else '', tg, tb, errs[0], errs[1]),
      flush=True)
