"""Streaming self-pairwise kernel (pairwise_self64_kernel; HECKA_PAIRWISE_X6=1: its bf16 six-product form) against fp64, with timings.
    python tools/pairwise_check.py [B ...]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'he-cka-ensembles_amd')]
import torch  # noqa: E402
from hyper_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
x6 = os.environ.get('HECKA_PAIRWISE_X6') == '1'
for B in [int(a) for a in sys.argv[1:]] or [512, 128, 64]:
    for K in (B * B, B * B // 8):
        if K % 32:
            continue
        torch.manual_seed(B)
        C = torch.randn(64, K, device=dev) * torch.rand(64, 1, device=dev)
        S = ops.pairwise_inner(C)
        ref = C.double() @ C.double().t()
        err = float((S.double() - ref).abs().max() / ref.abs().max())
        sym = float((S - S.t()).abs().max())
        for _ in range(3):
            ops.pairwise_inner(C)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.pairwise_inner(C)
        e1.record()
        torch.cuda.synchronize()
        print('%s K=%7d  rel err %.2e  asym %.1e  %.1f us' % ('bf16x6' if x6 else 'f32   ', K, err, sym,
                                                               e0.elapsed_time(e1) / 20 * 1e3), flush=True)
