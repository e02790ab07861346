"""Time hecka_colmean_f32 on the bench's five layer shapes (M = 64, B = 512): us per launch and GB/s of X read.
    python tools/colmean_bench.py [--noflush]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'he-cka-ensembles_amd')]
from hyper_amd import _lib  # noqa: E402

dev = torch.device('cuda:0')
M, B = 64, 512
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
tot = 0.0
for P in (8192, 4096, 2048, 1024, 512):
    X = torch.randn(M, B, P, device=dev)
    sbar = torch.empty(M, P, device=dev)
    xmax = torch.empty(M, device=dev)
    flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
    ts = []
    for it in range(12):
        if '--noflush' not in sys.argv:
            flush.zero_()                      # X out of the Infinity Cache, as in the pipeline (1 GB of features)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        _lib.check(L.hecka_colmean_f32(X.data_ptr(), M, B, P, sbar.data_ptr(), xmax.data_ptr(), st), 'colmean')
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts = sorted(ts[2:])
    med = ts[len(ts) // 2]
    ok = torch.allclose(sbar, X.mean(1), atol=1e-6) and torch.equal(xmax, X.abs().amax((1, 2)))
    tot += med
    print('P=%5d  %7.1f us  %6.0f GB/s  ok=%s' % (P, med, X.numel() * 4 / med / 1e3, ok), flush=True)
print('total %.1f us' % tot)
