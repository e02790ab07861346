"""Time (or expose to rocprofv3) ONE libhecka kernel at a given shape.

    python tools/kernel_bench.py gram_bwd 64 512 8192 [iters]
    python tools/kernel_bench.py gram_center|colmean|pairwise|combine M B P [iters]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'he-cka-ensembles_amd')]
import torch  # noqa: E402
from hyper_amd import _lib, ops  # noqa: E402

name, M, B, P = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 10
dev = torch.device('cuda:0')
torch.manual_seed(0)
X = torch.randn(M, B, P, device=dev)
L = _lib.lib()
MODE = int(os.environ.get('HECKA_GEMM_MODE', '0'))
st = torch.cuda.current_stream().cuda_stream
p = lambda t: t.data_ptr()
sbar = torch.empty(M, P, device=dev)
C = torch.empty(M, B, B, device=dev)
cvec = torch.empty(M, B, device=dev)
W = torch.randn(M, B, B, device=dev)
dX = torch.empty_like(X)
A = torch.randn(M, M, device=dev)
gamma = torch.randn(M, B, device=dev)
xmax = torch.empty(M, device=dev)
wmax = torch.empty(M, device=dev)
L.hecka_colmean_f32(p(X), M, B, P, p(sbar), p(xmax), st)
L.hecka_gram_center_f32(p(X), p(sbar), p(xmax), M, B, P, p(C), p(cvec), MODE, st)
nb = L.hecka_pairwise_workspace_bytes(M, M, B * B)
ws = torch.empty(max(nb, 4) // 4, device=dev)
S = torch.empty(M, M, device=dev)
fns = {
    'colmean': (lambda: L.hecka_colmean_f32(p(X), M, B, P, p(sbar), p(xmax), st), 0, 4.0 * M * B * P),
    'gram_center': (lambda: L.hecka_gram_center_f32(p(X), p(sbar), p(xmax), M, B, P, p(C), p(cvec), MODE, st), 2.0 * M * B * B * P, 0),
    'pairwise': (lambda: L.hecka_pairwise_f32(p(C), M, p(C), M, B * B, p(S), p(ws), nb, st), 2.0 * M * M * B * B, 4.0 * M * B * B),
    'combine': (lambda: L.hecka_combine_f32(p(A), p(C), p(gamma), M, M, B, 1, p(W), p(wmax), st), 2.0 * M * M * B * B, 8.0 * M * B * B),
    'combine_nomax': (lambda: L.hecka_combine_f32(p(A), p(C), p(gamma), M, M, B, 1, p(W), None, st), 2.0 * M * M * B * B, 8.0 * M * B * B),
    'head_fwd': (lambda: L.hecka_head_fwd_f32(p(S), p(nsq), M, M, 0, 1, 2.0, 0.0025, 0.031, 4, None, p(tot), 1.0, st), 0, 0),
    'gram_bwd': (lambda: L.hecka_gram_bwd_f32(p(W), p(X), p(wmax), p(xmax), None, M, B, P, p(dX), MODE, st), 2.0 * M * B * B * P, 0),
}
L.hecka_combine_f32(p(A), p(C), p(gamma), M, M, B, 1, p(W), p(wmax), st)   # W and wmax consistent for gram_bwd
L.hecka_pairwise_f32(p(C), M, p(C), M, B * B, p(S), p(ws), nb, st)
nsq = torch.diagonal(S).contiguous()
tot = torch.empty((), device=dev)
fn, flops, nbytes = fns[name]
for _ in range(3):
    assert fn() == 0
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    fn()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
print('%s M=%d B=%d P=%d: %.1f us  %.1f TFLOP/s  %.0f GB/s' % (name, M, B, P, ms * 1e3, flops / ms / 1e9, nbytes / ms / 1e6))
