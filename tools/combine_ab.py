"""Times hecka_combine_slice_batched_f32 / pairwise at the headline shape for the library HECKA_LIB_PATH names."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'he-cka-ensembles_amd')]
import torch
from hyper_amd import ops
M, L = 64, 5
B = int(os.environ.get('COMBINE_AB_B', '512'))
dev = torch.device('cuda:0')
torch.manual_seed(0)
A = torch.randn(L, M, M, device=dev) / M
C = torch.randn(L, M, B * B, device=dev)
gamma = torch.randn(L, M, B, device=dev)
def timed(fn, n=20, warm=3):
    for _ in range(warm): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2] * 1e3
t = timed(lambda: ops.combine_slice_batched(A, C, gamma, B, 0, True))
print('%-10s B=%d combine (5 layers) %7.1f us  %.2f TB/s (C read + W written)' % (sys.argv[1] if len(sys.argv) > 1 else '', B, t, 2 * L * M * B * B * 4 / t / 1e6), flush=True)
