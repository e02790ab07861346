"""Error of hecka_gram_bwd_f32 (dX = W X) against fp64: max and mean of (result - exact) / sum_k |w x|.
    python tools/bwd_accuracy.py B P      (B <= 1024: one accumulation chain; above: 64-k local chains)"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'he-cka-ensembles_amd')]
import torch  # noqa: E402
from hyper_amd import _lib  # noqa: E402

B, P, M = int(sys.argv[1]), int(sys.argv[2]), 8
dev = torch.device('cuda:0')
torch.manual_seed(1)
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
for name, W, X in (('mixed signs', torch.randn(M, B, B, device=dev), torch.randn(M, B, P, device=dev)),
                   ('all positive', torch.rand(M, B, B, device=dev) + 0.1, torch.rand(M, B, P, device=dev) + 0.1)):
    wmax = W.abs().amax(dim=(1, 2)).contiguous()
    xmax = X.abs().amax(dim=(1, 2)).contiguous()
    dX = torch.empty_like(X)
    assert L.hecka_gram_bwd_f32(W.data_ptr(), X.data_ptr(), wmax.data_ptr(), xmax.data_ptr(), None, M, B, P, dX.data_ptr(), 0, st) == 0
    ref = torch.bmm(W.double(), X.double())
    sabs = torch.bmm(W.double().abs(), X.double().abs())
    e = (dX.double() - ref) / sabs
    print('B=%d P=%d %-12s max |err|/sum|wx| %.3e   mean err/sum|wx| %+.3e   max |err|/max|dX| %.3e' % (
        B, P, name, e.abs().max().item(), e.mean().item(), ((dX.double() - ref).abs().max() / ref.abs().max()).item()))
