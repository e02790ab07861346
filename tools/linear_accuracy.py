import sys, os
sys.path[:0] = ['/root/repo', '/root/repo/he-cka-ensembles_amd']
import torch, torch.nn.functional as F
from hyper_amd.net import hip
dev = torch.device('cuda:0')
torch.manual_seed(0)
for N, I, O in [(6, 8, 64), (6, 64, 64), (128, 38, 256), (128, 256, 512), (6, 64, 2), (128, 512, 2294), (24, 12, 100)]:
    x = torch.randn(N, I, device=dev); W = torch.randn(O, I, device=dev) / I ** 0.5; b = torch.randn(O, device=dev); dy = torch.randn(N, O, device=dev)
    res = {}
    for name, fn in (('hip', hip.linear), ('torch', F.linear)):
        xs, Ws, bs = x.clone().requires_grad_(True), W.clone().requires_grad_(True), b.clone().requires_grad_(True)
        y = fn(xs, Ws, bs); y.backward(dy)
        res[name] = (y.detach(), xs.grad, Ws.grad, bs.grad)
    xd, Wd, bd = x.double().requires_grad_(True), W.double().requires_grad_(True), b.double().requires_grad_(True)
    yd = F.linear(xd, Wd, bd); yd.backward(dy.double())
    ref = (yd.detach(), xd.grad, Wd.grad, bd.grad)
    rel = lambda a, r: float((a.double() - r).abs().max() / r.abs().max())
    print(N, I, O, ' '.join('%s: y %.1e dx %.1e dW %.1e db %.1e |' % ((k,) + tuple(rel(a, r) for a, r in zip(v, ref))) for k, v in res.items()))
