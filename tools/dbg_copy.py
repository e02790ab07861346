import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'he-cka-ensembles_amd')]
import torch
from collections import OrderedDict
from hyper_amd import ops
from hyper_amd.diversity.model_kernels import build_model_kernel
import bench
dev = torch.device('cuda:0')
kern = build_model_kernel(bench.CFG).to(dev)
xs = [torch.randn(64, 512, p, device=dev).requires_grad_(True) for p in (512, 1024)]
ptrs = []
orig = ops.gram_backward
def rec(*a, **k):
    r = orig(*a, **k)
    ptrs.append(r.data_ptr())
    return r
ops.gram_backward = rec
from torch.profiler import profile, ProfilerActivity
for it in range(2):
    for x in xs: x.grad = None
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        loss = kern(None, OrderedDict((str(i), x) for i, x in enumerate(xs)), None, OrderedDict())
        loss.backward()
        torch.cuda.synchronize()
print('dX ptrs', ptrs[-2:], 'grad ptrs', [x.grad.data_ptr() for x in xs])
for e in prof.key_averages():
    if 'copy' in e.key.lower() or 'Memcpy' in e.key or 'clone' in e.key.lower():
        print(e.key[:60], e.count, e.device_time_total)
