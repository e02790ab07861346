"""Probe: why is the first evaluation after an RCCL collective slow (world size 1, one-GPU box)?"""
import os, sys, time
from collections import OrderedDict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'he-cka-ensembles_amd')]
import torch, torch.distributed as dist
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29545', RANK='0', WORLD_SIZE='1')
dev = torch.device('cuda:0'); torch.cuda.set_device(dev)
dist.init_process_group(os.environ.get('BACKEND', 'nccl'), device_id=dev)
import bench
from hyper_amd.diversity.model_kernels import build_model_kernel
from hyper_amd.parallel import ShardedModelLayersKernel
feats = bench.make_feats(dev, slice(0, 64), False)
kern = build_model_kernel(bench.CFG).to(dev)
if os.environ.get('SHARDED', '1') == '1':
    kern = ShardedModelLayersKernel(kern)
xs = [f.requires_grad_(True) for f in feats]
fd = OrderedDict((str(i), x) for i, x in enumerate(xs))
tok = torch.zeros(1, device=dev)
def phases():
    for x in xs: x.grad = None
    torch.cuda.synchronize(); t0 = time.perf_counter()
    loss = kern(None, fd, None, OrderedDict()); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    loss.backward(); t3 = time.perf_counter(); torch.cuda.synchronize(); t4 = time.perf_counter()
    return 'fwd issue %.2f + drain %.2f | bwd issue %.2f + drain %.2f ms' % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3)
for _ in range(5): phases()
for what in ('nothing', 'all_reduce', 'nothing', 'all_reduce', 'sync only', 'all_reduce no sync'):
    if what.startswith('all_reduce'): dist.all_reduce(tok)
    if what != 'all_reduce no sync': torch.cuda.synchronize()
    if os.environ.get('TIMING_EVENT'):
        t0 = time.perf_counter(); e = torch.cuda.Event(enable_timing=True); e.record(); print('timing event record: %.2f ms' % ((time.perf_counter() - t0) * 1e3))
    print('%-20s -> %s' % (what, phases()))
    print('%-20s -> %s' % ('  (next)', phases()))
def plain():
    for x in xs: x.grad = None
    loss = kern(None, fd, None, OrderedDict())
    loss.backward()
    return loss
for what in ('nothing', 'all_reduce', 'all_reduce', 'all_reduce + timing event'):
    if what.startswith('all_reduce'): dist.all_reduce(tok)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if what.endswith('event'):
        e = torch.cuda.Event(enable_timing=True); e.record()
    ts = []
    for _ in range(3):
        plain(); ts.append('%.2f' % ((time.perf_counter() - t0) * 1e3))
    torch.cuda.synchronize()
    print('%-26s -> wall after evals 1..3 %s, drained %.2f ms' % (what, ts, (time.perf_counter() - t0) * 1e3))
big = torch.zeros(8 << 20, device=dev); out = torch.zeros(8 << 20, device=dev)
for what in ('no collective', 'all_reduce(1) per eval', 'all_to_all(32 MB) + all_reduce per eval'):
    torch.cuda.synchronize()
    ts = []
    for i in range(150):
        t0 = time.perf_counter()
        if what.startswith('all_reduce'): dist.all_reduce(tok)
        if what.startswith('all_to_all'):
            dist.all_to_all_single(out, big); dist.all_reduce(tok)
        plain()
        if i % 3 == 2: torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    torch.cuda.synchronize()
    slow = [(i, round(t, 1)) for i, t in enumerate(ts) if t > 15]
    print('%-40s -> 150 evals, total %.1f ms, host stalls > 15 ms: %s' % (what, sum(ts), slow))
dist.destroy_process_group()
