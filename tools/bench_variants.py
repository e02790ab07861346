"""The measurement variants SURVEY.md 8d lists beside the headline bench (same M=64, B=512, five layers), on one GPU:
  base        - bench.py's workload (independent members, he power s=2)
  correlated  - X <- 0.7 S + 0.3 X with a shared S per layer (cos 0.7-0.9: stresses arccos)
  he_exp      - use_exp, eps = arc_eps = 0.001 (configs/cifar10/ens/he-lin.yml:13-21)
  bf16        - the same tensors .bfloat16() (enter the fp32 kernels through the exact up-cast; dX is cast back)
and a library ceiling for the two big products in 16-bit: torch.bmm (hipBLASLt) on fp16 copies of W and X.

    python tools/bench_variants.py [--steps 20]
"""
import argparse
import copy
import os
import sys
import time
from collections import OrderedDict

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'he-cka-ensembles_amd')]


def timed(fn, steps, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--only', default='', help='one variant (and no library ceiling)')
    args = ap.parse_args()
    import bench
    from hyper_amd.diversity.model_kernels import build_model_kernel
    dev = torch.device('cuda:0')
    base = bench.make_feats(dev, slice(0, bench.M))

    def variant(name):
        cfg = copy.deepcopy(bench.CFG)
        feats = base
        if name == 'correlated':
            torch.manual_seed(99)
            feats = [0.7 * torch.randn(1, bench.B, p, device=dev) + 0.3 * x for p, x in zip(bench.DIMS, base)]
        elif name == 'he_exp':
            cfg['layer_kernel'].update(use_exp=True, eps=0.001, arc_eps=0.001)
        elif name == 'bf16':
            feats = [x.bfloat16() for x in base]
        kern = build_model_kernel(cfg).to(dev)
        xs = [f.clone().requires_grad_(True) for f in feats]
        fd = OrderedDict((str(i), x) for i, x in enumerate(xs))
        out = {}

        def step():
            for x in xs:
                x.grad = None
            loss = kern(None, fd, None, OrderedDict())
            loss.backward()
            out['loss'] = loss
        dt = timed(step, args.steps)
        print('%-11s %7.1f evals/s  %6.3f ms/eval  loss %.6f' % (name, 1.0 / dt, 1e3 * dt, out['loss'].item()), flush=True)

    if args.only:
        variant(args.only)
        return
    for name in ('base', 'correlated', 'he_exp', 'bf16'):
        variant(name)

    # 16-bit library ceiling of the two products at the largest layer (not a parity-grade computation: one fp16 term)
    P = bench.DIMS[-1]
    X = base[-1].half()
    W = torch.randn(bench.M, bench.B, bench.B, device=dev).half()
    for label, fn, flops in (
            ('bmm fp16 X X^T (P=%d)' % P, lambda: torch.bmm(X, X.transpose(1, 2)), 2.0 * bench.M * bench.B * bench.B * P),
            ('bmm fp16 W X   (P=%d)' % P, lambda: torch.bmm(W, X), 2.0 * bench.M * bench.B * bench.B * P)):
        dt = timed(fn, 20)
        print('%-24s %7.1f us  %7.1f TFLOP/s (one fp16 product per multiply)' % (label, 1e6 * dt, flops / dt / 1e12), flush=True)


if __name__ == '__main__':
    main()
