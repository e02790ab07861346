"""Compute-side cost of ONE rank of a G-way member-sharded evaluation, measured on one GPU: the collectives of
hyper_amd.parallel are replaced by local stand-ins that move nothing between ranks (the received buffers are
filled from the rank's own data, shapes as in the real exchange), so the time is what rank 0 spends in its own
kernels, packing and unpacking.  The exchange itself (xGMI) is NOT in it - this bounds the scaling from above.

    python tools/shard_emulate.py [--modes slice,gather] [--worlds 1,2,4,8]
"""
import argparse
import os
import sys
import time
from collections import OrderedDict

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'he-cka-ensembles_amd')]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--modes', default='slice,gather')
    ap.add_argument('--worlds', default='1,2,4,8')
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--no-exchange-copies', action='store_true', help='skip the stand-in copies of the exchanges in the timed evaluations')
    args = ap.parse_args()
    import bench
    from hyper_amd import parallel
    from hyper_amd.diversity.model_kernels import build_model_kernel
    dev = torch.device('cuda:0')
    state = {'world': 1, 'copy': True}

    class FakeDist:
        ReduceOp = torch.distributed.ReduceOp

        @staticmethod
        def get_world_size(group=None):
            return state['world']

        @staticmethod
        def get_rank(group=None):
            return 0

    def exchange(sends, recvs, send_splits, recv_splits, group):
        # even splits: same element count; the rank's own data stands in for its peers'.  After the warm-up evaluations
        # the copy is skipped (--no-exchange-copies): the receive buffers come back from the caching allocator with the
        # previous evaluation's (valid) contents, and the time is then the rank's own kernels only
        if state['copy']:
            for snd, rcv in zip(sends, recvs):
                rcv.copy_(snd)
        return parallel._Done()

    def gather(out, inp, group):
        if state['copy']:
            out.view(state['world'], -1).copy_(inp.reshape(1, -1).expand(state['world'], -1))
        return parallel._Done()

    def a2a_equal(out, inp, group):
        if state['copy']:
            out.copy_(inp)
        return parallel._Done()

    parallel.dist = FakeDist
    parallel._exchange, parallel._all_gather, parallel._all_reduce = exchange, gather, (lambda t, group: None)
    parallel._a2a_equal = a2a_equal
    for mode in args.modes.split(','):
        os.environ['HECKA_SHARD_MODE'] = mode
        for world in [int(w) for w in args.worlds.split(',')]:
            state['world'] = world
            Ml = bench.M // world
            feats = bench.make_feats(dev, slice(0, Ml))
            kern = parallel.ShardedModelLayersKernel(build_model_kernel(bench.CFG).to(dev))
            xs = [f.requires_grad_(True) for f in feats]
            fd = OrderedDict((str(i), x) for i, x in enumerate(xs))

            def step():
                for x in xs:
                    x.grad = None
                kern(None, fd, None, OrderedDict()).backward()
            state['copy'] = True
            for _ in range(5):
                step()
            torch.cuda.synchronize()
            state['copy'] = not args.no_exchange_copies
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            ms = 1e3 * (time.perf_counter() - t0) / args.steps
            print('mode %-6s G=%d  members/rank %2d  rank-0 compute %.3f ms/eval  (%.0f evals/s if the exchange were free)%s'
                  % (mode, world, Ml, ms, 1e3 / ms, '' if state['copy'] else '  [no stand-in copies]'), flush=True)


if __name__ == '__main__':
    main()
