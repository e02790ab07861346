"""HE-CKA loss+grad microbenchmark (BASELINE.json configs[3]: M=64 members, B=512 samples, 5 tracked layers
with feature dims 512..8192, fp32, synthetic features).

One "step" = one eval = scalar `model_layers` HE loss AND dLoss/dX for all five layers, through the
drop-in `hyper_amd` model kernel (autograd.Function -> ctypes -> libhecka HIP kernels).  Inputs are
resident in HBM before the timed region.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line (rank 0)."""
import argparse
import json
import os
import sys
import time
from collections import OrderedDict

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'he-cka-ensembles_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

M, B = 64, 512
DIMS = [512, 1024, 2048, 4096, 8192]
CFG = {'name': 'model_layers',
       'layer_kernel': {'name': 'he', 'he_s': 2.0, 'eps': 0.0025, 'arc_eps': 0.031,
                        'feature_kernel': {'name': 'feature', 'detach_diag': True, 'kernel': 'linear'}},
       'layer_weighting': {'name': 'uniform'}}
PEAK_F32_MATRIX_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MATRIX_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 / fp16 MFMA
PEAK_HBM_GBS = 8000.0


def make_feats(dev, members):
    torch.manual_seed(1234)
    feats = []
    for p in DIMS:
        x = torch.randn(M, B, p)          # same global tensor on every rank, then the rank's member slice
        feats.append(x[members].contiguous().to(dev))
    return feats


def cpu_baseline():
    """The oracle (reference op sequence, stock PyTorch autograd) on this box's host cores,
    one full eval of the same workload."""
    from oracle import hecka_ref as R
    torch.manual_seed(1234)
    feats = [torch.randn(M, B, p) for p in DIMS]
    lk = CFG['layer_kernel']
    cfg = dict(kernel='linear', detach_diag=True, he_s=lk['he_s'], eps=lk['eps'], arc_eps=lk['arc_eps'])
    w = R.layer_weights('uniform', len(DIMS))
    R.loss_and_grads(feats, cfg, w)                  # untimed first eval (allocator, thread pool)
    n, t0 = 0, time.perf_counter()
    while n < 12 and (n == 0 or time.perf_counter() - t0 < 12.0):   # a bounded sample: about 12 s of CPU work
        R.loss_and_grads(feats, cfg, w)
        n += 1
    dt = time.perf_counter() - t0
    return {'value': n / dt, 'unit': 'evals/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': '%d full evals (M=64,B=512,P=512..8192 x5 layers, fp32, torch %s CPU autograd) in %.1f s after one '
                      'untimed eval' % (n, torch.__version__, dt)}


SHARD_NOTE = {'slice': 'all-to-all of Gram column slices (1/G of the exchange and of every stage per rank)',
              'gather': 'all_gather of centred Grams'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    # the contract is ONE JSON line on stdout: libraries that print banners there (RCCL prints its version block at
    # communicator creation) get stderr instead; the real stdout comes back for the result line
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if os.environ.get('HECKA_BENCH_ONE_CARD') == '1':     # rehearsal of the N > 1 path on a one-GPU box (with gloo below)
        local = 0
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)
    dist = None
    force_sharded = os.environ.get('HECKA_FORCE_SHARDED') == '1'   # rehearse the RCCL path with world_size 1
    if world > 1 or force_sharded:
        import torch.distributed as dist
        if force_sharded and 'RANK' not in os.environ:
            os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29533', RANK='0', WORLD_SIZE='1')
        backend = os.environ.get('HECKA_BENCH_BACKEND', 'nccl')      # 'nccl' = RCCL; 'gloo' only for rehearsals
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)

    from hyper_amd import ops, _lib as _hl
    from hyper_amd.diversity.model_kernels import build_model_kernel
    _hl.lib().hecka_set_gemm_mode(int(os.environ.get('HECKA_GEMM_MODE', '0')))   # 0 auto (f16x3) | 2 bf16x6 | 1 fp32 MFMA

    if M % world:
        raise SystemExit('M=%d members do not shard over %d GPUs' % (M, world))
    members = slice(rank * (M // world), (rank + 1) * (M // world))   # this rank's ensemble members
    feats = make_feats(dev, members)
    kern = build_model_kernel(CFG).to(dev)
    if world > 1 or force_sharded:
        from hyper_amd.parallel import ShardedModelLayersKernel
        kern = ShardedModelLayersKernel(kern)        # all-to-all of Gram column slices over RCCL (hyper_amd/parallel.py)
    xs = [f.requires_grad_(True) for f in feats]
    fdict = OrderedDict((str(i), x) for i, x in enumerate(xs))

    def step():
        for x in xs:
            x.grad = None
        loss = kern(None, fdict, None, OrderedDict())
        loss.backward()
        return loss

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    ops.PROFILE = {'gram_bwd': []}       # HIP events around the dominant kernel, on the launch stream
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    barrier()
    dt = time.perf_counter() - t0
    prof, ops.PROFILE = ops.PROFILE, None
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        ms = 1e3 * dt / args.steps
        # dominant kernel: dX = W X (gram_bwd_kernel), 2*M*B*B*P flop per launch (SURVEY.md 8d per-unit figure)
        kms = sum(a.elapsed_time(b) for a, b, _ in prof['gram_bwd'])
        kfl = sum(fl for _, _, fl in prof['gram_bwd'])      # rank 0's launches (M/world members each)
        ach = kfl / (kms * 1e-3) / 1e12 if kms > 0 else 0.0
        from hyper_amd import _lib
        mode = _lib.lib().hecka_get_gemm_mode()
        if mode == 0:
            # interior shapes, maxima supplied: every fp32 multiply is THREE fp16 MFMA products (gemm_bf16x6.hpp,
            # scheme f16x3), so the MFMA roof for ALGORITHMIC fp32 flops is the dense 16-bit peak / 3
            kname = 'gram_bwd_x6_kernel<1, false, 0> (dX = W X; fp32 operands split into 2 fp16 terms, 3 fp16 MFMA products per multiply)'
            peak, hw = PEAK_BF16_MATRIX_TFLOPS / 3.0, {'hw_f16_tflops': ach * 3.0, 'hw_f16_peak': PEAK_BF16_MATRIX_TFLOPS}
        elif mode == 2:
            kname = 'gram_bwd_x6_kernel<0> (dX = W X; fp32 operands split into 3 bf16 terms, 6 bf16 MFMA products per multiply)'
            peak, hw = PEAK_BF16_MATRIX_TFLOPS / 6.0, {'hw_bf16_tflops': ach * 6.0, 'hw_bf16_peak': PEAK_BF16_MATRIX_TFLOPS}
        else:
            kname, peak, hw = 'gram_bwd_kernel (dX = W X, fp32 MFMA)', PEAK_F32_MATRIX_TFLOPS, {}
        # HBM bytes per launch of the dominant kernel: not measurable from inside a run - taken from the committed
        # rocprofv3 PMC passes of this same command (profiles/r01_hbm_traffic.json says how), single-GPU f16x3 only
        traffic = None
        tfile = os.path.join(ROOT, 'profiles', 'r01_hbm_traffic.json')
        if world == 1 and mode == 0 and os.path.exists(tfile):
            with open(tfile) as fh:
                traffic = json.load(fh)['kernels']['gram_bwd_x6_kernel<1, false, 0>']['hbm_bytes']
        flops_eval = sum(4.0 * M * B * B * p + 4.0 * M * M * B * B for p in DIMS)
        out = {
            'metric': 'HE-CKA loss+grad evals/sec (M=64, B=512, 5 layers)', 'value': args.steps / dt, 'unit': 'evals/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms,
            'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'parallelism': 'members sharded %d-way, %s' % (world, SHARD_NOTE[os.environ.get('HECKA_SHARD_MODE', 'slice')]) if world > 1 else 'single GPU',
                       'workload': 'configs[3] synthetic loss-only microbench: M=64, B=512, P=[512,1024,2048,4096,8192], '
                                   'he (s=2, eps=0.0025, arc_eps=0.031), linear kernel, detach_diag, uniform layer weights',
                       'loss': float(loss.item()), 'algorithmic_tflops_per_s': flops_eval / (ms * 1e-3) / 1e12},
            'roofline': dict({'bound': 'mfma', 'kernel': kname, 'achieved': ach, 'peak': peak, 'unit': 'TFLOP/s',
                              'frac': ach / peak, 'avg_launch_ms': kms / max(len(prof['gram_bwd']), 1), 'traffic': traffic}, **hw),
        }
        if not args.no_cpu_baseline and world == 1:      # the CPU baseline is timed at N = 1 only
            out['cpu_baseline'] = cpu_baseline()
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
