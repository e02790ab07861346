"""HE-CKA loss+grad microbenchmark (BASELINE.json configs[3]: M=64 members, B=512 samples, 5 tracked layers
with feature dims 512..8192, fp32, synthetic features).

One "step" = one eval = scalar `model_layers` HE loss AND dLoss/dX for all five layers, through the
drop-in `hyper_amd` model kernel (autograd.Function -> ctypes -> libhecka HIP kernels).  Inputs are
resident in HBM before the timed region.

    python bench.py --gpus 1 --steps 100 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line (rank 0).  The run checks itself: the loss and every layer's gradient of the timed workload are
compared with the oracle evaluated in fp64 on the same tensors (`parity`), and the process exits with status 3 when
either is off by more than rel 1e-5 (the JSON line is still printed).  At N > 1 every rank checks its member slice: against
the unsharded hyper_amd evaluation of all 64 members run (untimed) on its own GPU, and rank 0 also against the fp64 oracle;
the line then carries `rccl_ranks`, `collectives_per_eval`, `exchanged_bytes_per_rank_per_eval` and
`exposed_wait_ms_per_eval` (how long the compute stream stood waiting for exchanged data, from HIP events around the waits)."""
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
FK = {'name': 'feature', 'detach_diag': True, 'kernel': 'linear'}
CFG = {'name': 'model_layers',      # configs/mnist/ens/he-lin.yml:28-35 head, uniform weights
       'layer_kernel': {'name': 'he', 'he_s': 2.0, 'eps': 0.0025, 'arc_eps': 0.031, 'feature_kernel': FK},
       'layer_weighting': {'name': 'uniform'}}
CFG_EXP = {'name': 'model_layers',  # configs/cifar10/ens/he-lin.yml:13-21 head
           'layer_kernel': {'name': 'he', 'he_s': 2.0, 'eps': 0.001, 'arc_eps': 0.001, 'use_exp': True, 'feature_kernel': FK},
           'layer_weighting': {'name': 'uniform'}}
PEAK_F32_MATRIX_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MATRIX_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 / fp16 MFMA
PEAK_HBM_GBS = 8000.0
PARITY_RTOL = 1e-5               # BASELINE.json north_star: rel 1e-5 (fp32) on loss and gradients


def make_feats(dev, members, correlated=False):
    """SURVEY.md 8d synthetic inputs: randn features, or the correlated set 0.7 S_l + 0.3 X_l (cos ~ 0.7 - 0.9)."""
    torch.manual_seed(1234)
    feats = []
    for p in DIMS:
        x = torch.randn(M, B, p)          # same global tensor on every rank, then the rank's member slice
        if correlated:
            x = 0.7 * torch.randn(1, B, p) + 0.3 * x
        feats.append(x[members].contiguous().to(dev))
    return feats


def cpu_baseline():
    """The oracle (reference op sequence, stock PyTorch autograd) on this box's host cores,
    one full eval of the same workload.  Returns (record, loss, grads) of the untimed first eval."""
    from oracle import hecka_ref as R
    torch.manual_seed(1234)
    feats = [torch.randn(M, B, p) for p in DIMS]
    lk = CFG['layer_kernel']
    cfg = dict(kernel='linear', detach_diag=True, he_s=lk['he_s'], eps=lk['eps'], arc_eps=lk['arc_eps'])
    w = R.layer_weights('uniform', len(DIMS))
    loss0, grads0 = R.loss_and_grads(feats, cfg, w)   # untimed first eval (allocator, thread pool); kept for `parity`
    n, t0 = 0, time.perf_counter()
    while n < 12 and (n == 0 or time.perf_counter() - t0 < 12.0):   # a bounded sample: about 12 s of CPU work
        R.loss_and_grads(feats, cfg, w)
        n += 1
    dt = time.perf_counter() - t0
    rec = {'value': n / dt, 'unit': 'evals/s', 'cores': torch.get_num_threads(), 'kind': 'port',
           'sample': '%d full evals (M=64,B=512,P=512..8192 x5 layers, fp32, torch %s CPU autograd) in %.1f s after one '
                     'untimed eval' % (n, torch.__version__, dt)}
    return rec, loss0, grads0


def oracle_fp64_on_gpu(cfg, xs):
    """The oracle (oracle/hecka_ref.py, the reference's op sequence in stock torch) in fp64 on the bench tensors, on the
    GPU (on the host cores the fp64 eval takes minutes).  Checker only - never timed, never part of `value`."""
    from oracle import hecka_ref as R
    xo = [x.detach().double().requires_grad_(True) for x in xs]
    ref = R.run_config(cfg, xo)
    ref.backward()
    out = (float(ref.item()), [o.grad for o in xo])
    del ref
    return out


def parity_record(loss, grads, ref_loss, ref_grads):
    lr = abs(loss - ref_loss) / max(abs(ref_loss), 1e-300)
    gr = [float((g.double() - r.to(g.device).double()).abs().max() / r.abs().max()) for g, r in zip(grads, ref_grads)]
    return {'loss_rel': lr, 'grad_rel_max': max(gr), 'grad_rel_per_layer': gr}


SHARD_NOTE = {'slice': 'all-to-all of Gram column slices (1/G of the exchange and of every stage per rank)',
              'gather': 'all_gather of centred Grams'}


def train_rates(budget_s):
    """Train steps/s of the model configurations BASELINE.json's metric also names (tools/train_bench.py: synthetic data
    of the dataset's shape, forward + HE-CKA method loss + backward + optimizer), a few steps each."""
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import train_bench
    out, t0 = {}, time.perf_counter()
    for key, name, kw in (('mnist_lenet_m5_b300', 'mnist', dict(graph=True, steps=60)),
                          ('toy_m30_b100', 'toy', dict(graph=True, steps=60)),
                          ('cifar_rn18_m10_b128', 'cifar', dict(steps=20, warmup=5)),
                          # the same step with the layer GEMMs in the 16-bit COMPUTE mode (ops.layer_products(1): operands rounded to
                          # fp16 after their power-of-two scale, fp32 accumulation and storage; the loss path untouched) - an extra
                          # line, NOT the fp32 step above and not what the reference computes (it is fp32-only)
                          ('cifar_rn18_m10_b128_16bit_compute', 'cifar', dict(steps=20, warmup=5, products=1)),
                          ('mnist_hyper_m128', 'hyper', dict(steps=10))):
        if time.perf_counter() - t0 > budget_s:
            out[key] = None
            continue
        try:
            kw = dict(kw)
            products = kw.pop('products', 3)
            from hyper_amd import ops
            with ops.layer_products(products):
                r = train_bench.run(name, **dict(dict(warmup=3), **kw))
            out[key] = {'steps_per_s': r['steps_per_s'], 'ms_per_step': r['ms_per_step'],
                        'mode': ('one HIP graph per step' if kw.get('graph') else 'eager') +
                                (', layer GEMMs with one fp16 product per multiply' if products == 1 else '')}
        except Exception as e:      # the headline line must survive a failure here
            out[key] = {'error': '%s: %s' % (type(e).__name__, str(e)[:200])}
        torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='skip the variant workloads and the train-step rates')
    ap.add_argument('--no-parity', action='store_true', help='skip the fp64 self-check (profiling runs)')
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

    from hyper_amd import ops
    from hyper_amd.diversity.model_kernels import build_model_kernel
    ops.set_gemm_mode(int(os.environ.get('HECKA_GEMM_MODE', '0')))   # 0 auto (f16x3) | 2 bf16x6 | 1 fp32 MFMA

    if M % world:
        raise SystemExit('M=%d members do not shard over %d GPUs' % (M, world))
    members = slice(rank * (M // world), (rank + 1) * (M // world))   # this rank's ensemble members
    sharded = world > 1 or force_sharded

    def workload(cfg, correlated):
        feats = make_feats(dev, members, correlated)
        kern = build_model_kernel(cfg).to(dev)
        if sharded:
            from hyper_amd.parallel import ShardedModelLayersKernel
            kern = ShardedModelLayersKernel(kern)    # all-to-all of Gram column slices over RCCL (hyper_amd/parallel.py)
        xs = [f.requires_grad_(True) for f in feats]
        fdict = OrderedDict((str(i), x) for i, x in enumerate(xs))

        def step():
            for x in xs:
                x.grad = None
            loss = kern(None, fdict, None, OrderedDict())
            loss.backward()
            return loss
        return xs, step

    tok = torch.zeros(1, device=dev)

    def barrier():
        # an all-reduce of one element on the compute stream + a device sync (dist.barrier() on RCCL costs tens of
        # milliseconds per call, which would sit inside the timed region)
        if dist is not None:
            dist.all_reduce(tok)
        torch.cuda.synchronize()

    def timed(step, steps, warmup, profile=False):
        """(wall seconds for `steps` evals, per-eval device-event milliseconds, last loss, profile records)."""
        for _ in range(warmup):
            step()
        if profile:
            ops.PROFILE = {'gram_bwd': []}       # HIP events around the dominant kernel, on the launch stream
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        import gc
        gc.collect()        # a full collection of the interpreter (torch imported: about 40 ms, the host thread stops issuing)
        gc.disable()        # would otherwise land in the timed region once in a few runs; nothing below makes cycles
        barrier()
        t0 = time.perf_counter()
        evs[0].record()
        for i in range(steps):
            loss = step()
            evs[i + 1].record()
        barrier()
        dt = time.perf_counter() - t0
        gc.enable()
        prof, ops.PROFILE = ops.PROFILE, None
        raw = [evs[i].elapsed_time(evs[i + 1]) for i in range(steps)]
        if os.environ.get('HECKA_BENCH_DEBUG'):
            print('slowest eval: #%d of %d, %.2f ms' % (max(range(steps), key=lambda i: raw[i]), steps, max(raw)), file=sys.stderr)
        per = sorted(raw)
        if dist is not None:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, per, loss, prof

    # the first collective of a process group sets up its channels and RCCL goes on initialising in the background for
    # some tens of milliseconds (seen as one sporadic 40 - 50 ms stall of the host thread): do it now, long before t0
    barrier()
    xs, step = workload(CFG, False)
    dt, per, loss, prof = timed(step, args.steps, args.warmup, profile=True)
    loss_val = float(loss.item())
    grads = [x.grad.detach().clone() for x in xs]
    sent_per_eval = coll_per_eval = 0
    wait_ms = None
    if sharded:      # bytes this rank sends per evaluation (hyper_amd.parallel counts what it hands to the collectives)
        from hyper_amd import parallel
        b0, c0 = parallel.EXCHANGED_BYTES, parallel.COLLECTIVES
        step()
        sent_per_eval, coll_per_eval = parallel.EXCHANGED_BYTES - b0, parallel.COLLECTIVES - c0
        # exposed exchange time: HIP events on the compute stream around every wait for exchanged data, a few untimed evals
        parallel.PROFILE = []
        nprobe = 5
        for _ in range(nprobe):
            step()
        torch.cuda.synchronize()
        recs, parallel.PROFILE = parallel.PROFILE, None
        wait_ms = sum(a.elapsed_time(b) for a, b in recs) / nprobe

    parity, bad = None, False
    if sharded and not args.no_parity:
        # every rank: its slice of the sharded result against the UNSHARDED evaluation of all members on its own GPU
        full = make_feats(dev, slice(None))
        fx = [f.requires_grad_(True) for f in full]
        floss = build_model_kernel(CFG).to(dev)(None, OrderedDict((str(i), x) for i, x in enumerate(fx)), None, OrderedDict())
        floss.backward()
        pr = parity_record(loss_val, grads, float(floss.item()), [x.grad[members] for x in fx])
        worst = torch.tensor([pr['loss_rel'], pr['grad_rel_max']], device=dev, dtype=torch.float64)
        if dist is not None and world > 1:
            dist.all_reduce(worst, op=dist.ReduceOp.MAX)
        parity = {'vs_unsharded_on_every_rank': {'loss_rel': float(worst[0]), 'grad_rel_max': float(worst[1])}, 'rtol': PARITY_RTOL}
        bad = not (float(worst[0]) <= PARITY_RTOL and float(worst[1]) <= PARITY_RTOL)
        for x in fx:
            x.grad = None
        if rank == 0:   # ... and rank 0 against the fp64 oracle on all members
            ref_loss, ref_grads = oracle_fp64_on_gpu(CFG, fx)
            po = parity_record(loss_val, grads, ref_loss, [g[members] for g in ref_grads])
            parity.update(po, oracle='oracle/hecka_ref.py in fp64 on all 64 members, on rank 0\'s GPU; rank 0\'s member slice compared')
            bad = bad or not (po['loss_rel'] <= PARITY_RTOL and po['grad_rel_max'] <= PARITY_RTOL)
            del ref_grads
        del full, fx, floss
        torch.cuda.empty_cache()
        if dist is not None and world > 1:     # every rank leaves with the same verdict
            flag = torch.tensor([1.0 if bad else 0.0], device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            bad = bool(flag.item() > 0)
    if world == 1 and not sharded and not args.no_parity:
        ref_loss, ref_grads = oracle_fp64_on_gpu(CFG, xs)
        parity = dict(parity_record(loss_val, grads, ref_loss, ref_grads), rtol=PARITY_RTOL,
                      oracle='oracle/hecka_ref.py (reference op sequence, torch autograd) in fp64 on the same tensors, on the GPU')
        bad = not (parity['loss_rel'] <= PARITY_RTOL and parity['grad_rel_max'] <= PARITY_RTOL)
        del ref_grads
        torch.cuda.empty_cache()

    variants = None
    if world == 1 and not args.no_extras:
        variants = {}
        for name, cfg, corr in (('correlated_power', CFG, True), ('randn_use_exp', CFG_EXP, False),
                                ('correlated_use_exp', CFG_EXP, True)):
            vx, vstep = workload(cfg, corr)
            vdt, vper, vloss, _ = timed(vstep, 20, 3)
            rec = {'evals_per_s': 20 / vdt, 'median_ms_device': vper[len(vper) // 2], 'loss': float(vloss.item())}
            if not args.no_parity:
                rl, rg = oracle_fp64_on_gpu(cfg, vx)
                pr = parity_record(rec['loss'], [x.grad for x in vx], rl, rg)
                rec.update(loss_rel=pr['loss_rel'], grad_rel_max=pr['grad_rel_max'])
                bad = bad or not (pr['loss_rel'] <= PARITY_RTOL and pr['grad_rel_max'] <= PARITY_RTOL)
                del rg
            variants[name] = rec
            del vx, vstep
            torch.cuda.empty_cache()

    if rank == 0:
        ms = 1e3 * dt / args.steps
        # dominant kernel: dX = W X (gram_bwd_kernel), 2*M*B*B*P flop per launch (SURVEY.md 8d per-unit figure)
        kms = sum(a.elapsed_time(b) for a, b, _ in prof['gram_bwd'])
        kfl = sum(fl for _, _, fl in prof['gram_bwd'])      # rank 0's launches (M/world members each)
        ach = kfl / (kms * 1e-3) / 1e12 if kms > 0 else 0.0
        mode = ops.get_gemm_mode()
        if mode == 0:
            # interior shapes, maxima supplied: every fp32 multiply is THREE fp16 MFMA products (gemm_bf16x6.hpp,
            # scheme f16x3), so the MFMA roof for ALGORITHMIC fp32 flops is the dense 16-bit peak / 3
            kname = ops.LAST_BWD_KERNEL or 'gram_bwd_x6_kernel'
            kname += ' (dX = W X; fp32 operands split into 2 fp16 terms, 3 fp16 MFMA products per multiply)'
            peak, hw = PEAK_BF16_MATRIX_TFLOPS / 3.0, {'hw_f16_tflops': ach * 3.0, 'hw_f16_peak': PEAK_BF16_MATRIX_TFLOPS}
        elif mode == 2:
            kname = 'gram_bwd_x6_kernel<0> (dX = W X; fp32 operands split into 3 bf16 terms, 6 bf16 MFMA products per multiply)'
            peak, hw = PEAK_BF16_MATRIX_TFLOPS / 6.0, {'hw_bf16_tflops': ach * 6.0, 'hw_bf16_peak': PEAK_BF16_MATRIX_TFLOPS}
        else:
            kname, peak, hw = 'gram_bwd_kernel (dX = W X, fp32 MFMA)', PEAK_F32_MATRIX_TFLOPS, {}
        # HBM bytes per launch of the dominant kernel: not measurable from inside a run - taken from the committed
        # rocprofv3 PMC passes of this same command (profiles/r03_hbm_traffic.json says how), single-GPU f16x3 only
        traffic = None
        tfile = os.path.join(ROOT, 'profiles', 'r03_hbm_traffic.json')
        if world == 1 and mode == 0 and os.path.exists(tfile):
            with open(tfile) as fh:
                tk = json.load(fh)['kernels']
            traffic = next((v['hbm_bytes'] for k, v in tk.items() if k.startswith('gram_bwd')), None)
        flops_eval = sum(4.0 * M * B * B * p + 4.0 * M * M * B * B for p in DIMS)
        out = {
            'metric': 'HE-CKA loss+grad evals/sec (M=64, B=512, 5 layers)', 'value': args.steps / dt, 'unit': 'evals/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms,
            'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'parallelism': 'members sharded %d-way, %s' % (world, SHARD_NOTE[os.environ.get('HECKA_SHARD_MODE', 'slice')]) if world > 1 else 'single GPU',
                       'workload': 'configs[3] synthetic loss-only microbench: M=64, B=512, P=[512,1024,2048,4096,8192], '
                                   'he (s=2, eps=0.0025, arc_eps=0.031), linear kernel, detach_diag, uniform layer weights',
                       'loss': loss_val, 'algorithmic_tflops_per_s': flops_eval / (ms * 1e-3) / 1e12},
            'median_ms_device': per[len(per) // 2], 'min_ms_device': per[0], 'p90_ms_device': per[(9 * len(per)) // 10],
            'max_ms_device': per[-1],
            'roofline': dict({'bound': 'mfma', 'kernel': kname, 'achieved': ach, 'peak': peak, 'unit': 'TFLOP/s',
                              'frac': ach / peak, 'avg_launch_ms': kms / max(len(prof['gram_bwd']), 1), 'traffic': traffic}, **hw),
        }
        if sharded:
            out['per_rank_ms_per_step'] = ms      # max over ranks (each rank runs M/world members)
            out['rccl_ranks'] = world if os.environ.get('HECKA_BENCH_BACKEND', 'nccl') == 'nccl' else 0
            out['collectives_per_eval'] = coll_per_eval
            out['exchanged_bytes_per_rank_per_eval'] = sent_per_eval
            out['exposed_wait_ms_per_eval'] = wait_ms
        if parity is not None:
            out['parity'] = parity
        if not args.no_cpu_baseline and world == 1:      # the CPU baseline is timed at N = 1 only
            rec, closs, cgrads = cpu_baseline()
            out['cpu_baseline'] = rec
            if parity is not None:   # the HIP path against the oracle's own fp32 CPU run (reported, not gated: both are fp32)
                pr = parity_record(loss_val, grads, float(closs), cgrads)
                parity['vs_cpu_fp32_oracle'] = {'loss_rel': pr['loss_rel'], 'grad_rel_max': pr['grad_rel_max']}
            del cgrads
        if variants is not None:
            out['variants'] = variants
        if world == 1 and not args.no_extras:
            del xs, grads
            torch.cuda.empty_cache()
            out['train_steps_per_s'] = train_rates(60.0)
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if dist is not None:
        dist.destroy_process_group()
    if bad:
        print('bench.py: parity check FAILED (rel > %g): %s %s' % (PARITY_RTOL, parity, variants), file=sys.stderr)
        sys.exit(3)


if __name__ == '__main__':
    main()
