"""Which torch OPS launch the small kernels of a training step: torch.profiler over a few steps of tools/train_bench.py's
configuration, grouped by op name (self device time, calls per step).  python tools/op_prof.py cifar [steps]"""
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'he-cka-ensembles_amd'), os.path.join(ROOT, 'tools')]
import torch  # noqa: E402
from torch.profiler import profile, ProfilerActivity  # noqa: E402
import train_bench  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'cifar'
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    train_bench.run(name, steps=2, warmup=2)
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=bool(os.environ.get('OP_PROF_STACK'))) as prof:
        train_bench.run(name, steps=steps, warmup=0)
    if os.environ.get('OP_PROF_OP'):      # one op, by input shapes (and by caller with OP_PROF_STACK=1)
        want = os.environ['OP_PROF_OP']
        for e in sorted(prof.key_averages(group_by_input_shape=True, group_by_stack_n=6 if os.environ.get('OP_PROF_STACK') else 0),
                        key=lambda e: -getattr(e, 'self_device_time_total', 0)):
            if e.key == want and getattr(e, 'self_device_time_total', 0) > 0:
                print('%7.1f calls/step %8.3f ms/step  %s' % (e.count / steps, e.self_device_time_total / 1e3 / steps, str(e.input_shapes)[:150]))
                for fr in (e.stack or [])[:6]:
                    print('        ', fr[:150])
        return
    rows = []
    for e in prof.key_averages():
        dev = getattr(e, 'self_device_time_total', None)
        if dev is None:
            dev = getattr(e, 'self_cuda_time_total', 0)
        if dev > 0:
            rows.append((dev, e.count, e.key))
    rows.sort(reverse=True)
    for dev, cnt, key in rows[:45]:
        print('%-70s calls/step %7.1f   %8.3f ms/step' % (key[:70], cnt / steps, dev / 1e3 / steps))


if __name__ == '__main__':
    main()
