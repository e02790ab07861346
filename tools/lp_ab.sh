#!/bin/bash
# One GPU call: conv tests, then the CIFAR ResNet18 x 10 step (three products, one product), then a profile.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_knobs.py -x -q -m gpu --timeout 120 --timeout-method thread -k "conv or single_product or resnet or train_step or tapmajor or batchnorm or norm" > $O/lp_tests.log 2>&1 || { tail -30 $O/lp_tests.log; exit 1; }
tail -3 $O/lp_tests.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 python $R/tools/train_bench.py cifar --steps 20 --warmup 5 > $O/lp3.txt 2>/dev/null || exit 1
HECKA_LAYER_PRODUCTS=1 timeout -k 10 200 python $R/tools/train_bench.py cifar --steps 20 --warmup 5 > $O/lp1.txt 2>/dev/null || exit 1
cat $O/lp3.txt $O/lp1.txt
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/lp3_prof -o s -- python3 $R/tools/train_bench.py cifar --steps 10 --warmup 3 > $O/lp3_prof.log 2>&1 || exit 1
find $O/lp3_prof -name "*kernel_trace.csv" -delete
