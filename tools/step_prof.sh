#!/bin/bash
# One GPU call: rocprofv3 kernel stats of a few training steps of one model configuration.  bash tools/step_prof.sh CONFIG TAG [extra train_bench args]
CFG=${1:-cifar}; TAG=${2:-step}; shift; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 python $R/tools/train_bench.py $CFG --steps 10 --warmup 3 "$@" > $O/${TAG}.txt 2>/dev/null || exit 1
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof -o s -- python3 $R/tools/train_bench.py $CFG --steps 10 --warmup 3 "$@" > $O/${TAG}_prof.log 2>&1 || exit 1
find $O/${TAG}_prof -name "*kernel_trace.csv" -delete
cat $O/${TAG}.txt
