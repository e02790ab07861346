#!/bin/bash
# One GPU call: hypernetwork train-step rate (eager, graphed) and its rocprofv3 kernel stats.  bash tools/hyper_prof.sh TAG
TAG=${1:-hyper}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 python $R/tools/train_bench.py hyper --steps 30 --warmup 5 > $O/${TAG}.txt 2>/dev/null || exit 1
timeout -k 10 200 python $R/tools/train_bench.py hyper --steps 30 --warmup 5 --graph >> $O/${TAG}.txt 2>/dev/null || exit 1
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof -o hyper -- python3 $R/tools/train_bench.py hyper --steps 20 --warmup 3 > $O/${TAG}_prof.log 2>&1 || exit 1
find $O/${TAG}_prof -name "*kernel_trace.csv" -delete
cat $O/${TAG}.txt
