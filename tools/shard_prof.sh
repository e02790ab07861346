#!/bin/bash
# One GPU call: per-rank compute of the member-sharded evaluation (tools/shard_emulate.py) and the kernel table at G = 8.
TAG=${1:-shard}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python $R/tools/shard_emulate.py --modes slice,gather --worlds 1,2,4,8 > $O/${TAG}.txt 2>/dev/null || exit 1
timeout -k 10 300 python $R/tools/shard_emulate.py --modes slice,gather --worlds 2,4,8 --no-exchange-copies >> $O/${TAG}.txt 2>/dev/null || exit 1
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof -o g8 -- python3 $R/tools/shard_emulate.py --modes slice --worlds 8 --steps 20 > $O/${TAG}_prof.log 2>&1 || exit 1
find $O/${TAG}_prof -name "*kernel_trace.csv" -delete
cat $O/${TAG}.txt
