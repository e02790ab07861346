#!/bin/bash
# One GPU call: per-rank compute of the member-sharded evaluation (tools/shard_emulate.py) and the kernel table at G = 8.
TAG=${1:-shard}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python $R/tools/shard_emulate.py --modes slice --worlds 1,2,4,8 > $O/${TAG}.txt 2>/dev/null || exit 1
timeout -k 10 300 python $R/tools/shard_emulate.py --modes slice --worlds 2,4,8 --no-exchange-copies >> $O/${TAG}.txt 2>/dev/null || exit 1
rm -rf $O/${TAG}_prof
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof -o g8 -- python3 $R/tools/shard_emulate.py --modes slice --worlds 8 --steps 20 > $O/${TAG}_prof.log 2>&1 || exit 1
python3 $R/tools/prof_summary.py $O/${TAG}_prof > $O/${TAG}_g8_kernels.txt
python3 - <<PY >> $O/${TAG}_g8_kernels.txt
import csv, glob
f = glob.glob('$O/${TAG}_prof/**/*_kernel_trace.csv', recursive=True)[0]
rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in csv.DictReader(open(f))))
rows = rows[len(rows) // 2:]          # the timed evaluations (second half of the trace)
busy = sum(e - s for s, e in rows); span = rows[-1][1] - rows[0][0]
print('second half of the trace: %d kernels, GPU busy %.1f%% of the span (%.3f ms busy per 1/20 of it)' % (len(rows), 100.0 * busy / span, busy / 1e6 / 10))
PY
find $O/${TAG}_prof -name "*kernel_trace.csv" -delete
cat $O/${TAG}.txt $O/${TAG}_g8_kernels.txt
