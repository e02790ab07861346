#!/bin/bash
# One GPU call for the member-sharded path: GPU tests of the sharded stages, bench.py's N > 1 code path rehearsed on one
# card (RCCL at world size 1; gloo with 2 and 8 ranks on cuda:0), and the per-rank compute emulation.
# Usage on the GPU box (via gpurun):  bash tools/shard_round.sh <tag>
set -o pipefail
TAG=${1:-r03_shard}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
timeout -k 10 400 python -u -m pytest tests/test_gpu_sharded.py -x -q --timeout 180 --timeout-method thread > $O/${TAG}_tests.log 2>&1 || { tail -30 $O/${TAG}_tests.log; exit 1; }
tail -3 $O/${TAG}_tests.log
FAST="--no-cpu-baseline --no-extras --steps 20 --warmup 5"
HECKA_FORCE_SHARDED=1 timeout -k 10 200 python bench.py $FAST > $O/${TAG}_w1.json 2> $O/${TAG}_w1.err || { tail -20 $O/${TAG}_w1.err; exit 1; }
cut -c1-400 $O/${TAG}_w1.json; python -c "import json,sys; d=json.load(open('$O/${TAG}_w1.json')); print({k: d.get(k) for k in ('parity','collectives_per_eval','exchanged_bytes_per_rank_per_eval','exposed_wait_ms_per_eval','rccl_ranks')})"
for N in 2 8; do
  HECKA_BENCH_BACKEND=gloo HECKA_BENCH_ONE_CARD=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2971$N bench.py --gpus $N --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/${TAG}_gloo$N.json 2> $O/${TAG}_gloo$N.err || { tail -30 $O/${TAG}_gloo$N.err; exit 1; }
  python -c "import json,sys; d=json.load(open('$O/${TAG}_gloo$N.json')); print($N, d['value'], {k: d.get(k) for k in ('parity','collectives_per_eval','exchanged_bytes_per_rank_per_eval','exposed_wait_ms_per_eval','rccl_ranks')})"
done
timeout -k 10 300 python tools/shard_emulate.py --modes slice --worlds 1,2,4,8 > $O/${TAG}_emulate.txt 2> $O/${TAG}_emulate.err || { tail -20 $O/${TAG}_emulate.err; exit 1; }
timeout -k 10 300 python tools/shard_emulate.py --modes slice --worlds 2,4,8 --no-exchange-copies >> $O/${TAG}_emulate.txt 2>> $O/${TAG}_emulate.err || exit 1
cat $O/${TAG}_emulate.txt
