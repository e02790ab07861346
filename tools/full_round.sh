#!/bin/bash
# One GPU call: the whole GPU suite, then (only if green) the CIFAR step profile.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
timeout -k 10 1000 python -u -m pytest tests -x -q -m gpu --timeout 120 --timeout-method thread > $O/full_tests.log 2>&1 || { tail -40 $O/full_tests.log; exit 1; }
tail -3 $O/full_tests.log
bash tools/step_prof.sh cifar cifar_r03
