#!/bin/bash
# sweep_env.sh VAR v1 v2 ... : the headline bench once per value of an environment switch (same box)
R=${GRAFT_REPO_ROOT:-/root/repo}
VAR=$1; shift
mkdir -p $R/gpurun_out/sweep
for v in "$@"; do
  env $VAR=$v timeout -k 10 200 python $R/bench.py --no-cpu-baseline --no-secondary --no-roofline $SWEEP_ARGS > $R/gpurun_out/sweep/${VAR}_$v.log 2>&1
  echo "$VAR=$v $(grep -h '^{' $R/gpurun_out/sweep/${VAR}_$v.log | sed -e 's/.*"value": \([0-9.]*\).*/\1/')"
done
