#!/bin/bash
# kernel stats of the optimisation-step bench (fused conv-weight optimiser on / off): prof_train.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for v in fused plain; do
  rm -rf $R/gpurun_out/prof_train_$v
  if [ $v = plain ]; then export MISEG_NO_OPT_PACK=1; else unset MISEG_NO_OPT_PACK; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_train_$v -o s -- python3 $R/bench.py --train-step --no-cpu-baseline --no-roofline --no-secondary > $R/gpurun_out/prof_train_$v.log 2>&1
  S=$(find $R/gpurun_out/prof_train_$v -name '*kernel_stats.csv' | head -1)
  echo "== $v"; grep -E "opt_|pack_conv3|param_cast|seg_loss" $S | awk -F, '{printf "%-60s calls %s avg %.1f us\n", substr($1,1,60), $2, $4/1000}'
  rm -f $(find $R/gpurun_out/prof_train_$v -name '*kernel_trace.csv')
done
