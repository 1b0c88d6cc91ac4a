#!/bin/bash
# kernel stats of the optimisation-step bench (fused conv-weight optimiser on / off): prof_train.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for v in ${VARIANTS:-fused plain}; do
  rm -rf $R/gpurun_out/prof_train_$v
  if [ $v = plain ]; then export MISEG_NO_OPT_PACK=1; else unset MISEG_NO_OPT_PACK; fi
  if [ -f $R/scripts/micro/libmiseg_$v.so ]; then export MISEG_HIP_LIB=$R/scripts/micro/libmiseg_$v.so; else unset MISEG_HIP_LIB; fi      # a variant named after a scratch build runs that build
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_train_$v -o s -- python3 $R/bench.py --train-step --no-cpu-baseline --no-roofline --no-secondary > $R/gpurun_out/prof_train_$v.log 2>&1
  S=$(find $R/gpurun_out/prof_train_$v -name '*kernel_stats.csv' | head -1)
  echo "== $v"; python3 $R/scripts/debug/prof_train_rows.py $S; grep -h "^{" $R/gpurun_out/prof_train_$v.log | sed -e 's/.*"value": \([0-9.]*\).*/train step \1 patches\/s/'
  rm -f $(find $R/gpurun_out/prof_train_$v -name '*kernel_trace.csv')
done
