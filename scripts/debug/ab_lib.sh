#!/bin/bash
# same-box A/B of two builds of the library: ab_lib.sh <old .so> [bench.py args...]; prints patches/s of old / new / old / new
R=${GRAFT_REPO_ROOT:-/root/repo}
OLD=$1; shift
mkdir -p $R/gpurun_out/ab
for i in 1 2; do
  MISEG_HIP_LIB=$OLD timeout -k 10 200 python $R/bench.py --no-cpu-baseline --no-secondary --no-roofline "$@" > $R/gpurun_out/ab/old_$i.log 2>&1
  echo "old $(grep -h '^{' $R/gpurun_out/ab/old_$i.log | sed -e 's/.*"value": \([0-9.]*\).*/\1/')"
  timeout -k 10 200 python $R/bench.py --no-cpu-baseline --no-secondary --no-roofline "$@" > $R/gpurun_out/ab/new_$i.log 2>&1
  echo "new $(grep -h '^{' $R/gpurun_out/ab/new_$i.log | sed -e 's/.*"value": \([0-9.]*\).*/\1/')"
done
