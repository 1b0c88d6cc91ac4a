#!/bin/bash
# same-box comparison of several builds of the library: abc_lib.sh lib1 lib2 ... (three rounds)
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/abc
for i in 1 2 3; do
  for L in "$@"; do
    MISEG_HIP_LIB=$L timeout -k 10 200 python $R/bench.py --no-cpu-baseline --no-secondary --no-roofline > $R/gpurun_out/abc/run.log 2>&1
    echo "$(basename $L) $(grep -h '^{' $R/gpurun_out/abc/run.log | sed -e 's/.*"value": \([0-9.]*\).*/\1/')"
  done
done
