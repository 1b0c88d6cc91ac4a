#!/bin/bash
# run on the GPU box: kernel trace of the default bench run (hipGraph), then the issue-order timeline of one replay -> gpurun_out/<tag>_timeline.txt
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-step}
shift || true
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/trace_$TAG
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/trace_$TAG -o t -- python3 $R/bench.py --no-cpu-baseline --no-roofline --no-secondary --steps 10 --warmup 3 "$@" > $R/gpurun_out/trace_$TAG.log 2>&1
cd $R
T=$(find gpurun_out/trace_$TAG -name '*kernel_trace.csv' | head -1)
S=$(find gpurun_out/trace_$TAG -name '*kernel_stats.csv' | head -1)
python3 scripts/step_timeline.py $T --period --all > gpurun_out/${TAG}_timeline.txt 2>&1 || python3 scripts/step_timeline.py $T --all > gpurun_out/${TAG}_timeline.txt 2>&1
python3 scripts/kernel_stats_top.py $S 60 > gpurun_out/${TAG}_stats_top.txt
cp $S gpurun_out/${TAG}_kernel_stats.csv
gzip -c $T > gpurun_out/${TAG}_kernel_trace.csv.gz
rm -f $T
grep "^{" gpurun_out/trace_$TAG.log | tail -1 | cut -c1-300
