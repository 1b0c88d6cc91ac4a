#!/bin/bash
# run on the GPU box (through gpurun): the three rocprofv3 passes scripts/make_profiles.py reads
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_stats $R/gpurun_out/prof_fetch $R/gpurun_out/prof_write
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -o s -- python3 $R/bench.py > $R/gpurun_out/prof_stats.log 2>&1
rm -f $R/gpurun_out/prof_stats/*/*kernel_trace.csv $R/gpurun_out/prof_stats/*kernel_trace.csv
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch -o f -- python3 $R/bench.py --no-graph --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $R/gpurun_out/prof_fetch.log 2>&1
rm -f $R/gpurun_out/prof_fetch/*/*kernel_trace.csv $R/gpurun_out/prof_fetch/*kernel_trace.csv
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_write -o w -- python3 $R/bench.py --no-graph --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $R/gpurun_out/prof_write.log 2>&1
rm -f $R/gpurun_out/prof_write/*/*kernel_trace.csv $R/gpurun_out/prof_write/*kernel_trace.csv
timeout -k 10 200 python3 $R/bench.py > $R/gpurun_out/bench_official.log 2>&1
tail -1 $R/gpurun_out/bench_official.log | cut -c1-400
