#!/bin/bash
# run on the GPU box (through gpurun): the three rocprofv3 passes scripts/make_profiles.py reads, then the bench lines of the round
# (official C2 line, fp32 line, configs[2] and configs[4] lines, C3 kernel stats).  Counters never share a run with traces other than the kernel trace.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_stats $R/gpurun_out/prof_fetch $R/gpurun_out/prof_write $R/gpurun_out/prof_c3
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -o s -- python3 $R/bench.py --no-cpu-baseline --no-secondary > $R/gpurun_out/prof_stats.log 2>&1
rm -f $R/gpurun_out/prof_stats/*/*kernel_trace.csv $R/gpurun_out/prof_stats/*kernel_trace.csv
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch -o f -- python3 $R/bench.py --no-graph --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary > $R/gpurun_out/prof_fetch.log 2>&1
rm -f $R/gpurun_out/prof_fetch/*/*kernel_trace.csv $R/gpurun_out/prof_fetch/*kernel_trace.csv
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_write -o w -- python3 $R/bench.py --no-graph --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary > $R/gpurun_out/prof_write.log 2>&1
rm -f $R/gpurun_out/prof_write/*/*kernel_trace.csv $R/gpurun_out/prof_write/*kernel_trace.csv
# matrix-core utilisation (north_star: "choices evidenced by rocprof HBM GB/s and MFMA utilisation"): SQ / GRBM counters only, in a pass of their own
rm -rf $R/gpurun_out/prof_mfma
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/prof_mfma -o m -- python3 $R/bench.py --no-graph --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary > $R/gpurun_out/prof_mfma.log 2>&1
rm -f $R/gpurun_out/prof_mfma/*/*kernel_trace.csv $R/gpurun_out/prof_mfma/*kernel_trace.csv
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c3 -o s -- python3 $R/bench.py --workload c3 --no-cpu-baseline --no-roofline > $R/gpurun_out/prof_c3.log 2>&1
rm -f $R/gpurun_out/prof_c3/*/*kernel_trace.csv $R/gpurun_out/prof_c3/*kernel_trace.csv
cd $R
timeout -k 10 400 python3 bench.py > gpurun_out/bench_official.log 2>&1
timeout -k 10 300 python3 bench.py --dtype f32 --no-cpu-baseline > gpurun_out/bench_f32.log 2>&1
timeout -k 10 400 python3 bench.py --workload c3 > gpurun_out/bench_c3.log 2>&1
timeout -k 10 300 python3 bench.py --workload c5 --steps 3 --warmup 1 > gpurun_out/bench_c5.log 2>&1
for f in official f32 c3 c5; do grep "^{" gpurun_out/bench_$f.log | tail -1 | cut -c1-260; done
