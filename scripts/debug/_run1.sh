cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_hip_kernels.py -q -x -m gpu -k "slabs" > gpurun_out/t_slabs.log 2>&1; tail -n 2 gpurun_out/t_slabs.log
for i in 1 2 3; do timeout -k 10 300 python bench.py --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | grep "^{" | cut -c68-100; done
