mkdir -p gpurun_out/s3
MISEG_FORK_AT=s1 timeout -k 10 400 python -m pytest tests/test_hip_modules.py -x -q -k "swin_unetr or graphed" > gpurun_out/s3/tests_s1.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/s3/tests_s1.log
for f in e10 s0 s1 s2 e10 s1; do
  MISEG_FORK_AT=$f timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --no-roofline > gpurun_out/s3/bench_$f.log 2>&1
  echo "$f $(grep -h '^{' gpurun_out/s3/bench_$f.log | cut -c75-130)"
done
MISEG_FORK_AT=s1 MISEG_STEP_STAMPS=1 timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --no-roofline > gpurun_out/s3/stamps_s1.log 2>&1
grep -h "step stamps" gpurun_out/s3/stamps_s1.log
