#!/bin/bash
# HBM traffic of the conv micro-benchmark: rocprofv3 FETCH_SIZE / WRITE_SIZE in separate passes (MI355X_MICROARCH.md, HBM section)
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/$1; LIB=$2
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/pmc_$c
  MISEG_HIP_LIB=$LIB timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -o p -- python3 $R/scripts/bench_conv.py fwd 2 > $OUT/pmc_$c.log 2>&1
  rm -f $OUT/pmc_$c/*kernel_trace.csv $OUT/pmc_$c/*/*kernel_trace.csv
done
python3 - <<PY
import csv, glob, collections
for c, mul in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
    f = glob.glob("$OUT/pmc_%s/**/*counter_collection.csv" % c, recursive=True)
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == c and "conv3_fwd96" in r["Kernel_Name"]:
            k = (r["Grid_Size"] if "Grid_Size" in r else "", r["Kernel_Name"][:60])
            acc[k][0] += 1; acc[k][1] += float(r["Counter_Value"])
    for k, (n, v) in acc.items():
        print(c, k, "launches", n, "MB/launch", round(mul * 1024 * v / n / 1e6, 1))
PY
