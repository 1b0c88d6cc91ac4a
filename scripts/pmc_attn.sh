#!/bin/bash
# SQ counters of the window-attention kernels on the stage shapes (one pass, counters only beside the kernel trace)
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/pmc_attn
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/p
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/p -o a -- python3 $R/scripts/bench_attn.py > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/p/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f[0])):
    if "winattn" in r["Kernel_Name"]:
        k = (r["Kernel_Name"][:48], r["Grid_Size"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[k] += 1
for k, v in acc.items():
    wc = v["SQ_WAVE_CYCLES"]
    print(k, "launches", cnt[k], {n: round(x / wc, 3) for n, x in v.items() if n != "SQ_WAVE_CYCLES"}, "wave_cycles/launch", round(wc / cnt[k]))
PY
