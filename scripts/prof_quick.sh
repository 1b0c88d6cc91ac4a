#!/bin/bash
# rocprofv3 kernel stats of a short bench run -> gpurun_out/quick/ (top kernels printed)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/quick
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/quick -o q -- python3 $R/bench.py --no-cpu-baseline --no-roofline "$@" > $R/gpurun_out/quick.log 2>&1
rm -f $R/gpurun_out/quick/q_kernel_trace.csv
python3 - <<'PY'
import csv, re, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
rows = list(csv.DictReader(open(R + "/gpurun_out/quick/q_kernel_stats.csv")))
def short(name):
    m = re.match(r"_ZN5miseg(\d+)", name)
    if m: return name[m.end():m.end() + int(m.group(1))]
    name = re.sub(r"^void ", "", name); name = re.sub(r"^miseg::", "", name)
    return re.split(r"[<(]", name)[0]
agg = {}
for r in rows:
    a = agg.setdefault(short(r["Name"]), [0, 0]); a[0] += int(r["Calls"]); a[1] += int(r["TotalDurationNs"])
tot = sum(v[1] for v in agg.values())
steps = max(1, agg.get("pack_conv3_batch_kernel", [1])[0])
print(f"all kernels: {tot/1e6:.2f} ms over ~{steps} steps = {tot/1e6/steps:.3f} ms per step (incl. capture warm-ups and checks)")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{k:40s} calls {v[0]:6d} total {v[1]/1e6:8.2f} ms  avg {v[1]/v[0]/1e3:7.1f} us {100*v[1]/tot:5.1f}%")
PY
