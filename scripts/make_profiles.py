"""Turn the rocprofv3 outputs of one round into the tracked files under profiles/.

On the GPU box (three separate passes; counters never share a run with traces other than the kernel trace):
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -o s -- python3 $R/bench.py
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch -o f -- python3 $R/bench.py --no-graph --steps 3 --warmup 1 --no-cpu-baseline --no-roofline
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_write -o w -- python3 $R/bench.py --no-graph --steps 3 --warmup 1 --no-cpu-baseline --no-roofline
Then here:  python scripts/make_profiles.py r01_e
"""
import csv, glob, json, os, re, shutil, sys
from collections import defaultdict

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
out = os.path.join(root, "profiles")


def one(pattern):
    files = glob.glob(os.path.join(root, "gpurun_out", pattern), recursive=True)
    if not files:
        raise SystemExit(f"missing {pattern}")
    return max(files, key=os.path.getmtime)


def short(name):
    m = re.match(r"_ZN5miseg(\d+)", name)
    if m:                                            # rocprofv3 leaves some template instances mangled
        return name[m.end():m.end() + int(m.group(1))]
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"^miseg::", "", name)
    return re.split(r"[<(]", name)[0]


def counter(path, which):
    acc = defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == which:
                a = acc[short(r["Kernel_Name"])]
                a[0] += 1
                a[1] += float(r["Counter_Value"])
    return acc


shutil.copy(one("prof_stats/**/*kernel_stats.csv"), os.path.join(out, f"{tag}_bench_bf16_kernel_stats.csv"))
fetch = counter(one("prof_fetch/**/*counter_collection.csv"), "FETCH_SIZE")
write = counter(one("prof_write/**/*counter_collection.csv"), "WRITE_SIZE")
kernels = {}
for k in sorted(set(fetch) & set(write)):
    if not k.startswith(("conv", "gemm", "winattn", "instnorm", "head", "patch", "colsum", "pack", "param", "ncdhw", "upcat", "copy", "fill")):
        continue
    n = fetch[k][0]
    fb = 2.0 * 1024.0 * fetch[k][1] / n           # KB -> bytes, doubled on gfx950 (MI355X_MICROARCH.md, HBM section)
    wb = 1024.0 * write[k][1] / write[k][0]
    kernels[k] = {"launches_profiled": n, "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb, "hbm_bytes_per_launch": fb + wb}
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `bench.py --no-graph --steps 3 --warmup 1`, all "
                     "dispatches of each kernel averaged; FETCH_SIZE (KB) doubled per the gfx950 correction of MI355X_MICROARCH.md (HBM "
                     "section), WRITE_SIZE (KB) as reported", "kernels": kernels},
          open(os.path.join(out, f"{tag}_pmc_traffic.json"), "w"), indent=1)
# matrix-core utilisation per kernel (its own counter pass): mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (launch duration x 2.4 GHz x 256 CUs x 4 SIMDs) -
# the gfx94x MfmaUtil formula rocprofv3 falls back to (MI355X_MICROARCH.md, "rocprofv3 PMC slots") with the launch's own start / end time stamps as the
# cycle base: GRBM_GUI_ACTIVE of a counter-collection dispatch covers the profiler's start / stop around it (294 k cycles for a 2.7 us copy)
mf = glob.glob(os.path.join(root, "gpurun_out", "prof_mfma/**/*counter_collection.csv"), recursive=True)
if mf:
    path = max(mf, key=os.path.getmtime)
    acc = defaultdict(lambda: defaultdict(float))
    seen = set()
    with open(path) as f:
        for r in csv.DictReader(f):
            k = short(r["Kernel_Name"])
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                acc[k]["_ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                acc[k]["_n"] += 1
    rowsm = {}
    for k, v in acc.items():
        if v["_ns"] <= 0:
            continue
        cyc = v["_ns"] * 2.4
        rowsm[k] = {"launches_profiled": int(v["_n"]), "avg_launch_us_under_counters": v["_ns"] / v["_n"] / 1e3,
                    "mfma_busy_cycles_per_launch": v["SQ_VALU_MFMA_BUSY_CYCLES"] / v["_n"], "mfma_util": v["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 256 * 4),
                    "valu_issue_share_of_wave_cycles": (v["SQ_ACTIVE_INST_VALU"] / v["SQ_WAVE_CYCLES"]) if v["SQ_WAVE_CYCLES"] else None,
                    "share_of_kernel_time": v["_ns"]}
    tot = sum(r["share_of_kernel_time"] for r in rowsm.values())
    for r in rowsm.values():
        r["share_of_kernel_time"] /= tot
    top = dict(sorted(rowsm.items(), key=lambda kv: -kv[1]["share_of_kernel_time"])[:24])
    json.dump({"source": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE over "
                         "`bench.py --no-graph --steps 3 --warmup 1` (a pass of its own); mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (launch duration * 2.4 GHz * 256 "
                         "CUs * 4 SIMDs), the share of the matrix pipes' cycles in which an MFMA was executing; kernels ordered by their share of the kernel time",
               "kernels": top}, open(os.path.join(out, f"{tag}_mfma.json"), "w"), indent=1)
for name, log in (("bench_line", "bench_official.log"), ("bench_f32_line", "bench_f32.log"), ("bench_c3_line", "bench_c3.log"), ("bench_c5_line", "bench_c5.log")):
    path = os.path.join(root, "gpurun_out", log)
    if os.path.exists(path):
        lines = [l for l in open(path) if l.startswith("{")]
        if lines:
            open(os.path.join(out, f"{tag}_{name}.json"), "w").write(lines[-1])
c3 = glob.glob(os.path.join(root, "gpurun_out", "prof_c3/**/*kernel_stats.csv"), recursive=True)
if c3:
    shutil.copy(max(c3, key=os.path.getmtime), os.path.join(out, f"{tag}_bench_c3_bf16_kernel_stats.csv"))
print("wrote", tag, "kernels:", len(kernels))
