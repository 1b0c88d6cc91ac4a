"""Per hardware queue busy time of ONE hipGraph replay (rocprofv3 --kernel-trace csv[.gz]): the step is cut between two consecutive
param_cast_batch_kernel launches (one per step); for every queue the busy time, and for the busiest (main) queue its idle time and the
kernel families by time.  Usage: python scripts/queue_breakdown.py trace.csv.gz [step index from the end, default 2]"""
import collections
import csv
import gzip
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.search(r"miseg::(\w+)", name) or re.search(r"_ZN5miseg\d+(\w+?)I", name) or re.search(r"_ZN5miseg\d+([a-z0-9_]+)", name)
    return m.group(1) if m else name[:40]


def main():
    f = gzip.open(sys.argv[1], "rt") if sys.argv[1].endswith(".gz") else open(sys.argv[1])
    rows = sorted(csv.DictReader(f), key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "param_cast_batch_kernel" in r["Kernel_Name"]]
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    lo, hi = marks[-back - 1], marks[-back]
    step = rows[lo:hi]
    t0, t1 = int(step[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in step)
    print(f"step of {len(step)} launches, {(t1 - t0) / 1e3:.1f} us wall")
    byq = collections.defaultdict(list)
    for r in step:
        byq[r["Queue_Id"]].append(r)
    main_q = max(byq, key=lambda q: sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in byq[q]))
    for q, rs in sorted(byq.items()):
        busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs) / 1e3
        print(f"queue {q}: {len(rs):4d} launches, busy {busy:8.1f} us, first +{(int(rs[0]['Start_Timestamp']) - t0) / 1e3:.0f} us, last ends +{(max(int(r['End_Timestamp']) for r in rs) - t0) / 1e3:.0f} us" + ("   <- main" if q == main_q else ""))
    rs = byq[main_q]
    idle, fam = 0.0, collections.defaultdict(lambda: [0, 0.0])
    gaps = []
    for i, r in enumerate(rs):
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        k = short(r["Kernel_Name"])
        fam[k][0] += 1
        fam[k][1] += d
        if i:
            g = (int(r["Start_Timestamp"]) - int(rs[i - 1]["End_Timestamp"])) / 1e3
            idle += max(g, 0.0)
            gaps.append((g, short(rs[i - 1]["Kernel_Name"]), k))
    print(f"main queue idle between its launches: {idle:.1f} us; the 12 longest waits:")
    for g, a, b in sorted(gaps, reverse=True)[:12]:
        print(f"   {g:7.1f} us  {a} -> {b}")
    for k, (n, d) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f"   {k:36s} n {n:3d}  {d:8.1f} us  avg {d / n:6.1f}")
    for q, rs in sorted(byq.items()):
        if q == main_q:
            continue
        fam = collections.defaultdict(lambda: [0, 0.0])
        for r in rs:
            k = short(r["Kernel_Name"])
            fam[k][0] += 1
            fam[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        print(f"queue {q}: " + ", ".join(f"{k} x{n} {d:.0f}" for k, (n, d) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:8]))


if __name__ == "__main__":
    main()
