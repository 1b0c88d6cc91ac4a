"""Concurrency inside ONE hipGraph replay from a rocprofv3 --kernel-trace csv: for the last period of back-to-back replays, the wall time,
the sum of kernel durations, the time with >= 1 and >= 2 kernels resident, and (--all) every kernel with start / end / how many others ran
beside it.  Usage: python scripts/step_overlap.py trace.csv [--all]"""
import csv
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.search(r"miseg::(\w+)", name) or re.search(r"_ZN5miseg\d+(\w+?)I", name) or re.search(r"_ZN5miseg\d+([a-z0-9_]+)", name)
    return m.group(1) if m else name[:40]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    names = [short(r["Kernel_Name"]) for r in rows]
    st = [int(r["Start_Timestamp"]) for r in rows]
    en = [int(r["End_Timestamp"]) for r in rows]
    fills = [i for i, n in enumerate(names) if n == "param_cast_batch_kernel"]
    lo, hi = fills[-2], fills[-1]
    ev = sorted([(st[i], 1) for i in range(lo, hi)] + [(en[i], -1) for i in range(lo, hi)])
    depth, last, t1, t2 = 0, ev[0][0], 0, 0
    for t, d in ev:
        if depth >= 1:
            t1 += t - last
        if depth >= 2:
            t2 += t - last
        depth += d
        last = t
    print(f"period = kernels {lo}..{hi - 1} ({hi - lo} launches), wall {(st[hi] - st[lo]) / 1e3:.1f} us, sum of durations "
          f"{sum(en[i] - st[i] for i in range(lo, hi)) / 1e3:.1f} us, >= 1 resident {t1 / 1e3:.1f} us, >= 2 resident {t2 / 1e3:.1f} us")
    if "--all" in sys.argv:
        for i in range(lo, hi):
            beside = sum(1 for j in range(max(lo, i - 40), min(hi, i + 40)) if j != i and st[j] < en[i] and en[j] > st[i])
            r = rows[i]
            print(f"{(st[i] - st[lo]) / 1e3:9.1f} {(en[i] - st[lo]) / 1e3:9.1f} {(en[i] - st[i]) / 1e3:7.1f} beside {beside:2d}  {names[i]:34s} "
                  f"grid {int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1)}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']} "
                  f"queue {r.get('Queue_Id', '?')} stream {r.get('Stream_Id', '?')}")


main()
