"""Timeline of ONE hipGraph replay from a rocprofv3 --kernel-trace csv: the kernels of the last full step in issue order with start offset,
duration and the idle gap in front of each, plus totals per kernel family.  Usage: python scripts/step_timeline.py trace.csv [--all]"""
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
    # a step starts at the arena fill: find the last two occurrences of the first kernel of the graph by looking for the longest gap pattern
    starts = [int(r["Start_Timestamp"]) for r in rows]
    ends = [int(r["End_Timestamp"]) for r in rows]
    gaps = [0] + [starts[i] - max(ends[:i][-8:]) for i in range(1, len(rows))]
    big = [i for i, g in enumerate(gaps) if g > 100_000]           # host-side pauses between replays
    big = [0] + big + [len(rows)]
    segs = [(big[i], big[i + 1]) for i in range(len(big) - 1)]
    want = int(sys.argv[sys.argv.index("--seg") + 1]) if "--seg" in sys.argv else None
    if want is None:                                                # the last segment that looks like a whole step
        sizes = sorted(b - a for a, b in segs)
        typical = max(set(sizes), key=sizes.count)
        want = max(i for i, (a, b) in enumerate(segs) if b - a == typical)
    if "--segs" in sys.argv:
        for i, (a, b) in enumerate(segs):
            print(i, a, b, b - a, f"{(ends[b - 1] - starts[a]) / 1e3:.1f} us")
    lo, hi = segs[want]
    if "--period" in sys.argv:                                      # back-to-back replays: cut one period out of the longest segment
        lo, hi = max(segs, key=lambda ab: ab[1] - ab[0])
        reg = names[lo:hi]
        # (launches of the side-branch stream interleave a little differently from replay to replay: take the best-matching period)
        per = max(range(50, min(len(reg) // 2, 2000)), key=lambda p: sum(reg[i] == reg[i + p] for i in range(len(reg) - p)) / (len(reg) - p))
        first = next(i for i in range(lo, hi) if names[i] == "fill_words_kernel" or "fill" in names[i])
        lo = first + per * ((hi - first) // per - 1)
        hi = lo + per
    print(f"step = kernels {lo}..{hi - 1} ({hi - lo} launches), wall {(ends[hi - 1] - starts[lo]) / 1e3:.1f} us")
    fam, tot_busy, tot_gap = {}, 0, 0
    for i in range(lo, hi):
        d = ends[i] - starts[i]
        g = gaps[i] if i > lo else 0
        tot_busy += d
        tot_gap += max(g, 0)
        f = fam.setdefault(names[i], [0, 0, 0])
        f[0] += 1
        f[1] += d
        f[2] += max(g, 0)
        if "--all" in sys.argv:
            r = rows[i]
            print(f"{(starts[i] - starts[lo]) / 1e3:9.1f} {d / 1e3:7.1f} gap {g / 1e3:5.1f}  {names[i]:34s} grid {int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1)}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']} wg {r['Workgroup_Size_X']} lds {r['LDS_Block_Size']}")
    print(f"busy {tot_busy / 1e3:.1f} us, gaps {tot_gap / 1e3:.1f} us")
    for k, v in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        print(f"{k:36s} n {v[0]:4d}  busy {v[1] / 1e3:8.1f}  avg {v[1] / v[0] / 1e3:6.1f}  gaps {v[2] / 1e3:7.1f}")


main()
