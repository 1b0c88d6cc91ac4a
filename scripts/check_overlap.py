"""per replayed step (bursts of kernels separated by > 150 us of nothing): wall span, time with at least one kernel running, sum of
kernel durations (> busy when graph branches run concurrently).  Input: rocprofv3 --kernel-trace csv."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
bursts, cur = [], [ev[0]]
for s, e in ev[1:]:
    if s - max(x[1] for x in cur[-8:]) > 150000:
        bursts.append(cur); cur = []
    cur.append((s, e))
bursts.append(cur)
for b in bursts:
    if len(b) < 300:
        continue
    span = max(e for _, e in b) - b[0][0]
    tot = sum(e - s for s, e in b)
    busy, hi = 0, b[0][0]
    for s, e in b:
        if e > hi:
            busy += e - max(s, hi); hi = e
    print(f"kernels {len(b):5d}  span {span/1e6:6.3f} ms  busy {busy/1e6:6.3f} ms ({100*busy/span:4.1f} %)  sum of durations {tot/1e6:6.3f} ms  concurrent {100*(tot-busy)/tot:4.1f} %")
