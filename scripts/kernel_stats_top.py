"""top kernels of a rocprofv3 --stats kernel_stats.csv: calls, average us, share"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 1e6:.1f} ms")
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 24]:
    print(f"{r['Name'][:84]:84s} {int(r['Calls']):7d} {float(r['AverageNs']) / 1e3:8.1f} {float(r['TotalDurationNs']) / tot * 100:6.2f}")
