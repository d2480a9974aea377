"""Prints a per-step table from a rocprofv3 *_kernel_stats.csv (development aid)."""
import csv, re, sys
path, steps = sys.argv[1], int(sys.argv[2])
rows = list(csv.DictReader(open(path)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f"total kernel time {tot/1e6/steps:.3f} ms/step")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 25]:
    name = re.sub(r'\(.*', '', r['Name'])[:72]
    print(f"{name:74s} calls/step {int(r['Calls'])/steps:6.1f} avg_us {float(r['AverageNs'])/1e3:8.1f} "
          f"ms/step {float(r['TotalDurationNs'])/1e6/steps:7.3f} {float(r['Percentage']):5.1f}%")
