"""Averages rocprofv3 --pmc counters per kernel from *_counter_collection.csv (development aid)."""
import csv, collections, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 else ''
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    name = re.sub(r'\(.*', '', r['Kernel_Name'])[:70]
    if pat in name:
        agg[name][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in agg.items():
    print(k, ' grid', rows[0]['Grid_Size'] if rows else '')
    for c, vals in sorted(v.items()):
        print(f'    {c:32s} n={len(vals):4d} avg={sum(vals)/len(vals):16.1f}')
