"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel, per counter, mean over dispatches."""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    name = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
    name = re.sub(r'\(.*', '', name).replace('void ', '')
    acc[name][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    print(k)
    for c, v in d.items():
        big = max(v)
        print(f'   {c:28s} n={len(v):3d} max={big:.4g} mean={sum(v)/len(v):.4g}')
