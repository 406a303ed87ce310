"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel: mean counter value per launch.
usage: python tools/pmc_table.py <counter_collection.csv> [name-substring]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2] if len(sys.argv) > 2 else ''
acc = defaultdict(lambda: defaultdict(float))
launches = defaultdict(set)
for r in rows:
    k = r['Kernel_Name']
    if flt not in k:
        continue
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    launches[k].add(r['Dispatch_Id'])
names = sorted({c for k in acc for c in acc[k]})
print('kernel'.ljust(84), 'n'.rjust(5), *[c[-18:].rjust(19) for c in names])
for k in sorted(acc, key=lambda k: -sum(acc[k].values())):
    n = len(launches[k])
    print(k[:84].ljust(84), str(n).rjust(5), *[f'{acc[k][c] / n:19.0f}' for c in names])
