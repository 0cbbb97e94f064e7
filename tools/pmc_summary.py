"""Summarise rocprofv3 --pmc CSVs (one dir per pass) for kernels matching a substring. Usage: pmc_summary.py DIR SUBSTR"""
import csv, glob, sys, collections
d, sub = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(acc):
    v = acc[k]
    print(f'{k:28s} n={len(v):3d} avg={sum(v)/len(v):.4g}')
