"""Summarise rocprofv3 --pmc CSVs: mean counter value per launch for kernels matching a pattern.
    python tools/pmc_summary.py <dir-or-glob> [kernel-substring]"""
import collections
import csv
import glob
import sys

pat = sys.argv[2] if len(sys.argv) > 2 else 'ctrl_accumulate'
files = sorted(glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True))
for f in files:
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if pat in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in sorted(acc.items()):
        print(f'{k:30s} n={len(v):3d} mean={sum(v)/len(v):.5g}')
