"""Print a readable per-kernel table from a rocprofv3 --stats output directory."""
import csv
import glob
import re
import sys

f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r['Name']
    m = re.search(r'(\w+_kernel|__amd_\w+)', n)
    t = re.search(r'<([^>]*)>', n)
    name = m.group(1) if m else n[:40]
    tmpl = '<' + t.group(1) + '>' if t else ''
    print(f"{name:28s} {tmpl:18s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:8.2f} "
          f"min={float(r['MinNs'])/1e3:8.2f} max={float(r['MaxNs'])/1e3:8.2f} pct={float(r['Percentage']):5.2f}")
