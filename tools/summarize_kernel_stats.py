"""Condense a rocprofv3 `--kernel-trace --stats --output-format csv` kernel_stats.csv into the
one-line-per-kernel text kept under profiles/:
    python tools/summarize_kernel_stats.py <kernel_stats.csv> ["# header line" ...]
"""
import csv
import re
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    m = re.match(r'(?:void )?(?:ffk::)?([A-Za-z_0-9]+)(<[^>(]*>)?', name.strip('"'))
    return (m.group(1), m.group(2) or '') if m else (name[:40], '')


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    for line in sys.argv[2:]:
        print(line)
    for r in rows:
        kernel, params = short(r['Name'])
        print(f"{kernel:28s} {params:18s} calls={int(r['Calls']):5d} avg_us={float(r['AverageNs'])/1e3:8.2f} "
              f"min={float(r['MinNs'])/1e3:8.2f} max={float(r['MaxNs'])/1e3:8.2f} pct={float(r['Percentage']):5.2f}")


if __name__ == '__main__':
    main()
