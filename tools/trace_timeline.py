"""Print a window of a rocprofv3 --kernel-trace CSV as a timeline (start, duration, stream/queue,
kernel), to see which kernels overlap.    python tools/trace_timeline.py <dir> [first] [count]"""
import csv
import glob
import re
import sys

f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
first = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
count = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
t0 = int(rows[first]['Start_Timestamp'])
for r in rows[first:first + count]:
    n = r['Kernel_Name']
    m = re.search(r'(\w+_kernel|nccl\w+|__amd_\w+)', n)
    name = m.group(1) if m else n[:50]
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"{(s - t0)/1e3:9.2f} us  +{(e - s)/1e3:7.2f}  queue {r.get('Queue_Id', '?'):>3s}  "
          f"grid {r.get('Grid_Size', '?'):>8s} wg {r.get('Workgroup_Size', '?'):>5s} "
          f"lds {r.get('LDS_Block_Size', '?'):>6s}  {name}")
