"""From a rocprofv3 --kernel-trace CSV of the bench's two-stream schedule: how do consecutive accumulate kernels
(alternating passes, they cannot share a CU: 101 KB of LDS each) hand the chip over?
    python tools/k3_handover.py <dir> [kernel-name fragment]
Prints the medians of: duration, start-to-start period, gap (next start - this end; negative = overlap), and which
other kernels were running when each accumulate kernel started."""
import csv
import glob
import sys

import numpy as np

f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
frag = sys.argv[2] if len(sys.argv) > 2 else 'ctrl_accumulate_pq_kernel'
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
k3 = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id', '?')) for r in rows if frag in r['Kernel_Name']]
k3 = k3[len(k3)//4:]                      # steady state
s = np.array([a for a, _, _ in k3], dtype=np.int64)
e = np.array([b for _, b, _ in k3], dtype=np.int64)
alt = np.mean([k3[i][2] != k3[i + 1][2] for i in range(len(k3) - 1)])
print(f'{len(k3)} accumulate kernels, {alt:.2f} of consecutive ones on different queues')
q = lambda a: f'median {np.median(a)/1e3:7.2f}  p10 {np.percentile(a, 10)/1e3:7.2f}  p90 {np.percentile(a, 90)/1e3:7.2f} us'
print('duration              ', q(e - s))
print('start-to-start period ', q(np.diff(s)))
print('gap next start - end  ', q(s[1:] - e[:-1]))
