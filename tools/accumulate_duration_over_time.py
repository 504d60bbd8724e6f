import csv, glob, sys
import numpy as np
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'ctrl_accumulate' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
d = np.array([(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows])
n = len(d)
print(n, 'mean', d.mean())
for a in range(0, n, max(1, n//20)):
    print(a, round(d[a:a+max(1, n//20)].mean(), 2))
