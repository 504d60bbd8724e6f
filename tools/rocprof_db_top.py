import sqlite3, sys
c=sqlite3.connect(sys.argv[1])
tabs=[r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd=[t for t in tabs if 'kernel_dispatch' in t][0]
ks=[t for t in tabs if 'kernel_symbol' in t][0]
q=f"select s.kernel_name, count(*), avg(k.end-k.start)/1e3 from {kd} k join {ks} s on k.kernel_id=s.id group by s.kernel_name order by 3 desc limit 3"
for r in c.execute(q): print(r[0][20:70], r[1], round(r[2],1))
