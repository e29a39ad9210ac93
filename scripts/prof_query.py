import sqlite3, sys
db=sqlite3.connect(sys.argv[1]); cur=db.cursor()
tabs=[r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd=[t for t in tabs if 'kernel_dispatch' in t][0]; ks=[t for t in tabs if 'kernel_symbol' in t][0]
pat=sys.argv[2] if len(sys.argv)>2 else ''
q=f"select s.kernel_name, count(*), avg(d.end-d.start)/1000.0, min(d.end-d.start)/1000.0, d.grid_size_x from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name, d.grid_size_x order by 1"
for r in cur.execute(q):
    if pat in r[0]:
        print("%-60s n=%5d avg %8.1f us min %8.1f grid %d"%(r[0][:60],r[1],r[2],r[3],r[4]))
