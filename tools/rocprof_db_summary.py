import sqlite3,glob,sys
for d in sys.argv[1:]:
    db=glob.glob(f'{d}/*.db')[0]
    c=sqlite3.connect(db)
    print(d)
    rows=c.execute("select name, count(*), avg(end-start), sum(end-start) from kernels group by name order by 4 desc limit 10").fetchall()
    for r in rows: print("  %-80s n=%d avg=%.1f us tot=%.1f us"%(r[0][:80],r[1],r[2]/1e3,r[3]/1e3))
    rows=c.execute("select name, start, end, grid_x, workgroup_x from kernels order by start desc limit 10").fetchall()
    prev=None
    for r in rows[::-1]:
        print("    %-60s %8.2f us  grid %d  gap %.2f us"%(r[0][:60], (r[2]-r[1])/1e3, r[3], ((r[1]-prev)/1e3 if prev else 0))); prev=r[2]
