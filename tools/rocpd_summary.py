"""Turn a rocprofv3 rocpd SQLite database (the default output of `rocprofv3 --kernel-trace --stats` in ROCm 7.2)
into the per-kernel summary that is committed under profiles/.  Usage: python tools/rocpd_summary.py DB OUT.csv"""
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
c = sqlite3.connect(db)
rows = c.execute("""select name, count(*), sum(duration), avg(duration), min(duration), max(duration),
                    max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), max(grid_x), max(workgroup_x)
                    from kernels group by name order by sum(duration) desc""").fetchall()
tot = sum(r[2] for r in rows)
with open(out, 'w') as f:
    f.write("kernel,calls,total_us,avg_us,min_us,max_us,pct,vgpr,agpr,sgpr,lds_bytes,scratch,grid_x,wg_x\n")
    for r in rows:
        f.write('"%s",%d,%.1f,%.2f,%.2f,%.2f,%.2f,%s,%s,%s,%s,%s,%s,%s\n' % (r[0].replace('"', "'"), r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3,
                                                                         r[5] / 1e3, 100.0 * r[2] / tot, *r[6:]))
print(open(out).read()[:6000])
