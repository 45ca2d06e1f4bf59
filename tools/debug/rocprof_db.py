"""kernel averages from a rocprofv3 results database (when no csv was written): python tools/debug/rocprof_db.py <dir> [pattern]"""
import sqlite3, glob, sys
f = glob.glob(sys.argv[1] + '/**/*.db', recursive=True)[0]
pat = sys.argv[2] if len(sys.argv) > 2 else ''
c = sqlite3.connect(f)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]; ks = [t for t in tabs if 'kernel_symbol' in t][0]
q = f"select s.kernel_name, count(*), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by sum(d.end-d.start) desc limit 14"
for r in c.execute(q):
    if pat in r[0]:
        print('%-90s %5d avg %9.1f us min %9.1f max %9.1f' % (r[0][:90], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3))
if pat:
    q = f"select s.kernel_name, d.start, d.end, d.grid_size_x, d.grid_size_y, d.workgroup_size_x from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"
    for r in c.execute(q):
        if pat in r[0]:
            print('  %9.1f us grid %s x %s wg %s' % ((r[2] - r[1]) / 1e3, r[3], r[4], r[5]))
