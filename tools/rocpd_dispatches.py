"""Per-dispatch listing of the kernels whose name contains a pattern: grid, workgroup size, duration (rocprofv3 rocpd database).
   python tools/rocpd_dispatches.py <p_results.db> <pattern> [...]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
t = lambda p: next(x for x in tabs if x.startswith(p))
kd, ks = t('rocpd_kernel_dispatch'), t('rocpd_info_kernel_symbol')
cols = [r[1] for r in db.execute(f'pragma table_info(`{kd}`)')]
gx = 'grid_size_x' if 'grid_size_x' in cols else 'grid_x'
wx = 'workgroup_size_x' if 'workgroup_size_x' in cols else 'workgroup_x'
for pat in sys.argv[2:]:
    q = (f'select s.kernel_name, d.{gx}, d.{wx}, count(*), avg(d.end - d.start), min(d.end - d.start), sum(d.end - d.start) from `{kd}` d '
         f'join `{ks}` s on d.kernel_id = s.id where s.kernel_name like ? group by 1, 2, 3 order by 7 desc')
    print(f'== {pat}')
    for name, g, w, n, avg, mn, tot in db.execute(q, (f'%{pat}%',)):
        print(f'{tot / 1e6:9.3f} ms  n={n:4d}  avg {avg / 1e3:8.1f} us  min {mn / 1e3:8.1f} us  grid {g:8d} wg {w:4d}  {name[:70]}')
