"""Idle time between consecutive kernel dispatches of a rocprofv3 kernel trace (rocpd database): how much of a step the device
spends between one kernel's end and the next kernel's start.   python tools/rocpd_gaps.py <p_results.db> [steps]"""
import sqlite3
import sys
import numpy as np

db = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = next(x for x in tabs if x.startswith('rocpd_kernel_dispatch'))
ks = next(x for x in tabs if x.startswith('rocpd_info_kernel_symbol'))
raw = db.execute(f'select d.start, d.end, s.kernel_name from `{kd}` d join `{ks}` s on d.kernel_id = s.id order by d.start').fetchall()
names = [r[2] for r in raw]
rows = np.array([(r[0], r[1]) for r in raw], dtype=np.int64)
# keep the steady part: the last `steps` steps' worth of dispatches (the run starts with warm-up / compilation-free setup work)
n = len(rows)
per = n // (steps + 2) if steps > 1 else n
if steps > 1:
    names = names[-per * steps:]
rows = rows[-per * steps:] if steps > 1 else rows
gap = rows[1:, 0] - np.maximum.accumulate(rows[:-1, 1])
busy = (rows[:, 1] - rows[:, 0]).sum()
span = rows[:, 1].max() - rows[0, 0]
pos = gap[gap > 0]
print(f'dispatches {len(rows)}, span {span / 1e6:.2f} ms, sum of kernel durations {busy / 1e6:.2f} ms, idle between kernels {pos.sum() / 1e6:.2f} ms '
      f'({100.0 * pos.sum() / span:.1f} % of the span); overlapping starts {int((gap <= 0).sum())}')
for lo, hi in ((0, 1000), (1000, 2000), (2000, 4000), (4000, 10000), (10000, 100000), (100000, 10**12)):
    sel = pos[(pos >= lo) & (pos < hi)]
    print(f'   gaps {lo / 1e3:6.1f} .. {hi / 1e3 if hi < 10**11 else float("inf"):8.1f} us: {len(sel):6d}  total {sel.sum() / 1e6:8.3f} ms')
print(f'   median gap {np.median(pos) / 1e3:.2f} us, p90 {np.percentile(pos, 90) / 1e3:.2f} us')

short = lambda n: n.split('(')[0][:60]
order = np.argsort(-gap)[:40]
print('largest gaps (us): previous kernel -> next kernel')
for i in sorted(order[:40], key=lambda i: -gap[i]):
    print(f'   {gap[i] / 1e3:8.1f}   {short(names[i])}  ->  {short(names[i + 1])}')
