"""How full is the device over a training step?  From a rocprofv3 kernel trace (rocpd database): the union of all kernel
intervals, split by what is running - at least one LARGE kernel (>= 256 workgroups: one per compute unit), only small kernels
(each occupies a few compute units), nothing - plus the small kernels that account for most of the "small only" time.
    python tools/rocpd_timeline.py <p_results.db> [steps_to_analyse=4]"""
import sqlite3
import sys
from collections import defaultdict

import numpy as np

db = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = next(x for x in tabs if x.startswith('rocpd_kernel_dispatch'))
ks = next(x for x in tabs if x.startswith('rocpd_info_kernel_symbol'))
cols = [r[1] for r in db.execute(f'pragma table_info(`{kd}`)')]
gx = [c for c in cols if c.startswith('grid_size')]
wx = [c for c in cols if c.startswith('workgroup_size')]
sel = ', '.join(f'd.{c}' for c in gx + wx)
raw = db.execute(f'select d.start, d.end, s.kernel_name, d.queue_id, {sel} from `{kd}` d join `{ks}` s on d.kernel_id = s.id order by d.start').fetchall()
# one step = the dispatches between two AdamW groups: find the adamw launches
adam = [i for i, r in enumerate(raw) if 'adamw' in r[2]]
ends = [adam[i] for i in range(len(adam)) if i + 1 == len(adam) or adam[i + 1] - adam[i] > 50]
assert len(ends) > steps + 1, f'only {len(ends)} steps in the trace'
lo, hi = ends[-steps - 1] + 1, ends[-1] + 1
rows = raw[lo:hi]
t0, t1 = rows[0][0], max(r[1] for r in rows)


def nwg(r):
    g = r[4:4 + len(gx)]
    w = r[4 + len(gx):]
    n = 1
    for a, b in zip(g, w):
        n *= max(1, (a + max(b, 1) - 1) // max(b, 1))
    return n


ev = []
for r in rows:
    big = nwg(r) >= 256
    ev.append((r[0], 1, big, r[2]))
    ev.append((r[1], -1, big, r[2]))
ev.sort(key=lambda e: (e[0], e[1]))
nbig = nsmall = 0
last = t0
tot = dict(big=0, small=0, idle=0)
running_small = defaultdict(int)
small_only_by_kernel = defaultdict(int)
for t, d, big, name in ev:
    dt = t - last
    if dt > 0:
        if nbig > 0:
            tot['big'] += dt
        elif nsmall > 0:
            tot['small'] += dt
            share = dt / max(1, sum(running_small.values()))
            for k, c in running_small.items():
                small_only_by_kernel[k] += share * c
        else:
            tot['idle'] += dt
    last = t
    if big:
        nbig += d
    else:
        nsmall += d
        running_small[name] += d
        if running_small[name] == 0:
            del running_small[name]
span = t1 - t0
queues = len({r[3] for r in rows})
print(f'{steps} steps, {len(rows)} dispatches on {queues} HSA queues, span {span / 1e6 / steps:.2f} ms/step; sum of kernel durations '
      f'{sum(r[1] - r[0] for r in rows) / 1e6 / steps:.2f} ms/step')
for k in ('big', 'small', 'idle'):
    print(f'  {k:6s} {tot[k] / 1e6 / steps:7.2f} ms/step  ({100.0 * tot[k] / span:5.1f} %)   '
          + {'big': 'at least one kernel with >= 256 workgroups is running', 'small': 'only kernels with < 256 workgroups are running',
             'idle': 'nothing is running'}[k])
print('kernels behind the "small only" time (ms/step):')
for k, v in sorted(small_only_by_kernel.items(), key=lambda kv: -kv[1])[:25]:
    print(f'  {v / 1e6 / steps:7.3f}  {k.split("(")[0][:110]}')
