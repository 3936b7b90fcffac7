"""Per HIP stream (HSA queue): how busy it is over a training step and what the END of backward looks like - which queue is
still running how long after the others went idle (the drain of the weight-gradient stream before the optimizer).
    python tools/rocpd_queues.py <p_results.db> [steps=3]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = next(x for x in tabs if x.startswith('rocpd_kernel_dispatch'))
ks = next(x for x in tabs if x.startswith('rocpd_info_kernel_symbol'))
raw = db.execute(f'select d.start, d.end, s.kernel_name, d.queue_id from `{kd}` d join `{ks}` s on d.kernel_id = s.id order by d.start').fetchall()
adam = [i for i, r in enumerate(raw) if 'adamw' in r[2]]
first = [adam[i] for i in range(len(adam)) if i == 0 or adam[i] - adam[i - 1] > 50]      # first AdamW launch of each step
assert len(first) > steps + 1
for si in range(len(first) - steps, len(first)):
    lo, hi = first[si - 1], first[si]
    rows = raw[lo:hi]
    t0, t1 = rows[0][0], raw[hi][0]
    print(f'step ending at AdamW #{si}: {(t1 - t0) / 1e6:.2f} ms')
    qs = sorted({r[3] for r in rows})
    for q in qs:
        rq = [r for r in rows if r[3] == q]
        busy = sum(r[1] - r[0] for r in rq) / 1e6
        print(f'   queue {q}: {len(rq):5d} kernels, sum of durations {busy:7.2f} ms, last kernel ends {(t1 - max(r[1] for r in rq)) / 1e6:6.2f} ms before AdamW starts')
    print('   last kernels before AdamW (start ms before AdamW, duration us, queue, name):')
    for r in rows[-14:]:
        print(f'      {(t1 - r[0]) / 1e6:7.3f}  {(r[1] - r[0]) / 1e3:8.1f}  q{r[3]}  {r[2][:90]}')
    # how long is exactly one queue busy at the end?
    ends = {q: max(r[1] for r in rows if r[3] == q) for q in qs}
    order = sorted(ends.values())
    if len(order) > 1:
        print(f'   the last queue runs alone for {(order[-1] - order[-2]) / 1e6:.2f} ms')

# the longest idle gaps of the busiest queue (the critical path) in the last step, and what follows them
rows = raw[first[-2]:first[-1]]
t1 = raw[first[-1]][0]
busy = {}
for r in rows:
    busy[r[3]] = busy.get(r[3], 0) + (r[1] - r[0])
main = max(busy, key=busy.get)
rq = [r for r in rows if r[3] == main]
gaps = []
for a, b in zip(rq[:-1], rq[1:]):
    if b[0] - a[1] > 0:
        gaps.append((b[0] - a[1], a, b))
tot = sum(g[0] for g in gaps) / 1e6
big = [g for g in gaps if g[0] > 20e3]
print(f'\nqueue {main} (busiest) in the last step: {len(rq)} kernels, idle between its kernels {tot:.2f} ms, of which {sum(g[0] for g in big) / 1e6:.2f} ms '
      f'in {len(big)} gaps > 20 us')
for g in sorted(big, reverse=True)[:16]:
    others = [r for r in rows if r[3] != main and r[0] < g[2][0] and r[1] > g[1][1]]
    names = sorted({(r[3], r[2][:40]) for r in others})
    print(f'   {g[0] / 1e3:8.1f} us at {(t1 - g[1][1]) / 1e6:6.2f} ms before AdamW: after {g[1][2][:44]} -> before {g[2][2][:44]}; meanwhile: '
          + ('; '.join(f'q{q} {n}' for q, n in names[:3]) or 'nothing'))

# detail of the three longest gaps: the busiest queue's kernels around them and the other queues' kernels during them
for g in sorted(big, reverse=True)[:3]:
    i = rq.index(g[1])
    print(f'\ngap of {g[0] / 1e3:.0f} us on queue {main}:')
    for r in rq[max(0, i - 4):i + 1]:
        print(f'   before  {(t1 - r[0]) / 1e6:7.3f} ms  {(r[1] - r[0]) / 1e3:7.1f} us  {r[2][:70]}')
    for r in rq[i + 1:i + 5]:
        print(f'   after   {(t1 - r[0]) / 1e6:7.3f} ms  {(r[1] - r[0]) / 1e3:7.1f} us  {r[2][:70]}')
    dur = [r for r in rows if r[3] != main and r[0] < g[2][0] and r[1] > g[1][1]]
    print(f'   during: {len(dur)} kernels on other queues; the first 6 and the last 8:')
    for r in dur[:6] + dur[-8:]:
        print(f'   during  q{r[3]} {(t1 - r[0]) / 1e6:7.3f} .. {(t1 - r[1]) / 1e6:7.3f} ms  {r[2][:70]}')
