"""Summarise a rocprofv3 rocpd SQLite database (`rocprofv3 --kernel-trace --stats -d DIR -o NAME`) as the
per-kernel statistics table that gets committed under profiles/.

    python tools/rocpd_stats.py gpurun_out/prof/x_results.db [--top 40] [--split-grid] > profiles/rNN_kernel_stats.txt

Columns: calls, total ms, avg us, min us, max us, % of GPU kernel time, VGPR/AGPR/SGPR/LDS, kernel name.
--split-grid additionally keys on the launch grid so that one templated kernel used at several problem shapes shows
up as one row per shape.
"""
import argparse
import re
import sqlite3
import sys


def short(name, n=110):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    return name if len(name) <= n else name[:n - 3] + '...'


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('db')
    ap.add_argument('--top', type=int, default=60)
    ap.add_argument('--split-grid', action='store_true')
    args = ap.parse_args()
    db = sqlite3.connect(args.db)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = next(t for t in tabs if t.startswith('rocpd_kernel_dispatch'))
    ks = next(t for t in tabs if t.startswith('rocpd_info_kernel_symbol'))
    key = 's.kernel_name' + (", d.grid_size_x || 'x' || d.grid_size_y || 'x' || d.grid_size_z" if args.split_grid else ", ''")
    q = (f'select {key}, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start), '
         f's.arch_vgpr_count, s.accum_vgpr_count, s.sgpr_count, max(d.group_segment_size) '
         f'from `{kd}` d join `{ks}` s on d.kernel_id = s.id group by 1, 2 order by 4 desc')
    rows = db.execute(q).fetchall()
    tot = sum(r[3] for r in rows) or 1
    span = db.execute(f'select min(start), max(end) from `{kd}`').fetchone()
    print(f'# source: {args.db}')
    print(f'# kernels: {len(rows)} distinct, {sum(r[2] for r in rows)} dispatches, '
          f'sum of kernel time {tot / 1e6:.2f} ms, first-start..last-end {(span[1] - span[0]) / 1e6:.2f} ms')
    print(f'{"calls":>7} {"total_ms":>10} {"avg_us":>10} {"min_us":>9} {"max_us":>9} {"pct":>6} {"vgpr":>4} {"agpr":>4} '
          f'{"sgpr":>4} {"lds":>6}  name')
    for name, grid, calls, t, mn, mx, vg, ag, sg, lds in rows[:args.top]:
        g = f' grid={grid}' if grid else ''
        print(f'{calls:7d} {t / 1e6:10.3f} {t / calls / 1e3:10.2f} {mn / 1e3:9.2f} {mx / 1e3:9.2f} {100.0 * t / tot:6.2f} '
              f'{vg or 0:4d} {ag or 0:4d} {sg or 0:4d} {lds or 0:6d}  {short(name)}{g}')
    if len(rows) > args.top:
        rest = rows[args.top:]
        print(f'{sum(r[2] for r in rest):7d} {sum(r[3] for r in rest) / 1e6:10.3f} {"":>10} {"":>9} {"":>9} '
              f'{100.0 * sum(r[3] for r in rest) / tot:6.2f}  ({len(rest)} more kernels)')


if __name__ == '__main__':
    sys.exit(main())
