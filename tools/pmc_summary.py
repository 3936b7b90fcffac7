"""HBM-side bytes per launch of the conv kernel classes from the two rocprofv3 --pmc passes of tools/pmc_step.sh.

    python tools/pmc_summary.py <fetch_pass.db> <write_pass.db> > profiles/rNN_hbm_traffic.json

FETCH_SIZE / WRITE_SIZE are per-dispatch KiB from the L2's memory-side request counters (Infinity-Cache hits included).  On gfx950
FETCH_SIZE tallies the 128-byte requests of 16-byte-per-lane streaming loads at 64 bytes (/opt/skills/guides/MI355X_MICROARCH.md,
section HBM): it is doubled for the kernels that stage their operands with 16-byte loads (the bf16x3 families); WRITE_SIZE as is."""
import json
import sqlite3
import sys

CLASSES = {   # class -> (kernel-name substrings, FETCH_SIZE correction)
    'bf16x3_implicit_gemm': (('conv_bf3_kernelILi256ELi128', 'conv_bf3_kernelILi128ELi256', 'conv_bf3_wgrad_pp_kernel'), 2.0),
    'bf16x3_small_tile': (('conv_bf3_kernelILi64ELi128', 'conv_bf3_wgrad_kernel'), 2.0),
    'vox_bf16x3': (('vox_bf3',), 2.0),
    'f32_implicit_gemm': (('conv_fwd_kernel', 'conv_wgrad_kernel'), 1.0),
}


def per_kernel(path):
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    t = lambda p: next(x for x in tabs if x.startswith(p))
    kd, ks, pe, pi = t('rocpd_kernel_dispatch'), t('rocpd_info_kernel_symbol'), t('rocpd_pmc_event'), t('rocpd_info_pmc')
    q = (f'select s.kernel_name, p.name, count(*), sum(e.value), sum(d.end - d.start) from `{pe}` e '
         f'join `{pi}` p on e.pmc_id = p.id join `{kd}` d on d.event_id = e.event_id '
         f'join `{ks}` s on d.kernel_id = s.id group by 1, 2')
    return [(k, c, n, v, dur) for k, c, n, v, dur in db.execute(q)]


def main():
    fetch, write = per_kernel(sys.argv[1]), per_kernel(sys.argv[2])
    out = {'source': 'tools/pmc_step.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over bench.py '
                     '--steps 2 --warmup 1; per-dispatch KiB summed per kernel class; FETCH_SIZE x2 for the 16-byte loads of the '
                     'bf16x3 kernels (gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md; Infinity-Cache hits included), '
                     'WRITE_SIZE as reported', 'steps_profiled': 3, 'classes': {}}
    for cls, (pats, corr) in CLASSES.items():
        n = sum(r[2] for r in fetch if r[1] == 'FETCH_SIZE' and any(p in r[0] for p in pats))
        if not n:
            continue
        fb = sum(r[3] for r in fetch if r[1] == 'FETCH_SIZE' and any(p in r[0] for p in pats)) * 1024.0 * corr / n
        dur = sum(r[4] for r in fetch if r[1] == 'FETCH_SIZE' and any(p in r[0] for p in pats)) / n * 1e-3
        nw = sum(r[2] for r in write if r[1] == 'WRITE_SIZE' and any(p in r[0] for p in pats))
        wb = sum(r[3] for r in write if r[1] == 'WRITE_SIZE' and any(p in r[0] for p in pats)) * 1024.0 / max(nw, 1)
        out['classes'][cls] = dict(dispatches=n, avg_us=round(dur, 1), fetch_bytes_per_launch=int(fb), fetch_correction=corr,
                                   write_bytes_per_launch=int(wb), hbm_bytes_per_launch=int(fb + wb),
                                   gb_per_s=round((fb + wb) / (dur * 1e-6) / 1e9, 1))
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
