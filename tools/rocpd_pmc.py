"""Per-kernel PMC summary of rocprofv3 --pmc rocpd databases: mean counter value per dispatch for each kernel.
    python tools/rocpd_pmc.py gpurun_out/pmc_x_pass1/p_results.db [more.db ...] [--match conv_bf3]"""
import argparse
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('dbs', nargs='+')
    ap.add_argument('--match', default='')
    args = ap.parse_args()
    for path in args.dbs:
        db = sqlite3.connect(path)
        tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
        t = lambda p: next(x for x in tabs if x.startswith(p))
        kd, ks, pe, pi, ev = t('rocpd_kernel_dispatch'), t('rocpd_info_kernel_symbol'), t('rocpd_pmc_event'), t('rocpd_info_pmc'), t('rocpd_event')
        q = (f'select s.kernel_name, p.name, count(*), avg(e.value), avg(d.end - d.start) from `{pe}` e '
             f'join `{pi}` p on e.pmc_id = p.id join `{kd}` d on d.event_id = e.event_id '
             f'join `{ks}` s on d.kernel_id = s.id group by 1, 2 order by 1, 2')
        cur = None
        print(f'# {path}')
        for kname, cname, n, val, dur in db.execute(q):
            if args.match and args.match not in kname:
                continue
            if kname != cur:
                cur = kname
                print(f'{kname[:100]}  (dispatches {n}, avg {dur / 1e3:.1f} us)')
            print(f'    {cname:32s} {val:16.1f}')


if __name__ == '__main__':
    main()
