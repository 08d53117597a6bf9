"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace run (rocpd sqlite database): where the GPU waits for
the host.  Gaps longer than --min-us are grouped by (kernel before -> kernel after); the table is per optimisation step when
--per NAME is given (the number of launches of kernel NAME = the number of steps in the trace)."""
import argparse
import glob
import re
import sqlite3


def short(n):
    n = re.sub(r'\.kd$', '', n)
    n = re.sub(r'^_ZN3tnp\d+', 'tnp::', n)
    n = re.sub(r'^_ZN2at6native\d+', 'at::', n)
    return re.split(r'[IE(<]', n)[0][:44]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('db', nargs='*')
    ap.add_argument('--min-us', type=float, default=3.0)
    ap.add_argument('--per', default='adam_step_kernel')
    ap.add_argument('--skip-first-ms', type=float, default=400.0, help='ignore start-up (module load, priming)')
    a = ap.parse_args()
    for path in a.db or glob.glob('gpurun_out/prof/*.db'):
        db = sqlite3.connect(path)
        cur = db.cursor()
        tab = [r[0] for r in cur.execute(
            "select name from sqlite_master where type='table' and name like 'rocpd_kernel_dispatch%'")][0]
        suffix = tab.replace('rocpd_kernel_dispatch', '')
        rows = list(cur.execute(
            "select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch%s d join rocpd_info_kernel_symbol%s s "
            "on d.kernel_id=s.id order by d.start" % (suffix, suffix)))
        if not rows:
            continue
        t0 = rows[0][1] + a.skip_first_ms * 1e6
        rows = [r for r in rows if r[1] >= t0]
        steps = sum(1 for r in rows if a.per in r[0]) or 1
        busy = sum(r[2] - r[1] for r in rows)
        span = rows[-1][2] - rows[0][1]
        gaps = {}
        for p, q in zip(rows, rows[1:]):
            g = (q[1] - p[2]) / 1e3
            if g >= a.min_us:
                k = (short(p[0]), short(q[0]))
                c = gaps.setdefault(k, [0, 0.0])
                c[0] += 1
                c[1] += g
        print('### %s: %d steps, span %.3f ms/step, kernels busy %.3f ms/step, idle %.3f ms/step\n'
              % (path, steps, span / 1e6 / steps, busy / 1e6 / steps, (span - busy) / 1e6 / steps))
        print('| after kernel | before kernel | gaps / step | idle us / step |\n|---|---|---|---|')
        for k, c in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]:
            print('| %s | %s | %.1f | %.1f |' % (k[0], k[1], c[0] / steps, c[1] / steps))
        print()


if __name__ == '__main__':
    main()
