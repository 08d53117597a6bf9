"""Kernel sequence of ONE optimisation step out of a rocprofv3 --kernel-trace database (rocpd sqlite): every dispatch between
the last two `adam_step_kernel` launches with its start offset, duration and the idle gap in front of it.
usage: python tools/diag/step_timeline.py <results.db> [which-step-from-the-end]"""
import re
import sqlite3
import sys


def main(path, back=1):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tab = [r[0] for r in cur.execute("select name from sqlite_master where type='table' and name like 'rocpd_kernel_dispatch%'")][0]
    suffix = tab.replace('rocpd_kernel_dispatch', '')
    rows = list(cur.execute("select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch%s d join rocpd_info_kernel_symbol%s s "
                            "on d.kernel_id=s.id order by d.start" % (suffix, suffix)))
    marks = [i for i, r in enumerate(rows) if 'adam_step_kernel' in r[0]]
    lo, hi = marks[-1 - back], marks[-back]
    t0 = rows[lo][2]
    prev = t0
    busy = 0
    glue = 0
    for name, a, b in rows[lo + 1:hi + 1]:
        short = re.sub(r'\.kd$', '', name)
        short = re.sub(r'^_ZN3tnp\d+', 'tnp::', short)
        short = re.sub(r'^_ZN2at6native\d*', 'at::', short)
        is_glue = not short.startswith('tnp::')
        busy += b - a
        glue += (b - a) if is_glue else 0
        print('%9.1f us  +%6.1f gap  %7.2f us  %s%s' % ((a - t0) / 1e3, (a - prev) / 1e3, (b - a) / 1e3, '* ' if is_glue else '  ', short[:100]))
        prev = b
    print('step: %.1f us from the end of one adam_step to the end of the next, %.1f us inside kernels (%.1f us in %s), %d launches'
          % ((rows[hi][2] - t0) / 1e3, busy / 1e3, glue / 1e3, 'kernels that are not tnp:: (marked *)', hi - lo))


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1)
