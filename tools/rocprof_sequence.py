"""Ordered kernel sequence of ONE optimisation step from a rocprofv3 --kernel-trace database: the kernels between the last two
adam_step_kernel launches, with runs of the recurrent step's native kernels collapsed -- shows where the framework's own
(ATen / runtime) kernels sit in the step.  usage: python tools/rocprof_sequence.py <rocpd .db>"""
import re
import sqlite3
import sys


def main(db_path):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    tab = [r[0] for r in cur.execute("select name from sqlite_master where type='table' and name like 'rocpd_kernel_dispatch%'")][0]
    suffix = tab.replace('rocpd_kernel_dispatch', '')
    rows = list(cur.execute("select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch%s d join rocpd_info_kernel_symbol%s s "
                            "on d.kernel_id=s.id order by d.start" % (suffix, suffix)))
    adam = [i for i, r in enumerate(rows) if 'adam_step' in r[0]]
    if len(adam) < 2:
        print('fewer than two optimiser steps in the trace')
        return
    seg = rows[adam[-2] + 1:adam[-1] + 1]
    t0 = seg[0][1]
    native_run, native_t = 0, 0.0
    foreign_total = 0.0
    for name, start, end in seg:
        short = re.sub(r'\.kd$', '', name)
        native = short.startswith('_ZN3tnp') or short.startswith('tnp::') or 'tnp' in short[:12]
        if native and 'adam' not in short:
            native_run += 1
            native_t += (end - start) / 1e3
            continue
        if native_run:
            print('   ... %d native kernels, %.1f us' % (native_run, native_t))
            native_run, native_t = 0, 0.0
        if not native:
            foreign_total += (end - start) / 1e3
        print('%9.1f us  %6.2f us  %s' % ((start - t0) / 1e3, (end - start) / 1e3, short[:150]))
    print('step span %.1f us, non-native kernels %.1f us' % ((seg[-1][2] - t0) / 1e3, foreign_total))


if __name__ == '__main__':
    main(sys.argv[1])
