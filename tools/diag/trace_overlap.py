"""Do kernels of different HIP streams overlap in time?  Reads a rocprofv3 --kernel-trace rocpd database and reports, over the
last part of the run: busy time per queue, wall span, time with >= 2 kernels running."""
import glob
import sqlite3
import sys

for path in sys.argv[1:] or glob.glob('gpurun_out/prof/*.db'):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tab = [r[0] for r in cur.execute("select name from sqlite_master where type='table' and name like 'rocpd_kernel_dispatch%'")][0]
    cols = [r[1] for r in cur.execute('pragma table_info(%s)' % tab)]
    print('columns:', cols)
    qcol = 'queue_id' if 'queue_id' in cols else ('stream_id' if 'stream_id' in cols else None)
    rows = list(cur.execute('select start, end, %s from %s order by start' % (qcol or '0', tab)))
    rows = rows[len(rows) // 2:]          # steady state
    span = rows[-1][1] - rows[0][0]
    ev = sorted([(s, 1) for s, e, q in rows] + [(e, -1) for s, e, q in rows])
    active, last, t_ge1, t_ge2 = 0, ev[0][0], 0, 0
    for t, d in ev:
        if active >= 1:
            t_ge1 += t - last
        if active >= 2:
            t_ge2 += t - last
        active += d
        last = t
    per_q = {}
    for s, e, q in rows:
        per_q[q] = per_q.get(q, 0) + (e - s)
    print('%d dispatches, span %.3f ms, some kernel running %.3f ms, >= 2 kernels running %.3f ms' % (len(rows), span / 1e6, t_ge1 / 1e6, t_ge2 / 1e6))
    for q, b in sorted(per_q.items()):
        print('  queue %s: busy %.3f ms' % (q, b / 1e6))
    durs = sorted(e - s for s, e, q in rows)
    print('  kernel duration: median %.1f us, mean %.1f us' % (durs[len(durs) // 2] / 1e3, sum(durs) / len(durs) / 1e3))
