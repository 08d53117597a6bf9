"""Summarise a rocprofv3 --kernel-trace run (rocpd sqlite database) into a per-kernel table (markdown)."""
import glob
import re
import sqlite3
import sys


def summarise(db_path):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    tab = [r[0] for r in cur.execute(
        "select name from sqlite_master where type='table' and name like 'rocpd_kernel_dispatch%'")][0]
    suffix = tab.replace('rocpd_kernel_dispatch', '')
    rows = list(cur.execute(
        "select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
        "from rocpd_kernel_dispatch%s d join rocpd_info_kernel_symbol%s s on d.kernel_id=s.id "
        "group by s.kernel_name order by 3 desc" % (suffix, suffix)))
    total = float(sum(r[2] for r in rows)) or 1.0
    out = ['| kernel | calls | total ms | avg us | min us | max us | % |', '|---|---|---|---|---|---|---|']
    for name, n, tot, avg, mn, mx in rows:
        short = re.sub(r'\.kd$', '', name)
        m = re.match(r'_ZN3tnp14gemm_nt_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb(\d)EEEvNS_8GemmArgsE', short)
        if m:
            short = 'tnp::gemm_nt_kernel<WM%s,WN%s,WK%s,AN%s,BK%s,PF%s,EPI%s,VEC%s>' % m.groups()
        short = short.replace('_ZN3tnp17grid_build_kernelENS_8GridArgsE', 'tnp::grid_build_kernel')
        short = short.replace('_ZN3tnp20track_prepare_kernelENS_8PrepArgsE', 'tnp::track_prepare_kernel')
        out.append('| %s | %d | %.3f | %.2f | %.2f | %.2f | %.1f |' % (short[:110], n, tot / 1e6, avg / 1e3, mn / 1e3,
                                                                        mx / 1e3, 100.0 * tot / total))
    return '\n'.join(out)


if __name__ == '__main__':
    paths = sys.argv[1:] or glob.glob('gpurun_out/prof/*.db')
    for p in paths:
        print('### %s\n' % p)
        print(summarise(p))
        print()
