"""DEVICE time per launch of tnp_linear_forward's tile variants (rocprofv3 kernel trace of a child run of this script): a
back-to-back loop from Python bottoms out at ~8.5 us of host time per launch, which hides everything a small GEMM does.

  python tools/diag/gemm_device_times.py            parent: runs the child under rocprofv3, prints the table
  python tools/diag/gemm_device_times.py child      the launches; writes the segment list to /tmp/gdt_segments.json
"""
import glob
import json
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

SHAPES = ((256, 1024, 'second embedding layer'), (1024, 256, 'its data gradient'), (448, 512, 'gates data gradient'),
          (256, 288, 'config-3 first layer'), (1024, 4096, 'dense first layer'))
MS = (4, 36, 128, 310, 512, 1024)
VARIANTS = (24, 25, 27, 29, 30, 31, 40, 41, 42, 43, 44, 45)
REPS = 12


def child():
    import torch
    from trajnetplusplusbaselines_amd import _lib
    segs = []
    for (N, K, what) in SHAPES:
        W = torch.randn(N, K, device='cuda') / K ** 0.5
        bias = torch.randn(N, device='cuda')
        for M in MS:
            x = torch.randn(M, K, device='cuda')
            out = torch.empty(M, N, device='cuda')
            for v in VARIANTS:
                try:
                    _lib.linear_forward(x, W, bias, relu=True, variant=v, out=out)
                except Exception:
                    continue
                for _ in range(REPS):
                    _lib.linear_forward(x, W, bias, relu=True, variant=v, out=out)
                torch.cuda.synchronize()
                segs.append([what, M, N, K, v, REPS + 1])
    json.dump(segs, open('/tmp/gdt_segments.json', 'w'))


def parent():
    out_dir = '/tmp/gdt_prof'
    subprocess.run('rm -rf %s' % out_dir, shell=True)
    env = dict(os.environ, TMPDIR='/tmp')
    subprocess.run(['rocprofv3', '--kernel-trace', '-d', out_dir, '-o', 't', '--', sys.executable, os.path.abspath(__file__), 'child'],
                   env=env, cwd='/tmp', stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
    segs = json.load(open('/tmp/gdt_segments.json'))
    db = sqlite3.connect(glob.glob(out_dir + '/**/*.db', recursive=True)[0])
    cur = db.cursor()
    tab = [r[0] for r in cur.execute("select name from sqlite_master where type='table' and name like 'rocpd_kernel_dispatch%'")][0]
    suffix = tab.replace('rocpd_kernel_dispatch', '')
    rows = list(cur.execute("select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch%s d join rocpd_info_kernel_symbol%s s "
                            "on d.kernel_id=s.id order by d.start" % (suffix, suffix)))
    rows = [(n, e - s) for n, s, e in rows if 'gemm_nt' in n or 'gemm_skinny' in n]
    assert len(rows) == sum(s[5] for s in segs), (len(rows), sum(s[5] for s in segs))
    i = 0
    table = {}
    for what, M, N, K, v, cnt in segs:
        d = sorted(r[1] for r in rows[i + 1:i + cnt])        # (first launch of a variant: code object / attribute set-up)
        i += cnt
        table.setdefault((what, N, K, M), []).append((v, d[len(d) // 2] / 1e3))
    for (what, N, K, M), vs in table.items():
        best = min(us for _, us in vs)
        print('%-24s M=%4d N=%4d K=%4d: %s' % (what, M, N, K, '  '.join(('v%d %s%.1f' % (v, '*' if us == best else '', us)) for v, us in vs)), flush=True)


if __name__ == '__main__':
    child() if len(sys.argv) > 1 and sys.argv[1] == 'child' else parent()
