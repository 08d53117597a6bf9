"""Round 6: the recurrent step at SMALL batches (one scene per call = the evaluator, batch_size 8 = the trainer).

  python tools/diag/small_step_probe.py gemm            tile variants of tnp_linear_forward at small M (old 24/29/30 vs 40..45)
  python tools/diag/small_step_probe.py forward [tag]   ms per inference forward of the headline model for several crowds under
                                                        the CURRENT environment (TNP_SPARSE_TILE, TNP_SKINNY_MAX_M, TNP_SPARSE_MIN_WG)
                                                        and gates variants; outputs saved to /tmp/ssp_<tag>.npz
  python tools/diag/small_step_probe.py sweep           runs `forward` as child processes under a list of environments and
                                                        compares every output with the first (max |difference|)
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

CROWDS = ((1, 4), (1, 36), (1, 64), (8, 39), (16, 32), (32, 32), (64, 32))


def timeit(f, n=200, warm=20):
    for _ in range(warm):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def gemm():
    from trajnetplusplusbaselines_amd import _lib
    shapes = ((256, 1024, 'second embedding layer'), (1024, 256, 'its data gradient'), (576, 512, 'gates data gradient'),
              (256, 288, 'config-3 first layer'))
    for (N, K, what) in shapes:
        W = torch.randn(N, K, device='cuda') / K ** 0.5
        bias = torch.randn(N, device='cuda')
        for M in (4, 36, 128, 310, 512, 1024, 2048):
            x = torch.randn(M, K, device='cuda')
            out = torch.empty(M, N, device='cuda')
            ref = torch.relu(x.double() @ W.double().t() + bias.double())
            row = []
            for v in (24, 29, 30, 40, 41, 42, 43, 44, 45):
                try:
                    us = timeit(lambda: _lib.linear_forward(x, W, bias, relu=True, variant=v, out=out))
                    err = float((out.double() - ref).abs().max())
                    row.append('v%d %.1f (%.0e)' % (v, us, err))
                except Exception:
                    row.append('v%d n/a' % v)
            print('%-24s M=%4d N=%4d K=%4d: %s' % (what, M, N, K, '  '.join(row)), flush=True)


def forward(tag):
    from trajnetplusplusbaselines_amd import synth
    from trajnetplusplusbaselines_amd.lstm import LSTM, GridBasedPooling
    torch.manual_seed(0)
    pool = GridBasedPooling(type_='social', hidden_dim=128, cell_side=0.6, n=16, out_dim=256, embedding_arch='two_layer',
                            layer_dims=[1024], latent_dim=16)
    model = LSTM(pool=pool).cuda().eval()
    gates = [int(v) for v in os.environ.get('SSP_GATES', '0').split(',')]
    saved = {}
    for scenes, agents in CROWDS:
        xy, split = synth.linear_crowd(scenes, agents, seed=1)
        obs, goals = xy[:9].cuda(), torch.zeros(xy.shape[1], 2).cuda()
        row = []
        for gv in gates:
            model.kernel_variant = gv << 8
            with torch.no_grad():
                us = timeit(lambda: model(obs, goals, split, n_predict=12), n=100, warm=10)
                rel, pred = model(obs, goals, split, n_predict=12)
            saved['%dx%d_g%d' % (scenes, agents, gv)] = pred.cpu().numpy()
            row.append('gates v%d %.3f ms' % (gv, us / 1e3))
        print('[%s] %3d scenes x %2d agents (M = %4d): %s' % (tag, scenes, agents, xy.shape[1], '   '.join(row)), flush=True)
    np.savez('/tmp/ssp_%s.npz' % tag, **saved)


def sweep():
    envs = [('old', {'TNP_SKINNY_MAX_M': '0', 'TNP_SPARSE_TILE': '64,2'}),
            ('skinny', {'TNP_SPARSE_TILE': '64,2', 'SSP_GATES': '0,30,31,32,33,34'}),
            ('auto', {}),
            ('minwg128', {'TNP_SPARSE_MIN_WG': '128'}),
            ('minwg64', {'TNP_SPARSE_MIN_WG': '64'}),
            ('minwg512', {'TNP_SPARSE_MIN_WG': '512'}),
            ('t32_2', {'TNP_SPARSE_TILE': '32,2'}), ('t32_1', {'TNP_SPARSE_TILE': '32,1'}), ('t16_1', {'TNP_SPARSE_TILE': '16,1'}),
            ('t8_1', {'TNP_SPARSE_TILE': '8,1'}), ('t4_1', {'TNP_SPARSE_TILE': '4,1'})]
    base = None
    for tag, env in envs:
        e = dict(os.environ)
        e.update(env)
        subprocess.run([sys.executable, os.path.abspath(__file__), 'forward', tag], env=e, check=False)
        try:
            cur = dict(np.load('/tmp/ssp_%s.npz' % tag))
        except Exception as ex:
            print('[%s] no output (%s)' % (tag, ex))
            continue
        if base is None:
            base = cur
            continue
        worst = {}
        for k, v in cur.items():
            ref = base.get(k.rsplit('_g', 1)[0] + '_g0')
            if ref is not None:
                same = np.array_equal(np.isnan(v), np.isnan(ref))
                worst[k] = (float(np.nanmax(np.abs(v - ref))) if same else float('inf'))
        print('[%s] max |pred - pred(old)| per crowd: %s' % (tag, '  '.join('%s %.1e' % kv for kv in sorted(worst.items()))), flush=True)


if __name__ == '__main__':
    what = sys.argv[1] if len(sys.argv) > 1 else 'sweep'
    if what == 'gemm':
        gemm()
    elif what == 'forward':
        forward(sys.argv[2] if len(sys.argv) > 2 else 'x')
    else:
        sweep()
