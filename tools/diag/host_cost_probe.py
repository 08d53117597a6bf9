"""Where the host time of one eager inference forward goes: the native call (tnp_lstm_forward_ex = ~40-80 kernel launches) against
the Python around it.  usage (gpurun): python tools/diag/host_cost_probe.py"""
import ctypes
import sys
import time

import torch

sys.path.insert(0, '.')
from trajnetplusplusbaselines_amd import _lib, synth                   # noqa: E402
from trajnetplusplusbaselines_amd.lstm import LSTM, GridBasedPooling   # noqa: E402

L = _lib.lib()
orig = L.tnp_lstm_forward_ex
acc = [0.0, 0]


class Timed(object):
    def __call__(self, *a):
        t0 = time.perf_counter()
        rc = orig(*a)
        acc[0] += time.perf_counter() - t0
        acc[1] += 1
        return rc


torch.manual_seed(0)
for kind in ('vanilla', 'directional', 'social'):
    pool = None
    if kind == 'directional':
        pool = GridBasedPooling(type_='directional', hidden_dim=128, cell_side=0.6, n=12, out_dim=256)
    if kind == 'social':
        pool = GridBasedPooling(type_='social', hidden_dim=128, cell_side=0.6, n=16, out_dim=256, embedding_arch='two_layer', layer_dims=[1024], latent_dim=16)
    model = LSTM(pool=pool).cuda().eval()
    for scenes, agents in ((1, 4), (64, 32)):
        xy, split = synth.linear_crowd(scenes, agents, seed=1)
        obs, goals = xy[:9].cuda(), torch.zeros(xy.shape[1], 2).cuda()
        with torch.no_grad():
            for _ in range(30):
                model(obs, goals, split, n_predict=12)
            torch.cuda.synchronize()
            _lib.lib().tnp_lstm_forward_ex = Timed()
            acc[0], acc[1] = 0.0, 0
            t0 = time.perf_counter()
            for _ in range(300):
                model(obs, goals, split, n_predict=12)
                if _ % 8 == 7:
                    torch.cuda.synchronize()       # keep the queue short: enqueue cost without back-pressure
            t = (time.perf_counter() - t0) / 300
            _lib.lib().tnp_lstm_forward_ex = orig
            torch.cuda.synchronize()
        print('%-12s %2d x %2d: %.3f ms per forward (incl. a sync every 8), native call %.3f ms, Python around it %.3f ms'
              % (kind, scenes, agents, t * 1e3, acc[0] / acc[1] * 1e3, (t - acc[0] / acc[1]) * 1e3))
