"""Per-scene LSTMPredictor calls with graph replay against eager calls and the stored reference predictions of
tests/golden/real_eval.npz (found the memset-node bug: replays of scenes with another NaN pattern were wrong).
usage (gpurun): python tools/diag/graph_predictor_probe.py"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from tests import helpers
from trajnetplusplusbaselines_amd import data
from trajnetplusplusbaselines_amd.lstm import LSTMPredictor
from trajnetplusplusbaselines_amd.lstm.lstm import _GraphedForward

z = np.load(os.path.join(helpers.GOLDEN, 'real_eval.npz'))
xy_all, split = z['f0_xy'], z['f0_split']
want = z['f0_pred_prim']
scenes = [xy_all[:, split[s]:split[s + 1]] for s in range(len(split) - 1)]
model, _ = helpers.real_model('cuda')
fast, slow = LSTMPredictor(model), LSTMPredictor(model)
fast.graph_replay, slow.graph_replay = True, False
for rnd in range(3):
    for s, xy in enumerate(scenes):
        paths = data.xy_to_paths(xy)
        n0 = sum(e.replays for e in (model._graphs or {}).values() if isinstance(e, _GraphedForward))
        a = fast(paths, np.zeros((xy.shape[1], 2)), n_predict=12, obs_length=9)
        n1 = sum(e.replays for e in (model._graphs or {}).values() if isinstance(e, _GraphedForward))
        b = slow(paths, np.zeros((xy.shape[1], 2)), n_predict=12, obs_length=9)
        ea, eb = np.abs(a[0][0] - want[:, s]).max(), np.abs(b[0][0] - want[:, s]).max()
        if ea > 1e-4 or eb > 1e-4 or n1 > n0:
            print('round %d scene %2d N=%2d replayed=%d  |fast-ref| %.2e  |slow-ref| %.2e' % (rnd, s, xy.shape[1], n1 - n0, ea, eb))
