"""Diagnostic: per-parameter gradient error of the headline model against tests/golden/train_full.npz, sparse and dense backward."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import helpers
from trajnetplusplusbaselines_amd import synth
from trajnetplusplusbaselines_amd.lstm import PredictionLoss

z = np.load(os.path.join(helpers.GOLDEN, 'train_full.npz'))
r = np.load(os.path.join(helpers.GOLDEN, 'real_cases.npz'))
for tag in sys.argv[1:] or ['synth', 'students']:
    if tag == 'synth':
        xy, split = synth.linear_crowd(64, 32, seed=100)
    else:
        xy, split = torch.tensor(r[tag + '_raw_xy'], dtype=torch.float32), torch.tensor(r[tag + '_split'])
    for sparse in (True, False):
        model, _ = helpers.real_model('cuda')
        model.train()
        model.sparse_backward = sparse
        M = xy.shape[1]
        rel, pred = model(xy[:9].clone(), torch.zeros(M, 2), split, xy[9:20].clone())
        loss = PredictionLoss()(rel[-12:], (xy[9:21] - xy[8:20]).cuda(), split) * (split.numel() - 1)
        loss.backward()
        print(tag, 'sparse_backward', sparse, 'loss', float(loss), 'ref', float(z[tag + '_loss']))
        for name, p in model.named_parameters():
            key = tag + '_grad_' + name
            if p.grad is None:
                continue
            g = p.grad.cpu().numpy().astype(np.float64)
            if key in z.files:
                w = z[key].astype(np.float64)
                print('   %-45s full   err %.2e (absmax %.2e)' % (name, np.abs(g - w).max() / np.abs(w).max(), np.abs(w).max()))
            else:
                s = helpers.sketch(g)
                e = {part: np.abs(s[part] - z[key + '@' + part]).max() / max(1e-30, np.abs(z[key + '@' + part]).max()) for part in ('rowsum', 'colsum', 'projR', 'projL', 'samples')}
                print('   %-45s sketch ' % name + ' '.join('%s %.2e' % kv for kv in e.items()))
