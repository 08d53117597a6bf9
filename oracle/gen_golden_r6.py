"""Round-6 fixtures, generated from the imported reference in the BUILD CONTAINER (never on the GPU box):

  python -m oracle.gen_golden_r6

tests/golden/lstm_poolblur.npz
    The reference's ``LSTM.forward`` with ``GridBasedPooling(pool_size != 1 and / or blur_size != 1)`` -- the options its own
    known-answer tests use (tests/test_pooling.py:9-22,86-99) and the trainer never sets (lstm/gridbased_pooling.py:297-304:
    fine grid of n * pool_size cells, avg_pool2d blur, lp_pool2d reduction) -- for the three grid types, odd and even blur,
    on a dense and on a ragged crowd (NaN tracks), free-running and teacher-forced.  Single-threaded reference.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import ref_import  # noqa: E402
from trajnetplusplusbaselines_amd import synth  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')

CASES = [dict(type='occupancy', n=4, pool_size=4, blur_size=3, arch='one_layer', dims=[], latent=16, out_dim=32),
         dict(type='directional', n=6, pool_size=2, blur_size=1, arch='one_layer', dims=[], latent=16, out_dim=64),
         dict(type='directional', n=5, pool_size=1, blur_size=3, arch='two_layer', dims=[48], latent=16, out_dim=32),
         dict(type='social', n=4, pool_size=2, blur_size=2, arch='two_layer', dims=[128], latent=8, out_dim=64),     # even blur: 9 x 9 map
         dict(type='social', n=8, pool_size=3, blur_size=5, arch='one_layer', dims=[], latent=4, out_dim=32)]


def main():
    torch.set_num_threads(1)
    ref = ref_import.import_reference()
    out = {'n_cases': np.int64(len(CASES))}
    for k, c in enumerate(CASES):
        torch.manual_seed(100 + k)
        pool = ref.GridBasedPooling(type_=c['type'], hidden_dim=128, cell_side=0.6, n=c['n'], pool_size=c['pool_size'],
                                    blur_size=c['blur_size'], out_dim=c['out_dim'], embedding_arch=c['arch'], layer_dims=c['dims'],
                                    latent_dim=c['latent'])
        model = ref.LSTM(pool=pool).eval()
        pre = 'c%d_' % k
        for key, v in c.items():
            out[pre + 'cfg_' + key] = np.asarray(v)
        for key, v in model.state_dict().items():
            out[pre + 'sd_' + key] = v.numpy().copy()
        for tag, (xy, split) in (('lin', synth.linear_crowd(3, 6, seed=21 + k)), ('rag', synth.ragged_crowd(5, 1, 9, seed=31 + k))):
            goals = torch.zeros(xy.shape[1], 2)
            with torch.no_grad():
                rel_n, pred_n = model(xy[:9].clone(), goals.clone(), split, n_predict=12)
                rel_t, pred_t = model(xy[:9].clone(), goals.clone(), split, prediction_truth=xy[9:20].clone())
            out.update({pre + tag + '_xy': xy.numpy(), pre + tag + '_split': split.numpy(),
                        pre + tag + '_rel_npredict': rel_n.numpy(), pre + tag + '_pred_npredict': pred_n.numpy(),
                        pre + tag + '_rel_truth': rel_t.numpy(), pre + tag + '_pred_truth': pred_t.numpy()})
        print('case %d: %s n=%d pool_size=%d blur_size=%d' % (k, c['type'], c['n'], c['pool_size'], c['blur_size']))
    np.savez_compressed(os.path.join(OUT, 'lstm_poolblur.npz'), **out)
    print('lstm_poolblur.npz')


if __name__ == '__main__':
    main()
