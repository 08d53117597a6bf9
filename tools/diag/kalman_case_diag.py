import sys, numpy as np
sys.path.insert(0, '/root/repo')
from tests import test_classical_ref as T
from oracle import oracle
ref = np.load('/root/repo/tests/golden/classical_ref.npz', allow_pickle=True)
i = 15
print('kind', ref['kinds'][i] if 'kinds' in ref else None)
tracks = T._kalman_tracks(T.case_paths(ref, i), int(ref['c%d_predict_all' % i]))
print([len(t) for t in tracks])
z = ref['c%d_kalman_noise' % i]
host = np.stack([oracle.kalman_predict(t[None], z[k:k + 1])[0, 1:] for k, t in enumerate(tracks)], axis=1)
want = T._pack(ref['c%d_kalman_core_primary' % i], ref['c%d_kalman_core_neigh' % i])
print('host vs fixture', np.abs(host - want).max())
try:
    import torch
    if torch.cuda.is_available():
        from trajnetplusplusbaselines_amd.classical import kalman
        for k, t in enumerate(tracks):
            got = kalman.predict_batch(t[None], 12, noise=z[k:k+1])
            h = oracle.kalman_predict(t[None], z[k:k + 1])
            print(k, len(t), 'gpu vs host', np.abs(np.asarray(got) - h[:, -np.asarray(got).shape[1]:]).max() if np.asarray(got).shape != h.shape else np.abs(np.asarray(got)-h).max())
except Exception as e:
    import traceback; traceback.print_exc()
