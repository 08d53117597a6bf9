"""classical.kalman.predict (reference classical/kalman.py:6-73): constant-velocity Kalman filter per track with EM,
smoothing and the mean of 5 sampled continuations, batched over tracks (tnp_kalman_predict, csrc/classical.hip).
The reference draws from numpy's global, unseeded RNG; here the standard-normal draws are an explicit input."""
import numpy as np
import torch

from .. import _lib
from ._common import device


def predict_batch(obs, n_predict=12, n_samples=5, noise=None, rng=None, n_iter=10):
    """obs [n_tracks, T, 2] float64 (complete observations) -> [n_tracks, n_predict, 2] float64."""
    dev = device()
    obs = np.ascontiguousarray(obs, dtype=np.float64)
    n_tracks, T = obs.shape[0], obs.shape[1]
    if noise is None:
        rng = rng or np.random
        noise = rng.standard_normal((n_tracks, n_samples, n_predict + 1, 6))
    z = torch.tensor(np.ascontiguousarray(noise, dtype=np.float64), device=dev)
    o = torch.tensor(obs, device=dev)
    out = torch.empty(n_tracks, n_predict + 1, 2, dtype=torch.float64, device=dev)
    _lib.check(_lib.lib().tnp_kalman_predict(_lib.ptr(o), n_tracks, T, n_iter, n_predict + 1, n_samples, _lib.ptr(z),
                                             1e-5, 0.05 ** 2, _lib.ptr(out), _lib.stream_ptr()), 'tnp_kalman_predict')
    return out.cpu().numpy()[:, 1:]       # first sample corresponds to the last state (classical/kalman.py:52-62)


def predict(paths, predict_all=True, n_predict=12, obs_length=9, rng=None):
    """``rng`` (not in the reference's signature): a ``numpy.random.RandomState``; default = numpy's global one, which
    is what pykalman's ``sample(random_state=None)`` draws from.  The draws are taken track by track in the order of
    ``paths``, 5 samples x (n_predict + 1) steps x 6 components each -- the order in which the reference's loop
    (classical/kalman.py:22-60) consumes them -- so a seeded run is reproducible against a seeded reference run."""
    primary = paths[0]
    start_frame = primary[obs_length - 1].frame
    if not predict_all:
        paths = paths[0:1]
    tracks, index = [], []
    for i, path in enumerate(paths):
        past_path = [t for t in path if t.frame <= start_frame]
        if start_frame not in [t.frame for t in past_path] or len(past_path) < 2:
            continue
        tracks.append(past_path)
        index.append(i)
    noise = (rng or np.random).standard_normal((len(tracks), 5, n_predict + 1, 6))
    # tracks may have different lengths: one launch per length
    results = {}
    by_len = {}
    for k, tr in enumerate(tracks):
        by_len.setdefault(len(tr), []).append(k)
    for length, ks in by_len.items():
        obs = np.array([[(r.x, r.y) for r in tracks[k]] for k in ks], dtype=np.float64)
        pred = predict_batch(obs, n_predict, noise=noise[ks])
        for k, p in zip(ks, pred):
            results[index[k]] = p
    primary_track = results.get(0)
    neighbours = [results[i] for i in sorted(results) if i != 0]
    neighbours_tracks = np.array(neighbours).transpose(1, 0, 2) if len(neighbours) else []
    return {0: (primary_track, neighbours_tracks)}
