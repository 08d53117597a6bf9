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


def _scene_tracks(paths, predict_all, obs_length):
    start_frame = paths[0][obs_length - 1].frame
    if not predict_all:
        paths = paths[0:1]
    tracks, index = [], []
    for i, path in enumerate(paths):
        past_path = [t for t in path if t.frame <= start_frame]
        if start_frame not in [t.frame for t in past_path] or len(past_path) < 2:
            continue
        tracks.append(past_path)
        index.append(i)
    return tracks, index


def _predict_tracks(tracks, noise, n_predict):
    """tracks of any lengths, noise [n_tracks, 5, n_predict + 1, 6] -> list of [n_predict, 2]: one launch per track length"""
    results = [None] * len(tracks)
    by_len = {}
    for k, tr in enumerate(tracks):
        by_len.setdefault(len(tr), []).append(k)
    for length, ks in by_len.items():
        obs = np.array([[(r.x, r.y) for r in tracks[k]] for k in ks], dtype=np.float64)
        pred = predict_batch(obs, n_predict, noise=noise[ks])
        for k, p in zip(ks, pred):
            results[k] = p
    return results


def _pack(results, index):
    by_path = dict(zip(index, results))
    primary_track = by_path.get(0)
    neighbours = [by_path[i] for i in sorted(by_path) if i != 0]
    neighbours_tracks = np.array(neighbours).transpose(1, 0, 2) if len(neighbours) else []
    return {0: (primary_track, neighbours_tracks)}


def predict(paths, predict_all=True, n_predict=12, obs_length=9, rng=None):
    """``rng`` (not in the reference's signature): a ``numpy.random.RandomState``; default = numpy's global one, which
    is what pykalman's ``sample(random_state=None)`` draws from.  The draws are taken track by track in the order of
    ``paths``, 5 samples x (n_predict + 1) steps x 6 components each -- the order in which the reference's loop
    (classical/kalman.py:22-60) asks for samples.  A seeded run is reproducible against ``oracle/classical_stubs.py`` (the
    driving stubs of tests/test_classical_ref.py), NOT against the real pykalman: ``KalmanFilter.sample(initial_state=...)``
    draws no transition noise at t = 0 and transforms its normals with ``rng.multivariate_normal`` (SVD), not with the
    Cholesky factors used here, so the two random streams part at the first step.  The predictive MEAN (what the reference
    returns is a mean of 5 noisy samples around it) is the comparable quantity."""
    tracks, index = _scene_tracks(paths, predict_all, obs_length)
    noise = (rng or np.random).standard_normal((len(tracks), 5, n_predict + 1, 6))
    return _pack(_predict_tracks(tracks, noise, n_predict), index)


def predict_scenes(scenes, n_predict=12, modes=1, obs_length=9, start_length=0, args=None, predict_all=True, rng=None):
    """``predict`` for many scenes: the tracks of ALL scenes go to the GPU together (one launch per track length); the normal
    draws are taken scene by scene, track by track -- the stream a loop of per-scene ``predict`` calls would consume -- so every
    result equals the per-scene call under the same seed bit for bit.  ``scenes``: list of ``paths`` or ``(paths, scene_goal)``."""
    paths_list = [sc[0] if isinstance(sc, tuple) else sc for sc in scenes]
    per_scene = [_scene_tracks(p, predict_all, obs_length) for p in paths_list]
    counts = [len(t) for t, _ in per_scene]
    noise = (rng or np.random).standard_normal((sum(counts), 5, n_predict + 1, 6))
    flat = _predict_tracks([tr for t, _ in per_scene for tr in t], noise, n_predict)
    out, lo = [], 0
    for (t, index), n in zip(per_scene, counts):
        out.append(_pack(flat[lo:lo + n], index))
        lo += n
    return out
