"""classical.socialforce.predict (reference classical/socialforce.py:10-111): social-force rollout of a scene, and a
batched form for many scenes per launch (tnp_sf_rollout, csrc/classical.hip)."""
import numpy as np
import torch

from .. import _lib
from ._common import device, scene_init, starts_tensor


def rollout_batch(states, sizes, sf_params=(0.5, 2.1, 0.3), n_predict=12, fps=20):
    """states [M, 6] float64 (x, y, vx, vy, goal_x, goal_y) of all scenes concatenated, sizes = agents per scene.
    -> [n_predict, M, 2] float64 positions, one row every 8 simulator steps starting with the first
    (classical/socialforce.py:84-95)."""
    dev = device()
    sampling_rate = int(fps / 2.5)
    st, M = starts_tensor(sizes, dev)
    state0 = torch.tensor(np.ascontiguousarray(states, dtype=np.float64), device=dev)
    out = torch.empty(n_predict, M, 2, dtype=torch.float64, device=dev)
    _lib.check(_lib.lib().tnp_sf_rollout(_lib.ptr(state0), _lib.ptr(st), len(sizes), M, int(max(sizes)),
                                         n_predict * sampling_rate, sampling_rate, float(sf_params[0]), float(sf_params[1]),
                                         float(sf_params[2]), 1.0 / fps, _lib.ptr(out), _lib.stream_ptr()), 'tnp_sf_rollout')
    return out.cpu().numpy()


def predict(input_paths, dest_dict=None, dest_type='interp', sf_params=[0.5, 2.1, 0.3],
            predict_all=True, n_predict=12, obs_length=9):
    rows = scene_init(input_paths, obs_length, n_predict, dest_dict, dest_type)
    if len(rows) != 0:
        states = np.array([[r[0], r[1], r[2], r[3], r[5], r[6]] for r in rows])
        out = rollout_batch(states, [len(rows)], sf_params, n_predict)
    else:   # stationary (classical/socialforce.py:96-99)
        start_frame = input_paths[0][obs_length - 1].frame
        p = [t for t in input_paths[0] if t.frame == start_frame][0]
        out = np.stack([[[p.x, p.y]] for _ in range(n_predict)])
    primary_track = out[:, 0, 0:2]
    neighbours_tracks = out[:, 1:, 0:2] if predict_all else []
    return {0: (primary_track, neighbours_tracks)}


def predict_scenes(scenes, n_predict=12, modes=1, obs_length=9, start_length=0, args=None, dest_dict=None, dest_type='interp',
                  sf_params=(0.5, 2.1, 0.3), predict_all=True):
    """``predict`` for MANY scenes in one launch (the evaluator feed of BASELINE config 5; the reference's classical
    evaluator predicts one scene per call on 12 joblib workers, classical/trajnet_evaluator.py:14-27, 86-92).  ``scenes``: a list
    of ``paths`` or of ``(paths, scene_goal)`` pairs (the goal is unused, as in the reference's call).  Scenes never interact, so
    every result equals ``predict(paths, ...)`` of that scene bit for bit (tests/test_classical_ref.py)."""
    paths_list = [sc[0] if isinstance(sc, tuple) else sc for sc in scenes]
    rows = [scene_init(p, obs_length, n_predict, dest_dict, dest_type) for p in paths_list]
    live = [k for k, r in enumerate(rows) if len(r)]
    results = [None] * len(paths_list)
    if live:
        states = np.array([[r[0], r[1], r[2], r[3], r[5], r[6]] for k in live for r in rows[k]])
        sizes = [len(rows[k]) for k in live]
        out = rollout_batch(states, sizes, sf_params, n_predict)
        lo = 0
        for k, n in zip(live, sizes):
            o = out[:, lo:lo + n]
            results[k] = {0: (o[:, 0, 0:2], o[:, 1:, 0:2] if predict_all else [])}
            lo += n
    for k, r in enumerate(results):
        if r is None:       # nobody present at the last observed frame: the wrapper's stationary branch
            results[k] = predict(paths_list[k], dest_dict, dest_type, list(sf_params), predict_all, n_predict, obs_length)
    return results
