"""classical.orca.predict (reference classical/orca.py:10-134): ORCA rollout of a scene, and a batched form
(tnp_orca_rollout, csrc/classical.hip)."""
import numpy as np
import torch

from .. import _lib
from ._common import device, scene_init, starts_tensor

MAX_SPEED_MULTIPLIER = 1.3  # with respect to initial speed (classical/orca.py:8)


def rollout_batch(pos, vel, speed, goals, sizes, orca_params=(1.5, 1.5, 0.4), n_predict=12, fps=20, want_neighbors=False):
    """pos, vel, goals [M, 2], speed [M] (float64 as the wrapper computes them), sizes = agents per scene.
    -> [n_predict, M, 2] float32 positions sampled at simulator steps 8, 16, ..., 96 (classical/orca.py:99-109)."""
    dev = device()
    sampling_rate = int(fps / 2.5)
    st, M = starts_tensor(sizes, dev)
    pos0 = torch.tensor(np.asarray(pos, dtype=np.float32), device=dev)
    vel0 = torch.tensor(np.asarray(vel, dtype=np.float32), device=dev)
    goals_t = torch.tensor(np.asarray(goals, dtype=np.float64), device=dev)
    speed_t = torch.tensor(np.asarray(speed, dtype=np.float64), device=dev)
    max_speed = torch.tensor(np.asarray(MAX_SPEED_MULTIPLIER * np.asarray(speed, dtype=np.float64), dtype=np.float32),
                             device=dev)
    out = torch.empty(n_predict, M, 2, dtype=torch.float32, device=dev)
    nbr = torch.full((M, 16), -1, dtype=torch.int32, device=dev) if want_neighbors else None
    _lib.check(_lib.lib().tnp_orca_rollout(_lib.ptr(pos0), _lib.ptr(vel0), _lib.ptr(goals_t), _lib.ptr(speed_t),
                                           _lib.ptr(max_speed), _lib.ptr(st), len(sizes), M, int(max(sizes)),
                                           sampling_rate * n_predict + 1, sampling_rate, 1.0 / fps, float(orca_params[0]),
                                           10, float(orca_params[1]), float(orca_params[2]), _lib.ptr(out), _lib.ptr(nbr),
                                           _lib.stream_ptr()), 'tnp_orca_rollout')
    res = out.cpu().numpy()
    return (res, nbr.cpu().numpy()) if want_neighbors else res


def predict(input_paths, dest_dict=None, dest_type='interp', orca_params=[1.5, 1.5, 0.4],
            predict_all=True, n_predict=12, obs_length=9):
    rows = scene_init(input_paths, obs_length, n_predict, dest_dict, dest_type, allow_vel_dest=False)
    if len(rows) == 0:
        return {0: (np.zeros((n_predict, 0)), [])}
    pos = np.array([[r[0], r[1]] for r in rows])
    vel = np.array([[r[2], r[3]] for r in rows])
    speed = np.array([r[4] for r in rows])
    goals = np.array([[r[5], r[6]] for r in rows])
    states = rollout_batch(pos, vel, speed, goals, [len(rows)], orca_params, n_predict)
    primary_track = states[:, 0, 0:2]
    neighbours_tracks = states[:, 1:, 0:2] if predict_all else []
    return {0: (primary_track, neighbours_tracks)}


def predict_scenes(scenes, n_predict=12, modes=1, obs_length=9, start_length=0, args=None, dest_dict=None, dest_type='interp',
                  orca_params=(1.5, 1.5, 0.4), predict_all=True):
    """``predict`` for many scenes in one launch; every result equals the per-scene call bit for bit (see
    ``socialforce.predict_scenes``)."""
    paths_list = [sc[0] if isinstance(sc, tuple) else sc for sc in scenes]
    rows = [scene_init(p, obs_length, n_predict, dest_dict, dest_type, allow_vel_dest=False) for p in paths_list]
    live = [k for k, r in enumerate(rows) if len(r)]
    results = [{0: (np.zeros((n_predict, 0)), [])} for _ in paths_list]
    if live:
        flat = [r for k in live for r in rows[k]]
        sizes = [len(rows[k]) for k in live]
        out = rollout_batch(np.array([[r[0], r[1]] for r in flat]), np.array([[r[2], r[3]] for r in flat]),
                            np.array([r[4] for r in flat]), np.array([[r[5], r[6]] for r in flat]), sizes, orca_params, n_predict)
        lo = 0
        for k, n in zip(live, sizes):
            o = out[:, lo:lo + n]
            results[k] = {0: (o[:, 0, 0:2], o[:, 1:, 0:2] if predict_all else [])}
            lo += n
    return results
