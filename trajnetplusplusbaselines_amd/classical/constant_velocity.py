"""classical.constant_velocity.predict (reference classical/constant_velocity.py:4-20) on the GPU."""
import numpy as np
import torch

from .. import _lib
from .. import data as trajdata
from ._common import device


def predict_xy(xy, n_predict=12):
    """xy [T, N, 2] float64 -> [n_predict, N, 2] float64."""
    dev = device()
    last = torch.tensor(np.ascontiguousarray(xy[-1]), dtype=torch.float64, device=dev)
    prev = torch.tensor(np.ascontiguousarray(xy[-2]), dtype=torch.float64, device=dev)
    N = last.shape[0]
    out = torch.empty(n_predict, N, 2, dtype=torch.float64, device=dev)
    _lib.check(_lib.lib().tnp_constant_velocity(_lib.ptr(last), _lib.ptr(prev), N, n_predict, _lib.ptr(out),
                                                _lib.stream_ptr()), 'tnp_constant_velocity')
    return out.cpu().numpy()


def predict(input_paths, predict_all=True, n_predict=12, obs_length=9):
    xy = trajdata.paths_to_xy(input_paths)
    # the reference extrapolates from the last two rows of xy (the full path, classical/constant_velocity.py:9-12)
    output_scenes = predict_xy(xy, n_predict)
    return {0: (output_scenes[-n_predict:, 0], output_scenes[-n_predict:, 1:])}


def predict_scenes(scenes, n_predict=12, modes=1, obs_length=9, start_length=0, args=None, predict_all=True):
    """``predict`` for many scenes in one launch (tracks of all scenes side by side); equal to the per-scene calls bit for bit."""
    paths_list = [sc[0] if isinstance(sc, tuple) else sc for sc in scenes]
    xys = [trajdata.paths_to_xy(p) for p in paths_list]
    last = np.concatenate([xy[-1] for xy in xys], axis=0)
    prev = np.concatenate([xy[-2] for xy in xys], axis=0)
    out = predict_xy(np.stack([prev, last]), n_predict)
    results, lo = [], 0
    for xy in xys:
        o = out[:, lo:lo + xy.shape[1]]
        results.append({0: (o[-n_predict:, 0], o[-n_predict:, 1:])})
        lo += xy.shape[1]
    return results
