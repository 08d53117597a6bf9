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
