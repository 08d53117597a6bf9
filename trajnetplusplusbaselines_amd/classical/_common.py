"""Host-side pieces the reference's classical wrappers share (classical/socialforce.py:15-72, classical/orca.py:14-79):
which agents take part (present at the last observed frame), initial velocity from a stride-3 finite difference,
goal by linear extrapolation 12 steps ahead.  numpy float64, exactly as the wrappers compute them."""
import numpy as np
import torch

from .. import _lib


def vel_state(prev, curr, stride):
    """classical/socialforce.py:57-63 / classical/orca.py:60-66 (speed via arctan2 / cos / sin)."""
    if stride == 0:
        return [0, 0], 0
    diff = np.array([curr.x - prev.x, curr.y - prev.y])
    theta = np.arctan2(diff[1], diff[0])
    speed = np.linalg.norm(diff) / (stride * 0.4)
    return [speed * np.cos(theta), speed * np.sin(theta)], speed


def dest_state(path, length, pred_length):
    """classical/socialforce.py:65-72: scipy interp1d(..., fill_value='extrapolate') evaluated pred_length steps past the
    last sample = the line through the last two samples."""
    if length == 1:
        return [path[-1].x, path[-1].y]
    x1, y1, x0, y0 = path[-1].x, path[-1].y, path[-2].x, path[-2].y
    slope_x, slope_y = (x1 - x0) / 1.0, (y1 - y0) / 1.0
    # scipy's linear extrapolation: y0 + slope * (t - t0) with (t0, y0) the second-to-last sample
    t = float(length - 1 + pred_length) - float(length - 2)
    return [x0 + slope_x * t, y0 + slope_y * t]


def scene_init(input_paths, obs_length, pred_length, dest_dict=None, dest_type='interp', allow_vel_dest=True):
    """-> list of (x, y, vx, vy, speed, gx, gy) for every agent present at the last observed frame, primary first."""
    primary = input_paths[0]
    start_frame = primary[obs_length - 1].frame
    rows = []
    for path in input_paths:
        ped_id = path[0].pedestrian
        past_path = [t for t in path if t.frame <= start_frame]
        future_path = [t for t in path if t.frame > start_frame]
        if start_frame not in [t.frame for t in past_path]:
            continue
        len_path = len(past_path)
        curr = past_path[-1]
        if len_path >= 4:
            stride, prev = 3, past_path[-4]
        else:
            stride, prev = len_path - 1, past_path[-len_path]
        (vx, vy), speed = vel_state(prev, curr, stride)
        if dest_type == 'true':
            if dest_dict is None:
                raise ValueError
            gx, gy = dest_dict[ped_id]
        elif dest_type == 'interp':
            gx, gy = dest_state(past_path, len_path, pred_length)
        elif dest_type == 'vel' and allow_vel_dest:
            gx, gy = pred_length * vx, pred_length * vy
        elif dest_type == 'pred_end':
            gx, gy = future_path[-1].x, future_path[-1].y
        else:
            raise NotImplementedError
        rows.append((curr.x, curr.y, vx, vy, speed, gx, gy))
    return rows


def starts_tensor(sizes, device):
    starts = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    return torch.tensor(starts, dtype=torch.int32, device=device), int(starts[-1])


def device():
    if not torch.cuda.is_available():
        raise RuntimeError('the classical predictors run on a ROCm device; no CPU fallback')
    _lib.lib()
    return torch.device('cuda', torch.cuda.current_device())
