"""Minimal host-side scene containers the predictor wrappers need.

The reference takes these from the un-vendored ``trajnetplusplustools`` package: ``TrackRow``,
``SceneRow``, ``Reader.paths_to_xy``; and from lstm/utils.py + augmentation.py: ``center_scene`` /
``inverse_scene``.  They are host-side preprocessing (numpy, float64) outside the GPU hot path.
"""
from collections import namedtuple

import numpy as np

TrackRow = namedtuple('TrackRow', ['frame', 'pedestrian', 'x', 'y', 'prediction_number', 'scene_id'])
TrackRow.__new__.__defaults__ = (None, None, None, None, None, None)
SceneRow = namedtuple('SceneRow', ['scene', 'pedestrian', 'start', 'end', 'fps', 'tag'])
SceneRow.__new__.__defaults__ = (None, None, None, None, None, None)


def paths_to_xy(paths):
    """list (primary first) of lists of rows with .frame .pedestrian .x .y -> float64 [T, N, 2], NaN = absent.
    Frames are the primary's frames (sorted); tracks keep the order of `paths`."""
    frames = sorted(set(r.frame for r in paths[0]))
    frame_to_index = {f: i for i, f in enumerate(frames)}
    xy = np.full((len(frames), len(paths), 2), np.nan)
    for ped_index, path in enumerate(paths):
        for row in path:
            i = frame_to_index.get(row.frame)
            if i is None:
                continue
            xy[i, ped_index] = [row.x, row.y]
    return xy


def rotate_path(xy, theta):
    """Rotate every [.., 2] point by `theta` (row-vector convention of reference lstm/utils.py:24-30)."""
    ct, st = np.cos(theta), np.sin(theta)
    r = np.array([[ct, st], [-st, ct]])
    return np.einsum('ptc,ci->pti', xy, r)


def center_scene(xy, obs_length=9, ped_id=0, goals=None):
    """Translate the scene so the primary's last observed position is the origin and rotate it so that the
    primary's last observed displacement points along +y (reference lstm/utils.py:32-51).
    Returns (xy, rotation, center[, goals])."""
    center = xy[obs_length - 1, ped_id].copy()
    xy = xy - center[np.newaxis, np.newaxis, :]
    last, second_last = xy[obs_length - 1, ped_id], xy[obs_length - 2, ped_id]
    heading = np.arctan2(last[1] - second_last[1], last[0] - second_last[0])
    rotation = -heading + np.pi / 2
    xy = rotate_path(xy, rotation)
    if goals is None:
        return xy, rotation, center
    goals = np.asarray(goals)[np.newaxis, :, :] - center[np.newaxis, np.newaxis, :]
    return xy, rotation, center, rotate_path(goals, rotation)[0]


def inverse_scene(xy, rotation, center):
    """Undo center_scene (reference augmentation.py:65-68)."""
    return rotate_path(xy, -rotation) + center[np.newaxis, np.newaxis, :]
