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


def read_ndjson_scenes(path, limit=None, scene_ids=None):
    """Read a TrajNet++ ``.ndjson`` file into ``[(scene_id, paths)]`` with ``paths`` = list of tracks (primary
    first, the others in order of first appearance), each a list of ``TrackRow`` within the scene's frame range.

    Stands in for ``trajnetplusplustools.Reader(path, scene_type='paths').scenes()`` as the reference calls it
    (lstm/data_load_utils.py:33-44, evaluator/trajnet_evaluator.py:25-36); the package itself is not vendored
    (SURVEY.md 8c), so the record layout is taken from the data files: ``{"scene": {id, p, s, e, fps, tag}}`` and
    ``{"track": {f, p, x, y[, prediction_number, scene_id]}}``.
    """
    import json
    by_frame = {}
    scenes = []
    with open(path, 'r') as f:
        for line in f:
            rec = json.loads(line)
            if 'track' in rec:
                t = rec['track']
                if t.get('prediction_number') is not None:
                    continue
                by_frame.setdefault(t['f'], []).append(TrackRow(t['f'], t['p'], t['x'], t['y']))
            elif 'scene' in rec:
                s = rec['scene']
                scenes.append(SceneRow(s['id'], s['p'], s['s'], s['e'], s.get('fps'), s.get('tag')))
    frames = sorted(by_frame)
    out = []
    for sc in scenes:
        if scene_ids is not None and sc.scene not in scene_ids:
            continue
        tracks = {}
        for fr in frames:
            if fr < sc.start or fr > sc.end:
                continue
            for row in by_frame[fr]:
                tracks.setdefault(row.pedestrian, []).append(row)
        if sc.pedestrian not in tracks:
            continue
        paths = [tracks[sc.pedestrian]] + [p for ped, p in tracks.items() if ped != sc.pedestrian]
        out.append((sc.scene, paths))
        if limit is not None and len(out) >= limit:
            break
    return out


def batch_scenes(scenes_xy):
    """Concatenate per-scene ``[T, N_s, 2]`` arrays along the track axis -> (``[T, M, 2]``, ``batch_split [B+1]``);
    the batch assembly of reference lstm/trainer.py:120-131."""
    split = np.cumsum([0] + [int(s.shape[1]) for s in scenes_xy])
    return np.concatenate(scenes_xy, axis=1), split
