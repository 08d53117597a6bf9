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
    Frames are the primary's frames (sorted); tracks keep the order of `paths`; of two rows of a track in one frame the later
    one stays.  (All rows of the scene are transposed with ``zip`` and placed with ONE numpy assignment: the per-row Python
    loop was the host cost of ``predict_batch``.)"""
    frames = np.array(sorted(set(r.frame for r in paths[0])))
    xy = np.full((len(frames), len(paths), 2), np.nan)
    rows = [r for path in paths for r in path]
    if not rows:
        return xy
    if isinstance(rows[0], tuple):                        # TrackRow (a namedtuple): frame, pedestrian, x, y, ...
        cols = tuple(zip(*rows))
        f, x, y = np.array(cols[0]), np.array(cols[2], dtype=np.float64), np.array(cols[3], dtype=np.float64)
    else:                                                 # any object with the four attributes
        f = np.array([r.frame for r in rows])
        x, y = np.array([r.x for r in rows], dtype=np.float64), np.array([r.y for r in rows], dtype=np.float64)
    ped = np.repeat(np.arange(len(paths)), [len(path) for path in paths])
    i = np.searchsorted(frames, f)
    i = np.minimum(i, len(frames) - 1)
    ok = frames[i] == f
    if not ok.all():
        i, ped, x, y = i[ok], ped[ok], x[ok], y[ok]
    xy[i, ped, 0] = x          # (duplicate (frame, track) pairs: numpy keeps the last assignment, as the row loop did)
    xy[i, ped, 1] = y
    return xy


def xy_to_paths(xy, frame_step=10, first_pedestrian=100):
    """Inverse of ``paths_to_xy``: float [T, N, 2] (NaN = absent, track 0 = the primary, present in every frame) -> list of
    lists of TrackRow.  For tests and for callers that hold scenes as arrays but talk to the predictors' path interface."""
    xy = np.asarray(xy)
    return [[TrackRow(frame_step * t, first_pedestrian + p, float(xy[t, p, 0]), float(xy[t, p, 1]))
             for t in range(xy.shape[0]) if not np.isnan(xy[t, p, 0])] for p in range(xy.shape[1])]


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
        lines = [ln for ln in f.read().split('\n') if ln.strip()]
    # ONE json parse of the whole file (the lines joined into an array) instead of one json.loads per line: the evaluator feed
    # spent two thirds of its time here (tools/diag/predict_dataset_throughput.py)
    for rec in json.loads('[' + ','.join(lines) + ']'):
        t = rec.get('track')
        if t is not None:
            if t.get('prediction_number') is not None:
                continue
            fr = t['f']
            row = TrackRow(fr, t['p'], t['x'], t['y'])
            rows = by_frame.get(fr)
            if rows is None:
                by_frame[fr] = [row]
            else:
                rows.append(row)
        else:
            sc = rec.get('scene')
            if sc is not None:
                scenes.append(SceneRow(sc['id'], sc['p'], sc['s'], sc['e'], sc.get('fps'), sc.get('tag')))
    import bisect
    frames = sorted(by_frame)
    out = []
    for sc in scenes:
        if scene_ids is not None and sc.scene not in scene_ids:
            continue
        tracks = {}
        for fr in frames[bisect.bisect_left(frames, sc.start):bisect.bisect_right(frames, sc.end)]:
            for row in by_frame[fr]:
                tracks.setdefault(row.pedestrian, []).append(row)
        if sc.pedestrian not in tracks:
            continue
        paths = [tracks[sc.pedestrian]] + [p for ped, p in tracks.items() if ped != sc.pedestrian]
        out.append((sc.scene, paths))
        if limit is not None and len(out) >= limit:
            break
    return out


def read_ndjson_columns(path):
    """The test file as COLUMNS -- track rows (frame, pedestrian, x, y) in file order and scene rows (id, primary, start, end) --
    parsed by the native reader (csrc/ndjson_io.cpp: tnp_ndjson_parse; one pass over the bytes instead of one Python dict per
    row).  Returns a dict of numpy arrays, or None when the file holds something the native reader does not take (non-integer
    frames / ids, malformed lines): the caller then uses ``read_ndjson_scenes``.  Track rows with a stored prediction are
    skipped, as there."""
    import ctypes
    from . import _lib
    with open(path, 'rb') as f:
        buf = f.read()
    # (one native pass: 75 ms for a 36 MB file -- chunks parsed on several threads were tried and bought nothing, the rest of this
    # function's 0.15 s is the file read, the line count and the first touch of the column arrays)
    cap = buf.count(b'\n') + 1
    t_f, t_p = np.empty(cap, dtype=np.int64), np.empty(cap, dtype=np.int64)
    t_x, t_y = np.empty(cap, dtype=np.float64), np.empty(cap, dtype=np.float64)
    s_cols = [np.empty(cap, dtype=np.int64) for _ in range(4)]
    nt, ns = ctypes.c_int64(0), ctypes.c_int64(0)
    cp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    rc = _lib.lib().tnp_ndjson_parse(buf, len(buf), cap, cp(t_f), cp(t_p), cp(t_x), cp(t_y), ctypes.byref(nt),
                                     cp(s_cols[0]), cp(s_cols[1]), cp(s_cols[2]), cp(s_cols[3]), ctypes.byref(ns))
    if rc != 0:
        return None
    nt, ns = nt.value, ns.value
    return dict(frame=t_f[:nt], ped=t_p[:nt], x=t_x[:nt], y=t_y[:nt], scene_id=s_cols[0][:ns], scene_ped=s_cols[1][:ns],
                scene_start=s_cols[2][:ns], scene_end=s_cols[3][:ns])


class SceneArrays(object):
    """One test scene as arrays, after ``preprocess_test``: what ``read_ndjson_scenes`` + ``preprocess_test`` + ``paths_to_xy``
    + the bookkeeping of ``write_predictions`` derive from the scene's paths (see ``scenes_from_columns``)."""
    __slots__ = ('scene_id', 'peds', 'xy', 'first_frame', 'frame_diff', 'start', 'end')


def scenes_from_columns(cols, obs_length=9, pred_length=12, limit=None):
    """``read_ndjson_scenes`` -> ``preprocess_test`` -> ``paths_to_xy`` on the columns of ``read_ndjson_columns``, scene by scene
    with numpy instead of row objects: rows of the scene's frame range in (frame, file order); tracks in order of first
    appearance, the primary first; tracks and rows after the primary's ``obs_length``-th frame dropped (evaluator/
    write_utils.py:34-40); ``xy [T, N, 2]`` on the primary's frames, NaN = absent.  Returns a list of ``SceneArrays``, or None
    when a scene needs the general path (a primary with two rows in one frame)."""
    F, P, X, Y = cols['frame'], cols['ped'], cols['x'], cols['y']
    order = np.argsort(F, kind='stable')
    Fs = F[order]
    los = np.searchsorted(Fs, cols['scene_start'], 'left')
    his = np.searchsorted(Fs, cols['scene_end'], 'right')
    out = []
    seq_length = obs_length + pred_length
    for k in range(len(los)):
        idx = order[los[k]:his[k]]
        f, pd = Fs[los[k]:his[k]], P[idx]
        prim = cols['scene_ped'][k]
        pf = f[pd == prim]
        if pf.size == 0:
            continue                                    # the reader drops scenes whose primary has no row in the range
        if pf.size > 1 and (pf[1:] == pf[:-1]).any():
            return None
        last_obs = pf[min(obs_length, pf.size) - 1]
        cut = np.searchsorted(f, last_obs, 'right')
        f, pd, idx = f[:cut], pd[:cut], idx[:cut]
        pf = pf[:min(obs_length, pf.size)]
        u, first, inv = np.unique(pd, return_index=True, return_inverse=True)
        rank = np.empty(u.size, dtype=np.int64)
        by_first = np.argsort(first, kind='stable')
        pi = int(np.searchsorted(u, prim))
        by_first = np.concatenate([[pi], by_first[by_first != pi]])           # primary first, then order of first appearance
        rank[by_first] = np.arange(u.size)
        col = rank[inv]
        T = pf.size
        ti = np.minimum(np.searchsorted(pf, f), T - 1)
        ok = pf[ti] == f
        xy = np.full((T, u.size, 2), np.nan)
        if not ok.all():
            ti, col, idx = ti[ok], col[ok], idx[ok]
        xy[ti, col, 0] = X[idx]
        xy[ti, col, 1] = Y[idx]
        sc = SceneArrays()
        sc.scene_id, sc.peds, sc.xy = int(cols['scene_id'][k]), u[by_first], xy
        if T >= 2:
            sc.frame_diff = int(pf[1] - pf[0])
            sc.first_frame = int(pf[min(obs_length, T) - 1]) + sc.frame_diff if T >= obs_length else None
            sc.start, sc.end = int(pf[0]), int(pf[0]) + (seq_length - 1) * sc.frame_diff
        else:
            sc.frame_diff = sc.first_frame = sc.start = sc.end = None
        out.append(sc)
        if limit is not None and len(out) >= limit:
            break
    return out


def format_predictions(pred, split, scenes):
    """bytes of the prediction-file lines of a batch (``write_predictions``' layout) from ``pred`` float64
    [modes, pred_length, M, 2] and the batch's ``SceneArrays``: csrc/ndjson_io.cpp, tnp_format_predictions."""
    import ctypes
    from . import _lib
    pred = np.ascontiguousarray(pred, dtype=np.float64)
    n_modes, pred_length, M = pred.shape[0], pred.shape[1], pred.shape[2]
    i64 = lambda v: np.ascontiguousarray(v, dtype=np.int64)
    split, ped = i64(split), i64(np.concatenate([sc.peds for sc in scenes]))
    sid, ff, fd = i64([sc.scene_id for sc in scenes]), i64([sc.first_frame for sc in scenes]), i64([sc.frame_diff for sc in scenes])
    st, en = i64([sc.start for sc in scenes]), i64([sc.end for sc in scenes])
    L = _lib.lib()
    cap = L.tnp_format_predictions_bound(n_modes, pred_length, M, len(scenes))
    buf = np.empty(cap, dtype=np.uint8)              # (not zero-filled: the bound is ~2x what is written)
    cp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    n = L.tnp_format_predictions(cp(pred), n_modes, pred_length, M, len(scenes), cp(split), cp(ped), cp(sid), cp(ff), cp(fd),
                                 cp(st), cp(en), buf.ctypes.data_as(ctypes.c_char_p), cap)
    if n < 0:
        raise RuntimeError('tnp_format_predictions: buffer too small')
    return buf[:n].data                              # a memoryview: file.write takes it as it is


def batch_scenes(scenes_xy):
    """Concatenate per-scene ``[T, N_s, 2]`` arrays along the track axis -> (``[T, M, 2]``, ``batch_split [B+1]``);
    the batch assembly of reference lstm/trainer.py:120-131."""
    split = np.cumsum([0] + [int(s.shape[1]) for s in scenes_xy])
    return np.concatenate(scenes_xy, axis=1), split


def random_rotation(xy, goals=None):
    """Rotate a scene (and its goals) by a uniformly random angle drawn from Python's ``random`` exactly as the
    reference does (lstm/utils.py:10-17), so seeded runs see the same augmentation."""
    import math
    import random
    theta = random.random() * 2.0 * math.pi
    ct, st = math.cos(theta), math.sin(theta)
    r = np.array([[ct, st], [-st, ct]])
    if goals is None:
        return np.einsum('ptc,ci->pti', xy, r)
    return np.einsum('ptc,ci->pti', xy, r), np.einsum('tc,ci->ti', goals, r)


def add_noise(observation, thresh=0.005, obs_length=9, ped='primary'):
    """Uniform noise on the observed frames of the primary or of the neighbours (reference augmentation.py:79-87,
    numpy's global RNG, in place)."""
    if ped == 'primary':
        observation[:obs_length, 0] += np.random.uniform(-thresh, thresh, observation[:obs_length, 0].shape)
    elif ped == 'neigh':
        observation[:obs_length, 1:] += np.random.uniform(-thresh, thresh, observation[:obs_length, 1:].shape)
    else:
        raise ValueError
    return observation


class SceneBatcher(object):
    """Pre-tensorised scenes for the training / evaluation loops.

    The reference re-runs ``paths_to_xy`` + ``drop_distant`` + ``center_scene`` per scene per epoch in Python
    (lstm/trainer.py:96-133).  Here every scene is converted once to a float32 ``[T, N_s, 2]`` array (after
    ``drop_distant`` and the optional ``center_scene``), all scenes are stored back to back in ONE device tensor, and a
    batch is a gather of the selected scenes' columns -- optionally rotated per scene on the device (the
    ``random_rotation`` augmentation: one angle per scene from Python's ``random``, applied as a 2x2 matrix to every
    frame).  ``batch(ids)`` returns (``batch_scene [T, M, 2]``, ``batch_scene_goal [M, 2]``, ``batch_split [B+1]``) as
    the trainer builds them (lstm/trainer.py:125-133)."""

    def __init__(self, scenes, goals=None, device='cuda', obs_length=9, normalize_scene=False, drop_distant_r=6.0):
        import torch
        xs, gs = [], []
        for k, xy in enumerate(scenes):
            xy = np.asarray(xy, dtype=np.float64)
            g = np.zeros((xy.shape[1], 2)) if goals is None else np.asarray(goals[k], dtype=np.float64)
            if drop_distant_r is not None:
                d2 = np.sum(np.square(xy - xy[:, 0:1]), axis=2)
                mask = np.nanmin(d2, axis=0) < drop_distant_r ** 2          # lstm/lstm.py:16-22
                xy, g = xy[:, mask], g[mask]
            if normalize_scene:
                xy, _, _, g = center_scene(xy, obs_length, goals=g)
            xs.append(xy)
            gs.append(g)
        self.sizes = np.array([x.shape[1] for x in xs], dtype=np.int64)
        self.starts = np.concatenate([[0], np.cumsum(self.sizes)])
        self.device = torch.device(device)
        self._xy_host = np.ascontiguousarray(np.concatenate(xs, axis=1).astype(np.float32).transpose(1, 0, 2))   # [sum N, T, 2]
        self._goals_host = np.concatenate(gs, axis=0).astype(np.float32)
        self.xy = torch.tensor(np.concatenate(xs, axis=1), dtype=torch.float32, device=self.device)     # [T, sum N, 2]
        self.goals = torch.tensor(np.concatenate(gs, axis=0), dtype=torch.float32, device=self.device)  # [sum N, 2]

    #: batches up to this many tracks are gathered / rotated on the host and sent in one copy (see ``batch``)
    HOST_ASSEMBLY_TRACKS = 2048

    def __len__(self):
        return len(self.sizes)

    def _h2d(self, t):
        # pinned + asynchronous: a pageable copy blocks the host until the stream has drained (see _lib.SceneIndex)
        from . import _lib
        return _lib.h2d_async(t, self.device)

    def batch(self, ids, augment=False):
        import math
        import random
        import torch
        ids = list(ids)
        n_tracks = int(self.sizes[ids].sum())
        if n_tracks <= self.HOST_ASSEMBLY_TRACKS:
            # small batches (the trainer's default batch_size 8: ~300 tracks, 50 KB): gather and rotate on the host, ONE pinned
            # asynchronous copy -- the device form below is a dozen small launches (index, stack, multiply ...) whose host cost
            # (0.27 ms) exceeds the whole batch's transfer; float32 arithmetic in the same order as the device form
            cols = np.concatenate([np.arange(self.starts[i], self.starts[i + 1]) for i in ids])
            xy = self._xy_host[cols]                                        # [M, T, 2]
            goals = self._goals_host[cols]
            if augment:
                theta = np.array([random.random() * 2.0 * math.pi for _ in ids], dtype=np.float64)
                per_track = np.repeat(theta, self.sizes[ids])
                ct, st = np.cos(per_track).astype(np.float32), np.sin(per_track).astype(np.float32)
                x, y = xy[..., 0], xy[..., 1]
                xy = np.stack([x * ct[:, None] - y * st[:, None], x * st[:, None] + y * ct[:, None]], axis=-1)
                gx, gy = goals[:, 0], goals[:, 1]
                goals = np.stack([gx * ct - gy * st, gx * st + gy * ct], axis=-1)
            T = xy.shape[1]
            flat = np.concatenate([np.ascontiguousarray(xy.transpose(1, 0, 2)).reshape(-1), goals.reshape(-1)])
            dev_flat = self._h2d(torch.from_numpy(flat))
            split = torch.from_numpy(np.concatenate([[0], np.cumsum(self.sizes[ids])]).astype(np.int64))
            return dev_flat[:T * n_tracks * 2].view(T, n_tracks, 2), dev_flat[T * n_tracks * 2:].view(n_tracks, 2), split
        cols = self._h2d(torch.from_numpy(np.concatenate([np.arange(self.starts[i], self.starts[i + 1]) for i in ids])))
        xy, goals = self.xy[:, cols], self.goals[cols]
        split = torch.tensor(np.concatenate([[0], np.cumsum(self.sizes[ids])]), dtype=torch.int64)
        if augment:
            theta = np.array([random.random() * 2.0 * math.pi for _ in ids], dtype=np.float64)
            per_track = np.repeat(theta, self.sizes[ids])             # (numpy: torch's CPU repeat_interleave wakes a thread pool)
            cs = self._h2d(torch.from_numpy(np.stack([np.cos(per_track), np.sin(per_track)]).astype(np.float32)))
            ct, st = cs[0], cs[1]
            x, y = xy[..., 0], xy[..., 1]
            xy = torch.stack([x * ct - y * st, x * st + y * ct], dim=-1)   # row vector times [[ct, st], [-st, ct]]
            gx, gy = goals[:, 0], goals[:, 1]
            goals = torch.stack([gx * ct - gy * st, gx * st + gy * ct], dim=-1)
        return xy, goals, split


# ---------------------------------------------------------------------------------------------------------------------
# Evaluator feed: test file -> batched predictions -> prediction file in the reference's layout
# ---------------------------------------------------------------------------------------------------------------------
def trajnet_format(row):
    """One ndjson line for a TrackRow / SceneRow.  Stands in for ``trajnetplusplustools.writers.trajnet`` (not vendored,
    SURVEY.md 8c): coordinates rounded to two decimals, prediction rows carry ``prediction_number`` and ``scene_id`` --
    the record layout of the data files ``read_ndjson_scenes`` parses."""
    import json
    if isinstance(row, SceneRow):
        return json.dumps({'scene': {'id': row.scene, 'p': row.pedestrian, 's': row.start, 'e': row.end, 'fps': row.fps,
                                     'tag': row.tag}})
    x, y = round(float(row.x), 2), round(float(row.y), 2)
    if row.prediction_number is None:
        return json.dumps({'track': {'f': row.frame, 'p': row.pedestrian, 'x': x, 'y': y}})
    return json.dumps({'track': {'f': row.frame, 'p': row.pedestrian, 'x': x, 'y': y,
                                 'prediction_number': row.prediction_number, 'scene_id': row.scene_id}})


def preprocess_test(scene, obs_len):
    """Drop tracks that only appear after the observation period and the rows after it (reference
    evaluator/write_utils.py:34-40; test files hold overlapping scenes)."""
    last_obs_frame = [r.frame for r in scene[0]][:obs_len][-1]
    return [[row for row in ped if row.frame <= last_obs_frame] for ped in scene if ped[0].frame <= last_obs_frame]


def write_predictions(pred_list, scenes, filename, obs_length=9, pred_length=12, mode='a'):
    """Append the predictions of ``scenes`` = [(scene_id, paths)] to ``filename`` exactly as the reference's
    ``write_predictions`` lays them out (evaluator/write_utils.py:42-81): per scene one SceneRow (primary id, first frame,
    first frame + (obs + pred - 1) frame steps, fps 2.5, tag 0), then per mode the primary's ``pred_length`` rows followed
    by every neighbour's rows, neighbours in the order of ``paths[1:]`` (which must be the paths the prediction was made
    from, i.e. after ``preprocess_test``); frames continue the primary's frame step."""
    seq_length = obs_length + pred_length

    def rows(frame0, frame_diff, ped, xy, m, scene_id):
        # the lines trajnet_format(TrackRow(...)) produces for one track, without a dict + json.dumps per row (the writer was 3/4
        # of the host time of predict_dataset).  json.dumps writes float.__repr__ for a float; xy [pred_length, 2] -> Python floats
        # of the float32 values, as the reference's .item() gives them.  Anything unusual (NaN / infinite coordinates, non-integer
        # frames or ids) takes the general path.
        arr = np.asarray(xy)
        pts = arr.tolist()
        if all(type(v) is int for v in (frame0, frame_diff, ped, m, scene_id)) and np.isfinite(arr).all():
            head = '{"track": {"f": %d, "p": ' + str(ped) + ', "x": %s, "y": %s, "prediction_number": ' + str(m) + \
                   ', "scene_id": ' + str(scene_id) + '}}\n'
            return ''.join([head % (frame0 + i * frame_diff, repr(round(x, 2)), repr(round(y, 2))) for i, (x, y) in enumerate(pts)])
        return ''.join(trajnet_format(TrackRow(frame0 + i * frame_diff, ped, x, y, m, scene_id)) + '\n' for i, (x, y) in enumerate(pts))

    import json
    with open(filename, mode) as f:
        for predictions, (scene_id, paths) in zip(pred_list, scenes):
            observed_path = paths[0]
            frame_diff = observed_path[1].frame - observed_path[0].frame
            first_frame = observed_path[obs_length - 1].frame + frame_diff
            ped_id = observed_path[0].pedestrian
            neigh_ids = [p[0].pedestrian for p in paths[1:]]
            f.write(trajnet_format(SceneRow(scene_id, ped_id, observed_path[0].frame,
                                            observed_path[0].frame + (seq_length - 1) * frame_diff, 2.5, 0)))
            f.write('\n')
            for m in range(len(predictions)):
                prediction, neigh_predictions = predictions[m]
                f.write(rows(first_frame, frame_diff, ped_id, prediction, m, scene_id))
                if len(neigh_predictions):
                    for n in range(neigh_predictions.shape[1]):
                        f.write(rows(first_frame, frame_diff, neigh_ids[n], neigh_predictions[:, n], m, scene_id))


def _predict_dataset_columns(ndjson_in, predictor, out_path, batch_scenes, obs_length, pred_length, modes, goals, in_flight, args,
                             limit):
    """``predict_dataset`` for predictors with an array-level entry (``predict_xy_launch`` / ``predict_xy_finish``): the test file
    is parsed into columns by the native reader, scenes become arrays without a Python object per row, the prediction file's
    lines are formatted natively, and the stages overlap -- while the GPU runs batch k the host assembles batch k + 1 and a small
    thread pool formats the batches before it (native calls release the GIL; the file is written in batch order).  ``in_flight`` = batches that stay queued
    on the GPU while the next one is assembled and launched.  Output: byte for byte what the general path writes.  Returns None when the file
    needs the general path."""
    import os
    cols = read_ndjson_columns(ndjson_in)
    if cols is None:
        return None
    scenes = scenes_from_columns(cols, obs_length, pred_length, limit)
    if scenes is None or any(sc.first_frame is None for sc in scenes):
        return None          # (a primary with fewer than obs_length frames: the general path raises the reference's error)
    d = os.path.dirname(os.path.abspath(out_path))
    if not os.path.isdir(d):
        os.makedirs(d)
    # formatting (native, releases the GIL) on a small pool, the file written in batch order by the main thread as results land
    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(max_workers=3)
    futures = []
    pending = []
    with open(out_path, 'wb') as f:
        def drain(block):
            while futures and (block or futures[0].done()):
                f.write(futures.pop(0).result())
        try:
            for lo in range(0, len(scenes), batch_scenes):
                chunk = scenes[lo:lo + batch_scenes]
                scene_goals = [np.array([goals[int(p)] for p in sc.peds], dtype=np.float64) if goals is not None
                               else np.zeros((len(sc.peds), 2)) for sc in chunk]
                handle = predictor.predict_xy_launch([sc.xy for sc in chunk], scene_goals, n_predict=pred_length, modes=modes,
                                                     obs_length=obs_length, args=args)
                pending.append((handle, chunk))
                while len(pending) > max(1, int(in_flight)):           # batch k is queued: read back batch k - in_flight
                    h, ch = pending.pop(0)
                    pred, split = predictor.predict_xy_finish(h, pred_length)
                    futures.append(pool.submit(format_predictions, pred, split, ch))
                drain(False)
                while len(futures) > 8:                                # bound the formatted bytes held in memory
                    f.write(futures.pop(0).result())
            for h, ch in pending:
                pred, split = predictor.predict_xy_finish(h, pred_length)
                futures.append(pool.submit(format_predictions, pred, split, ch))
            drain(True)
        finally:
            pool.shutdown(wait=True)
    return len(scenes)


def predict_dataset(ndjson_in, predictor, out_path, batch_scenes=64, obs_length=9, pred_length=12, modes=1, goals=None,
                    in_flight=1, args=None, limit=None, predict_kwargs=None):
    """The evaluator's prediction loop for one test file (reference lstm/trajnet_evaluator.py:29-65 ``get_predictions`` +
    evaluator/write_utils.py): read the scenes, ``preprocess_test`` each, predict them ``batch_scenes`` at a time through
    ``predictor.predict_batch`` (ONE ``LSTM.forward`` per batch instead of one call per scene on 12 joblib workers;
    ``in_flight`` > 1 keeps that many batches on the GPU at once through ``predict_batches``: +20 % on the GPU side, but the
    loop is bound by reading and writing the files -- 350-410 scenes/s end to end at 40 agents per scene against 6 300 scenes/s of
    prediction alone, tools/diag/predict_dataset_throughput.py -- so the default is one) and write the prediction file in
    the reference's layout.  ``goals``: {pedestrian id: (x, y)} (the reference's goal pickle, write_utils.py:21-26) or None
    = zeros.  ``predictor`` is anything with ``predict_batch(scenes, n_predict=, modes=, obs_length=, args=)`` -- the
    ``LSTMPredictor`` / ``SGANPredictor`` mirrors -- or a per-scene callable ``predictor(paths, scene_goal, ...)`` (the
    classical predictors).  A classical MODULE (``classical.socialforce`` / ``orca`` / ``kalman`` / ``constant_velocity``) is
    predicted through its ``predict_scenes`` (all scenes of a batch in one launch), a classical ``predict`` FUNCTION one scene per
    call as the reference's classical evaluator does (classical/trajnet_evaluator.py:14-27: no goal argument);
    ``predict_kwargs`` are handed on (``sf_params=``, ``orca_params=``, ...).  Returns the number of scenes written."""
    import os
    kw = dict(predict_kwargs or {})
    if hasattr(predictor, 'predict_xy_launch') and not kw:
        n = _predict_dataset_columns(ndjson_in, predictor, out_path, batch_scenes, obs_length, pred_length, modes, goals, in_flight,
                                     args, limit)
        if n is not None:
            return n
    scenes = read_ndjson_scenes(ndjson_in, limit=limit)
    scenes = [(sid, preprocess_test(paths, obs_length)) for sid, paths in scenes]
    scene_goals = [np.array([goals[p[0].pedestrian] for p in paths], dtype=np.float64) if goals is not None
                   else np.zeros((len(paths), 2)) for _, paths in scenes]
    d = os.path.dirname(os.path.abspath(out_path))
    if not os.path.isdir(d):
        os.makedirs(d)
    open(out_path, 'w').close()
    chunks = [list(range(lo, min(lo + batch_scenes, len(scenes)))) for lo in range(0, len(scenes), batch_scenes)]
    batches = [[(scenes[i][1], scene_goals[i]) for i in ids] for ids in chunks]
    if hasattr(predictor, 'predict_scenes'):
        results = [predictor.predict_scenes(b, n_predict=pred_length, obs_length=obs_length, **kw) for b in batches]
    elif hasattr(predictor, 'predict_batches') and in_flight > 1:
        results = predictor.predict_batches(batches, n_predict=pred_length, modes=modes, obs_length=obs_length, args=args,
                                            in_flight=in_flight)
    elif hasattr(predictor, 'predict_batch'):
        results = [predictor.predict_batch(b, n_predict=pred_length, modes=modes, obs_length=obs_length, args=args)
                   for b in batches]
    elif hasattr(predictor, 'model'):          # a Predictor object without a batched form: the evaluator's per-scene call
        results = [[predictor(paths, goal, n_predict=pred_length, obs_length=obs_length, modes=modes, args=args, **kw)
                    for paths, goal in b] for b in batches]
    else:                                      # a classical predict function
        results = [[predictor(paths, n_predict=pred_length, obs_length=obs_length, **kw) for paths, goal in b] for b in batches]
    for ids, preds in zip(chunks, results):
        write_predictions(preds, [scenes[i] for i in ids], out_path, obs_length, pred_length, mode='a')
    return len(scenes)
