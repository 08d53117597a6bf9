"""Evaluator feed as code (VERDICT r4 "missing 4"): ``data.predict_dataset`` = the reference's
``get_predictions`` loop (lstm/trajnet_evaluator.py:29-65) with batched prediction, ``data.write_predictions`` /
``data.preprocess_test`` = evaluator/write_utils.py:34-81.  Fixture: the reference's own ``write_predictions`` on 8 real scenes
with given two-mode predictions (oracle/gen_golden_r5.py:writer; ``trajnetplusplustools``' row classes and line writer are
stubbed with the package's published layout -- the reference's row order, frame arithmetic and ids are what is pinned)."""
import json
import os

import numpy as np
import pytest

from trajnetplusplusbaselines_amd import data as trajdata

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
INP = os.path.join(GOLDEN, 'writer_case.ndjson')
EXPECTED = os.path.join(GOLDEN, 'writer_case_expected.ndjson')


def stored_predictions():
    z = np.load(os.path.join(GOLDEN, 'writer_case_preds.npz'))
    n = len([k for k in z.files if k.endswith('_ids')])
    preds = []
    for k in range(n):
        d = {}
        for m in range(2):
            neigh = z['s%d_m%d_neigh' % (k, m)]
            d[m] = [z['s%d_m%d_prim' % (k, m)], neigh if neigh.size else []]
        preds.append(d)
    return preds, [z['s%d_ids' % k] for k in range(n)]


def test_preprocess_test_drops_late_tracks_and_future_rows():
    scenes = trajdata.read_ndjson_scenes(INP)
    for sid, paths in scenes:
        last = paths[0][8].frame
        got = trajdata.preprocess_test(paths, 9)
        assert len(got[0]) == 9 and all(r.frame <= last for p in got for r in p)
        assert [p[0].pedestrian for p in got] == [p[0].pedestrian for p in paths if p[0].frame <= last]


def test_write_predictions_equals_the_reference_writer(tmp_path):
    preds, ids = stored_predictions()
    scenes = [(sid, trajdata.preprocess_test(paths, 9)) for sid, paths in trajdata.read_ndjson_scenes(INP)]
    scenes[5] = (scenes[5][0], scenes[5][1][:1])                      # the fixture's "primary alone" scene
    for (sid, paths), want in zip(scenes, ids):
        assert [p[0].pedestrian for p in paths] == list(want)
    out = str(tmp_path / 'pred.ndjson')
    trajdata.write_predictions(preds, scenes, out, 9, 12)
    assert open(out).read() == open(EXPECTED).read()


class _StoredPredictor(object):
    """predict_batch that replays the fixture's predictions (host only): exercises the loop, the chunking and the writer"""

    def __init__(self):
        self.preds, _ = stored_predictions()
        self.calls, self.k = [], 0

    def predict_batch(self, scenes, n_predict=12, modes=1, obs_length=9, start_length=0, args=None):
        self.calls.append(len(scenes))
        out = []
        for paths, goal in scenes:
            assert goal.shape == (len(paths), 2)
            p = self.preds[self.k]
            n = len(paths) - 1
            neigh = lambda m: (p[m][1][:, :n] if len(p[m][1]) else np.zeros((12, n, 2), dtype=np.float32)) if n else []
            out.append({m: [p[m][0], neigh(m)] for m in range(modes)})
            self.k += 1
        return out


def parse_predictions(path):
    """prediction file -> {scene_id: {mode: {pedestrian: [[frame, x, y], ...]}}}, scene rows in file order"""
    scenes, tracks = [], {}
    for line in open(path):
        rec = json.loads(line)
        if 'scene' in rec:
            scenes.append(rec['scene'])
        else:
            t = rec['track']
            tracks.setdefault(t['scene_id'], {}).setdefault(t['prediction_number'], {}).setdefault(t['p'], []).append(
                [t['f'], t['x'], t['y']])
    return scenes, tracks


def test_predict_dataset_loop_chunks_and_parses_back(tmp_path):
    pred = _StoredPredictor()
    # scene 5 of the fixture was cut to its primary for the writer test; here every scene keeps its neighbours
    out = str(tmp_path / 'sub' / 'pred.ndjson')
    n = trajdata.predict_dataset(INP, pred, out, batch_scenes=3, modes=2)
    assert n == 8 and pred.calls == [3, 3, 2]
    scenes, tracks = parse_predictions(out)
    src = trajdata.read_ndjson_scenes(INP)
    assert [s['id'] for s in scenes] == [sid for sid, _ in src]
    for k, (sid, paths) in enumerate(src):
        paths = trajdata.preprocess_test(paths, 9)
        fd = paths[0][1].frame - paths[0][0].frame
        assert scenes[k] == {'id': sid, 'p': paths[0][0].pedestrian, 's': paths[0][0].frame,
                             'e': paths[0][0].frame + 20 * fd, 'fps': 2.5, 'tag': 0}
        for m in range(2):
            prim = np.array(tracks[sid][m][paths[0][0].pedestrian])
            assert list(prim[:, 0]) == [paths[0][8].frame + fd * (i + 1) for i in range(12)]
            want = pred.preds[k][m][0].astype(np.float64)
            assert np.abs(prim[:, 1:] - want).max() <= 0.005 + 1e-6           # two decimals
            assert set(tracks[sid][m]) == {p[0].pedestrian for p in paths}


@pytest.mark.gpu
def test_predict_dataset_with_the_lstm_predictor_on_device(tmp_path):
    from trajnetplusplusbaselines_amd.lstm import LSTMPredictor
    p = LSTMPredictor.load(os.path.join(GOLDEN, 'ref_predictor.pkl'))
    p.model.to('cuda')
    out1, out2 = str(tmp_path / 'a.ndjson'), str(tmp_path / 'b.ndjson')
    assert trajdata.predict_dataset(INP, p, out1, batch_scenes=3) == 8
    assert trajdata.predict_dataset(INP, p, out2, batch_scenes=2, in_flight=2) == 8
    assert open(out1).read() == open(out2).read()          # batching / batches in flight do not change a digit
    scenes, tracks = parse_predictions(out1)
    for sid, paths in trajdata.read_ndjson_scenes(INP):
        paths = trajdata.preprocess_test(paths, 9)
        res = p(paths, np.zeros((len(paths), 2)), n_predict=12)          # the evaluator's per-scene call
        prim = np.array(tracks[sid][0][paths[0][0].pedestrian])[:, 1:]
        assert np.array_equal(prim, np.round(res[0][0].astype(np.float64), 2))
        for n, path in enumerate(paths[1:]):
            got = np.array(tracks[sid][0][path[0].pedestrian])[:, 1:]
            want = res[0][1][:, n].astype(np.float64)
            assert np.array_equal(np.isnan(got), np.isnan(want))
            assert np.array_equal(got[~np.isnan(got)], np.round(want[~np.isnan(want)], 2))


def test_paths_to_xy_equals_the_row_loop_definition():
    """``paths_to_xy`` places all rows of a scene with one numpy assignment; its definition is the reference tool's row loop
    (frames of the primary, NaN for absent, the later of two rows of a track in one frame stays, rows outside the primary's frames
    are dropped)."""
    def rows_loop(paths):
        frames = sorted(set(r.frame for r in paths[0]))
        index = {f: i for i, f in enumerate(frames)}
        xy = np.full((len(frames), len(paths), 2), np.nan)
        for p, path in enumerate(paths):
            for r in path:
                if r.frame in index:
                    xy[index[r.frame], p] = [r.x, r.y]
        return xy

    for sid, paths in trajdata.read_ndjson_scenes(INP):
        assert np.array_equal(trajdata.paths_to_xy(paths), rows_loop(paths), equal_nan=True)
        cut = trajdata.preprocess_test(paths, 9)
        assert np.array_equal(trajdata.paths_to_xy(cut), rows_loop(cut), equal_nan=True)
    T = trajdata.TrackRow
    odd = [[T(10 * t, 1, float(t), 0.0) for t in range(5)], [T(10 * t, 2, 1.0 * k, 2.0) for k, t in enumerate((0, 1, 1, 7, 3))], []]
    assert np.array_equal(trajdata.paths_to_xy(odd), rows_loop(odd), equal_nan=True)

    class Row(object):
        def __init__(self, f, p, x, y):
            self.frame, self.pedestrian, self.x, self.y = f, p, x, y
    objs = [[Row(10 * t, 1, float(t), 0.5) for t in range(4)], [Row(10 * t, 2, 2.0, float(t)) for t in (1, 2, 9)]]
    assert np.array_equal(trajdata.paths_to_xy(objs), rows_loop(objs), equal_nan=True)
