"""Round-5 fixtures, generated from the imported reference in the BUILD CONTAINER (never on the GPU box):

  python -m oracle.gen_golden_r5 [classical] [writer] ...

classical -> tests/golden/classical_ref.npz
    The reference's OWN ``classical/socialforce.py``, ``classical/orca.py``, ``classical/kalman.py`` and
    ``classical/constant_velocity.py`` ``predict()`` functions, unmodified, run on real DATA_BLOCK scenes on top of the
    DRIVING STUBS of ``oracle/classical_stubs.py`` (simulator arithmetic = ``oracle/classical_numpy.py``).  This pins what the
    wrappers decide -- agent selection, stride-3 velocity, extrapolated goal, max_speed = 1.3 x speed, the 96-step /
    every-8th-state cadence, ORCA's 97 steps sampled at count 8..96 and its preferred-velocity rule, Kalman's matrices and
    13-sample / mean-of-5 rule, the output ordering and ``predict_all`` -- to the reference's lines
    (socialforce.py:15-111, orca.py:14-133, kalman.py:6-73).  The third-party arithmetic behind the stubs stays
    unpinned (our restatement).
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import ref_import  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')


def _rows(paths):
    rows = np.array([[r.frame, r.pedestrian, r.x, r.y] for p in paths for r in p], dtype=np.float64)
    return rows, np.array([len(p) for p in paths], dtype=np.int64)


def _edit_paths(paths, kind, rng):
    """Synthetic edits of a real scene that real data rarely shows (returns new paths; TrackRow is immutable)."""
    from trajnetplusplusbaselines_amd.data import TrackRow
    paths = [list(p) for p in paths]
    start_frame = paths[0][8].frame
    if kind == 'stationary_primary':         # the primary stands still during the observation
        p0 = paths[0][8]
        paths[0] = [TrackRow(r.frame, r.pedestrian, p0.x, p0.y) if r.frame <= start_frame else r for r in paths[0]]
    elif kind == 'late_entries':             # neighbours with 1, 2 and 3 past frames (stride 0, 1, 2)
        keep = 1
        for k in range(1, len(paths)):
            past = [r for r in paths[k] if r.frame <= start_frame]
            if len(past) >= 4 and past[-1].frame == start_frame:
                paths[k] = past[-keep:] + [r for r in paths[k] if r.frame > start_frame]
                keep = keep % 3 + 1
    elif kind == 'primary_only':
        paths = paths[:1]
    return paths


def classical():
    from oracle import classical_stubs as stubs
    from trajnetplusplusbaselines_amd import data
    from oracle import oracle as _oracle
    _oracle.build()
    ref_sf, ref_orca, ref_kalman, ref_cv = stubs.install()
    root = os.path.join(ref_import.REFERENCE_ROOT, 'DATA_BLOCK', 'trajdata', 'train')
    files = sorted(f for f in os.listdir(root) if f.endswith('.ndjson'))
    rng = np.random.RandomState(5)
    cases = []
    for fname in files:
        scenes = data.read_ndjson_scenes(os.path.join(root, fname))
        # scenes of moderate size (the numpy ORCA is O(97 n^2) Python): spread over the file
        ok = [s for s in scenes if 2 <= len(s[1]) <= 14 and len(s[1][0]) >= 21]
        if len(ok) < 8:                       # dense files (students): take their smallest scenes
            ok = sorted((s for s in scenes if len(s[1][0]) >= 21), key=lambda s: len(s[1]))[:16]
        pick = np.linspace(0, len(ok) - 1, 8).astype(int)
        for j, k in enumerate(pick):
            cases.append((fname, ok[k][0], ok[k][1], 'real'))
    # synthetic edits on a few of them
    for kind, idx in [('stationary_primary', 3), ('stationary_primary', 20), ('late_entries', 5), ('late_entries', 12),
                      ('late_entries', 33), ('primary_only', 7)]:
        fname, sid, paths, _ = cases[idx]
        cases.append((fname, sid, _edit_paths(paths, kind, rng), kind))
    out = {'n_cases': np.int64(len(cases)), 'files': np.asarray([c[0] for c in cases]),
           'scene_ids': np.asarray([c[1] for c in cases], dtype=np.int64), 'kinds': np.asarray([c[3] for c in cases])}
    t0 = time.time()
    stats = {'short': 0, 'absent': 0, 'stride0': 0}
    for i, (fname, sid, paths, kind) in enumerate(cases):
        pre = 'c%d_' % i
        out[pre + 'rows'], out[pre + 'lens'] = _rows(paths)
        start_frame = paths[0][8].frame
        for p in paths:
            past = [r for r in p if r.frame <= start_frame]
            if not past or past[-1].frame != start_frame:
                stats['absent'] += 1
            elif len(past) < 4:
                stats['short'] += 1
                stats['stride0'] += len(past) == 1
        predict_all = (i % 3 != 2)
        out[pre + 'predict_all'] = np.int64(predict_all)
        # option variants: a handful of cases use other destination rules / parameters
        sf_kw, orca_kw = {}, {}
        if i % 7 == 1:
            sf_kw, orca_kw = {'dest_type': 'pred_end'}, {'dest_type': 'pred_end'}
        elif i % 7 == 3:
            sf_kw = {'dest_type': 'vel'}
            dd = {p[0].pedestrian: [float(p[-1].x) + 0.25, float(p[-1].y) - 0.5] for p in paths}
            orca_kw = {'dest_type': 'true', 'dest_dict': dd}
            out[pre + 'dest_dict'] = np.array([[k, v[0], v[1]] for k, v in dd.items()], dtype=np.float64)
        elif i % 7 == 5:
            sf_kw, orca_kw = {'sf_params': [0.4, 1.6, 0.25]}, {'orca_params': [2.0, 1.0, 0.3]}
        out[pre + 'sf_kw'] = np.asarray(repr({k: v for k, v in sf_kw.items()}))
        out[pre + 'orca_kw'] = np.asarray(repr({k: v for k, v in orca_kw.items() if k != 'dest_dict'}))
        if 'pred_end' in (sf_kw.get('dest_type'), orca_kw.get('dest_type')) and \
                any(not [r for r in p if r.frame > start_frame] for p in paths
                    if [r for r in p if r.frame == start_frame]):
            sf_kw, orca_kw = {}, {}          # a present track without a future: the reference raises IndexError
            out[pre + 'sf_kw'], out[pre + 'orca_kw'] = np.asarray('{}'), np.asarray('{}')

        for backend, tag in (('numpy', ''), ('core', 'core_')):
            stubs.set_backend(backend)
            # ---- social force ----
            stubs.reset_record()
            with np.errstate(all='ignore'):
                res = ref_sf.predict(paths, predict_all=predict_all, **sf_kw)[0]
            rec = stubs.RECORD['sf']
            assert len(rec) == 1 and rec[0]['sim'].n_steps == 96 and rec[0]['delta_t'] == 1. / 20
            out[pre + 'sf_initial_state'] = rec[0]['initial_state']
            out[pre + 'sf_params'] = np.array([rec[0]['tau'], rec[0]['v0'], rec[0]['sigma']])
            out[pre + 'sf_' + tag + 'primary'] = np.asarray(res[0], dtype=np.float64)
            out[pre + 'sf_' + tag + 'neigh'] = np.asarray(res[1], dtype=np.float64)

            # ---- ORCA ----
            stubs.reset_record()
            res = ref_orca.predict(paths, predict_all=predict_all, **orca_kw)[0]
            rec = stubs.RECORD['orca']
            assert len(rec) == 1 and rec[0]['sim'].n_steps == 97
            out[pre + 'orca_sim_args'] = np.array(rec[0]['args'], dtype=np.float64)
            out[pre + 'orca_agents'] = np.array([[a[0][0], a[0][1], a[1], a[2][0], a[2][1]] for a in rec[0]['agents']],
                                                dtype=np.float64).reshape(-1, 5)
            out[pre + 'orca_' + tag + 'pref_calls'] = np.array(rec[0]['pref_calls'][:3 * len(rec[0]['agents'])], dtype=np.float64)
            out[pre + 'orca_' + tag + 'primary'] = np.asarray(res[0], dtype=np.float64)
            out[pre + 'orca_' + tag + 'neigh'] = np.asarray(res[1], dtype=np.float64)

            # ---- Kalman (numpy's global RNG, seeded here; the reference leaves it unseeded) ----
            stubs.reset_record()
            np.random.seed(1000 + i)
            res = ref_kalman.predict(paths, predict_all=predict_all)[0]
            rec = stubs.RECORD['kalman']
            assert all(len(r['sample_calls']) == 5 and all(n == 13 for n, _ in r['sample_calls']) for r in rec)
            out[pre + 'kalman_seed'] = np.int64(1000 + i)
            out[pre + 'kalman_noise'] = np.array([[z for _, z in r['sample_calls']] for r in rec]).reshape(-1, 5, 13, 6)
            out[pre + 'kalman_obs_lens'] = np.array([len(r['em_obs']) for r in rec], dtype=np.int64)
            out[pre + 'kalman_' + tag + 'primary'] = np.asarray(res[0], dtype=np.float64)
            out[pre + 'kalman_' + tag + 'neigh'] = np.asarray(res[1], dtype=np.float64)
        stubs.set_backend('numpy')

        # ---- constant velocity (no third-party arithmetic at all) ----
        res = ref_cv.predict(paths)[0]
        out[pre + 'cv_primary'] = np.asarray(res[0], dtype=np.float64)
        out[pre + 'cv_neigh'] = np.asarray(res[1], dtype=np.float64)
        if i % 10 == 0:
            print('case %d (%s, %s, %d tracks)  [%.0f s]' % (i, fname, kind, len(paths), time.time() - t0), flush=True)
    print('tracks over all cases: absent at the last observed frame %d, fewer than 4 past frames %d (single frame %d)'
          % (stats['absent'], stats['short'], stats['stride0']))
    np.savez_compressed(os.path.join(OUT, 'classical_ref.npz'), **out)
    print('classical_ref.npz  %d cases' % len(cases))


def pool_grad():
    """Autograd through the STAND-ALONE reference ``GridBasedPooling.forward(hidden_state, obs1, obs2)``
    (lstm/gridbased_pooling.py:94-110), called outside LSTM.forward: gradients of sum(out * R) (R a fixed random matrix)
    with respect to hidden_state, the pool's parameters and -- directional -- obs1 / obs2.  Crowds with absent (NaN) agents,
    several neighbours per cell and out-of-range neighbours (cell (0,0) clobbers).  single-threaded.
    tests/golden/pool_grad.npz"""
    import torch
    ref = ref_import.import_reference()
    torch.set_num_threads(1)
    out = {}
    rng = np.random.RandomState(17)
    cases = [('social', dict(type_='social', hidden_dim=32, cell_side=0.6, n=6, out_dim=16, embedding_arch='two_layer',
                             layer_dims=[64], latent_dim=8), 3, 7),
             ('social_one', dict(type_='social', hidden_dim=32, cell_side=0.5, n=4, out_dim=24, embedding_arch='one_layer',
                                 latent_dim=16), 2, 12),
             ('directional', dict(type_='directional', hidden_dim=32, cell_side=0.6, n=6, out_dim=16,
                                  embedding_arch='one_layer'), 3, 7),
             ('directional_const', dict(type_='directional', hidden_dim=32, cell_side=0.5, n=4, out_dim=8, constant=1,
                                        embedding_arch='two_layer', layer_dims=[32]), 2, 9),
             ('occupancy', dict(type_='occupancy', hidden_dim=32, cell_side=0.6, n=6, out_dim=16, embedding_arch='one_layer'), 2, 5)]
    for name, kw, B, N in cases:
        torch.manual_seed(11)
        pool = ref.GridBasedPooling(**kw)
        H = kw['hidden_dim']
        obs2 = rng.randn(B, N, 2) * 0.7
        obs1 = obs2 - rng.randn(B, N, 2) * 0.15
        obs2[0, 2] = np.nan; obs1[0, 2] = np.nan          # an absent agent
        obs2[B - 1, 1] = [40.0, 40.0]                      # out of range for everybody
        obs2[0, 4] = obs2[0, 3] + 0.01                     # two neighbours in one cell for most egos
        hid = rng.randn(B, N, H)
        hid[0, 2] = np.nan
        o1 = torch.tensor(obs1, dtype=torch.float32, requires_grad=name.startswith('directional'))
        o2 = torch.tensor(obs2, dtype=torch.float32, requires_grad=name.startswith('directional'))
        h = torch.tensor(hid, dtype=torch.float32, requires_grad=True)
        R = torch.tensor(rng.randn(B * N, kw['out_dim']), dtype=torch.float32)
        # (the reference writes -500 into absent rows of obs IN PLACE, :249: a leaf that requires grad cannot be passed)
        y = pool(h, o1 * 1.0 if o1.requires_grad else o1, o2 * 1.0 if o2.requires_grad else o2)
        (y * R).sum().backward()
        pre = name + '_'
        out[pre + 'kw'] = np.asarray(repr(kw))
        out[pre + 'obs1'], out[pre + 'obs2'], out[pre + 'hidden'], out[pre + 'R'] = obs1.astype(np.float32), obs2.astype(np.float32), hid.astype(np.float32), R.numpy()
        out[pre + 'out'] = y.detach().numpy()
        for k, v in pool.state_dict().items():
            out[pre + 'w_' + k] = v.numpy().copy()
        for k, p_ in pool.named_parameters():
            out[pre + 'g_' + k] = p_.grad.numpy().copy() if p_.grad is not None else np.zeros(0, dtype=np.float32)
        if h.grad is not None:
            out[pre + 'g_hidden'] = torch.nan_to_num(h.grad).numpy().copy()
        if o2.grad is not None:
            out[pre + 'g_obs1'], out[pre + 'g_obs2'] = torch.nan_to_num(o1.grad).numpy().copy(), torch.nan_to_num(o2.grad).numpy().copy()
        print(name, 'out', tuple(y.shape), 'grads:', sorted(k for k in out if k.startswith(pre + 'g_')))
    np.savez_compressed(os.path.join(OUT, 'pool_grad.npz'), **out)
    print('pool_grad.npz')


def ref_pickle():
    """A checkpoint written by the REFERENCE's ``LSTMPredictor.save`` (lstm/lstm.py:270-277: ``torch.save(self)`` + ``.state``)
    for a small Social-LSTM, and the reference predictor's own ``__call__`` on three real scenes.
    tests/golden/ref_predictor.pkl (+ .state), tests/golden/ref_predictor_cases.npz"""
    import types
    import torch
    ref = ref_import.import_reference()
    import trajnetbaselines.lstm.lstm as ref_lstm
    from oracle import classical_stubs
    from trajnetplusplusbaselines_amd import data
    sys.modules['trajnetplusplustools'].Reader = classical_stubs._Reader
    ref_lstm.trajnetplusplustools = sys.modules['trajnetplusplustools']
    torch.set_num_threads(1)
    torch.manual_seed(5)
    pool = ref.GridBasedPooling(type_='social', hidden_dim=64, cell_side=0.6, n=6, out_dim=32, embedding_arch='two_layer',
                                layer_dims=[64], latent_dim=8)
    model = ref.LSTM(embedding_dim=32, hidden_dim=64, pool=pool)
    predictor = ref_lstm.LSTMPredictor(model)
    state = {'epoch': 3, 'state_dict': model.state_dict(), 'optimizer': None, 'scheduler': None}
    path = os.path.join(OUT, 'ref_predictor.pkl')
    predictor.save(state, path)
    root = os.path.join(ref_import.REFERENCE_ROOT, 'DATA_BLOCK', 'trajdata', 'train')
    scenes = data.read_ndjson_scenes(os.path.join(root, 'biwi_hotel.ndjson'))
    out = {}
    out['normalize'] = np.array([0, 1, 0], dtype=np.int64)
    for k, si in enumerate((3, 40, 111)):
        args = types.SimpleNamespace(normalize_scene=bool(out['normalize'][k]))
        paths = scenes[si][1]
        xy = data.paths_to_xy(paths)
        res = predictor(paths, np.zeros((xy.shape[1], 2)), n_predict=12, obs_length=9, args=args)
        out['s%d_rows' % k], out['s%d_lens' % k] = _rows(paths)
        out['s%d_primary' % k], out['s%d_neigh' % k] = res[0][0], res[0][1]
    np.savez_compressed(os.path.join(OUT, 'ref_predictor_cases.npz'), **out)
    print('ref_predictor.pkl %d bytes' % os.path.getsize(path))


def writer():
    """The reference's ``write_predictions`` (evaluator/write_utils.py:42-81) and ``preprocess_test`` (:34-40), unmodified, on a
    small real test file with given predictions (two modes; one scene with the primary alone).  ``trajnetplusplustools`` is
    absent: TrackRow / SceneRow / writers.trajnet are stubbed with the package's published record layout (coordinates rounded
    to two decimals) -- what IS pinned is the reference's row order, frame arithmetic, ids and SceneRow fields.
    tests/golden/writer_case.ndjson (input: 8 scenes of DATA_BLOCK/trajdata/train/biwi_hotel.ndjson re-serialised),
    writer_case_expected.ndjson, writer_case_preds.npz"""
    import importlib.util
    import json
    import types
    from collections import namedtuple
    from trajnetplusplusbaselines_amd import data
    tools = types.ModuleType('trajnetplusplustools')
    tools.TrackRow = namedtuple('Row', ['frame', 'pedestrian', 'x', 'y', 'prediction_number', 'scene_id'])
    tools.TrackRow.__new__.__defaults__ = (None, None, None, None, None, None)
    tools.SceneRow = namedtuple('Row', ['scene', 'pedestrian', 'start', 'end', 'fps', 'tag'])
    tools.SceneRow.__new__.__defaults__ = (None, None, None, None, None, None)

    def trajnet(row):
        if isinstance(row, tools.TrackRow):
            x, y = round(row.x, 2), round(row.y, 2)
            if row.prediction_number is None:
                return json.dumps({'track': {'f': row.frame, 'p': row.pedestrian, 'x': x, 'y': y}})
            return json.dumps({'track': {'f': row.frame, 'p': row.pedestrian, 'x': x, 'y': y,
                                         'prediction_number': row.prediction_number, 'scene_id': row.scene_id}})
        return json.dumps({'scene': {'id': row.scene, 'p': row.pedestrian, 's': row.start, 'e': row.end, 'fps': row.fps, 'tag': row.tag}})
    tools.writers = types.ModuleType('trajnetplusplustools.writers')
    tools.writers.trajnet = trajnet
    saved = sys.modules.get('trajnetplusplustools')
    sys.modules['trajnetplusplustools'] = tools
    spec = importlib.util.spec_from_file_location('_ref_write_utils', os.path.join(ref_import.REFERENCE_ROOT, 'evaluator', 'write_utils.py'))
    wu = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(wu)
    if saved is not None:
        sys.modules['trajnetplusplustools'] = saved
    src = os.path.join(ref_import.REFERENCE_ROOT, 'DATA_BLOCK', 'trajdata', 'train', 'biwi_hotel.ndjson')
    all_scenes = data.read_ndjson_scenes(src)
    pick = [all_scenes[i] for i in (0, 7, 19, 33, 58, 90, 120, 150)]
    # the input file of the test: the picked scenes' rows re-serialised (scene rows first, then every track row once)
    inp = os.path.join(OUT, 'writer_case.ndjson')
    seen = set()
    with open(inp, 'w') as f:
        for sid, paths in pick:
            f.write(json.dumps({'scene': {'id': sid, 'p': paths[0][0].pedestrian, 's': paths[0][0].frame, 'e': paths[0][-1].frame,
                                          'fps': 2.5, 'tag': [0, []]}}) + '\n')
        for sid, paths in pick:
            for p in paths:
                for r in p:
                    if (r.frame, r.pedestrian) not in seen:
                        seen.add((r.frame, r.pedestrian))
                        f.write(json.dumps({'track': {'f': r.frame, 'p': r.pedestrian, 'x': r.x, 'y': r.y}}) + '\n')
    scenes = data.read_ndjson_scenes(inp)
    rng = np.random.RandomState(2)
    pred_list, ref_scenes, store = [], [], {}
    for k, (sid, paths) in enumerate(scenes):
        paths = [[tools.TrackRow(r.frame, r.pedestrian, r.x, r.y) for r in p] for p in paths]
        paths = wu.preprocess_test(paths, 9)
        if k == 5:
            paths = paths[:1]
        preds = {}
        for m in range(2):
            prim = np.array([paths[0][-1].x, paths[0][-1].y]) + np.cumsum(rng.randn(12, 2) * 0.3, axis=0)
            neigh = rng.randn(12, len(paths) - 1, 2) * 3 if len(paths) > 1 else []
            preds[m] = [prim.astype(np.float32), np.asarray(neigh, dtype=np.float32) if len(paths) > 1 else []]
            store['s%d_m%d_prim' % (k, m)] = preds[m][0]
            store['s%d_m%d_neigh' % (k, m)] = np.asarray(preds[m][1], dtype=np.float32)
        store['s%d_ids' % k] = np.array([p[0].pedestrian for p in paths], dtype=np.int64)
        pred_list.append(preds)
        ref_scenes.append(('biwi_hotel', sid, paths))
    exp = os.path.join(OUT, 'writer_case_expected.ndjson')
    if os.path.exists(exp):
        os.remove(exp)
    args = types.SimpleNamespace(obs_length=9, pred_length=12, path=OUT + '/')
    # write_predictions opens args.path + '{model_name}/{dataset_name}'
    os.makedirs(os.path.join(OUT, '_w'), exist_ok=True)
    wu.write_predictions(pred_list, ref_scenes, '_w', 'out.ndjson', args)
    os.replace(os.path.join(OUT, '_w', 'out.ndjson'), exp)
    os.rmdir(os.path.join(OUT, '_w'))
    np.savez_compressed(os.path.join(OUT, 'writer_case_preds.npz'), **store)
    print('writer_case: %d scenes, %d lines expected' % (len(scenes), sum(1 for _ in open(exp))))


if __name__ == '__main__':
    what = sys.argv[1:] or ['classical']
    for w in what:
        globals()[w]()
