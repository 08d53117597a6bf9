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


if __name__ == '__main__':
    what = sys.argv[1:] or ['classical']
    for w in what:
        globals()[w]()
