"""Pins the CPU oracle (oracle/trajnet_oracle.c) to the real reference through the
fixtures that oracle/gen_golden.py produced by running the Python reference, and to
the reference's own (adapted) known-answer vectors -- SURVEY.md section 4 / 8(c)."""
import numpy as np
import pytest

from oracle import oracle
from tests import helpers

GRID_CASES = helpers.load_grid_cases()


@pytest.mark.parametrize('rec', GRID_CASES, ids=[r['name'] for r in GRID_CASES])
def test_grid_matches_reference(rec):
    vals = helpers.social_enc(rec) if rec['type'] == 'social' else None
    g = oracle.grid(rec['type'], rec['obs1'], rec['obs2'], vals, n=rec['n'], cell_side=rec['cell_side'],
                    pool_size=rec['pool_size'], blur_size=rec['blur_size'], constant=rec['constant'],
                    front=bool(rec['front']), C=rec['grid'].shape[1])
    ref = rec['grid']
    if ref.shape[0] == 1 and g.shape[0] != 1:   # single-track shortcut returns one row
        g = g[:1]
    if rec['type'] == 'social' or rec['blur_size'] != 1:
        np.testing.assert_allclose(g, ref, rtol=0, atol=2e-6)
    else:
        # occupancy / directional values are copies or single fp32 subtractions: bit-exact
        assert np.array_equal(g, ref), np.abs(g - ref).max()


@pytest.mark.parametrize('rec', [r for r in GRID_CASES if 'tag_grid' in r],
                         ids=[r['name'] for r in GRID_CASES if 'tag_grid' in r])
def test_cell_ids_bit_exact(rec):
    """Rebuild the reference's tag grid (winner neighbour per cell) from the oracle's
    integer cell ids: pins fp32 indexing, last-writer-wins and the cell-0 clobber."""
    obs2 = rec['obs2']
    B, N = obs2.shape[:2]
    n = rec['n']
    oi, inr = oracle.cell_ids(obs2, n, rec['cell_side'], front=bool(rec['front']))
    tag = np.full((B * N, n * n), rec['constant'], dtype=np.float32)
    for r in range(B * N):
        for jj in range(N - 1):
            o = oi.reshape(B * N, N - 1)[r, jj]
            tag[r, o] = (jj + 1) if inr.reshape(B * N, N - 1)[r, jj] else rec['constant']
    assert np.array_equal(tag.reshape(B * N, 1, n, n), rec['tag_grid'])


def test_known_answer_simple_grid():
    # reference tests/test_pooling.py:9-22 adapted to the current (obs1, obs2) API
    rec = next(r for r in GRID_CASES if r['name'] == 'simple')
    g = oracle.grid('occupancy', rec['obs1'], rec['obs2'], n=2, cell_side=2.0, pool_size=4, blur_size=3)
    assert g.reshape(2, 4).tolist() == [[1, 0, 0, 0], [0, 0, 0, 1]]


def test_known_answer_nan_grid():
    # reference tests/test_pooling.py:86-99
    o = np.array([[[0.0, 0.0], [np.nan, np.nan]]], dtype=np.float32)
    g = oracle.grid('occupancy', o, o, n=2, cell_side=2.0)
    assert g.reshape(2, 4).tolist() == [[0, 0, 0, 0], [0, 0, 0, 0]]


def test_known_answer_midpoint():
    # reference tests/test_pooling.py:65-83 (abs 0.01)
    o = np.array([[[0.0, 0.0], [-1.0, 0.0]]], dtype=np.float32)
    g = oracle.grid('occupancy', o, o, n=2, cell_side=2.0, pool_size=100, blur_size=99)
    np.testing.assert_allclose(g.reshape(2, 4), [[0.5, 0.5, 0, 0], [0, 0, 0.5, 0.5]], atol=0.01)


def test_known_answer_clobber_order():
    # SURVEY.md 8(a) quirk 9, demonstrated on the reference itself
    a = next(r for r in GRID_CASES if r['name'] == 'clobber_a')
    b = next(r for r in GRID_CASES if r['name'] == 'clobber_b')
    ga = oracle.grid('occupancy', a['obs1'], a['obs2'], n=4, cell_side=1.0)
    gb = oracle.grid('occupancy', b['obs1'], b['obs2'], n=4, cell_side=1.0)
    assert ga[0, 0, 0, 0] == 0.0 and gb[0, 0, 0, 0] == 1.0


def test_fp32_cell_edges():
    # float32(-6*0.6)/0.6+6 -> cell 0 ; float32(-7*0.6)/0.6+6 -> out of range
    cs = np.float32(0.6)
    obs = np.array([[[0, 0], [np.float32(-6) * cs, 0], [np.float32(-7) * cs, 0], [np.float32(6) * cs, 0]]],
                   dtype=np.float32)
    oi, inr = oracle.cell_ids(obs, 12, 0.6)
    assert inr[0, 0].tolist() == [1, 0, 0]
    assert oi[0, 0, 0] == 0 * 12 + 6


@pytest.mark.parametrize('kind', ['vanilla', 'occupancy', 'directional', 'social', 'social_goals', 'lstmlayer', 'addhidden'])
@pytest.mark.parametrize('batch', ['lin', 'rag'])
def test_lstm_forward_matches_reference(kind, batch):
    sd, cfg, d = helpers.load_lstm_case(kind)
    om = helpers.oracle_model(sd, cfg)
    xy, split, goals = d[batch + '_xy'], d[batch + '_split'], d[batch + '_goals']
    rel, pred = om.forward(xy[:9], goals, split, n_predict=12)
    helpers.assert_close_nan(rel, d[batch + '_rel_npredict'], 2e-5, 'rel n_predict')
    helpers.assert_close_nan(pred, d[batch + '_pred_npredict'], 2e-5, 'pred n_predict')
    rel, pred = om.forward(xy[:9], goals, split, prediction_truth=xy[9:20])
    helpers.assert_close_nan(rel, d[batch + '_rel_truth'], 2e-5, 'rel truth')
    helpers.assert_close_nan(pred, d[batch + '_pred_truth'], 2e-5, 'pred truth')


def test_constant_velocity():
    # classical/constant_velocity.py:4-20
    rng = np.random.RandomState(0)
    xy = rng.randn(9, 5, 2)
    out = oracle.constant_velocity(xy, 12)
    want = np.stack([xy[-1] + t * (xy[-1] - xy[-2]) for t in range(1, 13)])
    assert np.array_equal(out, want)


@pytest.mark.parametrize('tag', ['hotel', 'students'])
def test_real_scenes_match_reference(tag):
    """Config-2 Social-LSTM on real TrajNet++ scenes (tests/golden/real_cases.npz): oracle vs the reference's
    outputs, and ADE/FDE of the primaries within 1e-4 m (the tolerance BASELINE.json states)."""
    model, z = helpers.real_model()
    sd = {k: v.numpy() for k, v in model.state_dict().items()}
    om = oracle.OracleModel(sd, pool_type='social', n=16, cell_side=0.6, goal_flag=False)
    split = z[tag + '_split']
    for name in ('raw', 'centered'):
        xy = z['%s_%s_xy' % (tag, name)].astype(np.float32)
        rel, pred = om.forward(xy[:9], np.zeros((xy.shape[1], 2), np.float32), split, n_predict=12)
        helpers.assert_close_nan(rel, z['%s_%s_rel' % (tag, name)], 5e-5, 'rel')
        helpers.assert_close_nan(pred, z['%s_%s_pred' % (tag, name)], 5e-5, 'pred')
        prim = split[:-1]
        a0, f0 = helpers.ade_fde(z['%s_%s_pred' % (tag, name)][-12:, prim], xy[9:21, prim])
        a1, f1 = helpers.ade_fde(pred[-12:, prim], xy[9:21, prim])
        assert np.abs(a0 - a1).max() < 1e-4 and np.abs(f0 - f1).max() < 1e-4


def test_center_scene_matches_reference_on_real_scenes():
    from trajnetplusplusbaselines_amd import data
    z = np.load(helpers.GOLDEN + '/real_cases.npz')
    for tag in ('hotel', 'students'):
        split = z[tag + '_split']
        raw, cen = z[tag + '_raw_xy'], z[tag + '_centered_xy']
        for s in range(len(split) - 1):
            mine, rot, center = data.center_scene(raw[:, split[s]:split[s + 1]].copy(), 9)
            helpers.assert_close_nan(mine, cen[:, split[s]:split[s + 1]], 1e-12, 'center_scene')
            back = data.inverse_scene(mine, rot, center)
            helpers.assert_close_nan(back, raw[:, split[s]:split[s + 1]], 1e-9, 'inverse_scene')


def test_random_rotation_and_batcher_match_reference_arithmetic():
    """data.random_rotation == reference lstm/utils.py:10-17 under the same `random` seed (value stored in the real-scene
    fixture is not needed: the formula is checked against the rotation matrix it documents); SceneBatcher on the CPU
    device reproduces drop_distant + batch assembly + per-scene rotation."""
    import math
    import random
    from trajnetplusplusbaselines_amd import data
    z = np.load(helpers.GOLDEN + '/real_cases.npz')
    split = z['hotel_split']
    raw = z['hotel_raw_xy'].astype(np.float64)
    scenes = [raw[:, split[s]:split[s + 1]] for s in range(len(split) - 1)]
    random.seed(5)
    theta = random.random() * 2.0 * math.pi
    random.seed(5)
    rot = data.random_rotation(scenes[0])
    r = np.array([[math.cos(theta), math.sin(theta)], [-math.sin(theta), math.cos(theta)]])
    helpers.assert_close_nan(rot, scenes[0] @ r, 1e-12, 'random_rotation')
    b = data.SceneBatcher(scenes, device='cpu', drop_distant_r=None)
    xy, goals, sp = b.batch([2, 0, 5])
    want = np.concatenate([scenes[2], scenes[0], scenes[5]], axis=1).astype(np.float32)
    helpers.assert_close_nan(xy.numpy(), want, 0.0, 'batch assembly')
    assert sp.tolist() == np.concatenate([[0], np.cumsum([scenes[i].shape[1] for i in (2, 0, 5)])]).tolist()
    random.seed(9)
    thetas = [random.random() * 2.0 * math.pi for _ in range(2)]
    random.seed(9)
    xy, goals, sp = b.batch([1, 3], augment=True)
    for k, (i, th) in enumerate(zip((1, 3), thetas)):
        r = np.array([[math.cos(th), math.sin(th)], [-math.sin(th), math.cos(th)]])
        helpers.assert_close_nan(xy[:, sp[k]:sp[k + 1]].numpy(), (scenes[i] @ r).astype(np.float32), 2e-5, 'rotated scene')


def test_duplicate_cells_single_threaded_reference():
    """tests/golden/dup_cells.npz: a [9 x 13] batch with 289 duplicate (ego, cell) entries, grids computed by the
    SINGLE-THREADED reference (the multi-threaded reference's index_put_ order differs on this very batch for social
    grids, tests/test_oracle_vs_reference.py).  Last writer in ascending j wins: occupancy / directional bit-exact."""
    import os
    z = np.load(os.path.join(helpers.GOLDEN, 'dup_cells.npz'))
    assert int(z['duplicate_pairs']) > 100
    for type_ in ('occupancy', 'directional'):
        g = oracle.grid(type_, z['obs1'], z['obs2'], n=12, cell_side=0.6)
        assert np.array_equal(g, z['grid_' + type_]), type_
    enc = helpers.social_enc(dict(hidden=z['hidden'], Wh=z['Wh'], bh=z['bh']))
    g = oracle.grid('social', z['obs1'], z['obs2'], enc, n=12, cell_side=0.6)
    np.testing.assert_allclose(g, z['grid_social'], rtol=0, atol=2e-6)    # a wrong winner would be off by O(1)


def test_scatter_backward_rule_matches_reference_autograd():
    """The rule the training kernels implement for the grid scatter's backward (helpers.pair_cells_autograd_numpy: every
    in-range neighbour receives its cell's gradient, duplicates included, EXCEPT in a cell whose final value is the constant 0
    -- cell (0, 0) clobbered by an out-of-range / absent / padded slot, where lp_pool2d's derivative is zero) against the
    reference's autograd on dense crowds (tests/golden/scatter_grad.npz): per-pair gradients of the scattered values, exact."""
    import os
    z = np.load(os.path.join(helpers.GOLDEN, 'scatter_grad.npz'))
    masked_total = 0
    for k in range(int(z['num_cases'])):
        pre = 's%d_' % k
        obs2, dgrid, want = z[pre + 'obs2'], z[pre + 'dgrid'], z[pre + 'dvalues']
        n, cs, const = int(z[pre + 'n']), float(z[pre + 'cell_side']), float(z[pre + 'constant'])
        B, N = obs2.shape[:2]
        got = np.zeros_like(want)
        for b in range(B):
            cells, _ = helpers.pair_cells_autograd_numpy(obs2[b], n, cs, const)
            raw, _ = helpers.pair_cells_autograd_numpy(obs2[b], n, cs, 1.0)      # constant != 0: nothing is masked
            masked_total += int(((raw >= 0) & (cells < 0)).sum())
            for i in range(N):
                for j in range(N):
                    if j != i and cells[i, j] >= 0:
                        c = cells[i, j]
                        got[b, i, j - (j > i)] = dgrid[b * N + i, :, c // n, c % n]
        np.testing.assert_array_equal(got, want)
    assert masked_total > 0, 'the fixture must contain in-range pairs in a clobbered cell (0, 0)'
