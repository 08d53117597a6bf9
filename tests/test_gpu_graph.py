"""Inference forwards replayed as hipGraphs (LSTM._forward_graphed, opt-in): a call shape that comes back is captured once --
the sequence driver's ~80 launches over static buffers -- and replayed with one host call.  Same kernels, same order, same
arguments as the eager path: every output must be BIT-identical to it."""
import os

import numpy as np
import pytest
import torch

from tests import helpers

pytestmark = pytest.mark.gpu


def _model(kind='social', goal_flag=False):
    from trajnetplusplusbaselines_amd import lstm as L
    torch.manual_seed(3)
    pool = {'social': lambda: L.GridBasedPooling(type_='social', hidden_dim=128, cell_side=0.6, n=16, out_dim=256,
                                                 embedding_arch='two_layer', layer_dims=[1024], latent_dim=16),
            'directional': lambda: L.GridBasedPooling(type_='directional', hidden_dim=128, cell_side=0.6, n=12, out_dim=256),
            'vanilla': lambda: None,
            'attentionmlp': lambda: L.AttentionMLPPooling(hidden_dim=128, out_dim=32),
            'nn_lstm': lambda: L.NearestNeighborLSTM(n=4, hidden_dim=64, out_dim=32)}[kind]()
    return L.LSTM(pool=pool, goal_flag=goal_flag).cuda().eval()


def _graphs(model):
    from trajnetplusplusbaselines_amd.lstm.lstm import _GraphedForward
    return [v for v in (model._graphs or {}).values() if isinstance(v, _GraphedForward)]


@pytest.mark.parametrize('kind', ['social', 'directional', 'vanilla', 'attentionmlp', 'nn_lstm'])
def test_replayed_forward_is_bitwise_the_eager_forward(kind):
    """Six calls of one shape with different crowds (same scene structure), free-running and teacher-forced: calls 1-2 run
    eagerly, call 3 captures, 4-6 replay; every result equals the eager model's, bit for bit, NaN pattern included."""
    from trajnetplusplusbaselines_amd import synth
    model = _model(kind, goal_flag=(kind == 'directional'))
    crowds = []
    for seed in range(6):
        xy, split = synth.ragged_crowd(5, 3, 40, seed=70)             # one structure ...
        g = torch.Generator().manual_seed(seed)
        xy = xy + 0.3 * torch.randn(xy.shape, generator=g)            # ... six different crowds (the NaN pattern stays)
        crowds.append((xy, split))
    M = crowds[0][0].shape[1]
    goals = torch.rand(M, 2, generator=torch.Generator().manual_seed(9)) * 6 - 3
    with torch.no_grad():
        for mode in ('free', 'truth'):
            eager = []
            for xy, split in crowds:
                kw = dict(n_predict=12) if mode == 'free' else dict(prediction_truth=xy[9:20].clone())
                eager.append([t.cpu().numpy() for t in model(xy[:9], goals, split, **kw)])
            for i, (xy, split) in enumerate(crowds):
                kw = dict(n_predict=12) if mode == 'free' else dict(prediction_truth=xy[9:20].clone())
                got = [t.cpu().numpy() for t in model(xy[:9], goals, split, graph=True, **kw)]
                for a, b in zip(got, eager[i]):
                    assert np.array_equal(a, b, equal_nan=True), (kind, mode, i)
    entries = _graphs(model)
    assert len(entries) == 2 and all(e.replays == 4 for e in entries), [(e.replays) for e in entries]


def test_weights_updated_in_place_are_seen_by_the_replay():
    """The captured kernel arguments are ADDRESSES: an in-place update of the parameters (an optimiser step between two
    evaluation passes) is seen by the next replay; the re-laid-out copies of the first embedding layer follow the
    parameter's version, so that shape is captured again."""
    from trajnetplusplusbaselines_amd import synth
    model = _model('social')
    model.graph_replay = True
    xy, split = synth.linear_crowd(8, 32, seed=5)
    goals = torch.zeros(xy.shape[1], 2)
    with torch.no_grad():
        for _ in range(5):
            a = model(xy[:9], goals, split, n_predict=12)[1].cpu().numpy()
        assert _graphs(model)[0].replays >= 2
        for p in model.parameters():
            p.mul_(1.01)
        outs = [model(xy[:9], goals, split, n_predict=12)[1].cpu().numpy() for _ in range(5)]
        model.graph_replay = False
        want = model(xy[:9], goals, split, n_predict=12)[1].cpu().numpy()
    assert not np.array_equal(a, want)
    for o in outs:
        assert np.array_equal(o, want)


def test_predictor_per_scene_calls_replay_and_match():
    """The evaluator's pattern (one LSTMPredictor call per scene, lstm/lstm.py:285-313): scenes with the same number of
    agents share a graph; predictions equal those of a predictor with graph_replay = False."""
    from trajnetplusplusbaselines_amd import data
    from trajnetplusplusbaselines_amd.lstm import LSTMPredictor
    z = np.load(os.path.join(helpers.GOLDEN, 'real_eval.npz'))
    xy_all, split = z['f0_xy'], z['f0_split']
    scenes = [xy_all[:, split[s]:split[s + 1]] for s in range(len(split) - 1)]
    scenes = scenes + scenes + scenes                                   # every agent count comes back
    model = _model('social')
    fast, slow = LSTMPredictor(model), LSTMPredictor(model)
    fast.graph_replay = True
    assert slow.graph_replay is False                                   # opt-in: see LSTMPredictor.graph_replay
    for xy in scenes:
        paths = data.xy_to_paths(xy)
        a = fast(paths, np.zeros((xy.shape[1], 2)), n_predict=12, obs_length=9)
        b = slow(paths, np.zeros((xy.shape[1], 2)), n_predict=12, obs_length=9)
        assert np.array_equal(a[0][0], b[0][0], equal_nan=True) and np.array_equal(a[0][1], b[0][1], equal_nan=True)
    assert sum(e.replays for e in _graphs(model)) >= len(scenes) // 3


def test_batches_in_flight_replay_on_their_own_streams():
    """predict_batches with two batches in flight: every stream owns its graph (static buffers and workspace are per graph);
    results equal predict_batch."""
    from trajnetplusplusbaselines_amd import data, synth
    from trajnetplusplusbaselines_amd.lstm import LSTMPredictor
    model = _model('social')
    pred = LSTMPredictor(model)
    pred.graph_replay = True
    batches = []
    for seed in range(2):
        xy, split = synth.linear_crowd(6, 20, seed=40 + seed)
        batches.append([(data.xy_to_paths(xy[:, split[s]:split[s + 1]].numpy()), None) for s in range(6)])
    many = batches * 5
    ref = LSTMPredictor(model)
    want = [ref.predict_batch(b) for b in batches]
    got = pred.predict_batches(many, in_flight=2)
    for i, res in enumerate(got):
        for s, scene in enumerate(res):
            assert np.array_equal(scene[0][0], want[i % 2][s][0][0]) and np.array_equal(scene[0][1], want[i % 2][s][0][1], equal_nan=True)
    assert len(_graphs(model)) == 2 and sum(e.replays for e in _graphs(model)) >= 4


def test_graph_cache_is_bounded():
    from trajnetplusplusbaselines_amd import synth
    model = _model('vanilla')
    model.graph_replay = True
    model._GRAPH_MAX = 3
    with torch.no_grad():
        for agents in (2, 3, 4, 5, 6):
            xy, split = synth.linear_crowd(2, agents, seed=1)
            for _ in range(4):
                model(xy[:9], torch.zeros(xy.shape[1], 2), split, n_predict=12)
    assert len(_graphs(model)) == 3


def test_replay_survives_a_cleared_scene_index_cache():
    """a graph is keyed by the CONTENT of the scene structure, its captured kernels hold raw pointers into the
    SceneIndex tables (starts / primary / slots).  The entry must keep that index alive: capture a shape, clear the index cache,
    collect, churn the allocator with fresh tensors of the tables' sizes, replay -- the result must still be the eager one."""
    import gc
    from trajnetplusplusbaselines_amd import _lib, synth
    model = _model('social')
    xy, split = synth.ragged_crowd(6, 3, 30, seed=12)
    goals = torch.zeros(xy.shape[1], 2)
    with torch.no_grad():
        eager = [t.cpu().numpy() for t in model(xy[:9], goals, split, n_predict=12)]
        for _ in range(3):
            model(xy[:9], goals, split, n_predict=12, graph=True)            # third call captures
        assert len(_graphs(model)) == 1 and _graphs(model)[0].idx is not None
        _lib.SceneIndex._cache.clear()
        gc.collect()
        torch.cuda.synchronize()
        junk = [torch.full((n,), 12345, dtype=torch.int32, device='cuda') for n in (7, 7, 64, 64, 128, 128, 256, 512) for _ in range(8)]
        junk += [torch.full((n,), 77, dtype=torch.uint8, device='cuda') for n in (64, 128, 256, 512) for _ in range(8)]
        torch.cuda.synchronize()
        other, osplit = synth.ragged_crowd(9, 2, 20, seed=13)                # new index tables land in recycled blocks
        model(other[:9], torch.zeros(other.shape[1], 2), osplit, n_predict=12)
        got = [t.cpu().numpy() for t in model(xy[:9], goals, split, n_predict=12, graph=True)]
        assert _graphs(model)[0].replays >= 1
    for a, b in zip(got, eager):
        assert np.array_equal(a, b, equal_nan=True)
    del junk
