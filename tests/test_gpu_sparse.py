"""GPU parity of the sparse first embedding layer (csrc/pool_embed_sparse.hip): winner table + per-track values
-> Linear on the (never materialised) social grid, against the oracle's dense grid + Linear, and against the dense
MFMA path of the same library."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import oracle
from trajnetplusplusbaselines_amd import _lib, synth
from trajnetplusplusbaselines_amd.lstm import LSTM, GridBasedPooling

pytestmark = pytest.mark.gpu


def run_sparse(obs1, obs2, enc, starts, n_max, n, cell_side, W, b, relu=True):
    dev = torch.device('cuda')
    M, C = enc.shape
    N1 = W.shape[0]
    ncell = n * n
    o1, o2 = torch.tensor(obs1).to(dev), torch.tensor(obs2).to(dev)
    e = torch.tensor(enc).to(dev)
    st = torch.tensor(starts, dtype=torch.int32).to(dev)
    B = len(starts) - 1
    winners = torch.empty(M, ncell, dtype=torch.int16, device=dev)
    L = _lib.lib()
    _lib.check(L.tnp_pool_grid_forward(_lib.POOL_SOCIAL, _lib.ptr(o1), _lib.ptr(o2), _lib.ptr(e), C, _lib.ptr(st), B,
                                       n_max, None, n, C, float(np.float32(cell_side)), n / 2, n / 2, 0.0, None, 0,
                                       _lib.ptr(winners), _lib.stream_ptr()), 'grid')
    row_base = torch.empty(M, dtype=torch.int32, device=dev)
    _lib.check(L.tnp_row_base(_lib.ptr(st), B, _lib.ptr(row_base), _lib.stream_ptr()), 'row_base')
    Wt = torch.tensor(W).to(dev)
    Wcm = Wt.view(N1, C, ncell).permute(2, 1, 0).contiguous()
    bt = torch.tensor(b).to(dev)
    need = L.tnp_pool_embed_sparse_workspace_bytes(M, N1, ncell)
    ws = torch.empty(max(need, 16), dtype=torch.uint8, device=dev)
    out = torch.empty(M, N1, dtype=torch.float32, device=dev)
    _lib.check(L.tnp_pool_embed_sparse_forward(_lib.ptr(winners), _lib.ptr(e), C, _lib.ptr(row_base), _lib.ptr(Wcm),
                                               _lib.ptr(bt), M, ncell, C, N1, int(relu), _lib.ptr(out), N1,
                                               _lib.ptr(ws), need, _lib.stream_ptr()), 'sparse')
    return out.cpu().numpy()


@pytest.mark.parametrize('scenes,agents,n,C,N1', [(3, 7, 8, 8, 64), (64, 32, 16, 16, 1024), (5, 40, 16, 16, 260),
                                                   (20, 13, 12, 4, 128), (2, 130, 16, 32, 256), (3, 130, 16, 16, 192),
                                                   (4, 20, 24, 8, 128)])  # 576 cells: cell-range fallback kernel
def test_sparse_embedding_matches_dense_oracle(scenes, agents, n, C, N1):
    rng = np.random.RandomState(scenes * 31 + agents)
    B, N = scenes, agents
    obs2 = (rng.rand(B, N, 2).astype(np.float32) * 8 - 4)
    obs2[rng.rand(B, N) < 0.15] = np.nan
    obs1 = obs2 - np.float32(0.1)
    enc = rng.randn(B, N, C).astype(np.float32)
    W = (rng.randn(N1, C * n * n) / np.sqrt(C * 30)).astype(np.float32)
    b = rng.randn(N1).astype(np.float32)
    grid = oracle.grid('social', obs1, obs2, enc, n=n, cell_side=0.6, C=C).reshape(B * N, -1)
    want = oracle.linear(grid, W, b, relu=True)
    starts = np.arange(0, B * N + 1, N)
    got = run_sparse(obs1.reshape(-1, 2), obs2.reshape(-1, 2), enc.reshape(-1, C), starts, N, n, 0.6, W, b)
    err = float(np.abs(got - want).max())
    assert err < 2e-5 * max(1.0, float(np.abs(want).max())), err


def test_sparse_ragged_scenes():
    rng = np.random.RandomState(7)
    sizes = [5, 1, 9, 3, 9, 2]
    n_max, n, C, N1 = max(sizes), 8, 8, 96
    starts = np.concatenate([[0], np.cumsum(sizes)])
    M = int(starts[-1])
    flat = (rng.rand(M, 2).astype(np.float32) * 4 - 2)
    enc = rng.randn(M, C).astype(np.float32)
    W = rng.randn(N1, C * n * n).astype(np.float32) / 8
    b = rng.randn(N1).astype(np.float32)
    padded = np.full((len(sizes), n_max, 2), np.nan, dtype=np.float32)
    pe = np.zeros((len(sizes), n_max, C), dtype=np.float32)
    for s, ns in enumerate(sizes):
        padded[s, :ns] = flat[starts[s]:starts[s + 1]]
        pe[s, :ns] = enc[starts[s]:starts[s + 1]]
    grid = oracle.grid('social', padded, padded, pe, n=n, cell_side=0.6, C=C).reshape(len(sizes), n_max, -1)
    got = run_sparse(flat, flat, enc, starts, n_max, n, 0.6, W, b)
    for s, ns in enumerate(sizes):
        want = oracle.linear(grid[s, :ns], W, b, relu=True)
        np.testing.assert_allclose(got[starts[s]:starts[s + 1]], want, atol=3e-5)


@pytest.mark.parametrize('n1,latent', [(1024, 16), (192, 8), (64, 4), (100, 16)])
def test_forward_sparse_equals_dense_path(n1, latent):
    """Whole Social-LSTM forward: the sparse first layer and the dense MFMA first layer agree to fp32 rounding.  The
    sequence driver hands the register-accumulator kernel the quad-major weight copy when N1 % 64 == 0 (192: the last
    column block is half empty; C = 8 / 4: shorter weight sets), else the cell-major one (100)."""
    torch.manual_seed(3)
    pool = GridBasedPooling(type_='social', hidden_dim=128, cell_side=0.6, n=16, out_dim=256,
                            embedding_arch='two_layer', layer_dims=[n1], latent_dim=latent)
    model = LSTM(pool=pool).eval().cuda()
    xy, split = synth.ragged_crowd(24, 3, 40, seed=4)
    goals = torch.zeros(xy.shape[1], 2)
    model.sparse_embedding = True
    rel_s, pred_s = model(xy[:9], goals, split, n_predict=12)
    model.sparse_embedding = False
    rel_d, pred_d = model(xy[:9], goals, split, n_predict=12)
    assert torch.equal(torch.isnan(pred_s), torch.isnan(pred_d))
    assert (torch.nan_to_num(pred_s) - torch.nan_to_num(pred_d)).abs().max().item() < 2e-5
    assert (torch.nan_to_num(rel_s) - torch.nan_to_num(rel_d)).abs().max().item() < 2e-5


def test_forward_with_scenes_larger_than_a_wave():
    """Scenes of up to 90 tracks: the fused winner build of the register-accumulator kernel votes in 64-neighbour chunks, a
    tile of 64 egos lies inside one scene, and the sparse path must still agree with the dense one."""
    torch.manual_seed(5)
    pool = GridBasedPooling(type_='social', hidden_dim=128, cell_side=0.6, n=16, out_dim=256,
                            embedding_arch='two_layer', layer_dims=[256], latent_dim=16)
    model = LSTM(pool=pool).eval().cuda()
    xy, split = synth.ragged_crowd(5, 60, 90, seed=11)
    goals = torch.zeros(xy.shape[1], 2)
    with torch.no_grad():
        model.sparse_embedding = True
        rel_s, pred_s = model(xy[:9], goals, split, n_predict=12)
        model.sparse_embedding = False
        rel_d, pred_d = model(xy[:9], goals, split, n_predict=12)
    assert torch.equal(torch.isnan(pred_s), torch.isnan(pred_d))
    assert (torch.nan_to_num(pred_s) - torch.nan_to_num(pred_d)).abs().max().item() < 3e-5


TILES = [(64, 2), (32, 2), (32, 1), (16, 1), (8, 1), (4, 1)]


@pytest.mark.parametrize('n1,latent,scenes,lo,hi', [(1024, 16, 6, 3, 40), (192, 8, 3, 20, 50), (64, 4, 1, 36, 36), (1024, 16, 2, 70, 90),
                                                    (128, 16, 9, 1, 6)])
def test_sparse_tile_does_not_change_the_result(n1, latent, scenes, lo, hi):
    """Round 6: the register-accumulator kernel picks its ego tile from the batch size (4 .. 64 egos x 64 / 128 columns).  The cell
    -> wave-group map and the order of the final sum are the tile's own business nowhere: every tile must give the SAME forward
    bit for bit (training forward with the winner table included), on ragged scenes, scenes larger than a tile / a wave,
    N1 % 128 == 64, C = 4 / 8 / 16."""
    torch.manual_seed(11)
    pool = GridBasedPooling(type_='social', hidden_dim=128, cell_side=0.6, n=16, out_dim=256,
                            embedding_arch='two_layer', layer_dims=[n1], latent_dim=latent)
    model = LSTM(pool=pool).eval().cuda()
    xy, split = synth.ragged_crowd(scenes, lo, hi, seed=5, nan_frac=0.1)
    goals = torch.zeros(xy.shape[1], 2)
    outs = []
    try:
        for te, ncs in TILES:
            _lib.tuning_set('sparse_tile', (te << 8) | ncs)
            with torch.no_grad():
                outs.append(model(xy[:9], goals, split, n_predict=12))
        _lib.tuning_set('sparse_tile', 0)
        with torch.no_grad():
            auto = model(xy[:9], goals, split, n_predict=12)
    finally:
        _lib.tuning_set('sparse_tile', 0)
    for rel, pred in outs[1:] + [auto]:
        assert torch.equal(torch.nan_to_num(rel), torch.nan_to_num(outs[0][0])) and torch.equal(torch.isnan(pred), torch.isnan(outs[0][1]))
        assert torch.equal(torch.nan_to_num(pred), torch.nan_to_num(outs[0][1]))
    # ... and the dense first layer agrees to rounding (the tiles are not merely equal to each other)
    model.sparse_embedding = False
    with torch.no_grad():
        rel_d, pred_d = model(xy[:9], goals, split, n_predict=12)
    assert (torch.nan_to_num(outs[0][1]) - torch.nan_to_num(pred_d)).abs().max().item() < 3e-5


def test_sparse_tile_training_gradients_agree():
    """The small tiles write the winner table the backward reads (FG + winners_out): gradients of a batch_size-8 step are the
    same whichever tile ran the forward."""
    from trajnetplusplusbaselines_amd.lstm import PredictionLoss
    torch.manual_seed(2)
    pool = GridBasedPooling(type_='social', hidden_dim=128, cell_side=0.6, n=16, out_dim=256,
                            embedding_arch='two_layer', layer_dims=[1024], latent_dim=16)
    model = LSTM(pool=pool).cuda().train()
    xy, split = synth.ragged_crowd(8, 8, 60, seed=3, nan_frac=0.15)
    xy = xy.cuda()
    goals = torch.zeros(xy.shape[1], 2, device='cuda')
    crit = PredictionLoss()
    grads = []
    try:
        for te, ncs in [(64, 2), (16, 1), (4, 1)]:
            _lib.tuning_set('sparse_tile', (te << 8) | ncs)
            model.zero_grad(set_to_none=True)
            rel, _ = model(xy[:9], goals, split, prediction_truth=xy[9:-1])
            loss = crit(rel[-12:], xy[9:21] - xy[8:20], split)
            loss.backward()
            grads.append([p.grad.clone() for p in model.parameters() if p.grad is not None])
    finally:
        _lib.tuning_set('sparse_tile', 0)
    for g in grads[1:]:
        assert len(g) == len(grads[0]) and all(torch.equal(a, b) for a, b in zip(g, grads[0]))


def _small_step_grads(model, xy, goals, split):
    from trajnetplusplusbaselines_amd.lstm import PredictionLoss
    model.zero_grad(set_to_none=True)
    rel, _ = model(xy[:9], goals, split, prediction_truth=xy[9:-1])
    PredictionLoss()(rel[-12:], xy[9:21] - xy[8:20], split).backward()
    return {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}


@pytest.mark.parametrize('n1,scenes,lo,hi', [(1024, 8, 8, 60), (512, 3, 30, 70), (1024, 1, 2, 2)])
def test_small_batch_backward_kernels_agree(n1, scenes, lo, hi):
    """Round 6, batch_size-8 regime of the backward sweep: the weight gradients' split plans (waves per cell of the sparse first
    layer, rows of K per split of the dense ones) follow the batch size; a plan only moves the order of the sums, so every plan
    gives the same gradients to rounding, and the same plan the same bits twice."""
    torch.manual_seed(4)
    pool = GridBasedPooling(type_='social', hidden_dim=128, cell_side=0.6, n=16, out_dim=256,
                            embedding_arch='two_layer', layer_dims=[n1], latent_dim=16)
    model = LSTM(pool=pool).cuda().train()
    xy, split = synth.ragged_crowd(scenes, lo, hi, seed=9, nan_frac=0.15)
    xy = xy.cuda()
    goals = torch.zeros(xy.shape[1], 2, device='cuda')
    try:
        base = _small_step_grads(model, xy, goals, split)
        again = _small_step_grads(model, xy, goals, split)
        plans = []
        for swg, rows in ((16 * 4 + 1, 256), (16 * 16 + 2, 32), (16 * 8 + 2, 64), (16 * 4 + 2, 128), (16 * 16 + 1, 256)):
            _lib.tuning_set('sparse_wgrad_plan', swg)
            _lib.tuning_set('wgrad_min_rows', rows)
            plans.append(_small_step_grads(model, xy, goals, split))
    finally:
        _lib.tuning_set('sparse_wgrad_plan', 0)
        _lib.tuning_set('wgrad_min_rows', 128)
    assert base.keys() == again.keys() and all(torch.equal(base[k], again[k]) for k in base)
    for g in plans:
        for k in base:
            scale = base[k].abs().max().item() + 1e-12
            assert (g[k] - base[k]).abs().max().item() <= 2e-5 * scale, k
    with pytest.raises(RuntimeError):
        _lib.tuning_set('sparse_wgrad_plan', 16 * 3 + 1)
        try:
            _small_step_grads(model, xy, goals, split)
        finally:
            _lib.tuning_set('sparse_wgrad_plan', 0)


def test_side_stream_weight_gradient_equals_the_single_stream_one(monkeypatch):
    """Opt-in (lstm/training.py _SIDE_STREAM, measured slower): the sparse first layer's weight gradient on a side stream beside
    the grouped weight-gradient launch.  Same kernels, same operands -- the same gradients bit for bit as on one stream, step
    after step (buffers the side kernels use must survive until the join: a freed block would be handed to the main stream's
    next allocation)."""
    from trajnetplusplusbaselines_amd.lstm import training
    torch.manual_seed(6)
    pool = GridBasedPooling(type_='social', hidden_dim=128, cell_side=0.6, n=16, out_dim=256,
                            embedding_arch='two_layer', layer_dims=[1024], latent_dim=16)
    model = LSTM(pool=pool).cuda().train()
    runs = {}
    for mode in ('0', '1'):
        monkeypatch.setattr(training, '_SIDE_STREAM', mode)
        out = []
        for seed in (1, 2, 3, 4):
            xy, split = synth.ragged_crowd(8, 8, 60, seed=seed, nan_frac=0.15)
            xy = xy.cuda()
            out.append(_small_step_grads(model, xy, torch.zeros(xy.shape[1], 2, device='cuda'), split))
        torch.cuda.synchronize()
        runs[mode] = out
    for mode in ('1',):
        for a, b in zip(runs['0'], runs[mode]):
            assert a.keys() == b.keys() and all(torch.equal(a[k], b[k]) for k in a), mode
