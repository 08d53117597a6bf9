"""Training path: loss.backward() through LSTM.forward on the GPU against the reference's autograd gradients on the
same weights and batch (tests/golden/grad_cases.npz: trainer loss PredictionLoss(rel[-12:], targets) * batch_size,
lstm/trainer.py:252-265, plus a term on the predicted positions)."""
import os

import numpy as np
import pytest
import torch

from tests import helpers

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(helpers.GOLDEN, 'grad_cases.npz'))


def build(kind):
    from trajnetplusplusbaselines_amd.lstm import LSTM, GridBasedPooling, NearestNeighborMLP
    pool = None
    if kind == 'nn':
        pool = NearestNeighborMLP(n=4, out_dim=32)
    if kind == 'social':
        pool = GridBasedPooling(type_='social', hidden_dim=128, cell_side=0.6, n=8, out_dim=64,
                                embedding_arch='two_layer', layer_dims=[128], latent_dim=8)
    elif kind == 'directional':
        pool = GridBasedPooling(type_='directional', hidden_dim=128, cell_side=0.6, n=12, out_dim=64)
    model = LSTM(pool=pool)
    pre = kind + '_sd_'
    model.load_state_dict({k[len(pre):]: torch.tensor(GOLD[k]) for k in GOLD.files if k.startswith(pre)})
    return model.cuda().train()


@pytest.mark.parametrize('kind', ['vanilla', 'directional', 'social', 'nn'])
def test_gradients_match_reference_autograd(kind):
    from trajnetplusplusbaselines_amd.lstm import PredictionLoss
    model = build(kind)
    xy, split = torch.tensor(GOLD[kind + '_xy']), torch.tensor(GOLD[kind + '_split'])
    M = xy.shape[1]
    observed, truth = xy[:9].clone(), xy[9:20].clone()
    targets = (xy[9:21] - xy[8:20]).cuda()
    rel, pred = model(observed, torch.zeros(M, 2), split, truth)
    assert rel.requires_grad and pred.requires_grad
    loss = PredictionLoss()(rel[-12:], targets, split) * 8 + 0.1 * torch.nan_to_num(pred[-12:, split[:-1].cuda()]).pow(2).mean()
    np.testing.assert_allclose(float(loss), float(GOLD[kind + '_loss']), rtol=2e-5)
    loss.backward()
    worst = 0.0
    for name, p in model.named_parameters():
        want = GOLD[kind + '_grad_' + name]
        if p.grad is None:   # unused parameter: the reference's grad is None too (stored as zeros)
            assert not np.any(want), name
            continue
        got = p.grad.cpu().numpy()
        scale = max(1e-6, float(np.abs(want).max()))
        err = float(np.abs(got - want).max()) / scale
        worst = max(worst, err)
        assert err < 2e-3, '%s: relative error %.2e (scale %.2e)' % (name, err, scale)
    print(kind, 'worst relative gradient error %.2e' % worst)


def test_one_adam_step_tracks_reference_free_run():
    """An optimizer step changes the predictions and keeps them finite; eval-mode forward (fused driver) agrees with
    the train-mode forward (step-by-step) on the same weights."""
    model = build('social')
    xy, split = torch.tensor(GOLD['social_xy']), torch.tensor(GOLD['social_split'])
    M = xy.shape[1]
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    from trajnetplusplusbaselines_amd.lstm import PredictionLoss
    rel, pred = model(xy[:9], torch.zeros(M, 2), split, xy[9:20].clone())
    loss0 = PredictionLoss()(rel[-12:], (xy[9:21] - xy[8:20]).cuda(), split) * 8
    loss0.backward()
    opt.step()
    opt.zero_grad()
    rel1, pred1 = model(xy[:9], torch.zeros(M, 2), split, xy[9:20].clone())
    loss1 = PredictionLoss()(rel1[-12:], (xy[9:21] - xy[8:20]).cuda(), split) * 8
    assert float(loss1) < float(loss0)
    model.eval()
    with torch.no_grad():
        rel_e, pred_e = model(xy[:9], torch.zeros(M, 2), split, xy[9:20].clone())
    helpers.assert_close_nan(pred_e.cpu().numpy(), pred1.detach().cpu().numpy(), 1e-5, 'train vs eval forward')


def test_training_curve_matches_reference():
    """Six optimisation steps (train_step.train_batch == Trainer.train_batch, lstm/trainer.py:229-269, Adam lr 1e-3
    wd 1e-4) from the reference's initial weights: loss trajectory, trained weights and the free-running prediction
    of the trained model against the reference's own run (tests/golden/train_curve.npz)."""
    from trajnetplusplusbaselines_amd.lstm import LSTM, GridBasedPooling, PredictionLoss
    from trajnetplusplusbaselines_amd.lstm.train_step import train_batch
    z = np.load(os.path.join(helpers.GOLDEN, 'train_curve.npz'))
    pool = GridBasedPooling(type_='social', hidden_dim=128, cell_side=0.6, n=8, out_dim=64,
                            embedding_arch='two_layer', layer_dims=[128], latent_dim=8)
    model = LSTM(pool=pool)
    model.load_state_dict({k[3:]: torch.tensor(z[k]) for k in z.files if k.startswith('sd_')})
    model = model.cuda()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
    crit = PredictionLoss()
    batches = [(torch.tensor(z['b%d_xy' % i]), torch.tensor(z['b%d_split' % i])) for i in range(2)]
    losses = []
    for it in range(6):
        xy, split = batches[it % 2]
        losses.append(train_batch(model, opt, crit, xy, torch.zeros(xy.shape[1], 2), split, 9, 12))
    np.testing.assert_allclose(losses, z['losses'], rtol=2e-5)
    # Adam amplifies rounding differences of tiny gradients (update = lr * g / (|g| + eps)): compare the weights
    # with a tolerance of half a step (lr = 1e-3); parameters the forward never touches must not move at all
    for k in z.files:
        if k.startswith('final_sd_'):
            got = model.state_dict()[k[len('final_sd_'):]].cpu().numpy()
            assert np.abs(got - z[k]).max() < 5e-4, k
    model.eval()
    xy, split = batches[0]
    with torch.no_grad():
        _, pred = model(xy[:9], torch.zeros(xy.shape[1], 2), split, n_predict=12)
    prim = z['b0_split'][:-1]
    truth = z['b0_xy'][9:21, prim]
    a0, f0 = helpers.ade_fde(z['final_pred'][-12:, prim], truth)
    a1, f1 = helpers.ade_fde(pred.cpu().numpy()[-12:, prim], truth)
    assert np.abs(a0 - a1).max() < 1e-4 and np.abs(f0 - f1).max() < 1e-4   # the 1e-4 m bar, after training
    print('losses', losses, 'ref', z['losses'].tolist(), 'dADE %.2e dFDE %.2e' % (np.abs(a0 - a1).max(), np.abs(f0 - f1).max()))


@pytest.mark.parametrize('latent_dim,n,layer', [(16, 12, 192), (4, 8, 64), (8, 16, 128), (32, 6, 256), (16, 16, 1024)])
def test_sparse_first_layer_backward_equals_dense_backward(latent_dim, n, layer):
    """The social first-layer gradients computed from the winner tables (tnp_social_dgrid_cells, tnp_sparse_wgrad)
    equal the dense GEMM backward on a crowd with duplicates, empty cells and ragged scenes, for every channel count
    the sparse kernels are instantiated for (C = 32 runs two 16-channel MFMA blocks); the golden social case above runs
    the sparse form against the reference's autograd."""
    import ctypes
    from trajnetplusplusbaselines_amd import _lib
    from trajnetplusplusbaselines_amd.lstm import LSTM, GridBasedPooling, PredictionLoss
    torch.manual_seed(5)
    g = torch.Generator().manual_seed(11)
    sizes = [7, 1, 19, 12, 3, 30]
    split = torch.tensor([0] + list(np.cumsum(sizes)))
    M = int(split[-1])
    start = torch.rand(M, 2, generator=g) * 5.0
    vel = (torch.rand(M, 2, generator=g) - 0.5) * 0.8
    xy = start[None] + vel[None] * torch.arange(21)[:, None, None] * 0.4 + 0.02 * torch.randn(21, M, 2, generator=g)
    xy[:5, 9] = float('nan')        # a late-appearing neighbour
    xy[15:, 20] = float('nan')      # one that leaves
    grads = {}
    for sparse in (True, False):
        torch.manual_seed(3)
        pool = GridBasedPooling(type_='social', hidden_dim=128, cell_side=0.5, n=n, out_dim=64,
                                embedding_arch='two_layer', layer_dims=[layer], latent_dim=latent_dim)
        model = LSTM(pool=pool).cuda().train()
        model.sparse_backward = sparse
        m, keep, _ = model._descriptor()
        assert _lib.lib().tnp_lstm_sparse_first_layer(ctypes.byref(m), M) == 1
        rel, pred = model(xy[:9].clone(), torch.zeros(M, 2), split, xy[9:20].clone())
        targets = (xy[9:21] - xy[8:20]).cuda()
        loss = PredictionLoss()(rel[-12:], targets, split) * len(sizes)
        loss.backward()
        grads[sparse] = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    assert grads[True].keys() == grads[False].keys()
    for n in grads[True]:
        a, b = grads[True][n], grads[False][n]
        scale = max(1e-6, float(b.abs().max()))
        assert float((a - b).abs().max()) / scale < 2e-5, n


@pytest.mark.parametrize('K,Mo,No', [(1000, 5, 128), (777, 62, 2), (4099, 512, 320), (38912, 256, 288), (33, 130, 70),
                                     (2048, 16, 128),
                                     # whole 64 x 64 blocks with 1, 2, 3 and 5 operand stages per split (the pipelined loop's
                                     # prologue / pair / odd-tail paths) and a long one with whole stages everywhere
                                     (16, 64, 64), (32, 128, 64), (48, 64, 128), (80, 128, 128), (8192, 192, 256)])
def test_wgrad_split_k_matches_fp64(K, Mo, No):
    """tnp_wgrad (dy^T x with K split across workgroups + bias column sums) against an fp64 product, ragged shapes,
    strided operands; two runs are bit-identical (fixed reduction order)."""
    from trajnetplusplusbaselines_amd import _lib
    g = torch.Generator().manual_seed(K + Mo)
    dy_full = torch.randn(K, Mo + 3, generator=g).cuda()
    x_full = torch.randn(K, No + 5, generator=g).cuda()
    dy, x = dy_full[:, 1:Mo + 1], x_full[:, 2:No + 2]
    L = _lib.lib()
    nbytes = L.tnp_wgrad_workspace_bytes(Mo, No, K)
    ws = torch.empty(nbytes, dtype=torch.uint8, device='cuda')
    outs = []
    for _ in range(2):
        dw = torch.full((Mo, No), float('nan'), device='cuda')
        db = torch.full((Mo,), float('nan'), device='cuda')
        _lib.check(L.tnp_wgrad(_lib.ptr(dy), dy.stride(0), _lib.ptr(x), x.stride(0), K, Mo, No, _lib.ptr(dw), No, _lib.ptr(db),
                               _lib.ptr(ws), nbytes, _lib.stream_ptr()), 'tnp_wgrad')
        outs.append((dw.clone(), db.clone()))
    want = dy.double().t() @ x.double()
    scale = float(want.abs().max())
    assert float((outs[0][0].double() - want).abs().max()) / scale < 2e-6
    wb = dy.double().sum(0)
    assert float((outs[0][1].double() - wb).abs().max()) / float(wb.abs().max()) < 2e-6
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize('kind', ['hiddenstatemlp', 'attentionmlp', 'nn_lstm', 'traj_pool', 'addhidden'])
def test_nongrid_gradients_match_reference_autograd(kind):
    """Training through HiddenStateMLPPooling (max-pool routing, Linear(2 -> dim) embeddings behind it, hidden embedding,
    out_projection) and AttentionMLPPooling (softmax attention over the slots with the linear maps folded, gradients
    un-folded to wq / wk / wv / in_proj / out_proj / out_projection) against the reference's autograd on the same weights
    and batch (grad_cases_nongrid.npz)."""
    from trajnetplusplusbaselines_amd.lstm import (LSTM, HiddenStateMLPPooling, AttentionMLPPooling, NearestNeighborLSTM,
                                                   TrajectronPooling, PredictionLoss)
    G = np.load(os.path.join(helpers.GOLDEN, 'grad_cases_nongrid.npz'))
    if kind in ('hiddenstatemlp', 'attentionmlp'):
        cls = HiddenStateMLPPooling if kind == 'hiddenstatemlp' else AttentionMLPPooling
        pool = cls(hidden_dim=128, mlp_dim=96, mlp_dim_spatial=32, mlp_dim_vel=32, out_dim=32)
    elif kind == 'nn_lstm':     # BPTT through the interaction encoder's own pool_lstm
        pool = NearestNeighborLSTM(n=4, hidden_dim=64, out_dim=32)
    elif kind == 'traj_pool':
        pool = TrajectronPooling(hidden_dim=64, out_dim=32)
    else:   # LSTM(pool_to_input=False): the interaction vector is added to the hidden state (lstm/lstm.py:150-151)
        from trajnetplusplusbaselines_amd.lstm import GridBasedPooling
        pool = GridBasedPooling(type_='social', hidden_dim=128, cell_side=0.6, n=8, out_dim=128,
                                embedding_arch='two_layer', layer_dims=[128], latent_dim=8)
    model = LSTM(pool=pool, pool_to_input=(kind != 'addhidden'))
    pre = kind + '_sd_'
    model.load_state_dict({k[len(pre):]: torch.tensor(G[k]) for k in G.files if k.startswith(pre)})
    model = model.cuda().train()
    xy, split = torch.tensor(G[kind + '_xy']), torch.tensor(G[kind + '_split'])
    M = xy.shape[1]
    targets = (xy[9:21] - xy[8:20]).cuda()
    rel, pred = model(xy[:9].clone(), torch.zeros(M, 2), split, xy[9:20].clone())
    loss = PredictionLoss()(rel[-12:], targets, split) * 8 + 0.1 * torch.nan_to_num(pred[-12:, split[:-1].cuda()]).pow(2).mean()
    np.testing.assert_allclose(float(loss.detach()), float(G[kind + '_loss']), rtol=2e-5)
    loss.backward()
    worst = 0.0
    for name, p in model.named_parameters():
        want = G[kind + '_grad_' + name]
        if p.grad is None:
            assert not np.any(want), name
            continue
        scale = max(1e-6, float(np.abs(want).max()))
        err = float(np.abs(p.grad.cpu().numpy() - want).max()) / scale
        worst = max(worst, err)
        assert err < 2e-3, '%s: relative error %.2e (scale %.2e)' % (name, err, scale)
    print(kind, 'worst relative gradient error %.2e' % worst)


def test_two_frame_free_running_gradients_match_reference():
    """LSTM.forward with two observed frames (pre-seeded positions, d_pred offset) and n_predict decoding (every track
    fed its own detached prediction) in training mode: outputs and gradients against the reference (grad_short.npz)."""
    from trajnetplusplusbaselines_amd.lstm import LSTM, GridBasedPooling
    G = np.load(os.path.join(helpers.GOLDEN, 'grad_short.npz'))
    pool = GridBasedPooling(type_='social', hidden_dim=128, cell_side=0.6, n=8, out_dim=64,
                            embedding_arch='two_layer', layer_dims=[128], latent_dim=8)
    model = LSTM(pool=pool)
    model.load_state_dict({k[3:]: torch.tensor(G[k]) for k in G.files if k.startswith('sd_')})
    model = model.cuda().train()
    xy, split = torch.tensor(G['xy']), torch.tensor(G['split'])
    M = xy.shape[1]
    rel, pred = model(xy[:2].clone(), torch.zeros(M, 2), split, n_predict=6)
    helpers.assert_close_nan(rel.detach().cpu().numpy(), G['rel'], 3e-5, 'rel')
    helpers.assert_close_nan(pred.detach().cpu().numpy(), G['pred'], 3e-5, 'pred')
    prim = split[:-1].cuda()
    loss = rel[-6:, prim, :2].pow(2).sum() + 0.3 * torch.nan_to_num(pred[-6:, prim]).pow(2).mean() + rel[-6:, prim, 2:].sum()
    np.testing.assert_allclose(float(loss.detach()), float(G['loss']), rtol=3e-5)
    loss.backward()
    for name, p in model.named_parameters():
        want = G['grad_' + name]
        if p.grad is None:
            assert not np.any(want), name
            continue
        scale = max(1e-6, float(np.abs(want).max()))
        err = float(np.abs(p.grad.cpu().numpy() - want).max()) / scale
        assert err < 2e-3, '%s: relative error %.2e (scale %.2e)' % (name, err, scale)


def test_training_step_is_bitwise_reproducible():
    """Every reduction of the training path has a fixed order (split-K partial tiles, per-cell hit lists, gathers per
    track, LDS integer max for the winners): two runs from the same weights give bit-identical gradients."""
    from trajnetplusplusbaselines_amd.lstm import PredictionLoss
    grads = []
    for _ in range(2):
        model = build('social')
        xy, split = torch.tensor(GOLD['social_xy']), torch.tensor(GOLD['social_split'])
        M = xy.shape[1]
        targets = (xy[9:21] - xy[8:20]).cuda()
        rel, pred = model(xy[:9].clone(), torch.zeros(M, 2), split, xy[9:20].clone())
        (PredictionLoss()(rel[-12:], targets, split) * 8).backward()
        grads.append({n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None})
    assert grads[0].keys() == grads[1].keys()
    for n in grads[0]:
        assert torch.equal(grads[0][n], grads[1][n]), n


def test_step_train_saves_equal_the_sequence_drivers_first_step():
    """tnp_lstm_step_train (one step with saves, the per-step form of the training forward) leaves exactly what
    tnp_lstm_forward_train leaves for the first step of the sequence: state, LSTMCell input, activations, gates,
    encodings, winner table -- bit for bit."""
    import ctypes
    from trajnetplusplusbaselines_amd import _lib
    from trajnetplusplusbaselines_amd.lstm import training
    model = build('social').eval()
    xy, split = torch.tensor(GOLD['social_xy']).cuda(), torch.tensor(GOLD['social_split'])
    M, H = xy.shape[1], model.hidden_dim
    L = _lib.lib()
    idx = _lib.SceneIndex.get(split, xy.device)
    m, keep, dev = model._descriptor()
    ws, need = model._workspace(m, M, idx.B, dev)
    pool = model.pool
    I = model.encoder.weight_ih.shape[1]
    N1, ncell, C = pool.embedding_layers()[0].weight.shape[0], pool.n * pool.n, pool.pooling_dim
    assert L.tnp_lstm_sparse_first_layer(ctypes.byref(m), M) == 1
    # --- sequence driver: two observed frames = one encoder step
    S = 1
    bufs = dict(h=torch.empty(S + 1, M, H), c=torch.empty(S + 1, M, H), X=torch.empty(S, M, I), g=torch.empty(S, M, 4 * H),
                a=torch.empty(S, M, N1), e=torch.empty(S, M, C), o1=torch.empty(S, M, 2), o2=torch.empty(S, M, 2))
    bufs = {k: v.cuda() for k, v in bufs.items()}
    win = torch.empty(S, M, ncell, dtype=torch.int16, device='cuda')
    sv = training.TrainSaves()
    sv.h_all, sv.c_all, sv.X_all, sv.gates_all = (bufs[k].data_ptr() for k in ('h', 'c', 'X', 'g'))
    sv.act_all[0], sv.enc_all, sv.winners_all = bufs['a'].data_ptr(), bufs['e'].data_ptr(), win.data_ptr()
    sv.obs1_all, sv.obs2_all = bufs['o1'].data_ptr(), bufs['o2'].data_ptr()
    rel, pos = torch.empty(S, M, 5, device='cuda'), torch.empty(S + 1, M, 2, device='cuda')
    ex = _lib.LstmExtras()
    obs = xy[:2].contiguous()
    _lib.check(L.tnp_lstm_forward_train(ctypes.byref(m), _lib.ptr(obs), 2, M, None, _lib.ptr(idx.starts), _lib.ptr(idx.primary),
                                        idx.B, idx.n_max, None, None, 0, _lib.ptr(rel), _lib.ptr(pos), _lib.ptr(ws), need,
                                        ctypes.byref(ex), ctypes.byref(sv), _lib.stream_ptr()), 'forward_train')
    # --- the same step through the per-step entry point
    h0, c0 = torch.zeros(M, H, device='cuda'), torch.zeros(M, H, device='cuda')
    h1, c1, nrm = torch.empty(M, H, device='cuda'), torch.empty(M, H, device='cuda'), torch.empty(M, 5, device='cuda')
    X1, g1, a1, e1 = (torch.empty_like(bufs[k][0]) for k in ('X', 'g', 'a', 'e'))
    w1 = torch.empty(M, ncell, dtype=torch.int16, device='cuda')
    ss = training.StepSaves()
    ss.X, ss.gates, ss.enc, ss.winners = X1.data_ptr(), g1.data_ptr(), e1.data_ptr(), w1.data_ptr()
    ss.act[0] = a1.data_ptr()
    _lib.check(L.tnp_lstm_step_train(ctypes.byref(m), 0, _lib.ptr(h0), _lib.ptr(c0), _lib.ptr(obs[0]), _lib.ptr(obs[1]), None,
                                     _lib.ptr(idx.starts), idx.B, M, idx.n_max, None, _lib.ptr(h1), _lib.ptr(c1), _lib.ptr(nrm),
                                     ctypes.byref(ss), _lib.ptr(ws), need, _lib.stream_ptr()), 'step_train')
    torch.cuda.synchronize()
    nan_eq = lambda a, b: torch.equal(torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(b, nan=-7.0))
    assert nan_eq(h1, bufs['h'][1]) and nan_eq(c1, bufs['c'][1]) and nan_eq(nrm, rel[0])
    present = ~torch.isnan(obs[0, :, 0]) & ~torch.isnan(obs[1, :, 0])
    assert torch.equal(X1[present], bufs['X'][0][present]) and torch.equal(g1[present], bufs['g'][0][present])
    assert torch.equal(a1, bufs['a'][0]) and torch.equal(e1, bufs['e'][0]) and torch.equal(w1, win[0])


@pytest.mark.parametrize('kind', ['nll', 'l2'])
def test_train_batch_with_collision_loss_matches_the_reference_trainer(kind):
    """train_step.train_batch against the reference's own Trainer.train_batch (lstm/trainer.py:229-269, instantiated
    unmodified by oracle/gen_golden_r2.py) with the auxiliary collision loss on (col_wt = 10): the loss of four Adam steps
    (each step starts from the previous update, so the later values pin the gradients) and, for the NLL run, every parameter
    gradient of the first step -- the collision term is most of that loss, and it back-propagates through the primaries'
    predicted positions (tnp_collision_loss_backward -> d_pred of the backward sweep)."""
    from trajnetplusplusbaselines_amd.lstm import LSTM, GridBasedPooling, PredictionLoss, L2Loss
    from trajnetplusplusbaselines_amd.lstm.train_step import train_batch
    z = np.load(os.path.join(helpers.GOLDEN, 'trainer_case.npz'))
    pre = kind + '_'
    pool = GridBasedPooling(type_='social', hidden_dim=128, cell_side=0.6, n=8, out_dim=64,
                            embedding_arch='two_layer', layer_dims=[128], latent_dim=8)
    model = LSTM(pool=pool)
    model.load_state_dict({k[len(pre) + 3:]: torch.tensor(z[k]) for k in z.files if k.startswith(pre + 'sd_')})
    model = model.cuda()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
    crit = (PredictionLoss if kind == 'nll' else L2Loss)(col_wt=10.0, col_distance=1.5)
    batches = [(torch.tensor(z[pre + 'b%d_xy' % i]), torch.tensor(z[pre + 'b%d_split' % i])) for i in range(2)]
    assert z[pre + 'losses'][0] > 2 * z[pre + 'first_loss_without_collision']      # the fixture's collisions matter
    losses = []
    for it in range(4):
        xy, split = batches[it % 2]
        losses.append(train_batch(model, opt, crit, xy, torch.zeros(xy.shape[1], 2), split, 9, 12, batch_size=6))
        if it == 0 and kind == 'nll':
            worst = 0.0
            for k, p in model.named_parameters():
                want = z[pre + 'grad_' + k]
                got = p.grad.cpu().numpy() if p.grad is not None else np.zeros_like(want)
                worst = max(worst, float(np.abs(got - want).max() / max(np.abs(want).max(), 1e-3)))
            assert worst < 5e-5, worst
    np.testing.assert_allclose(losses, z[pre + 'losses'], rtol=1e-4)


def test_scene_batcher_on_the_device_matches_the_trainers_host_assembly():
    """data.SceneBatcher (scenes tensorised once, a batch = a column gather, per-scene rotation on the
    device) against the reference trainer's per-scene host arithmetic: drop_distant (lstm/lstm.py:16-22), center_scene
    (lstm/utils.py:32-51), random_rotation (lstm/utils.py:10-17, one angle per scene from Python's `random`), and the
    batch assembly of lstm/trainer.py:120-131 -- then one optimisation step straight from the batcher's device tensors."""
    import random
    from trajnetplusplusbaselines_amd import data, synth
    from trajnetplusplusbaselines_amd.lstm import LSTM, GridBasedPooling, PredictionLoss, drop_distant
    from trajnetplusplusbaselines_amd.lstm.train_step import train_batch
    xy, split = synth.ragged_crowd(9, 2, 11, seed=23)
    xy = xy.numpy().astype(np.float64) * 1.6                      # spread out: drop_distant removes some neighbours
    scenes = [xy[:, split[s]:split[s + 1]] for s in range(9)]
    for normalize in (False, True):
        batcher = data.SceneBatcher(scenes, device='cuda', obs_length=9, normalize_scene=normalize)
        ids = [4, 0, 7, 2]
        random.seed(99)
        got_xy, got_goals, got_split = batcher.batch(ids, augment=True)
        random.seed(99)
        want, want_goals = [], []
        for i in ids:                                              # what lstm/trainer.py:96-133 does per scene, on the host
            sc, _ = drop_distant(scenes[i])
            g = np.zeros((sc.shape[1], 2))
            if normalize:
                sc, _, _, g = data.center_scene(sc, 9, goals=g)
            sc, g = data.random_rotation(sc, g)
            want.append(sc)
            want_goals.append(g)
        want_xy, want_split = data.batch_scenes(want)
        assert got_xy.is_cuda and got_split.tolist() == want_split.tolist()
        helpers.assert_close_nan(got_xy.cpu().numpy(), want_xy, 2e-5, 'batched scenes')
        helpers.assert_close_nan(got_goals.cpu().numpy(), np.concatenate(want_goals, axis=0), 2e-5, 'batched goals')
    assert (np.diff(want_split) < np.array([scenes[i].shape[1] for i in ids])).any()    # drop_distant did remove tracks
    pool = GridBasedPooling(type_='social', hidden_dim=128, cell_side=0.6, n=8, out_dim=64, embedding_arch='two_layer',
                            layer_dims=[128], latent_dim=8)
    torch.manual_seed(3)
    model = LSTM(pool=pool).cuda()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    losses = [train_batch(model, opt, PredictionLoss(), got_xy, got_goals, got_split, 9, 12) for _ in range(3)]
    assert np.isfinite(losses).all() and losses[-1] < losses[0]


class _Doubler(object):
    """stands in for parallel.GradReducer on one GPU: an "all-reduce" over two identical ranks = multiply by two in place"""

    def __init__(self):
        self.seen = []

    def __call__(self, t):
        assert t.is_contiguous()
        self.seen.append(t.numel())
        t.mul_(2.0)

        class _Work(object):
            def wait(self_inner):
                return True
        return _Work()


@pytest.mark.parametrize('kind', ['social', 'directional', 'nn', 'attentionmlp', 'hiddenstatemlp'])
def test_every_gradient_goes_through_the_in_backward_reducer_exactly_once(kind):
    """Data-parallel training all-reduces the gradients from inside the backward pass (lstm/training.py `publish`,
    parallel.GradReducer).  With a stand-in reducer that doubles a tensor in place, every parameter gradient must come out
    exactly twice the plain gradient -- none skipped, none reduced twice (re-laid-out / sliced / cloned gradients included),
    and the largest message (the sparse first layer's) goes first."""
    from trajnetplusplusbaselines_amd import synth
    from trajnetplusplusbaselines_amd.lstm import LSTM, GridBasedPooling, PredictionLoss
    from trajnetplusplusbaselines_amd.lstm.non_gridbased_pooling import NearestNeighborMLP, AttentionMLPPooling, HiddenStateMLPPooling
    torch.manual_seed(11)
    if kind == 'social':
        pool = GridBasedPooling(type_='social', hidden_dim=128, cell_side=0.6, n=8, out_dim=64, embedding_arch='two_layer',
                                layer_dims=[128], latent_dim=8)
    elif kind == 'directional':
        pool = GridBasedPooling(type_='directional', hidden_dim=128, cell_side=0.6, n=12, out_dim=64)
    elif kind == 'nn':
        pool = NearestNeighborMLP(n=4, out_dim=32)
    elif kind == 'attentionmlp':
        pool = AttentionMLPPooling(hidden_dim=128, out_dim=64)
    else:
        pool = HiddenStateMLPPooling(hidden_dim=128, out_dim=64)
    model = LSTM(pool=pool).cuda().train()
    xy, split = synth.ragged_crowd(5, 2, 9, seed=31)
    M = xy.shape[1]
    targets = (xy[9:21] - xy[8:20]).cuda()

    def grads(reducer):
        model.zero_grad()
        model._grad_reduce_fn = reducer
        rel, pred = model(xy[:9], torch.zeros(M, 2), split, xy[9:20].clone())
        model._grad_reduce_fn = None
        (PredictionLoss()(rel[-12:], targets, split) * 5 + 0.1 * torch.nan_to_num(pred[-12:, split[:-1]]).pow(2).mean()).backward()
        return {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    plain = grads(None)
    red = _Doubler()
    doubled = grads(red)
    assert set(plain) == set(doubled) and len(plain) >= 10
    for n in plain:
        assert torch.equal(doubled[n], plain[n] * 2.0), n
    if kind == 'social':
        assert red.seen[0] == max(red.seen)            # the sparse first-layer gradient is published first


def test_grouped_wgrad_is_bitwise_the_stand_alone_wgrad():
    """tnp_wgrad_grouped (all contractions of a step in one launch + one reduce) keeps every problem's split / summation plan:
    its results equal the stand-alone tnp_wgrad calls bit for bit, bias column sums included, for shapes like the step's."""
    import ctypes
    from trajnetplusplusbaselines_amd import _lib
    from trajnetplusplusbaselines_amd.lstm.training import WgradProblem
    L = _lib.lib()
    g = torch.Generator().manual_seed(5)
    shapes = [(4096, 256, 1024, True), (3000, 512, 320, True), (3000, 512, 128, False), (4096, 5, 128, True), (4096, 62, 2, True),
              (4096, 16, 128, True), (777, 33, 70, False)]
    probs, keep, singles = [], [], []
    for K, Mo, No, bias in shapes:
        dy = torch.randn(K, Mo, generator=g).cuda()
        x = torch.randn(K, No, generator=g).cuda()
        dw, db = torch.empty(Mo, No, device='cuda'), (torch.empty(Mo, device='cuda') if bias else None)
        dw1, db1 = torch.empty(Mo, No, device='cuda'), (torch.empty(Mo, device='cuda') if bias else None)
        nb = L.tnp_wgrad_workspace_bytes(Mo, No, K)
        ws = torch.empty(nb, dtype=torch.uint8, device='cuda')
        _lib.check(L.tnp_wgrad(_lib.ptr(dy), Mo, _lib.ptr(x), No, K, Mo, No, _lib.ptr(dw1), No, _lib.ptr(db1), _lib.ptr(ws), nb,
                               _lib.stream_ptr()), 'tnp_wgrad')
        probs.append(WgradProblem(dy.data_ptr(), Mo, x.data_ptr(), No, K, Mo, No, dw.data_ptr(), No, db.data_ptr() if bias else None))
        keep.append((dy, x, dw, db))
        singles.append((dw1, db1))
    table = (WgradProblem * len(probs))(*probs)
    nb = L.tnp_wgrad_grouped_workspace_bytes(table, len(probs))
    ws = torch.empty(nb, dtype=torch.uint8, device='cuda')
    _lib.check(L.tnp_wgrad_grouped(table, len(probs), _lib.ptr(ws), nb, _lib.stream_ptr()), 'tnp_wgrad_grouped')
    torch.cuda.synchronize()
    for (dy, x, dw, db), (dw1, db1) in zip(keep, singles):
        assert torch.equal(dw, dw1)
        if db is not None:
            assert torch.equal(db, db1)
        ref = dy.double().t() @ x.double()
        assert (dw.double() - ref).abs().max() <= 1e-3 * max(1.0, float(ref.abs().max()))


def test_grouped_wgrad_column_sums_without_a_weight_part():
    """A problem with No == 0 and a bias pointer is the column sums of dy alone (the sparse first layer's bias gradient rides in
    the grouped launch instead of an ATen sum behind the sweep): equal to the double-precision sums to rounding, the same bits
    twice, beside ordinary contractions and for ragged K / Mo."""
    from trajnetplusplusbaselines_amd import _lib
    from trajnetplusplusbaselines_amd.lstm.training import WgradProblem
    L = _lib.lib()
    g = torch.Generator().manual_seed(8)
    outs = []
    for _ in range(2):
        probs, keep = [], []
        g.manual_seed(8)
        for K, Mo, No in [(5890, 1024, 0), (777, 100, 0), (5, 130, 0), (3000, 512, 128), (1, 64, 0), (38912, 256, 0)]:
            dy = torch.randn(K, Mo, generator=g).cuda()
            x = torch.randn(K, No, generator=g).cuda() if No else None
            dw = torch.empty(Mo, No, device='cuda') if No else None
            db = torch.full((Mo,), float('nan'), device='cuda')
            probs.append(WgradProblem(dy.data_ptr(), Mo, x.data_ptr() if No else None, No, K, Mo, No, dw.data_ptr() if No else None, No,
                                      db.data_ptr()))
            keep.append((dy, x, dw, db))
        table = (WgradProblem * len(probs))(*probs)
        nb = L.tnp_wgrad_grouped_workspace_bytes(table, len(probs))
        assert nb > 0
        ws = torch.empty(nb, dtype=torch.uint8, device='cuda')
        _lib.check(L.tnp_wgrad_grouped(table, len(probs), _lib.ptr(ws), nb, _lib.stream_ptr()), 'tnp_wgrad_grouped')
        torch.cuda.synchronize()
        outs.append([k[3].clone() for k in keep])
        for dy, x, dw, db in keep:
            ref = dy.double().sum(0)
            assert (db.double() - ref).abs().max() <= 2e-5 * max(1.0, float(dy.abs().sum(0).max()))
            if dw is not None:
                refw = dy.double().t() @ x.double()
                assert (dw.double() - refw).abs().max() <= 1e-3 * max(1.0, float(refw.abs().max()))
    assert all(torch.equal(a, b) for a, b in zip(*outs))
    # rows with a leading dimension larger than Mo (a column slice of a wider matrix)
    wide = torch.randn(1000, 300, generator=g).cuda()
    view = wide[:, 17:17 + 200]
    db = torch.empty(200, device='cuda')
    one = (WgradProblem * 1)(WgradProblem(view.data_ptr(), view.stride(0), None, 0, 1000, 200, 0, None, 0, db.data_ptr()))
    nb = L.tnp_wgrad_grouped_workspace_bytes(one, 1)
    ws = torch.empty(nb, dtype=torch.uint8, device='cuda')
    _lib.check(L.tnp_wgrad_grouped(one, 1, _lib.ptr(ws), nb, _lib.stream_ptr()), 'tnp_wgrad_grouped')
    assert (db.double() - view.double().sum(0)).abs().max() <= 2e-5 * float(view.abs().sum(0).max())
    # a problem with neither a weight part nor a bias is skipped, as before
    empty = (WgradProblem * 1)(WgradProblem(keep[0][0].data_ptr(), 1024, None, 0, 5890, 1024, 0, None, 0, None))
    assert L.tnp_wgrad_grouped_workspace_bytes(empty, 1) == 0


def test_scaled_diff_is_nan_to_num_times_scale():
    from trajnetplusplusbaselines_amd import _lib
    g = torch.Generator().manual_seed(1)
    a, b = torch.randn(19, 333, 2, generator=g).cuda(), torch.randn(19, 333, 2, generator=g).cuda()
    a[0, :7] = float('nan'); b[3, 5] = float('nan'); a[4, 0, 0] = float('inf'); b[5, 1, 1] = float('inf'); a[6, 2] = 3e38; b[6, 2] = -3e38
    out = torch.empty_like(a)
    _lib.check(_lib.lib().tnp_scaled_diff(_lib.ptr(a), _lib.ptr(b), a.numel(), 4.0, _lib.ptr(out), _lib.stream_ptr()), 'scaled_diff')
    assert torch.equal(out, torch.nan_to_num(a - b) * 4.0)


def test_sharded_loss_with_the_real_collision_term_sums_to_the_single_process_gradient():
    """Scene-sharded training with PredictionLoss(col_wt > 0): the NLL term is a mean over frames x scenes, the
    collision term a sum over scenes (lstm/loss.py:85-91, :147-161); `train_step.batch_loss` scales them separately.  On one
    GPU: the gradients of two shards (each with the batch-wide slot count), added up as a SUM all-reduce would, equal the
    gradients of the whole batch; with the round-2 rule (everything x n_local / n_global) they do not."""
    from trajnetplusplusbaselines_amd import parallel, synth
    from trajnetplusplusbaselines_amd.lstm import LSTM, GridBasedPooling, PredictionLoss
    from trajnetplusplusbaselines_amd.lstm.train_step import batch_loss
    torch.manual_seed(3)
    pool = GridBasedPooling(type_='social', hidden_dim=128, cell_side=0.6, n=8, out_dim=64, embedding_arch='two_layer',
                            layer_dims=[128], latent_dim=8)
    model = LSTM(pool=pool).cuda().train()
    xy, split = synth.ragged_crowd(6, 3, 9, seed=41, nan_frac=0.0)
    xy = xy * 0.35                                                     # a dense crowd: primaries come within col_distance of neighbours
    crit = PredictionLoss(col_wt=10.0, col_distance=1.0)
    goals = torch.zeros(xy.shape[1], 2)

    def grads_of(batch, sp, shard, pad_to):
        model.zero_grad()
        scene = batch.cuda()
        rel, out = model(scene[:9].clone(), goals[:scene.shape[1]], sp, scene[9:20].clone(), pad_to=pad_to)
        targets = scene[9:21] - scene[8:20]
        loss = batch_loss(crit, rel, out, scene, targets, sp, 12, 6, shard=shard)
        loss.backward()
        return float(loss.detach()), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    full_loss, full = grads_of(xy, split, None, None)
    # the collision term is live and sizeable in this batch
    with torch.no_grad():
        model.eval()
        rel, out = model(xy[:9].cuda(), goals, split, xy[9:20].cuda().clone())
        model.train()
    total, parts = 0.0, []
    for r in range(2):
        sh = parallel.shard_batch(xy, goals, split, r, 2)
        lo, hi = sh.track_range
        l, g = grads_of(xy[:, lo:hi], sh.batch_split, (sh.n_scenes, sh.n_scenes_global), sh.pad_to)
        total += l
        parts.append(g)
    assert abs(total - full_loss) <= 2e-4 * max(1.0, abs(full_loss)), (total, full_loss)
    worst = 0.0
    for n, g in full.items():
        s = parts[0][n] + parts[1][n]
        worst = max(worst, float((s - g).abs().max()) / max(1e-6, float(g.abs().max())))
    assert worst < 2e-4, worst
    # the collision term really contributes: without it the loss is different
    crit0 = PredictionLoss()
    model.zero_grad()
    scene = xy.cuda()
    rel, out = model(scene[:9].clone(), goals, split, scene[9:20].clone())
    l0 = float(batch_loss(crit0, rel, out, scene, scene[9:21] - scene[8:20], split, 12, 6))
    assert abs(full_loss - l0) > 0.05 * abs(l0) + 1.0, (full_loss, l0)


def test_loss_read_back_is_the_value_at_record_time():
    """train_step._LossReadBack: the value travels to the host when it is recorded -- later in-place changes of the tensor
    (or any amount of work queued behind it) do not change what value() returns"""
    from trajnetplusplusbaselines_amd.lstm.train_step import _LossReadBack
    t = torch.full((), 3.25, device='cuda')
    rb = _LossReadBack(t)
    big = torch.randn(4096, 4096, device='cuda')
    for _ in range(4):
        big = big @ big.t() * 1e-4
    t.add_(1.0)
    assert rb.value() == 3.25 and float(t) == 4.25


# ---- the training path at BASELINE config 2's FULL size, against the reference's own autograd (train_full.npz) ----------------

def _train_full():
    return np.load(os.path.join(helpers.GOLDEN, 'train_full.npz'))


def _full_size_model():
    """helpers.real_model: the headline Social-LSTM (n=16, two_layer 1024, latent_dim 16) under the fixture's seed; the
    per-tensor weight sums stored by the reference run are checked."""
    model, _ = helpers.real_model('cuda')
    z = _train_full()
    for k, v in model.state_dict().items():
        assert abs(v.double().sum().item() - float(z['wsum_' + k])) < 1e-9, 'seeded weight differs from the reference run: ' + k
    return model.train(), z


def _full_batch(z, tag):
    if tag == 'synth':
        from trajnetplusplusbaselines_amd import synth
        return synth.linear_crowd(64, 32, seed=int(z['synth_seed']))
    r = np.load(os.path.join(helpers.GOLDEN, 'real_cases.npz'))
    return torch.tensor(r[tag + '_raw_xy'], dtype=torch.float32), torch.tensor(r[tag + '_split'])


@pytest.mark.parametrize('tag', ['synth', 'hotel', 'students'])
def test_full_size_gradients_match_reference_autograd(tag):
    """Trainer.train_batch's loss (lstm/trainer.py:252-265) back-propagated through the HEADLINE model -- layer_dims=[1024],
    C=16, n=16: the configuration whose backward runs dgrid_cells_xcd_kernel<8> (the N1 == 1024 specialisation),
    sparse_wgrad_mfma_kernel<16> and the grouped weight-gradient launch at K = 19 x 2048 -- on the bench's 64 x 32 batch and
    on the real scenes of real_cases.npz (ragged, up to 68 agents, NaN tracks), against the reference's autograd:
    loss 2e-5, primaries' outputs 5e-5, every parameter gradient within 1e-4 of its largest magnitude (measured ~1e-6),
    the 16.8 MB first-layer gradient through its sketch (row / column sums, seeded projections, 16 k sampled entries)."""
    from trajnetplusplusbaselines_amd.lstm import PredictionLoss
    model, z = _full_size_model()
    xy, split = _full_batch(z, tag)
    M = xy.shape[1]
    observed, truth = xy[:9].clone(), xy[9:20].clone()
    targets = (xy[9:21] - xy[8:20]).cuda()
    rel, pred = model(observed, torch.zeros(M, 2), split, truth)
    loss = PredictionLoss()(rel[-12:], targets, split) * (split.numel() - 1)
    np.testing.assert_allclose(float(loss), float(z[tag + '_loss']), rtol=2e-5)
    prim = split[:-1].cuda()
    helpers.assert_close_nan(rel.detach()[:, prim].cpu().numpy(), z[tag + '_rel_prim'], 5e-5, 'rel (primaries)')
    helpers.assert_close_nan(pred.detach()[:, prim].cpu().numpy(), z[tag + '_pred_prim'], 5e-5, 'pred (primaries)')
    loss.backward()
    worst = 0.0
    for name, p in model.named_parameters():
        if (tag + '_nograd_' + name) in z.files:
            assert p.grad is None, name + ': the reference leaves this gradient None'
            continue
        assert p.grad is not None, name
        worst = max(worst, helpers.assert_matches_stored(z, tag + '_grad_' + name, p.grad.cpu().numpy(), 1e-4, tag))
    print(tag, 'worst relative gradient error %.2e' % worst)


@pytest.mark.parametrize('tag', ['social', 'directional'])
def test_gradients_with_scenes_larger_than_a_wavefront_match_reference_autograd(tag):
    """Scenes of 97..117 agents (tests/golden/big_scene_grads.npz: synth.ragged_crowd(4, 70, 120, seed=31), dense -- dozens of
    neighbours per cell, clobbered corner cells, entering / leaving tracks), more slots per scene than a wavefront has lanes:
    the multi-pass vote / hit-list / pair kernels of the backward sweep, through the headline Social-LSTM (sparse first layer,
    N1 == 1024 kernels) and the config-3 D-LSTM.  Trainer loss against the reference's autograd: loss 2e-5, primaries 5e-5,
    every gradient within 1e-4 of its largest magnitude."""
    from trajnetplusplusbaselines_amd import synth
    from trajnetplusplusbaselines_amd.lstm import LSTM, GridBasedPooling, PredictionLoss
    z = np.load(os.path.join(helpers.GOLDEN, 'big_scene_grads.npz'))
    if tag == 'social':
        model, _ = helpers.real_model()
    else:
        torch.manual_seed(int(z[tag + '_seed']))
        model = LSTM(pool=GridBasedPooling(type_='directional', hidden_dim=128, cell_side=0.6, n=12, out_dim=256, embedding_arch='one_layer'))
    for k, v in model.state_dict().items():
        assert abs(v.double().sum().item() - float(z[tag + '_wsum_' + k])) < 1e-9, 'seeded weight differs from the reference run: ' + k
    model = model.cuda().train()
    xy, split = synth.ragged_crowd(4, 70, 120, seed=int(z['crowd_seed']))
    M = xy.shape[1]
    assert M == int(z['tracks']) and int(split.diff().max()) > 64
    targets = (xy[9:21] - xy[8:20]).cuda()
    rel, pred = model(xy[:9].clone(), torch.zeros(M, 2), split, xy[9:20].clone())
    loss = PredictionLoss()(rel[-12:], targets, split) * (split.numel() - 1)
    np.testing.assert_allclose(float(loss), float(z[tag + '_loss']), rtol=2e-5)
    prim = split[:-1].cuda()
    helpers.assert_close_nan(rel.detach()[:, prim].cpu().numpy(), z[tag + '_rel_prim'], 5e-5, 'rel (primaries)')
    helpers.assert_close_nan(pred.detach()[:, prim].cpu().numpy(), z[tag + '_pred_prim'], 5e-5, 'pred (primaries)')
    loss.backward()
    worst = 0.0
    for name, p in model.named_parameters():
        if (tag + '_nograd_' + name) in z.files:
            assert p.grad is None, name + ': the reference leaves this gradient None'
            continue
        assert p.grad is not None, name
        worst = max(worst, helpers.assert_matches_stored(z, tag + '_grad_' + name, p.grad.cpu().numpy(), 1e-4, tag))
    print(tag, 'worst relative gradient error %.2e' % worst)


def test_full_size_training_curve_matches_reference():
    """Four optimisation steps of the headline model (train_step.train_batch == Trainer.train_batch; Adam lr 1e-3, weight_decay
    1e-4 as lstm/trainer.py:497), alternating the 64 x 32 synthetic batch and the hotel scenes: the reference's loss trajectory
    (149.6 -> 32.1 -> 118.6 -> 16.4) within 1e-4 relative, trained weights within half an Adam step."""
    from trajnetplusplusbaselines_amd.lstm import PredictionLoss
    from trajnetplusplusbaselines_amd.lstm.train_step import train_batch
    for make_opt in (lambda ps: torch.optim.Adam(ps, lr=1e-3, weight_decay=1e-4), None):
        model, z = _full_size_model()
        if make_opt is None:
            from trajnetplusplusbaselines_amd import optim
            opt = optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)      # the one-launch update bench.py times
        else:
            opt = make_opt(model.parameters())
        batches = [_full_batch(z, 'synth'), _full_batch(z, 'hotel')]
        losses = []
        for it in range(4):
            xy, split = batches[it % 2]
            losses.append(train_batch(model, opt, PredictionLoss(), xy, torch.zeros(xy.shape[1], 2), split, 9, 12))
        np.testing.assert_allclose(losses, z['curve_losses'], rtol=1e-4)
        for k, v in model.state_dict().items():
            got = v.cpu().numpy()
            if ('curve_final_' + k) in z.files:
                assert np.abs(got - z['curve_final_' + k]).max() < 5e-4, k
            else:       # sketched: sampled entries within half a step, sums within half a step times the number of terms' root
                s = helpers.sketch(got)
                assert np.abs(s['samples'] - z['curve_final_' + k + '@samples']).max() < 5e-4, k
                assert np.abs(s['rowsum'] - z['curve_final_' + k + '@rowsum']).max() < 5e-4 * np.sqrt(got.size / got.shape[0]) * 4, k
        print('losses', losses, 'reference', z['curve_losses'].tolist())


@pytest.mark.parametrize('kind', ['occupancy', 'social3_front', 'social_const', 'directional_goals', 'lstm_layer'])
def test_gradient_variants_match_reference_autograd(kind):
    """Training-path options the earlier gradient fixtures did not exercise, on a DENSE ragged crowd (duplicates, clobbered
    corner cells): occupancy grid, social three_layer with front=True, social with constant = 0.5 (the clobbered cell (0, 0)
    then holds 0.5 and DOES pass gradient to the neighbours standing in it), directional with goals, embedding_arch =
    'lstm_layer' -- against the reference's autograd (tests/golden/grad_variants.npz, oracle/gen_golden_r4.py; weights =
    default init under the stored seed, per-tensor sums checked)."""
    from trajnetplusplusbaselines_amd.lstm import LSTM, GridBasedPooling, PredictionLoss
    z = np.load(os.path.join(helpers.GOLDEN, 'grad_variants.npz'))
    cfgs = {
        'occupancy': (dict(type_='occupancy', n=8, out_dim=32, embedding_arch='one_layer'), False),
        'social3_front': (dict(type_='social', n=8, out_dim=64, embedding_arch='three_layer', layer_dims=[64, 48], latent_dim=8, front=True), False),
        'social_const': (dict(type_='social', n=8, out_dim=64, embedding_arch='two_layer', layer_dims=[128], latent_dim=8, constant=0.5), False),
        'directional_goals': (dict(type_='directional', n=12, out_dim=64), True),
        'lstm_layer': (dict(type_='directional', n=12, out_dim=64, embedding_arch='lstm_layer'), False),
    }
    pool_kw, goal = cfgs[kind]
    pre = kind + '_'
    torch.manual_seed(int(z[pre + 'seed']))
    model = LSTM(pool=GridBasedPooling(hidden_dim=128, cell_side=0.6, **pool_kw), goal_flag=goal)
    for k, v in model.state_dict().items():
        assert abs(v.double().sum().item() - float(z[pre + 'wsum_' + k])) < 1e-9, 'seeded weight differs: ' + k
    model = model.cuda().train()
    xy, split, goals = torch.tensor(z[pre + 'xy']), torch.tensor(z[pre + 'split']), torch.tensor(z[pre + 'goals'])
    targets = (xy[9:21] - xy[8:20]).cuda()
    rel, pred = model(xy[:9].clone(), goals, split, xy[9:20].clone())
    loss = PredictionLoss()(rel[-12:], targets, split) * 5 + 0.1 * torch.nan_to_num(pred[-12:, split[:-1].cuda()]).pow(2).mean()
    np.testing.assert_allclose(float(loss.detach()), float(z[pre + 'loss']), rtol=1e-4, atol=2e-5)
    loss.backward()
    worst = 0.0
    for name, p in model.named_parameters():
        if (pre + 'nograd_' + name) in z.files:
            assert p.grad is None, name + ': the reference leaves this gradient None'
            continue
        assert p.grad is not None, name
        worst = max(worst, helpers.assert_matches_stored(z, pre + 'grad_' + name, p.grad.cpu().numpy(), 1e-4, kind))
    print(kind, 'worst relative gradient error %.2e' % worst)


@pytest.mark.parametrize('kind', ['social_h64_e32', 'directional_h256_e128', 'vanilla_h96_e64', 'occupancy_h32_e16'])
def test_gradients_with_other_state_sizes_match_reference_autograd(kind):
    """--hidden-dim / --coordinate-embedding-dim away from 128 / 64 (lstm/trainer.py:411-415) on the training path: hidden_dim
    32 / 64 / 96 / 256, embedding_dim 16 / 32 / 64 / 128 against the reference's autograd (tests/golden/grad_other_dims.npz,
    oracle/gen_golden_r4.py; default init under the stored seed, per-tensor sums checked)."""
    from trajnetplusplusbaselines_amd.lstm import LSTM, GridBasedPooling, PredictionLoss
    z = np.load(os.path.join(helpers.GOLDEN, 'grad_other_dims.npz'))
    cfgs = {
        'social_h64_e32': (64, 32, dict(type_='social', n=8, out_dim=32, embedding_arch='two_layer', layer_dims=[64], latent_dim=8)),
        'directional_h256_e128': (256, 128, dict(type_='directional', n=12, out_dim=64)),
        'vanilla_h96_e64': (96, 64, None),
        'occupancy_h32_e16': (32, 16, dict(type_='occupancy', n=8, out_dim=32, embedding_arch='one_layer')),
    }
    H, E, pool_kw = cfgs[kind]
    pre = kind + '_'
    torch.manual_seed(int(z[pre + 'seed']))
    pool = GridBasedPooling(hidden_dim=H, cell_side=0.6, **pool_kw) if pool_kw else None
    model = LSTM(embedding_dim=E, hidden_dim=H, pool=pool)
    for k, v in model.state_dict().items():
        assert abs(v.double().sum().item() - float(z[pre + 'wsum_' + k])) < 1e-9, 'seeded weight differs: ' + k
    model = model.cuda().train()
    xy, split = torch.tensor(z[pre + 'xy']), torch.tensor(z[pre + 'split'])
    targets = (xy[9:21] - xy[8:20]).cuda()
    rel, pred = model(xy[:9].clone(), torch.zeros(xy.shape[1], 2), split, xy[9:20].clone())
    loss = PredictionLoss()(rel[-12:], targets, split) * 5 + 0.1 * torch.nan_to_num(pred[-12:, split[:-1].cuda()]).pow(2).mean()
    np.testing.assert_allclose(float(loss.detach()), float(z[pre + 'loss']), rtol=1e-4, atol=2e-5)
    loss.backward()
    worst = 0.0
    for name, p in model.named_parameters():
        if (pre + 'nograd_' + name) in z.files:
            assert p.grad is None, name + ': the reference leaves this gradient None'
            continue
        assert p.grad is not None, name
        worst = max(worst, helpers.assert_matches_stored(z, pre + 'grad_' + name, p.grad.cpu().numpy(), 1e-4, kind))
    print(kind, 'worst relative gradient error %.2e' % worst)



def test_training_with_a_new_batch_split_every_step_does_not_depend_on_the_caches():
    """The reference trainer's operating point (lstm/trainer.py:30,125: batch_size 8, a new ragged batch_split every step;
    VERDICT r4 weak 5): every per-split cache -- _lib.SceneIndex._cache, the stacked row tables, the re-laid-out weight copies,
    the descriptor / parameter-list caches -- sees a new key at every step.  Twelve optimisation steps of the headline model on
    fresh SceneBatcher batches (rotation augmentation on) must leave bit-identical parameters whether the caches live across
    the steps or are thrown away in front of each one, and every loss must equal that of a cold model built from the same
    weights for that step alone."""
    import random
    from trajnetplusplusbaselines_amd import _lib, data as trajdata, synth
    from trajnetplusplusbaselines_amd.lstm import LSTM, GridBasedPooling, PredictionLoss
    from trajnetplusplusbaselines_amd.lstm.train_step import train_batch
    from trajnetplusplusbaselines_amd.optim import Adam
    xy, split = synth.ragged_crowd(40, 8, 72, seed=11, nan_frac=0.2)
    xy, split = xy.numpy(), split.numpy()
    scenes = [xy[:, split[i]:split[i + 1]] for i in range(len(split) - 1)]
    batcher = trajdata.SceneBatcher(scenes, device='cuda', drop_distant_r=None)

    def run(cold):
        torch.manual_seed(5)
        pool = GridBasedPooling(type_='social', hidden_dim=128, cell_side=0.6, n=16, out_dim=256, embedding_arch='two_layer',
                                layer_dims=[1024], latent_dim=16)
        model = LSTM(pool=pool).cuda().train()
        opt = Adam(model.parameters(), lr=1e-3)
        random.seed(3)
        rng = random.Random(9)
        losses, sizes = [], []
        for step in range(12):
            ids = [rng.randrange(len(scenes)) for _ in range(8)]
            bxy, bgoals, bsplit = batcher.batch(ids, augment=True)
            if cold:
                _lib.SceneIndex._cache.clear()
                for k in LSTM._CACHES:
                    if k != '_grad_reduce_fn':
                        model.__dict__[k] = None
            losses.append(train_batch(model, opt, PredictionLoss(), bxy, bgoals, bsplit, 9, 12, batch_size=8))
            sizes.append(int(bsplit[-1]))
        torch.cuda.synchronize()
        return losses, sizes, [p.detach().clone() for p in model.parameters()]

    warm_losses, sizes, warm = run(cold=False)
    cold_losses, _, cold = run(cold=True)
    assert len(set(sizes)) >= 10, 'the batches are meant to differ in size: %r' % (sizes,)
    assert all(np.isfinite(warm_losses)) and warm_losses == cold_losses
    for a, b in zip(warm, cold):
        assert torch.equal(a, b)
