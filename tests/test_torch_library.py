"""torch.library registration of the C-ABI entry points (trajnetplusplusbaselines_amd/ops.py): schemas and
fake (meta) implementations on CPU; the real kernels, ``opcheck`` and the autograd formula of ``trajnet::linear`` on a GPU."""
import numpy as np
import pytest
import torch

from trajnetplusplusbaselines_amd import ops  # noqa: F401  (registers the ops)

NAMES = ['pool_grid_winners', 'pool_grid', 'linear', 'pool_embed_sparse', 'constant_velocity', 'sf_rollout', 'orca_rollout',
         'kalman_predict', 'lstm_step', 'lstm_sequence']


def test_ops_are_registered_with_schemas():
    for name in NAMES:
        op = getattr(torch.ops.trajnet, name)
        schema = str(op.default._schema)
        assert schema.startswith('trajnet::' + name + '('), schema
    assert 'Tensor? bias' in str(torch.ops.trajnet.linear.default._schema)


def test_fake_implementations_propagate_shapes():
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        M, n, C, N1 = 37, 8, 16, 96
        obs = torch.empty(M, 2)
        starts = torch.empty(4, dtype=torch.int32)
        win = torch.ops.trajnet.pool_grid_winners(obs, obs, starts, 12, n, 0.6)
        assert win.shape == (M, n * n) and win.dtype == torch.int16
        assert torch.ops.trajnet.pool_grid('directional', obs, obs, None, starts, 12, n, 0.6, 0.0).shape == (M, 2 * n * n)
        enc = torch.empty(M, C)
        assert torch.ops.trajnet.pool_grid('social', obs, obs, enc, starts, 12, n, 0.6, 0.0).shape == (M, C * n * n)
        W = torch.empty(N1, C * n * n)
        y = torch.ops.trajnet.pool_embed_sparse(win, enc, starts, W, torch.empty(N1), True)
        assert y.shape == (M, N1) and y.dtype == torch.float32
        assert torch.ops.trajnet.linear(y, torch.empty(24, N1), None, False).shape == (M, 24)
        assert torch.ops.trajnet.constant_velocity(torch.empty(5, 2, dtype=torch.float64), torch.empty(5, 2, dtype=torch.float64), 12).shape == (12, 5, 2)
        assert torch.ops.trajnet.sf_rollout(torch.empty(9, 6, dtype=torch.float64), starts, 4, 12, 2.1, 0.3, 0.5).shape == (12, 9, 2)
        orc = torch.ops.trajnet.orca_rollout(torch.empty(9, 2), torch.empty(9, 2), torch.empty(9, 2, dtype=torch.float64),
                                             torch.empty(9, dtype=torch.float64), starts, 4, 12, 1.5, 1.5, 0.4)
        assert orc.shape == (12, 9, 2) and orc.dtype == torch.float32
        kal = torch.ops.trajnet.kalman_predict(torch.empty(7, 9, 2, dtype=torch.float64), torch.empty(7, 5, 13, 6, dtype=torch.float64), 10)
        assert kal.shape == (7, 13, 2) and kal.dtype == torch.float64
        h2, c2, nrm = torch.ops.trajnet.lstm_step(torch.empty(M, 128), torch.empty(M, 128), obs, obs, None,
                                                  torch.empty(4, dtype=torch.int64), True, 0, 0, [torch.empty(3)])
        assert h2.shape == (M, 128) and c2.shape == (M, 128) and nrm.shape == (M, 5)
        rel, pred = torch.ops.trajnet.lstm_sequence(torch.empty(9, M, 2), None, torch.empty(4, dtype=torch.int64), None, 11, 0, 0,
                                                    [torch.empty(3)])
        assert rel.shape == (19, M, 5) and pred.shape == (19, M, 2)
        rel, pred = torch.ops.trajnet.lstm_sequence(torch.empty(2, M, 2), None, torch.empty(4, dtype=torch.int64),
                                                    torch.empty(11, M, 2), 11, 0, 0, [torch.empty(3)])
        assert rel.shape == (12, M, 5) and pred.shape == (13, M, 2)        # T_obs == 2 pre-seeds `positions` (lstm/lstm.py:222-223)


def test_host_tensors_are_refused():
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        torch.ops.trajnet.linear(torch.zeros(4, 8), torch.zeros(3, 8), None, False)


@pytest.mark.gpu
def test_linear_op_matches_torch_and_has_gradients():
    torch.manual_seed(0)
    x = torch.randn(70, 96, device='cuda', requires_grad=True)
    w = torch.randn(40, 96, device='cuda', requires_grad=True)
    b = torch.randn(40, device='cuda', requires_grad=True)
    y = torch.ops.trajnet.linear(x, w, b, True)
    ref = torch.relu(x.double() @ w.double().t() + b.double())
    assert (y.double() - ref).abs().max() < 1e-4
    g = torch.randn_like(y)
    y.backward(g)
    gx, gw, gb = torch.autograd.grad(ref, (x, w, b), g.double())
    for got, want in ((x.grad, gx), (w.grad, gw), (b.grad, gb)):
        assert (got.double() - want).abs().max() < 1e-3 * max(1.0, float(want.abs().max()))
    torch.library.opcheck(torch.ops.trajnet.linear, (x.detach(), w.detach(), b.detach(), True),
                          test_utils=('test_schema', 'test_faketensor'))


@pytest.mark.gpu
def test_grid_and_sparse_ops_match_the_oracle():
    from oracle import oracle
    rng = np.random.RandomState(3)
    B, N, n, C, N1 = 5, 9, 8, 16, 64
    obs2 = (rng.rand(B, N, 2).astype(np.float32) * 6 - 3)
    obs1 = obs2 - np.float32(0.1)
    enc = rng.randn(B, N, C).astype(np.float32)
    W = (rng.randn(N1, C * n * n) / 20).astype(np.float32)
    b = rng.randn(N1).astype(np.float32)
    starts = torch.arange(0, B * N + 1, N, dtype=torch.int32)
    o1, o2 = torch.tensor(obs1.reshape(-1, 2)).cuda(), torch.tensor(obs2.reshape(-1, 2)).cuda()
    e = torch.tensor(enc.reshape(-1, C)).cuda()
    grid = torch.ops.trajnet.pool_grid('social', o1, o2, e, starts, N, n, 0.6, 0.0)
    want_grid = oracle.grid('social', obs1, obs2, enc, n=n, cell_side=0.6, C=C).reshape(B * N, -1)
    assert np.array_equal(grid.cpu().numpy(), want_grid)
    win = torch.ops.trajnet.pool_grid_winners(o1, o2, starts, N, n, 0.6)
    y = torch.ops.trajnet.pool_embed_sparse(win, e, starts, torch.tensor(W).cuda(), torch.tensor(b).cuda(), True)
    want = oracle.linear(want_grid, W, b, relu=True)
    assert np.abs(y.cpu().numpy() - want).max() < 2e-5 * max(1.0, float(np.abs(want).max()))
    torch.library.opcheck(torch.ops.trajnet.pool_grid_winners, (o1, o2, starts, N, n, 0.6), test_utils=('test_schema', 'test_faketensor'))


@pytest.mark.gpu
def test_constant_velocity_op():
    last = torch.tensor([[1.0, 2.0], [0.0, 0.5]], dtype=torch.float64).cuda()
    prev = torch.tensor([[0.5, 2.0], [0.0, 0.0]], dtype=torch.float64).cuda()
    out = torch.ops.trajnet.constant_velocity(last, prev, 3).cpu()
    want = torch.stack([last.cpu() + (k + 1) * (last.cpu() - prev.cpu()) for k in range(3)])
    assert torch.equal(out, want)


@pytest.mark.gpu
def test_orca_and_kalman_ops_equal_the_wrappers():
    """trajnet::orca_rollout / trajnet::kalman_predict (round 6) against classical.orca.rollout_batch / classical.kalman.
    predict_batch (the same C entry points behind the reference's wrapper logic): bit for bit; opcheck schema / fake tensors."""
    from tests.test_classical import crowd
    from trajnetplusplusbaselines_amd.classical import orca, kalman
    pos, vel, goals, speed, sizes = crowd(6, 9, 5)
    starts = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int32).cuda()
    dv = lambda a, dt: torch.tensor(np.asarray(a), dtype=dt).cuda()
    args = (dv(pos, torch.float32), dv(vel, torch.float32), dv(goals, torch.float64), dv(speed, torch.float64), starts, int(max(sizes)),
            12, 1.5, 1.5, 0.4)
    got = torch.ops.trajnet.orca_rollout(*args).cpu().numpy()
    assert np.array_equal(got, orca.rollout_batch(pos, vel, speed, goals, sizes))
    torch.library.opcheck(torch.ops.trajnet.orca_rollout, args, test_utils=('test_schema', 'test_faketensor'))
    rng = np.random.RandomState(2)
    t = np.arange(9)[None, :, None]
    obs = pos[:, None, :] + vel[:, None, :] * 0.4 * (t - 8) + rng.randn(len(pos), 9, 2) * 0.03
    z = rng.standard_normal((len(pos), 5, 13, 6))
    kargs = (dv(obs, torch.float64), dv(z, torch.float64), 10)
    kal = torch.ops.trajnet.kalman_predict(*kargs).cpu().numpy()
    assert np.array_equal(kal[:, 1:], kalman.predict_batch(obs, 12, noise=z))
    torch.library.opcheck(torch.ops.trajnet.kalman_predict, kargs, test_utils=('test_schema', 'test_faketensor'))


@pytest.mark.gpu
def test_lstm_step_op_equals_the_module_step():
    """trajnet::lstm_step == LSTM.step (dense state form) bit for bit, encoder and decoder cell, absent tracks passed through."""
    from trajnetplusplusbaselines_amd import ops as tops, synth
    from trajnetplusplusbaselines_amd.lstm import LSTM, GridBasedPooling
    torch.manual_seed(1)
    pool = GridBasedPooling(type_='social', hidden_dim=128, cell_side=0.6, n=8, out_dim=64, embedding_arch='two_layer',
                            layer_dims=[128], latent_dim=8)
    model = LSTM(pool=pool).cuda().eval()
    xy, split = synth.ragged_crowd(4, 2, 9, seed=5)
    M = xy.shape[1]
    h, c = torch.randn(M, 128).cuda(), torch.randn(M, 128).cuda()
    o1, o2, goals = xy[7].cuda(), xy[8].cuda(), torch.zeros(M, 2).cuda()
    assert torch.isnan(o2).any()
    with torch.no_grad():
        for dec in (False, True):
            (h2, c2), nrm = model.step(model.decoder if dec else model.encoder, (h, c), o1, o2, goals, split)
            args = (h, c, o1, o2, goals, split, dec, 0, tops.model_handle(model), list(model.parameters()))
            g = torch.ops.trajnet.lstm_step(*args)
            assert torch.equal(g[0], h2) and torch.equal(g[1], c2) and torch.equal(torch.nan_to_num(g[2]), torch.nan_to_num(nrm))
            torch.library.opcheck(torch.ops.trajnet.lstm_step, args, test_utils=('test_schema', 'test_faketensor'))
        absent = torch.isnan(o2[:, 0]) | torch.isnan(o1[:, 0])
        assert torch.equal(g[0][absent], h[absent]) and torch.equal(g[1][absent], c[absent])


@pytest.mark.gpu
def test_lstm_sequence_op_and_torch_compile_without_graph_break():
    """trajnet::lstm_sequence equals LSTM.forward bit for bit, passes opcheck's schema / fake-tensor tests, and an
    eval-mode model compiles with fullgraph=True (no graph break at the recurrent sequence)."""
    from trajnetplusplusbaselines_amd import ops as tops, synth
    from trajnetplusplusbaselines_amd.lstm import LSTM, GridBasedPooling
    torch.manual_seed(0)
    pool = GridBasedPooling(type_='social', hidden_dim=128, cell_side=0.6, n=8, out_dim=64, embedding_arch='two_layer',
                            layer_dims=[128], latent_dim=8)
    model = LSTM(pool=pool).cuda().eval()
    xy, split = synth.ragged_crowd(5, 2, 9, seed=3)
    obs, goals = xy[:9].cuda(), torch.zeros(xy.shape[1], 2).cuda()
    with torch.no_grad():
        rel, pred = model(obs, goals, split, n_predict=12)
        args = (obs, goals, split, None, 11, 0, tops.model_handle(model), list(model.parameters()))
        rel2, pred2 = torch.ops.trajnet.lstm_sequence(*args)
        assert torch.equal(torch.nan_to_num(rel), torch.nan_to_num(rel2)) and torch.equal(torch.nan_to_num(pred), torch.nan_to_num(pred2))
        torch.library.opcheck(torch.ops.trajnet.lstm_sequence, args, test_utils=('test_schema', 'test_faketensor'))
        compiled = torch.compile(model, backend='eager', fullgraph=True)
        rel3, pred3 = compiled(obs, goals, split, n_predict=12)
        assert torch.equal(torch.nan_to_num(pred), torch.nan_to_num(pred3)) and rel3.shape == rel.shape


@pytest.mark.gpu
def test_training_sequence_ops_and_torch_compile_without_graph_break():
    """trajnet::lstm_sequence_train / lstm_sequence_backward: the same numbers as the eager autograd.Function (outputs bit
    for bit, parameter gradients bit for bit where the Function returns one, None where it returns None), opcheck's schema /
    fake-tensor / autograd-registration tests, and a TRAIN-mode model compiles with fullgraph=True through AOT autograd."""
    from trajnetplusplusbaselines_amd import ops as tops, synth
    from trajnetplusplusbaselines_amd.lstm import LSTM, GridBasedPooling
    torch.manual_seed(0)
    pool = GridBasedPooling(type_='social', hidden_dim=128, cell_side=0.6, n=8, out_dim=64, embedding_arch='two_layer',
                            layer_dims=[128], latent_dim=8)
    model = LSTM(pool=pool).cuda().train()
    xy, split = synth.ragged_crowd(5, 2, 9, seed=3)
    xy = xy.cuda()
    obs, truth, goals = xy[:9], xy[9:20], torch.zeros(xy.shape[1], 2, device='cuda')
    targets = torch.nan_to_num(xy[9:21] - xy[8:20])

    def loss_of(m):
        rel, pred = m(obs, goals, split, truth)
        return (torch.nan_to_num(rel[-12:, :, :2]) - targets).square().mean() + 1e-3 * torch.nan_to_num(pred).square().mean()

    model.zero_grad()
    l_eager = loss_of(model)
    l_eager.backward()
    g_eager = {n: (p.grad.clone() if p.grad is not None else None) for n, p in model.named_parameters()}

    # the op pair called directly
    params = list(model.parameters())
    args = (obs, goals, split, truth, 11, 0, tops.model_handle(model), params)
    rel, pred, h_last, handle = torch.ops.trajnet.lstm_sequence_train(*args)
    with torch.no_grad():
        rel_e, pred_e = model.eval()(obs, goals, split, truth)
    model.train()
    assert handle.dtype == torch.int64 and handle.device.type == 'cpu' and rel.requires_grad
    assert torch.equal(torch.nan_to_num(rel), torch.nan_to_num(rel_e)) and torch.equal(torch.nan_to_num(pred), torch.nan_to_num(pred_e))
    torch.library.opcheck(torch.ops.trajnet.lstm_sequence_train, args, test_utils=('test_schema', 'test_faketensor',
                                                                                   'test_autograd_registration'))

    # compiled training step: one graph, gradients through the registered backward op
    model.zero_grad()
    compiled = torch.compile(loss_of, backend='aot_eager', fullgraph=True)
    l_comp = compiled(model)
    l_comp.backward()
    assert torch.equal(l_comp.detach(), l_eager.detach())
    for n, p in model.named_parameters():
        want = g_eager[n]
        if want is None:
            assert p.grad is None, n + ': unused parameters keep grad = None under torch.compile as in eager mode'
        else:
            assert p.grad is not None and torch.equal(p.grad, want), n
    # a handle serves one backward
    with pytest.raises(RuntimeError):
        torch.ops.trajnet.lstm_sequence_backward(handle.new_tensor(10 ** 9), rel.detach(), pred.detach(), h_last.detach(), params)
