"""Losses (reference lstm/loss.py): oracle vs the reference's golden values and adapted known-answer vectors
(reference tests/test_lstm_loss.py:12-43, 63-83) on CPU; HIP kernels vs both on the GPU."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import oracle
from tests import helpers

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'loss_cases.npz'))
CASES = [0, 1]


def _collision_known_answer():
    predictions = np.array([[[0, 0], [1, 0], [2, 0], [3, 0]],
                            [[0, 4], [1, 3], [2, 2], [3, 1]],
                            [[0, -3], [1, -2], [2, -1], [3, -1]],
                            [[0, -8], [1, -8], [2, -8], [3, -8]]], dtype=np.float32).transpose(1, 0, 2)
    return np.ascontiguousarray(predictions), np.array([0, 4])


def test_oracle_known_answers():
    params = np.array([[[0, 0, 1, 1, 0], [0, 0, 1, 1, 0]]], dtype=np.float32)
    coords = np.zeros((1, 2, 2), dtype=np.float32)
    got = oracle.primary_loss(0, params, coords, [0, 1, 2], background_rate=0.0, keep_batch_dim=True)
    want = -math.log(0.01 + 0.99 / (2 * math.pi))
    assert got.tolist() == pytest.approx([want, want], rel=1e-4)
    pred, split = _collision_known_answer()
    assert oracle.collision_loss(pred, split, 2.0, 2.0) == 3.0
    assert oracle.collision_loss(pred, split, 4.0, 2.0) == 6.0
    assert oracle.collision_loss(pred, split, 2.0, 4.0) == 7.5


@pytest.mark.parametrize('k', CASES)
def test_oracle_losses_match_reference(k):
    pre = 'l%d_' % k
    n, t, p, split = GOLD[pre + 'normals'], GOLD[pre + 'targets'], GOLD[pre + 'positions'], GOLD[pre + 'split']
    for bg in (0.2, 0.0):
        for keep in (False, True):
            got = np.atleast_1d(oracle.primary_loss(0, n, t, split, bg, keep))
            np.testing.assert_allclose(got, GOLD[pre + 'nll_bg%g_keep%d' % (bg, keep)], rtol=2e-5, atol=1e-6)
    for keep in (False, True):
        got = np.atleast_1d(oracle.primary_loss(1, n, t, split, 0.0, keep, multiplier=100.0))
        np.testing.assert_allclose(got, GOLD[pre + 'l2_keep%d' % keep], rtol=2e-5)
    for cw, cd in ((2.0, 0.2), (10.0, 0.5)):
        np.testing.assert_allclose(oracle.collision_loss(p, split, cw, cd), GOLD[pre + 'col_%g_%g' % (cw, cd)][0], rtol=2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize('k', CASES)
def test_gpu_losses_match_reference(k):
    from trajnetplusplusbaselines_amd.lstm import PredictionLoss, L2Loss, CollisionLoss
    pre = 'l%d_' % k
    n, t = torch.tensor(GOLD[pre + 'normals']).cuda(), torch.tensor(GOLD[pre + 'targets']).cuda()
    p, split = torch.tensor(GOLD[pre + 'positions']).cuda(), torch.tensor(GOLD[pre + 'split'])
    for bg in (0.2, 0.0):
        for keep in (False, True):
            got = np.atleast_1d(PredictionLoss(keep_batch_dim=keep, background_rate=bg)(n, t, split).cpu().numpy())
            np.testing.assert_allclose(got, GOLD[pre + 'nll_bg%g_keep%d' % (bg, keep)], rtol=2e-5, atol=1e-6)
    for keep in (False, True):
        got = np.atleast_1d(L2Loss(keep_batch_dim=keep)(n, t, split).cpu().numpy())
        np.testing.assert_allclose(got, GOLD[pre + 'l2_keep%d' % keep], rtol=2e-5)
    for cw, cd in ((2.0, 0.2), (10.0, 0.5)):
        got = float(CollisionLoss(p, split, cw, cd).cpu())
        np.testing.assert_allclose(got, GOLD[pre + 'col_%g_%g' % (cw, cd)][0], rtol=2e-5)
    # with the auxiliary collision term (lstm/loss.py:89-90)
    crit = PredictionLoss(col_wt=2.0, col_distance=0.2)
    got = float(crit(n, t, split, positions=p).cpu())
    want = GOLD[pre + 'nll_bg0.2_keep0'][0] + GOLD[pre + 'col_2_0.2'][0]
    np.testing.assert_allclose(got, want, rtol=2e-5)


@pytest.mark.gpu
def test_gpu_known_answers():
    from trajnetplusplusbaselines_amd.lstm import PredictionLoss, CollisionLoss
    params = torch.tensor([[[0, 0, 1, 1, 0], [0, 0, 1, 1, 0]]], dtype=torch.float32).cuda()
    coords = torch.zeros(1, 2, 2).cuda()
    got = PredictionLoss(keep_batch_dim=True, background_rate=0.0)(params, coords, torch.tensor([0, 1, 2])).cpu().tolist()
    want = -math.log(0.01 + 0.99 / (2 * math.pi))
    assert got == pytest.approx([want, want], rel=1e-4)
    pred, split = _collision_known_answer()
    pt = torch.tensor(pred).cuda()
    assert float(CollisionLoss(pt, torch.tensor(split), 2.0, 2.0)) == 3.0
    assert float(CollisionLoss(pt, torch.tensor(split), 4.0, 2.0)) == 6.0
    assert float(CollisionLoss(pt, torch.tensor(split), 2.0, 4.0)) == 7.5


def test_gan_losses_match_reference():
    """bce / generator / discriminator losses (host-side pointwise on [B] scores) against the reference's values with
    the same `random` seed (tests/golden/sgan_case.npz)."""
    import random
    import torch
    from trajnetplusplusbaselines_amd.lstm import bce_loss, gan_g_loss, gan_d_loss
    z = np.load(os.path.join(helpers.GOLDEN, 'sgan_case.npz'))
    sr, sf = torch.tensor(z['rand_scores_real']), torch.tensor(z['rand_scores_fake'])
    np.testing.assert_allclose(bce_loss(sr, (sf > 0).float()).item(), float(z['rand_bce']), rtol=1e-6)
    random.seed(10)
    np.testing.assert_allclose(gan_g_loss(sf).item(), float(z['rand_gan_g_loss']), rtol=1e-6)
    random.seed(10)
    np.testing.assert_allclose(gan_d_loss(sr, sf).item(), float(z['rand_gan_d_loss']), rtol=1e-6)


@pytest.mark.gpu
def test_variety_loss_matches_reference():
    """Top-k loss over the k = 3 generator samples of the S-GAN golden (reference Trainer.variety_loss)."""
    import torch
    from trajnetplusplusbaselines_amd.lstm import PredictionLoss, variety_loss
    z = np.load(os.path.join(helpers.GOLDEN, 'sgan_case.npz'))
    xy, split = torch.tensor(z['xy']), torch.tensor(z['split'])
    rel = [torch.tensor(z['truth_rel%d' % i]).cuda() for i in range(3)]
    targets = (xy[9:21] - xy[8:20]).cuda()
    got = variety_loss(PredictionLoss(keep_batch_dim=True), rel, targets, split)
    np.testing.assert_allclose(float(got), float(z['variety_loss']), rtol=2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize('mode,keep', [(0, False), (0, True), (1, False), (1, True)])
def test_fused_loss_backward_matches_the_tensor_expression(mode, keep):
    """tnp_primary_loss_backward (analytic derivatives, one launch) against autograd through the tensor restatement of
    lstm/loss.py:23-91, with scalar and per-scene upstream gradients"""
    from trajnetplusplusbaselines_amd.lstm import loss as L
    g = torch.Generator().manual_seed(3 + mode)
    T, sizes = 12, [5, 1, 9, 3]
    split = torch.tensor([0] + list(np.cumsum(sizes)))
    M = int(split[-1])
    raw = torch.randn(T, M, 5, generator=g)
    inputs = torch.cat([raw[..., :2], 0.01 + 0.2 * torch.sigmoid(raw[..., 2:4]), 0.7 * torch.sigmoid(raw[..., 4:])], dim=2).cuda()
    targets = (raw[..., :2] + 0.1 * torch.randn(T, M, 2, generator=g)).cuda()
    weights = torch.rand(len(sizes), generator=g).cuda()
    grads = []
    for fn in (L._PrimaryLossFn.apply, None):
        x = inputs.clone().requires_grad_(True)
        if fn is not None:
            out = fn(x, targets, split, mode, 0.2, keep, 1.5)
        else:
            out = helpers.primary_loss_autograd(mode, x, targets, split, 0.2, keep, 1.5)
        ((out * weights).sum() if keep else out * 0.7).backward()
        grads.append((out.detach().clone(), x.grad.clone()))
    assert torch.allclose(grads[0][0], grads[1][0], rtol=2e-5, atol=1e-6)
    scale = float(grads[1][1].abs().max())
    assert float((grads[0][1] - grads[1][1]).abs().max()) < 2e-5 * scale
    prim = split[:-1].cuda()
    mask = torch.ones(M, dtype=torch.bool, device='cuda')
    mask[prim] = False
    assert float(grads[0][1][:, mask].abs().max()) == 0.0      # only the primaries carry a gradient


COLG = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'collision_grad.npz'))
COL_PARAMS = [(10.0, 0.2), (2.0, 0.5)]


@pytest.mark.parametrize('k', range(int(COLG['num_cases'])))
def test_oracle_collision_gradient_matches_reference_autograd(k):
    """numpy restatement of d CollisionLoss / d positions vs the reference's autograd (oracle/gen_golden_r2.py)"""
    pos, split = COLG['c%d_positions' % k], COLG['c%d_split' % k]
    for cw, cd in COL_PARAMS:
        tag = 'c%d_w%g_d%g_' % (k, cw, cd)
        assert oracle.collision_loss(pos, split, cw, cd) == pytest.approx(float(COLG[tag + 'value']), rel=2e-5, abs=1e-6)
        got = oracle.collision_loss_grad(pos, split, cw, cd, grad_out=0.7)
        want = COLG[tag + 'grad']
        assert np.abs(got - want).max() <= 2e-5 * max(1.0, np.abs(want).max())
        prim = np.zeros(pos.shape[1], dtype=bool)
        prim[split[:-1]] = True
        assert not want[:, ~prim].any()                       # the reference detaches the neighbours


@pytest.mark.gpu
@pytest.mark.parametrize('k', range(int(COLG['num_cases'])))
def test_gpu_collision_gradient_matches_reference_autograd(k):
    """tnp_collision_loss_backward through CollisionLoss's autograd Function vs the reference's gradients"""
    from trajnetplusplusbaselines_amd.lstm import loss as L
    pos, split = COLG['c%d_positions' % k], COLG['c%d_split' % k]
    some = False
    for cw, cd in COL_PARAMS:
        tag = 'c%d_w%g_d%g_' % (k, cw, cd)
        p = torch.tensor(pos).cuda().requires_grad_(True)
        val = L.CollisionLoss(p * 1.0, torch.tensor(split), col_wt=cw, col_distance=cd)
        assert float(val) == pytest.approx(float(COLG[tag + 'value']), rel=2e-5, abs=1e-6)
        (val * 0.7).backward()
        want = COLG[tag + 'grad']
        got = p.grad.cpu().numpy()
        assert np.abs(got - want).max() <= 2e-5 * max(1.0, np.abs(want).max())
        assert np.array_equal(got != 0, np.abs(want) > 0) or np.abs(got - want).max() < 1e-6
        some |= bool(np.abs(want).max() > 0)
    assert some or k == 2                                       # every case but the single-track one collides


@pytest.mark.gpu
@pytest.mark.parametrize('kind', ['social', 'vanilla'])
def test_gpu_loss_fused_into_the_sequence_equals_the_stand_alone_loss(kind):
    """LSTM.forward_with_loss (the kernel that finishes a step evaluates the primaries' loss from registers) == the criterion
    applied to the returned normals, bit for bit, for PredictionLoss / L2Loss, both reductions, teacher-forced and free
    running; train_step.val_batch == Trainer.val_batch's two numbers computed the long way."""
    from trajnetplusplusbaselines_amd import synth
    from trajnetplusplusbaselines_amd.lstm import PredictionLoss, L2Loss
    from trajnetplusplusbaselines_amd.lstm.train_step import val_batch
    sd, cfg, d = helpers.load_lstm_case(kind)
    model = helpers.build_amd_model(sd, cfg)
    xy, split = synth.ragged_crowd(7, 2, 9, seed=77)
    M = xy.shape[1]
    goals = torch.zeros(M, 2)
    targets = (xy[9:21] - xy[8:20]).cuda()
    with torch.no_grad():
        for crit in (PredictionLoss(), PredictionLoss(keep_batch_dim=True, background_rate=0.1), L2Loss(), L2Loss(keep_batch_dim=True)):
            for kw in (dict(n_predict=12), dict(prediction_truth=xy[9:20].clone())):
                rel, pred, loss = model.forward_with_loss(xy[:9], goals, split, targets, crit, **kw)
                rel2, pred2 = model(xy[:9], goals, split, **kw)
                assert torch.equal(torch.nan_to_num(rel), torch.nan_to_num(rel2)) and torch.equal(torch.nan_to_num(pred), torch.nan_to_num(pred2))
                want = crit(rel2[-12:], targets, split)
                assert torch.equal(loss, want), (float(loss.sum()), float(want.sum()))
        crit = PredictionLoss()
        got = val_batch(model, crit, xy, goals, split, 9, 12, batch_size=8)
        rel_t, _ = model(xy[:9], goals, split, xy[9:20].clone())
        rel_f, _ = model(xy[:9], goals, split, n_predict=12)
        want = (float(crit(rel_t[-12:], targets, split)) * 8, float(crit(rel_f[-12:], targets, split)) * 8)
        assert got == want


@pytest.mark.gpu
@pytest.mark.parametrize('crit_name,col', [('PredictionLoss', 0.0), ('L2Loss', 0.0), ('PredictionLoss', 5.0)])
@pytest.mark.parametrize('batch_size', [8, 5])
def test_gpu_train_step_loss_takes_slice_and_batch_size_inside(crit_name, col, batch_size):
    """train_step.batch_loss hands our criteria the whole rel_outputs with ``tail=pred_length`` and ``times=batch_size``
    (no slice / multiplication nodes under autograd).  Value and gradient equal the trainer's expression
    ``criterion(rel_outputs[-pred_length:], targets, split, positions) * batch_size`` (lstm/trainer.py:258-264): bit for bit
    when batch_size is a power of two, to an ulp otherwise; frames in front of the slice get exact zeros."""
    from trajnetplusplusbaselines_amd.lstm import loss as loss_mod
    from trajnetplusplusbaselines_amd.lstm.train_step import batch_loss
    g = torch.Generator().manual_seed(3)
    S, M, pred = 20, 23, 12
    split = torch.tensor([0, 5, 6, 14, 23])
    rel = torch.randn(S, M, 5, generator=g)
    rel[..., 2:4] = rel[..., 2:4].abs() + 0.2
    rel[..., 4] = torch.tanh(rel[..., 4]) * 0.9
    rel[15, 7] = float('nan')            # a neighbour's row: not part of the loss
    rel = rel.cuda()
    targets = torch.randn(pred, M, 2, generator=g).cuda()
    scene = torch.randn(S + 1, M, 2, generator=g).cuda() * 0.3
    outputs = torch.randn(S, M, 2, generator=g).cuda() * 0.3
    crit = getattr(loss_mod, crit_name)(col_wt=col, col_distance=1.0) if col else getattr(loss_mod, crit_name)()

    def run(fast):
        r = rel.clone().requires_grad_(True)
        o = outputs.clone().requires_grad_(True)
        if fast:
            loss = batch_loss(crit, r, o, scene, targets, split, pred, batch_size)
        else:
            pos = None
            if col:
                pos = scene[-pred:].clone()
                pos[:, split[:-1].cuda()] = o[-pred:, split[:-1].cuda()]
            loss = crit(*((r[-pred:], targets, split) + ((pos,) if pos is not None else ()))) * batch_size
        loss.backward()
        return loss.detach(), r.grad, o.grad
    lf, gf, of = run(True)
    ls, gs, os_ = run(False)
    assert torch.equal(gf[:S - pred], torch.zeros_like(gf[:S - pred]))
    if batch_size & (batch_size - 1) == 0:
        assert torch.equal(lf, ls) and torch.equal(torch.nan_to_num(gf), torch.nan_to_num(gs))
    else:
        assert torch.allclose(lf, ls, rtol=3e-7, atol=0) and torch.allclose(torch.nan_to_num(gf), torch.nan_to_num(gs), rtol=3e-7, atol=1e-12)
    if col:
        assert torch.allclose(of, os_, rtol=3e-7, atol=1e-12)
