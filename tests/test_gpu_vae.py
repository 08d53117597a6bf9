"""The VAE forecaster (reference vae/vae.py) on the HIP sequence driver against the reference's own runs
(tests/golden/vae_cases.npz, oracle/gen_golden_r4.py): training-mode forward, the trainer's loss (vae/trainer.py:261-274) and
every parameter gradient with the same reparametrisation noise; evaluation-mode sampling with the same numpy draws."""
import os

import numpy as np
import pytest
import torch

from tests import helpers

pytestmark = pytest.mark.gpu
Z = np.load(os.path.join(helpers.GOLDEN, 'vae_cases.npz'))


def build(kind, num_modes):
    from trajnetplusplusbaselines_amd.lstm import GridBasedPooling
    from trajnetplusplusbaselines_amd.vae import VAE
    pool = None
    if kind == 'directional':
        pool = GridBasedPooling(type_='directional', hidden_dim=128, cell_side=0.6, n=12, out_dim=64)
    model = VAE(pool=pool, num_modes=num_modes, latent_dim=32)
    pre = kind + '_sd_'
    sd = {k[len(pre):]: torch.tensor(Z[k]) for k in Z.files if k.startswith(pre)}
    assert list(model.state_dict().keys()) == list(sd.keys())          # the reference's keys, in the reference's order
    model.load_state_dict(sd)
    return model.cuda()


def test_seeded_default_init_equals_the_reference():
    """sub-modules are created in the reference's order: the same seed gives the same weights"""
    from trajnetplusplusbaselines_amd.vae import VAE
    torch.manual_seed(91)
    model = VAE(pool=None, num_modes=2, latent_dim=32)
    for k, v in model.state_dict().items():
        assert torch.equal(v, torch.tensor(Z['vanilla_sd_' + k])), k


@pytest.mark.parametrize('kind', ['vanilla', 'directional'])
def test_training_forward_loss_and_gradients_match_reference(kind):
    from trajnetplusplusbaselines_amd.lstm import PredictionLoss
    from trajnetplusplusbaselines_amd.vae import KLDLoss
    model = build(kind, 2).train()
    pre = kind + '_'
    xy, split = torch.tensor(Z[pre + 'xy']), torch.tensor(Z[pre + 'split'])
    M, B = xy.shape[1], split.numel() - 1
    targets = (xy[9:21] - xy[8:20]).cuda()
    torch.manual_seed(17)                                                  # the reparametrisation noise of the two modes
    rel, pred, z_xy, z_x = model(xy[:9].clone(), torch.zeros(M, 2), split, xy[9:20].clone())
    assert z_x is None and len(rel) == 2 and len(pred) == 2
    np.testing.assert_allclose(z_xy.detach().cpu().numpy(), Z[pre + 'z_xy'], rtol=0, atol=2e-5)
    for i in range(2):
        helpers.assert_close_nan(rel[i].detach().cpu().numpy(), Z[pre + 'train_rel%d' % i], 3e-5, 'rel mode %d' % i)
        helpers.assert_close_nan(pred[i].detach().cpu().numpy(), Z[pre + 'train_pred%d' % i], 3e-5, 'pred mode %d' % i)
    crit, kld = PredictionLoss(), KLDLoss()
    reconstr = sum(crit(r[-12:], targets, split) * B for r in rel) / model.num_modes
    kld_loss = kld(z_xy, split, z_x) * B
    loss = reconstr + 0.7 * kld_loss
    np.testing.assert_allclose(float(reconstr.detach()), float(Z[pre + 'reconstr']), rtol=3e-5)
    np.testing.assert_allclose(float(kld_loss.detach()), float(Z[pre + 'kld']), rtol=3e-5)
    loss.backward()
    worst = 0.0
    for name, p in model.named_parameters():
        if (pre + 'nograd_' + name) in Z.files:
            assert p.grad is None, name + ': the reference leaves this gradient None'
            continue
        want = Z[pre + 'grad_' + name]
        assert p.grad is not None, name
        scale = max(1e-6, float(np.abs(want).max()))
        err = float(np.abs(p.grad.cpu().numpy() - want).max()) / scale
        worst = max(worst, err)
        assert err < 1e-4, '%s: relative error %.2e (scale %.2e)' % (name, err, scale)
    print(kind, 'VAE training step: worst relative gradient error %.2e' % worst)


@pytest.mark.parametrize('kind', ['vanilla', 'directional'])
def test_evaluation_sampling_matches_reference(kind):
    model = build(kind, 3).eval()
    pre = kind + '_'
    xy, split = torch.tensor(Z[pre + 'xy']), torch.tensor(Z[pre + 'split'])
    np.random.seed(23)
    with torch.no_grad():
        rel, pred, z_xy, z_x = model(xy[:9].clone(), torch.zeros(xy.shape[1], 2), split, n_predict=12)
    assert z_xy is None and z_x is None and len(rel) == 3
    for i in range(3):
        helpers.assert_close_nan(rel[i].cpu().numpy(), Z[pre + 'eval_rel%d' % i], 3e-5, 'rel mode %d' % i)
        helpers.assert_close_nan(pred[i].cpu().numpy(), Z[pre + 'eval_pred%d' % i], 3e-5, 'pred mode %d' % i)
    assert not np.array_equal(Z[pre + 'eval_pred0'], Z[pre + 'eval_pred1'])     # the modes differ


def test_predictor_and_one_by_one_modes():
    """VAEPredictor returns the reference's dict; the replica-batched modes equal mode-by-mode runs bit for bit."""
    from types import SimpleNamespace
    from trajnetplusplusbaselines_amd import data
    from trajnetplusplusbaselines_amd.vae import VAEPredictor
    import trajnetplusplusbaselines_amd.sgan.sgan as sg
    model = build('directional', 3).eval()
    xy, split = torch.tensor(Z['directional_xy']), torch.tensor(Z['directional_split'])
    np.random.seed(5)
    with torch.no_grad():
        _, pred_a, _, _ = model(xy[:9], torch.zeros(xy.shape[1], 2), split, n_predict=12)
    keep = sg._scene_local
    try:
        sg._scene_local = lambda pool: False                  # forces one sequence per mode
        np.random.seed(5)
        with torch.no_grad():
            _, pred_b, _, _ = model(xy[:9], torch.zeros(xy.shape[1], 2), split, n_predict=12)
    finally:
        sg._scene_local = keep
    for a, b in zip(pred_a, pred_b):
        assert torch.equal(torch.nan_to_num(a), torch.nan_to_num(b))
    sc = xy[:, int(split[0]):int(split[1])].numpy()
    paths = [[data.TrackRow(10 * t, 100 + p, float(sc[t, p, 0]), float(sc[t, p, 1])) for t in range(sc.shape[0])
              if not np.isnan(sc[t, p, 0])] for p in range(sc.shape[1])]
    out = VAEPredictor(model)(paths, np.zeros((sc.shape[1], 2)), n_predict=12, modes=2, args=SimpleNamespace(normalize_scene=False))
    assert sorted(out.keys()) == [0, 1] and out[0][0].shape == (12, 2) and out[0][1].shape == (12, sc.shape[1] - 1, 2)
    assert out[1][1] == []


def test_train_batch_reduces_the_loss():
    """vae/train_step.train_batch (= vae/trainer.py:229-280): first step's reconstruction loss is the reference's (same noise),
    repeated steps on one batch reduce it."""
    from trajnetplusplusbaselines_amd.lstm import PredictionLoss
    from trajnetplusplusbaselines_amd.vae.train_step import train_batch
    model = build('directional', 2)
    xy, split = torch.tensor(Z['directional_xy']), torch.tensor(Z['directional_split'])
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
    torch.manual_seed(17)
    losses = [train_batch(model, opt, PredictionLoss(), xy, torch.zeros(xy.shape[1], 2), split, alpha_kld=0.7) for _ in range(6)]
    np.testing.assert_allclose(losses[0], float(Z['directional_reconstr']), rtol=3e-5)
    assert np.isfinite(losses).all() and min(losses[3:]) < losses[0]


def test_stateful_interaction_modules_are_refused_loudly():
    """the reference carries NearestNeighborLSTM / TrajectronPooling's own LSTM state through obs_encoder ->
    pred_encoder -> every decoder mode (vae/vae.py:229-301); the sequence driver starts every run from a zero state, so the
    mirror refuses instead of computing something else."""
    from trajnetplusplusbaselines_amd import synth
    from trajnetplusplusbaselines_amd.lstm.non_gridbased_pooling import NearestNeighborLSTM
    from trajnetplusplusbaselines_amd.vae import VAE
    pool = NearestNeighborLSTM(n=4, hidden_dim=128, out_dim=32)
    model = VAE(pool=pool).cuda().eval()
    xy, split = synth.linear_crowd(2, 5, seed=1)
    with pytest.raises(NotImplementedError, match='stateful'):
        model(xy[:9].cuda(), torch.zeros(xy.shape[1], 2).cuda(), split, n_predict=12)
