"""Generate the golden fixtures under tests/golden/ by RUNNING THE PYTHON REFERENCE
(imported from /root/reference; possible only in the build container).

    python oracle/gen_golden.py

The fixtures pin the CPU oracle (and through it the HIP path) to the real
reference: every array stored here is either an input or a tensor the
reference's own code returned.  Nothing from the reference's source is copied.

Files
-----
grid_cases.npz   GridBasedPooling.{occupancies,directional,social} outputs and
                 "tag grids" (occupancy() called with other_values = neighbour
                 index + 1, which exposes which neighbour won each cell and so
                 pins cell ids, last-writer-wins and the cell-0 clobber).
lstm_<kind>.npz  LSTM.forward outputs (both n_predict and teacher-forced mode)
                 for small models, with the full state_dict.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import ref_import  # noqa: E402
from trajnetplusplusbaselines_amd import synth  # noqa: E402

NAN = float('nan')
OUT = os.path.join(ROOT, 'tests', 'golden')


def grid_cases(ref):
    rng = np.random.RandomState(1234)
    cases = []

    def add(name, type_, obs1, obs2, n, cell_side, pool_size=1, blur_size=1, constant=0, front=False,
            hidden=None, latent=4):
        obs1 = torch.tensor(np.asarray(obs1, dtype=np.float32))
        obs2 = torch.tensor(np.asarray(obs2, dtype=np.float32))
        B, N = obs2.shape[0], obs2.shape[1]
        hd = 8
        torch.manual_seed(len(cases))
        pool = ref.GridBasedPooling(cell_side=cell_side, n=n, hidden_dim=hd, type_=type_, pool_size=pool_size,
                                    blur_size=blur_size, front=front, constant=constant, latent_dim=latent,
                                    embedding_arch='one_layer', out_dim=8)
        rec = dict(name=name, type=type_, n=n, cell_side=cell_side, pool_size=pool_size, blur_size=blur_size,
                   constant=constant, front=int(front), obs1=obs1.numpy().copy(), obs2=obs2.numpy().copy())
        with torch.no_grad():
            if type_ == 'occupancy':
                g = pool.occupancies(obs1.clone(), obs2.clone())
            elif type_ == 'directional':
                g = pool.directional(obs1.clone(), obs2.clone())
            else:
                if hidden is None:
                    hidden = rng.randn(B, N, hd).astype(np.float32)
                    # padded slots carry NaN hidden states (lstm/lstm.py:33)
                    hidden[np.isnan(obs2.numpy()).any(-1) & (rng.rand(B, N) < 0.5)] = NAN
                hidden_t = torch.tensor(hidden)
                g = pool.social(hidden_t.clone(), obs1.clone(), obs2.clone())
                rec['hidden'] = hidden
                rec['Wh'] = pool.hidden_dim_encoding.weight.detach().numpy().copy()
                rec['bh'] = pool.hidden_dim_encoding.bias.detach().numpy().copy()
            rec['grid'] = g.numpy().copy()
            # tag grid: which neighbour (1-based slot j') won each cell
            if N > 1 and pool_size == 1 and blur_size == 1:
                tagpool = ref.GridBasedPooling(cell_side=cell_side, n=n, hidden_dim=hd, type_='occupancy',
                                               front=front, constant=constant)
                tags = torch.arange(1, N, dtype=torch.float32).view(1, 1, N - 1, 1).repeat(B, N, 1, 1)
                rec['tag_grid'] = tagpool.occupancy(obs2.clone(), tags, past_obs=obs1.clone()).numpy().copy()
        cases.append(rec)

    # 1-3: adapted known-answer inputs of reference tests/test_pooling.py:9-22, 65-83, 86-99
    o = [[[0.0, 0.0], [-1.0, -1.0]]]
    add('simple', 'occupancy', o, o, n=2, cell_side=2.0, pool_size=4, blur_size=3)
    o = [[[0.0, 0.0], [-1.0, 0.0]]]
    add('midpoint', 'occupancy', o, o, n=2, cell_side=2.0, pool_size=100, blur_size=99)
    o = [[[0.0, 0.0], [NAN, NAN]]]
    add('nan', 'occupancy', o, o, n=2, cell_side=2.0)
    # 4: directional known-answer inputs (test_pooling.py:45-62; current code stores v_j - v_i)
    add('dir_simple', 'directional', [[[0.0, 0.0], [-1.0, -1.0]]], [[[0.1, 0.1], [-1.1, -1.1]]], n=2,
        cell_side=2.0, pool_size=4)
    # 5: SURVEY quirk 9 -- cell-0 clobber depends on neighbour order
    add('clobber_a', 'occupancy', np.zeros((1, 3, 2)), [[[0, 0], [-1.5, -1.5], [50, 50]]], n=4, cell_side=1.0)
    add('clobber_b', 'occupancy', np.zeros((1, 3, 2)), [[[0, 0], [50, 50], [-1.5, -1.5]]], n=4, cell_side=1.0)
    # 6: exact cell-edge coordinates (multiples of float32(0.6)), n=12 and n=16
    for n in (12, 16):
        ks = np.arange(-n // 2 - 1, n // 2 + 2)
        edge = np.float32(0.6) * ks.astype(np.float32)
        pts = np.stack([edge, edge[::-1]], -1)
        pts = np.concatenate([[[0.0, 0.0]], pts]).astype(np.float32)[None]
        add('edges_n%d' % n, 'occupancy', pts, pts, n=n, cell_side=0.6)
        add('edges_dir_n%d' % n, 'directional', pts - 0.05, pts, n=n, cell_side=0.6)
    # 7: random crowds, all types, with NaNs, ragged padding, duplicates, constant != 0, front
    for k in range(12):
        B, N = int(rng.randint(1, 5)), int(rng.randint(2, 14))
        type_ = ['occupancy', 'directional', 'social'][k % 3]
        n = [4, 8, 12, 16][k % 4]
        cs = [0.6, 0.6, 1.0, 2.0][(k // 3) % 4]
        ext = n * cs * 0.45
        obs2 = (rng.rand(B, N, 2) * 2 - 1).astype(np.float32) * ext
        obs1 = obs2 - rng.randn(B, N, 2).astype(np.float32) * 0.3
        # snap some agents to the same spot (duplicate cells) and onto cell edges
        obs2[:, 1::4] = obs2[:, 0:1] + np.float32(0.1)
        obs2[:, 2::5] = np.round(obs2[:, 2::5] / np.float32(cs)) * np.float32(cs)
        # absent agents: NaN in obs2 and/or obs1; trailing padded slots
        drop2 = rng.rand(B, N) < 0.2
        drop1 = rng.rand(B, N) < 0.15
        obs2[drop2] = NAN
        obs1[drop1] = NAN
        if N > 3:
            obs2[-1, -2:] = NAN
            obs1[-1, -2:] = NAN
        add('rand%d' % k, type_, obs1, obs2, n=n, cell_side=cs, constant=[0, 0, 0, 1][k % 4],
            front=(k % 6 == 5))
    # 8: single track (gridbased_pooling.py:252-253)
    add('single', 'occupancy', [[[0.3, 0.2]]], [[[0.4, 0.1]]], n=4, cell_side=1.0)

    flat = {'num_cases': np.int64(len(cases))}
    for i, rec in enumerate(cases):
        for k, v in rec.items():
            flat['c%d_%s' % (i, k)] = np.asarray(v)
    np.savez_compressed(os.path.join(OUT, 'grid_cases.npz'), **flat)
    print('grid_cases.npz: %d cases' % len(cases))


def lstm_case(ref, kind):
    torch.manual_seed({'vanilla': 1, 'occupancy': 2, 'directional': 3, 'social': 4, 'social_goals': 5, 'lstmlayer': 6, 'addhidden': 7}[kind])
    goal_flag = kind == 'social_goals'
    cfg = dict(kind=kind, type='', n=0, cell_side=0.6, goal_flag=int(goal_flag))
    pool = None
    if kind == 'occupancy':
        cfg.update(type='occupancy', n=8)
        pool = ref.GridBasedPooling(type_='occupancy', hidden_dim=128, cell_side=0.6, n=8, out_dim=32,
                                    embedding_arch='one_layer')
    elif kind == 'directional':
        cfg.update(type='directional', n=12)
        pool = ref.GridBasedPooling(type_='directional', hidden_dim=128, cell_side=0.6, n=12, out_dim=64,
                                    embedding_arch='one_layer')
    elif kind == 'addhidden':   # LSTM(pool_to_input=False): interaction vector added to the hidden state, out_dim == H
        cfg.update(type='occupancy', n=8)
        pool = ref.GridBasedPooling(type_='occupancy', hidden_dim=128, cell_side=0.6, n=8, out_dim=128,
                                    embedding_arch='one_layer')
    elif kind == 'lstmlayer':   # embedding_arch='lstm_layer': Linear + ReLU, pool_lstm / hidden2pool unused by forward
        cfg.update(type='directional', n=12)
        pool = ref.GridBasedPooling(type_='directional', hidden_dim=128, cell_side=0.6, n=12, out_dim=64,
                                    embedding_arch='lstm_layer')
    elif kind in ('social', 'social_goals'):
        cfg.update(type='social', n=8)
        pool = ref.GridBasedPooling(type_='social', hidden_dim=128, cell_side=0.6, n=8, out_dim=64,
                                    embedding_arch='two_layer', layer_dims=[128], latent_dim=8)
    model = ref.LSTM(pool=pool, goal_flag=goal_flag, pool_to_input=(kind != 'addhidden')).eval()
    out = {'cfg_' + k: np.asarray(v) for k, v in cfg.items()}
    for k, v in model.state_dict().items():
        out['sd_' + k] = v.numpy().copy()
    # two batches: dense linear crowd and ragged crowd with entering/leaving tracks
    for tag, (xy, split) in (('lin', synth.linear_crowd(3, 6, seed=11)),
                             ('rag', synth.ragged_crowd(5, 1 if kind != 'vanilla' else 2, 9, seed=12))):
        M = xy.shape[1]
        g = torch.Generator().manual_seed(5)
        goals = (torch.rand(M, 2, generator=g) * 10 - 5) if goal_flag else torch.zeros(M, 2)
        with torch.no_grad():
            rel_n, pred_n = model(xy[:9].clone(), goals.clone(), split, n_predict=12)
            rel_t, pred_t = model(xy[:9].clone(), goals.clone(), split, prediction_truth=xy[9:20].clone())
        out.update({tag + '_xy': xy.numpy(), tag + '_split': split.numpy(), tag + '_goals': goals.numpy(),
                    tag + '_rel_npredict': rel_n.numpy(), tag + '_pred_npredict': pred_n.numpy(),
                    tag + '_rel_truth': rel_t.numpy(), tag + '_pred_truth': pred_t.numpy()})
    np.savez_compressed(os.path.join(OUT, 'lstm_%s.npz' % kind), **out)
    print('lstm_%s.npz' % kind)


def loss_cases(ref):
    """PredictionLoss / L2Loss / CollisionLoss values of the reference on random normals (lstm/loss.py)."""
    import trajnetbaselines.lstm.loss as ref_loss
    rng = np.random.RandomState(77)
    out = {}
    for k, (T, sizes) in enumerate(((12, [5, 3, 8, 1, 6]), (11, [32] * 8))):
        split = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        M = int(split[-1])
        normals = rng.randn(T, M, 5).astype(np.float32) * 0.3
        normals[:, :, 2:4] = 0.01 + 0.2 / (1 + np.exp(-normals[:, :, 2:4]))
        normals[:, :, 4] = 0.7 / (1 + np.exp(-normals[:, :, 4]))
        targets = (normals[:, :, :2] + rng.randn(T, M, 2).astype(np.float32) * 0.1).astype(np.float32)
        positions = np.cumsum(rng.randn(T, M, 2).astype(np.float32) * 0.3, axis=0)
        positions[:, 1::3] = positions[:, 0:1] + np.float32(0.05)          # some neighbours collide with a primary
        positions[2:, 2] = np.nan
        tsplit = torch.tensor(split)
        pre = 'l%d_' % k
        out.update({pre + 'normals': normals, pre + 'targets': targets, pre + 'positions': positions, pre + 'split': split})
        for bg in (0.2, 0.0):
            for keep in (False, True):
                crit = ref.PredictionLoss(keep_batch_dim=keep, background_rate=bg)
                out[pre + 'nll_bg%g_keep%d' % (bg, keep)] = np.atleast_1d(
                    crit(torch.tensor(normals), torch.tensor(targets), tsplit).numpy())
        for keep in (False, True):
            crit = ref.L2Loss(keep_batch_dim=keep)
            out[pre + 'l2_keep%d' % keep] = np.atleast_1d(crit(torch.tensor(normals), torch.tensor(targets), tsplit).numpy())
        for cw, cd in ((2.0, 0.2), (10.0, 0.5)):
            val = ref_loss.CollisionLoss(torch.tensor(positions.copy()), tsplit, col_wt=cw, col_distance=cd)
            out[pre + 'col_%g_%g' % (cw, cd)] = np.atleast_1d(np.float32(float(val)))
    np.savez_compressed(os.path.join(OUT, 'loss_cases.npz'), **out)
    print('loss_cases.npz')


def sgan_case(ref):
    """SGAN (directional generator with noise, discriminator) forward of the reference with seeded noise."""
    import trajnetbaselines.sgan.sgan as ref_sgan
    torch.manual_seed(21)
    gpool = ref.GridBasedPooling(type_='directional', hidden_dim=128, cell_side=0.6, n=12, out_dim=64,
                                 embedding_arch='one_layer')
    dpool = ref.GridBasedPooling(type_='directional', hidden_dim=128, cell_side=0.6, n=12, out_dim=64,
                                 embedding_arch='one_layer')
    gen = ref_sgan.LSTMGenerator(pool=gpool, noise_dim=16)
    disc = ref_sgan.LSTMDiscriminator(pool=dpool)
    model = ref_sgan.SGAN(generator=gen, discriminator=disc, k=3, d_steps=1, g_steps=1).eval()
    out = {}
    for k, v in model.state_dict().items():
        out['sd_' + k] = v.numpy().copy()
    xy, split = synth.ragged_crowd(4, 2, 7, seed=31)
    M = xy.shape[1]
    goals = torch.zeros(M, 2)
    out.update(xy=xy.numpy(), split=split.numpy())
    with torch.no_grad():
        torch.manual_seed(5)
        rel, pred, s_real, s_fake = model(xy[:9].clone(), goals, split, prediction_truth=xy[9:21].clone(), step_type='g',
                                          pred_length=12)
        for i in range(3):
            out['truth_rel%d' % i] = rel[i].numpy()
            out['truth_pred%d' % i] = pred[i].numpy()
        out['scores_real'] = s_real.numpy()
        out['scores_fake'] = s_fake.numpy()
        # S-GAN trainer losses of the reference on these outputs (sgan/trainer.py:330-400, lstm/loss.py:165-208)
        import random
        import types
        import trajnetbaselines.sgan.trainer as ref_tr
        import trajnetbaselines.lstm.loss as ref_loss
        fake_self = types.SimpleNamespace(criterion=ref_loss.PredictionLoss(keep_batch_dim=True), pred_length=12)
        targets = xy[9:21] - xy[8:20]
        out['variety_loss'] = np.float64(ref_tr.Trainer.variety_loss(fake_self, rel, targets, split).item())
        random.seed(9)
        out['gan_g_loss'] = np.float64(ref_loss.gan_g_loss(s_fake).item())
        random.seed(9)
        out['gan_d_loss'] = np.float64(ref_loss.gan_d_loss(s_real, s_fake).item())
        # the discriminator's last ReLU makes these scores 0 here; pin the formulas on non-trivial scores too
        g = torch.Generator().manual_seed(17)
        sr, sf = torch.randn(9, generator=g) * 3, torch.randn(9, generator=g) * 3
        out['rand_scores_real'], out['rand_scores_fake'] = sr.numpy(), sf.numpy()
        out['rand_bce'] = np.float64(ref_loss.bce_loss(sr, (sf > 0).float()).item())
        random.seed(10)
        out['rand_gan_g_loss'] = np.float64(ref_loss.gan_g_loss(sf).item())
        random.seed(10)
        out['rand_gan_d_loss'] = np.float64(ref_loss.gan_d_loss(sr, sf).item())
        torch.manual_seed(6)
        rel, pred, _, _ = model(xy[:9].clone(), goals, split, n_predict=12)
        for i in range(3):
            out['npred_rel%d' % i] = rel[i].numpy()
            out['npred_pred%d' % i] = pred[i].numpy()
    np.savez_compressed(os.path.join(OUT, 'sgan_case.npz'), **out)
    print('sgan_case.npz')


def grad_case(ref):
    """Parameter gradients of the reference (loss.backward()) for small models: the trainer's loss
    PredictionLoss(rel[-12:], targets) * batch_size (lstm/trainer.py:252-265) plus a term on the predicted positions."""
    out = {}
    import trajnetbaselines.lstm.non_gridbased_pooling as ng
    for kind in ('social', 'directional', 'vanilla', 'nn'):
        torch.manual_seed({'social': 41, 'directional': 42, 'vanilla': 43, 'nn': 44}[kind])
        pool = None
        if kind == 'social':
            pool = ref.GridBasedPooling(type_='social', hidden_dim=128, cell_side=0.6, n=8, out_dim=64,
                                        embedding_arch='two_layer', layer_dims=[128], latent_dim=8)
        elif kind == 'directional':
            pool = ref.GridBasedPooling(type_='directional', hidden_dim=128, cell_side=0.6, n=12, out_dim=64)
        elif kind == 'nn':
            pool = ng.NearestNeighborMLP(n=4, out_dim=32)
        model = ref.LSTM(pool=pool).train()
        xy, split = synth.ragged_crowd(4, 2, 8, seed=51)
        M = xy.shape[1]
        observed, truth = xy[:9].clone(), xy[9:20].clone()
        targets = xy[9:21] - xy[8:20]
        rel, pred = model(observed, torch.zeros(M, 2), split, truth)
        crit = ref.PredictionLoss()
        loss = crit(rel[-12:], targets, split) * 8 + 0.1 * torch.nan_to_num(pred[-12:, split[:-1]]).pow(2).mean()
        loss.backward()
        pre = kind + '_'
        out[pre + 'xy'], out[pre + 'split'] = xy.numpy(), split.numpy()
        out[pre + 'loss'] = np.float32(loss.item())
        for k, v in model.state_dict().items():
            out[pre + 'sd_' + k] = v.numpy().copy()
        for k, p in model.named_parameters():
            out[pre + 'grad_' + k] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy().copy()
    np.savez_compressed(os.path.join(OUT, 'grad_cases.npz'), **out)
    print('grad_cases.npz')


def grad_nongrid_case(ref):
    """grad_case for the non-grid interaction modules that train on the HIP path (tests/golden/grad_cases_nongrid.npz)."""
    out = {}
    import trajnetbaselines.lstm.non_gridbased_pooling as ng
    for kind in ('hiddenstatemlp', 'attentionmlp', 'nn_lstm', 'traj_pool', 'addhidden'):
        torch.manual_seed({'hiddenstatemlp': 45, 'attentionmlp': 46, 'nn_lstm': 47, 'traj_pool': 48, 'addhidden': 49}[kind])
        if kind in ('hiddenstatemlp', 'attentionmlp'):
            cls = ng.HiddenStateMLPPooling if kind == 'hiddenstatemlp' else ng.AttentionMLPPooling
            pool = cls(hidden_dim=128, mlp_dim=96, mlp_dim_spatial=32, mlp_dim_vel=32, out_dim=32)
        elif kind == 'nn_lstm':
            pool = ng.NearestNeighborLSTM(n=4, hidden_dim=64, out_dim=32)
        elif kind == 'traj_pool':
            pool = ng.TrajectronPooling(hidden_dim=64, out_dim=32)
        else:   # LSTM(pool_to_input=False): the interaction vector (out_dim == hidden_dim) is added to the hidden state
            pool = ref.GridBasedPooling(type_='social', hidden_dim=128, cell_side=0.6, n=8, out_dim=128,
                                        embedding_arch='two_layer', layer_dims=[128], latent_dim=8)
        model = ref.LSTM(pool=pool, pool_to_input=(kind != 'addhidden')).train()
        xy, split = synth.ragged_crowd(4, 2, 8, seed=52)
        M = xy.shape[1]
        observed, truth = xy[:9].clone(), xy[9:20].clone()
        targets = xy[9:21] - xy[8:20]
        rel, pred = model(observed, torch.zeros(M, 2), split, truth)
        crit = ref.PredictionLoss()
        loss = crit(rel[-12:], targets, split) * 8 + 0.1 * torch.nan_to_num(pred[-12:, split[:-1]]).pow(2).mean()
        loss.backward()
        pre = kind + '_'
        out[pre + 'xy'], out[pre + 'split'] = xy.numpy(), split.numpy()
        out[pre + 'loss'] = np.float32(loss.item())
        for k, v in model.state_dict().items():
            out[pre + 'sd_' + k] = v.numpy().copy()
        for k, p in model.named_parameters():
            out[pre + 'grad_' + k] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy().copy()
    np.savez_compressed(os.path.join(OUT, 'grad_cases_nongrid.npz'), **out)
    print('grad_cases_nongrid.npz')


def grad_short_case(ref):
    """Gradients in the two corner modes of LSTM.forward: two observed frames (positions pre-seeded, lstm/lstm.py:222-223)
    and free-running decoding (n_predict, every track fed its own detached prediction) -- tests/golden/grad_short.npz."""
    out = {}
    torch.manual_seed(57)
    pool = ref.GridBasedPooling(type_='social', hidden_dim=128, cell_side=0.6, n=8, out_dim=64,
                                embedding_arch='two_layer', layer_dims=[128], latent_dim=8)
    model = ref.LSTM(pool=pool).train()
    xy, split = synth.ragged_crowd(4, 1, 7, seed=58)
    M = xy.shape[1]
    rel, pred = model(xy[:2].clone(), torch.zeros(M, 2), split, n_predict=6)
    prim = split[:-1]
    loss = rel[-6:, prim, :2].pow(2).sum() + 0.3 * torch.nan_to_num(pred[-6:, prim]).pow(2).mean() + rel[-6:, prim, 2:].sum()
    loss.backward()
    out['xy'], out['split'] = xy.numpy(), split.numpy()
    out['loss'] = np.float32(loss.item())
    out['rel'], out['pred'] = rel.detach().numpy(), pred.detach().numpy()
    for k, v in model.state_dict().items():
        out['sd_' + k] = v.numpy().copy()
    for k, p in model.named_parameters():
        out['grad_' + k] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy().copy()
    np.savez_compressed(os.path.join(OUT, 'grad_short.npz'), **out)
    print('grad_short.npz', loss.item())


def train_curve_case(ref):
    """Loss trajectory of the reference over 6 optimisation steps of Trainer.train_batch's arithmetic
    (lstm/trainer.py:229-269: teacher-forced forward, PredictionLoss * batch_size, backward, Adam lr 1e-3 wd 1e-4
    as in lstm/trainer.py:497) on two alternating batches, and the free-running ADE/FDE of the trained model."""
    torch.manual_seed(61)
    pool = ref.GridBasedPooling(type_='social', hidden_dim=128, cell_side=0.6, n=8, out_dim=64,
                                embedding_arch='two_layer', layer_dims=[128], latent_dim=8)
    model = ref.LSTM(pool=pool).train()
    out = {}
    for k, v in model.state_dict().items():
        out['sd_' + k] = v.numpy().copy()
    batches = [synth.ragged_crowd(6, 2, 8, seed=71), synth.linear_crowd(5, 7, seed=72)]
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
    crit = ref.PredictionLoss()
    losses = []
    for it in range(6):
        xy, split = batches[it % 2]
        observed, truth = xy[:9].clone(), xy[9:20].clone()
        targets = xy[9:21] - xy[8:20]
        rel, _ = model(observed, torch.zeros(xy.shape[1], 2), split, truth)
        loss = crit(rel[-12:], targets, split) * (split.numel() - 1)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    for i, (xy, split) in enumerate(batches):
        out['b%d_xy' % i], out['b%d_split' % i] = xy.numpy(), split.numpy()
    model.eval()
    with torch.no_grad():
        xy, split = batches[0]
        _, pred = model(xy[:9].clone(), torch.zeros(xy.shape[1], 2), split, n_predict=12)
    out['losses'] = np.asarray(losses, dtype=np.float64)
    out['final_pred'] = pred.numpy()
    for k, v in model.state_dict().items():
        out['final_sd_' + k] = v.numpy().copy()
    np.savez_compressed(os.path.join(OUT, 'train_curve.npz'), **out)
    print('train_curve.npz', losses)


def nongrid_cases(ref):
    """Non-grid interaction modules of the reference (lstm/non_gridbased_pooling.py): stand-alone module outputs on
    padded [B,N,*] tensors with NaN slots, and LSTM.forward with the module as `pool` (both decoder modes)."""
    import trajnetbaselines.lstm.non_gridbased_pooling as ng
    out = {}
    for kind in ('nn', 'hiddenstatemlp', 'attentionmlp', 'nn_lstm', 'traj_pool'):
        torch.manual_seed({'nn': 81, 'hiddenstatemlp': 82, 'attentionmlp': 86, 'nn_lstm': 87, 'traj_pool': 88}[kind])
        pool = {'nn': lambda: ng.NearestNeighborMLP(n=4, out_dim=32),
                'hiddenstatemlp': lambda: ng.HiddenStateMLPPooling(hidden_dim=128, out_dim=48),
                'attentionmlp': lambda: ng.AttentionMLPPooling(hidden_dim=128, out_dim=48),
                'nn_lstm': lambda: ng.NearestNeighborLSTM(n=4, hidden_dim=256, out_dim=32),
                'traj_pool': lambda: ng.TrajectronPooling(hidden_dim=256, out_dim=32)}[kind]()
        model = ref.LSTM(pool=pool).eval()
        pre = kind + '_'
        for k, v in model.state_dict().items():
            out[pre + 'sd_' + k] = v.numpy().copy()
        # module level: 3 scenes x 6 slots, some slots absent (NaN), one scene with 2 padded slots, NaN hidden there
        g = torch.Generator().manual_seed(83)
        obs2 = torch.rand(3, 6, 2, generator=g) * 6 - 3
        obs1 = obs2 - torch.randn(3, 6, 2, generator=g) * 0.2
        hidden = torch.randn(3, 6, 128, generator=g)
        obs2[0, 2] = NAN; obs1[0, 2] = NAN              # absent track with a valid (frozen) hidden state
        obs1[1, 4] = NAN                                # position known, velocity unknown
        obs2[2, 4:] = NAN; obs1[2, 4:] = NAN; hidden[2, 4:] = NAN   # padded slots
        with torch.no_grad():
            pool.reset(3 * 6, 5, 'cpu')      # zero interaction-encoder state (stateful modules)
            y = pool(hidden.clone(), obs1.clone(), obs2.clone())
        out.update({pre + 'm_obs1': obs1.numpy(), pre + 'm_obs2': obs2.numpy(), pre + 'm_hidden': hidden.numpy(),
                    pre + 'm_out': y.numpy()})
        # a 3-slot scene for the "fewer than n neighbours" branch of NearestNeighborMLP (:134-137)
        with torch.no_grad():
            pool.reset(3, 2, 'cpu')
            y3 = pool(hidden[:1, :3].clone(), obs1[:1, :3].clone(), obs2[:1, :3].clone())
        out[pre + 'm3_out'] = y3.numpy()
        for tag, (xy, split) in (('lin', synth.linear_crowd(3, 6, seed=84)), ('rag', synth.ragged_crowd(5, 1, 9, seed=85))):
            M = xy.shape[1]
            with torch.no_grad():
                rel_n, pred_n = model(xy[:9].clone(), torch.zeros(M, 2), split, n_predict=12)
                rel_t, pred_t = model(xy[:9].clone(), torch.zeros(M, 2), split, prediction_truth=xy[9:20].clone())
            out.update({pre + tag + '_xy': xy.numpy(), pre + tag + '_split': split.numpy(),
                        pre + tag + '_rel_npredict': rel_n.numpy(), pre + tag + '_pred_npredict': pred_n.numpy(),
                        pre + tag + '_rel_truth': rel_t.numpy(), pre + tag + '_pred_truth': pred_t.numpy()})
    np.savez_compressed(os.path.join(OUT, 'nongrid_cases.npz'), **out)
    print('nongrid_cases.npz')


def sgan_train_case(ref):
    """Parameter gradients of the reference for one discriminator step and one generator step of S-GAN training
    (sgan/trainer.py:258-369: SGAN.forward in train mode, loss_criterion, loss.backward()) with seeded noise / labels.
    The discriminator's last ReLU is dead at initialisation (scores 0, zero gradients), so its last bias is raised to
    keep the adversarial terms alive."""
    import random
    import types
    import trajnetbaselines.sgan.sgan as ref_sgan
    import trajnetbaselines.sgan.trainer as ref_tr
    import trajnetbaselines.lstm.loss as ref_loss
    torch.manual_seed(23)
    mk = lambda: ref.GridBasedPooling(type_='directional', hidden_dim=128, cell_side=0.6, n=12, out_dim=64,
                                      embedding_arch='one_layer')
    gen = ref_sgan.LSTMGenerator(pool=mk(), noise_dim=16)
    disc = ref_sgan.LSTMDiscriminator(pool=mk())
    model = ref_sgan.SGAN(generator=gen, discriminator=disc, k=3, d_steps=1, g_steps=1).train()
    with torch.no_grad():
        last = [mod for mod in disc.real_classifier if isinstance(mod, torch.nn.Linear)][-1]
        last.bias.fill_(0.5)
    out = {}
    for k, v in model.state_dict().items():
        out['sd_' + k] = v.numpy().copy()
    xy, split = synth.ragged_crowd(4, 2, 7, seed=33)
    M = xy.shape[1]
    goals = torch.zeros(M, 2)
    out.update(xy=xy.numpy(), split=split.numpy())
    fake_self = types.SimpleNamespace(model=model, criterion=ref_loss.PredictionLoss(keep_batch_dim=True), pred_length=12)
    fake_self.variety_loss = lambda *a: ref_tr.Trainer.variety_loss(fake_self, *a)
    targets = xy[9:21] - xy[8:20]
    for step_type in ('d', 'g'):
        model.zero_grad()
        torch.manual_seed(41)
        random.seed(7)
        rel, outs, s_real, s_fake = model(xy[:9].clone(), goals, split, xy[9:21].clone(), step_type=step_type, pred_length=12)
        loss = ref_tr.Trainer.loss_criterion(fake_self, rel, targets, split, s_fake, s_real, step_type)
        loss.backward()
        out[step_type + '_loss'] = np.float64(loss.item())
        out[step_type + '_scores_real'] = s_real.detach().numpy()
        out[step_type + '_scores_fake'] = s_fake.detach().numpy()
        for k, p in model.named_parameters():
            out[step_type + '_grad_' + k] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy().copy()
            out[step_type + '_hasgrad_' + k] = np.asarray(p.grad is not None)
    # a short adversarial run: d, g, d, g with Adam(lr 1e-3, weight_decay 1e-4) on both networks (sgan/trainer.py:540-546)
    model.zero_grad()
    g_opt = torch.optim.Adam(model.generator.parameters(), lr=1e-3, weight_decay=1e-4)
    d_opt = torch.optim.Adam(model.discriminator.parameters(), lr=1e-3, weight_decay=1e-4)
    curve = []
    for it, step_type in enumerate(('d', 'g', 'd', 'g')):
        torch.manual_seed(50 + it)
        random.seed(60 + it)
        rel, outs, s_real, s_fake = model(xy[:9].clone(), goals, split, xy[9:21].clone(), step_type=step_type, pred_length=12)
        loss = ref_tr.Trainer.loss_criterion(fake_self, rel, targets, split, s_fake, s_real, step_type)
        opt = g_opt if step_type == 'g' else d_opt
        opt.zero_grad()
        loss.backward()
        opt.step()
        curve.append(loss.item())
    out['curve'] = np.asarray(curve, dtype=np.float64)
    print('curve', curve)
    np.savez_compressed(os.path.join(OUT, 'sgan_train_case.npz'), **out)
    print('sgan_train_case.npz', out['d_loss'], out['g_loss'], out['d_scores_fake'].ravel())


def sgan_train_goals_case(ref):
    """One generator step of S-GAN training with goals (goal_flag) in generator and discriminator, directional pooling:
    tests/golden/sgan_train_goals.npz."""
    import random
    import types
    import trajnetbaselines.sgan.sgan as ref_sgan
    import trajnetbaselines.sgan.trainer as ref_tr
    import trajnetbaselines.lstm.loss as ref_loss
    torch.manual_seed(31)
    mk = lambda: ref.GridBasedPooling(type_='directional', hidden_dim=128, cell_side=0.6, n=12, out_dim=64,
                                      embedding_arch='one_layer')
    gen = ref_sgan.LSTMGenerator(pool=mk(), noise_dim=16, goal_flag=True, goal_dim=64)
    disc = ref_sgan.LSTMDiscriminator(pool=mk(), goal_flag=True, goal_dim=64)
    model = ref_sgan.SGAN(generator=gen, discriminator=disc, k=2, d_steps=1, g_steps=1).train()
    with torch.no_grad():
        last = [mod for mod in disc.real_classifier if isinstance(mod, torch.nn.Linear)][-1]
        last.bias.fill_(0.5)
    out = {}
    for k, v in model.state_dict().items():
        out['sd_' + k] = v.numpy().copy()
    xy, split = synth.ragged_crowd(4, 2, 7, seed=35)
    M = xy.shape[1]
    goals = torch.nan_to_num(xy[20]) + 0.5 * torch.randn(M, 2)
    out.update(xy=xy.numpy(), split=split.numpy(), goals=goals.numpy())
    fake_self = types.SimpleNamespace(model=model, criterion=ref_loss.PredictionLoss(keep_batch_dim=True), pred_length=12)
    fake_self.variety_loss = lambda *a: ref_tr.Trainer.variety_loss(fake_self, *a)
    targets = xy[9:21] - xy[8:20]
    model.zero_grad()
    torch.manual_seed(44)
    random.seed(10)
    rel, outs, s_real, s_fake = model(xy[:9].clone(), goals, split, xy[9:21].clone(), step_type='g', pred_length=12)
    loss = ref_tr.Trainer.loss_criterion(fake_self, rel, targets, split, s_fake, s_real, 'g')
    loss.backward()
    out['g_loss'] = np.float64(loss.item())
    out['g_scores_real'] = s_real.detach().numpy()
    out['g_scores_fake'] = s_fake.detach().numpy()
    for k, p in model.named_parameters():
        out['g_grad_' + k] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy().copy()
    np.savez_compressed(os.path.join(OUT, 'sgan_train_goals.npz'), **out)
    print('sgan_train_goals.npz', out['g_loss'], out['g_scores_fake'].ravel())


def sgan_train_social_case(ref):
    """One generator step of S-GAN training with SOCIAL pooling in generator and discriminator (gradient from the scores
    through the discriminator's input positions into the generator): tests/golden/sgan_train_social.npz."""
    import random
    import types
    import trajnetbaselines.sgan.sgan as ref_sgan
    import trajnetbaselines.sgan.trainer as ref_tr
    import trajnetbaselines.lstm.loss as ref_loss
    torch.manual_seed(29)
    mk = lambda: ref.GridBasedPooling(type_='social', hidden_dim=128, cell_side=0.6, n=8, out_dim=64,
                                      embedding_arch='two_layer', layer_dims=[128], latent_dim=8)
    gen = ref_sgan.LSTMGenerator(pool=mk(), noise_dim=16)
    disc = ref_sgan.LSTMDiscriminator(pool=mk())
    model = ref_sgan.SGAN(generator=gen, discriminator=disc, k=2, d_steps=1, g_steps=1).train()
    with torch.no_grad():
        last = [mod for mod in disc.real_classifier if isinstance(mod, torch.nn.Linear)][-1]
        last.bias.fill_(0.5)
    out = {}
    for k, v in model.state_dict().items():
        out['sd_' + k] = v.numpy().copy()
    xy, split = synth.ragged_crowd(4, 2, 7, seed=34)
    M = xy.shape[1]
    goals = torch.zeros(M, 2)
    out.update(xy=xy.numpy(), split=split.numpy())
    fake_self = types.SimpleNamespace(model=model, criterion=ref_loss.PredictionLoss(keep_batch_dim=True), pred_length=12)
    fake_self.variety_loss = lambda *a: ref_tr.Trainer.variety_loss(fake_self, *a)
    targets = xy[9:21] - xy[8:20]
    model.zero_grad()
    torch.manual_seed(43)
    random.seed(9)
    rel, outs, s_real, s_fake = model(xy[:9].clone(), goals, split, xy[9:21].clone(), step_type='g', pred_length=12)
    loss = ref_tr.Trainer.loss_criterion(fake_self, rel, targets, split, s_fake, s_real, 'g')
    loss.backward()
    out['g_loss'] = np.float64(loss.item())
    out['g_scores_real'] = s_real.detach().numpy()
    out['g_scores_fake'] = s_fake.detach().numpy()
    for k, p in model.named_parameters():
        out['g_grad_' + k] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy().copy()
    np.savez_compressed(os.path.join(OUT, 'sgan_train_social.npz'), **out)
    print('sgan_train_social.npz', out['g_loss'], out['g_scores_fake'].ravel())


REAL_SEED = 123


def real_model(mod_pool, mod_lstm):
    """The headline (config 2) Social-LSTM, default torch init under a fixed seed; our module and the
    reference's construct their parameters in the same order, so the same seed gives the same weights."""
    torch.manual_seed(REAL_SEED)
    pool = mod_pool(type_='social', hidden_dim=128, cell_side=0.6, n=16, out_dim=256,
                    embedding_arch='two_layer', layer_dims=[1024], latent_dim=16)
    return mod_lstm(pool=pool)


def real_cases(ref):
    """Reference LSTM.forward on REAL TrajNet++ scenes (DATA_BLOCK/trajdata/train/*.ndjson of the reference
    checkout) with the full-size config-2 model: pins ADE/FDE parity on real crowds."""
    import trajnetbaselines.lstm.utils as ref_utils
    import trajnetbaselines.lstm.lstm as ref_lstm
    from trajnetplusplusbaselines_amd import data
    model = real_model(ref.GridBasedPooling, ref.LSTM).eval()
    out = {'seed': np.asarray(REAL_SEED)}
    for k, v in model.state_dict().items():
        out['wsum_' + k] = np.asarray(v.double().sum().item())
    root = os.path.join(ref_import.REFERENCE_ROOT, 'DATA_BLOCK', 'trajdata', 'train')
    for tag, fname, count, stride in (('hotel', 'biwi_hotel.ndjson', 16, 3), ('students', 'crowds_students001.ndjson', 8, 11)):
        scenes = data.read_ndjson_scenes(os.path.join(root, fname), limit=count * stride)[::stride][:count]
        xs, centered = [], []
        for _, paths in scenes:
            xy = data.paths_to_xy(paths)
            xy, _ = ref_lstm.drop_distant(xy)
            xs.append(xy)
            c, rot, center = ref_utils.center_scene(xy.copy(), 9)
            centered.append(c)
        xy, split = data.batch_scenes(xs)
        cxy, _ = data.batch_scenes(centered)
        for name, arr in (('raw', xy), ('centered', cxy)):
            t = torch.tensor(arr, dtype=torch.float32)
            with torch.no_grad():
                rel, pred = model(t[:9].clone(), torch.zeros(t.shape[1], 2), torch.tensor(split), n_predict=12)
            out.update({'%s_%s_xy' % (tag, name): arr, '%s_%s_rel' % (tag, name): rel.numpy(),
                        '%s_%s_pred' % (tag, name): pred.numpy()})
        out[tag + '_split'] = split
        print(tag, xy.shape, split)
    np.savez_compressed(os.path.join(OUT, 'real_cases.npz'), **out)
    print('real_cases.npz')


def main():
    os.makedirs(OUT, exist_ok=True)
    ref = ref_import.import_reference()
    if '--only-grad-nongrid' in sys.argv:
        return grad_nongrid_case(ref)
    if '--only-grad-short' in sys.argv:
        return grad_short_case(ref)
    grad_short_case(ref)
    if '--only-sgangoals' in sys.argv:
        return sgan_train_goals_case(ref)
    if '--only-sgansocial' in sys.argv:
        return sgan_train_social_case(ref)
    sgan_train_goals_case(ref)
    sgan_train_social_case(ref)
    if '--only-sgantrain' in sys.argv:
        return sgan_train_case(ref)
    sgan_train_case(ref)
    if '--only-lstmlayer' in sys.argv:
        lstm_case(ref, 'addhidden')
        return lstm_case(ref, 'lstmlayer')
    lstm_case(ref, 'lstmlayer')
    lstm_case(ref, 'addhidden')
    if '--only-nongrid' in sys.argv:
        return nongrid_cases(ref)
    nongrid_cases(ref)
    if '--only-curve' in sys.argv:
        return train_curve_case(ref)
    train_curve_case(ref)
    if '--only-real' in sys.argv:
        return real_cases(ref)
    real_cases(ref)
    grad_nongrid_case(ref)
    if '--only-grad' in sys.argv:
        return grad_case(ref)
    grad_case(ref)
    sgan_case(ref)
    if '--only-sgan' in sys.argv:
        return
    loss_cases(ref)
    if '--only-new' in sys.argv:
        return
    grid_cases(ref)
    for kind in ('vanilla', 'occupancy', 'directional', 'social', 'social_goals'):
        lstm_case(ref, kind)
    with open(os.path.join(OUT, 'PROVENANCE.txt'), 'w') as f:
        f.write('generated by oracle/gen_golden.py from the reference at %s\n' % ref_import.REFERENCE_ROOT)
        f.write('torch %s, numpy %s\n' % (torch.__version__, np.__version__))


if __name__ == '__main__':
    main()
