"""GPU parity of the recurrent path (tnp_lstm_forward / tnp_lstm_step) against the reference's golden outputs,
against the CPU oracle at BASELINE config sizes, and through size-independent properties."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle
from tests import helpers
from trajnetplusplusbaselines_amd import synth
from trajnetplusplusbaselines_amd.lstm import LSTM, GridBasedPooling

pytestmark = pytest.mark.gpu

KINDS = ['vanilla', 'occupancy', 'directional', 'social', 'social_goals', 'lstmlayer', 'addhidden']


@pytest.mark.parametrize('kind', KINDS)
@pytest.mark.parametrize('batch', ['lin', 'rag'])
def test_forward_matches_reference_golden(kind, batch):
    """FP32 tolerance 2e-5 on every predicted normal / position of the reference's own outputs."""
    sd, cfg, d = helpers.load_lstm_case(kind)
    model = helpers.build_amd_model(sd, cfg)
    xy = torch.tensor(d[batch + '_xy'])
    split = torch.tensor(d[batch + '_split'])
    goals = torch.tensor(d[batch + '_goals'])
    rel, pred = model(xy[:9], goals, split, n_predict=12)
    helpers.assert_close_nan(rel.cpu().numpy(), d[batch + '_rel_npredict'], 2e-5, 'rel n_predict')
    helpers.assert_close_nan(pred.cpu().numpy(), d[batch + '_pred_npredict'], 2e-5, 'pred n_predict')
    rel, pred = model(xy[:9], goals, split, prediction_truth=xy[9:20].clone())
    helpers.assert_close_nan(rel.cpu().numpy(), d[batch + '_rel_truth'], 2e-5, 'rel truth')
    helpers.assert_close_nan(pred.cpu().numpy(), d[batch + '_pred_truth'], 2e-5, 'pred truth')


@pytest.mark.parametrize('kind', ['directional', 'social'])
def test_step_matches_oracle(kind):
    """One masked step from a random state: state, normals and NaN pattern vs the oracle step."""
    sd, cfg, d = helpers.load_lstm_case(kind)
    model = helpers.build_amd_model(sd, cfg)
    om = helpers.oracle_model(sd, cfg)
    xy, split = d['rag_xy'], d['rag_split']
    M = xy.shape[1]
    rng = np.random.RandomState(1)
    h = (rng.randn(M, 128) * 0.3).astype(np.float32)
    c = (rng.randn(M, 128) * 0.3).astype(np.float32)
    for decoder, (t1, t2) in ((0, (3, 4)), (1, (10, 11))):
        h2, c2, normal, _ = om.step(decoder, h, c, xy[t1], xy[t2], None, split)
        cell = model.decoder if decoder else model.encoder
        (gh, gc), gn = model.step(cell, (torch.tensor(h).cuda(), torch.tensor(c).cuda()), torch.tensor(xy[t1]),
                                  torch.tensor(xy[t2]), None, torch.tensor(split))
        helpers.assert_close_nan(gn.cpu().numpy(), normal, 1e-5, 'normal')
        np.testing.assert_allclose(gh.cpu().numpy(), h2, atol=1e-5)
        np.testing.assert_allclose(gc.cpu().numpy(), c2, atol=1e-5)
        # the reference's list-of-tensors state form is accepted too
        hl = list(torch.tensor(h).cuda().unbind(0))
        cl = list(torch.tensor(c).cuda().unbind(0))
        (lh, lc), ln = model.step(cell, (hl, cl), torch.tensor(xy[t1]), torch.tensor(xy[t2]), None, torch.tensor(split))
        assert isinstance(lh, list) and len(lh) == M
        assert torch.equal(torch.stack(lh), gh)


def _config2_model(seed=0):
    torch.manual_seed(seed)
    pool = GridBasedPooling(type_='social', hidden_dim=128, cell_side=0.6, n=16, out_dim=256,
                            embedding_arch='two_layer', layer_dims=[1024], latent_dim=16)
    return LSTM(pool=pool).eval()


def _ade_fde(pred, truth, split):
    prim = split[:-1]
    d = np.linalg.norm(pred[:, prim].astype(np.float64) - truth[:, prim].astype(np.float64), axis=-1)
    return d.mean(axis=0), d[-1]


@pytest.mark.parametrize('scenes,agents', [(16, 32), (64, 32)])
def test_config2_social_lstm_vs_oracle(scenes, agents):
    """BASELINE config 2 (Social-LSTM n=16 two_layer 1024) at reduced and full size: per-scene ADE / FDE of the
    primary within 1e-4 m of the oracle, all positions within 1e-3."""
    model = _config2_model()
    sd = {k: v.numpy() for k, v in model.state_dict().items()}
    om = oracle.OracleModel(sd, pool_type='social', n=16, cell_side=0.6)
    model = model.cuda()
    xy, split = synth.linear_crowd(scenes, agents, seed=3)
    rel, pred = model(xy[:9], torch.zeros(xy.shape[1], 2), split, n_predict=12)
    orel, opred = om.forward(xy[:9].numpy(), None, split.numpy(), n_predict=12)
    pred, rel = pred.cpu().numpy(), rel.cpu().numpy()
    truth = xy[9:21].numpy()
    ade_g, fde_g = _ade_fde(pred[-12:], truth, split.numpy())
    ade_o, fde_o = _ade_fde(opred[-12:], truth, split.numpy())
    print('max |dADE| %.2e max |dFDE| %.2e max |dpos| %.2e' % (np.abs(ade_g - ade_o).max(),
          np.abs(fde_g - fde_o).max(), np.abs(pred - opred).max()))
    assert np.abs(ade_g - ade_o).max() < 1e-4
    assert np.abs(fde_g - fde_o).max() < 1e-4
    helpers.assert_close_nan(pred, opred, 1e-3, 'positions')
    helpers.assert_close_nan(rel, orel, 1e-3, 'normals')


def test_config3_directional_vs_oracle():
    """BASELINE config 3 per-GPU shard (D-LSTM n=12 one_layer; 32 scenes x 64 agents) with entering/leaving tracks."""
    torch.manual_seed(1)
    pool = GridBasedPooling(type_='directional', hidden_dim=128, cell_side=0.6, n=12, out_dim=256)
    model = LSTM(pool=pool).eval()
    sd = {k: v.numpy() for k, v in model.state_dict().items()}
    om = oracle.OracleModel(sd, pool_type='directional', n=12, cell_side=0.6)
    model = model.cuda()
    xy, split = synth.ragged_crowd(32, 40, 64, seed=5)
    M = xy.shape[1]
    for kw in (dict(n_predict=12), dict(prediction_truth=xy[9:20].clone())):
        rel, pred = model(xy[:9], torch.zeros(M, 2), split, **kw)
        okw = {k: (v.numpy() if hasattr(v, 'numpy') else v) for k, v in kw.items()}
        orel, opred = om.forward(xy[:9].numpy(), None, split.numpy(), **okw)
        helpers.assert_close_nan(pred.cpu().numpy(), opred, 2e-4, 'positions')
        helpers.assert_close_nan(rel.cpu().numpy(), orel, 2e-4, 'normals')


def test_properties_full_size():
    """Size-independent properties at BASELINE config 2 size: run-to-run bit reproducibility, scene-permutation
    equivariance (bitwise) and agreement of a joint batch with its two halves."""
    model = _config2_model(seed=2).cuda()
    xy, split = synth.linear_crowd(64, 32, seed=9)
    M = xy.shape[1]
    goals = torch.zeros(M, 2)
    rel_a, pred_a = model(xy[:9], goals, split, n_predict=12)
    rel_b, pred_b = model(xy[:9], goals, split, n_predict=12)
    assert torch.equal(pred_a, pred_b) and torch.equal(rel_a, rel_b)
    perm = torch.randperm(64, generator=torch.Generator().manual_seed(0))
    idx = (perm.view(-1, 1) * 32 + torch.arange(32).view(1, -1)).reshape(-1)
    rel_p, pred_p = model(xy[:9, idx], goals, split, n_predict=12)
    assert torch.equal(pred_p, pred_a[:, idx.cuda()])
    half = torch.arange(0, 32 * 32 + 1, 32)
    _, p0 = model(xy[:9, :1024], goals[:1024], half, n_predict=12)
    _, p1 = model(xy[:9, 1024:], goals[1024:], half, n_predict=12)
    # the halves may pick a different tile configuration (other summation order): fp32 rounding only
    assert (torch.cat([p0, p1], dim=1) - pred_a).abs().max().item() < 1e-4
    assert not torch.isnan(pred_a).any()


def test_edge_cases():
    sd, cfg, d = helpers.load_lstm_case('occupancy')
    model = helpers.build_amd_model(sd, cfg)
    om = helpers.oracle_model(sd, cfg)
    # (a) two observed frames only (reference lstm/lstm.py:222-223), (b) single-track scene, (c) all-NaN neighbour
    xy = np.array(d['rag_xy'])
    split = d['rag_split']
    rel, pred = model(torch.tensor(xy[7:9]), None, torch.tensor(split), n_predict=4)
    orel, opred = om.forward(xy[7:9], None, split, n_predict=4)
    assert pred.shape[0] == orel.shape[0] + 1
    helpers.assert_close_nan(pred.cpu().numpy(), opred, 2e-5, 'T_obs=2 positions')
    one = np.cumsum(np.random.RandomState(0).randn(21, 1, 2).astype(np.float32) * 0.3, axis=0)
    rel, pred = model(torch.tensor(one[:9]), None, torch.tensor([0, 1]), n_predict=12)
    orel, opred = om.forward(one[:9], None, np.array([0, 1]), n_predict=12)
    helpers.assert_close_nan(pred.cpu().numpy(), opred, 2e-5, 'single track')
    xy2 = xy.copy()
    xy2[:, 1] = np.nan
    rel, pred = model(torch.tensor(xy2[:9]), None, torch.tensor(split), n_predict=12)
    orel, opred = om.forward(xy2[:9], None, split, n_predict=12)
    helpers.assert_close_nan(pred.cpu().numpy(), opred, 2e-5, 'all-NaN neighbour')
    with pytest.raises(ValueError):
        model(torch.tensor(xy[:9]), None, torch.tensor([0, 3]), n_predict=12)


@pytest.mark.parametrize('tag', ['hotel', 'students'])
def test_real_scenes_match_reference(tag):
    """Headline model (Social-LSTM n=16, two_layer 1024) on real TrajNet++ scenes against the reference's own
    outputs (tests/golden/real_cases.npz): every normal within 5e-5, primaries' ADE/FDE within 1e-4 m."""
    model, z = helpers.real_model('cuda')
    split = z[tag + '_split']
    for name in ('raw', 'centered'):
        xy = z['%s_%s_xy' % (tag, name)].astype(np.float32)
        for dense in (False, True):
            model.sparse_embedding = not dense
            rel, pred = model(torch.tensor(xy[:9]), torch.zeros(xy.shape[1], 2), torch.tensor(split), n_predict=12)
            rel, pred = rel.cpu().numpy(), pred.cpu().numpy()
            helpers.assert_close_nan(rel, z['%s_%s_rel' % (tag, name)], 5e-5, 'rel')
            helpers.assert_close_nan(pred, z['%s_%s_pred' % (tag, name)], 5e-5, 'pred')
            prim = split[:-1]
            a0, f0 = helpers.ade_fde(z['%s_%s_pred' % (tag, name)][-12:, prim], xy[9:21, prim])
            a1, f1 = helpers.ade_fde(pred[-12:, prim], xy[9:21, prim])
            assert np.abs(a0 - a1).max() < 1e-4 and np.abs(f0 - f1).max() < 1e-4


def test_predict_batch_equals_per_scene_calls():
    """LSTMPredictor.predict_batch (one forward for many scenes) == LSTMPredictor.__call__ per scene."""
    from types import SimpleNamespace
    from trajnetplusplusbaselines_amd import data
    from trajnetplusplusbaselines_amd.lstm import LSTMPredictor
    sd, cfg, d = helpers.load_lstm_case('social')
    predictor = LSTMPredictor(helpers.build_amd_model(sd, cfg))
    xy, split = d['rag_xy'], d['rag_split']
    scenes = []
    for s in range(len(split) - 1):
        sc = xy[:, split[s]:split[s + 1]]
        paths = [[data.TrackRow(10 * t, 100 + p, float(sc[t, p, 0]), float(sc[t, p, 1]))
                  for t in range(sc.shape[0]) if not np.isnan(sc[t, p, 0])] for p in range(sc.shape[1])]
        scenes.append((paths, np.zeros((sc.shape[1], 2))))
    for normalize in (False, True):
        args = SimpleNamespace(normalize_scene=normalize)
        batched = predictor.predict_batch(scenes, n_predict=12, args=args)
        assert len(batched) == len(scenes)
        for (paths, goal), got in zip(scenes, batched):
            want = predictor(paths, goal, n_predict=12, args=args)
            helpers.assert_close_nan(got[0][0], want[0][0], 1e-5, 'primary')
            helpers.assert_close_nan(got[0][1], want[0][1], 1e-5, 'neighbours')
    assert predictor.predict_batch([]) == []


def test_config3_full_size_properties_and_sampled_oracle_parity():
    """BASELINE config 3 at its FULL size (256 scenes x up to 64 agents, D-LSTM n=12) in one process: run-to-run bit
    reproducibility; the 8 scene shards of the 8-GPU run, carrying the batch-wide slot count, against the unsharded batch
    (bit-equal whenever shard and batch select the same GEMM tile configuration -- tests/test_gpu_padding.py; here the
    16 384-track batch and the 2 048-track shards fall on different sides of the dispatcher's thresholds, so the
    summation order of the dense layers differs and the comparison is to fp32 rounding); and a sample of scenes against
    the oracle run on those scenes alone with the same slot count."""
    from trajnetplusplusbaselines_amd import parallel
    torch.manual_seed(1)
    pool = GridBasedPooling(type_='directional', hidden_dim=128, cell_side=0.6, n=12, out_dim=256)
    model = LSTM(pool=pool).eval()
    sd = {k: v.numpy() for k, v in model.state_dict().items()}
    om = oracle.OracleModel(sd, pool_type='directional', n=12, cell_side=0.6)
    model = model.cuda()
    xy, split = synth.ragged_crowd(256, 40, 64, seed=15)
    M = xy.shape[1]
    goals = torch.zeros(M, 2)
    nan_eq = lambda a, b: torch.equal(torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(b, nan=-7.0))
    with torch.no_grad():
        rel, pred = model(xy[:9], goals, split, n_predict=12)
        rel2, pred2 = model(xy[:9], goals, split, n_predict=12)
        assert nan_eq(rel, rel2) and nan_eq(pred, pred2)
        for r in range(8):
            sh = parallel.shard_batch(xy[:9], goals, split, r, 8)
            lo, hi = sh.track_range
            _, p = model(sh.observed, sh.goals, sh.batch_split, n_predict=12, pad_to=sh.pad_to)
            helpers.assert_close_nan(p.cpu().numpy(), pred[:, lo:hi].cpu().numpy(), 1e-4, 'shard %d' % r)
    n_max = int((split[1:] - split[:-1]).max())
    for s in (0, 100, 255):
        lo, hi = int(split[s]), int(split[s + 1])
        obs = xy[:9, lo:hi].numpy()
        # the oracle pads to the largest scene of ITS batch: an all-absent scene of n_max tracks gives it the batch-wide count
        ext = np.concatenate([obs, np.full((9, n_max, 2), np.nan, dtype=np.float32)], axis=1)
        _, opred = om.forward(ext, None, np.array([0, hi - lo, hi - lo + n_max]), n_predict=12)
        helpers.assert_close_nan(pred[:, lo:hi].cpu().numpy(), opred[:, :hi - lo], 2e-4, 'scene %d' % s)


def test_config4_full_size_sgan_forward_properties():
    """BASELINE config 4 at its FULL size (128 scenes x 32 agents, S-GAN directional n=12, k = 3): SGAN.forward is bit
    reproducible for fixed noise and equals, scene block by scene block, the forward of the 4 shards the 4-GPU run uses."""
    from trajnetplusplusbaselines_amd.sgan import SGAN, LSTMGenerator, LSTMDiscriminator
    torch.manual_seed(4)
    mk = lambda: GridBasedPooling(type_='directional', hidden_dim=128, cell_side=0.6, n=12, out_dim=256)
    model = SGAN(generator=LSTMGenerator(pool=mk(), noise_dim=16), discriminator=LSTMDiscriminator(pool=mk()), k=3,
                 d_steps=1, g_steps=1).eval().cuda()
    xy, split = synth.linear_crowd(128, 32, seed=44)
    M = xy.shape[1]
    gen = model.generator

    def run(obs, sp, seed):
        torch.manual_seed(seed)
        noise = torch.randn(16)
        with torch.no_grad():
            return gen(obs, torch.zeros(obs.shape[1], 2), sp, n_predict=12, noise=noise)
    rel_a, pred_a = run(xy[:9], split, 7)
    rel_b, pred_b = run(xy[:9], split, 7)
    assert torch.equal(pred_a, pred_b) and torch.equal(rel_a, rel_b) and not torch.isnan(pred_a).any()
    for r in range(4):
        lo, hi = r * 32 * 32, (r + 1) * 32 * 32
        _, p = run(xy[:9, lo:hi], torch.arange(0, 32 * 32 + 1, 32), 7)
        assert (p - pred_a[:, lo:hi]).abs().max().item() < 1e-4          # same noise, scenes independent (tile choice may differ)


def test_gates_kernel_variants_agree():
    """The LSTM-gates GEMM has four 32-track tilings (tnp_lstm_model.variant bits 8-15: 5 = 128-track tiles, 20 = split-K 4 per
    workgroup, 21 = one gate block per wave, 22 = 21 with the K range over two wave quartets) and, round 6, the 16-track
    register-operand tiles of csrc/gemm_skinny.hip (30 .. 34; 0 picks 34 up to 512 tracks): same forward within fp32
    summation order; the automatic choice is one of them, bit for bit."""
    model = _config2_model(seed=4).cuda().eval()
    xy, split = synth.ragged_crowd(12, 5, 30, seed=21)
    assert xy.shape[1] <= 512
    goals = torch.zeros(xy.shape[1], 2)
    outs = {}
    with torch.no_grad():
        for v in (0, 5, 6, 20, 21, 22, 30, 31, 32, 33, 34):
            model.kernel_variant = v << 8
            outs[v] = model(xy[:9], goals, split, n_predict=12)[1]
    model.kernel_variant = 0
    assert torch.equal(torch.nan_to_num(outs[0]), torch.nan_to_num(outs[34]))
    for v in (5, 6, 20, 21, 30, 31, 32, 33, 34):
        assert (torch.nan_to_num(outs[v]) - torch.nan_to_num(outs[22])).abs().max().item() < 2e-5
    assert torch.equal(torch.nan_to_num(outs[5]), torch.nan_to_num(outs[6]))        # the same 128-track tile, pipelined
    # the K split of a 16-track tile does not depend on how many column groups share the workgroup
    assert torch.equal(torch.nan_to_num(outs[30]), torch.nan_to_num(outs[31])) and torch.equal(torch.nan_to_num(outs[32]), torch.nan_to_num(outs[33]))
    assert torch.equal(torch.nan_to_num(outs[30]), torch.nan_to_num(outs[34]))


def _counted_flip_check(got, want, what, tol=2e-5, flip_frac=0.01, flip_tol=5e-3):
    """rows [T, R, k]: every row within `tol`, except rows whose fed-back positions sat within rounding distance of a 0.6 m
    cell edge (the neighbour is then seen one cell over and its path departs by ~1e-3): those are COUNTED -- at most
    `flip_frac` of the rows, each within `flip_tol`.  Returns the count."""
    na, nb = np.isnan(got), np.isnan(want)
    assert (na == nb).all(), what + ': NaN pattern differs'
    err = np.abs(np.nan_to_num(got).astype(np.float64) - np.nan_to_num(want).astype(np.float64)).max(axis=(0, 2))
    flipped = int((err > tol).sum())
    assert flipped <= max(1, int(len(err) * flip_frac)) and err.max() < flip_tol, (what, flipped, float(err.max()))
    return flipped


def test_config3_full_size_matches_reference():
    """BASELINE config 3 at its FULL size against the REFERENCE's own outputs (tests/golden/config3_full.npz,
    oracle/gen_golden_r4.py: D-LSTM directional n=12 one_layer out_dim 256, default init under seed 7, run single-threaded on
    synth.ragged_crowd(256, 40, 64, seed=15) = 256 scenes x 40..64 agents = 13 143 tracks with entering / leaving tracks),
    both decoder modes: the primaries' normals / positions within 2e-5 and their ADE / FDE within 1e-4 m; 512 sampled
    neighbour rows within 2e-5 under the counted cell-edge-flip rule."""
    z = np.load(os.path.join(helpers.GOLDEN, 'config3_full.npz'))
    torch.manual_seed(int(z['seed']))
    pool = GridBasedPooling(type_='directional', hidden_dim=128, cell_side=0.6, n=12, out_dim=256, embedding_arch='one_layer')
    model = LSTM(pool=pool).eval()
    for k, v in model.state_dict().items():
        assert abs(v.double().sum().item() - float(z['wsum_' + k])) < 1e-9, 'seeded weight differs: ' + k
    model = model.cuda()
    xy, split = synth.ragged_crowd(256, 40, 64, seed=int(z['crowd_seed']))
    M = xy.shape[1]
    assert M == int(z['tracks'])
    prim, rows = split[:-1].numpy(), z['rows']
    truth = xy[9:21, prim].numpy()
    flips = {}
    with torch.no_grad():
        for mode, kw in (('npredict', dict(n_predict=12)), ('truth', dict(prediction_truth=xy[9:20].clone()))):
            rel, pred = model(xy[:9], torch.zeros(M, 2), split, **kw)
            rel, pred = rel.cpu().numpy(), pred.cpu().numpy()
            helpers.assert_close_nan(rel[:, prim], z[mode + '_rel_prim'], 2e-5, mode + ' rel (primaries)')
            helpers.assert_close_nan(pred[:, prim], z[mode + '_pred_prim'], 2e-5, mode + ' pred (primaries)')
            a0, f0 = helpers.ade_fde(z[mode + '_pred_prim'][-12:], truth)
            a1, f1 = helpers.ade_fde(pred[-12:, prim], truth)
            assert np.abs(a0 - a1).max() < 1e-4 and np.abs(f0 - f1).max() < 1e-4
            flips[mode] = (_counted_flip_check(pred[:, rows], z[mode + '_pred_rows'], mode + ' pred rows'),
                           _counted_flip_check(rel[:, rows], z[mode + '_rel_rows'], mode + ' rel rows'))
    print('config 3 full size: sampled neighbour rows beyond 2e-5 (cell-edge flips):', flips)


def test_real_data_ade_fde_at_dataset_scale_matches_reference():
    """North-star "ADE / FDE vs the reference" on REAL scenes at dataset scale: 30 scenes of each of the reference's seven
    training files (tests/golden/real_eval.npz, oracle/gen_golden_r4.py:real_eval -- 210 scenes, 3 803 tracks, up to 54 agents,
    entering / leaving tracks as NaN), which the reference predicted ONE SCENE PER CALL with the headline model; here every
    file is ONE ragged batch.  Every primary's 12 predicted positions within 5e-5 m, per-scene ADE / FDE within 1e-4 m, and
    the per-file means within 2e-5 m."""
    model, _ = helpers.real_model()
    model = model.cuda()
    z = np.load(os.path.join(helpers.GOLDEN, 'real_eval.npz'))
    worst, table = 0.0, []
    for fi, fname in enumerate(z['files']):
        xy, split = torch.tensor(z['f%d_xy' % fi]), torch.tensor(z['f%d_split' % fi])
        prim = split[:-1].numpy()
        with torch.no_grad():
            _, pred = model(xy[:9], torch.zeros(xy.shape[1], 2), split, n_predict=12)
        got, want = pred[-12:].cpu().numpy()[:, prim], z['f%d_pred_prim' % fi]
        assert np.isfinite(got).all() and np.isfinite(want).all()
        truth = xy[9:21].numpy()[:, prim]
        a0, f0 = helpers.ade_fde(want, truth)
        a1, f1 = helpers.ade_fde(got, truth)
        err = float(np.abs(got - want).max())
        worst = max(worst, err)
        table.append((str(fname), len(prim), float(a0.mean()), float(a1.mean()), float(f0.mean()), float(f1.mean()), err))
        assert err < 5e-5, (fname, err)
        assert np.abs(a0 - a1).max() < 1e-4 and np.abs(f0 - f1).max() < 1e-4, fname
        assert abs(a0.mean() - a1.mean()) < 2e-5 and abs(f0.mean() - f1.mean()) < 2e-5, fname
    for row in table:
        print('%-28s %3d scenes  ADE ref %.5f ours %.5f   FDE ref %.5f ours %.5f   max |dpos| %.2e' % row)


def test_config2_full_size_neighbours_counted_flip_rule():
    """BASELINE config 2 at full size (64 x 32) against the oracle, ALL tracks: every row of positions / normals within 2e-5
    under the counted-flip rule (round 3 held the non-primaries to 1e-3 flat)."""
    model = _config2_model()
    sd = {k: v.numpy() for k, v in model.state_dict().items()}
    om = oracle.OracleModel(sd, pool_type='social', n=16, cell_side=0.6)
    model = model.cuda()
    xy, split = synth.linear_crowd(64, 32, seed=3)
    with torch.no_grad():
        rel, pred = model(xy[:9], torch.zeros(xy.shape[1], 2), split, n_predict=12)
    orel, opred = om.forward(xy[:9].numpy(), None, split.numpy(), n_predict=12)
    f = (_counted_flip_check(pred.cpu().numpy(), opred, 'positions'), _counted_flip_check(rel.cpu().numpy(), orel, 'normals'))
    print('config 2 full size: rows beyond 2e-5 of 2048 (positions, normals):', f)


def test_two_batches_in_flight_on_two_streams_equal_sequential_runs():
    """One model, two HIP streams, a different batch on each (LSTMPredictor.predict_batches keeps two forward passes in
    flight): per-stream workspaces and stream-aware caches (_lib.StreamMark) -> every output bit-equal to the same forward
    run alone, also when the re-laid-out weight copies and the scene index are built by the FIRST of the two."""
    model = _config2_model(seed=6).cuda()
    crowds = [synth.linear_crowd(64, 32, seed=31), synth.ragged_crowd(48, 5, 40, seed=32), synth.linear_crowd(64, 32, seed=33)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = []
    with torch.no_grad():
        for rep in range(3):                       # the first round builds the caches concurrently with the other stream's use
            outs = []
            for k, (xy, split) in enumerate(crowds):
                with torch.cuda.stream(streams[k % 2]):
                    outs.append(model(xy[:9].cuda(non_blocking=True), torch.zeros(xy.shape[1], 2), split, n_predict=12)[1])
            torch.cuda.synchronize()
        fresh = _config2_model(seed=6).cuda()
        for (xy, split), got in zip(crowds, outs):
            want = fresh(xy[:9], torch.zeros(xy.shape[1], 2), split, n_predict=12)[1]
            assert torch.equal(torch.nan_to_num(got), torch.nan_to_num(want))


def test_predict_batches_equals_predict_batch():
    from types import SimpleNamespace
    from trajnetplusplusbaselines_amd import data
    from trajnetplusplusbaselines_amd.lstm import LSTMPredictor
    sd, cfg, d = helpers.load_lstm_case('social')
    predictor = LSTMPredictor(helpers.build_amd_model(sd, cfg))
    xy, split = d['rag_xy'], d['rag_split']
    scenes = []
    for s in range(len(split) - 1):
        sc = xy[:, split[s]:split[s + 1]]
        paths = [[data.TrackRow(10 * t, 100 + p, float(sc[t, p, 0]), float(sc[t, p, 1]))
                  for t in range(sc.shape[0]) if not np.isnan(sc[t, p, 0])] for p in range(sc.shape[1])]
        scenes.append((paths, np.zeros((sc.shape[1], 2))))
    batches = [scenes[:2], scenes[2:], [], scenes[1:4]]
    args = SimpleNamespace(normalize_scene=True)
    got = predictor.predict_batches(batches, n_predict=12, args=args, in_flight=2)
    assert len(got) == 4 and got[2] == []
    for b, res in zip(batches, got):
        want = predictor.predict_batch(b, n_predict=12, args=args)
        assert len(res) == len(want)
        for r, w in zip(res, want):
            assert np.array_equal(r[0][0], w[0][0]) and np.array_equal(np.nan_to_num(r[0][1]), np.nan_to_num(w[0][1]))


def test_profile_hook_times_the_sparse_layer_by_its_dispatch():
    """include/trajnet_hip_profile.h: with the hook on, every launch of the sparse first layer carries its two HIP events in
    the dispatch (tnp_profile_dispatch_timed == launches) and the mean lies in the range of a kernel of that size."""
    import ctypes
    from trajnetplusplusbaselines_amd import _lib
    model = _config2_model().cuda()
    xy, split = synth.linear_crowd(64, 32, seed=3)
    L = _lib.lib()
    with torch.no_grad():
        model(xy[:9], torch.zeros(xy.shape[1], 2), split, n_predict=12)
        _lib.check(L.tnp_profile_begin(0), 'tnp_profile_begin')
        try:
            for _ in range(3):
                model(xy[:9], torch.zeros(xy.shape[1], 2), split, n_predict=12)
            torch.cuda.synchronize()
            ms, n = ctypes.c_double(0.0), ctypes.c_int(0)
            _lib.check(L.tnp_profile_read(ctypes.byref(ms), ctypes.byref(n)), 'tnp_profile_read')
            timed = int(L.tnp_profile_dispatch_timed())
        finally:
            L.tnp_profile_end()
    assert n.value == 3 * 19 and timed == n.value
    assert 5e-3 < ms.value / n.value < 0.5, ms.value / n.value        # ~0.033 ms per launch


@pytest.mark.parametrize('scenes,agents', [(1, 36), (8, 40), (31, 33), (64, 32)])
def test_round5_tiles_are_bitwise_repeatable(scenes, agents):
    """The eight-wave / small-batch GEMM tiles of round 5 (K range over several wave groups of a workgroup, partial tiles meeting in
    an LDS swap) across the batch sizes that select them: sixty forwards of the same input, every output equal to the first bit for
    bit (NaN pattern included) -- a missing barrier around the swap would show up as an occasional wrong row."""
    model = _config2_model(seed=6).cuda().eval()
    xy, split = synth.ragged_crowd(scenes, max(2, agents - 8), agents, seed=3 + scenes)
    obs, goals = xy[:9].cuda(), torch.zeros(xy.shape[1], 2).cuda()
    with torch.no_grad():
        first_rel, first = model(obs, goals, split, n_predict=12)
        for _ in range(60):
            rel, pred = model(obs, goals, split, n_predict=12)
            assert torch.equal(torch.nan_to_num(pred, nan=-7.0), torch.nan_to_num(first, nan=-7.0))
            assert torch.equal(torch.nan_to_num(rel, nan=-7.0), torch.nan_to_num(first_rel, nan=-7.0))


@pytest.mark.parametrize('case', range(5))
def test_forward_with_pool_size_and_blur_size_matches_reference(case):
    """Round 6: GridBasedPooling(pool_size, blur_size) through LSTM.forward -- fine grid, avg_pool2d blur (odd and even windows),
    lp_pool2d reduction (reference lstm/gridbased_pooling.py:297-304), csrc/pool_grid.hip grid_finish_kernel -- against the
    reference's own outputs (tests/golden/lstm_poolblur.npz, oracle/gen_golden_r6.py): three grid types, dense and ragged
    crowds, free-running and teacher-forced, 2e-5.  Training through these options raises."""
    z = np.load(os.path.join(helpers.GOLDEN, 'lstm_poolblur.npz'))
    pre = 'c%d_' % case
    cfg = {k[len(pre) + 4:]: z[k] for k in z.files if k.startswith(pre + 'cfg_')}
    sd = {k[len(pre) + 3:]: torch.tensor(z[k]) for k in z.files if k.startswith(pre + 'sd_')}
    pool = GridBasedPooling(type_=str(cfg['type']), hidden_dim=128, cell_side=0.6, n=int(cfg['n']), pool_size=int(cfg['pool_size']),
                            blur_size=int(cfg['blur_size']), out_dim=int(cfg['out_dim']), embedding_arch=str(cfg['arch']),
                            layer_dims=[int(v) for v in np.atleast_1d(cfg['dims'])], latent_dim=int(cfg['latent']))
    model = LSTM(pool=pool)
    model.load_state_dict(sd)
    model = model.cuda().eval()
    for tag in ('lin', 'rag'):
        xy, split = torch.tensor(z[pre + tag + '_xy']), torch.tensor(z[pre + tag + '_split'])
        goals = torch.zeros(xy.shape[1], 2)
        with torch.no_grad():
            rel, pred = model(xy[:9], goals, split, n_predict=12)
            helpers.assert_close_nan(rel.cpu().numpy(), z[pre + tag + '_rel_npredict'], 2e-5, 'rel n_predict')
            helpers.assert_close_nan(pred.cpu().numpy(), z[pre + tag + '_pred_npredict'], 2e-5, 'pred n_predict')
            rel, pred = model(xy[:9], goals, split, prediction_truth=xy[9:20].clone())
            helpers.assert_close_nan(rel.cpu().numpy(), z[pre + tag + '_rel_truth'], 2e-5, 'rel truth')
            helpers.assert_close_nan(pred.cpu().numpy(), z[pre + tag + '_pred_truth'], 2e-5, 'pred truth')
    model.train()
    with pytest.raises(NotImplementedError, match='pool_size'):
        model(xy[:9], goals, split, prediction_truth=xy[9:20].clone())


def test_predictor_modes_of_a_deterministic_model_share_one_forward():
    """LSTMPredictor(modes=3): the reference runs the same deterministic forward three times (lstm/lstm.py:299-311); here the plain
    LSTM runs it once per scene / batch and hands every mode its own copy -- per-scene call, predict_batch and the array-level
    entry of data.predict_dataset.  Same values as three forwards, independent arrays."""
    from trajnetplusplusbaselines_amd import data as trajdata
    from trajnetplusplusbaselines_amd.lstm import LSTMPredictor
    torch.manual_seed(5)
    pool = GridBasedPooling(type_='social', hidden_dim=128, cell_side=0.6, n=8, out_dim=64, embedding_arch='two_layer', layer_dims=[128], latent_dim=8)
    pred = LSTMPredictor(LSTM(pool=pool).cuda().eval())
    xy, split = synth.ragged_crowd(3, 3, 9, seed=2, nan_frac=0.0)
    scenes = [trajdata.xy_to_paths(xy[:, split[s]:split[s + 1]].numpy()) for s in range(3)]
    goals = [np.zeros((len(p), 2)) for p in scenes]
    one = pred(scenes[0], goals[0], n_predict=12, modes=1)
    three = pred(scenes[0], goals[0], n_predict=12, modes=3)
    assert sorted(three) == [0, 1, 2]
    for m in range(3):
        assert np.array_equal(three[m][0], one[0][0]) and np.array_equal(three[m][1], one[0][1], equal_nan=True)
    three[1][0][:] = 0.0
    assert not np.array_equal(three[0][0], three[1][0])                    # every mode owns its arrays
    batch = pred.predict_batch(list(zip(scenes, goals)), n_predict=12, modes=3)
    assert len(batch) == 3 and all(sorted(b) == [0, 1, 2] for b in batch)
    for b in batch:
        assert np.array_equal(b[0][0], b[2][0])
    arr, sp = pred.predict_xy_finish(pred.predict_xy_launch([trajdata.paths_to_xy(p) for p in scenes], goals, n_predict=12, modes=3), 12)
    assert arr.shape[0] == 3 and np.array_equal(arr[0], arr[2], equal_nan=True)


@pytest.mark.gpu
@pytest.mark.parametrize('type_,n_predict', [('directional', 12), ('occupancy', 12), ('directional', 0)])
def test_prepare_and_grid_in_one_launch_do_not_change_the_result(type_, n_predict):
    """Round 6: for occupancy / directional grids the sequence driver builds a step's grid in the SAME launch as track_prepare
    (the grid's workgroups form their scene's positions from the previous state themselves instead of waiting for the buffers
    track_prepare writes).  Same arithmetic in the same order: predictions (free-running and teacher-forced, ragged scenes with
    absent tracks, with and without goals) and training gradients are the same bit for bit as with two launches."""
    from trajnetplusplusbaselines_amd import _lib, synth
    from trajnetplusplusbaselines_amd.lstm import LSTM, GridBasedPooling, PredictionLoss
    torch.manual_seed(5)
    pool = GridBasedPooling(type_=type_, hidden_dim=128, cell_side=0.6, n=12, out_dim=256)
    model = LSTM(pool=pool, goal_flag=(type_ == 'occupancy')).cuda()
    xy, split = synth.ragged_crowd(9, 1, 70, seed=12, nan_frac=0.2)
    xy = xy.cuda()
    goals = torch.randn(xy.shape[1], 2, device='cuda')

    def run():
        model.eval()
        with torch.no_grad():
            if n_predict:
                out = model(xy[:9], goals, split, n_predict=n_predict)
            else:
                out = model(xy[:9], goals, split, prediction_truth=xy[9:20])
        model.train()
        model.zero_grad(set_to_none=True)
        rel, _ = model(xy[:9], goals, split, prediction_truth=xy[9:-1])
        PredictionLoss()(rel[-12:], xy[9:21] - xy[8:20], split).backward()
        return out, {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    try:
        _lib.tuning_set('fuse_prepare_grid', 2)
        (rel1, pred1), g1 = run()
        _lib.tuning_set('fuse_prepare_grid', 0)
        (rel0, pred0), g0 = run()
    finally:
        _lib.tuning_set('fuse_prepare_grid', 1)
    assert torch.equal(torch.isnan(pred1), torch.isnan(pred0)) and torch.isfinite(pred1[:, split[:-1]]).all()
    assert torch.equal(torch.nan_to_num(pred1), torch.nan_to_num(pred0)) and torch.equal(torch.nan_to_num(rel1), torch.nan_to_num(rel0))
    assert g1.keys() == g0.keys() and all(torch.equal(g1[k], g0[k]) for k in g1)
