"""N > 1 path on CPU: world_size-2 gloo processes shard a batch by scenes, run the (oracle) forward on their
shard, gather, and must reproduce the single-process result exactly; gradient bucketing and the loss-scaling rule
are checked against a single-process reference.  The GPU kernels are not involved (no GPU here): this covers the
sharding / gather / reduce logic that bench.py and multi-GPU training rely on."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import helpers
from trajnetplusplusbaselines_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _oracle_forward_padded(om, obs, lsplit, pad_to):
    """The oracle pads every scene to the largest scene of the batch it is given (lstm/lstm.py:29).  A shard must
    behave as part of the WHOLE batch: an extra scene of `pad_to` absent tracks gives the oracle the batch-wide slot
    count -- the CPU stand-in for ``LSTM.forward(..., pad_to=shard.pad_to)`` of the HIP path."""
    T, M = obs.shape[0], obs.shape[1]
    ext = np.concatenate([obs, np.full((T, pad_to, 2), np.nan, dtype=obs.dtype)], axis=1)
    _, pred = om.forward(ext, None, np.concatenate([lsplit, [M + pad_to]]), n_predict=12)
    return pred[:, :M]


def _worker(rank, world, port, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        sd, cfg, d = helpers.load_lstm_case('social')
        om = helpers.oracle_model(sd, cfg)
        xy = torch.tensor(d['rag_xy'])
        split = torch.tensor(d['rag_split'])
        M = xy.shape[1]
        shard = parallel.shard_batch(xy[:9], torch.zeros(M, 2), split, rank, world)
        obs, goals, lsplit, _, rng = shard
        pred = _oracle_forward_padded(om, obs.numpy(), lsplit.numpy(), shard.pad_to)
        full = parallel.gather_tracks(torch.tensor(pred), M, rng)
        t = parallel.max_over_ranks(float(rank + 1))
        # gradient all-reduce: each rank holds grad = rank+1 on two tensors, one tensor has no grad
        p1, p2, p3 = (torch.nn.Parameter(torch.zeros(5, 3)), torch.nn.Parameter(torch.zeros(7)),
                      torch.nn.Parameter(torch.zeros(2)))
        p1.grad = torch.full((5, 3), float(rank + 1))
        p2.grad = torch.full((7,), 10.0 * (rank + 1))
        n_msgs = parallel.allreduce_gradients([p1, p2, p3], bucket_bytes=16)
        reducer = parallel.GradReducer(coalesce_below=0)      # the in-backward form: asynchronous, one handle per gradient
        t1, t2 = torch.full((6,), float(rank + 1)), torch.full((3, 2), 10.0 * (rank + 1))
        handles = [reducer(t1), reducer(t2)]
        for h in handles:
            h.wait()
        # small gradients coalesced into one flattened message, a large one on its own
        red2 = parallel.GradReducer(coalesce_below=64)
        s1, s2, big = torch.full((5,), float(rank + 1)), torch.full((2, 3), 2.0 * (rank + 1)), torch.full((40,), 3.0 * (rank + 1))
        hs = [red2(s1), red2(big), red2(s2)]
        hs.append(red2.flush())
        assert red2.flush() is None
        for h in hs:
            h.wait()
        # a caller that only knows "call, then wait" (never flush()): the handle of a coalesced gradient sends the message
        red3 = parallel.GradReducer(coalesce_below=64)
        u1, u2 = torch.full((4,), float(rank + 1)), torch.full((3,), 5.0 * (rank + 1))
        h1, h2 = red3(u1), red3(u2)
        h1.wait()
        h2.wait()
        # a pass that raised before flush() leaves tensors behind: begin() drops them, they do not join the next message
        stale = torch.full((2,), 100.0)
        red3(stale)
        red3.begin()
        u3 = torch.full((2,), 7.0 * (rank + 1))
        red3(u3)
        red3.flush().wait()
        if rank == 0:
            ret['lazy'] = (u1.tolist(), u2.tolist(), stale.tolist(), u3.tolist(), red3.messages)
            ret['full'] = full.numpy()
            ret['ranges'] = rng
            ret['tmax'] = t
            ret['g1'] = p1.grad.numpy().copy()
            ret['g2'] = p2.grad.numpy().copy()
            ret['n_msgs'] = n_msgs
            ret['p3_none'] = p3.grad is None
            ret['reducer'] = (t1.tolist(), t2.reshape(-1).tolist(), reducer.messages, reducer.bytes)
            ret['coalesced'] = (s1.tolist(), s2.reshape(-1).tolist(), big.tolist(), red2.messages, red2.bytes)
    finally:
        dist.destroy_process_group()


def test_two_rank_scene_sharding_matches_single_process():
    world = 2
    mgr = mp.get_context('spawn').Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    sd, cfg, d = helpers.load_lstm_case('social')
    om = helpers.oracle_model(sd, cfg)
    xy, split = d['rag_xy'], d['rag_split']
    M = xy.shape[1]
    # the UNSHARDED batch in one process: shards that carry the batch-wide slot count reproduce it bit for bit
    _, want = om.forward(xy[:9], None, split, n_predict=12)
    sizes = np.diff(split)
    lo0, hi0 = parallel.shard_bounds(split, 0, world)
    assert sizes[lo0:hi0].max() != sizes[hi0:].max(), 'the fixture must give the two shards different largest scenes'
    got = ret['full']
    assert got.shape == want.shape
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.array_equal(np.nan_to_num(got), np.nan_to_num(want))
    assert ret['tmax'] == 2.0
    assert np.all(ret['g1'] == 3.0) and np.all(ret['g2'] == 30.0)
    assert ret['n_msgs'] == 2 and ret['p3_none']
    assert ret['reducer'] == ([3.0] * 6, [30.0] * 6, 2, 48)
    assert ret['lazy'] == ([3.0] * 4, [15.0] * 3, [100.0] * 2, [21.0] * 2, 2)
    assert ret['coalesced'] == ([3.0] * 5, [6.0] * 6, [9.0] * 40, 2, (5 + 6 + 40) * 4)      # two messages for three tensors


@pytest.mark.parametrize('balance', ['scenes', 'tracks', 'pairs'])
@pytest.mark.parametrize('world', [1, 2, 3, 8])
def test_shard_bounds_partition(balance, world):
    rng = np.random.RandomState(world)
    sizes = rng.randint(1, 40, size=37)
    split = np.concatenate([[0], np.cumsum(sizes)])
    covered = []
    for r in range(world):
        lo, hi = parallel.shard_bounds(split, r, world, balance)
        assert 0 <= lo <= hi <= len(sizes)
        covered.extend(range(lo, hi))
    assert covered == list(range(len(sizes)))      # contiguous, disjoint, complete
    if world > 1 and balance == 'pairs':
        w = [float(np.sum(sizes[slice(*parallel.shard_bounds(split, r, world, balance))].astype(float) ** 2))
             for r in range(world)]
        assert max(w) <= 2.5 * (sum(w) / world) + float(sizes.max()) ** 2


def test_loss_scaling_rule_reproduces_single_process_gradient():
    """mean over all primaries x batch_size == sum over ranks of (local mean x batch_size x n_local/n_global)
    when every scene contributes the same number of loss elements (reference lstm/trainer.py:263)."""
    torch.manual_seed(0)
    w = torch.nn.Parameter(torch.randn(4))
    x = torch.randn(6, 12, 4)          # 6 scenes x 12 predicted steps
    batch_size = 8                     # nominal batch size, may differ from the 6 scenes present
    loss = (x @ w).pow(2).mean() * batch_size
    g_ref, = torch.autograd.grad(loss, w)
    g_sum = torch.zeros(4)
    for lo, hi in ((0, 2), (2, 6)):
        local = (x[lo:hi] @ w).pow(2).mean()
        g, = torch.autograd.grad(parallel.scale_loss_for_sharding(local, batch_size, hi - lo, 6), w)
        g_sum += g
    assert torch.allclose(g_sum, g_ref, atol=1e-6)


class _StandInModel(torch.nn.Module):
    """CPU stand-in with LSTM.forward's signature (the HIP model needs a GPU): per-track linear maps of the velocities plus
    a per-scene interaction term, and one parameter that never receives a gradient (like the unused goal embedding)."""

    def __init__(self):
        super(_StandInModel, self).__init__()
        torch.manual_seed(5)
        self.a = torch.nn.Linear(2, 5)
        self.b = torch.nn.Linear(2, 2)
        self.unused = torch.nn.Parameter(torch.ones(3))

    def forward(self, observed, goals, batch_split, prediction_truth=None, n_predict=None, pad_to=None):
        frames = torch.cat([observed, prediction_truth], dim=0)
        vel = torch.nan_to_num(frames[1:] - frames[:-1])
        rel = self.a(vel)
        rel = torch.cat([rel[..., :2], 0.01 + 0.2 * torch.sigmoid(rel[..., 2:4]), 0.7 * torch.sigmoid(rel[..., 4:])], dim=-1)
        mean = torch.stack([vel[:, lo:hi].mean(dim=1) for lo, hi in zip(batch_split[:-1], batch_split[1:])], dim=1)
        sizes = batch_split[1:] - batch_split[:-1]
        pos = torch.nan_to_num(frames[1:]) + self.b(vel) + torch.repeat_interleave(self.b(mean), sizes, dim=1)
        return rel, pos


class _StandInLoss(torch.nn.Module):
    """A mean-type term (primaries' NLL, like PredictionLoss) plus a SUM-over-scenes collision penalty written with tensor
    ops as the reference's CollisionLoss is (lstm/loss.py:138-162; the HIP kernel needs a GPU) -- exposed through
    ``terms()`` like the product's losses so that the sharded step scales the two differently."""
    col_wt = 1.0      # train_batch then hands over primary_prediction (truth frames, primaries <- the model's positions)
    col_distance = 2.5

    def terms(self, inputs, targets, batch_split, positions=None):
        nll = helpers.primary_loss_autograd(0, inputs, torch.nan_to_num(targets), batch_split, 0.2, False, 1.0)
        mean_term = nll + 0.01 * positions[:, batch_split[:-1]].pow(2).mean()
        col = positions.new_zeros(())
        pos = torch.nan_to_num(positions, nan=-1000.0)
        for lo, hi in zip(batch_split[:-1].tolist(), batch_split[1:].tolist()):
            if hi - lo < 2:
                continue
            d = torch.norm(pos[:, lo:lo + 1] - pos[:, lo + 1:hi].detach(), dim=2)
            col = col + self.col_wt * torch.relu(1.0 - d / self.col_distance).sum()
        return mean_term, col

    def forward(self, inputs, targets, batch_split, positions=None):
        mean_term, col = self.terms(inputs, targets, batch_split, positions)
        return mean_term + col


def _train_worker(rank, world, port, ret, n_scenes, use_buckets):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from trajnetplusplusbaselines_amd import synth
        from trajnetplusplusbaselines_amd.lstm.train_step import train_batch
        xy, split = synth.ragged_crowd(n_scenes, 2, 7, seed=17)
        model = _StandInModel()
        opt = torch.optim.SGD(model.parameters(), lr=0.1)
        buckets = parallel.GradBuckets(model.parameters(), bucket_bytes=32) if use_buckets else None
        losses = []
        for _ in range(2):
            shard = parallel.shard_batch(xy, torch.zeros(xy.shape[1], 2), split, rank, world)
            losses.append(train_batch(model, opt, _StandInLoss(), shard.observed, shard.goals, shard.batch_split, 9, 12,
                                      batch_size=8, n_global_scenes=shard.n_scenes_global, pad_to=shard.pad_to, buckets=buckets))
        ret[rank] = dict(a=model.a.weight.detach().numpy().copy(), b=model.b.bias.detach().numpy().copy(),
                         unused_grad_none=model.unused.grad is None, losses=losses, n_local=shard.n_scenes,
                         n_buckets=len(buckets.buckets) if use_buckets else 0)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('n_scenes,use_buckets', [(5, False), (5, True), (1, False), (1, True)])
def test_train_batch_distributed_branch_equals_single_process(n_scenes, use_buckets):
    """lstm/train_step.train_batch with a process group (world_size 2, gloo): two optimisation steps on scene shards leave
    every rank with the weights of the single-process run on the whole batch -- loss-scaling rule, SUM all-reduce (flat
    bucket or GradBuckets), a parameter without gradient skipped on all ranks, and (n_scenes = 1) a rank whose shard is
    empty taking part with a zero-weighted dummy scene."""
    from trajnetplusplusbaselines_amd import synth
    from trajnetplusplusbaselines_amd.lstm.train_step import train_batch
    world = 2
    mgr = mp.get_context('spawn').Manager()
    ret = mgr.dict()
    mp.spawn(_train_worker, args=(world, _free_port(), ret, n_scenes, use_buckets), nprocs=world, join=True)
    xy, split = synth.ragged_crowd(n_scenes, 2, 7, seed=17)
    model = _StandInModel()
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    single = [train_batch(model, opt, _StandInLoss(), xy, torch.zeros(xy.shape[1], 2), split, 9, 12, batch_size=8)
              for _ in range(2)]
    for r in range(world):
        assert np.allclose(ret[r]['a'], model.a.weight.detach().numpy(), rtol=1e-5, atol=1e-6)
        assert np.allclose(ret[r]['b'], model.b.bias.detach().numpy(), rtol=1e-5, atol=1e-6)
        assert ret[r]['unused_grad_none']
    assert np.array_equal(ret[0]['a'], ret[1]['a'])                       # replicas stay bit-identical
    assert sum(ret[r]['n_local'] for r in range(world)) == n_scenes
    if n_scenes == 1:
        assert ret[1]['n_local'] == 0 and ret[1]['losses'] == [0.0, 0.0]
    # the collision-type term is live in this batch (otherwise the test would not see a mis-scaled SUM term)
    rel, pos = _StandInModel()(xy[:9], None, split, xy[9:20])
    crit = _StandInLoss()
    primary_prediction = xy[-12:].clone()
    primary_prediction[:, split[:-1]] = pos[-12:, split[:-1]]
    assert n_scenes == 1 or float(crit.terms(rel[-12:], xy[9:21] - xy[8:20], split, primary_prediction)[1]) > 0.1
    # the ranks' scaled losses add up to the single-process loss (mean over all primaries x batch_size + collision sum)
    assert np.allclose(np.sum([ret[r]['losses'] for r in range(world)], axis=0), single, rtol=1e-5)
    if use_buckets:
        assert ret[0]['n_buckets'] >= 2


def test_shard_bounds_never_leaves_a_rank_empty_when_there_are_enough_scenes():
    for split, world in (([0, 9, 10, 11, 12], 2), ([0, 1, 2, 3, 12], 4), ([0, 30, 31, 32, 33, 34, 35, 36, 37], 8)):
        b = [parallel.shard_bounds(split, r, world) for r in range(world)]
        assert all(hi > lo for lo, hi in b), (split, b)
        assert [lo for lo, _ in b][1:] == [hi for _, hi in b][:-1] and b[0][0] == 0 and b[-1][1] == len(split) - 1
    assert [parallel.shard_bounds([0, 5], r, 3) for r in range(3)] == [(0, 1), (1, 1), (1, 1)]
