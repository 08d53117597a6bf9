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
        obs, goals, lsplit, _, rng = parallel.shard_batch(xy[:9], torch.zeros(M, 2), split, rank, world)
        # every scene of a shard keeps the batch-wide padded slot count only if it holds the largest scene;
        # the shard forward is therefore compared with the same shard run stand-alone in rank 0 below
        rel, pred = om.forward(obs.numpy(), None, lsplit.numpy(), n_predict=12)
        full = parallel.gather_tracks(torch.tensor(pred), M, rng)
        t = parallel.max_over_ranks(float(rank + 1))
        # gradient all-reduce: each rank holds grad = rank+1 on two tensors, one tensor has no grad
        p1, p2, p3 = (torch.nn.Parameter(torch.zeros(5, 3)), torch.nn.Parameter(torch.zeros(7)),
                      torch.nn.Parameter(torch.zeros(2)))
        p1.grad = torch.full((5, 3), float(rank + 1))
        p2.grad = torch.full((7,), 10.0 * (rank + 1))
        n_msgs = parallel.allreduce_gradients([p1, p2, p3], bucket_bytes=16)
        if rank == 0:
            ret['full'] = full.numpy()
            ret['ranges'] = rng
            ret['tmax'] = t
            ret['g1'] = p1.grad.numpy().copy()
            ret['g2'] = p2.grad.numpy().copy()
            ret['n_msgs'] = n_msgs
            ret['p3_none'] = p3.grad is None
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
    # single process, shard by shard (same n_max per shard as the workers saw)
    want = np.empty((19, M, 2), dtype=np.float32)
    for r in range(world):
        obs, _, lsplit, _, (lo, hi) = parallel.shard_batch(torch.tensor(xy[:9]), None, torch.tensor(split), r, world)
        _, pred = om.forward(obs.numpy(), None, lsplit.numpy(), n_predict=12)
        want[:, lo:hi] = pred
    got = ret['full']
    assert got.shape == want.shape
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.array_equal(np.nan_to_num(got), np.nan_to_num(want))
    assert ret['tmax'] == 2.0
    assert np.all(ret['g1'] == 3.0) and np.all(ret['g2'] == 30.0)
    assert ret['n_msgs'] == 2 and ret['p3_none']


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
