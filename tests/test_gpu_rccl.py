"""RCCL itself (backend 'nccl' on ROCm), one process per GPU.  The two-rank tests are skipped on boxes with fewer than two
GPUs (a one-rank 'nccl' group runs c10d's RCCL code path on any box) -- the first multi-GPU box that runs `pytest -m gpu` executes the in-backward gradient all-reduce (parallel.GradReducer), the flat
buckets and the sharded train_batch over xGMI.  The same logic runs on gloo in tests/test_parallel_gloo.py (CPU)."""
import os
import socket

import numpy as np
import pytest
import torch

from tests import helpers

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _build_model():
    from trajnetplusplusbaselines_amd.lstm import LSTM, GridBasedPooling
    z = np.load(os.path.join(helpers.GOLDEN, 'train_curve.npz'))
    pool = GridBasedPooling(type_='social', hidden_dim=128, cell_side=0.6, n=8, out_dim=64, embedding_arch='two_layer',
                            layer_dims=[128], latent_dim=8)
    model = LSTM(pool=pool)
    model.load_state_dict({k[3:]: torch.tensor(z[k]) for k in z.files if k.startswith('sd_')})
    return model, z


def _worker(rank, world, port, mode, ret):
    import torch.distributed as dist
    from trajnetplusplusbaselines_amd import parallel
    from trajnetplusplusbaselines_amd.lstm import PredictionLoss
    from trajnetplusplusbaselines_amd.lstm.train_step import train_batch
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    torch.cuda.set_device(rank)
    dev = torch.device('cuda', rank)
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    try:
        model, z = _build_model()
        model = model.to(dev)
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
        buckets = parallel.GradBuckets(model.parameters()) if mode == 'buckets' else None
        losses = []
        for it in range(4):
            xy, split = torch.tensor(z['b%d_xy' % (it % 2)]), torch.tensor(z['b%d_split' % (it % 2)])
            sh = parallel.shard_batch(xy, torch.zeros(xy.shape[1], 2), split, rank, world)
            lo, hi = sh.track_range
            loss = train_batch(model, opt, PredictionLoss(), xy[:, lo:hi].contiguous().to(dev), torch.zeros(hi - lo, 2, device=dev),
                               sh.batch_split, 9, 12, batch_size=sh.n_scenes_global, n_global_scenes=sh.n_scenes_global,
                               pad_to=sh.pad_to, buckets=buckets, overlap=(mode == 'overlap'))
            t = torch.tensor([loss], dtype=torch.float64, device=dev)
            dist.all_reduce(t)                        # the shards' scaled losses add up to the single-process loss
            losses.append(float(t.item()))
        torch.cuda.synchronize()
        sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
        # every rank must hold the same weights after the same all-reduced steps
        for k, v in model.state_dict().items():
            ref = v.detach().clone()
            dist.broadcast(ref, src=0)
            assert torch.equal(ref, v.detach()), 'rank %d diverged from rank 0 in %s' % (rank, k)
        if rank == 0:
            ret['losses'] = losses
            ret['sd'] = sd
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs two GPUs (RCCL refuses two ranks on one device)')
@pytest.mark.parametrize('mode', ['overlap', 'buckets', 'flat'])
def test_two_rank_rccl_training_matches_reference_curve(mode):
    """Two ranks, one GPU each, scenes sharded (parallel.shard_batch), gradients summed over RCCL -- from inside the backward
    pass (GradReducer), through flat persistent buckets, or as one flattened bucket: the summed loss trajectory equals the
    REFERENCE's single-process one (tests/golden/train_curve.npz), the ranks end with identical weights, and those equal the
    reference's trained weights within half an Adam step."""
    import torch.multiprocessing as mp
    world = 2
    mgr = mp.get_context('spawn').Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), mode, ret), nprocs=world, join=True)
    _, z = _build_model()
    np.testing.assert_allclose(ret['losses'], z['losses'][:4], rtol=5e-5)
    # (the fixture's final weights are after six steps; four were run here: compare with a single-process run instead)
    from trajnetplusplusbaselines_amd.lstm import PredictionLoss
    from trajnetplusplusbaselines_amd.lstm.train_step import train_batch
    model, _ = _build_model()
    model = model.cuda()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
    for it in range(4):
        xy, split = torch.tensor(z['b%d_xy' % (it % 2)]), torch.tensor(z['b%d_split' % (it % 2)])
        train_batch(model, opt, PredictionLoss(), xy, torch.zeros(xy.shape[1], 2), split, 9, 12)
    for k, v in model.state_dict().items():
        assert np.abs(v.cpu().numpy() - ret['sd'][k]).max() < 5e-4, k


@pytest.mark.parametrize('mode', ['overlap', 'buckets', 'flat'])
def test_one_rank_rccl_backend_runs_the_reducers(mode):
    """What a ONE-GPU box can execute of the RCCL path: a one-rank 'nccl' process group.  c10d's ProcessGroupNCCL -- its side
    streams, the asynchronous work handles GradReducer waits on inside the backward pass, the flat buckets -- runs for real
    (gloo is synchronous on the host, so tests/test_parallel_gloo.py cannot catch a missing stream dependency); only the
    xGMI transfers are missing.  Four optimisation steps follow the REFERENCE's loss trajectory (train_curve.npz) and end
    with the weights of a run without any process group."""
    import torch.multiprocessing as mp
    mgr = mp.get_context('spawn').Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(1, _free_port(), mode, ret), nprocs=1, join=True)
    _, z = _build_model()
    np.testing.assert_allclose(ret['losses'], z['losses'][:4], rtol=5e-5)
    from trajnetplusplusbaselines_amd.lstm import PredictionLoss
    from trajnetplusplusbaselines_amd.lstm.train_step import train_batch
    model, _ = _build_model()
    model = model.cuda()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
    for it in range(4):
        xy, split = torch.tensor(z['b%d_xy' % (it % 2)]), torch.tensor(z['b%d_split' % (it % 2)])
        train_batch(model, opt, PredictionLoss(), xy, torch.zeros(xy.shape[1], 2), split, 9, 12)
    for k, v in model.state_dict().items():
        assert np.abs(v.cpu().numpy() - ret['sd'][k]).max() < 1e-6, k


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs two GPUs')
def test_bench_self_launches_two_ranks():
    """`python bench.py --gpus 2` without a launcher re-executes itself under torch.distributed.run and prints ONE JSON line
    with n_gpus = 2, the training leg's all-reduce bytes and config 3's strong-scaling leg."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '5', '--warmup', '2',
                          '--no-traffic', '--no-sustain'], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 2 and rec['scaling'] == 'weak' and rec['value'] > 0
    assert rec['training']['allreduce_bytes'] > 0 and rec['training']['first_step_check']['ok']
    s3 = rec['strong_scaling_config3']
    assert s3['scaling'] == 'strong' and s3['n_gpus'] == 2 and s3['inference']['value'] > 0 and s3['training']['allreduce_bytes'] > 0


def test_bench_self_launch_on_one_gpu_with_gloo_stand_in():
    """The same bare `python bench.py --gpus 2` on a ONE-GPU box: both ranks share the device and gloo stands in for RCCL
    (bench.py's test hooks TNP_BENCH_SHARE_GPU / TNP_BENCH_BACKEND; the numbers mean nothing).  Covers the self-launch, the
    N > 1 sharding, barrier / MAX-over-ranks timing, the in-backward all-reduce and the strong-scaling leg end to end."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    env.update(TNP_BENCH_SHARE_GPU='1', TNP_BENCH_BACKEND='gloo')
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1',
                          '--no-traffic', '--no-sustain', '--no-cpu-baseline'], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 2 and rec['config']['global_scenes'] == 128
    assert rec['training']['allreduce_bytes'] > 0 and rec['training']['first_step_check']['ok']
    s3 = rec['strong_scaling_config3']
    assert s3['n_gpus'] == 2 and s3['scenes_this_rank'] == 128 and s3['training']['allreduce_bytes'] > 0


def _sgan_model():
    from trajnetplusplusbaselines_amd.lstm import GridBasedPooling
    from trajnetplusplusbaselines_amd.sgan import SGAN, LSTMGenerator, LSTMDiscriminator
    z = np.load(os.path.join(helpers.GOLDEN, 'sgan_train_case.npz'))
    mk = lambda: GridBasedPooling(type_='directional', hidden_dim=128, cell_side=0.6, n=12, out_dim=64, embedding_arch='one_layer')
    model = SGAN(generator=LSTMGenerator(pool=mk(), noise_dim=16), discriminator=LSTMDiscriminator(pool=mk()), k=3,
                 d_steps=1, g_steps=1)
    model.load_state_dict({k[3:]: torch.tensor(z[k]) for k in z.files if k.startswith('sd_')})
    return model, z


def _sgan_steps(model, xy, split, goals, dev, **kw):
    import random
    from trajnetplusplusbaselines_amd.lstm import PredictionLoss
    from trajnetplusplusbaselines_amd.sgan.train_step import train_batch
    g_opt = torch.optim.Adam(model.generator.parameters(), lr=1e-3, weight_decay=1e-4)
    d_opt = torch.optim.Adam(model.discriminator.parameters(), lr=1e-3, weight_decay=1e-4)
    crit = PredictionLoss(keep_batch_dim=True)
    torch.manual_seed(77)            # the same noise vectors / noisy labels on every rank (and in the single-process run)
    random.seed(78)
    return [train_batch(model, g_opt, d_opt, crit, xy.to(dev), goals.to(dev), split, st, **kw) for st in ('d', 'g', 'd', 'g')]


def _sgan_worker(rank, world, port, ret):
    import torch.distributed as dist
    from trajnetplusplusbaselines_amd import parallel
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    dist.init_process_group('gloo', rank=rank, world_size=world)     # both ranks on ONE GPU: gloo stands in for RCCL
    try:
        model, z = _sgan_model()
        model = model.to(dev)
        xy, split = torch.tensor(z['xy']), torch.tensor(z['split'])
        sh = parallel.shard_batch(xy, torch.zeros(xy.shape[1], 2), split, rank, world)
        lo, hi = sh.track_range
        losses = _sgan_steps(model, xy[:, lo:hi].contiguous(), sh.batch_split, torch.zeros(hi - lo, 2), dev,
                             n_global_scenes=sh.n_scenes_global, pad_to=sh.pad_to)
        t = torch.tensor(losses, dtype=torch.float64)
        dist.all_reduce(t)
        if rank == 0:
            ret['losses'] = t.tolist()
            ret['sd'] = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    finally:
        dist.destroy_process_group()


def test_sharded_sgan_training_equals_single_process():
    """S-GAN training with the scenes sharded over two ranks (BASELINE config 4 runs on 4 GPUs): the variety loss enters as
    a SUM over scenes, the adversarial BCE terms as MEANS weighted n_local / n_global, gradients of the updated network are
    SUM-reduced -- d, g, d, g steps leave the weights of a single-process run on the whole batch (same noise and labels)."""
    import torch.multiprocessing as mp
    world = 2
    mgr = mp.get_context('spawn').Manager()
    ret = mgr.dict()
    mp.spawn(_sgan_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    model, z = _sgan_model()
    model = model.cuda()
    xy, split = torch.tensor(z['xy']), torch.tensor(z['split'])
    losses = _sgan_steps(model, xy, split, torch.zeros(xy.shape[1], 2), torch.device('cuda'))
    np.testing.assert_allclose(ret['losses'], losses, rtol=2e-5)
    for k, v in model.state_dict().items():
        assert np.abs(v.cpu().numpy() - ret['sd'][k]).max() < 2e-4, k
