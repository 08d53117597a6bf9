"""Small-batch workloads for rocprofv3 --kernel-trace --stats: `predict` = single-scene forwards (the evaluator's call,
~36 agents), `train` = batch_size-8 optimisation steps on fresh ragged batches (bench.py's trainer_default)."""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from trajnetplusplusbaselines_amd import synth, data as trajdata
from trajnetplusplusbaselines_amd.lstm import PredictionLoss
from trajnetplusplusbaselines_amd.lstm.train_step import train_batch

mode = sys.argv[1] if len(sys.argv) > 1 else 'predict'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
device = torch.device('cuda', 0)
model = bench.build_model(bench.CONFIGS['social'], device, seed=1)
xy, split = synth.ragged_crowd(256, 8, 72, seed=2024, nan_frac=0.2)
xy_np, split_np = xy.numpy(), split.numpy()
scenes = [xy_np[:, split_np[i]:split_np[i + 1]] for i in range(len(split_np) - 1)]
if mode == 'predict':
    model.eval()
    with torch.no_grad():
        for sc in scenes[:n]:
            obs = torch.tensor(sc[:9], dtype=torch.float32, device=device)
            model(obs, torch.zeros(sc.shape[1], 2, device=device), torch.tensor([0, sc.shape[1]]), n_predict=12)
elif mode == "train":
    optimizer = bench.make_adam(model.parameters())
    batcher = trajdata.SceneBatcher(scenes, device=device, drop_distant_r=None)
    rng = random.Random(7)
    for _ in range(n):
        ids = [rng.randrange(len(scenes)) for _ in range(8)]
        bxy, bgoals, bsplit = batcher.batch(ids, augment=True)
        train_batch(model, optimizer, PredictionLoss(), bxy, bgoals, bsplit, 9, 12, batch_size=8)
torch.cuda.synchronize()
if mode == 'train_time':
    # wall clock of the pipelined batch_size-8 loop (what bench.py's trainer_default reports), three times
    import time
    optimizer = bench.make_adam(model.parameters())
    batcher = trajdata.SceneBatcher(scenes, device=device, drop_distant_r=None)
    rng = random.Random(7)
    crit = PredictionLoss()
    def one():
        ids = [rng.randrange(len(scenes)) for _ in range(8)]
        bxy, bgoals, bsplit = batcher.batch(ids, augment=True)
        return train_batch(model, optimizer, crit, bxy, bgoals, bsplit, 9, 12, batch_size=8)
    for _ in range(80):
        one()
    torch.cuda.synchronize()
    res = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(n):
            one()
        th = time.perf_counter() - t0
        torch.cuda.synchronize()
        res.append('%.3f / %.3f' % (th / n * 1e3, (time.perf_counter() - t0) / n * 1e3))
    print('%s: host enqueue / wall ms per step: %s' % (os.environ.get('TAG', ''), '   '.join(res)))
if mode == 'train_hostprofile':
    # where the HOST spends a batch_size-8 optimisation step (the loop is host-enqueue bound)
    import cProfile, pstats, time
    optimizer = bench.make_adam(model.parameters())
    batcher = trajdata.SceneBatcher(scenes, device=device, drop_distant_r=None)
    rng = random.Random(7)
    crit = PredictionLoss()
    def one():
        ids = [rng.randrange(len(scenes)) for _ in range(8)]
        bxy, bgoals, bsplit = batcher.batch(ids, augment=True)
        train_batch(model, optimizer, crit, bxy, bgoals, bsplit, 9, 12, batch_size=8)
    for _ in range(80):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        one()
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    print('%.3f ms host enqueue, %.3f ms per step' % (th / n * 1e3, (time.perf_counter() - t0) / n * 1e3))
    torch.autograd.set_multithreading_enabled(False)      # backward on THIS thread: cProfile sees inside it
    for _ in range(20):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        one()
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    print('single-threaded autograd: %.3f ms host enqueue, %.3f ms per step' % (th / n * 1e3, (time.perf_counter() - t0) / n * 1e3))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(n):
        one()
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats('cumulative').print_stats(60)
    st.sort_stats('tottime').print_stats(35)
if mode == 'train_timeline':
    # wall-clock sections of a batch_size-8 optimisation step on the host (no profiler: perf_counter around the calls)
    import time
    from trajnetplusplusbaselines_amd.lstm import train_step as ts
    from trajnetplusplusbaselines_amd import _lib
    optimizer = bench.make_adam(model.parameters())
    batcher = trajdata.SceneBatcher(scenes, device=device, drop_distant_r=None)
    rng = random.Random(7)
    crit = PredictionLoss()
    acc = {}
    def tick(name, t0):
        t1 = time.perf_counter()
        acc[name] = acc.get(name, 0.0) + (t1 - t0)
        return t1
    L = _lib.lib()
    native = {}
    for nm in ('tnp_lstm_forward_train', 'tnp_lstm_backward_sweep', 'tnp_wgrad_grouped', 'tnp_sparse_wgrad', 'tnp_adam_step',
               'tnp_primary_loss_forward', 'tnp_primary_loss_backward', 'tnp_transpose_grouped', 'tnp_sparse_hits_build',
               'tnp_pool_embed_weight_layouts', 'tnp_pair_ego_lists'):
        if hasattr(L, nm):
            def mk(nm, orig):
                def f(*a):
                    t0 = time.perf_counter()
                    r = orig(*a)
                    native[nm] = native.get(nm, 0.0) + time.perf_counter() - t0
                    return r
                return f
            setattr(L, nm, mk(nm, getattr(L, nm)))
    def one(timed):
        t = time.perf_counter()
        ids = [rng.randrange(len(scenes)) for _ in range(8)]
        bxy, bgoals, bsplit = batcher.batch(ids, augment=True)
        if timed: t = tick('batch', t)
        split = torch.as_tensor(bsplit, dtype=torch.int64)
        observed, truth = bxy[0:9], bxy[9:20]
        targets = bxy[9:21] - bxy[8:20]
        if timed: t = tick('slices', t)
        rel, out = model(observed, bgoals, split, truth)
        if timed: t = tick('forward', t)
        loss = ts.batch_loss(crit, rel, out, bxy, targets, split, 12, 8)
        if timed: t = tick('loss', t)
        rb = ts._LossReadBack(loss)
        if timed: t = tick('readback_start', t)
        optimizer.zero_grad()
        if timed: t = tick('zero_grad', t)
        with torch.autograd.set_multithreading_enabled(False):
            loss.backward()
        if timed: t = tick('backward', t)
        optimizer.step()
        if timed: t = tick('optimizer', t)
        v = rb.value()
        if timed: t = tick('readback_wait', t)
    model.train()
    for _ in range(80):
        one(False)
    torch.cuda.synchronize()
    native.clear()
    t0 = time.perf_counter()
    for _ in range(n):
        one(True)
    tot = time.perf_counter() - t0
    torch.cuda.synchronize()
    print('%.3f ms per step on the host' % (tot / n * 1e3))
    for k, v in acc.items():
        print('  %-16s %.3f ms' % (k, v / n * 1e3))
    print('native calls (inside the sections above):')
    for k, v in sorted(native.items(), key=lambda kv: -kv[1]):
        print('  %-32s %.3f ms' % (k, v / n * 1e3))
