"""cProfile of the batch_size-8 training loop on fresh ragged batches (bench.py's trainer_default leg): where the HOST time goes."""
import cProfile, os, pstats, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from trajnetplusplusbaselines_amd import synth, data as trajdata
from trajnetplusplusbaselines_amd.lstm import PredictionLoss
from trajnetplusplusbaselines_amd.lstm.train_step import train_batch

device = torch.device('cuda', 0)
model = bench.build_model(bench.CONFIGS['social'], device, seed=1)
optimizer = bench.make_adam(model.parameters())
criterion = PredictionLoss()
xy, split = synth.ragged_crowd(256, 8, 72, seed=2024, nan_frac=0.2)
xy_np, split_np = xy.numpy(), split.numpy()
scenes = [xy_np[:, split_np[i]:split_np[i + 1]] for i in range(len(split_np) - 1)]
batcher = trajdata.SceneBatcher(scenes, device=device, drop_distant_r=None)
rng = random.Random(7)
order = [[rng.randrange(len(scenes)) for _ in range(8)] for _ in range(200)]


def one(ids):
    bxy, bgoals, bsplit = batcher.batch(ids, augment=True)
    return train_batch(model, optimizer, criterion, bxy, bgoals, bsplit, 9, 12, batch_size=8)


for ids in order[:20]:
    one(ids)
torch.cuda.synchronize()
t0 = time.perf_counter()
for ids in order[20:60]:
    one(ids)
th = time.perf_counter() - t0
torch.cuda.synchronize()
print('40 steps: host %.2f ms / step, wall %.2f ms / step' % (th / 40 * 1e3, (time.perf_counter() - t0) / 40 * 1e3))
import gc
gc.collect(); gc.disable()
t0 = time.perf_counter()
for ids in order[120:160]:
    one(ids)
th = time.perf_counter() - t0
torch.cuda.synchronize()
print('40 steps, gc disabled: host %.2f ms / step, wall %.2f ms / step' % (th / 40 * 1e3, (time.perf_counter() - t0) / 40 * 1e3))
gc.enable()
pr = cProfile.Profile()
pr.enable()
for ids in order[60:120]:
    one(ids)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(14)
st.sort_stats('cumtime').print_stats(30)
