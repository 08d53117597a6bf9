"""The Python REFERENCE at its real operating point, timed in the BUILD CONTAINER (it cannot travel to the GPU box): the headline
model (Social-LSTM n=16 two_layer 1024) (a) trained at the trainer's default batch_size = 8 on the ragged crowd bench.py's
`trainer_default` leg draws from (synth.ragged_crowd(256, 8, 72, seed=2024): forward + loss + backward + Adam as
Trainer.train_batch does, lstm/trainer.py:229-269) and (b) predicting one scene per call (lstm/lstm.py:285-313).
usage: python tools/ref_operating_point.py > profiles/round5_reference_operating_point.txt"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from oracle import ref_import
from trajnetplusplusbaselines_amd import synth

ref = ref_import.import_reference()
torch.manual_seed(1)
pool = ref.GridBasedPooling(type_='social', hidden_dim=128, cell_side=0.6, n=16, out_dim=256, embedding_arch='two_layer',
                            layer_dims=[1024], latent_dim=16)
model = ref.LSTM(pool=pool)
import trajnetbaselines.lstm.loss as ref_loss
criterion = ref_loss.PredictionLoss()
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
xy, split = synth.ragged_crowd(256, 8, 72, seed=2024, nan_frac=0.2)
split = split.numpy()
scenes = [xy[:, split[i]:split[i + 1]] for i in range(len(split) - 1)]
rng = random.Random(7)
print('threads', torch.get_num_threads(), 'cpu count', os.cpu_count())


def train_step(ids):
    batch = torch.cat([scenes[i] for i in ids], dim=1)
    bs = torch.tensor(np.concatenate([[0], np.cumsum([scenes[i].shape[1] for i in ids])]), dtype=torch.int64)
    observed, truth = batch[:9].clone(), batch[9:20].clone()
    targets = batch[9:21] - batch[8:20]
    rel, _ = model(observed, torch.zeros(batch.shape[1], 2), bs, truth)
    loss = criterion(rel[-12:], targets, bs) * 8
    opt.zero_grad()
    loss.backward()
    opt.step()
    return float(loss), batch.shape[1]


model.train()
train_step([rng.randrange(256) for _ in range(8)])
t0, tracks, n = time.time(), 0, 0
while time.time() - t0 < 60 and n < 12:
    _, m = train_step([rng.randrange(256) for _ in range(8)])
    tracks += m
    n += 1
t_train = (time.time() - t0) / n
print('reference optimisation step, batch_size 8, ragged 8..72 agents (%.0f tracks per step): %.1f ms per step (%d steps)' % (tracks / n, t_train * 1e3, n))

model.eval()
t0, n, agents = time.time(), 0, 0
with torch.no_grad():
    for sc in scenes[:40]:
        model(sc[:9].clone(), torch.zeros(sc.shape[1], 2), torch.tensor([0, sc.shape[1]]), n_predict=12)
        n += 1
        agents += sc.shape[1]
        if time.time() - t0 > 60:
            break
print('reference per-scene forward (n_predict=12), %.1f agents per scene: %.1f ms per scene (%d scenes)' % (agents / n, (time.time() - t0) / n * 1e3, n))
