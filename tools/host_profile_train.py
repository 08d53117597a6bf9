"""Where the HOST spends an optimisation step (cProfile over train_batch; the GPU idles while the host prepares the forward
call after the previous step's loss read-back).  python tools/host_profile_train.py [social|directional] [steps]"""
import cProfile
import pstats
import sys
import time

import torch

sys.path.insert(0, '.')
import bench
from trajnetplusplusbaselines_amd import synth
from trajnetplusplusbaselines_amd.lstm import PredictionLoss
from trajnetplusplusbaselines_amd.lstm.train_step import train_batch

cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else 'social']
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dev = torch.device('cuda', 0)
model = bench.build_model(cfg, dev)
xy, split = synth.linear_crowd(cfg['scenes'], cfg['agents'], seed=100)
scene = xy.to(dev)
goals = torch.zeros(xy.shape[1], 2, device=dev)
opt = bench.make_adam(model.parameters())
crit = PredictionLoss()
for _ in range(5):
    train_batch(model, opt, crit, scene, goals, split, 9, 12, batch_size=cfg['scenes'])
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    train_batch(model, opt, crit, scene, goals, split, 9, 12, batch_size=cfg['scenes'])
torch.cuda.synchronize()
print('%.3f ms per step (unprofiled)' % (1e3 * (time.perf_counter() - t0) / steps))
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    train_batch(model, opt, crit, scene, goals, split, 9, 12, batch_size=cfg['scenes'])
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('cumulative')
st.print_stats(45)

# ---- host timeline of one step: when the native calls are entered, relative to the start of train_batch ----
from trajnetplusplusbaselines_amd import _lib
L = _lib.lib()
marks = []
def wrap(name):
    orig = getattr(L, name)
    def f(*a):
        marks.append((name + ' >', time.perf_counter()))
        r = orig(*a)
        marks.append((name + ' <', time.perf_counter()))
        return r
    setattr(L, name, f)
for n in ('tnp_lstm_forward_train', 'tnp_lstm_backward_sweep', 'tnp_wgrad_grouped', 'tnp_sparse_wgrad', 'tnp_adam_step'):
    if hasattr(L, n):
        wrap(n)
acc = {}
for it in range(20):
    marks.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    train_batch(model, opt, crit, scene, goals, split, 9, 12, batch_size=cfg['scenes'])
    t1 = time.perf_counter()
    for k, t in marks:
        acc.setdefault(k, []).append(1e3 * (t - t0))
    acc.setdefault('train_batch returns', []).append(1e3 * (t1 - t0))
print('host timeline, ms after train_batch was entered (mean of 20 steps):')
for k, v in acc.items():
    print('  %-34s %.3f' % (k, sum(v) / len(v)))

# ---- the same for the python functions on the way to the first native call ----
import trajnetplusplusbaselines_amd.lstm.training as tr
import trajnetplusplusbaselines_amd.lstm.lstm as lm
def wrap_py(obj, name, label):
    orig = getattr(obj, name)
    def f(*a, **k):
        marks.append((label + ' >', time.perf_counter()))
        r = orig(*a, **k)
        marks.append((label + ' <', time.perf_counter()))
        return r
    setattr(obj, name, f)
wrap_py(type(model), '_descriptor', '_descriptor')
wrap_py(type(model), '_workspace', '_workspace')
wrap_py(tr, '_train_saves', '_train_saves')
wrap_py(tr, 'run_sequence_with_grad', 'run_sequence_with_grad')
wrap_py(type(opt), 'zero_grad', 'zero_grad')
wrap_py(_lib.SceneIndex, 'get', 'SceneIndex.get')
acc = {}
for it in range(20):
    marks.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    train_batch(model, opt, crit, scene, goals, split, 9, 12, batch_size=cfg['scenes'])
    seen = set()
    for k, t in marks:
        if k in seen:
            continue                      # first occurrence only (the backward calls some of them again)
        seen.add(k)
        acc.setdefault(k, []).append(1e3 * (t - t0))
print('python functions before the first native call (first call of each, ms after entry):')
for k, v in sorted(acc.items(), key=lambda kv: sum(kv[1])):
    print('  %-34s %.3f' % (k, sum(v) / len(v)))
