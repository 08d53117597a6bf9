"""Which ATen ops run in one training step besides our HIP kernels, and how long the host spends per phase
(measurement tool: python tools/torch_ops_train.py [social|directional])."""
import sys, time, torch
sys.path.insert(0, '.')
import bench
from torch.profiler import profile, ProfilerActivity
from trajnetplusplusbaselines_amd import synth
from trajnetplusplusbaselines_amd.lstm import PredictionLoss
from trajnetplusplusbaselines_amd.lstm.train_step import train_batch

cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else 'directional']
dev = torch.device('cuda', 0)
model = bench.build_model(cfg, dev)
xy, split = synth.linear_crowd(cfg['scenes'], cfg['agents'], seed=100)
scene = xy.to(dev)
goals = torch.zeros(xy.shape[1], 2, device=dev)
opt = bench.make_adam(model.parameters()) if hasattr(bench, 'make_adam') else torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
crit = PredictionLoss()
for _ in range(3):
    train_batch(model, opt, crit, scene, goals, split, 9, 12, batch_size=cfg['scenes'])
# host time per phase (enqueue only) and total with sync
model.train()
def phases():
    t = [time.perf_counter()]
    observed, truth = scene[:9].clone(), scene[9:-1].clone()
    targets = scene[9:21] - scene[8:20]
    rel, out = model(observed, goals, split, truth); t.append(time.perf_counter())
    loss = crit(rel[-12:], targets, split) * cfg['scenes']; t.append(time.perf_counter())
    opt.zero_grad(); loss.backward(); t.append(time.perf_counter())
    opt.step(); t.append(time.perf_counter())
    torch.cuda.synchronize(); t.append(time.perf_counter())
    return [1e3 * (b - a) for a, b in zip(t[:-1], t[1:])]
torch.cuda.synchronize()
acc = [phases() for _ in range(5)][1:]
print('host ms  forward %.2f  loss %.2f  backward %.2f  adam %.2f  wait-for-gpu %.2f' % tuple(sum(c) / len(c) for c in zip(*acc)))
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    train_batch(model, opt, crit, scene, goals, split, 9, 12, batch_size=cfg['scenes'])
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by='self_cuda_time_total', row_limit=45, max_name_column_width=60))

# every ATen op that launches something, with the innermost frame of this repository that issued it
import collections
seen = collections.Counter()
for ev in prof.events():
    if ev.device_time_total > 0 and ev.name.startswith('aten::') and ev.cpu_parent is not None and not ev.cpu_parent.name.startswith('aten::'):
        where = [f for f in (ev.stack or []) if 'trajnetplusplusbaselines_amd' in f or 'bench.py' in f]
        seen[(ev.name, where[0].split('/')[-1] if where else '?')] += 1
for (name, where), n in sorted(seen.items(), key=lambda kv: kv[0][1]):
    print('%3d x %-28s %s' % (n, name, where))
