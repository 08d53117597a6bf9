"""LSTM-gates tile variants (tnp_lstm_model.variant bits 8-15) at larger batches: whole-forward time of the headline model at
128 / 256 scenes x 32 agents and of the config-3 model (directional n=12 one_layer) at 256 x 64."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from trajnetplusplusbaselines_amd import synth


def timeit(f, n=40):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


dev = torch.device('cuda', 0)
SHARD = len(sys.argv) > 1 and sys.argv[1] == 'shard'      # one of eight shards of config 3 (32 x 64) and of config 2 x 4 (32 x 32)
CASES = (('directional', 32, 64), ('social', 32, 32), ('social', 64, 32)) if SHARD else (('social', 128, 32), ('social', 256, 32), ('directional', 256, 64))
VARIANTS = (0, 21, 22, 30, 31, 32, 33, 34) if SHARD else (0, 5, 21, 22)
for cfgname, scenes, agents in CASES:
    cfg = bench.CONFIGS[cfgname]
    model = bench.build_model(cfg, dev, seed=1).eval()
    xy, split = synth.linear_crowd(scenes, agents, seed=3)
    obs, goals = xy[:9].cuda(), torch.zeros(xy.shape[1], 2).cuda()
    row = []
    for gv in VARIANTS:
        model.kernel_variant = gv << 8
        with torch.no_grad():
            row.append('v%d %.3f ms' % (gv, timeit(lambda: model(obs, goals, split, n_predict=12), 200 if SHARD else 40)))
    print('%-12s %4d x %2d (M = %5d): %s' % (cfgname, scenes, agents, scenes * agents, '  '.join(row)), flush=True)
