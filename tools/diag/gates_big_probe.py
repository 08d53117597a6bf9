"""Gates tile variants at LARGE batches (config 3: 256 x 64 = 16384 tracks, and 8192 / 4096): ms per directional forward."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from trajnetplusplusbaselines_amd import synth
cfg = bench.CONFIGS['directional']
model = bench.build_model(cfg, torch.device('cuda', 0)).eval()
for scenes in (64, 128, 256):
    xy, split = synth.linear_crowd(scenes, 64, seed=100)
    obs, goals = xy[:9].cuda(), torch.zeros(xy.shape[1], 2).cuda()
    row, ref = [], None
    for v in (0, 5, 6, 7, 21):
        model.kernel_variant = v << 8
        try:
            with torch.no_grad():
                for _ in range(5):
                    out = model(obs, goals, split, n_predict=12)[1]
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(20):
                    model(obs, goals, split, n_predict=12)
                b.record()
                torch.cuda.synchronize()
            ref = out if ref is None else ref
            row.append('v%d %.3f ms (d %.1e)' % (v, a.elapsed_time(b) / 20, float((torch.nan_to_num(out) - torch.nan_to_num(ref)).abs().max())))
        except Exception as e:
            row.append('v%d n/a (%s)' % (v, str(e)[:40]))
    print('%d x 64 (%d tracks): %s' % (scenes, xy.shape[1], '  '.join(row)), flush=True)
