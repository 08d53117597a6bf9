"""cProfile of LSTMPredictor.__call__ on single scenes (bench.py's per_scene_predict leg)."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from trajnetplusplusbaselines_amd import _lib, synth, data as trajdata
from trajnetplusplusbaselines_amd.lstm import LSTMPredictor

dev = torch.device('cuda', 0)
model = bench.build_model(bench.CONFIGS['social'], dev, seed=1).eval()
p = LSTMPredictor(model)
xy, split = synth.ragged_crowd(256, 8, 72, seed=2024, nan_frac=0.2)
xy, split = xy.numpy(), split.numpy()
scenes = [xy[:, split[i]:split[i + 1]] for i in range(128)]
paths = [trajdata.xy_to_paths(sc) for sc in scenes]
goals = [np.zeros((sc.shape[1], 2)) for sc in scenes]
for a, g in zip(paths[:8], goals[:8]):
    p(a, g, n_predict=12)
for rep in range(2):
    _lib.SceneIndex._cache.clear()
    t0 = time.perf_counter()
    for a, g in zip(paths, goals):
        p(a, g, n_predict=12)
    print('pass %d: %.3f ms per call' % (rep, (time.perf_counter() - t0) / len(paths) * 1e3))
_lib.SceneIndex._cache.clear()
pr = cProfile.Profile(); pr.enable()
for a, g in zip(paths, goals):
    p(a, g, n_predict=12)
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(16)
