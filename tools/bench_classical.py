"""Throughput of the batched classical rollouts at BASELINE config 5 (4096 scenes x 128 agents, 9 obs + 12 pred) next
to the host execution of the same arithmetic (oracle, OpenMP over scenes) on a bounded sample of scenes."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from trajnetplusplusbaselines_amd.classical import socialforce, orca, kalman
from tests.test_classical import crowd

S, A = 4096, 128
pos, vel, goals, speed, sizes = crowd(S, A, 11)
st = np.concatenate([pos, vel, goals], axis=1)
starts = np.concatenate([[0], np.cumsum(sizes)])
rng = np.random.RandomState(0)
t = np.arange(9)[None, :, None]
obs = pos[:, None, :] + vel[:, None, :] * 0.4 * (t - 8) + rng.randn(S * A, 9, 2) * 0.03
z = rng.standard_normal((S * A, 5, 13, 6))

def timeit(f, n=3):
    f(); torch.cuda.synchronize(); best = 1e9
    for _ in range(n):
        t0 = time.perf_counter(); f(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best

res = {}
res['socialforce'] = timeit(lambda: socialforce.rollout_batch(st, sizes))
res['orca'] = timeit(lambda: orca.rollout_batch(pos, vel, speed, goals, sizes))
res['kalman'] = timeit(lambda: kalman.predict_batch(obs, 12, noise=z))
sub = 128   # scenes of the CPU sample (bench.py's cpu_baseline leg runs the oracle)
cpu = bench.cpu_baseline_classical(st, pos, vel, goals, speed, obs, z, starts, A, sub, S)
for k in res:
    print(json.dumps(dict(predictor=k, scenes=S, agents=A, gpu_seconds_incl_pcie=round(res[k], 4),
                          gpu_scene_steps_per_s=round(S * 21 / res[k]), cpu_port_seconds_extrapolated=round(cpu[k], 2),
                          cpu_scene_steps_per_s=round(S * 21 / cpu[k]), cpu_cores=os.cpu_count(),
                          cpu_sample='%d of %d scenes' % (sub, S))))
