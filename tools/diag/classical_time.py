"""Device time of the three classical rollouts at BASELINE config 5 (bench.classical_leg), one line."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
r = bench.classical_leg(torch.device('cuda', 0))
print(json.dumps(r['ms_per_predictor']), r['finite'])
