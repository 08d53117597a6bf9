"""bench.py's operating-point legs alone (trainer_default, per_scene_predict): python tools/diag/op_point.py [steps]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
out = bench.operating_point_legs(torch.device('cuda', 0), steps=int(sys.argv[1]) if len(sys.argv) > 1 else 200)
for k, v in out.items():
    print(k, json.dumps({a: b for a, b in v.items() if a not in ('workload', 'note', 'cpu_port_note')}))
