#!/bin/bash
# kernel stats of the inference forward of the remaining non-grid types (nn, nn_lstm, traj_pool) at the config-2 batch
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r3ab; export TMPDIR=/tmp; R=$PWD
cat > /tmp/nongrid_fwd.py <<'PY'
import sys, torch
sys.path.insert(0, sys.argv[1])
from trajnetplusplusbaselines_amd import synth
from trajnetplusplusbaselines_amd.lstm import LSTM, NearestNeighborMLP, NearestNeighborLSTM, TrajectronPooling
kind = sys.argv[2]
pool = {'nn': lambda: NearestNeighborMLP(n=4, out_dim=32), 'nn_lstm': lambda: NearestNeighborLSTM(n=4, hidden_dim=256, out_dim=32),
        'traj_pool': lambda: TrajectronPooling(hidden_dim=256, out_dim=32)}[kind]()
dev = torch.device('cuda', 0)
torch.manual_seed(0)
model = LSTM(pool=pool).to(dev).eval()
xy, split = synth.linear_crowd(64, 32, seed=100)
scene, goals = xy.to(dev), torch.zeros(xy.shape[1], 2, device=dev)
with torch.no_grad():
    for _ in range(30):
        model(scene[:9], goals, split, n_predict=12)
torch.cuda.synchronize()
PY
for K in nn nn_lstm traj_pool; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_ab -o bench -- python /tmp/nongrid_fwd.py $R $K > $R/gpurun_out/r3ab/rocprof_$K.log 2>&1)
  python tools/rocprof_summary.py gpurun_out/prof_ab/*.db > gpurun_out/r3ab/stats_$K.md 2>&1; rm -rf gpurun_out/prof_ab
  echo "== $K"; head -11 gpurun_out/r3ab/stats_$K.md | tail -7 | cut -d'|' -f2,3,5 | cut -c1-110
done
