#!/bin/bash
# kernel stats of one optimisation step of the non-grid interaction modules (LSTM --type attentionmlp / hiddenstatemlp)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r3aa; export TMPDIR=/tmp; R=$PWD
cat > /tmp/nongrid_train.py <<'PY'
import sys, torch
sys.path.insert(0, sys.argv[1])
from trajnetplusplusbaselines_amd import synth
from trajnetplusplusbaselines_amd.lstm import LSTM, HiddenStateMLPPooling, AttentionMLPPooling, PredictionLoss
from trajnetplusplusbaselines_amd.lstm.train_step import train_batch
from trajnetplusplusbaselines_amd.optim import Adam
kind = sys.argv[2]
pool = AttentionMLPPooling(hidden_dim=128, out_dim=128) if kind == 'attentionmlp' else HiddenStateMLPPooling(hidden_dim=128, out_dim=128)
dev = torch.device('cuda', 0)
torch.manual_seed(0)
model = LSTM(pool=pool).to(dev)
xy, split = synth.linear_crowd(64, 32, seed=100)
scene, goals = xy.to(dev), torch.zeros(xy.shape[1], 2, device=dev)
opt = Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
for _ in range(12):
    train_batch(model, opt, PredictionLoss(), scene, goals, split, 9, 12, batch_size=64)
torch.cuda.synchronize()
PY
for K in attentionmlp hiddenstatemlp; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_aa -o bench -- python /tmp/nongrid_train.py $R $K > $R/gpurun_out/r3aa/rocprof_$K.log 2>&1)
  python tools/rocprof_summary.py gpurun_out/prof_aa/*.db > gpurun_out/r3aa/stats_$K.md 2>&1; rm -rf gpurun_out/prof_aa
  echo "== $K"; head -16 gpurun_out/r3aa/stats_$K.md | tail -12 | cut -c1-80,100-170
done
