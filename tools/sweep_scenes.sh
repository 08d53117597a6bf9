#!/bin/bash
# Batch-size sweep of the headline workload (Social-LSTM n=16 two_layer 1024, 32 agents per scene): scenes 64 / 128 / 256 / 512
# per GPU with per-kernel times (rocprofv3 --kernel-trace) and the whole-step roofline -- separates "M = 2048 tracks cannot
# fill 256 CUs" from per-kernel inefficiency.  usage (through gpurun): bash tools/sweep_scenes.sh <tag> [shape]
#   -> gpurun_out/<tag>_sweep_scenes.md
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD; TAG=${1:-round}; VAR=${2:-0}
OUT=gpurun_out/${TAG}_sweep_scenes.md
echo "# scenes sweep (variant $VAR): bench.py --scenes S, 32 agents per scene, inference forward, 1 x MI355X" > $OUT
for S in 64 128 256 512; do
  python bench.py --scenes $S --steps 30 --warmup 5 --variant $VAR --no-cpu-baseline --no-traffic --no-train --no-sustain --no-strong --no-op-point 2>/dev/null | tail -1 > gpurun_out/${TAG}_sweep_$S.json
  (cd /tmp && TNP_BENCH_PRIME_S=0.3 timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_s$S -o bench -- python $R/bench.py --scenes $S --steps 30 --warmup 3 --variant $VAR --no-cpu-baseline --no-traffic --no-train --no-roofline --no-strong --no-op-point > $R/gpurun_out/rocprof_s$S.log 2>&1)
  echo >> $OUT; echo "## $S scenes ($((S*32)) tracks)" >> $OUT
  python - gpurun_out/${TAG}_sweep_$S.json >> $OUT <<'P'
import json, sys
d = json.load(open(sys.argv[1]))
r, s = d.get('roofline') or {}, d.get('step_roofline') or {}
print('%.0f scene-steps/s, %.3f ms per forward, %.1f us per recurrent step; dominant kernel %.1f us (frac %.3f, on hits %.3f); whole step: %.2f GFLOP -> %.1f TFLOP/s = %.3f of 157.3 (on hits %.3f)' % (
    d['value'], d['ms_per_step'], s.get('us_per_recurrent_step', 0), r.get('avg_launch_us', 0), r.get('frac', 0), r.get('frac_on_hits', 0),
    s.get('flops_per_recurrent_step', 0) / 1e9, s.get('achieved', 0), s.get('frac', 0), s.get('frac_on_hits', 0)))
print()
P
  python tools/rocprof_summary.py gpurun_out/prof_s$S/*.db 2>&1 | grep -v "^###" | head -8 | cut -c1-200 >> $OUT
  rm -rf gpurun_out/prof_s$S
done
cat $OUT | cut -c1-220
