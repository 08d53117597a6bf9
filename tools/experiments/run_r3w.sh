#!/bin/bash
# workgroups per contraction of the weight-gradient launch (TNP_WGRAD_WGS: a knob of the working tree that ran this sweep only)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r3w; export TMPDIR=/tmp; R=$PWD
for V in 256 384 512 768 1024 2048; do
  (cd /tmp && TNP_WGRAD_WGS=$V TNP_BENCH_PRIME_S=0.3 timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_w -o bench -- python $R/bench.py --train --steps 10 --warmup 3 --no-cpu-baseline --no-traffic --no-roofline > $R/gpurun_out/r3w/rocprof_$V.log 2>&1)
  python tools/rocprof_summary.py gpurun_out/prof_w/*.db > gpurun_out/r3w/stats_$V.md 2>&1; rm -rf gpurun_out/prof_w
  echo "target $V: $(grep -E 'wgrad_tn_group|wgrad_reduce_group' gpurun_out/r3w/stats_$V.md | cut -d'|' -f5 | tr '\n' ' ')"
done
