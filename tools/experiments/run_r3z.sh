#!/bin/bash
# cells per workgroup of dgrid_cells_xcd_kernel (TNP_DGRID_CPW: a knob of the working tree that ran this sweep only)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r3z; export TMPDIR=/tmp; R=$PWD
for V in 1 2 4; do
  TNP_DGRID_CPW=$V timeout 300 python -m pytest tests/test_gpu_training.py -m gpu -x -q 2>&1 | tail -1
  (cd /tmp && TNP_DGRID_CPW=$V TNP_BENCH_PRIME_S=0.3 timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_z -o bench -- python $R/bench.py --train --steps 10 --warmup 3 --no-cpu-baseline --no-traffic --no-roofline > $R/gpurun_out/r3z/rocprof_$V.log 2>&1)
  python tools/rocprof_summary.py gpurun_out/prof_z/*.db > gpurun_out/r3z/stats_$V.md 2>&1; rm -rf gpurun_out/prof_z
  echo "cells per workgroup $V: $(grep -E 'dgrid_cells_xcd' gpurun_out/r3z/stats_$V.md | cut -d'|' -f5-7)"
done
