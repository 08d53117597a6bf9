#!/bin/bash
# tile shape of the layer-1 backward GEMM ([2048 x 256] . [1024 x 256]^T, masked): TNP_L1B_VARIANT (a knob of the working tree
# that ran this sweep only)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r3y; export TMPDIR=/tmp; R=$PWD
for V in 0 24 25 26; do
  (cd /tmp && TNP_L1B_VARIANT=$V TNP_BENCH_PRIME_S=0.3 timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_y -o bench -- python $R/bench.py --train --steps 10 --warmup 3 --no-cpu-baseline --no-traffic --no-roofline > $R/gpurun_out/r3y/rocprof_$V.log 2>&1)
  python tools/rocprof_summary.py gpurun_out/prof_y/*.db > gpurun_out/r3y/stats_$V.md 2>&1; rm -rf gpurun_out/prof_y
  echo "variant $V"; grep -E "gemm_nt" gpurun_out/r3y/stats_$V.md | cut -c1-75,100-165
done
