#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r3r; export TMPDIR=/tmp; R=$PWD
(cd /tmp && TNP_BENCH_PRIME_S=0.3 timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r -o bench -- python $R/bench.py --train --steps 20 --warmup 3 --no-cpu-baseline --no-traffic --no-roofline > $R/gpurun_out/r3r/rocprof.log 2>&1)
python tools/rocprof_gaps.py gpurun_out/prof_r/*.db > gpurun_out/r3r/gaps.md 2>&1; python tools/rocprof_summary.py gpurun_out/prof_r/*.db > gpurun_out/r3r/train_stats.md; rm -rf gpurun_out/prof_r
cat gpurun_out/r3r/gaps.md | head -40
