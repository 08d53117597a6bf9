#!/bin/bash
# rocprofv3 kernel stats of the training step (bench.py --train) -> gpurun_out/train_kernel_stats.md
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
(cd /tmp && TNP_BENCH_PRIME_S=0 timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_train" -o bench -- python "$OLDPWD/bench.py" --train --steps 4 --warmup 2 --no-cpu-baseline --no-traffic --no-roofline ${TRAIN_ARGS:-} > "$OLDPWD/gpurun_out/rocprof_train.log" 2>&1); echo "rocprof rc=$?"
python tools/rocprof_summary.py gpurun_out/prof_train/*.db > gpurun_out/train_kernel_stats.md 2>&1
rm -rf gpurun_out/prof_train
head -40 gpurun_out/train_kernel_stats.md | cut -c1-200
