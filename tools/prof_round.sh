#!/bin/bash
# One profiling pass of a round: rocprofv3 kernel stats of the default bench (inference) and of the optimisation step.
# usage (through gpurun): bash tools/prof_round.sh <tag>   -> gpurun_out/<tag>_kernel_stats.md, <tag>_train_kernel_stats.md, <tag>_bench.json
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD; TAG=${1:-round}
python bench.py --steps 200 --warmup 10 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench.json
(cd /tmp && TNP_BENCH_PRIME_S=0.3 timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_i -o bench -- python $R/bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-traffic --no-train --no-sustain --no-strong > $R/gpurun_out/rocprof_i.log 2>&1); echo "rocprof inference rc=$?"
python tools/rocprof_summary.py gpurun_out/prof_i/*.db > gpurun_out/${TAG}_kernel_stats.md 2>&1; rm -rf gpurun_out/prof_i
(cd /tmp && TNP_BENCH_PRIME_S=0.3 timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_t -o bench -- python $R/bench.py --train --steps 20 --warmup 3 --no-cpu-baseline --no-traffic --no-roofline > $R/gpurun_out/rocprof_t.log 2>&1); echo "rocprof train rc=$?"
python tools/rocprof_summary.py gpurun_out/prof_t/*.db > gpurun_out/${TAG}_train_kernel_stats.md 2>&1; rm -rf gpurun_out/prof_t
head -12 gpurun_out/${TAG}_kernel_stats.md | cut -c1-170; head -30 gpurun_out/${TAG}_train_kernel_stats.md | cut -c1-170
