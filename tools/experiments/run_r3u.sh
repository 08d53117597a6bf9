#!/bin/bash
# per-rank shape of BASELINE config 3 at 8 GPUs (32 scenes x 64 agents, directional): kernel stats of inference and training
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r3u; export TMPDIR=/tmp; R=$PWD
(cd /tmp && TNP_BENCH_PRIME_S=0.3 timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_u -o bench -- python $R/bench.py --config directional --scenes 32 --steps 100 --warmup 5 --no-cpu-baseline --no-traffic --no-train --no-sustain > $R/gpurun_out/r3u/rocprof_i.log 2>&1)
python tools/rocprof_summary.py gpurun_out/prof_u/*.db > gpurun_out/r3u/dir_kernel_stats.md 2>&1; rm -rf gpurun_out/prof_u
head -12 gpurun_out/r3u/dir_kernel_stats.md | cut -c1-70,100-170
tail -1 gpurun_out/r3u/rocprof_i.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('inference', d['value'], d['ms_per_step'])"
(cd /tmp && TNP_BENCH_PRIME_S=0.3 timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_u -o bench -- python $R/bench.py --config directional --scenes 32 --train --steps 20 --warmup 3 --no-cpu-baseline --no-traffic --no-roofline > $R/gpurun_out/r3u/rocprof_t.log 2>&1)
python tools/rocprof_summary.py gpurun_out/prof_u/*.db > gpurun_out/r3u/dir_train_kernel_stats.md 2>&1; rm -rf gpurun_out/prof_u
head -26 gpurun_out/r3u/dir_train_kernel_stats.md | cut -c1-70,100-170
python bench.py --config directional --scenes 32 --train --steps 30 --warmup 5 --no-cpu-baseline --no-traffic --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train ms_per_step', d['ms_per_step'])"
