#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r3t; export TMPDIR=/tmp; R=$PWD
(cd /tmp && TNP_BENCH_PRIME_S=0.3 timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_t2 -o bench -- python $R/bench.py --train --config sgan --steps 10 --warmup 3 --no-cpu-baseline --no-traffic --no-roofline > $R/gpurun_out/r3t/rocprof.log 2>&1)
python tools/rocprof_summary.py gpurun_out/prof_t2/*.db > gpurun_out/r3t/sgan_train_stats.md 2>&1; rm -rf gpurun_out/prof_t2
head -30 gpurun_out/r3t/sgan_train_stats.md | cut -c1-70,100-170
python bench.py --train --config sgan --steps 10 --warmup 3 --no-cpu-baseline --no-traffic --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'])"
