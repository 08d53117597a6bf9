#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r3q; export TMPDIR=/tmp; R=$PWD
timeout 900 python -m pytest tests/test_gpu_training.py tests/test_gpu_sgan.py -m gpu -x -q 2>&1 | tail -3
(cd /tmp && TNP_BENCH_PRIME_S=0.3 timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_q -o bench -- python $R/bench.py --train --steps 20 --warmup 3 --no-cpu-baseline --no-traffic --no-roofline > $R/gpurun_out/r3q/rocprof.log 2>&1)
python tools/rocprof_summary.py gpurun_out/prof_q/*.db > gpurun_out/r3q/train_stats.md 2>&1; rm -rf gpurun_out/prof_q
head -24 gpurun_out/r3q/train_stats.md | cut -c1-60,100-175
python bench.py --train --steps 30 --warmup 5 --no-cpu-baseline --no-traffic --no-roofline 2>/dev/null | tail -1 > gpurun_out/r3q/bench_train.json
python -c "
import json; d=json.loads(open('gpurun_out/r3q/bench_train.json').read()); print(d['training'])"
