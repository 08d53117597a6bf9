#!/bin/bash
# Everything that goes into profiles/ for one state of a round.  usage (through gpurun): bash tools/round_evidence.sh <tag>
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp; TAG=${1:-round}
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.txt 2>&1; echo "smoke rc=$?"
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -3 > gpurun_out/${TAG}_pytest_gpu.txt; cat gpurun_out/${TAG}_pytest_gpu.txt
bash tools/prof_round.sh ${TAG} > gpurun_out/${TAG}_prof.log 2>&1; tail -3 gpurun_out/${TAG}_prof.log | cut -c1-150
python bench.py --config classical --steps 3 --warmup 1 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_classical.json
python bench.py --train --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_train_social.json
python bench.py --train --config directional --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_train_directional.json
python bench.py --train --config sgan --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_train_sgan.json
python bench.py --config sgan --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_sgan.json
for f in gpurun_out/${TAG}_bench*.json; do python - "$f" <<'P'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[1].split('/')[-1], round(d['value']), 'scene-steps/s', round(d['ms_per_step'], 3), 'ms', (d.get('roofline') or {}).get('frac'), (d.get('training') or {}).get('ms_per_step'))
P
done
# round 6: the small-batch regime (one scene per call, batch_size 8) and the evaluator feed
R=$PWD
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_sp -o t -- python $R/tools/diag/small_batch_workload.py predict 60 > /dev/null 2>&1)
python tools/rocprof_summary.py gpurun_out/prof_sp/*.db > gpurun_out/${TAG}_small_predict_kernel_stats.md 2>&1; rm -rf gpurun_out/prof_sp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_st -o t -- python $R/tools/diag/small_batch_workload.py train 60 > /dev/null 2>&1)
python tools/rocprof_summary.py gpurun_out/prof_st/*.db > gpurun_out/${TAG}_small_train_kernel_stats.md 2>&1
python tools/diag/step_timeline.py gpurun_out/prof_st/*.db 3 > gpurun_out/${TAG}_small_train_step_timeline.txt 2>&1; rm -rf gpurun_out/prof_st
python tools/diag/small_batch_workload.py train_time 200 2>&1 | grep "wall ms" > gpurun_out/${TAG}_small_train_wall.txt
python tools/diag/predict_dataset_throughput.py 1024 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_predict_dataset_throughput.txt
python tools/diag/small_batch_workload.py train_timeline 300 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_train_host_timeline.txt
python tests/parity_margin.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_parity_margin.txt
head -8 gpurun_out/${TAG}_small_predict_kernel_stats.md | cut -c1-150; cat gpurun_out/${TAG}_predict_dataset_throughput.txt
