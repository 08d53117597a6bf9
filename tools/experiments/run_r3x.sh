#!/bin/bash
# distributed legs of bench.py on one GPU (two ranks over gloo sharing the device): weak scaling of config 2 and strong scaling
# of config 3, plus config 3 at N = 1
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
python bench.py --steps 20 --warmup 5 --config directional --global-scenes 256 --no-cpu-baseline --no-traffic 2>/dev/null | tail -1 > gpurun_out/round3_t_bench_config3_strong_n1.json
TNP_BENCH_BACKEND=gloo TNP_BENCH_SHARE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-traffic 2>gpurun_out/weak2.err | tail -1 > gpurun_out/round3_t_bench_weak2_gloo_one_gpu.json; echo rc=$?
TNP_BENCH_BACKEND=gloo TNP_BENCH_SHARE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 2 --steps 5 --warmup 2 --config directional --global-scenes 256 --no-cpu-baseline --no-traffic 2>gpurun_out/strong2.err | tail -1 > gpurun_out/round3_t_bench_config3_strong2_gloo_one_gpu.json; echo rc=$?
python -c "
import json
for f in ('config3_strong_n1','weak2_gloo_one_gpu','config3_strong2_gloo_one_gpu'):
    d=json.load(open('gpurun_out/round3_t_bench_%s.json'%f)); print(f, round(d['value']), d['n_gpus'], d['scaling'], round(d['training']['ms_per_step'],3), d['training'].get('allreduce_bytes'), d['training'].get('scaling'))
"
tail -3 gpurun_out/weak2.err gpurun_out/strong2.err | cut -c1-200
