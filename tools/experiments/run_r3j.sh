set -x
bash tools/round_evidence.sh round3_j > gpurun_out/round3_j_evidence.log 2>&1; tail -12 gpurun_out/round3_j_evidence.log
bash tools/pmc_kernel.sh regacc gpurun_out/round3_j_pmc_regacc.md > /dev/null 2>&1
bash tools/sweep_scenes.sh round3_j 0 > gpurun_out/round3_j_sweep.log 2>&1; grep "scene-steps/s," gpurun_out/round3_j_sweep_scenes.md
python tools/bench_types.py > gpurun_out/round3_j_bench_types.jsonl 2>/dev/null; cat gpurun_out/round3_j_bench_types.jsonl | cut -c1-160
python bench.py --steps 20 --warmup 5 --config directional --global-scenes 256 --no-cpu-baseline --no-traffic 2>/dev/null | tail -1 > gpurun_out/round3_j_bench_config3_strong_n1.json
TNP_BENCH_BACKEND=gloo TNP_BENCH_SHARE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-traffic 2>/dev/null | tail -1 > gpurun_out/round3_j_bench_weak2_gloo_one_gpu.json; echo rc=$?
python -c "
import json
for f in ('config3_strong_n1','weak2_gloo_one_gpu'):
    d=json.load(open('gpurun_out/round3_j_bench_%s.json'%f)); print(f, d['value'], d['n_gpus'], d['scaling'], d['training']['ms_per_step'], d['training'].get('allreduce_bytes'))
"
