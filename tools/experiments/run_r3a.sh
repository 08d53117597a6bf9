set -x
mkdir -p gpurun_out/r3a
timeout 300 tools/experiments/sparse_ablate 10.0 3 > gpurun_out/r3a/ablate_t10.txt 2>&1
timeout 300 tools/experiments/sparse_ablate 3.0 3 > gpurun_out/r3a/ablate_t3.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_sparse.py tests/test_gpu_lstm.py tests/test_gpu_sgan.py tests/test_gpu_grid.py tests/test_gpu_padding.py -m gpu -x -q > gpurun_out/r3a/pytest_a.txt 2>&1
for v in 262144 524288 786432; do
  timeout 300 python bench.py --steps 50 --warmup 5 --variant $v --no-cpu-baseline --no-traffic --no-train > gpurun_out/r3a/bench_v$v.json 2> gpurun_out/r3a/bench_v$v.err
done
tail -3 gpurun_out/r3a/pytest_a.txt
grep -h "us$\|identical\|differ\|phase" gpurun_out/r3a/ablate_t10.txt | head -60
for v in 262144 524288 786432; do python -c "
import json,sys
d=json.loads(open('gpurun_out/r3a/bench_v$v.json').read().strip().splitlines()[-1])
print($v, d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])
"; done
