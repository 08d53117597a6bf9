set -x
timeout 600 python -m pytest tests/test_classical.py tests/test_gpu_optim.py -m gpu -x -q 2>&1 | tail -3
bash tools/pmc_classical.sh r3d
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-sustain 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('train', d['training'])"
TNP_BENCH_TORCH_ADAM=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-sustain 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('train torch adam', d['training']['ms_per_step'])"
