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
