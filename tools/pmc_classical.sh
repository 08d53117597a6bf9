#!/bin/bash
# rocprofv3 PMC passes (counters + kernel trace only) over the classical rollouts (bench.py --config classical): SQ counters of
# sf_rollout_kernel / orca_rollout_kernel / kalman_kernel, and their kernel times.  usage (through gpurun): bash tools/pmc_classical.sh <tag>
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; R=$PWD; mkdir -p gpurun_out; TAG=${1:-round}
run() { (cd /tmp && TNP_BENCH_PRIME_S=0 timeout 400 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmcc_$N -o p -- python $R/bench.py --config classical --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmcc_$N.log 2>&1); }
N=1 run SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
N=2 run SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_INST_CYCLES_VMEM_RD
{ echo "# PMC passes over bench.py --config classical (4096 scenes x 128 agents), mean per launch"; for n in 1 2; do echo; echo "## pass $n"; python tools/pmc_summary.py gpurun_out/pmcc_$n 'rollout|kalman'; rm -rf gpurun_out/pmcc_$n; done; } > gpurun_out/${TAG}_pmc_classical.md 2>&1
python bench.py --config classical --steps 3 --warmup 1 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_classical.json
cat gpurun_out/${TAG}_pmc_classical.md | cut -c1-160; python -c "
import json; d=json.load(open('gpurun_out/${TAG}_bench_classical.json')); print(d['config']['ms_per_predictor'], d['roofline']['frac'], d['value'])"
