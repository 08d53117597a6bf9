#!/bin/bash
# rocprofv3 PMC passes (counters + kernel trace only, separate passes) for one kernel of the inference bench.
# usage (through gpurun): bash tools/pmc_kernel.sh <kernel-regex> <out.md>
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; R=$PWD; mkdir -p gpurun_out; K=${1:-regacc}; OUT=${2:-gpurun_out/pmc_kernel.md}
run() { (cd /tmp && TNP_BENCH_PRIME_S=0 timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmck_$N -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-traffic --no-train > $R/gpurun_out/pmck_$N.log 2>&1); }
N=1 run SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM
N=2 run SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC
N=3 run FETCH_SIZE
N=4 run WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
for n in 1 2 3 4; do echo "## pass $n"; python tools/pmc_summary.py gpurun_out/pmck_$n "$K"; rm -rf gpurun_out/pmck_$n; done > $OUT 2>&1
cat $OUT
