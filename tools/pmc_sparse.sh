#!/bin/bash
# rocprofv3 PMC passes (counters + kernel trace only) on the sparse first-layer kernel at the config-2 shape.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; R=$PWD; mkdir -p gpurun_out
run() { (cd /tmp && SPREAD=${SPREAD:-8} timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmcs_$N -o p -- python $R/tools/sparse_bench.py > $R/gpurun_out/pmcs_$N.log 2>&1); }
N=1 run SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM
N=2 run SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC
N=3 run FETCH_SIZE
N=4 run WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
N=5 run TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE
for n in 1 2 3 4 5; do echo "## pass $n"; python tools/pmc_summary.py gpurun_out/pmcs_$n 'cellsplit|sparse_kernel'; tail -2 gpurun_out/pmcs_$n.log | cut -c1-200; rm -rf gpurun_out/pmcs_$n; done > gpurun_out/pmc_sparse.md 2>&1
cat gpurun_out/pmc_sparse.md
