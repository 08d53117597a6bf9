#!/bin/bash
# rocprofv3 PMC passes (counters + kernel trace only) over one optimisation step (bench.py --train): SQ counters of the
# training-only kernels (dgrid_cells, wgrad_tn, sparse_wgrad).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; R=$PWD; mkdir -p gpurun_out
run() { (cd /tmp && TNP_BENCH_PRIME_S=0 timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmct_$N -o p -- python $R/bench.py --train --steps 2 --warmup 1 --no-cpu-baseline --no-traffic --no-roofline > $R/gpurun_out/pmct_$N.log 2>&1); }
N=1 run SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE
N=2 run FETCH_SIZE
N=3 run WRITE_SIZE
for n in 1 2 3; do echo "## pass $n"; python tools/pmc_summary.py gpurun_out/pmct_$n 'dgrid_cells|wgrad_tn|sparse_wgrad|wgrad_reduce|scatter_backward_cells8'; rm -rf gpurun_out/pmct_$n; done > gpurun_out/pmc_train.md 2>&1
cat gpurun_out/pmc_train.md
