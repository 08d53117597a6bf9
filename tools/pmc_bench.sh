#!/bin/bash
# rocprofv3 PMC passes (counters + kernel trace only) over the bench's inference forward: per-kernel SQ counters.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; R=$PWD; mkdir -p gpurun_out
run() { (cd /tmp && TNP_BENCH_PRIME_S=0 timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmcb_$N -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-strong --no-train --no-sustain --no-traffic --no-roofline --no-op-point > $R/gpurun_out/pmcb_$N.log 2>&1); }
N=1 run SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU
N=2 run SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD
for n in 1 2; do echo "## pass $n"; python tools/pmc_summary.py gpurun_out/pmcb_$n 'gemm_nt_pipe|track_prepare|grid_build'; rm -rf gpurun_out/pmcb_$n; done > gpurun_out/pmc_bench.md 2>&1
cat gpurun_out/pmc_bench.md
