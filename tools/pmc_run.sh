#!/bin/bash
# rocprofv3 PMC passes (counters only, no tracing domains) for the GEMM-1 shape; CSV output under gpurun_out/pmc*
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; R=$PWD
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc1 -o p -- python $R/tools/pmc_gemm.py 4 12 2 > $R/gpurun_out/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/pmc2 -o p -- python $R/tools/pmc_gemm.py 4 12 2 > $R/gpurun_out/pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc3 -o p -- python $R/tools/pmc_gemm.py 4 12 2 > $R/gpurun_out/pmc3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/pmc4 -o p -- python $R/tools/pmc_gemm.py 4 12 2 > $R/gpurun_out/pmc4.log 2>&1
ls $R/gpurun_out/pmc1 $R/gpurun_out/pmc2 | head
