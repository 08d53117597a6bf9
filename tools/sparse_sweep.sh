#!/bin/bash
# Sparse first-layer kernel at two crowd densities, per-kernel durations from rocprofv3 (variants: see
# launch_pool_embed_sparse; 8 / 9 are timing-only ablations).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD
for sp in 8 3; do for v in ${SPARSE_VARIANTS:-0 1 2 8 9}; do
  (cd /tmp && SPREAD=$sp TNP_SPARSE_VARIANT=$v timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/sprof_${sp}_$v -o sp -- python $R/tools/sparse_bench.py > $R/gpurun_out/sprof_${sp}_$v.log 2>&1)
  echo "== spread $sp variant $v: $(grep -o 'hits per ego [0-9.]*' gpurun_out/sprof_${sp}_$v.log) $(grep -o 'max err vs dense [0-9.e+-]*' gpurun_out/sprof_${sp}_$v.log)"
  python tools/rocprof_summary.py gpurun_out/sprof_${sp}_$v/sp_results.db 2>&1 | grep -i "sparse\|cellsplit\|reduce_kernelEPK" | cut -c1-150 | head -4
  rm -rf gpurun_out/sprof_${sp}_$v   # keep gpurun_out small: only the log lines travel back
done; done 2>&1 | tee gpurun_out/sparse_sweep.log
