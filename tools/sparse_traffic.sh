#!/bin/bash
# duration + FETCH_SIZE / WRITE_SIZE of the sparse first-layer kernel for a list of TNP_SPARSE_VARIANTs (config-2 shape)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; R=$PWD; mkdir -p gpurun_out
for v in ${SPARSE_VARIANTS:-0 6 7}; do
  (cd /tmp && SPREAD=8 TNP_SPARSE_VARIANT=$v timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/st_t_$v -o sp -- python $R/tools/sparse_bench.py > $R/gpurun_out/st_t_$v.log 2>&1)
  echo "== variant $v $(grep -o 'max err vs dense [0-9.e+-]*' gpurun_out/st_t_$v.log)"
  python tools/rocprof_summary.py gpurun_out/st_t_$v/sp_results.db 2>&1 | grep -i "cellsplit\|sparse_kernel" | cut -c1-150 | head -2
  rm -rf gpurun_out/st_t_$v
  for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && SPREAD=8 TNP_SPARSE_VARIANT=$v timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/st_p_$v -o p -- python $R/tools/sparse_bench.py > /dev/null 2>&1)
    python tools/pmc_summary.py gpurun_out/st_p_$v 'cellsplit|sparse_kernel' 2>&1 | tail -3
    rm -rf gpurun_out/st_p_$v
  done
done 2>&1 | tee gpurun_out/sparse_traffic.log
