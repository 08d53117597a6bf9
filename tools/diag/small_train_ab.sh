#!/bin/bash
# A/B of the batch_size-8 optimisation step's kernels under tile / split knobs: one rocprofv3 kernel trace per setting.
# usage (through gpurun): bash tools/diag/small_train_ab.sh [steps]   -> gpurun_out/small_train_ab.txt
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD; N=${1:-60}
OUT=gpurun_out/small_train_ab.txt; : > $OUT
run() {   # tag, env assignments...
    local tag=$1; shift
    rm -rf gpurun_out/prof_ab
    (cd /tmp && env "$@" timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_ab -o t -- python $R/tools/diag/small_batch_workload.py train $N > /dev/null 2>&1)
    echo "== $tag ($*)" >> $OUT
    python tools/rocprof_summary.py gpurun_out/prof_ab/*.db 2>&1 | grep -E "sparse_wgrad|wgrad_tn_group|wgrad_reduce_group|dgrid_cells|social_scatter|pool_embed_regacc|gemm_|total" | cut -c1-160 >> $OUT
    rm -rf gpurun_out/prof_ab
}
run defaults X=1
run round5_plans TNP_SPARSE_WGRAD_PLAN=65 TNP_WGRAD_MIN_ROWS=256
for e in "X=1" "TNP_BWD_SIDE_STREAM=1" "X=1" "TNP_BWD_SIDE_STREAM=1" "TNP_SPARSE_WGRAD_PLAN=65 TNP_WGRAD_MIN_ROWS=256"; do
    env $e TAG="$e" python tools/diag/small_batch_workload.py train_time 200 2>&1 | grep "wall ms" >> $OUT
done
cat $OUT
