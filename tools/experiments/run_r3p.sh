#!/bin/bash
# sparse_wgrad_mfma sweep: pipeline depth (TNP_SWG_U), waves per workgroup (TNP_SWG_NW), block map (TNP_SWG_MAP).
# History: the three environment knobs existed only in the working tree that ran this sweep (the library reads no environment
# variables); results in profiles/round3_p_sparse_wgrad_sweep.md.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r3p; export TMPDIR=/tmp; R=$PWD
run() {
  tag=$1; shift
  (cd /tmp && env "$@" TNP_BENCH_PRIME_S=0.3 timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_p -o bench -- python $R/bench.py --train --steps 10 --warmup 3 --no-cpu-baseline --no-traffic --no-roofline > $R/gpurun_out/r3p/rocprof_$tag.log 2>&1)
  python tools/rocprof_summary.py gpurun_out/prof_p/*.db > gpurun_out/r3p/train_stats_$tag.md 2>&1; rm -rf gpurun_out/prof_p
  echo "$tag: $(grep -E 'sparse_wgrad_mfma' gpurun_out/r3p/train_stats_$tag.md | cut -d'|' -f4-7)"
}
run base X=1
TNP_SWG_NW=8 timeout 600 python -m pytest tests/test_gpu_training.py -m gpu -x -q 2>&1 | tail -2
run nw8 TNP_SWG_NW=8
run nw16 TNP_SWG_NW=16
run nw8u2 TNP_SWG_NW=8 TNP_SWG_U=2
run map1 TNP_SWG_MAP=1
run map2 TNP_SWG_MAP=2
run map1nw8 TNP_SWG_MAP=1 TNP_SWG_NW=8
