#!/bin/bash
# tile shape of the backward data-gradient GEMM ([2048 x 512] . [448 x 512]^T + masked copies): TNP_DG_VARIANT (a knob of the
# working tree that ran this sweep only; variant 25 is now chosen automatically, gemm_f32_mfma.hip launch_linear)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r3v; export TMPDIR=/tmp; R=$PWD
for V in 0 25 26 12; do
  (cd /tmp && TNP_DG_VARIANT=$V TNP_BENCH_PRIME_S=0.3 timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_v -o bench -- python $R/bench.py --train --steps 10 --warmup 3 --no-cpu-baseline --no-traffic --no-roofline > $R/gpurun_out/r3v/rocprof_$V.log 2>&1)
  python tools/rocprof_summary.py gpurun_out/prof_v/*.db > gpurun_out/r3v/stats_$V.md 2>&1; rm -rf gpurun_out/prof_v
  echo "variant $V"; grep -E "gemm_nt" gpurun_out/r3v/stats_$V.md | cut -c1-75,100-165
done
