#!/bin/bash
# Everything one gpurun call should collect: GPU tests, diagnostics, bench line, rocprofv3 kernel stats.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" | tee -a gpurun_out/summary.log
timeout 900 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/summary.log
tail -5 gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/bench.log 2>&1; echo "bench rc=$?" | tee -a gpurun_out/summary.log
tail -2 gpurun_out/bench.log
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof" -o bench -- python "$OLDPWD/bench.py" --steps 5 --warmup 2 --no-cpu-baseline > "$OLDPWD/gpurun_out/rocprof.log" 2>&1); echo "rocprof rc=$?" | tee -a gpurun_out/summary.log
python tools/rocprof_summary.py gpurun_out/prof/*.db > gpurun_out/kernel_stats.md 2>&1; head -14 gpurun_out/kernel_stats.md | cut -c1-170
