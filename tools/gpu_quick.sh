#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_sparse.py tests/test_gpu_lstm.py tests/test_gpu_grid.py -q -m gpu --tb=line -p no:cacheprovider -x -q > gpurun_out/pytest_quick.log 2>&1; echo "pytest rc=$?"
tail -4 gpurun_out/pytest_quick.log
timeout 600 python tools/gpu_check.py > gpurun_out/gpu_check.log 2>&1; echo "gpu_check rc=$?"
