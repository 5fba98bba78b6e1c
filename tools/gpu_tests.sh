#!/bin/bash
# the whole GPU suite (PYTEST_ARGS narrows it); log under gpurun_out/
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
t0=$(date +%s)
timeout ${PYTEST_TIMEOUT:-3000} python -m pytest tests -m gpu -q ${PYTEST_ARGS:-} --durations=15 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$? in $(( $(date +%s) - t0 )) s"
grep -E "passed|failed|error" gpurun_out/pytest_gpu.log | tail -5
grep -E "^FAILED|^ERROR" gpurun_out/pytest_gpu.log | head -20
