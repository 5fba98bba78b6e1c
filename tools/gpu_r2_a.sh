#!/bin/bash
# round 2, GPU session A: the whole -m gpu suite (incl. the new full-width parity tests), the bench line with its new objects,
# the per-shape GEMM breakdown and the serve-path host share.
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -m pytest tests -m gpu -q -rP -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee gpurun_out/rc.txt
tail -5 gpurun_out/pytest_gpu.log
KAI0_GEMM_BREAKDOWN=1 timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/bench.log 2>&1; echo "bench rc=$?" | tee -a gpurun_out/rc.txt
tail -1 gpurun_out/bench.log | cut -c1-1500
timeout 300 python tools/policy_latency.py 30 > gpurun_out/policy_latency.log 2>&1; tail -1 gpurun_out/policy_latency.log
nproc; free -g | head -2
