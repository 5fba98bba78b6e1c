#!/bin/bash
# round 4, call C: batched-read attention kernels: tests + A/B
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_fullwidth_gpu.py -m gpu -q -x -k "attention or attn" > gpurun_out/r4c_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r4c_pytest.log
for cfg in "KAI0_ATTN_PIPE=1" "KAI0_ATTN_PIPE=0" "KAI0_ATTN_PIPE=1 KAI0_ATTN_QT=2" "KAI0_ATTN_PIPE=0 KAI0_ATTN_QT=2"; do
  echo "=== $cfg"; env $cfg ATTN_BENCH=gemma timeout 300 python tools/attn_r4_bench.py 2>&1 | grep "gemma"
done 2>&1 | tee gpurun_out/r4c_bench.log
