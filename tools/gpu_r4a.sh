#!/bin/bash
# round 4, call A: attention tests + stored-P vs recompute A/B (several kernel variants, one process each)
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_fullwidth_gpu.py -m gpu -q -x -k "attention or attn" > gpurun_out/r4a_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r4a_pytest.log
for cfg in "default" "KAI0_ATTN_QT=2" "KAI0_SB_LDR=104" "KAI0_ATTN_ONEPASS_GRID=100000"; do
  echo "=== $cfg"
  if [ "$cfg" = default ]; then timeout 300 python tools/attn_r4_bench.py 2>&1 | tail -8
  elif [ "$cfg" = "KAI0_ATTN_QT=2" ]; then env $cfg ATTN_BENCH=gemma timeout 300 python tools/attn_r4_bench.py 2>&1 | tail -4
  else env $cfg ATTN_BENCH=siglip timeout 300 python tools/attn_r4_bench.py 2>&1 | tail -4; fi
done 2>&1 | tee gpurun_out/r4a_bench.log
