#!/bin/bash
# round 4, call B: SigLIP forward kernel tests + A/B (training shape and the B = 1 chunk)
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "siglip" > gpurun_out/r4b_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r4b_pytest.log
for cfg in "X=0" "KAI0_SF_ROWS=128" "KAI0_SIGLIP_FWD=general"; do
  echo "=== $cfg"; env $cfg ATTN_BENCH=siglip timeout 300 python tools/attn_r4_bench.py 2>&1 | grep "recompute (r4)"
done 2>&1 | tee gpurun_out/r4b_bench.log
for cfg in "X=0" "KAI0_SIGLIP_FWD=general"; do
  echo "=== infer $cfg"; env $cfg timeout 300 python tools/infer_bench.py 2>&1 | tail -2
done 2>&1 | tee -a gpurun_out/r4b_bench.log
