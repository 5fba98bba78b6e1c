#!/bin/bash
# round 4, call E: folded adaRMS in the denoise loop: kernel test, inference parity tests, chunk latency A/B
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "skinny or glue" -s 2>&1 | grep -E "folded|passed|failed|Error" | tail -5
timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_fullwidth_gpu.py tests/test_fullsize_gpu.py tests/test_fulldepth_gpu.py -m gpu -q -x -k "chunk or sample or infer or policy or graph or action" 2>&1 | tail -4
for cfg in "X=0" "KAI0_INFER_FOLD=0"; do echo "=== $cfg"; env $cfg timeout 300 python tools/infer_bench.py 2>&1 | tail -1 | cut -c1-260; done
