#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_kernels_gpu.py tests/test_fullwidth_gpu.py tests/test_model_gpu.py -m gpu -q --tb=short -k "attention or estimator or forward_loss or policy" 2>&1 | tail -3 | cut -c1-300
for i in 1 2; do python tools/attn_fwd_bench.py 2>&1 | tail -1; done
python tools/siglip_attn_bench.py 2>&1 | tail -1
python tools/attn_bwd_bench.py 2>&1 | tail -2
