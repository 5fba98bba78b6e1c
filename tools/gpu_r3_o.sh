#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
KAI0_ATTN_RB64=1 python -m pytest tests/test_kernels_gpu.py tests/test_fullwidth_gpu.py -m gpu -q --tb=short -k "attention" 2>&1 | tail -3 | cut -c1-300
for S in 0 1 0 1; do echo -n "rb64=$S "; KAI0_ATTN_RB64=$S python tools/attn_fwd_bench.py 2>&1 | tail -1; done
