#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_kernels_gpu.py tests/test_fullwidth_gpu.py tests/test_model_gpu.py -m gpu -q --tb=short -k "attention or estimator or forward_loss" 2>&1 | tail -3 | cut -c1-300
for S in 0 1 0 1; do echo -n "kc_lds=$S "; KAI0_ATTN_KC_LDS=$S python tools/attn_fwd_bench.py 2>&1 | tail -1; done
for S in 0 1; do echo -n "kc_lds=$S "; KAI0_ATTN_KC_LDS=$S python tools/siglip_attn_bench.py 2>&1 | tail -1; done
for A in 1 2 3; do echo -n "kc_lds=1 ablate=$A "; KAI0_ATTN_ABLATE=$A python tools/attn_fwd_bench.py 2>&1 | tail -1; done
