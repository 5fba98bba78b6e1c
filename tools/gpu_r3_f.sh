#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=short -k "geglu or attention" 2>&1 | tail -3 | cut -c1-300
for A in 0 1 2 4 8 16 3 7 31; do echo -n "ablate=$A "; KAI0_ATTN_ABLATE=$A python tools/attn_fwd_bench.py 2>&1 | tail -1; done
echo -n "QT=2 "; KAI0_ATTN_QT=2 python tools/attn_fwd_bench.py 2>&1 | tail -1
