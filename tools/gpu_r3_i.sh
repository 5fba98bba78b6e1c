#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=short -k "rope" 2>&1 | tail -3 | cut -c1-400
python -m pytest tests/test_model_gpu.py tests/test_fullwidth_gpu.py -m gpu -q --tb=short -k "sample_actions or chunk" 2>&1 | tail -3 | cut -c1-400
AB="-;-" bash tools/infer_ab.sh
