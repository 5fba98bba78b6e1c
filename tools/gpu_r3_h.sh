#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=short -k "glue" 2>&1 | tail -6 | cut -c1-400
python -m pytest tests/test_model_gpu.py tests/test_fullwidth_gpu.py tests/test_fullsize_gpu.py tests/test_fulldepth_gpu.py -m gpu -q --tb=short -k "sample_actions or chunk or policy or graph" 2>&1 | tail -5 | cut -c1-400
AB="KAI0_INFER_GLUE=0;-;KAI0_INFER_GLUE=0;-" bash tools/infer_ab.sh
