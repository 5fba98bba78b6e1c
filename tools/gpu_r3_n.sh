#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=short -k "norm or adarms" 2>&1 | tail -3 | cut -c1-300
python -m pytest tests/test_model_gpu.py -m gpu -q --tb=short -k "reference_executed or forward_loss or sample_actions" 2>&1 | tail -3 | cut -c1-300
AB="-;-" bash tools/infer_ab.sh
