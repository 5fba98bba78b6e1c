#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=short -k "gemm or mlp or linear" 2>&1 | tail -3 | cut -c1-300
python -m pytest tests/test_model_gpu.py tests/test_fullwidth_gpu.py -m gpu -q --tb=short -k "sample_actions or chunk or loss or reference_executed" 2>&1 | tail -3 | cut -c1-300
AB="-;-" bash tools/infer_ab.sh
AB="-" STEPS=8 BENCH_ARGS="--no-trim-extra --no-latency" bash tools/gpu_ab.sh
