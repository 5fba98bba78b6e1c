#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3e
python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=short -k "geglu" 2>&1 | tail -4 | cut -c1-300
AB="KAI0_GEGLU_PAIR=0;-;KAI0_GEGLU_PAIR=0;-" STEPS=8 BENCH_ARGS="--no-trim-extra" bash tools/gpu_ab.sh
