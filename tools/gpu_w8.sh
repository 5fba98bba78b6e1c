#!/bin/bash
# round 6: the eight-wave 128 x 128 tile and the two-blocks-per-CU A/B configurations
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py -q -x -k "eight_wave or rope_epilogue or gemm_nt or gemm_nn or gemm_tn" > gpurun_out/pytest_w8.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_w8.log
timeout 900 python tools/gemm_cfg_ab.py > gpurun_out/gemm_cfg_ab.txt 2>&1; tail -40 gpurun_out/gemm_cfg_ab.txt
AB="KAI0_GEMM_W8=0;-" bash tools/infer_ab.sh
