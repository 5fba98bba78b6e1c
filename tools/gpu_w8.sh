set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py -q -x -k "eight_wave or simple_epilogue or persistent_gemm or gemm_nt" > gpurun_out/pytest_w8.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_w8.log
python tools/probes/small_gemm_w8.py > gpurun_out/small_gemm_w8.txt 2>&1; cat gpurun_out/small_gemm_w8.txt | tail -12
AB="KAI0_GEMM_W8=1;-;KAI0_GEMM_W8=1;-" bash tools/infer_ab.sh
