#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3d; O=gpurun_out/r3d
python -m pytest tests/test_model_gpu.py -m gpu -q --tb=line -k "train_loop" -s > $O/pytest_loop.log 2>&1; echo "loop rc=$?"; tail -15 $O/pytest_loop.log | cut -c1-400
python -m pytest tests -m gpu -q --tb=short --deselect tests/test_model_gpu.py::test_train_loop_debug_pi05_resume_is_exact > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_gpu.log | cut -c1-300
