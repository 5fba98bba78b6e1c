#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3k; O=gpurun_out/r3k
python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=short -k "skinny or glue" 2>&1 | tail -3 | cut -c1-400
python -m pytest tests/test_model_gpu.py tests/test_fullwidth_gpu.py tests/test_fullsize_gpu.py -m gpu -q --tb=short -k "sample_actions or chunk or policy or graph" 2>&1 | tail -3 | cut -c1-400
AB="-;-" bash tools/infer_ab.sh
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_inf -o inf -- python tools/infer_once.py 5 1 > $O/infer_under_rocprof.log 2>&1
DB=$(find /tmp/prof_inf -name "*.db" | head -1)
python tools/infer_timeline.py $DB > $O/infer_timeline.txt 2>&1
head -20 $O/infer_timeline.txt | cut -c1-160
