#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/summary; mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_inf -o inf -- python tools/infer_once.py 5 1 > $OUT/infer_under_rocprof.log 2>&1
python tools/prof_summary.py $(find /tmp/prof_inf -name "*.db" | head -1) > $OUT/infer_kernel_stats.md 2>&1
python tools/infer_timeline.py $(find /tmp/prof_inf -name "*.db" | head -1) > $OUT/infer_timeline.txt 2>&1
tail -1 $OUT/infer_under_rocprof.log
