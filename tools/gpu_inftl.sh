#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/summary; mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_inf -o inf -- python tools/infer_once.py 5 1 > $OUT/infer_under_rocprof.log 2>&1
python tools/infer_timeline.py $(find /tmp/prof_inf -name "*.db" | head -1) > $OUT/infer_timeline.txt 2>&1
head -34 $OUT/infer_timeline.txt
