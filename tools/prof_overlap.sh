#!/bin/bash
# kernel trace of a few training steps WITH the action expert's second stream on: how much runs concurrently
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/summary; mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_ov -o ov -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-latency --no-gemm-timing --no-trim-extra > $OUT/overlap_under_rocprof.log 2>&1
python tools/overlap_summary.py $(find /tmp/prof_ov -name "*.db" | head -1) 3 > $OUT/overlap_summary.txt 2>&1
cat $OUT/overlap_summary.txt
