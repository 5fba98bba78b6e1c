#!/bin/bash
# kernel-trace of the bench command (training) and of the action chunk; summaries into gpurun_out/summary
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/summary; mkdir -p $OUT
BENCH="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-latency"
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_train -o train -- $BENCH > $OUT/bench_under_rocprof.log 2>&1
python tools/prof_summary.py $(find /tmp/prof_train -name "*.db" | head -1) > $OUT/train_kernel_stats.md 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_inf -o inf -- python tools/infer_once.py 5 1 > $OUT/infer_under_rocprof.log 2>&1
python tools/prof_summary.py $(find /tmp/prof_inf -name "*.db" | head -1) > $OUT/infer_kernel_stats.md 2>&1
python tools/infer_timeline.py $(find /tmp/prof_inf -name "*.db" | head -1) > $OUT/infer_timeline.txt 2>&1
grep -h '"metric"' $OUT/*.log | cut -c1-400
