#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/summary; mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_an -o an -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-latency --no-gemm-timing --no-trim-extra > $OUT/anatomy_under_rocprof.log 2>&1
DB=$(find /tmp/prof_an -name "*.db" | head -1)
python tools/step_anatomy.py $DB > $OUT/step_anatomy.txt 2>&1
python tools/prof_summary.py $DB > $OUT/anatomy_kernel_stats.md 2>&1
cat $OUT/step_anatomy.txt
