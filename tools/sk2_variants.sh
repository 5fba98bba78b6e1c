#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in ${VARIANTS:-0}; do
  rm -rf /tmp/pv$v
  KAI0_SK2_PAIR_VARIANT=${v%%:*} KAI0_SK2_VARIANT=${v##*:} timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pv$v -o inf -- python tools/infer_once.py 4 1 > /tmp/pv$v.log 2>&1
  echo "variant $v: $(grep 'chunk ms' /tmp/pv$v.log)"
  python tools/prof_by_grid.py $(find /tmp/pv$v -name "*.db" | head -1) skinny2
done
