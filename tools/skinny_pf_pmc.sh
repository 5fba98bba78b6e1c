#!/bin/bash
# L2 hits / misses per skinny launch with and without the chained prefetch (tools/skinny_cold_warm.py variants).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in ${PF_CASES:-0 1}; do
  KAI0_PF_PROBE=$([ $v = 0 ] && echo 0 || echo $v) timeout 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace -d /tmp/pfp_$v -o p --output-format csv -- python tools/skinny_cold_warm.py > /tmp/pfp_$v.log 2>&1
  echo "== prefetch $v"; python tools/pmc_summary.py /tmp/pfp_$v 2 2>&1 | grep -A4 skinny
done
