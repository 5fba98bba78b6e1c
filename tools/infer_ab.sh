#!/bin/bash
# A/B of environment switches on the B = 1 action chunk: one tools/infer_bench.py run per ;-separated entry of $AB ("-" = defaults)
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
if [ -n "${PYTEST_K:-}" ]; then python -m pytest tests -m gpu -q -x -k "$PYTEST_K" > gpurun_out/pytest_ab.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_ab.log; fi
IFS=';' read -ra CASES <<< "${AB:--}"
for c in "${CASES[@]}"; do
  [ "$c" = "-" ] && c=""
  echo -n "[${c:-defaults}] "
  env $c timeout 300 python tools/infer_bench.py 2>&1 | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('p50', d['p50_ms'], 'min', d['min_ms'], {k: round(v,3) for k,v in d.get('stages_ms',{}).items()})
except Exception as e: print('FAILED', e)"
done
