#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for sp in "0,0,0" "1,1,6" "1,1,8" "1,2,8" "2,2,8" "1,6,6" "1,3,4" "1,2,3"; do
  echo -n "splits $sp: "; KAI0_PREFIX_SPLITS=$sp python tools/infer_bench.py 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['p50_ms'], d['stages_ms'])"
done
