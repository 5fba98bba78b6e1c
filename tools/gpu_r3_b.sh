#!/bin/bash
# round 3, call B: whole -m gpu suite, and the N = 1 RCCL dry run of the bench line's `comm` object (both modes)
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3b; O=gpurun_out/r3b
python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log | cut -c1-300
for MODE in zero2 fsdp; do
KAI0_FORCE_COLLECTIVES=1 KAI0_SHARD_MODE=$MODE timeout 600 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-latency --no-trim-extra > $O/bench_comm_$MODE.log 2>&1
tail -1 $O/bench_comm_$MODE.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$MODE', round(d['value'],2), 'samples/s', d['config']['parallelism'], json.dumps(d.get('comm'))[:900])"
done
