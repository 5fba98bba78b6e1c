#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3c; O=gpurun_out/r3c
python -m pytest tests/test_model_gpu.py -m gpu -q -x -k "rccl or train_loop or policy or sample_actions" > $O/pytest_sel.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_sel.log | cut -c1-300
KAI0_FORCE_COLLECTIVES=1 KAI0_SHARD_MODE=fsdp timeout 600 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-latency --no-trim-extra > $O/bench_comm_fsdp.log 2>&1
tail -1 $O/bench_comm_fsdp.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('fsdp', round(d['value'],2), 'samples/s', d['config']['parallelism'], json.dumps(d.get('comm'))[:700])"
python tools/infer_host_breakdown.py 30 2>&1 | tail -8
KAI0_INFER_CACHE_MODS=0 python tools/infer_once.py 10 1 2>&1 | tail -1
python tools/infer_once.py 10 1 2>&1 | tail -1
