#!/bin/bash
# round 4, call D: new parity tests (full depth incl. bf16-oracle backward + estimator; 20-step trajectory) and the 1-rank RCCL dry run of bench.py
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_fulldepth_gpu.py tests/test_model_gpu.py -m gpu -q -x -k "fulldepth or trajectory" -s > gpurun_out/r4d_pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|worst|AdvantageEstimator|HIP /" gpurun_out/r4d_pytest.log | tail -8
KAI0_FORCE_COLLECTIVES=1 timeout 600 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-latency --no-trim-extra > gpurun_out/r4d_bench_coll.log 2>&1; echo "bench rc=$?"
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r4d_bench_coll.log') if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print(d['value'], d['config']['parallelism']); print(json.dumps(d.get('comm',{}).get('fsdp'))[:600])
else: print(open('gpurun_out/r4d_bench_coll.log').read()[-1500:])
PY
