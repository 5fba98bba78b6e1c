#!/bin/bash
# GPU tests + a short bench (no cpu baseline); logs under gpurun_out/
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x ${PYTEST_ARGS:-} > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu.log
KAI0_GEMM_BREAKDOWN=1 timeout 900 python bench.py --steps 8 --warmup 3 --no-cpu-baseline ${BENCH_ARGS:-} > gpurun_out/bench.log 2>&1; echo "bench rc=$?"
python - <<'PY'
import json
l=[x for x in open('gpurun_out/bench.log') if x.startswith('{')]
if l:
    d=json.loads(l[-1]); r=d.get('roofline',{}); i=d.get('inference',{})
    print('samples/s', round(d['value'],2), 'ms/step', round(d['ms_per_step'],1), 'gemm frac', round(r.get('frac',0),4), 'gemm ms', round(r.get('gemm_ms_per_step',0),1), 'p50', round(i.get('p50_ms',0),2), i.get('stages_ms'))
else:
    print(open('gpurun_out/bench.log').read()[-2000:])
PY
