#!/bin/bash
# round 5, first GPU call: the new bench-configuration parity tests + the new kernel test + a short bench with the family-resolved roofline
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_bench_config_gpu.py "tests/test_kernels_gpu.py::test_gemm_batch_strides_may_be_negative_or_span_two_allocations" tests/test_kernels_gpu.py::test_gemm_row_remaps_and_batch -m gpu -q -s > gpurun_out/pytest_r5a.log 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/pytest_r5a.log
KAI0_GEMM_BREAKDOWN=1 timeout 900 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/bench_r5a.log 2>&1; echo "bench rc=$?"
python - <<'PY'
import json
l=[x for x in open('gpurun_out/bench_r5a.log') if x.startswith('{')]
if l:
    d=json.loads(l[-1]); r=d.get('roofline',{}); i=d.get('inference',{})
    print('samples/s', round(d['value'],2), 'ms/step', round(d['ms_per_step'],1), 'gemm frac', round(r.get('frac',0),4), 'gemm ms', round(r.get('gemm_ms_per_step',0),1), 'p50', round(i.get('p50_ms',0),2), i.get('stages_ms'))
    for k,v in r.get('families',{}).items(): print('  ',k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items() if a!='note'})
else:
    print(open('gpurun_out/bench_r5a.log').read()[-3000:])
PY
