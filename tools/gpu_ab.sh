#!/bin/bash
# A/B of environment switches on the training step: one short bench per line of $AB (semicolon-separated env assignments, "-" = defaults).
# usage (through gpurun): AB="-;KAI0_SIDE_STREAM=0;KAI0_SIDE_STREAM=0 KAI0_ASYNC_OPT=0" bash tools/gpu_ab.sh
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
if [ -n "${PYTEST_K:-}" ]; then python -m pytest tests -m gpu -q -x -k "$PYTEST_K" > gpurun_out/pytest_ab.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_ab.log; fi
IFS=';' read -ra CASES <<< "${AB:--}"
i=0
for c in "${CASES[@]}"; do
  [ "$c" = "-" ] && c=""
  env $c timeout 600 python bench.py --steps ${STEPS:-8} --warmup 3 --no-cpu-baseline ${BENCH_ARGS:---no-latency} > gpurun_out/ab_$i.log 2>&1
  python - "$c" gpurun_out/ab_$i.log <<'PY'
import json, sys
l = [x for x in open(sys.argv[2]) if x.startswith('{')]
if l:
    d = json.loads(l[-1]); r = d.get('roofline', {}); inf = d.get('inference', {})
    print(f"[{sys.argv[1] or 'defaults'}] samples/s {d['value']:.2f} ms/step {d['ms_per_step']:.1f} gemm frac {r.get('frac', 0):.4f} gemm ms {r.get('gemm_ms_per_step', 0):.1f} loss {d['config'].get('final_loss')} p50 {inf.get('p50_ms')} {inf.get('stages_ms')}")
else:
    print(f"[{sys.argv[1]}] FAILED", open(sys.argv[2]).read()[-1500:])
PY
  i=$((i+1))
done
