#!/bin/bash
# per-shape GEMM rates inside the training step for each env setting in $AB (semicolon separated)
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
IFS=';' read -ra CASES <<< "${AB:--}"
i=0
for c in "${CASES[@]}"; do
  [ "$c" = "-" ] && c=""
  env $c KAI0_GEMM_BREAKDOWN=1 timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-latency --no-trim-extra > gpurun_out/bd_$i.log 2>&1
  cp gpurun_out/gemm_breakdown.json gpurun_out/gemm_breakdown_$i.json
  i=$((i+1))
done
python - <<'PY'
import json,glob
fs=sorted(glob.glob('gpurun_out/gemm_breakdown_[0-9].json'))
ds=[{(r['a_kc'],r['b_kc'],r['M'],r['N'],r['K'],r['batch']):r for r in json.load(open(f))} for f in fs]
keys=sorted(ds[0],key=lambda k:-ds[0][k]['ms'])[:16]
for k in keys:
    print(k, ' | '.join(f"{d[k]['ms']/2:7.2f} ms {d[k]['tflops']:7.1f}" if k in d else 'n/a' for d in ds))
print('total', ' | '.join(f"{sum(r['ms'] for r in d.values())/2:8.2f}" for d in ds))
PY
