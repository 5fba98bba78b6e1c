#!/bin/bash
# Round-end evidence run (tests are run separately): default bench line, per-shape GEMM rates, serve-path host share, profile collection.
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke.log | cut -c1-300
timeout 900 python bench.py > gpurun_out/bench_default.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench_default.log | cut -c1-300
KAI0_GEMM_BREAKDOWN=1 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench.log 2>&1; tail -1 gpurun_out/bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; i=d['inference']
print('samples/s', round(d['value'],2), 'ms/step', round(d['ms_per_step'],1), 'gemm frac', round(r['frac'],4), 'gemm ms', round(r['gemm_ms_per_step'],1), 'step frac', round(r['step_frac_of_mfma_peak'],4), 'p50', round(i['p50_ms'],2), i.get('stages_ms'))"
timeout 300 python tools/policy_latency.py 30 > gpurun_out/policy_latency.log 2>&1; tail -1 gpurun_out/policy_latency.log | cut -c1-600
bash tools/collect_profiles.sh > gpurun_out/collect_profiles.log 2>&1; tail -3 gpurun_out/collect_profiles.log
