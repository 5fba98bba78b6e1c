#!/bin/bash
# the whole -m gpu suite, smoke, the default bench line, a 1-rank RCCL dry run of the N > 1 bench path, the serve path's host share
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
t0=$(date +%s)
timeout 3000 python -m pytest tests -m gpu -q --durations=10 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$? in $(( $(date +%s) - t0 )) s"; grep -E "passed|failed" gpurun_out/pytest_gpu.log | tail -1; grep -E "^FAILED|^ERROR" gpurun_out/pytest_gpu.log | head
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/smoke.log | cut -c1-200
timeout 900 python bench.py > gpurun_out/bench_default.log 2>&1; echo "bench rc=$?"
tail -1 gpurun_out/bench_default.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; i=d['inference']; f=r['families']
print('samples/s', round(d['value'],2), 'ms/step', round(d['ms_per_step'],1), 'gemm frac', round(r['frac'],4), 'headline-schedule frac', round(r['headline_schedule']['frac'],4) if r.get('headline_schedule') else None, 'step frac', round(r['step_frac_of_mfma_peak'],4))
print('gemma_blocks', round(f['gemma_blocks']['frac'],3), 'vit_blocks', round(f['vit_blocks']['frac'],3), 'attention', round(f['attention']['frac'],3))
print('p50', round(i['p50_ms'],2), i.get('stages_ms'), 'trimmed', i.get('trimmed_prompt',{}).get('p50_ms'))
print('cpu', {k: d['cpu_baseline'][k] for k in ('value','cores','vocab','fwd_bwd_s_per_sample','optimizer_s')}, 'ranks', d['ranks']['rccl_ranks_seen'], d['ranks']['devices'][0]['device'][:40])"
KAI0_FORCE_COLLECTIVES=1 KAI0_BENCH_FSDP=0 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-latency --no-trim-extra > gpurun_out/bench_rccl1.log 2>&1; echo "1-rank RCCL bench rc=$?"; tail -1 gpurun_out/bench_rccl1.log | cut -c1-300
timeout 300 python tools/policy_latency.py 30 > gpurun_out/policy_latency.log 2>&1; tail -1 gpurun_out/policy_latency.log | cut -c1-500
