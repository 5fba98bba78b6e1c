#!/bin/bash
# round 3, call A: full-depth parity, this box's baseline bench line with the per-shape GEMM table, FETCH / WRITE counters over the action chunk
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3a
O=gpurun_out/r3a
timeout 1500 python -m pytest tests/test_fulldepth_gpu.py -m gpu -x -q -s > $O/pytest_fulldepth.log 2>&1; echo "fulldepth rc=$?"; tail -5 $O/pytest_fulldepth.log | cut -c1-300
cat gpurun_out/parity_r03.txt 2>/dev/null | cut -c1-400
KAI0_GEMM_BREAKDOWN=1 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-trim-extra > $O/bench.log 2>&1; tail -1 $O/bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; i=d['inference']
print('samples/s', round(d['value'],2), 'ms/step', round(d['ms_per_step'],1), 'gemm frac', round(r['frac'],4), 'gemm ms', round(r['gemm_ms_per_step'],1), 'p50', round(i['p50_ms'],2), i.get('stages_ms'))"
cp gpurun_out/gemm_breakdown.json $O/ 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_inf -o inf -- python tools/infer_once.py 5 1 > $O/infer_under_rocprof.log 2>&1
DB=$(find /tmp/prof_inf -name "*.db" | head -1)
python tools/infer_timeline.py $DB > $O/infer_timeline.txt 2>&1
for C in "FETCH_SIZE:fetch" "WRITE_SIZE:write"; do
  CN="${C%%:*}"; DN="${C##*:}"
  timeout 600 rocprofv3 --pmc $CN --kernel-trace -d /tmp/pmc_inf_$DN -o p --output-format csv -- python tools/infer_once.py 2 0 > $O/pmc_inf_$DN.log 2>&1
done
python tools/infer_pmc.py /tmp/pmc_inf_fetch /tmp/pmc_inf_write $DB 30 > $O/infer_chunk_pmc.txt 2>&1
head -30 $O/infer_chunk_pmc.txt | cut -c1-200
