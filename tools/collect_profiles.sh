#!/bin/bash
# Run on the GPU box (through gpurun): kernel-trace stats of the bench command and of one action chunk, plus three PMC
# passes over one training step.  Only small text summaries are kept (gpurun copies back <= 64 MiB).
# usage (from the build container; the box has no .git):  gpurun -- "KAI0_COMMIT=$(git rev-parse --short HEAD) bash tools/collect_profiles.sh"
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/summary; rm -rf $OUT; mkdir -p $OUT
# per-kernel durations are taken with the action expert's second stream off (KAI0_EXPERT_STREAM=0): every kernel then owns the
# chip while it runs, which is also how bench.py times the GEMM launches for its roofline object
export KAI0_EXPERT_STREAM=0
BENCH="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-latency --no-trim-extra"
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_train -o train -- $BENCH > $OUT/bench_under_rocprof.log 2>&1
python tools/prof_summary.py $(find /tmp/prof_train -name "*.db" | head -1) > $OUT/train_kernel_stats.md 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_inf -o inf -- python tools/infer_once.py 5 1 > $OUT/infer_under_rocprof.log 2>&1
python tools/prof_summary.py $(find /tmp/prof_inf -name "*.db" | head -1) > $OUT/infer_kernel_stats.md 2>&1
python tools/infer_timeline.py $(find /tmp/prof_inf -name "*.db" | head -1) > $OUT/infer_timeline.txt 2>&1
ONE="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-latency --no-gemm-timing --no-trim-extra"
for C in "FETCH_SIZE:fetch" "WRITE_SIZE:write" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA:mfma"; do
  CN="${C%%:*}"; DN="${C##*:}"
  timeout 900 rocprofv3 --pmc $CN --kernel-trace -d /tmp/pmc_$DN -o p --output-format csv -- $ONE > $OUT/pmc_$DN.log 2>&1
  python tools/pmc_summary.py /tmp/pmc_$DN 14 > $OUT/pmc_$DN.txt 2>&1
done
python tools/gemm_traffic.py /tmp/pmc_fetch /tmp/pmc_write $OUT/gemm_traffic.json > $OUT/gemm_traffic.log 2>&1
# the action chunk's HBM-side bytes per launch (FETCH_SIZE / WRITE_SIZE, eager launches) against the durations of the un-counted trace above
for C in "FETCH_SIZE:fetch" "WRITE_SIZE:write"; do
  CN="${C%%:*}"; DN="${C##*:}"
  timeout 600 rocprofv3 --pmc $CN --kernel-trace -d /tmp/pmc_inf_$DN -o p --output-format csv -- python tools/infer_once.py 2 0 > $OUT/pmc_inf_$DN.log 2>&1
done
python tools/infer_pmc.py /tmp/pmc_inf_fetch /tmp/pmc_inf_write $(find /tmp/prof_inf -name "*.db" | head -1) 30 > $OUT/infer_chunk_pmc.txt 2>&1
python tools/overlap_summary.py $(find /tmp/prof_train -name "*.db" | head -1) 2 > $OUT/train_overlap.txt 2>&1
grep -h '"metric"' $OUT/*.log | cut -c1-600
ls -la $OUT
