#!/bin/bash
# PMC passes (L2 hit rate, fabric-side fetch bytes) over single GEMM shapes of the training step.  Run through gpurun.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/gemm_pmc; rm -rf $OUT; mkdir -p $OUT
i=0
for SHAPE in "NT 30976 32768 2048" "NT 30976 2048 16384" "TN 16384 2048 30976" "NT 24576 4304 1152" ${EXTRA_SHAPES:-}; do
  i=$((i+1))
  for C in "FETCH_SIZE:fetch" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum:l2" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum:ea"; do
    CN="${C%%:*}"; DN="${C##*:}"
    timeout 300 rocprofv3 --pmc $CN --kernel-trace -d /tmp/gp_${i}_$DN -o p --output-format csv -- python tools/gemm_one.py $SHAPE 3 > $OUT/log_${i}_$DN.txt 2>&1
    echo "## $SHAPE  [$CN]" >> $OUT/summary.txt
    python tools/pmc_summary.py /tmp/gp_${i}_$DN 2 2>&1 | grep -v "^#" >> $OUT/summary.txt
  done
done
cat $OUT/summary.txt
