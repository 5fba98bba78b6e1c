#!/bin/bash
# LDS / wait counters of the two big GEMM schedules (NT quadrant, TN ring) on single shapes
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/gemm_lds_pmc; rm -rf $O; mkdir -p $O
i=0
for SHAPE in "TN 16384 2048 30976" "NT 30976 2048 16384" "NT 30976 16384 2048"; do
  i=$((i+1))
  for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY GRBM_GUI_ACTIVE:a" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VALU SQ_LDS_ADDR_CONFLICT:b"; do
    CN="${C%%:*}"; DN="${C##*:}"
    timeout 300 rocprofv3 --pmc $CN --kernel-trace -d /tmp/gq_${i}_$DN -o p --output-format csv -- python tools/gemm_one.py $SHAPE 4 > $O/log_${i}_$DN.txt 2>&1
    echo "## $SHAPE" >> $O/summary.txt
    python tools/pmc_summary.py /tmp/gq_${i}_$DN 1 2>&1 | grep -v "^#" >> $O/summary.txt
  done
done
cat $O/summary.txt
