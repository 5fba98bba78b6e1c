#!/bin/bash
# LDS / wait / MFMA counters of the joint attention forward kernel (two PMC passes over tools/attn_fwd_bench.py); run through gpurun
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/attn_pmc; mkdir -p $O
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ_[A-Z_0-9]+|GRBM_[A-Z_]+|TCC_[A-Z_0-9]+|TCP_[A-Z_0-9]+)\b" | sort -u | grep -E "LDS|MFMA|WAIT|ACTIVE_INST|BUSY|WAVE_CYCLES|VALU" > $O/counters.txt
cat $O/counters.txt | tr '\n' ' ' | cut -c1-3000
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY:a" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_MISC SQ_INSTS_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE:b"; do
  CN="${C%%:*}"; DN="${C##*:}"
  timeout 300 rocprofv3 --pmc $CN --kernel-trace -d /tmp/pmc_att_$DN -o p --output-format csv -- python tools/attn_fwd_bench.py > $O/pmc_$DN.log 2>&1
  python tools/pmc_summary.py /tmp/pmc_att_$DN 3 > $O/pmc_att_$DN.txt 2>&1
  grep -A12 "attn_fwd_kernel" $O/pmc_att_$DN.txt | head -14
done
