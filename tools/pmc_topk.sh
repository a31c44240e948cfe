#!/bin/bash
# PMC passes over the top-K probe (separate rocprofv3 runs, --kernel-trace only): issue / wait / co-execution breakdown.
# usage: bash tools/pmc_topk.sh <out.json>
OUT=${1:-gpurun_out/pmc_topk.json}
REPO=$PWD; export TMPDIR=/tmp
rm -rf /tmp/pmc_runs; mkdir -p /tmp/pmc_runs gpurun_out $(dirname $OUT)
i=0
for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_SALU" \
           "SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES SQ_INSTS_VMEM_RD SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d /tmp/pmc_runs/p$i -o run -- python $REPO/tools/kernel_probe.py topk 2 > /tmp/pmc_runs/p$i.log 2>&1; echo "pass $i ($SET) exit $?")
done
python $REPO/tools/pmc_aggregate.py /tmp/pmc_runs $OUT | grep -i topk
