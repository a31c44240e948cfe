#!/bin/bash
# PMC passes over one probe (separate rocprofv3 runs, --kernel-trace only, as MI355X_MICROARCH.md prescribes).
# usage: bash tools/pmc.sh <probe: fwd|wgrad|spmm|topk|all> <out.json>
PROBE=${1:-topk}; OUT=${2:-gpurun_out/pmc_$PROBE.json}
REPO=$PWD; export TMPDIR=/tmp
rm -rf /tmp/pmc_runs; mkdir -p /tmp/pmc_runs gpurun_out $(dirname $OUT)
i=0
for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d /tmp/pmc_runs/p$i -o run -- python $REPO/tools/kernel_probe.py $PROBE 2 > /tmp/pmc_runs/p$i.log 2>&1; echo "pass $i ($SET) exit $?")
done
python $REPO/tools/pmc_aggregate.py /tmp/pmc_runs $OUT
