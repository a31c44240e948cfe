#!/bin/bash
# every variant of tools/spmm_locality.py: a timing run, then three rocprofv3 --pmc passes (separate runs, --kernel-trace only)
OUT=${1:-gpurun_out/r03/spmm_locality.txt}; mkdir -p $(dirname $OUT); : > $OUT
REPO=$PWD; export TMPDIR=/tmp
for V in std std_items std_both comm comm_shuf comm_reord; do
  timeout 300 python tools/spmm_locality.py $V 10 2>&1 | grep LOCALITY | tee -a $OUT
  rm -rf /tmp/pmc_loc; mkdir -p /tmp/pmc_loc; i=0
  for SET in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d /tmp/pmc_loc/p$i -o run -- python $REPO/tools/spmm_locality.py $V 3 > /tmp/pmc_loc/p$i.log 2>&1)
  done
  python tools/pmc_aggregate.py /tmp/pmc_loc /tmp/pmc_loc/agg.json > /dev/null
  python - <<PY | tee -a $OUT
import json
d = json.load(open("/tmp/pmc_loc/agg.json"))
k = [n for n in d if "spmm_kernel" in n]
if k:
    c = d[k[0]]
    f, w = c.get("FETCH_SIZE", {}).get("mean", 0) * 1024, c.get("WRITE_SIZE", {}).get("mean", 0) * 1024
    h, m = c.get("TCC_HIT_sum", {}).get("mean", 0), c.get("TCC_MISS_sum", {}).get("mean", 0)
    print("   PMC $V: 2*FETCH+WRITE = %.2f GB per launch, L2 hit rate %.3f" % ((2 * f + w) / 1e9, h / max(h + m, 1)))
PY
done
