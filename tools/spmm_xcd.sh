#!/bin/bash
# Round 6: the SpMM's XCD-contiguous block -> row map against the linear one, on the standard generator and on the
# planted-community graph (ids contiguous per community / shuffled), both directions: a timing run, then three rocprofv3 --pmc passes
# (separate runs, --kernel-trace only). Usage: bash tools/spmm_xcd.sh [out.json]   -> the table of profiles/experiments/r06_spmm_xcd.md
OUT=${1:-gpurun_out/r06_pmc_spmm_xcd.json}; mkdir -p $(dirname $OUT)
REPO=$PWD; export TMPDIR=/tmp
LOG=/tmp/spmm_xcd.log; : > $LOG
for V in std comm comm_shuf; do for D in ui iu; do for X in 0 1; do
  export LLMREC_XCD=$X LLMREC_DIR=$D
  timeout 300 python tools/spmm_locality.py $V 10 2>&1 | grep LOCALITY | tee -a $LOG
  rm -rf /tmp/pmc_xcd; mkdir -p /tmp/pmc_xcd; i=0
  for SET in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d /tmp/pmc_xcd/p$i -o run -- python $REPO/tools/spmm_locality.py $V 3 > /tmp/pmc_xcd/p$i.log 2>&1)
  done
  python tools/pmc_aggregate.py /tmp/pmc_xcd /tmp/pmc_xcd/agg.json > /dev/null
  python - <<PY | tee -a $LOG
import json
d = json.load(open("/tmp/pmc_xcd/agg.json"))
k = [n for n in d if "spmm_kernel" in n]
if k:
    c = d[k[0]]
    f, w = c.get("FETCH_SIZE", {}).get("mean", 0) * 1024, c.get("WRITE_SIZE", {}).get("mean", 0) * 1024
    h, m = c.get("TCC_HIT_sum", {}).get("mean", 0), c.get("TCC_MISS_sum", {}).get("mean", 0)
    print("PMC $V dir $D xcd $X traffic_GB %.3f l2_hit %.3f" % ((2 * f + w) / 1e9, h / max(h + m, 1)))
PY
done; done; done
python - $LOG $OUT <<'PY'
import json, sys
rows, cur = [], None
for l in open(sys.argv[1]):
    t = l.split()
    if l.startswith("LOCALITY"):
        cur = {"graph": t[1], "dir": t[3], "xcd_contiguous": int(t[5]), "nnz": int(t[7]), "ms": float(t[9]), "gedges_per_s": float(t[11]),
               "frac_hbm_algorithmic": float(t[13]), "algorithmic_gb": float(t[17])}
        rows.append(cur)
    elif l.startswith("PMC") and cur is not None:
        cur["traffic_gb"] = float(t[7]); cur["l2_hit_rate"] = float(t[9]); cur["traffic_over_algorithmic"] = round(cur["traffic_gb"] / cur["algorithmic_gb"], 2)
json.dump({"what": "llmrec_spmm_f32 at 2 M x 1 M x 40 M edges (planted graphs: ~26 M), d = 64, linear vs XCD-contiguous block -> row map; traffic = 2 FETCH_SIZE + WRITE_SIZE per launch (gfx950 correction)",
           "command": "bash tools/spmm_xcd.sh", "rows": rows}, open(sys.argv[2], "w"), indent=1)
for r in rows:
    print(r)
PY
