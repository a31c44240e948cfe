#!/bin/bash
# PMC passes over the product SpMM at the bench's 2 M x 1 M x 40 M graph (tools/spmm_40m.py), separate rocprofv3 runs with --kernel-trace only
# (MI355X_MICROARCH.md). Output (profiles/r04_pmc_spmm_40M.json): per direction HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KB, L2 hit rate.
OUT=${1:-gpurun_out/pmc_spmm_40M.json}
REPO=$PWD; export TMPDIR=/tmp
rm -rf /tmp/pmc_spmm; mkdir -p /tmp/pmc_spmm $(dirname $OUT)
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d /tmp/pmc_spmm/p$i -o run -- python $REPO/tools/spmm_40m.py 3 > /tmp/pmc_spmm/p$i.log 2>&1; echo "pass $i ($SET) exit $?")
done
python - "$OUT" <<'PY'
import csv, glob, json, os, sys
out = sys.argv[1]
# per dispatch: (kernel, counter) -> value summed over instances; the spmm_kernel dispatches come in program order: n x ui, then n x iu
per = {}
for path in sorted(glob.glob("/tmp/pmc_spmm/**/*counter_collection.csv", recursive=True)):
    disp = {}
    with open(path) as f:
        for row in csv.DictReader(f):
            k = row.get("Kernel_Name") or ""
            if "spmm_kernel" not in k:
                continue
            key = (int(row["Dispatch_Id"]), row["Counter_Name"])
            disp[key] = disp.get(key, 0.0) + float(row["Counter_Value"] or 0)
            per.setdefault("kernel_name", k.split("(")[0])
    ids = sorted({d for d, _ in disp})
    half = len(ids) // 2
    for c in {c for _, c in disp}:
        per.setdefault(c, {})["ui"] = [disp[(d, c)] for d in ids[:half] if (d, c) in disp]
        per[c]["iu"] = [disp[(d, c)] for d in ids[half:] if (d, c) in disp]
mean = lambda v: sum(v) / len(v) if v else None
res = {"what": "rocprofv3 --pmc over tools/spmm_40m.py: spmm_kernel at 2 M x 1 M x 40 M edges, d = 64; ui = rows are users (gathers item rows), iu = rows are items",
       "kernel": per.get("kernel_name"), "units": "FETCH_SIZE / WRITE_SIZE in KB as reported; hbm_bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (the gfx950 correction of MI355X_MICROARCH.md)",
       "directions": {}}
for dname in ("ui", "iu"):
    f, w = mean(per.get("FETCH_SIZE", {}).get(dname, [])), mean(per.get("WRITE_SIZE", {}).get(dname, []))
    h, m = mean(per.get("TCC_HIT_sum", {}).get(dname, [])), mean(per.get("TCC_MISS_sum", {}).get(dname, []))
    if f is None or w is None:
        continue
    res["directions"][dname] = {"FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0,
                                "TCC_HIT_sum": h, "TCC_MISS_sum": m, "l2_hit_rate": (h / (h + m)) if h is not None and m else None,
                                "launches": len(per["FETCH_SIZE"][dname])}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res["directions"]))
PY
