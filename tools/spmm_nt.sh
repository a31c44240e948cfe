#!/bin/bash
# PMC passes over tools/spmm_nt.py (separate rocprofv3 runs with --kernel-trace only, MI355X_MICROARCH.md) + one plain timing run.
# Output: profiles-ready JSON (default gpurun_out/r05_pmc_spmm_nt.json): per (H, direction) ms, 2 x FETCH_SIZE + WRITE_SIZE per launch, L2 hit rate.
OUT=${1:-gpurun_out/r05_pmc_spmm_nt.json}
N=4
REPO=$PWD; export TMPDIR=/tmp
rm -rf /tmp/pmc_nt; mkdir -p /tmp/pmc_nt $(dirname $OUT)
timeout 300 python tools/spmm_nt.py 8 --json /tmp/pmc_nt/timing.json > /tmp/pmc_nt/timing.log 2>&1; echo "timing run exit $?"
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d /tmp/pmc_nt/p$i -o run -- python $REPO/tools/spmm_nt.py $N > /tmp/pmc_nt/p$i.log 2>&1; echo "pass $i ($SET) exit $?")
done
python - "$OUT" $N <<'PY'
import csv, glob, json, sys
out, n = sys.argv[1], int(sys.argv[2])
HS = [0, 4096, 16384, 65536, 262144]
timing = json.load(open("/tmp/pmc_nt/timing.json"))
cnt = {}
for path in sorted(glob.glob("/tmp/pmc_nt/**/*counter_collection.csv", recursive=True)):
    disp = {}
    with open(path) as f:
        for row in csv.DictReader(f):
            k = row.get("Kernel_Name") or ""
            if "spmm_kernel" not in k:
                continue
            key = (int(row["Dispatch_Id"]), row["Counter_Name"])
            disp[key] = disp.get(key, 0.0) + float(row["Counter_Value"] or 0)
    ids = sorted({d for d, _ in disp})
    assert len(ids) == len(HS) * 2 * n, (path, len(ids))
    for c in {c for _, c in disp}:
        for j, d_ in enumerate(ids):
            h, dr = HS[j // (2 * n)], ("ui", "iu")[(j // n) % 2]
            if j % n == 0:
                continue                                    # the first launch of each group is the cold one
            cnt.setdefault((h, dr, c), []).append(disp[(d_, c)])
mean = lambda v: sum(v) / len(v) if v else None
for rec in timing["variants"]:
    for dr in ("ui", "iu"):
        f, w = mean(cnt.get((rec["H"], dr, "FETCH_SIZE"))), mean(cnt.get((rec["H"], dr, "WRITE_SIZE")))
        h, m = mean(cnt.get((rec["H"], dr, "TCC_HIT_sum"))), mean(cnt.get((rec["H"], dr, "TCC_MISS_sum")))
        if f is not None and w is not None:
            rec[dr]["hbm_bytes_per_launch"] = (2.0 * f + w) * 1024.0
        if h is not None and m:
            rec[dr]["l2_hit_rate"] = h / (h + m)
timing["units"] = "hbm_bytes = (2 x FETCH_SIZE + WRITE_SIZE) KB x 1024 (the gfx950 correction of MI355X_MICROARCH.md); ms from the un-profiled run"
json.dump(timing, open(out, "w"), indent=1)
for rec in timing["variants"]:
    print(rec["H"], {dr: {k: (round(v, 4) if isinstance(v, float) else v) for k, v in rec[dr].items() if k in ("ms", "hbm_bytes_per_launch", "l2_hit_rate", "bit_identical_to_default_policy", "edge_share_of_rows_below_H")} for dr in ("ui", "iu")})
PY
