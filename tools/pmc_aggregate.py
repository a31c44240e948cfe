"""Aggregate rocprofv3 counter_collection CSVs (one directory per --pmc pass) into
{kernel: {counter: {"launches": n, "mean": value per launch}}}."""
import csv, glob, json, os, sys
root, out = sys.argv[1], sys.argv[2]
acc = {}
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    per_dispatch = {}
    with open(path) as f:
        for row in csv.DictReader(f):
            k = row.get("Kernel_Name") or row.get("Kernel Name")
            c = row.get("Counter_Name"); v = float(row.get("Counter_Value") or 0)
            d = row.get("Dispatch_Id")
            per_dispatch[(k, c, d)] = per_dispatch.get((k, c, d), 0.0) + v      # sum over dimensions (XCD / SE instances)
    for (k, c, d), v in per_dispatch.items():
        if not k or not k.startswith("llmrec") and "llmrec" not in k:
            continue
        k = k.split("(")[0].replace("void ", "")
        e = acc.setdefault(k, {}).setdefault(c, {"launches": 0, "sum": 0.0})
        e["launches"] += 1; e["sum"] += v
res = {k: {c: {"launches": e["launches"], "mean": e["sum"] / e["launches"]} for c, e in cs.items()} for k, cs in acc.items()}
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
for k, cs in sorted(res.items()):
    print(k, {c: round(e["mean"], 1) for c, e in sorted(cs.items())})
