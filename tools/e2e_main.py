"""The metric as the reference defines it (reference main.py:200,297,303,306-312): `python main.py` on the full Netflix-shaped
dataset, timed by the drop-in's own epoch timers (t2 - t1 = train, t3 - t2 = evaluation), in its two modes.

    python tools/e2e_main.py [--epochs 6] [--out profiles/r04_e2e_main.json] [--data /tmp/llmrec_e2e]

1. writes the dataset directory the reference's loader expects (llmrec_amd/synth.write_dataset: U = 13 187, I = 17 366,
   68 933 interactions, every user with one validation and one test item -> 13 187 test users as in BASELINE.md section 2,
   real feature widths 512 / 768 / 1536, 5 attribute keys, LLM-augmented sample dict);
2. runs `python main.py --dataset netflix_valid_item --data_path <dir>/ --epoch N` as a subprocess in
     default               : the reference's host sample stream + graph-replayed fused step + graph-replayed evaluation
     graph_device_sampler  : LLMREC_DEVICE_SAMPLER=1 (the HIP sampler inside the step graph: what bench.py's headline times)
   and keeps its `Epoch %d [%.1fs + %.1fs]` lines and the process wall time (start-up included);
3. runs the same `__main__` body in-process-per-mode (tools/e2e_main.py --child MODE) with Trainer._on_epoch set, which hands
   over the SAME two timer differences unrounded (the log line prints 0.1 s; an epoch here is tens of milliseconds) plus the host
   seconds the epoch spent in Data.sample().
Writes one JSON: per mode the per-epoch (train_s, eval_s, sample_s), the median over epochs >= 1 (epoch 0 carries the graph
captures), edges/s = n_batch * batch_size / train_s and users/s = n_test_users / eval_s - next to BASELINE.md's figures for the
unmodified reference on CPU (12.8 s + 42.3 s per epoch on 8 cores)."""
from __future__ import annotations

import argparse
import json
import os
import re
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODES = {"default": {}, "graph_device_sampler": {"LLMREC_DEVICE_SAMPLER": "1"}, "eager_no_graph": {"LLMREC_GRAPH": "0"}}
DATASET = "netflix_valid_item"


def write_dataset(data_root: str):
    sys.path.insert(0, ROOT)
    from llmrec_amd import synth
    ds = os.path.join(data_root, DATASET)
    if os.path.exists(os.path.join(ds, "augmented_sample_dict")):
        return ds, None
    t = time.time()
    stats = synth.write_dataset(ds, synth.NF_SHAPE.n_users, synth.NF_SHAPE.n_items, 68933, seed=0, keys=synth.DATASET_KEYS[DATASET],
                                max_deg=1000, min_deg=3, attr_rows_as_arrays=True)
    stats["write_s"] = round(time.time() - t, 1)
    return ds, stats


def child(mode: str, data_root: str, epochs: int):
    """The body of main.py's __main__ with the epoch hook set; prints one JSON line."""
    sys.path.insert(0, ROOT)
    os.chdir(ROOT)
    sys.argv = ["main.py", "--dataset", DATASET, "--data_path", data_root + "/", "--epoch", str(epochs), "--debug",
                "--early_stopping_patience", "1000"]
    t0 = time.time()
    import main as M
    import torch
    M.set_seed(M.args.seed)
    trainer = M.Trainer(data_config={"n_users": M.data_generator.n_users, "n_items": M.data_generator.n_items})
    torch.cuda.synchronize()
    t_init = time.time() - t0
    M._progress = lambda it: it
    rec = []
    trainer._on_epoch = lambda ep, loss, mf, emb, ret, t: rec.append(
        {"epoch": ep, "train_s": t[0], "eval_s": t[1], "sample_s": getattr(trainer, "sample_time", 0.0), "loss": float(loss), "mf_loss": float(mf),
         "emb_loss": float(emb), "recall20": float(ret["recall"][1]),
         "metrics": {m: [float(x) for x in ret[m]] for m in ("precision", "recall", "ndcg", "hit_ratio")}})
    t1 = time.time()
    trainer.train()
    n_batch = M.data_generator.n_train // M.args.batch_size + 1
    print("E2E_JSON " + json.dumps({"mode": mode, "init_s": t_init, "train_call_s": time.time() - t1, "n_batch": n_batch, "batch_size": M.args.batch_size,
                                    "n_test_users": len(M.data_generator.test_set), "n_train": M.data_generator.n_train, "epochs": rec}), flush=True)


def reference_record():
    """The newest committed record of the UNMODIFIED reference on this dataset that carries losses, metrics and content digests
    (oracle/time_reference.py, run where /root/reference exists -> profiles/r*_reference_cpu.json). None when absent."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_reference_cpu.json")), reverse=True):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        if "digests" in d and all("metrics" in e for e in d.get("epochs", [])):
            d["file"] = os.path.basename(path)
            return d
    return None


def vs_reference(epochs, ref, digests_here, loss_tol=1e-4, metric_tol=0.002):
    """The drop-in's epochs against the reference's on the same bytes, the same seed and (default mode: the host sampler replays the
    reference's RNG streams) the same batches: the logged epoch sums within loss_tol relative (north_star: 1e-4 on loss), all 12 metrics of
    every common epoch within metric_tol absolute (north_star: Recall@20 within 0.002). The reference's sums are its log line's 5 decimals."""
    same_bytes = ref.get("digests") == digests_here
    n = min(len(epochs), len(ref["epochs"]))
    rel = lambda a, b: abs(a - b) / max(abs(b), 1e-12)
    loss_rel = max(rel(epochs[i]["loss"], ref["epochs"][i]["loss"]) for i in range(n))
    mf_rel = max(rel(epochs[i]["mf_loss"], ref["epochs"][i]["mf_loss"]) for i in range(n))
    # emb_loss prints as 0.00000 in the reference's line (it is ~1e-7): compared absolutely at the line's resolution
    emb_abs = max(abs(epochs[i]["emb_loss"] - ref["epochs"][i]["emb_loss"]) for i in range(n))
    mmax, worst = 0.0, None
    for i in range(n):
        for m, vals in ref["epochs"][i]["metrics"].items():
            for j, v in enumerate(vals):
                dlt = abs(epochs[i]["metrics"][m][j] - v)
                if dlt >= mmax:
                    mmax, worst = dlt, "epoch%d/%s[%d]" % (i, m, j)
    return {"ok": bool(same_bytes and loss_rel <= loss_tol and mf_rel <= loss_tol and emb_abs <= 1e-5 and mmax <= metric_tol), "epochs": n,
            "same_dataset_bytes": same_bytes, "loss_rel": loss_rel, "mf_rel": mf_rel, "emb_abs": emb_abs, "metric_max_abs": mmax, "metric_worst": worst,
            "recall20": [epochs[i]["metrics"]["recall"][1] for i in range(n)], "recall20_reference": [ref["epochs"][i]["metrics"]["recall"][1] for i in range(n)],
            "loss": [epochs[i]["loss"] for i in range(n)], "loss_reference": [ref["epochs"][i]["loss"] for i in range(n)],
            "tolerances": {"loss_rel": loss_tol, "metric_abs": metric_tol}, "reference_file": ref.get("file"),
            "what": "python main.py (default mode: the reference's own batches) vs the unmodified reference's Trainer.train() on the same dataset bytes and seed"}


def summarise(c):
    later = [e for e in c["epochs"] if e["epoch"] >= 1] or c["epochs"]
    med = lambda k: statistics.median(e[k] for e in later)
    tr, ev, sm = med("train_s"), med("eval_s"), med("sample_s")
    return {"train_s": tr, "eval_s": ev, "sample_s": sm, "sample_share_of_train": sm / tr if tr else None,
            "edges_per_s": c["n_batch"] * c["batch_size"] / tr, "users_per_s": c["n_test_users"] / ev,
            "epoch0_train_s": c["epochs"][0]["train_s"], "epoch0_eval_s": c["epochs"][0]["eval_s"], "init_s": c["init_s"],
            "epochs_timed": len(later), "n_batch": c["n_batch"], "n_test_users": c["n_test_users"],
            "per_epoch_train_s": [round(e["train_s"], 5) for e in c["epochs"]], "per_epoch_eval_s": [round(e["eval_s"], 5) for e in c["epochs"]],
            "final_loss": c["epochs"][-1]["loss"], "final_recall20": c["epochs"][-1]["recall20"],
            "epochs": [{k: e[k] for k in ("epoch", "loss", "mf_loss", "emb_loss", "metrics")} for e in c["epochs"]]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=6)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "e2e_main.json"))
    ap.add_argument("--data", default="/tmp/llmrec_e2e")
    ap.add_argument("--child", default=None)
    ap.add_argument("--modes", default="default,graph_device_sampler,eager_no_graph")
    a = ap.parse_args()
    if a.child:
        return child(a.child, a.data, a.epochs)
    ds, stats = write_dataset(a.data)
    sys.path.insert(0, ROOT)
    from llmrec_amd import synth
    dig = synth.dataset_digests(ds)
    ref = reference_record()
    out = {"dataset": {"dir": ds, "stats": stats, "digests": dig, "shape": "U 13187 x I 17366, 68933 interactions (42559 train), 13187 test users, feats 512/768/1536 x (1 + 5)"},
           "command": "python main.py --dataset %s --data_path %s/ --epoch %d --debug" % (DATASET, a.data, a.epochs),
           "timers": "Trainer.train's own t2 - t1 (train) and t3 - t2 (evaluation), reference main.py:200,297,303; median over epochs >= 1",
           "reference_cpu_baseline_md": {"train_s": 12.8, "eval_s": 42.3, "edges_per_s": 3360, "users_per_s": 310, "cores": 8,
                                         "source": "BASELINE.md section 2 (the unmodified reference on the survey's NF-shaped set, 42 steps / 13187 test users)"}}
    for mode in a.modes.split(","):
        env = dict(os.environ, **MODES[mode])
        # (a) the plain command, as a user of the reference would run it
        t0 = time.time()
        r = subprocess.run([sys.executable, "main.py", "--dataset", DATASET, "--data_path", a.data + "/", "--epoch", str(a.epochs), "--debug",
                            "--early_stopping_patience", "1000"], cwd=ROOT, env=env, capture_output=True, text=True)
        wall = time.time() - t0
        lines = [l for l in (r.stdout + r.stderr).splitlines() if re.search(r"Epoch \d+ \[[0-9.]+s \+ [0-9.]+s\]", l)]
        rec = {"python_main_py": {"returncode": r.returncode, "wall_s_incl_startup": round(wall, 2),
                                  "epoch_lines": [re.search(r"Epoch \d+ \[[0-9.]+s \+ [0-9.]+s\]", l).group(0) for l in lines]}}
        if r.returncode != 0:
            rec["python_main_py"]["stderr_tail"] = r.stderr[-1500:]
        # (b) the same body with the unrounded timers
        r2 = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", mode, "--data", a.data, "--epochs", str(a.epochs)],
                            cwd=ROOT, env=env, capture_output=True, text=True)
        js = [l for l in r2.stdout.splitlines() if l.startswith("E2E_JSON ")]
        if js:
            rec.update(summarise(json.loads(js[-1][9:])))
            if ref is not None:
                # default mode draws the reference's batches (same RNG streams): held to north_star's tolerances. The device sampler draws
                # other batches from the same distribution: its figures are reported, not gated (ok is None)
                rec["vs_reference"] = vs_reference(rec["epochs"], ref, dig)
                if mode != "default":
                    rec["vs_reference"]["ok"] = None
                    rec["vs_reference"]["what"] = "other batches than the reference's (device sampler): reported, not gated"
        else:
            rec["error"] = (r2.stderr or r2.stdout)[-1500:]
        out[mode] = rec
        print("[e2e] %s: %s" % (mode, json.dumps({k: v for k, v in rec.items() if not k.startswith("per_epoch") and k != "epochs"})), flush=True)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(out, open(a.out, "w"), indent=1)
    print("[e2e] wrote", a.out)


if __name__ == "__main__":
    main()
