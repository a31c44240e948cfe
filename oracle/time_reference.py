"""TEST / MEASUREMENT INFRASTRUCTURE ONLY - times the UNMODIFIED reference on CPU on the dataset tools/e2e_main.py times the
drop-in on. Runs in the build container only (needs /root/reference); the GPU box never sees the reference,
so the result travels as a committed record:

    python oracle/time_reference.py [--epochs 6] [--out profiles/r05_reference_cpu.json] [--note "..."]

The reference is imported through oracle/ref_loader.py (import shims only; its evaluation pool becomes an in-process map - the
reference would use Pool(cpu_count() // 5) = one worker on this 8-core host, utility/batch_test.py:11,115) and its own
`Trainer.train()` runs `--epoch N`; the two timers are read from its own log line `Epoch %d [%.1fs + %.1fs]` (main.py:306-312).
bench.py copies the record into cpu_baseline.reference_unmodified next to the oracle port's live number."""
from __future__ import annotations

import argparse
import json
import os
import re
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def child(data_root: str, epochs: int):
    sys.path.insert(0, HERE)
    import ref_loader
    ref = ref_loader.load_reference(["--dataset", "netflix_valid_item", "--data_path", data_root + "/", "--epoch", str(epochs), "--debug"], cwd="/tmp")
    import torch
    ref.set_seed(ref.args.seed)
    t0 = time.time()
    trainer = ref.Trainer(data_config={"n_users": ref.data_generator.n_users, "n_items": ref.data_generator.n_items})
    init_s = time.time() - t0
    lines = []
    orig = trainer.logger.logging
    trainer.logger.logging = lambda s: (lines.append(str(s)), orig(s))[1]
    # read-only monitors (round 5: the headline shape pinned against the reference itself): the unrounded metric dict of every
    # test_torch call (utility/batch_test.py:112-169); they change no arithmetic and consume no RNG
    evals = []
    orig_test_torch = ref.test_torch

    def test_torch_wrap(ua, ia, users_to_test, is_val, *a, **k):
        res = orig_test_torch(ua, ia, users_to_test, is_val, *a, **k)
        evals.append({m: [float(x) for x in res[m]] for m in ("precision", "recall", "ndcg", "hit_ratio")})
        return res
    ref.test_torch = test_torch_wrap
    t1 = time.time()
    trainer.train()
    total = time.time() - t1
    ep = []
    num = r"(-?[0-9.]+(?:e-?[0-9]+)?|nan|inf)"
    for l in lines:
        m = re.search(r"Epoch (\d+) \[([0-9.]+)s \+ ([0-9.]+)s\]: train==\[%s=%s \+ %s \+ %s\]" % (num, num, num, num), l)
        if m:
            # the evaluation whose result this line prints is the FIRST test_torch call of the epoch (main.py:299); a second one follows
            # when recall@20 improved (main.py:316)
            ep.append({"epoch": int(m.group(1)), "train_s": float(m.group(2)), "eval_s": float(m.group(3)), "line": l.strip(),
                       "loss": float(m.group(4)), "mf_loss": float(m.group(5)), "emb_loss": float(m.group(6)), "reg_loss": float(m.group(7))})
    # pair each epoch line with its evaluation: metrics printed to 5 decimals in the line identify the call
    k = 0
    for e in ep:
        r20 = float(re.search(r"recall=\[[-0-9.e]+, ([-0-9.e]+),", e["line"]).group(1))
        while k < len(evals) and abs(evals[k]["recall"][1] - r20) > 6e-6:
            k += 1
        if k < len(evals):
            e["metrics"] = evals[k]
            k += 1
    n_batch = ref.data_generator.n_train // ref.args.batch_size + 1
    print("REF_JSON " + json.dumps({"epochs": ep, "n_evaluations": len(evals), "args": {k: (v if isinstance(v, (int, float, str, bool)) else str(v)) for k, v in vars(ref.args).items()}, "n_batch": n_batch, "batch_size": ref.args.batch_size, "n_train": ref.data_generator.n_train,
                                    "n_test_users": len(ref.data_generator.test_set), "init_s": init_s, "train_call_s": total,
                                    "torch_threads": torch.get_num_threads(), "torch": torch.__version__}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=2)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r05_reference_cpu.json"))
    ap.add_argument("--data", default="/tmp/llmrec_e2e")
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--note", default=None, help="free text stored as timing_note (e.g. the state of the host during the run)")
    a = ap.parse_args()
    if a.child:
        return child(a.data, a.epochs)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import e2e_main
    ds, stats = e2e_main.write_dataset(a.data)
    sys.path.insert(0, HERE)
    import make_trajectory
    dig = make_trajectory.digests(ds)                        # sha256 of the array content of every input file: the consumer trains on the same bytes
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--data", a.data, "--epochs", str(a.epochs)], capture_output=True, text=True, cwd="/tmp")
    js = [l for l in r.stdout.splitlines() if l.startswith("REF_JSON ")]
    if not js:
        raise SystemExit("reference run failed:\n" + (r.stderr or r.stdout)[-3000:])
    c = json.loads(js[-1][9:])
    best = min(c["epochs"], key=lambda e: e["train_s"])
    out = {"what": "the UNMODIFIED reference (/root/reference main.py, imported through oracle/ref_loader.py) on CPU, its own Trainer.train() and timers",
           "dataset": "tools/e2e_main.py's NF-shaped set: U 13187 x I 17366, 68933 interactions (%d train), %d test users, feats 512/768/1536 x (1 + 5)" % (c["n_train"], c["n_test_users"]),
           "host": {"cores": os.cpu_count(), "torch_threads": c["torch_threads"], "eval_pool": "in-process map (the reference's Pool(cpu_count() // 5) = 1 worker on this host)",
                    "torch": c["torch"], "where": "build container (no GPU)"},
           "digests": dig, "seed": c["args"].get("seed"), "args": c["args"], "n_evaluations": c["n_evaluations"],
           "epochs": c["epochs"], "n_batch": c["n_batch"], "batch_size": c["batch_size"], "n_test_users": c["n_test_users"],
           "train_s": best["train_s"], "eval_s": best["eval_s"],
           "edges_per_s": c["n_batch"] * c["batch_size"] / best["train_s"], "users_per_s": c["n_test_users"] / best["eval_s"], "init_s": c["init_s"]}
    if a.note:
        out["timing_note"] = a.note
    json.dump(out, open(a.out, "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
