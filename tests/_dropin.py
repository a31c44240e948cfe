"""Imports the repo's drop-in modules (main / Models / utility.*) the way the reference is driven:
sys.argv is set first, because four modules parse it at import time (reference main.py:34,
Models.py:15, utility/batch_test.py:13, utility/load_data.py:8)."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_MODS = ("main", "Models", "utility.batch_test", "utility.load_data")


def load_dropin(argv):
    for name in _MODS:
        sys.modules.pop(name, None)
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    old = sys.argv
    sys.argv = ["main.py"] + list(argv)
    try:
        return importlib.import_module("main")
    finally:
        sys.argv = old
        import gc
        gc.collect()          # the modules of the previous import (their data_generator holds device tensors) go NOW, not inside a later stream capture


def golden_argv(g):
    """CLI that reproduces a golden case's configuration."""
    a = g.args
    argv = ["--dataset", g.dataset, "--data_path", os.path.join(g.dir, "data") + "/", "--debug"]
    for k in ("batch_size", "epoch", "seed", "embed_size", "weight_size", "layers", "aug_sample_rate",
              "prune_loss_drop_rate", "lr", "regs", "Ks"):
        argv += ["--" + k, str(a[k])]
    return argv
