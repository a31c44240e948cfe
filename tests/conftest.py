"""pytest configuration: registers the ``gpu`` marker and puts the repo root on sys.path."""
import json
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # before torch / HIP initialise: see llmrec_amd/__init__.py

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
GOLDEN_CASES = ["nf_tiny", "nf_tiny_noaug", "ml_tiny"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _install_segv_bt():
    """Debug aid: LLMREC_SEGV_BT=<file> -> a native backtrace of a host-side SIGSEGV goes to that file (tools/dbg/segv_bt.c)."""
    path = os.environ.get("LLMREC_SEGV_BT")
    so = os.path.join(ROOT, "tools", "dbg", "segv_bt.so")
    if path and os.path.exists(so):
        import ctypes
        ctypes.CDLL(so).segv_bt_install(path.encode())


@pytest.fixture(autouse=True)
def _release_gpu_objects(request):
    """After every GPU test: collect cyclic garbage NOW (a test's Trainer / FusedStep / captured HIP graphs sit in reference cycles
    through the re-imported drop-in modules) and hand cached blocks back, so that graph executables and their private pools do not pile
    up over the ~150 tests of the suite. (Hygiene; the host fault of round 4 was hipGraphLaunch's own: llmrec_amd/__init__.py.)"""
    _install_segv_bt()                                       # (re-installed per test: runtimes loaded later may have replaced the handler)
    yield
    if request.node.get_closest_marker("gpu") is not None:
        import gc
        gc.collect()
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.synchronize()
                torch.cuda.empty_cache()
        except Exception:
            pass


class GoldenCase:
    def __init__(self, name):
        self.name = name
        self.dir = os.path.join(GOLDEN, name)
        self.meta = json.load(open(os.path.join(self.dir, "meta.json")))
        self.z = np.load(os.path.join(self.dir, "golden.npz"))
        self.args = self.meta["args"]
        self.dataset = self.meta["config"]["dataset"]
        self.data_dir = os.path.join(self.dir, "data", self.dataset)
        self.n_steps = int(self.z["n_steps"])

    def detail_steps(self):
        return sorted({int(k.split("/")[0][4:]) for k in self.z.files if k.startswith("step") and k.endswith("/E_u")})

    def init_params(self):
        return {k[5:]: self.z[k] for k in self.z.files if k.startswith("init/")}


@pytest.fixture(params=GOLDEN_CASES)
def golden(request):
    return GoldenCase(request.param)
