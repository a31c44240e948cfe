"""pytest configuration: registers the ``gpu`` marker and puts the repo root on sys.path."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
GOLDEN_CASES = ["nf_tiny", "nf_tiny_noaug", "ml_tiny"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
