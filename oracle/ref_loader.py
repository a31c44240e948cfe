"""TEST INFRASTRUCTURE ONLY - loads the UNMODIFIED reference (/root/reference) on CPU.

Only ``oracle/make_golden.py`` (run in the build container, where /root/reference exists) uses
this. Nothing under ``llmrec_amd/``, ``main.py``, ``Models.py`` or ``utility/`` may import it.
/root/reference does not exist on the GPU box; the vectors it produces travel as fixtures under
``tests/golden/``.

The reference hard-codes ``.cuda()`` and imports two things this image lacks, so four
import-time shims are installed. None of them touches arithmetic:

1. ``setproctitle``           - stub module (imported at reference main.py:32, unused).
2. ``np.asfarray``            - removed in numpy 2; used at reference utility/metrics.py:50,75.
3. ``Tensor.cuda/Module.cuda``- identity (reference main.py:96-97,134,159; Models.py:43-48,111).
4. ``torch.cuda.manual_seed_all`` - no-op without a GPU (reference main.py:359).

and one process-model shim so ranked lists can be captured in-process:

5. ``multiprocessing.Pool``   - in-process map (reference utility/batch_test.py:115,157 uses
   ``Pool(cpu_count()//5)``, which is ``Pool(1)`` on the 8-core build host and ``Pool(0)`` ->
   ValueError below 5 cores).
"""
from __future__ import annotations

import importlib
import os
import sys
import types

REFERENCE_ROOT = "/root/reference"


class _InProcessPool:
    def __init__(self, *a, **k):
        pass

    def map(self, fn, it):
        return [fn(x) for x in it]

    def close(self):
        pass

    def join(self):
        pass


def install_shims():
    import numpy as np
    import torch
    import multiprocessing

    sys.modules.setdefault("setproctitle", types.ModuleType("setproctitle"))
    if not hasattr(np, "asfarray"):
        np.asfarray = lambda a, dtype=np.float64: np.asarray(a, dtype=dtype)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    torch.cuda.manual_seed_all = lambda seed: None
    multiprocessing.Pool = _InProcessPool


def load_reference(argv, cwd=None):
    """Import the reference's ``main`` module with ``sys.argv = ['main.py'] + argv``.

    Returns the imported module (its ``__main__`` block does not run). The repo root is removed
    from sys.path first so the reference's ``utility``/``Models`` are the ones imported."""
    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError("reference tree not present; golden vectors can only be regenerated "
                           "in the build container")
    install_shims()
    repo_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:] = [p for p in sys.path if os.path.abspath(p or ".") != repo_root]
    for name in ("main", "Models", "utility", "utility.parser", "utility.batch_test",
                 "utility.load_data", "utility.metrics", "utility.logging", "utility.norm"):
        sys.modules.pop(name, None)
    sys.path.insert(0, REFERENCE_ROOT)
    sys.argv = ["main.py"] + list(argv)
    if cwd:
        os.chdir(cwd)
    return importlib.import_module("main")
