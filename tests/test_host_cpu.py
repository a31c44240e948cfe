"""CPU tests of the host side: C-ABI symbols, CLI surface, dataset container, metrics,
drop-in initialisation (no compute call is made without a GPU)."""
import ctypes
import os
import random

import numpy as np
import pytest
import torch

from llmrec_amd import _lib
from tests._dropin import load_dropin, golden_argv


def test_library_exports_every_header_symbol():
    protos = _lib.parse_header()
    assert len(protos) >= 25
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in protos:
        assert hasattr(lib, name), name
    lib2 = _lib.load()
    assert lib2.llmrec_abi_version() == _lib.CONST["LLMREC_ABI_VERSION"]
    assert lib2.llmrec_status_string(-3) == b"workspace too small"


def test_workspace_queries_run_without_gpu():
    assert _lib.query("llmrec_csr_build_workspace_bytes", 10, 1000) > 16 * 1000
    assert _lib.query("llmrec_linear_wgrad_workspace_bytes", 1000, 64, 512) >= 4 * 64 * 512
    assert _lib.query("llmrec_sumsq_workspace_bytes", 10, 10) > 0
    # top-K: 825 user tiles on 256 compute units (the default when no device answers) leave 57 tiles, cut in 4 parts each
    # the item table in fragment order: fp32 fragments (exact sweep) or bf16 (hi, mid) fragments with d padded to 32 (prefilter sweep),
    # whichever is larger, + the bf16 mode's 256-byte header, per-item factors (4 bytes per padded item) and one flag word per user tile
    up = lambda x: -(-x // 256) * 256
    frag = lambda n_items, d: max(-(-n_items // 32) * 2 * -(-d // 16) * 64 * 16, -(-n_items // 32) * 2 * -(-d // 32) * 2 * 64 * 16) + 256 + up(4 * -(-n_items // 32) * 32)
    packed = lambda n_items, d, n_query=None: frag(n_items, d) + (up(4 * -(-n_query // 16)) if n_query is not None else 0)
    # + the train rows as bitmaps, one word per item tile: all 16 rows of every block of the sweep (tile-major slices) while that fits 64 MB,
    # else the two longest rows of a block (off beyond 131 072 items / 64 MB)
    heavy = lambda blocks, n_items, rows=16: up(blocks * rows * -(-n_items // 32) * 4)
    assert _lib.query("llmrec_score_topk_workspace_bytes", 13187, 17366, 64) == 57 * 4 * 16 * 64 * 8 + packed(17366, 64, 13187) + heavy(768 + 57 * 4, 17366)
    # beyond 131 072 items the bf16 sweep cuts EVERY user tile into item parts (round 6): part lists for the smallest part (8 192 items) of any width
    assert _lib.query("llmrec_score_topk_workspace_bytes", 4096 * 16, 1_000_000, 64) == 4096 * 123 * 16 * 64 * 8 + packed(1_000_000, 64, 4096 * 16)
    assert _lib.query("llmrec_score_topk_workspace_bytes", 4096 * 16, 1_000_000, 128) == 4096 * 123 * 16 * 64 * 8 + packed(1_000_000, 128, 4096 * 16)
    assert _lib.query("llmrec_score_topk_workspace_bytes", 100, 500, 20) == packed(500, 20, 100) + heavy(7, 500)        # too few items to cut
    assert _lib.query("llmrec_score_topk_workspace_bytes", 16 * 4096, 131_072, 64) == packed(131_072, 64, 16 * 4096)      # 128 MB of two-row slices: off
    assert _lib.query("llmrec_score_topk_workspace_bytes", 16 * 1024, 65_536, 64) == packed(65_536, 64, 16 * 1024) + heavy(1024, 65_536, rows=2)   # 16 rows: 128 MB; two rows: 16 MB
    assert _lib.query("llmrec_score_topk_workspace_bytes", -1, 10, 64) == -1


def test_multi_target_weight_gradient_plan_without_gpu():
    """llmrec_linear_wgrad_multi_*: the fast-path test and the workspace size are host arithmetic on the argument block."""
    from llmrec_amd import ops
    lib = _lib.load()
    probs = (ops.WgradProblem * 2)()
    for j in range(2):
        probs[j].dY, probs[j].lddy, probs[j].X, probs[j].ldx, probs[j].M = 0x1000, 448, 0x2000, 1536, 17366
    tg = (ops.WgradTarget * 2)()
    tg[0].n_problems, tg[0].problems, tg[0].K, tg[0].dW, tg[0].lddw, tg[0].db, tg[0].accumulate = 2, ctypes.cast(probs, ctypes.c_void_p), 1536, 0x3000, 1536, 0x4000, 0
    tg[1].n_problems, tg[1].problems, tg[1].K, tg[1].dW, tg[1].lddw, tg[1].db, tg[1].accumulate = 1, ctypes.cast(probs, ctypes.c_void_p), 1536, 0x5000, 1536, None, 1
    need = _lib.query("llmrec_linear_wgrad_multi_workspace_bytes", 2, tg, 64)
    assert need >= 2 * 4 * 64 * 1536
    assert lib.llmrec_linear_wgrad_multi_bf16x3(2, tg, 64, None, 0, None) == -3                  # workspace too small
    tg[1].K = 100                                                                               # K % 64 != 0: outside the fast path
    assert _lib.query("llmrec_linear_wgrad_multi_workspace_bytes", 2, tg, 64) == -1
    assert lib.llmrec_linear_wgrad_multi_bf16x3(2, tg, 64, None, 0, None) == -4 and b"linear_wgrad_multi" in lib.llmrec_last_error()
    assert _lib.query("llmrec_linear_wgrad_multi_workspace_bytes", 5, tg, 64) == -1            # more than LLMREC_WGRAD_MAX_TARGETS


def test_argument_errors_are_reported_not_thrown():
    lib = _lib.load()
    st = lib.llmrec_spmm_f32(4, 4, None, None, None, None, None, None, 8, None, 8, 8, 0, None, None, None, None)
    assert st == -1 and b"spmm" in lib.llmrec_last_error()
    with pytest.raises(RuntimeError, match="invalid argument"):
        _lib.call("llmrec_degree_scale", -1, None, None, None)


def test_argument_errors_of_the_later_entry_points():
    """Every entry point validates its arguments before touching the device (status code + llmrec_last_error, no throw,
    no launch): exercised here without a GPU for the entry points added after the first ABI draft."""
    lib = _lib.load()
    cases = [
        ("llmrec_sample_batch", (1, None, 10, None, 5, None, None, 8, 0, 16, 0, None, None, None, None, None, None, None), b"sample_batch"),   # slice larger than the global batch
        ("llmrec_topk_metrics", (4, None, 50, None, None, None, 9, None, None, None), b"topk_metrics"),                                       # more than 8 cut-offs
        ("llmrec_bpr_multi_fwd_sharded_f32", (1, None, 64, None, None, None, 8, None, 0.3, 1e-5, 64.0, 3, None, None, 1, 0, 0, None, None, None), b"bpr_multi_fwd_sharded"),
        ("llmrec_linear_wgrad_grouped_bf16x3", (0, None, 64, 64, None, 64, None, 0, None, 0, None), b"linear_wgrad"),
        ("llmrec_fuse_bwd_f32", (4, 8, None, 8, 2, None, None, None, None, None, 0, 5, 0.0, None), b"fuse_bwd"),                              # n_reg_terms > n_norm
    ]
    for name, args, needle in cases:
        st = getattr(lib, name)(*args)
        assert st != 0, name
        assert needle in lib.llmrec_last_error(), (name, lib.llmrec_last_error())


def test_ops_refuse_cpu_tensors():
    from llmrec_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.softmax_rows(torch.zeros(4, 8))


def test_dropin_init_matches_reference_init(golden):
    """Same seed -> bit-identical initial parameters and the same sample stream as the reference."""
    m = load_dropin(golden_argv(golden))
    m.set_seed(golden.args["seed"])
    tr = m.Trainer(data_config={})
    init = golden.init_params()
    sd = tr.model_mm.state_dict()
    assert set(init) == set(sd)
    for k, v in init.items():
        assert np.array_equal(sd[k].cpu().numpy(), v), k
    if not torch.cuda.is_available():
        # sample stream (host sampler + aug triples), identical to the reference's
        for s in range(golden.n_steps):
            u, p, n = (t.tolist() for t in tr.sample_batch())
            assert u == golden.z["step%d/users" % s].tolist()
            assert p == golden.z["step%d/pos" % s].tolist()
            assert n == golden.z["step%d/neg" % s].tolist()
    assert m.data_generator.n_users == golden.meta["config"]["n_users"]
    assert m.data_generator.n_items == golden.meta["config"]["n_items"]
    assert (tr.n_users, tr.n_items) == (m.data_generator.n_users, m.data_generator.n_items)
    # graph tensors: values are diag(deg^-1/2) R
    ui = tr.ui_graph.coalesce()
    deg = torch.bincount(ui.indices()[0], minlength=tr.n_users).float()
    assert torch.allclose(ui.values(), deg[ui.indices()[0]].rsqrt(), rtol=1e-6)


def test_metrics_vectorised_equals_scalar():
    import utility.metrics as M
    rng = np.random.default_rng(0)
    hits = (rng.random((40, 50)) < 0.08).astype(np.uint8)
    hits[3] = 0
    n_pos = rng.integers(1, 6, size=40)
    Ks = [10, 20, 50]
    vec = M.metrics_from_hit_matrix(hits, n_pos, Ks)
    for u in range(40):
        r = hits[u].tolist()
        for j, K in enumerate(Ks):
            assert vec["precision"][u, j] == pytest.approx(M.precision_at_k(r, K), abs=1e-15)
            assert vec["recall"][u, j] == pytest.approx(M.recall_at_k(r, K, n_pos[u]), abs=1e-15)
            assert vec["ndcg"][u, j] == pytest.approx(M.ndcg_at_k(r, K), abs=1e-15)
            assert vec["hit_ratio"][u, j] == M.hit_at_k(r, K)


def test_synth_generator_exact_and_duplicate_free():
    from llmrec_amd.synth import bipartite_edges
    r, c = bipartite_edges(300, 200, 2500, seed=1, max_deg=150)
    assert r.size == 2500 and np.unique(r * 200 + c).size == 2500
    assert np.bincount(r, minlength=300).min() >= 1


def test_device_generator_gives_exactly_the_requested_distinct_edges():
    """"no duplicate (u, i); E exact" - the cfg 4 / cfg 5 generator (run here on the CPU device)."""
    import torch
    from llmrec_amd import synth
    for U, I, E, seed in ((20000, 5000, 400000, 1), (3000, 500, 200000, 2), (64, 4096, 70000, 3)):
        r, c = synth.bipartite_edges_device(U, I, E, seed, "cpu")
        key = r * I + c
        assert r.numel() == E and bool((key[1:] > key[:-1]).all())          # exact count, sorted, distinct
        assert int(r.min()) >= 0 and int(r.max()) < U and int(c.min()) >= 0 and int(c.max()) < I
        pop = torch.bincount(c, minlength=I).double()
        if seed == 1:                                                         # (the denser cases saturate the popular items)
            assert float(pop.max()) > 8 * float(pop.mean())                   # the popularity skew survives the top-up
    r2, c2 = synth.bipartite_edges_device(20000, 5000, 400000, 1, "cpu")
    r1, c1 = synth.bipartite_edges_device(20000, 5000, 400000, 1, "cpu")
    assert torch.equal(r1, r2) and torch.equal(c1, c2)                        # a function of the seed


def test_bench_finds_its_committed_profile_data():
    """bench.py quotes three things from profiles/: the PMC traffic of the roofline kernels, their in-step rocprofv3 averages and
    the per-class kernel time. The look-ups go by kernel NAME - a renamed kernel (or a stale profile) must fail here, not turn
    `roofline.traffic` into null in the driver's bench line."""
    import re
    import bench
    src = open(bench.__file__).read()
    # the (kernel-name substring, launches) lists kernel_rooflines() hands to the look-ups
    names = set(re.findall(r'"((?:linear|reduce)_[a-z0-9_]+kernel)"', src))
    assert {"linear_fwd_grouped_bf16x3_kernel", "linear_wgrad_bf16x3_v2_multi_kernel", "reduce_chunks_multi_kernel"} <= names
    for key in sorted(names - {"linear_fwd_grouped_kernel"}):            # (the exact-fp32 projection is not the default step's kernel)
        byts, src_info = bench.pmc_traffic_bytes([(key, 1)])
        assert byts is not None and byts > 0, (key, src_info)
        avg = bench.rocprof_avg_us([(key, 1)])
        assert avg is not None and avg["avg_us"] > 0, key
    shares = bench.kernel_time_shares()
    assert shares is not None and "spmm_kernel" in shares["classes"] and shares["file"].startswith("r06_")
    # and the kernels exist under these names in the library's source
    csrc = open(os.path.join(os.path.dirname(_lib.HEADER), "..", "llmrec_amd", "csrc", "dense.hip")).read()
    for key in names:
        assert key in csrc, key


def test_bench_launcher_decision():
    """`python bench.py --gpus N` (no launcher in the environment) must become N ranks or refuse - never a 1-GPU line labelled N
    (VERDICT r03 missing #2). The decision is a pure function of (--gpus, environment, visible devices)."""
    import bench
    d = bench.launch_decision
    assert d(1, {}, 1) == ("run", 1)
    assert d(1, {}, 8) == ("run", 1)
    assert d(8, {}, 8) == ("spawn", 8)
    assert d(2, {}, 8) == ("spawn", 2)
    assert d(8, {}, 1)[0] == "refuse" and d(2, {}, 0)[0] == "refuse"                      # fewer devices than ranks
    assert d(2, {"LLMREC_BENCH_SINGLE_DEVICE": "1"}, 1) == ("spawn", 2)                  # the 1-GPU test hook (gloo, all ranks on cuda:0)
    # under a launcher (the driver's torch.distributed.run command): the launcher's world counts, --gpus must agree with it
    assert d(8, {"WORLD_SIZE": "8"}, 8) == ("run", 8)
    assert d(1, {"WORLD_SIZE": "4"}, 8) == ("run", 4)                                     # --gpus left at its default
    assert d(8, {"WORLD_SIZE": "2"}, 8)[0] == "refuse"
    assert d(4, {"WORLD_SIZE": "4"}, 2)[0] == "refuse"
    assert d(2, {"WORLD_SIZE": "2", "LLMREC_BENCH_SINGLE_DEVICE": "1"}, 1) == ("run", 2)


def test_bench_spawn_command_is_the_drivers_launcher(monkeypatch):
    """spawn_ranks re-executes bench.py under torch.distributed.run with one rank per GPU on 127.0.0.1 and passes the flags through."""
    import subprocess
    import bench
    seen = {}
    monkeypatch.setattr(subprocess, "call", lambda cmd, env=None: seen.update(cmd=cmd, env=env) or 0)
    assert bench.spawn_ranks(4, ["--gpus", "4", "--steps", "7"]) == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-5:] == [os.path.abspath(bench.__file__), "--gpus", "4", "--steps", "7"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_block_sampler_and_c_helper_reproduce_the_per_call_stream(monkeypatch):
    """Data.sample(): the block form (raw MT19937 words replayed in Python) and the C helper (llmrec_amd/csrc/host_sampler.c) draw the
    same (users, pos, neg) as the reference's per-call np.random.randint form AND leave both global streams where it leaves them -
    users with one item (no word consumed), hub users (many rejections), item counts at and just past a power of two."""
    import random
    import sys
    monkeypatch.setattr(sys, "argv", ["main.py", "--dataset", "netflix_valid_item"])
    sys.modules.pop("utility.load_data", None)
    import utility.load_data as LD
    from llmrec_amd import build as _build
    _build.build_host(force=False)
    rng = np.random.default_rng(3)
    for n_items in (64, 65, 1000, 4097):
        n_users = 300
        D = LD.Data.__new__(LD.Data)
        D.batch_size, D.n_users, D.n_items = 128, n_users, n_items
        D.train_items = {}
        for u in range(n_users):
            deg = 1 if u % 5 == 0 else (n_items * 3 // 4 if u % 97 == 1 else int(rng.integers(1, 12)))
            D.train_items[u] = rng.choice(n_items, size=deg, replace=False).tolist()
        D.exist_users = list(D.train_items)
        D._train_sets = {u: set(v) for u, v in D.train_items.items()}
        results = {}
        for mode in ("reference", "python_block", "c_helper"):
            D._fast_sampler, D._host = (False if mode == "reference" else None), (False if mode == "python_block" else None)
            D._fast_users, D._users_scratch, D._exist_arr = (None if mode == "c_helper" else False), None, None
            np.random.seed(11); random.seed(11)
            out = [D.sample() for _ in range(6)]
            if mode == "c_helper":
                assert D._host not in (None, False), "libllmrec_host.so was not loaded"
                assert D._fast_users and all(v is True for v in D._fast_users.values())   # the C replay of random.sample agreed with the interpreter's, on each branch hit
            if mode != "reference":
                assert D._fast_sampler is True
            results[mode] = ([(u, p, [int(x) for x in n]) for u, p, n in out], np.random.get_state()[1].tolist(), np.random.get_state()[2], random.getstate())
        assert results["python_block"] == results["reference"]
        assert results["c_helper"] == results["reference"]


def test_py_sample_replay_is_verified_on_each_branch_of_random_sample(monkeypatch):
    """CPython's random.sample has two branches (pool: n <= setsize; set: larger populations). The production stream hits the set
    branch first (exist_users, n = 13 187, k = 1 024) and the pool branch every step after it (the augmented-triple draw, n = B, k = B / 10):
    the C replay is cross-checked against the interpreter the first time EACH branch is taken, and every draw equals random.sample's."""
    import random
    import sys
    monkeypatch.setattr(sys, "argv", ["main.py", "--dataset", "netflix_valid_item"])
    sys.modules.pop("utility.load_data", None)
    import utility.load_data as LD
    from llmrec_amd import build as _build
    _build.build_host(force=False)
    D = LD.Data.__new__(LD.Data)
    D._host, D._fast_users, D._users_scratch, D._exist_arr = None, {}, None, None
    D.n_users, D.train_items = 4, {0: [1], 1: [2, 3]}                # (what _host_lib lays out beside loading the helper)
    big, small = list(range(3, 13190)), list(range(100, 1124))      # n = 1024, k = 102: setsize = 21 + 4^5 = 1045 >= n -> the pool branch
    calls = [(big, 1024), (small, 102), (small, 102), (big, 1024), (small, 7), (big, 5)]
    random.seed(2022)
    want = [random.sample(p, k) for p, k in calls]
    state_want = random.getstate()
    random.seed(2022)
    got = []
    for i, (p, k) in enumerate(calls):
        got.append(D.py_sample(p, k))
        if i == 0:
            assert D._host not in (None, False), "libllmrec_host.so was not loaded"
            assert D._fast_users == {False: True}                  # only the set branch has been verified so far
        if i == 1:
            assert D._fast_users == {False: True, True: True}      # ... and now the pool branch, by its own draw-both-and-compare
    assert got == want and random.getstate() == state_want
    # a branch that fails its check is switched off alone
    D._fast_users = {False: True, True: False}
    random.seed(5); a = D.py_sample(small, 102); random.seed(5); b = random.sample(small, 102)
    assert a == b and D._fast_users == {False: True, True: False}


def test_host_helper_library_exports_what_its_header_declares():
    """include/llmrec_host.h <-> llmrec_amd/lib/libllmrec_host.so (plain C, gcc): both entry points load."""
    import ctypes
    import re
    from llmrec_amd import build as _build
    so = _build.build_host(force=False)
    hdr = open(os.path.join(os.path.dirname(_lib.HEADER), "llmrec_host.h")).read()
    names = re.findall(r"\b(llmrec_host_\w+)\s*\(", hdr)
    assert set(names) == {"llmrec_host_draw_items", "llmrec_host_py_sample"}
    lib = ctypes.CDLL(so)
    for n in names:
        getattr(lib, n)


def test_hardware_queue_pool_is_widened_before_hip_initialises():
    """llmrec_amd/__init__.py (and tests/conftest.py, bench.py, main.py, __graft_entry__.py) set GPU_MAX_HW_QUEUES - the work-around for
    hipGraphLaunch's unchecked walk over an executable's internal streams - unless the user chose a value."""
    import llmrec_amd                                             # noqa: F401
    assert int(os.environ.get("GPU_MAX_HW_QUEUES", "0")) >= 8
    for f in ("bench.py", "main.py", "__graft_entry__.py", os.path.join("tests", "conftest.py")):
        src = open(os.path.join(os.path.dirname(_lib.HEADER), "..", f)).read()
        assert 'os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")' in src, f
        assert src.index("GPU_MAX_HW_QUEUES") < (src.index("import torch") if "import torch" in src else len(src)), f


def test_queue_work_around_decision_is_loud():
    """VERDICT r04 next #8 / ADVICE r04: the work-around only holds if GPU_MAX_HW_QUEUES >= 8 is in the environment before the HIP runtime
    initialises; when it cannot hold, graph replay is refused (not silently left to fault)."""
    import llmrec_amd
    qd = llmrec_amd.queue_decision
    assert qd(None, False) == ("set", True, None)                       # the normal import: set it
    act, safe, msg = qd(None, True)                                      # the embedder touched the GPU first
    assert (act, safe) == ("keep", False) and "after the HIP runtime initialised" in msg
    act, safe, msg = qd("4", False)                                      # a smaller user value is kept, and flagged
    assert (act, safe) == ("keep", False) and "below 8" in msg
    assert qd("4", True)[1] is False and qd("junk", False)[1] is False
    assert qd("8", False) == ("keep", True, None) and qd("16", True) == ("keep", True, None)
    assert llmrec_amd.graph_replay_safe()                                # this process: conftest set it before torch was imported
    # a process that initialised torch.cuda first would be refused a capture; simulate the module state
    saved = (llmrec_amd._graph_safe, llmrec_amd._message)
    llmrec_amd._graph_safe, llmrec_amd._message = False, "simulated"
    try:
        assert not llmrec_amd.graph_replay_safe()
        try:
            llmrec_amd.require_graph_replay("test")
            raise AssertionError("capture was not refused")
        except RuntimeError as e:
            assert "refused" in str(e)
        os.environ["LLMREC_UNSAFE_GRAPH"] = "1"
        try:
            llmrec_amd.require_graph_replay("test")                      # the explicit override
        finally:
            del os.environ["LLMREC_UNSAFE_GRAPH"]
    finally:
        llmrec_amd._graph_safe, llmrec_amd._message = saved


def test_topk_key_order_is_the_reference_rank_rule():
    """csrc/topk.hip sorts its lists as one 64-bit key per entry, key = tk_ord(score) << 32 | ~item (tk_key): restated here in numpy, a larger
    key must mean (score desc, item id asc) - the reference's rank rule (utility/batch_test.py:21-36) - for every float incl. +-0 / +-inf /
    denormals, the empty slot (-inf, INT_MAX) must be the smallest key a list can hold, and the mapping must be a bijection on the score bits."""
    rng = np.random.default_rng(0)
    bits = np.concatenate([rng.integers(0, 2 ** 32, size=20000, dtype=np.uint64).astype(np.uint32),
                           np.array([0x00000000, 0x80000000, 0x7f800000, 0xff800000, 0x00000001, 0x80000001, 0x007fffff, 0x3f800000, 0xbf800000], dtype=np.uint32)])
    x = bits.view(np.float32)
    keep = ~np.isnan(x)                                        # (NaN scores never reach a list: the filter compare rejects them)
    bits, x = bits[keep], x[keep]
    sign = (bits.view(np.int32) >> 31).view(np.uint32)
    ordv = bits ^ (sign | np.uint32(0x80000000))               # tk_ord
    unord = np.where(ordv & np.uint32(0x80000000), ordv ^ np.uint32(0x80000000), ~ordv)
    assert np.array_equal(unord, bits)                         # tk_unord(tk_ord(x)) == x, bit for bit
    a, b = rng.integers(0, len(x), size=(2, 50000))
    lt = x[a] < x[b]
    assert np.all(ordv[a][lt] < ordv[b][lt])
    eq = x[a] == x[b]                                          # only -0 / +0 compare equal with different bits: -0 orders below +0 (never produced by the sweeps' fma chains)
    assert np.all((ordv[a][eq] == ordv[b][eq]) | ((x[a][eq] == 0) & (x[b][eq] == 0)))
    ids = rng.integers(0, 2 ** 31 - 1, size=len(x), dtype=np.int64)
    key = (ordv.astype(np.uint64) << np.uint64(32)) | (~ids.astype(np.uint32)).astype(np.uint64)
    same = x[a] == x[a]                                        # same score, different ids: the smaller id is the better (larger) key
    k1 = (ordv[a].astype(np.uint64) << np.uint64(32)) | (~ids[a].astype(np.uint32)).astype(np.uint64)
    k2 = (ordv[a].astype(np.uint64) << np.uint64(32)) | (~ids[b].astype(np.uint32)).astype(np.uint64)
    d = ids[a] != ids[b]
    assert np.all((k1[d] > k2[d]) == (ids[a][d] < ids[b][d])) and same.all()
    empty = (np.uint64(0x007FFFFF) << np.uint64(32)) | np.uint64(0x80000000)        # TK_KEY_EMPTY = tk_key(-inf, INT_MAX)
    ninf = np.array([0xff800000], dtype=np.uint32)
    assert int((ninf ^ np.uint32(0xffffffff))[0]) == 0x007FFFFF and np.all(key >= empty)
