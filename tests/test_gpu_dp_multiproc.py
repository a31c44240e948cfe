"""Two PROCESSES of the batch-sharded-replica step (llmrec_amd/dp.py) on one GPU: torch.distributed with the gloo
backend moving the two exchange buffers (device tensors staged through the host), everything else - sampler slices,
HIP kernels, the three graph segments, rank > 0 code paths - exactly what `bench.py --gpus N` runs over RCCL.
Checked against FusedStep on the concatenated global batch in a single process."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

from tests._dropin import load_dropin, golden_argv
from tests.conftest import GoldenCase, GOLDEN_CASES

STEPS = int(os.environ.get("LLMREC_TEST_DP_STEPS", "6"))
NAMES = ["item_trans.weight", "user_trans.bias", "image_trans.weight", "user_id_embedding.weight", "item_id_embedding.weight"]


def _build(case, cls, b_max, **kw):
    from llmrec_amd import ops
    golden = GoldenCase(case)
    m = load_dropin(golden_argv(golden))
    m.set_seed(golden.args["seed"])
    tr = m.Trainer(data_config={})
    graph = type("G", (), {"ui": ops.operand_from_sparse_tensor(tr.ui_graph), "iu": ops.operand_from_sparse_tensor(tr.iu_graph)})
    a = m.args
    step = cls(tr.model_mm, graph, tr.hyper, (a.model_cat_rate, a.user_cat_rate, a.item_cat_rate), tr.optimizer, b_max, **kw)
    return golden, m, tr, step


def _batcher(m, tr, rank, world, B):
    from llmrec_amd import engine
    st = m.data_generator.device_state(torch.device("cuda"))
    n_users = tr.n_users
    rng = np.random.default_rng(99)
    ap = torch.tensor(rng.integers(0, int(tr.n_items * 1.2), size=n_users)).cuda()
    an = torch.tensor(rng.integers(0, int(tr.n_items * 1.2), size=n_users)).cuda()
    return engine.DeviceBatcher(st["train"], st["exist_users"], tr.n_items, B, ap, an, 0.25, 4242, rank=rank, world=world)


def _worker(rank, world, port, case, out_dir, use_graph):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from llmrec_amd import dist as ld
    from llmrec_amd.dp import DataParallelStep
    golden = GoldenCase(case)
    B = 64
    cap = B + int(B * 0.25)
    golden, m, tr, step = _build(case, DataParallelStep, cap, comm=ld.Comm())
    batcher = _batcher(m, tr, rank, world, B)
    losses, seen = [], []
    for s in range(STEPS):
        if use_graph:
            if step.graphs is None:
                step.capture(batcher=batcher)                   # the capture's warm-up is a complete step
                losses.append(step.scal[1:4].clone().cpu().numpy())
            else:
                step.step()                                     # AdamW deferred into the next step's first graph:
                if s >= 2:                                      # the scalars visible now are those of step s - 1
                    losses.append(step.scal[1:4].clone().cpu().numpy())
        else:
            u, p, n, nv = batcher.next()
            out = torch.stack(step.step_eager(u, p, n, nv))
            seen.append((u.cpu(), p.cpu(), n.cpu(), int(nv)))
            losses.append(out.cpu().numpy())
    if use_graph:
        step.flush()                                            # the last step's update and scalars
        if STEPS >= 2:
            losses.append(step.scal[1:4].clone().cpu().numpy())
    assert len(losses) == STEPS
    params = dict(tr.model_mm.named_parameters())
    state = {}
    for k, p_ in params.items():                               # EVERY parameter and both Adam moments (bitwise replicas)
        state["p__" + k.replace(".", "_")] = p_.detach().cpu().numpy()
        st = tr.optimizer.state.get(p_)
        if st is not None:
            state["m__" + k.replace(".", "_")] = st[0].detach().cpu().numpy(); state["v__" + k.replace(".", "_")] = st[1].detach().cpu().numpy()
    np.savez(os.path.join(out_dir, "r%d.npz" % rank), losses=np.array(losses),
             **{k.replace(".", "_"): params[k].detach().cpu().numpy() for k in NAMES}, **state)
    torch.save(seen, os.path.join(out_dir, "batches%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("use_graph,world", [(False, 2), (True, 2), (True, 4), (False, 4)])
def test_two_process_replicas_match_single_process_global_batch(tmp_path, use_graph, world):
    """world = 4: four replicas on the one GPU (the exchange buffers hold four slices, the prune threshold covers four local batches)."""
    case = GOLDEN_CASES[0]
    mp.spawn(_worker, args=(world, _free_port(), case, str(tmp_path), use_graph), nprocs=world, join=True)
    r = [np.load(tmp_path / ("r%d.npz" % k)) for k in range(world)]
    full = [k for k in r[0].files if k[:3] in ("p__", "m__", "v__")]
    assert sum(k.startswith("m__") for k in full) >= 10
    for k in full:                                             # replicas bit-identical: every parameter, both Adam moments, after STEPS steps
        for j in range(1, world):
            assert np.array_equal(r[0][k].view(np.int32), r[j][k].view(np.int32)), (k, j)
    for j in range(1, world):
        assert np.allclose(r[0]["losses"], r[j]["losses"], rtol=0, atol=0)
    # single process, global batch = the ranks' slices concatenated (valid entries of rank 0, then of rank 1, ...)
    from llmrec_amd.fused import FusedStep
    B = 64
    cap = B + int(B * 0.25)
    golden, m, tr, single = _build(case, FusedStep, world * cap)
    batchers = [_batcher(m, tr, k, world, B) for k in range(world)]
    for s in range(STEPS):
        parts = [b.next() for b in batchers]
        u = torch.cat([p_[0][: int(p_[3])] for p_ in parts]); p = torch.cat([p_[1][: int(p_[3])] for p_ in parts])
        n = torch.cat([p_[2][: int(p_[3])] for p_ in parts])
        want = [float(x) for x in single.step_eager(u, p, n)]
        got = r[0]["losses"][s]
        for a_, b_ in zip(got, want):
            assert abs(a_ - b_) <= 2e-5 * abs(b_), (s, got, want)
    ref = dict(tr.model_mm.named_parameters())
    for k in NAMES:
        a_ = r[0][k.replace(".", "_")]; b_ = ref[k].detach().cpu().numpy()
        assert np.abs(a_ - b_).max() <= 1e-4 * np.abs(b_).max(), k
