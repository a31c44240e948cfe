"""CPU, gloo, world_size 2 (and 3 with an uneven user partition): the user-sharded path of llmrec_amd/dist.py (partitioning, global
item degrees, one all-reduce per layer forward/backward, sharded BPR + prune over the global
batch) reproduces the single-process oracle after optimiser steps. The local kernels are the
torch stand-ins of tests/_cpu_backend.py; the HIP kernels themselves are checked on the GPU."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import oracle as O

U, I, D, L, B_LOCAL, STEPS = 60, 45, 16, 2, 12, 3
DROP, DECAY, LR = 0.71, 1e-5, 1e-2


def _problem(world=2, n_users=U):
    """world = 2, n_users = U: the problem of the two-rank tests (same random stream as ever). Other values: the users are cut into the
    contiguous blocks of llmrec_amd.dist.user_block (the last one shorter when world does not divide n_users)."""
    from llmrec_amd.dist import user_block
    rng = np.random.default_rng(0)
    rows = np.repeat(np.arange(n_users), rng.integers(1, 9, size=n_users))
    cols = np.concatenate([rng.choice(I, size=c, replace=False) for c in np.bincount(rows)])
    u_tab = (rng.standard_normal((n_users, D)) * 0.1).astype(np.float32)
    i_tab = (rng.standard_normal((I, D)) * 0.1).astype(np.float32)
    batches = []
    for s in range(STEPS):
        per_rank = []
        for r in range(world):
            u0, u1 = user_block(n_users, r, world)
            us = rng.integers(u0, u1, size=B_LOCAL)
            per_rank.append((us, rng.integers(0, I, size=B_LOCAL), rng.integers(0, I, size=B_LOCAL)))
        batches.append(per_rank)
    return rows, cols, u_tab, i_tab, batches


def _oracle_run(n_aug=0, world=2, n_users=U):
    import scipy.sparse as sp
    rows, cols, u_tab, i_tab, batches = _problem(world, n_users)
    batches = _with_aug(batches, n_aug)
    R = sp.csr_matrix((np.ones(rows.size, dtype=np.float32), (rows, cols)), shape=(n_users, I))
    a_ui, a_iu = O.normalized_graphs(R)
    pu = torch.tensor(u_tab, requires_grad=True); pi = torch.tensor(i_tab, requires_grad=True)
    opt = torch.optim.AdamW([{"params": [pu, pi]}], lr=LR)
    cfg = O.Config(batch_size=world * B_LOCAL, decay=DECAY, prune_loss_drop_rate=DROP)
    losses = []
    for per_rank in batches:
        us = np.concatenate([b[0] for b in per_rank]); ps = np.concatenate([b[1] for b in per_rank]); ns = np.concatenate([b[2] for b in per_rank])
        u, i = pu, pi
        ul, il = [u], [i]
        for l in range(L):
            u = torch.sparse.mm(a_ui, i)
            if l == L - 1: u = torch.softmax(u, -1)
            i = torch.sparse.mm(a_iu, u)
            if l == L - 1: i = torch.softmax(i, -1)
            ul.append(u); il.append(i)
        eu, ei = torch.mean(torch.stack(ul), 0), torch.mean(torch.stack(il), 0)
        mf, emb = O.bpr_loss(eu[torch.tensor(us)], ei[torch.tensor(ps)], ei[torch.tensor(ns)], cfg)
        loss = mf + emb
        opt.zero_grad(); loss.backward(); opt.step()
        losses.append(float(loss))
    return pu.detach().numpy(), pi.detach().numpy(), losses


def _worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from llmrec_amd import dist as ld
    from tests._cpu_backend import CpuBackend
    rows, cols, u_tab, i_tab, batches = _problem()
    comm, be = ld.Comm(), CpuBackend()
    u0, u1 = ld.user_block(U, rank, world)
    sel = (rows >= u0) & (rows < u1)
    g = ld.ShardedGraph.build(torch.tensor(rows[sel] - u0), torch.tensor(cols[sel]), u1 - u0, I, u0, comm, be)
    model = ld.ShardedIDModel(g, comm, be, D, L, U, seed=1)
    with torch.no_grad():
        model.user_id_embedding.copy_(torch.tensor(u_tab[u0:u1])); model.item_id_embedding.copy_(torch.tensor(i_tab))
    tr = ld.ShardedTrainer(model, LR, B_LOCAL, DROP, DECAY, seed=1)
    losses = []
    for per_rank in batches:
        us, ps, ns = per_rank[rank]
        loss, _ = tr.step((torch.tensor(us - u0), torch.tensor(ps), torch.tensor(ns)))
        losses.append(float(loss))
    np.savez(os.path.join(out_dir, "r%d.npz" % rank), users=model.user_id_embedding.detach().numpy(),
             items=model.item_id_embedding.detach().numpy(), losses=np.array(losses))
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_rank_sharded_training_matches_single_process_oracle(tmp_path):
    ref_u, ref_i, ref_losses = _oracle_run()
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "r0.npz"), np.load(tmp_path / "r1.npz")
    got_u = np.concatenate([r0["users"], r1["users"]])
    assert np.allclose(r0["items"], r1["items"], rtol=0, atol=0)              # replicas stay identical
    assert np.allclose(r0["losses"], ref_losses, rtol=1e-5)
    assert np.allclose(r1["losses"], ref_losses, rtol=1e-5)
    assert np.abs(got_u - ref_u).max() <= 1e-4 * np.abs(ref_u).max()
    assert np.abs(r0["items"] - ref_i).max() <= 1e-4 * np.abs(ref_i).max()


def _fused_worker(rank, world, port, out_dir, exchange, n_aug, n_users=U):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from llmrec_amd import dist as ld
    from llmrec_amd.dist_fused import ShardedFusedID
    from tests._cpu_backend import CpuBackend
    rows, cols, u_tab, i_tab, batches = _problem(world, n_users)
    comm, be = ld.Comm(), CpuBackend()
    u0, u1 = ld.user_block(n_users, rank, world)
    sel = (rows >= u0) & (rows < u1)
    g = ld.ShardedGraph.build(torch.tensor(rows[sel] - u0), torch.tensor(cols[sel]), u1 - u0, I, u0, comm, be)
    st = ShardedFusedID(g, comm, be, D, L, n_users, seed=1, lr=LR, batch_local=B_LOCAL + n_aug, drop_rate=DROP, decay=DECAY, n_chunks=3,
                        user_init=torch.tensor(u_tab[u0:u1]), item_init=torch.tensor(i_tab),
                        batch_size_flag=float(world * B_LOCAL), exchange=exchange)
    assert len(st.chunks) == 3
    if exchange == "rs_ag" and world == 2:                     # 45 items: chunks of 16 rows split over the two ranks, the last (13 rows) does not
        assert [s_ is not None for s_ in st.shards] == [True, True, False]
    losses = []
    for per_rank in _with_aug(batches, n_aug):
        us, ps, ns = per_rank[rank]
        loss, _ = st.step((torch.tensor(us - u0), torch.tensor(ps), torch.tensor(ns)))
        losses.append(float(loss))
        assert float(st.dE_u.abs().max()) == 0.0 and float(st.dE_i.abs().max()) == 0.0    # scatter targets are clean between steps
    np.savez(os.path.join(out_dir, "f%d.npz" % rank), users=st.user_tab.detach().numpy(), items=st.item_tab.detach().numpy(),
             losses=np.array(losses), msg=np.array([st.allreduce_bytes]))
    dist.destroy_process_group()


def _with_aug(batches, n_aug):
    """The reference's LLM-augmented triples (main.py:216-224): extra (user, pos, neg) triples for users of the batch, appended."""
    if not n_aug:
        return batches
    rng = np.random.default_rng(77)
    out = []
    for per_rank in batches:
        out.append([(np.concatenate([us, us[:n_aug]]), np.concatenate([ps, rng.integers(0, I, size=n_aug)]),
                     np.concatenate([ns, rng.integers(0, I, size=n_aug)])) for us, ps, ns in per_rank])
    return out


@pytest.mark.parametrize("exchange,n_aug", [("all_reduce", 0), ("rs_ag", 0), ("rs_ag", 3)])
def test_two_rank_fused_sharded_step_matches_single_process_oracle(tmp_path, exchange, n_aug):
    """llmrec_amd/dist_fused.py (hand-written backward, chunked exchanges - all-reduce or reduce-scatter + all-gather -, BPR
    gradient rows exchanged by all-gather + deterministic scatter) on two gloo ranks vs the single-process oracle after 3
    optimiser steps; with augmented triples the regulariser's divisor stays the batch-size flag (main.py:340)."""
    ref_u, ref_i, ref_losses = _oracle_run(n_aug)
    mp.spawn(_fused_worker, args=(2, _free_port(), str(tmp_path), exchange, n_aug), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "f0.npz"), np.load(tmp_path / "f1.npz")
    got_u = np.concatenate([r0["users"], r1["users"]])
    assert np.array_equal(r0["items"], r1["items"])                            # replicas stay bit-identical
    assert np.allclose(r0["losses"], ref_losses, rtol=1e-5) and np.allclose(r1["losses"], ref_losses, rtol=1e-5)
    assert np.abs(got_u - ref_u).max() <= 1e-4 * np.abs(ref_u).max()
    assert np.abs(r0["items"] - ref_i).max() <= 1e-4 * np.abs(ref_i).max()
    # one message per layer and direction: I x d, except the last layer's forward message = the FIXED-SIZE block of the two batches' item rows
    # (world x 2 x batch_local slots, zeros behind the list's end: its length never reaches the host - round 5)
    assert int(r0["msg"][0]) == 4 * I * D * (2 * L - 1) + 4 * (2 * 2 * (B_LOCAL + n_aug)) * D


@pytest.mark.parametrize("exchange", ["all_reduce", "rs_ag"])
def test_three_rank_fused_sharded_step_with_an_uneven_user_partition(tmp_path, exchange):
    """The same step on THREE gloo ranks with 61 users (blocks of 21, 21 and 19: the last rank's shard is shorter, its graph block and user table
    too; the BPR rows of three ranks are gathered, the prune threshold covers 3 x 12 samples): the sharded result equals the single-process
    oracle's after 3 optimiser steps, the item replicas stay bit-identical on all three ranks."""
    world, n_users = 3, 61
    ref_u, ref_i, ref_losses = _oracle_run(0, world, n_users)
    mp.spawn(_fused_worker, args=(world, _free_port(), str(tmp_path), exchange, 0, n_users), nprocs=world, join=True)
    rs = [np.load(tmp_path / ("f%d.npz" % r)) for r in range(world)]
    assert [r["users"].shape[0] for r in rs] == [21, 21, 19]
    got_u = np.concatenate([r["users"] for r in rs])
    assert np.array_equal(rs[0]["items"], rs[1]["items"]) and np.array_equal(rs[0]["items"], rs[2]["items"])
    for r in rs:
        assert np.allclose(r["losses"], ref_losses, rtol=1e-5)
    assert np.abs(got_u - ref_u).max() <= 1e-4 * np.abs(ref_u).max()
    assert np.abs(rs[0]["items"] - ref_i).max() <= 1e-4 * np.abs(ref_i).max()
    assert int(rs[0]["msg"][0]) == 4 * I * D * (2 * L - 1) + 4 * (world * 2 * B_LOCAL) * D


def test_user_block_partition_covers_all_users():
    from llmrec_amd.dist import user_block
    for n, w in ((10, 3), (8, 8), (5, 8), (1000003, 8)):
        blocks = [user_block(n, r, w) for r in range(w)]
        assert blocks[0][0] == 0 and blocks[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))


# ---------------------------------------------------------------------------------------------
# full multi-modal model, two ranks vs the single-process oracle
# ---------------------------------------------------------------------------------------------
MM_KEYS = ("year", "title", "director", "country", "language")


def _mm_problem(world=2, n_users=U):
    rng = np.random.default_rng(5)
    rows, cols, u_tab, i_tab, batches = _problem(world, n_users)
    feats = {"image": rng.standard_normal((I, 12)).astype(np.float32), "text": rng.standard_normal((I, 20)).astype(np.float32),
             "user": rng.standard_normal((n_users, 28)).astype(np.float32)}
    for k in MM_KEYS:
        feats["attr/" + k] = rng.standard_normal((I, 28)).astype(np.float32)
    lin = {}
    for name, fin in (("image_trans", 12), ("text_trans", 20), ("user_trans", 28), ("item_trans", 28)):
        lin[name + ".weight"] = (rng.standard_normal((D, fin)) * 0.2).astype(np.float32)
        lin[name + ".bias"] = (rng.standard_normal(D) * 0.1).astype(np.float32)
    return rows, cols, u_tab, i_tab, batches, feats, lin


def _mm_cfg(world=2):
    return O.Config(embed_size=D, n_layers=L, batch_size=world * B_LOCAL, decay=DECAY, prune_loss_drop_rate=DROP, lr=LR, keys=MM_KEYS)


def _mm_oracle_run(world=2, n_users=U):
    import scipy.sparse as sp
    rows, cols, u_tab, i_tab, batches, feats, lin = _mm_problem(world, n_users)
    R = sp.csr_matrix((np.ones(rows.size, dtype=np.float32), (rows, cols)), shape=(n_users, I))
    a_ui, a_iu = O.normalized_graphs(R)
    cfg = _mm_cfg(world)
    params = {k: torch.tensor(v, requires_grad=True) for k, v in lin.items()}
    params["user_id_embedding.weight"] = torch.tensor(u_tab, requires_grad=True)
    params["item_id_embedding.weight"] = torch.tensor(i_tab, requires_grad=True)
    opt = torch.optim.AdamW([{"params": list(params.values())}], lr=LR)
    ft = {k: torch.tensor(v) for k, v in feats.items()}
    losses = []
    for per_rank in batches:
        us = np.concatenate([b[0] for b in per_rank]); ps = np.concatenate([b[1] for b in per_rank]); ns = np.concatenate([b[2] for b in per_rank])
        fw = O.forward(params, ft, a_ui, a_iu, cfg)
        loss, _ = O.step_loss(fw, us, ps, ns, I, cfg)
        opt.zero_grad(); loss.backward(); opt.step()
        losses.append(float(loss))
    with torch.no_grad():
        fw = O.forward(params, ft, a_ui, a_iu, cfg)
    return {k: v.detach().numpy() for k, v in params.items()}, losses, fw["E_u"].numpy(), fw["E_i"].numpy(), rows, cols


def _mm_worker(rank, world, port, out_dir, n_users=U):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from llmrec_amd import dist as ld
    from llmrec_amd.engine import Hyper
    from tests._cpu_backend import CpuBackend
    rows, cols, u_tab, i_tab, batches, feats, lin = _mm_problem(world, n_users)
    comm, be = ld.Comm(), CpuBackend()
    u0, u1 = ld.user_block(n_users, rank, world)
    sel = (rows >= u0) & (rows < u1)
    g = ld.ShardedGraph.build(torch.tensor(rows[sel] - u0), torch.tensor(cols[sel]), u1 - u0, I, u0, comm, be)
    item_feats = {k: torch.tensor(v) for k, v in feats.items() if k != "user"}
    model = ld.ShardedMMModel(g, comm, be, D, L, n_users, item_feats, torch.tensor(feats["user"][u0:u1]), MM_KEYS, (0.02, 2.8, 0.005), seed=3)
    with torch.no_grad():
        for name, v in lin.items():
            mod, attr = name.split(".")
            getattr(getattr(model, mod), attr).copy_(torch.tensor(v))
        model.user_id_embedding.copy_(torch.tensor(u_tab[u0:u1])); model.item_id_embedding.copy_(torch.tensor(i_tab))
    hp = Hyper(batch_size=world * B_LOCAL, decay=DECAY, prune_loss_drop_rate=DROP); hp.lr = LR
    tr = ld.ShardedMMTrainer(model, hp, B_LOCAL, I)
    losses = []
    for per_rank in batches:
        us, ps, ns = per_rank[rank]
        loss, _ = tr.step(torch.tensor(us - u0), torch.tensor(ps), torch.tensor(ns))
        losses.append(float(loss))
    with torch.no_grad():
        fw = model()
    out = {"losses": np.array(losses), "E_u": fw["E_u"].numpy(), "E_i": fw["E_i"].numpy()}
    for name, p_ in model.named_parameters():
        out["p/" + name] = p_.detach().numpy()
    np.savez(os.path.join(out_dir, "mm%d.npz" % rank), **out)
    dist.destroy_process_group()


def test_two_rank_full_model_matches_single_process_oracle(tmp_path):
    ref_params, ref_losses, ref_eu, ref_ei, rows, cols = _mm_oracle_run()
    mp.spawn(_mm_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "mm0.npz"), np.load(tmp_path / "mm1.npz")
    rel = lambda a, b: np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)
    assert np.allclose(r0["losses"], ref_losses, rtol=2e-5) and np.allclose(r1["losses"], ref_losses, rtol=2e-5)
    for name in ("image_trans.weight", "image_trans.bias", "text_trans.weight", "user_trans.weight", "user_trans.bias",
                 "item_trans.weight", "item_trans.bias"):
        assert np.array_equal(r0["p/" + name], r1["p/" + name]), name           # replicas stay identical
        assert rel(r0["p/" + name], ref_params[name]) < 1e-4, name
    assert rel(r0["p/item_id_embedding"], ref_params["item_id_embedding.weight"]) < 1e-4
    got_u = np.concatenate([r0["p/user_id_embedding"], r1["p/user_id_embedding"]])
    assert rel(got_u, ref_params["user_id_embedding.weight"]) < 1e-4
    assert rel(np.concatenate([r0["E_u"], r1["E_u"]]), ref_eu) < 1e-4
    assert rel(r0["E_i"], ref_ei) < 1e-4


def test_three_rank_full_model_with_an_uneven_user_partition(tmp_path):
    """The full multi-modal model on THREE gloo ranks, 61 users (21 + 21 + 19): parameters, losses and fused embeddings against the
    single-process oracle; the replicated parameters stay bit-identical on all three ranks."""
    world, n_users = 3, 61
    ref_params, ref_losses, ref_eu, ref_ei, rows, cols = _mm_oracle_run(world, n_users)
    mp.spawn(_mm_worker, args=(world, _free_port(), str(tmp_path), n_users), nprocs=world, join=True)
    rs = [np.load(tmp_path / ("mm%d.npz" % r)) for r in range(world)]
    rel = lambda a, b: np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)
    for r in rs:
        assert np.allclose(r["losses"], ref_losses, rtol=2e-5)
    for name in ("image_trans.weight", "image_trans.bias", "text_trans.weight", "user_trans.weight", "user_trans.bias",
                 "item_trans.weight", "item_trans.bias"):
        assert np.array_equal(rs[0]["p/" + name], rs[1]["p/" + name]) and np.array_equal(rs[0]["p/" + name], rs[2]["p/" + name]), name
        assert rel(rs[0]["p/" + name], ref_params[name]) < 1e-4, name
    assert rel(rs[0]["p/item_id_embedding"], ref_params["item_id_embedding.weight"]) < 1e-4
    assert [r["p/user_id_embedding"].shape[0] for r in rs] == [21, 21, 19]
    assert rel(np.concatenate([r["p/user_id_embedding"] for r in rs]), ref_params["user_id_embedding.weight"]) < 1e-4
    assert rel(np.concatenate([r["E_u"] for r in rs]), ref_eu) < 1e-4
    assert rel(rs[0]["E_i"], ref_ei) < 1e-4


# ---------------------------------------------------------------------------------------------
# the exchange helpers of the batch-sharded replicas (llmrec_amd/dp.py uses exactly these two)
# ---------------------------------------------------------------------------------------------
def _comm_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from llmrec_amd import dist as ld
    comm = ld.Comm()
    assert comm.rank == rank and comm.world == world
    block = torch.arange(5, dtype=torch.float32) + 10 * rank            # this rank's gather block
    gathered = comm.all_gather_into(torch.zeros(world * 5), block)
    bucket = torch.full((7,), float(rank + 1))
    comm.all_reduce_(bucket)
    np.savez(os.path.join(out_dir, "c%d.npz" % rank), gathered=gathered.numpy(), bucket=bucket.numpy())
    dist.destroy_process_group()


def test_comm_exchange_helpers_two_ranks(tmp_path):
    mp.spawn(_comm_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    want = np.concatenate([np.arange(5), np.arange(5) + 10]).astype(np.float32)
    for r in range(2):
        z = np.load(tmp_path / ("c%d.npz" % r))
        assert np.array_equal(z["gathered"], want)                       # blocks in rank order on every rank
        assert np.array_equal(z["bucket"], np.full(7, 3.0, dtype=np.float32))


def test_device_batcher_slices_tile_the_global_batch_shape():
    """engine.DeviceBatcher's rank / world bookkeeping (no kernel call): capacity and slice offsets."""
    from llmrec_amd import engine
    exist = torch.arange(100)
    for world in (1, 2, 8):
        caps = []
        for rank in range(world):
            b = engine.DeviceBatcher(None, exist, 50, 16, torch.zeros(100, dtype=torch.int64), torch.zeros(100, dtype=torch.int64), 0.25,
                                     seed=3, rank=rank, world=world)
            assert b.n_aug == 4 and b.capacity == 20 and b.rank * b.B == rank * 16
            caps.append(b.capacity)
        assert len(set(caps)) == 1


def test_row_restricted_last_layer_equals_dense_step_across_a_stamp_wrap():
    """The row-restricted last layer (marks with a per-step byte stamp, llmrec_amd/dist_fused.py) against the same step with every
    product dense, single rank on the CPU stand-in (which poisons every row the contract says is not read): six steps that start at
    step 252, so the stamp passes 255 -> 1 (the marks are cleared there) and a stamp value is reused; losses and both tables agree."""
    from llmrec_amd import dist as ld
    from llmrec_amd.dist_fused import ShardedFusedID
    from tests._cpu_backend import CpuBackend
    rows, cols, u_tab, i_tab, batches = _problem()
    res = []
    for sparse in (True, False):
        comm, be = ld.Comm(single=True), CpuBackend()
        g = ld.ShardedGraph.build(torch.tensor(rows), torch.tensor(cols), U, I, 0, comm, be)
        st = ShardedFusedID(g, comm, be, D, L, U, seed=1, lr=LR, batch_local=2 * B_LOCAL, drop_rate=DROP, decay=DECAY, n_chunks=2,
                            user_init=torch.tensor(u_tab), item_init=torch.tensor(i_tab), batch_size_flag=float(2 * B_LOCAL),
                            sparse_backward=sparse)
        st.step_id = 251
        if sparse:                                             # marks left from the beginning of this stamp cycle (steps 1 and 2): their values
            st.flag_i[:] = 3; st.flag_u[:] = 2                 # come round again right after the wrap - they must not survive it
            st.tmpI.fill_(float("nan"))                         # (rows of the first gradient outside the marked items are never read)
        losses = []
        for k in range(6):
            per = batches[k % len(batches)]
            us = np.concatenate([per[0][0], per[1][0]]); ps = np.concatenate([per[0][1], per[1][1]]); ns = np.concatenate([per[0][2], per[1][2]])
            loss, _ = st.step((torch.tensor(us), torch.tensor(ps), torch.tensor(ns)))
            losses.append(float(loss))
        res.append((np.array(losses), st.user_tab.detach().numpy().copy(), st.item_tab.detach().numpy().copy()))
    (l0, u0_, i0_), (l1, u1_, i1_) = res
    assert np.all(np.isfinite(i0_)) and np.all(np.isfinite(u0_))
    assert np.allclose(l0, l1, rtol=1e-6)
    assert np.abs(u0_ - u1_).max() <= 1e-5 * np.abs(u1_).max() and np.abs(i0_ - i1_).max() <= 1e-5 * np.abs(i1_).max()
