"""GPU tests of the row-sharded fused ID step (llmrec_amd/dist_fused.py) on the HIP backend:
* one rank (collectives are identities) against the single-process oracle, with the item rows cut into chunks and
  hub rows that cross every SpMM row bucket;
* two PROCESSES on one GPU (torch.distributed gloo moving device tensors through the host) against the same oracle -
  the code `bench.py --gpus N --workload synth` runs over RCCL;
* the two kernels the step adds (compact BPR gradient rows, deterministic sorted row scatter) against torch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

from oracle import oracle as O

U, I, L, B_LOCAL, STEPS = 900, 700, 2, 128, 3
DROP, DECAY, LR = 0.71, 1e-5, 1e-3
# (d, augmented triples per rank): cfg-4-like (d = 64) and cfg-5-like (BASELINE.json configs[4]: d = 128, 5 % pseudo-augmented
# triples appended to the batch - reference main.py:216-224 -, prune loss on)
CASES = [(64, 0), (128, 6)]


def _problem(world, D, n_aug):
    rng = np.random.default_rng(4)
    deg = rng.integers(1, 30, size=U); deg[7] = 600; deg[U - 3] = 200                     # hubs in both halves
    rows = np.repeat(np.arange(U), deg)
    cols = np.concatenate([rng.choice(I, size=c, replace=False) for c in deg])
    hot = rng.integers(0, U, size=2500); rows = np.concatenate([rows, hot]); cols = np.concatenate([cols, np.full(hot.size, 5)])   # a hub item
    key = np.unique(rows * I + cols); rows, cols = key // I, key % I
    u_tab = (rng.standard_normal((U, D)) * 0.1).astype(np.float32)
    i_tab = (rng.standard_normal((I, D)) * 0.1).astype(np.float32)
    per = (U + world - 1) // world
    batches = []
    for s in range(STEPS):
        per_rank = []
        for r in range(world):
            u0, u1 = r * per, min((r + 1) * per, U)
            pos = rng.integers(0, I, size=B_LOCAL); pos[:40] = 5                           # duplicate gradient rows (>= 3 per id)
            us, ns = rng.integers(u0, u1, size=B_LOCAL), rng.integers(0, I, size=B_LOCAL)
            if n_aug:                                                                      # extra triples for users of the batch
                us, pos, ns = np.concatenate([us, us[:n_aug]]), np.concatenate([pos, rng.integers(0, I, size=n_aug)]), np.concatenate([ns, rng.integers(0, I, size=n_aug)])
            per_rank.append((us, pos, ns))
        batches.append(per_rank)
    return rows, cols, u_tab, i_tab, batches


def _oracle_run(world, D, n_aug):
    import scipy.sparse as sp
    rows, cols, u_tab, i_tab, batches = _problem(world, D, n_aug)
    R = sp.csr_matrix((np.ones(rows.size, dtype=np.float32), (rows, cols)), shape=(U, I))
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
        opt.zero_grad(); (mf + emb).backward(); opt.step()
        losses.append(float(mf + emb))
    return pu.detach().numpy(), pi.detach().numpy(), losses


def _run_rank(rank, world, n_chunks, D, n_aug, exchange="all_reduce"):
    from llmrec_amd import dist as ld
    from llmrec_amd.dist_fused import ShardedFusedID
    rows, cols, u_tab, i_tab, batches = _problem(world, D, n_aug)
    comm, be = ld.Comm(), ld.HipBackend()
    u0, u1 = ld.user_block(U, rank, world)
    sel = (rows >= u0) & (rows < u1)
    g = ld.ShardedGraph.build(torch.tensor(rows[sel] - u0).cuda(), torch.tensor(cols[sel]).cuda(), u1 - u0, I, u0, comm, be)
    st = ShardedFusedID(g, comm, be, D, L, U, seed=1, lr=LR, batch_local=B_LOCAL + n_aug, drop_rate=DROP, decay=DECAY, n_chunks=n_chunks,
                        user_init=torch.tensor(u_tab[u0:u1]), item_init=torch.tensor(i_tab),
                        batch_size_flag=float(world * B_LOCAL), exchange=exchange)       # the divisor is the FLAG (main.py:340), not B + aug
    losses = []
    for per_rank in batches:
        us, ps, ns = per_rank[rank]
        loss, _ = st.step((torch.tensor(us - u0).cuda(), torch.tensor(ps).cuda(), torch.tensor(ns).cuda()))
        losses.append(float(loss))
        assert float(st.dE_u.abs().max()) == 0.0 and float(st.dE_i.abs().max()) == 0.0    # row-wise clean-up left the scatter targets zero
    return st, losses


@pytest.mark.parametrize("D,n_aug", CASES)
@pytest.mark.parametrize("n_chunks", [1, 5])
def test_fused_sharded_step_single_rank_matches_oracle(n_chunks, D, n_aug):
    ref_u, ref_i, ref_losses = _oracle_run(1, D, n_aug)
    st, losses = _run_rank(0, 1, n_chunks, D, n_aug)
    assert len(st.chunks) == n_chunks
    assert np.allclose(losses, ref_losses, rtol=2e-5), (losses, ref_losses)
    got_u, got_i = st.user_tab.detach().cpu().numpy(), st.item_tab.detach().cpu().numpy()
    assert np.abs(got_u - ref_u).max() <= 1e-4 * np.abs(ref_u).max()
    assert np.abs(got_i - ref_i).max() <= 1e-4 * np.abs(ref_i).max()
    if n_aug == 0:
        loss, _ = st.step()                                        # the device sampler path
        assert np.isfinite(float(loss))


def _worker(rank, world, port, out_dir, D, n_aug, exchange):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    st, losses = _run_rank(rank, world, 3, D, n_aug, exchange)
    np.savez(os.path.join(out_dir, "g%d.npz" % rank), users=st.user_tab.detach().cpu().numpy(), items=st.item_tab.detach().cpu().numpy(),
             losses=np.array(losses))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("D,n_aug,exchange,world", [(64, 0, "all_reduce", 2), (128, 6, "all_reduce", 2), (128, 6, "rs_ag", 2),
                                                    (64, 0, "rs_ag", 4), (128, 6, "all_reduce", 3)])
def test_fused_sharded_step_two_processes_one_gpu_match_oracle(tmp_path, D, n_aug, exchange, world):
    """world = 3 / 4: three / four user blocks of 300 / 225 users, chunk shards cut three / four ways (a world that does not divide the users:
    tests/test_dist_cpu.py, 61 users on three gloo ranks)."""
    ref_u, ref_i, ref_losses = _oracle_run(world, D, n_aug)
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), D, n_aug, exchange), nprocs=world, join=True)
    r = [np.load(tmp_path / ("g%d.npz" % k)) for k in range(world)]
    for j in range(1, world):
        assert np.array_equal(r[0]["items"], r[j]["items"])                               # replicas bit-identical (deterministic scatter)
    for j in range(world):
        assert np.allclose(r[j]["losses"], ref_losses, rtol=2e-5)
    got_u = np.concatenate([r[j]["users"] for j in range(world)])
    assert np.abs(got_u - ref_u).max() <= 1e-4 * np.abs(ref_u).max()
    assert np.abs(r[0]["items"] - ref_i).max() <= 1e-4 * np.abs(ref_i).max()


def _rccl_worker(rank, port, out_dir, D, n_aug, exchange, sparse_forward):
    """ONE rank over the real RCCL communicator with LLMREC_FORCE_COLLECTIVES=1: every collective of the step is issued (all-reduce /
    reduce_scatter_tensor / all_gather_into_tensor with async_op=True on the communicator's stream, the BPR rows' all-gather, the
    4-float exchange) - the non-gloo branches of llmrec_amd/dist_fused.py execute under pytest instead of only on an 8-GPU node."""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LLMREC_FORCE_COLLECTIVES="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from llmrec_amd import dist as ld
    assert ld.Comm().force and not ld.Comm()._host_staged()
    import llmrec_amd.dist_fused as df
    calls = {"rs": 0, "ag": 0, "ar": 0}
    orig = (dist.reduce_scatter_tensor, dist.all_gather_into_tensor, dist.all_reduce)
    def count(name, fn):
        def wrapped(*a, **k):
            calls[name] += 1
            return fn(*a, **k)
        return wrapped
    dist.reduce_scatter_tensor, dist.all_gather_into_tensor, dist.all_reduce = count("rs", orig[0]), count("ag", orig[1]), count("ar", orig[2])
    st, losses = _run_rank(0, 1, 5, D, n_aug, exchange) if not sparse_forward else _run_rank_variant(0, 1, 5, D, n_aug, exchange, sparse_forward=True)
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, "rccl.npz"), users=st.user_tab.detach().cpu().numpy(), items=st.item_tab.detach().cpu().numpy(),
             losses=np.array(losses), calls=np.array([calls["rs"], calls["ag"], calls["ar"]]))
    dist.destroy_process_group()


def _run_rank_variant(rank, world, n_chunks, D, n_aug, exchange, **kw):
    from llmrec_amd import dist as ld
    from llmrec_amd.dist_fused import ShardedFusedID
    rows, cols, u_tab, i_tab, batches = _problem(world, D, n_aug)
    comm, be = ld.Comm(), ld.HipBackend()
    u0, u1 = ld.user_block(U, rank, world)
    sel = (rows >= u0) & (rows < u1)
    g = ld.ShardedGraph.build(torch.tensor(rows[sel] - u0).cuda(), torch.tensor(cols[sel]).cuda(), u1 - u0, I, u0, comm, be)
    st = ShardedFusedID(g, comm, be, D, L, U, seed=1, lr=LR, batch_local=B_LOCAL + n_aug, drop_rate=DROP, decay=DECAY, n_chunks=n_chunks,
                        user_init=torch.tensor(u_tab[u0:u1]), item_init=torch.tensor(i_tab), batch_size_flag=float(world * B_LOCAL), exchange=exchange, **kw)
    losses = []
    for per_rank in batches:
        us, ps, ns = per_rank[rank]
        loss, _ = st.step((torch.tensor(us - u0).cuda(), torch.tensor(ps).cuda(), torch.tensor(ns).cuda()))
        losses.append(float(loss))
    return st, losses


@pytest.mark.parametrize("exchange,sparse_forward", [("all_reduce", False), ("rs_ag", False), ("rs_ag", True)])
def test_fused_sharded_step_one_rank_over_rccl_with_forced_collectives(tmp_path, exchange, sparse_forward):
    D, n_aug = 128, 6
    ref_u, ref_i, ref_losses = _oracle_run(1, D, n_aug)
    mp.spawn(_rccl_worker, args=(_free_port(), str(tmp_path), D, n_aug, exchange, sparse_forward), nprocs=1, join=True)
    r = np.load(tmp_path / "rccl.npz")
    assert np.allclose(r["losses"], ref_losses, rtol=2e-5), (r["losses"], ref_losses)
    assert np.abs(r["users"] - ref_u).max() <= 1e-4 * np.abs(ref_u).max()
    assert np.abs(r["items"] - ref_i).max() <= 1e-4 * np.abs(ref_i).max()
    rs, ag, ar = (int(x) for x in r["calls"])
    if exchange == "rs_ag":
        assert rs > 0 and ag > rs                            # the chunk exchanges (+ the ids / rows / log-sigmoid all-gathers)
    else:
        assert rs == 0 and ar > 0 and ag > 0


def test_sharded_step_with_the_dense_forward_never_synchronises_the_host():
    """no .item() / host read-back inside ShardedFusedID.step on the headline (dense-forward) path - asserted with
    torch's sync debug mode after the warm-up steps have built the row plans."""
    st, _ = _run_rank_variant(0, 1, 3, 64, 0, "all_reduce", sparse_forward=False)
    rows, cols, u_tab, i_tab, batches = _problem(1, 64, 0)
    us, ps, ns = batches[0][0]
    triples = (torch.tensor(us).cuda(), torch.tensor(ps).cuda(), torch.tensor(ns).cuda())
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        loss, parts = st.step(triples)
        loss2, _ = st.step()                                       # the device sampler path
    finally:
        torch.cuda.set_sync_debug_mode("default")
    assert np.isfinite(float(loss)) and np.isfinite(float(loss2))


def test_sharded_step_with_the_restricted_forward_never_synchronises_the_host_either():
    """the row-restricted-forward VARIANT builds its row list on the device (llmrec_sort_unique_ids_i32), forms the listed
    rows of the last product as a fixed-size compact block with a device-side count (llmrec_spmm_rows_compact_f32) and writes them back
    (llmrec_scatter_set_rows_f32): no torch.unique, no boolean indexing, no plan read-back - a step under torch's sync debug mode."""
    st, _ = _run_rank_variant(0, 1, 3, 64, 0, "all_reduce", sparse_forward=True)
    assert st.sparse_forward
    rows, cols, u_tab, i_tab, batches = _problem(1, 64, 0)
    us, ps, ns = batches[0][0]
    triples = (torch.tensor(us).cuda(), torch.tensor(ps).cuda(), torch.tensor(ns).cuda())
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        loss, parts = st.step(triples)
        loss2, _ = st.step()
    finally:
        torch.cuda.set_sync_debug_mode("default")
    assert np.isfinite(float(loss)) and np.isfinite(float(loss2))


@pytest.mark.parametrize("d", [64, 128, 20])
def test_listed_rows_of_a_product_with_the_list_on_the_device(d):
    """llmrec_sort_unique_ids_i32 + llmrec_spmm_rows_compact_f32 + llmrec_scatter_set_rows_f32 against torch: distinct ids ascending (negatives
    skipped, duplicates once), the listed rows of diag(s) P X incl. empty rows, rows around the 2 048-edge piece size and hubs cut into 8
    pieces, zeros behind the list's end, deterministic; the scatter writes exactly the listed rows."""
    from llmrec_amd import ops
    rng = np.random.default_rng(40 + d)
    n_rows, n_cols = 400, 30000
    degs = rng.integers(0, 30, size=n_rows)
    for k, dg in enumerate([0, 1, 2047, 2048, 2049, 4096, 9000, 16384, 20000, 29000]):
        degs[(k * 17 + 3) % n_rows] = dg
    r = np.repeat(np.arange(n_rows), degs)
    c = np.concatenate([rng.choice(n_cols, size=int(x), replace=False) for x in degs]).astype(np.int64)
    rp, ci, _ = ops.csr_from_coo(torch.tensor(r).cuda(), torch.tensor(c).cuda(), None, n_rows, n_cols)
    s = torch.tensor(rng.random(n_rows).astype(np.float32) + 0.5).cuda()
    a = ops.Csr(n_rows, n_cols, rp, ci, None, s, None, {})
    X = torch.tensor(rng.standard_normal((n_cols, d)).astype(np.float32)).cuda()
    ids = rng.integers(0, n_rows, size=300); ids[::7] = -1; ids[:12] = [(k * 17 + 3) % n_rows for k in range(10)] + [5, 5]
    ids_t = torch.tensor(ids).cuda()
    cap = ids.size
    lst, n = torch.full((cap,), -7, dtype=torch.int32, device="cuda"), torch.zeros(1, dtype=torch.int32, device="cuda")
    ops.sort_unique_ids(ids_t, lst, n)
    want_ids = np.unique(ids[ids >= 0])
    assert int(n[0]) == want_ids.size and np.array_equal(lst[:want_ids.size].cpu().numpy(), want_ids) and int(lst[want_ids.size:].abs().max()) == 0
    out = torch.full((cap, d), float("nan"), device="cuda")
    ops.spmm_rows_compact(a, X, lst, n, out)
    out2 = torch.full((cap, d), float("nan"), device="cuda")
    ops.spmm_rows_compact(a, X, lst, n, out2)
    assert torch.equal(out, out2)                                               # fixed piece order: deterministic
    A = torch.sparse_csr_tensor(rp.long(), ci.long(), torch.ones(ci.numel(), device="cuda", dtype=torch.float64), (n_rows, n_cols))
    ref = (A @ X.double()) * s.double()[:, None]
    got = out[:want_ids.size].double()
    assert float((got - ref[want_ids]).abs().max() / ref.abs().max()) < 2e-6
    assert float(out[want_ids.size:].abs().max()) == 0.0                        # zeros behind the list's end
    dst = torch.full((n_rows, d), 3.0, device="cuda")
    ops.scatter_set_rows(lst, n, out, dst)
    want = torch.full((n_rows, d), 3.0, device="cuda"); want[want_ids] = out[:want_ids.size]
    assert torch.equal(dst, want)
    # an empty list: nothing computed, everything zero
    n.zero_()
    ops.spmm_rows_compact(a, X, lst, n, out)
    assert float(out.abs().max()) == 0.0
    # sort_unique at its limit and beyond
    big = torch.randint(0, 5_000_000, (32768,), device="cuda")
    l2, n2 = torch.zeros(32768, dtype=torch.int32, device="cuda"), torch.zeros(1, dtype=torch.int32, device="cuda")
    ops.sort_unique_ids(big, l2, n2)
    u = torch.unique(big)
    assert int(n2[0]) == u.numel() and torch.equal(l2[:u.numel()].long(), u)
    with pytest.raises(RuntimeError):
        ops.sort_unique_ids(torch.zeros(32769, dtype=torch.int64, device="cuda"), torch.zeros(32769, dtype=torch.int32, device="cuda"), n2)


def test_zero_rows_clears_exactly_the_listed_rows():
    from llmrec_amd import dist as ld
    be = ld.HipBackend()
    big = torch.ones(500, 3 * 128, device="cuda")
    ids = torch.tensor([3, 499, -1, 3, 77], device="cuda")
    be.zero_rows(ids, big[:, 128:256])                                                    # a column slice (ld = 384)
    want = torch.ones(500, 3 * 128); want[[3, 499, 77], 128:256] = 0.0
    assert torch.equal(big.cpu(), want)


def test_scatter_rows_is_deterministic_and_matches_index_add():
    from llmrec_amd import dist as ld
    be = ld.HipBackend()
    rng = np.random.default_rng(9)
    n, rows_dst, d = 5000, 300, 64
    ids = rng.integers(0, rows_dst, size=n); ids[:700] = 17; ids[rng.integers(0, n, size=200)] = -1      # a long run + skipped entries
    rows = rng.standard_normal((n, d)).astype(np.float32)
    base = rng.standard_normal((rows_dst, d)).astype(np.float32)
    want = torch.tensor(base).double()
    keep = ids >= 0
    want.index_add_(0, torch.tensor(ids[keep]), 0.5 * torch.tensor(rows[keep]).double())
    outs = []
    for _ in range(3):
        dst = torch.tensor(base).cuda()
        be.scatter_rows(torch.tensor(ids).cuda(), torch.tensor(rows).cuda(), dst, 0.5)
        outs.append(dst.cpu())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])
    assert float((outs[0].double() - want).abs().max() / want.abs().max()) < 2e-6
    # strided source / destination (column slices)
    big_src = torch.zeros(n, 3 * d).cuda(); big_src[:, d:2 * d] = torch.tensor(rows).cuda()
    big_dst = torch.zeros(rows_dst, 2 * d).cuda(); big_dst[:, :d] = torch.tensor(base).cuda()
    be.scatter_rows(torch.tensor(ids).cuda(), big_src[:, d:2 * d], big_dst[:, :d], 0.5)
    assert torch.equal(big_dst[:, :d].cpu(), outs[0]) and float(big_dst[:, d:].abs().max()) == 0.0


@pytest.mark.parametrize("n", [1, 2, 3, 1000, 16383, 16384, 16385, 40000])
def test_scatter_rows_sorts_in_one_block_up_to_16384_rows_and_through_the_radix_sort_beyond(n):
    """llmrec_scatter_rows_f32: n <= 16 384 (the per-step case: 2 B x world gradient rows) orders its (id, j) keys with a one-block bitonic
    network in LDS - no vendor library on the row-sharded step's path -, larger n with rocPRIM's radix sort; the keys are
    distinct, so both give THE order: the same bits as a sequential accumulation in ascending j, run to run."""
    from llmrec_amd import dist as ld
    be = ld.HipBackend()
    rng = np.random.default_rng(n)
    rows_dst, d = 257, 64
    ids = rng.integers(0, rows_dst, size=n)
    if n > 10:
        ids[: n // 7] = 5; ids[rng.integers(0, n, size=n // 20 + 1)] = -1
    rows = rng.standard_normal((n, d)).astype(np.float32)
    base = rng.standard_normal((rows_dst, d)).astype(np.float32)
    outs = []
    for _ in range(2):
        dst = torch.tensor(base).cuda()
        be.scatter_rows(torch.tensor(ids).cuda(), torch.tensor(rows).cuda(), dst, 0.5)
        outs.append(dst.cpu())
    assert torch.equal(outs[0], outs[1])
    ref = torch.tensor(base).double()
    keep = ids >= 0
    ref.index_add_(0, torch.tensor(ids[keep]), 0.5 * torch.tensor(rows[keep]).double())
    assert float((outs[0].double() - ref).abs().max() / ref.abs().max()) < 3e-6


def test_bpr_gradient_rows_match_the_scatter_form():
    """llmrec_bpr_prune_bwd_rows_f32 = the same per-sample gradients llmrec_bpr_prune_bwd_f32 scatters."""
    from llmrec_amd import dist as ld
    be = ld.HipBackend()
    rng = np.random.default_rng(3)
    Un, In, d, B = 50, 40, 64, 96
    Eu = torch.tensor(rng.standard_normal((Un, d)).astype(np.float32)).cuda(); Ei = torch.tensor(rng.standard_normal((In, d)).astype(np.float32)).cuda()
    u = torch.tensor(rng.integers(0, Un, size=B)).cuda(); p = torch.tensor(rng.integers(0, In, size=B)).cuda(); n = torch.tensor(rng.integers(0, In, size=B)).cuda()
    _, s1 = be.bpr_fwd(Eu, Ei, u, p, n, 0.29, 1e-5, float(B), None, 0, 0, True)
    gm = be.bpr_local_m(s1, B).contiguous().clone()
    out, saved = be.bpr_fwd(Eu, Ei, u, p, n, 0.29, 1e-5, float(B), gm, B, 0, False)
    g2 = torch.tensor([1.0, 2.0]).cuda()
    dEu, dEi = be.bpr_bwd(Eu, Ei, u, p, n, 1e-5, float(B), saved, g2)
    rows3 = torch.empty(3, B, d).cuda()
    be.bpr_bwd_rows(Eu, Ei, u, p, n, 1e-5, float(B), saved, g2, rows3)
    wu = torch.zeros(Un, d).cuda(); wi = torch.zeros(In, d).cuda()
    be.scatter_rows(u, rows3[0], wu, 1.0)
    be.scatter_rows(torch.cat([p, n]), rows3[1:3].reshape(-1, d), wi, 1.0)
    assert float((wu - dEu).abs().max() / dEu.abs().max()) < 2e-6
    assert float((wi - dEi).abs().max() / dEi.abs().max()) < 2e-6
