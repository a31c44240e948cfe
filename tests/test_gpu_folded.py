"""GPU parity of the FOLDED launches of round 5 (36 -> 28 launches per step) against the launches they replace,
through the C ABI: same bits wherever the arithmetic is the same (loss values, saved slots, AdamW state, parameters, moments, stored
gradients, cleaned rows), 1e-6 where only a summation tree differs (the feature regulariser)."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"
p_ = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    from llmrec_amd import ops as _ops
    return _ops


def _stream():
    return torch.cuda.current_stream().cuda_stream


def test_fusion_launch_also_sums_the_regulariser(ops):
    """llmrec_fuse_fwd_multi_sumsq_f32: E_u / E_i bit-identical to llmrec_fuse_fwd_multi_f32; coef * sum(partials) = what the two
    llmrec_sumsq_f32 calls of the unfolded step leave in scal[0] (reference main.py:151-156), to summation-order rounding."""
    from llmrec_amd import _lib
    rng = np.random.default_rng(31)
    d, S = 64, 7
    rows = [1733, 2011]                                              # item side, user side
    cats = [torch.tensor(rng.standard_normal((r, S * d)).astype(np.float32)).to(DEV) for r in rows]
    profs = [torch.tensor(rng.standard_normal((r, d)).astype(np.float32)).to(DEV) for r in rows]
    means = [[torch.tensor(rng.standard_normal((r, d)).astype(np.float32)).to(DEV) for _ in range(3)] for r in rows]
    rates = (ctypes.c_float * (S + 1))(*([0.26, 0.26, 0.55] + [0.012] * (S - 2)))
    keep = []

    def problems(outs):
        arr = (ops.FuseFwdProblem * 2)()
        for k in range(2):
            norms = [cats[k][:, 0:d], cats[k][:, d:2 * d], profs[k]] + [cats[k][:, (2 + j) * d:(3 + j) * d] for j in range(S - 2)]
            mp = (ctypes.c_void_p * 3)(*[t.data_ptr() for t in means[k]]); ml = (ctypes.c_int64 * 3)(*[d] * 3)
            npt = (ctypes.c_void_p * len(norms))(*[t.data_ptr() for t in norms]); nl = (ctypes.c_int64 * len(norms))(*[t.stride(0) for t in norms])
            keep.extend((mp, ml, npt, nl))
            pr = arr[k]
            pr.rows, pr.mean_scale, pr.n_mean, pr.n_norm = rows[k], 1.0 / 3, 3, len(norms)
            pr.mean_terms, pr.mean_ld = ctypes.cast(mp, ctypes.c_void_p), ctypes.cast(ml, ctypes.c_void_p)
            pr.norm_terms, pr.norm_ld, pr.rates = ctypes.cast(npt, ctypes.c_void_p), ctypes.cast(nl, ctypes.c_void_p), ctypes.cast(rates, ctypes.c_void_p)
            pr.out, pr.ldo = outs[k].data_ptr(), d
        return arr
    plain = [torch.empty(r, d, device=DEV) for r in rows]
    folded = [torch.empty(r, d, device=DEV) for r in rows]
    _lib.call("llmrec_fuse_fwd_multi_f32", 2, problems(plain), d, _stream())
    cap = 4096
    partial = torch.full((cap,), float("nan"), device=DEV)
    n_part = ctypes.c_int32(0)
    _lib.call("llmrec_fuse_fwd_multi_sumsq_f32", 2, problems(folded), d, 2, p_(partial), cap, ctypes.byref(n_part), _stream())
    torch.cuda.synchronize()
    for a, b in zip(plain, folded):
        assert torch.equal(a, b)
    n = n_part.value
    assert 0 < n <= cap and bool(torch.isfinite(partial[:n]).all()) and bool(torch.isnan(partial[n:]).all())
    want = sum(float((c[:, :2 * d].double() ** 2).sum()) for c in cats)
    got = float(partial[:n].double().sum())
    assert abs(got - want) <= 2e-6 * want
    # the same call twice: same partials (fixed partition, fixed trees)
    partial2 = torch.zeros(cap, device=DEV)
    _lib.call("llmrec_fuse_fwd_multi_sumsq_f32", 2, problems(folded), d, 2, p_(partial2), cap, ctypes.byref(n_part), _stream())
    torch.cuda.synchronize()
    assert torch.equal(partial[:n], partial2[:n])
    # too small a partial buffer is an error, not an overrun
    with pytest.raises(RuntimeError):
        _lib.call("llmrec_fuse_fwd_multi_sumsq_f32", 2, problems(folded), d, 2, p_(partial2), 3, ctypes.byref(n_part), _stream())


@pytest.mark.parametrize("drop,cap,valid", [(0.71, 1126, 1126), (0.0, 1024, 1000), (0.5, 300, 123)])
def test_scores_step_and_losses_assemble_equal_the_separate_launches(ops, drop, cap, valid):
    """scores_step = scores + adamw_advance (state bits equal, `saved` equal); losses_assemble = losses + [sumsq x2] + loss_assemble mode 0:
    `out` and `saved` bit-identical, scal[1..3] and the double running sums equal given the same regulariser value."""
    from llmrec_amd import _lib
    rng = np.random.default_rng(7)
    U, I, P, d = 300, 420, 8, 64
    tabs = [(torch.tensor((rng.standard_normal((U, d)) * 0.3).astype(np.float32)).to(DEV), torch.tensor((rng.standard_normal((I, d)) * 0.3).astype(np.float32)).to(DEV))
            for _ in range(P)]
    idx = [torch.tensor(rng.integers(0, n, size=cap)).to(DEV) for n in (U, I, I)]
    nv = torch.tensor([valid], dtype=torch.int32, device=DEV)
    w_mf = [1.0, 0.02, 0.02] + [0.012] * (P - 3)
    wc = (ctypes.c_float * P)(*w_mf)
    partial = torch.tensor(rng.random(1911).astype(np.float32) * 50).to(DEV)
    coef = 1e-5 * 0.5 / I

    def run(folded):
        grads = [(torch.zeros(U, d, device=DEV), torch.zeros(I, d, device=DEV)) for _ in range(P)]
        arr = (ops.BprProblem * P)()
        for i in range(P):
            arr[i].Eu, arr[i].ldu, arr[i].Ei, arr[i].ldi = tabs[i][0].data_ptr(), d, tabs[i][1].data_ptr(), d
            arr[i].dEu, arr[i].lddu, arr[i].dEi, arr[i].lddi = grads[i][0].data_ptr(), d, grads[i][1].data_ptr(), d
            arr[i].g_mf, arr[i].g_emb = w_mf[i], 1.0 if i == 0 else 0.0
        saved = torch.full((P * ops.bpr_saved_floats(cap),), 7.0, device=DEV)
        out = torch.zeros(P, 2, device=DEV)
        stamp = torch.tensor([5], dtype=torch.int32, device=DEV)
        state = torch.zeros(3, device=DEV)
        scal = torch.zeros(4, device=DEV)
        running = torch.tensor([1.5, 2.5, 3.5], dtype=torch.float64, device=DEV)
        common = (P, arr, d, p_(idx[0]), p_(idx[1]), p_(idx[2]), cap, p_(nv))
        st = _stream()
        plan = ops.bpr_scatter_plan(idx[0], idx[1], idx[2], nv)
        for _ in range(3):                                         # three "steps": the counters advance three times
            if folded:
                _lib.call("llmrec_bpr_multi_scores_step_f32", *common, p_(saved), p_(stamp), p_(state), 1e-3, 0.9, 0.999, st)
            else:
                _lib.call("llmrec_adamw_advance", p_(state), 1e-3, 0.9, 0.999, st)
                _lib.call("llmrec_bpr_multi_scores_f32", *common, p_(saved), p_(stamp), st)
            _lib.call("llmrec_bpr_multi_select_bwd_f32", *common, 1 - drop, 1e-5, 64.0, p_(saved), None, None, p_(stamp), p_(plan), st)
            if folded:
                _lib.call("llmrec_bpr_multi_losses_assemble_f32", P, cap, p_(nv), 1 - drop, 1e-5, 64.0, p_(out), p_(saved), wc, p_(partial), partial.numel(),
                          coef, p_(scal), p_(running), st)
            else:
                _lib.call("llmrec_bpr_multi_losses_f32", P, cap, p_(nv), 1 - drop, 1e-5, 64.0, p_(out), p_(saved), st)
                scal[0] = float(scal_ref)                          # the regulariser's value, as the folded launch formed it
                _lib.call("llmrec_loss_assemble_f32", 0, P, p_(out), wc, p_(scal), None, 1.0, p_(running), st)
        torch.cuda.synchronize()
        return saved.cpu(), out.cpu(), state.cpu(), scal.cpu(), running.cpu(), int(stamp[0])
    s1, o1, a1, c1, r1, st1 = run(True)
    scal_ref = c1[0]
    s0, o0, a0, c0, r0, st0 = run(False)
    assert st0 == st1 == 8
    assert torch.equal(a0.view(torch.int32), a1.view(torch.int32)) and int(a1.view(torch.int32)[0]) == 3     # AdamW's state after three advances
    assert torch.equal(s0.view(torch.int32), s1.view(torch.int32))
    assert torch.equal(o0.view(torch.int32), o1.view(torch.int32))
    assert torch.equal(c0.view(torch.int32), c1.view(torch.int32))
    assert torch.equal(r0, r1)
    want = coef * float(partial.double().sum())
    assert abs(float(c1[0]) - want) <= 2e-6 * want
    assert abs(float(c1[1]) - (sum(float(o1[p, 0]) * w_mf[p] for p in range(P)) + float(o1[0, 1]) + float(c1[0]))) <= 1e-6 * abs(float(c1[1]))


def test_adamw_with_a_gradient_source_and_the_row_cleanup_in_one_launch(ops):
    """FusedAdamW.step_params(sources=..., zero_rows=...): (a) the update from `scale * G` stored into .grad by the same launch = axpy
    into .grad followed by the plain update, bit for bit (parameter, both moments, stored gradient); (b) the clean-up blocks zero exactly the
    listed rows of every job (b < n_valid), touch nothing else, and the parameter update beside them is the plain one."""
    from llmrec_amd import _lib
    rng = np.random.default_rng(3)
    U, I, d = 1300, 900, 64

    def table(n):
        p = torch.nn.Parameter(torch.tensor(rng.standard_normal((n, d)).astype(np.float32)).to(DEV))
        p.grad = torch.zeros_like(p)
        return p
    G = torch.tensor((rng.standard_normal((U, d)) * 1e-3).astype(np.float32)).to(DEV)
    inv = 1.0 / 3
    res = []
    for folded in (False, True):
        rng2 = np.random.default_rng(11)
        pu = torch.nn.Parameter(torch.tensor(rng2.standard_normal((U, d)).astype(np.float32)).to(DEV)); pu.grad = torch.zeros_like(pu)
        opt = ops.FusedAdamW([pu], lr=1e-3)
        for _ in range(3):
            opt.advance()
            if folded:
                opt.step_params([pu], sources={pu: (G, inv)})
            else:
                _lib.call("llmrec_axpy_f32", U, d, inv, None, p_(G), d, p_(pu.grad), d, 0, _stream())
                opt.step_params([pu])
        torch.cuda.synchronize()
        m, v = opt.moments(pu)
        res.append((pu.detach().clone(), pu.grad.clone(), m.clone(), v.clone()))
    for a, b in zip(*res):
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    # (b) clean-up beside an update
    cap, valid = 700, 613
    ids = [torch.tensor(rng.integers(0, n, size=cap)).to(DEV) for n in (U, I, I)]
    nv = torch.tensor([valid], dtype=torch.int32, device=DEV)
    wide = torch.tensor(rng.standard_normal((I, 7 * d)).astype(np.float32)).to(DEV)           # a [rows, 7 d] scatter target: whole rows
    narrow = torch.tensor(rng.standard_normal((U, d)).astype(np.float32)).to(DEV)
    sl = torch.tensor(rng.standard_normal((I, 3 * d)).astype(np.float32)).to(DEV)
    sl_view = sl[:, d:2 * d]                                                                   # a column slice: only these columns are cleaned
    before = (wide.clone(), narrow.clone(), sl.clone())
    outs = []
    for folded in (False, True):
        rng2 = np.random.default_rng(12)
        pi = torch.nn.Parameter(torch.tensor(rng2.standard_normal((I, d)).astype(np.float32)).to(DEV))
        pi.grad = torch.tensor((rng2.standard_normal((I, d)) * 1e-3).astype(np.float32)).to(DEV)
        opt = ops.FusedAdamW([pi], lr=1e-3)
        opt.advance()
        if folded:
            opt.step_params([pi], zero_rows=([(ids[1], wide), (ids[2], wide), (ids[0], narrow), (ids[1], sl_view)], cap, nv))
        else:
            opt.step_params([pi])
        torch.cuda.synchronize()
        outs.append((pi.detach().clone(),) + opt.moments(pi))
    for a, b in zip(*outs):
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    want_w, want_n, want_s = before[0].clone(), before[1].clone(), before[2].clone()
    want_w[ids[1][:valid]] = 0; want_w[ids[2][:valid]] = 0; want_n[ids[0][:valid]] = 0
    want_s[ids[1][:valid], d:2 * d] = 0
    assert torch.equal(wide, want_w) and torch.equal(narrow, want_n) and torch.equal(sl, want_s)
    # an empty batch cleans nothing
    nv.zero_()
    keep = wide.clone(); keep[5] = 1.0; wide.copy_(keep)
    opt.step_params([pi], zero_rows=([(ids[1], wide)], cap, nv))
    torch.cuda.synchronize()
    assert torch.equal(wide, keep)


def test_device_timestamps_are_monotonic_and_tick_at_the_advertised_rate(ops):
    """llmrec_timestamp (the bench's in-graph timing aid): stamps taken in stream order never decrease, a known kernel between two stamps
    takes about as long by the counter as by HIP events, and the step re-captured with stamps in it gives the same results as without."""
    from llmrec_amd import _lib
    rate = _lib.query("llmrec_timestamp_rate_hz")
    assert rate >= 1_000_000
    slots = torch.zeros(4, dtype=torch.int64, device=DEV)
    x = torch.randn(64 << 20, device=DEV)                               # 256 MB
    y = torch.empty_like(x)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(2):
        _lib.call("llmrec_timestamp", ctypes.c_void_p(slots.data_ptr()), _stream())
        e0.record()
        _lib.call("llmrec_axpy_f32", 1 << 20, 64, 2.0, None, p_(x), 64, p_(y), 64, 0, _stream())
        e1.record()
        _lib.call("llmrec_timestamp", ctypes.c_void_p(slots.data_ptr() + 8), _stream())
        _lib.call("llmrec_timestamp", ctypes.c_void_p(slots.data_ptr() + 16), _stream())
    torch.cuda.synchronize()
    t = slots.tolist()
    assert t[0] > 0 and t[0] <= t[1] <= t[2]
    by_counter_ms = (t[1] - t[0]) / rate * 1e3
    by_events_ms = e0.elapsed_time(e1)
    assert 0.5 * by_events_ms <= by_counter_ms <= 2.0 * by_events_ms + 0.05, (by_counter_ms, by_events_ms)
