"""TEST-ONLY torch-CPU stand-in for llmrec_amd.dist.HipBackend, so the sharding logic and the
placement of the collectives in llmrec_amd/dist.py can run under gloo without a GPU. It follows
the C ABI's contracts (saved-buffer layout of llmrec_bpr_prune_fwd_sharded_f32 etc.)."""
import torch
import torch.nn.functional as F


class _Pattern:
    def __init__(self, rows, cols, n_rows, n_cols, row_scale=None, col_scale=None):
        self.rows, self.cols, self.n_rows, self.n_cols = rows, cols, n_rows, n_cols
        self.row_scale, self.col_scale = row_scale, col_scale


class CpuBackend:
    def pattern_csr(self, rows, cols, n_rows, n_cols):
        return _Pattern(rows.long(), cols.long(), n_rows, n_cols)

    def with_scales(self, p, row_scale, col_scale):
        return _Pattern(p.rows, p.cols, p.n_rows, p.n_cols, row_scale, col_scale)

    def degrees(self, p):
        return torch.bincount(p.rows, minlength=p.n_rows).float()

    def spmm(self, p, X, out=None, epilogue=None):
        if epilogue is not None and epilogue.get("x_row_mask") is not None:      # the contract of llmrec_spmm_epilogue_t's operand sparsity
            act = epilogue["x_row_mask"] == epilogue["x_mask_active"]
            X = torch.where(act[:, None], X, torch.zeros_like(X))                # rows that are not active are never read
            if epilogue.get("y_row_gate") is not None:                           # gated-out rows are written as zeros unread: check the promise
                out_of_gate = epilogue["y_row_gate"] != epilogue["x_mask_active"]
                touched = torch.zeros(p.n_rows, dtype=torch.bool); touched[p.rows[act[p.cols]]] = True
                assert not bool((touched & out_of_gate).any()), "y_row_gate excludes a row with an active neighbour"
            if epilogue.get("y_row_flag") is not None:
                hit = torch.zeros(p.n_rows, dtype=torch.bool)
                hit[p.rows[act[p.cols]]] = True
                if epilogue.get("z_row_flag") is not None:
                    hit |= epilogue["z_row_flag"] == epilogue["x_mask_active"]
                epilogue["y_row_flag"].copy_(torch.where(hit, torch.full_like(epilogue["y_row_flag"], epilogue["x_mask_active"]),
                                                         torch.zeros_like(epilogue["y_row_flag"])))
        X = X if p.col_scale is None else X * p.col_scale[:, None]
        A = torch.sparse_coo_tensor(torch.stack([p.rows, p.cols]), torch.ones(p.rows.numel()), (p.n_rows, p.n_cols))
        Y = torch.sparse.mm(A, X)
        Y = Y if p.row_scale is None else Y * p.row_scale[:, None]
        if epilogue is not None:                                  # the contract of llmrec_spmm_epilogue_t
            if epilogue.get("Z") is not None:
                Y = epilogue.get("alpha", 0.0) * epilogue["Z"] + Y
            op = epilogue.get("op", "none")
            if op == "softmax":
                Y = torch.softmax(Y, dim=-1)
            elif op == "softmax_bwd":
                S = epilogue["S"]
                Y = S * (Y - (Y * S).sum(-1, keepdim=True))
            if epilogue.get("post_scale") is not None:
                Y = Y * epilogue["post_scale"][:, None]
        if epilogue is not None and epilogue.get("x_row_mask") is not None and epilogue.get("y_row_gate") is not None:
            Y = torch.where((epilogue["y_row_gate"] == epilogue["x_mask_active"])[:, None], Y, torch.zeros_like(Y))   # gated-out rows: zeros, unread
        if epilogue is not None and epilogue.get("y_row_needed") is not None:       # rows that are not needed keep their (stale) contents
            need = epilogue["y_row_needed"] == epilogue["x_mask_active"]
            out[need] = Y[need]
            out[~need] = float("nan")                                        # poison: nothing may read them
            return out
        if out is not None:
            out.copy_(Y)
            return out
        return Y

    def spmm_listed(self, p, X, rows, out):
        Y = self.spmm(p, X)
        out[:] = float("nan")                                                # poison everything that is not listed
        out[rows] = Y[rows]
        return out

    def sort_unique_ids(self, ids, out_list, out_n):
        u = torch.unique(ids[ids >= 0]).to(torch.int32)
        out_list.zero_(); out_list[:u.numel()] = u; out_n[0] = u.numel()

    def spmm_rows_compact(self, p, X, row_list, n_list, out):
        n = int(n_list[0])
        Y = self.spmm(p, X)
        out.zero_()                                                         # slots past the list's end are zeros (a fixed-size message)
        out[:n] = Y[row_list[:n].long()]

    def scatter_set_rows(self, row_list, n_list, src, dst):
        n = int(n_list[0])
        dst[row_list[:n].long()] = src[:n]

    def mark_rows(self, ids, value, flags):
        flags[ids[ids >= 0]] = value

    def softmax_bwd_listed_into(self, ids, alpha, Y, dY, post_scale, out):
        r = torch.unique(ids[ids >= 0])
        g = alpha * dY[r]
        z = Y[r] * (g - (g * Y[r]).sum(-1, keepdim=True))
        out[r] = z if post_scale is None else z * post_scale[r][:, None]

    def mark_neighbours(self, ids, p, value, flags):
        sel = torch.isin(p.rows, ids[ids >= 0])
        flags[p.cols[sel]] = value

    def row_chunk(self, p, r0, r1):
        sel = (p.rows >= r0) & (p.rows < r1)
        return _Pattern(p.rows[sel] - r0, p.cols[sel], r1 - r0, p.n_cols,
                        None if p.row_scale is None else p.row_scale[r0:r1], p.col_scale)

    def softmax_rows_into(self, Z, out):
        out.copy_(torch.softmax(Z, dim=-1))

    def softmax_bwd_into(self, Y, dY, out):
        out.copy_(Y * (dY - (dY * Y).sum(-1, keepdim=True)))

    def axpy_into(self, alpha, X, out):
        out.copy_(alpha * X)

    def scale_rows_into(self, s, X, out):
        out.copy_(X * s[:, None])

    def layer_mean_into(self, terms, out):
        out.copy_(torch.mean(torch.stack(list(terms)), dim=0))

    def gather_mean_into(self, terms, idx, out):
        out.copy_(torch.mean(torch.stack([t[idx] for t in terms]), dim=0))

    def optimizer_step(self, opt, grad_scales):
        with torch.no_grad():
            for p, s in grad_scales.items():
                p.grad.mul_(s)
        opt.step()

    def zero_(self, tensors):
        for t in tensors:
            t.zero_()

    def zero_rows(self, ids, dst):
        dst[ids[ids >= 0]] = 0.0

    def bpr_bwd_rows(self, Eu, Ei, u, p, n, decay, bsz, saved, grads2, rows3):
        B = u.numel()
        ds = grads2[0] * saved[:B]
        Su, Sp, Sq = saved[B], saved[B + 1], saved[B + 2]
        base = -4.0 * decay / bsz * grads2[1]
        cu, cp, cq = (base / (2 * S + 1e-8) ** 2 for S in (Su, Sp, Sq))
        eu, ep, en = Eu[u], Ei[p], Ei[n]
        rows3[0].copy_(ds[:, None] * (ep - en) + cu * eu)
        rows3[1].copy_(ds[:, None] * eu + cp * ep)
        rows3[2].copy_(-ds[:, None] * eu + cq * en)

    def scatter_rows(self, ids, rows, dst, alpha):
        keep = ids >= 0
        dst.index_add_(0, ids[keep], alpha * rows[keep])

    def sample(self, seed, step, exist, n_items, by_user, B):
        raise NotImplementedError("the CPU stand-in is driven with explicit triples")

    def softmax_rows(self, Z):
        return torch.softmax(Z, dim=-1)

    def layer_mean(self, terms):
        return torch.mean(torch.stack(list(terms)), dim=0)

    def linear(self, X, W, b):
        return F.linear(X, W, b)

    def fuse(self, mean_terms, norm_terms, rates):
        out = torch.mean(torch.stack(list(mean_terms)), dim=0)
        for r, t in zip(rates, norm_terms):
            out = out + r * F.normalize(t, p=2, dim=1)
        return out

    def sumsq(self, coef, Xs):
        return sum(coef * (x ** 2).sum() for x in Xs)

    def bpr_fwd(self, Eu, Ei, u, p, n, remember, decay, bsz, global_m, global_B, offset, scores_only):
        B = u.numel()
        eu, ep, en = Eu[u], Ei[p], Ei[n]
        x = (eu * ep).sum(1) - (eu * en).sum(1) + 1e-8
        m = F.logsigmoid(x)
        sg = torch.sigmoid(-x)
        norms = torch.stack([(eu ** 2).sum(), (ep ** 2).sum(), (en ** 2).sum()])
        saved = torch.zeros(B + 4)
        if scores_only:
            saved[:B] = m
            saved[B:B + 3] = norms
            return torch.zeros(2), saved
        k = int(remember * global_B)
        gidx = torch.arange(global_B)
        me = offset + torch.arange(B)
        rank = ((global_m[None, :] < m[:, None]) | ((global_m[None, :] == m[:, None]) & (gidx[None, :] < me[:, None]))).sum(1)
        keep = rank < k
        saved[:B] = torch.where(keep, -sg / k, torch.zeros(B))
        saved[B:B + 3] = norms
        saved[B + 3] = k
        out = torch.stack([-(m[keep].sum() / k), torch.tensor(0.0)])
        return out, saved

    def bpr_local_m(self, saved, B):
        return saved[:B]

    def bpr_bwd(self, Eu, Ei, u, p, n, decay, bsz, saved, grads2):
        B = u.numel()
        ds = grads2[0] * saved[:B]
        Su, Sp, Sq = saved[B], saved[B + 1], saved[B + 2]
        base = -4.0 * decay / bsz * grads2[1]
        cu, cp, cq = (base / (2 * S + 1e-8) ** 2 for S in (Su, Sp, Sq))
        dEu, dEi = torch.zeros_like(Eu), torch.zeros_like(Ei)
        eu, ep, en = Eu[u], Ei[p], Ei[n]
        dEu.index_add_(0, u, ds[:, None] * (ep - en) + cu * eu)
        dEi.index_add_(0, p, ds[:, None] * eu + cp * ep)
        dEi.index_add_(0, n, -ds[:, None] * eu + cq * en)
        return dEu, dEi

    def optimizer(self, params, lr):
        return _CpuAdamW(params, lr)

    def optimizer_advance(self, opt):
        opt.t += 1

    def optimizer_step_params(self, opt, params, grad_scales):
        with torch.no_grad():
            for p in params:
                opt.update(p, slice(None), p.grad * grad_scales.get(p, 1.0))

    def optimizer_step_rows(self, opt, param, row0, row1, grad_rows):
        with torch.no_grad():
            opt.update(param, slice(row0, row1), grad_rows)


class _CpuAdamW:
    """torch.optim.AdamW(lr, betas (.9, .999), eps 1e-8, weight_decay 0.01) written out, with an update over a row range."""

    def __init__(self, params, lr):
        self.params, self.lr, self.t = list(params), lr, 0
        self.m = {p: torch.zeros_like(p) for p in self.params}
        self.v = {p: torch.zeros_like(p) for p in self.params}

    def update(self, p, rows, g):
        b1, b2, eps, wd = 0.9, 0.999, 1e-8, 0.01
        m, v = self.m[p][rows], self.v[p][rows]
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1, bc2 = 1 - b1 ** self.t, 1 - b2 ** self.t
        p.data[rows] = p.data[rows] * (1 - self.lr * wd) - (self.lr / bc1) * m / (v.sqrt() / bc2 ** 0.5 + eps)

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            p.grad = None

    def step(self):
        self.t += 1
        with torch.no_grad():
            for p in self.params:
                if p.grad is not None:
                    self.update(p, slice(None), p.grad)
