"""Host side of the C ABI: device-CSR operands and torch.autograd Functions over the HIP kernels.

Everything here is plumbing: tensors stay torch-owned, kernels receive ``tensor.data_ptr()`` and
the hipStream_t of torch's current stream (so launches are ordered with torch's own work and can
be captured in a HIP graph). No function in this module computes on the CPU or falls back to
torch math: if the HIP library is missing or a tensor is not on the GPU, it raises.
"""
from __future__ import annotations

import ctypes
import os
import weakref
from dataclasses import dataclass
from typing import List, Optional, Sequence

import torch

from . import _lib

CONST = _lib.CONST
_c = ctypes


def _stream():
    return _c.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t: Optional[torch.Tensor]):
    return None if t is None else _c.c_void_p(t.data_ptr())


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("llmrec_amd: the HIP path needs GPU tensors; got a %s tensor "
                               "(there is no CPU fallback)" % t.device)


def _rowmajor(t: torch.Tensor) -> torch.Tensor:
    """2-D fp32 view whose inner stride is 1 (copy only if it is not)."""
    if t.dtype != torch.float32:
        raise RuntimeError("llmrec_amd: fp32 expected, got %s" % t.dtype)
    if t.dim() != 2:
        raise RuntimeError("llmrec_amd: 2-D tensor expected")
    if t.stride(1) != 1 or (t.shape[0] > 1 and t.stride(0) < t.shape[1]):
        t = t.contiguous()
    return t


def _ld(t: torch.Tensor) -> int:
    return t.stride(0) if t.shape[0] > 1 else max(t.shape[1], t.stride(0))


# ---------------------------------------------------------------------------------------------
# R1: CSR operands
# ---------------------------------------------------------------------------------------------
class SpmmPlanC(_c.Structure):
    """llmrec_spmm_plan_t"""
    _fields_ = [("t_wave", _c.c_int32), ("t_block", _c.c_int32), ("segment", _c.c_int32),
                ("n_wave_rows", _c.c_int32), ("wave_rows", _c.c_void_p), ("n_block_rows", _c.c_int32), ("block_rows", _c.c_void_p),
                ("n_split_rows", _c.c_int32), ("split_rows", _c.c_void_p), ("split_seg_begin", _c.c_void_p),
                ("n_segments", _c.c_int32), ("seg_split", _c.c_void_p), ("n_short_rows", _c.c_int32), ("slot_row", _c.c_void_p)]


class SpmmEpilogueC(_c.Structure):
    """llmrec_spmm_epilogue_t"""
    _fields_ = [("op", _c.c_int32), ("alpha", _c.c_float), ("Z", _c.c_void_p), ("ldz", _c.c_int64), ("S", _c.c_void_p), ("lds", _c.c_int64),
                ("post_scale", _c.c_void_p), ("x_row_mask", _c.c_void_p), ("x_mask_active", _c.c_int32), ("y_row_flag", _c.c_void_p),
                ("z_row_flag", _c.c_void_p), ("y_row_gate", _c.c_void_p), ("y_row_needed", _c.c_void_p), ("rows_listed_only", _c.c_int32),
                ("x_nt_from_row", _c.c_int32), ("xcd_contiguous", _c.c_int32), ("no_pipeline", _c.c_int32)]


EPI_NONE, EPI_SOFTMAX, EPI_SOFTMAX_BWD = 0, 1, 2
SPMM_LATENCY_NNZ = 4_000_000      # below: the graph is L2-resident and a step is launch-bound (Netflix scale)


def spmm_shape(d: int, nnz: int, whole_row: bool = False):
    """(slice_width, (t_wave, t_block, segment)) for llmrec_spmm_f32: the latency / throughput policy.
    Launch-bound graphs (nnz < SPMM_LATENCY_NNZ): the longest dependent gather chain of any wave sets a kernel's
    duration, so a lane group gets at most 32 nnz (wavefront bucket) / 64 nnz (block bucket), and wide operands whose
    width is a multiple of 64 are cut into 64-column slices ((row, slice) tasks) unless an epilogue needs the whole row.
    HBM-bound graphs: 128 / 512 nnz per lane group (fewer, longer pieces; partial sums only for hubs)."""
    small = nnz < SPMM_LATENCY_NNZ
    sw = 64 if (small and not whole_row and d > 64 and d % 64 == 0) else 0
    w = sw or d
    lpr = 4
    while lpr < 64 and lpr * 4 < w:
        lpr *= 2
    groups = 64 // lpr
    if small:
        t_wave, t_block = max(32, 32 * groups), 8 * groups * 64
    else:
        t_wave, t_block = 128 * groups, 8 * groups * 512
    return sw, (t_wave, t_block, t_block)


@dataclass
class SpmmPlan:
    """Row buckets of one rowptr for one threshold triple (include/llmrec_hip.h, llmrec_spmm_plan_*)."""
    t_wave: int = 128
    t_block: int = 2048
    segment: int = 2048
    n_wave: int = 0
    n_block: int = 0
    n_split: int = 0
    n_seg: int = 0
    wave_rows: Optional[torch.Tensor] = None
    block_rows: Optional[torch.Tensor] = None
    split_rows: Optional[torch.Tensor] = None
    split_seg_begin: Optional[torch.Tensor] = None
    seg_split: Optional[torch.Tensor] = None
    # the permuted CSR (llmrec_spmm_plan_t.slot_row): rows stored in visiting order, slot -> output row; None = the operand's own CSR
    n_short: int = 0
    slot_row: Optional[torch.Tensor] = None
    p_rowptr: Optional[torch.Tensor] = None
    p_colidx: Optional[torch.Tensor] = None
    colidx_src: int = 0               # data_ptr of the colidx the permuted copy was made from

    def csr_of(self, a: "Csr"):
        """(rowptr, colidx) llmrec_spmm_f32 gets with this plan: the permuted copy when the plan has one (made from THIS operand's pattern)."""
        if self.slot_row is None:
            return a.rowptr, a.colidx
        if a.val is not None or a.colidx.data_ptr() != self.colidx_src:
            raise RuntimeError("SpmmPlan: the permuted CSR of this plan belongs to another pattern (plans are shared by operands over ONE pattern)")
        return self.p_rowptr, self.p_colidx

    @property
    def n_long(self) -> int:
        return self.n_wave + self.n_block + self.n_split

    def c_struct(self) -> SpmmPlanC:
        c = getattr(self, "_c", None)
        if c is None:
            dp = lambda t, n: t.data_ptr() if n else None
            c = self._c = SpmmPlanC(self.t_wave, self.t_block, self.segment, self.n_wave, dp(self.wave_rows, self.n_wave),
                                    self.n_block, dp(self.block_rows, self.n_block), self.n_split, dp(self.split_rows, self.n_split),
                                    dp(self.split_seg_begin, self.n_split), self.n_seg, dp(self.seg_split, self.n_split),
                                    self.n_short if self.slot_row is not None else 0, dp(self.slot_row, self.slot_row is not None))
        return c

    def scratch(self, d: int, device) -> Optional[torch.Tensor]:
        """The plan's own partial-sum scratch for width d, one per (d, current stream): plans are shared by the operands over
        one rowptr (ui.fwd / iu.bwd, iu.fwd / ui.bwd), and two SpMMs over such operands issued on different streams must
        not share the partial sums of their split rows."""
        if not self.n_seg:
            return None
        cache = self.__dict__.setdefault("_scratch", {})
        key = (d, torch.cuda.current_stream().cuda_stream)
        if key not in cache:
            cache[key] = torch.empty(self.n_seg * d, dtype=torch.float32, device=device)
        return cache[key]

    @staticmethod
    def by_length_class(rows: torch.Tensor, deg: torch.Tensor) -> torch.Tensor:
        """`rows` (ascending ids) reordered by descending length class - the power-of-two bucket of the row's nnz, empty rows last - and
        ascending id within a class (stable): rows that share a wavefront / a round of blocks then take (almost) equally long."""
        if os.environ.get("LLMREC_SPMM_CLASS", "log2") == "exact":       # (experiment: one class per length)
            cls = deg.to(torch.int64)
        else:
            cls = torch.where(deg > 0, torch.floor(torch.log2(deg.clamp(min=1).to(torch.float64))).to(torch.int64) + 1, torch.zeros_like(deg, dtype=torch.int64))
        return rows[torch.sort(-cls, stable=True).indices].to(torch.int32).contiguous()

    @staticmethod
    def build(rowptr: torch.Tensor, t_wave: int = 128, t_block: int = 2048, segment: int = 2048, colidx: Optional[torch.Tensor] = None,
              order_rows: Optional[bool] = None) -> "SpmmPlan":
        """colidx (pattern-only operands) + order_rows: the plan carries a PERMUTED copy of the CSR - every bucket's rows stored in the order
        they are visited, by descending length class (by_length_class). order_rows None = on (LLMREC_SPMM_ORDER=0: the operand's own CSR in
        row order). Measured (profiles/experiments/r06_spmm_order.md): 2 M x 1 M x 40 M edges, d = 64: 1.65 -> 1.39 ms (rows = users),
        1.14 -> 0.99 ms (rows = items); cfg 4 whole 57.2 -> 53.5 ms per step; the Netflix-shaped step 0.460 -> 0.447 ms. Results never
        depend on the order (bit-identical: tests/test_gpu_ops.py)."""
        n_rows = rowptr.numel() - 1
        dev = rowptr.device
        if order_rows is None:
            order_rows = os.environ.get("LLMREC_SPMM_ORDER", "1") != "0"
        order_rows = bool(order_rows) and colidx is not None and n_rows > 0
        scratch = torch.zeros(4, dtype=torch.int32, device=dev)
        counts = (_c.c_int32 * 4)()
        _lib.call("llmrec_spmm_plan_count", n_rows, _p(rowptr), t_wave, t_block, segment, _p(scratch), counts, _stream())
        nw, nb, nsp, nseg = (int(x) for x in counts)
        wr = br = sr = sb = ss = None
        if nw + nb + nsp:
            i32 = lambda n: torch.empty(max(n, 1), dtype=torch.int32, device=dev)
            wr, br, sr, sb, ss = i32(nw), i32(nb), i32(nsp), i32(nsp), i32(nseg)
            _lib.call("llmrec_spmm_plan_fill", n_rows, _p(rowptr), t_wave, t_block, segment, _p(scratch), _p(wr), _p(br), _p(sr), _p(sb), _p(ss), _stream())
            # the fill compacts with atomics: sort the two independent row lists so that the plan (and the order rows are
            # visited in) is reproducible run to run; results never depend on the order
            wr = torch.sort(wr[:nw]).values.contiguous() if nw else None
            br = torch.sort(br[:nb]).values.contiguous() if nb else None
            sr, sb, ss = (sr, sb, ss) if nsp else (None, None, None)
        if not order_rows:
            torch.cuda.current_stream().synchronize()                  # (see below: the lists are read by every stream)
            return SpmmPlan(t_wave, t_block, segment, nw, nb, nsp, nseg, wr, br, sr, sb, ss)
        deg = (rowptr[1:] - rowptr[:-1]).to(torch.int64)
        ids = torch.nonzero(deg <= CONST["LLMREC_SPMM_LONG_ROW"]).flatten()
        parts = [SpmmPlan.by_length_class(ids, deg[ids])]
        if nw:
            parts.append(SpmmPlan.by_length_class(wr.long(), deg[wr.long()]))
        if nb:
            parts.append(SpmmPlan.by_length_class(br.long(), deg[br.long()]))
        if nsp:
            parts.append(sr[:nsp])
        n_short = parts[0].numel()
        slot_row = torch.cat(parts).contiguous()                       # slot -> row
        assert slot_row.numel() == n_rows
        p_rowptr = torch.zeros(n_rows + 1, dtype=torch.int64, device=dev)
        torch.cumsum(deg[slot_row.long()], 0, out=p_rowptr[1:])
        p_rowptr = p_rowptr.to(torch.int32)
        p_colidx = torch.empty_like(colidx)
        _lib.call("llmrec_csr_permute_rows", n_rows, _p(rowptr), _p(colidx), _p(slot_row), _p(p_rowptr), _p(p_colidx), _stream())
        ar = lambda lo, n: torch.arange(lo, lo + n, dtype=torch.int32, device=dev) if n else None     # the lists hold slots
        lists = (ar(n_short, nw), ar(n_short + nw, nb), ar(n_short + nw + nb, nsp))
        # A plan is built lazily, on whatever stream runs the first product over its operand (a SIDE stream of the fused step's warm-up), and
        # is then shared by every stream: its arrays must be complete before any other stream reads them. (Round 6: four replica processes
        # on one GPU faulted in the first captured step - the chain's stream read a permuted CSR the profile stream was still writing.)
        torch.cuda.current_stream().synchronize()
        return SpmmPlan(t_wave, t_block, segment, nw, nb, nsp, nseg, lists[0], lists[1], lists[2], sb, ss,
                        n_short, slot_row, p_rowptr, p_colidx, colidx.data_ptr())


@dataclass
class Csr:
    """One SpMM operand: Y = diag(row_scale) (P . val) diag(col_scale) X."""
    n_rows: int
    n_cols: int
    rowptr: torch.Tensor            # int32 [n_rows + 1]
    colidx: torch.Tensor            # int32 [nnz]
    val: Optional[torch.Tensor]     # fp32 [nnz] or None (pattern only)
    row_scale: Optional[torch.Tensor]
    col_scale: Optional[torch.Tensor]
    plans: Optional[dict] = None    # {(t_wave, t_block, segment): SpmmPlan}, shared by the operands that share this rowptr

    def __post_init__(self):
        if not isinstance(self.plans, dict):
            self.plans = {}

    @property
    def nnz(self) -> int:
        return self.colidx.numel()

    def plan_for(self, d: int, whole_row: bool = False):
        """(slice_width, SpmmPlan) for an operand of width d."""
        sw, key = spmm_shape(d, self.nnz, whole_row)
        pl = self.plans.get(key)
        if pl is None:
            if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
                # building a plan synchronises (llmrec_spmm_plan_count returns counts to the host) - illegal under capture
                raise RuntimeError("Csr.plan_for: no row plan for (d = %d, whole_row = %s) yet and the stream is being captured; "
                                   "run the same product once eagerly (a warm-up step) before capturing" % (d, whole_row))
            pl = self.plans[key] = SpmmPlan.build(self.rowptr, *key, colidx=self.colidx if self.val is None else None)
        return sw, pl

    @property
    def plan(self) -> SpmmPlan:
        return self.plan_for(64)[1]


def csr_from_coo(rows: torch.Tensor, cols: torch.Tensor, vals: Optional[torch.Tensor], n_rows: int, n_cols: int):
    """Device COO (int64) -> (rowptr, colidx, val) with ascending columns per row."""
    _need_gpu(rows, cols, vals)
    nnz = rows.numel()
    dev = rows.device
    rows = rows.to(torch.int64).contiguous()
    cols = cols.to(torch.int64).contiguous()
    if vals is not None:
        vals = vals.to(torch.float32).contiguous()
    rowptr = torch.empty(n_rows + 1, dtype=torch.int32, device=dev)
    colidx = torch.empty(nnz, dtype=torch.int32, device=dev)
    val = torch.empty(nnz, dtype=torch.float32, device=dev) if vals is not None else None
    ws_bytes = _lib.query("llmrec_csr_build_workspace_bytes", n_rows, nnz)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    _lib.call("llmrec_csr_build", n_rows, n_cols, nnz, _p(rows), _p(cols), _p(vals), _p(rowptr), _p(colidx), _p(val),
              _p(ws), ws_bytes, _stream())
    return rowptr, colidx, val


def degree_scale(rowptr: torch.Tensor) -> torch.Tensor:
    out = torch.empty(rowptr.numel() - 1, dtype=torch.float32, device=rowptr.device)
    _lib.call("llmrec_degree_scale", out.numel(), _p(rowptr), _p(out), _stream())
    return out


@dataclass
class SparseOperand:
    """A sparse matrix A prepared for Y = A X (``fwd``) and dX = A^T dY (``bwd``)."""
    fwd: Csr
    bwd: Csr

    @property
    def shape(self):
        return (self.fwd.n_rows, self.fwd.n_cols)

    @staticmethod
    def from_coo(rows, cols, vals, n_rows: int, n_cols: int) -> "SparseOperand":
        """General path (any values). If every row's values are one constant - the reference's
        normalised adjacency diag(s) R, main.py:123-126 - the 4 B/nnz value stream is dropped."""
        rowptr, colidx, val = csr_from_coo(rows, cols, vals, n_rows, n_cols)
        t_rowptr, t_colidx, t_val = csr_from_coo(cols, rows, vals, n_cols, n_rows)
        row_const = None
        if val is not None:
            rc = torch.empty(n_rows, dtype=torch.float32, device=rowptr.device)
            flag = torch.zeros(1, dtype=torch.int32, device=rowptr.device)
            _lib.call("llmrec_csr_row_constant", n_rows, _p(rowptr), _p(val), _p(rc), _p(flag), _stream())
            if int(flag.item()) == 1:
                row_const = rc
        plan_f, plan_b = {}, {}                                  # built lazily per piece length, shared by the operands below
        if val is None or row_const is not None:
            fwd = Csr(n_rows, n_cols, rowptr, colidx, None, row_const, None, plan_f)
            bwd = Csr(n_cols, n_rows, t_rowptr, t_colidx, None, None, row_const, plan_b)
        else:
            fwd = Csr(n_rows, n_cols, rowptr, colidx, val, None, None, plan_f)
            bwd = Csr(n_cols, n_rows, t_rowptr, t_colidx, t_val, None, None, plan_b)
        return SparseOperand(fwd, bwd)


_operand_cache = {}


def operand_from_sparse_tensor(t: torch.Tensor) -> SparseOperand:
    """Drop-in entry: accepts the torch sparse COO tensor the reference passes to
    MM_Model.forward (reference main.py:128-134) and caches the device CSR on tensor identity."""
    key = id(t)
    hit = _operand_cache.get(key)
    if hit is not None and hit[0]() is t:
        return hit[1]
    if not t.is_sparse:
        raise RuntimeError("llmrec_amd: expected a torch sparse COO tensor")
    _need_gpu(t)
    idx = t._indices()
    op = SparseOperand.from_coo(idx[0], idx[1], t._values(), t.shape[0], t.shape[1])
    _operand_cache[key] = (weakref.ref(t), op)
    return op


@dataclass
class BipartiteGraph:
    """User-item interaction pattern R with the reference's one-sided normalisation:
    A_ui = diag(s_u) R, A_iu = diag(s_i) R^T (reference main.py:84-91). Holds two pattern-only
    CSRs (by user, by item) shared by the four SpMM directions."""
    n_users: int
    n_items: int
    ui: SparseOperand
    iu: SparseOperand
    by_user: Csr            # pattern of R (train items per user, ascending) - also the eval mask
    s_u: torch.Tensor
    s_i: torch.Tensor

    @staticmethod
    def from_edges(users: torch.Tensor, items: torch.Tensor, n_users: int, n_items: int) -> "BipartiteGraph":
        ru, cu, _ = csr_from_coo(users, items, None, n_users, n_items)
        ri, ci, _ = csr_from_coo(items, users, None, n_items, n_users)
        s_u, s_i = degree_scale(ru), degree_scale(ri)
        pu, pi = {}, {}                                          # plan caches shared by the operands over ru / ri
        ui = SparseOperand(Csr(n_users, n_items, ru, cu, None, s_u, None, pu), Csr(n_items, n_users, ri, ci, None, None, s_u, pi))
        iu = SparseOperand(Csr(n_items, n_users, ri, ci, None, s_i, None, pi), Csr(n_users, n_items, ru, cu, None, None, s_i, pu))
        return BipartiteGraph(n_users, n_items, ui, iu, ui.fwd, s_u, s_i)


# ---------------------------------------------------------------------------------------------
# R2: SpMM
# ---------------------------------------------------------------------------------------------
def spmm_epilogue(op: int = EPI_NONE, alpha: float = 0.0, Z: Optional[torch.Tensor] = None, S: Optional[torch.Tensor] = None,
                  post_scale: Optional[torch.Tensor] = None, x_row_mask: Optional[torch.Tensor] = None, x_mask_active: int = 0,
                  y_row_flag: Optional[torch.Tensor] = None, z_row_flag: Optional[torch.Tensor] = None,
                  y_row_gate: Optional[torch.Tensor] = None, y_row_needed: Optional[torch.Tensor] = None, rows_listed_only: bool = False,
                  x_nt_from_row: int = 0, xcd_contiguous: bool = False, no_pipeline: Optional[bool] = None):
    """llmrec_spmm_epilogue_t: Y = post_scale . op(alpha * Z + A X); S = forward softmax rows for EPI_SOFTMAX_BWD.
    x_row_mask (uint8 [n_cols]) / x_mask_active: X rows whose byte differs from the active value are promised all-zero and not read;
    y_row_flag (uint8 [n_rows]): receives the active value for rows whose result can be non-zero (z_row_flag: the non-zero rows of Z);
    y_row_gate (uint8 [n_rows]): rows without the active value are promised zero results and written as zeros unread;
    y_row_needed (uint8 [n_rows]): rows without the active value are neither computed nor written; rows_listed_only: compute the rows
    in the plan's lists only (spmm_listed); x_nt_from_row > 0: X rows from that index on are gathered with non-temporal loads (cache hint);
    xcd_contiguous: the workgroups of one XCD take a contiguous piece of the rows (same bits, another block -> row map)."""
    for t in (x_row_mask, y_row_flag, z_row_flag, y_row_gate, y_row_needed):
        if t is not None and (t.dtype != torch.uint8 or not t.is_contiguous()):
            raise RuntimeError("spmm_epilogue: row masks / flags are contiguous uint8 tensors")
    return SpmmEpilogueC(op, float(alpha), Z.data_ptr() if Z is not None else None, _ld(Z) if Z is not None else 0,
                         S.data_ptr() if S is not None else None, _ld(S) if S is not None else 0,
                         post_scale.data_ptr() if post_scale is not None else None,
                         x_row_mask.data_ptr() if x_row_mask is not None else None, int(x_mask_active),
                         y_row_flag.data_ptr() if y_row_flag is not None else None,
                         z_row_flag.data_ptr() if z_row_flag is not None else None,
                         y_row_gate.data_ptr() if y_row_gate is not None else None,
                         y_row_needed.data_ptr() if y_row_needed is not None else None, 1 if rows_listed_only else 0, int(x_nt_from_row),
                         1 if xcd_contiguous else 0,
                         1 if (no_pipeline if no_pipeline is not None else os.environ.get("LLMREC_SPMM_PIPELINE", "1") == "0") else 0)


def listed_plan(a: Csr, rows: torch.Tensor, d: int, whole_row: bool = False) -> SpmmPlan:
    """A row plan over an explicit list of DISTINCT rows (int64 device tensor): every listed row lands in the wavefront, block or split
    list by its length - with rows_listed_only the launch computes exactly these rows. Built with torch ops on the device; reading the
    three list lengths back synchronises (once per call: the row-sharded training step, whose batch decides the list)."""
    _, (t_wave, t_block, segment) = spmm_shape(d, a.nnz, whole_row)
    rp = a.rowptr
    rows = rows.to(torch.int64)
    deg = (rp[rows + 1] - rp[rows]).to(torch.int64)
    i32 = lambda t: t.to(torch.int32).contiguous()
    w = i32(rows[deg <= t_wave]); b = i32(rows[(deg > t_wave) & (deg <= t_block)])
    big = deg > t_block
    sp = i32(rows[big])
    nseg = (deg[big] + segment - 1) // segment
    seg_begin = i32(torch.cumsum(nseg, 0) - nseg)
    seg_split = i32(torch.repeat_interleave(torch.arange(sp.numel(), device=rows.device), nseg))
    return SpmmPlan(t_wave, t_block, segment, w.numel(), b.numel(), sp.numel(), seg_split.numel(), w if w.numel() else None,
                    b if b.numel() else None, sp if sp.numel() else None, seg_begin if sp.numel() else None, seg_split if sp.numel() else None)


def spmm_listed(a: Csr, X: torch.Tensor, rows: torch.Tensor, out: torch.Tensor, epilogue=None) -> torch.Tensor:
    """out[r] = epilogue(A X)[r] for the listed DISTINCT rows r only (all other rows of out keep their contents): llmrec_spmm_f32 with a
    plan over the list and rows_listed_only."""
    d = X.shape[1]
    whole = epilogue is not None and epilogue.op != EPI_NONE
    pl = listed_plan(a, rows, d, whole_row=whole)
    if epilogue is None:
        epilogue = spmm_epilogue(EPI_NONE)
    epilogue.rows_listed_only = 1
    part = torch.empty(max(pl.n_seg, 1) * d, dtype=torch.float32, device=X.device)
    return spmm_raw(a, X, out=out, epilogue=epilogue, partials=part, plan=pl)


def sort_unique_ids(ids: torch.Tensor, out_list: torch.Tensor, out_n: torch.Tensor):
    """out_list[0 .. out_n[0]) = the distinct ids >= 0 of `ids`, ascending (int32), entirely on the device (llmrec_sort_unique_ids_i32:
    one block, <= 32 768 ids): the host never learns the count."""
    _need_gpu(ids, out_list, out_n)
    n = ids.numel()
    if out_list.dtype != torch.int32 or out_n.dtype != torch.int32 or out_list.numel() < n or ids.dtype != torch.int64 or not ids.is_contiguous():
        raise RuntimeError("sort_unique_ids: ids int64 contiguous, out_list int32 with room for every id, out_n int32[1]")
    _lib.call("llmrec_sort_unique_ids_i32", n, _p(ids), _p(out_list), _p(out_n), _stream())


def spmm_rows_compact(a: Csr, X: torch.Tensor, row_list: torch.Tensor, n_list: torch.Tensor, out: torch.Tensor, ws: Optional[torch.Tensor] = None):
    """out[j] = (A X)[row_list[j]] for j < n_list[0] (device count), zeros for the other slots of out [capacity, d]: "these rows of A X" as a
    fixed-size compact block, no host read-back (llmrec_spmm_rows_compact_f32; pattern-only operands)."""
    _need_gpu(X, out, row_list, n_list)
    if a.val is not None or a.col_scale is not None:
        raise RuntimeError("spmm_rows_compact: pattern-only operands (A = diag(row_scale) P)")
    X = _rowmajor(X)
    cap, d = out.shape
    need = _lib.query("llmrec_spmm_rows_compact_workspace_bytes", cap, d)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=X.device)
    _lib.call("llmrec_spmm_rows_compact_f32", a.n_rows, a.n_cols, _p(a.rowptr), _p(a.colidx), _p(a.row_scale), _p(X), _ld(X), d, _p(row_list),
              _p(n_list), cap, _p(out), _ld(out), _p(ws), ws.numel(), _stream())
    return ws


def scatter_set_rows(row_list: torch.Tensor, n_list: torch.Tensor, src: torch.Tensor, dst: torch.Tensor):
    """dst[row_list[j]] = src[j] for j < n_list[0] (llmrec_scatter_set_rows_f32)."""
    _need_gpu(src, dst, row_list, n_list)
    _lib.call("llmrec_scatter_set_rows_f32", src.shape[0], _p(row_list), _p(n_list), src.shape[1], _p(src), _ld(src), _p(dst), _ld(dst), _stream())


def spmm_raw(a: Csr, X: torch.Tensor, out: Optional[torch.Tensor] = None, accumulate: bool = False, epilogue=None,
             partials: Optional[torch.Tensor] = None, plan: Optional[SpmmPlan] = None) -> torch.Tensor:
    """Y = epilogue(A X) through llmrec_spmm_f32. accumulate: Y += A X (epilogue Z = Y, alpha = 1). partials: scratch of
    plan.n_seg * d floats when the caller runs several SpMMs over this operand at once (default: the plan's own)."""
    _need_gpu(X, a.rowptr)
    X = _rowmajor(X)
    if X.shape[0] != a.n_cols:
        raise RuntimeError("spmm: X has %d rows, operand has %d columns" % (X.shape[0], a.n_cols))
    d = X.shape[1]
    Y = out if out is not None else torch.empty(a.n_rows, d, dtype=torch.float32, device=X.device)
    if plan is not None:                                       # (a caller-built plan, e.g. over a row list: whole rows, no column slices;
        sw, pl = 0, plan                                       #  the operand's own full plan is not built for it - ADVICE r03)
    else:
        sw, pl = a.plan_for(d, whole_row=epilogue is not None and epilogue.op != EPI_NONE)
    if partials is None:
        partials = pl.scratch(d, X.device)
    if accumulate:
        if epilogue is not None:
            raise RuntimeError("spmm: accumulate and an explicit epilogue are exclusive (pass Z = Y, alpha = 1)")
        epilogue = spmm_epilogue(EPI_NONE, 1.0, Y)
    col_scale = a.col_scale
    if col_scale is not None and a.val is None and a.nnz >= SPMM_LATENCY_NNZ:
        # HBM-bound graphs: A diag(c) X = A (c . X) - one dense pass over X (8 d bytes per row) instead of a second random 4-byte
        # gather per EDGE (PMC, 36.6 M edges: 9.9 GB of memory traffic per launch with the per-edge scale, 6.9 GB without)
        # (the scaled copy lives in a scratch cached per operand, shape and stream - as SpmmPlan.scratch is - so that a captured step does
        #  not add an X-sized block to the graph pool per product; rounding: acc += A (c x) instead of fma(c, x, acc), only in this regime)
        cache = a.plans.setdefault("_scaled_x", {})
        key = (tuple(X.shape), torch.cuda.current_stream().cuda_stream)
        Xs = cache.get(key)
        if Xs is None or Xs.device != X.device:
            Xs = cache[key] = torch.empty(X.shape, dtype=torch.float32, device=X.device)
        _lib.call("llmrec_scale_rows_f32", X.shape[0], d, _p(col_scale), _p(X), _ld(X), _p(Xs), _ld(Xs), _stream())
        X, col_scale = Xs, None
    if epilogue is None and os.environ.get("LLMREC_SPMM_PIPELINE", "1") == "0":          # (A/B switch: one task per lane group)
        epilogue = spmm_epilogue()
    rp, ci = pl.csr_of(a)
    _lib.call("llmrec_spmm_f32", a.n_rows, a.n_cols, _p(rp), _p(ci), _p(a.val), _p(a.row_scale),
              _p(col_scale), _p(X), _ld(X), _p(Y), _ld(Y), d, sw, _c.byref(pl.c_struct()), _p(partials),
              _c.byref(epilogue) if epilogue is not None else None, _stream())
    return Y


class _SpMM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, op: SparseOperand, X):
        ctx.op = op
        return spmm_raw(op.fwd, X)

    @staticmethod
    def backward(ctx, dY):
        return None, spmm_raw(ctx.op.bwd, dY)


def spmm(op, X: torch.Tensor) -> torch.Tensor:
    """Y = A X with autograd (dX = A^T dY). ``op``: SparseOperand or a torch sparse COO tensor."""
    if isinstance(op, torch.Tensor):
        op = operand_from_sparse_tensor(op)
    return _SpMM.apply(op, X)


# ---------------------------------------------------------------------------------------------
# R4: projection
# ---------------------------------------------------------------------------------------------
def linear_fwd_raw(X, W, b, out=None):
    _need_gpu(X, W, b)
    X, W = _rowmajor(X), _rowmajor(W)
    M, K = X.shape
    N = W.shape[0]
    Y = out if out is not None else torch.empty(M, N, dtype=torch.float32, device=X.device)
    _lib.call("llmrec_linear_fwd_f32", M, N, K, _p(X), _ld(X), _p(W), _ld(W), _p(b), _p(Y), _ld(Y), _stream())
    return Y


def linear_wgrad_raw(dY, X, dW, db, accumulate: bool):
    dY, X = _rowmajor(dY), _rowmajor(X)
    M, K = X.shape
    N = dY.shape[1]
    ws_bytes = _lib.query("llmrec_linear_wgrad_workspace_bytes", M, N, K)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=X.device)
    _lib.call("llmrec_linear_wgrad_f32", M, N, K, _p(dY), _ld(dY), _p(X), _ld(X), _p(dW), _ld(dW), _p(db),
              1 if accumulate else 0, _p(ws), ws_bytes, _stream())


class _Linear(torch.autograd.Function):
    """Y = X W^T + b for CONSTANT X (the reference's feature matrices): no dX."""

    @staticmethod
    def forward(ctx, X, W, b):
        ctx.save_for_backward(X)
        return linear_fwd_raw(X, W, b)

    @staticmethod
    def backward(ctx, dY):
        (X,) = ctx.saved_tensors
        dW = torch.empty(dY.shape[1], X.shape[1], dtype=torch.float32, device=X.device)
        db = torch.empty(dY.shape[1], dtype=torch.float32, device=X.device)
        linear_wgrad_raw(dY, X, dW, db, accumulate=False)
        return None, dW, db


def linear(X, W, b):
    return _Linear.apply(X, W, b)


class _LinearMulti(torch.autograd.Function):
    """Several constant inputs through ONE shared Linear (the reference's item_trans for its 5
    attribute keys, Models.py:33,150): n outputs, one accumulated weight gradient."""

    @staticmethod
    def forward(ctx, W, b, *Xs):
        ctx.Xs = Xs
        return tuple(linear_fwd_raw(X, W, b) for X in Xs)

    @staticmethod
    def backward(ctx, *dYs):
        Xs = ctx.Xs
        N = Xs and dYs[0].shape[1]
        dW = torch.empty(N, Xs[0].shape[1], dtype=torch.float32, device=dYs[0].device)
        db = torch.empty(N, dtype=torch.float32, device=dYs[0].device)
        for k, (X, dY) in enumerate(zip(Xs, dYs)):
            linear_wgrad_raw(dY, X, dW, db, accumulate=k > 0)
        return (dW, db) + (None,) * len(Xs)


def linear_multi(W, b, Xs: Sequence[torch.Tensor]):
    return _LinearMulti.apply(W, b, *Xs)


# ---------------------------------------------------------------------------------------------
# R3: softmax over d
# ---------------------------------------------------------------------------------------------
class _SoftmaxRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, Z):
        _need_gpu(Z)
        Z = _rowmajor(Z)
        Y = torch.empty(Z.shape, dtype=torch.float32, device=Z.device)
        _lib.call("llmrec_softmax_rows_fwd_f32", Z.shape[0], Z.shape[1], _p(Z), _ld(Z), _p(Y), _ld(Y), _stream())
        ctx.save_for_backward(Y)
        return Y

    @staticmethod
    def backward(ctx, dY):
        (Y,) = ctx.saved_tensors
        dY = _rowmajor(dY)
        dZ = torch.empty_like(Y)
        _lib.call("llmrec_softmax_rows_bwd_f32", Y.shape[0], Y.shape[1], _p(Y), _ld(Y), _p(dY), _ld(dY), _p(dZ), _ld(dZ), _stream())
        return dZ


def softmax_rows(Z):
    return _SoftmaxRows.apply(Z)


# ---------------------------------------------------------------------------------------------
# R6: fusion
# ---------------------------------------------------------------------------------------------
def _ptr_table(ts: Sequence[torch.Tensor]):
    n = len(ts)
    return (_c.c_void_p * max(n, 1))(*[t.data_ptr() for t in ts]), (_c.c_int64 * max(n, 1))(*[_ld(t) for t in ts])


class _Fuse(torch.autograd.Function):
    """out = mean_scale * sum(mean_terms) + sum_t rate_t * normalize(norm_terms[t])."""

    @staticmethod
    def forward(ctx, mean_scale, n_mean, rates, *terms):
        _need_gpu(*terms)
        terms = [_rowmajor(t) for t in terms]
        mean_terms, norm_terms = terms[:n_mean], terms[n_mean:]
        rows, d = terms[0].shape
        out = torch.empty(rows, d, dtype=torch.float32, device=terms[0].device)
        mp, ml = _ptr_table(mean_terms)
        npt, nl = _ptr_table(norm_terms)
        r = (_c.c_float * max(len(rates), 1))(*rates)
        _lib.call("llmrec_fuse_fwd_f32", rows, d, float(mean_scale), len(mean_terms), mp, ml, len(norm_terms), npt, nl, r,
                  _p(out), _ld(out), _stream())
        ctx.mean_scale, ctx.n_mean, ctx.rates = float(mean_scale), n_mean, list(rates)
        ctx.save_for_backward(*norm_terms)
        return out

    @staticmethod
    def backward(ctx, dOut):
        norm_terms = ctx.saved_tensors
        dOut = _rowmajor(dOut)
        rows, d = dOut.shape
        d_mean = None
        if ctx.n_mean:
            d_mean = torch.empty(rows, d, dtype=torch.float32, device=dOut.device)
            _lib.call("llmrec_axpy_f32", rows, d, ctx.mean_scale, None, _p(dOut), _ld(dOut), _p(d_mean), _ld(d_mean), 0, _stream())
        d_terms = [torch.empty(rows, d, dtype=torch.float32, device=dOut.device) for _ in norm_terms]
        if norm_terms:
            npt, nl = _ptr_table(norm_terms)
            dp, dl = _ptr_table(d_terms)
            r = (_c.c_float * len(ctx.rates))(*ctx.rates)
            _lib.call("llmrec_fuse_bwd_f32", rows, d, _p(dOut), _ld(dOut), len(norm_terms), npt, nl, r, dp, dl, 0, 0, 0.0, _stream())
        return (None, None, None) + (d_mean,) * ctx.n_mean + tuple(d_terms)


def fuse(mean_terms: List[torch.Tensor], norm_terms: List[torch.Tensor], rates: List[float]):
    """Layer mean + normalise-and-add (reference Models.py:185-197)."""
    if len(mean_terms) > CONST["LLMREC_MAX_TERMS"] or len(norm_terms) > CONST["LLMREC_MAX_TERMS"]:
        raise RuntimeError("fuse: too many terms")
    return _Fuse.apply(1.0 / max(len(mean_terms), 1), len(mean_terms), list(rates), *mean_terms, *norm_terms)


# ---------------------------------------------------------------------------------------------
# R7: BPR + prune
# ---------------------------------------------------------------------------------------------
def bpr_saved_floats(B: int) -> int:
    """LLMREC_BPR_SAVED_FLOATS(B) of include/llmrec_hip.h."""
    return 6 * B + 8


def bpr_plan_words(B: int) -> int:
    """LLMREC_BPR_PLAN_WORDS(B): 3 B sorted keys + 3 B int32 run lengths."""
    return 5 * B


def bpr_scatter_plan(users, pos, neg, n_valid=None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The deterministic scatter plan of one batch (llmrec_bpr_scatter_plan): sorted (id, slot) keys of the user side and of the item
    side + the run lengths, LLMREC_BPR_PLAN_WORDS(B) 64-bit words. Every BPR backward entry point takes it - rows that several samples share are added in one fixed order,
    as the reference's CPU index_put does (main.py:232-254), never with float atomics."""
    _need_gpu(users, pos, neg)
    B = users.numel()
    if out is None:
        out = torch.empty(max(bpr_plan_words(B), 1), dtype=torch.int64, device=users.device)
    elif out.numel() < bpr_plan_words(B) or out.dtype != torch.int64:
        raise RuntimeError("bpr_scatter_plan: plan buffer needs %d int64 words" % bpr_plan_words(B))
    _lib.call("llmrec_bpr_scatter_plan", _p(users), _p(pos), _p(neg), B, _p(n_valid), _p(out), _stream())
    return out


class BprProblem(_c.Structure):
    """llmrec_bpr_problem_t"""
    _fields_ = [("Eu", _c.c_void_p), ("ldu", _c.c_int64), ("Ei", _c.c_void_p), ("ldi", _c.c_int64),
                ("dEu", _c.c_void_p), ("lddu", _c.c_int64), ("dEi", _c.c_void_p), ("lddi", _c.c_int64),
                ("g_mf", _c.c_float), ("g_emb", _c.c_float)]


class LinearProblem(_c.Structure):
    """llmrec_linear_problem_t"""
    _fields_ = [("X", _c.c_void_p), ("ldx", _c.c_int64), ("M", _c.c_int64), ("K", _c.c_int32),
                ("W", _c.c_void_p), ("ldw", _c.c_int64), ("bias", _c.c_void_p), ("Y", _c.c_void_p), ("ldy", _c.c_int64),
                ("bias_scale", _c.c_void_p)]


class ZeroTensor(_c.Structure):
    """llmrec_zero_tensor_t"""
    _fields_ = [("p", _c.c_void_p), ("n", _c.c_int64)]


class FuseFwdProblem(_c.Structure):
    """llmrec_fuse_fwd_problem_t"""
    _fields_ = [("rows", _c.c_int64), ("mean_scale", _c.c_float), ("n_mean", _c.c_int32), ("mean_terms", _c.c_void_p), ("mean_ld", _c.c_void_p),
                ("n_norm", _c.c_int32), ("norm_terms", _c.c_void_p), ("norm_ld", _c.c_void_p), ("rates", _c.c_void_p),
                ("out", _c.c_void_p), ("ldo", _c.c_int64)]


class FuseBwdProblem(_c.Structure):
    """llmrec_fuse_bwd_problem_t"""
    _fields_ = [("rows", _c.c_int64), ("dOut", _c.c_void_p), ("lddo", _c.c_int64), ("n_norm", _c.c_int32), ("norm_terms", _c.c_void_p),
                ("norm_ld", _c.c_void_p), ("rates", _c.c_void_p), ("d_terms", _c.c_void_p), ("d_ld", _c.c_void_p),
                ("src_terms", _c.c_void_p), ("src_ld", _c.c_void_p), ("n_reg_terms", _c.c_int32), ("reg_two_coef", _c.c_float),
                ("row_flags", _c.c_void_p), ("row_stamp", _c.c_void_p)]


class WgradProblem(_c.Structure):
    """llmrec_wgrad_problem_t"""
    _fields_ = [("dY", _c.c_void_p), ("lddy", _c.c_int64), ("X", _c.c_void_p), ("ldx", _c.c_int64), ("M", _c.c_int64),
                ("db_row_weight", _c.c_void_p), ("row_list", _c.c_void_p), ("n_rows", _c.c_void_p), ("rows_expected", _c.c_int64)]


class WgradTarget(_c.Structure):
    """llmrec_wgrad_target_t"""
    _fields_ = [("n_problems", _c.c_int32), ("problems", _c.c_void_p), ("K", _c.c_int32), ("dW", _c.c_void_p), ("lddw", _c.c_int64),
                ("db", _c.c_void_p), ("accumulate", _c.c_int32), ("block_budget", _c.c_int32)]


class WgradUpdate(_c.Structure):
    """llmrec_wgrad_update_t"""
    _fields_ = [("W", _c.c_void_p), ("m_W", _c.c_void_p), ("v_W", _c.c_void_p), ("b", _c.c_void_p), ("m_b", _c.c_void_p), ("v_b", _c.c_void_p),
                ("g_scale", _c.c_float)]


def _wgrad_targets(targets, block_budget: int = 0):
    """targets: [(pairs, dW, db, accumulate)] -> (ctypes array, keep-alive list, N); block_budget: llmrec_wgrad_target_t.block_budget"""
    arr = (WgradTarget * len(targets))()
    arr[0].block_budget = int(block_budget)
    keep = []
    N = targets[0][1].shape[0]
    for i, (pairs, dW, db, accumulate) in enumerate(targets):
        probs = (WgradProblem * len(pairs))()
        for j, pair in enumerate(pairs):                       # (dY, X[, db_row_weight[, (row_list, n_rows, rows_expected)]])
            dY, X = pair[:2]
            _need_gpu(dY, X)
            probs[j].dY, probs[j].lddy, probs[j].X, probs[j].ldx, probs[j].M = dY.data_ptr(), _ld(dY), X.data_ptr(), _ld(X), X.shape[0]
            probs[j].db_row_weight = pair[2].data_ptr() if len(pair) > 2 and pair[2] is not None else None
            if len(pair) > 3 and pair[3] is not None:          # the rows of dY that can be non-zero (llmrec_wgrad_problem_t.row_list)
                rl, nr, expected = pair[3]
                if rl.dtype != torch.int32 or nr.dtype != torch.int32 or rl.numel() < X.shape[0] + 32:
                    raise RuntimeError("linear_wgrad_multi: row_list must be int32 with M + 32 entries, n_rows an int32 device scalar")
                probs[j].row_list, probs[j].n_rows, probs[j].rows_expected = rl.data_ptr(), nr.data_ptr(), int(expected or 0)
                keep.append((rl, nr))
        keep.append(probs)
        arr[i].n_problems, arr[i].problems, arr[i].K = len(pairs), _c.cast(probs, _c.c_void_p), dW.shape[1]
        arr[i].dW, arr[i].lddw, arr[i].db, arr[i].accumulate = dW.data_ptr(), _ld(dW), (db.data_ptr() if db is not None else None), 1 if accumulate else 0
    return arr, keep, N


def batch_reach_rows(users, pos, neg, n_valid, by_item: "Csr", flags: torch.Tensor, row_list: torch.Tensor, n_rows: torch.Tensor):
    """llmrec_batch_reach_rows: the ascending list of the user rows a batch reaches (its users + every user adjacent to one of its
    positive / negative items; by_item = the CSR whose rows are items and whose columns are users). flags: [n_users] uint8 scratch,
    all-zero on entry and on return; row_list: int32 [n_users + 32]; n_rows: int32 [1]. Two launches, no host sync."""
    _need_gpu(users, pos, neg, flags, row_list, n_rows)
    n_users, n_items = by_item.n_cols, by_item.n_rows
    if flags.numel() < n_users or row_list.numel() < n_users + 32 or flags.dtype != torch.uint8 or row_list.dtype != torch.int32:
        raise RuntimeError("batch_reach_rows: flags [n_users] uint8, row_list [n_users + 32] int32")
    _lib.call("llmrec_batch_reach_rows", n_users, n_items, _p(users), _p(pos), _p(neg), users.numel(), _p(n_valid), _p(by_item.rowptr),
              _p(by_item.colidx), _p(flags), _p(row_list), _p(n_rows), _stream())


def linear_wgrad_multi_workspace(targets, block_budget: int = 0) -> int:
    """Bytes of workspace llmrec_linear_wgrad_multi_bf16x3 needs for these targets; -1 = outside its fast path."""
    arr, keep, N = _wgrad_targets(targets, block_budget)
    return _lib.query("llmrec_linear_wgrad_multi_workspace_bytes", len(targets), arr, N)


def linear_wgrad_multi(targets, ws: Optional[torch.Tensor] = None, update=None, block_budget: int = 0):
    """The bf16x3 weight gradients of several Linears in one launch + one reduction launch (llmrec_linear_wgrad_multi_bf16x3).
    targets: [(pairs, dW, db, accumulate)], pairs = [(dY, X)] as in linear_wgrad_grouped; all dW have N rows.
    update = (FusedAdamW, [(W, b)] per target): the AdamW update of those parameters rides in the reduction launch
    (llmrec_linear_wgrad_multi_adamw_bf16x3; the optimizer's step counter has been advanced already).
    block_budget: resident blocks per round the launch is laid out for (0 = 256: one per CU)."""
    arr, keep, N = _wgrad_targets(targets, block_budget)
    need = _lib.query("llmrec_linear_wgrad_multi_workspace_bytes", len(targets), arr, N)
    if need < 0:
        raise RuntimeError("linear_wgrad_multi: shapes outside the fast path (use linear_wgrad_grouped per target)")
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=targets[0][1].device)
    if update is None:
        _lib.call("llmrec_linear_wgrad_multi_bf16x3", len(targets), arr, N, _p(ws), ws.numel(), _stream())
        return
    opt, params = update
    if opt.dev_state is None:
        raise RuntimeError("linear_wgrad_multi: the optimizer's step counter has not been advanced (FusedAdamW.advance())")
    upd = (WgradUpdate * len(targets))()
    for i, (W, b) in enumerate(params):
        stW = opt.moments(W)
        upd[i].W, upd[i].m_W, upd[i].v_W = W.data_ptr(), stW[0].data_ptr(), stW[1].data_ptr()
        if b is not None:
            stb = opt.moments(b)
            upd[i].b, upd[i].m_b, upd[i].v_b = b.data_ptr(), stb[0].data_ptr(), stb[1].data_ptr()
        gs = float(opt.grad_scale.get(W, 1.0))
        if b is not None and float(opt.grad_scale.get(b, 1.0)) != gs:
            raise RuntimeError("linear_wgrad_multi: W and b of one target need the same grad_scale")
        upd[i].g_scale = gs
    _lib.call("llmrec_linear_wgrad_multi_adamw_bf16x3", len(targets), arr, N, _p(ws), ws.numel(), upd, _p(opt.dev_state), opt.lr, opt.betas[0],
              opt.betas[1], opt.eps, opt.wd, _stream())


def linear_wgrad_grouped(pairs, dW, db, accumulate: bool, ws: Optional[torch.Tensor] = None, precision: str = "f32"):
    """dW (+)= sum_p dY_p^T X_p for several (dY, X) pairs sharing one weight (one launch + reduce).
    precision "bf16x3": llmrec_linear_wgrad_grouped_bf16x3 (three-term bf16 split, fp32-class error)."""
    arr = (WgradProblem * len(pairs))()
    M_total = 0
    for i, pair in enumerate(pairs):                           # (dY, X) or (dY, X, db_row_weight)
        dY, X = pair[:2]
        _need_gpu(dY, X)
        arr[i].dY, arr[i].lddy, arr[i].X, arr[i].ldx, arr[i].M = dY.data_ptr(), _ld(dY), X.data_ptr(), _ld(X), X.shape[0]
        arr[i].db_row_weight = pair[2].data_ptr() if len(pair) > 2 and pair[2] is not None else None
        if len(pair) > 3 and pair[3] is not None:              # row list: served by the bf16x3 128-wide organisation only (else refused)
            arr[i].row_list, arr[i].n_rows, arr[i].rows_expected = pair[3][0].data_ptr(), pair[3][1].data_ptr(), int(pair[3][2] or 0)
        M_total += X.shape[0]
    N, K = dW.shape
    need = _lib.query("llmrec_linear_wgrad_workspace_bytes", M_total, N, K)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=dW.device)
    _lib.call("llmrec_linear_wgrad_grouped_bf16x3" if precision == "bf16x3" else "llmrec_linear_wgrad_grouped_f32",
              len(pairs), arr, N, K, _p(dW), _ld(dW), _p(db), 1 if accumulate else 0, _p(ws), ws.numel(), _stream())


def linear_fwd_grouped(jobs, N: int, precision: str = "f32"):
    """jobs: list of (X, W, bias, out) or (X, W, bias, out, bias_scale) - one launch. precision "f32": exact fp32 MFMA
    (llmrec_linear_fwd_grouped_f32); "bf16x3": three-term bf16 split, six bf16 MFMAs, fp32-class
    error (llmrec_linear_fwd_grouped_bf16x3). bias_scale [M]: out[r] = X[r] W^T + bias_scale[r] * bias."""
    arr = (LinearProblem * len(jobs))()
    for i, job in enumerate(jobs):
        X, W, b, out = job[:4]
        _need_gpu(X, W, b, out)
        arr[i].bias_scale = job[4].data_ptr() if len(job) > 4 and job[4] is not None else None
        arr[i].X, arr[i].ldx, arr[i].M, arr[i].K = X.data_ptr(), _ld(X), X.shape[0], X.shape[1]
        arr[i].W, arr[i].ldw, arr[i].bias = W.data_ptr(), _ld(W), (b.data_ptr() if b is not None else None)
        arr[i].Y, arr[i].ldy = out.data_ptr(), _ld(out)
    name = {"f32": "llmrec_linear_fwd_grouped_f32", "bf16x3": "llmrec_linear_fwd_grouped_bf16x3"}[precision]
    _lib.call(name, len(jobs), arr, N, _stream())


class _BprPrune(torch.autograd.Function):
    @staticmethod
    def forward(ctx, Eu, Ei, users, pos, neg, remember_rate, decay, batch_size_flag, n_valid):
        _need_gpu(Eu, Ei, users, pos, neg)
        Eu, Ei = _rowmajor(Eu), _rowmajor(Ei)
        B = users.numel()
        d = Eu.shape[1]
        out = torch.empty(2, dtype=torch.float32, device=Eu.device)
        saved = torch.empty(bpr_saved_floats(B), dtype=torch.float32, device=Eu.device)
        _lib.call("llmrec_bpr_prune_fwd_f32", _p(Eu), _ld(Eu), _p(Ei), _ld(Ei), d, _p(users), _p(pos), _p(neg), B, _p(n_valid),
                  float(remember_rate), float(decay), float(batch_size_flag), _p(out), _p(saved), _stream())
        ctx.save_for_backward(Eu, Ei, users, pos, neg, saved)
        ctx.n_valid, ctx.decay, ctx.bsz = n_valid, float(decay), float(batch_size_flag)
        return out

    @staticmethod
    def backward(ctx, g):
        Eu, Ei, users, pos, neg, saved = ctx.saved_tensors
        g = g.contiguous()
        dEu, dEi = torch.zeros_like(Eu), torch.zeros_like(Ei)
        plan = bpr_scatter_plan(users, pos, neg, ctx.n_valid)
        _lib.call("llmrec_bpr_prune_bwd_f32", _p(Eu), _ld(Eu), _p(Ei), _ld(Ei), Eu.shape[1], _p(users), _p(pos), _p(neg),
                  users.numel(), _p(ctx.n_valid), ctx.decay, ctx.bsz, _p(saved), _p(g), _p(dEu), _ld(dEu), _p(dEi), _ld(dEi), _p(plan), _stream())
        return dEu, dEi, None, None, None, None, None, None, None


def bpr_prune(Eu, Ei, users, pos, neg, drop_rate: float, decay: float, batch_size_flag: float, n_valid=None):
    """Returns a 2-vector [mf_loss, emb_loss] (reference main.py:330-342 with prune_loss :158-165).
    users/pos/neg: int64 device vectors indexing rows of Eu / Ei."""
    if users.dtype != torch.int64 or pos.dtype != torch.int64 or neg.dtype != torch.int64:
        raise RuntimeError("bpr_prune: int64 index tensors expected")
    if users.numel() > CONST["LLMREC_BPR_MAX_B"]:
        raise RuntimeError("bpr_prune: batch of %d exceeds LLMREC_BPR_MAX_B" % users.numel())
    remember = 1 - drop_rate                                       # python float, as main.py:161
    return _BprPrune.apply(Eu, Ei, users, pos, neg, remember, decay, batch_size_flag, n_valid)


# ---------------------------------------------------------------------------------------------
# R8: regulariser
# ---------------------------------------------------------------------------------------------
class _SumSq(torch.autograd.Function):
    """coef * sum_t ||X_t||_F^2 as a 1-element tensor."""

    @staticmethod
    def forward(ctx, coef, *Xs):
        _need_gpu(*Xs)
        Xs = [_rowmajor(x) for x in Xs]
        out = torch.empty(1, dtype=torch.float32, device=Xs[0].device)
        ws_bytes = _lib.query("llmrec_sumsq_workspace_bytes", 0, 0)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=Xs[0].device)
        for k, x in enumerate(Xs):
            _lib.call("llmrec_sumsq_f32", x.shape[0], x.shape[1], _p(x), _ld(x), float(coef), 1 if k else 0, _p(out), _p(ws), ws_bytes, _stream())
        ctx.coef = float(coef)
        ctx.save_for_backward(*Xs)
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        outs = []
        for x in ctx.saved_tensors:
            dx = torch.empty(x.shape, dtype=torch.float32, device=x.device)
            _lib.call("llmrec_axpy_f32", x.shape[0], x.shape[1], 2.0 * ctx.coef, _p(g), _p(x), _ld(x), _p(dx), _ld(dx), 0, _stream())
            outs.append(dx)
        return (None,) + tuple(outs)


def sumsq(coef: float, Xs: Sequence[torch.Tensor]):
    return _SumSq.apply(coef, *Xs)


# ---------------------------------------------------------------------------------------------
# R8: AdamW
# ---------------------------------------------------------------------------------------------
class AdamwTensor(_c.Structure):
    """llmrec_adamw_tensor_t"""
    _fields_ = [("p", _c.c_void_p), ("g", _c.c_void_p), ("m", _c.c_void_p), ("v", _c.c_void_p), ("n", _c.c_int64), ("g_scale", _c.c_float),
                ("g_out", _c.c_void_p)]


class ZeroRowsJob(_c.Structure):
    """llmrec_zero_rows_job_t"""
    _fields_ = [("ids", _c.c_void_p), ("dst", _c.c_void_p), ("ldd", _c.c_int64), ("d", _c.c_int32)]


class FusedAdamW:
    """torch.optim.AdamW(lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01) semantics as built
    at reference main.py:100-104, one HIP launch per parameter, bias corrections kept on device
    (graph-capturable). Parameters without a gradient are skipped, as torch does."""

    def __init__(self, params, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01):
        self.params = [p for p in params]
        self.lr, self.betas, self.eps, self.wd = float(lr), betas, float(eps), float(weight_decay)
        self.state = {}
        self.dev_state = None
        self.grad_scale = {}                    # {param: s}: the gradient of that parameter is s * param.grad (default 1)

    def zero_grad(self, set_to_none: bool = True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    @torch.no_grad()
    def advance(self):
        """Advance the device step counter / bias corrections (the first half of step()). A fused step calls it
        early on a side stream so that only the multi-tensor update itself remains on the critical path."""
        if self.dev_state is None:
            dev = next(p for p in self.params if p.grad is not None).device
            self.dev_state = torch.zeros(3, dtype=torch.float32, device=dev)
        _lib.call("llmrec_adamw_advance", _p(self.dev_state), self.lr, self.betas[0], self.betas[1], _stream())

    @torch.no_grad()
    def step_params(self, params, sources=None, zero_rows=None):
        """The update of a SUBSET of the parameters (their gradients are final), the step counter having been advanced already
        (advance()): a fused step updates its embedding tables while the weight-gradient GEMM of the Linears is still running.
        sources: {param: (G, scale)} - the gradient of that parameter is scale * G (another buffer); it is ALSO stored into param.grad by the
        same launch (llmrec_adamw_tensor_t.g_out: no separate scaling pass). zero_rows: (jobs [(ids int64, dst [rows, width])], B_cap, n_valid_dev) -
        the row-wise clean-up of the step's scatter targets rides in the same launch (llmrec_adamw_multi_zero_rows_f32)."""
        self._update([p for p in params if p.grad is not None], sources=sources, zero_rows=zero_rows)

    @torch.no_grad()
    def step(self, advanced: bool = False):
        live = [p for p in self.params if p.grad is not None]
        if not live:
            return
        if not advanced:
            self.advance()
        self._update(live)

    def moments(self, p):
        """(exp_avg, exp_avg_sq) of a parameter, created on first use."""
        st = self.state.get(p)
        if st is None:
            if not p.is_contiguous():
                raise RuntimeError("FusedAdamW: contiguous parameters expected")
            st = self.state[p] = (torch.zeros_like(p), torch.zeros_like(p))
        return st

    def _update(self, live, sources=None, zero_rows=None):
        if not live and zero_rows is None:
            return
        _need_gpu(*live)
        cap = CONST["LLMREC_ADAMW_MAX_TENSORS"]
        if zero_rows is not None and len(live) > cap:
            raise RuntimeError("FusedAdamW: the clean-up rides with at most %d tensors" % cap)
        for lo in range(0, max(len(live), 1), cap):
            group = live[lo:lo + cap]
            arr = (AdamwTensor * max(len(group), 1))()
            keep = []
            for i, p in enumerate(group):
                st = self.state.get(p)
                if st is None:
                    st = self.state[p] = (torch.zeros_like(p), torch.zeros_like(p))
                if not p.is_contiguous():
                    raise RuntimeError("FusedAdamW: contiguous parameters expected")
                src = sources.get(p) if sources else None
                if src is not None:
                    g, scale = src
                    if not g.is_contiguous() or g.numel() != p.numel() or not p.grad.is_contiguous():
                        raise RuntimeError("FusedAdamW: a gradient source is a contiguous tensor of the parameter's size")
                    arr[i].g_scale, arr[i].g_out = float(scale) * float(self.grad_scale.get(p, 1.0)), p.grad.data_ptr()
                else:
                    g = p.grad.contiguous()
                    arr[i].g_scale, arr[i].g_out = float(self.grad_scale.get(p, 1.0)), None
                keep.append(g)
                arr[i].p, arr[i].g, arr[i].m, arr[i].v, arr[i].n = p.data_ptr(), g.data_ptr(), st[0].data_ptr(), st[1].data_ptr(), p.numel()
            if zero_rows is None:
                _lib.call("llmrec_adamw_multi_f32", len(group), arr, _p(self.dev_state), self.lr, self.betas[0], self.betas[1],
                          self.eps, self.wd, _stream())
            else:
                jobs, b_cap, n_valid = zero_rows
                if len(jobs) > CONST["LLMREC_ZERO_ROWS_MAX_JOBS"]:
                    raise RuntimeError("FusedAdamW: %d clean-up jobs exceed LLMREC_ZERO_ROWS_MAX_JOBS" % len(jobs))
                jarr = (ZeroRowsJob * max(len(jobs), 1))()
                for j, (ids, dst) in enumerate(jobs):
                    jarr[j].ids, jarr[j].dst, jarr[j].ldd, jarr[j].d = ids.data_ptr(), dst.data_ptr(), _ld(dst), dst.shape[1]
                if self.dev_state is None:
                    raise RuntimeError("FusedAdamW: advance() first")
                _lib.call("llmrec_adamw_multi_zero_rows_f32", len(group), arr, _p(self.dev_state), self.lr, self.betas[0], self.betas[1],
                          self.eps, self.wd, len(jobs), jarr, int(b_cap), _p(n_valid), _stream())


# ---------------------------------------------------------------------------------------------
# R9/R10: scoring + top-K, R11: sampler
# ---------------------------------------------------------------------------------------------
TOPK_MODES = {"exact": 0, "prefilter": 1}


def topk_mode(mode=None, n_items: int = 0, d: int = 64, K: int = 50) -> int:
    """llmrec_score_topk_mode_f32's mode: "exact" = every score by the exact-fp32 MFMA chain; "prefilter" = bf16 sweep keeping the 64 best
    by approximate score, exact re-ranking + verification, exact sweep for the tiles that fail it (bit-identical lists and scores);
    "auto" (default; LLMREC_TOPK_MODE overrides) = prefilter whenever K leaves room to verify. Measured on MI355X at 16 384 users, d = 64
    (profiles/r06_topk_crossover.json): 1.41 x (10 K items), 1.96 x (33 K), 2.09 x (66 K), 2.03 x (131 K), 1.93 x (262 K), 1.96 x (524 K),
    1.99 x (10^6); 65 536 users x 10^6 items 107.7 -> 46.4 ms, d = 128 50.2 -> 29.7 ms. Up to round 5 the mode stopped paying at 524 K
    items (0.84 x at 10^6: every 16-user tile streamed the 256 MB fragment table through the L2-miss path); since round 6 tables beyond
    131 072 items are swept in item PARTS of 8 MB of fragments with part-major block ids (csrc/topk.hip plan_parts), so that the blocks
    resident at one time share each part through the L2."""
    if mode is None:
        mode = os.environ.get("LLMREC_TOPK_MODE", "auto")
    if mode == "auto":
        mode = "prefilter" if K <= CONST["LLMREC_TOPK_PREFILTER_MAX_K"] else "exact"
    if mode not in TOPK_MODES:
        raise RuntimeError("top-K mode %r (auto | exact | prefilter)" % (mode,))
    return TOPK_MODES[mode]


def topk_set_part_items(items: int = 0):
    """llmrec_topk_set_part_items: the bf16 sweep's item parts (0 = the library's policy, -1 = never, else items per part). Process-wide;
    workspaces sized before a change are stale."""
    _lib.call("llmrec_topk_set_part_items", int(items))


def score_topk(Eu, Ei, query_users: torch.Tensor, train: Optional[Csr], K: int, mode=None, stats: Optional[dict] = None):
    """Masked top-K item ids (int32 [n_query, K], -1 = none) and scores for the listed users. stats (a dict; synchronises): receives
    "fallback_tiles" / "tiles" - the user tiles the bf16 mode's verification sent to the exact sweep -, "drains" - the pool drains of the bf16
    sweep, summed over its blocks - and "bitmap_rows" - the train rows its blocks swept as bitmaps (None outside the bf16 mode)."""
    _need_gpu(Eu, Ei, query_users)
    Eu, Ei = _rowmajor(Eu.detach()), _rowmajor(Ei.detach())
    q = query_users.to(torch.int64).contiguous()
    n = q.numel()
    idx = torch.empty(n, K, dtype=torch.int32, device=Eu.device)
    sc = torch.empty(n, K, dtype=torch.float32, device=Eu.device)
    ws = topk_workspace(n, Ei.shape[0], Eu.device, Eu.shape[1])
    _lib.call("llmrec_score_topk_mode_f32", n, _p(q), _p(Eu), _ld(Eu), _p(Ei), _ld(Ei), Ei.shape[0], Eu.shape[1],
              _p(train.rowptr) if train is not None else None, _p(train.colidx) if train is not None else None,
              K, _p(idx), _p(sc), _p(ws), ws.numel() if ws is not None else 0, topk_mode(mode, Ei.shape[0], Eu.shape[1], K), _stream())
    if stats is not None:
        off = _lib.query("llmrec_score_topk_stats_offset", n, Ei.shape[0])
        stats["tiles"] = (n + 15) // 16
        pre = topk_mode(mode, Ei.shape[0], Eu.shape[1], K) == 1 and K <= CONST["LLMREC_TOPK_PREFILTER_MAX_K"]
        stats["fallback_tiles"] = int(ws[off + 4:off + 8].view(torch.int32).item()) if pre else 0
        stats["drains"] = int(ws[off:off + 4].view(torch.int32).item()) if pre else None                # pool drains of the bf16 sweep, summed over its blocks
        stats["bitmap_rows"] = int(ws[off + 8:off + 12].view(torch.int32).item()) if pre else None     # long train rows swept as per-block bitmaps (bf16 mode's launch)
    return idx, sc


def topk_workspace(n_query: int, n_items: int, device, d: int = 64) -> Optional[torch.Tensor]:
    """Scratch for llmrec_score_topk_ws_f32: the item table in fragment order + the part lists of the left-over user tiles."""
    nbytes = _lib.query("llmrec_score_topk_workspace_bytes", n_query, n_items, d)
    return torch.empty(nbytes, dtype=torch.uint8, device=device) if nbytes > 0 else None


def export_candidates(Eu, Ei, k: int = 10, query_users: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Stage-1 candidate export: the top-k item ids per user WITHOUT masking, i.e. what the reference
    README's `torch.topk(user_emb @ item_emb.T, k=10)` snippet (README.md:243-247) produces for
    `candidate_indices`, as an int64 [n, k] tensor - computed by the scoring kernel, never
    materialising U x I. Ties are broken by ascending item id."""
    q = torch.arange(Eu.shape[0], device=Eu.device) if query_users is None else query_users
    idx, _ = score_topk(Eu, Ei, q, None, k)
    return idx.to(torch.int64)


def scores(Eu, Ei, query_users: torch.Tensor):
    _need_gpu(Eu, Ei, query_users)
    Eu, Ei = _rowmajor(Eu.detach()), _rowmajor(Ei.detach())
    q = query_users.to(torch.int64).contiguous()
    S = torch.empty(q.numel(), Ei.shape[0], dtype=torch.float32, device=Eu.device)
    _lib.call("llmrec_scores_f32", q.numel(), _p(q), _p(Eu), _ld(Eu), _p(Ei), _ld(Ei), Ei.shape[0], Eu.shape[1], _p(S), _ld(S), _stream())
    return S


def topk_hits(topk_idx: torch.Tensor, query_users: torch.Tensor, test_rowptr: torch.Tensor, test_colidx: torch.Tensor):
    q = query_users.to(torch.int64).contiguous()
    hits = torch.empty(topk_idx.shape, dtype=torch.uint8, device=topk_idx.device)
    _lib.call("llmrec_topk_hits", q.numel(), _p(q), topk_idx.shape[1], _p(topk_idx), _p(test_rowptr), _p(test_colidx), _p(hits), _stream())
    return hits


def topk_eval_sums(idx: torch.Tensor, query_users: torch.Tensor, test_rowptr: torch.Tensor, test_colidx: torch.Tensor, Ks,
                   out: Optional[torch.Tensor] = None, ws: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[4, len(Ks)] float64: the sums over the query users of precision / recall / ndcg / hit-ratio at every cut-off (llmrec_topk_eval_sums:
    hits, per-user metrics and their sums in two launches). out: a device tensor or a PINNED host tensor (written by the kernel itself)."""
    _need_gpu(idx, query_users, test_rowptr, test_colidx)
    n, K = idx.shape
    Ks = [int(k) for k in Ks]
    if out is None:
        out = torch.empty(4, len(Ks), dtype=torch.float64, device=idx.device)
    need = _lib.query("llmrec_topk_eval_sums_workspace_bytes", n, len(Ks))
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=idx.device)
    arr = (_c.c_int32 * len(Ks))(*Ks)
    _lib.call("llmrec_topk_eval_sums", n, _p(query_users.to(torch.int64).contiguous()), K, _p(idx.contiguous()), _p(test_rowptr), _p(test_colidx),
              len(Ks), arr, _p(ws), ws.numel(), _c.c_void_p(out.data_ptr()), _stream())
    return out


def sample_bpr(seed: int, step: int, exist_users: torch.Tensor, n_items: int, train: Csr, B: int):
    dev = exist_users.device
    u = torch.empty(B, dtype=torch.int64, device=dev)
    p = torch.empty(B, dtype=torch.int64, device=dev)
    n = torch.empty(B, dtype=torch.int64, device=dev)
    _lib.call("llmrec_sample_bpr", seed, step, exist_users.numel(), _p(exist_users), n_items, _p(train.rowptr), _p(train.colidx),
              B, _p(u), _p(p), _p(n), _stream())
    return u, p, n


def sample_batch(seed: int, step_dev: torch.Tensor, exist_users: torch.Tensor, n_items: int, train: Csr, B_global: int,
                 slice_begin: int, B: int, n_aug: int, aug_pos: Optional[torch.Tensor], aug_neg: Optional[torch.Tensor],
                 users: torch.Tensor, pos: torch.Tensor, neg: torch.Tensor, n_valid: torch.Tensor):
    """llmrec_sample_batch: this rank's slice of the step's BPR triples + the LLM-augmented triples, written
    into the given (static) buffers; step_dev (int64[1], device) is the step counter the launch advances."""
    _need_gpu(step_dev, exist_users, users, pos, neg, n_valid)
    if users.numel() < B + n_aug or step_dev.dtype != torch.int64 or n_valid.dtype != torch.int32:
        raise RuntimeError("sample_batch: buffers of B + n_aug int64 entries, int64 step counter, int32 n_valid expected")
    _lib.call("llmrec_sample_batch", seed, _p(step_dev), exist_users.numel(), _p(exist_users), n_items, _p(train.rowptr), _p(train.colidx),
              B_global, slice_begin, B, n_aug, _p(aug_pos), _p(aug_neg), _p(users), _p(pos), _p(neg), _p(n_valid), _stream())


def topk_metrics(idx: torch.Tensor, hits: torch.Tensor, query_users: torch.Tensor, test_rowptr: torch.Tensor, Ks) -> torch.Tensor:
    """Per-user [n, 4, len(Ks)] float64 (precision, recall, ndcg, hit_ratio) on the device (llmrec_topk_metrics)."""
    _need_gpu(idx, hits, query_users, test_rowptr)
    n, K = idx.shape
    Ks = [int(k) for k in Ks]
    out = torch.empty(n, 4, len(Ks), dtype=torch.float64, device=idx.device)
    arr = (_c.c_int32 * len(Ks))(*Ks)
    _lib.call("llmrec_topk_metrics", n, _p(query_users.to(torch.int64).contiguous()), K, _p(hits.contiguous()), _p(idx.contiguous()),
              _p(test_rowptr), len(Ks), arr, _p(out), _stream())
    return out
