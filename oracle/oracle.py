"""TEST INFRASTRUCTURE ONLY - CPU restatement of LLMRec's Stage-2 hot path.

This module is the parity oracle for the HIP path. It is imported only by ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``; nothing under
``llmrec_amd/`` or the drop-in modules (``main.py``, ``Models.py``, ``utility/``) may import it.

It restates, in plain torch-CPU / numpy (the same arithmetic libraries the reference itself
calls - the reference has no arithmetic of its own), exactly what the reference
computes on the path, function by function, each citing the reference lines it follows.
Gradients come from torch autograd, as in the reference, which makes this an independent check
of the hand-written backward kernels.

Pinning: the reference ships no golden vectors, so this oracle is pinned against
vectors captured from the unmodified reference run in the build container
(``oracle/make_golden.py`` -> ``tests/golden/*``; ``tests/test_oracle_golden.py``).
"""
from __future__ import annotations

import heapq
import math
import random as _pyrandom
from dataclasses import dataclass, field
from typing import Dict, List, Sequence

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------------
# configuration (defaults = reference utility/parser.py:7-54)
# --------------------------------------------------------------------------------------------
@dataclass
class Config:
    embed_size: int = 64
    n_layers: int = 2                      # len(eval(--weight_size)), reference main.py:49-50
    batch_size: int = 1024                 # the FLAG value; divisor of the BPR regulariser
    decay: float = 1e-5                    # eval(--regs)[0], main.py:51-52
    feat_reg_decay: float = 1e-5
    model_cat_rate: float = 0.02
    user_cat_rate: float = 2.8
    item_cat_rate: float = 0.005
    aug_mf_rate: float = 0.012
    mm_mf_rate: float = 0.0001
    prune_loss_drop_rate: float = 0.71
    aug_sample_rate: float = 0.1
    lr: float = 1e-4
    Ks: Sequence[int] = (10, 20, 50)
    keys: Sequence[str] = ("year", "title", "director", "country", "language")

    @staticmethod
    def from_args(a: dict, keys) -> "Config":
        ws = eval(a["weight_size"]) if isinstance(a["weight_size"], str) else a["weight_size"]
        regs = eval(a["regs"]) if isinstance(a["regs"], str) else a["regs"]
        ks = eval(a["Ks"]) if isinstance(a["Ks"], str) else a["Ks"]
        return Config(embed_size=a["embed_size"], n_layers=len(ws), batch_size=a["batch_size"],
                      decay=regs[0], feat_reg_decay=a["feat_reg_decay"],
                      model_cat_rate=a["model_cat_rate"], user_cat_rate=a["user_cat_rate"],
                      item_cat_rate=a["item_cat_rate"], aug_mf_rate=a["aug_mf_rate"],
                      mm_mf_rate=a["mm_mf_rate"], prune_loss_drop_rate=a["prune_loss_drop_rate"],
                      aug_sample_rate=a["aug_sample_rate"], lr=a["lr"], Ks=tuple(ks), keys=tuple(keys))


# --------------------------------------------------------------------------------------------
# R1 - graph normalisation: reference main.py:114-126 (csr_norm, mean_flag=True), :128-134
# --------------------------------------------------------------------------------------------
def row_scale(csr) -> np.ndarray:
    """(rowsum + 1e-8) ** -0.5, inf -> 0; float64 like numpy does it in the reference."""
    rowsum = np.asarray(csr.sum(1)).reshape(-1)
    s = np.power(rowsum + 1e-8, -0.5)
    s[np.isinf(s)] = 0.0
    return s


def normalized_graphs(train_mat, dtype=torch.float32):
    """Return (A_ui, A_iu) as torch sparse COO tensors: diag(s_u) R and diag(s_i) R^T."""
    import scipy.sparse as sp

    def to_tensor(m):
        m = m.tocoo()
        idx = torch.from_numpy(np.vstack((m.row, m.col)).astype(np.int64))
        val = torch.from_numpy(m.data)
        return torch.sparse_coo_tensor(idx, val, torch.Size(m.shape)).to(dtype)

    ui = sp.csr_matrix(train_mat)
    iu = ui.T
    a_ui = sp.diags(row_scale(ui)) * ui
    a_iu = sp.diags(row_scale(iu)) * iu
    return to_tensor(a_ui), to_tensor(a_iu)


# --------------------------------------------------------------------------------------------
# R2-R6 - forward: reference Models.py:127-199
# --------------------------------------------------------------------------------------------
def forward(params: Dict[str, torch.Tensor], feats: Dict[str, torch.Tensor], a_ui, a_iu, cfg: Config):
    """params: image_trans.{weight,bias}, text_trans.*, user_trans.*, item_trans.*,
    user_id_embedding.weight, item_id_embedding.weight.  feats: image, text, user, attr/<key>.
    Returns a dict with the tensors of the reference's 14-tuple (Models.py:199)."""
    spmm = torch.sparse.mm
    lin = lambda x, name: F.linear(x, params[name + ".weight"], params[name + ".bias"])  # Models.py:145-150
    p_img = lin(feats["image"], "image_trans")
    p_txt = lin(feats["text"], "text_trans")
    p_usr = lin(feats["user"], "user_trans")
    p_att = {k: lin(feats["attr/" + k], "item_trans") for k in cfg.keys}          # ONE shared item_trans

    img_u = spmm(a_ui, p_img); img_i = spmm(a_iu, img_u)                          # Models.py:153-154
    txt_u = spmm(a_ui, p_txt); txt_i = spmm(a_iu, txt_u)                          # Models.py:156-157
    att_u, att_i = {}, {}
    for k in cfg.keys:                                                            # Models.py:161-163
        att_u[k] = spmm(a_ui, p_att[k])
        att_i[k] = spmm(a_iu, att_u[k])
    prof_i = spmm(a_iu, p_usr)                                                    # Models.py:166 (items first)
    prof_u = spmm(a_ui, prof_i)                                                   # Models.py:167

    u = params["user_id_embedding.weight"]
    i = params["item_id_embedding.weight"]
    us, is_ = [u], [i]
    for l in range(cfg.n_layers):                                                 # Models.py:174-183
        u = spmm(a_ui, i)
        if l == cfg.n_layers - 1:
            u = torch.softmax(u, dim=-1)
        i = spmm(a_iu, u)                                                         # uses the NEW u
        if l == cfg.n_layers - 1:
            i = torch.softmax(i, dim=-1)
        us.append(u); is_.append(i)
    e_u = torch.mean(torch.stack(us), dim=0)                                      # Models.py:185-186
    e_i = torch.mean(torch.stack(is_), dim=0)

    n = lambda x: F.normalize(x, p=2, dim=1)                                      # Models.py:188-197
    e_u = e_u + cfg.model_cat_rate * n(img_u) + cfg.model_cat_rate * n(txt_u)
    e_i = e_i + cfg.model_cat_rate * n(img_i) + cfg.model_cat_rate * n(txt_i)
    e_u = e_u + cfg.user_cat_rate * n(prof_u)
    e_i = e_i + cfg.user_cat_rate * n(prof_i)
    for k in cfg.keys:
        e_u = e_u + cfg.item_cat_rate * n(att_u[k])
        e_i = e_i + cfg.item_cat_rate * n(att_i[k])
    return {"E_u": e_u, "E_i": e_i, "img_i": img_i, "txt_i": txt_i, "img_u": img_u, "txt_u": txt_u,
            "P_usr": p_usr, "prof_u": prof_u, "prof_i": prof_i, "att_u": att_u, "att_i": att_i}


# --------------------------------------------------------------------------------------------
# R7 - BPR + prune: reference main.py:330-342, :158-165
# --------------------------------------------------------------------------------------------
def prune_loss(pred: torch.Tensor, drop_rate: float) -> torch.Tensor:
    """Mean of the int((1-drop)*B) SMALLEST entries of pred (main.py:158-165).

    The reference sorts with numpy's default (unstable) argsort; ties at the cut are therefore
    unspecified there. Here ties are broken by ascending sample index (stable)."""
    order = torch.argsort(pred.detach(), stable=True)
    keep = int((1 - drop_rate) * len(pred))
    return pred[order[:keep]].mean()


def bpr_loss(users, pos, neg, cfg: Config):
    pos_s = torch.sum(users * pos, dim=1)
    neg_s = torch.sum(users * neg, dim=1)
    reg = 1. / (2 * (users ** 2).sum() + 1e-8) + 1. / (2 * (pos ** 2).sum() + 1e-8) + 1. / (2 * (neg ** 2).sum() + 1e-8)
    reg = reg / cfg.batch_size                                                    # the FLAG value (main.py:335)
    maxi = F.logsigmoid(pos_s - neg_s + 1e-8)
    mf = -prune_loss(maxi, cfg.prune_loss_drop_rate)
    return mf, cfg.decay * reg


def feat_reg(img_i, txt_i, img_u, txt_u, n_items: int, cfg: Config):              # main.py:151-156
    r = 0.5 * (img_i ** 2).sum() + 0.5 * (txt_i ** 2).sum() + 0.5 * (img_u ** 2).sum() + 0.5 * (txt_u ** 2).sum()
    return cfg.feat_reg_decay * (r / n_items)


def step_loss(fw: dict, users, pos, neg, n_items: int, cfg: Config):
    """Loss assembly of reference main.py:232-273 (mask branch off). Returns (loss, parts)."""
    users = torch.as_tensor(users, dtype=torch.long)
    pos = torch.as_tensor(pos, dtype=torch.long)
    neg = torch.as_tensor(neg, dtype=torch.long)
    parts = []
    mf, emb = bpr_loss(fw["E_u"][users], fw["E_i"][pos], fw["E_i"][neg], cfg)
    parts.append((mf, emb))
    img_mf, e1 = bpr_loss(fw["img_u"][users], fw["img_i"][pos], fw["img_i"][neg], cfg)
    txt_mf, e2 = bpr_loss(fw["txt_u"][users], fw["txt_i"][pos], fw["txt_i"][neg], cfg)
    parts += [(img_mf, e1), (txt_mf, e2)]
    aug = 0
    for k in cfg.keys:                                                            # main.py:249-254
        a_mf, e = bpr_loss(fw["prof_u"][users], fw["att_i"][k][pos], fw["att_i"][k][neg], cfg)
        parts.append((a_mf, e))
        aug = aug + a_mf
    fr = feat_reg(fw["img_i"], fw["txt_i"], fw["img_u"], fw["txt_u"], n_items, cfg)
    loss = mf + emb + fr + cfg.aug_mf_rate * aug + cfg.mm_mf_rate * (img_mf + txt_mf)   # main.py:273
    return loss, {"bpr": parts, "feat_reg": fr}


# --------------------------------------------------------------------------------------------
# R8 - optimiser: torch.optim.AdamW defaults as constructed at reference main.py:100-104
# --------------------------------------------------------------------------------------------
class AdamW:
    """Plain decoupled-weight-decay Adam, lr from the flag, betas (0.9, 0.999), eps 1e-8,
    weight_decay 0.01 (the torch default; the --weight_decay flag is unused in the reference)."""

    def __init__(self, params: Dict[str, torch.Tensor], lr: float, wd: float = 0.01,
                 b1: float = 0.9, b2: float = 0.999, eps: float = 1e-8):
        self.params, self.lr, self.wd, self.b1, self.b2, self.eps = params, lr, wd, b1, b2, eps
        self.t = 0
        self.m = {k: torch.zeros_like(v) for k, v in params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in params.items()}

    @torch.no_grad()
    def step(self, grads: Dict[str, torch.Tensor]):
        self.t += 1
        bc1 = 1 - self.b1 ** self.t
        bc2 = 1 - self.b2 ** self.t
        for k, p in self.params.items():
            g = grads.get(k)
            if g is None:
                continue
            p.mul_(1 - self.lr * self.wd)
            self.m[k].mul_(self.b1).add_(g, alpha=1 - self.b1)
            self.v[k].mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            denom = (self.v[k].sqrt() / math.sqrt(bc2)).add_(self.eps)
            p.addcdiv_(self.m[k], denom, value=-self.lr / bc1)


# --------------------------------------------------------------------------------------------
# R9/R10 - ranking + metrics: reference utility/batch_test.py:21-36,70-109 ; utility/metrics.py
# --------------------------------------------------------------------------------------------
def rank_topk(scores: np.ndarray, train_items: Sequence[int], k: int) -> List[int]:
    """Top-k item ids among all items minus train_items, by score descending; ties keep the
    ascending-item-id order (heapq.nlargest over the ascending candidate list is stable)."""
    banned = set(train_items)
    cand = [i for i in range(scores.shape[0]) if i not in banned]                 # batch_test.py:100-102
    return heapq.nlargest(k, cand, key=lambda i: scores[i])                       # batch_test.py:27


def rank_topk_np(scores: np.ndarray, train_items: Sequence[int], k: int) -> np.ndarray:
    """Vectorised equivalent of rank_topk (lexsort on (-score, id)); used at larger sizes."""
    s = np.array(scores, dtype=np.float64, copy=True)
    mask = np.zeros(s.shape[0], dtype=bool)
    if len(train_items):
        mask[np.asarray(train_items, dtype=np.int64)] = True
    ids = np.flatnonzero(~mask)
    order = np.lexsort((ids, -s[ids]))[:k]
    return ids[order]


def metrics_from_hits(r: Sequence[int], n_pos: int, Ks: Sequence[int]) -> dict:
    """precision/recall/ndcg/hit at each K from the hit vector of the top-max(Ks) list.
    metrics.py:8-18 (precision), :74-79 (recall), :43-71 (dcg/ndcg: IDCG from the retrieved
    hit vector itself, sorted descending), :82-87 (hit)."""
    r = np.asarray(r, dtype=np.float64)
    out = {"precision": [], "recall": [], "ndcg": [], "hit_ratio": []}
    for K in Ks:
        rk = r[:K]
        out["precision"].append(np.mean(np.asarray(r)[:K]) if K >= 1 else 0.0)
        out["recall"].append(0 if n_pos == 0 else np.sum(rk) / n_pos)
        disc = np.log2(np.arange(2, rk.size + 2))
        dcg = np.sum(rk / disc) if rk.size else 0.0
        ideal = np.asarray(sorted(r.tolist(), reverse=True), dtype=np.float64)[:K]
        idcg = np.sum(ideal / np.log2(np.arange(2, ideal.size + 2))) if ideal.size else 0.0
        out["ndcg"].append(dcg / idcg if idcg else 0.0)
        out["hit_ratio"].append(1.0 if np.sum(rk) > 0 else 0.0)
    return {k: np.asarray(v, dtype=np.float64) for k, v in out.items()}


def evaluate(e_u: np.ndarray, e_i: np.ndarray, users_to_test: Sequence[int], train_items: dict,
             test_set: dict, Ks: Sequence[int], batch_size: int = 1024, scores_fn=None):
    """test_torch restated (batch_test.py:112-169): blocks of 2*batch_size users, fp32 GEMM,
    per-user masked top-max(Ks), metrics averaged over len(users_to_test).
    Returns (result dict, topk lists [n_users_to_test][<=Kmax])."""
    kmax = max(Ks)
    n = len(users_to_test)
    res = {k: np.zeros(len(Ks)) for k in ("precision", "recall", "ndcg", "hit_ratio")}
    lists = []
    ub = 2 * batch_size
    eu_t, ei_t = torch.from_numpy(np.ascontiguousarray(e_u)), torch.from_numpy(np.ascontiguousarray(e_i))
    for s in range(0, n // ub + 1):
        blk = users_to_test[s * ub:(s + 1) * ub]
        if len(blk) == 0:
            continue
        if scores_fn is None:
            S = torch.matmul(eu_t[torch.as_tensor(blk, dtype=torch.long)], ei_t.t()).numpy()
        else:
            S = scores_fn(blk)
        for row, u in enumerate(blk):
            top = rank_topk_np(S[row], train_items.get(u, []), kmax)
            lists.append(top)
            pos = set(test_set[u])
            r = [1 if int(i) in pos else 0 for i in top]
            m = metrics_from_hits(r, len(test_set[u]), Ks)
            for k in res:
                res[k] += m[k] / n
    res["auc"] = 0.0
    return res, lists


def scores_fma_chain(e_u: np.ndarray, e_i: np.ndarray, order: str = "natural") -> np.ndarray:
    """S[u, i] as a fp32 fma chain over k, the exact arithmetic of gfx950's v_mfma_f32_*
    (cdna_hip_programming.md section 3: 'bit-for-bit a k-ordered f32 fmaf chain').
    order="mfma16x16x4" visits k as llmrec_amd/csrc/topk.hip documents it
    (for c, for s, for q: k = 16c + 4q + s). Each step is computed in float64 (the product of two
    fp32 numbers is exact there) and rounded once to fp32; used at small sizes to state the GPU
    scoring kernel's results bit for bit."""
    eu = np.asarray(e_u, dtype=np.float32)
    ei = np.asarray(e_i, dtype=np.float32)
    d = eu.shape[1]
    if order == "natural":
        ks = list(range(d))
    elif order == "mfma16x16x4":
        ks = [16 * c + 4 * q + s for c in range((d + 15) // 16) for s in range(4) for q in range(4)]
        ks = [k for k in ks if k < d]
    else:
        raise ValueError(order)
    acc = np.zeros((eu.shape[0], ei.shape[0]), dtype=np.float32)
    for k in ks:
        prod = eu[:, k:k + 1].astype(np.float64) * ei[:, k].astype(np.float64)[None, :]
        acc = (prod + acc.astype(np.float64)).astype(np.float32)
    return acc


# --------------------------------------------------------------------------------------------
# R11 - host sampler: reference utility/load_data.py:157-195 and main.py:216-224
# --------------------------------------------------------------------------------------------
def sample_batch(exist_users, train_items: dict, n_items: int, n_users: int, batch_size: int,
                 rd=_pyrandom, nprng=np.random):
    if batch_size <= n_users:
        users = rd.sample(exist_users, batch_size)
    else:
        users = [rd.choice(exist_users) for _ in range(batch_size)]
    pos, neg = [], []
    for u in users:
        items = train_items[u]
        pos.append(items[nprng.randint(low=0, high=len(items), size=1)[0]])
        while True:
            j = nprng.randint(low=0, high=n_items, size=1)[0]
            if j not in items:
                neg.append(j)
                break
    return users, pos, neg


def augment_batch(users, pos, neg, aug_dict: dict, n_items: int, rate: float, rd=_pyrandom):
    ua = rd.sample(users, int(len(users) * rate))
    ok = [u for u in ua if aug_dict[u][0] < n_items and aug_dict[u][1] < n_items]
    return users + ok, pos + [aug_dict[u][0] for u in ok], neg + [aug_dict[u][1] for u in ok]


# --------------------------------------------------------------------------------------------
# dataset loading for the oracle (reference utility/load_data.py:10-92, main.py:54-79)
# --------------------------------------------------------------------------------------------
@dataclass
class OracleData:
    n_users: int
    n_items: int
    train_items: dict
    test_set: dict
    val_set: dict
    train_mat: object
    feats: Dict[str, torch.Tensor]
    aug_dict: dict
    keys: Sequence[str] = field(default_factory=tuple)
    n_train: int = 0


def load_dataset(path: str, keys: Sequence[str], dtype=torch.float32) -> OracleData:
    import json
    import pickle

    tr = json.load(open(path + "/train.json"))
    te = json.load(open(path + "/test.json"))
    va = json.load(open(path + "/val.json"))
    train_items = {int(u): v for u, v in tr.items() if len(v)}
    test_set = {int(u): v for u, v in te.items() if len(v)}
    val_set = {int(u): v for u, v in va.items() if len(v)}
    text = np.load(path + "/text_feat.npy")
    image = np.load(path + "/image_feat.npy")
    train_mat = pickle.load(open(path + "/train_mat", "rb"))
    ue = pickle.load(open(path + "/augmented_user_init_embedding", "rb"))
    user = np.array([ue[i] for i in range(len(ue))])
    ad = pickle.load(open(path + "/augmented_atttribute_embedding_dict", "rb"))
    feats = {"image": torch.tensor(image).to(dtype), "text": torch.tensor(text).to(dtype),
             "user": torch.tensor(user).to(dtype)}
    for k in keys:
        feats["attr/" + k] = torch.tensor(np.array([ad[k][i] for i in range(len(ad[k]))])).to(dtype)
    aug = pickle.load(open(path + "/augmented_sample_dict", "rb"))
    return OracleData(n_users=train_mat.shape[0], n_items=train_mat.shape[1], train_items=train_items,
                      test_set=test_set, val_set=val_set, train_mat=train_mat, feats=feats, aug_dict=aug,
                      keys=tuple(keys), n_train=sum(len(v) for v in train_items.values()))
