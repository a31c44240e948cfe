"""One training step / one evaluation of the Stage-2 path, shared by main.Trainer and bench.py.

Mirrors the loop body of the reference's Trainer.train (main.py:228-283) and Trainer.test
(main.py:182-187) on top of llmrec_amd.ops; the drop-in modules bind their argparse namespace to
the ``Hyper`` record below.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch

from . import ops


@dataclass
class Hyper:
    """The flags the step reads (reference utility/parser.py defaults)."""
    batch_size: int = 1024                  # the FLAG value: divisor of the BPR regulariser
    decay: float = 1e-5                     # eval(--regs)[0]
    feat_reg_decay: float = 1e-5
    aug_mf_rate: float = 0.012
    mm_mf_rate: float = 0.0001
    prune_loss_drop_rate: float = 0.71
    aug_sample_rate: float = 0.1

    @staticmethod
    def from_args(args) -> "Hyper":
        return Hyper(batch_size=args.batch_size, decay=eval(args.regs)[0], feat_reg_decay=args.feat_reg_decay,
                     aug_mf_rate=args.aug_mf_rate, mm_mf_rate=args.mm_mf_rate,
                     prune_loss_drop_rate=args.prune_loss_drop_rate, aug_sample_rate=args.aug_sample_rate)


def bpr(hp: Hyper, user_table, item_table, users, pos, neg, n_valid=None):
    """(mf_loss, emb_loss) of reference Trainer.bpr_loss on table rows (main.py:330-342)."""
    out = ops.bpr_prune(user_table, item_table, users, pos, neg, hp.prune_loss_drop_rate, hp.decay, hp.batch_size, n_valid)
    return out[0], out[1]


def step_loss(hp: Hyper, fw, users, pos, neg, n_items: int, n_valid=None, on_bpr=None):
    """Loss assembly of reference main.py:232-273 (mask branch handled by the caller).
    ``fw`` is the model's 14-tuple. Returns (batch_loss, mf_loss, emb_loss)."""
    (e_u, e_i, img_i, txt_i, img_u, txt_u, _p_usr, _att_i_dup, prof_u, _prof_i, _att_u, att_i, _im, _um) = fw
    record = on_bpr or (lambda *a: None)
    mf, emb = bpr(hp, e_u, e_i, users, pos, neg, n_valid)
    record(mf, emb)
    img_mf, e1 = bpr(hp, img_u, img_i, users, pos, neg, n_valid)
    record(img_mf, e1)
    txt_mf, e2 = bpr(hp, txt_u, txt_i, users, pos, neg, n_valid)
    record(txt_mf, e2)
    aug_mf = 0
    for key in att_i:                                       # user side is the LLM user profile (main.py:250)
        a_mf, e = bpr(hp, prof_u, att_i[key], users, pos, neg, n_valid)
        record(a_mf, e)
        aug_mf = aug_mf + a_mf
    feat_reg = ops.sumsq(hp.feat_reg_decay * 0.5 / n_items, [img_i, txt_i, img_u, txt_u])[0]    # main.py:151-156
    loss = mf + emb + feat_reg + hp.aug_mf_rate * aug_mf + hp.mm_mf_rate * (img_mf + txt_mf)
    return loss, mf, emb


def train_step(model, optimizer, ui, iu, users, pos, neg, hp: Hyper, n_valid=None, on_bpr=None, extra_loss=None):
    """forward -> losses -> backward -> AdamW. Returns detached device scalars (loss, mf, emb)."""
    model.train()
    fw = model(ui, iu, ui, iu, ui, iu)
    loss, mf, emb = step_loss(hp, fw, users, pos, neg, model.n_items, n_valid, on_bpr)
    if extra_loss is not None:
        loss = loss + extra_loss(fw)
    optimizer.zero_grad()
    loss.backward()
    optimizer.step()
    return loss.detach(), mf.detach(), emb.detach()


class DeviceBatcher:
    """On-device mini-batch construction in ONE launch (llmrec_sample_batch): BPR triples from the HIP
    sampler plus the LLM-augmented triples of reference main.py:216-224, no host synchronisation. The
    number of valid triples and the step counter stay on the device, so a captured graph can contain
    the sampler and a training step is a single graph replay.

    aug_pos / aug_neg: int64 [n_users] device arrays (the augmented_sample_dict columns); a pair
    is used only if both ids are < n_items, as in the reference.
    rank/world: batch-sharded replicas (llmrec_amd.dp) - the sampler draws the GLOBAL batch of
    world * batch_size users (one keyed permutation, so still without replacement) and this rank
    keeps its slice; augmented triples are drawn from the slice."""

    def __init__(self, train: ops.Csr, exist_users: torch.Tensor, n_items: int, batch_size: int,
                 aug_pos: Optional[torch.Tensor], aug_neg: Optional[torch.Tensor], aug_rate: float, seed: int,
                 rank: int = 0, world: int = 1):
        self.train, self.exist_users, self.n_items, self.B = train, exist_users, n_items, batch_size
        self.rank, self.world = rank, world
        self.aug_pos, self.aug_neg = aug_pos, aug_neg
        self.n_aug = int(batch_size * aug_rate) if aug_pos is not None else 0
        self.seed = seed
        self.capacity = self.B + self.n_aug
        self.step_dev = torch.zeros(1, dtype=torch.int64, device=exist_users.device)

    def fill(self, users, pos, neg, n_valid):
        """Sample the next step's batch into the given buffers (capacity entries each); advances the device counter."""
        ops.sample_batch(self.seed, self.step_dev, self.exist_users, self.n_items, self.train, self.B * self.world,
                         self.rank * self.B, self.B, self.n_aug, self.aug_pos, self.aug_neg, users, pos, neg, n_valid)

    def next(self, step: Optional[int] = None):
        """(users, pos, neg, n_valid) in fresh buffers; step = None continues the device counter."""
        dev = self.exist_users.device
        if step is not None:
            self.step_dev.fill_(int(step))
        u, p, n = (torch.empty(self.capacity, dtype=torch.int64, device=dev) for _ in range(3))
        nv = torch.empty(1, dtype=torch.int32, device=dev)
        self.fill(u, p, n, nv)
        return u, p, n, nv


@torch.no_grad()
def evaluate_topk(model, ui, iu, query_users: torch.Tensor, train: ops.Csr, k: int):
    """Reference Trainer.test up to the ranked lists: eval-mode forward + masked top-k."""
    model.eval()
    fw = model(ui, iu, ui, iu, ui, iu)
    return ops.score_topk(fw[0], fw[1], query_users, train, k)
