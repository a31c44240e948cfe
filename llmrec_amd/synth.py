"""Synthetic bipartite-graph and dataset generators.

The real Netflix / MovieLens files are not distributable (reference README.md:166 is a
Google-Drive link), so every workload in this repo is generated here from a seed:

* ``bipartite_edges``      - numpy, exact edge count, no duplicate (u, i) pairs, power-law user
                             degrees and item popularity ~ rank^-0.8.
* ``bipartite_edges_device`` - torch-on-device variant for the 200 M / 1 B edge configs (never
                             materialises the COO on the host).
* ``write_dataset``        - writes every on-disk file Stage 2 consumes (SURVEY.md Appendix A;
                             reference main.py:54-79,216, utility/load_data.py:15-28).
* ``NF_SHAPE`` / ``ML_SHAPE`` - the Netflix- and MovieLens-shaped size presets.
"""
from __future__ import annotations

import json
import os
import pickle
from dataclasses import dataclass

import numpy as np

NETFLIX_KEYS = ("year", "title", "director", "country", "language")       # reference main.py:72
MOVIELENS_KEYS = ("title", "genre", "director", "country", "language")    # reference main.py:70
DATASET_KEYS = {
    "netflix_valid_item": NETFLIX_KEYS,
    "preprocessed_raw_MovieLens": MOVIELENS_KEYS,
}


@dataclass(frozen=True)
class Shape:
    n_users: int
    n_items: int
    n_train: int
    image_dim: int = 512
    text_dim: int = 768
    llm_dim: int = 1536


NF_SHAPE = Shape(n_users=13187, n_items=17366, n_train=55146)   # 80 % of 68933 (image/datasets.png)
ML_SHAPE = Shape(n_users=12495, n_items=10322, n_train=46368)   # 80 % of 57960


def _user_degrees(rng: np.random.Generator, n_users: int, n_edges: int, max_deg: int, min_deg: int = 1) -> np.ndarray:
    """Power-law degrees clipped to [min_deg, max_deg], rescaled so they sum to exactly n_edges."""
    max_deg = max(1, min(max_deg, n_edges))
    raw = rng.zipf(1.8, size=n_users).astype(np.float64)
    raw = np.clip(raw, 1, max_deg)
    deg = np.maximum(min_deg, np.floor(raw * (n_edges / raw.sum()))).astype(np.int64)
    deg = np.minimum(deg, max_deg)
    # fix the remainder one edge at a time on random users (keeps 1 <= deg <= max_deg)
    diff = int(n_edges - deg.sum())
    guard = 0
    while diff != 0 and guard < (64 if min_deg == 1 else 4096):
        guard += 1
        if diff > 0:
            cand = np.flatnonzero(deg < max_deg)
            take = rng.choice(cand, size=min(diff, cand.size), replace=False)
            deg[take] += 1
        else:
            cand = np.flatnonzero(deg > min_deg)
            take = rng.choice(cand, size=min(-diff, cand.size), replace=False)
            deg[take] -= 1
        diff = int(n_edges - deg.sum())
    if diff != 0:
        raise ValueError("cannot realise %d edges over %d users" % (n_edges, n_users))
    return deg


def bipartite_edges(n_users: int, n_items: int, n_edges: int, seed: int = 0,
                    max_deg: int = 10_000, item_alpha: float = 0.8, n_communities: int = 0, p_in: float = 0.85,
                    min_deg: int = 1):
    """Return (rows, cols) int64 arrays, sorted by (row, col), without duplicate pairs.

    n_communities > 0 plants a block structure a collaborative-filtering model can learn (the long-horizon accuracy fixtures,
    oracle/make_trajectory.py): user u belongs to block u % C, the item of popularity rank r to block r % C; a fraction p_in of
    every user's draws is moved to the nearest rank of the user's own block (same popularity law). 0 (default): popularity only,
    the generator of every other fixture and workload, unchanged. min_deg = 3 gives every user a validation and a test item
    (split_train_test), as in the survey's probe of the reference (13 187 test users)."""
    rng = np.random.default_rng(seed)
    max_deg = min(max_deg, n_items)
    deg = _user_degrees(rng, n_users, n_edges, max_deg, min_deg)
    pop = np.arange(1, n_items + 1, dtype=np.float64) ** (-item_alpha)
    cdf = np.cumsum(pop / pop.sum())
    item_perm = rng.permutation(n_items)            # popularity rank -> item id
    rows = np.repeat(np.arange(n_users, dtype=np.int64), deg)
    ranks = np.minimum(np.searchsorted(cdf, rng.random(rows.size)), n_items - 1)
    if n_communities > 0:
        C = int(n_communities)
        own = (ranks // C) * C + rows % C
        inside = (rng.random(rows.size) < p_in) & (own < n_items)
        ranks = np.where(inside, own, ranks)
    cols = item_perm[ranks]
    # de-duplicate (u, i) and top up with fresh draws until exact
    for _ in range(200):
        key = rows * n_items + cols
        order = np.argsort(key, kind="stable")
        key_s = key[order]
        dup = np.zeros(key.size, dtype=bool)
        dup[order[1:]] = key_s[1:] == key_s[:-1]
        n_dup = int(dup.sum())
        if n_dup == 0:
            break
        # dense users: redraw uniformly among items (guarantees progress for high-degree rows)
        cols[dup] = rng.integers(0, n_items, size=n_dup)
    else:  # pragma: no cover
        raise RuntimeError("could not de-duplicate edges")
    order = np.lexsort((cols, rows))
    return rows[order], cols[order].astype(np.int64)


def bipartite_edges_device(n_users: int, n_items: int, n_edges: int, seed: int, device,
                           item_alpha: float = 0.8, max_deg: int = 10_000):
    """Device generator for the large synthetic configs (cfg 4/5).

    Returns (rows, cols) int64 device tensors, sorted by (row, col), with EXACTLY ``n_edges`` distinct pairs (SURVEY.md
    8(d): "no duplicate (u, i); E exact"): the first draw loses a few per cent to duplicates (hub users x popular items);
    the deficit is topped up with fresh draws - users taken in proportion to their degree (a random existing edge's user),
    items from the same popularity law - until the count is exact."""
    import torch

    g = torch.Generator(device=device)
    g.manual_seed(seed)
    if n_edges > n_users * n_items:
        raise ValueError("bipartite_edges_device: %d distinct pairs do not fit %d x %d" % (n_edges, n_users, n_items))
    # degrees: Pareto-ish via inverse CDF, clipped, scaled to the target sum
    u = torch.rand(n_users, generator=g, device=device, dtype=torch.float64)
    raw = torch.clamp((1.0 - u) ** (-1.0 / 0.8), 1.0, float(min(max_deg, n_items)))
    deg = torch.clamp((raw * (n_edges / raw.sum())).floor(), min=1.0).to(torch.int64)
    rows = torch.repeat_interleave(torch.arange(n_users, device=device, dtype=torch.int64), deg)
    del u, raw, deg
    # spread ranks over ids with a multiplicative hash so hot items are not adjacent rows
    mult = 2654435761 % n_items
    while np.gcd(mult, n_items) != 1:
        mult += 1
    a = 1.0 - item_alpha

    def draw_items(n):
        # popularity ~ rank^-alpha by inverse CDF of the continuous approximation
        r = torch.rand(n, generator=g, device=device, dtype=torch.float64)
        rank = ((r * ((n_items + 1.0) ** a - 1.0) + 1.0) ** (1.0 / a) - 1.0).floor().to(torch.int64)
        rank.clamp_(0, n_items - 1)
        return (rank * mult) % n_items

    key = torch.unique(rows * n_items + draw_items(rows.numel()))       # sorted, distinct
    del rows
    for _ in range(64):
        need = n_edges - key.numel()
        if need <= 0:
            break
        m = int(need * 1.25) + 1024
        pick = torch.randint(0, key.numel(), (m,), generator=g, device=device)
        new = torch.unique(torch.div(key[pick], n_items, rounding_mode="floor") * n_items + draw_items(m))
        pos = torch.searchsorted(key, new).clamp_(max=key.numel() - 1)
        new = new[key[pos] != new]                                       # not present yet
        if new.numel() > need:                                           # an unbiased subset, not the lowest keys
            new = new[torch.randperm(new.numel(), generator=g, device=device)[:need]]
        key = torch.sort(torch.cat([key, new])).values
        del pick, new, pos
    if key.numel() > n_edges:                                            # the first draw can overshoot by the floor()s' slack: drop at random
        keep = torch.randperm(key.numel(), generator=g, device=device)[:n_edges]
        key = key[torch.sort(keep).values]
    if key.numel() != n_edges:
        raise RuntimeError("bipartite_edges_device: %d of %d edges after the top-up rounds" % (key.numel(), n_edges))
    rows = torch.div(key, n_items, rounding_mode="floor")
    cols = key - rows * n_items
    return rows, cols


def split_train_test(rows: np.ndarray, cols: np.ndarray, n_users: int, seed: int = 0):
    """Hold out 1 val + 1 test item for users with >= 3 interactions (rest stay in train)."""
    rng = np.random.default_rng(seed + 1)
    order = np.lexsort((rng.random(rows.size), rows))
    rows, cols = rows[order], cols[order]
    start = np.searchsorted(rows, np.arange(n_users))
    end = np.searchsorted(rows, np.arange(n_users), side="right")
    deg = end - start
    role = np.zeros(rows.size, dtype=np.int8)        # 0 train, 1 val, 2 test
    big = deg >= 3
    role[start[big]] = 2
    role[start[big] + 1] = 1
    return rows, cols, role


def write_dataset(path: str, n_users: int, n_items: int, n_edges: int, seed: int = 0,
                  image_dim: int = 512, text_dim: int = 768, llm_dim: int = 1536,
                  keys=NETFLIX_KEYS, feat_dtype=np.float32, aug_out_of_range: float = 0.05,
                  max_deg: int = 10_000, n_communities: int = 0, min_deg: int = 1, attr_rows_as_arrays: bool = False) -> dict:
    """Write a complete Stage-2 dataset directory (every file of SURVEY.md Appendix A).

    ``n_edges`` counts all interactions; users with >= 3 give one to val and one to test."""
    import scipy.sparse as sp

    os.makedirs(path, exist_ok=True)
    rng = np.random.default_rng(seed + 7)
    rows, cols = bipartite_edges(n_users, n_items, n_edges, seed=seed, max_deg=max_deg, n_communities=n_communities, min_deg=min_deg)
    rows, cols, role = split_train_test(rows, cols, n_users, seed)

    def as_dict(mask):
        out = {}
        for u, i in zip(rows[mask].tolist(), cols[mask].tolist()):
            out.setdefault(str(u), []).append(i)
        return out

    train, val, test = as_dict(role == 0), as_dict(role == 1), as_dict(role == 2)
    for name, obj in (("train", train), ("val", val), ("test", test)):
        with open(os.path.join(path, name + ".json"), "w") as f:
            json.dump(obj, f)

    tr = role == 0
    train_mat = sp.csr_matrix((np.ones(int(tr.sum()), dtype=np.float32), (rows[tr], cols[tr])),
                              shape=(n_users, n_items))
    with open(os.path.join(path, "train_mat"), "wb") as f:
        pickle.dump(train_mat, f)

    np.save(os.path.join(path, "image_feat.npy"),
            rng.standard_normal((n_items, image_dim)).astype(feat_dtype))
    np.save(os.path.join(path, "text_feat.npy"),
            rng.standard_normal((n_items, text_dim)).astype(feat_dtype))

    user_emb = {u: rng.standard_normal(llm_dim).astype(np.float64) for u in range(n_users)}
    with open(os.path.join(path, "augmented_user_init_embedding"), "wb") as f:
        pickle.dump(user_emb, f)

    # the reference's file holds python lists (gpt_i_attribute_generate_aug.py:325-355); attr_rows_as_arrays keeps each row an
    # ndarray - the same values to both readers (np.array([d[k][i] ...]), main.py:73-77), without 133 M python floats at Netflix shape
    row = (lambda v: v) if attr_rows_as_arrays else (lambda v: v.tolist())
    attr = {k: {i: row(rng.standard_normal(llm_dim).astype(np.float64)) for i in range(n_items)}
            for k in keys}
    with open(os.path.join(path, "augmented_atttribute_embedding_dict"), "wb") as f:
        pickle.dump(attr, f)

    # augmented_sample_dict: user -> {0: pos, 1: neg}; a fraction points past n_items so the
    # reference's `< n_items` filter (main.py:218-220) is exercised.
    hi = int(n_items * (1.0 + aug_out_of_range)) + 1
    aug = {u: {0: int(rng.integers(0, hi)), 1: int(rng.integers(0, hi))} for u in range(n_users)}
    with open(os.path.join(path, "augmented_sample_dict"), "wb") as f:
        pickle.dump(aug, f)

    return {"n_users": n_users, "n_items": n_items, "n_train": int(tr.sum()),
            "n_test": int((role == 2).sum()), "n_val": int((role == 1).sum())}


DATASET_INPUT_FILES = ("train.json", "val.json", "test.json", "train_mat", "image_feat.npy", "text_feat.npy",
                       "augmented_user_init_embedding", "augmented_atttribute_embedding_dict", "augmented_sample_dict")


def dataset_digests(ds_dir: str):
    """sha256 of the array CONTENT of every input file (pickle byte streams are not stable across library versions)."""
    import hashlib
    out = {}
    for fn in DATASET_INPUT_FILES:
        p = os.path.join(ds_dir, fn)
        h = hashlib.sha256()
        if fn.endswith(".json"):
            d = json.load(open(p))
            for k in sorted(d, key=int):
                h.update(np.asarray([int(k)] + list(d[k]), dtype=np.int64).tobytes())
        elif fn.endswith(".npy"):
            h.update(np.ascontiguousarray(np.load(p)).tobytes())
        else:
            obj = pickle.load(open(p, "rb"))
            if fn == "train_mat":
                m = obj.tocsr(); m.sort_indices()
                h.update(m.indptr.astype(np.int64).tobytes()); h.update(m.indices.astype(np.int64).tobytes())
            elif fn == "augmented_user_init_embedding":
                h.update(np.asarray([obj[i] for i in range(len(obj))], dtype=np.float64).tobytes())
            elif fn == "augmented_atttribute_embedding_dict":
                for k in sorted(obj):
                    h.update(k.encode()); h.update(np.asarray([obj[k][i] for i in range(len(obj[k]))], dtype=np.float64).tobytes())
            else:
                h.update(np.asarray([[u, obj[u][0], obj[u][1]] for u in sorted(obj)], dtype=np.int64).tobytes())
        out[fn] = h.hexdigest()
    return out
