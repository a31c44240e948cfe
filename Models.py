"""MM_Model / Decoder with the reference's constructor and forward signatures
(reference Models.py:19-225), computed by the gfx950 HIP kernels behind llmrec_amd.ops.

forward() returns the reference's 14-tuple in the same order (Models.py:199). Parameters have the
reference's names, shapes and (same seed) the same initial values, because the modules are
created and initialised in the same order on the CPU generator before moving to the GPU.
"""
import numpy as np
import torch
import torch.nn as nn

from llmrec_amd import ops
from utility.parser import parse_args

args = parse_args()

device = torch.device("cuda" if torch.cuda.is_available() else "cpu")


def _as_feature(x):
    """Constant side-feature matrix: fp32, contiguous, on the compute device."""
    if isinstance(x, torch.Tensor):
        return x.detach().float().contiguous().to(device)
    return torch.as_tensor(np.asarray(x)).float().contiguous().to(device)


class MM_Model(nn.Module):
    def __init__(self, n_users, n_items, embedding_dim, weight_size, dropout_list, image_feats, text_feats,
                 user_init_embedding, item_attribute_dict):
        super().__init__()
        self.n_users, self.n_items = n_users, n_items
        self.embedding_dim = embedding_dim
        self.n_ui_layers = len(weight_size)                 # the real propagation depth
        self.weight_size = [embedding_dim] + list(weight_size)

        d = args.embed_size
        # creation order == the reference's, so default inits draw the same RNG values
        self.image_trans = nn.Linear(image_feats.shape[1], d)
        self.text_trans = nn.Linear(text_feats.shape[1], d)
        self.user_trans = nn.Linear(user_init_embedding.shape[1], d)
        self.item_trans = nn.Linear(item_attribute_dict['title'].shape[1], d)   # shared by every attribute key
        for lin in (self.image_trans, self.text_trans, self.user_trans, self.item_trans):
            nn.init.xavier_uniform_(lin.weight)
        self.user_id_embedding = nn.Embedding(n_users, embedding_dim)
        self.item_id_embedding = nn.Embedding(n_items, embedding_dim)
        nn.init.xavier_uniform_(self.user_id_embedding.weight)
        nn.init.xavier_uniform_(self.item_id_embedding.weight)

        self.image_feats = _as_feature(image_feats)
        self.text_feats = _as_feature(text_feats)
        self.user_feats = _as_feature(user_init_embedding)
        self.item_feats = {key: _as_feature(val) for key, val in item_attribute_dict.items()}

        self.softmax = nn.Softmax(dim=-1)
        self.act = nn.Sigmoid()
        self.sigmoid = nn.Sigmoid()
        self.dropout = nn.Dropout(p=args.drop_rate)
        self.batch_norm = nn.BatchNorm1d(d)                 # never used in forward; kept for state_dict parity
        self.tau = 0.5

    def mm(self, x, y):
        """Sparse x dense product; x: torch sparse COO tensor or ops.SparseOperand."""
        return ops.spmm(x, y)

    def _drop(self, x):
        return self.dropout(x) if (self.training and args.drop_rate > 0) else x

    def _mask_features(self):
        """Feature masking (reference Models.py:131-142); a no-op at the default mask_rate 0.
        The reference draws torch.randperm(n_users) every forward even then; nothing on the
        path consumes the CPU torch generator afterwards, so the draw is skipped here."""
        i_mask_nodes, u_mask_nodes = None, None
        if args.mask:
            i_perm = torch.randperm(self.n_items)
            i_mask_nodes = i_perm[: int(args.mask_rate * self.n_items)]
            for key in self.item_feats:
                self.item_feats[key][i_mask_nodes] = self.item_feats[key].mean(0)
        n_mask_u = int(args.mask_rate * self.n_users)
        if n_mask_u > 0:
            u_mask_nodes = torch.randperm(self.n_users)[:n_mask_u]
            self.user_feats[u_mask_nodes] = self.user_feats.mean(0)
        else:
            u_mask_nodes = torch.empty(0, dtype=torch.long)
        return i_mask_nodes, u_mask_nodes

    def forward(self, ui_graph, iu_graph, image_ui_graph=None, image_iu_graph=None, text_ui_graph=None, text_iu_graph=None):
        i_mask_nodes, u_mask_nodes = self._mask_features()
        if isinstance(ui_graph, torch.Tensor):
            ui_graph = ops.operand_from_sparse_tensor(ui_graph)
        if isinstance(iu_graph, torch.Tensor):
            iu_graph = ops.operand_from_sparse_tensor(iu_graph)

        # R4: projections (fp32 MFMA). One shared item_trans for all attribute keys.
        image_feats = self._drop(ops.linear(self.image_feats, self.image_trans.weight, self.image_trans.bias))
        text_feats = self._drop(ops.linear(self.text_feats, self.text_trans.weight, self.text_trans.bias))
        user_feats = self._drop(ops.linear(self.user_feats, self.user_trans.weight, self.user_trans.bias))
        keys = list(self.item_feats.keys())
        proj = ops.linear_multi(self.item_trans.weight, self.item_trans.bias, [self.item_feats[k] for k in keys])
        item_feats = {k: self._drop(p) for k, p in zip(keys, proj)}

        # R5: side-feature propagation. The reference repeats this block args.layers times with
        # identical inputs (Models.py:152-157); the result does not depend on the repeat count.
        image_user_feats = ops.spmm(ui_graph, image_feats)
        image_item_feats = ops.spmm(iu_graph, image_user_feats)
        text_user_feats = ops.spmm(ui_graph, text_feats)
        text_item_feats = ops.spmm(iu_graph, text_user_feats)
        user_feat_from_item = {}
        for key in keys:
            user_feat_from_item[key] = ops.spmm(ui_graph, item_feats[key])
            item_feats[key] = ops.spmm(iu_graph, user_feat_from_item[key])
        item_prof_feat = ops.spmm(iu_graph, user_feats)            # profile stream goes items first
        user_prof_feat = ops.spmm(ui_graph, item_prof_feat)

        # R3: ID-embedding chain; items of layer l+1 use the NEW users; softmax over d on the last layer
        u_g, i_g = self.user_id_embedding.weight, self.item_id_embedding.weight
        user_emb_list, item_emb_list = [u_g], [i_g]
        for layer in range(self.n_ui_layers):
            last = layer == self.n_ui_layers - 1
            u_g = ops.spmm(ui_graph, i_g)
            if last:
                u_g = ops.softmax_rows(u_g)
            i_g = ops.spmm(iu_graph, u_g)
            if last:
                i_g = ops.softmax_rows(i_g)
            user_emb_list.append(u_g)
            item_emb_list.append(i_g)

        # R6: layer mean + normalise-and-add fusion, one kernel per side
        rates = [args.model_cat_rate, args.model_cat_rate, args.user_cat_rate] + [args.item_cat_rate] * len(keys)
        u_g_embeddings = ops.fuse(user_emb_list, [image_user_feats, text_user_feats, user_prof_feat]
                                  + [user_feat_from_item[k] for k in keys], rates)
        i_g_embeddings = ops.fuse(item_emb_list, [image_item_feats, text_item_feats, item_prof_feat]
                                  + [item_feats[k] for k in keys], rates)

        return (u_g_embeddings, i_g_embeddings, image_item_feats, text_item_feats, image_user_feats, text_user_feats,
                user_feats, item_feats, user_prof_feat, item_prof_feat, user_feat_from_item, item_feats,
                i_mask_nodes, u_mask_nodes)


class Decoder(nn.Module):
    """Attribute-restoration decoder (reference Models.py:203-225); only used with --mask."""

    def __init__(self, feat_size):
        super().__init__()
        self.feat_size = feat_size
        self.u_net = nn.Sequential(nn.Linear(args.embed_size, int(feat_size)), nn.LeakyReLU(True))
        self.i_net = nn.Sequential(nn.Linear(args.embed_size, int(feat_size)), nn.LeakyReLU(True))

    def forward(self, u, i):
        u_output = self.u_net(u.float())
        i_output = self.i_net(torch.stack([i[key] for key in i.keys()]).float())
        return u_output, i_output
