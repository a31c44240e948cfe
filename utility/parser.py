"""Command-line surface of Stage 2 - every flag of the reference with its default and type
(reference utility/parser.py:7-54), declared as one table so the drop-in contract is auditable.

``parse_args()`` keeps the reference's behaviour (parse ``sys.argv``; unknown flags are errors).
Several flags are dead in the reference too (weight_decay, sc, norm_type, mess_dropout, cf_model,
de_drop*, mf_emb_rate, title, point); they are accepted and ignored the same way."""
import argparse

# (flag, kind, default, help); kind: a type, "str?" (nargs='?'), or "flag" (store_true)
FLAGS = [
    ("data_path", "str?", "./data/", "Input data path"),
    ("seed", int, 2022, "Random seed"),
    ("dataset", "str?", "netflix", "Choose a dataset from {movieLens, netflix}"),
    ("verbose", int, 5, "Interval of evaluation."),
    ("epoch", int, 1000, "Number of epoch."),
    ("regs", "str?", "[1e-5,1e-5,1e-2]", "Regularizations."),
    ("embed_size", int, 64, "Embedding size."),
    ("weight_size", "str?", "[64, 64]", "Output sizes of every layer"),
    ("early_stopping_patience", int, 7, "Early Stop Patience"),
    ("mess_dropout", "str?", "[0.1, 0.1]", "Message dropout per layer (unused)"),
    ("sparse", int, 1, "Sparse or dense adjacency matrix"),
    ("debug", "flag", False, "Do not write ./logs/"),
    ("norm_type", "str?", "sym", "Adjacency matrix normalization operation (unused)"),
    ("gpu_id", int, 0, "GPU ID"),
    ("Ks", "str?", "[10, 20, 50]", "K value of ndcg/recall @ k"),
    ("test_flag", "str?", "part", "{part, full}: full also computes AUC"),
    ("sc", float, 1.0, "GCN self connection (unused)"),
    ("feat_reg_decay", float, 1e-5, "Feature Reg Decay"),
    ("title", str, "try_to_draw_line", ""),
    ("cf_model", "str?", "lightgcn", "Downstream CF model (unused)"),
    ("point", str, "", ""),
    # train
    ("batch_size", int, 1024, "Batch size."),
    ("lr", float, 0.0001, "Learning rate."),
    ("de_lr", float, 0.0002, "Decoder learning rate."),
    ("weight_decay", float, 1e-4, "Weight_decay (unused: AdamW's own default 0.01 applies)"),
    # model
    ("layers", int, 1, "Repeat count of the modal propagation loop"),
    ("drop_rate", float, 0.0, "Dropout rate"),
    ("mask_rate", float, 0.0, "Mask rate"),
    ("mask", bool, False, "If mask (argparse type=bool: any non-empty string is True)"),
    ("user_cat_rate", float, 2.8, "User cat rate"),
    ("item_cat_rate", float, 0.005, "Item cat rate"),
    ("model_cat_rate", float, 0.02, "Model cat rate"),
    ("de_drop1", float, 0.31, "(unused)"),
    ("de_drop2", float, 0.5, "(unused)"),
    # loss
    ("aug_mf_rate", float, 0.012, "Augmentation mf rate"),
    ("prune_loss_drop_rate", float, 0.71, "Prune loss drop rate"),
    ("mm_mf_rate", float, 0.0001, "MM mf rate"),
    ("feat_loss_type", str, "sce", "Feature loss type"),
    ("att_re_rate", float, 0.0, "Attribute restoration rate"),
    ("alpha_l", float, 2, "`pow` index for `sce` loss"),
    ("aug_sample_rate", float, 0.1, "Augmentation sample rate"),
    ("mf_emb_rate", float, 0.0, "MF embedding rate (unused)"),
]


def build_parser() -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser(description="")
    for name, kind, default, text in FLAGS:
        opt = "--" + name
        if kind == "flag":
            parser.add_argument(opt, action="store_true", help=text)
        elif kind == "str?":
            parser.add_argument(opt, nargs="?", default=default, help=text)
        else:
            parser.add_argument(opt, type=kind, default=default, help=text)
    return parser


def parse_args(argv=None):
    return build_parser().parse_args(argv)
