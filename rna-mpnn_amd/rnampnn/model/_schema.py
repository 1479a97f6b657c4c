"""State-dict schema of the reference's ``RNAMPNN`` (key -> shape), derived from its
constructor (``rnampnn/model/rnampnn.py:94-134``; ``feature.py:184-203``; ``mpnn.py:132-152``;
``functional.py:15-16,62-74,105-127,179-189``).  Key names and order equal torch's
``state_dict()`` of the reference model, so a Lightning ``.ckpt['state_dict']`` loads as is.
The C library registers the same list (``rnampnn_weight_info``); a CPU test compares the two
and both against the key list recorded in the golden fixtures.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Tuple

DEFAULT_HPARAMS = OrderedDict(
    num_res_neighbours=3, num_inside_dist_atoms=7, num_inside_angle_atoms=6, num_inside_dihedral_atoms=6,
    num_cross_dist_atoms=7, num_cross_angle_atoms=6, num_cross_dihedral_atoms=6,
    res_embedding_dim=128, num_embedding_attn_layers=0, num_embedding_heads=8, embedding_ffn_dim=512,
    num_embedding_ffn_layers=3, res_edge_embedding_dim=128, depth_res_edge_feature=2,
    num_res_mpnn_layers=10, depth_res_mpnn=2, num_mpnn_edge_layers=2, padding_len=4500,
    num_post_fusion_attn_layers=2, num_post_fusion_heads=8, post_fusion_ffn_dim=512,
    num_post_fusion_ffn_layers=3, num_raw_ffn_dim=512, num_raw_ffn_layers=3, raw_embedding_dim=128,
    readout_hidden_dim=512, num_readout_layers=2, dropout=0.4, lr=2e-3, weight_decay=0.0002,
    n_estimators=150, xgb_max_depth=8, xgb_learning_rate=0.1, xgb_subsample=0.8, xgb_colsample_bytree=0.8)

# fields of the C struct RnaMpnnConfig, in order (include/rnampnn_hip.h)
NATIVE_FIELDS = ("num_res_neighbours", "res_embedding_dim", "num_embedding_attn_layers", "num_embedding_heads",
                 "embedding_ffn_dim", "num_embedding_ffn_layers", "res_edge_embedding_dim", "depth_res_edge_feature",
                 "num_res_mpnn_layers", "depth_res_mpnn", "num_mpnn_edge_layers", "padding_len",
                 "num_post_fusion_attn_layers", "num_post_fusion_heads", "post_fusion_ffn_dim",
                 "num_post_fusion_ffn_layers", "num_raw_ffn_dim", "num_raw_ffn_layers", "raw_embedding_dim",
                 "readout_hidden_dim", "num_readout_layers")

D = 128


def _linear(out: Dict, prefix: str, n_in: int, n_out: int):
    out[prefix + ".weight"] = (n_out, n_in)
    out[prefix + ".bias"] = (n_out,)


def _gn(out: Dict, prefix: str):
    out[prefix + ".scale"] = (1, 1, D)
    out[prefix + ".shift"] = (1, 1, D)


def _bert(out: Dict, prefix: str, n_attn: int, ffn: int, n_ffn: int):
    for j in range(n_attn):
        p = f"{prefix}.bi_attention_layers.{j}"
        out[p + ".in_proj_weight"] = (3 * D, D)
        out[p + ".in_proj_bias"] = (3 * D,)
        _linear(out, p + ".out_proj", D, D)
    for j in range(n_attn):
        _gn(out, f"{prefix}.graph_norm_layers.{j}")
    n_in = D
    for i in range(n_ffn):
        _linear(out, f"{prefix}.ffn_layers.{3 * i}", n_in, ffn)
        n_in = ffn
    _linear(out, f"{prefix}.ffn_layers.{3 * n_ffn}", ffn, D)


def state_dict_shapes(hp) -> "OrderedDict[str, Tuple[int, ...]]":
    out: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    _linear(out, "res_feature.raw_project", 28, D)
    _bert(out, "res_feature.res_embedding", hp["num_embedding_attn_layers"], hp["embedding_ffn_dim"],
          hp["num_embedding_ffn_layers"])
    _gn(out, "res_feature.graph_norm")
    for i in range(hp["depth_res_edge_feature"]):
        _linear(out, f"res_feature.res_edge_embedding_layers.{3 * i}", 90 if i == 0 else D, D)
    for l in range(hp["num_res_mpnn_layers"]):
        p = f"res_mpnn_layers.{l}"
        _gn(out, p + ".graph_norm")
        for i in range(hp["depth_res_mpnn"]):
            _linear(out, f"{p}.message_layers.{3 * i}", 3 * D if i == 0 else D, D)
        for i in range(hp["num_mpnn_edge_layers"]):
            _linear(out, f"{p}.edge_layers.{3 * i}", 3 * D if i == 0 else D, D)
    _bert(out, "post_fusion", hp["num_post_fusion_attn_layers"], hp["post_fusion_ffn_dim"],
          hp["num_post_fusion_ffn_layers"])
    n_in = 28
    for i in range(hp["num_raw_ffn_layers"]):
        _linear(out, f"raw_embedding.raw_ffn.{3 * i}", n_in, hp["num_raw_ffn_dim"])
        n_in = hp["num_raw_ffn_dim"]
    _linear(out, f"raw_embedding.raw_ffn.{3 * hp['num_raw_ffn_layers']}", hp["num_raw_ffn_dim"], D)
    _gn(out, "raw_embedding.graph_norm")
    n_in = 2 * D
    for i in range(hp["num_readout_layers"] - 1):
        _linear(out, f"readout.readout_layers.{3 * i}", n_in, hp["readout_hidden_dim"])
        n_in = hp["readout_hidden_dim"]
    _linear(out, f"readout.readout_layers.{3 * (hp['num_readout_layers'] - 1)}", n_in, 4)
    return out
