"""Device-side gradient-boosted-tree read-out: stands in for the reference's fitted
``xgb.XGBClassifier`` (``rnampnn/model/rnampnn.py:136-145``; ``self.xgb_readout.predict(embedding)``, ``:297-298``).

PARITY UNPINNED: xgboost (``requirements.txt``: ``xgboost~=2.1.1``) is not installed in the build image and the reference ships no
fitted model (its ``XGB-V*.pkl`` is a pickle, which this project never loads), so no XGBoost-produced vector backs this path.  What is
implemented is XGBoost's published prediction rule for ``gbtree`` / ``multi:softmax`` models, read from XGBoost's JSON model format
(``Booster.save_model("model.json")`` - a maintainer converts the pickle once with xgboost installed):

    learner.learner_model_param.{num_class, num_feature, base_score}
    learner.gradient_booster.model.tree_info[t]                      class of tree t
    learner.gradient_booster.model.trees[t].{left_children, right_children, split_indices, split_conditions, default_left}

(left child -1 = leaf, whose value sits in ``split_conditions``; go left iff ``x[f] < split_condition``, NaN follows ``default_left``).
"""
from __future__ import annotations

import ctypes as C
import json
from typing import Sequence, Union

import numpy as np
import torch

from .. import _native
from ._base import _stream


def parse_xgboost_json(model: Union[str, dict]) -> dict:
    """XGBoost JSON model (path or parsed dict) -> flat arrays (see the module docstring for the fields read)."""
    if isinstance(model, str):
        with open(model) as f:
            model = json.load(f)
    learner = model["learner"]
    lp = learner["learner_model_param"]
    gb = learner["gradient_booster"]
    if gb.get("name", "gbtree") != "gbtree":
        raise NotImplementedError(f"booster '{gb.get('name')}' is not supported (gbtree only)")
    m = gb["model"]
    trees = m["trees"]
    num_class = max(int(lp.get("num_class", "0")), 1)
    for t in trees:
        if t.get("categories_nodes"):
            raise NotImplementedError("categorical splits are not supported")
    offs = np.zeros(len(trees) + 1, np.int32)
    offs[1:] = np.cumsum([len(t["left_children"]) for t in trees])
    cat = lambda key, dt: np.concatenate([np.asarray(t[key], dtype=dt) for t in trees])
    return dict(num_class=num_class, num_feature=int(lp["num_feature"]), base_score=float(lp.get("base_score", "0.5")),
                tree_offsets=offs, tree_class=np.asarray(m["tree_info"], np.int32),
                left_children=cat("left_children", np.int32), right_children=cat("right_children", np.int32),
                split_indices=cat("split_indices", np.int32), split_conditions=cat("split_conditions", np.float32),
                default_left=cat("default_left", np.uint8))


class GBDTReadout:
    """``predict(embedding)`` of a fitted multi:softmax XGBoost model on the MI355X (``rnampnn_gbdt_*``, csrc/gbdt.hip)."""

    def __init__(self, arrays: dict):
        self.arrays = {k: (np.ascontiguousarray(v) if isinstance(v, np.ndarray) else v) for k, v in arrays.items()}
        a = self.arrays
        self.num_class, self.num_feature = int(a["num_class"]), int(a["num_feature"])
        self._h = C.c_void_p()
        p = lambda k: a[k].ctypes.data_as(C.c_void_p)
        rc = _native.lib().rnampnn_gbdt_create(len(a["tree_class"]), self.num_class, self.num_feature, float(a["base_score"]),
                                               p("tree_offsets"), p("tree_class"), p("left_children"), p("right_children"),
                                               p("split_indices"), p("split_conditions"), p("default_left"), C.byref(self._h))
        if rc != 0:
            msg = _native.lib().rnampnn_gbdt_last_error().decode()
            raise (ValueError if rc == _native.ERR_BAD_ARG else RuntimeError)(msg)

    @classmethod
    def from_xgboost_json(cls, model: Union[str, dict]) -> "GBDTReadout":
        return cls(parse_xgboost_json(model))

    def _run(self, x: torch.Tensor, want_margin: bool):
        if x.device.type != "cuda":
            raise RuntimeError("the GBDT read-out runs on an MI355X: pass a CUDA tensor (there is no CPU fallback)")
        x = x.detach().to(torch.float32)
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        if x2.shape[1] < self.num_feature:
            raise ValueError(f"model expects {self.num_feature} features, got {x2.shape[1]}")
        n = int(x2.shape[0])
        pred = torch.empty(n, dtype=torch.int32, device=x.device)
        margin = torch.empty(n, self.num_class, dtype=torch.float32, device=x.device) if want_margin else None
        with torch.cuda.device(x.device):
            rc = _native.lib().rnampnn_gbdt_predict(self._h, C.c_void_p(x2.data_ptr()), n, int(x2.shape[1]),
                                                    C.c_void_p(margin.data_ptr()) if want_margin else None,
                                                    C.c_void_p(pred.data_ptr()), _stream(x.device))
        if rc != 0:
            raise RuntimeError(_native.lib().rnampnn_gbdt_last_error().decode())
        return pred.reshape(x.shape[:-1]).to(torch.int64), (margin.reshape(*x.shape[:-1], self.num_class) if want_margin else None)

    def predict(self, embedding: torch.Tensor) -> torch.Tensor:
        """``XGBClassifier.predict``: class ids, shape ``embedding.shape[:-1]``."""
        return self._run(embedding, False)[0]

    def margins(self, embedding: torch.Tensor) -> torch.Tensor:
        """``predict(output_margin=True)``: per-class sums of the leaf values (+ base_score)."""
        return self._run(embedding, True)[1]

    def close(self):
        if self._h:
            _native.lib().rnampnn_gbdt_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
