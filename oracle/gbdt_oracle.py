"""CPU ORACLE of the gradient-boosted-tree read-out.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED.  The reference calls `xgb.XGBClassifier(objective='multi:softmax', ...).predict` (rnampnn/model/rnampnn.py:136-145,
297-298); xgboost (requirements.txt: xgboost~=2.1.1) is absent from this image and the reference holds no fitted model, so nothing
XGBoost-produced can pin this file.  It restates XGBoost's published `gbtree` prediction rule on the arrays of its JSON model format:
walk from node 0, left iff x[split_index] < split_condition (NaN follows default_left), a leaf (left child -1) adds
split_conditions[leaf] to the margin of tree_info[t]; margins start at base_score; multi:softmax returns the first argmax.
Pure-Python loops: small cases only.
"""
import numpy as np


def predict(arrays: dict, X: np.ndarray):
    """-> (class ids (N,), margins (N, num_class) float32), trees added in model order (float32 accumulation)."""
    X = np.asarray(X, np.float32)
    N, C = X.shape[0], int(arrays["num_class"])
    off, cls = arrays["tree_offsets"], arrays["tree_class"]
    left, right = arrays["left_children"], arrays["right_children"]
    feat, thr, dl = arrays["split_indices"], arrays["split_conditions"], arrays["default_left"]
    margins = np.full((N, C), np.float32(arrays["base_score"]), np.float32)
    for i in range(N):
        for t in range(len(cls)):
            b, node = int(off[t]), 0
            while left[b + node] >= 0:
                v = X[i, feat[b + node]]
                go_left = bool(dl[b + node]) if np.isnan(v) else bool(v < thr[b + node])
                node = int(left[b + node] if go_left else right[b + node])
            margins[i, cls[t]] = np.float32(margins[i, cls[t]] + thr[b + node])
    return margins.argmax(1), margins
