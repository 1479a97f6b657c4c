"""Counter-based deterministic generators shared by the oracle, the golden-vector script,
the tests and ``bench.py``.

Everything here is a pure function of integer counters (no ``torch.Generator`` /
``numpy.random`` state), so the same weights and the same synthetic RNA backbones come
out on every box and in every process of a data-parallel job.

* ``uniform01(stream, idx)``   - 24-bit uniform in [0, 1) from a splitmix64-style mix.
* ``closed_form_state_dict``   - weights for every key of an RNAMPNN ``state_dict``:
  ``Linear`` tensors ~ U(-1/sqrt(fan_in), 1/sqrt(fan_in)) (the torch default range),
  ``GraphNormalization`` scale = 1 + 0.1 u, shift = 0.1 u, so norms are exercised.
* ``synth_rna``                - random-walk backbone: unit steps scaled to 6.0 A, atoms at
  centre + 1.5 A * N(0, 1) (SURVEY.md section 8d "Synthetic inputs").
"""
from __future__ import annotations

import math
from typing import Dict, Iterable, Sequence, Tuple

import numpy as np

_M64 = (1 << 64) - 1
_GOLDEN = 0x9E3779B97F4A7C15


def _fnv1a64(text: str) -> int:
    h = 0xCBF29CE484222325
    for ch in text.encode("utf-8"):
        h ^= ch
        h = (h * 0x100000001B3) & _M64
    return h


def _mix64(x: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on a uint64 array (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        x = x.astype(np.uint64, copy=True)
        x ^= x >> np.uint64(30)
        x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27)
        x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    return x


def uniform01(stream: int, idx: np.ndarray) -> np.ndarray:
    """24-bit uniforms in [0,1) for counters ``idx`` of stream ``stream`` (float64 array)."""
    idx = np.asarray(idx, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = (idx + np.uint64(1)) * np.uint64(_GOLDEN) + np.uint64(stream & _M64)
    bits = _mix64(_mix64(x) ^ np.uint64((stream * 0xD6E8FEB86659FD93) & _M64))
    return (bits >> np.uint64(40)).astype(np.float64) * (1.0 / (1 << 24))


def normal01(stream: int, idx: np.ndarray) -> np.ndarray:
    """Standard normals (Box-Muller on two uniform streams), float64."""
    idx = np.asarray(idx, dtype=np.uint64)
    u1 = uniform01(stream, idx * np.uint64(2))
    u2 = uniform01(stream, idx * np.uint64(2) + np.uint64(1))
    r = np.sqrt(-2.0 * np.log(1.0 - u1))          # 1-u1 in (0,1]
    return r * np.cos(2.0 * math.pi * u2)


def _fan_in(key: str, shape: Sequence[int]) -> int:
    if len(shape) >= 2:
        return int(shape[-1])
    return 0


def closed_form_tensor(key: str, shape: Sequence[int], fan_in: int | None = None) -> np.ndarray:
    """Deterministic fp32 fill for one state-dict entry (function of key + flat index)."""
    n = int(np.prod(shape)) if len(shape) else 1
    u = uniform01(_fnv1a64(key), np.arange(n, dtype=np.uint64)) * 2.0 - 1.0
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "scale":
        vals = 1.0 + 0.1 * u
    elif leaf == "shift":
        vals = 0.1 * u
    else:
        f = fan_in if fan_in else _fan_in(key, shape)
        bound = 1.0 / math.sqrt(f) if f > 0 else 0.05
        vals = bound * u
    return vals.astype(np.float32).reshape(tuple(shape))


def closed_form_state_dict(shapes: Dict[str, Sequence[int]]) -> Dict[str, np.ndarray]:
    """Fill every key of ``shapes`` (key -> shape).  Biases take the fan-in of their
    sibling ``weight`` (``in_proj_bias`` of ``in_proj_weight``), as torch's default init does."""
    out: Dict[str, np.ndarray] = {}
    for key, shape in shapes.items():
        fan = None
        if key.endswith("bias"):
            sib = key[: -len("bias")] + "weight"
            if sib in shapes:
                fan = int(shapes[sib][-1])
        out[key] = closed_form_tensor(key, shape, fan)
    return out


def synth_rna(length: int, rna_index: int, seed: int = 0) -> np.ndarray:
    """Synthetic backbone of ``length`` residues -> coords (length, 7, 3) float32."""
    stream = _fnv1a64(f"synth_rna/{seed}/{rna_index}")
    L = int(length)
    steps = normal01(stream, np.arange(L * 3, dtype=np.uint64)).reshape(L, 3)
    norm = np.sqrt((steps ** 2).sum(-1, keepdims=True))
    steps = steps / np.maximum(norm, 1e-12) * 6.0
    centres = np.cumsum(steps, axis=0)
    jitter = normal01(stream ^ 0x5A5A5A5A5A5A5A5A, np.arange(L * 21, dtype=np.uint64)).reshape(L, 7, 3)
    return (centres[:, None, :] + 1.5 * jitter).astype(np.float32)


def synth_lengths(batch: int, lo: int, hi: int, seed: int = 0, first_index: int = 0) -> np.ndarray:
    """Lengths ~ U[lo, hi] (inclusive) for RNAs first_index .. first_index+batch-1."""
    u = uniform01(_fnv1a64(f"synth_len/{seed}"), np.arange(first_index, first_index + batch, dtype=np.uint64))
    return (lo + np.floor(u * (hi - lo + 1))).astype(np.int64)


def synth_labels(length: int, rna_index: int, seed: int = 0) -> np.ndarray:
    """Uniform labels over {A,U,C,G} -> int64 (length,)."""
    u = uniform01(_fnv1a64(f"synth_seq/{seed}/{rna_index}"), np.arange(length, dtype=np.uint64))
    return np.floor(u * 4).astype(np.int64)


def synth_batch(lengths: Iterable[int], first_index: int = 0, seed: int = 0,
                max_len: int | None = None) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Padded batch in the layout of the reference collate (``utils/data.py:110-142``):
    coords (B,T,7,3) zero-padded, mask (B,T) prefix of ones, labels (B,T) int64 (0 on padding)."""
    lengths = [int(x) for x in lengths]
    B = len(lengths)
    T = int(max_len) if max_len is not None else max(lengths)
    coords = np.zeros((B, T, 7, 3), dtype=np.float32)
    mask = np.zeros((B, T), dtype=np.float32)
    labels = np.zeros((B, T), dtype=np.int64)
    for b, n in enumerate(lengths):
        coords[b, :n] = synth_rna(n, first_index + b, seed)
        mask[b, :n] = 1.0
        labels[b, :n] = synth_labels(n, first_index + b, seed)
    return coords, mask, labels


def synth_xgb_json(n_rounds, num_class, num_feature, max_depth, seed):
    """A seeded random forest in XGBoost's JSON model schema (``Booster.save_model``): n_rounds x num_class trees, depth-first node
    numbering (children after parents) - synthetic stand-in for the fitted read-out the reference does not ship (row F4)."""
    rng = np.random.RandomState(seed)
    trees, info = [], []
    for r in range(n_rounds):
        for c in range(num_class):
            L, R, F, T, D = [], [], [], [], []

            def grow(depth):
                i = len(L)
                L.append(-1); R.append(-1); F.append(0); T.append(0.0); D.append(0)
                if depth < max_depth and (depth < 2 or rng.rand() < 0.75):
                    F[i] = int(rng.randint(num_feature)); T[i] = float(np.float32(rng.randn())); D[i] = int(rng.rand() < 0.5)
                    L[i] = grow(depth + 1)
                    R[i] = grow(depth + 1)
                else:
                    T[i] = float(np.float32(0.3 * rng.randn()))          # leaf value
                return i
            grow(0)
            trees.append(dict(left_children=L, right_children=R, split_indices=F, split_conditions=T, default_left=D,
                              categories_nodes=[], id=len(trees)))
            info.append(c)
    return {"learner": {"learner_model_param": {"num_class": str(num_class), "num_feature": str(num_feature), "base_score": "5E-1"},
                        "gradient_booster": {"name": "gbtree", "model": {"tree_info": info, "trees": trees}},
                        "objective": {"name": "multi:softmax"}}, "version": [2, 1, 1]}
