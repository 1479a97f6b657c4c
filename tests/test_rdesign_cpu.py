"""CPU checks of the `rdesign` row: the C ABI of include/rdesign_hip.h is exported, the parameter table
equals the oracle's restatement of the reference state_dict, and the oracle (PARITY UNPINNED - oracle/rdesign_oracle.py) is
self-consistent: shapes, rigid-motion invariance, batch/padding independence, neighbour ordering."""
import os
import re

import numpy as np
import pytest
import torch

from conftest import REPO
from oracle import rdesign_oracle as O


def _batch(lengths, seed=0):
    from rnampnn.utils import synth
    T = max(lengths)
    X = np.zeros((len(lengths), T, 6, 3), np.float32)
    mask = np.zeros((len(lengths), T), np.float32)
    for i, n in enumerate(lengths):
        X[i, :n] = synth.synth_rna(n, i, seed=seed)[:, :6]
        mask[i, :n] = 1
    return torch.from_numpy(X), torch.from_numpy(mask)


def _weights(cfg, seed=0):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, shp in O.state_dict_shapes(cfg).items():
        if k.endswith("gain") or (("norm1" in k or "norm2" in k) and k.endswith("weight")):
            sd[k] = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif len(shp) == 2:
            sd[k] = torch.randn(shp, generator=g) / shp[1] ** 0.5
        else:
            sd[k] = 0.1 * torch.randn(shp, generator=g)
    return sd


def test_library_exports_every_declared_rdesign_symbol():
    import __graft_entry__ as g
    g.build()
    from rdesign import _native
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(REPO, "include", "rdesign_hip.h")).read(), flags=re.S)
    declared = sorted(set(re.findall(r"\b(rdesign_[a-z0-9_]+)\s*\(", text)))
    lib = _native.lib()
    assert len(declared) >= 10 and sorted(_native.SYMBOLS) == declared
    for name in declared:
        assert hasattr(lib, name)


@pytest.mark.parametrize("kw", [dict(), dict(num_readout_layers=2, num_mpnn_layers=2, dim_dense_layers=64, k_neighbors=5,
                                              num_message_layers=2, num_dense_layers=1)])
def test_module_tree_matches_reference_state_dict(kw):
    from rdesign.model.rdesign import RNAModel
    m = RNAModel(**kw)
    shapes = O.state_dict_shapes(O.RDesignConfig(**kw))
    sd = m.state_dict()
    assert list(sd) == list(shapes) and all(tuple(sd[k].shape) == shapes[k] for k in shapes)
    if not kw:
        assert sum(v.numel() for v in sd.values()) == 2_551_812


def test_errors_and_no_cpu_fallback():
    from rdesign.model.rdesign import RNAModel
    with pytest.raises(NotImplementedError):
        RNAModel(hidden_dim=64)
    with pytest.raises(NotImplementedError):
        RNAModel(k_neighbors=100)
    with pytest.raises(NotImplementedError):
        RNAModel(node_feat_types=["angle"])
    m = RNAModel(num_mpnn_layers=1).eval()
    X, mask = _batch([5])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(X, torch.zeros(1, 5, dtype=torch.long), mask)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.readout(torch.zeros(3, 128))


def test_oracle_shapes_and_feature_ranges():
    cfg = O.RDesignConfig(k_neighbors=6, num_mpnn_layers=2)
    X, mask = _batch([12, 4, 9])
    node, edge, E_idx, attend = O.raw_features(X, mask, cfg)
    assert node.shape == (3, 12, 101) and edge.shape == (3, 12, 6, 115) and E_idx.shape == (3, 12, 6)
    # an RNA shorter than k has only n attended slots per residue (feature.py:186-194)
    assert attend[1, :4].sum(-1).tolist() == [4] * 4 and attend[1, 4:].sum() == 0
    assert attend[0].sum(-1).tolist() == [6] * 12
    # first neighbour = the residue itself (distance sqrt(1e-6)), distances ascending
    assert (E_idx[0, :, 0] == torch.arange(12)).all()
    v = node[mask == 1]
    assert torch.isfinite(v).all() and (v[:, 12:92] >= 0).all() and (v[:, 12:92] <= 1).all()
    assert torch.allclose(v[:, :6] ** 2 + v[:, 6:12] ** 2, torch.ones(v.shape[0], 6), atol=1e-5)      # cos^2 + sin^2
    q = edge[attend][:, :4]
    assert torch.allclose((q ** 2).sum(-1), torch.ones(q.shape[0]), atol=1e-5)                         # unit quaternions
    h_V, logits = O.forward(X, mask, _weights(cfg), cfg)
    assert h_V.shape == (25, 128) and logits.shape == (25, 4) and torch.isfinite(logits).all()


def test_oracle_invariances():
    """Distances, dihedrals and local-frame directions are rigid-motion invariant.  The quaternion block is only translation
    invariant: the reference forms R = Q_i^T Q_j with the frame vectors as ROWS of Q (feature.py:104,127), which conjugates
    with the rotation - restated as written, not "fixed"."""
    cfg = O.RDesignConfig(k_neighbors=5, num_mpnn_layers=2)
    X, mask = _batch([11])
    g = torch.Generator().manual_seed(3)
    Q, _ = torch.linalg.qr(torch.randn(3, 3, generator=g, dtype=torch.float64))
    if torch.det(Q) < 0:
        Q[:, 0] = -Q[:, 0]
    Xr = (X.double() @ Q.T + torch.tensor([3.0, -7.0, 11.0], dtype=torch.float64)).float()
    Xt = (X.double() + torch.tensor([3.0, -7.0, 11.0], dtype=torch.float64)).float()
    n0, e0, i0, _ = O.raw_features(X, mask, cfg)
    n1, e1, i1, _ = O.raw_features(Xr, mask, cfg)
    assert (i0 == i1).all() and (n0 - n1).abs().max() < 1e-4 and (e0[..., 4:] - e1[..., 4:]).abs().max() < 1e-4
    sd = _weights(cfg)
    a, _ = O.forward(X, mask, sd, cfg)
    b, _ = O.forward(Xt, mask, sd, cfg)
    assert (a - b).abs().max() < 2e-3


def test_oracle_interior_rows_do_not_depend_on_batch_padding():
    """The flattened-chain dihedrals / frames of the LAST residue of an RNA see the zero padding (feature.py:85-101,138-153) -
    that is reference behaviour, restated - but every other residue's features are those of the RNA alone."""
    cfg = O.RDesignConfig(k_neighbors=4, num_mpnn_layers=1)
    X, mask = _batch([7, 12])
    node, edge, _, attend = O.raw_features(X, mask, cfg)
    X1, m1 = X[:1, :7], mask[:1, :7]
    node1, edge1, _, attend1 = O.raw_features(X1, m1, cfg)
    assert torch.allclose(node[0, :5], node1[0, :5], atol=1e-6)
    assert (attend[0, :7] == attend1[0]).all()
