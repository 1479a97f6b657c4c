#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE's own L1 modules.

Runs only in the build container (needs /root/reference; never on the GPU box, never from
tests).  It imports ``rnampnn.model.{feature,mpnn,functional}`` from the reference, composes
them exactly as ``rnampnn/model/rnampnn.py:94-134,173-185`` does (the LightningModule adds
no arithmetic and cannot be imported here: pytorch_lightning / xgboost are absent), fills
the weights with the closed-form deterministic init of ``rnampnn/utils/synth.py`` and dumps
inputs + outputs as small ``.npz`` fixtures.  ``seeding()`` is deliberately not called
(it would set float32 matmul precision to 'medium').

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden.py
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, REF)                     # the reference's ``rnampnn`` package
from rnampnn.model.feature import ResFeature          # noqa: E402
from rnampnn.model.mpnn import ResMPNN                # noqa: E402
from rnampnn.model.functional import RNABert, RawFFN, Readout  # noqa: E402

# our own deterministic generators (loaded by path: the package name collides with the reference's)
import importlib.util  # noqa: E402
_spec = importlib.util.spec_from_file_location(
    "synth", os.path.join(REPO, "rna-mpnn_amd", "rnampnn", "utils", "synth.py"))
synth = importlib.util.module_from_spec(_spec)
sys.modules["synth"] = synth
_spec.loader.exec_module(synth)

DEFAULTS = dict(num_res_neighbours=3, res_embedding_dim=128, num_embedding_attn_layers=0,
                num_embedding_heads=8, embedding_ffn_dim=512, num_embedding_ffn_layers=3,
                res_edge_embedding_dim=128, depth_res_edge_feature=2, num_res_mpnn_layers=10,
                depth_res_mpnn=2, num_mpnn_edge_layers=2, padding_len=4500,
                num_post_fusion_attn_layers=2, num_post_fusion_heads=8, post_fusion_ffn_dim=512,
                num_post_fusion_ffn_layers=3, num_raw_ffn_dim=512, num_raw_ffn_layers=3,
                raw_embedding_dim=128, readout_hidden_dim=512, num_readout_layers=2)


class RefComposite(torch.nn.Module):
    """The five reference modules wired as RNAMPNN.__init__ wires them (rnampnn.py:94-134)."""

    def __init__(self, **hp):
        super().__init__()
        c = dict(DEFAULTS)
        c.update(hp)
        self.cfg = c
        drop = 0.4
        self.res_feature = ResFeature(num_neighbours=c["num_res_neighbours"],
                                      res_embedding_dim=c["res_embedding_dim"],
                                      padding_len=c["padding_len"],
                                      num_attn_layers=c["num_embedding_attn_layers"],
                                      num_heads=c["num_embedding_heads"],
                                      ffn_dim=c["embedding_ffn_dim"],
                                      num_ffn_layers=c["num_embedding_ffn_layers"],
                                      res_edge_embedding_dim=c["res_edge_embedding_dim"],
                                      num_edge_layers=c["depth_res_edge_feature"], dropout=drop)
        self.res_mpnn_layers = torch.nn.ModuleList([
            ResMPNN(res_embedding_dim=c["res_embedding_dim"],
                    res_edge_embedding_dim=c["res_edge_embedding_dim"],
                    depth_res_mpnn=c["depth_res_mpnn"], num_edge_layers=c["num_mpnn_edge_layers"],
                    dropout=drop) for _ in range(c["num_res_mpnn_layers"])])
        self.post_fusion = RNABert(padding_len=c["padding_len"], res_embedding_dim=c["res_embedding_dim"],
                                   num_attn_layers=c["num_post_fusion_attn_layers"],
                                   num_heads=c["num_post_fusion_heads"], ffn_dim=c["post_fusion_ffn_dim"],
                                   num_ffn_layers=c["num_post_fusion_ffn_layers"], dropout=drop)
        self.raw_embedding = RawFFN(raw_dim=self.res_feature.raw_dim,
                                    num_raw_ffn_layers=c["num_raw_ffn_layers"],
                                    num_raw_ffn_dim=c["num_raw_ffn_dim"],
                                    raw_embedding_dim=c["raw_embedding_dim"], dropout=drop)
        self.readout = Readout(embedding_dim=c["raw_embedding_dim"] + c["res_embedding_dim"],
                               readout_hidden_dim=c["readout_hidden_dim"],
                               num_layers=c["num_readout_layers"], dropout=drop)

    @torch.no_grad()
    def run(self, coords, mask, tap_layers=(1,)):
        out = {}
        raw, h, e, idx = self.res_feature(coords, mask)                 # rnampnn.py:173
        out.update(raw=raw.clone(), h0=h.clone(), e0=e.clone(), edge_index=idx.clone())
        for l, layer in enumerate(self.res_mpnn_layers):                # rnampnn.py:177-178
            h, e = layer(h, e, idx, mask)
            if (l + 1) in tap_layers:
                out[f"h{l + 1}"] = h.clone()
                out[f"e{l + 1}"] = e.clone()
        out["hL"] = h.clone()
        hp = self.post_fusion(h, mask)                                  # rnampnn.py:179
        re = self.raw_embedding(raw, mask)                              # rnampnn.py:180
        emb = torch.cat((hp, re), dim=-1)
        logits = self.readout(emb.clone(), mask)                        # rnampnn.py:181
        out.update(h_post=hp, raw_emb=re, embedding=emb, logits=logits)
        # loss of rnampnn.py:151-154,200-204 on synthetic labels (set by caller) is added there
        return out


def build(hp, dtype):
    model = RefComposite(**hp).eval()
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = synth.closed_form_state_dict(shapes)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return model.to(dtype), shapes


def edge_raw(model, coords, mask, idx):
    rf = model.res_feature
    return torch.cat([rf._cross_dists(coords, mask, idx), rf._cross_angles(coords, mask, idx),
                      rf._cross_dihedrals(coords, mask, idx)], dim=-1)


def loss_fn(logits, mask, labels):
    probs = torch.softmax(logits, dim=-1)[mask.bool()]
    return torch.nn.functional.cross_entropy(probs, labels[mask.bool()], reduction="mean")


LITE_KEYS = ("edge_index", "hL", "logits", "loss")


def run_case(name, hp, coords, mask, labels, e_nodes=4, outdir=None, full=True):
    """Run fp32 (the golden) and fp64 (noise-floor reference) and store a compact fixture.
    ``full`` keeps every stage tap; lite cases keep edge_index, hL, logits and the loss."""
    res = {}
    for dtype, tag in ((torch.float32, ""), (torch.float64, "_f64")):
        model, shapes = build(hp, dtype)
        c = torch.from_numpy(coords).to(dtype)
        m = torch.from_numpy(mask).to(dtype)
        out = model.run(c, m)
        idx = out["edge_index"]
        out["edge_raw"] = edge_raw(model, c, m, idx)
        out["loss"] = loss_fn(out["logits"], m, torch.from_numpy(labels))
        if tag == "":
            keep = {}
            for k, v in out.items():
                if k == "embedding" or (not full and k not in LITE_KEYS):
                    continue                            # embedding = cat(h_post, raw_emb)
                v = v.numpy()
                if k in ("e0", "e1", "edge_raw"):
                    v = v[:, :e_nodes]                  # first few nodes only: keeps fixtures small
                if k == "edge_index":
                    v = v.astype(np.int32)
                keep[k] = v
            res.update(keep)
        else:
            res["logits" + tag] = out["logits"].numpy()
    res["coords"] = coords
    res["mask"] = mask
    res["labels"] = labels
    res["hparams"] = np.frombuffer(json.dumps({**DEFAULTS, **hp}).encode(), dtype=np.uint8)
    res["state_keys"] = np.frombuffer(json.dumps({k: list(s) for k, s in shapes.items()}).encode(), dtype=np.uint8)
    path = os.path.join(outdir, name + ".npz")
    np.savez_compressed(path, **res)
    d = float(np.abs(res["logits"] - res["logits_f64"]).max())
    print(f"{name}: B,T={mask.shape} k={hp.get('num_res_neighbours', 3)} "
          f"|logits32-logits64|max={d:.2e} loss={float(res['loss']):.6f} -> {os.path.getsize(path) / 1024:.0f} KB")


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    outdir = os.path.join(REPO, "tests", "golden")
    os.makedirs(outdir, exist_ok=True)

    # C1: the real 66-nt RNA 1B23_1_R (NaN-free), k=16, T=n, P in {66, 4500}
    real = np.load(os.path.join(REF, "data", "coords", "1B23_1_R.npy")).astype(np.float32)
    n = real.shape[0]
    vocab = {"A": 0, "U": 1, "C": 2, "G": 3}
    seq = "".join(l.strip() for l in open(os.path.join(REF, "data", "seqs", "1B23_1_R.fasta")) if not l.startswith(">"))
    lab = np.array([[vocab[ch] for ch in seq]], dtype=np.int64)
    c1, m1 = real[None], np.ones((1, n), dtype=np.float32)
    run_case("c1_1b23_k16_P66", dict(num_res_neighbours=16, padding_len=n), c1, m1, lab, outdir=outdir)
    run_case("c1_1b23_k16_P4500", dict(num_res_neighbours=16, padding_len=4500), c1, m1, lab, outdir=outdir, full=False)
    # default hyper-parameters (k=3)
    run_case("c1_1b23_k3_default", dict(padding_len=128), c1, m1, lab, outdir=outdir, full=False)
    # the same RNA padded to T=80 (padding-dependent GraphNorm), k=30
    c80 = np.zeros((1, 80, 7, 3), np.float32); c80[0, :n] = real
    m80 = np.zeros((1, 80), np.float32); m80[0, :n] = 1
    l80 = np.zeros((1, 80), np.int64); l80[0, :n] = lab[0]
    run_case("c1_1b23_k30_T80", dict(num_res_neighbours=30, padding_len=80), c80, m80, l80, outdir=outdir, full=False)

    # phantom-edge rule: n=5, T=8, k=6 (tie-break dependent)
    co, ma, la = synth.synth_batch([5], first_index=900, max_len=8)
    run_case("phantom_n5_T8_k6", dict(num_res_neighbours=6, padding_len=8, num_res_mpnn_layers=2), co, ma, la, outdir=outdir)

    # ragged synthetic batch with short RNAs (n < k), k=30, P=T and a smaller stack with embedding attention
    lens = [40, 7, 33, 1, 25]
    co, ma, la = synth.synth_batch(lens, first_index=100)
    run_case("ragged_k30", dict(num_res_neighbours=30, padding_len=40, num_res_mpnn_layers=3), co, ma, la, outdir=outdir)
    # the commented train.py:9-43 configuration (k=4, 6 layers, 1 attention layer in the embedding, ...)
    alt = dict(num_res_neighbours=4, num_embedding_attn_layers=1, embedding_ffn_dim=256, num_embedding_ffn_layers=1,
               num_res_mpnn_layers=6, num_mpnn_edge_layers=1, num_post_fusion_attn_layers=1,
               post_fusion_ffn_dim=256, num_post_fusion_ffn_layers=1, num_raw_ffn_layers=1, num_raw_ffn_dim=256,
               readout_hidden_dim=256, num_readout_layers=1, padding_len=64)
    co, ma, la = synth.synth_batch([30, 21], first_index=200)
    run_case("alt_cfg_k4", alt, co, ma, la, outdir=outdir)
    # C2-shaped miniature: 4 RNAs of 100-140 nt, k=30, P=T
    lens = synth.synth_lengths(3, 100, 140, seed=0)
    co, ma, la = synth.synth_batch(lens, first_index=0)
    run_case("c2_mini_k30", dict(num_res_neighbours=30, padding_len=int(max(lens))), co, ma, la, outdir=outdir, full=False)


if __name__ == "__main__":
    main()
