"""Host-side input contract of the hot path: the reference's collate layout and ``separate``
(``rnampnn/utils/data.py:110-142, 594-604``).  Dataset / PDB parsing / plotting of that file are
host I/O outside the hot path and are not mirrored."""
from __future__ import annotations

from typing import Dict, List, Tuple, Union

import torch

from ..config.glob import NUM_MAIN_SEQ_ATOMS, NUM_RES_TYPES


def featurize(batch: List[Dict[str, Union[str, torch.Tensor]]]) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, List[str]]:
    """Pad a list of {'sequence' (L,4) one-hot, 'coordinates' (L,7,3), 'id'} items to the batch
    max length: sequences (B,T,4), coords (B,T,7,3) zero-padded, mask (B,T) prefix of ones, ids."""
    batch_size = len(batch)
    max_len = max(item['sequence'].shape[0] for item in batch)
    sequences = torch.zeros((batch_size, max_len, NUM_RES_TYPES), dtype=torch.float32)
    coords = torch.zeros((batch_size, max_len, NUM_MAIN_SEQ_ATOMS, 3), dtype=torch.float32)
    mask = torch.zeros((batch_size, max_len), dtype=torch.float32)
    ids = []
    for i, item in enumerate(batch):
        n = item['sequence'].shape[0]
        sequences[i, :n] = item['sequence']
        coords[i, :n] = item['coordinates']
        mask[i, :n] = 1
        ids.append(item['id'])
    return sequences, coords, mask, ids


def separate(concat: torch.Tensor, lengths: torch.Tensor) -> torch.Tensor:
    """Split a concatenation of per-RNA vectors back into a zero-padded (B, max_len) tensor."""
    lens = [int(x) for x in lengths.tolist()]
    out = torch.zeros((len(lens), max(lens) if lens else 0), dtype=concat.dtype)
    start = 0
    for i, n in enumerate(lens):
        out[i, :n] = concat[start:start + n]
        start += n
    return out


def check_prefix_mask(mask: torch.Tensor) -> None:
    """The kernels assume the collate's prefix masks (ones then zeros per row)."""
    m = mask.detach().cpu() != 0
    if bool((m[:, 1:] & ~m[:, :-1]).any()):
        raise ValueError("mask rows must be prefixes of ones (as produced by featurize)")
